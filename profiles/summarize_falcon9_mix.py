#!/usr/bin/env python3
"""Condense profiles/collect_falcon9_mix.sh's counter passes: per WAVE and TICK averages of the generated Falcon 9 kernel
(512 waves x 1000 ticks per launch).  SQ_*_CYCLES / ACTIVE / WAIT counters tick once per 4 clocks (one issue slot of a wave64)."""
import collections
import csv
import glob
import sys

out = sys.argv[1]
agg = collections.defaultdict(list)
for f in glob.glob(f"{out}/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f, newline="")):
        if "sixdof_step_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
waves = (sum(agg["SQ_WAVES"]) / len(agg["SQ_WAVES"])) if agg.get("SQ_WAVES") else 512.0
per = {k: sum(v) / len(v) / waves / 1000.0 for k, v in agg.items()}
g = lambda k: per.get(k, float("nan"))
print("| per wave and tick | |")
print("|---|---|")
print(f"| wave cycles (4-clock issue slots) | {g('SQ_WAVE_CYCLES'):.0f} |")
print(f"| issuing (any) / VALU / scalar / branch | {g('SQ_ACTIVE_INST_ANY'):.0f} / {g('SQ_ACTIVE_INST_VALU'):.0f} / {g('SQ_ACTIVE_INST_SCA'):.0f} / {g('SQ_ACTIVE_INST_MISC'):.0f} |")
print(f"| waiting (s_waitcnt and the like) / issue-stalled | {g('SQ_WAIT_ANY'):.0f} / {g('SQ_WAIT_INST_ANY'):.0f} |")
print(f"| VALU instructions | {g('SQ_INSTS_VALU'):.0f} |")
print(f"| of which f32 add / mul / fma / transcendental / convert / int32 | {g('SQ_INSTS_VALU_ADD_F32'):.0f} / {g('SQ_INSTS_VALU_MUL_F32'):.0f} / "
      f"{g('SQ_INSTS_VALU_FMA_F32'):.0f} / {g('SQ_INSTS_VALU_TRANS_F32'):.0f} / {g('SQ_INSTS_VALU_CVT'):.0f} / {g('SQ_INSTS_VALU_INT32'):.0f} |")
print(f"| scalar ALU / scalar memory / branches | {g('SQ_INSTS_SALU'):.0f} / {g('SQ_INSTS_SMEM'):.2f} / {g('SQ_INSTS_BRANCH'):.1f} |")
print(f"| vector memory reads / writes | {g('SQ_INSTS_VMEM_RD'):.2f} / {g('SQ_INSTS_VMEM_WR'):.2f} |")
print(f"| instruction fetches / instruction-cache misses | {g('SQ_IFETCH'):.0f} / {g('SQC_ICACHE_MISSES'):.3f} |")
print(f"| scalar-cache requests / misses | {g('SQC_DCACHE_REQ'):.2f} / {g('SQC_DCACHE_MISSES'):.3f} |")
