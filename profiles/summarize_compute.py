#!/usr/bin/env python3
"""Condense profiles/collect_compute.sh's SQ counter pass into a small markdown table (per-launch averages)."""
import collections
import csv
import glob
import sys

out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{out}/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f, newline="")):
        agg[(r["Kernel_Name"], int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for f in glob.glob(f"{out}/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f, newline="")):
        dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("| kernel | grid | waves | VALU instr / wave | active (any / VALU) | issue-stalled | parked | avg us |")
print("|---|---|---|---|---|---|---|---|")
for (name, grid), c in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("SQ_WAVE_CYCLES", [0]))):
    m = {k: sum(v) / len(v) for k, v in c.items()}
    wc = m["SQ_WAVE_CYCLES"]
    d = dur.get(name, [0.0])
    print(f"| `{name[:90]}` | {grid} | {m['SQ_WAVES']:.0f} | {m['SQ_INSTS_VALU'] / m['SQ_WAVES']:.0f} | "
          f"{m['SQ_ACTIVE_INST_ANY'] / wc:.1%} / {m['SQ_ACTIVE_INST_VALU'] / wc:.1%} | {m['SQ_WAIT_INST_ANY'] / wc:.1%} | "
          f"{m['SQ_WAIT_ANY'] / wc:.1%} | {sum(d) / len(d):.1f} |")
