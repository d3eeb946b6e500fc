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
dur_grid = collections.defaultdict(list)
for f in glob.glob(f"{out}/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f, newline="")):
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        dur[r["Kernel_Name"]].append(us)
        dur_grid[(r["Kernel_Name"], int(r.get("Grid_Size") or r.get("Grid_Size_X") or 0))].append(us)
print("| kernel | grid | waves | VALU instr / wave | active (any / VALU) | issue-stalled | parked | avg us |")
print("|---|---|---|---|---|---|---|---|")
for (name, grid), c in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("SQ_WAVE_CYCLES", [0]))):
    m = {k: sum(v) / len(v) for k, v in c.items()}
    wc = m["SQ_WAVE_CYCLES"]
    d = dur_grid.get((name, grid)) or dur.get(name, [0.0])
    print(f"| `{name[:90]}` | {grid} | {m['SQ_WAVES']:.0f} | {m['SQ_INSTS_VALU'] / m['SQ_WAVES']:.0f} | "
          f"{m['SQ_ACTIVE_INST_ANY'] / wc:.1%} / {m['SQ_ACTIVE_INST_VALU'] / wc:.1%} | {m['SQ_WAIT_INST_ANY'] / wc:.1%} | "
          f"{m['SQ_WAIT_ANY'] / wc:.1%} | {sum(d) / len(d):.1f} |")


# ---- the VALU-issue roofline's inputs for bench.py (configs 3 / 4 / 5): instructions per wave and TICK, from the counters above ----
# python profiles/summarize_compute.py <dir> --json profiles/pmc_valu.json
if "--json" in sys.argv:
    import json
    TICKS = 1000                                     # tools/prof_compute_kernels.py TICKS_PER_LAUNCH (campaign kernels); the n-body tick is one sweep
    doc = {"source": "profiles/collect_compute.sh -> rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU ... (its own pass, no trace domains)",
           "peak": {"simds": 1024, "clock_hz": 2.4e9, "clocks_per_wave_instruction": 4, "wave_instructions_per_s": 1024 * 2.4e9 / 4,
                    "what": "one VALU instruction of a 64-wide wave occupies its SIMD's 16 lanes for 4 clocks (MI355X_MICROARCH.md)"}, "kernels": {}}
    for (name, grid), c in agg.items():
        m = {k: sum(v) / len(v) for k, v in c.items()}
        key = ("falcon9" if ("sixdof_step_kernel<float" in name and "PipeCustom" in name) else "apollo" if "apollo_rollout" in name else
               "nbody_allpairs" if "allpairs_kernel" in name else None)
        if key is None:
            continue
        ticks = TICKS if key in ("falcon9", "apollo") else 1
        d = dur.get(name, [0.0])
        doc["kernels"][key] = {"kernel": name[:120], "grid": grid, "waves": round(m["SQ_WAVES"]), "ticks_per_launch": ticks,
                               "valu_per_wave_per_tick": round(m["SQ_INSTS_VALU"] / m["SQ_WAVES"] / ticks, 2),
                               "issuing_fraction_of_wave_cycles": round(m["SQ_ACTIVE_INST_ANY"] / m["SQ_WAVE_CYCLES"], 4),
                               "profiled_us_per_launch": round(sum(d) / len(d), 2)}
    path = sys.argv[sys.argv.index("--json") + 1]
    open(path, "w").write(json.dumps(doc, indent=1))


# ---- the whole-world StableHLO ticks (tools/prof_world_modules.py): one kernel template, told apart by the grid ----
# python profiles/summarize_compute.py <dir> --world profiles/pmc_valu_world.json
if "--world" in sys.argv:
    import json
    keys = json.load(open(f"{out}/world_keys.json"))
    doc = {"source": "profiles/collect_world.sh -> rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU ... (its own pass, no trace domains); tools/prof_world_modules.py",
           "peak": {"simds": 1024, "clock_hz": 2.4e9, "clocks_per_wave_instruction": 4, "wave_instructions_per_s": 1024 * 2.4e9 / 4}, "kernels": {}}
    for (name, grid), c in agg.items():
        key = keys["grids"].get(str(grid))
        if key is None or "sixdof_step_kernel" not in name:
            continue
        m = {k: sum(v) / len(v) for k, v in c.items()}
        d = dur_grid.get((name, grid)) or [0.0]
        d = d[1:] if len(d) > 1 else d                       # (the first launch of a program pays its code upload)
        doc["kernels"][key] = {"kernel": name[:120], "grid": grid, "waves": round(m["SQ_WAVES"]), "ticks_per_launch": keys["ticks_per_launch"],
                               "valu_per_wave_per_tick": round(m["SQ_INSTS_VALU"] / m["SQ_WAVES"] / keys["ticks_per_launch"], 2),
                               "issuing_fraction_of_wave_cycles": round(m["SQ_ACTIVE_INST_ANY"] / m["SQ_WAVE_CYCLES"], 4),
                               "profiled_us_per_launch": round(sum(d) / len(d), 2)}
    path = sys.argv[sys.argv.index("--world") + 1]
    open(path, "w").write(json.dumps(doc, indent=1))
