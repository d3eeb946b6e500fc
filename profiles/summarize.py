#!/usr/bin/env python3
"""Condense the rocprofv3 CSVs of profiles/collect.sh into small committed files:
  <out>/summary_<tag>.md        per-kernel count / avg / min / max duration (from *_kernel_trace.csv), split by grid
  <out>/pmc_traffic.json        HBM bytes per launch per entity count (FETCH_SIZE*2 + WRITE_SIZE, KiB units)
FETCH_SIZE on gfx950 reports half of a wide coalesced read (MI355X_MICROARCH.md §HBM): doubled here; WRITE_SIZE
matched the byte count of this kernel's stores exactly (200 B/entity) and is used as is."""
import csv
import glob
import json
import sys
from collections import defaultdict

out, tag = sys.argv[1], sys.argv[2]


def rows(pattern):
    for f in glob.glob(pattern, recursive=True):
        with open(f, newline="") as fh:
            yield from csv.DictReader(fh)


stats = defaultdict(list)
for r in rows(f"{out}/trace/**/*kernel_trace.csv"):
    grid = int(r.get("Grid_Size_X") or r.get("Grid_Size") or 0)
    dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    stats[(r["Kernel_Name"], grid)].append(dur)

lines = [f"# rocprofv3 --kernel-trace --stats summary ({tag})", "",
         "Command: `python bench.py --steps 2048 --warmup 128 --no-cpu-baseline --extras --skip-legs telemetry_commit,history_stream,monte_carlo_example,world_module,build` (profiles/collect.sh).",
         "Durations in microseconds; `grid` = total work-items (= entities for the step kernel). The 65536-entity",
         "step kernel appears with ticks_per_launch = 1 (timed region + warmup), = 64 (`fused`) and = 64 with the",
         "telemetry ring (`recording`): rows are split by duration.  PipeStatic<2, 3> = gravity | body_torque;",
         "the trailing int is the cache-policy code of the instantiation (load * 8 + store: 1 = plain loads + nt stores,",
         "2 = plain loads + sc1 stores, 9 = nt both ways; csrc/step_kernel.hpp); PipeCustom = a generated pipe / program",
         "(`<float, 1, PipeCustom, 1>` at grid 32768 = the Falcon 9 ascent campaign, 1000 ticks per launch).", "",
         "| kernel | grid | launches | avg us | min us | max us | total ms |", "|---|---|---|---|---|---|---|"]
for (name, grid), d in sorted(stats.items(), key=lambda kv: -sum(kv[1])):
    groups = [("", d)]
    if "sixdof_step_kernel" in name and grid == 65536:
        # the bench runs this grid with 1 tick/launch (timed region), 64 ticks/launch (`fused`, ~40 us) and
        # 64 ticks/launch with the telemetry ring (`recording`, ~130-160 us)
        groups = [(" [1 tick/launch]", [x for x in d if x < 15000]),
                  (" [8 ticks/launch, `telemetry_commit` leg]", [x for x in d if 15000 <= x < 30000]),
                  (" [64 ticks/launch]", [x for x in d if 30000 <= x < 80000]),
                  (" [64 ticks/launch + telemetry ring]", [x for x in d if x >= 80000])]
    for label, g in groups:
        if g:
            lines.append(f"| `{name}`{label} | {grid} | {len(g)} | {sum(g)/len(g)/1e3:.3f} | {min(g)/1e3:.3f} | "
                         f"{max(g)/1e3:.3f} | {sum(g)/1e6:.3f} |")

pmc = defaultdict(lambda: defaultdict(list))
for which in ("fetch", "write"):
    for r in rows(f"{out}/pmc_{which}_*/**/*counter_collection.csv"):
        if "sixdof_step_kernel" not in r["Kernel_Name"]:
            continue
        grid = int(r.get("Grid_Size") or 0)
        pmc[grid][r["Counter_Name"]].append(float(r["Counter_Value"]))
traffic = {}
lines += ["", "## HBM traffic per launch (separate --pmc passes)", "",
          "| entities | FETCH_SIZE KiB (raw) | read bytes (x2, gfx950) | WRITE_SIZE KiB | write bytes | total bytes | algorithmic (384 B/entity) |",
          "|---|---|---|---|---|---|---|"]
for grid, c in sorted(pmc.items()):
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        f = sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"])
        w = sum(c["WRITE_SIZE"]) / len(c["WRITE_SIZE"])
        rd, wr = 2 * f * 1024, w * 1024
        traffic[str(grid)] = {"fetch_size_kib_raw": f, "write_size_kib": w, "read_bytes": rd, "write_bytes": wr,
                              "hbm_bytes_per_launch": rd + wr, "correction": "FETCH_SIZE x2 (gfx950), KiB units"}
        lines.append(f"| {grid} | {f:.1f} | {rd:.0f} | {w:.1f} | {wr:.0f} | {rd+wr:.0f} | {384*grid} |")
# stamp: the collection is valid for THESE kernel sources (bench.py reports traffic: null once they change)
import hashlib
import os
from pathlib import Path
root = Path(__file__).resolve().parent.parent
h = hashlib.sha256()
for name in ("step_kernel.hpp", "effectors.hpp", "spatial.hpp", "kernels.hpp", "sixdof_kernels.hip"):      # = bench.STEP_KERNEL_SOURCES
    h.update((root / "elodin_amd" / "csrc" / name).read_bytes())
traffic["_stamp"] = {"step_kernel_hash": h.hexdigest()[:16], "tag": tag, "commit": os.environ.get("SIXDOF_COMMIT", "unknown (no .git on the GPU box)")}
open(f"{out}/summary_{tag}.md", "w").write("\n".join(lines) + "\n")
json.dump(traffic, open(f"{out}/pmc_traffic.json", "w"), indent=1)
print("\n".join(lines))
