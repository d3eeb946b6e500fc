#!/usr/bin/env python3
"""One 20-replay window of BASELINE configs[1] under the profiler: where the time between `t0` and `t1` of bench.py goes.

  run:      rocprofv3 --kernel-trace --hip-trace --output-format csv -d <dir> -o k1 -- python tools/k1_window.py run
  analyse:  python tools/k1_window.py analyse <dir>            -> the table of profiles/r06_k1_floor.md

`run` = bench.py's window, 12 times: idle gap (what the barrier leaves), then invoke_batch(20) = one hipGraphLaunch of a
20-launch chain + hipStreamSynchronize.  `analyse` joins the kernel trace with the HIP API trace on their common clock."""
import csv
import glob
import statistics
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def run():
    import torch
    import bench
    ex, w, eff = bench.make_exec(65536, 0, 0, 1, True)
    ex.prepare(20)
    ex.invoke_batch(5)
    for _ in range(12):
        torch.cuda.synchronize()
        time.sleep(0.003)
        ex.invoke_batch(20)
    ex.prepare(4096)
    ex.invoke_batch(4096)
    ex.close()


def rows(pattern):
    for f in glob.glob(pattern, recursive=True):
        with open(f, newline="") as fh:
            yield from csv.DictReader(fh)


def analyse(d):
    ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows(f"{d}/**/*kernel_trace.csv") if "sixdof_step_kernel" in r["Kernel_Name"]))
    api = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"]) for r in rows(f"{d}/**/*hip_api_trace.csv")))
    launches = [(a, b) for a, b, f in api if f == "hipGraphLaunch"]
    syncs = [(a, b) for a, b, f in api if f == "hipStreamSynchronize"]
    win = []
    for (la, lb) in launches:
        mine = [k for k in ks if k[0] >= la]
        nxt = [x for x, _ in launches if x > la]
        if nxt:
            mine = [k for k in mine if k[0] < nxt[0]]
        if len(mine) != 20:
            continue                                            # the 32 / 128-launch chains of the long batch
        sy = [s for s in syncs if s[0] >= la][0]
        dur = [b - a for a, b in mine]
        gap = [mine[i + 1][0] - mine[i][1] for i in range(19)]
        win.append({"graph_launch_call": lb - la, "launch_to_first_kernel": mine[0][0] - la, "first_kernel": dur[0], "kernels_2_20_mean": statistics.mean(dur[1:]),
                    "kernel_min": min(dur), "gaps_mean": statistics.mean(gap), "gap_max": max(gap), "device_span": mine[-1][1] - mine[0][0],
                    "last_kernel_end_to_sync_return": sy[1] - mine[-1][1], "host_window": sy[1] - la})
    long_ = [b - a for a, b in ks[-2048:]]
    long_gaps = [ks[i + 1][0] - ks[i][1] for i in range(len(ks) - 2048, len(ks) - 1)]
    print(f"# {len(win)} windows of 20 launches (65,536 bodies, RK4 f64, one tick per launch), nanoseconds; median over the windows\n")
    print("| part of the window | median ns | per step (/20) us |\n|---|---|---|")
    med = lambda k: statistics.median(w_[k] for w_ in win)
    for k, label in (("graph_launch_call", "hipGraphLaunch call (host)"), ("launch_to_first_kernel", "hipGraphLaunch entry -> first kernel starts"),
                     ("device_span", "first kernel start -> last kernel end (device span)"), ("last_kernel_end_to_sync_return", "last kernel end -> hipStreamSynchronize returns"),
                     ("host_window", "hipGraphLaunch entry -> hipStreamSynchronize returns (= t1 - t0 of bench.py, minus Python)")):
        print(f"| {label} | {med(k):.0f} | {med(k) / 20e3:.3f} |")
    print("\n| inside the device span | median ns |\n|---|---|")
    for k, label in (("first_kernel", "duration of kernel 1"), ("kernels_2_20_mean", "mean duration of kernels 2..20"), ("kernel_min", "shortest kernel"),
                     ("gaps_mean", "mean gap between consecutive kernels (end -> start)"), ("gap_max", "largest gap")):
        print(f"| {label} | {med(k):.0f} |")
    print(f"\nlong batch (last 2,048 launches of a 4,096-launch batch): kernel duration mean {statistics.mean(long_):.0f} ns, min {min(long_)} ns; "
          f"gap mean {statistics.mean(long_gaps):.0f} ns, so {statistics.mean(long_) + statistics.mean(long_gaps):.0f} ns per launch")


if __name__ == "__main__":
    run() if sys.argv[1] == "run" else analyse(sys.argv[2])
