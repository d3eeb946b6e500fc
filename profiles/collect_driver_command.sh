#!/bin/bash
# rocprofv3 --kernel-trace --stats of EXACTLY the driver's command (python bench.py --gpus 1 --steps 20 --warmup 5): the per-kernel
# summary that stands beside the line's `roofline` (profiles/collect.sh traces a longer run with the side legs).
#   gpurun -- 'bash profiles/collect_driver_command.sh r06'   ->  gpurun_out/prof_driver_<tag>/driver_command.md
set -u
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_driver_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_line.json 2> $OUT/trace.err
cd $R
python - $OUT <<'PY' > $OUT/driver_command.md
import csv, glob, json, sys
O = sys.argv[1]
rows = []
for f in glob.glob(f"{O}/trace/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f, newline="")))
by = {}
for r in rows:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    by.setdefault((r["Kernel_Name"], r.get("Grid_Size") or r.get("Grid_Size_X") or ""), []).append(d)
line = json.loads(open(f"{O}/bench_line.json").read().strip().splitlines()[-1])
print("# rocprofv3 --kernel-trace --stats of the driver's command: `python bench.py --gpus 1 --steps 20 --warmup 5`\n")
print("Every kernel the command launched (warm-up 5 + timed 20 launches, the line's `long_batch` of 256 + 4,096 launches of the SAME kernel on the same handle, the")
print("4,096-row parity run).  Durations in microseconds (per-dispatch timestamps; the profiler serialises dispatches and adds about 1 µs to each).\n")
print("| kernel | grid | launches | avg us | min us | max us | total ms |\n|---|---|---|---|---|---|---|")
for (k, g), v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    print(f"| `{k[:110]}` | {g} | {len(v)} | {sum(v)/len(v):.3f} | {min(v):.3f} | {max(v):.3f} | {sum(v)/1e3:.3f} |")
step = sorted((v for (k, g), v in by.items() if "sixdof_step_kernel<double, 0" in k and str(g) == "65536"), key=len)
if step:
    v = step[-1]                      # the steady instantiation (the single launch with the first-tick `accel_in_check` is its own row)
    avg = sum(v) / len(v)
    print(f"\nThe step kernel at 65,536 bodies: {len(v)} launches, average {avg:.3f} us -> 384 B x 65,536 / {avg:.3f} us / 8 TB/s = {384*65536/avg/1e3/8000:.3f} of the HBM peak")
    print(f"(360 B, SURVEY 8(d)'s figure without the torque column: {360*65536/avg/1e3/8000:.3f}); the shortest launch: {min(v):.3f} us.  The line printed UNDER the profiler (every dispatch carries its overhead): value {line['value']:.4g}, roofline.avg_launch_us {line['roofline']['avg_launch_us']}, frac {line['roofline']['frac']}, long_batch {line['roofline'].get('long_batch')}.")
PY
cat $OUT/driver_command.md
