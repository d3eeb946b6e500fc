#!/bin/bash
# Collect the rocprofv3 evidence for bench.py on the GPU box (run through gpurun from the repo root):
#   gpurun -- 'bash profiles/collect.sh r01'
# Three separate passes of the SAME command (kernel trace; FETCH_SIZE; WRITE_SIZE — the two TCC counters do
# not fit one pass, and PMC passes must not be combined with other tracing).  Outputs land in
# gpurun_out/prof_<tag>/ ; profiles/summarize.py condenses them into the files committed under profiles/.
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# telemetry_commit is left out of the traced run: its launches share the device with D2H copies and would blur the
# per-kernel averages of the timed region
CMD="python $R/bench.py --steps 2048 --warmup 128 --no-cpu-baseline --extras --extras-out $OUT/bench_extras_trace.json --skip-legs telemetry_commit,history_stream,monte_carlo_example,world_module,build"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- $CMD > $OUT/bench_trace.json 2> $OUT/trace.err
# PMC passes: only the timed region's kernel (1 tick per launch), no graph replay (counters are attributed per
# dispatch), fewer steps; once at the bench size and once at 4,194,304 bodies (the roofline_hbm leg's size)
for SIZE in 65536 4194304; do
  CMDP="python $R/bench.py --entities $SIZE --steps 32 --warmup 4 --no-cpu-baseline --no-graph"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$SIZE -o bench -- $CMDP > $OUT/bench_fetch_$SIZE.json 2> $OUT/fetch_$SIZE.err
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$SIZE -o bench -- $CMDP > $OUT/bench_write_$SIZE.json 2> $OUT/write_$SIZE.err
done
python $R/profiles/summarize.py $OUT $TAG
find $OUT -type f | head -40
