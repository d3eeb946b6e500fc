#!/bin/bash
# LDS / wave-level counters of the fused step kernel at 65,536 bodies, 1 tick per launch (one --pmc pass, no graph).
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_lds_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY \
  --output-format csv -d $OUT/pmc -o k -- python $R/bench.py --steps 32 --warmup 4 --no-cpu-baseline --no-graph > $OUT/run.log 2> $OUT/run.err
find $OUT -name "*counter_collection.csv" | head -2
