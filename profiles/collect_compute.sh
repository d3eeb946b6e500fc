#!/bin/bash
# SQ counters for the two compute-bound kernels (n-body all-pairs, generated Falcon 9 program): one --pmc pass.
#   gpurun -- 'bash profiles/collect_compute.sh r01'   then   python profiles/summarize_compute.py gpurun_out/prof_compute_r01
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_compute_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY \
  --output-format csv -d $OUT/pmc -o k -- python $R/tools/prof_compute_kernels.py > $OUT/run.log 2> $OUT/run.err
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o k -- python $R/tools/prof_compute_kernels.py >> $OUT/run.log 2>> $OUT/run.err
find $OUT -name "*.csv" | head
