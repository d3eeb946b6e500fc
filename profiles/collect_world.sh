#!/bin/bash
# SQ counters for the whole-world StableHLO ticks (bench.py `world_module`): one --pmc pass, one kernel-trace pass.
#   gpurun -- 'bash profiles/collect_world.sh r05'   then   python profiles/summarize_compute.py gpurun_out/prof_world_r05 --world profiles/pmc_valu_world.json
set -u
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_world_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY \
  --output-format csv -d $OUT/pmc -o k -- python $R/tools/prof_world_modules.py $OUT/world_keys.json > $OUT/run.log 2> $OUT/run.err
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o k -- python $R/tools/prof_world_modules.py $OUT/world_keys.json >> $OUT/run.log 2>> $OUT/run.err
cd $R && python profiles/summarize_compute.py $OUT --world profiles/pmc_valu_world.json > $OUT/world_kernels_pmc.md 2> $OUT/summarize.err
cat $OUT/world_kernels_pmc.md; tail -3 $OUT/run.err
