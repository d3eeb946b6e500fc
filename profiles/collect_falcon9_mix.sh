#!/bin/bash
# Instruction mix of the generated Falcon 9 campaign kernel (32,768 rollouts, 1000 ticks per launch): separate rocprofv3 --pmc
# passes of tools/prof_falcon9_kernel.py, nothing else traced.   gpurun -- 'bash profiles/collect_falcon9_mix.sh r04'
set -u
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_f9mix_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
k=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_BRANCH" \
           "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQC_DCACHE_REQ SQC_DCACHE_MISSES"; do
  k=$((k+1))
  timeout 200 rocprofv3 --pmc $set --output-format csv -d $OUT/p$k -o k -- python $R/tools/prof_falcon9_kernel.py > $OUT/run$k.log 2>&1
done
python $R/profiles/summarize_falcon9_mix.py $OUT
