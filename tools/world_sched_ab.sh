#!/bin/bash
# A/B of LLVM scheduling strategies for the generated whole-world configs[1] module kernels (result of round 6: profiles/r06_world_sched_ab.txt — no gain).
#   gpurun -- 'bash tools/world_sched_ab.sh'
for f in "" "-mllvm -amdgpu-sched-strategy=iterative-ilp" "-mllvm -amdgpu-sched-strategy=max-ilp" "-mllvm -amdgpu-sched-strategy=max-memory-clause"; do
  echo "== SIXDOF_JIT_FLAGS='$f'"
  SIXDOF_JIT_FLAGS="$f" python tools/world_relaxed_ab.py gpurun_out/world_sched_ab.json 2>&1 | grep "^reference \|^relaxed_one_world " | cut -c1-200
done
