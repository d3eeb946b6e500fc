#!/bin/bash
# A/B of LLVM scheduling strategies for the hand-written Apollo rollout kernel: build the variant libraries first (on the build host):
#   cd elodin_amd/csrc && for v in max-ilp iterative-ilp; do mkdir -p build/$v; hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -mllvm -amdgpu-sched-strategy=$v \
#     -c apollo_kernels.hip -o build/$v/apollo.o; hipcc --offload-arch=gfx950 -shared -fPIC -o ../../gpurun_ab_lib_$v.so build/$v/apollo.o $(ls build/*.o | grep -v apollo) -ldl; done
# then   gpurun -- 'bash tools/apollo_sched_ab.sh'   (result of round 6: profiles/r06_apollo_sched_ab.txt — no gain)
for v in base max-ilp iterative-ilp; do
  if [ $v = base ]; then unset SIXDOF_LIBRARY; else export SIXDOF_LIBRARY=$PWD/gpurun_ab_lib_$v.so; fi
  for r in 1 2; do
    python bench.py --extras --only-legs apollo_mc --no-cpu-baseline --extras-out gpurun_out/apollo_ab_${v}_$r.json > /dev/null 2>gpurun_out/apollo_ab_${v}_$r.err
    python - <<PY
import json
d=json.load(open("gpurun_out/apollo_ab_${v}_$r.json"))["apollo_mc"]
print("$v", $r, d["seconds"], d["roofline"]["frac_of_occupied_simds"], d["rollouts_vs_time"]["by_rollouts"]["8192"])
PY
  done
done
