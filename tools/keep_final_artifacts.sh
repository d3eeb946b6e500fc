#!/bin/bash
# After `gpurun -- 'bash tools/final_gpu_run.sh <tag>'` has merged its outputs into gpurun_out/: copy what is to be judged into profiles/
# (tracked) and unpack the JIT objects the box compiled.   bash tools/keep_final_artifacts.sh r06
TAG=${1:-r06}
cd "$(dirname "$0")/.."; O=gpurun_out/final_$TAG; P=profiles
cp $O/bench_steps20.json $P/${TAG}_bench_steps20.json
cp $O/bench_steps20_first.json $P/${TAG}_bench_steps20_first_cold_box.json
cp $O/bench_default.json $P/${TAG}_bench_unprofiled.json
cp $O/bench_extras.json $P/${TAG}_bench_extras.json
cp $O/pytest.log $P/${TAG}_gpu_pytest.log
cp $O/k1_split.txt $P/${TAG}_k1_split_raw.txt
[ -s gpurun_out/prof_$TAG/summary_$TAG.md ] && cp gpurun_out/prof_$TAG/summary_$TAG.md $P/${TAG}_rocprof_summary.md
[ -s gpurun_out/prof_$TAG/trace/bench_kernel_stats.csv ] && cp gpurun_out/prof_$TAG/trace/bench_kernel_stats.csv $P/${TAG}_kernel_stats.csv
[ -s gpurun_out/prof_$TAG/bench_trace.json ] && cp gpurun_out/prof_$TAG/bench_trace.json $P/${TAG}_bench_under_rocprof.json
[ -s gpurun_out/prof_$TAG/pmc_traffic.json ] && cp gpurun_out/prof_$TAG/pmc_traffic.json $P/pmc_traffic.json
[ -s $O/compute_kernels_pmc.md ] && cp $O/compute_kernels_pmc.md $P/${TAG}_compute_kernels_pmc.md
python profiles/summarize_compute.py gpurun_out/prof_compute_$TAG --json $P/pmc_valu.json > /dev/null 2>&1
python profiles/summarize_compute.py gpurun_out/prof_world_$TAG --world $P/pmc_valu_world.json > $P/${TAG}_world_kernels_pmc.md 2> /dev/null
[ -s gpurun_out/pair_$TAG/pair_kernels.md ] && cp gpurun_out/pair_$TAG/pair_kernels.md $P/${TAG}_pair_kernels_final.md
tar xzf $O/jit_new.tgz -C . 2>/dev/null; echo "jit objects now: $(ls elodin_amd/_jit/*.so | wc -l)"
tail -3 $P/${TAG}_gpu_pytest.log; python -c "
import json; d=json.loads(open('$P/${TAG}_bench_steps20.json').read().strip().splitlines()[-1]); print(len(open('$P/${TAG}_bench_steps20.json').read()), d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'])"
