#!/bin/bash
# The round's closing GPU run (through gpurun from the repo root):  gpurun --timeout 2700 -- 'bash tools/final_gpu_run.sh r05 [quick]'
# The whole GPU suite, the rocprofv3 collections (skipped with `quick`), then the bench lines (the driver's command and the
# default run) — after the PMC passes, so that the lines carry `roofline.traffic` measured on THIS kernel source (the fresh
# pmc_traffic.json is put in place first).  Outputs under gpurun_out/final_<tag>/ — copy what is to be judged into profiles/.
# Also lists which JIT objects the run used and packs the ones it had to compile, so the build container can keep
# elodin_amd/_jit exact.
TAG=${1:-r05}
cd $GRAFT_REPO_ROOT; O=gpurun_out/final_$TAG; mkdir -p $O
touch /tmp/jit_marker; sleep 1
find elodin_amd/_jit -name '*.so' | sort > /tmp/jit_before.txt
python bench.py --steps 20 --warmup 5 --no-campaigns > $O/bench_steps20_first.json 2> $O/bench_steps20_first.err
timeout 1700 python -m pytest tests -m gpu -q --durations=15 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
find elodin_amd/_jit -name '*.so' -newer /tmp/jit_marker | sort > $O/jit_used.txt
find elodin_amd/_jit -name '*.so' | sort > /tmp/jit_after.txt
comm -13 /tmp/jit_before.txt /tmp/jit_after.txt | sed 's/\.so$//' | while read f; do ls $f.so $f.json $f.hip 2>/dev/null; done > /tmp/jit_new.txt
tar czf $O/jit_new.tgz -T /tmp/jit_new.txt; wc -l < /tmp/jit_new.txt > $O/jit_new_count.txt
if [ "${2:-}" != "quick" ]; then
  bash profiles/collect.sh $TAG > $O/collect.log 2>&1
  [ -s gpurun_out/prof_$TAG/pmc_traffic.json ] && cp gpurun_out/prof_$TAG/pmc_traffic.json profiles/pmc_traffic.json
  bash profiles/collect_compute.sh $TAG > $O/collect_compute.log 2>&1
  python profiles/summarize_compute.py gpurun_out/prof_compute_$TAG --json profiles/pmc_valu.json > $O/compute_kernels_pmc.md 2> $O/summarize_compute.err
  bash profiles/collect_falcon9_mix.sh $TAG > $O/falcon9_instruction_mix.md 2> $O/falcon9_mix.err
fi
python bench.py --steps 20 --warmup 5 > $O/bench_steps20.json 2> $O/bench_steps20.err
python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 240 python tools/falcon9_k1.py > $O/falcon9_k1_policy_ab.txt 2> $O/falcon9_k1.err
timeout 60 python tools/apollo_perf.py 8192 10000 > $O/apollo_perf.txt 2>&1
timeout 300 python tools/falcon9_pk_ab.py 32768 20000 > $O/falcon9_pk_ab.txt 2>&1
( cd tools/ubench && ./pk_f32 ) > $O/ubench_pk_f32.txt 2>&1
cut -c1-400 $O/bench_steps20.json; tail -25 $O/pytest.log; cat $O/jit_new_count.txt
