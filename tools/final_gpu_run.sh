#!/bin/bash
# The round's closing GPU run (through gpurun from the repo root):  gpurun --timeout 3000 -- 'bash tools/final_gpu_run.sh r06 [quick]'
# The driver's bench command first (cold box, timed), the whole GPU suite, the rocprofv3 collections (skipped with `quick`), then the
# bench lines again — after the PMC passes, so that they carry `roofline.traffic` measured on THIS kernel source (the fresh
# pmc_traffic.json is put in place first) — and the side legs into their sidecar.  Outputs under gpurun_out/final_<tag>/ — copy what
# is to be judged into profiles/.  Also packs the JIT objects the box had to compile, so the build container can keep elodin_amd/_jit exact.
TAG=${1:-r06}
cd $GRAFT_REPO_ROOT; O=gpurun_out/final_$TAG; mkdir -p $O
find elodin_amd/_jit -name '*.so' | sort > /tmp/jit_before.txt
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_steps20_first.json 2> $O/bench_steps20_first.err ) 2> $O/bench_steps20_first.time
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
timeout 1700 python -m pytest tests -m gpu -q --durations=15 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
find elodin_amd/_jit -name '*.so' | sort > /tmp/jit_after.txt
comm -13 /tmp/jit_before.txt /tmp/jit_after.txt | sed 's/\.so$//' | while read f; do ls $f.so $f.json $f.hip 2>/dev/null; done > /tmp/jit_new.txt
tar czf $O/jit_new.tgz -T /tmp/jit_new.txt; wc -l < /tmp/jit_new.txt > $O/jit_new_count.txt
if [ "${2:-}" != "quick" ]; then
  bash profiles/collect.sh $TAG > $O/collect.log 2>&1
  [ -s gpurun_out/prof_$TAG/pmc_traffic.json ] && cp gpurun_out/prof_$TAG/pmc_traffic.json profiles/pmc_traffic.json
  bash profiles/collect_compute.sh $TAG > $O/collect_compute.log 2>&1
  python profiles/summarize_compute.py gpurun_out/prof_compute_$TAG --json profiles/pmc_valu.json > $O/compute_kernels_pmc.md 2> $O/summarize_compute.err
  bash profiles/collect_world.sh $TAG > $O/collect_world.log 2>&1
  bash tools/r06_pair_prof.sh pair_$TAG > $O/pair_prof.log 2>&1
fi
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_steps20.json 2> $O/bench_steps20.err ) 2> $O/bench_steps20.time
python bench.py > $O/bench_default.json 2> $O/bench_default.err
( time python bench.py --steps 20 --warmup 5 --extras --extras-out $O/bench_extras.json > $O/bench_extras_line.json 2> $O/bench_extras.err; echo "extras rc=$?" >> $O/bench_extras.err ) 2> $O/bench_extras.time
timeout 600 python tools/k1_floor.py > $O/k1_split.txt 2>&1
cut -c1-700 $O/bench_steps20.json; echo; cat $O/bench_steps20_first.time; tail -25 $O/pytest.log; tail -2 $O/smoke.log; tail -3 $O/bench_extras.err; cat $O/jit_new_count.txt
