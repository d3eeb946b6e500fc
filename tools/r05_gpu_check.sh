#!/bin/bash
# A mid-round GPU check (through gpurun from the repo root):  gpurun --timeout 1500 -- 'bash tools/r05_gpu_check.sh <tag>'
# the whole GPU suite, smoke(), the driver's bench command; packs the JIT objects the box had to compile.
TAG=${1:-check}
cd $GRAFT_REPO_ROOT; O=gpurun_out/$TAG; mkdir -p $O
find elodin_amd/_jit -name '*.so' | sort > /tmp/jit_before.txt
ls elodin_amd/_jit | grep -c '^pch_' > $O/pch_files_that_travelled.txt
timeout 1100 python -m pytest tests -m gpu -q --durations=12 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
find elodin_amd/_jit -name '*.so' | sort > /tmp/jit_after.txt
comm -13 /tmp/jit_before.txt /tmp/jit_after.txt | sed 's/\.so$//' | while read f; do ls $f.so $f.json $f.hip 2>/dev/null; done > /tmp/jit_new.txt
tar czf $O/jit_new.tgz -T /tmp/jit_new.txt; wc -l < /tmp/jit_new.txt > $O/jit_new_count.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_steps20.json 2> $O/bench_steps20.err
cut -c1-600 $O/bench_steps20.json; tail -22 $O/pytest.log; cat $O/smoke.log | tail -2; cat $O/jit_new_count.txt $O/pch_files_that_travelled.txt
