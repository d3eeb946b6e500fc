#!/bin/bash
# A mid-round GPU check (through gpurun from the repo root):  gpurun --timeout 1500 -- 'bash tools/r06_gpu_check.sh <tag> [suite]'
# the driver's bench command first (timed by `time`), the side legs into the sidecar, smoke(); with `suite` the whole GPU suite.
TAG=${1:-check}
cd $GRAFT_REPO_ROOT; O=gpurun_out/$TAG; mkdir -p $O
find elodin_amd/_jit -name '*.so' | sort > /tmp/jit_before.txt
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_steps20.json 2> $O/bench_steps20.err ) 2> $O/bench_steps20.time
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
if [ "${2:-}" = "suite" ]; then
  timeout 1500 python -m pytest tests -m gpu -q --durations=12 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
else
  timeout 600 python -m pytest tests/test_gpu_multirank_shared.py tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
fi
( time timeout 900 python bench.py --steps 20 --warmup 5 --extras --extras-out $O/bench_extras.json > $O/bench_extras_line.json 2> $O/bench_extras.err; echo "extras rc=$?" >> $O/bench_extras.err ) 2> $O/bench_extras.time
( time python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time
find elodin_amd/_jit -name '*.so' | sort > /tmp/jit_after.txt
comm -13 /tmp/jit_before.txt /tmp/jit_after.txt | sed 's/\.so$//' | while read f; do ls $f.so $f.json $f.hip 2>/dev/null; done > /tmp/jit_new.txt
tar czf $O/jit_new.tgz -T /tmp/jit_new.txt; wc -l < /tmp/jit_new.txt > $O/jit_new_count.txt
wc -c $O/bench_steps20.json; cat $O/bench_steps20.json; cat $O/bench_steps20.time; tail -5 $O/pytest.log; tail -2 $O/smoke.log; tail -3 $O/bench_extras.err; cat $O/bench_extras.time; cat $O/jit_new_count.txt
