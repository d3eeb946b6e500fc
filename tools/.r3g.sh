cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3g; O=gpurun_out/r3g
find elodin_amd/_jit -name '*.so' | sort > /tmp/jit_before.txt
python bench.py --steps 20 --warmup 5 > $O/bench_steps20.json 2> $O/bench_steps20.err
python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 1300 python -m pytest tests -m gpu -q --durations=25 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
find elodin_amd/_jit -name '*.so' | sort > /tmp/jit_after.txt
comm -13 /tmp/jit_before.txt /tmp/jit_after.txt | sed 's/\.so$//' | while read f; do ls $f.so $f.json $f.hip 2>/dev/null; done > /tmp/jit_new.txt
tar czf $O/jit_new.tgz -T /tmp/jit_new.txt; wc -l /tmp/jit_new.txt
bash profiles/collect.sh r03 > $O/collect.log 2>&1
cat $O/bench_steps20.json; tail -45 $O/pytest.log
