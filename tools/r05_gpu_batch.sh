mkdir -p gpurun_out/r05d
touch /tmp/jit_marker; sleep 1
find elodin_amd/_jit -name '*.so' | sort > /tmp/jit_before.txt
timeout 1700 python -m pytest tests -m gpu -q --durations=10 -p no:cacheprovider > gpurun_out/r05d/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r05d/pytest.log
find elodin_amd/_jit -name '*.so' | sort > /tmp/jit_after.txt
comm -13 /tmp/jit_before.txt /tmp/jit_after.txt | sed 's/\.so$//' | while read f; do ls $f.so $f.json $f.hip 2>/dev/null; done > /tmp/jit_new.txt
tar czf gpurun_out/r05d/jit_new.tgz -T /tmp/jit_new.txt; wc -l < /tmp/jit_new.txt > gpurun_out/r05d/jit_new_count.txt
python __graft_entry__.py smoke > gpurun_out/r05d/smoke.txt 2>&1
tail -25 gpurun_out/r05d/pytest.log; cat gpurun_out/r05d/jit_new_count.txt; tail -2 gpurun_out/r05d/smoke.txt
