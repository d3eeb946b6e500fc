mkdir -p gpurun_out/r05c
python tools/spill_repro/run.py > gpurun_out/r05c/spill_repro.txt 2>&1
python tools/falcon9_pk_ab.py 32768 20000 > gpurun_out/r05c/falcon9_pk_ab.txt 2>&1
python -m pytest tests -m gpu -q -x > gpurun_out/r05c/gpu_tests.log 2>&1; echo rc=$? >> gpurun_out/r05c/gpu_tests.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r05c/bench_steps20.json 2> gpurun_out/r05c/bench_steps20.err
cat gpurun_out/r05c/spill_repro.txt | tail -40; cat gpurun_out/r05c/falcon9_pk_ab.txt | tail -8; tail -4 gpurun_out/r05c/gpu_tests.log; wc -c gpurun_out/r05c/bench_steps20.json
