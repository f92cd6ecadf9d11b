mkdir -p gpurun_out/s14
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/s14/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/s14/tests.log
tail -3 gpurun_out/s14/tests.log
timeout -k 10 400 python tools/c5_ab.py > gpurun_out/s14/c5.txt 2>&1
cat gpurun_out/s14/c5.txt
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/s14/bench_c3.json 2> gpurun_out/s14/bench_c3.err; echo "bench rc=$?"
