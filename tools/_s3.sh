mkdir -p gpurun_out/s3
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/s3/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/s3/tests.log
tail -3 gpurun_out/s3/tests.log
MRA_TRACE_PLAN=1 MRA_TRACE_REPLAY=1 timeout -k 10 300 python tools/e2e_breakdown.py > gpurun_out/s3/e2e.txt 2>&1
grep "^rep" gpurun_out/s3/e2e.txt
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/s3/bench_c3.json 2> gpurun_out/s3/bench_c3.err; echo "bench rc=$?"
