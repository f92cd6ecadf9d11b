mkdir -p gpurun_out/s8
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/s8/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/s8/tests.log
tail -3 gpurun_out/s8/tests.log
MRA_TRACE_PLAN=1 MRA_TRACE_REPLAY=1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/s8/bench_c3.json 2> gpurun_out/s8/bench_c3.err; echo "bench rc=$?"
grep "total\|replay_q" gpurun_out/s8/bench_c3.err | tail -8 | cut -c1-250
