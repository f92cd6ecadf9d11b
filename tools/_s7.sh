mkdir -p gpurun_out/s7
MRA_TRACE_PLAN=1 timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/s7/bench_c3.json 2> gpurun_out/s7/bench_c3.err; echo "bench rc=$?"
grep "total\|uploads" gpurun_out/s7/bench_c3.err | tail -12
