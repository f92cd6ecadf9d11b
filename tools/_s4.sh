mkdir -p gpurun_out/s4
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/s4/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/s4/tests.log
tail -3 gpurun_out/s4/tests.log
MRA_TRACE_PLAN=1 MRA_TRACE_REPLAY=1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/s4/bench_c3.json 2> gpurun_out/s4/bench_c3.err; echo "bench rc=$?"
for v in e0 e1; do
  PYMRA_AMD_LIB=$PWD/pymra_amd/libmra_hip_$v.so timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end > gpurun_out/s4/bench_$v.json 2> gpurun_out/s4/bench_$v.err; echo "bench $v rc=$?"
  PYMRA_AMD_LIB=$PWD/pymra_amd/libmra_hip_$v.so timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -k "ulps or goldens" > gpurun_out/s4/ulps_$v.log 2>&1; tail -2 gpurun_out/s4/ulps_$v.log
done
