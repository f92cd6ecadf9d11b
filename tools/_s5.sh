mkdir -p gpurun_out/s5
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -Wno-unused-result -Wno-pass-failed tools/chol_latency.hip -o /tmp/chol_latency 2> gpurun_out/s5/cl_build.log && timeout -k 10 120 /tmp/chol_latency > gpurun_out/s5/chol_latency.txt 2>&1
cat gpurun_out/s5/chol_latency.txt
MRA_BENCH_PROFILE_E2E=1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/s5/bench_c3.json 2> gpurun_out/s5/bench_c3.err; echo "bench rc=$?"
PYMRA_AMD_LIB=$PWD/pymra_amd/libmra_hip_sq.so timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end > gpurun_out/s5/bench_sq.json 2> gpurun_out/s5/bench_sq.err; echo "bench sq rc=$?"
PYMRA_AMD_LIB=$PWD/pymra_amd/libmra_hip_sq.so timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -k "ulps or goldens" > gpurun_out/s5/ulps_sq.log 2>&1; tail -2 gpurun_out/s5/ulps_sq.log
