mkdir -p gpurun_out/s6
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -Wno-unused-result -Wno-pass-failed tools/chol_latency.hip -o /tmp/chol_latency 2> gpurun_out/s6/cl_build.log && timeout -k 10 120 /tmp/chol_latency > gpurun_out/s6/chol_latency.txt 2>&1
head -30 gpurun_out/s6/chol_latency.txt
