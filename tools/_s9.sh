mkdir -p gpurun_out/s9
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -Wno-unused-result -Wno-pass-failed tools/chol_latency.hip -o /tmp/chol_latency 2> gpurun_out/s9/cl_build.log && timeout -k 10 120 /tmp/chol_latency > gpurun_out/s9/chol_latency.txt 2>&1
head -22 gpurun_out/s9/chol_latency.txt
bash tools/collect_round.sh r04 > gpurun_out/s9/collect.log 2>&1; echo "collect rc=$?"
tail -5 gpurun_out/s9/collect.log
