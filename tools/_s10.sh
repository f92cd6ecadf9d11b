mkdir -p gpurun_out/s10
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/s10/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/s10/tests.log
tail -3 gpurun_out/s10/tests.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end > gpurun_out/s10/bench_new.json 2> gpurun_out/s10/bench_new.err; echo "bench rc=$?"
PYMRA_AMD_LIB=$PWD/pymra_amd/libmra_hip_oldatom.so timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end > gpurun_out/s10/bench_old.json 2> gpurun_out/s10/bench_old.err; echo "bench old rc=$?"
timeout -k 10 300 python tools/shard_timing.py 1 8 > gpurun_out/s10/shard_new.txt 2>&1
PYMRA_AMD_LIB=$PWD/pymra_amd/libmra_hip_oldatom.so timeout -k 10 300 python tools/shard_timing.py 1 8 > gpurun_out/s10/shard_old.txt 2>&1
grep "plain\|world" gpurun_out/s10/shard_new.txt gpurun_out/s10/shard_old.txt
timeout -k 10 300 python bench.py --config c5 --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end > gpurun_out/s10/bench_c5_new.json 2> gpurun_out/s10/bench_c5_new.err; echo "c5 rc=$?"
