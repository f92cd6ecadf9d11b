#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/s22
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/s22/t1.log 2>&1; rc=$?; echo "parity rc=$rc"; tail -12 gpurun_out/s22/t1.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-end-to-end > gpurun_out/s22/bench.json 2> gpurun_out/s22/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
b=json.load(open('gpurun_out/s22/bench.json'))
print(b['value'], b['ms_per_step'], b['roofline']['avg_launch_ms'], b['roofline']['frac'])
for k in b['roofline']['kernels']: print('%-70s %7.3f'%(k['name'][:70],k['ms_per_step']))
PY
timeout -k 10 200 python tools/shard_timing.py 8 > gpurun_out/s22/s8.txt 2>&1; grep "plain\|predict" gpurun_out/s22/s8.txt
