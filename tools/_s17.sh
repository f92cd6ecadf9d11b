#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/s17
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "deep_wide or wide_blocks" > gpurun_out/s17/t1.log 2>&1; echo "deep rc=$?"; tail -3 gpurun_out/s17/t1.log
timeout -k 10 300 python tools/shard_timing.py --config c5 1 > gpurun_out/s17/c5_new.txt 2>&1; echo "c5 rc=$?"; cat gpurun_out/s17/c5_new.txt
