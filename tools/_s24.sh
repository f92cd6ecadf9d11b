#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/s24
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "deep_wide or wide_blocks" > gpurun_out/s24/t1.log 2>&1; rc=$?; echo "deep rc=$rc"; tail -15 gpurun_out/s24/t1.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/shard_timing.py --config c5 1 8 > gpurun_out/s24/c5.txt 2>&1; echo "c5 rc=$?"; grep "world\|predict\|update\|plain" gpurun_out/s24/c5.txt
