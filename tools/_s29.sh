#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/s29
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/s29/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -12 gpurun_out/s29/tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/shard_timing.py --config c5 --likelihood-only 1 8 > gpurun_out/s29/c5l.txt 2>&1; echo "rc=$?"; grep "world\|prior\|plain" gpurun_out/s29/c5l.txt
