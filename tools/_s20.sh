#!/bin/bash
mkdir -p gpurun_out/s20
timeout -k 10 300 python tools/shard_timing.py --config c5 --likelihood-only 1 8 > gpurun_out/s20/c5_lik.txt 2>&1; echo "rc=$?"; cat gpurun_out/s20/c5_lik.txt
