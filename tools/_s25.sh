#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/s25
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/s25/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 gpurun_out/s25/tests.log
[ $rc -eq 0 ] || exit 1
bash tools/collect_round.sh r04 > gpurun_out/s25/collect.log 2>&1; echo "collect rc=$?"; tail -3 gpurun_out/s25/collect.log
