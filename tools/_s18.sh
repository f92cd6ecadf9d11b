#!/bin/bash
mkdir -p gpurun_out/s18
for v in 0 1; do
timeout -k 10 300 tools/syrk_lab_w$v 1024 > gpurun_out/s18/syrk_lab_w$v.txt 2>&1; echo "rc=$?"; cat gpurun_out/s18/syrk_lab_w$v.txt
done
