#!/bin/bash
mkdir -p gpurun_out/s26
for rep in 1 2; do
for lib in g3 g0 g2 g1; do
  if [ $lib = g3 ]; then unset PYMRA_AMD_LIB; else export PYMRA_AMD_LIB=pymra_amd/libmra_hip_$lib.so; fi
  timeout -k 10 200 python tools/shard_timing.py 1 8 > gpurun_out/s26/c3_${lib}_$rep.txt 2>&1
  echo "== $lib $rep: $(grep 'predict' gpurun_out/s26/c3_${lib}_$rep.txt | awk '{print $(NF-1)}' | tr '\n' ' ') | $(grep 'plain' gpurun_out/s26/c3_${lib}_$rep.txt | awk '{print $2}' | tr '\n' ' ')"
done; done
for lib in g3 g0; do
  if [ $lib = g3 ]; then unset PYMRA_AMD_LIB; else export PYMRA_AMD_LIB=pymra_amd/libmra_hip_$lib.so; fi
  timeout -k 10 200 python tools/shard_timing.py --config c5 1 > gpurun_out/s26/c5_${lib}.txt 2>&1
  echo "== c5 $lib: $(grep 'predict_hi\|plain' gpurun_out/s26/c5_${lib}.txt | awk '{print $(NF-1), $2}' | tr '\n' ' ')"
done
