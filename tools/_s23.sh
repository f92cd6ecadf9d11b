#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/s23
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/s23/kt -- python3 tools/shard_timing.py --config c5 1 > gpurun_out/s23/kt.log 2>&1
f=$(ls -t gpurun_out/s23/kt/*/*_kernel_stats.csv | head -1)
grep "k_predict\|k_syrk\|k_leaf_gemm<2, 2, 0, 2" $f | cut -c1-200
