#!/bin/bash
# Per-launch timeline (start, end, queue) of one C3 pass from a rocprofv3 kernel trace; run through gpurun.
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/ov; mkdir -p gpurun_out/ov
cat > gpurun_out/ov_run.py <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import bench
from pymra_amd import plan as P
from pymra_amd.topology import build_topology
import pymra_amd.MRATools as mt
c = bench.CONFIGS["c3"]
locs, y_obs = bench.make_inputs(c)
topo = build_topology(locs, c["r"], c["M"], c["J"])
pl = P.HipPlan(topo, 0); pl.set_locs(locs); pl.set_obs(y_obs, c["R"]); pl.set_kernel(mt.KIND_MATERN32, c["l"], c["sig"], 1.0)
for _ in range(4): pl.run(True, True)
pl.close()
PY
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ov -- python3 gpurun_out/ov_run.py > gpurun_out/ov/log.txt 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/ov/*/*_kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
seq=[(r['Kernel_Name'][:40],int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Queue_Id'],r['Stream_Id']) for r in rows]
seq.sort(key=lambda x:x[1])
last=seq[-40:]
t0=last[0][1]
for n,s,e,q,st in last: print("%-42s start %9.1f end %9.1f us  dur %8.1f  queue %s stream %s"%(n,(s-t0)/1e3,(e-t0)/1e3,(e-s)/1e3,q,st))
PY
