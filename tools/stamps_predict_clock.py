"""Shader clock during the predictive cascade: shader-clock stamps against the 100 MHz real-time counter (diagnostic build).
   PYMRA_AMD_LIB=pymra_amd/libmra_hip_stamps.so python tools/stamps_predict_clock.py [dbg]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from pymra_amd import plan as P
from pymra_amd.topology import build_topology
import pymra_amd.MRATools as mt
c = bench.CONFIGS["c3"]
locs, y_obs = bench.make_inputs(c)
topo = build_topology(locs, c["r"], c["M"], c["J"])
pl = P.HipPlan(topo, 0); pl.set_locs(locs); pl.set_obs(y_obs, c["R"]); pl.set_kernel(mt.KIND_MATERN32, c["l"], c["sig"], 1.0)
if len(sys.argv) > 1: pl.set_option(99, int(sys.argv[1]))
for _ in range(3): pl.run(True, True)
st = pl.buffer(3).view(np.uint64).reshape(-1, 16).astype(np.int64)
st = st[st[:, 15] > 0]
cyc = st[:, 15] - st[:, 0]; wall = (st[:, 14] - st[:, 13]) * 10.0     # ns
print("workgroup: median %d shader cycles in %.1f us -> %.3f GHz" % (np.median(cyc), np.median(wall) / 1e3, np.median(cyc) / np.median(wall)))
span = (st[:, 14].max() - st[:, 13].min()) * 10.0 / 1e3
print("kernel span %.1f us; mean workgroups in flight %.1f" % (span, wall.sum() / 4 / 1e3 / span))
