"""Where does a row tile of the prior cascade spend its cycles?  Needs the diagnostic build (make stamps):
   PYMRA_AMD_LIB=pymra_amd/libmra_hip_stamps.so python tools/stamps_cascade.py"""
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
for _ in range(3): pl.run(True, True)
st = pl.buffer(2).view(np.uint64).reshape(-1, 16).astype(np.int64)
st = st[st[:, 15] > 0]
print("tiles with stamps:", len(st))
d = np.diff(st, axis=1)
names = ["staging (wg start -> tile start)"] + [x for m in range(6) for x in ("level %d compute" % m, "level %d stores" % m)]
tot = st[:, 15] - st[:, 0]
print("cycles per tile, wg start -> end: median %d  p10 %d  p90 %d" % (np.median(tot), np.percentile(tot, 10), np.percentile(tot, 90)))
for k, nm in enumerate(names):
    print("  %-36s median %7d   mean %7d   p90 %7d" % (nm, np.median(d[:, k]), d[:, k].mean(), np.percentile(d[:, k], 90)))
print("  %-36s median %7d" % ("tail (var, y block, Ut y)", np.median(st[:, 15] - st[:, 13])))
first = st[::16]
print("second tile of a wave starts %d cycles after the workgroup (median)" % np.median(st[8::16, 1] - st[8::16, 0]))
