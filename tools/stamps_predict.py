"""Where does a row tile of the predictive cascade spend its cycles?  Needs the diagnostic build (make stamps):
   PYMRA_AMD_LIB=pymra_amd/libmra_hip_stamps.so python tools/stamps_predict.py"""
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
st = pl.buffer(3).view(np.uint64).reshape(-1, 16).astype(np.int64)
st = st[st[:, 15] > 0]
print("tiles with stamps:", len(st))
nl = c["M"]
tot = st[:, 15] - st[:, 0]
print("cycles per tile, start -> end: median %d  p10 %d  p90 %d" % (np.median(tot), np.percentile(tot, 10), np.percentile(tot, 90)))
def show(nm, d): print("  %-44s median %7d   mean %7d   p90 %7d" % (nm, np.median(d), d.mean(), np.percentile(d, 90)))
show("W/Tt loads + first Ut chunk staged", st[:, 1] - st[:, 0])
show("leaf update (8 chunks of 16 k)", st[:, 2] - st[:, 1])
prev = st[:, 2]
for m in range(nl - 1, -1, -1):
    show("level %d: barriers + LDS image written" % m, st[:, 3 + 2 * m] - prev)
    show("level %d: products" % m, st[:, 4 + 2 * m] - st[:, 3 + 2 * m])
    prev = st[:, 4 + 2 * m]
show("tail", st[:, 15] - prev)
