"""Where does a workgroup of the knot chain spend its time?  Needs the diagnostic build (make stamps):
   PYMRA_AMD_LIB=pymra_amd/libmra_hip_stamps.so python tools/stamps_knot_chain.py [world]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from pymra_amd import plan as P
from pymra_amd.sharding import shard_topology
from pymra_amd.topology import build_topology
import pymra_amd.MRATools as mt
world = int(sys.argv[1]) if len(sys.argv) > 1 else 1
c = bench.CONFIGS["c3"]
locs, y_obs = bench.make_inputs(c)
topo = build_topology(locs, c["r"], c["M"], c["J"])
local, red = shard_topology(topo, world, 0)
pl = P.HipPlan(local, 0); pl.set_locs(locs); pl.set_obs(y_obs, c["R"]); pl.set_kernel(mt.KIND_MATERN32, c["l"], c["sig"], 1.0)
if red >= 0: pl.set_reduce_level(red)
for _ in range(3):
    if red < 0: pl.run(True, True)
    else: pl.run(True, True, split=True); pl.resume()
st = pl.buffer(5).view(np.uint64).reshape(-1, 64).astype(np.int64)
st = st[st[:, 1] > 0]
n = int((st[0] > 0).sum())
print("workgroups with stamps: %d, stamps per workgroup: %d (10 ns ticks)" % (len(st), n))
d = np.diff(st[:, :n], axis=1) * 0.01
names = ["prologue (prefetch + sync)"] + [x for m in range((n - 2) // 4) for x in ("level %d: 1 knot rows down" % m, "level %d: 2 kInv" % m, "level %d: 3 Cholesky" % m, "level %d: 4 outputs" % m)]
for k, nm in enumerate(names):
    print("  %-30s median %6.2f us   p90 %6.2f us" % (nm, np.median(d[:, k]), np.percentile(d[:, k], 90)))
print("  whole workgroup: median %.2f us" % np.median((st[:, n - 1] - st[:, 0]) * 0.01))
# ---- the fused leaf solve + update of the same pass (side stream of a sharded run)
st = pl.buffer(6)
if st.size:
    st = st.view(np.uint64).reshape(-1, 8).astype(np.int64)
    st = st[st[:, 5] > 0]
    d = np.diff(st[:, :6], axis=1) * 0.01
    print("k_leaf_solve_update: %d workgroups" % len(st))
    for k, nm in enumerate(["stage Lc", "V load + row solve", "W load + first Ut stage", "update (7 chunks)", "stores"]):
        print("  %-30s median %6.2f us   p90 %6.2f us" % (nm, np.median(d[:, k]), np.percentile(d[:, k], 90)))
    print("  whole workgroup: median %.2f us;  kernel span %.2f us" % (np.median((st[:, 5] - st[:, 0]) * 0.01), (st[:, 5].max() - st[:, 0].min()) * 0.01))
    order = np.argsort(st[:, 0]); t0 = st[:, 0].min()
    print("  start times (us) of every 64th workgroup:", " ".join("%.0f" % ((st[i, 0] - t0) * 0.01) for i in order[::64]))
