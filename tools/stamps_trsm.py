"""Where does a row tile of the leaf solves (k_trsm_rows2) spend its cycles?  Diagnostic build (make stamps):
   PYMRA_AMD_LIB=pymra_amd/libmra_hip_stamps.so python tools/stamps_trsm.py"""
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
st = pl.buffer(4).view(np.uint64).reshape(-1, 64, 8).astype(np.int64)
ok = st[:, :, 5] > 0
print("problems %d, row tiles with stamps %d" % (st.shape[0], ok.sum()))
wg_end = np.where(ok, st[:, :, 5], 0).max(axis=1); wg0 = np.where(ok, st[:, :, 0], 1 << 62).min(axis=1)
print("workgroup start -> last tile done: median %d cycles" % np.median(wg_end - wg0))
s = st[ok]
def show(nm, d): print("  %-40s median %7d   mean %7d   p90 %7d" % (nm, np.median(d), d.mean(), np.percentile(d, 90)))
show("staging L into LDS", s[:, 1] - s[:, 0])
show("tile: loads issued -> arrived", s[:, 3] - s[:, 2])
show("tile: solve (column tiles 0..nt-1)", s[:, 4] - s[:, 3])
show("tile: variance update", s[:, 5] - s[:, 4])
show("tile: whole", s[:, 5] - s[:, 2])
# the shader clock is per XCD: group the workgroups by clock domain (gaps of more than 1e8 cycles between start times)
order = np.argsort(wg0); w0 = wg0[order]; w1 = wg_end[order]
cuts = np.flatnonzero(np.diff(w0) > 1e8) + 1
for g, (a, b) in enumerate(zip(np.r_[0, cuts], np.r_[cuts, len(w0)])):
    span = w1[a:b].max() - w0[a:b].min()
    print("  clock domain %d: %4d workgroups, span %8d cycles, mean in flight %.1f (32 CUs)" % (g, b - a, span, (w1[a:b] - w0[a:b]).sum() / span))
a, b = 0, (cuts[0] if len(cuts) else len(w0))
print("domain 0 start-time quantiles (cycles after first):", [int(np.percentile(w0[a:b] - w0[a], p)) for p in (0, 10, 25, 50, 75, 90, 100)])
print("domain 0 end-time quantiles:", [int(np.percentile(w1[a:b] - w0[a], p)) for p in (0, 10, 25, 50, 75, 90, 100)])
r0 = np.where(ok, st[:, :, 6], 1 << 62).min(axis=1) * 10.0; r1 = np.where(ok, st[:, :, 7], 0).max(axis=1) * 10.0   # ns
print("real time: kernel span %.1f us, workgroup median %.1f us, mean in flight %.1f" % ((r1.max() - r0.min()) / 1e3, np.median(r1 - r0) / 1e3, (r1 - r0).sum() / (r1.max() - r0.min())))
print("real start quantiles (us):", [round(float(np.percentile(r0 - r0.min(), p)) / 1e3, 1) for p in (0, 10, 25, 50, 75, 90, 100)])
