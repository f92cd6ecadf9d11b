"""BASELINE config 5 geometry on ONE GPU: 2-D 2048x2048 grid, M=8, J=4, r0=64, Matern32 (the general
level-by-level path: 4 column tiles x 8 levels exceeds the fused cascades' register budget)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pymra_amd.MRATools as mt
from pymra_amd import plan as P
from pymra_amd.topology import build_topology

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
M = int(sys.argv[2]) if len(sys.argv) > 2 else 8
r = int(sys.argv[3]) if len(sys.argv) > 3 else 64
np.random.seed(11)
t0 = time.time()
locs = mt.genLocations2d(Nx=n, Ny=n)
y = np.random.normal(size=(n * n, 1))
oi = np.sort(np.random.choice(n * n, int(0.4 * n * n), replace=False))
y_obs = np.full((n * n, 1), np.nan); y_obs[oi] = y[oi]
t1 = time.time()
topo = build_topology(locs, r, M, 4)
t2 = time.time()
print("inputs %.2f s, tree %.2f s: %d nodes, P=%d, levels %s" % (t1 - t0, t2 - t1, topo.n_nodes, topo.P, list(np.diff(topo.level_ptr))), flush=True)
pl = P.HipPlan(topo, 0); pl.set_locs(locs); pl.set_obs(y_obs, 1e-2); pl.set_kernel(mt.KIND_MATERN32, 0.3, 1.0, 1.0)
t3 = time.time()
info = pl.info()
print("plan %.2f s; W %.1f GB, panels %.1f GB, Gt %.1f GB" % (t3 - t2, info["bytes_W"] / 1e9, info["bytes_panels"] / 1e9, info["bytes_Gt"] / 1e9), flush=True)
for rep in range(3):
    ta = time.time(); pl.run(True, True); tb = time.time()
    d, u = pl.likelihood()
    print("run %d: lik+predict %.1f ms  lik=%.10e  %s" % (rep, 1e3 * (tb - ta), d + u, {k: round(v, 1) for k, v in pl.timers().items()}), flush=True)
ta = time.time(); pl.run(True, False); tb = time.time()
d2, u2 = pl.likelihood()
print("likelihood-only %.1f ms, rel diff to full %.2e ; nodes/s (full) %.0f" % (1e3 * (tb - ta), abs(d2 + u2 - d - u) / abs(d + u), topo.n_nodes / (1e-3 * pl.timers()["total_ms"])), flush=True)
pl.set_option(1, 1)
for pred in (True, False):
    pl.run(True, pred)
    print("per-kernel (likelihood%s):" % ("+predict" if pred else " only"))
    for k in pl.kernel_stats():
        if k["launches"]:
            print("   %-60s %3d launches %9.3f ms %7.2f TFLOP/s" % (k["name"], k["launches"], k["ms"], k["flops"] / max(k["ms"], 1e-9) / 1e9))
    print("   phases", {k: round(v, 2) for k, v in pl.timers().items()})
