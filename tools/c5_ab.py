"""C5 geometry: per-kernel times with a plan option set (python tools/c5_ab.py [opt=val ...])."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from pymra_amd import plan as P
from pymra_amd.topology import build_topology
import pymra_amd.MRATools as mt
c = bench.CONFIGS["c5"]
locs, y_obs = bench.make_inputs(c)
topo = build_topology(locs, c["r"], c["M"], c["J"])
pl = P.HipPlan(topo, 0); pl.set_locs(locs); pl.set_obs(y_obs, c["R"]); pl.set_kernel(mt.KIND_MATERN32, c["l"], c["sig"], 1.0)
for variant in [()] + [tuple(a.split("=")) for a in sys.argv[1:]]:
    if variant: pl.set_option(int(variant[0]), int(variant[1]))
    pl.set_option(1, 0)
    for _ in range(2): pl.run(True, True)
    t0 = time.perf_counter()
    for _ in range(3): pl.run(True, True)
    dt = (time.perf_counter() - t0) / 3 * 1e3
    pl.set_option(1, 1); pl.run(True, True)
    print("options %s: %.1f ms/pass  lik %.10e" % (variant, dt, sum(pl.likelihood())))
    for k in pl.kernel_stats():
        if k["launches"]: print("   %-74s %3d launches %9.3f ms %7.2f TFLOP/s" % (k["name"], k["launches"], k["ms"], k["flops"] / max(k["ms"], 1e-9) / 1e9))
