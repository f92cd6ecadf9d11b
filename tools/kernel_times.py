"""Per-kernel-family event times of config c3 with plan options set: python tools/kernel_times.py [opt=val ...]"""
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
for a in sys.argv[1:]:
    o, v = a.split("="); pl.set_option(int(o), int(v))
for _ in range(3): pl.run(True, True)
pl.set_option(1, 1)
acc = {}
N = 10
for _ in range(N):
    pl.run(True, True)
    for k in pl.kernel_stats():
        a = acc.setdefault(k["name"], [0.0, 0.0, 0]); a[0] += k["ms"]; a[1] += k["flops"]; a[2] += k["launches"]
for nm, (ms, fl, ln) in acc.items():
    if ms > 0: print("  %-78s %5.1f launches %8.3f ms %7.2f TFLOP/s" % (nm, ln / N, ms / N, fl / ms / 1e9 if ms else 0))
