"""500 passes on one C3 plan (likelihood-only passes mixed in, a detour through other kernel parameters): buffers that are prepared once
and trusted afterwards must leave every later pass bit-identical to the first.  Run through gpurun."""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, bench
from pymra_amd import plan as P
from pymra_amd.topology import build_topology
import pymra_amd.MRATools as mt
c = bench.CONFIGS["c3"]
locs, y_obs = bench.make_inputs(c)
topo = build_topology(locs, c["r"], c["M"], c["J"])
pl = P.HipPlan(topo, 0); pl.set_locs(locs); pl.set_obs(y_obs, c["R"]); pl.set_kernel(mt.KIND_MATERN32, c["l"], c["sig"], 1.0)
pl.run(True, True); d0, u0 = pl.likelihood(); m0, v0 = pl.predict(); m0 = m0.copy(); v0 = v0.copy()
for i in range(500):
    pl.run(True, i % 3 != 0)
d, u = pl.likelihood()
pl.run(True, True); d1, u1 = pl.likelihood(); m1, v1 = pl.predict()
print("after 500 passes: lik diff %.3e, mean max diff %.3e, var max diff %.3e" % (abs(d1 + u1 - d0 - u0), np.max(np.abs(m1 - m0)), np.max(np.abs(v1 - v0))))
# kernel parameters changed and changed back
pl.set_kernel(mt.KIND_MATERN32, 0.7 * c["l"], c["sig"], 1.0); pl.run(True, True)
pl.set_kernel(mt.KIND_MATERN32, c["l"], c["sig"], 1.0); pl.run(True, True); d2, u2 = pl.likelihood(); m2, v2 = pl.predict()
print("after a detour through other parameters: lik diff %.3e, mean %.3e, var %.3e" % (abs(d2 + u2 - d0 - u0), np.max(np.abs(m2 - m0)), np.max(np.abs(v2 - v0))))
