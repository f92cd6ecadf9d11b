"""Objective (-2 loglik = d + u) of the resident plan over a grid of range parameters, on the RFF GP sample bench.py's MLE block uses."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from pymra_amd import plan as P
from pymra_amd.topology import build_topology
import pymra_amd.MRATools as mt
cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
c = bench.CONFIGS[cfg]
locs, y_obs = bench.make_inputs(c)
topo = build_topology(locs, c["r"], c["M"], c["J"])
pl = P.HipPlan(topo, 0); pl.set_locs(locs)
f, rng = bench.gp_sample_rff(c, 0.2)
mask = np.isfinite(y_obs)
y_gp = np.where(mask, f + np.sqrt(c["R"]) * rng.normal(size=f.shape), np.nan)
print("field std %.3f, observed %d" % (float(f.std()), int(mask.sum())))
pl.set_obs(y_gp, c["R"])
grid = [float(a) for a in sys.argv[2].split(",")] if len(sys.argv) > 2 else (0.05, 0.1, 0.15, 0.18, 0.2, 0.22, 0.25, 0.3, 0.315, 0.4, 0.6)
for k in grid:
    pl.set_kernel(mt.KIND_MATERN32, k, c["sig"], 1.0); pl.run(True, False)
    d, u = pl.likelihood()
    print("kappa %.3f  d %.6e  u %.6e  d+u %.8e" % (k, d, u, d + u), flush=True)
