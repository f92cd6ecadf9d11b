"""Where the end-to-end MRATree(...) wall-clock goes (host tree replay, plan creation, uploads, device pass, download)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import pymra_amd.MRATools as mt
from pymra_amd import plan as P
from pymra_amd.topology import build_topology
c = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "c3"]
for rep in range(3):
    locs, y_obs = bench.make_inputs(c)
    t = [time.perf_counter()]
    topo = build_topology(locs, c["r"], c["M"], c["J"]); t.append(time.perf_counter())
    pl = P.HipPlan(topo, 0); t.append(time.perf_counter())
    pl.set_locs(locs); t.append(time.perf_counter())
    pl.set_obs(y_obs, c["R"]); t.append(time.perf_counter())
    pl.set_kernel(mt.KIND_MATERN32, c["l"], c["sig"], 1.0); t.append(time.perf_counter())
    pl.run(True, True); t.append(time.perf_counter())
    d, u = pl.likelihood(); mean, var = pl.predict(); t.append(time.perf_counter())
    pl.close(); t.append(time.perf_counter())
    names = ["tree replay", "plan create", "set_locs", "set_obs", "set_kernel", "run", "results D2H+unpermute", "destroy"]
    print("rep %d total %.3f s: " % (rep, t[-1] - t[0]) + ", ".join("%s %.1f ms" % (n, 1e3 * (b - a)) for n, a, b in zip(names, t[:-1], t[1:])))
