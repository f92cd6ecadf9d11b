"""A/B timing of a plan option on config c3 (or rank 0 of an N-way shard): interleaved rounds in one process.
usage: python tools/ab_option.py <option id> [world]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from pymra_amd import plan as P
from pymra_amd.sharding import shard_topology
from pymra_amd.topology import build_topology
import pymra_amd.MRATools as mt
opt = int(sys.argv[1]); world = int(sys.argv[2]) if len(sys.argv) > 2 else 1
c = bench.CONFIGS["c3"]
locs, y_obs = bench.make_inputs(c)
topo = build_topology(locs, c["r"], c["M"], c["J"])
local, red = shard_topology(topo, world, 0)
pl = P.HipPlan(local, 0); pl.set_locs(locs); pl.set_obs(y_obs, c["R"]); pl.set_kernel(mt.KIND_MATERN32, c["l"], c["sig"], 1.0)
if red >= 0: pl.set_reduce_level(red)
def step():
    if red < 0: pl.run(True, True)
    else: pl.run(True, True, split=True); pl.resume()
VALS = [int(a) for a in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 0]
res = {v: [] for v in VALS}
lik = {}
for rnd in range(6):
    for v in VALS:
        pl.set_option(opt, v)
        for _ in range(3): step()
        t0 = time.perf_counter()
        for _ in range(20): step()
        res[v].append((time.perf_counter() - t0) / 20 * 1e3)
        lik[v] = sum(pl.likelihood())
for v in VALS:
    print("option %d = %d: median %.3f ms  min %.3f ms  (%s)" % (opt, v, float(np.median(res[v])), min(res[v]), " ".join("%.3f" % x for x in res[v])), " d+u = %.10f" % lik[v])
