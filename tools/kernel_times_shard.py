"""Per-kernel-family event times of rank 0 of an N-way shard of config c3 (one-GPU emulation, front exchange skipped):
python tools/kernel_times_shard.py <world> [opt=val ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pymra_amd import plan as P
from pymra_amd.sharding import shard_topology
from pymra_amd.topology import build_topology
import pymra_amd.MRATools as mt
world = int(sys.argv[1])
c = bench.CONFIGS["c3"]
locs, y_obs = bench.make_inputs(c)
topo = build_topology(locs, c["r"], c["M"], c["J"])
local, red = shard_topology(topo, world, 0)
pl = P.HipPlan(local, 0); pl.set_locs(locs); pl.set_obs(y_obs, c["R"]); pl.set_kernel(mt.KIND_MATERN32, c["l"], c["sig"], 1.0)
if red >= 0: pl.set_reduce_level(red)
for a in sys.argv[2:]:
    o, v = a.split("="); pl.set_option(int(o), int(v))
def step():
    if red < 0: pl.run(True, True)
    else: pl.run(True, True, split=True); pl.resume()
for _ in range(3): step()
pl.set_option(1, 1)
acc = {}
N = 10
for _ in range(N):
    step()
    for k in pl.kernel_stats():
        a = acc.setdefault(k["name"], [0.0, 0]); a[0] += k["ms"]; a[1] += k["launches"]
for nm, (ms, ln) in acc.items():
    if ms > 0: print("  %-78s %5.1f launches %8.3f ms" % (nm, ln / N, ms / N))
