"""A few resident passes of config c3 (optionally rank 0 of an N-way shard, emulated): the command tools/trace_kernels.sh traces."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pymra_amd import plan as P
from pymra_amd.sharding import shard_topology
from pymra_amd.topology import build_topology
import pymra_amd.MRATools as mt
world = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cfg = sys.argv[2] if len(sys.argv) > 2 else "c3"
c = bench.CONFIGS[cfg]
locs, y_obs = bench.make_inputs(c)
topo = build_topology(locs, c["r"], c["M"], c["J"])
local, red = shard_topology(topo, world, 0)
pl = P.HipPlan(local, 0); pl.set_locs(locs); pl.set_obs(y_obs, c["R"]); pl.set_kernel(mt.KIND_MATERN32, c["l"], c["sig"], 1.0)
if red >= 0: pl.set_reduce_level(red)
for _ in range(4):
    if red < 0: pl.run(True, True)
    else:
        pl.run(True, True, split=True); pl.resume()
print(sum(pl.likelihood()))
