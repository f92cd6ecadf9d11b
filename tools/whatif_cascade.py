"""What-if timing of the prior row cascade at c3: which ingredient costs what (results are wrong with a switch on)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pymra_amd import plan as P
from pymra_amd.topology import build_topology
import pymra_amd.MRATools as mt
c = bench.CONFIGS["c3"]
locs, y_obs = bench.make_inputs(c)
topo = build_topology(locs, c["r"], c["M"], c["J"])
pl = P.HipPlan(topo, 0); pl.set_locs(locs); pl.set_obs(y_obs, c["R"]); pl.set_kernel(mt.KIND_MATERN32, c["l"], c["sig"], 1.0)
pl.set_option(1, 1)
for dbg, name in ((0, "baseline"), (1, "no Ut scatter"), (2, "no W stores"), (3, "no stores at all"), (4, "no kernel evaluation"), (7, "MFMA + LDS only")):
    pl.set_option(99, dbg)
    try:
        for _ in range(3): pl.run(True, False)
    except Exception as e:
        pass
    acc = 0.0
    for _ in range(5):
        try: pl.run(True, False)
        except Exception: pass
        acc += [k for k in pl.kernel_stats() if "row pass" in k["name"]][0]["ms"]
    print("%-24s row cascade %.3f ms" % (name, acc / 5), flush=True)
pl.set_option(99, 0)
