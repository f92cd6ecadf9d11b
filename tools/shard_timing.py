"""Timing-only emulation of one rank of an N-way sharded run on a single GPU: rank 0's local tree is
run with the split entry points, re-importing its own front buffer (numbers are NOT the full
likelihood - only the time per step and the per-kernel breakdown are meaningful).

    python tools/shard_timing.py [--config c3|c5] [--likelihood-only] [world ...]          (default: c3, worlds 1 2 4 8)"""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from pymra_amd import plan as P
from pymra_amd.sharding import shard_topology
from pymra_amd.topology import build_topology
import pymra_amd.MRATools as mt

argv = sys.argv[1:]
cfg = "c3"
if "--config" in argv:
    k = argv.index("--config"); cfg = argv[k + 1]; del argv[k:k + 2]
PRED = "--likelihood-only" not in argv
argv = [a for a in argv if a != "--likelihood-only"]
c = bench.CONFIGS[cfg]
locs, y_obs = bench.make_inputs(c)
topo = build_topology(locs, c["r"], c["M"], c["J"])
print("config %s (%s), %d nodes" % (cfg, "likelihood+predict" if PRED else "likelihood only", topo.n_nodes))
for world in [int(a) for a in argv] or [1, 2, 4, 8]:
    local, red = shard_topology(topo, world, 0)
    pl = P.HipPlan(local, 0); pl.set_locs(locs); pl.set_obs(y_obs, c["R"]); pl.set_kernel(mt.KIND_MATERN32, c["l"], c["sig"], 1.0)
    if red >= 0: pl.set_reduce_level(red)
    def step():
        if red < 0: pl.run(True, PRED); return
        pl.run(True, PRED, split=True); buf = pl.reduce_export(); pl.reduce_import(buf); pl.resume()
    for _ in range(3): step()
    pl.set_option(1, 1)
    t0 = time.perf_counter(); n = 10 if cfg != "c5" else 3
    acc = None
    for _ in range(n):
        step(); ks = pl.kernel_stats()
        if acc is None: acc = ks
        else:
            for a, b in zip(acc, ks): a["ms"] += b["ms"]; a["launches"] += b["launches"]
    dt = (time.perf_counter() - t0) / n * 1e3
    print("world %d: rank-0 local P=%d nodes=%d  %.3f ms/step (incl. host export/import)  reduce buf %.1f KB" % (world, local.P, local.n_nodes, dt, (len(pl.reduce_export()) * 8 / 1024) if False else 0))
    for k in acc:
        if k["launches"]: print("      %-58s %5.1f launches %7.3f ms" % (k["name"], k["launches"] / n, k["ms"] / n))
    # device wall-clock per step without per-kernel events and without the host export/import hop
    pl.set_option(1, 0)
    def step2():
        if red < 0: pl.run(True, PRED); return
        pl.run(True, PRED, split=True); pl.resume()
    for _ in range(3): step2()
    n2 = 50 if cfg != "c5" else 5; t0 = time.perf_counter()
    for _ in range(n2): step2()
    print("      plain: %.3f ms/step wall (no kernel events, no host hop); device total %.3f ms" % ((time.perf_counter() - t0) / n2 * 1e3, pl.timers()["total_ms"]))
    pl.close()
