"""Randomized sweep over REGULAR 2-D trees (the fused kernels: row cascade, knot chain, leaf kernels, fronts, predictive cascade) and
their 2/4/8-way shards emulated on one GPU (k_chol_tiles, fused solve + update, split runs): HIP vs the CPU oracle, shards vs the single
plan.  Run through gpurun:  SEEDS=0:24 python tools/stress_regular.py"""
import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import _cases as K
import pymra_amd.MRATools as mt
from pymra_amd import plan as P
from pymra_amd.topology import build_topology
from oracle.mra_levelwise import run_levelwise
lo, hi = (int(v) for v in os.environ.get("SEEDS", "0:16").split(":"))
bad = 0
t0 = time.time()
for seed in range(lo, hi):
    rng = np.random.RandomState(1000 + seed); np.random.seed(1000 + seed)
    n = int(rng.choice([96, 128, 160, 192, 256, 320]))
    r = int(rng.choice([16, 32])); M = int(rng.choice([3, 4, 5]))
    frac = float(rng.choice([0.05, 0.4, 0.4, 0.9, 1.0]))
    locs = mt.genLocations2d(Nx=n, Ny=n)
    if seed % 3 == 1: locs = locs + rng.uniform(-0.3, 0.3, size=locs.shape) / n
    N = len(locs)
    y = rng.normal(size=(N, 1)); y_obs = np.where(rng.uniform(size=(N, 1)) < frac, y, np.nan)
    specs = [mt.KernelSpec(mt.KIND_EXP, 0.2), mt.KernelSpec(mt.KIND_MATERN32, 0.15, 1.3), mt.KernelSpec(mt.KIND_MATERN52, 0.2, 0.7), mt.KernelSpec(mt.KIND_MATERN32, 0.4, 1.0, 2.5)]
    spec = specs[seed % len(specs)]; R = float(rng.choice([1e-2, 5e-2, 0.5]))
    topo = build_topology(locs, r, M, 4)
    regular = topo.n_levels == M + 1 and len(set(np.asarray(topo.cw)[:M])) == 1
    tag = "seed %d n=%d r=%d M=%d frac=%.2f kind=%d R=%g nodes=%d %s" % (seed, n, r, M, frac, spec.kind, R, topo.n_nodes, "regular" if regular else "IRREGULAR")
    try:
        ref = run_levelwise(topo, locs, spec, y_obs, R)
        pl = P.HipPlan(topo, 0); pl.set_locs(locs); pl.set_obs(y_obs, R); pl.set_kernel(spec.kind, spec.l, spec.sig, spec.scale)
        pl.run(True, True); d, u = pl.likelihood(); mean, var = pl.predict()
        fused = any("k_predict_cascade" in k["name"] and k["launches"] for k in pl.kernel_stats())
        pl.run(True, False); d1, u1 = pl.likelihood()
        pl.run(True, True); d2, u2 = pl.likelihood(); m2, v2 = pl.predict(); pl.close()
    except Exception as e:
        print(tag, "FAILED:", type(e).__name__, str(e)[:120]); bad += 1; continue
    lik = d + u
    e1 = abs(lik - ref["lik"]) / max(1.0, abs(ref["lik"])); e2 = float(np.max(np.abs(mean - ref["mean"]))); e3 = float(np.max(np.abs(np.sqrt(np.maximum(var, 0)) - ref["sd"])))
    again = max(abs(d1 + u1 - lik), abs(d2 + u2 - lik)) / max(1.0, abs(lik)) + float(np.max(np.abs(m2 - mean))) + float(np.max(np.abs(v2 - var)))
    es = 0.0
    worlds = []
    for world in (2, 4, 8):
        try:
            liks, ms, vs, lev = K.emulate_world_on_one_gpu(P, topo, locs, y_obs, R, spec, world)
        except Exception as e:
            if "shard" in str(e).lower() or "level" in str(e).lower() or "world" in str(e).lower(): continue     # tree too shallow for this world size
            print(tag, "world %d FAILED:" % world, type(e).__name__, str(e)[:120]); bad += 1; continue
        worlds.append(world)
        es = max(es, max(abs(l - lik) for l in liks) / max(1.0, abs(lik)), float(np.max(np.abs(ms - mean))), float(np.max(np.abs(np.sqrt(np.maximum(vs, 0)) - np.sqrt(np.maximum(var, 0))))))
    ok = e1 < 1e-9 and e2 < 1e-7 and e3 < 1e-6 and again < 1e-11 and es < 1e-9
    if not ok: bad += 1
    print("%s fused=%s: lik %.1e mean %.1e sd %.1e  again %.1e  shards %s %.1e%s" % (tag, fused, e1, e2, e3, again, worlds, es, "" if ok else "  <-- CHECK"), flush=True)
print("bad:", bad, "time %.1f s" % (time.time() - t0))
