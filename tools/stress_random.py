"""Randomized HIP-vs-oracle sweep (1-D and 2-D, every kernel family, odd block widths); run through gpurun."""
import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import pymra_amd.MRATools as mt
from pymra_amd import plan as P
from pymra_amd.topology import build_topology
from oracle.mra_levelwise import run_levelwise
bad = 0
t0 = time.time()
lo, hi = (int(v) for v in os.environ.get("SEEDS", "100:160").split(":"))
for seed in range(lo, hi):
    rng = np.random.RandomState(seed); np.random.seed(seed)
    d = 1 if seed % 5 == 0 else 2
    if d == 2:
        nx, ny = int(rng.randint(16, 90)), int(rng.randint(16, 90))
        kind = seed % 3
        if kind == 0: locs = mt.genLocations2d(Nx=nx, Ny=ny)
        elif kind == 1: locs = mt.genLocations2d(Nx=nx, Ny=ny) + rng.uniform(-0.3, 0.3, size=(nx * ny, 2)) / max(nx, ny)
        else: locs = rng.uniform(0, 1, size=(nx * ny, 2))
        J = 4
    else:
        n = int(rng.randint(200, 3000)); locs = mt.genLocations(n); J = int(rng.choice([2, 3, 4]))
    N = len(locs)
    r = int(rng.choice([3, 5, 8, 16, 20, 32, 40, 64])); M = int(rng.randint(1, 5))
    frac = float(rng.choice([0.05, 0.4, 0.9, 1.0]))
    y = rng.normal(size=(N, 1)); y_obs = np.where(rng.uniform(size=(N, 1)) < frac, y, np.nan)
    specs = [mt.KernelSpec(mt.KIND_EXP, 0.2), mt.KernelSpec(mt.KIND_MATERN32, 0.15, 1.3), mt.KernelSpec(mt.KIND_MATERN52, 0.2, 0.7), mt.KernelSpec(mt.KIND_GAUSSIAN, 0.05, 1.0), mt.KernelSpec(mt.KIND_MATERN32, 0.4, 1.0, 2.5)]
    spec = specs[seed % len(specs)]; R = float(rng.choice([1e-3, 5e-2, 0.5]))
    try:
        topo = build_topology(locs, r, M, J)
    except Exception as e:
        print(seed, "topology:", type(e).__name__, str(e)[:80]); continue
    try:
        ref = run_levelwise(topo, locs, spec, y_obs, R)
    except Exception as e:
        print(seed, "oracle:", type(e).__name__, str(e)[:80]); continue
    try:
        pl = P.HipPlan(topo, 0); pl.set_locs(locs); pl.set_obs(y_obs, R); pl.set_kernel(spec.kind, spec.l, spec.sig, spec.scale)
        pl.run(True, True); dd, u = pl.likelihood(); mean, var = pl.predict()
        # the same plan again (buffers prepared once must still be right), a likelihood-only pass in between
        pl.run(True, False); d1, u1 = pl.likelihood()
        pl.run(True, True); d2, u2 = pl.likelihood(); m2, v2 = pl.predict(); pl.close()
        again = max(abs(d1 + u1 - dd - u), abs(d2 + u2 - dd - u)) / max(1.0, abs(dd + u)) + float(np.max(np.abs(m2 - mean))) + float(np.max(np.abs(v2 - var)))
        if again > 1e-11:
            print(seed, "REPEATED PASSES DIFFER: %.3e" % again); bad += 1
    except Exception as e:
        print(seed, "HIP:", type(e).__name__, str(e)[:100]); bad += 1; continue
    lik = dd + u
    e1 = abs(lik - ref["lik"]) / max(1.0, abs(ref["lik"])); e2 = np.max(np.abs(mean - ref["mean"])); e3 = np.max(np.abs(np.sqrt(np.maximum(var, 0)) - ref["sd"]))
    flag = "" if (e1 < 1e-9 and e2 < 1e-7 and e3 < 1e-6) else "  <-- CHECK"
    if flag: bad += 1
    print("seed %d d=%d N=%d r=%d M=%d J=%d frac=%.2f kind=%d R=%g nodes=%d: lik %.2e mean %.2e sd %.2e%s" % (seed, d, N, r, M, J, frac, spec.kind, R, topo.n_nodes, e1, e2, e3, flag))
print("bad:", bad, "time %.1f s" % (time.time() - t0))
