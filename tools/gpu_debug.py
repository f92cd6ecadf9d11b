"""Debug helper (GPU box): run a few cases through the HIP path and compare with the oracle,
including the whitened-basis buffer."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import warnings; warnings.filterwarnings("ignore")
import make_golden as mg
import pymra_amd.MRATools as mt
from pymra_amd.topology import build_topology, resolve_tree_shape
from pymra_amd.plan import HipPlan
from oracle.mra_levelwise import run_levelwise

def spec_of(c):
    return mt.KernelSpec(mt.KIND_EXP if c["kern"] == "exp" else mt.KIND_MATERN32, c["l"], c["sig"])

for name in sys.argv[1:]:
    c = mg.CASES[name]
    if c.get("data"):
        g = np.load(os.path.join(ROOT, "tests/golden/%s.npz" % name)); locs, y_obs = g["locs"], g["y_obs"]
    else:
        locs, y_obs, _ = mg.make_inputs(c)
    M, J = resolve_tree_shape(len(locs), locs.shape[1], c["r"], c["M"], c["J"])
    topo = build_topology(locs, c["r"], M, J)
    spec = spec_of(c)
    t0 = time.time()
    pl = HipPlan(topo, 0); pl.set_locs(locs); pl.set_obs(y_obs, c["R"]); pl.set_kernel(spec.kind, spec.l, spec.sig, spec.scale)
    pl.set_option(1, 1)
    pl.set_option(2, int(os.environ.get('MRA_FUSED', '1')))
    pl.run()   # warm-up (first-launch overheads)
    try:
        pl.run()
    except Exception as e:
        print(name, "RUN FAILED", e); continue
    t1 = time.time()
    d, u = pl.likelihood(); mean, var = pl.predict()
    print(name, "info", pl.info(), "timers", pl.timers(), "wall %.3f" % (t1 - t0))
    if topo.N <= 70000:
        ref = run_levelwise(topo, locs, spec, y_obs, c["R"], keep=True)
        W = pl.buffer(0).reshape(topo.P, -1)
        print("   d %.12g ref %.12g | u %.12g ref %.12g" % (d, ref["d"], u, ref["u"]))
        print("   lik rel %.3e  mean abs %.3e  sd rel %.3e" % (abs(d + u - ref["lik"]) / abs(ref["lik"]),
              np.max(np.abs(mean - ref["mean"])), np.max(np.abs(np.sqrt(var) - ref["sd"]) / np.maximum(ref["sd"], 1e-300))))
        dn = pl.buffer(1)
        print("   dnode max abs diff %.3e" % np.max(np.abs(dn - ref["dnode"])))
        print("   W (final) max abs diff %.3e (max |W| %.3e)" % (np.max(np.abs(W - ref["W"])), np.max(np.abs(ref["W"]))))
    gp = os.path.join(ROOT, "tests/golden/%s.npz" % name)
    if os.path.exists(gp):
        g = np.load(gp)
        gm, gs = g["mean"], g["sd"]
        if "sample_idx" in g: idx = g["sample_idx"]; mean_s, sd_s = mean[idx], np.sqrt(var)[idx]
        else: mean_s, sd_s = mean, np.sqrt(var)
        print("   vs REFERENCE golden: lik rel %.3e mean abs %.3e sd rel %.3e" % (abs(d + u - g["lik"]) / abs(g["lik"]),
              np.max(np.abs(mean_s - gm)), np.max(np.abs(sd_s - gs) / np.maximum(gs, 1e-300))))
    for k in pl.kernel_stats():
        if k["launches"]: print("      %-34s launches %3d  %9.3f ms  %8.2f GFLOP/s" % (k["name"], k["launches"], k["ms"], k["flops"] / max(k["ms"], 1e-9) / 1e6))
    pl.close()
