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

# ---- inside the tree replay: the native call alone, then the export into NumPy arrays
import ctypes as C
lib = P.load_library()
for rep in range(3):
    locs, y_obs = bench.make_inputs(c)
    xy = np.ascontiguousarray(locs, dtype=np.float64)
    st = np.random.get_state()
    key = np.ascontiguousarray(st[1], dtype=np.uint32).copy(); pos = C.c_int32(int(st[2])); h = C.c_void_p()
    t0 = time.perf_counter()
    rc = lib.mra_tree_replay_2d(xy.ctypes.data_as(C.c_void_p), len(xy), int(c["r"]), int(c["M"]), key.ctypes.data_as(C.c_void_p), C.byref(pos), C.byref(h))
    t1 = time.perf_counter()
    lib.mra_tree_free(h)
    t2 = time.perf_counter()
    topo = build_topology(locs, c["r"], c["M"], c["J"])
    t3 = time.perf_counter()
    print("native replay call %.1f ms (rc %d), free %.1f ms, whole build_topology %.1f ms" % (1e3 * (t1 - t0), rc, 1e3 * (t2 - t1), 1e3 * (t3 - t2)))
