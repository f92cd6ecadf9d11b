"""README example 1 (pyMRA/README.md:19-46) on sub-grids of the packaged 100x100 sample: M = 0 means ONE dense leaf
(exact kriging on every point), the worst case for the leaf kernels.  Prints the wall-clock per size."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import pymra_amd.DataLoader as dl
import pymra_amd.MRATools as mt
from pymra_amd import MRATree
y, locs, y_obs = dl.load_data("large", True)
for n in [int(a) for a in sys.argv[1:]] or [30, 50]:
    sel = np.array([j * 100 + i for j in range(n) for i in range(n)])
    t0 = time.perf_counter()
    tree = MRATree(locs[sel], 4, lambda a, b: mt.ExpCovFun(a, b, l=2), y_obs[sel].reshape(-1, 1), 1e-4, 0)
    yP, sdP = tree.predict()
    print("README example 1 on %dx%d points (M=%d): %.3f s, lik %.6f, device %.1f ms" % (n, n, tree.M, time.perf_counter() - t0, float(tree.getLikelihood()[0, 0]), tree.plan.timers()["total_ms"]), flush=True)
