#!/usr/bin/env python3
"""Maximum-likelihood estimation of the kernel range with Nelder-Mead, as in pyMRA's README
(README.md:96-104) and pyMRA/tests/test-param-est.py:81-123 - but the tree, the locations and the
observations stay on the GPU between objective calls (pyMRA builds a new MRATree per call, and in
2-D re-draws the knots every time, which makes its objective noisy).

    python examples/mle_nelder_mead.py [grid_side] [M] [r0]
"""
import sys
import time

import numpy as np
import scipy.optimize as opt

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import pymra_amd.MRATools as mt
from pymra_amd import MRATree


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    M = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    r0 = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    np.random.seed(11)
    locs = mt.genLocations2d(Nx=n, Ny=n)
    true_kappa, sig, me = 0.3, 1.0, 1e-2
    # random-Fourier-feature sample of a Matern32 field (dense Cholesky is impossible at this size)
    nf = 512
    # Matern-3/2 spectral measure in 2-D = bivariate Student-t with 3 degrees of freedom, scale 1/kappa
    w = np.random.normal(size=(nf, 2)) / np.sqrt(np.random.chisquare(3, size=(nf, 1)) / 3.0) / true_kappa
    ph = np.random.uniform(0, 2 * np.pi, nf)
    x = np.sqrt(2.0 * sig / nf) * np.cos(locs @ w.T + ph).sum(axis=1, keepdims=True)
    y = x + np.sqrt(me) * np.random.normal(size=x.shape)
    oi = np.sort(np.random.choice(n * n, int(0.4 * n * n), replace=False))
    y_obs = np.full_like(y, np.nan)
    y_obs[oi] = y[oi]

    t0 = time.time()
    tree = MRATree(locs, r0, lambda a, b: mt.Matern32(a, b, l=true_kappa, sig=sig), y_obs, me, M=M, J=4, want_predict=False)
    print("tree + first likelihood: %.3f s, lik(kappa=%.2f) = %.6f" % (time.time() - t0, true_kappa, tree.getLikelihood()[0, 0]))
    calls = [0]

    def objective(p):
        calls[0] += 1
        kappa = float(np.abs(p[0]))
        return float(tree.reevaluate(lambda a, b: mt.Matern32(a, b, l=kappa, sig=sig))[0, 0])

    t0 = time.time()
    res = opt.minimize(objective, [0.5], method="nelder-mead", options={"xatol": 1e-3, "disp": False})
    dt = time.time() - t0
    print("Nelder-Mead: kappa_hat = %.4f after %d likelihood evaluations in %.3f s (%.1f ms each, %d nodes -> %.0f nodes/s)"
          % (res.x[0], calls[0], dt, 1e3 * dt / calls[0], tree.topology.n_nodes, tree.topology.n_nodes * calls[0] / dt))


if __name__ == "__main__":
    main()
