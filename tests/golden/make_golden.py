#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by RUNNING THE REFERENCE (pyMRA) here.

Run only in the build container, where /root/reference exists:

    python tests/golden/make_golden.py [case ...]        # default: every case but c3
    python tests/golden/make_golden.py c3                # ~30 min, 2.4 GB
    python tests/golden/make_golden.py --nodes [case ...]  # per-node blocks of the tiny cases -> <case>_nodes.npz

The fixtures are data only (inputs or their recipe, the tree the reference built, and the
reference's outputs); no reference code is stored.  The reference imports the third-party
package ``numpy_indexed`` which is not installed in this image; the one function it uses,
``contains(this, that)`` (row-membership mask), is provided by the small stand-in written to
a temporary directory below.  It consumes no random numbers.

Inputs are generated with THIS repo's generators (pymra_amd.MRATools) following the recipe
of SURVEY.md section 8(d); the very same arrays are handed to the reference.
The tree is captured by wrapping ``Node.calculatePrior`` (pyMRA/MRANode.py:378) - the
children are discarded after construction (MRANode.py:108-111), so it cannot be read later.
"""
import json
import os
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"

_SHIM = '''
import numpy as np
def contains(this, that, axis=0):
    this = np.asarray(this); that = np.asarray(that)
    if this.ndim == 1: this = this.reshape(-1, 1)
    if that.ndim == 1: that = that.reshape(-1, 1)
    if len(this) == 0: return np.zeros(len(that), dtype=bool)
    s = set(map(bytes, np.ascontiguousarray(this, dtype=np.float64)))
    return np.array([bytes(r) in s for r in np.ascontiguousarray(that, dtype=np.float64)], dtype=bool)
'''


def _import_reference():
    tmp = tempfile.mkdtemp(prefix="npi_shim_")
    os.makedirs(os.path.join(tmp, "numpy_indexed"))
    with open(os.path.join(tmp, "numpy_indexed", "__init__.py"), "w") as f:
        f.write(_SHIM)
    os.environ.setdefault("MPLBACKEND", "Agg")
    sys.path.insert(0, tmp)
    sys.path.insert(0, REF)
    import warnings
    warnings.filterwarnings("ignore")
    import logging
    logging.disable(logging.CRITICAL)
    from pyMRA.MRATree import MRATree
    from pyMRA import MRANode
    import pyMRA.MRATools as rmt
    return MRATree, MRANode, rmt


sys.path.insert(0, REPO)
import pymra_amd.MRATools as mt        # noqa: E402  (this repo's generators)


# ------------------------------------------------------------------------------------------
#  input recipes
# ------------------------------------------------------------------------------------------
def grid_case(n, seed, frac=0.4):
    """seed; genLocations2d; y ~ N(0,1); 40 % observed; RNG left running."""
    np.random.seed(seed)
    locs = mt.genLocations2d(Nx=n, Ny=n)
    y = np.random.normal(size=(n * n, 1))
    oi = np.sort(np.random.choice(n * n, int(frac * n * n), replace=False))
    y_obs = np.empty((n * n, 1))
    y_obs[:] = np.nan
    y_obs[oi] = y[oi]
    return locs, y_obs


def line_case(n, seed, kernel, l, sig, R, frac=0.4):
    """README recipe (README.md:62-93): GP sample by dense Cholesky + noise, frac observed."""
    import scipy.linalg as lng
    np.random.seed(seed)
    locs = mt.genLocations(n)
    Sig = sig * np.asarray(kernel(locs))
    SigC = lng.cholesky(Sig)
    x = SigC.T @ np.random.normal(size=(n, 1))
    y = x + np.sqrt(R) * np.random.normal(size=(n, 1))
    oi = np.sort(np.random.choice(n, int(n * frac), replace=False))
    y_obs = np.empty((n, 1))
    y_obs[:] = np.nan
    y_obs[oi] = y[oi]
    return locs, y_obs


CASES = {
    # name: dict(kind, ...)
    "kat1":  dict(dim=1, n=20,  seed=5,  kern="exp", l=1.0, sig=1.0, R=1e-2, r=20, M=0, J=2, frac=0.5),
    "kat2":  dict(dim=1, n=3,   seed=6,  kern="exp", l=1.0, sig=1.0, R=1e-2, r=1,  M=1, J=2, frac=0.67),
    "kat3":  dict(dim=1, n=100, seed=11, kern="exp", l=0.3, sig=1.0, R=1e-2, r=2,  M=3, J=3, frac=0.4),
    "c1":    dict(dim=1, n=100, seed=11, kern="m32", l=0.3, sig=1.0, R=1e-2, r=2,  M=3, J=3, frac=0.4),
    "t1000": dict(dim=1, n=1000, seed=3, kern="exp", l=0.3, sig=1.0, R=1e-2, r=2,  M=4, J=3, frac=0.4),
    "t201":  dict(dim=1, n=201, seed=4, kern="exp", l=0.3, sig=1.0, R=1e-2, r=2,  M=3, J=3, frac=0.4),
    "kat4":  dict(dim=2, data="small", kern="exp", l=0.1, sig=1.0, R=1e-2, r=100, M=0, J=4),
    "u3":    dict(dim=2, n=10,  seed=12, kern="m32", l=0.5, sig=1.0, R=1e-6, r=2,  M=2, J=3, frac=0.7, gp=True),
    "g32":   dict(dim=2, n=32,  seed=7,  kern="exp", l=0.3, sig=1.0, R=1e-2, r=16, M=2, J=4),
    "g64":   dict(dim=2, n=64,  seed=11, kern="exp", l=0.3, sig=1.0, R=1e-2, r=16, M=3, J=4),
    "g64m":  dict(dim=2, n=64,  seed=11, kern="m32", l=0.3, sig=1.0, R=1e-2, r=16, M=3, J=4),
    "g128m": dict(dim=2, n=128, seed=11, kern="m32", l=0.3, sig=1.0, R=1e-2, r=16, M=3, J=4),
    "c2":    dict(dim=2, n=256, seed=11, kern="exp", l=0.3, sig=1.0, R=1e-2, r=16, M=4, J=4),
    "c3":    dict(dim=2, n=1024, seed=11, kern="m32", l=0.3, sig=1.0, R=1e-2, r=32, M=6, J=4),
}
DEFAULT = [k for k in CASES if k != "c3"]


def make_inputs(c):
    kern = {"exp": lambda a, b=np.array([]): mt.ExpCovFun(a, b, l=c["l"]),
            "m32": lambda a, b=np.array([]): mt.Matern32(a, b, l=c["l"], sig=c["sig"])}[c["kern"]]
    if c.get("data") == "small":
        locs = np.load(os.path.join(REF, "pyMRA/data/small/locs.npy"))
        y_obs = np.load(os.path.join(REF, "pyMRA/data/small/y_obs.npy")).reshape(-1, 1)
    elif c["dim"] == 1:
        locs, y_obs = line_case(c["n"], c["seed"], kern, c["l"], c["sig"], c["R"], c["frac"])
    elif c.get("gp"):
        import scipy.linalg as lng
        np.random.seed(c["seed"])
        n = c["n"]
        locs = mt.genLocations2d(Nx=n, Ny=n)
        SigC = lng.cholesky(c["sig"] * np.asarray(kern(locs)))
        x = SigC.T @ np.random.normal(size=(n * n, 1))
        y = x + np.sqrt(c["R"]) * np.random.normal(size=(n * n, 1))
        oi = np.sort(np.random.choice(n * n, int(n * n * c["frac"]), replace=False))
        y_obs = np.empty((n * n, 1))
        y_obs[:] = np.nan
        y_obs[oi] = y[oi]
    else:
        locs, y_obs = grid_case(c["n"], c["seed"], c.get("frac", 0.4))
    return locs, y_obs, kern


def run_case(name, MRATree, MRANode, rmt):
    c = CASES[name]
    locs, y_obs, _ = make_inputs(c)
    N = len(locs)
    rcov = {"exp": lambda a, b: rmt.ExpCovFun(a, b, l=c["l"]),
            "m32": lambda a, b: rmt.Matern32(a, b, l=c["l"], sig=c["sig"])}[c["kern"]]
    # capture the tree
    key = {np.ascontiguousarray(locs[i]).tobytes(): i for i in range(N)}
    rec = []
    orig = MRANode.Node.calculatePrior

    def spy(self, cov):
        rows = np.array([key[np.ascontiguousarray(p).tobytes()] for p in self.locs], dtype=np.int64)
        rec.append((self.ID, rows, rows[np.asarray(self.kInds, dtype=np.int64)], bool(self.leaf)))
        return orig(self, cov)
    MRANode.Node.calculatePrior = spy
    rng_after_inputs = np.random.get_state()
    t0 = time.time()
    try:
        tree = MRATree(locs, c["r"], rcov, y_obs, c["R"], M=c["M"], J=c["J"])
    finally:
        MRANode.Node.calculatePrior = orig
    wall = time.time() - t0
    lik = float(np.asarray(tree.getLikelihood()).reshape(-1)[0])
    mean, sd = tree.predict()
    mean = np.asarray(mean).reshape(-1)
    sd = np.asarray(sd).reshape(-1)
    idents = np.array([t[0] for t in rec])
    nrows = np.array([len(t[1]) for t in rec], dtype=np.int64)
    knot_ptr = np.concatenate([[0], np.cumsum([len(t[2]) for t in rec])]).astype(np.int64)
    knots = np.concatenate([t[2] for t in rec]).astype(np.int32)
    leaf = np.array([t[3] for t in rec])
    rowsum = np.array([int(t[1].sum()) for t in rec], dtype=np.int64)     # cheap row-set fingerprint
    out = dict(params=json.dumps(c), M_eff=tree.M, J_eff=tree.J, lik=lik,
               d=float(np.asarray(tree.root.d).reshape(-1)[0]), u=float(np.asarray(tree.root.u).reshape(-1)[0]),
               node_ident=idents, node_nrows=nrows, node_rowsum=rowsum, node_leaf=leaf,
               knot_ptr=knot_ptr, knots=knots, ref_wall_s=wall,
               y_checksum=float(np.nansum(y_obs)), n_obs=int(np.isfinite(y_obs).sum()))
    if N <= 70000:
        out.update(mean=mean, sd=sd)
    else:
        idx = np.arange(0, N, 257)
        out.update(sample_idx=idx, mean=mean[idx], sd=sd[idx],
                   mean_sum=float(mean.sum()), sd_sum=float(sd.sum()),
                   mean_sq=float((mean ** 2).sum()), sd_sq=float((sd ** 2).sum()))
    if N <= 1100 or c.get("data"):
        out.update(locs=locs, y_obs=y_obs)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("%-6s N=%-8d nodes=%-5d lik=%.10f wall=%.1fs" % (name, N, len(rec), lik, wall), flush=True)


NODE_CASES = ["kat2", "kat3", "c1", "u3", "g32"]


def run_case_nodes(name, MRATree, MRANode, rmt):
    """Per-node blocks of a tiny case -> <name>_nodes.npz: for every node (keyed by its position in the
    reference's construction order) what the reference holds right after calculatePosterior
    (pyMRA/MRANode.py:403-523): B, kInv, kC, kTil, kTilC, A[res][res], omg[res] (the children's omgTil summed, or
    the leaf's own B^T H^T obs / R), BTil[res], d, u, mean, var.  The main fixture <name>.npz is not touched."""
    c = CASES[name]
    locs, y_obs, _ = make_inputs(c)
    rcov = {"exp": lambda a, b: rmt.ExpCovFun(a, b, l=c["l"]),
            "m32": lambda a, b: rmt.Matern32(a, b, l=c["l"], sig=c["sig"])}[c["kern"]]
    out = {}
    order = []
    orig_prior = MRANode.Node.calculatePrior
    orig_post = MRANode.Node.calculatePosterior

    def spy_prior(self, cov):
        order.append(self.ID)
        return orig_prior(self, cov)

    def spy_post(self, obs, R):
        res = orig_post(self, obs, R)
        t = order.index(self.ID)
        a = lambda x: np.array(x, dtype=np.float64)
        if self.leaf:
            oi = np.array(np.isfinite(obs)).ravel()
            omg = a(self._getB_lk(self.res))[oi].T @ a(obs)[oi] / R
        else:
            omg = sum(a(ch.omgTil[self.res]) for ch in self.children)
        blk = dict(B=a(self.B), kInv=a(self.kInv), kC=a(self.kC), kTil=a(self.kTil), kTilC=a(self.kTilC),
                   Amm=a(self.A[self.res][self.res]), omg=a(omg).ravel(), BTil=a(self.BTil[self.res]),
                   d=a(self.d).ravel()[:1], u=a(self.u).ravel()[:1], mean=a(self.mean).ravel(), var=a(self.var).ravel())
        for k, v in blk.items():
            out["n%d_%s" % (t, k)] = v
        return res
    MRANode.Node.calculatePrior = spy_prior
    MRANode.Node.calculatePosterior = spy_post
    try:
        MRATree(locs, c["r"], rcov, y_obs, c["R"], M=c["M"], J=c["J"])
    finally:
        MRANode.Node.calculatePrior = orig_prior
        MRANode.Node.calculatePosterior = orig_post
    out["node_ident"] = np.array(order)
    out["params"] = json.dumps(c)
    np.savez_compressed(os.path.join(HERE, name + "_nodes.npz"), **out)
    print("%-6s per-node blocks of %d nodes" % (name, len(order)), flush=True)


if __name__ == "__main__":
    args = sys.argv[1:]
    MRATree, MRANode, rmt = _import_reference()
    if args and args[0] == "--nodes":
        for nm in (args[1:] or NODE_CASES):
            run_case_nodes(nm, MRATree, MRANode, rmt)
    else:
        for nm in (args or DEFAULT):
            run_case(nm, MRATree, MRANode, rmt)
