"""ORACLE (test infrastructure, not product code) - faithful CPU restatement.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the product path (pymra_amd) never does.

What it is: a NumPy/SciPy restatement of the reference's per-node prior/posterior
inference, node by node, with the SAME dense primitives the reference calls, so that
both its results and its CPU cost profile stand in for the reference on machines where
the reference cannot be present:

* prior  B = v_m(S,Q), kInv = B[knots], k = inv(kInv), kC = chol(k)   pyMRA/MRANode.py:378-391
* conditional covariance passed to the children                        pyMRA/MRANode.py:73-80
* leaf A/omega through a dense (1/R) H^T H                             pyMRA/MRANode.py:415-430
* non-leaf A/omega as sums over children                               pyMRA/MRANode.py:434-440
* kTil = inv(kInv + A_mm), kTilInv = inv(kTil) (scipy.linalg.inv)     pyMRA/MRANode.py:444-445
* u, d (slogdet at leaves, log(det) above)                             pyMRA/MRANode.py:450-468
* ATil, omgTil                                                         pyMRA/MRANode.py:474-480
* BTil                                                                 pyMRA/MRANode.py:486-495
* eigh square root with sign-flipped negative eigenvalues              pyMRA/MRANode.py:504-507
* mean, var and the children's accumulation                            pyMRA/MRANode.py:510-520
* per-node gc.collect()                                                pyMRA/MRANode.py:108-111
* lik = d + u, sd = sqrt(var)                                          pyMRA/MRATree.py:82-94

Mechanism differences (rounding-level only): the tree comes from pymra_amd.topology
(index bookkeeping instead of coordinate matching), rows are addressed in the padded
leaf order of that topology (phantom padding rows behave as extra unobserved points and
are never reported), and the conditional covariance v_m(S,Q) is evaluated from the
ancestors' stored B blocks,
    v_m(S,Q) = C(S,Q) - sum_{k<m} B_k[S] K_k B_k[Q]^T,
the closed form of the reference's closure chain (MRANode.py:80).  One deliberate repair:
BTil[k] is allocated with ancestor k's own rank (the reference allocates every block with
the node's own rank, MRANode.py:492, which only works when all ranks are equal).

Pinned against the imported reference by tests/golden (tests/golden/make_golden.py) and
against exact kriging by the identities of pyMRA/tests/unit-tests.py:22-71, 75-130.
"""
from __future__ import annotations

import gc
import sys
import time
from typing import Optional

import numpy as np
import scipy.linalg as sla
from numpy.linalg import slogdet


class _NodeState:
    __slots__ = ("B", "kInv", "K", "kC", "A", "omg", "kTil", "ATil", "omgTil", "BTil", "d", "u",
                 "mean", "var")


def _as_cov(cov, coords):
    """cov(rows_a, rows_b) over caller row indices, from a callable or a dense matrix."""
    if callable(cov):
        return lambda ra, rb: np.asarray(cov(coords[ra], coords[rb]), dtype=np.float64)
    dense = np.asarray(cov, dtype=np.float64)
    return lambda ra, rb: dense[np.ix_(ra, rb)]


class _Recursion:
    """Per-node work.  ``pruned`` optionally supplies ancestors that exist only on a subset of
    rows (bounded CPU-baseline sample); everything else is the plain recursion."""

    def __init__(self, topo, locs, cov, obs, R, do_gc, pruned=None):
        coords = np.asarray(locs, dtype=np.float64)
        if coords.ndim == 1:
            coords = coords.reshape(-1, 1)
        self.t = topo
        self.covf = _as_cov(cov, coords)
        self.y = np.asarray(obs, dtype=np.float64).reshape(-1)
        self.R = float(R)
        self.do_gc = do_gc
        self.st = {}
        self.pruned = pruned or {}
        self.tm = dict(prior=0.0, posterior=0.0, gc=0.0)
        self.count = 0
        self.keep = None                      # dict: finished nodes are parked here instead of being deleted (tests)

    # -- topology helpers ----------------------------------------------------------------
    def rows(self, i):
        return np.arange(self.t.node_row0[i], self.t.node_row1[i])

    def knots(self, i):
        return self.t.knot_rows[self.t.knot_ptr[i]:self.t.knot_ptr[i + 1]]

    def kids(self, i):
        return [int(c) for c in self.t.child_list[self.t.child_ptr[i]:self.t.child_ptr[i + 1]]]

    def lineage(self, i):
        chain = []
        p = int(self.t.node_parent[i])
        while p >= 0:
            chain.append(p)
            p = int(self.t.node_parent[p])
        return chain[::-1]

    def B_rows(self, a, rows):
        if a in self.pruned:
            pr = self.pruned[a]
            return pr["B"][pr["pos"](rows)]
        return self.st[a].B[rows - self.t.node_row0[a]]

    def K_of(self, a):
        return self.pruned[a]["K"] if a in self.pruned else self.st[a].K

    def rank_of(self, a):
        return self.K_of(a).shape[0]

    # -- prior (MRANode.py:378-391 with the conditional covariance of :73-80) --------------
    def prior(self, i):
        t0 = time.perf_counter()
        s = _NodeState()
        rp, kq = self.rows(i), self.knots(i)
        src = self.t.src
        B = self.covf(src[rp], src[kq])
        for a in self.lineage(i):
            B = B - self.B_rows(a, rp) @ self.K_of(a) @ self.B_rows(a, kq).T
        s.B = B
        s.kInv = B[kq - self.t.node_row0[i]]
        s.K = np.linalg.inv(s.kInv)
        s.kC = np.linalg.cholesky(s.K)
        self.st[i] = s
        self.tm["prior"] += time.perf_counter() - t0

    # -- posterior (MRANode.py:403-523) -------------------------------------------------------
    def posterior(self, i):
        t0 = time.perf_counter()
        t, st, R = self.t, self.st, self.R
        s = st[i]
        m = int(t.node_level[i])
        rp = self.rows(i)
        chain = self.lineage(i) + [i]
        kids = self.kids(i)
        leaf = bool(t.node_leaf[i])
        if leaf:
            yv = np.where(t.perm[rp] >= 0, self.y[t.src[rp]], np.nan)
            oi = np.isfinite(yv)
            H = np.eye(len(rp))[oi, :]
            HRinvH = (1.0 / R) * (H.T @ H)
            HRinvObs = H.T @ ((1.0 / R) * yv[oi])
            Bs = [self.B_rows(a, rp) for a in chain]
            s.A = [[Bs[k].T @ HRinvH @ Bs[l] for l in range(m + 1)] for k in range(m + 1)]
            s.omg = [Bs[k].T @ HRinvObs for k in range(m + 1)]
        elif not kids:
            # a non-leaf node without children: only on PRUNED trees (a rank's local tree keeps every upper node but only its
            # own subtrees, pymra_amd.sharding); the sums over children are empty
            rk = [self.rank_of(a) for a in chain]
            s.A = [[np.zeros((rk[k], rk[l])) for l in range(m + 1)] for k in range(m + 1)]
            s.omg = [np.zeros(rk[k]) for k in range(m + 1)]
        else:
            s.A = [[sum(st[c].ATil[k][l] for c in kids) for l in range(m + 1)] for k in range(m + 1)]
            s.omg = [sum(st[c].omgTil[k] for c in kids) for k in range(m + 1)]
        s.kTil = sla.inv(s.kInv + s.A[m][m])
        kTilInv = sla.inv(s.kTil)
        if leaf:
            s.u = -s.omg[m] @ s.kTil @ s.omg[m] + yv[oi] @ yv[oi] / R
            s.d = slogdet(kTilInv)[1] - slogdet(s.kInv)[1] + slogdet(R * np.eye(int(oi.sum())))[1]
        else:
            s.d = -np.log(np.linalg.det(s.kTil)) - np.log(np.linalg.det(s.kInv))
            s.u = -s.omg[m] @ s.kTil @ s.omg[m]
            for c in kids:
                s.d += st[c].d
                s.u += st[c].u
        s.ATil = [[s.A[k][l] - s.A[k][m] @ s.kTil @ s.A[m][l] for l in range(m + 1)] for k in range(m)]
        s.omgTil = [s.omg[k] - s.A[k][m] @ s.kTil @ s.omg[m] for k in range(m)]
        if leaf:
            s.BTil = [self.B_rows(a, rp) for a in chain]
        else:
            s.BTil = [np.zeros((len(rp), self.rank_of(a))) for a in chain]
            for c in kids:
                sc = st[c]
                sel = self.rows(c) - t.node_row0[i]
                for k in range(m + 1):
                    s.BTil[k][sel] = sc.BTil[k] - sc.BTil[m + 1] @ sc.kTil @ sc.A[m + 1][k]
        W, V = np.linalg.eigh(s.kTil)
        W = np.where(W < 0, -W, W)
        kTilC = V @ np.diag(np.sqrt(W))
        s.mean = s.BTil[m] @ s.kTil @ s.omg[m]
        s.var = np.linalg.norm(s.BTil[m] @ kTilC, axis=1) ** 2
        for c in kids:
            sel = self.rows(c) - t.node_row0[i]
            s.mean[sel] += st[c].mean
            s.var[sel] += st[c].var
        self.tm["posterior"] += time.perf_counter() - t0
        for c in kids:                       # the reference pops and deletes the children here
            if self.keep is None:
                del st[c]
            else:
                self.keep[c] = st.pop(c)
        if self.do_gc:
            t1 = time.perf_counter()
            gc.collect()
            self.tm["gc"] += time.perf_counter() - t1

    def visit(self, i):
        self.prior(i)
        for c in self.kids(i):
            self.visit(c)
        self.posterior(i)
        self.count += 1


def run_faithful(topo, locs, cov, obs, R: float, *, do_gc: bool = False, timers: Optional[dict] = None,
                 keep_nodes: Optional[dict] = None):
    """Evaluate the whole tree.  Returns dict(lik, d, u, mean[N], var[N], sd[N]).  ``keep_nodes``: a dict that
    receives every node's finished state (B, kInv, K, kC, A, omg, kTil, BTil, d, u, mean, var; rows in padded
    leaf order) keyed by node number, for the per-node golden blocks."""
    sys.setrecursionlimit(max(10000, sys.getrecursionlimit()))
    rec = _Recursion(topo, locs, cov, obs, R, do_gc)
    rec.keep = keep_nodes
    root = 0
    rec.visit(root)
    s = rec.st[root]
    if keep_nodes is not None:
        keep_nodes[root] = s
    mean = np.zeros(topo.N)
    var = np.zeros(topo.N)
    real = topo.perm >= 0
    mean[topo.perm[real]] = s.mean[real]
    var[topo.perm[real]] = s.var[real]
    d = float(np.asarray(s.d).reshape(-1)[0])
    u = float(np.asarray(s.u).reshape(-1)[0])
    if timers is not None:
        timers.update(rec.tm)
    return dict(lik=d + u, d=d, u=u, mean=mean, var=var, sd=np.sqrt(var))


def run_subtree_sample(topo, locs, cov, obs, R: float, top: int, *, do_gc: bool = True):
    """Bounded CPU-baseline sample: prior + posterior of every node of the subtree rooted at
    node number ``top`` - exactly the reference's per-node work for those nodes.  The
    ancestors of ``top`` are evaluated first on the subtree's rows only (set-up, not timed and
    not counted).  Returns (n_nodes, seconds, timers)."""
    sys.setrecursionlimit(max(10000, sys.getrecursionlimit()))
    rec = _Recursion(topo, locs, cov, obs, R, do_gc)
    chain = rec.lineage(int(top))
    r0, r1 = int(topo.node_row0[top]), int(topo.node_row1[top])
    nloc = r1 - r0
    src = topo.src
    # the pruned ancestors live on [subtree rows] + [every chain node's knots]
    ext = np.concatenate([rec.knots(a) for a in chain]) if chain else np.zeros(0, dtype=np.int64)
    rows = np.concatenate([np.arange(r0, r1), ext])
    index = {}
    for j, q in enumerate(ext):
        index.setdefault(int(q), nloc + j)

    def pos(rr):
        rr = np.asarray(rr)
        inside = (rr >= r0) & (rr < r1)
        out = np.where(inside, rr - r0, 0)
        for j in np.nonzero(~inside)[0]:
            out[j] = index[int(rr[j])]
        return out

    for a in chain:
        kq = rec.knots(a)
        B = rec.covf(src[rows], src[kq])
        for b in chain:
            if b == a:
                break
            B = B - rec.B_rows(b, rows) @ rec.K_of(b) @ rec.B_rows(b, kq).T
        rec.pruned[a] = dict(B=B, K=np.linalg.inv(B[pos(kq)]), pos=pos)
    t0 = time.perf_counter()
    rec.visit(int(top))
    return rec.count, time.perf_counter() - t0, rec.tm
