"""ORACLE (test infrastructure, not product code) - level-wise executable spec of the HIP path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the product path (pymra_amd) never does.

Same model as the reference (pyMRA/MRANode.py:378-523), restated in the factorised form
the GPU kernels execute, on the same padded leaf-ordered layout, so that
every intermediate device buffer has a NumPy twin here:

  prior      W^m[S_j] = (C(S_j,Q_j) - W^{<m}[S_j] W^{<m}[Q_j]^T) L_j^{-T},  L_j L_j^T = v_m(Q_j,Q_j)
             (whitened B of MRANode.py:384-387: B_j = W^m L_j^T, k = L_j^{-T} L_j^{-1})
  leaf       observation-space form of MRANode.py:415-459: with o = observed rows,
             C = v_m(o,o) + R I = Lc Lc^T,  U = Lc^{-1}[W^{<m}[o] | y_o],  T = Lc^{-1} v_m(o,S)
             d_leaf = logdet C,  Gt = U^T U  (= ATil/omgTil/u of :452-480 in whitened form),
             mean = T^T u_y,  var = diag v_m(S,S) - colsumsq(T),  W^{<m}[S] -= T^T U  (BTil, :489-495)
  non-leaf   F = I_own + sum_children Gt (MRANode.py:434-445); partial Cholesky of the own block:
             Lt = chol(F_oo), Zt = F_ao Lt^{-T}, d = logdet F_oo (:463), Gt = F_aa - Zt Zt^T (:476-480)
  predict    X = W^m[S_j] Lt^{-T}; var += rowsumsq(X) (:511); [W^{<m}|ycol][S_j] -= X Zt^T (:495, :510)
  outputs    lik = sum d + Gt_root[y,y] (MRATree.py:84), mean = -ycol, sd = sqrt(var) (MRATree.py:92-93)

The equivalence of the observation-space leaf with the reference's knot-space leaf is the
Woodbury identity (leaf knots are all of the leaf's not-yet-used locations, MRANode.py:42-45,
so B K B^T restricted to the leaf is the residual covariance itself); it is exact in exact
arithmetic and is checked numerically against oracle/mra_faithful.py and the goldens.
"""
from __future__ import annotations

import numpy as np
from scipy.linalg import solve_triangular

YB = 16          # width of the y column block (col 0 = y, rest zero)


class Layout:
    """Column layout of the whitened-basis array and the front sizes (shared with the HIP host)."""

    def __init__(self, topo):
        L = topo.n_levels
        cw = np.asarray(topo.cw, dtype=np.int64)
        self.cw = cw
        self.Ka = int(cw.sum())
        self.coff = np.array([int(cw[m + 1:].sum()) for m in range(L)], dtype=np.int64)  # deepest first
        self.asuf = self.coff + cw                   # first ancestor column of a level-m node
        self.ldw = self.Ka + YB
        self.nf = self.ldw - self.coff               # front size of a non-leaf level-m node
        self.na = self.ldw - self.asuf               # ancestors + y block (Schur / leaf Gt size)


def _cov_callable(cov):
    if hasattr(cov, "evaluate"):
        return lambda a, b: np.asarray(cov.evaluate(a, b), dtype=np.float64)
    return lambda a, b: np.asarray(cov(a, b), dtype=np.float64)


def run_levelwise(topo, locs, cov, obs, R: float, *, predict: bool = True, keep: bool = False,
                  reduce_level: int = -1, allreduce=None, lean: bool = False):
    """Returns dict(lik, d, u, mean[N], var[N], sd[N]) (+ intermediate buffers if keep)."""
    coords = np.asarray(locs, dtype=np.float64)
    if coords.ndim == 1:
        coords = coords.reshape(-1, 1)
    covf = _cov_callable(cov)
    lay = Layout(topo)
    P, Ka, ldw = topo.P, lay.Ka, lay.ldw
    X = coords[topo.src]
    yp = np.asarray(obs, dtype=np.float64).reshape(-1)[topo.src].copy()
    yp[topo.perm < 0] = np.nan
    W = np.zeros((P, ldw))
    W[:, Ka] = np.where(np.isfinite(yp), yp, 0.0)
    nn = topo.n_nodes
    Lp = [None] * nn
    dnode = np.zeros(nn)
    Gt = [None] * nn
    Lt = [None] * nn
    Zt = [None] * nn
    var = np.zeros(P)
    R = float(R)

    def rows(i):
        return int(topo.node_row0[i]), int(topo.node_row1[i])

    # ---------------- prior, top-down -------------------------------------------------------
    for m in range(topo.n_levels):
        cwm = int(lay.cw[m])
        if cwm == 0:
            continue
        a0 = int(lay.asuf[m])
        c0 = int(lay.coff[m])
        for i in range(int(topo.level_ptr[m]), int(topo.level_ptr[m + 1])):
            if topo.node_leaf[i]:
                continue
            r0, r1 = rows(i)
            kq = topo.knot_rows[topo.knot_ptr[i]:topo.knot_ptr[i + 1]]
            rj = len(kq)
            Rf = covf(X[r0:r1], X[kq]) - W[r0:r1, a0:Ka] @ W[kq, a0:Ka].T
            kInv = np.eye(cwm)
            kInv[:rj, :rj] = Rf[kq - r0]
            L = np.linalg.cholesky(kInv)
            Lp[i] = L
            W[r0:r1, c0:c0 + rj] = solve_triangular(L[:rj, :rj], Rf.T, lower=True).T
            W[r0:r1, c0 + rj:c0 + cwm] = 0.0

    # ---------------- leaves (observation space) ------------------------------------------------
    def leaf_node(i):
        m = int(topo.node_level[i])
        a0 = int(lay.asuf[m])
        na = int(lay.na[m])
        r0, r1 = rows(i)
        yv = yp[r0:r1]
        o = np.nonzero(np.isfinite(yv))[0]
        Wa = W[r0:r1, a0:Ka]
        Xs = X[r0:r1]
        # prior residual variance of every row: C(x,x) - |W^{<m}[x]|^2
        # (a KernelSpec is stationary: one number; an opaque callable is evaluated per leaf)
        if hasattr(cov, "evaluate"):
            dv = np.full(len(Xs), covf(Xs[:1], Xs[:1])[0, 0])
        else:
            dv = np.diag(covf(Xs, Xs)).copy()
        dv = dv - np.einsum("ij,ij->i", Wa, Wa)
        G = np.zeros((na, na))
        if len(o):
            Vso = covf(Xs, Xs[o]) - Wa @ Wa[o].T
            C = Vso[o] + R * np.eye(len(o))
            C = 0.5 * (C + C.T)
            Lc = np.linalg.cholesky(C)
            dnode[i] = 2.0 * np.log(np.diag(Lc)).sum()
            rhs = np.zeros((len(o), na))
            rhs[:, :na - YB] = Wa[o]
            rhs[:, na - YB] = yv[o]
            U = solve_triangular(Lc, rhs, lower=True)
            G = U.T @ U
            if predict:
                T = solve_triangular(Lc, Vso.T, lower=True)           # n_o x N_j
                var[r0:r1] = np.maximum(dv - np.einsum("ij,ij->j", T, T), 0.0)
                upd = T.T @ U                                            # N_j x na
                W[r0:r1, a0:Ka] -= upd[:, :na - YB]
                W[r0:r1, Ka] = -upd[:, na - YB]
        else:
            var[r0:r1] = np.maximum(dv, 0.0)
            W[r0:r1, Ka] = 0.0
        Gt[i] = G

    if not lean:
        for i in range(nn):
            if topo.node_leaf[i]:
                leaf_node(i)

    # ---------------- non-leaf fronts, bottom-up ----------------------------------------------------
    d_below = None
    if lean:
        # same arithmetic in depth-first post-order: a node's Schur block dies as soon as its parent has summed it, so only
        # O(depth * J) blocks are alive at any time (level by level, config 5 would hold 65536 leaf blocks of 528^2)
        if reduce_level >= 0:
            raise ValueError("lean=True is for unsharded runs")

        def post(i):
            if topo.node_leaf[i]:
                leaf_node(i)
                g = Gt[i]
                Gt[i] = None
                return g
            m = int(topo.node_level[i])
            cwm, nf = int(lay.cw[m]), int(lay.nf[m])
            F = np.zeros((nf, nf))
            for c in topo.child_list[topo.child_ptr[i]:topo.child_ptr[i + 1]]:
                F += post(int(c))
            F[:cwm, :cwm] += np.eye(cwm)
            L = np.linalg.cholesky(F[:cwm, :cwm])
            Z = solve_triangular(L, F[:cwm, cwm:], lower=True).T
            dnode[i] = 2.0 * np.log(np.diag(L)).sum()
            Lt[i], Zt[i] = L, Z
            return F[cwm:, cwm:] - Z @ Z.T

        Gt[0] = post(0)
    for m in range(topo.n_levels - 1, -1, -1) if not lean else []:
        cwm = int(lay.cw[m])
        ids = [i for i in range(int(topo.level_ptr[m]), int(topo.level_ptr[m + 1])) if not topo.node_leaf[i]]
        nf = int(lay.nf[m])
        fronts = []
        for i in ids:
            F = np.zeros((nf, nf))
            for c in topo.child_list[topo.child_ptr[i]:topo.child_ptr[i + 1]]:
                F += Gt[c]
                if not keep:
                    Gt[c] = None
            fronts.append(F)
        if m == reduce_level and allreduce is not None:
            # the one exchange of the sharded path: fronts without identity + the local log-det sum
            lo = int(topo.level_ptr[m + 1])
            buf = np.concatenate([f.ravel() for f in fronts] + [np.array([dnode[lo:].sum()])])
            buf = allreduce(buf)
            fronts = [buf[t * nf * nf:(t + 1) * nf * nf].reshape(nf, nf).copy() for t in range(len(ids))]
            d_below = float(buf[-1])
            dnode[lo:] = 0.0
        for i, F in zip(ids, fronts):
            F[:cwm, :cwm] += np.eye(cwm)
            L = np.linalg.cholesky(F[:cwm, :cwm])
            Z = solve_triangular(L, F[:cwm, cwm:], lower=True).T        # na x cw  (rows-below form)
            dnode[i] = 2.0 * np.log(np.diag(L)).sum()
            Gt[i] = F[cwm:, cwm:] - Z @ Z.T
            Lt[i], Zt[i] = L, Z

    # ---------------- predictive moments, bottom-up -------------------------------------------------
    if predict:
        for m in range(topo.n_levels - 1, -1, -1):
            cwm = int(lay.cw[m])
            c0, a0 = int(lay.coff[m]), int(lay.asuf[m])
            for i in range(int(topo.level_ptr[m]), int(topo.level_ptr[m + 1])):
                if topo.node_leaf[i]:
                    continue
                r0, r1 = rows(i)
                Xm = solve_triangular(Lt[i], W[r0:r1, c0:c0 + cwm].T, lower=True).T
                var[r0:r1] += np.einsum("ij,ij->i", Xm, Xm)
                W[r0:r1, a0:ldw] -= Xm @ Zt[i].T

    root = 0
    d = float(dnode.sum()) + (d_below if d_below is not None else 0.0)
    u = float(Gt[root][-YB, -YB])
    mean = np.zeros(topo.N)
    v = np.zeros(topo.N)
    good = topo.in_leaf
    mean[topo.perm[good]] = -W[good, Ka]
    v[topo.perm[good]] = var[good]
    out = dict(lik=d + u, d=d, u=u, mean=mean, var=v, sd=np.sqrt(v))
    if keep:
        out.update(W=W, Lp=Lp, Lt=Lt, Zt=Zt, Gt=Gt, dnode=dnode, layout=lay, var_p=var)
    return out
