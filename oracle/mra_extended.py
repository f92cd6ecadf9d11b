"""ORACLE (test infrastructure, not product code) - the level-wise algorithm in 80-bit extended precision.

Only tests/ may import this module.  Purpose: adjudicate between the reference's float64 numbers
and this implementation's where they disagree by more than rounding (Matern32 predictive sd, see
DESIGN.md section 7 and SURVEY.md section 7 hard part 5): the same model is evaluated with
np.longdouble arithmetic (64-bit mantissa on x86), hand-written Cholesky / triangular solves and
longdouble kernel evaluation, for small trees only (pure NumPy loops).
Same formulas as oracle/mra_levelwise.py (pyMRA/MRANode.py:378-523 in factorised form).
"""
from __future__ import annotations

import numpy as np

from .mra_levelwise import Layout, YB

LD = np.longdouble


def _chol(A):
    A = np.array(A, dtype=LD)
    n = A.shape[0]
    L = np.zeros_like(A)
    for j in range(n):
        d = A[j, j] - L[j, :j] @ L[j, :j]
        if not d > 0:
            raise np.linalg.LinAlgError("not positive definite")
        L[j, j] = np.sqrt(d)
        if j + 1 < n:
            L[j + 1:, j] = (A[j + 1:, j] - L[j + 1:, :j] @ L[j, :j]) / L[j, j]
    return L


def _solve_lower(L, B):
    B = np.array(B, dtype=LD)
    X = np.zeros_like(B)
    for i in range(L.shape[0]):
        X[i] = (B[i] - L[i, :i] @ X[:i]) / L[i, i]
    return X


def _kernel(spec, a, b):
    a = np.asarray(a, dtype=LD)
    b = np.asarray(b, dtype=LD)
    D2 = np.zeros((len(a), len(b)), dtype=LD)
    for c in range(a.shape[1]):
        D2 += (a[:, c:c + 1] - b[:, c:c + 1].T) ** 2
    D = np.sqrt(D2)
    l, sig, scale = LD(spec.l), LD(spec.sig), LD(spec.scale)
    if spec.kind == 0:
        v = np.exp(-D / l)
    elif spec.kind == 1:
        t = np.sqrt(LD(3)) * D / l
        v = sig * (1 + t) * np.exp(-t)
    elif spec.kind == 2:
        t = np.sqrt(LD(5)) * D / l
        v = sig * (1 + t + (LD(5) / 3) * (D / l) ** 2) * np.exp(-t)
    elif spec.kind == 3:
        v = sig * np.exp(-D2 / (2 * l * l))
    else:
        v = (D == 0).astype(LD)
    return scale * v


def run_extended(topo, locs, spec, obs, R):
    """dict(lik, mean, var, sd) as float64 arrays, computed in extended precision."""
    lay = Layout(topo)
    P, Ka, ldw = topo.P, lay.Ka, lay.ldw
    X = np.asarray(locs, dtype=np.float64)
    if X.ndim == 1:
        X = X.reshape(-1, 1)
    X = X[topo.src]
    yp = np.asarray(obs, dtype=np.float64).reshape(-1)[topo.src].copy()
    yp[topo.perm < 0] = np.nan
    W = np.zeros((P, ldw), dtype=LD)
    nn = topo.n_nodes
    dn = np.zeros(nn, dtype=LD)
    Gt, Lt, Zt = [None] * nn, [None] * nn, [None] * nn
    var = np.zeros(P, dtype=LD)
    R = LD(R)
    for m in range(topo.n_levels):
        cwm = int(lay.cw[m])
        if cwm == 0:
            continue
        a0, c0 = int(lay.asuf[m]), int(lay.coff[m])
        for i in range(int(topo.level_ptr[m]), int(topo.level_ptr[m + 1])):
            if topo.node_leaf[i]:
                continue
            r0, r1 = int(topo.node_row0[i]), int(topo.node_row1[i])
            kq = topo.knot_rows[topo.knot_ptr[i]:topo.knot_ptr[i + 1]]
            rj = len(kq)
            Rf = _kernel(spec, X[r0:r1], X[kq]) - W[r0:r1, a0:Ka] @ W[kq, a0:Ka].T
            L = _chol(Rf[kq - r0])
            W[r0:r1, c0:c0 + rj] = _solve_lower(L, Rf.T).T
    for i in range(nn):
        if not topo.node_leaf[i]:
            continue
        m = int(topo.node_level[i])
        a0, na = int(lay.asuf[m]), int(lay.na[m])
        r0, r1 = int(topo.node_row0[i]), int(topo.node_row1[i])
        yv = yp[r0:r1]
        o = np.nonzero(np.isfinite(yv))[0]
        Wa = W[r0:r1, a0:Ka]
        Xs = X[r0:r1]
        dv = np.full(len(Xs), _kernel(spec, Xs[:1], Xs[:1])[0, 0], dtype=LD) - np.einsum("ij,ij->i", Wa, Wa)
        G = np.zeros((na, na), dtype=LD)
        if len(o):
            Vso = _kernel(spec, Xs, Xs[o]) - Wa @ Wa[o].T
            C = Vso[o] + R * np.eye(len(o), dtype=LD)
            Lc = _chol(0.5 * (C + C.T))
            dn[i] = 2 * np.log(np.diag(Lc)).sum()
            rhs = np.zeros((len(o), na), dtype=LD)
            rhs[:, :na - YB] = Wa[o]
            rhs[:, na - YB] = yv[o]
            U = _solve_lower(Lc, rhs)
            G = U.T @ U
            T = _solve_lower(Lc, Vso.T)
            var[r0:r1] = np.maximum(dv - np.einsum("ij,ij->j", T, T), 0)
            upd = T.T @ U
            W[r0:r1, a0:Ka] -= upd[:, :na - YB]
            W[r0:r1, Ka] = -upd[:, na - YB]
        else:
            var[r0:r1] = np.maximum(dv, 0)
        Gt[i] = G
    for m in range(topo.n_levels - 1, -1, -1):
        cwm = int(lay.cw[m])
        for i in range(int(topo.level_ptr[m]), int(topo.level_ptr[m + 1])):
            if topo.node_leaf[i]:
                continue
            nf = int(lay.nf[m])
            F = np.zeros((nf, nf), dtype=LD)
            F[:cwm, :cwm] = np.eye(cwm, dtype=LD)
            for c in topo.child_list[topo.child_ptr[i]:topo.child_ptr[i + 1]]:
                F += Gt[c]
            L = _chol(F[:cwm, :cwm])
            Z = _solve_lower(L, F[:cwm, cwm:]).T
            dn[i] = 2 * np.log(np.diag(L)).sum()
            Gt[i] = F[cwm:, cwm:] - Z @ Z.T
            Lt[i], Zt[i] = L, Z
    for m in range(topo.n_levels - 1, -1, -1):
        cwm = int(lay.cw[m])
        c0, a0 = int(lay.coff[m]), int(lay.asuf[m])
        for i in range(int(topo.level_ptr[m]), int(topo.level_ptr[m + 1])):
            if topo.node_leaf[i]:
                continue
            r0, r1 = int(topo.node_row0[i]), int(topo.node_row1[i])
            Xm = _solve_lower(Lt[i], W[r0:r1, c0:c0 + cwm].T).T
            var[r0:r1] += np.einsum("ij,ij->i", Xm, Xm)
            W[r0:r1, a0:ldw] -= Xm @ Zt[i].T
    lik = dn.sum() + Gt[0][-YB, -YB]
    mean = np.zeros(topo.N)
    v = np.zeros(topo.N)
    good = topo.in_leaf
    mean[topo.perm[good]] = np.asarray(-W[good, Ka], dtype=np.float64)
    v[topo.perm[good]] = np.asarray(var[good], dtype=np.float64)
    return dict(lik=float(lik), mean=mean, var=v, sd=np.sqrt(v))
