"""Covariance plug-ins, distances and grid generators with the reference's call surface.

Mirrors the part of pyMRA/MRATools.py that the inference path and its callers use:

* ``dist``              pyMRA/MRATools.py:229-245
* ``Iden``              pyMRA/MRATools.py:256-262
* ``ExpCovFun``         pyMRA/MRATools.py:265-269
* ``Matern52``          pyMRA/MRATools.py:281-285
* ``Matern32``          pyMRA/MRATools.py:289-293
* ``GaussianCovFun``    pyMRA/MRATools.py:297-301
* ``genLocations``      pyMRA/MRATools.py:180-187
* ``genLocations2d``    pyMRA/MRATools.py:192-203

Each kernel keeps the reference's signature and, for array inputs, its ``np.matrix`` result.
The device path adds one thing: when a kernel is called with the *symbolic* location handles
that ``MRATree`` uses to probe a user's ``cov`` callable, it returns a ``KernelSpec`` (kind +
parameters, closed under multiplication by a scalar such as ``sigma*mt.Matern32(...)``,
cf. pyMRA/tests/test-param-est.py:84) instead of numbers; ``MRATree`` then evaluates that
kernel on the GPU.  Any callable that does not produce a ``KernelSpec`` is treated as an
opaque function and evaluated on the host block by block.
"""
from __future__ import annotations

import numpy as np
from scipy.spatial.distance import cdist, pdist, squareform

# kernel kinds understood by the HIP library (include/mra_hip.h: MRA_KERNEL_*)
KIND_EXP = 0
KIND_MATERN32 = 1
KIND_MATERN52 = 2
KIND_GAUSSIAN = 3
KIND_IDEN = 4
KIND_KANTER = 5


class SymbolicLocs:
    """Stand-in for a location array while ``MRATree`` probes the ``cov`` callable."""
    __slots__ = ("tag", "ndim", "shape", "data")

    def __init__(self, tag, d, data=None):
        self.tag = tag
        self.ndim = 2
        self.shape = (0, d)
        self.data = data            # the caller's full location array (needed by KanterCovFun(radius=int))

    def __len__(self):
        return 1            # `if len(locs2)` in the reference's dist(): "second set given"


class KernelSpec:
    """kind + (l, sig) + an overall scale; value = scale * k_kind(D; l, sig)."""
    __slots__ = ("kind", "l", "sig", "scale", "circular")

    def __init__(self, kind, l=1.0, sig=1.0, scale=1.0, circular=False):
        self.kind, self.l, self.sig, self.scale, self.circular = int(kind), float(l), float(sig), float(scale), bool(circular)

    def __mul__(self, c):
        if isinstance(c, (int, float, np.floating, np.integer)):
            return KernelSpec(self.kind, self.l, self.sig, self.scale * float(c), self.circular)
        return NotImplemented

    __rmul__ = __mul__

    def __truediv__(self, c):
        return self.__mul__(1.0 / float(c))

    def params(self):
        return np.array([self.l, self.sig, self.scale], dtype=np.float64)

    def evaluate(self, locs1, locs2):
        """Host evaluation (tests / diagnostics); same arithmetic as the array path below."""
        fn = _HOST_KERNELS[self.kind]
        return self.scale * np.asarray(fn(np.asarray(locs1), np.asarray(locs2), self.l, self.sig, self.circular))

    def __repr__(self):
        return "KernelSpec(kind=%d, l=%g, sig=%g, scale=%g)" % (self.kind, self.l, self.sig, self.scale)


def _symbolic(*args):
    return any(isinstance(a, SymbolicLocs) for a in args)


# ----------------------------------------------------------------------------------------
def genLocations(NGrid, lb=0, ub=1, random=False):
    if random:
        pts = np.random.uniform(lb, ub, NGrid)
    else:
        pts = np.linspace(lb, ub, num=NGrid + 1)[1:]
    return pts.reshape((NGrid, 1))


def genLocations2d(Nx, lbx=0, ubx=1, Ny=0, lby=0, uby=1):
    """Regular grid, x fastest: row i = (x[i mod Nx], y[i div Nx])."""
    if not Ny:
        Ny = Nx
    gx, gy = np.meshgrid(np.linspace(lbx, ubx, num=Nx), np.linspace(lby, uby, num=Ny))
    return np.hstack((gx.reshape(Nx * Ny, 1), gy.reshape(Nx * Ny, 1)))


def dist(locs, locs2=np.array([]), circular=False):
    locs = locs if np.ndim(locs) == 2 else np.reshape(locs, [len(locs), 1])
    if circular:
        other = locs2 if len(locs2) else locs
        a, b = np.meshgrid(locs, other)
        lo, hi = np.minimum(a, b), np.maximum(a, b)
        return np.matrix(np.minimum(hi - lo, lo + 1 - hi).T)
    if len(locs2):
        locs2 = locs2 if np.ndim(locs2) == 2 else np.reshape(locs2, [len(locs2), 1])
        return np.matrix(cdist(locs, locs2))
    return np.matrix(squareform(pdist(locs)))


def _exp(l1, l2, l, sig, circular):
    return np.exp(-dist(l1, l2, circular) / l)


def _m32(l1, l2, l, sig, circular):
    D = dist(l1, l2, circular)
    return sig * np.multiply(1 + np.sqrt(3) * D / l, np.exp(-np.sqrt(3) * D / l))


def _m52(l1, l2, l, sig, circular):
    D = dist(l1, l2, circular)
    return sig * np.multiply(1 + np.sqrt(5) * D / l + (5 / 3) * np.square(D / l), np.exp(-np.sqrt(5) * D / l))


def _gauss(l1, l2, l, sig, circular):
    D = dist(l1, l2, circular)
    return sig * np.exp(-np.square(D) / (2 * (l ** 2)))


def _iden(l1, l2, l, sig, circular):
    D = dist(l1, l2, circular)
    out = np.matrix(np.zeros(D.shape))
    out[np.where(D == 0)] = 1
    return out


def _kanter(l1, l2, l, sig, circular):
    """Kanter's compactly supported taper with radius l (pyMRA/MRATools.py:318-323)."""
    D = np.array(dist(l1, l2, circular)) / l
    with np.errstate(divide="ignore", invalid="ignore"):
        p2 = 2 * np.pi * D
        R = (1 - D) * np.sin(p2) / p2 + 1 / np.pi * (1 - np.cos(p2)) / p2
    R[D > 1] = 0
    R[D == 0] = 1
    return R


_HOST_KERNELS = {KIND_EXP: _exp, KIND_MATERN32: _m32, KIND_MATERN52: _m52, KIND_GAUSSIAN: _gauss,
                 KIND_IDEN: _iden, KIND_KANTER: _kanter}


def determine_radius(k, h, ndim=2):
    """Taper radius that leaves about k grid points inside the support (mesh width h); same rule as
    pyMRA/MRATools.py:329-388: 1 grid row -> int(k/2) h; otherwise the ring of the square lattice whose
    cumulative point count is nearest to k."""
    if ndim == 1:
        return int(k / 2) * h
    if k == 0:
        raise ValueError("Ensemble size must be stricly positive")
    s = np.floor(np.sqrt(k))
    sf = s - 1 if s % 2 == 0 else s
    if k == sf ** 2:
        return h * 1.01 * (sf - 1) / 2 * np.sqrt(2)
    base = (sf - 1) / 2.0
    counts = [sf ** 2]
    while counts[-1] < (sf + 2) ** 2:
        step = 4 if (len(counts) == 1 or (sf + 2) ** 2 - counts[-1] == 4) else 8
        counts.append(counts[-1] + step)
    counts = np.asarray(counts)
    ind = counts.searchsorted(k)
    pick = ind - 1 if k <= (counts[ind - 1] + counts[ind]) / 2.0 else ind
    if pick == 0:
        return h * base * np.sqrt(2) + h * 0.01
    return h * np.sqrt((base + 1) ** 2 + (pick - 1) ** 2) + h * 0.01


def KanterCovFun(locs, locs2=np.array([]), radius=1.0, circular=False):
    """radius: float = taper radius; int = target number of points in the support (resolved from the grid
    spacing of ``locs`` like the reference does; on the device path from the tree's full location set)."""
    src = locs.data if isinstance(locs, SymbolicLocs) else locs
    if isinstance(radius, int):
        if src is None:
            raise ValueError("KanterCovFun(radius=int) needs the locations to derive the mesh width")
        xs = np.sort(np.unique(src[:, 0]))
        ndim = len(np.unique(src[:, 1])) if src.shape[1] > 1 else 1
        radius = determine_radius(radius, xs[1] - xs[0], ndim=ndim)
    if _symbolic(locs, locs2):
        return KernelSpec(KIND_KANTER, radius, 1.0, 1.0, circular)
    return _kanter(locs, locs2, radius, 1.0, circular)


def Iden(locs, locs2=np.array([]), l=1, circular=False):
    if _symbolic(locs, locs2):
        return KernelSpec(KIND_IDEN, l, 1.0, 1.0, circular)
    return _iden(locs, locs2, l, 1.0, circular)


def ExpCovFun(locs, locs2=np.array([]), l=1, circular=False):
    if _symbolic(locs, locs2):
        return KernelSpec(KIND_EXP, l, 1.0, 1.0, circular)
    return _exp(locs, locs2, l, 1.0, circular)


def Matern52(locs, locs2=np.array([]), l=1, sig=1, circular=False):
    if _symbolic(locs, locs2):
        return KernelSpec(KIND_MATERN52, l, sig, 1.0, circular)
    return np.matrix(_m52(locs, locs2, l, sig, circular))


def Matern32(locs, locs2=np.array([]), l=1, sig=1, circular=False):
    if _symbolic(locs, locs2):
        return KernelSpec(KIND_MATERN32, l, sig, 1.0, circular)
    return np.matrix(_m32(locs, locs2, l, sig, circular))


def GaussianCovFun(locs, locs2=np.array([]), l=1, sig=1, circular=False):
    if _symbolic(locs, locs2):
        return KernelSpec(KIND_GAUSSIAN, l, sig, 1.0, circular)
    return np.matrix(_gauss(locs, locs2, l, sig, circular))
