"""Drop-in ``MRATree`` with pyMRA's constructor, ``getLikelihood()`` and ``predict()``.

Mirrors the public surface of pyMRA/MRATree.py:

* ``MRATree(locs, r, cov, obs, R, M=-1, J=-1, critDepth=-1, verbose=True)``   MRATree.py:23
* ``getLikelihood()`` -> 1x1 ``np.matrix`` (root.d + root.u)                  MRATree.py:82-84
* ``predict()`` -> (N x 1 ``np.matrix`` mean, (N,) ``ndarray`` sd)            MRATree.py:90-94
* attributes ``locs, d, r, J, M, obs_inds, root`` with ``root.d, root.u, root.mean, root.var``

As in the reference all the work happens in the constructor - here it is: replay the tree on
the host (pymra_amd.topology), hand the flat tree + locations + observations to libmra_hip.so
through ctypes, run the HIP path once, copy the results back.  No NumPy fallback exists: without
the library or without a GPU the constructor raises.

``cov`` plug-in: the user's two-argument callable is probed once with symbolic location handles.
If it is built from this package's kernels (``mt.ExpCovFun``, ``mt.Matern32`` ... possibly times a
scalar) the probe yields a ``KernelSpec`` and the kernel is evaluated on the GPU.  Other callables
and dense ``np.matrix``/``ndarray`` covariances are evaluated block by block on the host and uploaded
(``MRA_KERNEL_HOST``, ``HipPlan.upload_host_cov``); everything else still runs on the GPU.

``critDepth`` is accepted for signature compatibility.  The reference's parallel mode forks one
process per subtree at that depth (MRANode.py:90-104); the forked children inherit the same RNG
state, so its 2-D knot draws - and therefore its results - differ from the serial mode.  This
implementation always reproduces the serial (default) mode; GPU parallelism is over all nodes of
a level, and multi-GPU sharding lives in pymra_amd.sharding.
"""
from __future__ import annotations

import logging

import numpy as np

from . import MRATools as mt
from .plan import HipPlan, create_with_replay
from .topology import build_topology, resolve_tree_shape

logger = logging.getLogger("pyMRA.MRATree")


class RootView:
    """The attributes of the reference's root ``Node`` that callers read (pyMRA/MRANode.py:26-45, 378-391,
    444-520).  ``d, u, mean, var`` come from the GPU pass; ``knots, B, kInv, k`` (the root's prior blocks,
    cov(locs, knots) etc.) are evaluated lazily on the host from the kernel when somebody asks."""

    def __init__(self, d, u, mean, var, N, topo, locs=None, kernel=None):
        self.d = np.matrix([[d]])
        self.u = np.matrix([[u]])
        self.mean = np.asmatrix(mean).reshape(N, 1) if mean is not None else None      # a view: no 8 MB copy
        self.var = var
        self.N = N
        self.res = 0
        self.ID = "r"
        self.leaf = bool(topo.node_leaf[0])
        kq = topo.knot_rows[topo.knot_ptr[0]:topo.knot_ptr[1]]
        self.kInds = np.sort(topo.perm[kq])
        self.children = []
        self.locs = locs
        self._kernel = kernel

    @property
    def knots(self):
        return np.asarray(self.locs)[self.kInds]

    @property
    def B(self):
        if self._kernel is None:
            raise AttributeError("root.B needs a device kernel (KernelSpec) to be evaluated lazily")
        return np.matrix(self._kernel.evaluate(np.asarray(self.locs), self.knots))

    @property
    def kInv(self):
        return self.B[self.kInds, :]

    @property
    def k(self):
        return np.linalg.inv(self.kInv)


def probe_cov(cov, d, locs=None):
    """Return a KernelSpec if ``cov`` is expressible as one of the device kernels, else None."""
    if isinstance(cov, mt.KernelSpec):
        return cov
    if not callable(cov):
        return None
    try:
        out = cov(mt.SymbolicLocs("a", d, locs), mt.SymbolicLocs("b", d, locs))
    except Exception:
        return None
    return out if isinstance(out, mt.KernelSpec) else None


class MRATree(object):

    def __init__(self, locs, r, cov, obs, R, M=-1, J=-1, critDepth=-1, verbose=True, device=0,
                 want_predict=True):
        self.locs = locs
        self.d = np.shape(self.locs)[1]
        N = len(locs)
        self.r = r
        self.M, self.J = resolve_tree_shape(N, self.d, r, M, J)
        if critDepth < 0:
            critDepth = self.M + 1
        self.critDepth = critDepth
        if not isinstance(R, float):
            # the reference indexes a non-float R as a matrix (MRANode.py:85-88) and fails on an int
            raise TypeError("R must be a Python float (scalar nugget variance); got %r" % type(R))
        obs_arr = np.asarray(obs, dtype=np.float64)
        self._obs_inds = None                   # (the reference's obs_inds, MRATree.py:62: computed when somebody reads it)
        self.obs, self.R = obs_arr, R

        spec = probe_cov(cov, self.d, np.asarray(locs, dtype=np.float64))
        if spec is not None and spec.circular and self.d != 1:
            raise ValueError("circular distances are defined for 1-D locations only")
        if spec is None and not callable(cov) and not isinstance(cov, np.ndarray):
            raise TypeError("cov must be a callable cov(locs1, locs2) or a dense N x N matrix")
        self.kernel = spec          # None: opaque callable / dense matrix -> values come from the host

        logger.debug('r: %d, \tJ: %d,\tM: %d' % (self.r, self.J, self.M))
        # large 2-D trees: tree replay, plan construction and the uploads in one library call (the plan is built beside the
        # sequential knot draws); everything else - 1-D, small nodes, KMeans rules - the general path
        both = create_with_replay(locs, r, self.M, self.J, obs_arr, R, device) if self.d == 2 else None
        if both is not None:
            self.plan, self.topology = both
        else:
            self.topology = build_topology(np.asarray(locs, dtype=np.float64), r, self.M, self.J)
            self.plan = HipPlan(self.topology, device=device)
            self.plan.set_locs(locs)
            self.plan.set_obs(obs_arr, R)
        if spec is not None:
            self.plan.set_kernel(spec.kind, spec.l, spec.sig, spec.scale, spec.circular)
        else:
            # cov-callable plug-in for anything the device cannot evaluate itself (pyMRA/MRANode.py:73-80,
            # 381-384): the host evaluates the raw covariance blocks, the GPU does everything else
            self.plan.upload_host_cov(cov, locs, obs_arr)
        self.plan.run(likelihood=True, predict=want_predict)
        d, u = self.plan.likelihood()
        if want_predict:
            mean, var, self._sd = self.plan.predict(with_sd=True)      # (sd beside var: predict() returns it, as the reference's np.sqrt(root.var))
        else:
            mean, var, self._sd = None, None, None
        self.root = RootView(d, u, mean, var, N, self.topology, np.asarray(locs, dtype=np.float64), spec)
        self._node_blocks = {}

    @property
    def obs_inds(self):
        if self._obs_inds is None:
            self._obs_inds = np.where(np.logical_not(np.isnan(self.obs)))[0]
        return self._obs_inds

    def getLikelihood(self):
        return self.root.d + self.root.u

    def predict(self):
        if self.root.mean is None:
            raise RuntimeError("constructed with want_predict=False")
        xP = self.root.mean
        sdP = self._sd if self._sd is not None else np.sqrt(self.root.var)
        return xP, sdP

    # ---- diagnostics surface (pyMRA/MRATree.py:101-132, 445-511); host-side de-whitening, see pymra_amd.diagnostics
    def getNodeBlocks(self, posterior=True):
        """Per-node ``B, kInv, k, kC`` (+ ``A, omg, kTil, kTilC, BTil[res]``, cumulative ``d, u``) of every node
        in level order - the attributes the reference keeps on its ``Node`` objects (pyMRA/MRANode.py:384-391,
        415-445, 486-507).  Costs two device passes."""
        from .diagnostics import collect_node_blocks
        key = bool(posterior)
        if key not in self._node_blocks:                    # two device passes + one block download per node: kept until reevaluate()
            self._node_blocks[key] = collect_node_blocks(self, posterior=posterior)
        return self._node_blocks[key]

    def getNodesBFS(self, groupByResolution=False):
        """Nodes in breadth-first order (pyMRA/MRATree.py:101-120) as ``NodeBlocks`` records.  (In the reference
        the tree is torn down at the end of construction, MRANode.py:108-111, and only the root is ever seen.)"""
        nodes = self.getNodeBlocks(posterior=True)
        if not groupByResolution:
            return nodes
        return [[nb for nb in nodes if nb.res == m] for m in range(self.topology.n_levels)]

    def getBasisFunctionsMatrix(self, distr="prior", groupByResolution=False, order='root', timesKC=False):
        """pyMRA/MRATree.py:445-511."""
        from .diagnostics import basis_functions_matrix
        return basis_functions_matrix(self, distr, groupByResolution, order, timesKC)

    # re-evaluate with other kernel parameters on the same tree (plan reuse for MLE loops,
    # README.md:96-104 builds a new MRATree per objective call)
    def reevaluate(self, cov, want_predict=False):
        spec = probe_cov(cov, self.d, np.asarray(self.locs, dtype=np.float64))
        if spec is None:
            raise NotImplementedError("cov must be a device kernel")
        self.kernel = spec
        self._node_blocks = {}
        self.plan.set_kernel(spec.kind, spec.l, spec.sig, spec.scale, spec.circular)
        self.plan.run(likelihood=True, predict=want_predict)
        d, u = self.plan.likelihood()
        mean, var, self._sd = self.plan.predict(with_sd=True) if want_predict else (None, None, None)
        self.root = RootView(d, u, mean, var, len(self.locs), self.topology, np.asarray(self.locs, dtype=np.float64), spec)
        return self.getLikelihood()
