"""ctypes binding of libmra_hip.so (include/mra_hip.h) - the only way the Python host reaches the GPU.

There is deliberately no CPU fallback: if the shared library is missing, or no AMD GPU is
visible when a plan is created, an exception is raised.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# PYMRA_AMD_LIB: another build of the same library (tests/test_asan_host.py points it at the host-sanitizer build)
LIB_PATH = os.environ.get("PYMRA_AMD_LIB") or os.path.join(_HERE, "libmra_hip.so")

MRA_RUN_LIKELIHOOD = 1
MRA_RUN_PREDICT = 2
MRA_RUN_SPLIT = 4
MRA_KERNEL_HOST = 100
MRA_OPT_KERNEL_TIMING = 1
MRA_OPT_FUSED = 2
MRA_OPT_GEMM_LDS = 3
MRA_OPT_FRONT_FUSED = 4
MRA_OPT_KNOT_CHAIN = 5
MRA_OPT_LEAF_GEMM = 6
MRA_OPT_LEAF_SOLVE = 7
MRA_OPT_PRED_UPDATE = 8
MRA_OPT_LEAF_SOLVE_SPLIT = 10
MRA_OPT_CHOL_TILES = 11
MRA_OPT_SEG_GEMM_LDS = 12
MRA_OPT_UT_GATHER = 13
MRA_OPT_SYRK_BLK = 14
MRA_OPT_PRIOR_LEVEL = 15
MRA_OPT_HI_FOLD = 16
MRA_OPT_LIK_ROWS = 17
MRA_BLOCK_W_ROWS, MRA_BLOCK_LPRIOR, MRA_BLOCK_FRONT, MRA_BLOCK_LEAF = 0, 1, 2, 3

ERR_NAMES = {-1: "MRA_ERR_INVALID", -2: "MRA_ERR_HIP", -3: "MRA_ERR_NOT_SPD", -4: "MRA_ERR_STATE",
             -5: "MRA_ERR_COMM"}

# every symbol include/mra_hip.h declares (tests check that the library exports all of them)
EXPORTS = [
    "mra_device_count", "mra_release_cached_memory", "mra_plan_create", "mra_plan_destroy", "mra_plan_set_locs", "mra_plan_set_obs",
    "mra_plan_set_kernel", "mra_plan_set_locs_rows", "mra_plan_set_obs_rows", "mra_get_predict_rows", "mra_get_predict_rows_sd", "mra_eval_kernel", "mra_plan_set_cov_block", "mra_run", "mra_get_likelihood", "mra_get_predict",
    "mra_get_buffer", "mra_get_node_block", "mra_get_timers", "mra_plan_set_option", "mra_plan_get_option", "mra_plan_prepare", "mra_kernel_family_count",
    "mra_get_kernel_stats", "mra_get_kernel_work", "mra_device_synchronize", "mra_plan_info", "mra_comm_unique_id", "mra_comm_init",
    "mra_plan_set_reduce_level", "mra_reduce_size", "mra_reduce_export", "mra_reduce_import",
    "mra_run_resume", "mra_last_error", "mra_version",
    "mra_tree_replay_2d", "mra_tree_replay_2d_into", "mra_plan_create_replay_2d", "mra_tree_sizes", "mra_tree_export", "mra_tree_free",
]


class MraTopologyStruct(C.Structure):
    _fields_ = [("P", C.c_int64), ("d", C.c_int32), ("n_levels", C.c_int32), ("n_nodes", C.c_int32),
                ("level_ptr", C.c_void_p), ("node_row0", C.c_void_p), ("node_row1", C.c_void_p),
                ("node_leaf", C.c_void_p), ("node_parent", C.c_void_p), ("child_ptr", C.c_void_p),
                ("child_list", C.c_void_p), ("knot_ptr", C.c_void_p), ("knot_rows", C.c_void_p),
                ("cw", C.c_void_p)]


class MraError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s: %s" % (ERR_NAMES.get(code, str(code)), msg))
        self.code = code


_lib = None


def load_library():
    """Load libmra_hip.so (built by __graft_entry__.build()).  Raises if it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("pymra_amd: %s not found - build it with `python -c 'import __graft_entry__ as g; "
                          "g.build()'` (hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64, u32, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_uint32, C.c_double
    sig = {
        "mra_device_count": (C.c_int, []),
        "mra_release_cached_memory": (C.c_int, []),
        "mra_plan_create": (C.c_int, [C.POINTER(vp), C.POINTER(MraTopologyStruct), C.c_int]),
        "mra_plan_destroy": (C.c_int, [vp]),
        "mra_plan_set_locs": (C.c_int, [vp, vp]),
        "mra_plan_set_obs": (C.c_int, [vp, vp, dbl]),
        "mra_plan_set_locs_rows": (C.c_int, [vp, vp, vp]),
        "mra_plan_set_obs_rows": (C.c_int, [vp, vp, vp, vp, dbl]),
        "mra_get_predict_rows": (C.c_int, [vp, vp, vp, i64, vp, vp]),
        "mra_get_predict_rows_sd": (C.c_int, [vp, vp, vp, i64, vp, vp, vp]),
        "mra_plan_set_kernel": (C.c_int, [vp, C.c_int, vp, C.c_int]),
        "mra_plan_set_cov_block": (C.c_int, [vp, i32, vp, i64, i64, vp]),
        "mra_eval_kernel": (C.c_int, [C.c_int, vp, C.c_int, vp, i64, vp]),
        "mra_run": (C.c_int, [vp, u32]),
        "mra_run_resume": (C.c_int, [vp]),
        "mra_get_likelihood": (C.c_int, [vp, C.POINTER(dbl), C.POINTER(dbl)]),
        "mra_get_predict": (C.c_int, [vp, vp, vp]),
        "mra_get_buffer": (C.c_int, [vp, C.c_int, vp, i64, C.POINTER(i64)]),
        "mra_get_node_block": (C.c_int, [vp, i32, C.c_int, vp, i64, C.POINTER(i64), C.POINTER(i64)]),
        "mra_get_timers": (C.c_int, [vp, vp, C.c_int]),
        "mra_plan_set_option": (C.c_int, [vp, C.c_int, i64]),
        "mra_plan_get_option": (C.c_int, [vp, C.c_int, C.POINTER(i64)]),
        "mra_plan_prepare": (C.c_int, [vp, C.POINTER(i64)]),
        "mra_kernel_family_count": (C.c_int, []),
        "mra_get_kernel_stats": (C.c_int, [vp, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(dbl), C.POINTER(dbl)]),
        "mra_get_kernel_work": (C.c_int, [vp, C.c_int, vp, C.c_int]),
        "mra_device_synchronize": (C.c_int, [C.c_int]),
        "mra_plan_info": (C.c_int, [vp, vp, C.c_int]),
        "mra_comm_unique_id": (C.c_int, [C.c_char_p, C.c_int]),
        "mra_comm_init": (C.c_int, [vp, C.c_char_p, C.c_int, C.c_int]),
        "mra_plan_set_reduce_level": (C.c_int, [vp, C.c_int]),
        "mra_reduce_size": (C.c_int, [vp, C.POINTER(i64)]),
        "mra_reduce_export": (C.c_int, [vp, vp]),
        "mra_reduce_import": (C.c_int, [vp, vp]),
        "mra_tree_replay_2d": (C.c_int, [vp, i64, i32, i32, vp, C.POINTER(i32), C.POINTER(vp)]),
        "mra_tree_replay_2d_into": (C.c_int, [vp, i64, i32, i32, vp, C.POINTER(i32), i64, vp, vp, vp, vp, C.POINTER(vp)]),
        "mra_plan_create_replay_2d": (C.c_int, [vp, i64, i32, i32, vp, C.POINTER(i32), vp, dbl, C.c_int, i64, vp, vp, vp, vp,
                                                C.POINTER(vp), C.POINTER(vp)]),
        "mra_tree_sizes": (C.c_int, [vp, vp]),
        "mra_tree_export": (C.c_int, [vp] + [vp] * 15),
        "mra_tree_free": (C.c_int, [vp]),
        "mra_last_error": (C.c_char_p, [vp]),
        "mra_version": (C.c_char_p, []),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class HipPlan:
    """One MRA tree on one GPU.  Thin, stateful wrapper of the C ABI."""

    def __init__(self, topo, device: int = 0, _handle=None):
        self.lib = load_library()
        self.topo = topo
        self._h = C.c_void_p()
        if _handle is not None:                  # a plan the library has already built (create_with_replay)
            self._h = _handle
            self.P = int(topo.P)
            self.d = int(topo.d)
            return
        # keep the arrays alive (and of the exact dtypes the header declares) during the call
        arrs = dict(
            level_ptr=np.ascontiguousarray(topo.level_ptr, dtype=np.int64),
            node_row0=np.ascontiguousarray(topo.node_row0, dtype=np.int64),
            node_row1=np.ascontiguousarray(topo.node_row1, dtype=np.int64),
            node_leaf=np.ascontiguousarray(topo.node_leaf, dtype=np.uint8),
            node_parent=np.ascontiguousarray(topo.node_parent, dtype=np.int32),
            child_ptr=np.ascontiguousarray(topo.child_ptr, dtype=np.int32),
            child_list=np.ascontiguousarray(topo.child_list, dtype=np.int32),
            knot_ptr=np.ascontiguousarray(topo.knot_ptr, dtype=np.int64),
            knot_rows=np.ascontiguousarray(topo.knot_rows, dtype=np.int64),
            cw=np.ascontiguousarray(topo.cw, dtype=np.int32),
        )
        st = MraTopologyStruct(P=int(topo.P), d=int(topo.d), n_levels=int(topo.n_levels), n_nodes=int(topo.n_nodes),
                               **{k: _ptr(v) for k, v in arrs.items()})
        rc = self.lib.mra_plan_create(C.byref(self._h), C.byref(st), int(device))
        if rc != 0:
            msg = self.lib.mra_last_error(None)
            self._h = C.c_void_p()
            raise MraError(rc, msg.decode() if msg else "")
        self.P = int(topo.P)
        self.d = int(topo.d)

    # -- helpers -------------------------------------------------------------------------------
    def _check(self, rc):
        if rc != 0:
            msg = self.lib.mra_last_error(self._h)
            raise MraError(rc, msg.decode() if msg else "")

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self.lib.mra_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- inputs --------------------------------------------------------------------------------
    def _row_maps(self):
        if getattr(self, "_maps", None) is None:
            t = self.topo
            self._maps = (np.ascontiguousarray(t.src, dtype=np.int64), np.ascontiguousarray(t.perm, dtype=np.int64),
                          np.ascontiguousarray(t.in_leaf, dtype=np.uint8))
        return self._maps

    def set_locs(self, locs):
        """locs: the caller's N x d array; permuted/padded inside the library."""
        X = np.ascontiguousarray(locs, dtype=np.float64)
        if X.ndim == 1:
            X = X.reshape(-1, 1)
        if X.shape != (self.topo.N, self.d):
            raise ValueError("locs must be N x d")
        src, _, _ = self._row_maps()
        self._check(self.lib.mra_plan_set_locs_rows(self._h, _ptr(X), _ptr(src)))

    def set_obs(self, obs, R):
        y = np.ascontiguousarray(np.asarray(obs, dtype=np.float64).reshape(-1))
        if len(y) != self.topo.N:
            raise ValueError("obs must have N entries")
        src, perm, _ = self._row_maps()
        self._check(self.lib.mra_plan_set_obs_rows(self._h, _ptr(y), _ptr(src), _ptr(perm), float(R)))

    def set_kernel(self, kind, l, sig=1.0, scale=1.0, circular=False):
        par = np.array([l, sig, scale, 1.0 if circular else 0.0], dtype=np.float64)
        self._check(self.lib.mra_plan_set_kernel(self._h, int(kind), _ptr(par), 4))

    def set_kernel_host(self):
        """Covariance values will be supplied block by block (``set_cov_block``); call after ``set_obs``."""
        self._check(self.lib.mra_plan_set_kernel(self._h, MRA_KERNEL_HOST, None, 0))

    def set_cov_block(self, node, C, diag=None):
        C = np.ascontiguousarray(C, dtype=np.float64)
        d = None if diag is None else np.ascontiguousarray(diag, dtype=np.float64)
        self._check(self.lib.mra_plan_set_cov_block(self._h, int(node), _ptr(C), C.shape[0], C.shape[1],
                                                    None if d is None else _ptr(d)))

    def upload_host_cov(self, cov, locs, obs):
        """Evaluate an opaque ``cov`` (callable on location arrays, or a dense N x N matrix) block by block
        on the host - C(S_j, Q_j) for non-leaf nodes (pyMRA/MRANode.py:384), C(S_j, O_j) and C(x,x) for
        leaves - and hand the blocks to the library."""
        t = self.topo
        X = np.asarray(locs, dtype=np.float64)
        if X.ndim == 1:
            X = X.reshape(-1, 1)
        y = np.asarray(obs, dtype=np.float64).reshape(-1)
        dense = None if callable(cov) else np.asarray(cov, dtype=np.float64)
        if dense is not None and dense.shape != (t.N, t.N):
            raise ValueError("a dense cov must be N x N")

        def block(ra, rb):
            if dense is not None:
                return dense[np.ix_(ra, rb)]
            return np.asarray(cov(X[ra], X[rb]), dtype=np.float64)

        self.set_kernel_host()
        for i in range(t.n_nodes):
            rows = np.arange(t.node_row0[i], t.node_row1[i])
            ra = t.src[rows]
            if t.node_leaf[i]:
                real = t.perm[rows] >= 0
                obs_rows = rows[real & np.isfinite(np.where(real, y[ra], np.nan))]
                rb = t.src[obs_rows]
                C = block(ra, rb) if len(rb) else np.zeros((len(ra), 0))
                if dense is not None:
                    dg = np.diag(dense)[ra]
                else:
                    dg = np.diag(block(ra, ra))
                self.set_cov_block(i, C, dg)
            else:
                kq = t.knot_rows[t.knot_ptr[i]:t.knot_ptr[i + 1]]
                self.set_cov_block(i, block(ra, t.src[kq]))

    def set_option(self, option, value):
        self._check(self.lib.mra_plan_set_option(self._h, int(option), int(value)))

    def get_option(self, option):
        v = C.c_int64()
        self._check(self.lib.mra_plan_get_option(self._h, int(option), C.byref(v)))
        return int(v.value)

    def prepare(self):
        """Raise the dynamic-LDS limits of this plan's kernels on its device now; returns how many kernels are on record."""
        n = C.c_int64()
        self._check(self.lib.mra_plan_prepare(self._h, C.byref(n)))
        return int(n.value)

    # -- run / results -----------------------------------------------------------------------------
    def run(self, likelihood=True, predict=True, split=False):
        flags = (MRA_RUN_LIKELIHOOD if likelihood else 0) | (MRA_RUN_PREDICT if predict else 0) | (MRA_RUN_SPLIT if split else 0)
        self._check(self.lib.mra_run(self._h, flags))

    def resume(self):
        self._check(self.lib.mra_run_resume(self._h))

    def likelihood(self):
        d, u = C.c_double(), C.c_double()
        self._check(self.lib.mra_get_likelihood(self._h, C.byref(d), C.byref(u)))
        return d.value, u.value

    def predict(self, with_sd=False):
        """(mean[N], var[N]) in the caller's row order - with_sd: (mean, var, sd = sqrt(var)) -; rows outside every leaf report 0
        (see topology: rows that a partition drops, pyMRA/MRANode.py:222-228, 492-495)."""
        t = self.topo
        _, perm, in_leaf = self._row_maps()
        mean = np.empty(t.N)
        var = np.empty(t.N)
        sd = np.empty(t.N) if with_sd else None
        self._check(self.lib.mra_get_predict_rows_sd(self._h, _ptr(perm), _ptr(in_leaf), int(t.N), _ptr(mean), _ptr(var),
                                                     None if sd is None else _ptr(sd)))
        return (mean, var, sd) if with_sd else (mean, var)

    def buffer(self, what):
        n = C.c_int64()
        self._check(self.lib.mra_get_buffer(self._h, what, None, 0, C.byref(n)))
        out = np.empty(n.value)
        self._check(self.lib.mra_get_buffer(self._h, what, _ptr(out), n.value, C.byref(n)))
        return out

    def node_block(self, node, what):
        """Raw per-node device block (include/mra_hip.h: MRA_BLOCK_*), as a 2-D array."""
        nr, nc = C.c_int64(), C.c_int64()
        self._check(self.lib.mra_get_node_block(self._h, int(node), int(what), None, 0, C.byref(nr), C.byref(nc)))
        out = np.empty((nr.value, nc.value))
        if out.size:
            self._check(self.lib.mra_get_node_block(self._h, int(node), int(what), _ptr(out), out.size, C.byref(nr), C.byref(nc)))
        return out

    def timers(self):
        out = np.zeros(5)
        k = self.lib.mra_get_timers(self._h, _ptr(out), 5)
        names = ["prior_ms", "leaf_ms", "fronts_ms", "predict_ms", "total_ms"]
        return dict(zip(names[:k], out[:k]))

    def kernel_stats(self):
        res = []
        for k in range(self.lib.mra_kernel_family_count()):
            name = C.create_string_buffer(128)
            n, ms, fl = C.c_int(), C.c_double(), C.c_double()
            self._check(self.lib.mra_get_kernel_stats(self._h, k, name, 128, C.byref(n), C.byref(ms), C.byref(fl)))
            w = np.zeros(4)
            self._check(self.lib.mra_get_kernel_work(self._h, k, _ptr(w), 4))
            res.append(dict(name=name.value.decode(), launches=n.value, ms=ms.value, flops=fl.value, flops_exec=float(w[1]), bytes=float(w[2])))
        return res

    def info(self):
        out = np.zeros(8, dtype=np.int64)
        self.lib.mra_plan_info(self._h, _ptr(out), 8)
        keys = ["P", "ldw", "Ka", "n_leaves", "bytes_W", "bytes_panels", "bytes_Gt", "n_nodes"]
        return dict(zip(keys, (int(v) for v in out)))

    # -- multi-GPU ---------------------------------------------------------------------------------
    def set_reduce_level(self, level):
        self._check(self.lib.mra_plan_set_reduce_level(self._h, int(level)))

    def reduce_export(self):
        n = C.c_int64()
        self._check(self.lib.mra_reduce_size(self._h, C.byref(n)))
        out = np.empty(n.value)
        self._check(self.lib.mra_reduce_export(self._h, _ptr(out)))
        return out

    def reduce_import(self, buf):
        b = np.ascontiguousarray(buf, dtype=np.float64)
        self._check(self.lib.mra_reduce_import(self._h, _ptr(b)))

    def comm_init(self, uid: bytes, n_ranks: int, rank: int):
        self._check(self.lib.mra_comm_init(self._h, uid, int(n_ranks), int(rank)))


def create_with_replay(locs, r, M, J, obs, R, device: int = 0):
    """MRATree.__init__ for large 2-D trees in one library call (mra_plan_create_replay_2d): the tree replay and, beside its
    sequential knot draws, the plan with locations and observations already uploaded.  Consumes NumPy's global RNG exactly like
    pymra_amd.topology.build_topology.  Returns (HipPlan, Topology), or None when the tree does not follow the large-2-D rules
    (RNG state untouched: the caller takes the general path)."""
    from .topology import topology_from_tree
    coords = np.ascontiguousarray(np.asarray(locs, dtype=np.float64))
    if coords.ndim != 2 or coords.shape[1] != 2 or J != 4 or M < 1:
        return None
    y = np.ascontiguousarray(np.asarray(obs, dtype=np.float64).reshape(-1))
    N = len(coords)
    if len(y) != N:
        raise ValueError("obs must have N entries")
    state = np.random.get_state()
    if state[0] != "MT19937":
        return None
    lib = load_library()
    key = np.ascontiguousarray(state[1], dtype=np.uint32).copy()
    pos = C.c_int32(int(state[2]))
    cap = N + 15 * 4 ** int(M)
    perm, src = np.empty(cap, np.int64), np.empty(cap, np.int64)
    in_leaf, knot_rows = np.empty(cap, np.bool_), np.empty(N, np.int64)
    tree, ph = C.c_void_p(), C.c_void_p()
    rc = lib.mra_plan_create_replay_2d(_ptr(coords), N, int(r), int(M), _ptr(key), C.byref(pos), _ptr(y), float(R), int(device), cap,
                                       _ptr(perm), _ptr(src), _ptr(in_leaf), _ptr(knot_rows), C.byref(tree), C.byref(ph))
    if rc == 1:
        return None
    if rc != 0:
        msg = lib.mra_last_error(None)
        raise MraError(rc, msg.decode() if msg else "")
    try:
        topo = topology_from_tree(lib, tree, N, int(r), int(M), int(J), perm, src, in_leaf, knot_rows)
    finally:
        lib.mra_tree_free(tree)
    np.random.set_state((state[0], key, int(pos.value), state[3], state[4]))
    return HipPlan(topo, device, _handle=ph), topo


def comm_unique_id() -> bytes:
    lib = load_library()
    buf = C.create_string_buffer(128)
    rc = lib.mra_comm_unique_id(buf, 128)
    if rc != 0:
        raise MraError(rc, (lib.mra_last_error(None) or b"").decode())
    return buf.raw


def eval_kernel(kind, l, sig, scale, D):
    """Device evaluation of a stationary kernel on an array of distances (tests)."""
    lib = load_library()
    D = np.ascontiguousarray(D, dtype=np.float64).ravel()
    out = np.empty_like(D)
    par = np.array([l, sig, scale], dtype=np.float64)
    rc = lib.mra_eval_kernel(int(kind), _ptr(par), 3, _ptr(D), D.size, _ptr(out))
    if rc != 0:
        raise MraError(rc, (lib.mra_last_error(None) or b"").decode())
    return out


def device_count() -> int:
    return int(load_library().mra_device_count())


def device_synchronize(device: int = 0) -> None:
    """hipDeviceSynchronize on `device` through the library (no other GPU runtime needed for a barrier)."""
    rc = load_library().mra_device_synchronize(int(device))
    if rc != 0:
        raise MraError(rc, "hipDeviceSynchronize failed")


def release_cached_memory() -> None:
    """Return the device blocks the library keeps from destroyed plans (reused by later plans of the same sizes) to the driver."""
    load_library().mra_release_cached_memory()
