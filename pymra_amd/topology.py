"""Host-side tree builder: knots, partitions and the leaf-ordered row layout.

Pure index arithmetic (NumPy, host).  It replays the *rules* of the reference's
recursive constructor so that the same inputs and the same global NumPy RNG state
give the same tree:

* resolution / fan-out defaults and the M clip ......... pyMRA/MRATree.py:27-59
* leaf test and early leaves .............................. pyMRA/MRANode.py:34-45
* knot choice: 1-D nearest-rank quantiles, 2-D random
  draw (global RNG, DFS pre-order) or KMeans ............ pyMRA/MRANode.py:179-205
* candidate bookkeeping ("notKnots") ...................... pyMRA/MRANode.py:53, 83
* partition: quadrants / terciles / knot-boundary / KMeans  pyMRA/MRANode.py:213-242, 289-340

Differences in mechanism (not in result): membership is tracked by global row index
instead of by coordinate value (the reference uses numpy_indexed.contains on the
coordinates, which is the same thing for distinct locations), and the output is a flat,
device-friendly description instead of a graph of Python objects:

* every node's rows are one contiguous range of a *permuted, padded* row order
  (depth-first leaf order; each leaf - and each group of rows that a partition drops,
  see ``orphans`` - is padded to a multiple of ``ROW_ALIGN`` rows with phantom rows),
* nodes are numbered level by level (``level_ptr``), children are contiguous.

No reference code is imported or needed here.
"""
from __future__ import annotations

import logging
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np
from collections.abc import Sequence

logger = logging.getLogger("pyMRA.MRATree")

ROW_ALIGN = 16          # MFMA f64 tile height: every leaf row range is a multiple of this
COL_ALIGN = 16          # MFMA f64 tile width: every per-level knot block is padded to this


# ----------------------------------------------------------------------------------------
#  M / J / critDepth resolution (pyMRA/MRATree.py:27-59)
# ----------------------------------------------------------------------------------------
def resolve_tree_shape(N: int, d: int, r: int, M: int = -1, J: int = -1):
    """Return (M, J) exactly as the reference constructor resolves them.

    J<0: 2-D -> 4; 1-D -> the reference evaluates ``self.J == r+1`` (a comparison on an
    attribute that does not exist) and raises AttributeError (MRATree.py:31-33); we raise
    the same exception type with a readable message.
    maxM = int(log(N*J/r + 1)/log(J)) - 1 (MRATree.py:41-43); M<0 -> maxM; M>maxM -> clip
    with a warning (MRATree.py:44-50).
    """
    if J < 0:
        if d == 2:
            J = 4
        else:
            raise AttributeError("'MRATree' object has no attribute 'J' "
                                 "(pass J explicitly for %d-D locations)" % d)
    num = np.log(N * J / r + 1)
    denom = np.log(J)
    if denom == 0:
        raise OverflowError("cannot convert float infinity to integer")   # J == 1, as reference
    maxM = int(num / denom) - 1
    if M < 0:
        M = maxM
    elif M > maxM:
        logger.warning("The number of resolutions M=%d you requested is to large for your "
                       "grid. Setting M:=%d" % (M, maxM))
        M = maxM
    return int(M), int(J)


# ----------------------------------------------------------------------------------------
#  recursive replay
# ----------------------------------------------------------------------------------------
@dataclass
class _TNode:
    ident: str
    level: int
    rows: np.ndarray                 # global row indices, ascending (== reference's locs order)
    knots: np.ndarray                # global row indices of the knots, ascending
    leaf: bool
    children: List["_TNode"] = field(default_factory=list)
    child_local: List[np.ndarray] = field(default_factory=list)   # reference's inds[chID]


def _nearest_rank_percentile(sorted_vals: np.ndarray, q: float) -> float:
    return float(np.percentile(sorted_vals, q, method="nearest"))


def _knots_1d(coords: np.ndarray, rows: np.ndarray, cand: np.ndarray, r: int) -> np.ndarray:
    """Quantile knots in 1-D (MRANode.py:181-187): the r interior nearest-rank percentiles
    of the candidate coordinates; a row is a knot when its coordinate equals one of them
    (so repeated percentiles give fewer than r knots)."""
    c1 = coords[cand, 0]
    vals = [np.percentile(c1, 100.0 * i / (r + 1), method="nearest") for i in range(r + 2)][1:-1]
    mask = np.isin(coords[rows, 0], np.asarray(vals))
    return rows[mask]


def _knots_kmeans(coords: np.ndarray, rows: np.ndarray, cand: np.ndarray, r: int) -> np.ndarray:
    """Small-node knots (MRANode.py:195-201): KMeans(r, random_state=0) on the candidates,
    then the candidate nearest to each centroid (first minimum)."""
    from sklearn.cluster import KMeans
    from scipy.spatial.distance import cdist
    pts = coords[cand]
    km = KMeans(n_clusters=r, random_state=0).fit(pts)
    D = cdist(pts, km.cluster_centers_)
    picked = np.unique(cand[np.argmin(D, axis=0)])
    return picked


def _splits_geometric(coords: np.ndarray, rows: np.ndarray) -> List[np.ndarray]:
    """Partition of a node with more than 100 rows (MRANode.py:213-242), as *local* index
    arrays.  1-D: strict terciles about np.percentile(.,(33,66)) - rows equal to a percentile
    belong to no child.  2-D: quadrants about the coordinate means, order (<=,<=),(<=,>),
    (>,<=),(>,>)."""
    X = coords[rows]
    if X.shape[1] == 1:
        p = np.percentile(X, (33, 66))
        x = X[:, 0]
        return [np.where(x < p[0])[0],
                np.where(np.logical_and(x > p[0], x < p[1]))[0],
                np.where(x > p[1])[0]]
    m = np.mean(X, axis=0)
    lx = X[:, 0] <= m[0]
    ly = X[:, 1] <= m[1]
    return [np.where(lx & ly)[0], np.where(lx & ~ly)[0], np.where(~lx & ly)[0], np.where(~lx & ~ly)[0]]


def _splits_small(coords: np.ndarray, rows: np.ndarray, knots: np.ndarray, newcand: np.ndarray,
                  Jeff: int) -> List[np.ndarray]:
    """Partition of a node with at most 100 rows (MRANode.py:289-340), local index arrays.

    1-D with Jeff == #knots+1 and enough rows: cut at the knots (each knot starts the next
    child).  Otherwise KMeans(min(Jeff, #newcand), random_state=0) on the remaining candidates;
    every other row of the node (own and ancestors' knots) joins the nearest cluster centre;
    empty clusters are skipped; 1-D children are ordered by their first row."""
    N = len(rows)
    klocal = np.searchsorted(rows, knots)
    rk = len(klocal)
    if Jeff == rk + 1 and coords.shape[1] == 1 and N >= Jeff + rk:
        return [a for a in np.split(np.arange(N), klocal)]
    from sklearn.cluster import KMeans
    from scipy.spatial.distance import cdist
    nk_local = np.searchsorted(rows, newcand)
    pts = coords[newcand]
    ncl = min(Jeff, len(newcand))
    km = KMeans(n_clusters=ncl, random_state=0).fit(pts)
    labels = km.labels_
    other_local = np.setdiff1d(np.arange(N), nk_local)
    out = []
    if len(other_local):
        D = cdist(coords[rows[other_local]], km.cluster_centers_)
        olab = np.argmin(D, axis=1)
    else:
        olab = np.zeros(0, dtype=int)
    for j in range(Jeff):
        inds = np.sort(np.hstack((other_local[olab == j], nk_local[labels == j]))).astype(np.int64)
        if len(inds):
            out.append(inds)
    if coords.shape[1] == 1:
        out = sorted(out, key=lambda a: np.min(a))
    return out


def _grow(coords, rows, is_cand, levels_left, level, ident, r, J, flat: list) -> _TNode:
    """One node of the replay.  ``is_cand`` is a global boolean array over caller rows: True while a
    row has not been used as a knot by this node's ancestors (the reference's ``notKnots``,
    MRANode.py:53, 83); a node's candidates are ``rows[is_cand[rows]]`` (ascending, like the
    reference's)."""
    cand = rows[is_cand[rows]]
    leaf = (levels_left == 0)
    n_c = len(cand)
    splitting = (not leaf) and n_c > max(r, J)
    if splitting:
        if coords.shape[1] == 1:
            knots = _knots_1d(coords, rows, cand, r)
        elif n_c > 1e2:
            # MRANode.py:191-193: consumes the *global* NumPy RNG, in DFS pre-order
            # (an int population draws the same permutation as the reference's np.arange(n_c))
            pick = np.random.choice(n_c, size=r, replace=False)
            knots = np.sort(cand[pick])
        else:
            knots = _knots_kmeans(coords, rows, cand, r)
    else:
        knots = cand
        leaf = True
    node = _TNode(ident, level, rows, knots, leaf)
    flat.append(node)
    if splitting:
        is_cand[knots] = False            # stays False for the whole subtree (and nobody else owns these rows)
        if len(rows) > 1e2:
            parts = _splits_geometric(coords, rows)
        else:
            newcand = rows[is_cand[rows]]
            Jeff = min(J, len(newcand))
            parts = _splits_small(coords, rows, knots, newcand, Jeff)
        for j, loc_idx in enumerate(parts):
            ch_rows = rows[loc_idx]
            node.child_local.append(np.asarray(loc_idx, dtype=np.int64))
            node.children.append(_grow(coords, ch_rows, is_cand, levels_left - 1, level + 1,
                                       ident + str(j + 1), r, J, flat))
    return node


# ----------------------------------------------------------------------------------------
#  flat, padded, leaf-ordered description
# ----------------------------------------------------------------------------------------
@dataclass
class Topology:
    """Flat description of one MRA tree (all arrays host NumPy).

    Row space: ``P`` padded rows in depth-first leaf order; ``perm[p]`` is the caller's row
    index of padded row p, or -1 for a phantom (padding) row.  ``src[p]`` is a valid caller
    row for *every* p (phantoms copy a real row of the same leaf so that kernel evaluations
    stay finite; they are never observed and never reported).

    Nodes are numbered level by level: level m owns nodes ``level_ptr[m]:level_ptr[m+1]``.
    """
    N: int
    d: int
    M: int
    J: int
    r: int
    P: int
    perm: np.ndarray            # int64[P]
    src: np.ndarray             # int64[P]
    in_leaf: np.ndarray         # bool[P]  real row that belongs to some leaf (else: orphan/phantom)
    n_nodes: int
    n_levels: int               # deepest level + 1
    level_ptr: np.ndarray       # int64[n_levels+1]
    node_level: np.ndarray      # int32[n]
    node_row0: np.ndarray       # int64[n]
    node_row1: np.ndarray       # int64[n]   (padded range end)
    node_leaf: np.ndarray       # bool[n]
    node_parent: np.ndarray     # int32[n]   (-1 root)
    child_ptr: np.ndarray       # int32[n+1] children are nodes child_ptr[i]..child_ptr[i+1]-1 of child_list
    child_list: np.ndarray      # int32[.]
    knot_ptr: np.ndarray        # int64[n+1]
    knot_rows: np.ndarray       # int64[.]   padded-row positions of each node's knots, in the
                                #            reference's column order (ascending caller index)
    node_ident: List[str]
    cw: np.ndarray              # int32[n_levels] knot-block width of the NON-LEAF nodes of a level,
                                #                 padded to COL_ALIGN (0 if the level has none)
    order_preorder: np.ndarray  # int32[n] node numbers in the reference's construction order

    def rank(self, i: int) -> int:
        return int(self.knot_ptr[i + 1] - self.knot_ptr[i])

    @property
    def nodes_per_level(self):
        return np.diff(self.level_ptr)


class _LazyIdents(Sequence):
    """Node names of a native quadtree replay ("r", "r1", "r13", ...: the reference's Node.ID, pyMRA/MRANode.py:26-28), built
    on first use: diagnostics and tests read them, the inference path never does, and the Python loop over all nodes costs
    milliseconds at 1024^2."""

    def __init__(self, parent, level, level_ptr):
        self._parent, self._level, self._lp, self._ids = parent, level, level_ptr, None

    def _build(self):
        if self._ids is None:
            n = len(self._parent)
            ids = [""] * n
            if n:
                ids[0] = "r"
            for i in range(1, n):
                ids[i] = ids[self._parent[i]] + str(int((i - self._lp[self._level[i]]) % 4) + 1)
            self._ids = ids
        return self._ids

    def __len__(self):
        return len(self._parent)

    def __getitem__(self, i):
        return self._build()[i]

    def __eq__(self, other):
        return list(self) == list(other)


def _replay_native(coords: np.ndarray, r: int, M: int, J: int) -> Optional[Topology]:
    """Large 2-D trees: the replay runs in libmra_hip.so (csrc/mra_topology.h; host code, no GPU needed),
    driven by and returning NumPy's global MT19937 state.  None when the library is not built or the tree
    does not follow the large-2-D rules (then the Python replay below does the job; results are identical,
    tests/test_topology.py)."""
    if coords.shape[1] != 2 or J != 4 or M < 1:
        return None
    try:
        import ctypes as C
        from .plan import load_library
        lib = load_library()
    except Exception:
        return None
    state = np.random.get_state()
    if state[0] != "MT19937":
        return None
    key = np.ascontiguousarray(state[1], dtype=np.uint32).copy()
    pos = C.c_int32(int(state[2]))
    handle = C.c_void_p()
    xy = np.ascontiguousarray(coords, dtype=np.float64)
    N = len(xy)
    cap = N + 15 * 4 ** int(M)                          # every leaf is padded to a multiple of 16 rows
    perm, src = np.empty(cap, np.int64), np.empty(cap, np.int64)
    in_leaf, knot_rows = np.empty(cap, np.bool_), np.empty(N, np.int64)
    vp = lambda v: v.ctypes.data_as(C.c_void_p)
    rc = lib.mra_tree_replay_2d_into(vp(xy), N, int(r), int(M), vp(key), C.byref(pos), cap, vp(perm), vp(src), vp(in_leaf),
                                     vp(knot_rows), C.byref(handle))
    if rc != 0:
        return None
    try:
        topo = topology_from_tree(lib, handle, N, r, M, J, perm, src, in_leaf, knot_rows)
    finally:
        lib.mra_tree_free(handle)
    np.random.set_state((state[0], key, int(pos.value), state[3], state[4]))
    return topo


def topology_from_tree(lib, handle, N, r, M, J, perm, src, in_leaf, knot_rows) -> Topology:
    """The Topology of a native replay: the node arrays exported from the library's tree handle, the four long arrays
    (already written into the caller's buffers by the replay) cut to their sizes."""
    import ctypes as C
    vp = lambda v: v.ctypes.data_as(C.c_void_p)
    sz = np.zeros(5, dtype=np.int64)
    lib.mra_tree_sizes(handle, vp(sz))
    P, n_nodes, n_levels, n_child, n_knots = (int(v) for v in sz)
    a = dict(perm=None, src=None, in_leaf=None,
             level_ptr=np.empty(n_levels + 1, np.int64), node_level=np.empty(n_nodes, np.int32),
             row0=np.empty(n_nodes, np.int64), row1=np.empty(n_nodes, np.int64), leaf=np.empty(n_nodes, np.uint8),
             parent=np.empty(n_nodes, np.int32), child_ptr=np.empty(n_nodes + 1, np.int32),
             child_list=np.empty(n_child, np.int32), knot_ptr=np.empty(n_nodes + 1, np.int64),
             knot_rows=None, cw=np.empty(n_levels, np.int32), pre=np.empty(n_nodes, np.int32))
    lib.mra_tree_export(handle, *[None if v is None else vp(v) for v in a.values()])
    a.update(perm=perm[:P], src=src[:P], in_leaf=in_leaf[:P], knot_rows=knot_rows[:n_knots])
    idents = _LazyIdents(a["parent"], a["node_level"], a["level_ptr"])
    return Topology(N=N, d=2, M=M, J=J, r=r, P=P, perm=a["perm"], src=a["src"], in_leaf=a["in_leaf"],
                    n_nodes=n_nodes, n_levels=n_levels, level_ptr=a["level_ptr"], node_level=a["node_level"],
                    node_row0=a["row0"], node_row1=a["row1"], node_leaf=a["leaf"].astype(bool), node_parent=a["parent"],
                    child_ptr=a["child_ptr"], child_list=a["child_list"], knot_ptr=a["knot_ptr"], knot_rows=a["knot_rows"],
                    node_ident=idents, cw=a["cw"], order_preorder=a["pre"])


def build_topology(locs: np.ndarray, r: int, M: int, J: int, native: bool = True) -> Topology:
    """Replay the reference's tree construction for ``locs`` (N x d, d in {1,2}).

    M and J must already be resolved (``resolve_tree_shape``).  Consumes the global NumPy
    RNG exactly like the reference does (one ``np.random.choice`` per 2-D node with more
    than 100 candidates, depth-first pre-order)."""
    coords = np.ascontiguousarray(np.asarray(locs, dtype=np.float64))
    if coords.ndim == 1:
        coords = coords.reshape(-1, 1)
    N, d = coords.shape
    if native:
        topo = _replay_native(coords, r, M, J)
        if topo is not None:
            return topo
    flat: List[_TNode] = []
    all_rows = np.arange(N, dtype=np.int64)
    import sys
    sys.setrecursionlimit(max(10000, sys.getrecursionlimit()))
    root = _grow(coords, all_rows, np.ones(N, dtype=bool), M, 0, "r", r, J, flat)

    # ---- leaf-ordered padded row layout --------------------------------------------------
    perm_chunks: List[np.ndarray] = []
    src_chunks: List[np.ndarray] = []
    inleaf_chunks: List[np.ndarray] = []
    cursor = [0]
    span = {}

    def _emit(real_rows: np.ndarray, is_leaf_rows: bool):
        n = len(real_rows)
        npad = (-n) % ROW_ALIGN
        p = np.concatenate([real_rows, np.full(npad, -1, dtype=np.int64)])
        filler = real_rows[0] if n else 0
        s = np.concatenate([real_rows, np.full(npad, filler, dtype=np.int64)])
        perm_chunks.append(p)
        src_chunks.append(s)
        fl = np.zeros(n + npad, dtype=bool)
        fl[:n] = is_leaf_rows
        inleaf_chunks.append(fl)
        cursor[0] += n + npad

    def _layout(nd: _TNode):
        start = cursor[0]
        if nd.leaf:
            _emit(nd.rows, True)
        else:
            covered = np.zeros(len(nd.rows), dtype=bool)
            for ch, li in zip(nd.children, nd.child_local):
                covered[li] = True
                _layout(ch)
            orphans = nd.rows[~covered]
            if len(orphans):
                _emit(orphans, False)
        span[id(nd)] = (start, cursor[0])

    import sys
    sys.setrecursionlimit(max(10000, sys.getrecursionlimit()))
    _layout(root)
    P = cursor[0]
    perm = np.concatenate(perm_chunks) if perm_chunks else np.zeros(0, dtype=np.int64)
    src = np.concatenate(src_chunks) if src_chunks else np.zeros(0, dtype=np.int64)
    in_leaf = np.concatenate(inleaf_chunks) if inleaf_chunks else np.zeros(0, dtype=bool)
    pos_of = np.full(N, -1, dtype=np.int64)
    real = perm >= 0
    pos_of[perm[real]] = np.nonzero(real)[0]

    # ---- level-major node numbering (children of a node stay contiguous) --------------------
    n_nodes = len(flat)
    n_levels = max(nd.level for nd in flat) + 1
    by_level: List[List[_TNode]] = [[] for _ in range(n_levels)]
    # breadth-first keeps siblings adjacent and parents ordered
    frontier = [root]
    while frontier:
        nxt = []
        for nd in frontier:
            by_level[nd.level].append(nd)
            nxt.extend(nd.children)
        frontier = nxt
    number = {}
    k = 0
    level_ptr = np.zeros(n_levels + 1, dtype=np.int64)
    for m in range(n_levels):
        for nd in by_level[m]:
            number[id(nd)] = k
            k += 1
        level_ptr[m + 1] = k

    node_level = np.zeros(n_nodes, dtype=np.int32)
    node_row0 = np.zeros(n_nodes, dtype=np.int64)
    node_row1 = np.zeros(n_nodes, dtype=np.int64)
    node_leaf = np.zeros(n_nodes, dtype=bool)
    node_parent = np.full(n_nodes, -1, dtype=np.int32)
    child_ptr = np.zeros(n_nodes + 1, dtype=np.int32)
    knot_ptr = np.zeros(n_nodes + 1, dtype=np.int64)
    idents = [""] * n_nodes
    child_list: List[int] = []
    knot_chunks: List[np.ndarray] = [None] * n_nodes
    ordered = [nd for m in range(n_levels) for nd in by_level[m]]
    for i, nd in enumerate(ordered):
        node_level[i] = nd.level
        node_row0[i], node_row1[i] = span[id(nd)]
        node_leaf[i] = nd.leaf
        idents[i] = nd.ident
        for ch in nd.children:
            cj = number[id(ch)]
            node_parent[cj] = i
            child_list.append(cj)
        child_ptr[i + 1] = len(child_list)
        knot_chunks[i] = pos_of[nd.knots]
        knot_ptr[i + 1] = knot_ptr[i] + len(nd.knots)
    knot_rows = np.concatenate(knot_chunks) if n_nodes else np.zeros(0, dtype=np.int64)

    cw = np.zeros(n_levels, dtype=np.int32)
    for m in range(n_levels):
        ranks = [len(nd.knots) for nd in by_level[m] if not nd.leaf]
        if ranks:
            cw[m] = -(-max(ranks) // COL_ALIGN) * COL_ALIGN
    preorder = np.array([number[id(nd)] for nd in flat], dtype=np.int32)

    return Topology(N=N, d=d, M=M, J=J, r=r, P=P, perm=perm, src=src, in_leaf=in_leaf,
                    n_nodes=n_nodes, n_levels=n_levels, level_ptr=level_ptr,
                    node_level=node_level, node_row0=node_row0, node_row1=node_row1,
                    node_leaf=node_leaf, node_parent=node_parent, child_ptr=child_ptr,
                    child_list=np.asarray(child_list, dtype=np.int32), knot_ptr=knot_ptr,
                    knot_rows=knot_rows, node_ident=idents, cw=cw, order_preorder=preorder)
