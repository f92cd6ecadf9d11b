"""Subtree sharding across GPUs: one process per GPU, one all-reduce.

The reference's only parallel mode forks a process per child subtree at ``critDepth`` and pickles
the finished ``Node`` objects back through pipes (pyMRA/MRANode.py:64-65, 90-104, 114-115).  The
structural fact it uses - siblings meet only in their parent's sums (MRANode.py:434-440,
466-468) - is what is kept here:

* pick the shard level ``s`` = first level with at least ``n_ranks`` nodes; every rank owns a
  contiguous block of the level-``s`` subtrees (all their rows, nodes and leaves),
* every rank also keeps ALL nodes of the levels above ``s`` (they are tiny: r knots each) together
  with the rows of their knots, so the upper part of the prior is recomputed redundantly and
  bit-identically everywhere - no communication for the prior,
* after the leaves and the levels >= ``s`` are done, the level ``s-1`` fronts (sums of the
  children's Schur blocks, without their identity) plus one scalar (the rank-local log-det sum)
  are all-reduced ONCE; every rank then finishes the levels above redundantly and applies the
  predictive updates to its own rows.

``shard_topology`` only does index bookkeeping and returns an ordinary ``Topology`` whose
``perm`` still refers to the caller's rows; rows that are present only because an upper node needs
them as knots are flagged ``in_leaf = False`` and are never reported.
"""
from __future__ import annotations

from dataclasses import replace
from typing import List, Tuple

import numpy as np

from .topology import ROW_ALIGN, Topology


def choose_shard_level(topo: Topology, n_ranks: int) -> int:
    for m in range(topo.n_levels):
        if topo.level_ptr[m + 1] - topo.level_ptr[m] >= n_ranks:
            return m
    raise ValueError("tree has no level with %d nodes" % n_ranks)


def shard_topology(topo: Topology, n_ranks: int, rank: int) -> Tuple[Topology, int]:
    """Local topology of ``rank`` and the reduce level (shard level - 1; -1 when not sharded)."""
    if n_ranks <= 1:
        return topo, -1
    s = choose_shard_level(topo, n_ranks)
    if s == 0:
        return topo, -1
    if np.any(topo.node_leaf[:topo.level_ptr[s]]):
        raise NotImplementedError("leaves above the shard level are not supported by the sharded path")
    lvl_nodes = np.arange(topo.level_ptr[s], topo.level_ptr[s + 1])
    blocks = np.array_split(lvl_nodes, n_ranks)
    mine = set(int(i) for i in blocks[rank])

    kids = lambda i: [int(c) for c in topo.child_list[topo.child_ptr[i]:topo.child_ptr[i + 1]]]
    keep = np.zeros(topo.n_nodes, dtype=bool)
    keep[:topo.level_ptr[s]] = True
    stack = list(mine)
    while stack:
        i = stack.pop()
        keep[i] = True
        stack.extend(kids(i))

    # ---- new row layout: DFS over kept nodes; old padded ranges of owned subtrees are copied verbatim
    new_perm: List[np.ndarray] = []
    new_src: List[np.ndarray] = []
    new_inleaf: List[np.ndarray] = []
    cursor = [0]
    new_span = {}
    old_to_new_row = {}          # old padded row -> new padded row (only for rows that are kept)

    def emit_block(old_rows: np.ndarray, as_leaf_rows: np.ndarray):
        n = len(old_rows)
        npad = (-n) % ROW_ALIGN
        base = cursor[0]
        for t, p in enumerate(old_rows):
            old_to_new_row.setdefault(int(p), base + t)
        perm = np.concatenate([topo.perm[old_rows], np.full(npad, -1, dtype=np.int64)])
        filler = topo.src[old_rows[0]] if n else 0
        src = np.concatenate([topo.src[old_rows], np.full(npad, filler, dtype=np.int64)])
        fl = np.concatenate([as_leaf_rows, np.zeros(npad, dtype=bool)])
        new_perm.append(perm); new_src.append(src); new_inleaf.append(fl)
        cursor[0] += n + npad

    def layout(i):
        start = cursor[0]
        if topo.node_level[i] >= s:
            rows = np.arange(topo.node_row0[i], topo.node_row1[i])
            emit_block(rows, topo.in_leaf[rows])                 # owned subtree: verbatim, already aligned
            new_span[i] = (start, cursor[0])
            base_old = int(topo.node_row0[i])
            stack2 = kids(i)
            while stack2:
                c = stack2.pop()
                new_span[c] = (start + int(topo.node_row0[c]) - base_old, start + int(topo.node_row1[c]) - base_old)
                stack2.extend(kids(c))
            return
        for c in kids(i):
            if keep[c]:
                layout(c)
        # rows this upper node needs that are not inside a kept child: its knots (as orphan rows)
        kq = topo.knot_rows[topo.knot_ptr[i]:topo.knot_ptr[i + 1]]
        extra = np.array([int(q) for q in kq if int(q) not in old_to_new_row], dtype=np.int64)
        if len(extra):
            emit_block(extra, np.zeros(len(extra), dtype=bool))
        new_span[i] = (start, cursor[0])

    import sys
    sys.setrecursionlimit(max(10000, sys.getrecursionlimit()))
    layout(0)
    P = cursor[0]

    # ---- renumber kept nodes level by level
    old_ids = [int(i) for i in range(topo.n_nodes) if keep[i]]
    new_id = {o: n for n, o in enumerate(old_ids)}
    n_nodes = len(old_ids)
    level_ptr = np.zeros(topo.n_levels + 1, dtype=np.int64)
    for o in old_ids:
        level_ptr[topo.node_level[o] + 1] += 1
    level_ptr = np.cumsum(level_ptr)
    node_level = np.array([topo.node_level[o] for o in old_ids], dtype=np.int32)
    row0 = np.array([new_span[o][0] for o in old_ids], dtype=np.int64)
    row1 = np.array([new_span[o][1] for o in old_ids], dtype=np.int64)
    leaf = np.array([topo.node_leaf[o] for o in old_ids], dtype=bool)
    parent = np.array([new_id[int(topo.node_parent[o])] if topo.node_parent[o] >= 0 else -1 for o in old_ids], dtype=np.int32)
    child_ptr = np.zeros(n_nodes + 1, dtype=np.int32)
    child_list: List[int] = []
    knot_ptr = np.zeros(n_nodes + 1, dtype=np.int64)
    knot_chunks = []
    for n, o in enumerate(old_ids):
        for c in kids(o):
            if keep[c]:
                child_list.append(new_id[c])
        child_ptr[n + 1] = len(child_list)
        kq = topo.knot_rows[topo.knot_ptr[o]:topo.knot_ptr[o + 1]]
        if topo.node_level[o] >= s:
            base_old, base_new = int(topo.node_row0[o]), new_span[o][0]
            # rows of an owned subtree keep their relative position
            top = o
            while topo.node_level[top] > s:
                top = int(topo.node_parent[top])
            shift = new_span[top][0] - int(topo.node_row0[top])
            knot_chunks.append(kq + shift)
        else:
            knot_chunks.append(np.array([old_to_new_row[int(q)] for q in kq], dtype=np.int64))
        knot_ptr[n + 1] = knot_ptr[n] + len(kq)
    local = replace(
        topo, P=P, perm=np.concatenate(new_perm), src=np.concatenate(new_src), in_leaf=np.concatenate(new_inleaf),
        n_nodes=n_nodes, level_ptr=level_ptr, node_level=node_level, node_row0=row0, node_row1=row1,
        node_leaf=leaf, node_parent=parent, child_ptr=child_ptr, child_list=np.asarray(child_list, dtype=np.int32),
        knot_ptr=knot_ptr, knot_rows=np.concatenate(knot_chunks) if knot_chunks else np.zeros(0, dtype=np.int64),
        node_ident=[topo.node_ident[o] for o in old_ids], order_preorder=np.arange(n_nodes, dtype=np.int32))
    return local, s - 1


def sharded_minimize(evaluate, x0, rank: int, bcast, transform=None, **options):
    """Nelder-Mead over one kernel parameter with a SHARDED objective (BASELINE config 5; the reference's pattern is
    scipy.optimize.minimize(..., method='nelder-mead') around a new MRATree per call, pyMRA/README.md:96-104,
    pyMRA/tests/test-param-est.py:81-123).

    ``evaluate(kappa)`` is a collective: every rank calls it with the same ``kappa`` (it runs the rank's local plan up to
    the one front all-reduce and finishes the upper levels redundantly), and every rank gets the same -2 loglik back.
    Only rank 0 runs the optimiser; before every objective call it broadcasts ``[1, kappa]`` so that all ranks evaluate
    the same point, and ``[0, kappa_hat]`` when it is done.  ``bcast(buf)`` broadcasts a float64 array of 2 from rank 0
    in place (torch.distributed.broadcast on a CPU tensor; any other transport will do).

    Returns (kappa_hat, objective at kappa_hat, [(kappa, objective), ...] of this rank's evaluations, scipy result or None)."""
    tf = transform or (lambda p: float(abs(p)) + 1e-3)
    calls = []
    if rank == 0:
        import scipy.optimize as opt

        def obj(p):
            kappa = tf(p[0])
            bcast(np.array([1.0, kappa]))
            v = float(evaluate(kappa))
            calls.append((kappa, v))
            return v
        opts = {"xatol": 1e-3, "disp": False}
        opts.update(options)
        res = opt.minimize(obj, [float(x0)], method="nelder-mead", options=opts)
        k_hat = tf(res.x[0])
        bcast(np.array([0.0, k_hat]))
        return k_hat, float(res.fun), calls, res
    while True:
        msg = np.zeros(2)
        bcast(msg)
        if msg[0] == 0.0:
            best = min(calls, key=lambda kv: kv[1])[1] if calls else float("nan")
            return float(msg[1]), best, calls, None
        calls.append((float(msg[1]), float(evaluate(float(msg[1])))))


def sharded_run(plan, reduce_level: int, allreduce=None, likelihood=True, predict=True):
    """Run a (local) plan with the single front all-reduce.

    ``allreduce``: None -> the plan's own RCCL communicator (``plan.comm_init`` was called) does it
    on the device, in stream order; otherwise a callable ``buf -> summed buf`` on host NumPy arrays
    (used with torch.distributed/gloo in tests, or any other transport)."""
    if reduce_level < 0:
        plan.run(likelihood, predict)
        return
    plan.set_reduce_level(reduce_level)
    if allreduce is None:
        plan.run(likelihood, predict)
        return
    plan.run(likelihood, predict, split=True)
    buf = plan.reduce_export()
    plan.reduce_import(allreduce(buf))
    plan.resume()
