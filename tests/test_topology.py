"""Host tree replay == the tree the reference built (captured in the golden fixtures)."""
import numpy as np
import pytest

import _cases as K


@pytest.mark.parametrize("name", [n for n in K.SMALL + K.MEDIUM + K.LARGE + K.HEADLINE if K.have(n)])
def test_topology_matches_reference(name):
    cs = K.load_case(name)
    g, topo = cs["g"], cs["topo"]
    ids = list(g["node_ident"])
    assert topo.n_nodes == len(ids)
    assert int(g["M_eff"]) == cs["M"] and int(g["J_eff"]) == cs["J"]
    for t, i in enumerate(topo.order_preorder):            # reference construction order
        assert topo.node_ident[i] == ids[t]
        kq = topo.perm[topo.knot_rows[topo.knot_ptr[i]:topo.knot_ptr[i + 1]]]
        gk = g["knots"][g["knot_ptr"][t]:g["knot_ptr"][t + 1]]
        assert np.array_equal(kq, gk), "knots of node %s" % ids[t]
        rows = topo.perm[topo.node_row0[i]:topo.node_row1[i]]
        rows = rows[rows >= 0]
        assert len(rows) == g["node_nrows"][t] and rows.sum() == g["node_rowsum"][t]
        assert bool(topo.node_leaf[i]) == bool(g["node_leaf"][t])


def test_layout_invariants():
    cs = K.load_case("t201")                                # 1-D terciles: percentiles hit grid points -> dropped rows
    t = cs["topo"]
    assert t.P % 16 == 0 and np.all(t.node_row0 % 16 == 0) and np.all(t.node_row1 % 16 == 0)
    real = t.perm[t.perm >= 0]
    assert np.array_equal(np.sort(real), np.arange(t.N))    # every caller row exactly once
    assert (~t.in_leaf[t.perm >= 0]).sum() > 0              # some rows belong to no leaf
    for i in range(t.n_nodes):                              # children nest inside the parent
        for c in t.child_list[t.child_ptr[i]:t.child_ptr[i + 1]]:
            assert t.node_row0[i] <= t.node_row0[c] and t.node_row1[c] <= t.node_row1[i]
            assert t.node_parent[c] == i
        kq = t.knot_rows[t.knot_ptr[i]:t.knot_ptr[i + 1]]
        assert np.all((kq >= t.node_row0[i]) & (kq < t.node_row1[i]))


def test_shape_resolution_matches_reference_rules():
    from pymra_amd.topology import resolve_tree_shape
    assert resolve_tree_shape(65536, 2, 16, 4, 4) == (4, 4)
    assert resolve_tree_shape(65536, 2, 16, -1, -1) == (6, 4)          # maxM = int(log(N J/r+1)/log J)-1
    assert resolve_tree_shape(100, 1, 2, 9, 3)[0] == 3                  # clipped with a warning
    with pytest.raises(AttributeError):
        resolve_tree_shape(100, 1, 2, 3, -1)                            # MRATree.py:31-33
    with pytest.raises(OverflowError):
        resolve_tree_shape(2, 1, 2, 0, 1)                               # J=1 -> log(J)=0 (unit-tests.py #1)


@pytest.mark.parametrize("n,r,M,grid", [(64, 16, 2, True), (129, 16, 2, True), (301, 16, 3, True), (256, 16, 4, True),
                                         (100, 8, 2, False), (40, 16, 3, True)])
def test_native_replay_equals_python_replay(built_library, n, r, M, grid):
    """csrc/mra_topology.h (C++: exact sequential means, MT19937 + NumPy's legacy shuffle) against the general
    Python replay: every array identical, and NumPy's global RNG left in the same state (also when the
    native path declines - (40,16,3) has nodes with <= 100 rows - and the Python replay takes over)."""
    import pymra_amd.MRATools as mt
    from pymra_amd.topology import build_topology, _replay_native
    locs = mt.genLocations2d(Nx=n, Ny=n) if grid else np.random.RandomState(1).uniform(size=(n * n, 2))
    np.random.seed(5); np.random.normal(size=3)             # leaves a cached Gaussian in the state
    a = build_topology(locs, r, M, 4, native=True)
    sa = np.random.get_state()
    np.random.seed(5); np.random.normal(size=3)
    b = build_topology(locs, r, M, 4, native=False)
    sb = np.random.get_state()
    for f in ("N", "d", "M", "J", "r", "P", "n_nodes", "n_levels", "node_ident"):
        assert getattr(a, f) == getattr(b, f), f
    for f in ("perm", "src", "in_leaf", "level_ptr", "node_level", "node_row0", "node_row1", "node_leaf", "node_parent",
              "child_ptr", "child_list", "knot_ptr", "knot_rows", "cw", "order_preorder"):
        assert np.array_equal(getattr(a, f), getattr(b, f)), f
    assert np.array_equal(sa[1], sb[1]) and sa[2:] == sb[2:]
    if n == 40:
        np.random.seed(5)
        assert _replay_native(np.ascontiguousarray(locs), r, M, 4) is None
