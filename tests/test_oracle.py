"""The CPU oracles against (a) the golden vectors produced by running the reference and
(b) exact-GP identities that need no reference at all.  CPU only."""
import numpy as np
import pytest

import _cases as K
from oracle.mra_faithful import run_faithful, run_subtree_sample
from oracle.mra_levelwise import run_levelwise

WELL = ["kat2", "kat3", "c1", "t1000", "t201", "kat4", "g32", "g64"]       # well conditioned cases


@pytest.mark.parametrize("name", WELL)
def test_faithful_oracle_matches_reference_goldens(name):
    cs = K.load_case(name)
    out = run_faithful(cs["topo"], cs["locs"], cs["covfun"], cs["y_obs"], cs["c"]["R"])
    g = cs["g"]
    assert abs(out["lik"] - g["lik"]) <= 1e-10 * abs(g["lik"])
    assert np.max(np.abs(out["mean"] - g["mean"])) < 1e-8
    assert K.rel(out["sd"], g["sd"]) < 1e-8


@pytest.mark.parametrize("name", WELL + ["kat1", "g64m"])
def test_levelwise_oracle_matches_reference_goldens(name):
    cs = K.load_case(name)
    out = run_levelwise(cs["topo"], cs["locs"], cs["spec"], cs["y_obs"], cs["c"]["R"])
    g = cs["g"]
    assert abs(out["lik"] - g["lik"]) <= 1e-10 * abs(g["lik"])
    assert np.max(np.abs(out["mean"] - g["mean"])) < 1e-8
    # Matern32: the reference's own sd carries error from its explicit inverses / eigh square root
    # (SURVEY.md section 7 hard part 5); ExpCovFun sd agrees to ~1e-9
    assert K.rel(out["sd"], g["sd"]) < (1e-8 if cs["c"]["kern"] == "exp" else 1e-5)


@pytest.mark.parametrize("name", ["kat1", "kat2", "kat3", "kat4"])
def test_oracles_reproduce_exact_kriging(name):
    """MRA == exact GP for M=0 and for 1-D exponential with knots on the boundaries
    (pyMRA/tests/unit-tests.py:22-71, 75-130)."""
    cs = K.load_case(name)
    lik, mean, sd = K.kriging(cs["locs"], cs["y_obs"], cs["spec"], cs["c"]["R"])
    for fn, cov in ((run_levelwise, cs["spec"]), (run_faithful, cs["covfun"])):
        out = fn(cs["topo"], cs["locs"], cov, cs["y_obs"], cs["c"]["R"])
        assert abs(out["lik"] - lik) <= 1e-9 * abs(lik)
        assert np.max(np.abs(out["mean"] - mean)) < 1e-9
        assert K.rel(out["sd"], sd) < 1e-8


def test_dropped_rows_report_zero_like_the_reference():
    """1-D terciles drop the rows that equal a percentile (MRANode.py:222-228); the reference then
    reports mean = sd = 0 there (BTil rows never filled, :492-495)."""
    cs = K.load_case("t201")
    t = cs["topo"]
    dropped = t.perm[(t.perm >= 0) & ~t.in_leaf]
    assert len(dropped) > 0
    assert np.all(cs["g"]["mean"][dropped] == 0) and np.all(cs["g"]["sd"][dropped] == 0)
    out = run_levelwise(t, cs["locs"], cs["spec"], cs["y_obs"], cs["c"]["R"])
    assert np.all(out["mean"][dropped] == 0) and np.all(out["sd"][dropped] == 0)


def test_levelwise_likelihood_only_mode_and_opaque_callable():
    cs = K.load_case("g32")
    a = run_levelwise(cs["topo"], cs["locs"], cs["spec"], cs["y_obs"], cs["c"]["R"])
    b = run_levelwise(cs["topo"], cs["locs"], cs["covfun"], cs["y_obs"], cs["c"]["R"], predict=False)
    assert abs(a["lik"] - b["lik"]) < 1e-9 * abs(a["lik"])


def test_ill_conditioned_case_reference_vs_stable():
    """u3 (10x10, Matern32, R=1e-6, KMeans knots and splits): the reference's explicit-inverse
    arithmetic is only good to ~1e-4 here (the faithful restatement differs from the reference by
    that much from a different row order alone); the stable factorisation stays within that."""
    cs = K.load_case("u3")
    g = cs["g"]
    f = run_faithful(cs["topo"], cs["locs"], cs["covfun"], cs["y_obs"], cs["c"]["R"])
    l = run_levelwise(cs["topo"], cs["locs"], cs["spec"], cs["y_obs"], cs["c"]["R"])
    assert abs(f["lik"] - g["lik"]) < 1e-3 * abs(g["lik"])
    assert abs(l["lik"] - g["lik"]) < 1e-3 * abs(g["lik"])
    assert np.max(np.abs(l["mean"] - g["mean"])) < 1e-3


def test_subtree_sample_is_the_same_arithmetic():
    """The bounded CPU-baseline sample (bench.py) processes a subtree exactly like the full run."""
    cs = K.load_case("g64")
    t = cs["topo"]
    top = int(t.level_ptr[1])                               # first level-1 node
    n, secs, tm = run_subtree_sample(t, cs["locs"], cs["covfun"], cs["y_obs"], cs["c"]["R"], top, do_gc=False)
    assert n == 21 and secs > 0 and tm["posterior"] > 0


@pytest.mark.parametrize("name", ["kat2", "kat3", "c1", "g32"])
def test_faithful_oracle_per_node_blocks_match_the_reference(name):
    """Every intermediate block of every node (B, kInv, kTil, A[res][res], omg[res], BTil[res], d, u) against what
    the reference held on its Node objects (tests/golden/<name>_nodes.npz, pyMRA/MRANode.py:384-391, 426-511)."""
    cs = K.load_case(name)
    t = cs["topo"]
    nodes = {}
    run_faithful(t, cs["locs"], cs["covfun"], cs["y_obs"], cs["c"]["R"], keep_nodes=nodes)
    gold = K.load_node_goldens(name)
    assert len(gold) == t.n_nodes
    for i in range(t.n_nodes):
        g, s = gold[t.node_ident[i]], nodes[i]
        m = int(t.node_level[i])
        pr = K.node_real_rows(t, i) - t.node_row0[i]
        sc = max(1.0, float(np.abs(g["Amm"]).max()))
        assert g["B"].shape == (len(pr), s.B.shape[1])
        assert np.max(np.abs(s.B[pr] - g["B"])) < 1e-11
        assert np.max(np.abs(s.kInv - g["kInv"])) < 1e-11
        assert np.max(np.abs(s.A[m][m] - g["Amm"])) < 1e-9 * sc
        assert np.max(np.abs(np.ravel(s.omg[m]) - g["omg"])) < 1e-9 * max(1.0, float(np.abs(g["omg"]).max()))
        assert np.max(np.abs(s.kTil - g["kTil"])) < 1e-9 * max(1.0, float(np.abs(g["kTil"]).max()))
        assert np.max(np.abs(s.BTil[m][pr] - g["BTil"])) < 1e-9
        assert abs(float(np.ravel(s.d)[0]) - float(g["d"][0])) < 1e-8 * max(1.0, abs(float(g["d"][0])))
        assert abs(float(np.ravel(s.u)[0]) - float(g["u"][0])) < 1e-8 * max(1.0, abs(float(g["u"][0])))


def test_data_loader_mirrors_the_reference():
    """pyMRA/DataLoader.py:5-19: sizes, NaN counts and the KAT4 inputs."""
    import pymra_amd.DataLoader as dl
    y, locs, y_obs = dl.load_data("small", True)
    assert y.shape == (10, 10) and locs.shape == (100, 2) and y_obs.shape == (100, 1) and int(np.isnan(y_obs).sum()) == 14
    g = np.load(K.os.path.join(K.GOLD, "kat4.npz"))
    assert np.array_equal(locs, g["locs"]) and np.array_equal(y_obs, g["y_obs"], equal_nan=True)
    y, locs, y_obs = dl.load_data("large", include_truth=True)
    assert y.shape == (100, 100) and locs.shape == (10000, 2) and int(np.isnan(y_obs).sum()) == 1440
    yo, lo = dl.load_data("small")
    assert yo.shape == (100, 1) and lo.shape == (100, 2)
    with pytest.raises(ValueError):
        dl.load_data("medium")
