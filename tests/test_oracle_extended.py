"""Who is right where the reference and this implementation disagree?  An 80-bit evaluation of the
same model (oracle/mra_extended.py) says: the float64 level-wise algorithm (the one the GPU runs) is
accurate to ~1e-12, the reference's Matern32 predictive sd is not (explicit inverses + eigh square
root, pyMRA/MRANode.py:444-445, 504-507).  CPU only, small trees."""
import numpy as np
import pytest

import _cases as K
from oracle.mra_extended import run_extended
from oracle.mra_levelwise import run_levelwise


@pytest.mark.parametrize("name", ["g32", "c1", "g64m", "g128m", "u3"])
def test_float64_algorithm_against_extended_precision(name):
    cs = K.load_case(name)
    g = cs["g"]
    ext = run_extended(cs["topo"], cs["locs"], cs["spec"], cs["y_obs"], cs["c"]["R"])
    lw = run_levelwise(cs["topo"], cs["locs"], cs["spec"], cs["y_obs"], cs["c"]["R"])
    hard = name == "u3"
    # (i) the algorithm the GPU executes vs extended-precision truth: meets the 1e-6 bar with margin
    assert abs(lw["lik"] - ext["lik"]) <= (1e-7 if hard else 1e-11) * abs(ext["lik"])
    assert np.max(np.abs(lw["mean"] - ext["mean"])) < (1e-7 if hard else 1e-9)
    assert K.rel(lw["sd"], ext["sd"]) < (1e-6 if hard else 1e-9)
    # (ii) the reference vs the same truth
    ref_sd = K.rel(g["sd"], ext["sd"])
    ref_lik = abs(float(g["lik"]) - ext["lik"]) / abs(ext["lik"])
    if name == "g64m":
        assert ref_sd > 1e-8          # the reference's own sd error is far above rounding here
        assert ref_sd < 1e-5 and ref_lik < 1e-10
    if name == "g128m":
        # 128^2 (SURVEY.md section 7.5's own adjudication size): the reference's sd is off by up to 7e-4, a tenth of the
        # locations by more than north_star's 1e-6, while its likelihood and mean are fine
        e = np.abs(g["sd"] - ext["sd"]) / ext["sd"]
        assert 1e-4 < e.max() < 2e-3 and (e > 1e-6).mean() > 0.05 and ref_lik < 1e-10
        assert np.max(np.abs(np.asarray(g["mean"]).ravel() - ext["mean"])) < 1e-7
    if name == "u3":
        assert ref_lik > 1e-5         # ill-conditioned: the reference is off by ~2e-4 in the likelihood itself
    if name in ("g32", "c1"):
        assert ref_sd < 1e-8 and ref_lik < 1e-10


def test_c3_geometry_subtree_adjudication():
    """C3's own geometry: one level-3 subtree of the 1024^2 tree with pruned ancestors (tests/golden/make_c3_subtree_ext.py).
    The float64 level-wise algorithm reproduces the stored 80-bit results to 1e-9 (sd) on all 16 384 locations; the numbers the
    generating script recorded for the reference's algorithm say where the C3 sd gap comes from."""
    import json
    import os
    import sys
    sys.path.insert(0, K.GOLD)
    import make_c3_subtree_ext as mk
    g = np.load(os.path.join(K.GOLD, "c3_subtree_ext.npz"))
    prm = json.loads(str(g["params"]))
    c, locs, y_obs, lt, red, spec = mk.local_problem(prm["which"])
    assert abs(float(np.nansum(y_obs)) - float(g["y_checksum"])) < 1e-9
    own = lt.perm[lt.in_leaf]
    assert np.array_equal(own, g["own"])
    lw = run_levelwise(lt, locs, spec, y_obs, c["R"], reduce_level=red, allreduce=lambda b: b)
    assert abs(lw["lik"] - float(g["lik"])) <= 1e-12 * abs(float(g["lik"]))
    assert np.max(np.abs(lw["mean"][own] - g["mean"])) < 1e-9
    assert K.rel(lw["sd"][own], g["sd"]) < 1e-9
    assert float(g["ref_alg_sd_rel_max"]) > 1e-5 and float(g["ref_alg_sd_frac_above_1e6"]) > 0.05
    assert float(g["ref_alg_lik_rel"]) < 1e-10
