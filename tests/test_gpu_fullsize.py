"""BASELINE.json's headline configuration (1024x1024 grid, M=6, J=4, r0=32, Matern32) on the GPU:
golden vector from the reference (when present), the reference's published-in-SURVEY likelihood,
and size-independent properties (linearity of the predictive mean in y, y-independence of the
predictive sd, likelihood-only == full)."""
import os

import numpy as np
import pytest

import _cases as K

pytestmark = pytest.mark.gpu

C3_LIK_SURVEY = 3.9562815625e+07          # reference run, 11 printed digits


@pytest.fixture(scope="module")
def c3(built_library):
    from pymra_amd import plan
    if plan.device_count() < 1:
        pytest.fail("no GPU visible")
    import make_golden as mg
    from pymra_amd.topology import build_topology
    import pymra_amd.MRATools as mt
    c = mg.CASES["c3"]
    locs, y_obs, _ = mg.make_inputs(c)
    topo = build_topology(locs, c["r"], c["M"], c["J"])
    pl = plan.HipPlan(topo, 0)
    pl.set_locs(locs)
    pl.set_obs(y_obs, c["R"])
    pl.set_kernel(mt.KIND_MATERN32, c["l"], c["sig"], 1.0)
    pl.run(True, True)
    d, u = pl.likelihood()
    mean, var = pl.predict()
    return dict(pl=pl, topo=topo, locs=locs, y_obs=y_obs, c=c, lik=d + u, mean=mean, var=var)


def test_c3_tree_shape(c3):
    t = c3["topo"]
    assert t.n_nodes == 5461 and list(np.diff(t.level_ptr)) == [4 ** m for m in range(7)]
    ranks = np.diff(t.knot_ptr)[t.node_leaf]
    assert ranks.min() >= 230 and ranks.max() <= 256


def test_c3_likelihood_matches_reference(c3):
    assert abs(c3["lik"] - C3_LIK_SURVEY) <= 1e-10 * C3_LIK_SURVEY
    # spot values printed by the reference run recorded in SURVEY.md section 8(c)
    assert np.allclose(c3["mean"][:3], [-0.1460093, -0.13918176, -0.13169202], atol=5e-8)
    assert np.allclose(np.sqrt(c3["var"][:3]), [0.02140315, 0.02007744, 0.01889032], rtol=2e-3)


def test_c3_against_golden_vector(c3):
    p = os.path.join(K.GOLD, "c3.npz")
    if not os.path.exists(p):
        pytest.skip("tests/golden/c3.npz not generated (30 min reference run)")
    g = np.load(p)
    assert abs(c3["lik"] - float(g["lik"])) <= 1e-9 * abs(float(g["lik"]))
    idx = g["sample_idx"]
    assert np.max(np.abs(c3["mean"][idx] - g["mean"])) < 1e-6
    sd = np.sqrt(c3["var"])
    assert K.rel(sd[idx], g["sd"]) < 5e-3                     # Matern32: reference sd error, see parity policy
    assert abs(c3["mean"].sum() - float(g["mean_sum"])) < 1e-4
    assert abs(sd.sum() - float(g["sd_sum"])) < 1e-4 * float(g["sd_sum"])


def test_c3_linearity_and_sd_invariance(c3):
    """mean is linear in y for a fixed mask, sd does not depend on y at all."""
    pl, y = c3["pl"], c3["y_obs"]
    rng = np.random.RandomState(5)
    y2 = np.where(np.isfinite(y), rng.normal(size=y.shape), np.nan)
    pl.set_obs(y2, c3["c"]["R"]); pl.run(True, True)
    m2, v2 = pl.predict()
    pl.set_obs(2.0 * y - 3.0 * y2, c3["c"]["R"]); pl.run(True, True)
    m3, v3 = pl.predict()
    assert np.max(np.abs(m3 - (2.0 * c3["mean"] - 3.0 * m2))) < 1e-9
    assert K.rel(v2, c3["var"]) < 1e-12 and K.rel(v3, c3["var"]) < 1e-12
    pl.set_obs(y, c3["c"]["R"]); pl.run(True, False)
    d, u = pl.likelihood()
    assert abs(d + u - c3["lik"]) <= 1e-13 * abs(c3["lik"])
