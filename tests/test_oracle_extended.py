"""Who is right where the reference and this implementation disagree?  An 80-bit evaluation of the
same model (oracle/mra_extended.py) says: the float64 level-wise algorithm (the one the GPU runs) is
accurate to ~1e-12, the reference's Matern32 predictive sd is not (explicit inverses + eigh square
root, pyMRA/MRANode.py:444-445, 504-507).  CPU only, small trees."""
import numpy as np
import pytest

import _cases as K
from oracle.mra_extended import run_extended
from oracle.mra_levelwise import run_levelwise


@pytest.mark.parametrize("name", ["g32", "c1", "g64m", "u3"])
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
    if name == "u3":
        assert ref_lik > 1e-5         # ill-conditioned: the reference is off by ~2e-4 in the likelihood itself
    if name in ("g32", "c1"):
        assert ref_sd < 1e-8 and ref_lik < 1e-10
