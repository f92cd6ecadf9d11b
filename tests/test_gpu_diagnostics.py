"""SURVEY.md section 8(f)4 on the GPU: per-node blocks through mra_get_node_block de-whitened by
pymra_amd.diagnostics against what the reference held on its Node objects (tests/golden/*_nodes.npz),
MRATree.getBasisFunctionsMatrix against the reconstruction identities of pyMRA/tests/debug-posterior.py:97-109,
and README example 1 (pyMRA/README.md:19-46) through DataLoader."""
import numpy as np
import pytest

import _cases as K

pytestmark = pytest.mark.gpu


def _tree(cs, **kw):
    import pymra_amd
    import pymra_amd.MRATools as mt
    c = cs["c"]
    cov = (lambda a, b: mt.ExpCovFun(a, b, l=c["l"])) if c["kern"] == "exp" else (lambda a, b: mt.Matern32(a, b, l=c["l"], sig=c["sig"]))
    if c["dim"] == 2 and not c.get("data"):
        import make_golden as mg
        mg.make_inputs(c)                                  # global RNG where the reference had it when it drew its knots
    return pymra_amd.MRATree(cs["locs"], c["r"], cov, cs["y_obs"], c["R"], M=c["M"], J=c["J"], **kw)


@pytest.mark.parametrize("name", ["kat2", "kat3", "c1", "g32"])
def test_node_blocks_match_the_reference(built_library, name):
    cs = K.load_case(name)
    tree = _tree(cs)
    lik0 = float(tree.getLikelihood()[0, 0])
    nodes = tree.getNodeBlocks()
    gold = K.load_node_goldens(name)
    assert len(nodes) == len(gold)
    for nb in nodes:
        g = gold[nb.ID]
        sc = lambda a: max(1.0, float(np.abs(a).max()))
        assert nb.B.shape == g["B"].shape, nb.ID
        assert np.max(np.abs(nb.B - g["B"])) < 1e-9, nb.ID
        assert np.max(np.abs(nb.kInv - g["kInv"])) < 1e-9
        assert np.max(np.abs(nb.kC @ nb.kC.T - g["kC"] @ g["kC"].T)) < 1e-7 * sc(g["kC"]) ** 2
        assert np.max(np.abs(nb.kTil - g["kTil"])) < 1e-7 * sc(g["kTil"]), nb.ID
        Amm = nb.A[nb.res][nb.res]                 # leaves carry the same A[k][l] / omg[k] structure as every node
        omg = nb.omg[nb.res]
        assert np.max(np.abs(Amm - g["Amm"])) < 1e-8 * sc(g["Amm"]), nb.ID
        assert np.max(np.abs(omg - g["omg"])) < 1e-8 * sc(g["omg"]), nb.ID
        assert np.max(np.abs(nb.BTil - g["BTil"])) < 1e-8, nb.ID
        assert abs(nb.d - float(g["d"][0])) < 1e-8 * max(1.0, abs(float(g["d"][0]))), nb.ID
        assert abs(nb.u - float(g["u"][0])) < 1e-7 * max(1.0, abs(float(g["u"][0]))), nb.ID
    # the diagnostics passes leave the plan usable and unchanged
    assert abs(float(tree.reevaluate(lambda a, b: cs["spec"])[0, 0]) - lik0) <= 1e-12 * abs(lik0)


@pytest.mark.parametrize("name", ["kat1", "kat3", "c1", "g32"])
def test_basis_functions_matrix_identities(built_library, name):
    """debug-posterior.py:97-109: with B = getBasisFunctionsMatrix('prior', timesKC=True),
    B (I + B^T H^T R^-1 H B)^-1 B^T is the posterior covariance of the MRA prior B B^T.  Its diagonal must equal
    predict()'s variance for ANY tree (the MRA algorithm is exact inference under its own prior), the posterior
    basis BP must satisfy BP BP^T = the same matrix, and for the exact-screening cases (kat1: M=0, kat3: 1-D
    exponential) B B^T is the kernel itself and the result equals kriging."""
    import scipy.linalg as lng
    cs = K.load_case(name)
    tree = _tree(cs)
    xP, sdP = tree.predict()
    N = len(cs["locs"])
    B = tree.getBasisFunctionsMatrix(distr="prior", timesKC=True)
    BP = tree.getBasisFunctionsMatrix(distr="posterior", timesKC=True)
    assert isinstance(B, np.matrix) and B.shape[0] == N and BP.shape == B.shape
    obs = np.isfinite(np.asarray(cs["y_obs"]).ravel())
    H = np.matrix(np.eye(N))[obs, :]
    Rm = np.matrix(cs["c"]["R"] * np.eye(int(obs.sum())))
    recSigP = B * lng.inv(np.eye(B.shape[1]) + B.T * H.T * lng.inv(Rm) * H * B) * B.T
    t = tree.topology
    kept = np.zeros(N, dtype=bool)
    kept[t.perm[t.in_leaf]] = True                          # rows a 1-D tercile split drops report 0 (MRANode.py:222-228)
    assert np.max(np.abs(np.diag(recSigP)[kept] - sdP[kept] ** 2)) < 1e-9
    post = np.asarray(BP * BP.T)
    assert np.max(np.abs(post - np.asarray(recSigP))[np.ix_(kept, kept)]) < 1e-8
    mean_rec = recSigP * H.T * (np.asarray(cs["y_obs"])[obs] / cs["c"]["R"])
    assert np.max(np.abs(np.asarray(mean_rec).ravel()[kept] - np.asarray(xP).ravel()[kept])) < 1e-8
    if name in ("kat1", "kat3"):
        Sig = np.asarray(cs["spec"].evaluate(cs["locs"], cs["locs"]))
        assert np.max(np.abs(np.asarray(B * B.T) - Sig)) < 1e-9
        _, _, sd = K.kriging(cs["locs"], cs["y_obs"], cs["spec"], cs["c"]["R"])
        assert np.max(np.abs(np.sqrt(np.diag(recSigP)) - sd)) < 1e-8
    # options of the reference's signature
    Bl = tree.getBasisFunctionsMatrix(distr="prior", groupByResolution=True, order="leaves")
    assert isinstance(Bl, list) and len(Bl) == t.n_levels and sum(b.shape[1] for b in Bl) == B.shape[1]
    B0 = tree.getBasisFunctionsMatrix(distr="prior")
    assert np.max(np.abs(np.asarray(B0[:, :Bl[0].shape[1]]) - np.asarray(tree.root.B))) < 1e-10


def test_readme_example_1_small(built_library):
    """README.md:19-46 with the small data set: `MRATree(locs, r0, cov, y_obs, me_scale, critDepth)` passes
    critDepth = 0 into the M slot (a single dense leaf: exact kriging)."""
    import pymra_amd.DataLoader as dl
    import pymra_amd.MRATools as mt
    from pymra_amd import MRATree
    r0, me_scale, critDepth = 4, 1e-4, 0
    y, locs, y_obs = dl.load_data("small", True)
    Nx, Ny = y.shape
    y_obs = y_obs.reshape((Nx * Ny, 1))
    cov = lambda _locs1, _locs2: mt.ExpCovFun(_locs1, _locs2, l=2)
    mraTree = MRATree(locs, r0, cov, y_obs, me_scale, critDepth)
    yP, sdP = mraTree.predict()
    assert mraTree.M == 0
    lik, mean, sd = K.kriging(locs, y_obs, mt.KernelSpec(mt.KIND_EXP, 2.0), me_scale)
    assert np.max(np.abs(np.asarray(yP).ravel() - mean)) < 1e-8 and K.rel(sdP, sd) < 1e-7
    assert abs(float(mraTree.getLikelihood()[0, 0]) - lik) <= 1e-9 * abs(lik)
    assert sdP.reshape((Nx, Ny)).shape == (10, 10) and yP.reshape((Nx, Ny)).shape == (10, 10)


def test_readme_example_1_on_a_large_single_leaf(built_library):
    """README example 1 on a 40 x 40 sub-grid of the packaged 100 x 100 sample: M = 0 puts all 1600 points (about 1370
    observed: 86 column tiles) into ONE leaf, the blocked right-looking Cholesky path for large leaves, against exact
    kriging.  (The full 100 x 100 case is the same code with 535 column tiles; tools/readme_example1.py times it.)"""
    import pymra_amd.DataLoader as dl
    import pymra_amd.MRATools as mt
    from pymra_amd import MRATree
    y, locs, y_obs = dl.load_data("large", True)
    n = 40
    sel = np.array([j * 100 + i for j in range(n) for i in range(n)])
    locs, y_obs = locs[sel], y_obs[sel].reshape(-1, 1)
    tree = MRATree(locs, 4, lambda a, b: mt.ExpCovFun(a, b, l=2), y_obs, 1e-4, 0)
    assert tree.M == 0 and tree.topology.n_nodes == 1
    yP, sdP = tree.predict()
    lik, mean, sd = K.kriging(locs, y_obs, mt.KernelSpec(mt.KIND_EXP, 2.0), 1e-4)
    assert abs(float(tree.getLikelihood()[0, 0]) - lik) <= 1e-8 * abs(lik)
    assert np.max(np.abs(np.asarray(yP).ravel() - mean)) < 1e-6
    assert K.rel(sdP, sd) < 1e-5                              # cond(Sigma_y) ~ 1e8 at me_scale = 1e-4: both sides lose digits
