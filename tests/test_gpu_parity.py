"""Parity of the HIP path (through the C ABI) with the CPU oracle and with the golden vectors that
were produced by running the reference.  GPU only.

Tolerances: north_star asks for 1e-6 relative on getLikelihood() and predict(); against the
level-wise oracle (same algorithm) the HIP path is held to 1e-10 / 1e-9; against the reference's
own numbers it is held to 1e-9 (lik), 1e-7 (mean) and 1e-6 (sd, ExpCovFun).  For Matern32 the
REFERENCE's sd is itself only good to ~1e-3 relative (explicit inverses + eigh square root,
SURVEY.md section 7 hard part 5; adjudicated in extended precision by tests/test_oracle_extended.py),
so there the 1e-6 bar is applied against the oracle AND against 80-bit runs of the same model (1e-9), and the golden sd is
checked to the reference's own measured error per case (c1 1e-7, g64m 1e-5, g128m 1e-3; C3 at full size 5e-5).
"""
import numpy as np
import pytest

import _cases as K

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip(built_library):
    from pymra_amd import plan
    if plan.device_count() < 1:
        pytest.fail("no GPU visible: the gpu-marked tests must run on the MI355X box")
    return plan


def run_hip(plan_mod, cs, predict=True, topo=None):
    t = topo if topo is not None else cs["topo"]
    pl = plan_mod.HipPlan(t, 0)
    pl.set_locs(cs["locs"])
    pl.set_obs(cs["y_obs"], cs["c"]["R"])
    s = cs["spec"]
    pl.set_kernel(s.kind, s.l, s.sig, s.scale)
    pl.run(True, predict)
    d, u = pl.likelihood()
    mean, var = pl.predict() if predict else (None, None)
    return pl, d + u, mean, var


@pytest.mark.parametrize("name", [n for n in K.SMALL + K.MEDIUM + K.LARGE if K.have(n)])
def test_hip_matches_oracle_and_reference_goldens(hip, name):
    from oracle.mra_levelwise import run_levelwise
    cs = K.load_case(name)
    pl, lik, mean, var = run_hip(hip, cs)
    sd = np.sqrt(var)
    ref = run_levelwise(cs["topo"], cs["locs"], cs["spec"], cs["y_obs"], cs["c"]["R"])
    hard = name == "u3"                                   # R=1e-6 Matern32: cond ~1e9
    assert abs(lik - ref["lik"]) <= (1e-8 if hard else 1e-11) * abs(ref["lik"])
    assert np.max(np.abs(mean - ref["mean"])) < (1e-8 if hard else 1e-10)
    assert K.rel(sd, ref["sd"]) < (1e-6 if hard else 1e-9)
    g = cs["g"]
    if not hard:
        assert abs(lik - g["lik"]) <= 1e-9 * abs(g["lik"])
        assert np.max(np.abs(mean - g["mean"])) < 1e-7
        # sd against the reference: ExpCovFun goldens are good to ~1e-8; the reference's Matern32 sd carries its own error
        # (explicit inverses + eigh square root, pyMRA/MRANode.py:444-445, 504-507), measured against 80-bit runs in
        # tests/test_oracle_extended.py: c1 < 1e-8, g64m < 1e-5, g128m 6.6e-4 - so the bound is per case, and the HIP sd is held
        # to the 80-bit truth itself below
        ref_sd_bound = {"c1": 1e-7, "g64m": 1e-5, "g128m": 1e-3}
        assert K.rel(sd, g["sd"]) < (1e-6 if cs["c"]["kern"] == "exp" else ref_sd_bound[name])
    if cs["c"]["kern"] == "m32" and name in ("c1", "g64m", "g128m"):
        from oracle.mra_extended import run_extended
        ext = run_extended(cs["topo"], cs["locs"], cs["spec"], cs["y_obs"], cs["c"]["R"])
        assert K.rel(sd, ext["sd"]) < 1e-9 and np.max(np.abs(mean - ext["mean"])) < 1e-9      # north_star: 1e-6
        if name == "g128m":
            e_ref = np.abs(g["sd"] - ext["sd"]) / ext["sd"]
            e_us = np.abs(sd - g["sd"]) / ext["sd"]
            assert np.max(np.abs(e_us - e_ref)) < 1e-8       # we differ from the reference exactly where it differs from the truth
    pl.close()


@pytest.mark.parametrize("name", ["kat1", "kat2", "kat3", "kat4"])
def test_mratree_reproduces_exact_kriging(hip, name):
    """The reference's own unit-test identities (pyMRA/tests/unit-tests.py:22-71, 75-130) through
    the drop-in MRATree API."""
    import pymra_amd
    import pymra_amd.MRATools as mt
    cs = K.load_case(name)
    c = cs["c"]
    cov = lambda a, b: mt.ExpCovFun(a, b, l=c["l"])
    tree = pymra_amd.MRATree(cs["locs"], c["r"], cov, cs["y_obs"], c["R"], M=c["M"], J=c["J"])
    xP, sdP = tree.predict()
    lik, mean, sd = K.kriging(cs["locs"], cs["y_obs"], cs["spec"], c["R"])
    assert isinstance(xP, np.matrix) and xP.shape == (len(cs["locs"]), 1)
    assert isinstance(sdP, np.ndarray) and sdP.shape == (len(cs["locs"]),)
    L = tree.getLikelihood()
    assert isinstance(L, np.matrix) and L.shape == (1, 1)
    assert abs(L[0, 0] - lik) <= 1e-9 * abs(lik)
    assert np.max(np.abs(np.asarray(xP).ravel() - mean)) < 1e-9
    assert K.rel(sdP, sd) < 1e-8


def test_likelihood_only_and_rerun_are_consistent(hip):
    cs = K.load_case("g64m")
    pl, lik, mean, var = run_hip(hip, cs)
    pl.run(True, True)
    d, u = pl.likelihood()
    m2, v2 = pl.predict()
    assert d + u == lik and np.array_equal(m2, mean) and np.array_equal(v2, var)       # bit-identical re-run
    pl.run(True, False)
    d, u = pl.likelihood()
    assert abs(d + u - lik) <= 1e-13 * abs(lik)
    with pytest.raises(hip.MraError):
        pl.predict()                                       # not computed in likelihood-only mode
    pl.close()


def test_observation_patterns(hip):
    """Edge cases of the observation mask: none observed anywhere, one leaf without observations,
    everything observed, a single observation."""
    from oracle.mra_levelwise import run_levelwise
    base = K.load_case("g32")
    N = len(base["locs"])
    rng = np.random.RandomState(0)
    full = rng.normal(size=(N, 1))
    t = base["topo"]
    leaf0 = t.perm[t.node_row0[t.n_nodes - 1]:t.node_row1[t.n_nodes - 1]]
    leaf0 = leaf0[leaf0 >= 0]
    pats = {"all": full.copy(), "none": np.full((N, 1), np.nan), "single": np.full((N, 1), np.nan),
            "hole": base["y_obs"].copy()}
    pats["single"][137] = 0.7
    pats["hole"][leaf0] = np.nan
    for tag, y in pats.items():
        cs = dict(base, y_obs=y)
        pl, lik, mean, var = run_hip(hip, cs)
        ref = run_levelwise(t, cs["locs"], cs["spec"], y, cs["c"]["R"])
        assert abs(lik - ref["lik"]) <= 1e-11 * max(abs(ref["lik"]), 1.0), tag
        assert np.max(np.abs(mean - ref["mean"])) < 1e-10, tag
        assert K.rel(np.sqrt(var), ref["sd"]) < 1e-9, tag
        if tag == "none":
            assert lik == 0.0 and np.all(mean == 0.0)
        pl.close()


def test_sharded_on_one_gpu_equals_single(hip):
    """Two 'ranks' emulated one after the other on the same GPU, exchanging the front buffer through
    the export/import entry points (what RCCL does on the device in bench.py --gpus N)."""
    from pymra_amd.sharding import shard_topology
    cs = K.load_case("g64m")
    pl, lik, mean, var = run_hip(hip, cs)
    pl.close()
    world = 2
    plans, bufs = [], []
    for r in range(world):
        lt, red = shard_topology(cs["topo"], world, r)
        p = hip.HipPlan(lt, 0)
        p.set_locs(cs["locs"]); p.set_obs(cs["y_obs"], cs["c"]["R"])
        p.set_kernel(cs["spec"].kind, cs["spec"].l, cs["spec"].sig, cs["spec"].scale)
        p.set_reduce_level(red)
        p.run(True, True, split=True)
        bufs.append(p.reduce_export())
        plans.append(p)
    tot = bufs[0] + bufs[1]
    m = np.zeros_like(mean); v = np.zeros_like(var)
    for p in plans:
        p.reduce_import(tot)
        p.resume()
        d, u = p.likelihood()
        assert abs(d + u - lik) <= 1e-12 * abs(lik)
        mm, vv = p.predict()
        m += mm; v += vv
        p.close()
    assert np.max(np.abs(m - mean)) < 1e-12 and np.max(np.abs(v - var)) < 1e-13


@pytest.mark.parametrize("case,world", [("c2", 4), ("c2", 8), ("g64m", 8), ("g64", 5)])
def test_sharded_4_and_8_ways_on_one_gpu(hip, case, world):
    """BASELINE config 4's layouts (4 GPUs: shard level 1; 8 GPUs: shard level 2 with orphan knot rows of two
    upper levels and reduce level 1) through the HIP path: all ranks emulated on one GPU with the split /
    export / import entry points, against the single-plan HIP result and the oracle."""
    from oracle.mra_levelwise import run_levelwise
    cs = K.load_case(case)
    pl, lik, mean, var = run_hip(hip, cs)
    pl.close()
    liks, m, v, s = K.emulate_world_on_one_gpu(hip, cs["topo"], cs["locs"], cs["y_obs"], cs["c"]["R"], cs["spec"], world)
    assert s == (1 if world <= 4 else 2)
    assert max(abs(l - lik) for l in liks) <= 1e-12 * abs(lik)
    assert np.max(np.abs(m - mean)) < 1e-11 and np.max(np.abs(v - var)) < 1e-12
    if case != "c2":
        ref = run_levelwise(cs["topo"], cs["locs"], cs["spec"], cs["y_obs"], cs["c"]["R"])
        assert abs(liks[0] - ref["lik"]) <= 1e-11 * abs(ref["lik"]) and np.max(np.abs(m - ref["mean"])) < 1e-10


def test_state_errors_are_status_codes_not_faults(hip):
    """(1) new observations on a plan that uses host-evaluated covariance blocks invalidate the blocks: mra_run must
    return MRA_ERR_STATE until they are uploaded again (it used to launch with a null block pointer);
    (2) a reduce level without communicator and without MRA_RUN_SPLIT must not silently skip the exchange;
    (3) a plan is usable again after an abandoned split run and after an error."""
    import pymra_amd.MRATools as mt
    cs = K.load_case("g32")
    cov = lambda a, b: np.asarray(mt.ExpCovFun(a, b, l=cs["c"]["l"]))
    pl = hip.HipPlan(cs["topo"], 0)
    pl.set_locs(cs["locs"]); pl.set_obs(cs["y_obs"], cs["c"]["R"])
    pl.upload_host_cov(cov, cs["locs"], cs["y_obs"])
    pl.run(True, True)
    lik0 = sum(pl.likelihood())
    y2 = cs["y_obs"].copy()
    y2[np.nonzero(np.isfinite(y2.ravel()))[0][:7]] = np.nan
    pl.set_obs(y2, cs["c"]["R"])
    with pytest.raises(hip.MraError) as e:
        pl.run(True, True)
    assert e.value.code == -4
    pl.upload_host_cov(cov, cs["locs"], y2)
    pl.run(True, True)
    assert sum(pl.likelihood()) != lik0
    pl.close()
    # (2) + (3)
    from pymra_amd.sharding import shard_topology
    lt, red = shard_topology(cs["topo"], 2, 0)
    p = hip.HipPlan(lt, 0)
    p.set_locs(cs["locs"]); p.set_obs(cs["y_obs"], cs["c"]["R"]); p.set_kernel(cs["spec"].kind, cs["spec"].l, cs["spec"].sig, 1.0)
    p.set_reduce_level(red)
    with pytest.raises(hip.MraError) as e:
        p.run(True, True)
    assert e.value.code == -4
    p.run(True, True, split=True)                   # abandoned: never resumed
    p.run(True, True, split=True)
    b1 = p.reduce_export()
    p.reduce_import(b1); p.resume()
    l1 = sum(p.likelihood())
    p.run(True, True, split=True); p.reduce_import(p.reduce_export()); p.resume()
    assert sum(p.likelihood()) == l1
    p.close()


def test_other_device_kernels_against_oracle(hip):
    """Matern52 / Gaussian / scaled kernels (pyMRA/MRATools.py:281-301) on the device."""
    import pymra_amd.MRATools as mt
    from oracle.mra_levelwise import run_levelwise
    base = K.load_case("g32")
    for spec in (mt.KernelSpec(mt.KIND_MATERN52, 0.25, 1.3), mt.KernelSpec(mt.KIND_GAUSSIAN, 0.05, 0.8),
                 mt.KernelSpec(mt.KIND_MATERN32, 0.3, 1.0, scale=2.5), mt.KernelSpec(mt.KIND_KANTER, 0.35),
                 mt.KernelSpec(mt.KIND_IDEN, 1.0, 1.0, scale=0.7)):
        cs = dict(base, spec=spec)
        pl, lik, mean, var = run_hip(hip, cs)
        ref = run_levelwise(cs["topo"], cs["locs"], spec, cs["y_obs"], cs["c"]["R"])
        assert abs(lik - ref["lik"]) <= 1e-9 * abs(ref["lik"])
        assert np.max(np.abs(mean - ref["mean"])) < 1e-8
        assert K.rel(np.sqrt(var), ref["sd"]) < 1e-7
        pl.close()


def test_api_errors(hip):
    import pymra_amd
    import pymra_amd.MRATools as mt
    cs = K.load_case("c1")
    cov = lambda a, b: mt.Matern32(a, b, l=0.3, sig=1.0)
    with pytest.raises(TypeError):
        pymra_amd.MRATree(cs["locs"], 2, cov, cs["y_obs"], 1, M=3, J=3)               # int R (MRANode.py:85-88)
    with pytest.raises(AttributeError):
        pymra_amd.MRATree(cs["locs"], 2, cov, cs["y_obs"], 1e-2, M=3)                 # 1-D needs J (MRATree.py:31-33)
    with pytest.raises(TypeError):
        pymra_amd.MRATree(cs["locs"], 2, "not a covariance", cs["y_obs"], 1e-2, M=3, J=3)
    tree = pymra_amd.MRATree(cs["locs"], 2, cov, cs["y_obs"], 1e-2, M=9, J=3)         # M clipped to 3
    assert tree.M == 3 and tree.J == 3 and tree.d == 1 and tree.r == 2
    assert abs(tree.getLikelihood()[0, 0] - float(cs["g"]["lik"])) < 1e-9
    assert np.array_equal(tree.obs_inds, np.where(np.isfinite(cs["y_obs"]))[0])


def test_device_kernels_match_numpy_to_ulps(hip):
    """ExpCovFun / Matern32 / Matern52 / Gaussian as the inference kernels evaluate them on the GPU vs
    the NumPy formulas of pyMRA/MRATools.py:265-301 (row a1-a3 of SURVEY section 8)."""
    import pymra_amd.MRATools as mt
    rng = np.random.RandomState(1)
    D = np.concatenate([[0.0], np.abs(rng.normal(size=20000)) * 0.7, 10.0 ** rng.uniform(-9, 2, size=20000)])
    x = np.zeros((1, 1))
    for kind, l, sig in ((mt.KIND_EXP, 0.3, 1.0), (mt.KIND_MATERN32, 0.3, 1.7), (mt.KIND_MATERN52, 2.0, 0.4),
                         (mt.KIND_GAUSSIAN, 0.8, 1.2), (mt.KIND_KANTER, 0.9, 1.0)):
        spec = mt.KernelSpec(kind, l, sig, 1.5)
        want = np.asarray(spec.evaluate(x, D.reshape(-1, 1))).ravel()
        got = hip.eval_kernel(kind, l, sig, 1.5, D)
        # exp(-t) carries the rounding of its argument: relative error ~ t * 2^-53, so compare where t <~ 30
        if kind == mt.KIND_KANTER:      # the taper's two terms cancel near D = 1: compare on the kernel's own scale
            assert np.max(np.abs(got - want)) < 1e-14 * 1.5
            continue
        big = want > 1e-13
        assert np.max(np.abs(got[big] - want[big]) / want[big]) < 2e-14, kind
        assert np.all(np.abs(got[~big] - want[~big]) <= 1e-11 * want[~big] + 1e-300)


def test_rccl_allreduce_path_single_rank(hip):
    """The on-device exchange (ncclAllReduce on the plan's stream, librccl loaded with dlopen) with a
    communicator of size 1: must reproduce the plain run bit for bit in lik and to rounding elsewhere."""
    from pymra_amd.sharding import sharded_run
    cs = K.load_case("g64m")
    pl, lik, mean, var = run_hip(hip, cs)
    pl.close()
    p = hip.HipPlan(cs["topo"], 0)
    p.set_locs(cs["locs"]); p.set_obs(cs["y_obs"], cs["c"]["R"])
    p.set_kernel(cs["spec"].kind, cs["spec"].l, cs["spec"].sig, cs["spec"].scale)
    p.comm_init(hip.comm_unique_id(), 1, 0)
    sharded_run(p, 0, None, True, True)                    # reduce level 0: the root front goes through RCCL
    d, u = p.likelihood()
    m2, v2 = p.predict()
    assert abs(d + u - lik) <= 1e-13 * abs(lik)
    assert np.max(np.abs(m2 - mean)) < 1e-12 and np.max(np.abs(v2 - var)) < 1e-13
    p.close()


def test_opaque_callable_and_dense_matrix_cov(hip):
    """cov plug-in surface (pyMRA/MRANode.py:73-80, 381-384): an arbitrary callable, and a dense N x N
    matrix, go through the host-evaluated block path and agree with the device-kernel path."""
    import pymra_amd
    import pymra_amd.MRATools as mt
    for name in ("g32", "c1", "u3"):
        cs = K.load_case(name)
        c = cs["c"]
        np.random.seed(123)
        dev = pymra_amd.MRATree(cs["locs"], c["r"], lambda a, b: cs["spec"].evaluate(a, b) if False else
                                (mt.ExpCovFun(a, b, l=c["l"]) if c["kern"] == "exp" else mt.Matern32(a, b, l=c["l"], sig=c["sig"])),
                                cs["y_obs"], c["R"], M=c["M"], J=c["J"])
        assert dev.kernel is not None
        opaque = lambda a, b: np.asarray(cs["spec"].evaluate(a, b)) + 0.0          # not recognisable by the probe
        np.random.seed(123)
        t1 = pymra_amd.MRATree(cs["locs"], c["r"], opaque, cs["y_obs"], c["R"], M=c["M"], J=c["J"])
        assert t1.kernel is None
        dense = np.asarray(cs["spec"].evaluate(cs["locs"], cs["locs"]))
        np.random.seed(123)
        t2 = pymra_amd.MRATree(cs["locs"], c["r"], np.matrix(dense), cs["y_obs"], c["R"], M=c["M"], J=c["J"])
        tol = 1e-7 if name == "u3" else 1e-10
        for t in (t1, t2):
            assert abs(t.getLikelihood()[0, 0] - dev.getLikelihood()[0, 0]) <= tol * abs(dev.getLikelihood()[0, 0])
            assert np.max(np.abs(np.asarray(t.predict()[0]) - np.asarray(dev.predict()[0]))) < max(tol, 1e-9)
            assert K.rel(t.predict()[1], dev.predict()[1]) < (1e-5 if name == "u3" else 1e-8)


def test_circular_distance_kernel(hip):
    """1-D torus distances (pyMRA/MRATools.py:232-239) on the device vs the NumPy formula and vs the oracle."""
    import pymra_amd
    import pymra_amd.MRATools as mt
    from oracle.mra_levelwise import run_levelwise
    cs = K.load_case("c1")
    c = cs["c"]
    spec = mt.KernelSpec(mt.KIND_EXP, 0.2, 1.0, 1.0, circular=True)
    tree = pymra_amd.MRATree(cs["locs"], c["r"], lambda a, b: mt.ExpCovFun(a, b, l=0.2, circular=True), cs["y_obs"], c["R"],
                             M=c["M"], J=c["J"])
    assert tree.kernel.circular
    ref = run_levelwise(tree.topology, cs["locs"], spec, cs["y_obs"], c["R"])
    assert abs(tree.getLikelihood()[0, 0] - ref["lik"]) <= 1e-10 * abs(ref["lik"])
    assert np.max(np.abs(np.asarray(tree.predict()[0]).ravel() - ref["mean"])) < 1e-10
    assert K.rel(tree.predict()[1], ref["sd"]) < 1e-9


def test_level_by_level_kernels_equal_fused_kernels(hip):
    """Regular trees run the fused cascade kernels; the general level-by-level kernels (used for
    irregular trees and host-evaluated covariances) must give the same numbers."""
    for name in ("c1", "g64m", "c2"):
        cs = K.load_case(name)
        pl, lik, mean, var = run_hip(hip, cs)
        pl.set_option(2, 0)                       # MRA_OPT_FUSED off
        pl.set_option(3, 0)                       # MRA_OPT_GEMM_LDS off: direct-load GEMM
        pl.run(True, True)
        d, u = pl.likelihood()
        m2, v2 = pl.predict()
        assert abs(d + u - lik) <= 1e-12 * abs(lik)
        assert np.max(np.abs(m2 - mean)) < 1e-11 and K.rel(np.sqrt(v2), np.sqrt(var)) < 1e-10
        pl.run(True, False)                       # likelihood only on the level-by-level path: C from the small gathered product
        d, u = pl.likelihood()
        assert abs(d + u - lik) <= 1e-12 * abs(lik)
        pl.close()


@pytest.mark.parametrize("name", ["g64m", "c2", "g128m"])
def test_single_launch_chains_equal_per_level_launches(hip, name):
    """MRA_OPT_KNOT_CHAIN (knot pass of all levels in one launch), MRA_OPT_FRONT_FUSED (assembly + partial Cholesky +
    Schur per level / children's Ut -> parent front in one launch) and MRA_OPT_LEAF_GEMM (leaf-resident products) against
    the launch sequences / kernels they replace."""
    cs = K.load_case(name)
    pl, lik, mean, var = run_hip(hip, cs)
    for opts in ((4, 0), (5, 0), (6, 0), (6, 2), (7, 0), (7, 1), (7, 1, 10, 1), (8, 0), (11, 0), (11, 2), (13, 0), (13, 0, 7, 1), (4, 0, 5, 0, 6, 0, 7, 0, 8, 0, 11, 0)):
        pl.set_option(4, 1); pl.set_option(5, 1); pl.set_option(6, 1); pl.set_option(7, 2); pl.set_option(8, 1); pl.set_option(10, 2)
        pl.set_option(11, 1); pl.set_option(13, 1)
        for o, v in zip(opts[::2], opts[1::2]):
            pl.set_option(o, v)
        pl.run(True, True)
        d, u = pl.likelihood()
        m2, v2 = pl.predict()
        assert abs(d + u - lik) <= 1e-12 * abs(lik), opts
        assert np.max(np.abs(m2 - mean)) < 1e-11 and K.rel(np.sqrt(v2), np.sqrt(var)) < 1e-10, opts
    pl.close()


@pytest.mark.parametrize("n,r,M", [(768, 16, 7), (768, 32, 7), (384, 16, 6), (192, 32, 4)])
def test_fused_cascades_at_every_depth(hip, n, r, M):
    """The fused cascades are instantiated for up to 6 levels (three workgroups per CU in the predictive cascade) and up to
    8; config C3 only reaches the first.  Regular trees of 7, 6 and 4 levels with one and two column tiles per level:
    the fused launches must have run (kernel statistics) and agree with the level-by-level kernels on the same plan; the
    6-level tree is also checked against the oracle."""
    import pymra_amd.MRATools as mt
    from pymra_amd.topology import build_topology
    np.random.seed(5)
    locs = mt.genLocations2d(Nx=n, Ny=n)
    y = np.random.normal(size=(n * n, 1))
    y_obs = np.where(np.random.uniform(size=(n * n, 1)) < 0.4, y, np.nan)
    topo = build_topology(locs, r, M, 4)
    assert [int(v) for v in np.diff(topo.level_ptr)] == [4 ** k for k in range(M + 1)]
    spec = mt.KernelSpec(mt.KIND_MATERN32, 0.25, 1.2)
    cs = dict(topo=topo, locs=locs, y_obs=y_obs, spec=spec, c=dict(R=2e-2))
    pl = hip.HipPlan(topo, 0); pl.set_locs(locs); pl.set_obs(y_obs, 2e-2); pl.set_kernel(spec.kind, spec.l, spec.sig, spec.scale)
    pl.set_option(1, 1)                                   # kernel timing: launches per kernel family
    pl.run(True, True)
    d, u = pl.likelihood(); lik = d + u
    mean, var = pl.predict()
    ran = {k["name"]: k["launches"] for k in pl.kernel_stats()}
    assert any("k_predict_cascade" in nm and c > 0 for nm, c in ran.items()), ran
    assert any("k_prior_cascade row pass" in nm and c > 0 for nm, c in ran.items()), ran
    pl.set_option(8, 0)                                   # leaf update as a product of its own, then the cascade
    pl.run(True, True)
    m1, v1 = pl.predict()
    assert abs(sum(pl.likelihood()) - lik) <= 1e-12 * abs(lik)
    assert np.max(np.abs(m1 - mean)) < 1e-10 and K.rel(np.sqrt(v1), np.sqrt(var)) < 1e-9
    pl.set_option(8, 1); pl.set_option(2, 0)              # MRA_OPT_FUSED off: level-by-level kernels
    pl.run(True, True)
    m2, v2 = pl.predict()
    assert abs(sum(pl.likelihood()) - lik) <= 1e-11 * abs(lik)
    assert np.max(np.abs(m2 - mean)) < 1e-9 and K.rel(np.sqrt(v2), np.sqrt(var)) < 1e-8
    if M == 6:
        from oracle.mra_levelwise import run_levelwise
        ref = run_levelwise(topo, locs, spec, y_obs, 2e-2)
        assert abs(lik - ref["lik"]) <= 1e-10 * abs(ref["lik"])
        assert np.max(np.abs(mean - ref["mean"])) < 1e-9 and K.rel(np.sqrt(var), ref["sd"]) < 1e-8
    pl.close()


@pytest.mark.parametrize("n,r,M,fused", [(128, 64, 3, True), (128, 64, 5, False), (96, 48, 2, True), (64, 16, 3, True)])
def test_wide_blocks_and_deep_trees(hip, n, r, M, fused):
    """r0 = 64 (4 column tiles per level: C5-like) on the fused path (CWT=4) and, when the cascades'
    register budget is exceeded (4 tiles x 5 levels), on the general level-by-level path; r0 = 48 (3 tiles) is
    not a fused block width and also takes the general path."""
    import pymra_amd.MRATools as mt
    from pymra_amd.topology import build_topology
    from oracle.mra_levelwise import run_levelwise
    np.random.seed(21)
    locs = mt.genLocations2d(Nx=n, Ny=n)
    y = np.random.normal(size=(n * n, 1))
    y_obs = np.where(np.random.uniform(size=(n * n, 1)) < 0.35, y, np.nan)
    topo = build_topology(locs, r, M, 4)
    spec = mt.KernelSpec(mt.KIND_MATERN32, 0.25, 1.2)
    cs = dict(topo=topo, locs=locs, y_obs=y_obs, spec=spec, c=dict(R=2e-2))
    pl, lik, mean, var = run_hip(hip, cs)
    ref = run_levelwise(topo, locs, spec, y_obs, 2e-2)
    assert abs(lik - ref["lik"]) <= 1e-10 * abs(ref["lik"])
    assert np.max(np.abs(mean - ref["mean"])) < 1e-9
    assert K.rel(np.sqrt(var), ref["sd"]) < 1e-8
    pl.close()


def test_large_fully_observed_leaves(hip):
    """Leaves with several hundred observations (more than the 192 that fit the LDS-resident solve): the
    general panel Cholesky path."""
    import pymra_amd.MRATools as mt
    from pymra_amd.topology import build_topology
    from oracle.mra_levelwise import run_levelwise
    np.random.seed(4)
    n = 48
    locs = mt.genLocations2d(Nx=n, Ny=n)
    y_obs = np.random.normal(size=(n * n, 1))                     # everything observed
    topo = build_topology(locs, 16, 1, 4)
    assert topo.n_nodes == 5 and (topo.node_row1[1] - topo.node_row0[1]) >= 576
    spec = mt.KernelSpec(mt.KIND_EXP, 0.3)
    cs = dict(topo=topo, locs=locs, y_obs=y_obs, spec=spec, c=dict(R=5e-2))
    pl, lik, mean, var = run_hip(hip, cs)
    ref = run_levelwise(topo, locs, spec, y_obs, 5e-2)
    assert abs(lik - ref["lik"]) <= 1e-10 * abs(ref["lik"])
    assert np.max(np.abs(mean - ref["mean"])) < 1e-9
    assert K.rel(np.sqrt(var), ref["sd"]) < 1e-8
    pl.run(True, False)
    d, u = pl.likelihood()
    assert abs(d + u - lik) <= 1e-12 * abs(lik)
    pl.close()


def test_not_positive_definite_is_reported(hip):
    """A numerically singular knot block: the reference drops into pdb (pyMRA/MRANode.py:386-390); here the
    Cholesky flags the non-positive pivot and the run raises MRA_ERR_NOT_SPD."""
    import pymra_amd
    import pymra_amd.MRATools as mt
    cs = K.load_case("c1")
    cov = lambda a, b: mt.GaussianCovFun(a, b, l=50.0, sig=1.0)      # 16 knots, condition number ~1e60
    with pytest.raises(hip.MraError) as ei:
        pymra_amd.MRATree(cs["locs"], 16, cov, cs["y_obs"], 1e-2, M=1, J=3)
    assert ei.value.code == -3


def test_a_failed_pass_leaves_nothing_behind(hip):
    """Some buffers are prepared once and trusted afterwards (the identity rows of the leaves' phantom observations survive the
    in-place Cholesky exactly).  A pass that fails - kernel parameters that make a knot block numerically singular: NaN spreads
    through the leaves - must not poison the next pass on the same plan."""
    import pymra_amd.MRATools as mt
    cs = K.load_case("c2")
    pl, lik, mean, var = run_hip(hip, cs)
    s = cs["spec"]
    pl.set_kernel(mt.KIND_GAUSSIAN, 50.0, 1.0, 1.0)                 # a nearly constant field: numerically singular knot blocks
    with pytest.raises(hip.MraError) as ei:
        pl.run(True, True)
    assert ei.value.code == -3
    pl.set_kernel(s.kind, s.l, s.sig, s.scale)
    pl.run(True, True)
    d, u = pl.likelihood()
    m2, v2 = pl.predict()
    assert abs(d + u - lik) <= 1e-13 * abs(lik)
    assert np.max(np.abs(m2 - mean)) < 1e-12 and K.rel(np.sqrt(v2), np.sqrt(var)) < 1e-12
    pl.close()


def test_repeated_passes_are_bit_identical(hip):
    """Nothing a pass leaves behind may change the next one: twenty passes on one plan (likelihood-only passes mixed in, a detour
    through other kernel parameters) reproduce the first bit for bit (tools/soak_passes.py does 500 on C3)."""
    cs = K.load_case("c2")
    pl, lik, mean, var = run_hip(hip, cs)
    mean, var = mean.copy(), var.copy()
    s = cs["spec"]
    for i in range(20):
        if i == 7:
            pl.set_kernel(s.kind, 0.6 * s.l, s.sig, s.scale); pl.run(True, True)
            pl.set_kernel(s.kind, s.l, s.sig, s.scale)
        pl.run(True, i % 3 != 0)
    pl.run(True, True)
    d, u = pl.likelihood()
    m2, v2 = pl.predict()
    assert d + u == lik and np.array_equal(m2, mean) and np.array_equal(v2, var)
    pl.close()


def test_fused_path_with_leaves_above_128_observations(hip):
    """Regular tree whose leaves hold 144 observations each (192 x 192 grid, r0 = 32, M = 4, every row observed: nine
    observation tiles, more than the eight the fused row solve / predictive cascade take): those leaves go through the
    12-tile row solve and the PLAIN update product (gLeafUpdatePlain + n_trsm_small) on the fused path.  A lost upload of
    that descriptor array once faulted the GPU here while the suite stayed green (no other fused-path case has such
    leaves).  Checked against the oracle, against the level-by-level kernels, and with the leaf-stage options in all
    combinations, likelihood-only and predict passes back to back on one plan."""
    import pymra_amd.MRATools as mt
    from pymra_amd.topology import build_topology
    from oracle.mra_levelwise import run_levelwise
    np.random.seed(17)
    n, r, M = 192, 32, 4
    locs = mt.genLocations2d(Nx=n, Ny=n)
    y_obs = np.random.normal(size=(n * n, 1))                     # everything observed
    topo = build_topology(locs, r, M, 4)
    leaf_rows = (topo.node_row1 - topo.node_row0)[topo.node_leaf]
    assert [int(v) for v in np.diff(topo.level_ptr)] == [4 ** k for k in range(M + 1)] and leaf_rows.min() >= 129 and leaf_rows.max() <= 192
    spec = mt.KernelSpec(mt.KIND_MATERN32, 0.25, 1.2)
    R = 2e-2
    pl = hip.HipPlan(topo, 0); pl.set_locs(locs); pl.set_obs(y_obs, R); pl.set_kernel(spec.kind, spec.l, spec.sig, spec.scale)
    pl.set_option(1, 1)
    pl.run(True, True)
    lik = sum(pl.likelihood())
    mean, var = pl.predict()
    ran = {k["name"]: k["launches"] for k in pl.kernel_stats()}
    assert any("k_predict_cascade" in nm and c > 0 for nm, c in ran.items()), ran      # the fused path did run
    pl.set_option(1, 0)
    ref = run_levelwise(topo, locs, spec, y_obs, R)
    assert abs(lik - ref["lik"]) <= 1e-10 * abs(ref["lik"])
    assert np.max(np.abs(mean - ref["mean"])) < 1e-9 and K.rel(np.sqrt(var), ref["sd"]) < 1e-8
    for o7 in (0, 1, 2):
        for o8 in (0, 1):
            for o13 in (0, 1):
                pl.set_option(7, o7); pl.set_option(8, o8); pl.set_option(13, o13)
                pl.run(True, False)
                assert abs(sum(pl.likelihood()) - lik) <= 1e-12 * abs(lik), (o7, o8, o13)
                pl.run(True, True)
                m2, v2 = pl.predict()
                assert abs(sum(pl.likelihood()) - lik) <= 1e-12 * abs(lik), (o7, o8, o13)
                assert np.max(np.abs(m2 - mean)) < 1e-10 and K.rel(np.sqrt(v2), np.sqrt(var)) < 1e-9, (o7, o8, o13)
    pl.set_option(7, 2); pl.set_option(8, 1); pl.set_option(13, 1); pl.set_option(2, 0)      # level-by-level kernels
    pl.run(True, True)
    m3, v3 = pl.predict()
    assert abs(sum(pl.likelihood()) - lik) <= 1e-11 * abs(lik)
    assert np.max(np.abs(m3 - mean)) < 1e-9 and K.rel(np.sqrt(v3), np.sqrt(var)) < 1e-8
    pl.close()


def test_chol_tiles_write_back_survives_repeated_passes(hip):
    """k_chol_tiles factorises the leaves' C blocks in place and relies on the phantom observations' identity rows surviving
    from pass to pass (they are written once, cphantom_valid).  Force that kernel (option 11 = 2) ONCE, then run predict /
    likelihood-only / predict passes without touching any option in between: bit-identical results."""
    cs = K.load_case("g64m")
    pl, lik, mean, var = run_hip(hip, cs)
    mean, var = mean.copy(), var.copy()
    pl.set_option(11, 2)
    pl.run(True, True)
    lik_t = sum(pl.likelihood())
    m_t, v_t = (a.copy() for a in pl.predict())
    assert abs(lik_t - lik) <= 1e-12 * abs(lik) and np.max(np.abs(m_t - mean)) < 1e-11
    for want_predict in (False, True, False, True):
        pl.run(True, want_predict)
        assert sum(pl.likelihood()) == lik_t
        if want_predict:
            m2, v2 = pl.predict()
            assert np.array_equal(m2, m_t) and np.array_equal(v2, v_t)
    pl.close()


def test_no_buffer_relies_on_zero_initialised_memory(hip):
    """The device-block cache hands the blocks of a destroyed plan to the next plan as they are.  With MRA_POOL_POISON=1 every
    block the library hands out (fresh or reused, large or small) is filled with NaNs first: the oracle parity cases must come
    out the same - fused path, general level-by-level path (1-D, irregular), sharded split run, likelihood-only first pass."""
    import os
    from oracle.mra_levelwise import run_levelwise
    from pymra_amd.sharding import shard_topology
    os.environ["MRA_POOL_POISON"] = "1"
    try:
        for name in ("g32", "c1", "u3", "g64m", "t201"):
            cs = K.load_case(name)
            ref = run_levelwise(cs["topo"], cs["locs"], cs["spec"], cs["y_obs"], cs["c"]["R"])
            for first_predict in (False, True):
                pl = hip.HipPlan(cs["topo"], 0)
                pl.set_locs(cs["locs"]); pl.set_obs(cs["y_obs"], cs["c"]["R"])
                s = cs["spec"]
                pl.set_kernel(s.kind, s.l, s.sig, s.scale)
                pl.run(True, first_predict)
                assert abs(sum(pl.likelihood()) - ref["lik"]) <= 1e-10 * abs(ref["lik"]), (name, first_predict)
                pl.run(True, True)
                mean, var = pl.predict()
                own = cs["topo"].perm[cs["topo"].in_leaf]
                assert np.max(np.abs(mean[own] - ref["mean"][own])) < 1e-9, name
                assert K.rel(np.sqrt(var[own]), ref["sd"][own]) < 1e-8, name
                pl.close()
        cs = K.load_case("g64m")
        ref = run_levelwise(cs["topo"], cs["locs"], cs["spec"], cs["y_obs"], cs["c"]["R"])
        liks, mean, var, _ = K.emulate_world_on_one_gpu(hip, cs["topo"], cs["locs"], cs["y_obs"], cs["c"]["R"], cs["spec"], 8)
        assert max(abs(l - ref["lik"]) for l in liks) <= 1e-10 * abs(ref["lik"])
        assert np.max(np.abs(mean - ref["mean"])) < 1e-9 and K.rel(np.sqrt(var), ref["sd"]) < 1e-8
    finally:
        del os.environ["MRA_POOL_POISON"]
        hip.release_cached_memory()


@pytest.mark.parametrize("n,r,M", [(64, 16, 2), (200, 16, 3), (256, 32, 3)])
def test_one_call_constructor_equals_the_stepwise_path(hip, n, r, M):
    """MRATree(...) builds large 2-D trees through mra_plan_create_replay_2d (tree replay with the plan constructed beside the
    knot draws).  Same seed, same inputs: the tree, NumPy's RNG state afterwards, the likelihood and the predictions must be
    bit-identical to build_topology -> HipPlan -> set_locs -> set_obs."""
    import pymra_amd.MRATools as mt
    from pymra_amd import MRATree
    from pymra_amd.topology import build_topology
    np.random.seed(31)
    locs = mt.genLocations2d(Nx=n, Ny=n)
    y_obs = np.where(np.random.uniform(size=(n * n, 1)) < 0.4, np.random.normal(size=(n * n, 1)), np.nan)
    st0 = np.random.get_state()
    cov = lambda a, b: mt.Matern32(a, b, l=0.25, sig=1.1)
    tree = MRATree(locs, r, cov, y_obs, 2e-2, M=M, J=4)
    st1 = np.random.get_state()
    lik = float(tree.getLikelihood()[0, 0])
    mean, sd = tree.predict()
    np.random.set_state(st0)
    topo = build_topology(locs, r, M, 4)
    st2 = np.random.get_state()
    assert np.array_equal(st1[1], st2[1]) and st1[2] == st2[2]
    for f in ("perm", "src", "in_leaf", "node_row0", "node_row1", "knot_ptr", "knot_rows", "child_list", "order_preorder"):
        assert np.array_equal(getattr(tree.topology, f), getattr(topo, f)), f
    pl = hip.HipPlan(topo, 0); pl.set_locs(locs); pl.set_obs(y_obs, 2e-2); pl.set_kernel(mt.KIND_MATERN32, 0.25, 1.1, 1.0)
    pl.run(True, True)
    m2, v2 = pl.predict()
    assert sum(pl.likelihood()) == lik
    assert np.array_equal(np.asarray(mean).ravel(), m2) and np.array_equal(sd, np.sqrt(v2))
    pl.close(); tree.plan.close()


def test_root_view_attributes(hip):
    import pymra_amd
    import pymra_amd.MRATools as mt
    cs = K.load_case("g32")
    c = cs["c"]
    np.random.seed(7); K.mg.make_inputs(c)
    tree = pymra_amd.MRATree(cs["locs"], c["r"], lambda a, b: mt.ExpCovFun(a, b, l=c["l"]), cs["y_obs"], c["R"], M=c["M"], J=c["J"])
    g = cs["g"]
    assert np.array_equal(tree.root.kInds, g["knots"][g["knot_ptr"][0]:g["knot_ptr"][1]])     # the reference's root knots
    assert tree.root.knots.shape == (c["r"], 2) and tree.root.B.shape == (len(cs["locs"]), c["r"])
    assert np.allclose(np.asarray(tree.root.kInv), np.asarray(mt.ExpCovFun(tree.root.knots, tree.root.knots, l=c["l"])))
    assert abs(tree.root.d[0, 0] - float(g["d"])) <= 1e-8 * abs(float(g["d"])) and abs(tree.root.u[0, 0] - float(g["u"])) <= 1e-8 * abs(float(g["u"]))


def test_plan_reuse_for_mle(hip):
    """MLE loops (README.md:96-104) re-evaluate the likelihood for new kernel parameters on the SAME tree:
    `reevaluate` must equal a fresh MRATree built with the same seed."""
    import pymra_amd
    import pymra_amd.MRATools as mt
    cs = K.load_case("g64m")
    c = cs["c"]

    def fresh(kappa, sig):
        np.random.seed(c["seed"]); K.mg.make_inputs(c)          # RNG where the recipe leaves it: same knots
        return pymra_amd.MRATree(cs["locs"], c["r"], lambda a, b: sig * mt.Matern32(a, b, l=kappa), cs["y_obs"], c["R"],
                                 M=c["M"], J=c["J"])
    base = fresh(0.3, 1.0)
    for kappa, sig in ((0.2, 1.0), (0.45, 1.7)):
        lik = base.reevaluate(lambda a, b: sig * mt.Matern32(a, b, l=kappa))
        ref = fresh(kappa, sig)
        assert isinstance(lik, np.matrix) and abs(lik[0, 0] - ref.getLikelihood()[0, 0]) <= 1e-12 * abs(ref.getLikelihood()[0, 0])
    full = base.reevaluate(lambda a, b: mt.Matern32(a, b, l=0.3, sig=1.0), want_predict=True)
    assert abs(full[0, 0] - float(cs["g"]["lik"])) <= 1e-9 * abs(float(cs["g"]["lik"]))
    assert np.max(np.abs(np.asarray(base.predict()[0]).ravel() - cs["g"]["mean"])) < 1e-7


@pytest.mark.parametrize("frac", [0.2, 0.6])
def test_deep_wide_tree_leaf_update_inside_predict_hi(hip, frac):
    """Deep 64-wide trees whose leaves have at most 64 padded observations (BASELINE config 5: 64 rows per leaf, up to four k tiles) take
    the leaf update W[S, anc | y] -= Tt Ut^T inside k_predict_hi - on the sixteen deep tiles in registers, on every coarse tile as it
    passes through the sweep - instead of as a pass over all of W.  Leaves with four, three, two, one and no observation tile, against the
    separate product (option 16 = 0) and against the CPU oracle."""
    import pymra_amd.MRATools as mt
    from pymra_amd.topology import build_topology
    from oracle.mra_levelwise import run_levelwise
    n, r, M = 256, 64, 5
    np.random.seed(29)
    locs = mt.genLocations2d(Nx=n, Ny=n)
    y = np.random.normal(size=(n * n, 1))
    y_obs = np.where(np.random.uniform(size=(n * n, 1)) < frac, y, np.nan)
    topo = build_topology(locs, r, M, 4)
    leaves = [i for i in range(topo.n_nodes) if topo.node_leaf[i]]
    for i in (leaves[0], leaves[501], leaves[-1]):                 # three leaves without any observation
        rows = topo.perm[int(topo.node_row0[i]):int(topo.node_row1[i])]
        y_obs[rows[rows >= 0]] = np.nan
    nobs = np.array([np.isfinite(y_obs[topo.perm[int(topo.node_row0[i]):int(topo.node_row1[i])][topo.perm[int(topo.node_row0[i]):int(topo.node_row1[i])] >= 0]]).sum() for i in leaves])
    if frac < 0.5: assert nobs.max() <= 32 and (nobs <= 16).any() and (nobs > 16).any() and (nobs == 0).sum() >= 3      # k tiles: 0, 1, 2
    else: assert ((nobs > 32) & (nobs <= 48)).any() and (nobs > 48).any() and (nobs == 0).sum() >= 3                     # ... 3, 4
    spec = mt.KernelSpec(mt.KIND_MATERN32, 0.25, 1.2)
    cs = dict(topo=topo, locs=locs, y_obs=y_obs, spec=spec, c=dict(R=2e-2))
    pl, lik, mean, var = run_hip(hip, cs)
    assert pl.get_option(16) == 1
    names = [k["name"] for k in pl.kernel_stats() if k["launches"]]
    assert any("k_predict_hi" in k for k in names)
    pl.set_option(16, 0)                                            # the update as a product of its own
    pl.run(True, True)
    d, u = pl.likelihood()
    m2, v2 = pl.predict()
    assert abs(d + u - lik) <= 1e-13 * abs(lik)
    assert np.max(np.abs(m2 - mean)) < 1e-11 and K.rel(np.sqrt(v2), np.sqrt(var)) < 1e-11
    pl.close()
    ref = run_levelwise(topo, locs, spec, y_obs, 2e-2)
    assert abs(lik - ref["lik"]) <= 1e-10 * abs(ref["lik"])
    assert np.max(np.abs(mean - ref["mean"])) < 1e-9
    assert K.rel(np.sqrt(var), ref["sd"]) < 1e-8


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6, 7, 8])
def test_random_geometries(hip, seed):
    """Seeded random problems away from the benchmark geometry: rectangular grids, jittered and fully
    scattered 2-D locations, every kernel family, r0 that is not a multiple of 16, shallow and deep trees,
    sparse and dense observations.  HIP path vs the level-wise oracle on the same tree."""
    import pymra_amd.MRATools as mt
    from pymra_amd.topology import build_topology
    from oracle.mra_levelwise import run_levelwise
    rng = np.random.RandomState(1000 + seed)
    np.random.seed(seed)                      # the 2-D knot draws of the tree replay use the global RNG
    nx, ny = int(rng.randint(24, 72)), int(rng.randint(24, 72))
    kind = seed % 3
    if kind == 0:
        locs = mt.genLocations2d(Nx=nx, Ny=ny)
    elif kind == 1:
        locs = mt.genLocations2d(Nx=nx, Ny=ny) + rng.uniform(-0.3, 0.3, size=(nx * ny, 2)) / max(nx, ny)
    else:
        locs = rng.uniform(0, 1, size=(nx * ny, 2))
    N = len(locs)
    r = int(rng.choice([5, 8, 16, 20, 32]))
    M = int(rng.randint(1, 4))
    frac = float(rng.choice([0.1, 0.4, 0.9]))
    y = rng.normal(size=(N, 1))
    y_obs = np.where(rng.uniform(size=(N, 1)) < frac, y, np.nan)
    specs = [mt.KernelSpec(mt.KIND_EXP, 0.2), mt.KernelSpec(mt.KIND_MATERN32, 0.15, 1.3), mt.KernelSpec(mt.KIND_MATERN52, 0.2, 0.7),
             mt.KernelSpec(mt.KIND_GAUSSIAN, 0.05, 1.0), mt.KernelSpec(mt.KIND_MATERN32, 0.4, 1.0, 2.5)]
    spec = specs[seed % len(specs)]
    R = float(rng.choice([1e-3, 5e-2, 0.5]))
    topo = build_topology(locs, r, M, 4)
    cs = dict(topo=topo, locs=locs, y_obs=y_obs, spec=spec, c=dict(R=R))
    pl, lik, mean, var = run_hip(hip, cs)
    ref = run_levelwise(topo, locs, spec, y_obs, R)
    assert abs(lik - ref["lik"]) <= 1e-9 * max(1.0, abs(ref["lik"]))
    assert np.max(np.abs(mean - ref["mean"])) < 1e-8
    assert np.max(np.abs(np.sqrt(np.maximum(var, 0)) - ref["sd"])) < 1e-7
    # likelihood-only rerun on the same plan
    pl.run(True, False)
    d, u = pl.likelihood()
    assert abs(d + u - lik) <= 1e-11 * max(1.0, abs(lik))
    # ... whose row cascade walks gathered tiles of the OBSERVED rows only (option 17, the default where the cascade stages all its
    # levels at once); every row's arithmetic is the same whichever tile it sits in: bit-identical to the pass over all rows
    pl.set_option(17, 0)
    pl.run(True, False)
    d2, u2 = pl.likelihood()
    assert (d2, u2) == (d, u)
    pl.set_option(17, 1)
    pl.set_option(13, 0)                                   # ... also when the cascade scatters Ut itself instead of the row solve gathering it
    pl.run(True, False)
    d3, u3 = pl.likelihood()
    assert abs(d3 + u3 - (d + u)) <= 1e-12 * max(1.0, abs(d + u))
    pl.set_option(13, 1)
    # the level-by-level kernels on the same plan: the prior of a level in one launch (option 15, the default there) and as
    # residual product + gather + factorisation + row solve
    for one_launch in (1, 0):
        pl.set_option(2, 0); pl.set_option(15, one_launch)
        pl.run(True, True)
        d, u = pl.likelihood()
        m2, v2 = pl.predict()
        assert abs(d + u - lik) <= 1e-10 * max(1.0, abs(lik)), one_launch
        assert np.max(np.abs(m2 - mean)) < 1e-9 and np.max(np.abs(np.sqrt(np.maximum(v2, 0)) - np.sqrt(np.maximum(var, 0)))) < 1e-8, one_launch
    # ... and their likelihood-only passes over the rows a likelihood needs (observed rows and knots) against all rows: bit-identical
    pl.set_option(15, 1)
    liks = []
    for rows_needed in (1, 0):
        pl.set_option(17, rows_needed)
        pl.run(True, False)
        liks.append(pl.likelihood())
    assert liks[0] == liks[1]
    assert abs(sum(liks[0]) - lik) <= 1e-10 * max(1.0, abs(lik))
    pl.close()


def test_leaves_without_any_observation(hip):
    """A leaf with no observed row has a zero-sized panel (nop = 0).  The fused row solve + update (MRA_OPT_LEAF_SOLVE = 1, the
    default of 8-way sharded runs) and the update folded into the predictive cascade (MRA_OPT_PRED_UPDATE = 1) must skip it
    without reading past the panel allocation - the LAST leaf is one of the empty ones - and agree with the level-by-level
    kernels and with the CPU oracle."""
    from oracle.mra_levelwise import run_levelwise
    cs = dict(K.load_case("c2"))
    t = cs["topo"]
    y = np.array(cs["y_obs"], dtype=np.float64).reshape(-1, 1)
    leaves = [i for i in range(t.n_nodes) if t.node_leaf[i]]
    for i in (leaves[0], leaves[37], leaves[-1]):
        rows = t.perm[int(t.node_row0[i]):int(t.node_row1[i])]
        y[rows[rows >= 0]] = np.nan
    cs["y_obs"] = y
    ref = run_levelwise(t, cs["locs"], cs["spec"], y, cs["c"]["R"])
    pl, lik, mean, var = run_hip(hip, cs)
    assert abs(lik - ref["lik"]) <= 1e-11 * abs(ref["lik"])
    assert np.max(np.abs(mean - ref["mean"])) < 1e-10 and K.rel(np.sqrt(var), ref["sd"]) < 1e-9
    for opts in ((7, 1), (8, 1, 7, 0), (8, 0, 7, 0), (11, 0), (13, 0), (13, 0, 7, 1), (2, 0)):
        pl.set_option(2, 1); pl.set_option(7, 2); pl.set_option(8, 1); pl.set_option(11, 1); pl.set_option(13, 1)
        for o, v in zip(opts[::2], opts[1::2]):
            pl.set_option(o, v)
        pl.run(True, True)
        d, u = pl.likelihood()
        m2, v2 = pl.predict()
        assert abs(d + u - lik) <= 1e-12 * abs(lik), opts
        assert np.max(np.abs(m2 - mean)) < 1e-11 and K.rel(np.sqrt(v2), np.sqrt(var)) < 1e-10, opts
    pl.close()


@pytest.mark.parametrize("n,r,M,oracle", [(256, 64, 5, True), (512, 64, 6, True), (1024, 64, 7, False)])
def test_deep_wide_trees_two_kernel_predictive_cascade(hip, n, r, M, oracle):
    """Trees with 64-wide blocks and 5 - 8 non-leaf levels (BASELINE config 5 is the 8-level one, tests/test_gpu_fullsize.py) have
    too many tiles per row for the one-kernel cascades: their predictive pass is k_predict_hi (four deepest levels in registers, one
    sweep over the coarse columns) + k_predict_cascade (coarse levels).  Against the level-by-level kernels at every depth and
    against the CPU oracle where it finishes in seconds."""
    import pymra_amd.MRATools as mt
    from pymra_amd.topology import build_topology
    from oracle.mra_levelwise import run_levelwise
    np.random.seed(23)
    locs = mt.genLocations2d(Nx=n, Ny=n)
    y = np.random.normal(size=(n * n, 1))
    y_obs = np.where(np.random.uniform(size=(n * n, 1)) < 0.4, y, np.nan)
    topo = build_topology(locs, r, M, 4)
    assert topo.n_levels == M + 1 and np.all(np.asarray(topo.cw)[:M] == 64)
    spec = mt.KernelSpec(mt.KIND_MATERN32, 0.25, 1.2)
    cs = dict(topo=topo, locs=locs, y_obs=y_obs, spec=spec, c=dict(R=2e-2))
    pl, lik, mean, var = run_hip(hip, cs)
    assert any("k_predict_hi" in k["name"] and k["launches"] for k in pl.kernel_stats())
    pl.set_option(2, 0)                                    # level-by-level predictive pass
    pl.run(True, True)
    assert not any("k_predict_hi" in k["name"] for k in pl.kernel_stats())
    d, u = pl.likelihood()
    m2, v2 = pl.predict()
    assert abs(d + u - lik) <= 1e-13 * abs(lik)
    assert np.max(np.abs(m2 - mean)) < 1e-10 and K.rel(np.sqrt(v2), np.sqrt(var)) < 1e-10
    pl.set_option(2, 1); pl.set_option(12, 0)              # the parents' panel product on the direct-load GEMM instead of the LDS-tiled one
    pl.run(True, False)
    d, u = pl.likelihood()
    assert abs(d + u - lik) <= 1e-13 * abs(lik)
    # the grandparents' signed SYRK: 96 x 96 blocks through LDS (k_syrk_blk, the default; ragged last blocks of 5, 3 and 1 sub-tile
    # rows at these depths) against the 32 x 32 wave tiles of k_gemm_nt
    assert pl.get_option(14) == 1                          # stage by LDS DMA (k_syrk_dma); 2: through registers (k_syrk_blk); 0: k_gemm_nt
    pl.set_option(12, 1)
    for form in (2, 0):
        pl.set_option(14, form)
        pl.run(True, True)
        d, u = pl.likelihood()
        m3, v3 = pl.predict()
        assert abs(d + u - lik) <= 1e-13 * abs(lik), form
        assert np.max(np.abs(m3 - mean)) < 1e-11 and K.rel(np.sqrt(v3), np.sqrt(var)) < 1e-10, form
    # the prior of a level in one launch (knots' residual block -> Lp, then residual + kernel + row solve: k_leaf_gemm<COV, SOLVE>,
    # the default) against residual product, gather, factorisation and row solve as four launches
    assert pl.get_option(15) == 1
    pl.set_option(14, 1); pl.set_option(15, 0)
    pl.run(True, True)
    d, u = pl.likelihood()
    m4, v4 = pl.predict()
    assert abs(d + u - lik) <= 1e-12 * abs(lik)
    assert np.max(np.abs(m4 - mean)) < 1e-10 and K.rel(np.sqrt(v4), np.sqrt(var)) < 1e-9
    # likelihood-only passes walk the rows a likelihood needs (observed rows and knots: option 17, the default) - same arithmetic
    # per row, bit-identical to the pass over all rows
    pl.set_option(15, 1)
    pl.run(True, False)
    dl, ul = pl.likelihood()
    assert any("prior of a level" in k["name"] and k["launches"] for k in pl.kernel_stats())
    pl.set_option(17, 0)
    pl.run(True, False)
    assert pl.likelihood() == (dl, ul)
    assert abs(dl + ul - lik) <= 1e-12 * abs(lik)
    pl.set_option(17, 1)
    pl.run(True, True)                                     # and a predictive pass behind a likelihood-only one finds nothing stale
    m5, v5 = pl.predict()
    assert np.max(np.abs(m5 - mean)) < 1e-10 and K.rel(np.sqrt(v5), np.sqrt(var)) < 1e-9
    pl.close()
    if oracle:
        ref = run_levelwise(topo, locs, spec, y_obs, 2e-2)
        assert abs(lik - ref["lik"]) <= 1e-10 * abs(ref["lik"])
        assert np.max(np.abs(mean - ref["mean"])) < 1e-9
        assert K.rel(np.sqrt(var), ref["sd"]) < 1e-8
