"""The C-ABI library on a machine WITHOUT a GPU: it builds, loads, exports every symbol the header
declares, and refuses to create a plan (no CPU fallback)."""
import os
import re

import numpy as np
import pytest

import _cases as K


def _header_functions():
    txt = open(os.path.join(K.ROOT, "include", "mra_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mra_[a-z_0-9]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(built_library):
    from pymra_amd import plan
    declared = _header_functions()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(built_library, name), "libmra_hip.so lacks %s" % name
    assert sorted(plan.EXPORTS) == declared, "pymra_amd.plan.EXPORTS and include/mra_hip.h disagree"
    assert b"gfx950" in built_library.mra_version()


def test_product_path_never_imports_the_oracle():
    """pymra_amd must not reach into oracle/ (only tests, smoke() and bench's cpu_baseline may)."""
    pkg = os.path.join(K.ROOT, "pymra_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "oracle" not in re.sub(r'""".*?"""', "", src, flags=re.S), fn


def test_no_gpu_means_loud_failure(built_library):
    from pymra_amd import plan
    if plan.device_count() > 0:
        pytest.skip("a GPU is present")
    cs = K.load_case("kat2")
    with pytest.raises(plan.MraError) as ei:
        plan.HipPlan(cs["topo"], 0)
    assert ei.value.code == -2 and "no CPU fallback" in str(ei.value)
    import pymra_amd
    with pytest.raises(plan.MraError):
        pymra_amd.MRATree(cs["locs"], cs["c"]["r"], lambda a, b: pymra_amd.MRATools.ExpCovFun(a, b, l=1.0),
                          cs["y_obs"], cs["c"]["R"], M=cs["c"]["M"], J=cs["c"]["J"])


def test_cov_probe_recognises_device_kernels():
    import pymra_amd.MRATools as mt
    from pymra_amd.MRATree import probe_cov
    s = probe_cov(lambda a, b: 2.5 * mt.Matern32(a, b, l=0.3, sig=1.5), 2)
    assert (s.kind, s.l, s.sig, s.scale) == (mt.KIND_MATERN32, 0.3, 1.5, 2.5)
    assert probe_cov(lambda a, b: mt.ExpCovFun(a, b, l=0.7), 1).kind == mt.KIND_EXP
    assert probe_cov(lambda a, b: np.exp(-np.abs(a - b.T)), 1) is None          # opaque callable
    assert probe_cov(np.eye(3), 1) is None                                       # dense matrix
    # array inputs still give the reference's np.matrix values
    x = mt.genLocations(5)
    v = mt.Matern32(x, x, l=0.3, sig=2.0)
    assert isinstance(v, np.matrix) and v.shape == (5, 5) and abs(v[0, 0] - 2.0) < 1e-15
    assert np.allclose(np.asarray(s.evaluate(x, x)), 2.5 * np.asarray(mt.Matern32(x, x, l=0.3, sig=1.5)))


def test_option_numbers_agree_with_the_header():
    """pymra_amd/plan.py repeats the MRA_OPT_* numbers of include/mra_hip.h (ctypes has no header): they must not drift."""
    import re
    from pymra_amd import plan as P
    hdr = open(os.path.join(K.ROOT, "include", "mra_hip.h")).read()
    defs = dict((m.group(1), int(m.group(2))) for m in re.finditer(r"#define\s+(MRA_OPT_\w+)\s+(\d+)", hdr))
    assert len(defs) >= 12
    for name, val in defs.items():
        assert getattr(P, name) == val, name


def test_every_plan_raises_its_own_lds_limits(built_library, tmp_path):
    """The 160 KB dynamic-LDS attribute is per kernel AND per device: it is recorded per plan (a plan lives on one device),
    not behind a process-wide flag - two plans on different device ordinals in one process both take the path.  Host dry
    run (MRA_HOST_DRYRUN=1: plans are built in host memory, nothing is launched), so this runs without a GPU."""
    import subprocess
    import sys
    child = tmp_path / "child.py"
    child.write_text(r'''
import os, sys
sys.path.insert(0, os.environ["MRA_ROOT"]); sys.path.insert(0, os.path.join(os.environ["MRA_ROOT"], "tests"))
import _cases as K
import pymra_amd.MRATools as mt
from pymra_amd import plan as P
cs = K.load_case("g64")
counts = []
for dev in (0, 1, 0):
    pl = P.HipPlan(cs["topo"], dev)
    pl.set_locs(cs["locs"]); pl.set_obs(cs["y_obs"], cs["c"]["R"]); pl.set_kernel(mt.KIND_EXP, 0.3, 1.0, 1.0)
    counts.append(pl.prepare())
    assert pl.prepare() == counts[-1]            # idempotent
    try:
        pl.set_option(99, 2)                     # what-if bits that change results are not in the product library
        raise SystemExit("option 99 bit 2 must be refused")
    except P.MraError as e:
        assert e.code == -1
    pl.set_option(99, 8); assert pl.get_option(99) == 8
    pl.set_option(P.MRA_OPT_FUSED, 0); assert pl.get_option(P.MRA_OPT_FUSED) == 0
    for opt, val in ((P.MRA_OPT_LEAF_SOLVE_SPLIT, 1), (P.MRA_OPT_CHOL_TILES, 2), (P.MRA_OPT_SEG_GEMM_LDS, 0), (P.MRA_OPT_UT_GATHER, 0)):
        pl.set_option(opt, val); assert pl.get_option(opt) == val
    try:
        pl.set_option(57, 1)
        raise SystemExit("an unknown option must be refused")
    except P.MraError as e:
        assert e.code == -1
    pl.close()
print("LDS_ATTR_OK", counts)
assert counts[0] >= 10 and counts[0] == counts[1] == counts[2]
''')
    env = dict(os.environ, MRA_ROOT=K.ROOT, MRA_HOST_DRYRUN="1")
    res = subprocess.run([sys.executable, str(child)], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "LDS_ATTR_OK" in res.stdout, (res.stdout + res.stderr)[-2000:]
