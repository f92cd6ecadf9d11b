"""Host sanitizers: the AddressSanitizer + UBSan build of libmra_hip.so (`make asan`) runs the native
tree replay and the whole plan construction - topology copy, level / front / leaf descriptors, observation bookkeeping,
host-covariance block offsets, shard-local topologies - in host memory (MRA_HOST_DRYRUN=1: every device buffer is a malloc,
nothing is launched).  A sanitizer report aborts the child process.  CPU only; the first run builds the library (~3 min)."""
import glob
import os
import subprocess
import sys

import pytest

import _cases as K

ASAN_LIB = os.path.join(K.ROOT, "pymra_amd", "libmra_hip_asan.so")

CHILD = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.environ["MRA_ROOT"]); sys.path.insert(0, os.path.join(os.environ["MRA_ROOT"], "tests"))
import _cases as K
import pymra_amd.MRATools as mt
from pymra_amd import plan as P
from pymra_amd.sharding import shard_topology
from pymra_amd.topology import build_topology
assert "asan" in P.LIB_PATH
n_plans = 0
def build(topo, locs, y_obs, R, host_cov=None):
    global n_plans
    pl = P.HipPlan(topo, 0)
    pl.set_locs(locs); pl.set_obs(y_obs, R)
    if host_cov is None:
        pl.set_kernel(mt.KIND_MATERN32, 0.3, 1.0, 1.0)
    else:
        pl.upload_host_cov(host_cov, locs, y_obs)
        y2 = np.array(y_obs, dtype=float); y2[np.nonzero(np.isfinite(y2.ravel()))[0][:3]] = np.nan
        pl.set_obs(y2, R)                                  # invalidates the blocks (state path)
    try:
        pl.run(True, True)
        raise SystemExit("a dry-run plan must refuse to run")
    except P.MraError as e:
        assert e.code == -4
    info = pl.info(); assert info["P"] == topo.P
    pl.close(); n_plans += 1
# golden cases: 1-D (terciles, knot-boundary splits), KMeans knots/splits, regular 2-D trees
for name in ["kat1", "kat2", "kat3", "c1", "t201", "t1000", "kat4", "u3", "g32", "g64"]:
    cs = K.load_case(name)
    build(cs["topo"], cs["locs"], cs["y_obs"], cs["c"]["R"])
    if name in ("kat2", "c1", "u3", "g32"):
        build(cs["topo"], cs["locs"], cs["y_obs"], cs["c"]["R"], host_cov=cs["covfun"])
    if name in ("g32", "g64", "c1"):
        for world in (2, 3, 5, 8):
            for rank in range(world):
                try:
                    lt, red = shard_topology(cs["topo"], world, rank)
                except (ValueError, NotImplementedError):
                    continue
                build(lt, cs["locs"], cs["y_obs"], cs["c"]["R"])
# native replay (C++) at three sizes, wide blocks, empty and full observation patterns
for n, r, M in ((48, 16, 2), (128, 32, 3), (256, 64, 3), (200, 16, 4)):
    np.random.seed(3)
    locs = mt.genLocations2d(Nx=n, Ny=n)
    topo = build_topology(locs, r, M, 4)
    for frac in (0.0, 0.4, 1.0):
        y = np.where(np.random.uniform(size=(n * n, 1)) < frac, np.random.normal(size=(n * n, 1)), np.nan)
        build(topo, locs, y, 1e-2)
    # the one-call constructor path (replay + plan beside the knot draws): same tree, same RNG state, a plan that is complete
    np.random.seed(3)
    st0 = np.random.get_state()
    both = P.create_with_replay(locs, r, M, 4, y, 1e-2)
    assert both is not None
    pl2, topo2 = both
    st1 = np.random.get_state()
    np.random.set_state(st0)
    topo3 = build_topology(locs, r, M, 4)
    st2 = np.random.get_state()
    assert np.array_equal(st1[1], st2[1]) and st1[2] == st2[2]
    for f in ("perm", "src", "in_leaf", "level_ptr", "node_row0", "node_row1", "node_leaf", "node_parent", "child_ptr", "child_list",
              "knot_ptr", "knot_rows", "cw", "order_preorder"):
        assert np.array_equal(getattr(topo2, f), getattr(topo3, f)), f
    pl2.set_kernel(mt.KIND_MATERN32, 0.3, 1.0, 1.0)
    assert pl2.info()["P"] == topo3.P
    pl2.close(); n_plans += 1
# a tree the one-call path must decline (nodes with <= 100 rows), leaving the RNG state alone
np.random.seed(4)
locs = mt.genLocations2d(Nx=40, Ny=40)
st0 = np.random.get_state()
assert P.create_with_replay(locs, 16, 3, 4, np.zeros(1600), 1e-2) is None
st1 = np.random.get_state()
assert np.array_equal(st0[1], st1[1]) and st0[2] == st1[2]
print("ASAN_HOST_OK", n_plans)
'''


def _runtime():
    g = glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so")
    return g[0] if g else None


def test_plan_construction_and_tree_replay_under_asan_ubsan(tmp_path):
    rt = _runtime()
    if rt is None or not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc / sanitizer runtime in this image")
    subprocess.run(["make", "-s", "asan"], cwd=K.ROOT, check=True, timeout=1500)
    script = tmp_path / "child.py"
    script.write_text(CHILD)
    env = dict(os.environ, MRA_ROOT=K.ROOT, MRA_HOST_DRYRUN="1", PYMRA_AMD_LIB=ASAN_LIB, LD_PRELOAD=rt,
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1")
    res = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=1500)
    tail = (res.stdout + res.stderr)[-3000:]
    assert res.returncode == 0 and "ASAN_HOST_OK" in res.stdout, tail
    assert "AddressSanitizer" not in res.stderr and "runtime error" not in res.stderr, tail
