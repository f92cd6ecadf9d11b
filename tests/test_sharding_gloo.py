"""The N>1 path on CPU: world_size-2 (and 3) torch.distributed/gloo processes, each building its
local shard topology and doing the ONE front all-reduce; compute is done by the NumPy oracle here
(tests may), so what is under test is the sharding + exchange logic that bench.py --gpus N uses."""
import os
import subprocess
import sys

import numpy as np
import pytest

import _cases as K

WORKER = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, os.environ["MRA_ROOT"]); sys.path.insert(0, os.path.join(os.environ["MRA_ROOT"], "tests"))
import torch, torch.distributed as dist
import _cases as K
from pymra_amd.sharding import shard_topology
from oracle.mra_levelwise import run_levelwise
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
cs = K.load_case(os.environ["MRA_CASE"])
local, red = shard_topology(cs["topo"], world, rank)
def allreduce(buf):
    t = torch.from_numpy(np.ascontiguousarray(buf)); dist.all_reduce(t); return t.numpy()
out = run_levelwise(local, cs["locs"], cs["spec"], cs["y_obs"], cs["c"]["R"], reduce_level=red, allreduce=allreduce)
mean = torch.from_numpy(out["mean"].copy()); var = torch.from_numpy(out["var"].copy())
dist.all_reduce(mean); dist.all_reduce(var)          # disjoint supports: sum == concatenation
if rank == 0:
    np.savez(os.environ["MRA_OUT"], lik=out["lik"], mean=mean.numpy(), var=var.numpy(), P_local=local.P, red=red)
dist.barrier(); dist.destroy_process_group()
'''


@pytest.mark.parametrize("case,world", [("g32", 2), ("g64", 2), ("c1", 3), ("g64", 8)])
def test_sharded_equals_single(tmp_path, case, world):
    from oracle.mra_levelwise import run_levelwise
    cs = K.load_case(case)
    full = run_levelwise(cs["topo"], cs["locs"], cs["spec"], cs["y_obs"], cs["c"]["R"])
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    out = tmp_path / "out.npz"
    env = dict(os.environ, MRA_ROOT=K.ROOT, MRA_CASE=case, MRA_OUT=str(out), OMP_NUM_THREADS="2")
    port = 29500 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)]
    subprocess.run(cmd, env=env, check=True, timeout=600, capture_output=True)
    r = np.load(out)
    assert int(r["red"]) >= 0 and int(r["P_local"]) < cs["topo"].P
    if world == 8:
        assert int(r["red"]) == 1          # the 8-GPU layout of BASELINE config 4: shard level 2, orphan knot rows of two upper levels
    assert abs(float(r["lik"]) - full["lik"]) <= 1e-12 * abs(full["lik"])
    assert np.max(np.abs(r["mean"] - full["mean"])) < 1e-12
    assert np.max(np.abs(r["var"] - full["var"])) < 1e-12


def test_shard_bookkeeping():
    from pymra_amd.sharding import shard_topology, choose_shard_level
    cs = K.load_case("g64")
    t = cs["topo"]
    assert choose_shard_level(t, 2) == 1 and choose_shard_level(t, 4) == 1 and choose_shard_level(t, 8) == 2
    seen = np.zeros(t.N, dtype=int)
    for r in range(8):
        lt, red = shard_topology(t, 8, r)
        assert red == 1 and lt.P % 16 == 0
        own = lt.perm[lt.in_leaf]
        seen[own] += 1
        # all upper nodes are replicated, children nest, knots are inside their node
        assert lt.level_ptr[2] == t.level_ptr[2]
        for i in range(lt.n_nodes):
            kq = lt.knot_rows[lt.knot_ptr[i]:lt.knot_ptr[i + 1]]
            assert np.all((kq >= lt.node_row0[i]) & (kq < lt.node_row1[i]))
            assert np.array_equal(np.sort(lt.perm[kq]), np.sort(t.perm[t.knot_rows[t.knot_ptr[0]:t.knot_ptr[1]]])) or i > 0
    assert np.all(seen == 1)                                # every caller row is owned by exactly one rank


@pytest.mark.parametrize("case,world,s_expected", [("g64", 4, 1), ("g64", 5, 2), ("g64", 8, 2), ("g64m", 8, 2), ("g128m", 8, 2)])
def test_shard_levels_1_and_2_through_the_oracle(case, world, s_expected):
    """Every rank of a 4/5/8-way sharded run (shard level 1 and 2: what BASELINE config 4 uses at 4 and 8 GPUs),
    emulated in-process with the level-wise oracle: identical likelihood on every rank, means and variances
    assembled from the ranks' own rows equal the single-plan result."""
    from oracle.mra_levelwise import run_levelwise
    cs = K.load_case(case)
    full = run_levelwise(cs["topo"], cs["locs"], cs["spec"], cs["y_obs"], cs["c"]["R"])
    liks, mean, var, s = K.emulate_world_oracle(cs["topo"], cs["locs"], cs["y_obs"], cs["c"]["R"], cs["spec"], world)
    assert s == s_expected
    assert max(abs(l - full["lik"]) for l in liks) <= 1e-13 * abs(full["lik"])
    assert np.max(np.abs(mean - full["mean"])) < 1e-12 and np.max(np.abs(var - full["var"])) < 1e-13


MLE_WORKER = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, os.environ["MRA_ROOT"]); sys.path.insert(0, os.path.join(os.environ["MRA_ROOT"], "tests"))
import torch, torch.distributed as dist
import _cases as K
import pymra_amd.MRATools as mt
from pymra_amd.sharding import shard_topology, sharded_minimize
from oracle.mra_levelwise import run_levelwise
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
cs = K.load_case(os.environ["MRA_CASE"])
y = np.load(os.environ["MRA_Y"])
local, red = shard_topology(cs["topo"], world, rank)
def allreduce(buf):
    t = torch.from_numpy(np.ascontiguousarray(buf)); dist.all_reduce(t); return t.numpy()
def evaluate(kappa):                       # the sharded objective: one all-reduce per call, the oracle as compute
    spec = mt.KernelSpec(mt.KIND_MATERN32, kappa, 1.0)
    return run_levelwise(local, cs["locs"], spec, y, cs["c"]["R"], reduce_level=red, allreduce=allreduce, predict=False)["lik"]
def bcast(buf):
    t = torch.from_numpy(buf); dist.broadcast(t, src=0); return buf
k_hat, fun, calls, res = sharded_minimize(evaluate, 0.3, rank, bcast, maxfev=60)
allk = [None] * world
dist.all_gather_object(allk, (k_hat, len(calls), [c[1] for c in calls]))
if rank == 0:
    json.dump({"k_hat": k_hat, "fun": fun, "n": len(calls), "all": allk, "red": red}, open(os.environ["MRA_OUT"], "w"))
dist.barrier(); dist.destroy_process_group()
'''


def test_sharded_nelder_mead_equals_single_rank(tmp_path):
    """BASELINE config 5's pattern at a size the oracle finishes in seconds: Nelder-Mead over the range parameter with the
    SHARDED objective (2 gloo ranks, one all-reduce per evaluation, kappa broadcast from rank 0) converges to the same
    kappa_hat as the single-rank run, every rank sees every objective value, and the optimum is interior."""
    import json
    from oracle.mra_levelwise import run_levelwise
    from pymra_amd.sharding import sharded_minimize
    import pymra_amd.MRATools as mt
    cs = K.load_case("g32")
    # a GP sample of the kernel family (dense Cholesky at 1024 points) + noise on the case's observation mask: interior optimum
    rng = np.random.RandomState(5)
    S = np.asarray(mt.Matern32(cs["locs"], cs["locs"], l=0.15, sig=1.0))
    f = np.linalg.cholesky(S + 1e-10 * np.eye(len(S))) @ rng.normal(size=len(S))
    y = np.where(np.isfinite(np.asarray(cs["y_obs"]).ravel()), f + np.sqrt(cs["c"]["R"]) * rng.normal(size=len(S)), np.nan).reshape(-1, 1)
    np.save(tmp_path / "y.npy", y)

    def evaluate(kappa):
        return run_levelwise(cs["topo"], cs["locs"], mt.KernelSpec(mt.KIND_MATERN32, kappa, 1.0), y, cs["c"]["R"], predict=False)["lik"]
    k1, f1, calls1, _ = sharded_minimize(evaluate, 0.3, 0, lambda b: b, maxfev=60)
    assert 0.05 < k1 < 0.29 and f1 < calls1[0][1]            # moved away from the start to an interior optimum

    script = tmp_path / "worker.py"
    script.write_text(MLE_WORKER)
    out = tmp_path / "out.json"
    env = dict(os.environ, MRA_ROOT=K.ROOT, MRA_CASE="g32", MRA_OUT=str(out), MRA_Y=str(tmp_path / "y.npy"), OMP_NUM_THREADS="2")
    port = 31500 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)]
    subprocess.run(cmd, env=env, check=True, timeout=900, capture_output=True)
    r = json.load(open(out))
    assert r["red"] == 0
    assert abs(r["k_hat"] - k1) <= 1e-6
    assert abs(r["fun"] - f1) <= 1e-10 * abs(f1)
    (ka, na, va), (kb, nb, vb) = r["all"]
    assert ka == kb and na == nb == r["n"] and va == vb     # both ranks evaluated the same points and got the same values


def test_bench_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with no launcher in the environment starts its ranks as a child torch.distributed.run
    (no exec) and hands their exit code back.  On a machine without a GPU the ranks stop at the library's "needs an AMD
    GPU" check - which is exactly how we see here that they were started."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    p = subprocess.run([sys.executable, os.path.join(K.ROOT, "bench.py"), "--gpus", "2", "--exchange", "gloo", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=600)
    from pymra_amd import plan
    if plan.device_count() >= 1:
        pytest.skip("a GPU is visible: the rehearsal on the GPU box covers this path")
    assert p.returncode != 0
    assert "launch with torch.distributed.run" not in p.stderr
    assert p.stderr.count("needs an AMD GPU") >= 2           # both ranks ran bench.py's main()
