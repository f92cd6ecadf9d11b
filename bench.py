#!/usr/bin/env python3
"""bench.py - BASELINE.json's metric on BASELINE.json's configuration.

Workload (config.workload = "c3"): 2-D 1024x1024 grid, M=6, J=4, r0=32, Matern32(l=0.3, sig=1),
40 % observed, R=1e-2, synthetic data by the recipe of SURVEY.md section 8(d) (seed 11).
A "step" is one full pass of the hot path - prior for every node, posterior for every node,
likelihood and predictive mean/variance at all 2^20 locations (everything pyMRA's MRATree
constructor computes) - with the tree, locations and observations already resident in HBM.
value = MRA nodes processed per second by the whole job (5461 nodes per step).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

N > 1: strong scaling - the same tree, level-s subtrees sharded over the ranks, ONE RCCL all-reduce
of the level s-1 fronts per step (pymra_amd/sharding.py); torch.distributed (gloo) is used only for
the rendezvous, the barriers and the max-over-ranks of the timings.

"value" (= "value_resident") is the device pass with tree, locations and observations resident in HBM - what an MLE
loop pays per evaluation; "value_end_to_end" is the same node count over the wall-clock of the drop-in constructor
MRATree(...) + getLikelihood() + predict() (host tree replay + uploads + device pass + download), the quantity
SURVEY.md section 8(d) defines for the reference.  Extra objects in the JSON line: "roofline" (dominant kernel,
hipEvent-timed inside the timed steps; "peak" is the datasheet FP64 matrix rate, "peak_measured" the back-to-back
v_mfma_f64_16x16x4 rate measured on this GPU, profiles/r02_mfma_f64_peak_microbench.txt), "cpu_baseline"
(N=1, rank 0: the faithful NumPy/SciPy restatement of the reference timed on a bounded sample of the same tree, with
and without the reference's per-node gc.collect() and with one BLAS thread), "host" (set-up timings).

--config c5 (BASELINE config 5 geometry, 2048^2 grid, M=8, r0=64) adds "mle": Nelder-Mead objective evaluations
per second on the resident plan (likelihood-only passes with a new kernel parameter each).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6     # MI355X datasheet FP64 matrix (= vector) peak; DESIGN.md section 5
FP64_MFMA_MEASURED_TFLOPS = 77.3  # back-to-back v_mfma_f64_16x16x4 on this GPU: 64.0 cycles per instruction per SIMD at 2.39 GHz
                                  # (tools/mfma_f64_peak.hip, profiles/r02_mfma_f64_peak_microbench.txt)
HBM_PEAK_GBS = 8000.0

CONFIGS = {
    "c3": dict(n=1024, M=6, J=4, r=32, kern="m32", l=0.3, sig=1.0, R=1e-2, seed=11, frac=0.4),
    "c2": dict(n=256, M=4, J=4, r=16, kern="exp", l=0.3, sig=1.0, R=1e-2, seed=11, frac=0.4),
    # BASELINE.json config 5 geometry (general level-by-level kernels; 29 GB on one GPU)
    "c5": dict(n=2048, M=8, J=4, r=64, kern="m32", l=0.3, sig=1.0, R=1e-2, seed=11, frac=0.4),
}


def make_inputs(c):
    import pymra_amd.MRATools as mt
    np.random.seed(c["seed"])
    n = c["n"]
    locs = mt.genLocations2d(Nx=n, Ny=n)
    y = np.random.normal(size=(n * n, 1))
    oi = np.sort(np.random.choice(n * n, int(c["frac"] * n * n), replace=False))
    y_obs = np.full((n * n, 1), np.nan)
    y_obs[oi] = y[oi]
    return locs, y_obs


def cpu_baseline(topo, locs, y_obs, c, budget_nodes=85):
    """Faithful restatement of the reference's per-node work (oracle/mra_faithful.py) on a bounded
    sample: one level-3 subtree of the same tree (85 nodes incl. 64 leaves at c3).  Four variants, as BASELINE.md
    section 3 asks - with / without the reference's per-node gc.collect() (MRANode.py:111), on the default BLAS thread
    pool / on one BLAS thread (the small LAPACK calls oversubscribe a large pool) - and "value" is the FASTEST of them."""
    import pymra_amd.MRATools as mt
    from oracle.mra_faithful import run_subtree_sample
    cov = (lambda a, b: mt.Matern32(a, b, l=c["l"], sig=c["sig"])) if c["kern"] == "m32" else \
          (lambda a, b: mt.ExpCovFun(a, b, l=c["l"]))
    level = 0
    for m in range(topo.n_levels):
        # deepest level whose subtrees still have at least budget_nodes nodes
        n_sub = sum(int(topo.level_ptr[k + 1] - topo.level_ptr[k]) for k in range(m, topo.n_levels)) / \
                max(1, int(topo.level_ptr[m + 1] - topo.level_ptr[m]))
        if n_sub >= budget_nodes:
            level = m
    top = int(topo.level_ptr[level])
    # threads actually used: the BLAS/LAPACK pool NumPy and SciPy run on (the Python loop itself is serial)
    cores = None
    runs = {}
    n, secs, tm = run_subtree_sample(topo, locs, cov, y_obs, c["R"], top, do_gc=True)
    runs["default_blas_threads_with_gc_collect"] = (n, secs, tm)
    n2, secs_nogc, tm2 = run_subtree_sample(topo, locs, cov, y_obs, c["R"], top, do_gc=False)
    runs["default_blas_threads_without_gc_collect"] = (n2, secs_nogc, tm2)
    try:
        from threadpoolctl import threadpool_info, threadpool_limits
        pools = [p.get("num_threads", 0) for p in threadpool_info() if p.get("user_api") == "blas"]
        cores = max(pools) if pools else None
        with threadpool_limits(limits=1, user_api="blas"):
            runs["one_blas_thread_with_gc_collect"] = run_subtree_sample(topo, locs, cov, y_obs, c["R"], top, do_gc=True)
            runs["one_blas_thread_without_gc_collect"] = run_subtree_sample(topo, locs, cov, y_obs, c["R"], top, do_gc=False)
    except Exception:
        pass
    if not cores:
        try:
            cores = len(os.sched_getaffinity(0))
        except Exception:
            cores = os.cpu_count()
    # the reported baseline is the FASTEST of the variants (the reference as written is the first one: default threads and a
    # gc.collect() per node, MRANode.py:111; its many small LAPACK calls oversubscribe a large BLAS pool, so one thread wins)
    best = max(runs, key=lambda k: runs[k][0] / runs[k][1])
    bn, bsecs, btm = runs[best]
    bcores = 1 if best.startswith("one_blas_thread") else cores
    return {"value": bn / bsecs, "unit": "nodes/s", "cores": bcores, "kind": "port", "variant": best,
            "sample": "subtree of level-%d node %d of the same tree: %d nodes (%.1f s; prior %.1f s, posterior %.1f s, "
                      "per-node gc.collect %.1f s), fastest of four variants: %s"
                      % (level, top, bn, bsecs, btm["prior"], btm["posterior"], btm["gc"], best.replace("_", " ")),
            "seconds": bsecs, "nodes": bn,
            "variants": {k: {"value": v[0] / v[1], "seconds": v[1], "cores": 1 if k.startswith("one_blas_thread") else cores}
                         for k, v in runs.items()}}


def cpu_baseline_whole_tree(topo, locs, y_obs, c):
    """The WHOLE tree through the faithful restatement (oracle/mra_faithful.run_faithful, the reference's per-node recursion with
    its own LAPACK calls and its per-node gc.collect(), MRANode.py:111): what BASELINE.md section 2 measured for the reference
    itself (config 2: 89.6 s on the 8-core survey box).  Not an extrapolation from a subtree - minutes at c2, so on request only
    (--whole-tree-cpu-baseline)."""
    import pymra_amd.MRATools as mt
    from oracle.mra_faithful import run_faithful
    cov = (lambda a, b: mt.Matern32(a, b, l=c["l"], sig=c["sig"])) if c["kern"] == "m32" else \
          (lambda a, b: mt.ExpCovFun(a, b, l=c["l"]))
    tm = {}
    t0 = time.perf_counter()
    out = run_faithful(topo, locs, cov, y_obs, c["R"], do_gc=True, timers=tm)
    secs = time.perf_counter() - t0
    try:
        from threadpoolctl import threadpool_info
        pools = [p.get("num_threads", 0) for p in threadpool_info() if p.get("user_api") == "blas"]
        cores = max(pools) if pools else os.cpu_count()
    except Exception:
        cores = os.cpu_count()
    return {"value": topo.n_nodes / secs, "unit": "nodes/s", "cores": cores, "kind": "port",
            "sample": "the whole tree (%d nodes), constructor-equivalent work incl. the per-node gc.collect(), NumPy/SciPy default BLAS "
                      "threads" % topo.n_nodes, "seconds": secs, "nodes": int(topo.n_nodes), "likelihood": float(out["lik"]),
            "timers": {k: float(v) for k, v in tm.items()}}


def gp_sample_rff(c, l_true, seed=12345, n_feat=1024):
    """A Matern-3/2 Gaussian-process sample on the n x n grid of `c` by random Fourier features ("a smoother
    variant (random-Fourier-feature GP sample, same seed) may be added"): f(x) = sqrt(2 sig / D) sum_k cos(w_k . x + b_k) with w_k
    from the kernel's spectral measure (a bivariate Student-t with 2 nu = 3 degrees of freedom, scale 1 / l) and b_k ~ U(0, 2 pi).
    On a grid cos(wx x_i + wy y_j + b) separates into outer products, so the whole field is two D-column GEMMs."""
    rng = np.random.RandomState(seed)
    n = c["n"]
    g = np.linspace(0.0, 1.0, n)
    z = rng.normal(size=(n_feat, 2))
    u = rng.chisquare(3.0, size=n_feat)
    w = z * np.sqrt(3.0 / u)[:, None] / l_true             # Matern nu = 3/2, range l_true (MRATools.Matern32: exp(-sqrt(3) D / l) (1 + sqrt(3) D / l))
    b = rng.uniform(0.0, 2.0 * np.pi, size=n_feat)
    ax = np.outer(g, w[:, 0]) + b[None, :]                  # x index i
    ay = np.outer(g, w[:, 1])                               # y index j
    f = np.cos(ay) @ np.cos(ax).T - np.sin(ay) @ np.sin(ax).T        # F[j][i]: x fastest, as genLocations2d orders the rows
    return (np.sqrt(2.0 * c["sig"] / n_feat) * f).reshape(-1, 1), rng


def mle_throughput(pl, c, kind, y_obs, red=-1, host_allreduce=None, rank=0, dist=None, l_true=0.1, max_evals=80):
    """End-to-end MLE of the range parameter on the resident plan(s), as README.md:96-104 / tests/test-param-est.py:81-123 do it with
    a new MRATree per call: Nelder-Mead from kappa_0 = 0.3 with xatol 1e-3 on -2 loglik = d + u (MRATree.getLikelihood), every
    objective call one likelihood-only device pass with new kernel parameters.  With more than one rank (BASELINE config 5:
    8 GPUs) the pass is the SHARDED one - every rank runs its subtrees, ONE all-reduce of the reduce level's fronts per
    objective call - and rank 0 drives the optimiser, broadcasting every kappa so that all ranks evaluate the same point
    (pymra_amd.sharding.sharded_minimize).  The data are a GP sample of the same kernel family with range l_true (random
    Fourier features) plus N(0, R) noise on the benchmark's observation mask, so the optimiser has an interior optimum to
    converge to (on the iid-noise data of the throughput recipe it only walks to the lower bound)."""
    from pymra_amd.sharding import sharded_minimize, sharded_run
    f, rng = gp_sample_rff(c, l_true)
    mask = np.isfinite(np.asarray(y_obs).reshape(-1, 1))
    y_gp = np.where(mask, f + np.sqrt(c["R"]) * rng.normal(size=f.shape), np.nan)
    pl.set_obs(y_gp, c["R"])                               # (a local plan picks its own rows through its row maps)

    def evaluate(kappa):
        pl.set_kernel(kind, kappa, c["sig"], 1.0)
        sharded_run(pl, red, host_allreduce, True, False)
        d, u = pl.likelihood()
        return d + u

    def bcast(buf):
        if dist is not None:
            import torch
            t = torch.from_numpy(buf)
            dist.broadcast(t, src=0)
        return buf
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    k_hat, fun, calls, res = sharded_minimize(evaluate, c["l"], rank, bcast, maxfev=max_evals)
    dt = time.perf_counter() - t0
    out = None
    if rank == 0:
        out = {"data": "Matern32 GP sample by %d random Fourier features, range %.3g, + N(0, R) noise, benchmark mask" % (1024, l_true),
               "kappa_true": l_true, "kappa_start": c["l"], "kappa_hat": k_hat, "converged": bool(res.success),
               "objective_evaluations": len(calls), "seconds_to_convergence": dt, "evaluations_per_s": len(calls) / dt,
               "ms_per_evaluation": 1e3 * dt / len(calls), "objective_at_start": calls[0][1], "objective_at_kappa_hat": fun,
               "interior_optimum": bool(0.02 < k_hat < 5.0 and fun < calls[0][1]),
               "sharded": bool(red >= 0), "all_reduces_per_evaluation": 1 if red >= 0 else 0}
    pl.set_obs(y_obs, c["R"])                              # back to the throughput recipe's data
    pl.set_kernel(kind, c["l"], c["sig"], 1.0)
    return out


def traffic_from_profiles(kernel_family, profiles_dir=None):
    """(hbm bytes per launch, raw counters, source file, note) of a kernel family from the newest PMC summary under profiles/.
    The counters cannot be collected inside this process (they need rocprofv3 around it), so the summary of the round's
    collection is quoted - but only if it was collected from the kernel sources that are in the tree now (tools/csrc_hash.py):
    after any change under pymra_amd/csrc the bytes are unknown until tools/collect_profiles.sh has run again."""
    import glob
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from csrc_hash import csrc_sha256
    cands = sorted(glob.glob(os.path.join(profiles_dir or os.path.join(ROOT, "profiles"), "*_pmc_traffic_by_kernel_family.json")))
    if not cands:
        return None, None, None, "no PMC summary under profiles/"
    src = os.path.basename(cands[-1])
    try:
        summ = json.load(open(cands[-1]))
    except Exception as e:                                    # pragma: no cover
        return None, None, src, "unreadable summary: %r" % (e,)
    stamp = (summ.get("_collected_from") or {}).get("csrc_sha256")
    if stamp is None:
        return None, None, src, "%s carries no source stamp (collected before round 4)" % src
    if stamp != csrc_sha256():
        return None, None, src, "%s was collected from other kernel sources (stamp %s...): run tools/collect_profiles.sh again" % (src, stamp[:12])
    fam = summ.get(kernel_family)
    if not fam or "hbm_bytes_per_launch" not in fam:
        return None, None, src, "%s has no counters for this kernel family" % src
    return (fam["hbm_bytes_per_launch"], {"FETCH_SIZE_KB": fam.get("FETCH_SIZE_KB_per_launch_mean"), "WRITE_SIZE_KB": fam.get("WRITE_SIZE_KB_per_launch_mean")},
            src, None)


def relaunch_with_ranks(n):
    """`python bench.py --gpus N` without a launcher (WORLD_SIZE unset): start the N ranks ourselves, as a CHILD process -
    `python -m torch.distributed.run --nproc-per-node N ... bench.py <same arguments>` on a free local port - before anything
    in this process has touched the GPU or loaded libmra_hip.so; rank 0's JSON line arrives on our stdout through the
    inherited descriptor and the child's exit code is ours.  (The reference's parallel mode needs no launcher either: it forks
    its workers itself, pyMRA/MRATree.py:53-59, MRANode.py:90-104.)"""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    return subprocess.call(cmd, env=env)


def main():
    # multi-process GPU work on this pool needs dmabuf IPC (RCCL fails with hipIpcGetMemHandle: invalid argument otherwise); the
    # variable is read when the HSA runtime starts, i.e. before the first HIP call of this process - normally the launcher exports it
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the end-to-end MRATree(...) constructor timing")
    ap.add_argument("--likelihood-only", action="store_true")
    ap.add_argument("--whole-tree-cpu-baseline", action="store_true",
                    help="time the faithful CPU restatement on the WHOLE tree of the configuration (about 90 s at c2, 30 min at c3) "
                         "instead of the bounded subtree sample")
    ap.add_argument("--mle", action="store_true", help="add the end-to-end Nelder-Mead MLE block (always on for --config c5)")
    ap.add_argument("--allow-gloo-fallback", action="store_true",
                    help="with --exchange rccl: if RCCL cannot initialise, re-plan onto the host/gloo exchange instead of failing "
                         "(the line then carries \"exchange_fallback\": true)")
    ap.add_argument("--exchange", default="rccl", choices=["rccl", "gloo"],
                    help="transport of the one front all-reduce for --gpus > 1: RCCL on the device (default) or "
                         "host export/import + torch.distributed gloo (rehearsal on a box with fewer GPUs than ranks)")
    args = ap.parse_args()
    c = CONFIGS[args.config]

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(relaunch_with_ranks(args.gpus))     # self-contained: the ranks are our child processes
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d was started by a launcher with WORLD_SIZE=%d" % (args.gpus, world))
    # PyTorch is rendezvous plumbing for N > 1 only (gloo barrier, broadcast of the RCCL id, max over ranks); a single-GPU run
    # never imports it: the data path is ctypes -> libmra_hip.so, and the device barrier is the library's hipDeviceSynchronize
    dist = torch = None
    json_fd = None
    if world > 1:
        # rank 0 prints ONE JSON line: gloo announces its connections on stdout (C++ side) when the first collective runs, so
        # file descriptor 1 of every rank is pointed at stderr for the lifetime of the process and the line goes out through a
        # copy of the original descriptor
        sys.stdout.flush()
        json_fd = os.dup(1)
        os.dup2(2, 1)
        import torch
        import torch.distributed as dist
        dist.init_process_group("gloo")

    import pymra_amd.MRATools as mt
    from pymra_amd import plan as P
    from pymra_amd.sharding import shard_topology, sharded_run
    from pymra_amd.topology import build_topology

    if P.device_count() < 1:
        raise SystemExit("bench.py needs an AMD GPU (libmra_hip has no CPU fallback)")
    if os.environ.get("MRA_BENCH_SINGLE_DEVICE") == "1":      # rehearsal: every rank on GPU 0 (only with --exchange gloo)
        local_rank = 0

    # ---- set-up (untimed): inputs, tree, plan, upload ----------------------------------------------
    t0 = time.perf_counter()
    locs, y_obs = make_inputs(c)
    t1 = time.perf_counter()
    topo = build_topology(locs, c["r"], c["M"], c["J"])
    t2 = time.perf_counter()
    local, red = shard_topology(topo, world, rank)
    pl = P.HipPlan(local, local_rank)
    pl.set_locs(locs)
    pl.set_obs(y_obs, c["R"])
    kind = mt.KIND_MATERN32 if c["kern"] == "m32" else mt.KIND_EXP
    pl.set_kernel(kind, c["l"], c["sig"], 1.0)
    host_allreduce = None
    exchange = args.exchange
    fallback = False
    if world > 1 and exchange == "rccl":
        ok = 1
        try:
            uid = [P.comm_unique_id() if rank == 0 else None]
        except Exception as e:                                # pragma: no cover - depends on the box
            uid, ok = [None], 0
            print("rank %d: RCCL unique id failed: %s" % (rank, e), file=sys.stderr)
        dist.broadcast_object_list(uid, src=0)
        if uid[0] is None:
            ok = 0
        if ok:
            try:
                pl.comm_init(uid[0], world, rank)
            except Exception as e:                            # pragma: no cover
                ok = 0
                print("rank %d: RCCL init failed: %s" % (rank, e), file=sys.stderr)
        flag = torch.tensor([ok], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag[0]) == 0:
            if not args.allow_gloo_fallback:
                raise SystemExit("RCCL could not initialise on every rank; refusing to measure the host exchange instead "
                                 "(pass --allow-gloo-fallback or --exchange gloo to do that on purpose)")
            # every rank falls back together: host export/import + gloo all-reduce of the same buffer
            exchange = "gloo (RCCL unavailable)"
            fallback = True
            pl.close()
            pl = P.HipPlan(local, local_rank)
            pl.set_locs(locs); pl.set_obs(y_obs, c["R"]); pl.set_kernel(kind, c["l"], c["sig"], 1.0)
    if world > 1 and exchange != "rccl":
        def host_allreduce(buf):
            t = torch.from_numpy(np.ascontiguousarray(buf))
            dist.all_reduce(t)
            return t.numpy()
    t3 = time.perf_counter()
    predict = not args.likelihood_only

    def step():
        sharded_run(pl, red, host_allreduce, True, predict)

    def barrier():
        if dist is not None:
            dist.barrier()
        P.device_synchronize(local_rank)         # = torch.cuda.synchronize() of the contract, without a second runtime on the GPU

    for _ in range(args.warmup):
        step()
    pl.set_option(P.MRA_OPT_KERNEL_TIMING, 1)
    kacc = None
    barrier()
    ts = time.perf_counter()
    for _ in range(args.steps):
        step()
        ks = pl.kernel_stats()
        if kacc is None:
            kacc = ks
        else:
            for a, b in zip(kacc, ks):
                for f in ("launches", "ms", "flops", "flops_exec", "bytes"):
                    a[f] += b[f]
    barrier()
    elapsed = time.perf_counter() - ts
    pl.set_option(P.MRA_OPT_KERNEL_TIMING, 0)
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt[0])
    d, u = pl.likelihood()
    timers = pl.timers()

    # un-instrumented repeat of the same steps (event bracketing costs a little host time)
    barrier()
    per_step = []
    ts = time.perf_counter()
    for _ in range(args.steps):
        t_s = time.perf_counter()
        step()                                   # blocking: mra_run returns after the pass has completed on the device
        per_step.append(time.perf_counter() - t_s)
    barrier()
    elapsed_plain = time.perf_counter() - ts
    if dist is not None:
        tt = torch.tensor([elapsed_plain], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed_plain = float(tt[0])

    n_nodes = topo.n_nodes
    out = None
    if rank == 0:
        dom = max(kacc, key=lambda k: k["ms"])
        dom_ms = dom["ms"] / max(dom["launches"], 1)
        dom_fl = dom["flops"] / max(dom["launches"], 1)
        ach = dom_fl / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0
        # HBM traffic of the dominant kernel from the committed PMC summary of this same command
        # (tools/collect_profiles.sh: separate --pmc FETCH_SIZE / WRITE_SIZE passes, gfx950 correction) - quoted only while the
        # summary's stamp matches the kernel sources this library was built from, null (with a note) otherwise
        traffic, traffic_raw, traffic_src, traffic_note = traffic_from_profiles(dom["name"])
        if traffic_note:
            print("bench.py: roofline.traffic is null: %s" % traffic_note, file=sys.stderr)
        # "achieved"/"frac" are strictly algorithmic: flops counted with the true ranks, the true observation counts and a
        # one-column y (mra_plan.hip: struct Work); the rate of the MFMAs the 16-padded tiles actually execute is reported
        # separately as mfma_executed_tflops.  Every family also shows its algorithmic HBM bytes against the 8 TB/s roof,
        # so that each kernel can be read against both roofs.
        def fam_entry(k):
            sec = k["ms"] * 1e-3
            tf = k["flops"] / sec / 1e12 if sec > 0 else 0.0
            gbs = k["bytes"] / sec / 1e9 if sec > 0 else 0.0
            return {"name": k["name"], "launches_per_step": k["launches"] / args.steps, "ms_per_step": k["ms"] / args.steps,
                    "tflops": tf, "mfma_frac": tf / FP64_MFMA_PEAK_TFLOPS,
                    "mfma_executed_tflops": k["flops_exec"] / sec / 1e12 if sec > 0 else 0.0,
                    "hbm_bytes": k["bytes"] / max(k["launches"], 1), "hbm_gbs": gbs, "hbm_frac": gbs / HBM_PEAK_GBS}
        dom_exec = dom["flops_exec"] / max(dom["launches"], 1)
        roof = {"bound": "mfma", "kernel": dom["name"], "achieved": ach, "peak": FP64_MFMA_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": ach / FP64_MFMA_PEAK_TFLOPS, "peak_measured": FP64_MFMA_MEASURED_TFLOPS,
                "frac_of_measured": ach / FP64_MFMA_MEASURED_TFLOPS,
                "mfma_executed_tflops": dom_exec / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0,
                "traffic": traffic, "traffic_raw_counters": traffic_raw, "traffic_source": traffic_src, "traffic_note": traffic_note,
                "hbm_bytes_algorithmic": dom["bytes"] / max(dom["launches"], 1),
                "hbm_frac": (dom["bytes"] / max(dom["launches"], 1)) / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if dom_ms > 0 else 0.0,
                "avg_launch_ms": dom_ms, "flop_per_launch": dom_fl,
                "kernels": [fam_entry(k) for k in kacc if k["launches"]]}
        whole_flop = sum(k["flops"] for k in kacc) / args.steps
        whole_bytes = sum(k["bytes"] for k in kacc) / args.steps
        out = {
            "metric": "MRA nodes/sec, resident device pass (prior+posterior+likelihood+predict with tree and data in HBM; value_end_to_end = "
                      "the same nodes over the wall-clock of MRATree(...)+getLikelihood()+predict()), 1024^2 grid M=6 J=4 r0=32" if args.config == "c3" else
                      "MRA nodes/sec, resident device pass, %s" % args.config,
            "value": n_nodes * args.steps / elapsed_plain, "unit": "nodes/s", "n_gpus": world,
            "value_resident": n_nodes * args.steps / elapsed_plain, "value_end_to_end": None,
            "whole_pass": {"flop_algorithmic": whole_flop, "tflops": whole_flop / (elapsed_plain / args.steps) / 1e12,
                           "hbm_bytes_algorithmic": whole_bytes, "hbm_gbs": whole_bytes / (elapsed_plain / args.steps) / 1e9},
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed_plain / args.steps,
            "ms_per_step_median": 1e3 * float(np.median(per_step)), "ms_per_step_min": 1e3 * float(np.min(per_step)),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": args.config, "grid": "%dx%d" % (c["n"], c["n"]), "M": c["M"], "J": c["J"],
                       "r0": c["r"], "kernel": "Matern32" if c["kern"] == "m32" else "ExpCovFun", "l": c["l"],
                       "frac_obs": c["frac"], "R": c["R"], "nodes": n_nodes,
                       "mode": "likelihood" if args.likelihood_only else "likelihood+predict",
                       "parallelism": ("subtree-shard x%d, 1 all-reduce (%s)" % (world, exchange)) if world > 1 else "single GPU"},
            "exchange_fallback": fallback,
            "likelihood": d + u,
            "ms_per_step_with_kernel_events": 1e3 * elapsed / args.steps,
            "device_phase_ms": timers,
            "roofline": roof,
            "host": {"input_synthesis_s": t1 - t0, "tree_build_s": t2 - t1, "plan_and_upload_s": t3 - t2},
        }
    if args.config == "c5" or args.mle:
        # every rank takes part: the objective is the sharded pass (one all-reduce per evaluation), rank 0 drives Nelder-Mead
        m = mle_throughput(pl, c, kind, y_obs, red, host_allreduce, rank, dist)
        if rank == 0:
            out["mle"] = m
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(topo, locs, y_obs, c)
        if args.whole_tree_cpu_baseline:
            wt = cpu_baseline_whole_tree(topo, locs, y_obs, c)
            wt["likelihood_rel_diff_vs_gpu"] = abs(wt["likelihood"] - (d + u)) / abs(d + u)
            out["cpu_baseline_whole_tree"] = wt
    if rank == 0 and world == 1 and not args.no_end_to_end:
        # end-to-end constructor wall-clock of the drop-in API (host tree build + H2D + device pass + D2H): BASELINE.json's
        # "nodes/sec + getLikelihood() wall-clock" as SURVEY.md section 8(d) defines it for the reference
        from pymra_amd import MRATree
        np.random.seed(c["seed"]); make_inputs(c)            # RNG where the recipe leaves it
        cov = (lambda a, b: mt.Matern32(a, b, l=c["l"], sig=c["sig"])) if c["kern"] == "m32" else \
              (lambda a, b: mt.ExpCovFun(a, b, l=c["l"]))
        pl.close()
        walls = []
        for _ in range(5):                                   # fresh constructions (same seed => same tree); the median is reported
            np.random.seed(c["seed"]); make_inputs(c)
            tA = time.perf_counter()
            tree = MRATree(locs, c["r"], cov, y_obs, c["R"], M=c["M"], J=c["J"])
            lik = tree.getLikelihood(); xP, sdP = tree.predict()
            walls.append(time.perf_counter() - tA)
            tree.plan.close()
        if os.environ.get("MRA_BENCH_PROFILE_E2E") == "1":     # where the constructor's wall-clock goes, Python side included (stderr)
            import cProfile
            import pstats
            np.random.seed(c["seed"]); make_inputs(c)
            pr = cProfile.Profile()
            pr.enable()
            tree = MRATree(locs, c["r"], cov, y_obs, c["R"], M=c["M"], J=c["J"])
            lik = tree.getLikelihood(); xP, sdP = tree.predict()
            pr.disable()
            tree.plan.close()
            pstats.Stats(pr, stream=sys.stderr).sort_stats("cumulative").print_stats(28)
        wall = float(np.median(walls))
        e2e = {"what": "MRATree(locs, r0, cov, y_obs, R, M, J) + getLikelihood() + predict(): host tree replay + uploads + device pass + download",
               "wall_s": wall, "wall_s_all": walls, "value": n_nodes / wall, "unit": "nodes/s",
               "likelihood_rel_diff_vs_resident_pass": abs(float(lik[0, 0]) - (d + u)) / abs(d + u)}
        if "cpu_baseline" in out:
            # like for like: constructor-equivalent CPU work per node against the constructor wall-clock per node
            e2e["vs_cpu_baseline"] = e2e["value"] / out["cpu_baseline"]["value"]
        out["end_to_end"] = e2e
        out["value_end_to_end"] = e2e["value"]
    if rank == 0:
        if json_fd is None:
            print(json.dumps(out))
        else:
            sys.stdout.flush()
            os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
