#!/usr/bin/env python3
"""Generate tests/golden/c5_levelwise.npz: the CPU level-wise oracle (oracle/mra_levelwise.py, depth-first
memory-lean order) on the full BASELINE config 5 geometry (2048x2048 grid, M=8, J=4, r0=64, Matern32;
87381 nodes, 4.2 M rows), sampled.

The reference itself cannot be run at this size in the build container (its per-node Python recursion took 27 min at
C3 and holds every BTil list), so the whole-tree anchor of C5 is the level-wise oracle, which is
pinned against reference runs on 14 smaller cases and against the reference's own C3 run
(tests/test_oracle.py, c3.npz vs c3_levelwise.npz).

    python tests/golden/make_levelwise_c5.py          # ~25 GB RSS, build container or GPU box

Data only: inputs come from bench.make_inputs (the seeded recipe of SURVEY.md section 8(d)), outputs are stored sampled
(every 251st location + the first 4096) plus whole-vector sums.
"""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)

import bench                                              # noqa: E402  (input recipe only)
import pymra_amd.MRATools as mt                           # noqa: E402
from pymra_amd.topology import build_topology             # noqa: E402
from oracle.mra_levelwise import run_levelwise            # noqa: E402


def main(name="c5", stride=251):
    c = bench.CONFIGS[name]
    locs, y_obs = bench.make_inputs(c)
    topo = build_topology(locs, c["r"], c["M"], c["J"])
    spec = mt.KernelSpec(mt.KIND_MATERN32 if c["kern"] == "m32" else mt.KIND_EXP, c["l"], c["sig"])
    t0 = time.time()
    out = run_levelwise(topo, locs, spec, y_obs, c["R"], lean=True)
    wall = time.time() - t0
    N = len(locs)
    idx = np.unique(np.concatenate([np.arange(0, N, int(stride)), np.arange(0, 4096)]))
    mean, sd = out["mean"], out["sd"]
    np.savez_compressed(os.path.join(HERE, name + "_levelwise.npz"), params=json.dumps(c), lik=out["lik"], d=out["d"], u=out["u"],
                        sample_idx=idx, mean=mean[idx], sd=sd[idx], mean_sum=float(mean.sum()), sd_sum=float(sd.sum()),
                        mean_sq=float((mean ** 2).sum()), sd_sq=float((sd ** 2).sum()),
                        y_checksum=float(np.nansum(y_obs)), oracle_wall_s=wall, n_nodes=int(topo.n_nodes))
    print("%s levelwise: lik=%.10f wall=%.1fs samples=%d" % (name, out["lik"], wall, len(idx)), flush=True)


if __name__ == "__main__":
    main(*(sys.argv[1:3]))
