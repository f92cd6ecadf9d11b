"""Register / scratch budget of the kernels in the product build ("a kernel-resource report with no scratch
is part of done").  __graft_entry__.build() compiles every translation unit with -Rpass-analysis=kernel-resource-usage and keeps
the remarks of that very build next to the library (pymra_amd/libmra_hip.resource_usage.txt); this test parses them - no
second compile - and fails when a kernel that a C3 / C5 pass launches (one GPU or a shard) uses scratch memory, or when any
other kernel starts to.  Exceptions among the production kernels are PINNED (exact byte counts: growth fails the test, and so
does an improvement that forgets to delete the entry); none at present.

ALLOWED lists the known exceptions with their bound and the reason; anything else must report 0 bytes of scratch."""
import os
import re
import sys

import pytest

import _cases as K

sys.path.insert(0, os.path.join(K.ROOT, "tools"))

# kernels launched by a likelihood+predict pass of BASELINE config 3 (fused path, 2-D, Matern32 = mode 0) and of config 5
# (level-by-level path), sharded or not
PRODUCTION = [
    r"k_knot_chain<2, 8, 2, 0>", r"k_prior_cascade<2, 8, 2, 0, true>", r"k_leaf_gemm<2, 2, 0, 1, 8, 512, 4, 0>",
    r"k_chol_wave<12>", r"k_trsm_rows2<8>", r"k_trsm_rows2<12>", r"k_front<true>", r"k_leaf_solve_update<8, 13, true>",
    r"k_gemm_nt_lds<2, 2, 0>", r"k_gemm_nt_lds<1, 2, 0>", r"k_gemm_nt<0, 2, 0>", r"k_gemm_nt<1, 2, 0>", r"k_panel_chol",
    r"k_trsm_rows2<4>", r"k_front<false>", r"k_sum_dnode", r"k_leaf_cphantom", r"k_assemble", r"k_leaf_moments",
    # shards (at most two leaves per CU) and config 5
    r"k_chol_tiles<8, 4>", r"k_chol_tiles<10, 4>", r"k_predict_hi<4, true>", r"k_predict_hi<4, false>", r"k_parent_front<2>", r"k_parent_front<4>", r"k_parent_front<8>",
    r"k_predict_cascade<2, 6, 4, true, 3, 2>", r"k_predict_cascade<2, 6, 4, false, 3, 2>", r"k_predict_cascade<4, 4, 8, false, 1, 2>",
    # (round 4: two barriers per level and the Ut chunks by LDS DMA: every instantiation is without scratch, whatever its shape)
    r"k_predict_cascade<",
    r"k_parent_front<12>", r"k_syrk_blk<0, 0>", r"k_syrk_dma<0>",
]
# production kernels with a known, pinned amount of scratch: (pattern, exact bytes per lane)
PINNED = []      # (round 4: k_parent_front<12> lost its 36 B with the blocked factorisation atom, the predictive cascades theirs with half staging)
# (pattern, max scratch bytes per lane, reason)
ALLOWED = [
    # Kanter taper: ocml sin/cos (Payne-Hanek range reduction keeps a private table on the stack); not a spill
    (r"k_prior_cascade<\d, \d, \d, 3, (true|false)>", 544, "ocml sincos stack"),
    (r"k_knot_chain<\d, \d, \d, 3>", 544, "ocml sincos stack"),
    (r"k_leaf_gemm<2, \d, 3,", 544, "ocml sincos stack"),
    (r"k_gemm_nt(_lds)?<2, \d, 3>", 544, "ocml sincos stack"),
    (r"k_eval_kernel", 544, "ocml sincos stack"),
    # one-launch prior level at three workgroups per CU (168 registers; two are slower: 26.1 vs 25.4 ms at config 5): row pointers
    # and lane offsets of the epilogue are parked in scratch across the K loop (ISA: no scratch access inside the loop of 32 MFMAs)
    (r"k_leaf_gemm<2, \d, [012], 2, 4, 256, 3, 1>", 64, "epilogue addresses parked across the K loop"),
    (r"k_prior_cascade<4, 4, \d, \d, true>", 40, "CWT = 4 trees with <= 4 levels only"),
    # the 8-level row cascade sits at the 256-register limit of two waves per SIMD (253-255 in every covariance family); the row
    # number of a gathered (likelihood-only) tile is one more live value and costs the family with the longest polynomial four
    # registers (ISA: parked across the level loop, not inside a level's products).  The C3 instantiation (<2, 8, 2, 0, true>) has none.
    (r"k_prior_cascade<2, 8, 2, 2, true>", 24, "gathered-tile row number at the 256-register limit"),
]


@pytest.fixture(scope="module")
def usage(built_library):
    import resource_usage
    p = os.path.join(K.ROOT, "pymra_amd", "libmra_hip.resource_usage.txt")
    if not os.path.exists(p):
        pytest.skip("no resource-usage remarks next to the library (built outside __graft_entry__.build / make lib)")
    res = resource_usage.parse(open(p).read())
    assert len(res) > 150
    return res


def _find(usage, pat):
    return {n: r for n, r in usage.items() if pat in n}


def test_production_kernels_use_no_scratch(usage):
    for pat in PRODUCTION:
        hits = _find(usage, pat)
        assert hits, "kernel %s not found in the remarks" % pat
        for n, r in hits.items():
            assert r["scratch"] == 0, "%s: %d B/lane of scratch, %s spilled registers" % (n, r["scratch"], r["spills"])


def test_pinned_exceptions_are_exact(usage):
    for pat, nbytes in PINNED:
        hits = _find(usage, pat)
        assert hits, "kernel %s not found in the remarks" % pat
        for n, r in hits.items():
            assert r["scratch"] == nbytes, "%s: %d B/lane of scratch, pinned at %d" % (n, r["scratch"], nbytes)


def test_no_other_kernel_starts_to_spill(usage):
    bad = []
    for n, r in sorted(usage.items()):
        if not r["scratch"]:
            continue
        lim = [b for p, b, _ in ALLOWED if re.search(p, n)]
        if not lim or r["scratch"] > max(lim):
            bad.append("%s: %d B/lane (%s spills)" % (n, r["scratch"], r["spills"]))
    assert not bad, "\n".join(bad)


def test_mfma_kernels_keep_their_accumulators_out_of_agprs(usage):
    """With the full 512-register budget hipcc parks MFMA accumulators in AGPRs and copies them around every loop iteration
    (DESIGN.md section 5); every MFMA kernel of the pass is cut for <= 256 registers.  The knot chain (one workgroup per CU,
    latency-bound) is the one kernel allowed to use them."""
    for n, r in usage.items():
        if "k_knot_chain" in n:
            continue
        assert not r["agprs"], "%s uses %s AGPRs" % (n, r["agprs"])
