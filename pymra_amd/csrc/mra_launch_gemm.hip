// mra_launch_gemm.hip - dispatch of the batched GEMM kernels (k_gemm_nt, k_gemm_nt_lds, k_leaf_gemm) over epilogue, dimension and
// covariance family; a translation unit of its own so that libmra_hip.so builds in parallel.
#define MRA_KERNELS_TEMPLATES_ONLY
#include "mra_plan_types.h"

static inline unsigned gemm_grid_x(long M, long N, bool lower_tri) {
    const long tm = (M + 31) / 32, tn = (N + 31) / 32;
    return (unsigned)(((lower_tri ? tm * (tm + 1) / 2 : tm * tn) + 3) / 4);
}

template <int EPI>
// lower_tri: every problem of the batch has .lower set and M == N (direct kernel only)
static void launch_gemm(mra_plan* pl, const GemmProb* probs, size_t nprob, long maxM, long maxN, bool allow_lds = true, bool lower_tri = false) {
    if (!nprob || maxM <= 0 || maxN <= 0) return;
    const bool lds = pl->gemm_lds && allow_lds;
    const unsigned gx = lds ? (unsigned)(((maxM + 63) / 64) * ((maxN + 63) / 64)) : gemm_grid_x(maxM, maxN, lower_tri);
    const int mode = (EPI == EPI_COV) ? pl->kp.mode : 0;
    // 1-D grid, XCD-aware (xcd_problem_tile): gx workgroups per problem, problems rounded up to 8
    const size_t chunk = (size_t)std::max<long>(8, ((0x7fffffffL / (long)gx) / 8) * 8);
    for (size_t off = 0; off < nprob; off += chunk) {
        unsigned gy = (unsigned)std::min<size_t>(chunk, nprob - off);
        // fewer than 8 problems: every problem cut into S runs of tiles, S * gy virtual problems over the 8 XCD lanes (xcd_problem_tile)
        unsigned S = 1, gxs = gx;
        if (gy < 8u && gx >= 8u) {
            S = gy == 1 ? 8u : (gy == 2 ? 4u : (gy <= 4 ? 2u : 1u));
            gxs = (gx + S - 1) / S;
            gy *= S;
        }
        dim3 grid(gxs * (((gy + 7u) / 8u) * 8u));
#define MRA_GEMM_LAUNCH(KERN, D, MD) hipLaunchKernelGGL((KERN<EPI, D, (EPI == EPI_COV ? MD : 0)>), grid, dim3(256), 0, pl->stream, probs + off, pl->kp, gxs, gy, S)
        if (lds) {
            if (pl->d == 1) { if (mode == 0) MRA_GEMM_LAUNCH(k_gemm_nt_lds, 1, 0); else if (mode == 1) MRA_GEMM_LAUNCH(k_gemm_nt_lds, 1, 1); else if (mode == 2) MRA_GEMM_LAUNCH(k_gemm_nt_lds, 1, 2); else MRA_GEMM_LAUNCH(k_gemm_nt_lds, 1, 3); }
            else { if (mode == 0) MRA_GEMM_LAUNCH(k_gemm_nt_lds, 2, 0); else if (mode == 1) MRA_GEMM_LAUNCH(k_gemm_nt_lds, 2, 1); else if (mode == 2) MRA_GEMM_LAUNCH(k_gemm_nt_lds, 2, 2); else MRA_GEMM_LAUNCH(k_gemm_nt_lds, 2, 3); }
        } else {
            if (pl->d == 1) { if (mode == 0) MRA_GEMM_LAUNCH(k_gemm_nt, 1, 0); else if (mode == 1) MRA_GEMM_LAUNCH(k_gemm_nt, 1, 1); else if (mode == 2) MRA_GEMM_LAUNCH(k_gemm_nt, 1, 2); else MRA_GEMM_LAUNCH(k_gemm_nt, 1, 3); }
            else { if (mode == 0) MRA_GEMM_LAUNCH(k_gemm_nt, 2, 0); else if (mode == 1) MRA_GEMM_LAUNCH(k_gemm_nt, 2, 1); else if (mode == 2) MRA_GEMM_LAUNCH(k_gemm_nt, 2, 2); else MRA_GEMM_LAUNCH(k_gemm_nt, 2, 3); }
        }
#undef MRA_GEMM_LAUNCH
    }
}

// leaf-resident product (one workgroup per problem): the two big leaf GEMMs of a pass.  The residual (few column tiles, long K)
// runs two row tiles per wave, the update (13 column tiles at C3, short K) one row tile per wave with all columns in one pass.
template <int EPI>
static void launch_leaf_gemm(mra_plan* pl, const GemmProb* probs, size_t nprob) {
    if (!nprob) return;
    const int mode = (EPI == EPI_COV) ? pl->kp.mode : 0;
    const dim3 grid((unsigned)nprob);
    constexpr int CT = (EPI == EPI_SUB) ? 13 : 7;
    // COV: one row tile per wave, 8 waves, 8 column tiles per pass (leaves of up to 128 observations in one pass), 128 registers =
    // four waves per SIMD (the Kanter taper's sin/cos need 217: two); dbg bit 32: the 2-row-tile, 4-wave, 7-column, 239-register shape
    const bool wide = (EPI == EPI_COV) && !(pl->dbg & 32);
#define MRA_LG_LAUNCH(D, MD) do { \
        if (wide) hipLaunchKernelGGL((k_leaf_gemm<EPI, D, (EPI == EPI_COV ? MD : 0), 1, 8, 512, (MD == 3 ? 2 : 4)>), grid, dim3(512), 0, pl->stream, probs, pl->kp); \
        else hipLaunchKernelGGL((k_leaf_gemm<EPI, D, (EPI == EPI_COV ? MD : 0), (EPI == EPI_SUB ? 1 : 2), CT, 256, 2>), grid, dim3(256), 0, pl->stream, probs, pl->kp); \
    } while (0)
    if (pl->d == 1) { if (mode == 0) MRA_LG_LAUNCH(1, 0); else if (mode == 1) MRA_LG_LAUNCH(1, 1); else if (mode == 2) MRA_LG_LAUNCH(1, 2); else MRA_LG_LAUNCH(1, 3); }
    else { if (mode == 0) MRA_LG_LAUNCH(2, 0); else if (mode == 1) MRA_LG_LAUNCH(2, 1); else if (mode == 2) MRA_LG_LAUNCH(2, 2); else MRA_LG_LAUNCH(2, 3); }
#undef MRA_LG_LAUNCH
}

// blocked segmented SYRK (k_syrk_blk): every problem is C = [I] + sum_seg (+|-) A_seg A_seg^T with .lower set, M == N == a multiple of 16,
// nseg > 0 and A == B in every segment
void mra_launch_syrk_blk(mra_plan* pl, const GemmProb* probs, size_t nprob, long M, bool dma) {
    if (!nprob || M <= 0) return;
    const long nbk = (M / 16 + SB_T - 1) / SB_T;
    const unsigned gx = (unsigned)(nbk * (nbk + 1) / 2);
    const size_t chunk = (size_t)std::max<long>(8, ((0x7fffffffL / (long)gx) / 8) * 8);
    for (size_t off = 0; off < nprob; off += chunk) {
        unsigned gy = (unsigned)std::min<size_t>(chunk, nprob - off);
        unsigned S = 1, gxs = gx;
        if (gy < 8u && gx >= 8u) {
            S = gy == 1 ? 8u : (gy == 2 ? 4u : (gy <= 4 ? 2u : 1u));
            gxs = (gx + S - 1) / S;
            gy *= S;
        }
        // dma: the stage filled by LDS DMA (k_syrk_dma: SD_MAXST steps at most, three workgroups per CU); else through registers
        if (dma) hipLaunchKernelGGL((k_syrk_dma<EPI_SET>), dim3(gxs * (((gy + 7u) / 8u) * 8u)), dim3(256), 0, pl->stream, probs + off, gxs, gy, S);
        else hipLaunchKernelGGL((k_syrk_blk<EPI_SET>), dim3(gxs * (((gy + 7u) / 8u) * 8u)), dim3(256), 0, pl->stream, probs + off, gxs, gy, S);
    }
}

// one level of the level-by-level prior in one launch (k_leaf_gemm<COV, ..., SOLVE>): problems are blocks of rows of a node, N = the
// node's block width (<= 64), solveL / solveI its factor.  Two row tiles per wave, four waves: 32 MFMAs per wave, chunk and barrier.
#ifndef MRA_PL_WPE
#define MRA_PL_WPE 3
#endif
void mra_launch_prior_level(mra_plan* pl, const GemmProb* probs, size_t nprob) {
    if (!nprob) return;
    const int mode = pl->kp.mode;
    const dim3 grid((unsigned)nprob);
#define MRA_PL_LAUNCH(D, MD) hipLaunchKernelGGL((k_leaf_gemm<EPI_COV, D, MD, 2, 4, 256, (MD == 3 ? 2 : MRA_PL_WPE), 1>), grid, dim3(256), 0, pl->stream, probs, pl->kp)
    if (pl->d == 1) { if (mode == 0) MRA_PL_LAUNCH(1, 0); else if (mode == 1) MRA_PL_LAUNCH(1, 1); else if (mode == 2) MRA_PL_LAUNCH(1, 2); else MRA_PL_LAUNCH(1, 3); }
    else { if (mode == 0) MRA_PL_LAUNCH(2, 0); else if (mode == 1) MRA_PL_LAUNCH(2, 1); else if (mode == 2) MRA_PL_LAUNCH(2, 2); else MRA_PL_LAUNCH(2, 3); }
#undef MRA_PL_LAUNCH
}

void mra_launch_gemm(mra_plan* pl, int epi, const GemmProb* probs, size_t nprob, long maxM, long maxN, bool allow_lds, bool lower_tri) {
    switch (epi) {
        case EPI_SET: launch_gemm<EPI_SET>(pl, probs, nprob, maxM, maxN, allow_lds, lower_tri); break;
        case EPI_SUB: launch_gemm<EPI_SUB>(pl, probs, nprob, maxM, maxN, allow_lds, lower_tri); break;
        case EPI_COV: launch_gemm<EPI_COV>(pl, probs, nprob, maxM, maxN, allow_lds, lower_tri); break;
        default: launch_gemm<EPI_HOSTCOV>(pl, probs, nprob, maxM, maxN, allow_lds, lower_tri); break;
    }
}
void mra_launch_leaf_gemm(mra_plan* pl, int epi, const GemmProb* probs, size_t nprob) {
    switch (epi) {
        case EPI_SUB: launch_leaf_gemm<EPI_SUB>(pl, probs, nprob); break;
        case EPI_COV: launch_leaf_gemm<EPI_COV>(pl, probs, nprob); break;
        default: launch_leaf_gemm<EPI_HOSTCOV>(pl, probs, nprob); break;
    }
}
