// mra_kernels.h - hand-written gfx950 (CDNA4) kernels of the MRA hot path.  FP64 throughout.
//
// Building blocks (all batched over the nodes of one tree level; every row range, column block
// and front size is a multiple of 16, so every MFMA tile is full):
//
//   k_gemm_nt      C (=|-=) A B^T on v_mfma_f64_16x16x4_f64, 32x32 output tile per wave, with three
//                  epilogues: SET, SUB and COV (C = kernel(x_i, x_c) - A B^T : the pairwise
//                  covariance evaluation of pyMRA/MRATools.py:229-293 fused into the conditional
//                  covariance of pyMRA/MRANode.py:73-80, 384)
//   k_panel_chol   left-looking blocked Cholesky of the leading ne*16 columns of a tall panel
//                  (one workgroup per node; 16x16 diagonal blocks factorised and inverted in
//                  registers with cross-lane shuffles, everything else on MFMA); rows below the
//                  square part come out multiplied by L^{-T}  (np.linalg.inv / cholesky / slogdet /
//                  scipy.linalg.inv of pyMRA/MRANode.py:387-391, 444-463)
//   k_trsm_rows    X = R L^{-T} for tall row ranges, 16 rows per wave, X^T blocks stay in the MFMA
//                  accumulator layout and are fed back as B operands (pyMRA/MRANode.py:384-387,
//                  504-511)
//   small gather / assemble / moment kernels around them.
//
// MFMA f64 16x16x4 operand layout (guide: cdna_hip_programming.md section 3): lane l, r = l & 15, q = l >> 4
//   A[r][k = q]   B[k = q][r]   D[row = q + 4*reg][col = r]
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

// Pointers that reach a kernel inside a descriptor read from memory (GemmProb, PanelProb, ...) are
// generic to the compiler, and every access through them becomes a FLAT instruction: those count on
// lgkmcnt as well as vmcnt, so each s_waitcnt lgkmcnt(0) in front of an LDS-fed MFMA group also waits
// for the global prefetches in flight, and the loop falls back to vmcnt(0).  Telling the compiler that
// Accesses through such pointers therefore go through these helpers, which cast to the global address
// space at the access and give global_load / global_store.
#define MRA_AS1 __attribute__((address_space(1)))
__device__ __forceinline__ double gld(const double* p) { return *(const double MRA_AS1*)p; }
__device__ __forceinline__ int gldi(const int* p) { return *(const int MRA_AS1*)p; }
__device__ __forceinline__ d2 gld2(const double* p) { return *(const d2 MRA_AS1*)p; }
__device__ __forceinline__ d4 gld4(const double* p) { return *(const d4 MRA_AS1*)p; }
__device__ __forceinline__ void gst(double* p, double v) { *(double MRA_AS1*)p = v; }
__device__ __forceinline__ void gst2(double* p, d2 v) { *(d2 MRA_AS1*)p = v; }
__device__ __forceinline__ void gst4(double* p, d4 v) { *(d4 MRA_AS1*)p = v; }


#define MRA_YB 16
// what-if timing switches that change results exist only in the diagnostic build (-DMRA_WHATIF); the product compiles them out
#ifdef MRA_WHATIF
#define MRA_WHATIF_BIT(dbg, bit) ((dbg) & (bit))
#else
#define MRA_WHATIF_BIT(dbg, bit) false
#endif
#define FT_LD 18                /* row stride (doubles) of a 16x16 tile held in LDS by the front / knot-chain kernels */
#define FT_SZ (16 * FT_LD)

// row permutation of the "vec" tile layout (see k_trsm_rows2)
// 16 bytes per lane from global memory straight into LDS (global_load_lds_dwordx4, gfx950): lane l's piece lands at
// lds_wave_base + 16 l - no staging registers, no ds_write.  lds_wave_base must be wave-uniform; completion is counted by vmcnt
// (mra_wait_vm0() before the barrier that publishes the data).
__device__ __forceinline__ void gld_lds16(const void* g, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ void mra_wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ int pi16(int rho) { return ((rho & 3) << 2) | (rho >> 2); }

__device__ __forceinline__ d4 mfma16(double a, double b, d4 c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// "row-on-lane" tile access: element s of lane (r,q) is tile[r][q + 4 s].  The same pattern is
// (i) the A operand of k-step s, (ii) the B operand of k-step s for a transposed product, and
// (iii) the accumulator layout of a transposed 16x16 result - so results chain without shuffles.
__device__ __forceinline__ d4 load_rowlane(const double* __restrict__ p, long ld, int r, int q) {
    const double* a = p + (long)r * ld + q;
    d4 v;
    v[0] = gld(a); v[1] = gld(a + 4); v[2] = gld(a + 8); v[3] = gld(a + 12);
    return v;
}
__device__ __forceinline__ void store_rowlane(double* __restrict__ p, long ld, int r, int q, d4 v) {
    double* a = p + (long)r * ld + q;
    gst(a, v[0]); gst(a + 4, v[1]); gst(a + 8, v[2]); gst(a + 12, v[3]);
}

// ------------------------------------------------------------------------------------------------
//  covariance kernels (pyMRA/MRATools.py:256-301); same operation order as the reference
// ------------------------------------------------------------------------------------------------
struct KernelParams {
    int kind;
    int d;
    double l, sig, scale;
    // derived on the host (derive_kernel_params): value = amp * (1 + a1 t + a2 t^2) * exp(-t)
    //   mode 0: t = c * D / l   (ExpCovFun: c=1,a1=a2=0; Matern32: c=sqrt3,a1=1; Matern52: c=sqrt5,a1=1,a2=1/3)
    //   mode 1: t = D^2 / (2 l^2)   (GaussianCovFun)          mode 2: [D == 0]   (Iden)
    //   mode 3: Kanter taper of radius l (KanterCovFun, compact support), c_inv_l = 1/radius
    int mode;
    int circular;           // 1-D only: D = min(|a-b|, 1-|a-b|)  (torus of length 1, pyMRA/MRATools.py:232-239)
    double c_inv_l, inv_2l2, a1, a2, amp;
};
#define MRA_FAR_AWAY 1.0e150   /* coordinate of a phantom knot: every kernel gives exactly 0 there */

// exp(-t), t >= 0.  The ocml exp()/sqrt() are an order of magnitude more instructions (denormal/NaN/overflow care that cannot
// occur here) and made the covariance evaluation the bottleneck of every kernel using it; and FP64 VALU work shares the pipe
// with the FP64 MFMAs, so every instruction of an evaluation is taken from the matrix rate.
//   MRA_EXP_VARIANT 0: t = k ln2 + r, |r| <= ln2/2, degree-13 Taylor (remainder < 4e-18), v_ldexp_f64: 21 FP64 instructions
//   MRA_EXP_VARIANT 2: t = (64 k + j) ln2/64 + r, |r| <= ln2/128, degree-5 Taylor (remainder 3.5e-17 relative), times 2^(-j/64)
//                      from a 64-entry table that lives ONE ENTRY PER LANE in a register pair (computed from the lane number,
//                      which the compiler hoists to the top of the kernel) and is read with two ds_bpermute_b32 - the LDS
//                      crossbar, not the FP64 pipe: 12 FP64 instructions
//   MRA_EXP_VARIANT 1: quarter steps, t = (4 k + j) ln2/4 + r, |r| <= ln2/8, degree 9, 2^(-j/4) by selects: 16 FP64 instructions
// Measured at C3 (round 4, A/B builds by tools/build_variant.sh): variant 0 IS the fastest - row cascade 1.021 ms against 1.051
// (variant 1) and 1.058 (variant 2), leaf residual 0.942 against 0.953 / 0.961, pass 5.61 against 5.65 / 5.67 ms.  What the
// evaluation costs is VALU ISSUE slots, not FP64 operations: the selects, shifts and conversions that pay for a shorter polynomial
// are instructions too, and the table's register pair pushes the row cascade (255 registers) into 80 B of scratch.
#ifndef MRA_EXP_VARIANT
#define MRA_EXP_VARIANT 0
#endif
#ifndef MRA_SQRT_SHORT
#define MRA_SQRT_SHORT 0
#endif
__device__ __forceinline__ double exp_neg_poly13(double t) {
    const double kf = __builtin_rint(t * 1.4426950408889634074);
    double r = __builtin_fma(kf, 6.93147180369123816490e-01, -t);      // k*ln2_hi - t
    r = __builtin_fma(kf, 1.90821492927058770002e-10, r);              // + k*ln2_lo  ->  r = -(t - k ln2)
    double p = 1.6059043836821613e-10;                                  // 1/13!
    p = __builtin_fma(p, r, 2.08767569878681e-09);
    p = __builtin_fma(p, r, 2.505210838544172e-08);
    p = __builtin_fma(p, r, 2.755731922398589e-07);
    p = __builtin_fma(p, r, 2.7557319223985893e-06);
    p = __builtin_fma(p, r, 2.48015873015873e-05);
    p = __builtin_fma(p, r, 1.984126984126984e-04);
    p = __builtin_fma(p, r, 1.388888888888889e-03);
    p = __builtin_fma(p, r, 8.333333333333333e-03);
    p = __builtin_fma(p, r, 4.1666666666666664e-02);
    p = __builtin_fma(p, r, 1.6666666666666666e-01);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    return __builtin_amdgcn_ldexp(p, -(int)kf);
}
__device__ __forceinline__ double exp_neg(double t) {
    t = fmin(t, 1100.0);
#if MRA_EXP_VARIANT == 0
    return exp_neg_poly13(t);
#elif MRA_EXP_VARIANT == 1
    const double kf = __builtin_rint(t * 5.7707801635558536296);       // 4 / ln2
    double r = __builtin_fma(kf, 1.73286795092280954123e-01, -t);      // k*(ln2/4)_hi - t   (hi: 33 bits, exact products)
    r = __builtin_fma(kf, 4.77053732317646925e-11, r);                 // + k*(ln2/4)_lo
    double p = 2.7557319223985893e-06;                                  // 1/9!
    p = __builtin_fma(p, r, 2.48015873015873e-05);
    p = __builtin_fma(p, r, 1.984126984126984e-04);
    p = __builtin_fma(p, r, 1.388888888888889e-03);
    p = __builtin_fma(p, r, 8.333333333333333e-03);
    p = __builtin_fma(p, r, 4.1666666666666664e-02);
    p = __builtin_fma(p, r, 1.6666666666666666e-01);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    const int ki = (int)kf;
    const double c01 = (ki & 1) ? 8.40896415253714543e-01 : 1.0;       // 2^(-1/4) : 1
    const double c23 = (ki & 1) ? 5.94603557501360533e-01 : 7.07106781186547524e-01;      // 2^(-3/4) : 2^(-1/2)
    p *= (ki & 2) ? c23 : c01;
    return __builtin_amdgcn_ldexp(p, -(ki >> 2));
#else
    // this lane's table entry 2^(-lane/64): a pure function of the lane number, hoisted out of every loop by the compiler
    const int lane_ = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const double tab = exp_neg_poly13((double)lane_ * 1.0830424696249145e-02);      // lane * ln2 / 64
    const double kf = __builtin_rint(t * 92.332482616893658074);       // 64 / ln2
    double r = __builtin_fma(kf, 1.08304246932675596327e-02, -t);      // k*(ln2/64)_hi - t
    r = __builtin_fma(kf, 2.98158582698529328e-12, r);                 // + k*(ln2/64)_lo
    double p = 8.333333333333333e-03;                                   // 1/5!
    p = __builtin_fma(p, r, 4.1666666666666664e-02);
    p = __builtin_fma(p, r, 1.6666666666666666e-01);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    const int ki = (int)kf;
    const int sel = (ki & 63) << 2;
    const double tj = __hiloint2double(__builtin_amdgcn_ds_bpermute(sel, __double2hiint(tab)), __builtin_amdgcn_ds_bpermute(sel, __double2loint(tab)));
    return __builtin_amdgcn_ldexp(p * tj, -(ki >> 6));
#endif
}

// sqrt(x), x >= 0 finite: v_rsq_f64 seed (2^-23 relative), two Goldschmidt steps and one Newton correction with the residual
// x - g g (<= 1 ulp).  MRA_SQRT_SHORT: ONE Goldschmidt step (-> 2^-45) before the correction (-> rounding level): A/B switch.
__device__ __forceinline__ double sqrt_pos(double x) {
    const double y = __builtin_amdgcn_rsq(fmax(x, 1.0e-300));      // x == 0 stays 0 through every step below
    double g = x * y, h = 0.5 * y;
    const double e = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, e, g);
    h = __builtin_fma(h, e, h);
#if !MRA_SQRT_SHORT
    {
        const double e2 = __builtin_fma(-h, g, 0.5);
        g = __builtin_fma(g, e2, g);
        h = __builtin_fma(h, e2, h);
    }
#endif
    // one residual correction: g += (x - g*g) * h
    const double res = __builtin_fma(-g, g, x);
    return __builtin_fma(res, h, g);
}

// value of the kernel at squared distance D2 (the reference evaluates the same formulas on
// D = cdist(...), pyMRA/MRATools.py:265-301; differences are at the 1-2 ulp level).  Straight-line
// code: the only branches are on wave-uniform kernel parameters.
template <int MODE>
__device__ __forceinline__ double cov_of_dist2(const KernelParams& kp, double D2) {
    if (MODE == 2) return (D2 == 0.0) ? kp.amp : 0.0;
    if (MODE == 3) {                                 // pyMRA/MRATools.py:318-323
        const double D = fmin(sqrt_pos(D2) * kp.c_inv_l, 2.0);
        const double p2 = 6.283185307179586 * D;
        const double v = (1.0 - D) * sin(p2) / p2 + 0.3183098861837907 * (1.0 - cos(p2)) / p2;
        return kp.amp * ((D == 0.0) ? 1.0 : ((D > 1.0) ? 0.0 : v));
    }
    const double t = (MODE == 1) ? D2 * kp.inv_2l2 : sqrt_pos(D2) * kp.c_inv_l;
    const double poly = __builtin_fma(__builtin_fma(kp.a2, t, kp.a1), t, 1.0);
    return kp.amp * (poly * exp_neg(t));
}
__device__ __forceinline__ double cov_of_dist(const KernelParams& kp, double D) {
    return kp.mode == 0 ? cov_of_dist2<0>(kp, D * D) : (kp.mode == 1 ? cov_of_dist2<1>(kp, D * D) : (kp.mode == 2 ? cov_of_dist2<2>(kp, D * D) : cov_of_dist2<3>(kp, D * D)));
}

template <int DIM>
__device__ __forceinline__ double pair_dist2(const double* __restrict__ xa, const double* __restrict__ xb, int circular = 0) {
    if (DIM == 1) {
        double dx = fabs(xa[0] - xb[0]);
        if (circular) dx = fmin(dx, fabs(1.0 - dx));
        return dx * dx;
    }
    const double dx = xa[0] - xb[0], dy = xa[1] - xb[1];
    return dx * dx + dy * dy;
}

// diagnostics / tests: kernel values for an array of distances
#ifndef MRA_KERNELS_TEMPLATES_ONLY      /* non-template kernel: defined once, in the translation unit of mra_plan.hip */
__global__ void k_eval_kernel(const double* __restrict__ D, double* __restrict__ out, long n, KernelParams kp) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = cov_of_dist(kp, D[i]);
}
#endif

// ------------------------------------------------------------------------------------------------
//  batched C = f(A B^T)
// ------------------------------------------------------------------------------------------------
struct GemmProb {
    const double* A;        // M x K, row-major, lda
    const double* B;        // N x K, row-major, ldb (rows optionally gathered through idxB)
    double* C;              // M x N, row-major, ldc
    const int* idxB;        // nullptr, or N row numbers into B (-1: zero row)
    const double* XA;       // COV: coordinates of A's rows (row 0 of this problem), stride d
    const double* XB;       // COV: coordinates addressed by idxB (or by column when idxB == nullptr)
    const double* Csrc;     // HOSTCOV: host-evaluated covariance block, ldcs
    long lda, ldb, ldc, ldcs;
    int M, N, K;
    int lower;              // 1: only 32x32 tiles with mt >= nt
    int zc;                 // SUB: columns >= zc take C_in = 0 (0 = off)
    // COV, leaves: rows that are observed are also copied (plus diag_add on the diagonal) into the
    // square block C2 (ldc): C2[rowmap[row]][col]
    const int* rowmap;
    double* C2;
    double diag_add;
    // K split into segments (sum over a node's children): when nseg > 0, A/B/lda/ldb/K are ignored
    const struct GemmSeg* segs;
    int nseg;
    int diag_one;           // SET: add 1 on the diagonal for row == col < diag_one (front identity block)
    // k_gemm_nt_lds only: rows of A gathered through idxA (-1: phantom).  With sym_diag (COV) the result is
    // the square block kernel(x_a, x_b) - A B^T + diag_add * I over the SAME index list on both sides,
    // identity on phantoms: the leaves' C = v_m(o,o) + R I in likelihood-only runs
    const int* idxA;
    int sym_diag;
    // k_leaf_gemm<..., SOLVE>: the row solve rides on the COV epilogue, C = (kernel - A B^T) L^-T with the N x N lower factor solveL
    // (row-major, ld = N) and its inverted 16 x 16 diagonal blocks solveI; var[row] -= |C[row]|^2
    const double* solveL;
    const double* solveI;
    double* var;
};
struct GemmSeg { const double* A; const double* B; long lda, ldb; int K; int neg; };   // neg: the segment is SUBTRACTED (k_gemm_nt only)

enum { EPI_SET = 0, EPI_SUB = 1, EPI_COV = 2, EPI_HOSTCOV = 3 };

// ------------------------------------------------------------------------------------------------
//  XCD-aware problem mapping for the batched kernels.  Workgroups are dealt round-robin over the 8
//  XCDs (blocks b and b+8 share one, each XCD has its own 4 MiB L2), and the G workgroups of one
//  problem (one leaf / one node) re-read the same A/B panels.  The grid is 1-D, G * 8 * ceil(nprob/8)
//  blocks; block b -> XCD lane b & 7, slot b >> 3; problem = (slot / G) * 8 + lane, tile = slot % G:
//  all tiles of a problem run back to back behind ONE L2 instead of being spread over eight.
//  Placement is a speed matter only; nothing depends on it for correctness.
// ------------------------------------------------------------------------------------------------
// Fewer than 8 problems (the top levels of a tree: ONE problem of 4 M rows at level 0 of config 5) would leave XCDs idle - all tiles
// of a problem behind one L2 means all of them on one XCD: the host then cuts every problem into S = 8, 4 or 2 runs of G consecutive
// tiles and launches S * nprob virtual problems (virtual problem v = real problem v / S, tiles [(v % S) G, (v % S + 1) G)).
__device__ __forceinline__ bool xcd_problem_tile(unsigned G, unsigned nprob, unsigned& prob, unsigned& tile, unsigned S = 1) {
    const unsigned lane = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    const unsigned grp = slot / G;
    tile = slot - grp * G;
    prob = grp * 8u + lane;
    if (prob >= nprob) return false;
    if (S > 1) { tile += (prob % S) * G; prob /= S; }
    return true;
}

// The descriptor is read field by field through a wave-uniform pointer (scalar loads into SGPRs).  A by-value copy
// whose address escapes (a by-reference lambda capture, pb.segs[sg]) lives in scratch memory: every pb.* access of
// the epilogue was a scratch load (176 B per lane).  __launch_bounds__(256, 4) keeps the kernel within 128
// registers: with the full 512 the compiler parks the accumulators in AGPRs and copies all 32 registers of them to
// VGPRs and back around every K-step (v_accvgpr_read/write: 64 VALU moves per 16 MFMAs).
template <int EPI, int DIM, int MODE>
__device__ __forceinline__ void gemm_nt_emit(const GemmProb* __restrict__ pp, const KernelParams& kp, d4 acc, int mb, int nb, int bcol, int r, int q) {
    const int col = nb + r;
    double* const C = pp->C;
    const long ldc = pp->ldc;
    double xb[DIM];
    if (EPI == EPI_COV) {
        const long bc = bcol < 0 ? 0 : bcol;
#pragma unroll
        for (int c = 0; c < DIM; ++c) xb[c] = gld(pp->XB + bc * DIM + c);
    }
    const int* rowmap = (EPI == EPI_COV || EPI == EPI_HOSTCOV) ? pp->rowmap : nullptr;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int row = mb + q + 4 * s;
        double* cp = C + (long)row * ldc + col;
        double v;
        if (EPI == EPI_SET) v = acc[s] + ((row == col && row < pp->diag_one) ? 1.0 : 0.0);
        else if (EPI == EPI_SUB) v = -acc[s];
        else if (EPI == EPI_COV) {
            double xa[DIM];
#pragma unroll
            for (int c = 0; c < DIM; ++c) xa[c] = gld(pp->XA + (long)row * DIM + c);
            const double cv = cov_of_dist2<MODE>(kp, pair_dist2<DIM>(xa, xb, kp.circular)) - acc[s];
            v = (bcol < 0) ? 0.0 : cv;
            if (rowmap) {
                const int op = gldi(rowmap + row);
                if (op >= 0) gst(pp->C2 + (long)op * ldc + col, v + (op == col ? pp->diag_add : 0.0));
            }
        } else {
            v = (bcol < 0) ? 0.0 : gld(pp->Csrc + (long)row * pp->ldcs + col) - acc[s];
            if (rowmap) {
                const int op = gldi(rowmap + row);
                if (op >= 0) gst(pp->C2 + (long)op * ldc + col, v + (op == col ? pp->diag_add : 0.0));
            }
        }
        gst(cp, v);
    }
}

template <int EPI, int DIM, int MODE>
__global__ __launch_bounds__(256, 4) void k_gemm_nt(const GemmProb* __restrict__ probs, KernelParams kp, unsigned G, unsigned nprob, unsigned S = 1) {
    unsigned prob_i, wg_i;
    if (!xcd_problem_tile(G, nprob, prob_i, wg_i, S)) return;
    const GemmProb* __restrict__ pp = probs + __builtin_amdgcn_readfirstlane(prob_i);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    const int pM = pp->M, pN = pp->N;
    const int tn = (pN + 31) >> 5, tm = (pM + 31) >> 5;
    const long tile = (long)wg_i * 4 + wave;
    int mt, nt;
    if (pp->lower) {
        // lower-triangular products (SYRK, Schur): only the tm(tm+1)/2 tiles on or below the diagonal are
        // enumerated (the host sizes G for them), so no workgroup is launched just to exit
        if (tile >= (long)tm * (tm + 1) / 2) return;
        mt = 0;
        while ((long)(mt + 1) * (mt + 2) / 2 <= tile) ++mt;
        nt = (int)(tile - (long)mt * (mt + 1) / 2);
    } else {
        if (tile >= (long)tm * tn) return;
        mt = (int)(tile / tn); nt = (int)(tile % tn);
    }
    const int m0 = mt << 5, n0 = nt << 5;
    const bool mv1 = (m0 + 16) < pM, nv1 = (n0 + 16) < pN;

    const int* idxB = pp->idxB;
    int br0 = n0 + r, br1 = n0 + 16 + r;
    if (idxB) {
        br0 = gldi(idxB + n0 + r);
        br1 = nv1 ? gldi(idxB + n0 + 16 + r) : -1;
    }
    const bool bz0 = br0 < 0, bz1 = (!nv1) || br1 < 0;
    const d4 zero = {0, 0, 0, 0};
    d4 c00 = zero, c01 = zero, c10 = zero, c11 = zero;
    // SUB: the C tile is fetched before the products (independent loads in flight together), not element
    // by element behind the stores of the epilogue, and goes straight into the accumulators (negated; the epilogue
    // writes -acc = C_in - A B^T)
    if (EPI == EPI_SUB) {
        const double* Cin = pp->C;
        const long ldc = pp->ldc;
        const int zc = pp->zc;
        d4 cin[4] = {zero, zero, zero, zero};
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) {
            const int mb = m0 + (t4 >> 1) * 16, nb = n0 + (t4 & 1) * 16;
            const int col = nb + r;
            if (mb < pM && nb < pN && !(zc > 0 && col >= zc)) {
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) cin[t4][s4] = gld(Cin + (long)(mb + q + 4 * s4) * ldc + col);
            }
        }
        c00 = -cin[0]; c01 = -cin[1]; c10 = -cin[2]; c11 = -cin[3];
    }
    const int pnseg = pp->nseg;
    const GemmSeg* segs = pp->segs;
    const int nseg = pnseg > 0 ? pnseg : 1;
    // The K dimension is a flat sequence of (segment, 16-column step) pairs: the operands of step i+1 are in flight while step i is on
    // the MFMA pipe ACROSS segment boundaries too.  (Segments are short when they are the leaves' Ut blocks - 32 observations at
    // config 5, two steps - and a prefetch pipeline that restarts in every segment spends most of its time waiting for loads.)
    // Every load is unconditional (rows that do not exist read a valid row and are zeroed by a select): branches around loads force
    // s_waitcnt vmcnt(0) in the loop.
    auto seg_get = [&](int sg, const double*& Ap, const double*& Bp, long& lda, long& ldb, int& K, bool& neg) {
        if (pnseg > 0) { const GemmSeg* g = segs + sg; Ap = g->A; Bp = g->B; lda = g->lda; ldb = g->ldb; K = g->K; neg = g->neg != 0; }
        else { Ap = pp->A; Bp = pp->B; lda = pp->lda; ldb = pp->ldb; K = pp->K; neg = false; }
    };
    int sg = 0, Kc = 0, k0 = 0;
    const double *a0p = nullptr, *a1p = nullptr, *b0p = nullptr, *b1p = nullptr;
    bool negc = false;
    auto seg_enter = [&](int s_) -> bool {                  // first segment >= s_ with K > 0; sets the operand pointers
        for (sg = s_; sg < nseg; ++sg) {
            const double *Ap, *Bp;
            long lda, ldb;
            int Ks;
            bool negs;
            seg_get(sg, Ap, Bp, lda, ldb, Ks, negs);
            if (Ks > 0) {
                Kc = Ks; negc = negs;                       // (an empty segment leaves the state of the last real one alone)
                a0p = Ap + (long)(m0 + r) * lda + 4 * q;
                a1p = mv1 ? a0p + 16 * lda : a0p;
                b0p = Bp + (long)(bz0 ? 0 : br0) * ldb + 4 * q;
                b1p = Bp + (long)(bz1 ? 0 : br1) * ldb + 4 * q;
                k0 = 0;
                return true;
            }
        }
        return false;
    };
    bool more = seg_enter(0);
    d4 na0 = zero, na1 = zero, nb0 = zero, nb1 = zero;
    bool nneg = false;
    if (more) { na0 = gld4(a0p); na1 = gld4(a1p); nb0 = gld4(b0p); nb1 = gld4(b1p); nneg = negc; }
    while (more) {
        const d4 a0 = nneg ? -na0 : na0, a1 = mv1 ? (nneg ? -na1 : na1) : zero, b0 = bz0 ? zero : nb0, b1 = bz1 ? zero : nb1;
        k0 += 16;
        if (k0 >= Kc) more = seg_enter(sg + 1);
        {
            const int ko = more ? k0 : Kc - 16;             // past the end: re-read the last step (unconditional loads, see above)
            na0 = gld4(a0p + ko); na1 = gld4(a1p + ko); nb0 = gld4(b0p + ko); nb1 = gld4(b1p + ko); nneg = negc;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            c00 = mfma16(a0[j], b0[j], c00);
            c01 = mfma16(a0[j], b1[j], c01);
            c10 = mfma16(a1[j], b0[j], c10);
            c11 = mfma16(a1[j], b1[j], c11);
        }
    }
    // epilogue: accumulator element s of lane (r,q) is C[m + q + 4 s][n + r]
    const int bc0 = idxB ? br0 : n0 + r, bc1 = idxB ? br1 : n0 + 16 + r;
    gemm_nt_emit<EPI, DIM, MODE>(pp, kp, c00, m0, n0, bc0, r, q);
    if (nv1) gemm_nt_emit<EPI, DIM, MODE>(pp, kp, c01, m0, n0 + 16, bc1, r, q);
    if (mv1) gemm_nt_emit<EPI, DIM, MODE>(pp, kp, c10, m0 + 16, n0, bc0, r, q);
    if (mv1 && nv1) gemm_nt_emit<EPI, DIM, MODE>(pp, kp, c11, m0 + 16, n0 + 16, bc1, r, q);
}

// ------------------------------------------------------------------------------------------------
//  LDS-tiled variant of k_gemm_nt: 64x64 output tile per workgroup (4 waves, 32x32 each), K stepped
//  by 16 through a double-buffered LDS stage (one barrier per step, global loads of step k+1 in
//  flight while step k is on the MFMA pipe).  Halves the L1/L2 operand traffic per MFMA of the
//  direct-load kernel above; same GemmProb, same epilogues.
// ------------------------------------------------------------------------------------------------
// the K dimension of a segmented product as a flat table of 16-column steps (built in LDS by the kernels that walk it)
#define SB_MAXST 256     /* 16-column steps (and segments) per problem: K <= 4096 */
struct SbStep { const double* A; int lda; int k0s; };                            // k0s: column offset of the step | sign bit of the segment
#define SB_MAXST2 96
struct SbStep2 { const double* A; const double* B; int lda, ldb, k0s, pad; };    // k_gemm_nt_lds (A and B sides; SB_MAXST2 steps)
#ifndef GL_LDS_LD
#define GL_LDS_LD 18      /* doubles per staged row: 16 + 2 pad -> conflict-free 32-byte fragment reads */
#endif
#define GL_PF 2           /* K-steps the global loads run ahead (register sets); even */

template <int EPI, int DIM, int MODE>
__global__ __launch_bounds__(256, 4) void k_gemm_nt_lds(const GemmProb* __restrict__ probs, KernelParams kp, unsigned G, unsigned nprob, unsigned S = 1) {
    __shared__ __attribute__((aligned(16))) double sA[2][64 * GL_LDS_LD];
    __shared__ __attribute__((aligned(16))) double sB[2][64 * GL_LDS_LD];
    constexpr bool SEGS = EPI == EPI_SET;
    // SET with segments: the K steps as a flat table in LDS (see k_syrk_blk: walking the segment descriptors inside the loop reads
    // them with vector loads and waits with vmcnt(0) - for the prefetches too - at every segment boundary)
    // (SB_MAXST2 steps at most, the host checks: with 3 KB of table four workgroups still share a CU's LDS; the per-segment step
    // counts sit in the not yet used stage while the table is built)
    __shared__ __attribute__((aligned(16))) SbStep2 sTab[SEGS ? SB_MAXST2 : 1];
    __shared__ int sNk;
    int* const sK = (int*)&sA[0][0];
    unsigned prob_i, wg_i;
    if (!xcd_problem_tile(G, nprob, prob_i, wg_i, S)) return;
    const GemmProb pb = probs[prob_i];
    const int tn = (pb.N + 63) >> 6, tm = (pb.M + 63) >> 6;
    if ((int)wg_i >= tm * tn) return;
    const int mt = wg_i / tn, nt = wg_i % tn;
    if (pb.lower && nt > mt) return;
    const int M0 = mt << 6, N0 = nt << 6;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    // wave tiles: 2 x 2 waves of 32 x 32; when at most 32 columns are left in this column tile (N = 224: the fourth
    // tile) the waves are stacked 4 x 1 with 16 x 32 each instead of leaving two of them idle
    const bool narrow = (pb.N - N0) <= 32;
    const int wr0 = narrow ? wave * 16 : (wave >> 1) * 32, wc0 = narrow ? 0 : (wave & 1) * 32;
    // ---- staging role: thread -> (row, 32-byte chunk) of A and of B
    const int srow = threadIdx.x >> 2, sch = (threadIdx.x & 3) << 2;
    bool a_ok = (M0 + srow) < pb.M;
    long arow = M0 + srow;
    if (a_ok && pb.idxA) { const int ia = gldi(pb.idxA + arow); a_ok = ia >= 0; arow = a_ok ? ia : 0; }
    const double* ap = pb.A + (a_ok ? arow : 0) * pb.lda + sch;
    long brow = N0 + srow;
    bool b_ok = brow < pb.N;
    if (b_ok && pb.idxB) { const int ib = gldi(pb.idxB + brow); b_ok = ib >= 0; brow = b_ok ? ib : 0; }
    const double* bp = pb.B + (b_ok ? brow : 0) * pb.ldb + sch;
    const d4 zero = {0, 0, 0, 0};
    d4 c00 = zero, c01 = zero, c10 = zero, c11 = zero;
    // SET: K may be split into segments (sum over a node's children, GemmSeg; `neg` segments are subtracted): the K loop below then
    // walks the step table, the prefetch running ahead across segment boundaries
    const bool segmented = SEGS && pb.nseg > 0;
    int nk = pb.K >> 4;
    int it_s = 0, it_k = 0, it_sgn = 0;
    if (SEGS && segmented) {
        const int nseg = pb.nseg;
        for (int sg = threadIdx.x; sg < nseg; sg += 256) sK[sg] = pb.segs[sg].K >> 4;
        __syncthreads();
        for (int sg = threadIdx.x; sg < nseg; sg += 256) {
            int start = 0;
            for (int j = 0; j < sg; ++j) start += sK[j];
            const int cnt = sK[sg];
            const GemmSeg g = pb.segs[sg];
            for (int t = 0; t < cnt; ++t) sTab[start + t] = SbStep2{g.A, g.B, (int)g.lda, (int)g.ldb, (16 * t) | (g.neg ? (int)0x80000000 : 0), 0};
            if (sg == nseg - 1) sNk = start + cnt;
        }
        __syncthreads();                                     // (also: sK is dead before the first stage is written)
        nk = sNk;
        if (nk <= 0) { ap = bp = (const double*)probs; nk = 0; }      // (no columns at all: the unconditional prefetches read something valid)
    }
    auto seg_fetch = [&]() {                                 // pointers of step it_s (past the end: the last step again); then one step on
        if (nk <= 0) return;
        const SbStep2 e = sTab[min(it_s, nk - 1)];
        ap = e.A + (a_ok ? arow : 0) * e.lda + sch;
        bp = e.B + (b_ok ? brow : 0) * e.ldb + sch;
        it_k = e.k0s & 0x7fffffff;
        it_sgn = e.k0s & (int)0x80000000;
        ++it_s;
    };
    // Global loads run GL_PF K-steps ahead of the MFMAs through GL_PF register sets (statically indexed:
    // the loop is unrolled by GL_PF); the LDS stage is double-buffered.  What-if runs showed this
    // kernel bound by the latency of its operand stream, not by the MFMA pipe (all MFMAs removed:
    // 1.52 -> 1.27 ms for the leaf update), with loads only two steps ahead.
    // Every load is unconditional (rows that do not exist read row 0 and are zeroed on their way into
    // LDS; steps past the end re-read the last step): a load inside a branch makes the compiler fall
    // back to s_waitcnt vmcnt(0) in the loop, which waits for the prefetches just issued as well.
    d4 ra[GL_PF], rb[GL_PF];
    int rn[GL_PF];                              // sign bit of the segment the set was loaded from
    const int klast = (nk > 0 ? nk - 1 : 0) * 16;
#pragma unroll
    for (int i = 0; i < GL_PF; ++i) {
        if (SEGS && segmented) {
            seg_fetch();
            ra[i] = gld4(ap + it_k); rb[i] = gld4(bp + it_k); rn[i] = it_sgn;
        } else {
            const int ko = i < nk ? i * 16 : klast;
            ra[i] = gld4(ap + ko);
            rb[i] = gld4(bp + ko);
            rn[i] = 0;
        }
    }
    // SUB: the C tile this wave is going to update is fetched now, not after the K loop (sixteen
    // dependent 8-byte loads at the tail of every workgroup cost ~0.27 ms of the leaf update)
    const int pm0 = M0 + wr0, pn0 = N0 + wc0;
    d4 cin[4] = {zero, zero, zero, zero};
    if (EPI == EPI_SUB && pm0 < pb.M && pn0 < pb.N) {
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) {
            const int mb = pm0 + (t4 >> 1) * 16, nb = pn0 + (t4 & 1) * 16;
            const int col = nb + r;
            if (mb < pb.M && nb < pb.N && !(narrow && t4 >= 2) && !(pb.zc > 0 && col >= pb.zc)) {
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) cin[t4][s4] = gld(pb.C + (long)(mb + q + 4 * s4) * pb.ldc + col);
            }
        }
        // ... and goes straight into the accumulators (negated; the epilogue writes -acc = C_in - A B^T): the
        // tile then costs no registers during the K loop (32 VGPRs = one wave per SIMD of occupancy)
        c00 = -cin[0]; c01 = -cin[1]; c10 = -cin[2]; c11 = -cin[3];
    }
    *(d4*)(&sA[0][srow * GL_LDS_LD + sch]) = a_ok ? ((SEGS && rn[0]) ? -ra[0] : ra[0]) : zero;
    *(d4*)(&sB[0][srow * GL_LDS_LD + sch]) = b_ok ? rb[0] : zero;
    __syncthreads();
    const int arow0 = (wr0 + r) * GL_LDS_LD + 4 * q, arow1 = arow0 + 16 * GL_LDS_LD;
    const int brow0 = (wc0 + r) * GL_LDS_LD + 4 * q, brow1 = brow0 + 16 * GL_LDS_LD;
    // which of this wave's four 16x16 sub-tiles exist (wave-uniform): padded sub-tiles issue no MFMA
    const bool w_m0 = (M0 + wr0) < pb.M, w_m1 = !narrow && (M0 + wr0 + 16) < pb.M;
    const bool w_n0 = (N0 + wc0) < pb.N, w_n1 = (N0 + wc0 + 16) < pb.N;
    auto compute = [&](int cur) {
        if (w_m0 && w_n0) {
            const d4 a0 = *(const d4*)(&sA[cur][arow0]);
            const d4 b0 = *(const d4*)(&sB[cur][brow0]);
            if (w_m1 && w_n1) {
                const d4 a1 = *(const d4*)(&sA[cur][arow1]);
                const d4 b1 = *(const d4*)(&sB[cur][brow1]);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    c00 = mfma16(a0[j], b0[j], c00);
                    c01 = mfma16(a0[j], b1[j], c01);
                    c10 = mfma16(a1[j], b0[j], c10);
                    c11 = mfma16(a1[j], b1[j], c11);
                }
            } else if (w_m1) {
                const d4 a1 = *(const d4*)(&sA[cur][arow1]);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    c00 = mfma16(a0[j], b0[j], c00);
                    c10 = mfma16(a1[j], b0[j], c10);
                }
            } else if (w_n1) {
                const d4 b1 = *(const d4*)(&sB[cur][brow1]);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    c00 = mfma16(a0[j], b0[j], c00);
                    c01 = mfma16(a0[j], b1[j], c01);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) c00 = mfma16(a0[j], b0[j], c00);
            }
        }
    };
    for (int ks0 = 0; ks0 < nk; ks0 += GL_PF) {
#pragma unroll
        for (int i = 0; i < GL_PF; ++i) {
            const int ks = ks0 + i;                        // LDS buffer i & 1 holds step ks (GL_PF is even)
            if (ks < nk) {
                if (SEGS && segmented) {                   // register set i is free again: refill it GL_PF steps ahead
                    seg_fetch();
                    ra[i] = gld4(ap + it_k); rb[i] = gld4(bp + it_k); rn[i] = it_sgn;
                } else {
                    const int ko = (ks + GL_PF < nk) ? (ks + GL_PF) * 16 : klast;
                    ra[i] = gld4(ap + ko);
                    rb[i] = gld4(bp + ko);
                }
                compute(i & 1);
                *(d4*)(&sA[(i + 1) & 1][srow * GL_LDS_LD + sch]) = a_ok ? ((SEGS && rn[(i + 1) % GL_PF]) ? -ra[(i + 1) % GL_PF] : ra[(i + 1) % GL_PF]) : zero;
                *(d4*)(&sB[(i + 1) & 1][srow * GL_LDS_LD + sch]) = b_ok ? rb[(i + 1) % GL_PF] : zero;
                __syncthreads();
            }
        }
    }
    // ---- epilogue (as k_gemm_nt): accumulator element s of lane (r,q) is C[m + q + 4 s][n + r]
    const int m0 = M0 + wr0, n0 = N0 + wc0;
    if (m0 >= pb.M || n0 >= pb.N) return;
    const bool mv1 = !narrow && (m0 + 16) < pb.M, nv1 = (n0 + 16) < pb.N;
    int bc0 = n0 + r, bc1 = n0 + 16 + r;
    if (pb.idxB) { bc0 = gldi(pb.idxB + n0 + r); bc1 = nv1 ? gldi(pb.idxB + n0 + 16 + r) : -1; }
    // COV / HOSTCOV: everything the epilogue reads per row and per column (gather indices, coordinates, the leaf's observed-row
    // map) is fetched in rounds of independent loads, one 16-row half of the wave's tile at a time (both halves at once kept 64
    // registers alive beside the accumulators and spilled 6 of them at the 128-register budget)
    int xrow4[4];
    int op4[4];
    double xa4[4][DIM], xb2[2][DIM];
    if (EPI == EPI_COV) {
#pragma unroll
        for (int c = 0; c < DIM; ++c) {
            xb2[0][c] = gld(pb.XB + (long)(bc0 < 0 ? 0 : bc0) * DIM + c);
            xb2[1][c] = gld(pb.XB + (long)(bc1 < 0 ? 0 : bc1) * DIM + c);
        }
    }
    auto fetch_rows = [&](int mb) {
        if (EPI == EPI_COV || EPI == EPI_HOSTCOV) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = min(mb + q + 4 * e, pb.M - 1);
                xrow4[e] = pb.idxA ? gldi(pb.idxA + row) : row;
                op4[e] = pb.rowmap ? gldi(pb.rowmap + row) : -1;
            }
            if (EPI == EPI_COV) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int c = 0; c < DIM; ++c) xa4[e][c] = gld(pb.XA + (long)(xrow4[e] < 0 ? 0 : xrow4[e]) * DIM + c);
            }
        }
    };
    auto emit = [&](d4 acc, int mb, int nb, int bcol, int ti) {
        const int col = nb + r;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int row = mb + q + 4 * s;
            double* cp = pb.C + (long)row * pb.ldc + col;
            double v;
            if (EPI == EPI_SET) v = acc[s] + ((row == col && row < pb.diag_one) ? 1.0 : 0.0);
            else if (EPI == EPI_SUB) v = -acc[s];
            else if (EPI == EPI_COV) {
                const int xrow = xrow4[s];
                const double cv = cov_of_dist2<MODE>(kp, pair_dist2<DIM>(xa4[s], xb2[ti & 1], kp.circular)) - acc[s];
                v = (bcol < 0 || xrow < 0) ? 0.0 : cv;
                if (pb.sym_diag && row == col) v = (bcol < 0) ? 1.0 : v + pb.diag_add;
                const int op = op4[s];
                if (op >= 0) gst(pb.C2 + (long)op * pb.ldc + col, v + (op == col ? pb.diag_add : 0.0));
            } else {
                v = (bcol < 0) ? 0.0 : gld(pb.Csrc + (long)row * pb.ldcs + col) - acc[s];
                const int op = op4[s];
                if (op >= 0) gst(pb.C2 + (long)op * pb.ldc + col, v + (op == col ? pb.diag_add : 0.0));
            }
            gst(cp, v);
        }
    };
    fetch_rows(m0);
    emit(c00, m0, n0, bc0, 0);
    if (nv1) emit(c01, m0, n0 + 16, bc1, 1);
    if (mv1) {
        fetch_rows(m0 + 16);
        emit(c10, m0 + 16, n0, bc0, 2);
        if (nv1) emit(c11, m0 + 16, n0 + 16, bc1, 3);
    }
}

// ------------------------------------------------------------------------------------------------
//  Blocked segmented SYRK: C = [I] + sum_seg (+|-) A_seg A_seg^T (lower part) for fronts that do not fit in LDS
//  (the grandparents' fronts of deep 64-wide trees: 464 x 464 from 20 segments, K = 768, at config 5).
//  One workgroup (4 waves) per 96 x 96 block of 16 x 16 sub-tiles on or below the diagonal; the 96 + 96 operand
//  rows of a 16-k step go through a double-buffered LDS stage (one barrier per step, global loads two steps
//  ahead).  Against the 32 x 32 wave tiles of k_gemm_nt (every wave pulls its own 32 + 32 rows through L1 / L2
//  per step: 4 flop per operand byte, the L2 -> CU path is the limit) a block moves 12 flop per byte, and only the
//  sub-tiles on or below the diagonal are computed: off the diagonal a wave owns 3 x 3 sub-tiles (36 MFMAs per
//  step and barrier), on it the 21 lower sub-tiles are dealt 6 / 5 / 5 / 5.
//  Measured (tools/syrk_lab.hip, 4096 problems of config 5's shape): k_gemm_nt 21.4 ms, this kernel 13.0 ms = 54 TFLOP/s executed;
//  its MFMAs alone 10.7 ms (the wave slots of the 6 / 5 / 5 / 5 deal at 2.2 GHz), its loads + LDS writes + barriers alone 5.0 ms.
//  Layout of the stage: role 0 = the rows of block row bi (negated for `neg` segments on their way in), role 1 = the
//  rows of block column bj (diagonal blocks load their rows twice - the second copy is an L1 hit - and run the same
//  code).  Sub-tiles strictly above the diagonal are not written (k_gemm_nt wrote the upper half of its 32 x 32
//  diagonal tiles; nothing reads it: k_panel_chol, k_front, k_assemble and the Schur products work on lower tiles).
// ------------------------------------------------------------------------------------------------
#define SB_T 6
#define SB_ROWS (SB_T * 16)
// VAR: what-if switches of tools/syrk_lab.hip (timing only, wrong results): 1 no barriers, 2 no global loads in the loop, 4 operand
// fragments read from LDS once, 8 no LDS writes in the loop, 16 no MFMAs; 0 in the library
template <int EPI, int VAR = 0>
__global__ __launch_bounds__(256, 2) void k_syrk_blk(const GemmProb* __restrict__ probs, unsigned G, unsigned nprob, unsigned S = 1) {
    static_assert(EPI == EPI_SET, "k_syrk_blk: SET epilogue only");
    __shared__ __attribute__((aligned(16))) double sR[2][2 * SB_ROWS * GL_LDS_LD];
    __shared__ __attribute__((aligned(16))) SbStep sTab[SB_MAXST];
    __shared__ int sK[SB_MAXST];
    __shared__ int sNk;
    unsigned prob_i, wg_i;
    if (!xcd_problem_tile(G, nprob, prob_i, wg_i, S)) return;
    const GemmProb* __restrict__ pp = probs + __builtin_amdgcn_readfirstlane(prob_i);
    const int M = pp->M, Mt = M >> 4;
    const int nbk = (Mt + SB_T - 1) / SB_T;
    const int tile = __builtin_amdgcn_readfirstlane((int)wg_i);
    if (tile >= nbk * (nbk + 1) / 2) return;
    int bi = 0;
    while ((bi + 1) * (bi + 2) / 2 <= tile) ++bi;
    const int bj = tile - bi * (bi + 1) / 2;
    const bool diag = bi == bj;
    const int nri = min(SB_T, Mt - SB_T * bi);              // sub-tile rows of this block row that exist
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    // ---- staging role: thread -> three (role, row, 32-byte chunk) slots of the 192-row stage
    const int srow = threadIdx.x >> 2, sch = (threadIdx.x & 3) << 2;
    long grow[3];
    int lofs[3];
    bool sneg[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int f = srow + 64 * i, role = f >= SB_ROWS ? 1 : 0, row = f - role * SB_ROWS;
        grow[i] = min((role ? bj : bi) * SB_ROWS + row, M - 1);          // rows past the end re-read the last one: their sub-tiles are never written
        lofs[i] = (role * SB_ROWS + row) * GL_LDS_LD + sch;
        sneg[i] = role == 0;
    }
    // ---- the K dimension as a flat table of 16-column steps in LDS: (operand base, leading dimension, column offset | sign).  Walking
    // the segment descriptors inside the loop (as k_gemm_nt does) reads them with VECTOR loads - the compiler cannot prove the walk
    // uniform - and every segment boundary then waits with s_waitcnt vmcnt(0) for the prefetches just issued as well.
    const GemmSeg* __restrict__ segs = pp->segs;
    const int nseg = pp->nseg;
    for (int sg = threadIdx.x; sg < nseg; sg += 256) sK[sg] = segs[sg].K >> 4;
    __syncthreads();
    for (int sg = threadIdx.x; sg < nseg; sg += 256) {
        int start = 0;
        for (int j = 0; j < sg; ++j) start += sK[j];
        const int cnt = sK[sg];
        const double* A = segs[sg].A;
        const int lda = (int)segs[sg].lda, sgn = segs[sg].neg ? (int)0x80000000 : 0;
        for (int t = 0; t < cnt; ++t) sTab[start + t] = SbStep{A, lda, (16 * t) | sgn};
        if (sg == nseg - 1) sNk = start + cnt;
    }
    __syncthreads();
    const int nk = __builtin_amdgcn_readfirstlane(nseg > 0 ? sNk : 0);
    int growi[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) growi[i] = (int)grow[i];
    int it_s = 0;                                            // next step to load
    auto issue = [&](d4* dst, int& sgn) {                    // loads of step it_s (past the end: the last step again, unconditional loads)
        const SbStep e = sTab[min(it_s, nk - 1)];
        const double* base = e.A + (e.k0s & 0x7fffffff) + sch;
        sgn = e.k0s & (int)0x80000000;
        if (!(VAR & 2)) {
            dst[0] = gld4(base + (long)growi[0] * e.lda); dst[1] = gld4(base + (long)growi[1] * e.lda); dst[2] = gld4(base + (long)growi[2] * e.lda);
        }
        ++it_s;
    };
    auto flip = [&](const d4 v, int sgn) -> d4 {              // v with the sign bit of every element XORed with sgn (0 or 0x80000000)
        d4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = __hiloint2double(__double2hiint(v[e]) ^ sgn, __double2loint(v[e]));
        return o;
    };
    const d4 zero = {0, 0, 0, 0};
    d4 acc[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) acc[i] = zero;
    // roles of the wave: off the diagonal sub-tile rows 3 wr .. 3 wr + 2 x columns 3 wc .. 3 wc + 2; on it row rowA with
    // columns 0 .. nA - 1 (accumulators 0 ..) and row rowB with columns 0 .. nB - 1 (accumulators 6, 7)
    const int wr = wave >> 1, wc = wave & 1;
    const int nvr = diag ? 0 : max(0, min(3, nri - 3 * wr));
    int rowA = 5 - wave, nA = 6 - wave, rowB = wave == 2 ? 0 : 1, nB = wave < 2 ? 0 : wave - 1;
    if (!diag || rowA >= nri) nA = 0;
    if (!diag || rowB >= nri) nB = 0;
    // the K loop, once for blocks on the diagonal and once for the others (two loops, not one loop with both bodies: the register
    // allocation of the union spilled)
    auto kloop = [&](auto on_diag) {
        constexpr bool DG = decltype(on_diag)::value;
        d4 ra[GL_PF][3];
        int rn[GL_PF];
#pragma unroll
        for (int i = 0; i < GL_PF; ++i) {
#pragma unroll
            for (int j = 0; j < 3; ++j) ra[i][j] = zero;
            issue(ra[i], rn[i]);
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) *(d4*)(&sR[0][lofs[j]]) = flip(ra[0][j], sneg[j] ? rn[0] : 0);
        __syncthreads();
        const int fo = r * GL_LDS_LD + 4 * q;
        for (int ks0 = 0; ks0 < nk; ks0 += GL_PF) {
#pragma unroll
            for (int i = 0; i < GL_PF; ++i) {
                const int ks = ks0 + i;                        // LDS buffer i & 1 holds step ks (GL_PF is even)
                if (ks < nk) {
                    issue(ra[i], rn[i]);
                    const double* sA = &sR[(VAR & 4) ? 0 : (i & 1)][(VAR & 4) ? 0 : fo];
                    const double* sB = sA + SB_ROWS * GL_LDS_LD;
                    if (!DG) {
                        if (nvr > 0) {
                            d4 a[3], b[3];
#pragma unroll
                            for (int x = 0; x < 3; ++x) {
                                a[x] = *(const d4*)(sA + (3 * wr + x) * 16 * GL_LDS_LD);
                                b[x] = *(const d4*)(sB + (3 * wc + x) * 16 * GL_LDS_LD);
                            }
#pragma unroll
                            for (int j = 0; j < 4; ++j)
#pragma unroll
                                for (int x = 0; x < 3; ++x)
                                    if (x < nvr) {
#pragma unroll
                                        for (int y = 0; y < 3; ++y) acc[3 * x + y] = (VAR & 16) ? acc[3 * x + y] : mfma16(a[x][j], b[y][j], acc[3 * x + y]);
                                    }
                        }
                    } else if ((nA | nB) != 0) {
                        const d4 aA = *(const d4*)(sA + rowA * 16 * GL_LDS_LD);
                        const d4 aB = *(const d4*)(sA + rowB * 16 * GL_LDS_LD);
                        // columns in two halves of three (three B fragments live at a time; three independent accumulator chains)
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            if (h == 0 || 3 < nA) {
                                d4 b[3];
#pragma unroll
                                for (int c = 0; c < 3; ++c) b[c] = *(const d4*)(sB + (3 * h + c) * 16 * GL_LDS_LD);
#pragma unroll
                                for (int j = 0; j < 4; ++j) {
#pragma unroll
                                    for (int c = 0; c < 3; ++c)
                                        if (3 * h + c < nA) acc[3 * h + c] = (VAR & 16) ? acc[3 * h + c] : mfma16(aA[j], b[c][j], acc[3 * h + c]);
                                    if (h == 0) {
#pragma unroll
                                        for (int c = 0; c < 2; ++c)
                                            if (c < nB) acc[6 + c] = (VAR & 16) ? acc[6 + c] : mfma16(aB[j], b[c][j], acc[6 + c]);
                                    }
                                }
                            }
                        }
                    }
                    if (!(VAR & 8))
#pragma unroll
                    for (int j = 0; j < 3; ++j)
                        *(d4*)(&sR[(i + 1) & 1][lofs[j]]) = flip(ra[(i + 1) % GL_PF][j], sneg[j] ? rn[(i + 1) % GL_PF] : 0);
                    if (!(VAR & 1)) __syncthreads();
                }
            }
        }
    };
    if (nk > 0) {
        if (diag) kloop(std::true_type{});
        else kloop(std::false_type{});
    }
    // ---- epilogue: accumulator element s of lane (r,q) is C[row0 + q + 4 s][col0 + r]
    double* const C = pp->C;
    const long ldc = pp->ldc;
    const int done = pp->diag_one;
    auto emit = [&](const d4 v, int ti, int tj) {
        const int row0 = (SB_T * bi + ti) * 16, col = (SB_T * bj + tj) * 16 + r;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const int row = row0 + q + 4 * s4;
            gst(C + (long)row * ldc + col, v[s4] + ((row == col && row < done) ? 1.0 : 0.0));
        }
    };
    if (!diag) {
#pragma unroll
        for (int x = 0; x < 3; ++x)
            if (x < nvr) {
#pragma unroll
                for (int y = 0; y < 3; ++y) emit(acc[3 * x + y], 3 * wr + x, 3 * wc + y);
            }
    } else {
#pragma unroll
        for (int c = 0; c < 6; ++c)
            if (c < nA) emit(acc[c], rowA, c);
#pragma unroll
        for (int c = 0; c < 2; ++c)
            if (c < nB) emit(acc[6 + c], rowB, c);
    }
}

// ------------------------------------------------------------------------------------------------
//  k_syrk_blk with the stage filled by LDS DMA (global_load_lds_dwordx4): same blocks, same deal of sub-tiles, same results.
//  No staging registers (three workgroups per CU instead of two), no ds_write.  The DMA writes lane-linear - lane l of a wave
//  fills 16 bytes at position l of a 1 KB piece = 8 rows of 16 doubles - so the stage cannot be padded; rows of 128 bytes all
//  start in bank 0, and the 16-byte chunks of a row are stored XOR-swizzled instead (chunk c of row p at position c ^ (p & 7),
//  done on the SOURCE address of the DMA): a fragment (32 bytes of a row) is read as two 16-byte halves, the eight rows a group of
//  lanes reads fall on eight different positions.  Diagonal blocks load their rows once (both operand roles read them).  The sign
//  of `neg` segments is applied to the A fragments as they are read.  tools/syrk_lab.hip: 12.2-12.6 ms against 12.9-13.0 for the
//  register-staged k_syrk_blk on the same box (requesting the next stage AFTER the fragment reads instead of before them: 34 ms).
// ------------------------------------------------------------------------------------------------
#define SD_MAXST 128
#ifndef SD_WPE
#define SD_WPE 3
#endif
template <int EPI>
__global__ __launch_bounds__(256, SD_WPE) void k_syrk_dma(const GemmProb* __restrict__ probs, unsigned G, unsigned nprob, unsigned S = 1) {
    static_assert(EPI == EPI_SET, "k_syrk_dma: SET epilogue only");
    __shared__ __attribute__((aligned(16))) double sR[2][2 * SB_ROWS * 16];
    __shared__ __attribute__((aligned(16))) SbStep sTab[SD_MAXST];
    __shared__ int sNk;
    int* const sK = (int*)&sR[0][0];                        // per-segment step counts while the table is built
    unsigned prob_i, wg_i;
    if (!xcd_problem_tile(G, nprob, prob_i, wg_i, S)) return;
    const GemmProb* __restrict__ pp = probs + __builtin_amdgcn_readfirstlane(prob_i);
    const int M = pp->M, Mt = M >> 4;
    const int nbk = (Mt + SB_T - 1) / SB_T;
    const int tile = __builtin_amdgcn_readfirstlane((int)wg_i);
    if (tile >= nbk * (nbk + 1) / 2) return;
    int bi = 0;
    while ((bi + 1) * (bi + 2) / 2 <= tile) ++bi;
    const int bj = tile - bi * (bi + 1) / 2;
    const bool diag = bi == bj;
    const int nri = min(SB_T, Mt - SB_T * bi);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    // ---- staging role: wave w moves the six 1 KB pieces 6 w .. 6 w + 5 of the 24 of a stage (piece = 8 rows; pieces 0-11 are the
    // rows of block row bi, 12-23 those of block column bj); lane l: row l >> 3 of the piece, position l & 7, chunk (l & 7) ^ (row & 7)
    int growi[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int srow = (6 * wave + i) * 8 + (lane >> 3);                       // row of the stage
        const int role = srow >= SB_ROWS ? 1 : 0;
        growi[i] = min((role ? bj : bi) * SB_ROWS + srow - role * SB_ROWS, M - 1);
    }
    const int schunk = ((lane & 7) ^ ((lane >> 3) & 7)) * 2;                     // doubles: the logical chunk this lane fetches (rows of a piece start at a multiple of 8)
    const GemmSeg* __restrict__ segs = pp->segs;
    const int nseg = pp->nseg;
    for (int sg = threadIdx.x; sg < nseg; sg += 256) sK[sg] = segs[sg].K >> 4;
    __syncthreads();
    for (int sg = threadIdx.x; sg < nseg; sg += 256) {
        int start = 0;
        for (int j = 0; j < sg; ++j) start += sK[j];
        const int cnt = sK[sg];
        const double* A = segs[sg].A;
        const int lda = (int)segs[sg].lda, sgn = segs[sg].neg ? (int)0x80000000 : 0;
        for (int t = 0; t < cnt; ++t) sTab[start + t] = SbStep{A, lda, (16 * t) | sgn};
        if (sg == nseg - 1) sNk = start + cnt;
    }
    __syncthreads();                                         // (also: sK is dead before the first stage lands)
    const int nk = __builtin_amdgcn_readfirstlane(nseg > 0 ? sNk : 0);
    const bool loads_b = !diag;                              // diagonal blocks: both roles read the rows of role 0
    auto request = [&](int step, int buf) {                  // DMA of step `step` into buffer `buf`
        const SbStep e = sTab[step];
        const double* base = e.A + (e.k0s & 0x7fffffff) + schunk;
        char* const db = (char*)&sR[buf][0] + (long)(6 * wave) * 1024;
#pragma unroll
        for (int i = 0; i < 6; ++i)
            if (loads_b || 6 * wave + i < 12) gld_lds16(base + (long)growi[i] * e.lda, db + i * 1024);
    };
    const d4 zero = {0, 0, 0, 0};
    d4 acc[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) acc[i] = zero;
    const int wr = wave >> 1, wc = wave & 1;
    const int nvr = diag ? 0 : max(0, min(3, nri - 3 * wr));
    int rowA = 5 - wave, nA = 6 - wave, rowB = wave == 2 ? 0 : 1, nB = wave < 2 ? 0 : wave - 1;
    if (!diag || rowA >= nri) nA = 0;
    if (!diag || rowB >= nri) nB = 0;
    // fragment of lane (r, q) in a 16-row sub-tile: row r, doubles 4 q .. 4 q + 3 = chunks 2 q, 2 q + 1 at their swizzled positions
    const int fo0 = r * 16 + (((2 * q) ^ (r & 7)) << 1), fo1 = r * 16 + (((2 * q + 1) ^ (r & 7)) << 1);
    auto frag = [&](const double* sub) -> d4 {
        const d2 lo = *(const d2*)(sub + fo0), hi = *(const d2*)(sub + fo1);
        return d4{lo[0], lo[1], hi[0], hi[1]};
    };
    auto flip = [&](const d4 v, int sgn) -> d4 {
        d4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = __hiloint2double(__double2hiint(v[e]) ^ sgn, __double2loint(v[e]));
        return o;
    };
    auto kloop = [&](auto on_diag) {
        constexpr bool DG = decltype(on_diag)::value;
        request(0, 0);
        mra_wait_vm0();
        __syncthreads();
        for (int ks = 0; ks < nk; ++ks) {
            const int cur = ks & 1;
            if (ks + 1 < nk) request(ks + 1, cur ^ 1);       // (that buffer was read in step ks - 1: everybody has passed a barrier since)
            const int sgn = sTab[ks].k0s & (int)0x80000000;
            const double* sA = &sR[cur][0];
            const double* sB = DG ? sA : sA + SB_ROWS * 16;
            if (!DG) {
                if (nvr > 0) {
                    d4 a[3], b[3];
#pragma unroll
                    for (int x = 0; x < 3; ++x) {
                        a[x] = flip(frag(sA + (3 * wr + x) * 256), sgn);
                        b[x] = frag(sB + (3 * wc + x) * 256);
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int x = 0; x < 3; ++x)
                            if (x < nvr) {
#pragma unroll
                                for (int y = 0; y < 3; ++y) acc[3 * x + y] = mfma16(a[x][j], b[y][j], acc[3 * x + y]);
                            }
                }
            } else if ((nA | nB) != 0) {
                const d4 aA = flip(frag(sA + rowA * 256), sgn);
                const d4 aB = flip(frag(sA + rowB * 256), sgn);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    if (h == 0 || 3 < nA) {
                        d4 b[3];
#pragma unroll
                        for (int c = 0; c < 3; ++c) b[c] = frag(sB + (3 * h + c) * 256);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
#pragma unroll
                            for (int c = 0; c < 3; ++c)
                                if (3 * h + c < nA) acc[3 * h + c] = mfma16(aA[j], b[c][j], acc[3 * h + c]);
                            if (h == 0) {
#pragma unroll
                                for (int c = 0; c < 2; ++c)
                                    if (c < nB) acc[6 + c] = mfma16(aB[j], b[c][j], acc[6 + c]);
                            }
                        }
                    }
                }
            }
            mra_wait_vm0();
            __syncthreads();
        }
    };
    if (nk > 0) {
        if (diag) kloop(std::true_type{});
        else kloop(std::false_type{});
    }
    double* const C = pp->C;
    const long ldc = pp->ldc;
    const int done = pp->diag_one;
    auto emit = [&](const d4 v, int ti, int tj) {
        const int row0 = (SB_T * bi + ti) * 16, col = (SB_T * bj + tj) * 16 + r;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const int row = row0 + q + 4 * s4;
            gst(C + (long)row * ldc + col, v[s4] + ((row == col && row < done) ? 1.0 : 0.0));
        }
    };
    if (!diag) {
#pragma unroll
        for (int x = 0; x < 3; ++x)
            if (x < nvr) {
#pragma unroll
                for (int y = 0; y < 3; ++y) emit(acc[3 * x + y], 3 * wr + x, 3 * wc + y);
            }
    } else {
#pragma unroll
        for (int c = 0; c < 6; ++c)
            if (c < nA) emit(acc[c], rowA, c);
#pragma unroll
        for (int c = 0; c < 2; ++c)
            if (c < nB) emit(acc[6 + c], rowB, c);
    }
}

// ------------------------------------------------------------------------------------------------
//  Leaf-resident batched product: one workgroup (8 waves) per problem, C (=|-=) f(A B^T) with
//    * every wave owning two 16-row tiles of A (rows w and w + 8 of each group of 16 row tiles): its A fragments
//      come straight from global memory (32 contiguous bytes per lane, nobody else needs them),
//    * B streamed through a double-buffered LDS stage, 16 k at a time, shared by all 8 waves,
//    * up to LG_CT column tiles per pass held in registers (2 x 7 accumulators): 56 MFMAs per wave and barrier
//      instead of the 16 of the 64x64-tile kernel above, whose operands also cross L2 four times more often.
//  Accumulators come out in vec layout (lane (r,q): C[row r][4q .. 4q+3]): the epilogues move 32 bytes per lane.
//  Same GemmProb, same SUB / COV epilogue semantics as k_gemm_nt_lds (used for the two big leaf products of the
//  fused path: V[S,o] = kernel - W_S W_o^T and W[S,anc|y] -= Tt Ut^T; MRANode.py:73-80, 489-495 in factorised form).
// ------------------------------------------------------------------------------------------------
#define LG_LD 20
// RT row tiles per wave, CT column tiles per pass: <2, 7> for the leaf residual (N = observations, <= 7 tiles in one pass at
// C3), <1, 13> for the leaf update (N = ancestors + y = 13 tiles: one pass, so that Tt is read from HBM once)
// NTHR threads per workgroup, WPE waves per SIMD the register budget is cut for (the second __launch_bounds__ argument is
// waves per SIMD on HIP): <2, 7, 256, 2> holds 2 x 7 accumulators per wave at two waves per SIMD; <1, 7, 512, 4> one row
// tile per wave at 128 registers, two 8-wave workgroups = four waves per SIMD
// SOLVE (COV, N <= CT * 16 <= 64: one pass): the prior of one level of the level-by-level path in one launch - residual product,
// kernel evaluation and the row solve W^m = R L^-T (MRANode.py:83-91) on the accumulators in vec layout, the way the cascades do it:
//     x_jb = (r_jb - sum_{kb < jb} x_kb L[jb][kb]^T) inv_jb^T ,   var -= |x|^2 .
// The factor's strictly-lower tiles (jb (jb-1)/2 + kb) and inverted diagonal blocks (6 + jb) sit in LDS as 16 x 16 tiles.  Against
// k_gemm_nt_lds<COV> + k_trsm_rows2 the residual never visits HBM (one write of 64 columns and one read less per level).
__device__ __forceinline__ int pp_N(const GemmProb* __restrict__ probs) { return probs[blockIdx.x].N; }
template <int EPI, int DIM, int MODE, int RT, int CT, int NTHR, int WPE, int SOLVE = 0>
__global__ __launch_bounds__(NTHR, WPE) void k_leaf_gemm(const GemmProb* __restrict__ probs, KernelParams kp) {
    static_assert(!SOLVE || (EPI == EPI_COV && CT <= 4), "k_leaf_gemm: the row solve rides on the COV epilogue of at most four column tiles");
    __shared__ __attribute__((aligned(16))) double sB[2][CT * 16 * LG_LD];
    __shared__ double sXB[CT * 16 * DIM];
    __shared__ int sIB[CT * 16];
    __shared__ __attribute__((aligned(16))) double sL[SOLVE ? 10 * 256 : 2];
    if (SOLVE) {
        // factor tiles of this problem's node: staged once (the barrier in front of the first K loop covers them)
        const int nct = (pp_N(probs) + 15) >> 4;
        const double* L = probs[blockIdx.x].solveL;
        const double* I = probs[blockIdx.x].solveI;
        const long ldl = probs[blockIdx.x].N;
        for (int e = threadIdx.x; e < 10 * 128; e += NTHR) {
            const int tile = e >> 7, rr = (e >> 3) & 15, c2 = (e & 7) << 1;
            d2 v = d2{0.0, 0.0};
            if (tile < 6) {
                const int jb = tile < 1 ? 1 : (tile < 3 ? 2 : 3), kb = tile - jb * (jb - 1) / 2;
                if (jb < nct) v = gld2(L + (long)(jb * 16 + rr) * ldl + kb * 16 + c2);
            } else if (tile - 6 < nct) v = gld2(I + (long)(tile - 6) * 256 + rr * 16 + c2);
            *(d2*)(sL + tile * 256 + rr * 16 + c2) = v;
        }
    }
    const GemmProb* __restrict__ pp = probs + blockIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwave = blockDim.x >> 6;                           // 4: two workgroups (two leaves) share a CU, one's epilogue
    const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;   // and prologue run beside the other's K loop
    const int prow = pi16(r);
    const int M = pp->M, N = pp->N, K = pp->K;
    const int ntm = M >> 4, ntn = (N + 15) >> 4, nk = K >> 4;
    const double* const A = pp->A; const long lda = pp->lda;
    const double* const B = pp->B; const long ldb = pp->ldb;
    const int* const idxB = pp->idxB;
    // SOLVE: the rows of A (and of XA, C, var) may be a GATHERED list (idxA: absolute row numbers, -1: padding; A, XA, C and var are
    // then row 0's) - likelihood-only passes of the level-by-level path walk the rows a likelihood needs (observed rows and knots)
    const int* const idxA = SOLVE ? pp->idxA : nullptr;
    double* const C = pp->C; const long ldc = pp->ldc;
    const int zc = pp->zc;
    const d4 zero = {0, 0, 0, 0};
    constexpr int NST = (CT * 64 + NTHR - 1) / NTHR;             // 32-byte staging chunks per thread
    for (int rg = 0; rg < ntm; rg += RT * nwave) {
        bool vr[RT], rok[RT];
        long rowh[RT];
        const double* ap[RT];
        int qa = q;
        asm volatile("" : "+v"(qa));         // opaque: the lane part of the A address is rebuilt per row group instead of being hoisted and spilled
#pragma unroll
        for (int h = 0; h < RT; ++h) {
            const int ih = rg + wave + h * nwave;
            vr[h] = ih < ntm;
            rowh[h] = (long)(vr[h] ? ih : 0) * 16 + r;
            rok[h] = true;
            if (SOLVE && idxA) { const int gi = gldi(idxA + rowh[h]); rok[h] = gi >= 0; rowh[h] = gi < 0 ? 0 : gi; }
            ap[h] = A + rowh[h] * lda + 4 * qa;
        }
        for (int j0 = 0; j0 < ntn; j0 += CT) {
            const int nc = min(CT, ntn - j0);                    // column tiles of this pass
            const int nchunk = nc * 64;                          // 32-byte chunks (row, 4 columns) of the staged nc*16 x 16 slab
            bool on[NST], ok[NST];
            int so[NST];
            const double* bp[NST];
#pragma unroll
            for (int g = 0; g < NST; ++g) {
                const int e = threadIdx.x + g * NTHR;
                on[g] = e < nchunk;
                const int sr = e >> 2, sc = (e & 3) << 2;
                so[g] = sr * LG_LD + sc;
                long br = j0 * 16 + sr;
                ok[g] = on[g] && br < N;
                if (ok[g] && idxB) { const int ib = gldi(idxB + br); ok[g] = ib >= 0; br = ib; }
                bp[g] = B + (ok[g] ? br : 0) * ldb + sc;
            }
            __syncthreads();                                     // previous pass is done with sB / sXB / sIB
            if (EPI == EPI_COV || EPI == EPI_HOSTCOV) {
                for (int t = threadIdx.x; t < nc * 16; t += blockDim.x) {
                    const int col = j0 * 16 + t;
                    const int ib = col < N ? (idxB ? gldi(idxB + col) : col) : -1;
                    sIB[t] = ib;
                    if (EPI == EPI_COV) {
#pragma unroll
                        for (int c = 0; c < DIM; ++c) sXB[t * DIM + c] = gld(pp->XB + (long)(ib < 0 ? 0 : ib) * DIM + c);
                    }
                }
            }
            d4 acc[RT][CT];
#pragma unroll
            for (int h = 0; h < RT; ++h)
#pragma unroll
                for (int j = 0; j < CT; ++j) acc[h][j] = zero;
            if (EPI == EPI_SUB) {
                // the C tiles go straight into the accumulators (negated; the epilogue writes -acc = C_in - A B^T): their
                // loads are in flight beside the first chunk and cost no registers
#pragma unroll
                for (int j = 0; j < CT; ++j) {
                    if (j < nc) {
                        const int col = (j0 + j) * 16 + 4 * q;
                        if (!(zc > 0 && col >= zc)) {
#pragma unroll
                            for (int h = 0; h < RT; ++h) if (vr[h]) acc[h][j] = -gld4(C + rowh[h] * ldc + col);
                        }
                    }
                }
            }
            d4 sb[NST], fa[RT];
#pragma unroll
            for (int g = 0; g < NST; ++g) sb[g] = nk > 0 ? gld4(bp[g]) : zero;
#pragma unroll
            for (int h = 0; h < RT; ++h) fa[h] = nk > 0 ? gld4(ap[h]) : zero;
#pragma unroll
            for (int g = 0; g < NST; ++g) if (on[g]) *(d4*)(&sB[0][so[g]]) = ok[g] ? sb[g] : zero;
            __syncthreads();
            for (int ks = 0; ks < nk; ++ks) {
                const int cur = ks & 1;
                const int kn = (ks + 1 < nk ? ks + 1 : ks) * 16;
                d4 a[RT];
#pragma unroll
                for (int h = 0; h < RT; ++h) a[h] = fa[h];
#pragma unroll
                for (int g = 0; g < NST; ++g) sb[g] = gld4(bp[g] + kn);       // next chunk in flight during the products
#pragma unroll
                for (int h = 0; h < RT; ++h) fa[h] = gld4(ap[h] + kn);
#pragma unroll
                for (int j = 0; j < CT; ++j) {
                    if (j < nc) {
                        const d4 b = *(const d4*)(&sB[cur][(j * 16 + prow) * LG_LD + 4 * q]);
#pragma unroll
                        for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                            for (int h = 0; h < RT; ++h) acc[h][j] = mfma16(b[s4], a[h][s4], acc[h][j]);
                    }
                }
                if (ks + 1 < nk) {
#pragma unroll
                    for (int g = 0; g < NST; ++g) if (on[g]) *(d4*)(&sB[cur ^ 1][so[g]]) = ok[g] ? sb[g] : zero;
                    __syncthreads();
                }
            }
            // ---- epilogue: lane (r,q) of tile (i,j) holds C[i*16 + r][j*16 + 4q .. 4q+3]
            int qe = q;
            asm volatile("" : "+v"(qe));     // opaque, as qa above: keeps the epilogue's lane offsets out of the K loop's register budget
#pragma unroll
            for (int h = 0; h < RT; ++h) {
                if (!vr[h]) continue;
                const long row = rowh[h];
                double xa[DIM];
                int op = -1;
                if (EPI == EPI_COV) {
#pragma unroll
                    for (int c = 0; c < DIM; ++c) xa[c] = gld(pp->XA + row * DIM + c);
                }
                if (EPI == EPI_COV || EPI == EPI_HOSTCOV) op = pp->rowmap ? gldi(pp->rowmap + row) : -1;
                if (SOLVE) {
                    d4 x[CT];
                    double ssq = 0.0;
                    const int fo = prow * 16 + 4 * qe;
#pragma unroll
                    for (int jb = 0; jb < CT; ++jb) {
                        if (jb < nc) {
                            d4 v;
#pragma unroll
                            for (int s4 = 0; s4 < 4; ++s4) {
                                const int lc = jb * 16 + 4 * qe + s4;
                                const double cv = cov_of_dist2<MODE>(kp, pair_dist2<DIM>(xa, &sXB[lc * DIM], kp.circular));
                                v[s4] = sIB[lc] < 0 ? 0.0 : cv - acc[h][jb][s4];
                            }
                            d4 upd = zero;
#pragma unroll
                            for (int kb = 0; kb < CT; ++kb) {
                                if (kb < jb) {
                                    const d4 a = *(const d4*)(sL + (jb * (jb - 1) / 2 + kb) * 256 + fo);
#pragma unroll
                                    for (int s4 = 0; s4 < 4; ++s4) upd = mfma16(a[s4], x[kb][s4], upd);
                                }
                            }
                            v -= upd;
                            const d4 ia = *(const d4*)(sL + (6 + jb) * 256 + fo);
                            d4 xx = zero;
#pragma unroll
                            for (int s4 = 0; s4 < 4; ++s4) xx = mfma16(ia[s4], v[s4], xx);
                            x[jb] = xx;
                            ssq += xx[0] * xx[0] + xx[1] * xx[1] + xx[2] * xx[2] + xx[3] * xx[3];
                            if (rok[h]) gst4(C + row * ldc + jb * 16 + 4 * qe, xx);
                        }
                    }
                    ssq += __shfl_xor(ssq, 16, 64);
                    ssq += __shfl_xor(ssq, 32, 64);
                    if (qe == 0 && pp->var && rok[h]) { double* vp = pp->var + row; gst(vp, gld(vp) - ssq); }
                    continue;
                }
#pragma unroll
                for (int j = 0; j < CT; ++j) {
                    if (j < nc) {
                        const d4 ac = acc[h][j];
                        const int col = (j0 + j) * 16 + 4 * qe;
                        double* cp = C + row * ldc + col;
                        d4 v;
                        if (EPI == EPI_SET) v = ac;
                        else if (EPI == EPI_SUB) v = -ac;
                        else {
#pragma unroll
                            for (int s4 = 0; s4 < 4; ++s4) {
                                const int lc = j * 16 + 4 * qe + s4;
                                const int ib = sIB[lc];
                                double cv;
                                if (EPI == EPI_COV) cv = cov_of_dist2<MODE>(kp, pair_dist2<DIM>(xa, &sXB[lc * DIM], kp.circular));
                                else cv = gld(pp->Csrc + row * pp->ldcs + col + s4);
                                v[s4] = ib < 0 ? 0.0 : cv - ac[s4];
                            }
                            if (op >= 0) {
                                d4 v2 = v;
#pragma unroll
                                for (int s4 = 0; s4 < 4; ++s4) v2[s4] += (op == col + s4) ? pp->diag_add : 0.0;
                                gst4(pp->C2 + (long)op * ldc + col, v2);
                            }
                        }
                        gst4(cp, v);
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
//  16x16 diagonal block: Cholesky + inverse of the factor, one row per lane (lanes 0..15)
// ------------------------------------------------------------------------------------------------
// a[k] = T[lane][k] (k <= lane valid, 0 above the diagonal).  On return a[k] = L[lane][k] (0 above the
// diagonal), mr[k] = (L^{-1})[lane][k] (row `lane` of the inverse, 0 above the diagonal).  Returns
// sum(log diag L) and sets bad when a pivot is not positive.  All 64 lanes call it with
// lane = (lane id & 15): the four rows of 16 lanes do the same work.
//
// This serial 16x16 step is the latency floor of every small factorisation on the path (knot pass,
// fronts, the leaves' C blocks: 3.3 us per call), so it is written for a short dependent chain:
// right-looking with one row per lane; per column every lane fetches the un-normalised entries
// c_k = A[k][j] from their owner lanes with DPP row broadcasts (VALU moves: no LDS round trip, no
// v_readlane scalar hazards) and applies A[i][k] -= (c_i / d_j) c_k with its own Newton reciprocal of
// the pivot d_j = c_j.  Square roots (v_rsq_f64 + Goldschmidt, no division) are off the critical
// path: L[i][j] = c_i * rsqrt(d_j).  The inverse comes from applying the same row operations to the
// identity (row j of the unit-lower inverse is final at step j), scaled by rsqrt(d_i) at the end.
// What is left on the chain is sixteen dependent reciprocals.
__device__ __forceinline__ void lds_wave_sync() {
    // LDS operations of one wave complete in program order; this only stops the compiler from moving
    // or forwarding memory accesses across the exchange
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ double rcp_pos(double d) {            // 1/d, d > 0 normal: v_rcp_f64 + 2 Newton steps
    double y = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-d, y, 1.0);
    return __builtin_fma(y, e, y);
}
__device__ __forceinline__ double rsqrt_pos(double d) {          // 1/sqrt(d), d > 0 normal
    const double y = __builtin_amdgcn_rsq(d);
    double g = d * y, h = 0.5 * y;
    double e = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, e, g);
    h = __builtin_fma(h, e, h);
    e = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, e, g);
    h = __builtin_fma(h, e, h);
    e = __builtin_fma(-h, g, 0.5);
    h = __builtin_fma(h, e, h);
    return h + h;
}
// value of lane k of the caller's row of 16 lanes (DPP row_newbcast: one VALU move per 32 bits, no LDS round trip and
// no scalar-register hazard); k is a constant after unrolling, the switch folds to one instruction
__device__ __forceinline__ int row_bcast_i(int v, int k) {
    switch (k) {
        case 0: return __builtin_amdgcn_update_dpp(0, v, 0x150 + 0, 0xf, 0xf, false);
        case 1: return __builtin_amdgcn_update_dpp(0, v, 0x150 + 1, 0xf, 0xf, false);
        case 2: return __builtin_amdgcn_update_dpp(0, v, 0x150 + 2, 0xf, 0xf, false);
        case 3: return __builtin_amdgcn_update_dpp(0, v, 0x150 + 3, 0xf, 0xf, false);
        case 4: return __builtin_amdgcn_update_dpp(0, v, 0x150 + 4, 0xf, 0xf, false);
        case 5: return __builtin_amdgcn_update_dpp(0, v, 0x150 + 5, 0xf, 0xf, false);
        case 6: return __builtin_amdgcn_update_dpp(0, v, 0x150 + 6, 0xf, 0xf, false);
        case 7: return __builtin_amdgcn_update_dpp(0, v, 0x150 + 7, 0xf, 0xf, false);
        case 8: return __builtin_amdgcn_update_dpp(0, v, 0x150 + 8, 0xf, 0xf, false);
        case 9: return __builtin_amdgcn_update_dpp(0, v, 0x150 + 9, 0xf, 0xf, false);
        case 10: return __builtin_amdgcn_update_dpp(0, v, 0x150 + 10, 0xf, 0xf, false);
        case 11: return __builtin_amdgcn_update_dpp(0, v, 0x150 + 11, 0xf, 0xf, false);
        case 12: return __builtin_amdgcn_update_dpp(0, v, 0x150 + 12, 0xf, 0xf, false);
        case 13: return __builtin_amdgcn_update_dpp(0, v, 0x150 + 13, 0xf, 0xf, false);
        case 14: return __builtin_amdgcn_update_dpp(0, v, 0x150 + 14, 0xf, 0xf, false);
        case 15: return __builtin_amdgcn_update_dpp(0, v, 0x150 + 15, 0xf, 0xf, false);
    }
    return v;
}
__device__ __forceinline__ double row_bcast(double x, int k) {
    return __hiloint2double(row_bcast_i(__double2hiint(x), k), row_bcast_i(__double2loint(x), k));
}
__device__ __forceinline__ double chol16_inv(double a[16], double mr[16], int lane, bool& bad, double* __restrict__ /*sc: unused*/) {
    // opaque copy of the lane number: keeps the per-lane masks / unit vectors below from being hoisted
    // out of the caller's tile loop (dozens of live registers across the whole factorisation)
    asm volatile("" : "+v"(lane));
    double dsel = 1.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) mr[k] = (k == lane) ? 1.0 : 0.0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        // column j of the trailing matrix (entry k comes from lane k) and row j of the unit-lower inverse (from lane j)
        double c[16], mj[16];
#pragma unroll
        for (int k = j; k < 16; ++k) c[k] = row_bcast(a[j], k);
#pragma unroll
        for (int k = 0; k <= j; ++k) mj[k] = row_bcast(mr[k], j);
        double d = c[j];
        if (!(d > 0.0)) { bad = true; d = 1.0; }
        const double t = (lane > j) ? a[j] * rcp_pos(d) : 0.0;
        // the empty asm pins each update to its step: left alone the compiler sinks all of them to
        // their first use (a left-looking schedule that keeps every earlier column live in registers)
#pragma unroll
        for (int k = j + 1; k < 16; ++k) { a[k] = __builtin_fma(-t, c[k], a[k]); asm volatile("" : "+v"(a[k])); }
#pragma unroll
        for (int k = 0; k <= j; ++k) { mr[k] = __builtin_fma(-t, mj[k], mr[k]); asm volatile("" : "+v"(mr[k])); }
        a[j] *= rsqrt_pos(d);
        dsel = (lane == j) ? d : dsel;
    }
    const double rsl = rsqrt_pos(dsel);
#pragma unroll
    for (int k = 0; k < 16; ++k) mr[k] *= rsl;
    double lg = 0.5 * log(dsel);
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) lg += __shfl_xor(lg, o, 64);
    return lg;
}

// ------------------------------------------------------------------------------------------------
//  The same 16 x 16 factorise-and-invert atom, BLOCKED by four columns with the trailing updates on the matrix cores.
//  chol16_inv above eliminates column by column with the tile row-per-lane (four identical copies in the wave): per column 34 DPP
//  moves + 16 FMAs + a reciprocal on the dependency chain, 16 times: 3.4 us, the latency atom of every small factorisation
//  (knot chain, fronts, leaf Cholesky of a shard).  Here the symmetric tile S lives in the MFMA accumulator layout - lane (r, q)
//  holds S[4 i + q][r], i = 0 .. 3: no replication, 4 doubles per lane - and a panel step p is
//      B   = S[4p .. 4p+3, 4p .. 4p+3]                 10 numbers, fetched to every lane with v_readlane (uniform values)
//      B   = U D U^T, Linv = D^-1/2 U^-1                4 x 4, every lane the same arithmetic: three reciprocals on the chain,
//                                                       the square roots beside it
//      Y   = Linv S[4p .. 4p+3, :]                      ONE v_mfma_f64_16x16x4: A = Linv (rows 0 .. 3 of a zero-padded operand),
//                                                       B = register p of S; Y[q][r] = L[r][4p + q] lands in register 0 of lane (r, q)
//      S  -= Y^T Y                                      ONE v_mfma: lane (r, q) supplies Y[q][r] as A[m = r][k = q] AND as B[k = q][n = r]
//  and the inverse is carried along on the identity, M <- E_p M, two more MFMAs per panel off the critical path:
//      M[4p .. 4p+3, :] = Linv M[4p .. 4p+3, :],    M[rows below] -= L[rows below, panel] M[4p .. 4p+3, :].
//  The symmetry of S is what makes register p serve as both the panel (A operand) and its transpose (B operand) without a
//  single cross-lane move.  Results go to LDS row-major (L lower triangle, L^-1 lower triangle); returns sum(log diag L).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double readlane_f64(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
template <int ABL = 0>      // ABL: ablation bits for the timing lab (tools/chol_latency.hip) only - wrong results; 0 everywhere else
__device__ __forceinline__ double chol16_mfma(const double* __restrict__ tile /* LDS, 16 x 16 row-major, lower triangle valid */,
                                              double* __restrict__ Lout, double* __restrict__ Iout /* LDS 16 x 16 row-major each */,
                                              int lane, bool& bad) {
    asm volatile("" : "+v"(lane));
    const int r = lane & 15, q = lane >> 4;
    d4 s, m;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 4 * i + q;
        s[i] = row >= r ? tile[row * 16 + r] : tile[r * 16 + row];
        m[i] = row == r ? 1.0 : 0.0;
    }
    double dsel = 1.0;
    double lcol[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        // ---- the diagonal block, uniform: b[a][c] = S[4p + a][4p + c] sits in lane (r = 4p + c, q = a), register p
        double b[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int c = 0; c <= a; ++c) b[a][c] = (ABL & 16) ? s[p] + (double)(a + c) : readlane_f64(s[p], (4 * p + c) + 16 * a);
        // ---- B = U D U^T (unit lower U), reciprocals of the pivots on the chain, square roots beside it
        double d0 = b[0][0];
        if (!(d0 > 0.0)) { bad = true; d0 = 1.0; }
#define MRA_ATOM_RCP(x) ((ABL & 2) ? __builtin_amdgcn_rcp(x) : rcp_pos(x))
#define MRA_ATOM_RSQ(x) ((ABL & 4) ? __builtin_amdgcn_rsq(x) : rsqrt_pos(x))
        const double i0 = MRA_ATOM_RCP(d0);
        const double u10 = b[1][0] * i0, u20 = b[2][0] * i0, u30 = b[3][0] * i0;
        double d1 = __builtin_fma(-u10, b[1][0], b[1][1]);
        if (!(d1 > 0.0)) { bad = true; d1 = 1.0; }
        const double i1 = MRA_ATOM_RCP(d1);
        const double c21 = __builtin_fma(-u20, b[1][0], b[2][1]), c31 = __builtin_fma(-u30, b[1][0], b[3][1]);
        const double u21 = c21 * i1, u31 = c31 * i1;
        double d2 = __builtin_fma(-u21, c21, __builtin_fma(-u20, b[2][0], b[2][2]));
        if (!(d2 > 0.0)) { bad = true; d2 = 1.0; }
        const double i2 = MRA_ATOM_RCP(d2);
        const double c32 = __builtin_fma(-u31, c21, __builtin_fma(-u30, b[2][0], b[3][2]));
        const double u32 = c32 * i2;
        double d3 = __builtin_fma(-u32, c32, __builtin_fma(-u31, c31, __builtin_fma(-u30, b[3][0], b[3][3])));
        if (!(d3 > 0.0)) { bad = true; d3 = 1.0; }
        const double rs0 = MRA_ATOM_RSQ(d0), rs1 = MRA_ATOM_RSQ(d1), rs2 = MRA_ATOM_RSQ(d2), rs3 = MRA_ATOM_RSQ(d3);
#undef MRA_ATOM_RCP
#undef MRA_ATOM_RSQ
        // U^-1 (unit lower)
        const double v10 = -u10, v21 = -u21, v32 = -u32;
        const double v20 = __builtin_fma(u21, u10, -u20), v31 = __builtin_fma(u32, u21, -u31);
        const double v30 = -__builtin_fma(u32, v20, __builtin_fma(u31, v10, u30));
        // ---- this lane's element of the zero-padded A operand: Linv[r][q] = rs_r v_rq for q <= r < 4
        const double vr = (r == 0) ? 1.0 : (r == 1) ? (q == 0 ? v10 : 1.0) : (r == 2) ? (q == 0 ? v20 : (q == 1 ? v21 : 1.0)) : (q == 0 ? v30 : (q == 1 ? v31 : (q == 2 ? v32 : 1.0)));
        const double rsr = (r == 0) ? rs0 : (r == 1) ? rs1 : (r == 2) ? rs2 : rs3;
        const double la = (r < 4 && q <= r) ? rsr * vr : 0.0;
        const d4 zero = {0, 0, 0, 0};
        // ---- Y = Linv S[panel rows, :]; the rows of the panel and above are finished: only columns r >= 4p carry the factor
        const d4 yy = mfma16(la, s[p], zero);
        const double y = (r >= 4 * p) ? yy[0] : 0.0;
        lcol[p] = (r >= 4 * p + q) ? y : 0.0;                 // L[r][4p + q]
        // ---- S -= Y^T Y (the finished rows / columns are not looked at again)
        if (p < 3) s = mfma16(-y, y, s);
        // ---- the inverse: M[panel rows] = Linv M[panel rows]; rows below -= L[below, panel] M[panel rows]
        if (!(ABL & 1)) {
            const d4 mp = mfma16(la, m[p], zero);
            m[p] = mp[0];
            if (p < 3) {
                const double yb = (r >= 4 * p + 4) ? y : 0.0;     // (zero rows of the A operand leave the panel rows and everything above alone)
                m = mfma16(-yb, mp[0], m);
            }
        }
        dsel = (r == 4 * p) ? d0 : (r == 4 * p + 1) ? d1 : (r == 4 * p + 2) ? d2 : (r == 4 * p + 3) ? d3 : dsel;
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        Lout[r * 16 + 4 * p + q] = lcol[p];
        Iout[(4 * p + q) * 16 + r] = (4 * p + q >= r) ? m[p] : 0.0;
    }
    if (ABL & 8) return dsel;
    double lg = (q == 0) ? 0.5 * log(dsel) : 0.0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) lg += __shfl_xor(lg, o, 64);
    return lg;
}

// The same blocked atom in square-root-free form (what the measurements of tools/chol_latency.hip asked for: the first version spent
// its time ISSUING the 4 x 4 work that every lane repeats - 16 rsqrt_pos of 11 instructions each, nested selects, an ocml log - not
// waiting for its chain).  S = U D U^T with unit-lower U:
//      Z   = U_pp^-1 S[panel rows, :]                   one MFMA (A = the 4 x 4 unit-lower inverse, zero-padded)
//      S  -= (D_p^-1 Z)^T Z                             one MFMA; D_p^-1 Z = the multipliers U[:, panel]^T, a per-lane scaling by 1 / d_q
//      N[panel rows] = U_pp^-1 N[panel rows];  N[below] -= U[below, panel] N[panel rows]        (N: the unit-lower inverse)
// and only at the end L[r][4p + q] = Z_p[q][r] / sqrt(d_{4p+q}),  L^-1[4p + q][c] = N[4p + q][c] / sqrt(d_{4p+q}): FOUR rsqrt per
// lane in all (its own pivots).  The log-determinant comes from the product of the pivots (exponents kept apart): one logarithm
// of one uniform number, no reduction.
__device__ __forceinline__ double log_pos_uniform(double x) {           // log(x), x > 0 normal: frexp + atanh series; ~1 ulp of the result
    int e;
    double m = __builtin_frexp(x, &e);                                   // m in [0.5, 1)
    if (m < 0.70710678118654752) { m += m; --e; }                       // m in [sqrt(1/2), sqrt(2))
    const double t = (m - 1.0) * rcp_pos(m + 1.0), t2 = t * t;         // |t| <= 0.1716
    double p = 1.0 / 21.0;
    p = __builtin_fma(p, t2, 1.0 / 19.0); p = __builtin_fma(p, t2, 1.0 / 17.0); p = __builtin_fma(p, t2, 1.0 / 15.0);
    p = __builtin_fma(p, t2, 1.0 / 13.0); p = __builtin_fma(p, t2, 1.0 / 11.0); p = __builtin_fma(p, t2, 1.0 / 9.0);
    p = __builtin_fma(p, t2, 1.0 / 7.0); p = __builtin_fma(p, t2, 1.0 / 5.0); p = __builtin_fma(p, t2, 1.0 / 3.0);
    p = __builtin_fma(p, t2, 1.0);
    return __builtin_fma((double)e, 0.69314718055994530942, 2.0 * t * p);
}
// tile / Lout / Iout: LDS, rows of ldt / ldl / ldi doubles (Lout may be the tile itself: every lane has read its share before
// anybody writes); gL / gI (optional, global memory): the same two results once more, rows of gldl / 16 doubles
template <int ABL = 0>
__device__ __forceinline__ double chol16_ldl(const double* tile, int ldt, double* Lout, int ldl, double* Iout, int ldi,
                                             int lane, bool& bad, double* __restrict__ gL = nullptr, long gldl = 0, double* __restrict__ gI = nullptr) {
    asm volatile("" : "+v"(lane));
    const int r = lane & 15, q = lane >> 4;
    d4 s, n;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 4 * i + q;
        s[i] = row >= r ? tile[row * ldt + r] : tile[r * ldt + row];
        n[i] = row == r ? 1.0 : 0.0;
    }
    const d4 zero = {0, 0, 0, 0};
    double lcol[4], dq[4];
    double prod = 1.0;
    int pexp = 0;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        double b[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int c = 0; c <= a; ++c) b[a][c] = readlane_f64(s[p], (4 * p + c) + 16 * a);
        double d0 = b[0][0];
        const double i0 = rcp_pos(d0);
        const double u10 = b[1][0] * i0, u20 = b[2][0] * i0, u30 = b[3][0] * i0;
        double d1 = __builtin_fma(-u10, b[1][0], b[1][1]);
        const double i1 = rcp_pos(d1);
        const double c21 = __builtin_fma(-u20, b[1][0], b[2][1]), c31 = __builtin_fma(-u30, b[1][0], b[3][1]);
        const double u21 = c21 * i1, u31 = c31 * i1;
        double d2 = __builtin_fma(-u21, c21, __builtin_fma(-u20, b[2][0], b[2][2]));
        const double i2 = rcp_pos(d2);
        const double c32 = __builtin_fma(-u31, c21, __builtin_fma(-u30, b[2][0], b[3][2]));
        const double u32 = c32 * i2;
        double d3 = __builtin_fma(-u32, c32, __builtin_fma(-u31, c31, __builtin_fma(-u30, b[3][0], b[3][3])));
        const double i3 = rcp_pos(d3);
        // not positive definite (or NaN): flagged once per panel off the chain; the numbers that follow are garbage either way
        if (!(fmin(fmin(d0, d1), fmin(d2, d3)) > 0.0)) bad = true;
        // U_pp^-1 (unit lower), this lane's element of the zero-padded A operand: V[r][q], q <= r < 4
        const double v20 = __builtin_fma(u21, u10, -u20), v31 = __builtin_fma(u32, u21, -u31);
        const double v30 = -__builtin_fma(u32, v20, __builtin_fma(-u31, u10, u30));
        const double vq0 = (r == 1) ? -u10 : (r == 2) ? v20 : v30;           // column q = 0, rows 1 .. 3
        const double vq1 = (r == 2) ? -u21 : v31;                             // column q = 1, rows 2, 3
        const double vlow = (q == 0) ? vq0 : (q == 1) ? vq1 : -u32;          // strictly lower part
        const double va = (r < 4) ? ((q == r) ? 1.0 : (q < r ? vlow : 0.0)) : 0.0;
        const double iq = (q == 0) ? i0 : (q == 1) ? i1 : (q == 2) ? i2 : i3;
        dq[p] = (q == 0) ? d0 : (q == 1) ? d1 : (q == 2) ? d2 : d3;
        // Z = U_pp^-1 S[panel rows, :]; the columns left of the panel are finished
        const d4 zz = mfma16(va, s[p], zero);
        const double z = (r >= 4 * p) ? zz[0] : 0.0;
        const double w = z * iq;                             // U[r][4p + q]
        lcol[p] = (r >= 4 * p + q) ? z : 0.0;
        if (p < 3) s = mfma16(-w, z, s);
        if (!(ABL & 1)) {
            const d4 np = mfma16(va, n[p], zero);
            n[p] = np[0];
            if (p < 3) {
                const double wb = (r >= 4 * p + 4) ? w : 0.0;
                n = mfma16(-wb, np[0], n);
            }
        }
        // product of the pivots, exponent kept apart (sixteen pivots of 1e-20 would underflow a plain product)
        prod *= (d0 * d1) * (d2 * d3);
        int e2;
        prod = __builtin_frexp(prod, &e2);
        pexp += e2;
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const double rs = rsqrt_pos(dq[p]);
        const double lv = lcol[p] * rs, iv = (4 * p + q >= r) ? n[p] * rs : 0.0;
        Lout[r * ldl + 4 * p + q] = lv;
        Iout[(4 * p + q) * ldi + r] = iv;
        if (gL) gst(gL + (long)r * gldl + 4 * p + q, lv);
        if (gI) gst(gI + (4 * p + q) * 16 + r, iv);
    }
    if (ABL & 8) return prod;
    return 0.5 * __builtin_fma((double)pexp, 0.69314718055994530942, log_pos_uniform(prod));
}

// ------------------------------------------------------------------------------------------------
//  blocked partial Cholesky of a tall panel (one workgroup per problem)
// ------------------------------------------------------------------------------------------------
struct PanelProb {
    double* P;          // ht*16 rows, row-major, ld; leading ne*16 columns are eliminated
    double* invd;       // ne blocks of 16x16: inverses of the diagonal blocks of L (row-major)
    long ld;
    int ht;             // row tiles
    int ne;             // column tiles to eliminate
    int node;           // where to put 2*sum(log diag L)
};

#ifndef MRA_KERNELS_TEMPLATES_ONLY      /* non-template kernel: defined once, in the translation unit of mra_plan.hip */
__global__ __launch_bounds__(256, 2) void k_panel_chol(const PanelProb* __restrict__ probs,
                                                     double* __restrict__ dnode, int* __restrict__ err, int accumulate = 0) {
    const PanelProb pb = probs[blockIdx.x];
    __shared__ __attribute__((aligned(16))) double sd[16][17];
    __shared__ double sinv[16][17];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    double logacc = 0.0;
    const d4 zero = {0, 0, 0, 0};
    for (int jb = 0; jb < pb.ne; ++jb) {
        // ---- phase 1: column tile jb of every row tile ib >= jb gets its left-looking update
        for (int ib = jb + wave; ib < pb.ht; ib += 4) {
            double* tp = pb.P + (long)ib * 16 * pb.ld + jb * 16;
            d4 acc = load_rowlane(tp, pb.ld, r, q);
            d4 upd = zero;
            const double* arow = pb.P + (long)jb * 16 * pb.ld;
            const double* brow = pb.P + (long)ib * 16 * pb.ld;
            for (int kb = 0; kb < jb; ++kb) {
                d4 a = load_rowlane(arow + kb * 16, pb.ld, r, q);
                d4 b = load_rowlane(brow + kb * 16, pb.ld, r, q);
#pragma unroll
                for (int s = 0; s < 4; ++s) upd = mfma16(a[s], b[s], upd);
            }
            acc -= upd;
            if (ib == jb) {
#pragma unroll
                for (int s = 0; s < 4; ++s) sd[r][q + 4 * s] = acc[s];
            } else {
                store_rowlane(tp, pb.ld, r, q, acc);
            }
        }
        __syncthreads();
        // ---- phase 2: factorise + invert the diagonal block (wave 0)
        if (wave == 0) {
            bool bad = false;
#ifdef MRA_ATOM_OLD
            double a[16], m[16];
            const int rl = lane & 15;
#pragma unroll
            for (int k = 0; k < 16; ++k) a[k] = (k <= rl) ? sd[rl][k] : 0.0;
            double ls = chol16_inv(a, m, rl, bad, &sd[0][0]);
            logacc += ls;
            if (lane < 16) {
                double* dp = pb.P + (long)(jb * 16 + lane) * pb.ld + jb * 16;
#pragma unroll
                for (int k = 0; k < 16; k += 2) gst2(dp + k, d2{a[k], a[k + 1]});
                double* ip = pb.invd + (long)jb * 256;
#pragma unroll
                for (int k = 0; k < 16; ++k) sinv[lane][k] = m[k];
#pragma unroll
                for (int k = 0; k < 16; k += 2) gst2(ip + lane * 16 + k, d2{m[k], m[k + 1]});
            }
#else
            logacc += chol16_ldl(&sd[0][0], 17, &sd[0][0], 17, &sinv[0][0], 17, lane, bad, pb.P + (long)(jb * 16) * pb.ld + jb * 16, pb.ld, pb.invd + (long)jb * 256);
#endif
            if (bad && lane == 0) atomicMax(err, pb.node + 1);
        }
        __syncthreads();
        // ---- phase 3: rows below: X^T = L_jj^{-1} T^T
        {
            d4 a;
#pragma unroll
            for (int s = 0; s < 4; ++s) a[s] = sinv[r][q + 4 * s];
            for (int ib = jb + wave; ib < pb.ht; ib += 4) {
                if (ib == jb) continue;
                double* tp = pb.P + (long)ib * 16 * pb.ld + jb * 16;
                d4 b = load_rowlane(tp, pb.ld, r, q);
                d4 x = zero;
#pragma unroll
                for (int s = 0; s < 4; ++s) x = mfma16(a[s], b[s], x);
                store_rowlane(tp, pb.ld, r, q, x);
            }
        }
        __syncthreads();
    }
    // accumulate: a later 64-column step of a blocked factorisation (large leaves) adds its share of the log-determinant
    if (threadIdx.x == 0) dnode[pb.node] = (accumulate ? dnode[pb.node] : 0.0) + 2.0 * logacc;
}
#endif

// ------------------------------------------------------------------------------------------------
//  Cholesky of many small matrices (the leaves' C blocks): ONE WAVE per matrix, no workgroup barriers,
//  so the serial 16x16 diagonal-block factorisations of different matrices overlap on a CU instead of
//  idling three of four waves (k_panel_chol).  Left-looking over 16x16 tiles in vec layout; the
//  matrix stays in global memory (L1/L2-hot, written and re-read by the same wave).
// ------------------------------------------------------------------------------------------------
// One wave factorises an nt*16 square matrix in place (lower part), writes the inverted diagonal blocks
// and returns sum(log diag L).  sd / si: 2 KB of LDS scratch each, private to the wave.
__device__ __forceinline__ double chol_wave_body(double* __restrict__ P, long ld, int nt, double* __restrict__ invd,
                                                 double* sd, double* si, int lane, bool& bad) {
    const int r = lane & 15, q = lane >> 4;
    const int prow = pi16(r);
    const d4 zero = {0, 0, 0, 0};
    double logacc = 0.0;
#pragma unroll 1
    for (int jb = 0; jb < nt; ++jb) {
        // ---- diagonal tile: left-looking update, then factor + invert
        {
            double* tp = P + (long)(jb * 16 + r) * ld + jb * 16 + 4 * q;
            d4 acc = gld4(tp);
            d4 upd0 = zero, upd1 = zero;          // two chains: dependent f64 MFMAs issue every ~73 ns, independent ones every ~29 ns
            for (int kb = 0; kb < jb; ++kb) {
                const d4 a = gld4(P + (long)(jb * 16 + prow) * ld + kb * 16 + 4 * q);
                const d4 b = gld4(P + (long)(jb * 16 + r) * ld + kb * 16 + 4 * q);
                upd0 = mfma16(a[0], b[0], upd0);
                upd1 = mfma16(a[1], b[1], upd1);
                upd0 = mfma16(a[2], b[2], upd0);
                upd1 = mfma16(a[3], b[3], upd1);
            }
            acc -= upd0 + upd1;
            *(d4*)(sd + r * 16 + 4 * q) = acc;
        }
        __builtin_amdgcn_wave_barrier();
#ifdef MRA_ATOM_OLD
        {
            double a[16], m[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) a[k] = (k <= r) ? sd[r * 16 + k] : 0.0;
            logacc += chol16_inv(a, m, r, bad, sd);
            if (lane < 16) {
                double* dp = P + (long)(jb * 16 + lane) * ld + jb * 16;
                double* ip2 = invd + (long)jb * 256;
#pragma unroll
                for (int k = 0; k < 16; k += 2) gst2(dp + k, d2{a[k], a[k + 1]});
#pragma unroll
                for (int k = 0; k < 16; k += 2) { *(d2*)(si + lane * 16 + k) = d2{m[k], m[k + 1]}; gst2(ip2 + lane * 16 + k, d2{m[k], m[k + 1]}); }
            }
        }
#else
        // factor + inverse on the matrix cores (chol16_ldl): L over the tile in LDS and to the matrix in global memory, the inverse
        // into si and to invd
        logacc += chol16_ldl(sd, 16, sd, 16, si, 16, lane, bad, P + (long)(jb * 16) * ld + jb * 16, ld, invd + (long)jb * 256);
#endif
        __builtin_amdgcn_wave_barrier();
        const d4 ia = *(const d4*)(si + prow * 16 + 4 * q);
        // ---- rows below
        for (int ib = jb + 1; ib < nt; ++ib) {
            double* tp = P + (long)(ib * 16 + r) * ld + jb * 16 + 4 * q;
            d4 acc = gld4(tp);
            d4 upd0 = zero, upd1 = zero;
            for (int kb = 0; kb < jb; ++kb) {
                const d4 a = gld4(P + (long)(jb * 16 + prow) * ld + kb * 16 + 4 * q);
                const d4 b = gld4(P + (long)(ib * 16 + r) * ld + kb * 16 + 4 * q);
                upd0 = mfma16(a[0], b[0], upd0);
                upd1 = mfma16(a[1], b[1], upd1);
                upd0 = mfma16(a[2], b[2], upd0);
                upd1 = mfma16(a[3], b[3], upd1);
            }
            acc -= upd0 + upd1;
            d4 x0 = mfma16(ia[0], acc[0], zero), x1 = mfma16(ia[1], acc[1], zero);
            x0 = mfma16(ia[2], acc[2], x0);
            x1 = mfma16(ia[3], acc[3], x1);
            gst4(tp, x0 + x1);
        }
        __threadfence_block();          // this wave's stores before its own later loads (other lanes)
    }
    return logacc;
}

template <int NTMAX>
__global__ __launch_bounds__(256, 4) void k_chol_wave(const PanelProb* __restrict__ probs, int nprob,
                                                    double* __restrict__ dnode, int* __restrict__ err) {
    __shared__ __attribute__((aligned(16))) double sdiag[4][16 * 16 + 32];     // 272 used: tile + exchange column
    __shared__ __attribute__((aligned(16))) double sinv[4][16 * 16];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ip = blockIdx.x * 4 + wave;
    if (ip >= nprob) return;
    const PanelProb pb = probs[ip];
    bool bad = false;
    const double logacc = chol_wave_body(pb.P, pb.ld, pb.ne, pb.invd, sdiag[wave], sinv[wave], lane, bad);
    if (lane == 0) {
        dnode[pb.node] = 2.0 * logacc;
        if (bad) atomicMax(err, pb.node + 1);
    }
}

// ------------------------------------------------------------------------------------------------
//  The same factorisation with one WORKGROUP per matrix and the matrix in REGISTERS.  k_chol_wave is one wave per matrix walking
//  a chain of global round trips (L2-hot, but 90 us for 112 x 112 however few matrices there are - a fixed cost of every shard).
//  A 16 x 16 tile is 4 doubles per lane: the strictly-lower tiles are dealt to the NW - 1 worker waves (tile t to worker
//  t mod (NW - 1), NSLOT tiles each, never moved), the diagonal tiles live in LDS, and wave 0 does nothing but factorise + invert
//  diagonal blocks (the 3.3 us atom) - the NEXT one while the workers apply the rest of the trailing update (right-looking):
//      A   workers: column jb below the diagonal, X = T L_jj^{-T}, to LDS (exchange buffer, by row tile) and to global memory
//      B   wave 0:  tile (jb+1, jb+1) -= X X^T, factorise + invert it;   workers: their tiles (ib, jc) -= X_ib X_jc^T, jc > jb,
//                   and the later diagonal tiles
//  Two barriers per column, about 4.5 us per column.  Wave 0 and the workers run separate loops (same barrier count), so that the
//  register allocation is the larger of the two working sets, not their sum.  LDS: 2 NT + 1 tiles of 2 KB.
// ------------------------------------------------------------------------------------------------
template <int NT, int NW>
__global__ __launch_bounds__(NW * 64, NT <= 8 ? 3 : 2) void k_chol_tiles(const PanelProb* __restrict__ probs, double* __restrict__ dnode,
                                                                      int* __restrict__ err) {
    constexpr int NWK = NW - 1;
    constexpr int NOFF = NT * (NT - 1) / 2;
    constexpr int NSLOT = (NOFF + NWK - 1) / NWK;
    constexpr int NTH = NW * 64;
    __shared__ __attribute__((aligned(16))) double sdiag[NT * 256];
    __shared__ __attribute__((aligned(16))) double sx[NT * 256];
    __shared__ __attribute__((aligned(16))) double finv[256];
    const PanelProb pb = probs[blockIdx.x];
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    const int prow = pi16(r);
    const int nt = pb.ne;
    const d4 zero = {0, 0, 0, 0};
    if (nt == 0) {                                   // a leaf without observations
        if (threadIdx.x == 0) dnode[pb.node] = 0.0;
        return;
    }
    // ---- diagonal tiles into LDS (16-byte pieces, all loads of a thread in flight before its first write)
    {
        constexpr int PD = (NT * 128 + NTH - 1) / NTH;
        const int total = nt * 128;
        d2 v[PD];
#pragma unroll
        for (int i = 0; i < PD; ++i) {
            int e = (int)threadIdx.x + i * NTH;
            e = e < total ? e : total - 1;
            const int tile = e >> 7, row = (e >> 3) & 15, c2 = (e & 7) << 1;
            v[i] = gld2(pb.P + (long)(tile * 16 + row) * pb.ld + tile * 16 + c2);
        }
#pragma unroll
        for (int i = 0; i < PD; ++i) {
            const int e = (int)threadIdx.x + i * NTH;
            if (e < total) *(d2*)(sdiag + (long)e * 2) = v[i];
        }
    }
    __syncthreads();
    if (wave == 0) {
        double logacc = 0.0;
        bool bad = false;
        // factorise + invert diagonal tile jd: factor to global memory, inverse into finv and to global memory
        auto factor = [&](int jd) {
#ifdef MRA_ATOM_OLD
            double a[16], m[16];
            const double* dt = sdiag + jd * 256;
#pragma unroll
            for (int k = 0; k < 16; ++k) a[k] = (k <= r) ? dt[r * 16 + k] : 0.0;
            logacc += chol16_inv(a, m, r, bad, nullptr);
            if (lane < 16) {
                double* dp = pb.P + (long)(jd * 16 + lane) * pb.ld + jd * 16;
                double* ip2 = pb.invd + (long)jd * 256;
#pragma unroll
                for (int k = 0; k < 16; k += 2) {
                    gst2(dp + k, d2{a[k], a[k + 1]});
                    *(d2*)(finv + lane * 16 + k) = d2{m[k], m[k + 1]};
                    gst2(ip2 + lane * 16 + k, d2{m[k], m[k + 1]});
                }
            }
#else
            double* dt = sdiag + jd * 256;
            logacc += chol16_ldl(dt, 16, dt, 16, finv, 16, lane, bad, pb.P + (long)(jd * 16) * pb.ld + jd * 16, pb.ld, pb.invd + (long)jd * 256);
#endif
        };
        factor(0);
        __syncthreads();
#pragma unroll 1
        for (int jb = 0; jb < nt - 1; ++jb) {
            __syncthreads();                                   // column jb is in sx
            const int j1 = jb + 1;
            const d4 a = *(const d4*)(sx + j1 * 256 + prow * 16 + 4 * q);
            const d4 b = *(const d4*)(sx + j1 * 256 + r * 16 + 4 * q);
            d4 u0 = mfma16(a[0], b[0], zero), u1 = mfma16(a[1], b[1], zero);
            u0 = mfma16(a[2], b[2], u0);
            u1 = mfma16(a[3], b[3], u1);
            d4* tp = (d4*)(sdiag + j1 * 256 + r * 16 + 4 * q);
            *tp = *tp - (u0 + u1);
            lds_wave_sync();
            factor(j1);
            __syncthreads();
        }
        if (lane == 0) {
            dnode[pb.node] = 2.0 * logacc;
            if (bad) atomicMax(err, pb.node + 1);
        }
    } else {
        const int w = wave - 1;
        d4 T[NSLOT];
        int tib[NSLOT], tjc[NSLOT];
#pragma unroll
        for (int s = 0; s < NSLOT; ++s) {
            const int t = s * NWK + w;
            int ib = 1;
            while ((ib + 1) * ib / 2 <= t) ++ib;
            tib[s] = (t < NOFF && ib < nt) ? ib : 0;            // 0: no such tile in this matrix
            tjc[s] = t - ib * (ib - 1) / 2;
            T[s] = zero;
            if (tib[s]) T[s] = gld4(pb.P + (long)(tib[s] * 16 + r) * pb.ld + tjc[s] * 16 + 4 * q);
        }
        __syncthreads();
#pragma unroll 1
        for (int jb = 0; jb < nt - 1; ++jb) {
            // ---- A
            {
                const d4 ia = *(const d4*)(finv + prow * 16 + 4 * q);
#pragma unroll
                for (int s = 0; s < NSLOT; ++s) {
                    if (tib[s] && tjc[s] == jb) {
                        d4 x0 = mfma16(ia[0], T[s][0], zero), x1 = mfma16(ia[1], T[s][1], zero);
                        x0 = mfma16(ia[2], T[s][2], x0);
                        x1 = mfma16(ia[3], T[s][3], x1);
                        const d4 x = x0 + x1;
                        *(d4*)(sx + tib[s] * 256 + r * 16 + 4 * q) = x;
                        gst4(pb.P + (long)(tib[s] * 16 + r) * pb.ld + jb * 16 + 4 * q, x);
                    }
                }
            }
            __syncthreads();
            // ---- B
#pragma unroll
            for (int s = 0; s < NSLOT; ++s) {
                if (tib[s] && tjc[s] > jb) {
                    const d4 a = *(const d4*)(sx + tjc[s] * 256 + prow * 16 + 4 * q);
                    const d4 b = *(const d4*)(sx + tib[s] * 256 + r * 16 + 4 * q);
                    d4 u = mfma16(-a[1], b[1], zero);
                    T[s] = mfma16(-a[0], b[0], T[s]);
                    u = mfma16(-a[3], b[3], u);
                    T[s] = mfma16(-a[2], b[2], T[s]);
                    T[s] += u;
                }
            }
            for (int j = jb + 2 + w; j < nt; j += NWK) {
                const d4 a = *(const d4*)(sx + j * 256 + prow * 16 + 4 * q);
                const d4 b = *(const d4*)(sx + j * 256 + r * 16 + 4 * q);
                d4 u0 = mfma16(a[0], b[0], zero), u1 = mfma16(a[1], b[1], zero);
                u0 = mfma16(a[2], b[2], u0);
                u1 = mfma16(a[3], b[3], u1);
                d4* tp = (d4*)(sdiag + j * 256 + r * 16 + 4 * q);
                *tp = *tp - (u0 + u1);
            }
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------------
//  X = R L^{-T} on 16-row tiles (in place), optional var += rowsumsq(X)
// ------------------------------------------------------------------------------------------------
struct TrsmNode {
    const double* L;      // cwt*16 square, row-major (lower part used)
    const double* invd;   // cwt inverted diagonal blocks
    long ldL;
    int cwt;
};

#ifndef MRA_KERNELS_TEMPLATES_ONLY      /* non-template kernel: defined once, in the translation unit of mra_plan.hip */
__global__ __launch_bounds__(256, 2) void k_trsm_rows(const TrsmNode* __restrict__ nodes,
                                                    const int* __restrict__ tile_node,
                                                    const long* __restrict__ tile_row0, long ntiles,
                                                    double* __restrict__ W, long ldw, int c0,
                                                    double* __restrict__ var) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    const long t = (long)blockIdx.x * 4 + wave;
    if (t >= ntiles) return;
    const TrsmNode nd = nodes[tile_node[t]];
    double* wp = W + tile_row0[t] * ldw + c0;
    const d4 zero = {0, 0, 0, 0};
    double ssq = 0.0;
    for (int jb = 0; jb < nd.cwt; ++jb) {
        d4 acc = load_rowlane(wp + jb * 16, ldw, r, q);
        d4 upd = zero;
        for (int kb = 0; kb < jb; ++kb) {
            d4 a = load_rowlane(nd.L + (long)jb * 16 * nd.ldL + kb * 16, nd.ldL, r, q);
            d4 b = load_rowlane(wp + kb * 16, ldw, r, q);
#pragma unroll
            for (int s = 0; s < 4; ++s) upd = mfma16(a[s], b[s], upd);
        }
        acc -= upd;
        d4 ia = load_rowlane(nd.invd + (long)jb * 256, 16, r, q);
        d4 x = zero;
#pragma unroll
        for (int s = 0; s < 4; ++s) x = mfma16(ia[s], acc[s], x);
        store_rowlane(wp + jb * 16, ldw, r, q, x);
        ssq += x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3];
    }
    if (var) {
        ssq += __shfl_xor(ssq, 16, 64);
        ssq += __shfl_xor(ssq, 32, 64);
        if (q == 0) var[tile_row0[t] + r] += ssq;
    }
}
#endif

// ------------------------------------------------------------------------------------------------
//  X = R L^{-T} on 16-row tiles, second generation: L and the inverted diagonal blocks are staged in
//  LDS once per workgroup, the solved tiles X_kb stay in registers, and every global access moves
//  32 contiguous bytes per lane.
//
//  "vec" tile layout: element j of lane (r,q) is tile[r][4q + j].  It is the B operand of k-step j
//  (k <-> column 4q+j).  The A operand of k-step j is then Amat[pi(r)][4q + j] with the row
//  permutation pi(rho) = 4 (rho & 3) + (rho >> 2): with it the accumulator (row rho = q + 4 s) lands
//  in vec layout again (column pi(q + 4 s) = 4 q + s), so solved tiles chain as B operands and are
//  stored with 32-byte accesses.
// ------------------------------------------------------------------------------------------------
//  Staging of 16x16 operand tiles (2 KB each = 128 16-byte chunks, LDS image linear in the chunk
//  number) with G loads of every thread in flight at once: a plain load -> ds_write loop serialises
//  one L2 round trip (~1 us) per chunk.  (Register arrays use the native vector type d2: arrays of
//  HIP's double2 class are not promoted to registers and end up in scratch memory.)
// ------------------------------------------------------------------------------------------------
template <int G, class SrcFn>
__device__ __forceinline__ void stage_chunks(double* __restrict__ ldsb, int total, SrcFn src) {
    for (int e0 = threadIdx.x; e0 < total; e0 += G * (int)blockDim.x) {
        d2 v[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int e = e0 + g * (int)blockDim.x;
            v[g] = gld2(src(e < total ? e : total - 1));             // clamped: always a valid address
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int e = e0 + g * (int)blockDim.x;
            if (e < total) *(d2*)(ldsb + 2 * e) = v[g];
        }
    }
}

// ------------------------------------------------------------------------------------------------
struct Trsm2Prob {
    const double* L;      // nt*16 square lower-triangular factor, row-major
    const double* invd;   // nt inverted 16x16 diagonal blocks (row-major, 256 doubles each)
    double* X;            // first row of the row range, row-major
    double* var;          // nullptr or per-row accumulator (+= rowsumsq(X)), indexed from the first row
    long ldL, ldx;
    int nt;               // column tiles
    int ntiles;           // 16-row tiles
    int var_tile0;        // tiles >= var_tile0 update var[(t - var_tile0)*16 + r]
    double var_sign;      // +1: var += |x|^2, -1: var -= |x|^2
    // the first gtiles row tiles take their right-hand side from a transposed gather:
    // R[a][k] = gW[gidx[k] * gld + a]  (0 where gidx[k] < 0): the leaf's Ut = [W_anc[o] | y_o]^T
    const int* gidx;
    const double* gW;
    long gld;
    int gtiles;
};


#ifdef MRA_STAMPS
#define MRA_TSTAMP(slot) do { if (stamps && (threadIdx.x & 63) == 0) stamps[((long)blockIdx.y * 64 + t) * 8 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#define MRA_TSTAMP_ARG , unsigned long long* stamps
#else
#define MRA_TSTAMP(slot) do { } while (0)
#define MRA_TSTAMP_ARG
#endif
template <int NTMAX>
__global__ __launch_bounds__(512) void k_trsm_rows2(const Trsm2Prob* __restrict__ probs, int tiles_per_wg MRA_TSTAMP_ARG) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const Trsm2Prob pb = probs[blockIdx.y];
#ifdef MRA_STAMPS
    const unsigned long long t_wg0 = __builtin_amdgcn_s_memtime();
    const unsigned long long t_real0 = __builtin_amdgcn_s_memrealtime();
#endif
    const int t_begin = blockIdx.x * tiles_per_wg;
    if (t_begin >= pb.ntiles) return;
    const int t_end = min(pb.ntiles, t_begin + tiles_per_wg);
    const int nt = pb.nt;
    const int ntri = nt * (nt - 1) / 2;
    // ---- stage L (strictly-lower tiles, index jb(jb-1)/2+kb) and invd (after them) in LDS.
    // (Measured at C3: this 72 KB image lets ONE workgroup run per CU although 2 x 72 KB <= 160 KB; with the inverted
    // blocks read from L2 instead the image is 56 KB, two workgroups do run per CU -- and the launch is slower, 0.93 ms
    // against 0.77: the solve moves 3.4 GB in place and sits at ~75% of what HBM streams, more waves only contend.)
    stage_chunks<8>(lds, (ntri + nt) * 128, [&](int e) -> const double* {
        const int tile = e >> 7, chunk = e & 127, row = chunk >> 3, c2 = (chunk & 7) << 1;
        if (tile < ntri) {
            int jb = 1;
            while ((jb + 1) * jb / 2 <= tile) ++jb;
            const int kb = tile - jb * (jb - 1) / 2;
            return pb.L + (long)(jb * 16 + row) * pb.ldL + kb * 16 + c2;
        }
        return pb.invd + (long)(tile - ntri) * 256 + row * 16 + c2;
    });
    // the gather list of the Ut tiles (the leaf's observed rows) once per workgroup: read per tile from global memory it put a
    // second round trip in front of every gathered tile's loads
    int* const sidx = (int*)(lds + (long)(ntri + nt) * 256);
    if (pb.gtiles > 0 && t_begin < pb.gtiles)
        for (int e = threadIdx.x; e < nt * 16; e += blockDim.x) sidx[e] = gldi(pb.gidx + e);
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    const int nwave = blockDim.x >> 6;
    const int prow = pi16(r);
    const d4 zero = {0, 0, 0, 0};
#ifdef MRA_STAMPS
    const unsigned long long t_wg1 = __builtin_amdgcn_s_memtime();
#endif
    for (int t = t_begin + wave; t < t_end; t += nwave) {
#ifdef MRA_STAMPS
        if (stamps && lane == 0) { stamps[((long)blockIdx.y * 64 + t) * 8 + 0] = t_wg0; stamps[((long)blockIdx.y * 64 + t) * 8 + 1] = t_wg1; }
#endif
        MRA_TSTAMP(2);
        double* xp = pb.X + (long)t * 16 * pb.ldx + (long)r * pb.ldx + 4 * q;
        d4 x[NTMAX];
        double ssq = 0.0;
        // the whole row of tiles is fetched first (independent loads in flight together); read one
        // tile at a time each load would wait behind the store of the previous tile
#pragma unroll
        for (int jb = 0; jb < NTMAX; ++jb) {
            if (jb < nt) {
                if (t < pb.gtiles) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int gi = sidx[jb * 16 + 4 * q + j];
                        x[jb][j] = gi < 0 ? 0.0 : gld(pb.gW + (long)gi * pb.gld + t * 16 + r);
                    }
                } else {
                    x[jb] = gld4(xp + jb * 16);
                }
            }
        }
#ifdef MRA_STAMPS
        if (stamps) { __builtin_amdgcn_s_waitcnt(0); MRA_TSTAMP(3); }
#endif
#pragma unroll
        for (int jb = 0; jb < NTMAX; ++jb) {
            if (jb < nt) {
                d4 acc = x[jb];
                // two accumulators: a dependent f64 MFMA issues every ~73 ns, an independent one every ~29 ns
                d4 upd0 = zero, upd1 = zero;
#pragma unroll
                for (int kb = 0; kb < jb; ++kb) {
                    const d4 a = *(const d4*)(lds + (jb * (jb - 1) / 2 + kb) * 256 + prow * 16 + 4 * q);
                    upd0 = mfma16(a[0], x[kb][0], upd0);
                    upd1 = mfma16(a[1], x[kb][1], upd1);
                    upd0 = mfma16(a[2], x[kb][2], upd0);
                    upd1 = mfma16(a[3], x[kb][3], upd1);
                }
                if (jb > 0) acc -= upd0 + upd1;
                const d4 ia = *(const d4*)(lds + (ntri + jb) * 256 + prow * 16 + 4 * q);
                d4 xx0 = mfma16(ia[0], acc[0], zero), xx1 = mfma16(ia[1], acc[1], zero);
                xx0 = mfma16(ia[2], acc[2], xx0);
                xx1 = mfma16(ia[3], acc[3], xx1);
                const d4 xx = xx0 + xx1;
                x[jb] = xx;
                gst4(xp + jb * 16, xx);
                ssq += xx[0] * xx[0] + xx[1] * xx[1] + xx[2] * xx[2] + xx[3] * xx[3];
            }
        }
        MRA_TSTAMP(4);
        if (pb.var && t >= pb.var_tile0) {
            ssq += __shfl_xor(ssq, 16, 64);
            ssq += __shfl_xor(ssq, 32, 64);
            if (q == 0) { double* vp = pb.var + (long)(t - pb.var_tile0) * 16 + r; gst(vp, gld(vp) + pb.var_sign * ssq); }
        }
        MRA_TSTAMP(5);
#ifdef MRA_STAMPS
        if (stamps && lane == 0) { stamps[((long)blockIdx.y * 64 + t) * 8 + 6] = t_real0; stamps[((long)blockIdx.y * 64 + t) * 8 + 7] = __builtin_amdgcn_s_memrealtime(); }
#endif
    }
}

// ------------------------------------------------------------------------------------------------
//  Leaf solve + update in one launch:  Tt = V[S,o] Lc^-T  (row TRSM, as k_trsm_rows2),  var -= rowsumsq(Tt),
//  W[S, anc | y] -= Tt Ut^T  (as k_leaf_gemm<SUB>)  - with Tt never written to memory: the solved tiles of a row tile stay
//  in registers and ARE the A fragments of the update product.  Separately the two steps move 3.4 + 4.4 GB at C3 (both
//  HBM-bound at 3.4-4.3 TB/s); fused, V is read once and W read and written once (MRANode.py:489-495, 510-511).
//  One workgroup (8 waves) per leaf: Lc (strictly-lower tiles + inverted diagonal blocks) is staged in LDS once, each wave
//  owns one 16-row tile per row group; Ut streams through a double-buffered LDS stage, 16 observations at a time, shared by
//  the 8 waves (52 MFMAs per wave and barrier at C3).  The rows of Ut themselves are solved beforehand by k_trsm_rows2.
// ------------------------------------------------------------------------------------------------
struct LeafSolveProb {
    const double* L;      // nt*16 square lower factor of C (row-major, ldL)
    const double* invd;   // nt inverted diagonal blocks (256 doubles each)
    const double* V;      // nrt row tiles of V[S,o] (row-major, ldx)
    const double* Ut;     // nat*16 rows (ancestors + y), solved (row-major, ldx)
    double* Wr;           // the leaf's rows of W at the first ancestor column (ldw)
    double* var;          // per-row variance of the leaf's rows
    long ldL, ldx, ldw;
    int nt;               // column tiles of Lc = 16-observation chunks
    int nrt;              // row tiles
    int nat;              // column tiles of the update (ancestors + y block)
    int zt;               // column tiles >= zt take C_in = 0 (the y block)
};

// LLDS = true: Lc staged in LDS, one 8-wave workgroup per CU.  (LLDS = false - Lc read through L1/L2, two 4-wave workgroups
// per CU - was measured: 300 B/lane of spills and exposed L2 latency in the solve, 15 % slower over the whole pass; not launched.)
#ifdef MRA_STAMPS
#define MRA_LSTAMP(slot) do { if (stamps && threadIdx.x == 0) stamps[(long)blockIdx.x * 8 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define MRA_LSTAMP_ARG , unsigned long long* stamps
#else
#define MRA_LSTAMP(slot) do { } while (0)
#define MRA_LSTAMP_ARG
#endif
template <int NTMAX, int CT, bool LLDS>
__global__ __launch_bounds__(LLDS ? 512 : 256, LLDS ? 1 : 2) void k_leaf_solve_update(const LeafSolveProb* __restrict__ probs MRA_LSTAMP_ARG) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    MRA_LSTAMP(0);
    constexpr int NTHR = LLDS ? 512 : 256;
    constexpr int NWAVE = NTHR / 64;
    constexpr int NST = (CT * 64 + NTHR - 1) / NTHR;            // 32-byte Ut staging pieces per thread
    const LeafSolveProb* __restrict__ pp = probs + blockIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    const int prow = pi16(r);
    const int nt = pp->nt, nrt = pp->nrt, nat = pp->nat, zt = pp->zt;
    const int ntri = nt * (nt - 1) / 2;
    const long ldx = pp->ldx, ldw = pp->ldw, ldL = pp->ldL;
    const double* const Lg = pp->L;
    const double* const invg = pp->invd;
    const d4 zero = {0, 0, 0, 0};
    double* const sL = lds;                                     // LLDS: (ntri + nt) tiles, FT_SZ doubles each
    double* const sU0 = lds + (LLDS ? (long)(ntri + nt) * FT_SZ : 0);        // two stages of nat*16 rows x LG_LD
    double* const sU1 = sU0 + (long)nat * 16 * LG_LD;
    if (LLDS) {
        // ---- stage Lc: 16-byte chunks, eight in flight per thread
        const int total = (ntri + nt) * 128;
        for (int e0 = threadIdx.x; e0 < total; e0 += 8 * NTHR) {
            d2 v[8];
            int dst[8];
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                int e = e0 + g * NTHR;
                e = e < total ? e : total - 1;
                const int tile = e >> 7, chunk = e & 127, row = chunk >> 3, c2 = (chunk & 7) << 1;
                const double* src;
                if (tile < ntri) {
                    int jb = 1;
                    while ((jb + 1) * jb / 2 <= tile) ++jb;
                    const int kb = tile - jb * (jb - 1) / 2;
                    src = Lg + (long)(jb * 16 + row) * ldL + kb * 16 + c2;
                } else src = invg + (long)(tile - ntri) * 256 + row * 16 + c2;
                v[g] = gld2(src);
                dst[g] = tile * FT_SZ + row * FT_LD + c2;
            }
#pragma unroll
            for (int g = 0; g < 8; ++g) if (e0 + g * NTHR < total) *(d2*)(sL + dst[g]) = v[g];
        }
    }
    // staging role for Ut chunks: 32-byte pieces (row, 4 columns) of the nat*16 x 16 slab
    const int nch = nat * 64;
    bool on[NST];
    int so[NST];
    const double* up[NST];
#pragma unroll
    for (int g = 0; g < NST; ++g) {
        const int e = threadIdx.x + g * NTHR;
        on[g] = e < nch;
        so[g] = (e >> 2) * LG_LD + ((e & 3) << 2);
        up[g] = pp->Ut + (long)(on[g] ? e >> 2 : 0) * ldx + ((e & 3) << 2);
    }
    __syncthreads();
    MRA_LSTAMP(1);
    for (int rg = 0; rg < nrt; rg += NWAVE) {
        const int t = rg + wave;
        const bool valid = t < nrt;
        const long row = (long)(valid ? t : 0) * 16 + r;
        // ---- Tt tile: x = V L^-T in registers (vec layout), var -= |x|^2
        d4 x[NTMAX];
        {
            const double* vp = pp->V + row * ldx + 4 * q;
#pragma unroll
            for (int jb = 0; jb < NTMAX; ++jb) x[jb] = (jb < nt) ? gld4(vp + jb * 16) : zero;
        }
        // first Ut chunk in flight beside the solve
        d4 sg[NST];
#pragma unroll
        for (int g = 0; g < NST; ++g) sg[g] = nt > 0 ? gld4(up[g]) : zero;    // a leaf without observations has a zero-sized panel
        double ssq = 0.0;
#pragma unroll
        for (int jb = 0; jb < NTMAX; ++jb) {
            if (jb < nt) {
                d4 acc = x[jb];
                d4 u0 = zero, u1 = zero;
#pragma unroll
                for (int kb = 0; kb < jb; ++kb) {
                    const d4 a = LLDS ? *(const d4*)(sL + (long)(jb * (jb - 1) / 2 + kb) * FT_SZ + prow * FT_LD + 4 * q)
                                      : gld4(Lg + (long)(jb * 16 + prow) * ldL + kb * 16 + 4 * q);
                    u0 = mfma16(a[0], x[kb][0], u0); u1 = mfma16(a[1], x[kb][1], u1);
                    u0 = mfma16(a[2], x[kb][2], u0); u1 = mfma16(a[3], x[kb][3], u1);
                }
                if (jb > 0) acc -= u0 + u1;
                const d4 ia = LLDS ? *(const d4*)(sL + (long)(ntri + jb) * FT_SZ + prow * FT_LD + 4 * q)
                                   : gld4(invg + (long)jb * 256 + prow * 16 + 4 * q);
                d4 y0 = mfma16(ia[0], acc[0], zero), y1 = mfma16(ia[1], acc[1], zero);
                y0 = mfma16(ia[2], acc[2], y0); y1 = mfma16(ia[3], acc[3], y1);
                const d4 xx = y0 + y1;
                x[jb] = xx;
                ssq += xx[0] * xx[0] + xx[1] * xx[1] + xx[2] * xx[2] + xx[3] * xx[3];
            }
        }
        ssq += __shfl_xor(ssq, 16, 64);
        ssq += __shfl_xor(ssq, 32, 64);
        if (valid && q == 0) { double* vp = pp->var + row; gst(vp, gld(vp) - ssq); }
        if (rg == 0) MRA_LSTAMP(2);
        // ---- update accumulators start from -W (the epilogue writes -acc = W - Tt Ut^T)
        d4 acc[CT];
        double* wp = pp->Wr + row * ldw + 4 * q;
#pragma unroll
        for (int j = 0; j < CT; ++j) acc[j] = (valid && j < nat && j < zt) ? -gld4(wp + j * 16) : zero;
        __syncthreads();                                        // the previous row group is done with the stages
#pragma unroll
        for (int g = 0; g < NST; ++g) if (on[g]) *(d4*)(sU0 + so[g]) = sg[g];
        __syncthreads();
        if (rg == 0) MRA_LSTAMP(3);
#pragma unroll
        for (int kb = 0; kb < NTMAX; ++kb) {
            if (kb < nt) {
                const double* st = (kb & 1) ? sU1 : sU0;
                const int kn = (kb + 1 < nt ? kb + 1 : kb) * 16;
#pragma unroll
                for (int g = 0; g < NST; ++g) sg[g] = gld4(up[g] + kn);
#pragma unroll
                for (int j = 0; j < CT; ++j) {
                    if (j < nat) {
                        const d4 b = *(const d4*)(st + (j * 16 + prow) * LG_LD + 4 * q);
#pragma unroll
                        for (int s4 = 0; s4 < 4; ++s4) acc[j] = mfma16(b[s4], x[kb][s4], acc[j]);
                    }
                }
                if (kb + 1 < nt) {
                    double* sn = (kb & 1) ? sU0 : sU1;
#pragma unroll
                    for (int g = 0; g < NST; ++g) if (on[g]) *(d4*)(sn + so[g]) = sg[g];
                    __syncthreads();
                }
            }
        }
        if (rg == 0) MRA_LSTAMP(4);
        if (valid) {
#pragma unroll
            for (int j = 0; j < CT; ++j) if (j < nat) gst4(wp + j * 16, -acc[j]);
        }
        if (rg == 0) MRA_LSTAMP(5);
    }
}

// ------------------------------------------------------------------------------------------------
//  Fused prior cascade ("regular" trees: every non-leaf level has the same knot-block width
//  CWT*16 and all leaves sit on the last level).
//
//  One wave owns 16 rows and walks down the levels with the whitened basis tiles of all coarser
//  levels in registers (vec layout, see k_trsm_rows2):
//      R^m   = kernel(x_i, knots_m) - sum_{k<m} W^k[i] Wk_m[:, block k]^T        (MRANode.py:73-80, 384)
//      W^m   = R^m L_m^{-T}                                                       (MRANode.py:385-387)
//  The per-node operands (knot coordinates kx, knot rows of the coarser bases Wk, factor L, inverted
//  diagonal blocks) are small, shared by every row of the node and come through L1/L2.
//  FULL mode: rows are a contiguous tile, all levels 0..mlast are processed and W is written once.
//  KNOT mode: rows are the knots of one level-(mlast+1) node (16 per tile, by index); the output is
//  that node's Wk rows (its kInv then comes from k_knot_kinv, its factor from k_panel_chol).
// ------------------------------------------------------------------------------------------------
struct CascadeLevel {
    const double* kx;     // [node][cw][DIM] knot coordinates
    const int* kvalid;    // [node][cw] 1 = real knot, 0 = phantom
    const double* Wk;     // [node][cw][m*cw] knot rows of the coarser whitened bases, level-major columns
    const double* L;      // [node][cw][cw]
    const double* invd;   // [node][cwt][256]
};
struct CascadeArgs {
    int knot_threads;       // host side only: workgroup size of a knot-mode launch
    int ycol;               // row mode: column of W's y block to initialise (-1: leave it)
    CascadeLevel lev[8];
    const double* X;          // coordinates of all rows [P][DIM]
    double* W;                // FULL: whitened basis, ldw
    long ldw;
    int coff[8];              // column offset of level m's block in W
    const int* tile_rows;     // KNOT: [ntiles][16] row numbers (-1: phantom knot)
    const int* tile_chain;    // [ntiles][8]  node slot per level
    const long* tile_row0;    // FULL: first row of each tile
    // FULL, likelihood-only passes: the row tiles are GATHERED lists of the leaves' observed rows (16 per tile, -1: padding) -
    // a likelihood needs W at the observed rows only (they feed Ut and C; the knots have their own pass), at C3 7 tiles of a leaf's 16
    const int* row_gather;    // nullptr, or [ntiles][16] row numbers
    const int* tile_knot0;    // KNOT: index of the tile's first knot inside its node (multiple of 16)
    double* Wk_out;           // KNOT: [node][cw][(mlast+1)*cw] of level mlast+1
    const long* wg_tile0;     // per workgroup: first tile and number of tiles (<= 8), all sharing one chain
    const int* wg_ntiles;
    long n_wg;
    int mlast;
    int knot_mode;
    double* var_out;          // FULL: prior residual variance C(x,x) - |W[x]|^2 (nullptr: skip)
    double cov0;
    // FULL: observed rows also scatter their whitened basis (transposed) and y into the leaf panel's
    // Ut block: Ut[a][k] = W[row][a], Ut[ycol][k] = y[row] with k = obs_pos[row]  (nullptr: skip)
    const int* obs_pos;       // [P] position of a row among its leaf's observations, -1 if unobserved
    const int* tile_leaf;     // [ntiles] leaf number of each tile
    double* const* leaf_ut;   // [leaf] pointer to the Ut block (row-major, ld = nop)
    const int* leaf_nop;      // [leaf]
    const double* y;          // [P]
    int ut_off[8];            // Ut row of level m's first column = ut_off[m]
    int ut_yrow;              // Ut row of y
    // KNOT: finish the node in the same launch: kInv = kernel(knots,knots) - Wk Wk^T, its Cholesky factor
    // and inverted diagonal blocks (nullptr: leave that to separate launches)
    double* Lp_out;           // [node][cw][cw]
    double* invd_out;         // [node][cwt][256]
    int* err;
    unsigned long long* stamps;   // diagnostic build (-DMRA_STAMPS) only: 16 s_memtime stamps per row tile
    int dbg;                  // what-if timing switches (results are wrong when set): 1 no Ut scatter, 2 no W stores, 4 constant instead of kernel evaluation, 8 (valid results) predictive cascade at two workgroups per CU
    int node_base;            // KNOT: node number of slot 0 of the level being factorised (error reports name node + 1)
};

// tiles staged for level m: Wk [jb][k*CWT+kt] (CWT*m*CWT), strictly-lower L (NTRI), inverted diagonals (CWT)
template <int CWT>
__device__ __forceinline__ constexpr int cascade_level_tiles(int m) { return CWT * m * CWT + CWT * (CWT - 1) / 2 + CWT; }
template <int CWT>
__device__ __forceinline__ constexpr int cascade_level_off(int m) {      // tiles of all levels < m
    return CWT * CWT * (m * (m - 1) / 2) + m * (CWT * (CWT - 1) / 2 + CWT);
}

template <int CWT>
__device__ __forceinline__ void cascade_stage_level(const CascadeArgs& ar, int m, int slot, double* __restrict__ ldsb) {
    constexpr int CW = CWT * 16;
    constexpr int NTRI = CWT * (CWT - 1) / 2;
    const CascadeLevel lv = ar.lev[m];
    const double* Wk = lv.Wk + (long)slot * CW * (m * CW);
    const double* Lm = lv.L + (long)slot * CW * CW;
    const double* inv = lv.invd + (long)slot * CWT * 256;
    const int nwk = CWT * m * CWT;
    stage_chunks<8>(ldsb, (nwk + NTRI + CWT) * 128, [&](int e) -> const double* {
        const int tile = e >> 7, chunk = e & 127, row = chunk >> 3, c2 = (chunk & 7) << 1;
        if (tile < nwk) {
            const int jb = tile / (m * CWT), kk = tile % (m * CWT);
            return Wk + (long)(jb * 16 + row) * (m * CW) + kk * 16 + c2;
        }
        if (tile < nwk + NTRI) {
            const int tt = tile - nwk;
            int jb = 1;
            while ((jb + 1) * jb / 2 <= tt) ++jb;
            const int kb = tt - jb * (jb - 1) / 2;
            return Lm + (long)(jb * 16 + row) * CW + kb * 16 + c2;
        }
        return inv + (long)(tile - nwk - NTRI) * 256 + row * 16 + c2;
    });
}

template <int CWT, int NLMAX, int DIM, int MODE, bool LAT = false>
__device__ __forceinline__ void cascade_compute_level_kx(const double* __restrict__ kx, const KernelParams& kp, int m,
                                                         const double* __restrict__ ldsb, d4 (&w)[NLMAX][CWT],
                                                         const double* xr, int prow, int q, int dbg, const d4* cvp = nullptr);
template <int CWT, int NLMAX, int DIM, int MODE>
__device__ __forceinline__ void cascade_compute_level(const CascadeArgs& ar, const KernelParams& kp, int m, int slot,
                                                      const double* __restrict__ ldsb, d4 (&w)[NLMAX][CWT],
                                                      const double* xr, int prow, int q) {
    cascade_compute_level_kx<CWT, NLMAX, DIM, MODE>(ar.lev[m].kx + (long)slot * (CWT * 16) * DIM, kp, m, ldsb, w, xr, prow, q, ar.dbg);
}
// LAT: the caller is a latency chain (k_knot_chain: one or two waves per workgroup walk a few rows down while everybody else
// waits) - no scheduling fences, the kernel evaluations may come precomputed (cvp); the row cascade (many waves per SIMD,
// register-bound) keeps the fenced form.  (Four accumulator chains instead of one buy nothing: v_mfma_f64_16x16x4 issues every
// 64 cycles dependent or not)
// cvp (LAT only): the covariances of the 16 rows with this level's knots, evaluated beforehand (cascade_cov_tiles)
template <int CWT, int NLMAX, int DIM, int MODE, bool LAT>
__device__ __forceinline__ void cascade_compute_level_kx(const double* __restrict__ kx, const KernelParams& kp, int m,
                                                         const double* __restrict__ ldsb, d4 (&w)[NLMAX][CWT],
                                                         const double* xr, int prow, int q, int dbg, const d4* cvp) {
    constexpr int NTRI = CWT * (CWT - 1) / 2;
    const d4 zero = {0, 0, 0, 0};
    const int nwk = CWT * m * CWT;
#pragma unroll
    for (int jb = 0; jb < CWT; ++jb) {
        d4 acc = zero;
        if (LAT && m > 0) {
            d4 a0 = zero, a1 = zero, a2 = zero, a3 = zero;
            const double* ab = ldsb + (jb * (m * CWT)) * 256 + prow * 16 + 4 * q;
#pragma unroll
            for (int i = 0; i < NLMAX * CWT; ++i) {
                if (i < m * CWT) {
                    const d4 a = *(const d4*)(ab + i * 256);
                    a0 = mfma16(a[0], w[i / CWT][i % CWT][0], a0);
                    a1 = mfma16(a[1], w[i / CWT][i % CWT][1], a1);
                    a2 = mfma16(a[2], w[i / CWT][i % CWT][2], a2);
                    a3 = mfma16(a[3], w[i / CWT][i % CWT][3], a3);
                }
            }
            acc = (a0 + a1) + (a2 + a3);
        } else if (m > 0) {
            // the Wk fragment of product i+1 is read while product i issues, and no further ahead (left alone the scheduler
            // hoists every LDS read of the level to the top: up to 56 registers)
            const double* ab = ldsb + (jb * (m * CWT)) * 256 + prow * 16 + 4 * q;
            d4 an = *(const d4*)ab;
#pragma unroll
            for (int i = 0; i < NLMAX * CWT; ++i) {
                if (i < m * CWT) {
                    const d4 a = an;
                    if (i + 1 < m * CWT) an = *(const d4*)(ab + (i + 1) * 256);
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc = mfma16(a[j], w[i / CWT][i % CWT][j], acc);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        d4 res;
        if (LAT && cvp) res = cvp[jb] - acc;
        else {
            double kc[4 * DIM];                     // coordinates of this lane's 4 knots (phantoms: far away)
            const double* kp4 = kx + (long)(jb * 16 + 4 * q) * DIM;
#pragma unroll
            for (int e = 0; e < 4 * DIM; e += 4) *(d4*)(kc + e) = *(const d4*)(kp4 + e);
#pragma unroll
            for (int j = 0; j < 4; ++j) res[j] = (MRA_WHATIF_BIT(dbg, 4) ? kc[j * DIM] * 1e-3 : cov_of_dist2<MODE>(kp, pair_dist2<DIM>(xr, kc + j * DIM, kp.circular))) - acc[j];
        }
        d4 upd = zero, upd1 = zero;
#pragma unroll
        for (int kb = 0; kb < CWT; ++kb) {
            if (kb < jb) {
                const d4 a = *(const d4*)(ldsb + (nwk + jb * (jb - 1) / 2 + kb) * 256 + prow * 16 + 4 * q);
                if (LAT) {
                    upd = mfma16(a[0], w[m][kb][0], upd); upd1 = mfma16(a[1], w[m][kb][1], upd1);
                    upd = mfma16(a[2], w[m][kb][2], upd); upd1 = mfma16(a[3], w[m][kb][3], upd1);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) upd = mfma16(a[j], w[m][kb][j], upd);
                }
            }
        }
        res -= LAT ? upd + upd1 : upd;
        const d4 ia = *(const d4*)(ldsb + (nwk + NTRI + jb) * 256 + prow * 16 + 4 * q);
        d4 xx = zero;
        if (LAT) {
            d4 x1 = mfma16(ia[1], res[1], zero);
            xx = mfma16(ia[0], res[0], zero);
            x1 = mfma16(ia[3], res[3], x1);
            xx = mfma16(ia[2], res[2], xx);
            xx += x1;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) xx = mfma16(ia[j], res[j], xx);
        }
        w[m][jb] = xx;
    }
}

// kernel(rows, knots of one level) for the 16 rows of a wave: CWT tiles in the layout cascade_compute_level_kx consumes
template <int CWT, int DIM, int MODE>
__device__ __forceinline__ void cascade_cov_tiles(const double* __restrict__ kx, const KernelParams& kp, const double* xr, int q, d4* cv) {
#pragma unroll
    for (int jb = 0; jb < CWT; ++jb) {
        double kc[4 * DIM];
        const double* kp4 = kx + (long)(jb * 16 + 4 * q) * DIM;
#pragma unroll
        for (int e = 0; e < 4 * DIM; e += 4) *(d4*)(kc + e) = *(const d4*)(kp4 + e);
#pragma unroll
        for (int j = 0; j < 4; ++j) cv[jb][j] = cov_of_dist2<MODE>(kp, pair_dist2<DIM>(xr, kc + j * DIM, kp.circular));
    }
}

template <int CWT, int NLMAX>
__device__ __forceinline__ void cascade_outputs(const CascadeArgs& ar, const int* chain, long t, long myrow, bool phantom_row,
                                                d4 (&w)[NLMAX][CWT], int r, int q, double ssq) {
    constexpr int CW = CWT * 16;
    const d4 zero = {0, 0, 0, 0};
    if (ar.knot_mode) {
        const int mo = ar.mlast + 1;
        const int slotk = ar.tile_chain[t * 8 + mo];       // the tile's own node (a workgroup holds a family of siblings)
        const int krow = ar.tile_knot0[t] + r;
        double* o = ar.Wk_out + ((long)slotk * CW + krow) * (mo * CW);
#pragma unroll
        for (int k = 0; k < NLMAX; ++k) {
            if (k < mo) {
#pragma unroll
                for (int kt = 0; kt < CWT; ++kt) {
                    d4 v = phantom_row ? zero : w[k][kt];
                    *(d4*)(o + (k * CWT + kt) * 16 + 4 * q) = v;
                }
            }
        }
        return;
    }
    // row mode: the W tiles and the Ut columns of each level were written by cascade_output_level right
    // after the level was computed; what is left is the residual prior variance and the y entry of Ut
    if (ar.var_out) {
        ssq += __shfl_xor(ssq, 16, 64);
        ssq += __shfl_xor(ssq, 32, 64);
        if (q == 0 && !phantom_row) ar.var_out[myrow] = ar.cov0 - ssq;
    }
    if (ar.ycol >= 0 && !phantom_row) { // the 16-wide y block of W: y (0 where missing) in its first column
        const double yy = ar.y[myrow];
        const d4 yt = {(q == 0 && isfinite(yy)) ? yy : 0.0, 0.0, 0.0, 0.0};
        int qo = q;
        asm volatile("" : "+v"(qo));     // opaque: the lane part of the address is rebuilt here, not hoisted out of the tile loop and spilled
        *(d4*)(ar.W + myrow * ar.ldw + ar.ycol + 4 * qo) = yt;
    }
    if (ar.obs_pos && !phantom_row) {
        const int op = ar.obs_pos[myrow];
        if (op >= 0 && q == 0) {
            const int lf = ar.tile_leaf[t];
            gst(ar.leaf_ut[lf] + (long)ar.ut_yrow * ar.leaf_nop[lf] + op, ar.y[myrow]);
        }
    }
}

// Row mode, level m just computed: its W tiles go out now (stores spread over the whole cascade drain while
// the next levels are on the MFMA pipe; issued in one burst at the end they cost ~0.4 ms at C3), the
// observed rows also into the leaf's Ut (transposed).  Returns the row's sum of squares of the level.
template <int CWT, int NLMAX>
__device__ __forceinline__ double cascade_output_level(const CascadeArgs& ar, int m, long t, long myrow, int op,
                                                       const d4 (&w)[NLMAX][CWT], int q) {
    const bool phantom_row = op == -2;                     // padding row of a gathered (likelihood-only) tile
    double* o = ar.W + myrow * ar.ldw;
    double ssq = 0.0;
#pragma unroll
    for (int jb = 0; jb < CWT; ++jb) {
        const d4 v = w[m][jb];
        if (!MRA_WHATIF_BIT(ar.dbg, 2) && !phantom_row) *(d4*)(o + ar.coff[m] + jb * 16 + 4 * q) = v;
        ssq += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    if (op >= 0 && !MRA_WHATIF_BIT(ar.dbg, 1)) {
        const int lf = ar.tile_leaf[t];
        double* ut = ar.leaf_ut[lf];
        const long nop = ar.leaf_nop[lf];
#pragma unroll
        for (int jb = 0; jb < CWT; ++jb) {
            const d4 v = w[m][jb];
#pragma unroll
            for (int j = 0; j < 4; ++j) gst(ut + (long)(ar.ut_off[m] + jb * 16 + 4 * q + j) * nop + op, v[j]);
        }
    }
    return ssq;
}

// In-kernel stamps (cdna_hip_programming.md section 7): a diagnostic build records the shader clock at the phase boundaries
// of every row tile into a buffer of its own; the product build compiles none of it.
#ifdef MRA_STAMPS
#define MRA_STAMP(slot) do { if (ar.stamps && !ar.knot_mode && lane == 0) ar.stamps[(t) * 16 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define MRA_STAMP(slot) do { } while (0)
#endif
// One workgroup = the row tiles of one leaf (FULL) or the knot tiles of one node (KNOT): they share
// the same node on every level.  STAGE_ALL: the operands of ALL levels (Wk, strictly-lower tiles of L,
// inverted diagonal blocks) are staged in LDS once, then every wave walks its tiles down the levels
// with no further barrier.  Otherwise (operands do not fit in 160 KB) they are staged level by level
// and each wave owns exactly one tile.
template <int CWT, int NLMAX, int DIM, int MODE, bool STAGE_ALL>
__global__ __launch_bounds__(512) void k_prior_cascade(CascadeArgs ar, KernelParams kp) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    const int nwave = blockDim.x >> 6;
    const long t0 = ar.wg_tile0[blockIdx.x];
    const int nt_wg = ar.wg_ntiles[blockIdx.x];
    const int prow = pi16(r);
    const int* chain = ar.tile_chain + t0 * 8;          // identical for every tile of the workgroup
    if (STAGE_ALL) {
#ifdef MRA_STAMPS
        const unsigned long long t_wg0 = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll
        for (int m = 0; m < NLMAX; ++m)
            if (m <= ar.mlast) cascade_stage_level<CWT>(ar, m, chain[m], lds + cascade_level_off<CWT>(m) * 256);
        __syncthreads();
        for (int tw = wave; tw < nt_wg; tw += nwave) {
            const long t = t0 + tw;
#ifdef MRA_STAMPS
            if (ar.stamps && !ar.knot_mode && lane == 0) ar.stamps[t * 16 + 0] = t_wg0;
#endif
            MRA_STAMP(1);
            long myrow;
            bool phantom_row = false;
            if (ar.knot_mode || ar.row_gather) {
                const int rr = (ar.knot_mode ? ar.tile_rows : ar.row_gather)[t * 16 + r];
                phantom_row = rr < 0;
                myrow = phantom_row ? 0 : rr;
            } else {
                myrow = ar.tile_row0[t] + r;
            }
            double xr[DIM];
#pragma unroll
            for (int c = 0; c < DIM; ++c) xr[c] = ar.X[myrow * DIM + c];
            d4 w[NLMAX][CWT];
            double ssq = 0.0;
            // (row mode: -2 marks a padding row of a gathered tile - its outputs are skipped - without a second live register)
            const int op = ar.knot_mode ? -1 : (phantom_row ? -2 : (ar.obs_pos ? ar.obs_pos[myrow] : -1));
#pragma unroll
            for (int m = 0; m < NLMAX; ++m)
                if (m <= ar.mlast) {
                    cascade_compute_level<CWT, NLMAX, DIM, MODE>(ar, kp, m, chain[m], lds + cascade_level_off<CWT>(m) * 256, w, xr, prow, q);
                    MRA_STAMP(2 + 2 * m);
                    if (!ar.knot_mode) ssq += cascade_output_level<CWT, NLMAX>(ar, m, t, myrow, op, w, q);
                    MRA_STAMP(3 + 2 * m);
                }
            cascade_outputs<CWT, NLMAX>(ar, chain, t, myrow, ar.knot_mode ? phantom_row : op == -2, w, r, q, ssq);
            MRA_STAMP(15);
        }
    } else {
        const bool active = wave < nt_wg;
        const long t = t0 + (active ? wave : 0);
        long myrow;
        bool phantom_row = false;
        if (ar.knot_mode || ar.row_gather) {
            const int rr = (ar.knot_mode ? ar.tile_rows : ar.row_gather)[t * 16 + r];
            phantom_row = rr < 0;
            myrow = phantom_row ? 0 : rr;
        } else {
            myrow = ar.tile_row0[t] + r;
        }
        double xr[DIM];
#pragma unroll
        for (int c = 0; c < DIM; ++c) xr[c] = ar.X[myrow * DIM + c];
        d4 w[NLMAX][CWT];
        double ssq = 0.0;
        const int op = ar.knot_mode ? -1 : (phantom_row ? -2 : (ar.obs_pos ? ar.obs_pos[myrow] : -1));
#pragma unroll
        for (int m = 0; m < NLMAX; ++m) {
            if (m <= ar.mlast) {
                __syncthreads();
                cascade_stage_level<CWT>(ar, m, chain[m], lds);
                __syncthreads();
                if (active) {
                    cascade_compute_level<CWT, NLMAX, DIM, MODE>(ar, kp, m, chain[m], lds, w, xr, prow, q);
                    if (!ar.knot_mode) ssq += cascade_output_level<CWT, NLMAX>(ar, m, t, myrow, op, w, q);
                }
            }
        }
        if (active) cascade_outputs<CWT, NLMAX>(ar, chain, t, myrow, ar.knot_mode ? phantom_row : op == -2, w, r, q, ssq);
    }
    if (ar.knot_mode && ar.Lp_out) {
        // ---- kInv, factor and inverted diagonal blocks of every node of the workgroup, still in this launch
        constexpr int CW = CWT * 16;
        constexpr int NT2 = CWT * (CWT + 1) / 2;
        const int mo = ar.mlast + 1;
        const int Kw = mo * CW;
        const int nnode = nt_wg / CWT;
        const d4 zero = {0, 0, 0, 0};
        __syncthreads();                                  // every wave's Wk rows are in global memory
        for (int w = wave; w < nnode * NT2; w += nwave) {
            const int ni = w / NT2, idx = w - ni * NT2;
            const int slotk = ar.tile_chain[(t0 + (long)ni * CWT) * 8 + mo];
            const double* Wk = ar.Wk_out + (long)slotk * CW * Kw;
            const double* kx = ar.lev[mo].kx + (long)slotk * CW * DIM;
            double* Lp = ar.Lp_out + (long)slotk * CW * CW;
            int ib = 0;
            while ((ib + 1) * (ib + 2) / 2 <= idx) ++ib;
            const int jb = idx - ib * (ib + 1) / 2;
            d4 acc = zero;
            for (int kk = 0; kk < Kw; kk += 16) {
                const d4 a = *(const d4*)(Wk + (long)(jb * 16 + prow) * Kw + kk + 4 * q);
                const d4 b = *(const d4*)(Wk + (long)(ib * 16 + r) * Kw + kk + 4 * q);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc = mfma16(a[j], b[j], acc);
            }
            double xi[DIM];
#pragma unroll
            for (int c = 0; c < DIM; ++c) xi[c] = kx[(long)(ib * 16 + r) * DIM + c];
            d4 res;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                res[j] = cov_of_dist2<MODE>(kp, pair_dist2<DIM>(xi, kx + (long)(jb * 16 + 4 * q + j) * DIM, kp.circular)) - acc[j];
            *(d4*)(Lp + (long)(ib * 16 + r) * CW + jb * 16 + 4 * q) = res;
        }
        __syncthreads();
        // one wave per node; the staged operands are dead by now: the dynamic LDS is reused as per-wave scratch
        // (272 doubles tile + exchange column of chol16_inv, 256 doubles inverse block)
        for (int ni = wave; ni < nnode; ni += nwave) {
            const int slotk = ar.tile_chain[(t0 + (long)ni * CWT) * 8 + mo];
            double* s_sd = lds + wave * 576;
            double* s_si = s_sd + 288;
            bool bad = false;
            chol_wave_body(ar.Lp_out + (long)slotk * CW * CW, CW, CWT, ar.invd_out + (long)slotk * CWT * 256, s_sd, s_si, lane, bad);
            if (bad && lane == 0) atomicMax(ar.err, ar.node_base + slotk + 1);
        }
    }
}

// ------------------------------------------------------------------------------------------------
//  Knot pass of ALL levels in one launch.  The per-level launches of the knot mode above form a chain of NL dependent
//  kernels, each a few microseconds of work behind global round trips (stage the ancestors' operands, write Wk,
//  read it back for kInv, factorise in global memory): 0.22 ms at C3 however few nodes a level has - a fixed cost that
//  does not shrink when the tree is sharded.  Here one workgroup per node of the LAST non-leaf level walks down its own
//  ancestor chain, level by level, recomputing every ancestor redundantly (bit-identical in every workgroup) with the
//  operand images kept in LDS from one level to the next; nothing is exchanged between workgroups and nothing is read
//  back from global memory.  Results go to the level arrays from the first workgroup below each node ("owner").
//  LDS: images of levels 0 .. NL-2 in the row cascade's layout (cascade_level_off), then NT2 + CWT padded tiles of
//  factorisation scratch.
// ------------------------------------------------------------------------------------------------
struct KnotChainArgs {
    CascadeLevel lev[8];          // level arrays: Wk / L / invd (out)
    const int* chain;             // [workgroup][8] node slot per level
    const int* ownmask;           // [workgroup] bit m: this workgroup writes the results of its level-m ancestor (the first workgroup below it)
    const double* knots;          // [workgroup][level][CW * (DIM + 1)]: the knots' coordinates, then 1.0 (real knot) / 0.0 (phantom) per knot -
                                  // everything the chain needs from outside, packed per workgroup by the host (mra_plan_set_locs)
    int nl;
    int node_base[8];             // node number of slot 0 per level (error reports)
    int* err;
    unsigned long long* stamps;   // diagnostic build (-DMRA_STAMPS) only: 64 wall-clock stamps (10 ns) per workgroup
};
#ifdef MRA_STAMPS
#define MRA_KSTAMP() do { if (ka.stamps && threadIdx.x == 0) ka.stamps[(long)blockIdx.x * 64 + kst] = __builtin_amdgcn_s_memrealtime(); ++kst; } while (0)
#else
#define MRA_KSTAMP() do { } while (0)
#endif

template <int CWT, int NLMAX, int DIM, int MODE>
__global__ __launch_bounds__(256, 1) void k_knot_chain(KnotChainArgs ka, KernelParams kp) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int CW = CWT * 16;
    constexpr int NTRI = CWT * (CWT - 1) / 2;
    constexpr int NT2 = CWT * (CWT + 1) / 2;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    const int nwave = blockDim.x >> 6;
    const int prow = pi16(r);
    const d4 zero = {0, 0, 0, 0};
    const int* chain = ka.chain + (long)blockIdx.x * 8;
    const int nl = ka.nl;
    double* const fs = lds + (long)cascade_level_off<CWT>(nl - 1) * 256;     // NT2 lower tiles (FT_SZ each), then CWT inverse tiles
    double* const finv = fs + NT2 * FT_SZ;
    double* const kxs = finv + CWT * FT_SZ;                                  // [level][CW][DIM] knot coordinates of this chain
#ifdef MRA_STAMPS
    int kst = 0;
#endif
    MRA_KSTAMP();
    // ---- everything the chain needs from outside in ONE round trip (addresses depend on the workgroup number only): node slots, owner
    // bits and the packed knots of every level, into LDS.  (Fetched where they were used - slot -> knot rows -> coordinates, and the
    // coarser levels' knots in front of every kernel evaluation - these were 2-3 dependent global round trips per level.)
    constexpr int KREC = CW * (DIM + 1);
    int slots[NLMAX];
#pragma unroll
    for (int k = 0; k < NLMAX; ++k) slots[k] = k < nl ? chain[k] : 0;
    const int ownbits = ka.ownmask[blockIdx.x];
    {
        const double* src = ka.knots + (long)blockIdx.x * nl * KREC;
        for (int e = threadIdx.x; e < nl * KREC; e += blockDim.x) kxs[e] = src[e];
    }
    __syncthreads();
    MRA_KSTAMP();
    // The knot rows of level m walk down the levels k < m (residual against the coarser bases, whitening by L_k): only the LAST of
    // these steps needs what the previous iteration has just factorised.  CWT == 2: the walker waves (0, 1) do the steps k < m - 1 of
    // the NEXT level's rows, and the kernel evaluation of its last step, while the last wave factorises the current level (a
    // 3.4 us diagonal block, a row solve, an update, another diagonal block: 8.8 us in which they used to wait); after the barrier
    // they finish with one step.  Other widths (every wave works on the factorisation) walk in one go at the top of the iteration.
    d4 w[NLMAX][CWT];                                           // the walker's rows: whitened tiles of the levels done so far
    d4 cvl[CWT];                                                // kernel(rows, knots of the level of the last step)
    const int fwave = CWT == 2 ? nwave - 1 : 0;
    auto walk_early = [&](int lvl) {                            // rows of level lvl: steps k < lvl - 1, then the kernel tiles of step lvl - 1
        double xr[DIM];                                         // this lane's knot row = knot wave * 16 + r of the level's node
#pragma unroll
        for (int c = 0; c < DIM; ++c) xr[c] = kxs[lvl * KREC + (wave * 16 + r) * DIM + c];
#pragma unroll
        for (int k = 0; k < NLMAX; ++k)
            if (k < lvl - 1) cascade_compute_level_kx<CWT, NLMAX, DIM, MODE, true>(kxs + k * KREC, kp, k,
                                                                                  lds + (long)cascade_level_off<CWT>(k) * 256, w, xr, prow, q, 0, nullptr);
#pragma unroll
        for (int k = 0; k < NLMAX; ++k)
            if (k == lvl - 1) cascade_cov_tiles<CWT, DIM, MODE>(kxs + k * KREC, kp, xr, q, cvl);
    };
    auto walk_finish = [&](int lvl) {                           // step lvl - 1 (its level's image is in LDS by now)
        double xr[DIM];
#pragma unroll
        for (int c = 0; c < DIM; ++c) xr[c] = kxs[lvl * KREC + (wave * 16 + r) * DIM + c];
#pragma unroll
        for (int k = 0; k < NLMAX; ++k)
            if (k == lvl - 1) cascade_compute_level_kx<CWT, NLMAX, DIM, MODE, true>(kxs + k * KREC, kp, k,
                                                                                   lds + (long)cascade_level_off<CWT>(k) * 256, w, xr, prow, q, 0, cvl);
    };
#pragma unroll 1
    for (int m = 0; m < nl; ++m) {
        int slot = 0;
#pragma unroll
        for (int k = 0; k < NLMAX; ++k) if (k == m) slot = slots[k];
        const bool own = (ownbits >> m) & 1;
        const bool last = m == nl - 1;
        double* const img = lds + (long)cascade_level_off<CWT>(m) * 256;      // image of level m (unused for the last level)
        const int Kw = m * CW;
        double* const Wk_g = const_cast<double*>(ka.lev[m].Wk) + (long)slot * CW * Kw;
        const double* kxn = kxs + m * KREC;
        // ---- 1. the knot rows' last step, then the Wk tiles out.  The last level has no image of its own (LDS is full), but the images
        //         of the first levels are dead by then: its Wk tiles go there when they end before the image its walkers still read,
        //         and step 2 takes them from LDS instead of making the round trip through global memory (8.7 -> 3 us)
        const bool walker = wave < CWT && m > 0;
        const bool wk_lds = !last || (m >= 1 && CWT * m * CWT <= cascade_level_off<CWT>(m - 1));
        double* const wimg = last ? lds : img;
        if (walker) {
            if (CWT != 2) walk_early(m);
            walk_finish(m);
        }
        if (last && CWT != 2) __syncthreads();                  // every walker is done with the early images
        if (walker) {
            const bool phantom_row = kxn[CW * DIM + wave * 16 + r] == 0.0;
#pragma unroll
            for (int k = 0; k < NLMAX; ++k) {
                if (k < m) {
#pragma unroll
                    for (int kt = 0; kt < CWT; ++kt) {
                        const d4 v = phantom_row ? zero : w[k][kt];
                        if (wk_lds) *(d4*)(wimg + (long)(wave * (m * CWT) + k * CWT + kt) * 256 + r * 16 + 4 * q) = v;
                        if (own || last) *(d4*)(Wk_g + (long)(wave * 16 + r) * Kw + (k * CWT + kt) * 16 + 4 * q) = v;
                    }
                }
            }
        }
        __syncthreads();
        MRA_KSTAMP();
        // ---- 2. kInv = kernel(knots, knots) - Wk Wk^T, lower tiles into the factorisation scratch
        for (int idx = wave; idx < NT2; idx += nwave) {
            int ib = 0;
            while ((ib + 1) * (ib + 2) / 2 <= idx) ++ib;
            const int jb = idx - ib * (ib + 1) / 2;
            d4 acc = zero, acc1 = zero;
            for (int kk = 0; kk < m * CWT; ++kk) {
                d4 a, b;
                if (wk_lds) {
                    a = *(const d4*)(wimg + (long)(jb * (m * CWT) + kk) * 256 + prow * 16 + 4 * q);
                    b = *(const d4*)(wimg + (long)(ib * (m * CWT) + kk) * 256 + r * 16 + 4 * q);
                } else {
                    a = *(const d4*)(Wk_g + (long)(jb * 16 + prow) * Kw + kk * 16 + 4 * q);
                    b = *(const d4*)(Wk_g + (long)(ib * 16 + r) * Kw + kk * 16 + 4 * q);
                }
                acc = mfma16(a[0], b[0], acc); acc1 = mfma16(a[1], b[1], acc1);
                acc = mfma16(a[2], b[2], acc); acc1 = mfma16(a[3], b[3], acc1);
            }
            acc += acc1;
            double xi[DIM];
#pragma unroll
            for (int c = 0; c < DIM; ++c) xi[c] = kxn[(long)(ib * 16 + r) * DIM + c];
            d4 res;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                res[j] = cov_of_dist2<MODE>(kp, pair_dist2<DIM>(xi, kxn + (long)(jb * 16 + 4 * q + j) * DIM, kp.circular)) - acc[j];
            *(d4*)(fs + (long)idx * FT_SZ + r * FT_LD + 4 * q) = res;
        }
        __syncthreads();
        MRA_KSTAMP();
        // ---- 3. Cholesky of the CW x CW block in LDS (left-looking over column tiles, diagonal blocks on wave 0)
        if (CWT == 2) {
            // two column tiles: the whole chain (factor, one row solve, one update, factor) on one wave without workgroup barriers
            if (wave < CWT && m + 1 < nl) walk_early(m + 1);
            if (wave == fwave) {
                bool bad = false;
#pragma unroll
                for (int jb = 0; jb < 2; ++jb) {
                    double* dt = fs + (long)(jb * (jb + 1) / 2 + jb) * FT_SZ;
#ifdef MRA_ATOM_OLD
                    double a[16], mi[16];
#pragma unroll
                    for (int k = 0; k < 16; ++k) a[k] = (k <= r) ? dt[r * FT_LD + k] : 0.0;
                    chol16_inv(a, mi, r, bad, nullptr);
                    if (lane < 16) {
#pragma unroll
                        for (int k = 0; k < 16; k += 2) {
                            *(d2*)(dt + lane * FT_LD + k) = d2{a[k], a[k + 1]};
                            *(d2*)(finv + jb * FT_SZ + lane * FT_LD + k) = d2{mi[k], mi[k + 1]};
                        }
                    }
#else
                    chol16_ldl<8>(dt, FT_LD, dt, FT_LD, finv + jb * FT_SZ, FT_LD, lane, bad);       // (<8>: the log-determinant is not wanted here)
#endif
                    if (jb == 0) {
                        lds_wave_sync();
                        const d4 ia = *(const d4*)(finv + prow * FT_LD + 4 * q);
                        d4* tp = (d4*)(fs + (long)1 * FT_SZ + r * FT_LD + 4 * q);
                        const d4 b = *tp;
                        d4 x0 = mfma16(ia[0], b[0], zero), x1 = mfma16(ia[1], b[1], zero);
                        x0 = mfma16(ia[2], b[2], x0); x1 = mfma16(ia[3], b[3], x1);
                        *tp = x0 + x1;
                        lds_wave_sync();
                        const d4 xa = *(const d4*)(fs + (long)1 * FT_SZ + prow * FT_LD + 4 * q);
                        const d4 xb = x0 + x1;
                        d4 u0 = mfma16(xa[0], xb[0], zero), u1 = mfma16(xa[1], xb[1], zero);
                        u0 = mfma16(xa[2], xb[2], u0); u1 = mfma16(xa[3], xb[3], u1);
                        d4* dp = (d4*)(fs + (long)2 * FT_SZ + r * FT_LD + 4 * q);
                        *dp = *dp - (u0 + u1);
                        lds_wave_sync();
                    }
                }
                if (bad && lane == 0) atomicMax(ka.err, ka.node_base[m] + slot + 1);
            }
            __syncthreads();
        } else
        for (int jb = 0; jb < CWT; ++jb) {
            if (jb > 0) {
                for (int ib = jb + wave; ib < CWT; ib += nwave) {
                    d4 u = zero;
                    for (int kb = 0; kb < jb; ++kb) {
                        const d4 a = *(const d4*)(fs + (long)(jb * (jb + 1) / 2 + kb) * FT_SZ + prow * FT_LD + 4 * q);
                        const d4 b = *(const d4*)(fs + (long)(ib * (ib + 1) / 2 + kb) * FT_SZ + r * FT_LD + 4 * q);
#pragma unroll
                        for (int j = 0; j < 4; ++j) u = mfma16(a[j], b[j], u);
                    }
                    d4* tp = (d4*)(fs + (long)(ib * (ib + 1) / 2 + jb) * FT_SZ + r * FT_LD + 4 * q);
                    *tp = *tp - u;
                }
                __syncthreads();
            }
            if (wave == 0) {
                double* dt = fs + (long)(jb * (jb + 1) / 2 + jb) * FT_SZ;
                bool bad = false;
#ifdef MRA_ATOM_OLD
                double a[16], mi[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) a[k] = (k <= r) ? dt[r * FT_LD + k] : 0.0;
                chol16_inv(a, mi, r, bad, nullptr);
                if (lane < 16) {
#pragma unroll
                    for (int k = 0; k < 16; k += 2) {
                        *(d2*)(dt + lane * FT_LD + k) = d2{a[k], a[k + 1]};
                        *(d2*)(finv + jb * FT_SZ + lane * FT_LD + k) = d2{mi[k], mi[k + 1]};
                    }
                }
#else
                chol16_ldl<8>(dt, FT_LD, dt, FT_LD, finv + jb * FT_SZ, FT_LD, lane, bad);
#endif
                if (bad && lane == 0) atomicMax(ka.err, ka.node_base[m] + slot + 1);
            }
            __syncthreads();
            {
                const d4 ia = *(const d4*)(finv + jb * FT_SZ + prow * FT_LD + 4 * q);
                for (int ib = jb + 1 + wave; ib < CWT; ib += nwave) {
                    d4* tp = (d4*)(fs + (long)(ib * (ib + 1) / 2 + jb) * FT_SZ + r * FT_LD + 4 * q);
                    const d4 b = *tp;
                    d4 x = zero;
#pragma unroll
                    for (int j = 0; j < 4; ++j) x = mfma16(ia[j], b[j], x);
                    *tp = x;
                }
            }
            __syncthreads();
        }
        MRA_KSTAMP();
        // ---- 4. factor into the level's image (strictly-lower tiles, inverted diagonal blocks) and, from the owner, into
        //         the level arrays (row-major CW x CW factor with zeros above the diagonal, CWT inverse blocks)
        for (int idx = wave; idx < NT2 + CWT; idx += nwave) {
            if (idx < NT2) {
                int ib = 0;
                while ((ib + 1) * (ib + 2) / 2 <= idx) ++ib;
                const int jb = idx - ib * (ib + 1) / 2;
                const d4 v = *(const d4*)(fs + (long)idx * FT_SZ + r * FT_LD + 4 * q);
                if (!last && ib > jb) *(d4*)(img + (long)(CWT * m * CWT + ib * (ib - 1) / 2 + jb) * 256 + r * 16 + 4 * q) = v;
                if (own || last) {
                    double* Lg = const_cast<double*>(ka.lev[m].L) + (long)slot * CW * CW;
                    *(d4*)(Lg + (long)(ib * 16 + r) * CW + jb * 16 + 4 * q) = v;
                    if (ib > jb) *(d4*)(Lg + (long)(jb * 16 + r) * CW + ib * 16 + 4 * q) = zero;
                }
            } else {
                const int jb = idx - NT2;
                const d4 v = *(const d4*)(finv + jb * FT_SZ + r * FT_LD + 4 * q);
                if (!last) *(d4*)(img + (long)(CWT * m * CWT + NTRI + jb) * 256 + r * 16 + 4 * q) = v;
                if (own || last) *(d4*)(const_cast<double*>(ka.lev[m].invd) + ((long)slot * CWT + jb) * 256 + r * 16 + 4 * q) = v;
            }
        }
        __syncthreads();
        MRA_KSTAMP();
    }
}

// ------------------------------------------------------------------------------------------------
//  Fused predictive cascade (regular trees), bottom-up over the non-leaf levels, one pass over W:
//      X_m = W~^m[i] Lt_m^{-T};  var += |X_m|^2;  [W~^{<m} | ycol][i] -= X_m Zt_m^T      (MRANode.py:486-511)
//  with all coarser tiles of the 16 rows in registers.  Inputs: W after the leaf update, var after the
//  leaf part (prior residual variance minus |Tt|^2, may be slightly negative -> clamped here).
//  Outputs: mean (= -ycol) and var only; W is not written back.
// ------------------------------------------------------------------------------------------------
struct PredLevel {
    const double* F;      // [node] factorised fronts of this level (Lt, Zt in place): nf rows of ld doubles each, `stride` doubles per node
    const double* invF;   // [node][cwt][256]
    int nf;
    int ld;               // = nf, or the block width when only the panel columns [Lt ; Zt] of the level's fronts are kept
    long stride;          // = nf * ld
};
struct PredArgs {
    PredLevel lev[8];
    PredLevel deep;           // = lev[nl - 1] (a run-time index into lev[] would put the whole argument block in scratch)
    const double* W;
    double* mean;
    double* var;
    long ldw;
    int coff[8];
    int ycol;                 // first column of the y block in W
    const long* tile_row0;
    const int* tile_chain;
    const long* wg_tile0;
    const int* wg_ntiles;
    long n_wg;
    int nl;                   // number of non-leaf levels
    // UPD: the leaf update W[S, anc | y] -= Tt Ut^T is applied to the row tile in registers before the levels are walked
    // (the updated W is never written: 1.7 GB of stores and 1.7 GB of re-reads less at C3)
    const int* tile_leaf;     // [tile] leaf number
    const int* wg_leaf;       // [workgroup] leaf number of the workgroup's tiles (they share one): known one round trip before tile_leaf[t0]
    double* const* leaf_ut;   // [leaf] solved Ut block (na x nop), followed by the leaf's Tt rows (N_j x nop)
    const int* leaf_nop;      // [leaf]
    const long* leaf_row0;    // [leaf] first padded row
    const unsigned char* leaf_upd;   // [leaf] 1: update here (nop <= 128); 0: a separate product has done it
    int na;                   // rows of Ut = ancestors + y block
    unsigned long long* stamps;   // diagnostic build (-DMRA_STAMPS) only: 16 s_memtime stamps per row tile
};
#ifdef MRA_STAMPS
#define MRA_PSTAMP(slot) do { if (ar.stamps && active && lane == 0) ar.stamps[(t) * 16 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#define MRA_PSTAMP_WALL(slot) do { if (ar.stamps && active && lane == 0) ar.stamps[(t) * 16 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define MRA_PSTAMP(slot) do { } while (0)
#define MRA_PSTAMP_WALL(slot) do { } while (0)
#endif

// ---- staging of a level's operand tiles in TWO HALVES, straight into LDS ----------------------------------------------------
// The LDS image of level m lists its tiles in the order the products consume them:
//     slot 0 .. NTRI-1            strictly-lower tiles of Lt            (solve)
//     slot NTRI .. NTRI+CWT-1     inverted diagonal blocks              (solve)
//     then CWT tiles [jb] per front row block, ancestors first: k = 0 .. m-1, kt = 0 .. CWT-1  (front row block a = (m-1-k) CWT + kt,
//     the level updates w[k][kt]), then the y row block (a = m CWT).
// Half A = the solve tiles + the row blocks of the first kA = m / 2 ancestor levels, half B = the rest (never empty: the y row
// block at least).  Half A lives at the start of the LDS image, half B at a fixed offset (the size of the deepest level's half A).
// The pieces go from global memory straight into LDS (gld_lds16): half B of level m is requested when everybody has passed the
// barrier that publishes half A (the region is free: level m+1's half-B products are done) and flies during the solve and the
// half-A products; half A of level m-1 is requested behind the barrier that publishes half B and flies during the half-B products.
// TWO barriers per level, no staging registers, no ds_write.  (Rounds 2-4 staged through registers: a whole level cost the
// predictive cascade 84 B / lane of scratch at three workgroups per CU, half a level 7 register pairs and a third barrier.)
template <int CWT> __host__ __device__ __forceinline__ constexpr int pred_tiles_a(int m, int ka) { return CWT * (CWT - 1) / 2 + CWT + ka * CWT * CWT; }
template <int CWT> __host__ __device__ __forceinline__ constexpr int pred_tiles_all(int m) { return CWT * (CWT - 1) / 2 + CWT + (m * CWT + 1) * CWT; }
// steps (of TPS tiles) of the largest half of any level
template <int CWT, int NLMAX, int NTH> __host__ __device__ __forceinline__ constexpr int pred_half_steps() {
    int mx = 0;
    for (int m = 0; m < NLMAX; ++m) {
        const int ta = pred_tiles_a<CWT>(m, m / 2), tb = pred_tiles_all<CWT>(m) - ta;
        mx = ta > mx ? ta : mx;
        mx = tb > mx ? tb : mx;
    }
    return (mx * 128 + NTH - 1) / NTH;
}
// tiles of level m in half A
template <int CWT> __host__ __device__ __forceinline__ constexpr int pred_sa(int m) { return pred_tiles_a<CWT>(m, m / 2); }
// request LDS slots [S0, S0 + CNT) of level ms (operands of node chain[ms]) into the tiles T0 .. of the LDS image.  A tile is 128
// pieces of 16 bytes, a wave moves half a tile per step, so WHICH tile a wave moves in step i is wave-uniform (tl = i TPS + wv): the
// tile decoding runs on the scalar unit, and the global address is a scalar base plus ONE per-thread offset (row and column of the
// piece inside its tile: thread constants; times the front's row stride for the tiles that come out of F).
#define MRA_PRED_LOAD_RANGE(ms, lvl, S0, CNT, T0) do { \
                const int ms_ = (ms); \
                const PredLevel ls = (lvl); \
                const int slot_s = chain[ms_]; \
                const double* Fs = ls.F + (long)slot_s * ls.stride; \
                const double* invs = ls.invF + (long)slot_s * CWT * 256; \
                const int cnt_ = (CNT), s0_ = (S0); \
                const unsigned vz = (unsigned)(prow_c * ls.ld + pcol_c) * 8u, vi = (unsigned)(prow_c * 16 + pcol_c) * 8u; \
                char* const lb_ = (char*)lds + (long)(T0) * 2048 + whalf * 1024; \
_Pragma("unroll") \
                for (int i = 0; i < PH; ++i) { \
                    const int tl = i * TPS + wv; \
                    if (i * TPS < cnt_ && tl < cnt_) { \
                        const int tile = s0_ + tl; \
                        const bool is_tri = tile < NTRI, is_inv = !is_tri && tile < NTRI + CWT; \
                        int jbt = 1; \
_Pragma("unroll") \
                        for (int c = 1; c < CWT - 1; ++c) jbt += (tile >= c * (c + 1) / 2) ? 1 : 0; \
                        const int kbt = tile - jbt * (jbt - 1) / 2; \
                        const int zt = tile >= NTRI + CWT ? tile - NTRI - CWT : 0, g = zt / CWT, jbz = zt % CWT;   /* g: consumption order of the row blocks */ \
                        const int az = g < ms_ * CWT ? (ms_ - 1 - g / CWT) * CWT + g % CWT : ms_ * CWT; \
                        const long rowoff = is_tri ? jbt * 16 : CW + az * 16; \
                        const int coloff = is_tri ? kbt * 16 : jbz * 16; \
                        const char* sbF = (const char*)(Fs + rowoff * ls.ld + coloff); \
                        const char* sbI = (const char*)(invs + (long)(is_inv ? tile - NTRI : 0) * 256); \
                        if (GLDS & 1) gld_lds16((is_inv ? sbI : sbF) + (is_inv ? vi : vz), lb_ + (long)tl * 2048); \
                        else pre[i] = *(const d2*)((is_inv ? sbI : sbF) + (is_inv ? vi : vz)); \
                    } \
                } \
} while (0)
// the requested range has landed: DMA requests are waited for; the register-staged form (GLDS = false: one or two workgroups per
// CU, where the LDS DMA measured slower - 0.132 against 0.118 ms for the cascade of an 8-way shard of C3) writes its pieces now.
// Either way the barrier that follows publishes the range.
#define MRA_PRED_LAND_RANGE(CNT, T0) do { \
                if (GLDS & 1) mra_wait_vm0(); \
                else { \
                    const int cnt_ = (CNT); \
                    char* const lb_ = (char*)lds + (long)(T0) * 2048 + whalf * 1024 + (long)lane * 16; \
_Pragma("unroll") \
                    for (int i = 0; i < PH; ++i) { \
                        const int tl = i * TPS + wv; \
                        if (i * TPS < cnt_ && tl < cnt_) *(d2*)(lb_ + (long)tl * 2048) = pre[i]; \
                    } \
                } \
} while (0)
// GLDS: bit 0 - the level operands by LDS DMA (else through registers), bit 1 - the Ut chunks of the leaf update
template <int CWT, int NLMAX, int WPW, bool UPD, int MINB, int GLDS = 3>
__global__ __launch_bounds__(64 * WPW, MINB) void k_predict_cascade(PredArgs ar) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int CW = CWT * 16;
    constexpr int NTRI = CWT * (CWT - 1) / 2;
    constexpr int NTH = 64 * WPW;
    // staging steps of HALF a level's operand tiles (see MRA_PRED_LOAD_RANGE)
    constexpr int PH = pred_half_steps<CWT, NLMAX, NTH>();
    constexpr int TPS = NTH / 128;                            // tiles the workgroup moves per staging step
    d2 pre[(GLDS & 1) ? 1 : PH];                                    // staging registers of the register-staged form
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    // staging coordinates: this thread's piece of a tile (pidx_c of 128: row prow_c, first column pcol_c) and, wave-uniform,
    // which of the TPS tiles of a step its wave works on (wv) and which half of that tile (whalf: 64 pieces = 1 KB per wave)
    const int wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 7);
    const int whalf = __builtin_amdgcn_readfirstlane(((int)threadIdx.x >> 6) & 1);
    const int pidx_c = (int)threadIdx.x & 127, prow_c = pidx_c >> 3, pcol_c = (pidx_c & 7) << 1;
    const long t0 = ar.wg_tile0[blockIdx.x];
    const int nt_wg = ar.wg_ntiles[blockIdx.x];
    if (nt_wg == 0) return;                                 // padding slot of the XCD-dealt workgroup order
    const bool active = wave < nt_wg;
    const long t = t0 + (active ? wave : 0);
    const int prow = pi16(r);
    const d4 zero = {0, 0, 0, 0};
    const long myrow = ar.tile_row0[t] + r;
    const int* chain = ar.tile_chain + t0 * 8;
    const double* wrow = ar.W + myrow * ar.ldw + 4 * q;
    d4 w[NLMAX][CWT];
#pragma unroll
    for (int m = 0; m < NLMAX; ++m) {
        if (m < ar.nl) {
#pragma unroll
            for (int jb = 0; jb < CWT; ++jb) w[m][jb] = *(const d4*)(wrow + ar.coff[m] + jb * 16);
        }
    }
    // The y block of W is 16 columns wide for the sake of the tile layout, but only its first column is data: it is carried as ONE
    // number per row (this lane's share of it: the sum over the four lanes of a row is the value) and updated with a few FMAs on
    // the row fragments the lane holds anyway, instead of as a thirteenth MFMA tile (7.7 % of the update's and ~8 % of every
    // level's matrix instructions at C3, and one LDS fragment per step)
    double yv = (q == 0) ? ar.W[myrow * ar.ldw + ar.ycol] : 0.0;
    double ssq = 0.0;
    MRA_PSTAMP(0);
    MRA_PSTAMP_WALL(13);
    bool pre_issued = false;
    // tile offset of half B in the LDS image: the size of the deepest level's half A (no shallower level's half A is larger)
    const int offb = pred_sa<CWT>(ar.nl - 1);
// half A of the deepest level (a run-time level number: ar.deep = ar.lev[ar.nl - 1], a run-time index into lev[] would put the
// whole argument block in scratch)
#define MRA_PRED_LOAD_DEEP_A() MRA_PRED_LOAD_RANGE(ar.nl - 1, ar.deep, 0, offb, 0)
    if (UPD) {
        const int lf = ar.wg_leaf[blockIdx.x];              // all tiles of a workgroup belong to one leaf
        const int nc = ar.leaf_upd[lf] ? (ar.leaf_nop[lf] >> 3) : 0;      // chunks of 8 k
        if (nc > 0) {
            const long nop = ar.leaf_nop[lf];
            const double* ut = ar.leaf_ut[lf];
            // Tt fragments of the row tile, one 16 x 8 chunk at a time (lane (r, q) holds k = 2q, 2q+1 of the chunk), fetched
            // one chunk ahead: holding all of Tt would cost 32 registers and the third wave per SIMD with them
            const double* tt = ut + (long)ar.na * nop + (myrow - ar.leaf_row0[lf]) * nop + 2 * q;
            yv = 0.0;                                        // the y block takes C_in = 0 (the column still holds y itself)
            const int nat = ar.nl * CWT + 1;
            const int nch = nat * 64;                        // 16-byte pieces of a (nat*16) x 8 chunk of Ut, 4 per row
            constexpr int NSTU = ((NLMAX * CWT + 1) * 64 + NTH - 1) / NTH;
            // the two Ut chunk buffers sit in the half-B region of the LDS image: the deepest level's half A is requested into its
            // own region while the last chunk's products issue
            double* cur = lds + (long)offb * 256;
            double* nxt = cur + (long)nat * 128;
            // Ut chunk staging, straight into LDS: step g moves rows g NTH/4 .. of the chunk (4 pieces per row, piece e = thread + g NTH
            // lands at byte 16 e): a scalar base per step plus ONE per-thread offset; nch is a multiple of 64, so a wave is in or out
            const unsigned uvo = (unsigned)(((int)threadIdx.x >> 2) * (int)nop + (((int)threadIdx.x & 3) << 1)) * 8u;
            d2 sg[(GLDS & 2) ? 1 : NSTU];
#define MRA_PRED_UT_LOADS(kcol, DST) do { \
_Pragma("unroll") \
                for (int g = 0; g < NSTU; ++g) { \
                    if (g * NTH + wave * 64 < nch) { \
                        const char* ub = (const char*)(ut + (long)(g * (NTH / 4)) * nop + (kcol)); \
                        if (GLDS & 2) gld_lds16(ub + uvo, (char*)(DST) + (long)(g * NTH + wave * 64) * 16); \
                        else sg[g] = *(const d2*)(ub + uvo); \
                    } \
                } \
} while (0)
#define MRA_PRED_UT_LAND(DST) do { \
                if (GLDS & 2) mra_wait_vm0(); \
                else { \
_Pragma("unroll") \
                    for (int g = 0; g < NSTU; ++g) \
                        if (g * NTH + wave * 64 < nch) *(d2*)((char*)(DST) + (long)(g * NTH + (int)threadIdx.x) * 16) = sg[g]; \
                } \
} while (0)
            MRA_PRED_UT_LOADS(0, cur);
            d2 xc = *(const d2*)tt;
            MRA_PRED_UT_LAND(cur);
            __syncthreads();
            MRA_PSTAMP(1);
// one 8-k chunk: the Ut fragments of level k+1 are read while the products of level k issue, and no further ahead (left
// alone the scheduler hoists all 13 LDS reads to the top: 52 registers, and with them the third wave per SIMD)
#define MRA_PRED_UPD_CHUNK() do { \
                if (active) { \
                    const double x0 = -xc[0], x1 = -xc[1]; \
                    const double* cb = cur + prow * 8 + 2 * q; \
                    d2 bq[2][CWT]; \
                    const d2 uy = *(const d2*)(cur + (ar.nl * CWT) * 128 + 2 * q);      /* row 0 of the y tile: Ut_y[2q], Ut_y[2q+1] */ \
_Pragma("unroll") \
                    for (int kt = 0; kt < CWT; ++kt) bq[0][kt] = *(const d2*)(cb + ((ar.nl - 1) * CWT + kt) * 128); \
_Pragma("unroll") \
                    for (int k = 0; k < NLMAX; ++k) { \
                        if (k < ar.nl) { \
                            if (k + 1 < ar.nl) { \
_Pragma("unroll") \
                                for (int kt = 0; kt < CWT; ++kt) bq[(k + 1) & 1][kt] = *(const d2*)(cb + ((ar.nl - 2 - k) * CWT + kt) * 128); \
                            } \
_Pragma("unroll") \
                            for (int kt = 0; kt < CWT; ++kt) { \
                                w[k][kt] = mfma16(bq[k & 1][kt][0], x0, w[k][kt]); \
                                w[k][kt] = mfma16(bq[k & 1][kt][1], x1, w[k][kt]); \
                            } \
                            __builtin_amdgcn_sched_barrier(0); \
                            if (k + 1 == ar.nl) yv = __builtin_fma(uy[0], x0, __builtin_fma(uy[1], x1, yv)); \
                        } \
                    } \
                } \
} while (0)
#pragma nounroll
            for (int c = 0; c + 1 < nc; ++c) {
                const int kn = (c + 1) * 8;
                MRA_PRED_UT_LOADS(kn, nxt);                   // (nxt was read two chunks ago: everybody has passed a barrier since)
                const d2 xn = *(const d2*)(tt + kn);
                MRA_PRED_UPD_CHUNK();
                MRA_PRED_UT_LAND(nxt);
                __syncthreads();
                xc = xn;
                double* const sw = cur; cur = nxt; nxt = sw;
            }
            // the deepest level's half A rides behind the last chunk, into its own region of the image
            MRA_PRED_LOAD_DEEP_A();
            pre_issued = true;
            MRA_PRED_UPD_CHUNK();
        }
    }
    MRA_PSTAMP(2);
#pragma unroll
    for (int mm = 0; mm < NLMAX; ++mm) {
        const int m = NLMAX - 1 - mm;
        if (m < ar.nl) {
            const int SA = pred_sa<CWT>(m);                   // tiles in half A; half B: the remaining TB tiles at tile offset offb
            const int TB = pred_tiles_all<CWT>(m) - SA;
            // ---- stage: a half is requested (global -> LDS, no registers) BEFORE the products of the half in front of it, so its L2
            // round trip (a load -> ds_write loop pays ~1 us per chunk, which was half of this kernel's run time) hides behind MFMA work
            if (m == ar.nl - 1 && !pre_issued) MRA_PRED_LOAD_RANGE(m, ar.lev[m], 0, SA, 0);   // the first (deepest) level of this tree, no leaf update in front of it: LDS is untouched
            MRA_PRED_LAND_RANGE(SA, 0);
            __syncthreads();                                  // half A has landed, and everybody is done with the previous level's half B
            MRA_PRED_LOAD_RANGE(m, ar.lev[m], SA, TB, offb);
            if (m < 5) MRA_PSTAMP(3 + 2 * m);
            d4 x[CWT];                                        // MINUS X_m: every product below accumulates straight into its w tile (no separate accumulator: 8 registers)
            // tile slot -> LDS address (slots below SA in half A, the others in half B)
#define MRA_PRED_TILE(slot) (lds + (long)((slot) < SA ? (slot) : offb + (slot) - SA) * 256)
            if (active) {
#pragma unroll
                for (int jb = 0; jb < CWT; ++jb) {
                    d4 acc = w[m][jb];
#pragma unroll
                    for (int kb = 0; kb < CWT; ++kb) {
                        if (kb < jb) {
                            const d4 a = *(const d4*)(lds + (jb * (jb - 1) / 2 + kb) * 256 + prow * 16 + 4 * q);
#pragma unroll
                            for (int j = 0; j < 4; ++j) acc = mfma16(a[j], x[kb][j], acc);
                        }
                    }
                    const d4 ia = *(const d4*)(lds + (NTRI + jb) * 256 + prow * 16 + 4 * q);
                    d4 xx = zero;
#pragma unroll
                    for (int j = 0; j < 4; ++j) xx = mfma16(ia[j], acc[j], xx);
                    ssq += xx[0] * xx[0] + xx[1] * xx[1] + xx[2] * xx[2] + xx[3] * xx[3];
                    x[jb] = -xx;
                }
            }
            // the row blocks of the front below the own block, in consumption order g = k CWT + kt (ancestor level k, tile kt), then y:
            // those staged with half A now, the others after half B has landed
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                if (half == 1) {
                    MRA_PRED_LAND_RANGE(TB, offb);
                    __syncthreads();                          // half B has landed, and everybody is done with half A
                    if (m > 0) MRA_PRED_LOAD_RANGE(m - 1, ar.lev[m - 1], 0, (pred_sa<CWT>(m - 1)), 0);
                }
                if (active) {
#pragma unroll
                    for (int k = 0; k < NLMAX; ++k) {
                        if (k < m) {
#pragma unroll
                            for (int kt = 0; kt < CWT; ++kt) {
                                const int s0 = NTRI + CWT + (k * CWT + kt) * CWT;
                                if ((s0 < SA) == (half == 0)) {
#pragma unroll
                                    for (int jb = 0; jb < CWT; ++jb) {
                                        const d4 z = *(const d4*)(MRA_PRED_TILE(s0 + jb) + prow * 16 + 4 * q);
#pragma unroll
                                        for (int j = 0; j < 4; ++j) w[k][kt] = mfma16(z[j], x[jb][j], w[k][kt]);
                                    }
                                }
                            }
                        }
                    }
                    {
                        const int s0 = NTRI + CWT + m * CWT * CWT;      // the y row block follows the ancestors: its first row is Zt's y row
                        if ((s0 < SA) == (half == 0)) {
#pragma unroll
                            for (int jb = 0; jb < CWT; ++jb) {
                                const d4 z = *(const d4*)(MRA_PRED_TILE(s0 + jb) + 4 * q);
#pragma unroll
                                for (int j = 0; j < 4; ++j) yv = __builtin_fma(z[j], x[jb][j], yv);
                            }
                        }
                    }
                }
            }
#undef MRA_PRED_TILE
            if (m < 5) MRA_PSTAMP(4 + 2 * m);
        }
    }
    MRA_PSTAMP(15);
    MRA_PSTAMP_WALL(14);
    if (!active) return;
    ssq += __shfl_xor(ssq, 16, 64);
    ssq += __shfl_xor(ssq, 32, 64);
    yv += __shfl_xor(yv, 16, 64);
    yv += __shfl_xor(yv, 32, 64);
    if (q == 0) {
        const double v0 = ar.var[myrow];
        ar.var[myrow] = (v0 > 0.0 ? v0 : 0.0) + ssq;
        ar.mean[myrow] = -yv;
    }
}

// ------------------------------------------------------------------------------------------------
//  Predictive cascade of DEEP trees with 64-wide blocks (five to eight non-leaf levels of four tiles: BASELINE config 5,
//  M = 8, r0 = 64), upper half.  A row tile of such a tree has 32 basis tiles: twice the register file.  The walk is split:
//      k_predict_hi      the four deepest levels with their 16 tiles in registers, X_m = W~^m Lt_m^-T in place (MRANode.py:504-511),
//                        the updates among those four levels in registers, then ONE sweep over the row's coarser tiles
//                        (levels < nl-4 and the y block):  w -= sum_{m >= nl-4} X_m Zt_m[w's rows]^T, read and written once;
//      k_predict_cascade the remaining (at most four) coarse levels, as for any shallow tree.
//  Level by level the same work re-reads and re-writes the coarse columns once per level: 129 GB at config 5 against 35 GB here.
//  Node operands do not fit LDS (Zt of a level-7 node: 232 KB), so they stream through a double-buffered stage in chunks of
//  16 tiles (32 KB): the solve operands of a level, the Zt rows of one finer-on-coarser level pair, or the Zt rows of all four
//  levels for one coarse tile - 40 to 64 MFMAs per wave, chunk and barrier.
// ------------------------------------------------------------------------------------------------
struct PredHiArgs {
    PredLevel hi[4];          // the four deepest non-leaf levels, hi[h] = level nl - 4 + h
    double* W;                // in: whitened basis after the leaf update; out: columns [levels < nl - 4 | y block] updated in place
    double* var;              // in: leaf part (clamped at 0 here), out: + sum_h |X_h|^2
    long ldw;
    int coff_hi[4];           // W column of hi level h
    int col_low;              // W column of the first coarse tile (levels nl-5 .. 0 and the y block are contiguous from here)
    int n_low;                // coarse tiles, y tile included: (nl - 4) * 4 + 1
    int lev0;                 // nl - 4: position of hi level 0 in a tile's chain
    const long* tile_row0;
    const int* tile_chain;
    const long* wg_tile0;     // per workgroup: first tile and number of tiles (<= 4), all sharing one chain
    const int* wg_ntiles;
    // UPD: the leaf update W[S, anc | y] -= Tt Ut^T rides in this launch (leaves of at most 64 padded observations: four k tiles) -
    // on the sixteen deep tiles in registers before the levels are walked, on every coarse tile as it passes through the sweep.
    // The updated W never exists in memory: a read and a write of all of W less (36 GB at config 5).
    const int* wg_leaf;       // [workgroup] leaf number of the workgroup's tiles (they share one)
    double* const* leaf_ut;   // [leaf] solved Ut block (na x nop, rows: deepest level first, then coarser, then the y block), then the leaf's Tt rows
    const int* leaf_nop;      // [leaf]
    const long* leaf_row0;    // [leaf] first padded row
    int na;                   // rows of Ut
};

// the 16-byte pieces of a 16-tile chunk this thread moves: piece i is row (chunk >> 3), doubles c2, c2+1 of tile tsel + 2 i; they go
// from global memory straight into the LDS buffer DST (gld_lds16: lane-linear, a wave fills half a tile = 1 KB per instruction)
#define MRA_PH_LOAD_Z(DST, F0, N0, A0, F1, N1, A1, F2, N2, A2, F3, N3, A3) do { \
        char* const db_ = (char*)(DST) + (long)stile0 * 2048 + whalf * 1024; \
        gld_lds16((F0) + (long)(64 + (A0) * 16 + srow) * (N0) + sjb0 * 16 + sc2, db_ + 0 * 4096); \
        gld_lds16((F0) + (long)(64 + (A0) * 16 + srow) * (N0) + (sjb0 + 2) * 16 + sc2, db_ + 1 * 4096); \
        gld_lds16((F1) + (long)(64 + (A1) * 16 + srow) * (N1) + sjb0 * 16 + sc2, db_ + 2 * 4096); \
        gld_lds16((F1) + (long)(64 + (A1) * 16 + srow) * (N1) + (sjb0 + 2) * 16 + sc2, db_ + 3 * 4096); \
        gld_lds16((F2) + (long)(64 + (A2) * 16 + srow) * (N2) + sjb0 * 16 + sc2, db_ + 4 * 4096); \
        gld_lds16((F2) + (long)(64 + (A2) * 16 + srow) * (N2) + (sjb0 + 2) * 16 + sc2, db_ + 5 * 4096); \
        gld_lds16((F3) + (long)(64 + (A3) * 16 + srow) * (N3) + sjb0 * 16 + sc2, db_ + 6 * 4096); \
        gld_lds16((F3) + (long)(64 + (A3) * 16 + srow) * (N3) + (sjb0 + 2) * 16 + sc2, db_ + 7 * 4096); \
} while (0)
// solve operands of hi level H: tiles 0..5 = strictly-lower tiles of Lt ((1,0) (2,0) (2,1) (3,0) (3,1) (3,2)), 6..9 = inverted diagonal blocks
#define MRA_PH_LOAD_T(DST, H) do { \
        char* const db_ = (char*)(DST) + (long)stile0 * 2048 + whalf * 1024; \
_Pragma("unroll") \
        for (int i = 0; i < 5; ++i) { \
            const int tile = stile0 + 2 * i; \
            const double* src; \
            if (tile < 6) { \
                const int jb = tile < 1 ? 1 : (tile < 3 ? 2 : 3), kb = tile - jb * (jb - 1) / 2; \
                src = Fh[H] + (long)(jb * 16 + srow) * nfh[H] + kb * 16 + sc2; \
            } else src = invh[H] + (long)(tile - 6) * 256 + srow * 16 + sc2; \
            gld_lds16(src, db_ + i * 4096); \
        } \
} while (0)
// the requested chunk has landed and everybody is done with the buffer the next request overwrites
#define MRA_PH_SYNC() do { mra_wait_vm0(); __syncthreads(); } while (0)
// leaf update, deep part: k tile KT of the Ut rows of the sixteen deep tiles (Ut row tile a = (3 - h) 4 + jb: the deepest level first)
#define MRA_PH_LOAD_UT0(DST, KT) do { \
        char* const db_ = (char*)(DST) + (long)stile0 * 2048 + whalf * 1024; \
_Pragma("unroll") \
        for (int i = 0; i < 8; ++i) gld_lds16(ut + (long)((stile0 + 2 * i) * 16 + srow) * nop + (KT) * 16 + sc2, db_ + i * 4096); \
} while (0)
// leaf update, sweep part: the (at most four) k tiles of the Ut rows of coarse tile I, behind the sixteen Zt tiles of the chunk
#define MRA_PH_LOAD_UTV(DST, I) do { \
        if (UPD) { \
_Pragma("unroll") \
            for (int i2 = 0; i2 < 2; ++i2) \
                if (stile0 + 2 * i2 < nkt) \
                    gld_lds16(ut + (long)((16 + (I)) * 16 + srow) * nop + (stile0 + 2 * i2) * 16 + sc2, (char*)(DST) + (long)(16 + stile0 + 2 * i2) * 2048 + whalf * 1024); \
        } \
} while (0)

template <int CWT, bool UPD>        // CWT = 4 (a template so that only the translation unit that launches it instantiates it)
__global__ __launch_bounds__(256, 2) void k_predict_hi(PredHiArgs ar) {
    static_assert(CWT == 4, "k_predict_hi is written for 64-wide blocks");
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int CHT = UPD ? 20 : 16;                     // tiles per chunk buffer: sixteen operand tiles (+ up to four k tiles of Ut in the sweep)
    double* const buf0 = lds;
    double* const buf1 = lds + CHT * 256;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    const long t0 = ar.wg_tile0[blockIdx.x];
    const int nt_wg = ar.wg_ntiles[blockIdx.x];
    if (nt_wg == 0) return;
    const bool active = wave < nt_wg;
    const long t = t0 + (active ? wave : 0);
    const int prow = pi16(r);
    const d4 zero = {0, 0, 0, 0};
    const long myrow = ar.tile_row0[t] + r;
    const int* chain = ar.tile_chain + t0 * 8 + ar.lev0;
    double* const wrow = ar.W + myrow * ar.ldw + 4 * q;
    // staging role of this thread inside a 16-tile chunk
    const int stile0 = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 7), sjb0 = stile0, schunk = threadIdx.x & 127, srow = schunk >> 3, sc2 = (schunk & 7) << 1;
    const int whalf = __builtin_amdgcn_readfirstlane(((int)threadIdx.x >> 6) & 1);
    const double* Fh[4];
    const double* invh[4];
    long nfh[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        const long slot = chain[h];
        nfh[h] = ar.hi[h].ld;
        Fh[h] = ar.hi[h].F + slot * ar.hi[h].stride;
        invh[h] = ar.hi[h].invF + slot * 4 * 256;
    }
    d4 w[4][4];
#pragma unroll
    for (int h = 0; h < 4; ++h)
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) w[h][jb] = *(const d4*)(wrow + ar.coff_hi[h] + jb * 16);
    double ssq = 0.0;
    const int fo = prow * 16 + 4 * q;                      // this lane's fragment inside a staged tile
    // ---- leaf update of the sixteen deep tiles: w[h][jb] -= Tt Ut[(3 - h) 4 + jb]^T, one chunk of sixteen Ut tiles per k tile
    int nkt = 0;
    long nop = 0;
    const double* ut = nullptr;
    const double* ttp = nullptr;                           // this lane's share of the row tile's Tt rows (vec layout: row r, k = 4 q ..)
    if (UPD) {
        const int lf = ar.wg_leaf[blockIdx.x];
        nop = ar.leaf_nop[lf];
        nkt = __builtin_amdgcn_readfirstlane((int)(nop >> 4));
        ut = ar.leaf_ut[lf];
        ttp = ut + (long)ar.na * nop + (myrow - ar.leaf_row0[lf]) * nop + 4 * q;
    }
#define MRA_PH_UPD0(TT, BUF) do { \
        if (active) { \
_Pragma("unroll") \
            for (int h = 0; h < 4; ++h) { \
_Pragma("unroll") \
                for (int jb = 0; jb < 4; ++jb) { \
                    const d4 z = *(const d4*)((BUF) + ((3 - h) * 4 + jb) * 256 + fo); \
_Pragma("unroll") \
                    for (int j = 0; j < 4; ++j) w[h][jb] = mfma16(z[j], (TT)[j], w[h][jb]); \
                } \
                __builtin_amdgcn_sched_barrier(0);   /* four fragment reads in flight, not sixteen (128 registers) */ \
            } \
        } \
} while (0)
    // ---- the four deepest levels, deepest first; the solve operands of the deepest level end up in buf0 whatever came before: k
    // tile kt of the leaf update sits in buf1 when nkt - 1 - kt is even (the last one always), in buf0 otherwise
    d4 tt[4] = {zero, zero, zero, zero};                   // MINUS Tt, k tiles 0 .. 3, for the sweep (loaded behind the deep part: sixteen w tiles + these is the register peak)
    if (UPD && nkt > 0) {
        double* bc = ((nkt - 1) & 1) ? buf0 : buf1;
        double* bn = ((nkt - 1) & 1) ? buf1 : buf0;
        MRA_PH_LOAD_UT0(bc, 0);
        d4 tk = -*(const d4*)ttp;
        MRA_PH_SYNC();
#pragma nounroll
        for (int kt = 0; kt + 1 < nkt; ++kt) {
            MRA_PH_LOAD_UT0(bn, kt + 1);
            const d4 tn = -*(const d4*)(ttp + (kt + 1) * 16);
            MRA_PH_UPD0(tk, bc);
            MRA_PH_SYNC();
            tk = tn;
            double* const sw = bc; bc = bn; bn = sw;
        }
        MRA_PH_LOAD_T(buf0, 3);    MRA_PH_UPD0(tk, buf1);  MRA_PH_SYNC();
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) if (kt < nkt) tt[kt] = -*(const d4*)(ttp + kt * 16);
    } else {
        MRA_PH_LOAD_T(buf0, 3);
        MRA_PH_SYNC();
    }
#define MRA_PH_SOLVE(H, BUF) do { \
        if (active) { \
            d4 x[4]; \
_Pragma("unroll") \
            for (int jb = 0; jb < 4; ++jb) { \
                d4 acc = w[H][jb]; \
                d4 upd = zero; \
_Pragma("unroll") \
                for (int kb = 0; kb < 4; ++kb) { \
                    if (kb < jb) { \
                        const d4 a = *(const d4*)((BUF) + (jb * (jb - 1) / 2 + kb) * 256 + fo); \
_Pragma("unroll") \
                        for (int j = 0; j < 4; ++j) upd = mfma16(a[j], x[kb][j], upd); \
                    } \
                } \
                acc -= upd; \
                const d4 ia = *(const d4*)((BUF) + (6 + jb) * 256 + fo); \
                d4 xx = zero; \
_Pragma("unroll") \
                for (int j = 0; j < 4; ++j) xx = mfma16(ia[j], acc[j], xx); \
                x[jb] = xx; \
                ssq += xx[0] * xx[0] + xx[1] * xx[1] + xx[2] * xx[2] + xx[3] * xx[3]; \
            } \
_Pragma("unroll") \
            for (int jb = 0; jb < 4; ++jb) w[H][jb] = x[jb]; \
        } \
} while (0)
    // w[HP][kt] -= X_H Zt_H[rows of level HP]^T : tiles (kt, jb) of the chunk
#define MRA_PH_UPDATE(H, HP, BUF) do { \
        if (active) { \
_Pragma("unroll") \
            for (int kt = 0; kt < 4; ++kt) { \
                d4 acc = zero; \
_Pragma("unroll") \
                for (int jb = 0; jb < 4; ++jb) { \
                    const d4 z = *(const d4*)((BUF) + (kt * 4 + jb) * 256 + fo); \
_Pragma("unroll") \
                    for (int j = 0; j < 4; ++j) acc = mfma16(z[j], w[H][jb][j], acc); \
                } \
                w[HP][kt] -= acc; \
            } \
        } \
} while (0)
#define MRA_PH_LOAD_U(DST, H, HP) MRA_PH_LOAD_Z(DST, Fh[H], nfh[H], ((H) - 1 - (HP)) * 4 + 0, Fh[H], nfh[H], ((H) - 1 - (HP)) * 4 + 1, \
                                                Fh[H], nfh[H], ((H) - 1 - (HP)) * 4 + 2, Fh[H], nfh[H], ((H) - 1 - (HP)) * 4 + 3)
    // level 3 (deepest): the operands of the step behind are requested into the other buffer, the products of this step issue,
    // then wait + barrier
    MRA_PH_LOAD_U(buf1, 3, 2);  MRA_PH_SOLVE(3, buf0);      MRA_PH_SYNC();
    MRA_PH_LOAD_U(buf0, 3, 1);  MRA_PH_UPDATE(3, 2, buf1);  MRA_PH_SYNC();
    MRA_PH_LOAD_U(buf1, 3, 0);  MRA_PH_UPDATE(3, 1, buf0);  MRA_PH_SYNC();
    MRA_PH_LOAD_T(buf0, 2);     MRA_PH_UPDATE(3, 0, buf1);  MRA_PH_SYNC();
    // level 2
    MRA_PH_LOAD_U(buf1, 2, 1);  MRA_PH_SOLVE(2, buf0);      MRA_PH_SYNC();
    MRA_PH_LOAD_U(buf0, 2, 0);  MRA_PH_UPDATE(2, 1, buf1);  MRA_PH_SYNC();
    MRA_PH_LOAD_T(buf1, 1);     MRA_PH_UPDATE(2, 0, buf0);  MRA_PH_SYNC();
    // level 1
    MRA_PH_LOAD_U(buf0, 1, 0);  MRA_PH_SOLVE(1, buf1);      MRA_PH_SYNC();
    MRA_PH_LOAD_T(buf1, 0);     MRA_PH_UPDATE(1, 0, buf0);  MRA_PH_SYNC();
    // level 0 of the four; behind it the first coarse tile's operands
    const int n_low = ar.n_low;
    double* const wlow = wrow + ar.col_low;
#define MRA_PH_LOAD_V(DST, I) MRA_PH_LOAD_Z(DST, Fh[0], nfh[0], 0 + (I), Fh[1], nfh[1], 4 + (I), Fh[2], nfh[2], 8 + (I), Fh[3], nfh[3], 12 + (I))
    MRA_PH_LOAD_V(buf0, 0);
    MRA_PH_LOAD_UTV(buf0, 0);
    d4 wl = *(const d4*)wlow;
    MRA_PH_SOLVE(0, buf1);
    MRA_PH_SYNC();
    // ---- one sweep over the coarse tiles: w_i -= sum_h X_h Zt_h[4 h + i]^T  (tile (h, jb) of the chunk)
#pragma nounroll
    for (int i = 0; i < n_low; ++i) {
        const double* bufc = (i & 1) ? buf1 : buf0;
        double* bufn = (i & 1) ? buf0 : buf1;
        const int in = (i + 1 < n_low) ? i + 1 : i;
        if (i + 1 < n_low) { MRA_PH_LOAD_V(bufn, in); MRA_PH_LOAD_UTV(bufn, in); }
        const d4 wn = *(const d4*)(wlow + in * 16);
        if (UPD && i == n_low - 1) wl = zero;              // the y block takes C_in = 0 in the leaf update (the column still holds y itself)
        if (active) {
            d4 acc0 = zero, acc1 = zero;
            if (UPD) {
                // leaf update of this coarse tile: + Tt Ut[16 + i]^T with tt = -Tt, i.e. subtracted from the sum that is subtracted below
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
                    if (kt < nkt) {
                        const d4 z = *(const d4*)(bufc + (16 + kt) * 256 + fo);
                        if (kt & 1) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) acc1 = mfma16(z[j], tt[kt][j], acc1);
                        } else {
#pragma unroll
                            for (int j = 0; j < 4; ++j) acc0 = mfma16(z[j], tt[kt][j], acc0);
                        }
                    }
                acc0 = -acc0; acc1 = -acc1;
            }
#pragma unroll
            for (int h = 0; h < 4; ++h) {
#pragma unroll
                for (int jb = 0; jb < 4; ++jb) {
                    const d4 z = *(const d4*)(bufc + (h * 4 + jb) * 256 + fo);
                    if (h & 1) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc1 = mfma16(z[j], w[h][jb][j], acc1);
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc0 = mfma16(z[j], w[h][jb][j], acc0);
                    }
                }
            }
            *(d4*)(wlow + i * 16) = wl - (acc0 + acc1);
        }
        wl = wn;
        MRA_PH_SYNC();
    }
    if (!active) return;
    ssq += __shfl_xor(ssq, 16, 64);
    ssq += __shfl_xor(ssq, 32, 64);
    if (q == 0) {
        const double v0 = ar.var[myrow];
        ar.var[myrow] = (v0 > 0.0 ? v0 : 0.0) + ssq;
    }
}

// ------------------------------------------------------------------------------------------------
//  small kernels
// ------------------------------------------------------------------------------------------------
// W[:, Ka] = y (0 where missing), W[:, Ka+1 .. Ka+15] = 0
#ifndef MRA_KERNELS_TEMPLATES_ONLY      /* non-template kernel: defined once, in the translation unit of mra_plan.hip */
// var (optional): the prior variance C(x,x) of every row, from which the level-by-level row solves then subtract |W^m[x]|^2 level by
// level and the leaf solve |Tt[x]|^2 (instead of a separate pass over all of W at the end: k_leaf_moments)
__global__ void k_init_yblock(double* __restrict__ W, long ldw, int Ka, const double* __restrict__ y, long P,
                              double* __restrict__ var, double cov0, const double* __restrict__ diag_host) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P * MRA_YB) return;
    long p = i / MRA_YB; int c = (int)(i % MRA_YB);
    double v = 0.0;
    if (c == 0) {
        double yy = y[p]; v = isfinite(yy) ? yy : 0.0;
        if (var) var[p] = diag_host ? diag_host[p] : cov0;
    }
    W[p * ldw + Ka + c] = v;
}
// what is left of k_leaf_moments when the row solves have accumulated the variance: clamp at 0, reset the y column of the leaves' rows
__global__ void k_leaf_finish_var(const int* __restrict__ row_leaf, double* __restrict__ W, long ldw, int Ka, double* __restrict__ var, long P) {
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P || row_leaf[p] < 0) return;
    const double v = var[p];
    var[p] = v > 0.0 ? v : 0.0;
    W[p * ldw + Ka] = 0.0;
}
#endif

// prior: kInv block of a node = rows of its knots out of its own conditional-covariance block
struct KinvProb { double* Lp; const long* knots; int rank; int cw; };
#ifndef MRA_KERNELS_TEMPLATES_ONLY      /* non-template kernel: defined once, in the translation unit of mra_plan.hip */
__global__ void k_gather_kinv(const KinvProb* __restrict__ probs, const double* __restrict__ W, long ldw, int c0) {
    const KinvProb pb = probs[blockIdx.y];
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= pb.cw * pb.cw) return;
    int a = e / pb.cw, b = e % pb.cw;
    double v = (a == b) ? 1.0 : 0.0;
    if (a < pb.rank && b < pb.rank) v = W[pb.knots[a] * ldw + c0 + b];
    pb.Lp[(long)a * pb.cw + b] = v;
}
#endif

// leaf panel: [ C = V[o,o] + R I ; Ut = [W_anc[o] | y_o]^T ; V[S,o] ] -> fill the first two parts
struct LeafProb {
    double* Pn;           // panel base
    const int* obs;       // nop entries: padded-row numbers of observed rows (-1 phantom)
    long row0;            // first padded row of the leaf
    long ld;              // = nop
    int nrows;            // N_j
    int nop;              // padded number of observations
    int na;               // ancestors + y block
    int a0;               // first ancestor column in W
    int node;
};
#ifndef MRA_KERNELS_TEMPLATES_ONLY      /* non-template kernel: defined once, in the translation unit of mra_plan.hip */
__global__ void k_leaf_fill(const LeafProb* __restrict__ probs, const double* __restrict__ W, long ldw, double R) {
    const LeafProb pb = probs[blockIdx.y];
    const long total = (long)(pb.nop + pb.na) * pb.nop;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        int row = (int)(e / pb.nop), k = (int)(e % pb.nop);
        double v;
        if (row < pb.nop) {
            int oa = pb.obs[row], ob = pb.obs[k];
            if (oa >= 0 && ob >= 0) {
                v = pb.Pn[(long)(pb.nop + pb.na + (oa - pb.row0)) * pb.ld + k];
                if (row == k) v += R;
            } else v = (row == k) ? 1.0 : 0.0;
        } else {
            int t = row - pb.nop, ok = pb.obs[k];
            v = (ok >= 0) ? W[(long)ok * ldw + pb.a0 + t] : 0.0;
        }
        pb.Pn[(long)row * pb.ld + k] = v;
    }
}
#endif

// phantom observation rows of a leaf's C block: identity (the real rows come from the COV epilogue)
#ifndef MRA_KERNELS_TEMPLATES_ONLY      /* non-template kernel: defined once, in the translation unit of mra_plan.hip */
__global__ void k_leaf_cphantom(const LeafProb* __restrict__ probs, const int* __restrict__ n_obs) {
    const LeafProb pb = probs[blockIdx.x];
    const int no = n_obs[blockIdx.x];
    const int total = (pb.nop - no) * pb.nop;
    for (int e = threadIdx.x; e < total; e += blockDim.x) {
        const int row = no + e / pb.nop, col = e % pb.nop;
        pb.Pn[(long)row * pb.ld + col] = (row == col) ? 1.0 : 0.0;
    }
}
#endif

// leaf moments: var = max(C(x,x) - |W_anc[x]|^2 - |Tt[x]|^2, 0); y column of W reset to 0.
// One wave per row.
#ifndef MRA_KERNELS_TEMPLATES_ONLY      /* non-template kernel: defined once, in the translation unit of mra_plan.hip */
__global__ __launch_bounds__(256) void k_leaf_moments(const LeafProb* __restrict__ probs, const int* __restrict__ row_leaf,
                                                       double* __restrict__ W, long ldw, int Ka,
                                                       double* __restrict__ var, double cov0,
                                                       const double* __restrict__ diag_host, long P) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long p = (long)blockIdx.x * 4 + wave;
    if (p >= P) return;
    const int lf = row_leaf[p];
    if (lf < 0) return;
    const LeafProb pb = probs[lf];
    double s = 0.0;
    const double* wr = W + p * ldw;
    for (int c = pb.a0 + lane; c < Ka; c += 64) { double v = wr[c]; s += v * v; }
    const double* tr = pb.Pn + (long)(pb.nop + pb.na + (p - pb.row0)) * pb.ld;
    for (int k = lane; k < pb.nop; k += 64) { double v = tr[k]; s += v * v; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) {
        double base = diag_host ? diag_host[p] : cov0;
        double v = base - s;
        var[p] = v > 0.0 ? v : 0.0;
        W[p * ldw + Ka] = 0.0;
    }
}
#endif

// sum of n numbers by one 256-thread workgroup in a FIXED order (the result must not depend on scheduling): thread t adds up
// x[t], x[t + 256], ... (coalesced, four independent partial sums), thread 0 adds the 256 partial results in thread order.
// (Round 1-2: thread t summed a contiguous chunk - every load of a wave a different cache line: 144 us for the 87381 nodes of
// config 5, 9 us at C3.)
__device__ __forceinline__ double block_sum_ordered(const double* __restrict__ x, int n, double* part /* 256 doubles of LDS */) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int i = threadIdx.x;
    for (; i + 768 < n; i += 1024) { s0 += x[i]; s1 += x[i + 256]; s2 += x[i + 512]; s3 += x[i + 768]; }
    for (; i < n; i += 256) s0 += x[i];
    part[threadIdx.x] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x == 0)
        for (int k = 0; k < 256; ++k) t += part[k];
    return t;                                    // valid in thread 0
}

// non-leaf front: F = [I on the own block] + sum of the children's Schur blocks (lower triangle)
struct AsmProb { double* F; int nf; int cw; int child0; int nchild; int add_identity; };
struct AsmChild { const double* G; long ld; };
#ifndef MRA_KERNELS_TEMPLATES_ONLY      /* non-template kernel: defined once, in the translation unit of mra_plan.hip */
__global__ void k_assemble(const AsmProb* __restrict__ probs, const AsmChild* __restrict__ kids, int add_identity,
                           const double* __restrict__ sum_in = nullptr, int sum_n = 0, double* __restrict__ sum_out = nullptr) {
    // sum_in: one more row of workgroups (blockIdx.y == gridDim.y - 1) adds up sum_n log-determinants in order - the reduce level of
    // a sharded run needs that number beside its assembled fronts, and a launch of its own for it sat on the rank's critical path
    if (sum_in && blockIdx.y == gridDim.y - 1) {
        if (blockIdx.x != 0) return;
        __shared__ double part[256];
        const double t = block_sum_ordered(sum_in, sum_n, part);
        if (threadIdx.x == 0) sum_out[0] = t;
        return;
    }
    const AsmProb pb = probs[blockIdx.y];
    const long total = (long)pb.nf * pb.nf;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        int a = (int)(e / pb.nf), b = (int)(e % pb.nf);
        if (b > a) continue;
        double v = (add_identity && a == b && a < pb.cw) ? 1.0 : 0.0;
        for (int c = 0; c < pb.nchild; ++c) {
            const AsmChild ch = kids[pb.child0 + c];
            v += ch.G[(long)a * ch.ld + b];
        }
        pb.F[(long)a * pb.nf + b] = v;
    }
}
#endif
#ifndef MRA_KERNELS_TEMPLATES_ONLY      /* non-template kernel: defined once, in the translation unit of mra_plan.hip */
__global__ void k_add_identity(const AsmProb* __restrict__ probs) {
    const AsmProb pb = probs[blockIdx.y];
    int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a < pb.cw) pb.F[(long)a * pb.nf + a] += 1.0;
}
#endif

// ------------------------------------------------------------------------------------------------
//  Fused non-leaf front: assembly + partial Cholesky + Schur complement in ONE launch per level, the front held
//  in LDS as 16x16 tiles (pyMRA/MRANode.py:434-445, 463, 476-480):
//      F = [I on the own block] + sum_children Gt_c ;  Lt = chol(F_oo), Zt = F_ao Lt^-T ;  Gt = F_aa - Zt Zt^T
//  Replaces k_assemble / k_add_identity + k_panel_chol + k_gemm_nt<SUB> (three dependent launches per level, each
//  a few tiles of work behind a full launch + global round trips): what bounds the upper levels of the tree is the
//  length of that dependent chain, not arithmetic.
//  FULL  : every lower tile of the front fits in LDS (nt (nt+1)/2 + cwt tiles): assemble -> factor -> Schur in LDS, one
//          coalesced write of the finished front.
//  PANEL : only the cwt panel columns fit: the front is read from global memory (built by the parent SYRK or by
//          k_assemble), the Schur update is applied to the trailing tiles in global memory.
//  Tiles are stored with a row stride of 18 doubles (32-byte fragment reads without bank conflicts); the product idiom is
//  the vec layout of k_trsm_rows2: with a = rows pi(r) of tile P and b = rows r of tile Q (32 bytes per lane each),
//  the accumulator is tile (Q P^T) in vec layout again.
// ------------------------------------------------------------------------------------------------
struct FrontProb {
    double* F;          // nf x nf, row-major, lower part
    double* invd;       // cwt inverted diagonal blocks (256 doubles each) for the predictive pass
    int nf;
    int cwt;            // panel column tiles (own block)
    int node;
    int child0, nchild; // children's Schur blocks in the AsmChild array (FULL with assembly)
};

template <bool FULL>
__global__ __launch_bounds__(512, 1) void k_front(const FrontProb* __restrict__ probs, const AsmChild* __restrict__ kids,
                                                   double* __restrict__ dnode, int* __restrict__ err, int do_assemble, int add_identity) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const FrontProb* __restrict__ pp = probs + blockIdx.x;
    double* const F = pp->F;
    const int nf = pp->nf, cwt = pp->cwt, nt = nf >> 4, cw = cwt << 4;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    const int nwave = blockDim.x >> 6;
    const int prow = pi16(r);
    const d4 zero = {0, 0, 0, 0};
    // tile (i, j), j <= i: FULL keeps the whole lower triangle, PANEL the first cwt column tiles
    auto tix = [&](int i, int j) -> int { return FULL ? (i * (i + 1) / 2 + j) : (j * nt - j * (j - 1) / 2 + (i - j)); };
    const int n_lds_tiles = FULL ? nt * (nt + 1) / 2 : cwt * nt - cwt * (cwt - 1) / 2;
    double* const inv = lds + (long)n_lds_tiles * FT_SZ;            // cwt inverted diagonal blocks
    // ---- A. bring the front in.  16-byte chunks enumerated row by row over the lower trapezoid (tile row block i holds
    // 16 rows of 8 (i+1) chunks; PANEL: at most cwt column tiles), so consecutive lanes read consecutive chunks of one
    // front row; FA_G chunks (x up to four children) of every thread are in flight together - a plain load -> add ->
    // ds_write loop pays one L2 round trip per chunk, which was most of this kernel's run time.
    {
        constexpr int FA_G = 4;
        const int wmax = FULL ? nt : cwt;                     // column tiles kept per row block, at most
        int nchunk = 0;
        for (int i = 0; i < nt; ++i) nchunk += 128 * min(i + 1, wmax);
        const int nchild = (FULL && do_assemble) ? pp->nchild : 0;
        const AsmChild* kd = kids + pp->child0;
        for (int e0 = threadIdx.x; e0 < nchunk; e0 += FA_G * (int)blockDim.x) {
            d2 v[FA_G];
            int dst[FA_G];
            long off[FA_G];
            int grow[FA_G], gcol[FA_G];
            bool diag0[FA_G], diag1[FA_G];
#pragma unroll
            for (int g = 0; g < FA_G; ++g) {
                int e = e0 + g * (int)blockDim.x;
                e = e < nchunk ? e : nchunk - 1;
                int i = 0, base = 0;
                while (i + 1 < nt && e >= base + 128 * min(i + 1, wmax)) { base += 128 * min(i + 1, wmax); ++i; }
                const int w8 = 8 * min(i + 1, wmax), rem = e - base;
                const int row = rem / w8, cc = rem - row * w8, j = cc >> 3, c2 = (cc & 7) << 1;
                const int gr = i * 16 + row, gc = j * 16 + c2;
                off[g] = (long)gr * nf + gc;
                grow[g] = gr; gcol[g] = gc;
                dst[g] = tix(i, j) * FT_SZ + row * FT_LD + c2;
                const bool idn = add_identity && i == j && gr < cw;
                diag0[g] = idn && row == c2;
                diag1[g] = idn && row == c2 + 1;
                v[g] = d2{0.0, 0.0};
            }
            if (FULL && do_assemble) {
                for (int c0 = 0; c0 < nchild; c0 += 4) {
                    d2 t[FA_G][4];
#pragma unroll
                    for (int ch = 0; ch < 4; ++ch) {
                        const int cc = (c0 + ch < nchild) ? c0 + ch : c0;
                        const double* G = kd[cc].G;
                        const long ldc = kd[cc].ld;              // a child's Schur block sits inside the child's own (larger) front
#pragma unroll
                        for (int g = 0; g < FA_G; ++g) t[g][ch] = gld2(G + (long)grow[g] * ldc + gcol[g]);
                    }
#pragma unroll
                    for (int ch = 0; ch < 4; ++ch)
                        if (c0 + ch < nchild) {
#pragma unroll
                            for (int g = 0; g < FA_G; ++g) v[g] += t[g][ch];
                        }
                }
            } else {
#pragma unroll
                for (int g = 0; g < FA_G; ++g) v[g] = gld2(F + off[g]);
            }
#pragma unroll
            for (int g = 0; g < FA_G; ++g) {
                if (e0 + g * (int)blockDim.x < nchunk) {
                    d2 x = v[g];
                    if (diag0[g]) x[0] += 1.0;
                    if (diag1[g]) x[1] += 1.0;
                    *(d2*)(lds + dst[g]) = x;
                }
            }
        }
    }
    __syncthreads();
    // ---- B. partial Cholesky of the cwt panel columns (left-looking over the panel)
    double logacc = 0.0;
    for (int jb = 0; jb < cwt; ++jb) {
        if (jb > 0) {
            for (int ib = jb + wave; ib < nt; ib += nwave) {
                d4 u0 = zero, u1 = zero;
                for (int kb = 0; kb < jb; ++kb) {
                    const d4 a = *(const d4*)(lds + (long)tix(jb, kb) * FT_SZ + prow * FT_LD + 4 * q);
                    const d4 b = *(const d4*)(lds + (long)tix(ib, kb) * FT_SZ + r * FT_LD + 4 * q);
                    u0 = mfma16(a[0], b[0], u0); u1 = mfma16(a[1], b[1], u1);
                    u0 = mfma16(a[2], b[2], u0); u1 = mfma16(a[3], b[3], u1);
                }
                d4* tp = (d4*)(lds + (long)tix(ib, jb) * FT_SZ + r * FT_LD + 4 * q);
                *tp = *tp - (u0 + u1);
            }
            __syncthreads();
        }
        if (wave == 0) {
            double* dt = lds + (long)tix(jb, jb) * FT_SZ;
            bool bad = false;
#ifdef MRA_ATOM_OLD
            double a[16], m[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) a[k] = (k <= r) ? dt[r * FT_LD + k] : 0.0;
            logacc += chol16_inv(a, m, r, bad, nullptr);
            if (lane < 16) {
#pragma unroll
                for (int k = 0; k < 16; k += 2) {
                    *(d2*)(dt + lane * FT_LD + k) = d2{a[k], a[k + 1]};
                    *(d2*)(inv + jb * FT_SZ + lane * FT_LD + k) = d2{m[k], m[k + 1]};
                    gst2(pp->invd + (long)jb * 256 + lane * 16 + k, d2{m[k], m[k + 1]});
                }
            }
#else
            logacc += chol16_ldl(dt, FT_LD, dt, FT_LD, inv + jb * FT_SZ, FT_LD, lane, bad, nullptr, 0, pp->invd + (long)jb * 256);
#endif
            if (bad && lane == 0) atomicMax(err, pp->node + 1);
        }
        __syncthreads();
        {
            const d4 ia = *(const d4*)(inv + jb * FT_SZ + prow * FT_LD + 4 * q);
            for (int ib = jb + 1 + wave; ib < nt; ib += nwave) {
                d4* tp = (d4*)(lds + (long)tix(ib, jb) * FT_SZ + r * FT_LD + 4 * q);
                const d4 b = *tp;
                d4 x0 = mfma16(ia[0], b[0], zero), x1 = mfma16(ia[1], b[1], zero);
                x0 = mfma16(ia[2], b[2], x0); x1 = mfma16(ia[3], b[3], x1);
                *tp = x0 + x1;
            }
        }
        __syncthreads();
    }
    // ---- C. Schur complement of the trailing block, D. write back
    const int ntr = nt - cwt;
    const int ntrail = ntr * (ntr + 1) / 2;
    auto trail_ij = [&](int t, int& i, int& j) {
        int ii = 0;
        while ((ii + 1) * (ii + 2) / 2 <= t) ++ii;
        i = cwt + ii; j = cwt + (t - ii * (ii + 1) / 2);
    };
    d4 tnext = zero;
    if (!FULL && wave < ntrail) {                  // PANEL: the trailing tile comes from global memory, one tile ahead
        int i, j;
        trail_ij(wave, i, j);
        tnext = gld4(F + (long)(i * 16 + r) * nf + j * 16 + 4 * q);
    }
    for (int t = wave; t < ntrail; t += nwave) {
        int i, j;
        trail_ij(t, i, j);
        d4 tcur = tnext;
        if (!FULL) {
            int i2, j2;
            trail_ij(t + nwave < ntrail ? t + nwave : t, i2, j2);
            tnext = gld4(F + (long)(i2 * 16 + r) * nf + j2 * 16 + 4 * q);
        } else {
            tcur = *(const d4*)(lds + (long)tix(i, j) * FT_SZ + r * FT_LD + 4 * q);
        }
        d4 u0 = zero, u1 = zero;
        for (int kb = 0; kb < cwt; ++kb) {
            const d4 a = *(const d4*)(lds + (long)tix(j, kb) * FT_SZ + prow * FT_LD + 4 * q);
            const d4 b = *(const d4*)(lds + (long)tix(i, kb) * FT_SZ + r * FT_LD + 4 * q);
            u0 = mfma16(a[0], b[0], u0); u1 = mfma16(a[1], b[1], u1);
            u0 = mfma16(a[2], b[2], u0); u1 = mfma16(a[3], b[3], u1);
        }
        gst4(F + (long)(i * 16 + r) * nf + j * 16 + 4 * q, tcur - (u0 + u1));
    }
    // panel columns: Lt and Zt
    for (int t = wave; t < cwt * nt - cwt * (cwt - 1) / 2; t += nwave) {
        int j = 0, base = 0;
        while (j + 1 < cwt && t >= base + (nt - j)) { base += nt - j; ++j; }
        const int i = j + (t - base);
        gst4(F + (long)(i * 16 + r) * nf + j * 16 + 4 * q, *(const d4*)(lds + (long)tix(i, j) * FT_SZ + r * FT_LD + 4 * q));
    }
    if (threadIdx.x == 0) dnode[pp->node] = 2.0 * logacc;
}

// ------------------------------------------------------------------------------------------------
//  Parents of the leaves: front assembly from the children's Ut blocks AND its factorisation in one launch
//      F = I_own + sum_c Ut_c Ut_c^T  (MRANode.py:434-440 with the leaves in observation space)  ->  Lt, Zt, Gt as k_front
//  One workgroup per node.  The K dimension (the children's observations, 16 at a time) streams through a
//  double-buffered LDS stage shared by the 8 waves; every wave accumulates up to NACC of the node's nt(nt+1)/2 lower
//  tiles in registers (48 MFMAs per wave and barrier at C3).  The front never makes the HBM round trip between a
//  SYRK kernel and a factorisation kernel (2 x 177 MB at C3), and the operand fragments come from LDS instead of L2.
//  Panel tiles go through LDS for the partial Cholesky (as k_front), the trailing tiles take their Schur update in
//  registers and are written once.
// ------------------------------------------------------------------------------------------------
#define PF_LD 20          /* doubles per staged row of 16 k */
template <int NACC>
__global__ __launch_bounds__(512, 1) void k_parent_front(const FrontProb* __restrict__ probs, const GemmSeg* __restrict__ segs_all,
                                                          double* __restrict__ dnode, int* __restrict__ err) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const FrontProb* __restrict__ pp = probs + blockIdx.x;
    double* const F = pp->F;
    const int nf = pp->nf, cwt = pp->cwt, nt = nf >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    const int prow = pi16(r);
    const d4 zero = {0, 0, 0, 0};
    const int ntiles = nt * (nt + 1) / 2;
    const int npanel = cwt * nt - cwt * (cwt - 1) / 2;
    // LDS: two staging buffers of nf rows x PF_LD, then the panel tiles (k_front's PANEL layout) and the inverted diagonal blocks
    double* const stage0 = lds;
    double* const stage1 = lds + (long)nf * PF_LD;
    double* const pan = lds + 2L * nf * PF_LD;
    double* const inv = pan + (long)npanel * FT_SZ;
    auto tix = [&](int i, int j) -> int { return j * nt - j * (j - 1) / 2 + (i - j); };
    // this wave's tiles: t = wave + 8 n, row-major over the lower triangle (t = i (i+1)/2 + j)
    int ti[NACC], tj[NACC];
#pragma unroll
    for (int n = 0; n < NACC; ++n) {
        const int t = wave + 8 * n;
        int i = 0;
        while ((i + 1) * (i + 2) / 2 <= t) ++i;
        ti[n] = i; tj[n] = t - i * (i + 1) / 2;
    }
    d4 acc[NACC];
#pragma unroll
    for (int n = 0; n < NACC; ++n) acc[n] = zero;
    // ---- K loop over the children's observation columns
    const GemmSeg* segs = segs_all + pp->child0;
    const int nseg = pp->nchild;
    // staging role: 32-byte chunk (row, 4 columns) of the nf x 16 slab; nf * 4 chunks, up to two per thread
    const int c0 = threadIdx.x, c1 = threadIdx.x + 512;
    const int nch = nf * 4;
    const int row0 = (c0 < nch ? c0 : 0) >> 2, col0 = (c0 & 3) << 2;
    const int row1 = (c1 < nch ? c1 : 0) >> 2, col1 = (c1 & 3) << 2;
    int sg = 0, k0 = 0;
    while (sg < nseg && segs[sg].K == 0) ++sg;
    d4 s0 = zero, s1 = zero;
    if (sg < nseg) {
        const double* A = segs[sg].A; const long lda = segs[sg].lda;
        s0 = gld4(A + (long)row0 * lda + col0);
        s1 = gld4(A + (long)row1 * lda + col1);
    }
    int cur = 0;
    if (sg < nseg) {
        if (c0 < nch) *(d4*)(stage0 + row0 * PF_LD + col0) = s0;
        if (c1 < nch) *(d4*)(stage0 + row1 * PF_LD + col1) = s1;
    }
    __syncthreads();
    while (sg < nseg) {
        // next chunk (if any) into registers while this one is on the MFMA pipe
        int sg2 = sg, k2 = k0 + 16;
        if (k2 >= segs[sg2].K) { k2 = 0; ++sg2; while (sg2 < nseg && segs[sg2].K == 0) ++sg2; }
        const bool more = sg2 < nseg;
        {
            const int sl = more ? sg2 : sg;
            const int kl = more ? k2 : k0;
            const double* A = segs[sl].A; const long lda = segs[sl].lda;
            s0 = gld4(A + (long)row0 * lda + kl + col0);
            s1 = gld4(A + (long)row1 * lda + kl + col1);
        }
        const double* st = cur ? stage1 : stage0;
#pragma unroll
        for (int n = 0; n < NACC; ++n) {
            if (wave + 8 * n < ntiles) {
                const d4 a = *(const d4*)(st + (tj[n] * 16 + prow) * PF_LD + 4 * q);
                const d4 b = *(const d4*)(st + (ti[n] * 16 + r) * PF_LD + 4 * q);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[n] = mfma16(a[j], b[j], acc[n]);
            }
        }
        if (more) {
            double* sn = cur ? stage0 : stage1;
            if (c0 < nch) *(d4*)(sn + row0 * PF_LD + col0) = s0;
            if (c1 < nch) *(d4*)(sn + row1 * PF_LD + col1) = s1;
        }
        __syncthreads();
        cur ^= 1; sg = sg2; k0 = k2;
    }
    // ---- identity on the own block, panel tiles to LDS
#pragma unroll
    for (int n = 0; n < NACC; ++n) {
        if (wave + 8 * n < ntiles) {
            if (ti[n] == tj[n] && ti[n] < cwt) {
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[n][j] += (r == 4 * q + j) ? 1.0 : 0.0;
            }
            if (tj[n] < cwt) *(d4*)(pan + (long)tix(ti[n], tj[n]) * FT_SZ + r * FT_LD + 4 * q) = acc[n];
        }
    }
    __syncthreads();
    // ---- partial Cholesky of the panel (k_front, phase B); the diagonal blocks go to the last wave (fewest tiles in registers)
    double logacc = 0.0;
    const int nwave = 8;
    for (int jb = 0; jb < cwt; ++jb) {
        if (jb > 0) {
            for (int ib = jb + wave; ib < nt; ib += nwave) {
                d4 u0 = zero, u1 = zero;
                for (int kb = 0; kb < jb; ++kb) {
                    const d4 a = *(const d4*)(pan + (long)tix(jb, kb) * FT_SZ + prow * FT_LD + 4 * q);
                    const d4 b = *(const d4*)(pan + (long)tix(ib, kb) * FT_SZ + r * FT_LD + 4 * q);
                    u0 = mfma16(a[0], b[0], u0); u1 = mfma16(a[1], b[1], u1);
                    u0 = mfma16(a[2], b[2], u0); u1 = mfma16(a[3], b[3], u1);
                }
                d4* tp = (d4*)(pan + (long)tix(ib, jb) * FT_SZ + r * FT_LD + 4 * q);
                *tp = *tp - (u0 + u1);
            }
            __syncthreads();
        }
        if (wave == nwave - 1) {
            double* dt = pan + (long)tix(jb, jb) * FT_SZ;
            bool bad = false;
#ifdef MRA_ATOM_OLD
            double a[16], m[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) a[k] = (k <= r) ? dt[r * FT_LD + k] : 0.0;
            logacc += chol16_inv(a, m, r, bad, nullptr);
            if (lane < 16) {
#pragma unroll
                for (int k = 0; k < 16; k += 2) {
                    *(d2*)(dt + lane * FT_LD + k) = d2{a[k], a[k + 1]};
                    *(d2*)(inv + jb * FT_SZ + lane * FT_LD + k) = d2{m[k], m[k + 1]};
                    gst2(pp->invd + (long)jb * 256 + lane * 16 + k, d2{m[k], m[k + 1]});
                }
            }
#else
            logacc += chol16_ldl(dt, FT_LD, dt, FT_LD, inv + jb * FT_SZ, FT_LD, lane, bad, nullptr, 0, pp->invd + (long)jb * 256);
#endif
            if (bad && lane == 0) atomicMax(err, pp->node + 1);
        }
        __syncthreads();
        {
            const d4 ia = *(const d4*)(inv + jb * FT_SZ + prow * FT_LD + 4 * q);
            for (int ib = jb + 1 + wave; ib < nt; ib += nwave) {
                d4* tp = (d4*)(pan + (long)tix(ib, jb) * FT_SZ + r * FT_LD + 4 * q);
                const d4 b = *tp;
                d4 x0 = mfma16(ia[0], b[0], zero), x1 = mfma16(ia[1], b[1], zero);
                x0 = mfma16(ia[2], b[2], x0); x1 = mfma16(ia[3], b[3], x1);
                *tp = x0 + x1;
            }
        }
        __syncthreads();
    }
    // ---- Schur update of the trailing tiles in registers, everything out.  The tile coordinates are worked out again from an
    // opaque copy of the wave number (scalar arithmetic): kept from the top of the kernel they sit in registers through the K loop
    // and the factorisation, and were spilled (19 registers, 80 B/lane)
    int wv = wave;
    asm volatile("" : "+s"(wv));
#pragma unroll
    for (int n = 0; n < NACC; ++n) {
        if (wave + 8 * n < ntiles) {
            int i = 0;
            {
                const int t = wv + 8 * n;
                while ((i + 1) * (i + 2) / 2 <= t) ++i;
            }
            const int j = wv + 8 * n - i * (i + 1) / 2;
            double* gp = F + (long)(i * 16 + r) * nf + j * 16 + 4 * q;
            if (j < cwt) {
                gst4(gp, *(const d4*)(pan + (long)tix(i, j) * FT_SZ + r * FT_LD + 4 * q));
            } else {
                d4 u0 = zero, u1 = zero;
                for (int kb = 0; kb < cwt; ++kb) {
                    const d4 a = *(const d4*)(pan + (long)tix(j, kb) * FT_SZ + prow * FT_LD + 4 * q);
                    const d4 b = *(const d4*)(pan + (long)tix(i, kb) * FT_SZ + r * FT_LD + 4 * q);
                    u0 = mfma16(a[0], b[0], u0); u1 = mfma16(a[1], b[1], u1);
                    u0 = mfma16(a[2], b[2], u0); u1 = mfma16(a[3], b[3], u1);
                }
                gst4(gp, acc[n] - (u0 + u1));
            }
        }
    }
    if (threadIdx.x == 64 * (nwave - 1)) dnode[pp->node] = 2.0 * logacc;
}

// mean = -W[:, Ka]
#ifndef MRA_KERNELS_TEMPLATES_ONLY      /* non-template kernel: defined once, in the translation unit of mra_plan.hip */
__global__ void k_extract_mean(const double* __restrict__ W, long ldw, int Ka, double* __restrict__ mean, long P) {
    long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < P) mean[p] = -W[p * ldw + Ka];
}
#endif

// d = sum of per-node log-determinants, in node order (deterministic), one workgroup.  With `up` set
// this is the last launch of a pass and out[0..3] = {d, u, log-det sum carried by the reduce level,
// error flag} is the one record the host reads back
#ifndef MRA_KERNELS_TEMPLATES_ONLY      /* non-template kernel: defined once, in the translation unit of mra_plan.hip */
__global__ __launch_bounds__(256) void k_sum_dnode(const double* __restrict__ dnode, int n, double* __restrict__ out,
                                                   const double* __restrict__ up = nullptr, const double* __restrict__ below = nullptr,
                                                   const int* err = nullptr) {
    __shared__ double part[256];
    const double t = block_sum_ordered(dnode, n, part);
    if (threadIdx.x == 0) {
        out[0] = t;
        if (up) {
            out[1] = *up;
            out[2] = below ? *below : 0.0;
            out[3] = (double)(*err);
            *const_cast<int*>(err) = 0;          // ready for the next pass (no memset launch per run)
        }
    }
}
#endif
