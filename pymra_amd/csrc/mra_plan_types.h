// mra_plan_types.h - the plan object and the small host helpers shared by the translation units of libmra_hip.so
// (mra_plan.hip: plan construction, launch sequence, C ABI; mra_launch_*.hip: the dispatchers of the heavily templated
// kernels, compiled separately so that the library builds in parallel).
#pragma once
#include "mra_kernels.h"
#include "../../include/mra_hip.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <initializer_list>
#include <rccl/rccl.h>      // types and enum values only: the library itself is loaded with dlopen at run time
#include <set>
#include <string>
#include <vector>
#include <time.h>

#define MRA_VERSION_STR "mra_hip 0.2 (gfx950)"

// ---- host dry run (test instrumentation, never a compute path) ------------------------------------------------------
// With MRA_HOST_DRYRUN=1 in the environment the plan is built entirely in host memory: every "device" buffer is a
// malloc, uploads are memcpy, no stream / event / kernel is ever created and mra_run refuses to run.  The point is
// to push the ~1500 lines of index arithmetic of plan construction (build_static, build_leaf, the host-cov block
// bookkeeping) and the native tree replay through AddressSanitizer / UBSan on a machine without a GPU
// (`make asan`, tests/test_asan_host.py).  Nothing is computed in this mode.
static const bool g_dry = []() { const char* e = getenv("MRA_HOST_DRYRUN"); return e && e[0] == '1'; }();
// Device allocations go through a small process-wide cache (mra_plan.hip): blocks of a destroyed plan are kept and handed to the
// next plan that asks for the same size on the same device.  The reference's MLE pattern builds a NEW MRATree per objective call
// (README.md:96-104): without the cache every call pays hipMalloc + hipFree of ~4 GB (~30 ms at 1024^2) around a 6 ms pass.
hipError_t mraMalloc(void** p, size_t n);
hipError_t mraFree(void* p);
static inline hipError_t mraMemcpy(void* d, const void* s_, size_t n, hipMemcpyKind k) { if (g_dry) { memcpy(d, s_, n); return hipSuccess; } return hipMemcpy(d, s_, n, k); }
static inline hipError_t mraMemset(void* d, int v, size_t n) { if (g_dry) { memset(d, v, n); return hipSuccess; } return hipMemset(d, v, n); }
static inline hipError_t mraMemcpy2D(void* d, size_t dp, const void* s_, size_t sp, size_t w, size_t h, hipMemcpyKind k) {
    if (g_dry) { for (size_t i = 0; i < h; ++i) memcpy((char*)d + i * dp, (const char*)s_ + i * sp, w); return hipSuccess; }
    return hipMemcpy2D(d, dp, s_, sp, w, h, k);
}
static inline hipError_t mraSetDevice(int dev) { return g_dry ? hipSuccess : hipSetDevice(dev); }

static thread_local std::string g_last_error;

#define HIP_TRY(expr)                                                                     \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess) {                                                           \
            char _b[512];                                                                 \
            snprintf(_b, sizeof _b, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                     __FILE__, __LINE__);                                                 \
            throw MraError(MRA_ERR_HIP, _b);                                              \
        }                                                                                 \
    } while (0)

struct MraError {
    int code;
    std::string msg;
    MraError(int c, const std::string& m) : code(c), msg(m) {}
};

enum KFam {
    KF_PRIOR_RESID = 0, KF_PRIOR_CHOL, KF_PRIOR_TRSM, KF_LEAF_RESID, KF_LEAF_CHOL, KF_LEAF_SYRK,
    KF_LEAF_UPDATE, KF_FRONT_CHOL, KF_FRONT_SCHUR, KF_PRED_TRSM, KF_PRED_UPDATE, KF_MISC, KF_COUNT
};
// family names = the kernels that actually run (rocprofv3 kernel names), by path: [0] fused cascades on regular trees,
// [1] general level-by-level path, [2] level-by-level prior and fronts with the two-kernel predictive cascade of deep 64-wide trees.  tools/summarize_profiles.py maps the trace's kernel names onto the same strings.
static const char* kfam_name[3][KF_COUNT] = {
    {"k_gemm_nt_lds<COV> prior residual (unused on the fused path)",
     "k_knot_chain + k_prior_cascade<KNOT> knot pass (knot rows, kInv, Cholesky)",
     "k_prior_cascade row pass (W of all levels)",
     "k_leaf_gemm<COV> leaf residual V[S,o] and C",
     "k_chol_wave + k_trsm_rows2 leaf factor and solves (Lc, Ut, Tt)",
     "k_parent_front (children's Ut -> parent front -> Lt, Zt, Schur)",
     "k_gemm_nt_lds<SUB> / k_leaf_solve_update leaf update (separate launch)",
     "k_front (assembly + partial Cholesky + Schur per level)",
     "k_gemm_nt<SUB> front Schur complement (fronts too large for LDS)",
     "k_trsm_rows2 predict (unused on the fused path)",
     "k_predict_cascade (leaf update + all levels, mean/var)",
     "small kernels (k_assemble, k_leaf_cphantom, k_sum_dnode, ...)"},
    {"k_leaf_gemm<COV,SOLVE> prior of a level: residual + kernel + row solve (or k_gemm_nt_lds<COV> residual only)",
     "k_gemm_nt_lds<COV> knots' residual block + k_panel_chol prior kInv Cholesky per level",
     "k_trsm_rows2 prior W = R L^-T per level",
     "k_leaf_gemm<COV> leaf residual V[S,o] and C",
     "k_chol_wave + k_trsm_rows2 (or k_panel_chol) leaf factor and solves",
     "k_gemm_nt<SET> / k_parent_front leaf or parent SYRK",
     "k_gemm_nt_lds<SUB> leaf update W[S,anc] -= Tt^T Ut",
     "k_front / k_panel_chol front partial Cholesky",
     "k_gemm_nt<SUB> front Schur complement",
     "k_trsm_rows2 predict X = W Lt^-T per level",
     "k_gemm_nt_lds<SUB> predict update per level",
     "small kernels (k_assemble, k_gather_kinv, k_leaf_moments, k_sum_dnode, ...)"},
    {"k_leaf_gemm<COV,SOLVE> prior of a level: residual + kernel + row solve (or k_gemm_nt_lds<COV> residual only)",
     "k_gemm_nt_lds<COV> knots' residual block + k_panel_chol prior kInv Cholesky per level",
     "k_trsm_rows2 prior W = R L^-T per level",
     "k_leaf_gemm<COV> leaf residual V[S,o] and C",
     "k_chol_wave + k_trsm_rows2 (or k_panel_chol) leaf factor and solves",
     "k_gemm_nt<SET> / k_parent_front leaf or parent SYRK",
     "k_gemm_nt_lds<SUB> leaf update W[S,anc] -= Tt^T Ut",
     "k_front / k_panel_chol front partial Cholesky",
     "k_gemm_nt<SUB> front Schur complement",
     "k_predict_cascade (the four coarse levels of a deep tree)",
     "k_predict_hi (four deepest levels in registers + one sweep over the coarse columns)",
     "small kernels (k_assemble, k_gather_kinv, k_leaf_moments, k_sum_dnode, ...)"}};

// Work of a launch (or a family of launches): algorithmic flops with TRUE sizes (true ranks, true observation counts, a
// one-column y), the flops the MFMA tiles actually execute on the 16-padded layout, and the algorithmic HBM bytes of the
// launch (each operand array of the padded layout read or written once).  bench.py prices `alg` against the FP64 roof and
// `bytes` against the HBM roof; `exec` is reported as the executed-MFMA rate.
struct Work {
    double alg = 0, exec = 0, bytes = 0;
    Work() {}
    Work(double a) : alg(a), exec(a) {}
    Work(double a, double e, double b = 0) : alg(a), exec(e), bytes(b) {}
    Work& operator+=(const Work& o) { alg += o.alg; exec += o.exec; bytes += o.bytes; return *this; }
    Work operator+(const Work& o) const { Work w = *this; w += o; return w; }
    Work with_bytes(double b) const { Work w = *this; w.bytes = b; return w; }
};

// (MRA_TRACE_PLAN: number of descriptor uploads and the time spent in them)
struct UploadStats { long n = 0; double ms = 0; };
inline UploadStats& upload_stats() { static thread_local UploadStats s; return s; }

// Descriptor arena: a plan uploads ~170 small arrays (problem descriptors, index lists) while it is built - each one a device
// allocation and a synchronous pageable copy, ~20 us apiece, a third of the construction time of a 1024^2 tree.  Inside an
// ArenaScope the small ones (< 256 KB) are bump-allocated from ONE device block of the plan and staged in a pinned host mirror;
// the scope's end sends what was added in one copy.  Pointers are final at once (descriptors embed each other's addresses).
// A full arena simply falls back to the stand-alone path.
struct UploadArena {
    char* dev = nullptr;
    char* host = nullptr;           // pinned mirror (plain memory in a host dry run)
    size_t cap = 0, used = 0, flushed = 0;
    static constexpr size_t SMALL = 256 * 1024;
    void* take(size_t bytes, const void* src) {
        const size_t at = (used + 255) & ~(size_t)255;
        if (!dev || at + bytes > cap) return nullptr;
        memcpy(host + at, src, bytes);
        used = at + bytes;
        return dev + at;
    }
    void flush() {
        if (used > flushed) {
            if (mraMemcpy(dev + flushed, host + flushed, used - flushed, hipMemcpyHostToDevice) != hipSuccess) throw MraError(MRA_ERR_HIP, "hipMemcpy H2D failed (descriptor arena)");
            flushed = used;
        }
    }
};
inline UploadArena*& current_arena() { static thread_local UploadArena* a = nullptr; return a; }
struct ArenaScope {
    UploadArena* prev;
    UploadArena* mine;
    explicit ArenaScope(UploadArena* a) : prev(current_arena()), mine(a) { current_arena() = a; }
    void finish() { current_arena() = prev; if (mine) { UploadArena* a = mine; mine = nullptr; a->flush(); } }
    ~ArenaScope() { current_arena() = prev; if (mine) { try { mine->flush(); } catch (...) {} } }      // (error paths: the plan is torn down anyway)
};

template <class T>
struct DevVec {
    T* p = nullptr;
    size_t n = 0;
    bool in_arena = false;          // p points into the plan's descriptor arena: nothing to free
    void upload(const std::vector<T>& h) {
        static const bool trace = getenv("MRA_TRACE_PLAN") != nullptr;
        timespec t0{}, t1{};
        if (trace) clock_gettime(CLOCK_MONOTONIC, &t0);
        release();
        n = h.size();
        if (n) {
            UploadArena* a = current_arena();
            void* q = (a && n * sizeof(T) < UploadArena::SMALL) ? a->take(n * sizeof(T), h.data()) : nullptr;
            if (q) { p = (T*)q; in_arena = true; }
            else {
                if (mraMalloc((void**)&p, n * sizeof(T)) != hipSuccess) throw MraError(MRA_ERR_HIP, "hipMalloc failed (descriptor array)");
                if (mraMemcpy(p, h.data(), n * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) throw MraError(MRA_ERR_HIP, "hipMemcpy H2D failed");
            }
        }
        if (trace) { clock_gettime(CLOCK_MONOTONIC, &t1); upload_stats().n += 1; upload_stats().ms += (t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6; }
    }
    // copy into the EXISTING allocation (descriptors already point into it)
    void fill(const std::vector<T>& h) {
        if (h.size() != n) throw MraError(MRA_ERR_STATE, "DevVec::fill: size changed");
        if (n && mraMemcpy(p, h.data(), n * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) throw MraError(MRA_ERR_HIP, "hipMemcpy H2D failed");
    }
    void alloc(size_t count) {
        release();
        n = count;
        if (n && mraMalloc((void**)&p, n * sizeof(T)) != hipSuccess) {
            char b[160];
            snprintf(b, sizeof b, "hipMalloc of %.3f GB failed", (double)(n * sizeof(T)) / 1e9);
            p = nullptr;
            throw MraError(MRA_ERR_HIP, b);
        }
    }
    void release() {
        if (p && !in_arena) mraFree(p);
        p = nullptr;
        n = 0;
        in_arena = false;
    }
    ~DevVec() { release(); }
};

struct LevelData {
    std::vector<int> nodes;          // non-leaf nodes of this level
    int cw = 0, cwt = 0, c0 = 0, a0 = 0, nf = 0, na = 0;
    int ldf = 0;                     // row stride of a front in F: nf, or cw when only the panel columns [Lt ; Zt] are kept (panel_only)
    bool panel_only = false;
    long max_rows = 0;               // largest row range among the nodes
    DevVec<double> Lp, invP, F, invF;
    DevVec<GemmProb> gResid, gSchur, gUpdate;
    // one-launch prior of the level (k_leaf_gemm<COV, SOLVE>): the knots' residual block straight into Lp (both sides gathered through
    // the knot list), then residual + kernel + row solve on blocks of rows
    DevVec<GemmProb> gKnotResid, gResidFused;
    DevVec<GemmProb> gResidLik;               // ... over the gathered rows a likelihood needs (ensure_lik_general)
    double lik_share = 1.0;                   // their share of the level's rows
    bool prior_level_ok = false;
    Work fl_knot_resid;
    std::vector<GemmProb> hResid;
    DevVec<KinvProb> gKinv;
    DevVec<PanelProb> gPriorChol, gFrontChol;
    DevVec<TrsmNode> gTrsmPrior, gTrsmPost;
    DevVec<Trsm2Prob> gTrsm2Prior, gTrsm2Post;
    long max_tiles = 0;
    DevVec<int> tile_node;
    DevVec<long> tile_row0;
    long ntiles = 0;
    DevVec<AsmProb> gAsm;
    DevVec<FrontProb> gFront;
    int front_mode = 0;               // 0: separate launches, 1: k_front<PANEL>, 2: k_front<FULL> (whole front in LDS)
    size_t front_lds = 0;
    Work fl_resid, fl_pchol, fl_trsm, fl_fchol, fl_schur, fl_update;
};

struct mra_plan {
    int device = 0;
    UploadArena arena;                   // the small descriptor arrays of this plan (device block + pinned mirror, see UploadArena)
    hipStream_t stream = nullptr;        // the pass; carries the chain of small dependent launches and the all-reduce (high priority)
    hipStream_t stream2 = nullptr;       // side stream: the leaf update runs here, beside the front chain / all-reduce (low priority)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool side_pending = false;           // work on stream2 that the predictive pass has to wait for
    std::string err;
    // topology (host)
    long P = 0;
    int d = 0, n_levels = 0, n_nodes = 0;
    std::vector<long> level_ptr, row0, row1, knot_ptr, knot_rows;
    bool knots_pending = false;         // mra_plan_create_replay_2d: built beside the knot draws; knot_rows (and the leaves' knot_ptr) arrive later (plan_set_knots)
    std::vector<uint8_t> leaf;
    std::vector<int> parent, child_ptr, child_list, cw, node_level;
    // layout
    int Ka = 0, ldw = 0;
    std::vector<int> coff, asuf, nf, na;
    // state
    bool have_locs = false, have_obs = false, have_kernel = false, ran = false, split_pending = false;
    uint32_t run_flags = 0;
    KernelParams kp{};
    bool host_cov = false;
    double R = 0.0;
    int reduce_level = -1;
    // device data
    DevVec<double> X, y, W, var, mean, dnode, scal, covsrc, covdiag, stamps, pstamps, tstamps, kstamps, kstamps2;
    double* host_res = nullptr;      // pinned, device-mapped {d, u, below, err} record of the last pass
    double* host_res_dev = nullptr;  // the same memory as the device sees it
    DevVec<int> errflag, knot_idx, row_leaf;
    DevVec<long> knots_dev;
    std::vector<long> knot_idx_off;      // per node offset into knot_idx (padded to cw)
    std::vector<LevelData> lev;
    std::vector<int> node_slot;          // index of a non-leaf node inside its level's arrays
    // leaves
    std::vector<int> leaf_nodes;         // node numbers
    std::vector<int> leaf_slot;          // node -> leaf index or -1
    std::vector<int> leaf_nop;
    std::vector<long> leaf_poff, leaf_goff, leaf_ioff;
    DevVec<double> panel, leafInv, Gt;
    DevVec<int> obs_idx, obs_pos, leaf_nobs, leaf_nop_dev, ft_leaf;
    DevVec<double*> leaf_ut;
    DevVec<LeafProb> gLeaf;
    DevVec<GemmProb> gLeafResid, gLeafSyrk, gLeafUpdate, gLeafResidLik;
    std::vector<GemmProb> hLeafResid;
    std::vector<int> leaf_nobs_host;
    DevVec<PanelProb> gLeafCholFull, gLeafCholLik, gLeafCholC, gLeafCholSorted;
    // leaves with more than 192 observations: right-looking blocked factorisation, 64 columns per step (one panel launch +
    // one trailing-update GEMM per step); [variant 0 full / 1 likelihood-only][step] -> descriptors of all leaves
    std::vector<DevVec<PanelProb>> gBigPanel[2];
    std::vector<DevVec<GemmProb>> gBigTrail[2];
    std::vector<long> bigM[2], bigN[2];
    DevVec<Trsm2Prob> gLeafTrsmFull, gLeafTrsmLik, gLeafTrsmFullPlain, gLeafTrsmLikPlain, gLeafTrsmFullPlainG, gLeafTrsmLikPlainG;
    DevVec<LeafSolveProb> gLeafSolve;     // k_leaf_solve_update, same order as the *Plain arrays (leaves with nt <= 8 first)
    DevVec<LeafSolveProb> gLeafSolveHalf; // the small leaves again, two workgroups each (row tiles split in two): shorter workgroups on the
    size_t n_leaf_solve_half = 0;         // side stream free their CUs sooner for the high-priority front chain (MRA_OPT_LEAF_SOLVE_SPLIT)
    int leaf_solve_split = 2;             // measured on an eighth of C3: 1.086 -> 1.066 ms
    DevVec<GemmProb> gLeafUpdatePlain;    // the leaf update in that order (for the leaves the fused kernel does not take)
    bool use_leaf_solve = true, leaf_solve_ok = false;
    int use_chol_lds = 1;                     // leaf Cholesky with one workgroup per matrix (k_chol_tiles): 1 when a CU sees at most two leaves, 2 always, 0 never
    size_t n_chol_small = 0;
    bool ut_gather = true;                    // fused path: the leaves' Ut rows gathered from W by the row solve (no scatter in the row cascade: 1.21 -> 1.03 ms there, +0.16 ms in the solve)
    bool cphantom_valid = false;              // the phantom observation rows of the leaves' C blocks hold their identity rows
    int use_hi_fold = 1;                      // deep 64-wide trees: the leaf update W -= Tt Ut^T inside k_predict_hi instead of a pass over all of W (option 16)
    bool hi_fold_now = false;                 // ... decided for the pass that is running
    bool use_prior_level = true;              // levels of the level-by-level prior with blocks <= 64 wide: residual + kernel + row solve in one launch (option 15)
    bool parent_panel_lds_ok = false;         // every parent-panel problem fits the step table of the LDS-tiled segmented product
    bool grand_syrk_dma_ok = false;           // ... and the smaller one of its DMA-staged form
    bool grand_syrk_blk_ok = false;           // every grandparent problem fits k_syrk_blk's step table
    int use_syrk_blk = 1;                     // the grandparents' signed SYRK on 96 x 96 blocks through LDS (k_syrk_blk) instead of 32 x 32 wave tiles (option 14)
    bool seg_gemm_lds = true;                 // the parents' panel product (segmented: sum over the children's Ut blocks) on the LDS-tiled GEMM (6.3 -> 5.4 ms at config 5)
    bool use_pred_update = true, pred_update_now = false;   // leaf update folded into the predictive cascade
    DevVec<long> leaf_row0_dev;
    DevVec<unsigned char> leaf_upd_dev;
    size_t leaf_solve_lds = 0;
    int leaf_solve_mode = 2;              // MRA_OPT_LEAF_SOLVE: 0 off, 1 always, 2 (default) when the leaves are few per CU
    size_t n_trsm_small = 0;            // the *Plain arrays are ordered: leaves with nt <= 8 first
    int trsm_small_nt = 0, trsm_small_tiles_full = 0, trsm_small_tiles_lik = 0;
    DevVec<GemmProb> gParentSyrk;        // fused path: fronts of the leaves' parents straight from the children's Ut
    DevVec<GemmSeg> parentSegs;
    DevVec<FrontProb> gParentFront;      // the same nodes for k_parent_front (SYRK + factorisation in one launch)
    int parent_front_nacc = 0;           // 0: not available (front too large for the register-resident SYRK)
    size_t parent_front_lds = 0;
    bool parent_syrk = false, direct_parent = false;
    // Large fronts at the level of the leaves' parents (config 5: 528^2 per node, 36 GB in all) are never formed: F = I + U U^T with
    // U = the children's Ut blocks side by side has rank <= sum(n_obs), so only its panel columns [F_oo ; F_ao] = U U_o^T (+ I) are
    // built and factorised ([Lt ; Zt]), and the grandparents' fronts come straight from the leaves and the panels:
    //     F_grand = I + sum_leaves Ut[anc] Ut[anc]^T - sum_parents Zt Zt^T        (Schur complement = F_aa - Zt Zt^T, MRANode.py:476-480)
    // one signed, segmented SYRK instead of SYRK + Schur update + assembly of 528^2 blocks (95 GB of HBM traffic -> 37 GB).
    bool lowrank_parent = false;
    DevVec<GemmProb> gParentPanel, gGrandSyrk;
    DevVec<GemmSeg> grandSegs, parentSegsAo;
    DevVec<FrontProb> gParentOwn;         // k_parent_front on the own (cw x cw) block alone: F_oo = I + U_o U_o^T -> Lt, inverted diagonal blocks, log det
    DevVec<Trsm2Prob> gParentZt;          // Zt = F_ao Lt^-T, row solve of the panel's lower part in place
    size_t parent_own_lds = 0;
    Work fl_parent_panel, fl_grand_syrk;
    bool shape_regular = false;          // every leaf sits on the last level (all other levels hold non-leaf nodes only)
    std::vector<AsmChild> hKids;         // host copies: the leaves' Gt blocks are allocated only when something needs them
    std::vector<int> kid_leaf;
    std::vector<GemmProb> hLeafSyrk;
    int leaf_max_tiles_full = 0, leaf_max_tiles_lik = 0;
    DevVec<AsmChild> asmKids;
    long leaf_max_rows = 0;
    int leaf_max_nop = 0, leaf_max_na = 0, leaf_max_ht = 0;
    Work fl_leaf_resid, fl_leaf_chol, fl_leaf_chol_lik, fl_leaf_syrk, fl_leaf_update, fl_leaf_c_only;
    double by_leaf_ut = 0, by_leaf_tt = 0, by_leaf_c = 0;     // bytes of all leaves' Ut / Tt (= V) / C blocks
    std::vector<long> anc_rank;          // per node: sum of the TRUE ranks of its ancestors
    // fused ("regular tree") path
    // deep trees with 64-wide blocks (5 - 8 non-leaf levels of four tiles: BASELINE config 5): too many tiles per row for the one-kernel
    // cascades; the predictive pass runs as k_predict_hi (four deepest levels) + k_predict_cascade (coarse levels)
    bool var_accumulated = false;       // level-by-level path: the row solves have accumulated the prior / leaf variance (no k_leaf_moments pass)
    bool regular_hi = false;
    DevVec<long> hi_wg0_8;
    DevVec<int> hi_wgleaf;                    // [k_predict_hi workgroup] leaf slot of its tiles
    DevVec<int> hi_wgn_8;
    long n_hi_wg8 = 0;
    bool regular = false, use_fused = true, gemm_lds = true, use_front_fused = true, use_leaf_gemm = true, leaf_gemm_update = false;
    int dbg = 0;
    int NL = 0, CWT = 0;
    struct FusedLevel {
        DevVec<double> kx, Wk;
        DevVec<int> kvalid, kt_rows, kt_chain, kt_knot0, kt_wgn;
        DevVec<GemmProb> gKinv;
        DevVec<long> kt_wg0;
        long n_ktiles = 0, n_kwg = 0;
        int k_threads = 256;        // workgroup size of the level's knot launch (512 when sibling families share a workgroup)
    };
    std::vector<FusedLevel> fl;
    DevVec<long> ft_row0, ft_wg0, ft_wg0_x;
    DevVec<int> ft_chain, ft_wgn, ft_wgn_x, ft_wgleaf_x;
    long n_ftiles = 0, n_fwg = 0, n_fwg_x = 0;
    size_t cascade_lds = 0, cascade_lds_all = 0;
    bool cascade_stage_all = false;   // all levels' operands fit in LDS: one workgroup per leaf, staged once
    bool cascade_group_siblings = false;
    int n_cu = 256;
    // knot pass of all levels in one launch (k_knot_chain)
    bool use_knot_chain = true, knot_chain_ok = false;
    int kc_levels = 0;                    // levels 0 .. kc_levels-1 go through the chain kernel
    size_t knot_chain_lds = 0;
    DevVec<int> kc_chain;                 // [bottom slot][8]
    std::vector<int> kc_chain_host;
    DevVec<int> kc_ownmask;               // [bottom slot] bit m: the workgroup writes its level-m ancestor's results
    DevVec<double> kc_knots;              // [bottom slot][level][cw * (d + 1)] packed knots of the chain (mra_plan_set_locs)
    int cascade_wpw = 4;          // row tiles (= waves) per workgroup of the per-level cascade kernels (4 or 8; 4 measured faster)
    // likelihood-only passes: the row cascade walks gathered tiles of the leaves' OBSERVED rows only (obs_idx is the gather list);
    // per-tile chain / leaf and the workgroup list are built when such a pass first runs (ensure_lik_tiles)
    DevVec<int> lik_chain, lik_leaf, lik_wgn;
    DevVec<long> lik_wg0;
    long n_lik_wg = 0, n_lik_tiles = 0;
    bool lik_tiles_valid = false;
    // level-by-level path: the rows a likelihood needs = observed rows and knots, per leaf (padded to 16 with -1), leaves concatenated
    DevVec<int> need_idx;
    std::vector<unsigned char> y_finite_host; // [P] 1 where the row is observed (set_obs)
    bool lik_general_valid = false, lik_general_ok = false;
    bool use_lik_rows = true;                 // option 17
    std::vector<long> obs_off_host;           // [leaf + 1] offset of the leaf's list in obs_idx (multiples of 16)
    DevVec<long> ft_wg0_leaf;
    DevVec<int> ft_wgn_leaf;
    long n_fwg_leaf = 0;
    // host cov staging (MRA_KERNEL_HOST)
    std::vector<long> cov_off;           // per node offset into covsrc
    std::vector<double> cov_host, covdiag_host;
    // results
    double res_d = 0, res_u = 0;
    // timers
    hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    double phase_ms[5] = {0, 0, 0, 0, 0};
    bool ktiming = false;
    struct KStat { int launches = 0; double ms = 0, flops = 0, flops_exec = 0, bytes = 0; } kstat[KF_COUNT];
    std::vector<std::pair<int, std::pair<hipEvent_t, hipEvent_t>>> kev;
    // comm
    void* rccl = nullptr;
    ncclComm_t comm = nullptr;
    ncclResult_t (*allreduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    int n_ranks = 1, rank = 0;
    bool pass_open = false;              // a pass was started and has not reached finish_run (error or abandoned split)
    // kernels whose dynamic-LDS limit was raised for THIS plan's device (the attribute is per device, a plan lives on one)
    std::set<const void*> big_lds_done;
    bool prepare_only = false;           // mra_plan_prepare: the launch helpers set their attributes and return
};

// Dynamic LDS above 64 KiB has to be requested per kernel and per DEVICE.  The record is kept per plan (a plan lives on one
// device; mra_plan_prepare reports its size) AND per (device, kernel) for the process, so that the second plan of a process - the
// reference's MLE pattern builds a new MRATree per objective call - does not pay ~20 us per kernel again in its first pass.
bool mra_big_lds_once(int device, const void* fn);      // mra_plan.hip: true the first time this (device, kernel) pair is seen
static inline void ensure_big_lds(mra_plan* pl, std::initializer_list<const void*> fns) {
    for (const void* f : fns) {
        if (pl->big_lds_done.count(f)) continue;
        if (!g_dry && mra_big_lds_once(pl->device, f)) hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        pl->big_lds_done.insert(f);
    }
}


// ---- dispatchers defined in the mra_launch_*.hip translation units ----------------------------------------------------
// batched C (=|-=) f(A B^T) with epilogue EPI_SET / EPI_SUB / EPI_COV / EPI_HOSTCOV; lower_tri: every problem has .lower set and M == N
void mra_launch_gemm(mra_plan* pl, int epi, const GemmProb* probs, size_t nprob, long maxM, long maxN, bool allow_lds = true, bool lower_tri = false);
void mra_launch_leaf_gemm(mra_plan* pl, int epi, const GemmProb* probs, size_t nprob);
void mra_launch_syrk_blk(mra_plan* pl, const GemmProb* probs, size_t nprob, long M, bool dma);
void mra_launch_prior_level(mra_plan* pl, const GemmProb* probs, size_t nprob);
void launch_cascade_d1(mra_plan* pl, const CascadeArgs& ar);       // one translation unit per spatial dimension
void launch_cascade_d2(mra_plan* pl, const CascadeArgs& ar);
void launch_knot_chain_d1(mra_plan* pl, const KnotChainArgs& ka);
void launch_knot_chain_d2(mra_plan* pl, const KnotChainArgs& ka);
static inline void launch_cascade_any(mra_plan* pl, const CascadeArgs& ar) { if (pl->d == 1) launch_cascade_d1(pl, ar); else launch_cascade_d2(pl, ar); }
static inline void launch_knot_chain(mra_plan* pl, const KnotChainArgs& ka) { if (pl->d == 1) launch_knot_chain_d1(pl, ka); else launch_knot_chain_d2(pl, ka); }
void launch_predict_any(mra_plan* pl, const PredArgs& ar, size_t lds);
void launch_predict_hi(mra_plan* pl, const PredHiArgs& hi, const PredArgs& low, size_t lds_low, bool upd);
