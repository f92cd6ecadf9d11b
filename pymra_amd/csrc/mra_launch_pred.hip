// mra_launch_pred.hip - dispatch of the predictive cascade (k_predict_cascade) over block width, depth, workgroup shape and the
// folded leaf update; a translation unit of its own so that libmra_hip.so builds in parallel.
#define MRA_KERNELS_TEMPLATES_ONLY
#include "mra_plan_types.h"

// MINB workgroups per CU: three when both the LDS image and the register budget (168 with three waves per SIMD) allow it.
// MRA_PRED_GLDS: which operands take the LDS DMA path (bit 0 the level operands, bit 1 the Ut chunks of the leaf update)
#ifndef MRA_PRED_GLDS
#define MRA_PRED_GLDS 2
#endif
template <int CWT, int NLMAX, int WPW, int MINB>
static void launch_predict_cascade_w(mra_plan* pl, const PredArgs& ar, size_t lds) {
    ensure_big_lds(pl, {(const void*)k_predict_cascade<CWT, NLMAX, WPW, false, MINB, MRA_PRED_GLDS>, (const void*)k_predict_cascade<CWT, NLMAX, WPW, true, MINB, MRA_PRED_GLDS>});
    if (pl->prepare_only) return;
    if (ar.leaf_upd) hipLaunchKernelGGL((k_predict_cascade<CWT, NLMAX, WPW, true, MINB, MRA_PRED_GLDS>), dim3((unsigned)ar.n_wg), dim3(64 * WPW), lds, pl->stream, ar);
    else hipLaunchKernelGGL((k_predict_cascade<CWT, NLMAX, WPW, false, MINB, MRA_PRED_GLDS>), dim3((unsigned)ar.n_wg), dim3(64 * WPW), lds, pl->stream, ar);
}
template <int CWT, int NLMAX>
static void launch_predict_cascade(mra_plan* pl, const PredArgs& ar, size_t lds) {
    if (pl->cascade_wpw == 8) launch_predict_cascade_w<CWT, NLMAX, 8, 1>(pl, ar, lds);
    else if (pl->cascade_wpw == 4) {
        if (CWT <= 2 && NLMAX <= 6 && 3 * lds <= 160 * 1024 && !(pl->dbg & 8)) launch_predict_cascade_w<CWT, NLMAX, 4, (CWT <= 2 && NLMAX <= 6) ? 3 : 2>(pl, ar, lds);
        else launch_predict_cascade_w<CWT, NLMAX, 4, 2>(pl, ar, lds);
    }
    else throw MraError(MRA_ERR_STATE, "cascade_wpw must be 4 or 8");
}

void launch_predict_any(mra_plan* pl, const PredArgs& ar, size_t lds) {
    const int cwt = pl->CWT;
    if (cwt == 1) launch_predict_cascade<1, 8>(pl, ar, lds);
    else if (cwt == 2) { if (pl->NL <= 6) launch_predict_cascade<2, 6>(pl, ar, lds); else launch_predict_cascade<2, 8>(pl, ar, lds); }
    else launch_predict_cascade<4, 4>(pl, ar, lds);
}


// deep 64-wide trees: the four deepest levels + the sweep over the coarse columns, then the (at most four) coarse levels with
// eight row tiles per workgroup (k_predict_cascade<4, 4, 8, false, 1>: 236 registers, no scratch)
void launch_predict_hi(mra_plan* pl, const PredHiArgs& hi, const PredArgs& low, size_t lds_low, bool upd) {
    ensure_big_lds(pl, {(const void*)k_predict_hi<4, false>, (const void*)k_predict_hi<4, true>, (const void*)k_predict_cascade<4, 4, 8, false, 1, MRA_PRED_GLDS>});
    if (pl->prepare_only) return;
    // two chunk buffers of sixteen tiles (twenty with the leaf update: up to four k tiles of Ut ride behind the Zt tiles of the sweep:
    // 2 x 80 KB = all of a CU's LDS for two workgroups)
    if (upd) hipLaunchKernelGGL((k_predict_hi<4, true>), dim3((unsigned)pl->n_fwg), dim3(256), 2 * 20 * 256 * sizeof(double), pl->stream, hi);
    else hipLaunchKernelGGL((k_predict_hi<4, false>), dim3((unsigned)pl->n_fwg), dim3(256), 2 * 16 * 256 * sizeof(double), pl->stream, hi);
    hipLaunchKernelGGL((k_predict_cascade<4, 4, 8, false, 1, MRA_PRED_GLDS>), dim3((unsigned)low.n_wg), dim3(512), lds_low, pl->stream, low);
}
