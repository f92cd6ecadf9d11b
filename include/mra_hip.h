/*
 * mra_hip.h - C ABI of libmra_hip.so: the MI355X (gfx950) implementation of pyMRA's per-node
 * prior/posterior inference hot path.
 *
 * The reference (marcinjurek/pyMRA) is pure Python and has no FFI boundary of its own: all of the
 * work below happens inside its MRATree constructor.  Each entry point states which reference
 * interface (file:line under the reference checkout) it stands in for.  The Python facade that a
 * pyMRA user sees (pymra_amd.MRATree) binds these with ctypes; INTEGRATION.md shows the stub a
 * pyMRA maintainer would add.
 *
 * Conventions: every function returns 0 on success and a negative MRA_ERR_* code on failure and
 * never throws; mra_last_error() gives the message.  Host pointers passed in are copied before the
 * call returns and never retained.  Results are written only into caller-allocated host buffers.
 * A plan is bound to one GPU and one internal HIP stream; it is not thread-safe.
 * All floating point data is IEEE binary64, all row indices refer to the caller's padded,
 * leaf-ordered row space (see mra_topology).
 */
#ifndef MRA_HIP_H
#define MRA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MRA_OK                 0
#define MRA_ERR_INVALID       -1   /* bad argument / inconsistent topology                      */
#define MRA_ERR_HIP           -2   /* a HIP runtime call failed (no GPU, out of memory, ...)    */
#define MRA_ERR_NOT_SPD       -3   /* a Cholesky pivot was <= 0 or NaN (reference: pdb prompt,   */
                                   /*   pyMRA/MRANode.py:386-390)                               */
#define MRA_ERR_STATE         -4   /* call order violated (e.g. run before set_obs)             */
#define MRA_ERR_COMM          -5   /* RCCL failure                                              */

/* stationary kernels evaluated on the device (pyMRA/MRATools.py:256-301) */
#define MRA_KERNEL_EXP         0   /* ExpCovFun      exp(-D/l)                          :265-269 */
#define MRA_KERNEL_MATERN32    1   /* Matern32       sig (1+sqrt3 D/l) exp(-sqrt3 D/l)  :289-293 */
#define MRA_KERNEL_MATERN52    2   /* Matern52                                          :281-285 */
#define MRA_KERNEL_GAUSSIAN    3   /* GaussianCovFun sig exp(-D^2/(2 l^2))              :297-301 */
#define MRA_KERNEL_IDEN        4   /* Iden           [D == 0]                           :256-262 */
#define MRA_KERNEL_KANTER      5   /* KanterCovFun   compact taper, radius = l          :305-324 */
#define MRA_KERNEL_HOST        100 /* values supplied block by block (any cov callable or a dense
                                      matrix, pyMRA/MRANode.py:73-80, 381-384)                   */

/* mra_run flags */
#define MRA_RUN_LIKELIHOOD     1u  /* d and u                 (MRATree.getLikelihood, MRATree.py:82-84) */
#define MRA_RUN_PREDICT        2u  /* + predictive mean / var (MRATree.predict,       MRATree.py:90-94) */

typedef struct mra_plan mra_plan;

/*
 * The tree that pyMRA's Node.__init__ recursion (pyMRA/MRANode.py:23-115) would build, flattened.
 * Rows are in depth-first leaf order and every leaf's row range is padded to a multiple of 16
 * (phantom rows); node i owns rows [node_row0[i], node_row1[i]); nodes are numbered level by
 * level: level m owns nodes level_ptr[m] .. level_ptr[m+1]-1; node 0 is the root.
 * knot_rows[knot_ptr[i] .. knot_ptr[i+1]) are the padded-row positions of node i's knots
 * (pyMRA/MRANode.py:36-45, 179-205); for a leaf they are informational only.
 * cw[m] = knot-block width of the non-leaf nodes of level m, a multiple of 16 (0: none).
 */
typedef struct {
    int64_t        P;            /* padded rows                                    */
    int32_t        d;            /* spatial dimension, 1 or 2                      */
    int32_t        n_levels;
    int32_t        n_nodes;
    const int64_t *level_ptr;    /* [n_levels + 1]                                 */
    const int64_t *node_row0;    /* [n_nodes]                                      */
    const int64_t *node_row1;    /* [n_nodes]                                      */
    const uint8_t *node_leaf;    /* [n_nodes]                                      */
    const int32_t *node_parent;  /* [n_nodes], -1 for the root                     */
    const int32_t *child_ptr;    /* [n_nodes + 1]                                  */
    const int32_t *child_list;   /* [child_ptr[n_nodes]]                           */
    const int64_t *knot_ptr;     /* [n_nodes + 1]                                  */
    const int64_t *knot_rows;    /* [knot_ptr[n_nodes]]                            */
    const int32_t *cw;           /* [n_levels]                                     */
} mra_topology;

/* Number of usable GPUs (0 if none); never fails. */
int mra_device_count(void);

/* Replaces: the allocation side of MRATree.__init__ -> Node(...) (pyMRA/MRATree.py:69).
 * device: HIP device ordinal.  The topology arrays are copied. */
int mra_plan_create(mra_plan **plan, const mra_topology *topo, int device);
int mra_plan_destroy(mra_plan *plan);

/* Replaces: the `locs` argument of MRATree (pyMRA/MRATree.py:23-28).
 * locs_perm: P x d row-major, padded leaf order (phantom rows must hold a valid location). */
int mra_plan_set_locs(mra_plan *plan, const double *locs_perm);

/* Replaces: the `obs` and `R` arguments of MRATree (pyMRA/MRATree.py:23, 61-62; NaN = missing,
 * pyMRA/MRANode.py:415).  y_perm: P values, padded leaf order, NaN for missing and for phantom rows.
 * R: scalar nugget variance (the reference requires a Python float, pyMRA/MRANode.py:85-88). */
int mra_plan_set_obs(mra_plan *plan, const double *y_perm, double R);

/* Replaces: the `cov` callable when it is one of pyMRA's stationary kernels
 * (pyMRA/MRATools.py:256-301).  params = {l, sig, scale[, circular]}; value = scale * k(D; l, sig);
 * circular != 0 (1-D only): D = min(|a-b|, 1-|a-b|), the torus branch of dist() (MRATools.py:232-239).
 * kind == MRA_KERNEL_HOST (params ignored, call after mra_plan_set_obs): covariance values come from
 * mra_plan_set_cov_block. */
int mra_plan_set_kernel(mra_plan *plan, int kind, const double *params, int n_params);

/* Diagnostics: out[i] = scale * k_kind(D[i]; l, sig) evaluated by the device code that the inference
 * kernels inline - the device counterpart of ExpCovFun/Matern32/... applied to D = dist(locs, locs2)
 * (pyMRA/MRATools.py:229-301). */
int mra_eval_kernel(int kind, const double *params, int n_params, const double *D, int64_t n, double *out);

/* Replaces: the `cov` callable for anything else (MRA_KERNEL_HOST): the host evaluates
 * C(S_j, Q_j) for a non-leaf node (N_j x rank, row-major, pyMRA/MRANode.py:384) or
 * C(S_j, O_j) for a leaf (N_j x n_obs_j; O_j = the leaf's observed rows in ascending row order)
 * and hands it over.  diag: C(x,x) for each of the node's N_j rows (leaves only, else NULL). */
int mra_plan_set_cov_block(mra_plan *plan, int32_t node, const double *C, int64_t n_rows,
                           int64_t n_cols, const double *diag);

/* Replaces: everything Node.__init__ computes - calculatePrior (pyMRA/MRANode.py:378-395) for
 * every node top-down, calculatePosterior (:403-523) bottom-up.  Blocking. */
int mra_run(mra_plan *plan, uint32_t flags);

/* Replaces: MRATree.getLikelihood (pyMRA/MRATree.py:82-84): lik = *d + *u
 * (root.d, root.u of pyMRA/MRANode.py:456-468). */
int mra_get_likelihood(mra_plan *plan, double *d, double *u);

/* Replaces: MRATree.predict (pyMRA/MRATree.py:90-94): root.mean and root.var
 * (pyMRA/MRANode.py:510-520) in padded leaf order, P values each. */
int mra_get_predict(mra_plan *plan, double *mean_perm, double *var_perm);

/* Diagnostics for tests (the reference exposes these as attributes of Node objects):
 * what = 0: whitened basis W (P x ldw, row-major) ; 1: per-node log-det terms (n_nodes);
 * copies min(capacity, available) doubles into out, returns the available count in *n_avail.
 * (W after a likelihood-only run is complete only with MRA_OPT_LIK_ROWS off: by default such a run computes W at the rows a
 * likelihood needs - the observed rows, and the knots on the level-by-level path - and leaves the others as they were.) */
int mra_get_buffer(mra_plan *plan, int what, double *out, int64_t capacity, int64_t *n_avail);

/* Replaces: the per-node attributes the reference leaves on its Node objects - B, kInv (pyMRA/MRANode.py:384-385),
 * kTil, A, omg (:426-445), BTil (:486-495) - which its diagnostics read (MRATree.getBasisFunctionsMatrix,
 * pyMRA/MRATree.py:445-511; pyMRA/tests/debug-posterior.py:97-109).  The device keeps them in whitened form
 * (DESIGN.md section 3); this call copies one node's raw block, pymra_amd.diagnostics de-whitens on the host:
 *   MRA_BLOCK_W_ROWS  the node's rows of the whitened basis array W, all ldw columns (N_j x ldw).  After a
 *                     likelihood-only run (MRA_OPT_LIK_ROWS off): the prior W (B_k[S_j] = W[:, block k] L_k^T).  After a predict run with
 *                     MRA_OPT_FUSED off: block m of a level-m non-leaf node's rows holds X = BTil_j[m] (L_j Lt_j)^-T
 *                     (X X^T = BTil[m] kTil BTil[m]^T), the blocks of coarser levels hold the rows of BTil[k] L_k^-T
 *                     as the node's parent sees them.
 *   MRA_BLOCK_LPRIOR  non-leaf: L_j, cw x cw lower Cholesky factor of kInv_j (padded with the identity)
 *   MRA_BLOCK_FRONT   non-leaf: the factorised front, nf x nf (lower triangle): Lt_j (cw x cw), Zt_j below it
 *                     (na x cw), the Schur block Gt_j (na x na) - whitened ATil / omgTil / u (last block = y)
 *   MRA_BLOCK_LEAF    leaf: the panel [C = v_m(o,o) + R I -> Lc ; Ut ; Tt], (nop + na + N_j) x nop
 * out receives min(capacity, rows*cols) doubles row-major; *n_rows, *n_cols the block shape. */
#define MRA_BLOCK_W_ROWS   0
#define MRA_BLOCK_LPRIOR   1
#define MRA_BLOCK_FRONT    2
#define MRA_BLOCK_LEAF     3
int mra_get_node_block(mra_plan *plan, int32_t node, int what, double *out, int64_t capacity,
                       int64_t *n_rows, int64_t *n_cols);

/* Device blocks of destroyed plans are cached per (device, size) and reused by later plans of the process - the reference's
 * MLE pattern builds a new MRATree per objective call (README.md:96-104).  This hands every cached block back to the driver. */
int mra_release_cached_memory(void);

/* Caller-order variants of mra_plan_set_locs / mra_plan_set_obs / mra_get_predict: the arrays are in the CALLER's row order
 * (locs N x d, y N; NaN = missing), src[P] / perm[P] / in_leaf[P] are the topology's row maps (padded row -> caller row it
 * copies; -1 in perm for a phantom row; in_leaf = row is reported).  The gather / scatter runs inside the library (threads +
 * a pinned staging area), which is what an end-to-end MRATree(...) call pays for instead of NumPy fancy indexing and
 * pageable copies (MRATree.__init__, pyMRA/MRATree.py:61-69, hands the arrays over in caller order). */
int mra_plan_set_locs_rows(mra_plan *plan, const double *locs, const int64_t *src);
int mra_plan_set_obs_rows(mra_plan *plan, const double *y, const int64_t *src, const int64_t *perm, double R);
int mra_get_predict_rows(mra_plan *plan, const int64_t *perm, const uint8_t *in_leaf, int64_t N, double *mean, double *var);
/* The same with sd = sqrt(var) written beside it (MRATree.predict returns np.sqrt(root.var), pyMRA/MRATree.py:93); sd may be NULL. */
int mra_get_predict_rows_sd(mra_plan *plan, const int64_t *perm, const uint8_t *in_leaf, int64_t N, double *mean, double *var,
                            double *sd);

/* Per-phase device milliseconds of the last mra_run (hipEvent deltas on the plan's stream):
 * out[0]=prior, [1]=leaf, [2]=fronts, [3]=predict, [4]=total; returns how many were written.
 * The four phase entries are measured only while MRA_OPT_KERNEL_TIMING is on (0 otherwise): an event
 * between dependent launches costs a few microseconds of stream time; the total is always measured. */
int mra_get_timers(mra_plan *plan, double *out_ms, int capacity);

/* Kernel-level timing: with option MRA_OPT_KERNEL_TIMING on, every launch of mra_run is bracketed by
 * hipEvents on the plan's stream; afterwards mra_get_kernel_stats reports, per kernel family
 * (0 .. mra_kernel_family_count()-1), its name, the number of launches, their summed device
 * milliseconds and the algorithmic flop count of those launches. */
#define MRA_OPT_KERNEL_TIMING  1
#define MRA_OPT_GEMM_LDS       3   /* 1 (default): LDS-tiled batched GEMM; 0: direct-load variant */
#define MRA_OPT_FUSED          2   /* 1 (default): fused cascade kernels on regular trees; 0: level-by-level kernels */
#define MRA_OPT_KNOT_CHAIN     5   /* 1 (default): knot pass of all levels in one launch (k_knot_chain); 0: one launch per level */
#define MRA_OPT_LEAF_GEMM      6   /* 1 (default): leaf-resident residual product (k_leaf_gemm, one workgroup per leaf); 2: also the leaf
                                      update; 0: 64x64-tile k_gemm_nt_lds for both */
#define MRA_OPT_LEAF_SOLVE     7   /* row solve Tt = V Lc^-T and leaf update in one launch (k_leaf_solve_update, Tt stays in registers):
                                      2 (default) when a CU sees at most two leaves (8-way sharded runs), 1 always, 0 never */
#define MRA_OPT_PRED_UPDATE    8   /* 1 (default): the leaf update is applied inside the predictive cascade (W is not rewritten); 0: separate product */
#define MRA_OPT_LEAF_SOLVE_SPLIT 10 /* 2 (default): the fused row solve + update runs two workgroups per leaf (half the row tiles each), so that
                                      the side stream frees CUs sooner for the front chain; 1: one workgroup per leaf */
#define MRA_OPT_CHOL_TILES     11  /* leaf Cholesky by one workgroup per matrix with the tiles in registers (k_chol_tiles): 1 (default) when a CU sees
                                      at most two leaves of at most 160 observations (sharded runs), 2 always, 0 never (one wave per matrix, k_chol_wave) */
#define MRA_OPT_SEG_GEMM_LDS   12  /* 1 (default): the panel columns of the leaves' parents (a segmented product: sum over the children's Ut blocks; deep
                                      64-wide trees) run on the LDS-tiled GEMM; 0: on the direct-load GEMM */
#define MRA_OPT_UT_GATHER      13  /* 1 (default): the leaves' Ut = [W_anc[o] | y_o]^T is gathered from W by the row solve (128-byte segments); 0: scattered
                                      by the prior row cascade in 8-byte pieces (the round-2 path: 0.02 ms slower per C3 pass) */
#define MRA_OPT_FRONT_FUSED    4   /* 1 (default): one LDS-resident launch per front level; 0: assemble / Cholesky / Schur launches */
#define MRA_OPT_SYRK_BLK       14  /* 1 (default): the grandparents' signed segmented SYRK of deep 64-wide trees on 96 x 96 blocks through LDS,
                                      the stage filled by LDS DMA (k_syrk_dma); 2: through registers (k_syrk_blk); 0: 32 x 32 wave tiles (k_gemm_nt) */
#define MRA_OPT_PRIOR_LEVEL    15  /* 1 (default): level-by-level path, blocks <= 64 wide: the prior of a level in one launch (residual product,
                                      kernel, row solve; the knots' block straight into its factor); 0: four launches per level */
#define MRA_OPT_HI_FOLD        16  /* 1 (default): deep 64-wide trees on one GPU: the leaf update W -= Tt Ut^T inside k_predict_hi (leaves of at most
                                      64 padded observations); 0: a product of its own over all of W; 2: inside k_predict_hi on a sharded rank too */
#define MRA_OPT_LIK_ROWS       17  /* 1 (default): likelihood-only passes of the fused path compute the whitened basis W at the OBSERVED rows only
                                      (gathered row tiles: a likelihood needs nothing else; W's other rows keep what an earlier pass left there);
                                      0: at every row - what mra_get_buffer(W) callers and the node-block diagnostics want */
int mra_plan_set_option(mra_plan *plan, int option, int64_t value);
/* current value of an option (so that a caller can change one temporarily and put it back) */
int mra_plan_get_option(mra_plan *plan, int option, int64_t *value);
/* Raise the dynamic-LDS limit (160 KB) of every kernel the plan may launch on the plan's own device now rather than at
 * first launch.  The attribute is per kernel AND per device, so it is recorded per plan: two MRATree objects on two
 * devices of one process each take this path.  *n_kernels (may be NULL): kernels on record for this plan. */
int mra_plan_prepare(mra_plan *plan, int64_t *n_kernels);
int mra_kernel_family_count(void);
int mra_get_kernel_stats(mra_plan *plan, int which, char *name, int name_cap, int *launches, double *ms,
                         double *flops);
/* Work of the family's launches in the last mra_run: out[0] algorithmic flops (true ranks, true observation counts, one y
 * column), out[1] flops the MFMA tiles execute on the 16-padded layout, out[2] algorithmic HBM bytes (every operand array of
 * the launch read or written once), out[3] summed device milliseconds (as mra_get_kernel_stats). */
int mra_get_kernel_work(mra_plan *plan, int which, double *out, int capacity);
/* hipDeviceSynchronize on `device` (bench.py's barrier: the timed region needs no second GPU runtime in the process) */
int mra_device_synchronize(int device);
/* out[0..7] = P, ldw, Ka, n_leaves, bytes(W), bytes(leaf panels), bytes(leaf Gt), n_nodes */
int mra_plan_info(mra_plan *plan, int64_t *out, int capacity);

/* ---- native host tree replay (no GPU involved) -------------------------------------------------------
 * Replaces, for large 2-D trees (every node above the leaves has > 100 rows and > 100 candidates, J = 4): the
 * knot selection and partition part of Node.__init__ (pyMRA/MRANode.py:34-59, 191-193, 232-239) and the flat
 * layout of pymra_amd.topology.  mt_key[624] / mt_pos are NumPy's global MT19937 state
 * (np.random.get_state()[1:3]); on success they hold the state after the reference's knot draws.
 * Returns 0, or 1 when the tree does not follow those rules (state untouched; use the Python replay). */
typedef struct mra_tree mra_tree;
int mra_tree_replay_2d(const double *locs, int64_t N, int32_t r, int32_t M, uint32_t *mt_key, int32_t *mt_pos,
                       mra_tree **out);
/* The same replay writing the four long arrays (perm, src, in_leaf: P <= cap_rows entries; knot_rows: N entries) straight into
 * caller buffers of capacity cap_rows >= N + 15 * 4^M (every leaf is padded to a multiple of 16 rows); mra_tree_export then
 * skips them (pass NULL).  Saves ~25 MB of copies and unmapping per tree: an end-to-end MRATree(...) call is mostly this. */
int mra_tree_replay_2d_into(const double *locs, int64_t N, int32_t r, int32_t M, uint32_t *mt_key, int32_t *mt_pos,
                            int64_t cap_rows, int64_t *perm, int64_t *src, uint8_t *in_leaf, int64_t *knot_rows,
                            mra_tree **out);
/* MRATree.__init__ for large 2-D trees in ONE call (pyMRA/MRATree.py:61-69 -> Node.__init__, MRANode.py:23-115): the replay
 * of mra_tree_replay_2d_into AND mra_plan_create + mra_plan_set_locs_rows + mra_plan_set_obs_rows, overlapped - the plan is sized,
 * allocated and fed its locations (locs: the caller's N x 2 rows) and observations (y: N values, NaN = missing; R: nugget) on a
 * helper thread while the calling thread draws the knots.  On success *tree_out is the tree (mra_tree_sizes / mra_tree_export /
 * mra_tree_free as usual; perm, src, in_leaf, knot_rows are written to the caller's buffers) and *plan_out a plan that only
 * needs mra_plan_set_kernel before mra_run.  Returns 1 - nothing created, RNG state untouched - when the tree does not follow
 * the large-2-D rules (use mra_tree_replay / the Python replay and mra_plan_create then), < 0 on errors. */
int mra_plan_create_replay_2d(const double *locs, int64_t N, int32_t r, int32_t M, uint32_t *mt_key, int32_t *mt_pos,
                              const double *y, double R, int device, int64_t cap_rows,
                              int64_t *perm, int64_t *src, uint8_t *in_leaf, int64_t *knot_rows,
                              mra_tree **tree_out, mra_plan **plan_out);
/* out5 = {P, n_nodes, n_levels, len(child_list), len(knot_rows)} */
int mra_tree_sizes(mra_tree *t, int64_t *out5);
/* copies the arrays of `mra_topology` (+ perm, src, in_leaf, node_level, pre-order) into caller buffers */
int mra_tree_export(mra_tree *t, int64_t *perm, int64_t *src, uint8_t *in_leaf, int64_t *level_ptr,
                    int32_t *node_level, int64_t *row0, int64_t *row1, uint8_t *leaf, int32_t *parent,
                    int32_t *child_ptr, int32_t *child_list, int64_t *knot_ptr, int64_t *knot_rows, int32_t *cw,
                    int32_t *preorder);
int mra_tree_free(mra_tree *t);

/* ---- multi-GPU: one process per GPU, subtrees sharded, ONE all-reduce of the shard-level fronts --
 * The reference's only parallel mode forks one process per child subtree at `critDepth` and pickles
 * the finished Node back (pyMRA/MRANode.py:64-65, 90-104, 114-115); here each rank owns whole
 * subtrees and the level above them is summed with one ncclAllReduce over xGMI.               */
int mra_comm_unique_id(char *out, int capacity);                 /* 128 bytes; rank 0 calls, then broadcast */
int mra_comm_init(mra_plan *plan, const char *unique_id, int n_ranks, int rank);
/* reduce_level: nodes of this level get their assembled fronts summed over ranks before they are
 * factorised (-1: no reduction). */
int mra_plan_set_reduce_level(mra_plan *plan, int level);

/* Front exchange without RCCL (tests on one GPU, or another transport): mra_run stops before the
 * reduce level's factorisation when MRA_RUN_SPLIT is set in flags; export/import the summed
 * fronts, then call mra_run_resume. */
#define MRA_RUN_SPLIT          4u
int mra_reduce_size(mra_plan *plan, int64_t *n_doubles);
int mra_reduce_export(mra_plan *plan, double *out);
int mra_reduce_import(mra_plan *plan, const double *in);
int mra_run_resume(mra_plan *plan);

const char *mra_last_error(mra_plan *plan);   /* plan may be NULL: last create error */
const char *mra_version(void);

#ifdef __cplusplus
}
#endif
#endif
