// mra_plan.hip - host side of libmra_hip.so: plan construction, batched-problem descriptors, the
// launch sequence of one inference pass and the C ABI of include/mra_hip.h.
//
// One pass (mra_run) = what pyMRA's Node.__init__ recursion computes (pyMRA/MRANode.py:23-115):
//   1. prior, top-down per level    (calculatePrior, MRANode.py:378-395, with the conditional
//                                    covariance of :73-80)            -> whitened basis W
//   2. leaves, observation space    (leaf branch of calculatePosterior, :411-430, :450-459)
//   3. non-leaf fronts, bottom-up   (:432-480)
//   4. predictive moments, bottom-up(:486-520)
// DESIGN.md section 3 derives the factorised form; oracle/mra_levelwise.py is its NumPy twin.
#include "mra_plan_types.h"
#include "mra_topology.h"
#include <atomic>
#include <chrono>
#include <mutex>
#include <thread>

// MRA_TRACE_PLAN=1: host-side phase times of plan construction on stderr (tools/e2e_breakdown.py)
namespace {
struct PlanTrace {
    bool on;
    double t0, last;
    const char* what;
    static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    explicit PlanTrace(const char* w) : on(getenv("MRA_TRACE_PLAN") != nullptr), t0(now()), last(t0), what(w) {}
    void mark(const char* label) { if (on) { const double t = now(); fprintf(stderr, "  [%s] %-34s %7.2f ms\n", what, label, t - last); last = t; } }
    ~PlanTrace() { if (on) fprintf(stderr, "  [%s] total %.2f ms (descriptor uploads of this thread so far: %ld, %.2f ms)\n", what, now() - t0, upload_stats().n, upload_stats().ms); }
};
}  // namespace

// ---- device memory cache --------------------------------------------------------------------------------------------
#include <map>
#include <unordered_map>
namespace {
struct DevPool {
    std::mutex mu;
    std::unordered_map<void*, std::pair<int, size_t>> live;           // pointer -> (device, bytes) of pooled-size blocks in use
    std::multimap<std::pair<int, size_t>, void*> idle;                // (device, bytes) -> cached block
    size_t idle_bytes = 0;
    static constexpr size_t MIN_BLOCK = 256 * 1024;                   // smaller blocks are not worth keeping
    // Idle bytes above the cap go back to the driver at once (another runtime in the process - RCCL, a caller's torch - must not
    // run out of memory while blocks sit here).  MRA_POOL_MAX_GB sets it (0 switches the cache off); the default is 16 GB or an
    // eighth of the device's memory, whichever is smaller: four C3-sized plans, not a config-5 one (18 GB of W alone - an MLE
    // loop at that size should keep its plan and call mra_plan_set_kernel, as bench.py does).
    size_t max_idle() {
        if (cap_known) return cap;
        cap = (size_t)16 << 30;
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) == hipSuccess && tot / 8 < cap) cap = tot / 8;
        if (const char* e = getenv("MRA_POOL_MAX_GB")) { const double g = atof(e); cap = g > 0 ? (size_t)(g * (double)((size_t)1 << 30)) : 0; }
        cap_known = true;
        return cap;
    }
    size_t cap = 0;
    bool cap_known = false;
    void flush_locked() {
        for (auto& e : idle) hipFree(e.second);
        idle.clear();
        idle_bytes = 0;
    }
    ~DevPool() { /* process exit: the driver reclaims everything; calling into HIP from a static destructor is not safe */ }
};
DevPool g_pool;
// MRA_POOL_POISON=1 (debug): every block handed out - fresh or reused - is filled with NaNs first, so that a buffer that relies on
// zero-initialised or left-over contents shows up in the results (tests/test_gpu_parity.py runs the oracle parity cases once this way)
inline bool pool_poison() { const char* e = getenv("MRA_POOL_POISON"); return e && e[0] && e[0] != '0'; }
inline hipError_t poison_block(void* p, size_t n) {
    if (!pool_poison() || !p || !n) return hipSuccess;
    hipError_t e = hipMemset(p, 0xFF, n);                              // 0xFFFF...: a quiet NaN in every double, -1 in every int
    if (e == hipSuccess) e = hipDeviceSynchronize();
    return e;
}
}  // namespace

hipError_t mraMalloc(void** p, size_t n) {
    if (g_dry) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
    if (n < DevPool::MIN_BLOCK) { hipError_t e = hipMalloc(p, n); return e == hipSuccess ? poison_block(*p, n) : e; }
    int dev = 0;
    hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(g_pool.mu);
    auto it = g_pool.idle.find({dev, n});
    if (it != g_pool.idle.end()) {
        *p = it->second;
        g_pool.idle.erase(it);
        g_pool.idle_bytes -= n;
        g_pool.live[*p] = {dev, n};
        return poison_block(*p, n);
    }
    hipError_t e = hipMalloc(p, n);
    if (e != hipSuccess && !g_pool.idle.empty()) {                    // out of memory with blocks cached: give them back and retry
        (void)hipGetLastError();
        g_pool.flush_locked();
        e = hipMalloc(p, n);
    }
    if (e == hipSuccess) { g_pool.live[*p] = {dev, n}; e = poison_block(*p, n); }
    return e;
}

hipError_t mraFree(void* p) {
    if (g_dry) { free(p); return hipSuccess; }
    if (!p) return hipSuccess;
    std::lock_guard<std::mutex> lock(g_pool.mu);
    auto it = g_pool.live.find(p);
    if (it == g_pool.live.end()) return hipFree(p);
    const std::pair<int, size_t> key = it->second;
    g_pool.live.erase(it);
    if (g_pool.idle_bytes + key.second > g_pool.max_idle()) return hipFree(p);
    g_pool.idle.insert({key, p});
    g_pool.idle_bytes += key.second;
    return hipSuccess;
}

// ---- stream cache: creating the two prioritised streams of a plan costs milliseconds (each is a hardware queue), destroying them
// as much; a process that builds a new MRATree per objective call (README.md:96-104) gets the pair of the previous plan back
namespace {
struct StreamPair { hipStream_t hi, lo; double* host_res; double* host_res_dev; char* arena_host; };      // + the pinned result record and the pinned mirror of the descriptor arena
std::mutex g_streams_mu;
std::multimap<int, StreamPair> g_streams;                           // device -> idle pair (both synchronised when they were returned)
}  // namespace
static void acquire_streams(mra_plan* pl) {
    {
        std::lock_guard<std::mutex> lock(g_streams_mu);
        auto it = g_streams.find(pl->device);
        if (it != g_streams.end()) {
            pl->stream = it->second.hi; pl->stream2 = it->second.lo; pl->host_res = it->second.host_res; pl->host_res_dev = it->second.host_res_dev;
            pl->arena.host = it->second.arena_host;
            g_streams.erase(it);
            return;
        }
    }
    int lo = 0, hi = 0;
    HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
    HIP_TRY(hipStreamCreateWithPriority(&pl->stream, hipStreamDefault, hi));
    HIP_TRY(hipStreamCreateWithPriority(&pl->stream2, hipStreamDefault, lo));
}
static void return_streams(mra_plan* pl) {
    if (!pl->stream || !pl->stream2) {
        if (pl->stream) hipStreamDestroy(pl->stream);
        if (pl->stream2) hipStreamDestroy(pl->stream2);
    } else {
        std::lock_guard<std::mutex> lock(g_streams_mu);
        if (g_streams.count(pl->device) < 4) { g_streams.insert({pl->device, StreamPair{pl->stream, pl->stream2, pl->host_res, pl->host_res_dev, pl->arena.host}}); pl->host_res = nullptr; pl->arena.host = nullptr; }
        else { hipStreamDestroy(pl->stream); hipStreamDestroy(pl->stream2); }
    }
    if (pl->host_res) hipHostFree(pl->host_res);
    if (pl->arena.host) { hipHostFree(pl->arena.host); pl->arena.host = nullptr; }
    pl->host_res = pl->host_res_dev = nullptr;
    pl->stream = pl->stream2 = nullptr;
}

// a few host threads over a range of rows (or of leaves: min_per_thread is the smallest share worth a thread)
template <class F>
static void parallel_rows(int64_t n, F fn, int64_t min_per_thread = 65536) {
    const int T = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(4, std::thread::hardware_concurrency()), n / min_per_thread));
    if (T <= 1) { fn(0, n); return; }
    std::vector<std::thread> pool;
    for (int t = 0; t < T; ++t) pool.emplace_back(fn, n * t / T, n * (t + 1) / T);
    for (auto& th : pool) th.join();
}

// the plan's descriptor arena: a 16 MB device block (from the block cache) and its pinned mirror (kept with the stream pair)
static void init_arena(mra_plan* pl) {
    UploadArena& a = pl->arena;
    a.cap = (size_t)16 << 20;
    a.used = a.flushed = 0;
    if (mraMalloc((void**)&a.dev, a.cap) != hipSuccess) { a.dev = nullptr; a.cap = 0; return; }       // no arena: stand-alone uploads
    if (!a.host) {
        if (g_dry) a.host = (char*)malloc(a.cap);
        else if (hipHostMalloc((void**)&a.host, a.cap, hipHostMallocDefault) != hipSuccess) a.host = nullptr;
    }
    if (!a.host) { mraFree(a.dev); a.dev = nullptr; a.cap = 0; }
}
static void drop_arena(mra_plan* pl) {
    if (pl->arena.dev) mraFree(pl->arena.dev);
    pl->arena.dev = nullptr;
    if (g_dry && pl->arena.host) { free(pl->arena.host); pl->arena.host = nullptr; }
}

bool mra_big_lds_once(int device, const void* fn) {
    static std::mutex mu;
    static std::set<std::pair<int, const void*>> done;
    std::lock_guard<std::mutex> lock(mu);
    return done.insert({device, fn}).second;
}

static void derive_kernel_params(KernelParams& kp) {
    kp.mode = 0; kp.a1 = 0.0; kp.a2 = 0.0; kp.amp = kp.scale * kp.sig;
    kp.inv_2l2 = 1.0 / (2.0 * (kp.l * kp.l));
    double c = 1.0;
    switch (kp.kind) {
        case 0: kp.amp = kp.scale; break;                       // ExpCovFun has no sig
        case 1: c = 1.7320508075688772; kp.a1 = 1.0; break;
        case 2: c = 2.23606797749979; kp.a1 = 1.0; kp.a2 = 1.0 / 3.0; break;
        case 3: kp.mode = 1; break;
        case 5: kp.mode = 3; kp.amp = kp.scale; break;                // KanterCovFun(radius = l)
        default: kp.mode = 2; kp.amp = kp.scale; break;
    }
    kp.c_inv_l = c / kp.l;
}

static int fail(mra_plan* p, const MraError& e) {
    if (p) p->err = e.msg;
    g_last_error = e.msg;
    return e.code;
}

static void fill_knot_arrays(mra_plan* pl);

// ------------------------------------------------------------------------------------------------
static void build_static(mra_plan* pl) {
    PlanTrace tr("build_static");
    ArenaScope arena(&pl->arena);
    const int L = pl->n_levels;
    pl->Ka = 0;
    for (int m = 0; m < L; ++m) {
        if (pl->cw[m] % 16) throw MraError(MRA_ERR_INVALID, "cw must be a multiple of 16");
        pl->Ka += pl->cw[m];
    }
    pl->ldw = pl->Ka + MRA_YB;
    pl->coff.assign(L, 0); pl->asuf.assign(L, 0); pl->nf.assign(L, 0); pl->na.assign(L, 0);
    for (int m = 0; m < L; ++m) {
        int s = 0;
        for (int k = m + 1; k < L; ++k) s += pl->cw[k];
        pl->coff[m] = s;
        pl->asuf[m] = s + pl->cw[m];
        pl->nf[m] = pl->ldw - pl->coff[m];
        pl->na[m] = pl->ldw - pl->asuf[m];
    }
    pl->node_level.assign(pl->n_nodes, 0);
    for (int m = 0; m < L; ++m)
        for (long i = pl->level_ptr[m]; i < pl->level_ptr[m + 1]; ++i) pl->node_level[i] = m;
    pl->anc_rank.assign(pl->n_nodes, 0);
    for (int i = 0; i < pl->n_nodes; ++i) {             // nodes are numbered level by level: a parent precedes its children
        const int pa = pl->parent[i];
        if (pa >= 0 && pa < i) pl->anc_rank[i] = pl->anc_rank[pa] + (pl->knot_ptr[pa + 1] - pl->knot_ptr[pa]);
    }
    for (int i = 0; i < pl->n_nodes; ++i) {
        if ((pl->row0[i] % 16) || (pl->row1[i] % 16) || pl->row1[i] < pl->row0[i] || pl->row1[i] > pl->P)
            throw MraError(MRA_ERR_INVALID, "node row ranges must be 16-aligned and inside [0,P)");
        const int m = pl->node_level[i];
        const long rk = pl->knot_ptr[i + 1] - pl->knot_ptr[i];
        if (!pl->leaf[i] && (rk > pl->cw[m] || rk <= 0)) throw MraError(MRA_ERR_INVALID, "non-leaf rank must be in 1..cw[level]");
    }

    // device arrays shared by everything
    pl->X.alloc((size_t)pl->P * pl->d);
    pl->y.alloc(pl->P);
    pl->W.alloc((size_t)pl->P * pl->ldw);
    pl->var.alloc(pl->P);
    pl->mean.alloc(pl->P);
    pl->dnode.alloc(pl->n_nodes);
    pl->scal.alloc(4);
    pl->errflag.alloc(1);
    HIP_TRY(mraMemset(pl->errflag.p, 0, sizeof(int)));       // k_sum_dnode reads it and clears it again at the end of every pass
    HIP_TRY(mraMemset(pl->W.p, 0, pl->W.n * sizeof(double)));
    HIP_TRY(mraMemset(pl->var.p, 0, pl->var.n * sizeof(double)));
    HIP_TRY(mraMemset(pl->dnode.p, 0, pl->dnode.n * sizeof(double)));
    pl->knots_dev.alloc((size_t)std::max<long>(pl->knot_ptr.back(), 1));       // contents: fill_knot_arrays
    tr.mark("checks, allocations, memsets");

    // knot index arrays for the gather side of the prior GEMM (padded to cw with -1): offsets here, contents in fill_knot_arrays
    pl->knot_idx_off.assign(pl->n_nodes, -1);
    {
        long tot = 0;
        for (int i = 0; i < pl->n_nodes; ++i) {
            if (pl->leaf[i]) continue;
            pl->knot_idx_off[i] = tot;
            tot += pl->cw[pl->node_level[i]];
        }
        pl->knot_idx.alloc((size_t)std::max<long>(tot, 1));
    }

    // per level buffers + static descriptors
    pl->lev.clear();
    pl->lev.resize(L);
    pl->node_slot.assign(pl->n_nodes, -1);
    {
        // shape: every leaf on the last level, every other level non-leaf only (also decided again below for the fused paths)
        bool ok = L >= 3;
        for (int m = 0; m < L - 1 && ok; ++m)
            for (long i = pl->level_ptr[m]; i < pl->level_ptr[m + 1] && ok; ++i) if (pl->leaf[i]) ok = false;
        for (long i = pl->level_ptr[L - 1]; i < pl->level_ptr[L] && ok; ++i) if (!pl->leaf[i]) ok = false;
        // fronts of the leaves' parents too large for the register-resident k_parent_front: keep their panels only
        pl->lowrank_parent = ok && pl->nf[L - 2] > 256 && pl->cw[L - 2] <= 7 * 16 && !getenv("MRA_NO_LOWRANK_PARENT");
    }
    for (int m = 0; m < L; ++m) {
        LevelData& lv = pl->lev[m];
        lv.cw = pl->cw[m]; lv.cwt = lv.cw / 16; lv.c0 = pl->coff[m]; lv.a0 = pl->asuf[m];
        lv.nf = pl->nf[m]; lv.na = pl->na[m];
        lv.panel_only = pl->lowrank_parent && m == L - 2;
        lv.ldf = lv.panel_only ? lv.cw : lv.nf;
        for (long i = pl->level_ptr[m]; i < pl->level_ptr[m + 1]; ++i)
            if (!pl->leaf[i]) { pl->node_slot[i] = (int)lv.nodes.size(); lv.nodes.push_back((int)i); }
        const size_t nn = lv.nodes.size();
        if (!nn) continue;
        if (lv.cw == 0) throw MraError(MRA_ERR_INVALID, "level with non-leaf nodes has cw == 0");
        lv.Lp.alloc(nn * (size_t)lv.cw * lv.cw);
        lv.invP.alloc(nn * (size_t)lv.cwt * 256);
        lv.F.alloc(nn * (size_t)lv.nf * lv.ldf + 16);
        lv.invF.alloc(nn * (size_t)lv.cwt * 256);
    }
    tr.mark("knot_idx, level buffers");
    // leaves
    pl->leaf_nodes.clear();
    pl->leaf_slot.assign(pl->n_nodes, -1);
    for (int i = 0; i < pl->n_nodes; ++i)
        if (pl->leaf[i]) { pl->leaf_slot[i] = (int)pl->leaf_nodes.size(); pl->leaf_nodes.push_back(i); }
    // (row_leaf - padded row -> leaf, 4 bytes per row - is read by two kernels of the general path only: built on first use, ensure_row_leaf)
    // Gt of the leaves
    pl->leaf_goff.assign(pl->leaf_nodes.size() + 1, 0);
    for (size_t t = 0; t < pl->leaf_nodes.size(); ++t) {
        const int na = pl->na[pl->node_level[pl->leaf_nodes[t]]];
        pl->leaf_goff[t + 1] = pl->leaf_goff[t] + (long)na * na;
    }
    {
        const int Lq = pl->n_levels;
        bool ok = Lq >= 2;
        for (int m = 0; m < Lq - 1 && ok; ++m)
            for (long i = pl->level_ptr[m]; i < pl->level_ptr[m + 1] && ok; ++i) if (pl->leaf[i]) ok = false;
        for (long i = pl->level_ptr[Lq - 1]; i < pl->level_ptr[Lq] && ok; ++i) if (!pl->leaf[i]) ok = false;
        pl->shape_regular = ok;
        if (ok) pl->NL = Lq - 1;
    }
    // shape-regular trees build the leaves' parents straight from the children's Ut blocks (segmented SYRK),
    // so the per-leaf Schur blocks Gt (146 GB at 2048^2, M=8, r0=64) are only allocated on demand (ensure_gt)
    if (!pl->shape_regular) pl->Gt.alloc(pl->leaf_goff.back());

    tr.mark("leaf maps (row_leaf upload)");
    // descriptors that do not depend on the observations
    std::vector<AsmChild> kids;
    pl->kid_leaf.clear();
    for (int m = 0; m < L; ++m) {
        LevelData& lv = pl->lev[m];
        const size_t nn = lv.nodes.size();
        if (!nn) continue;
        std::vector<GemmProb> resid(nn), schur(nn), upd(nn);
        std::vector<KinvProb> kinv(nn);
        std::vector<PanelProb> pch(nn), fch(nn);
        std::vector<TrsmNode> tpr(nn), tpo(nn);
        std::vector<Trsm2Prob> t2pr(nn), t2po(nn);
        std::vector<AsmProb> as(nn);
        std::vector<FrontProb> fr(nn);
        const int Kanc = pl->Ka - lv.a0;
        for (size_t s = 0; s < nn; ++s) {
            const int i = lv.nodes[s];
            const long r0 = pl->row0[i], nr = pl->row1[i] - r0;
            lv.max_rows = std::max(lv.max_rows, nr);
            const long rk = pl->knot_ptr[i + 1] - pl->knot_ptr[i];
            double* Lp = lv.Lp.p + s * (size_t)lv.cw * lv.cw;
            double* F = lv.F.p + s * (size_t)lv.nf * lv.ldf;
            GemmProb g{};
            g.A = pl->W.p + r0 * pl->ldw + lv.a0; g.lda = pl->ldw;
            g.B = pl->W.p + lv.a0; g.ldb = pl->ldw; g.idxB = pl->knot_idx.p + pl->knot_idx_off[i];
            g.C = pl->W.p + r0 * pl->ldw + lv.c0; g.ldc = pl->ldw;
            g.XA = pl->X.p + r0 * pl->d; g.XB = pl->X.p;
            g.M = (int)nr; g.N = lv.cw; g.K = Kanc; g.lower = 0;
            resid[s] = g;
            const double rkt = (double)rk, anct = (double)pl->anc_rank[i], nat = anct + 1.0;      // true rank, true ancestor columns, + y
            lv.fl_resid += Work(2.0 * nr * rkt * anct, 2.0 * nr * lv.cw * Kanc, 8.0 * nr * (Kanc + lv.cw + pl->d));
            kinv[s] = KinvProb{Lp, pl->knots_dev.p + pl->knot_ptr[i], (int)rk, lv.cw};
            pch[s] = PanelProb{Lp, lv.invP.p + s * (size_t)lv.cwt * 256, lv.cw, lv.cwt, lv.cwt, i};
            lv.fl_pchol += Work(rkt * rkt * rkt / 3.0, (double)lv.cw * lv.cw * lv.cw / 3.0, 8.0 * 2 * lv.cw * lv.cw);
            tpr[s] = TrsmNode{Lp, lv.invP.p + s * (size_t)lv.cwt * 256, lv.cw, lv.cwt};
            tpo[s] = TrsmNode{F, lv.invF.p + s * (size_t)lv.cwt * 256, lv.ldf, lv.cwt};
            lv.fl_trsm += Work((double)nr * rkt * rkt, (double)nr * lv.cw * lv.cw, 8.0 * 2 * nr * lv.cw);
            // (var -= |W^m[x]|^2 rides on the row solve: the prior variance of the level-by-level path, see k_init_yblock)
            t2pr[s] = Trsm2Prob{Lp, lv.invP.p + s * (size_t)lv.cwt * 256, pl->W.p + r0 * pl->ldw + lv.c0, pl->var.p + r0, lv.cw, pl->ldw, lv.cwt, (int)(nr / 16), 0, -1.0, nullptr, nullptr, 0, 0};
            t2po[s] = Trsm2Prob{F, lv.invF.p + s * (size_t)lv.cwt * 256, pl->W.p + r0 * pl->ldw + lv.c0, pl->var.p + r0, lv.ldf, pl->ldw, lv.cwt, (int)(nr / 16), 0, 1.0, nullptr, nullptr, 0, 0};
            lv.max_tiles = std::max(lv.max_tiles, nr / 16);
            fch[s] = PanelProb{F, lv.invF.p + s * (size_t)lv.cwt * 256, lv.ldf, lv.nf / 16, lv.cwt, i};
            lv.fl_fchol += Work(rkt * rkt * rkt / 3.0 + nat * rkt * rkt, (double)lv.cw * lv.cw * lv.cw / 3.0 + (double)lv.na * lv.cw * lv.cw,
                                8.0 * (2.5 * lv.nf * (lv.nf + 1) / 2));      // four children's Schur blocks in (lower halves; fewer for ragged trees), the front out
            GemmProb sc{};
            sc.A = F + (size_t)lv.cw * lv.ldf; sc.lda = lv.ldf; sc.B = sc.A; sc.ldb = lv.ldf;
            sc.C = lv.panel_only ? nullptr : F + (size_t)lv.cw * lv.nf + lv.cw; sc.ldc = lv.nf;      // panel_only: no Schur block is ever formed
            sc.M = lv.na; sc.N = lv.na; sc.K = lv.cw; sc.lower = 1;
            schur[s] = sc;
            lv.fl_schur += Work(nat * nat * rkt, (double)lv.na * lv.na * lv.cw, 0.0);
            GemmProb u{};
            u.A = pl->W.p + r0 * pl->ldw + lv.c0; u.lda = pl->ldw;
            u.B = F + (size_t)lv.cw * lv.ldf; u.ldb = lv.ldf;
            u.C = pl->W.p + r0 * pl->ldw + lv.a0; u.ldc = pl->ldw;
            u.M = (int)nr; u.N = lv.na; u.K = lv.cw; u.lower = 0;
            upd[s] = u;
            lv.fl_update += Work(2.0 * nr * nat * rkt, 2.0 * nr * lv.na * lv.cw, 8.0 * 2 * nr * lv.na);
            AsmProb a{};
            a.F = F; a.nf = lv.nf; a.cw = lv.cw; a.child0 = (int)kids.size();
            a.nchild = pl->child_ptr[i + 1] - pl->child_ptr[i]; a.add_identity = 1;
            for (int c = pl->child_ptr[i]; c < pl->child_ptr[i + 1]; ++c) {
                const int ch = pl->child_list[c];
                if (pl->node_level[ch] != m + 1) throw MraError(MRA_ERR_INVALID, "child must be one level below its parent");
                AsmChild k{};
                pl->kid_leaf.push_back(pl->leaf[ch] ? pl->leaf_slot[ch] : -1);
                if (pl->leaf[ch]) { k.G = pl->Gt.p ? pl->Gt.p + pl->leaf_goff[pl->leaf_slot[ch]] : nullptr; k.ld = pl->na[m + 1]; }
                else {
                    const LevelData& cl = pl->lev[m + 1];
                    // (children whose fronts are kept as panels have no Schur block: their parents are built by the signed SYRK)
                    k.G = cl.panel_only ? nullptr : cl.F.p + (size_t)pl->node_slot[ch] * cl.nf * cl.nf + (size_t)cl.cw * cl.nf + cl.cw;
                    k.ld = cl.nf;
                }
                if (pl->na[m + 1] != lv.nf) throw MraError(MRA_ERR_INVALID, "front size mismatch");
                kids.push_back(k);
            }
            as[s] = a;
            fr[s] = FrontProb{F, lv.invF.p + s * (size_t)lv.cwt * 256, lv.nf, lv.cwt, i, a.child0, a.nchild};
        }
        {
            const long nt = lv.nf / 16;
            const size_t full = (size_t)(nt * (nt + 1) / 2 + lv.cwt) * FT_SZ * sizeof(double);
            const size_t panel = (size_t)(lv.cwt * nt - lv.cwt * (lv.cwt - 1) / 2 + lv.cwt) * FT_SZ * sizeof(double);
            lv.front_mode = full <= 160 * 1024 ? 2 : (panel <= 160 * 1024 ? 1 : 0);
            if (lv.panel_only) lv.front_mode = 0;
            lv.front_lds = lv.front_mode == 2 ? full : panel;
            lv.gFront.upload(fr);
        }
        // one-launch prior of the level: blocks <= 64 wide (four column tiles in registers)
        lv.prior_level_ok = lv.cw <= 64 && lv.cw % 16 == 0;
        lv.fl_knot_resid = Work();
        if (lv.prior_level_ok) {
            std::vector<GemmProb> kn(nn), fz;
            const long blk = 512;                            // rows per workgroup: four passes of 4 waves x 2 row tiles
            for (size_t s = 0; s < nn; ++s) {
                const int i = lv.nodes[s];
                const long r0 = pl->row0[i], nr = pl->row1[i] - r0;
                double* Lp = lv.Lp.p + s * (size_t)lv.cw * lv.cw;
                const int* kix = pl->knot_idx.p + pl->knot_idx_off[i];
                GemmProb c{};
                c.A = pl->W.p + lv.a0; c.lda = pl->ldw; c.idxA = kix; c.B = c.A; c.ldb = pl->ldw; c.idxB = kix;
                c.C = Lp; c.ldc = lv.cw; c.XA = pl->X.p; c.XB = pl->X.p;
                c.M = lv.cw; c.N = lv.cw; c.K = Kanc; c.lower = 0; c.sym_diag = 1; c.diag_add = 0.0;
                kn[s] = c;
                const double rkt = (double)(pl->knot_ptr[i + 1] - pl->knot_ptr[i]), anct = (double)pl->anc_rank[i];
                lv.fl_knot_resid += Work(2.0 * rkt * rkt * anct, 2.0 * lv.cw * lv.cw * Kanc, 8.0 * (rkt * (Kanc + pl->d) + (double)lv.cw * lv.cw));
                for (long b0 = 0; b0 < nr; b0 += blk) {
                    GemmProb g = resid[s];
                    g.A += b0 * pl->ldw; g.C += b0 * pl->ldw; g.XA += b0 * pl->d;
                    g.M = (int)std::min(blk, nr - b0);
                    g.solveL = Lp; g.solveI = lv.invP.p + s * (size_t)lv.cwt * 256; g.var = pl->var.p + r0 + b0;
                    fz.push_back(g);
                }
            }
            lv.gKnotResid.upload(kn); lv.gResidFused.upload(fz);
        }
        lv.hResid = resid;
        lv.gResid.upload(resid); lv.gSchur.upload(schur); lv.gUpdate.upload(upd); lv.gKinv.upload(kinv);
        lv.gPriorChol.upload(pch); lv.gFrontChol.upload(fch); lv.gTrsmPrior.upload(tpr); lv.gTrsmPost.upload(tpo);
        lv.gAsm.upload(as);
        lv.gTrsm2Prior.upload(t2pr); lv.gTrsm2Post.upload(t2po);
        // (tile_node / tile_row0 - one entry per row tile of the level - serve the row solve of blocks wider than 192 only: ensure_tile_lists)
    }
    pl->asmKids.upload(kids);
    pl->hKids = kids;
    tr.mark("per-level descriptors");

    // ---- fused path eligibility: uniform block width on all non-leaf levels, leaves only on the last level
    pl->regular = false;
    const int NL = L - 1;
    if (NL >= 1 && NL <= 8) {
        bool ok = true;
        for (int m = 0; m < NL && ok; ++m) {
            if (pl->cw[m] != pl->cw[0]) ok = false;
            for (long i = pl->level_ptr[m]; i < pl->level_ptr[m + 1] && ok; ++i) if (pl->leaf[i]) ok = false;
        }
        for (long i = pl->level_ptr[NL]; i < pl->level_ptr[NL + 1] && ok; ++i) if (!pl->leaf[i]) ok = false;
        const int cwt = pl->cw[0] / 16;
        if (cwt != 1 && cwt != 2 && cwt != 4) ok = false;
        if (cwt * NL > 16) ok = false;                       // register budget of the cascade kernels
        if (ok) { pl->regular = true; pl->NL = NL; pl->CWT = cwt; }
    }
    pl->regular_hi = false;
    if (!pl->regular && pl->shape_regular && NL >= 5 && NL <= 8) {
        bool ok = true;
        for (int m = 0; m < NL && ok; ++m) if (pl->cw[m] != 64) ok = false;
        if (ok) {
            pl->regular_hi = true; pl->NL = NL; pl->CWT = 4;
            // row tiles of the leaves with their ancestor chains; one workgroup per leaf (<= 4 tiles) for k_predict_hi, groups of eight
            // consecutive tiles below one level-(NL-5) node for the coarse cascade
            std::vector<long> r0s, wg0, wg0_8;
            std::vector<int> chains, wgn, wgn_8, wgl;
            const int nlo = NL - 4;
            int prev_coarse = -1;
            for (size_t t = 0; t < pl->leaf_nodes.size(); ++t) {
                const int i = pl->leaf_nodes[t];
                int ch[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                for (int a = pl->parent[i]; a >= 0; a = pl->parent[a]) ch[pl->node_level[a]] = pl->node_slot[a];
                const int coarse = ch[nlo - 1];                          // slot of the deepest coarse level: same slot = same coarse chain
                for (long p = pl->row0[i]; p < pl->row1[i]; p += 16) {
                    const long k = (p - pl->row0[i]) / 16;
                    if (k % 4 == 0) { wg0.push_back((long)r0s.size()); wgn.push_back((int)std::min<long>(4, (pl->row1[i] - p) / 16)); wgl.push_back((int)t); }
                    if (wgn_8.empty() || wgn_8.back() == 8 || coarse != prev_coarse) { wg0_8.push_back((long)r0s.size()); wgn_8.push_back(0); }
                    ++wgn_8.back();
                    prev_coarse = coarse;
                    r0s.push_back(p);
                    for (int k2 = 0; k2 < 8; ++k2) chains.push_back(ch[k2]);
                }
            }
            pl->ft_row0.upload(r0s); pl->ft_chain.upload(chains);
            pl->n_ftiles = (long)r0s.size();
            pl->ft_wg0.upload(wg0); pl->ft_wgn.upload(wgn); pl->n_fwg = (long)wg0.size();
            pl->hi_wgleaf.upload(wgl);
            pl->hi_wg0_8.upload(wg0_8); pl->hi_wgn_8.upload(wgn_8); pl->n_hi_wg8 = (long)wg0_8.size();
        }
    }
    if (pl->regular) {
        const int cw = pl->cw[0];
        pl->fl.clear();
        pl->fl.resize(pl->NL);
        auto chain_of = [&](int node, int* out) {            // slots of the ancestors (and the node itself)
            for (int k = 0; k < 8; ++k) out[k] = 0;
            int i = node;
            while (i >= 0) {
                if (!pl->leaf[i]) out[pl->node_level[i]] = pl->node_slot[i];
                i = pl->parent[i];
            }
        };
        for (int m = 0; m < pl->NL; ++m) {
            mra_plan::FusedLevel& f = pl->fl[m];
            const LevelData& lv = pl->lev[m];
            const size_t nn = lv.nodes.size();
            f.kx.alloc(nn * (size_t)cw * pl->d);
            f.Wk.alloc(std::max<size_t>(nn * (size_t)cw * (m * cw), 1));
            std::vector<int> kv(nn * (size_t)cw), rows, chain, knot0, wgn;
            std::vector<long> wg0;
            for (size_t sl = 0; sl < nn; ++sl) {
                const int i = lv.nodes[sl];
                const long rk = pl->knot_ptr[i + 1] - pl->knot_ptr[i];
                for (int c = 0; c < cw; ++c) kv[sl * cw + c] = c < rk ? 1 : 0;
                int ch[8];
                chain_of(i, ch);
                // one workgroup per family of siblings (same ancestor chain: the staged operands are shared),
                // as long as it stays within 8 row tiles (one per wave on the level-by-level staging path)
                // (only on levels with many nodes, where workgroups run in several rounds per CU: on small levels
                // - the levels of a sharded rank - the longer per-workgroup chain costs more than it saves)
                const bool same_family = nn >= 512 && sl > 0 && pl->parent[i] >= 0 && pl->parent[i] == pl->parent[lv.nodes[sl - 1]] &&
                                         !wgn.empty() && wgn.back() + cw / 16 <= 8;
                if (same_family) wgn.back() += cw / 16;
                else { wg0.push_back((long)knot0.size()); wgn.push_back(cw / 16); }
                for (int tt = 0; tt < cw / 16; ++tt) {
                    for (int r = 0; r < 16; ++r) rows.push_back(-1);          // contents: fill_knot_arrays
                    for (int k = 0; k < 8; ++k) chain.push_back(ch[k]);
                    knot0.push_back(tt * 16);
                }
            }
            f.kvalid.upload(kv); f.kt_rows.upload(rows); f.kt_chain.upload(chain); f.kt_knot0.upload(knot0);
            f.n_ktiles = (long)knot0.size();
            f.kt_wg0.upload(wg0); f.kt_wgn.upload(wgn); f.n_kwg = (long)wg0.size();
            f.k_threads = 256;
            for (int v : wgn) if (v > 4) f.k_threads = 512;
            {
                // kInv of every node of the level: kernel(knots, knots) - Wk Wk^T as one batched COV product
                std::vector<GemmProb> gk(nn);
                for (size_t sl = 0; sl < nn; ++sl) {
                    GemmProb g{};
                    g.A = f.Wk.p + sl * (size_t)cw * (m * cw); g.lda = m * cw; g.B = g.A; g.ldb = m * cw;
                    g.C = pl->lev[m].Lp.p + sl * (size_t)cw * cw; g.ldc = cw;
                    g.XA = f.kx.p + sl * (size_t)cw * pl->d; g.XB = g.XA;
                    g.M = cw; g.N = cw; g.K = m * cw; g.lower = 0;
                    gk[sl] = g;
                }
                f.gKinv.upload(gk);
            }
        }
        {
            // k_knot_chain: one workgroup per node of the last non-leaf level; owner of an upper node = first workgroup below it
            // (it recomputes every ancestor in every workgroup, so it only pays while one round of workgroups covers the
            // level: the chain covers levels 0 .. kc_levels-1, the deepest level with at most one workgroup per CU; deeper
            // levels - many nodes, throughput-bound - keep their per-level launches)
            int ncu = 256;
            {
                hipDeviceProp_t prop;
                if (!g_dry && hipGetDeviceProperties(&prop, pl->device) == hipSuccess && prop.multiProcessorCount > 0) ncu = prop.multiProcessorCount;
            }
            pl->n_cu = ncu;
            int nlv = 1;
            while (nlv < pl->NL && (long)pl->lev[nlv].nodes.size() <= ncu) ++nlv;
            pl->kc_levels = nlv;
            const int cwt = pl->CWT;
            const LevelData& lb = pl->lev[nlv - 1];
            std::vector<int> kch(lb.nodes.size() * 8, 0);
            std::vector<std::vector<int>> own(nlv);
            for (int m = 0; m < nlv; ++m) own[m].assign(pl->lev[m].nodes.size(), -1);
            for (size_t b = 0; b < lb.nodes.size(); ++b) {
                int ch[8];
                chain_of(lb.nodes[b], ch);
                for (int k = 0; k < 8; ++k) kch[b * 8 + k] = ch[k];
                for (int m = 0; m < nlv; ++m) if (own[m][ch[m]] < 0) own[m][ch[m]] = (int)b;
            }
            pl->kc_chain.upload(kch);
            pl->kc_chain_host = kch;
            std::vector<int> mask(lb.nodes.size(), 0);
            for (size_t b = 0; b < lb.nodes.size(); ++b)
                for (int m = 0; m < nlv; ++m) if (own[m][kch[b * 8 + m]] == (int)b) mask[b] |= 1 << m;
            pl->kc_ownmask.upload(mask);
            pl->kc_knots.alloc(lb.nodes.size() * (size_t)nlv * cwt * 16 * (pl->d + 1));        // filled by mra_plan_set_locs
            const long off = (long)cwt * cwt * ((nlv - 1) * (nlv - 2) / 2) + (long)(nlv - 1) * (cwt * (cwt - 1) / 2 + cwt);
            pl->knot_chain_lds = (size_t)off * 2048 + (size_t)(cwt * (cwt + 1) / 2 + cwt) * FT_SZ * sizeof(double)
                                 + (size_t)nlv * cwt * 16 * (pl->d + 1) * sizeof(double);    // + the chain's knots (coordinates, real/phantom flags)
            pl->knot_chain_ok = pl->knot_chain_lds <= 160 * 1024;
        }
        tr.mark("fused levels, knot chain");
        std::vector<long> r0s, fwg0, lwg0;
        std::vector<int> chains, fwgn, tleaf, lwgn;
        {
            int ndev_cu = 256;
            hipDeviceProp_t prop;
            if (!g_dry && hipGetDeviceProperties(&prop, pl->device) == hipSuccess && prop.multiProcessorCount > 0) ndev_cu = prop.multiProcessorCount;
            size_t nparents = 0;
            long ntiles_all = 0;
            for (size_t t = 0; t < pl->leaf_nodes.size(); ++t) {
                if (t == 0 || pl->parent[pl->leaf_nodes[t]] != pl->parent[pl->leaf_nodes[t - 1]]) ++nparents;
                ntiles_all += (pl->row1[pl->leaf_nodes[t]] - pl->row0[pl->leaf_nodes[t]]) / 16;
            }
            // one workgroup (8 waves, the whole LDS) per leaf or per family of sibling leaves: rounds of workgroups over the CUs times
            // [staging the operand image (~5 us, exposed: one workgroup per CU) + rounds of 8 row tiles (~32 us each)].  Measured on
            // shards of C3: 256 families on 256 CUs = one round of 64 tiles (261 us) against four rounds of 16 (316 us); 128 families
            // leave half the CUs idle (261 against 158 us).
            const size_t nl_ = std::max<size_t>(1, pl->leaf_nodes.size()), np_ = std::max<size_t>(1, nparents);
            const double t_leaf = std::ceil((double)ntiles_all / nl_ / 8.0), t_fam = std::ceil((double)ntiles_all / np_ / 8.0);
            const double cost_leaf = std::ceil((double)nl_ / ndev_cu) * (5.0 + 32.0 * t_leaf), cost_fam = std::ceil((double)np_ / ndev_cu) * (5.0 + 32.0 * t_fam);
            pl->cascade_group_siblings = cost_fam < cost_leaf;
        }
        for (size_t t = 0; t < pl->leaf_nodes.size(); ++t) {
            const int i = pl->leaf_nodes[t];
            int ch[8];
            chain_of(i, ch);
            // one workgroup per leaf ... or per family of sibling leaves (same parent = same operand image on every level,
            // staged once for all of them) when that still leaves at least two workgroups per CU
            const bool join = pl->cascade_group_siblings && t > 0 && pl->parent[i] >= 0 && pl->parent[i] == pl->parent[pl->leaf_nodes[t - 1]] && !lwgn.empty();
            if (join) lwgn.back() += (int)((pl->row1[i] - pl->row0[i]) / 16);
            else { lwg0.push_back((long)r0s.size()); lwgn.push_back((int)((pl->row1[i] - pl->row0[i]) / 16)); }
            for (long p = pl->row0[i]; p < pl->row1[i]; p += 16) {
                if (((p - pl->row0[i]) / 16) % pl->cascade_wpw == 0) {
                    fwg0.push_back((long)r0s.size());
                    fwgn.push_back((int)std::min<long>(pl->cascade_wpw, (pl->row1[i] - p) / 16));
                }
                r0s.push_back(p);
                tleaf.push_back((int)t);
                for (int k = 0; k < 8; ++k) chains.push_back(ch[k]);
            }
        }
        pl->ft_row0.upload(r0s); pl->ft_chain.upload(chains); pl->ft_leaf.upload(tleaf);
        pl->n_ftiles = (long)r0s.size();
        pl->ft_wg0.upload(fwg0); pl->ft_wgn.upload(fwgn); pl->n_fwg = (long)fwg0.size();
        {
            // the same workgroups ordered for the predictive cascade: workgroup b runs on XCD b % 8, and the workgroups of
            // one leaf all stream that leaf's Ut block, so they are dealt to one XCD (one L2) -- empty slots pad the lanes
            std::vector<std::vector<size_t>> lane(8);
            for (size_t g = 0; g < fwg0.size(); ++g) lane[(size_t)tleaf[(size_t)fwg0[g]] % 8].push_back(g);
            size_t deep = 0;
            for (const auto& l : lane) deep = std::max(deep, l.size());
            std::vector<long> x0(deep * 8, 0);
            std::vector<int> xn(deep * 8, 0), xl(deep * 8, 0);
            for (size_t k = 0; k < deep; ++k)
                for (size_t x = 0; x < 8; ++x)
                    if (k < lane[x].size()) { x0[k * 8 + x] = fwg0[lane[x][k]]; xn[k * 8 + x] = fwgn[lane[x][k]]; xl[k * 8 + x] = tleaf[(size_t)fwg0[lane[x][k]]]; }
            pl->ft_wg0_x.upload(x0); pl->ft_wgn_x.upload(xn); pl->ft_wgleaf_x.upload(xl); pl->n_fwg_x = (long)x0.size();
        }
        {
            const int cwt = pl->CWT, mmax = pl->NL - 1;
            pl->cascade_lds = (size_t)(cwt * mmax * cwt + cwt * (cwt - 1) / 2 + cwt) * 2048;
            const int nl = pl->NL;
            pl->cascade_lds_all = (size_t)(cwt * cwt * (nl * (nl - 1) / 2) + nl * (cwt * (cwt - 1) / 2 + cwt)) * 2048;
            pl->cascade_stage_all = pl->cascade_lds_all <= 160 * 1024;
            pl->ft_wg0_leaf.upload(lwg0); pl->ft_wgn_leaf.upload(lwgn); pl->n_fwg_leaf = (long)lwg0.size();
        }
        tr.mark("row tiles of the cascades");
    }
    arena.finish();
    tr.mark("descriptor arena to the device");
    if (!pl->knots_pending) { fill_knot_arrays(pl); tr.mark("knot arrays"); }
}

// Everything in the plan that holds knot ROWS (as opposed to knot counts): the device copy of knot_rows, the padded per-node
// index lists of the prior products, the knot row tiles of the fused levels.  The arrays are sized by build_static; a plan
// built beside the knot draws (mra_plan_create_replay_2d) gets their contents here, once the draws are done.
static void fill_knot_arrays(mra_plan* pl) {
    if ((long)pl->knot_rows.size() != pl->knot_ptr.back()) throw MraError(MRA_ERR_INVALID, "knot_rows does not match knot_ptr");
    // (the device only ever looks at the knot rows of NON-leaf nodes - the kInv gather of the level-by-level prior; the leaves'
    //  entries, nine tenths of the 8 MB at 1024^2, stay on the host)
    {
        long kmax = 0;
        for (int i = 0; i < pl->n_nodes; ++i) if (!pl->leaf[i]) kmax = std::max(kmax, pl->knot_ptr[i + 1]);
        if (kmax > 0) HIP_TRY(mraMemcpy(pl->knots_dev.p, pl->knot_rows.data(), (size_t)kmax * sizeof(long), hipMemcpyHostToDevice));
    }
    std::vector<int> kidx(pl->knot_idx.n, -1);
    for (int i = 0; i < pl->n_nodes; ++i) {
        if (pl->leaf[i]) continue;
        const int m = pl->node_level[i];
        const long rk = pl->knot_ptr[i + 1] - pl->knot_ptr[i];
        for (int c = 0; c < pl->cw[m] && c < rk; ++c) kidx[pl->knot_idx_off[i] + c] = (int)pl->knot_rows[pl->knot_ptr[i] + c];
    }
    pl->knot_idx.fill(kidx);
    if (pl->regular) {
        const int cw = pl->cw[0];
        for (int m = 0; m < pl->NL; ++m) {
            const LevelData& lv = pl->lev[m];
            std::vector<int> rows(lv.nodes.size() * (size_t)cw, -1);
            for (size_t sl = 0; sl < lv.nodes.size(); ++sl) {
                const int i = lv.nodes[sl];
                const long rk = pl->knot_ptr[i + 1] - pl->knot_ptr[i];
                for (int c = 0; c < cw && c < rk; ++c) rows[sl * cw + c] = (int)pl->knot_rows[pl->knot_ptr[i] + c];
            }
            pl->fl[m].kt_rows.fill(rows);
        }
    }
}

// leaf descriptors: depend on which rows are observed
static void build_leaf(mra_plan* pl, const double* y) {
    PlanTrace tr("build_leaf");
    ArenaScope arena(&pl->arena);
    const size_t nl = pl->leaf_nodes.size();
    pl->cphantom_valid = false;
    pl->leaf_nop.assign(nl, 0);
    pl->leaf_poff.assign(nl + 1, 0);
    pl->leaf_ioff.assign(nl + 1, 0);
    // observed rows per leaf: counted and listed by a few threads over runs of leaves (one pass over y each)
    std::vector<int> obs, opos(pl->P), nobs(nl, 0);
    std::vector<long> obs_off(nl + 1, 0);
    const int64_t leaves_per_thread = std::max<int64_t>(1, (int64_t)(65536 * nl / std::max<long>(pl->P, 1)));
    parallel_rows((int64_t)nl, [&](int64_t a, int64_t b) {
        for (int64_t t = a; t < b; ++t) {
            const int i = pl->leaf_nodes[t];
            int no = 0;
            for (long p = pl->row0[i]; p < pl->row1[i]; ++p) no += std::isfinite(y[p]) ? 1 : 0;
            nobs[t] = no;
        }
    }, leaves_per_thread);
    pl->leaf_max_rows = 0; pl->leaf_max_nop = 0; pl->leaf_max_na = 0; pl->leaf_max_ht = 0;
    for (size_t t = 0; t < nl; ++t) {
        const int i = pl->leaf_nodes[t];
        const int na = pl->na[pl->node_level[i]];
        const int no = nobs[t], nop = (no + 15) / 16 * 16;
        obs_off[t + 1] = obs_off[t] + nop;
        pl->leaf_nop[t] = nop;
        const long nr = pl->row1[i] - pl->row0[i];
        pl->leaf_poff[t + 1] = pl->leaf_poff[t] + (long)(nop + na + nr) * nop;
        pl->leaf_ioff[t + 1] = pl->leaf_ioff[t] + (long)(nop / 16) * 256;
        pl->leaf_max_rows = std::max(pl->leaf_max_rows, nr);
        pl->leaf_max_nop = std::max(pl->leaf_max_nop, nop);
        pl->leaf_max_na = std::max(pl->leaf_max_na, na);
        pl->leaf_max_ht = std::max(pl->leaf_max_ht, (int)((nop + na + nr) / 16));
    }
    obs.resize((size_t)obs_off[nl]);
    {
        // rows outside every leaf (orphan knot rows of a shard) are nobody's observation
        long covered = 0;
        for (size_t t = 0; t < nl; ++t) covered += pl->row1[pl->leaf_nodes[t]] - pl->row0[pl->leaf_nodes[t]];
        if (covered != pl->P) std::fill(opos.begin(), opos.end(), -1);
    }
    parallel_rows((int64_t)nl, [&](int64_t a, int64_t b) {
        for (int64_t t = a; t < b; ++t) {
            const int i = pl->leaf_nodes[t];
            int* o = obs.data() + obs_off[t];
            int no = 0;
            for (long p = pl->row0[i]; p < pl->row1[i]; ++p) {
                const bool f = std::isfinite(y[p]);
                opos[p] = f ? no : -1;
                if (f) o[no++] = (int)p;
            }
            for (int k = no; k < pl->leaf_nop[t]; ++k) o[k] = -1;
        }
    }, leaves_per_thread);
    pl->obs_idx.upload(obs); pl->obs_pos.upload(opos); pl->leaf_nobs.upload(nobs);
    pl->obs_off_host = obs_off; pl->lik_tiles_valid = false; pl->lik_general_valid = false;
    pl->y_finite_host.assign(pl->P, 0);
    for (long p2 = 0; p2 < pl->P; ++p2) pl->y_finite_host[p2] = std::isfinite(y[p2]) ? 1 : 0;
    tr.mark("observation lists");
    pl->panel.alloc(std::max<long>(pl->leaf_poff.back(), 1));
    pl->leafInv.alloc(std::max<long>(pl->leaf_ioff.back(), 1));
    tr.mark("panel allocation");
    std::vector<LeafProb> lp(nl);
    std::vector<GemmProb> gr(nl), gs(nl), gu(nl), grl(nl);
    std::vector<PanelProb> pf(nl), pk(nl), pc(nl);
    std::vector<Trsm2Prob> tf(nl), tk(nl);
    pl->leaf_max_tiles_full = pl->leaf_max_tiles_lik = 0;
    pl->fl_leaf_resid = pl->fl_leaf_chol = pl->fl_leaf_chol_lik = pl->fl_leaf_syrk = pl->fl_leaf_update = pl->fl_leaf_c_only = Work();
    pl->by_leaf_ut = pl->by_leaf_tt = pl->by_leaf_c = 0;
    for (size_t t = 0; t < nl; ++t) {
        const int i = pl->leaf_nodes[t];
        const int m = pl->node_level[i];
        const int na = pl->na[m], a0 = pl->asuf[m], nop = pl->leaf_nop[t];
        const long r0 = pl->row0[i], nr = pl->row1[i] - r0;
        const int Kanc = pl->Ka - a0;
        double* Pn = pl->panel.p + pl->leaf_poff[t];
        LeafProb q{};
        q.Pn = Pn; q.obs = pl->obs_idx.p + obs_off[t]; q.row0 = r0; q.ld = nop; q.nrows = (int)nr;
        q.nop = nop; q.na = na; q.a0 = a0; q.node = i;
        lp[t] = q;
        GemmProb g{};
        g.A = pl->W.p + r0 * pl->ldw + a0; g.lda = pl->ldw;
        g.B = pl->W.p + a0; g.ldb = pl->ldw; g.idxB = q.obs;
        g.C = Pn + (size_t)(nop + na) * nop; g.ldc = nop;
        g.XA = pl->X.p + r0 * pl->d; g.XB = pl->X.p;
        g.M = (int)nr; g.N = nop; g.K = Kanc; g.lower = 0;
        g.rowmap = pl->obs_pos.p + r0; g.C2 = Pn; g.diag_add = pl->R;
        gr[t] = g;
        {
            // likelihood-only runs need just C = v_m(o,o) + R I: both sides gathered through the observation list
            GemmProb c{};
            c.A = pl->W.p + a0; c.lda = pl->ldw; c.idxA = q.obs; c.B = c.A; c.ldb = pl->ldw; c.idxB = q.obs;
            c.C = Pn; c.ldc = nop; c.XA = pl->X.p; c.XB = pl->X.p;
            c.M = nop; c.N = nop; c.K = Kanc; c.lower = 1; c.sym_diag = 1; c.diag_add = pl->R;
            grl[t] = c;
        }
        const double no = (double)nobs[t], anct = (double)pl->anc_rank[i], nat = anct + 1.0;      // true sizes: observations, ancestor columns, + y
        pl->by_leaf_ut += 8.0 * na * nop; pl->by_leaf_tt += 8.0 * nr * nop; pl->by_leaf_c += 8.0 * nop * nop;
        pl->fl_leaf_resid += Work(2.0 * nr * no * anct, 2.0 * nr * nop * Kanc, 8.0 * (nr * (Kanc + pl->d) + (double)nr * nop + (double)nop * nop));
        pl->fl_leaf_c_only += Work(no * no * anct, (double)nop * nop * Kanc, 8.0 * (no * (Kanc + pl->d) + (double)nop * nop));
        double* inv = pl->leafInv.p + pl->leaf_ioff[t];
        pf[t] = PanelProb{Pn, inv, nop, (int)((nop + na + nr) / 16), nop / 16, i};
        pk[t] = PanelProb{Pn, inv, nop, (nop + na) / 16, nop / 16, i};
        pc[t] = PanelProb{Pn, inv, nop, nop / 16, nop / 16, i};
        tf[t] = Trsm2Prob{Pn, inv, Pn + (size_t)nop * nop, pl->var.p + r0, nop, nop, nop / 16, (int)((na + nr) / 16), na / 16, -1.0, q.obs, pl->W.p + a0, pl->ldw, na / 16};
        tk[t] = Trsm2Prob{Pn, inv, Pn + (size_t)nop * nop, nullptr, nop, nop, nop / 16, na / 16, 0, 1.0, q.obs, pl->W.p + a0, pl->ldw, na / 16};
        pl->leaf_max_tiles_full = std::max(pl->leaf_max_tiles_full, (int)((na + nr) / 16));
        pl->leaf_max_tiles_lik = std::max(pl->leaf_max_tiles_lik, na / 16);
        pl->fl_leaf_chol += Work(no * no * no / 3.0 + (nat + nr) * no * no, (double)nop * nop * nop / 3.0 + (double)(na + nr) * nop * nop,
                                 8.0 * (1.5 * nop * nop + 2.0 * (na + nr) * nop + 2.0 * nr));
        pl->fl_leaf_chol_lik += Work(no * no * no / 3.0 + nat * no * no, (double)nop * nop * nop / 3.0 + (double)na * nop * nop,
                                     8.0 * (1.5 * nop * nop + 2.0 * na * nop));
        GemmProb s{};
        s.A = Pn + (size_t)nop * nop; s.lda = nop; s.B = s.A; s.ldb = nop;
        s.C = pl->Gt.p ? pl->Gt.p + pl->leaf_goff[t] : nullptr; s.ldc = na; s.M = na; s.N = na; s.K = nop; s.lower = 1;
        gs[t] = s;
        pl->fl_leaf_syrk += Work(nat * nat * no, (double)na * na * nop, 8.0 * na * nop);
        GemmProb u{};
        u.A = Pn + (size_t)(nop + na) * nop; u.lda = nop; u.B = Pn + (size_t)nop * nop; u.ldb = nop;
        u.C = pl->W.p + r0 * pl->ldw + a0; u.ldc = pl->ldw; u.M = (int)nr; u.N = na; u.K = nop; u.lower = 0;
        u.zc = na - MRA_YB;                  // y block: C_in = 0 (the column still holds y itself)
        gu[t] = u;
        pl->fl_leaf_update += Work(2.0 * nr * nat * no, 2.0 * nr * na * nop, 8.0 * ((double)nr * nop + (double)na * nop + 2.0 * nr * na));
    }
    tr.mark("leaf descriptors (host)");
    {
        std::vector<double*> uts(nl);
        for (size_t t = 0; t < nl; ++t) uts[t] = pl->panel.p + pl->leaf_poff[t] + (size_t)pl->leaf_nop[t] * pl->leaf_nop[t];
        pl->leaf_ut.upload(uts);
        pl->leaf_nop_dev.upload(pl->leaf_nop);
        // the Ut blocks start from zero: rows of the y block beyond y itself and phantom observation columns stay zero
        if (pl->panel.n) HIP_TRY(mraMemset(pl->panel.p, 0, pl->panel.n * sizeof(double)));
    }
    tr.mark("panel memset");
    pl->parent_syrk = false;
    pl->hLeafSyrk = gs;
    if (pl->shape_regular && pl->NL >= 1) {
        const LevelData& lv = pl->lev[pl->NL - 1];
        std::vector<GemmProb> ps(lv.nodes.size());
        std::vector<GemmSeg> segs;
        std::vector<std::pair<size_t, int>> where(lv.nodes.size());
        for (size_t sidx = 0; sidx < lv.nodes.size(); ++sidx) {
            const int i = lv.nodes[sidx];
            where[sidx] = {segs.size(), pl->child_ptr[i + 1] - pl->child_ptr[i]};
            for (int c = pl->child_ptr[i]; c < pl->child_ptr[i + 1]; ++c) {
                const int lt = pl->leaf_slot[pl->child_list[c]];
                const int nop = pl->leaf_nop[lt];
                double* ut = pl->panel.p + pl->leaf_poff[lt] + (size_t)nop * nop;
                segs.push_back(GemmSeg{ut, ut, nop, nop, nop, 0});
            }
        }
        pl->parentSegs.upload(segs);
        for (size_t sidx = 0; sidx < lv.nodes.size(); ++sidx) {
            GemmProb g{};
            g.C = lv.panel_only ? nullptr : lv.F.p + sidx * (size_t)lv.nf * lv.nf; g.ldc = lv.nf; g.M = lv.nf; g.N = lv.nf; g.lower = 1;      // (not launched when panel_only)
            g.segs = pl->parentSegs.p + where[sidx].first; g.nseg = where[sidx].second; g.diag_one = lv.cw;
            if (g.nseg == 0) { g.nseg = 0; g.K = 0; g.A = g.B = pl->W.p; }
            ps[sidx] = g;
        }
        pl->gParentSyrk.upload(ps);
        pl->parent_syrk = true;
        pl->fl_parent_panel = pl->fl_grand_syrk = Work();
        if (pl->lowrank_parent && pl->NL >= 2) {
            // panels of the leaves' parents, in three launches: the own block F_oo = I + U_o U_o^T with its Cholesky (k_parent_front on
            // a front of cw rows: everything in registers / LDS), the rows below F_ao = U_a U_o^T (one GEMM over the children's Ut
            // segments, A = their ancestor rows, B = their own-block rows), and the row solve Zt = F_ao Lt^-T in place
            std::vector<GemmSeg> segs_ao;
            for (const GemmSeg& sg : segs) segs_ao.push_back(GemmSeg{sg.A + (size_t)lv.cw * sg.lda, sg.B, sg.lda, sg.ldb, sg.K, 0});
            pl->parentSegsAo.upload(segs_ao);
            std::vector<GemmProb> pp(lv.nodes.size());
            std::vector<FrontProb> pown(lv.nodes.size());
            std::vector<Trsm2Prob> pzt(lv.nodes.size());
            for (size_t sidx = 0; sidx < lv.nodes.size(); ++sidx) {
                double* panel_s = lv.F.p + sidx * (size_t)lv.nf * lv.ldf;
                double* inv_s = lv.invF.p + sidx * (size_t)lv.cwt * 256;
                GemmProb g{};
                g.C = panel_s + (size_t)lv.cw * lv.ldf; g.ldc = lv.ldf; g.M = lv.na; g.N = lv.cw; g.lower = 0;
                g.segs = pl->parentSegsAo.p + where[sidx].first; g.nseg = where[sidx].second; g.diag_one = 0;
                if (g.nseg == 0) { g.K = 0; g.A = g.B = pl->W.p; }
                pp[sidx] = g;
                pown[sidx] = FrontProb{panel_s, inv_s, lv.cw, lv.cwt, lv.nodes[sidx], (int)where[sidx].first, where[sidx].second};
                pzt[sidx] = Trsm2Prob{panel_s, inv_s, panel_s + (size_t)lv.cw * lv.ldf, nullptr, lv.ldf, lv.ldf, lv.cwt, lv.na / 16, 0, 1.0, nullptr, nullptr, 0, 0};
                const int i = lv.nodes[sidx];
                const double rkt = (double)(pl->knot_ptr[i + 1] - pl->knot_ptr[i]), nat = (double)pl->anc_rank[i] + 1.0;
                double kobs = 0, kpad = 0;
                for (int c = pl->child_ptr[i]; c < pl->child_ptr[i + 1]; ++c) { const int lt = pl->leaf_slot[pl->child_list[c]]; kobs += nobs[lt]; kpad += pl->leaf_nop[lt]; }
                pl->fl_parent_panel += Work(2.0 * (rkt + nat) * rkt * kobs, 2.0 * lv.nf * lv.cw * kpad, 8.0 * (lv.nf * kpad + (double)lv.nf * lv.cw));
            }
            pl->gParentPanel.upload(pp);
            // the LDS-tiled segmented product keeps its K steps as a table of SB_MAXST2 entries
            pl->parent_panel_lds_ok = true;
            for (size_t sidx = 0; sidx < lv.nodes.size() && pl->parent_panel_lds_ok; ++sidx) {
                long steps = 0;
                for (int k = 0; k < where[sidx].second; ++k) steps += segs_ao[where[sidx].first + k].K / 16;
                if (where[sidx].second > SB_MAXST2 || steps > SB_MAXST2) pl->parent_panel_lds_ok = false;
            }
            pl->gParentOwn.upload(pown);
            pl->gParentZt.upload(pzt);
            {
                const long nt = lv.cwt, npanel = lv.cwt * nt - lv.cwt * (lv.cwt - 1) / 2;
                pl->parent_own_lds = (size_t)(2 * lv.cw * PF_LD + (npanel + lv.cwt) * FT_SZ) * sizeof(double);
            }
            // fronts of the grandparents: I + sum over grandchild leaves Ut[anc] Ut[anc]^T - sum over children Zt Zt^T
            const LevelData& lg = pl->lev[pl->NL - 2];
            std::vector<GemmSeg> gsegs;
            std::vector<std::pair<size_t, int>> gwhere(lg.nodes.size());
            for (size_t sidx = 0; sidx < lg.nodes.size(); ++sidx) {
                const int i = lg.nodes[sidx];
                const size_t first = gsegs.size();
                const double rkt = (double)(pl->knot_ptr[i + 1] - pl->knot_ptr[i]), nft = rkt + (double)pl->anc_rank[i] + 1.0;
                double kalg = 0, kpad = 0;
                for (int c = pl->child_ptr[i]; c < pl->child_ptr[i + 1]; ++c) {
                    const int ch = pl->child_list[c];                  // a parent of leaves
                    for (int c2 = pl->child_ptr[ch]; c2 < pl->child_ptr[ch + 1]; ++c2) {
                        const int lt = pl->leaf_slot[pl->child_list[c2]];
                        const int nop = pl->leaf_nop[lt];
                        if (!nop) continue;
                        double* ut = pl->panel.p + pl->leaf_poff[lt] + (size_t)nop * nop + (size_t)lv.cw * nop;      // rows of the ancestors of `ch`
                        gsegs.push_back(GemmSeg{ut, ut, nop, nop, nop, 0});
                        kalg += nobs[lt]; kpad += nop;
                    }
                    double* zt = lv.F.p + (size_t)pl->node_slot[ch] * lv.nf * lv.ldf + (size_t)lv.cw * lv.ldf;
                    gsegs.push_back(GemmSeg{zt, zt, lv.ldf, lv.ldf, lv.cw, 1});
                    kalg += (double)(pl->knot_ptr[ch + 1] - pl->knot_ptr[ch]); kpad += lv.cw;
                }
                gwhere[sidx] = {first, (int)(gsegs.size() - first)};
                pl->fl_grand_syrk += Work(nft * nft * kalg, (double)lg.nf * lg.nf * kpad, 8.0 * (lg.nf * kpad + 0.5 * lg.nf * (lg.nf + 1)));
            }
            pl->grandSegs.upload(gsegs);
            std::vector<GemmProb> gp(lg.nodes.size());
            for (size_t sidx = 0; sidx < lg.nodes.size(); ++sidx) {
                GemmProb g{};
                g.C = lg.F.p + sidx * (size_t)lg.nf * lg.nf; g.ldc = lg.nf; g.M = lg.nf; g.N = lg.nf; g.lower = 1;
                g.segs = pl->grandSegs.p + gwhere[sidx].first; g.nseg = gwhere[sidx].second; g.diag_one = lg.cw;
                if (g.nseg == 0) { g.K = 0; g.A = g.B = pl->W.p; }
                gp[sidx] = g;
            }
            pl->gGrandSyrk.upload(gp);
            // k_syrk_blk keeps the K steps of a problem as a table in LDS: SB_MAXST segments / 16-column steps at most
            pl->grand_syrk_blk_ok = pl->grand_syrk_dma_ok = (lg.nf % 16) == 0;
            for (size_t sidx = 0; sidx < lg.nodes.size() && pl->grand_syrk_blk_ok; ++sidx) {
                long steps = 0;
                for (int k = 0; k < gwhere[sidx].second; ++k) steps += gsegs[gwhere[sidx].first + k].K / 16;
                if (gwhere[sidx].second == 0 || gwhere[sidx].second > SB_MAXST || steps > SB_MAXST) pl->grand_syrk_blk_ok = false;
                if (gwhere[sidx].second == 0 || gwhere[sidx].second > SD_MAXST || steps > SD_MAXST) pl->grand_syrk_dma_ok = false;
            }
        }
        {
            std::vector<FrontProb> pf(lv.nodes.size());
            for (size_t sidx = 0; sidx < lv.nodes.size(); ++sidx)
                pf[sidx] = FrontProb{lv.F.p + sidx * (size_t)lv.nf * lv.ldf, lv.invF.p + sidx * (size_t)lv.cwt * 256, lv.nf, lv.cwt,
                                     lv.nodes[sidx], (int)where[sidx].first, where[sidx].second};
            pl->gParentFront.upload(pf);
            const long nt = lv.nf / 16, ntiles = nt * (nt + 1) / 2;
            const long npanel = lv.cwt * nt - lv.cwt * (lv.cwt - 1) / 2;
            pl->parent_front_lds = (size_t)(2 * lv.nf * PF_LD + (npanel + lv.cwt) * FT_SZ) * sizeof(double);
            pl->parent_front_nacc = ntiles <= 16 ? 2 : (ntiles <= 32 ? 4 : (ntiles <= 64 ? 8 : (ntiles <= 96 ? 12 : 0)));
            if (pl->parent_front_lds > 160 * 1024 || lv.panel_only) pl->parent_front_nacc = 0;
        }
    }
    tr.mark("parent / grandparent descriptors");
    pl->hLeafResid = gr; pl->leaf_nobs_host = nobs;
    pl->gLeafResidLik.upload(grl);
    pl->gLeaf.upload(lp); pl->gLeafResid.upload(gr); pl->gLeafSyrk.upload(gs); pl->gLeafUpdate.upload(gu);
    pl->gLeafCholFull.upload(pf); pl->gLeafCholLik.upload(pk); pl->gLeafCholC.upload(pc);
    {
        // the same problems with the matrices of at most 8 tiles first (k_chol_tiles<8> at three workgroups per CU, then the few larger ones)
        std::vector<PanelProb> ps;
        for (int pass = 0; pass < 2; ++pass)
            for (const PanelProb& p : pc) if ((p.ne <= 8) == (pass == 0)) ps.push_back(p);
        pl->n_chol_small = 0;
        for (const PanelProb& p : pc) if (p.ne <= 8) ++pl->n_chol_small;
        pl->gLeafCholSorted.upload(ps);
    }
    for (int v = 0; v < 2; ++v) { pl->gBigPanel[v].clear(); pl->gBigTrail[v].clear(); pl->bigM[v].clear(); pl->bigN[v].clear(); }
    if (pl->leaf_max_nop / 16 > 12) {
        const int NBT = 4;                                        // column tiles per step
        const int nsteps = (pl->leaf_max_nop / 16 + NBT - 1) / NBT;
        for (int v = 0; v < 2; ++v) {
            pl->gBigPanel[v].resize(nsteps); pl->gBigTrail[v].resize(nsteps);
            pl->bigM[v].assign(nsteps, 0); pl->bigN[v].assign(nsteps, 0);
            for (int st = 0; st < nsteps; ++st) {
                std::vector<PanelProb> pp(nl);
                std::vector<GemmProb> gg(nl);
                for (size_t t = 0; t < nl; ++t) {
                    const PanelProb& full = v == 0 ? pf[t] : pk[t];
                    const int ntl = full.ne, c0 = st * NBT;               // column tiles of this leaf, first tile of the step
                    const int ne = std::max(0, std::min(NBT, ntl - c0));
                    const long nop = full.ld;
                    PanelProb q = full;
                    q.P = full.P + (size_t)c0 * 16 * nop + (size_t)c0 * 16;
                    q.invd = full.invd + (size_t)c0 * 256;
                    q.ht = ne > 0 ? full.ht - c0 : 0;
                    q.ne = ne;
                    pp[t] = q;
                    GemmProb g{};
                    const int c1 = c0 + ne;
                    g.M = ne > 0 ? (full.ht - c1) * 16 : 0;
                    g.N = ne > 0 ? (ntl - c1) * 16 : 0;
                    if (g.N <= 0 || g.M <= 0) { g.M = 0; g.N = 0; }
                    g.K = ne * 16;
                    g.A = full.P + (size_t)c1 * 16 * nop + (size_t)c0 * 16; g.lda = nop;
                    g.B = g.A; g.ldb = nop;
                    g.C = full.P + (size_t)c1 * 16 * nop + (size_t)c1 * 16; g.ldc = nop;
                    gg[t] = g;
                    pl->bigM[v][st] = std::max<long>(pl->bigM[v][st], g.M);
                    pl->bigN[v][st] = std::max<long>(pl->bigN[v][st], g.N);
                }
                pl->gBigPanel[v][st].upload(pp);
                pl->gBigTrail[v][st].upload(gg);
            }
        }
    }
    pl->gLeafTrsmFull.upload(tf); pl->gLeafTrsmLik.upload(tk);
    const std::vector<Trsm2Prob> tfg = tf, tkg = tk;          // with the Ut gather
    for (auto& e : tf) e.gtiles = 0;
    for (auto& e : tk) e.gtiles = 0;
    {
        // most leaves need at most 8 column tiles: they get the 8-tile instance (fewer registers, two
        // workgroups per CU), the few larger ones the 12-tile instance
        std::vector<Trsm2Prob> tf2, tk2;
        pl->trsm_small_nt = pl->trsm_small_tiles_full = pl->trsm_small_tiles_lik = 0;
        for (int pass = 0; pass < 2; ++pass)
            for (size_t t = 0; t < nl; ++t) {
                const bool small = tf[t].nt <= 8;
                if (small != (pass == 0)) continue;
                tf2.push_back(tf[t]); tk2.push_back(tk[t]);
                if (small) {
                    pl->trsm_small_nt = std::max(pl->trsm_small_nt, tf[t].nt);
                    pl->trsm_small_tiles_full = std::max(pl->trsm_small_tiles_full, tf[t].ntiles);
                    pl->trsm_small_tiles_lik = std::max(pl->trsm_small_tiles_lik, tk[t].ntiles);
                }
            }
        pl->n_trsm_small = 0;
        for (size_t t = 0; t < nl; ++t) if (tf[t].nt <= 8) ++pl->n_trsm_small;
        pl->gLeafTrsmFullPlain.upload(tf2); pl->gLeafTrsmLikPlain.upload(tk2);
        {
            std::vector<Trsm2Prob> tf2g, tk2g;                 // same order, Ut gathered from W by the row solve itself
            for (int pass = 0; pass < 2; ++pass)
                for (size_t t = 0; t < nl; ++t)
                    if ((tf[t].nt <= 8) == (pass == 0)) { tf2g.push_back(tfg[t]); tk2g.push_back(tkg[t]); }
            pl->gLeafTrsmFullPlainG.upload(tf2g); pl->gLeafTrsmLikPlainG.upload(tk2g);
        }
        {
            // fused row solve + update (k_leaf_solve_update) for the leaves with at most 8 observation tiles, in the same order
            std::vector<LeafSolveProb> sp;
            std::vector<GemmProb> up;
            int nat_max = 0;
            for (int pass = 0; pass < 2; ++pass)
                for (size_t t = 0; t < nl; ++t) {
                    const bool small = tf[t].nt <= 8;
                    if (small != (pass == 0)) continue;
                    const int i = pl->leaf_nodes[t];
                    const int m = pl->node_level[i];
                    const int na = pl->na[m], a0 = pl->asuf[m], nop = pl->leaf_nop[t];
                    const long r0 = pl->row0[i], nr = pl->row1[i] - r0;
                    double* Pn = pl->panel.p + pl->leaf_poff[t];
                    LeafSolveProb q{};
                    q.L = Pn; q.invd = pl->leafInv.p + pl->leaf_ioff[t];
                    q.V = Pn + (size_t)(nop + na) * nop; q.Ut = Pn + (size_t)nop * nop;
                    q.Wr = pl->W.p + r0 * pl->ldw + a0; q.var = pl->var.p + r0;
                    q.ldL = nop; q.ldx = nop; q.ldw = pl->ldw;
                    q.nt = nop / 16; q.nrt = (int)(nr / 16); q.nat = na / 16; q.zt = (na - MRA_YB) / 16;
                    sp.push_back(q);
                    up.push_back(gu[t]);
                    if (small) nat_max = std::max(nat_max, q.nat);
                }
            pl->gLeafSolve.upload(sp);
            pl->gLeafUpdatePlain.upload(up);
            {
                std::vector<LeafSolveProb> half;
                for (size_t k = 0; k < pl->n_trsm_small && k < sp.size(); ++k) {
                    const LeafSolveProb& q = sp[k];
                    const int h0 = (q.nrt + 1) / 2;
                    for (int part = 0; part < 2; ++part) {
                        LeafSolveProb h = q;
                        const int t0 = part ? h0 : 0, t1 = part ? q.nrt : h0;
                        if (t1 <= t0) continue;
                        h.V = q.V + (size_t)t0 * 16 * q.ldx; h.Wr = q.Wr + (size_t)t0 * 16 * q.ldw; h.var = q.var + (size_t)t0 * 16; h.nrt = t1 - t0;
                        half.push_back(h);
                    }
                }
                pl->n_leaf_solve_half = half.size();
                pl->gLeafSolveHalf.upload(half);
            }
            {
                std::vector<long> lr0(nl);
                std::vector<unsigned char> lup(nl);
                for (size_t t = 0; t < nl; ++t) { lr0[t] = pl->row0[pl->leaf_nodes[t]]; lup[t] = tf[t].nt <= 8 ? 1 : 0; }
                pl->leaf_row0_dev.upload(lr0); pl->leaf_upd_dev.upload(lup);
            }
            const int nts = pl->trsm_small_nt;
            pl->leaf_solve_lds = (size_t)((nts * (nts - 1) / 2 + nts) * FT_SZ + 2 * nat_max * 16 * LG_LD) * sizeof(double);
            pl->leaf_solve_ok = pl->n_trsm_small > 0 && nat_max <= 13 && nat_max > 0 && pl->leaf_solve_lds <= 160 * 1024 && pl->leaf_max_rows >= 128;
        }
    }
    arena.finish();
    tr.mark("descriptor arena to the device");
}

// ------------------------------------------------------------------------------------------------
//  launch helpers
// ------------------------------------------------------------------------------------------------
// ---- arrays that only the general (level-by-level) kernels read, built when one of those kernels is about to be launched ------
static void ensure_row_leaf(mra_plan* pl) {
    if (pl->row_leaf.p) return;
    std::vector<int> rl(pl->P, -1);
    for (size_t t = 0; t < pl->leaf_nodes.size(); ++t) {
        const int i = pl->leaf_nodes[t];
        for (long p = pl->row0[i]; p < pl->row1[i]; ++p) rl[p] = (int)t;
    }
    pl->row_leaf.upload(rl);
}
static void ensure_tile_lists(mra_plan* pl, int m) {
    LevelData& lv = pl->lev[m];
    if (lv.tile_node.p) return;
    std::vector<int> tnode;
    std::vector<long> trow;
    for (size_t s = 0; s < lv.nodes.size(); ++s) {
        const int i = lv.nodes[s];
        for (long t = pl->row0[i]; t < pl->row1[i]; t += 16) { tnode.push_back((int)s); trow.push_back(t); }
    }
    lv.ntiles = (long)tnode.size();
    lv.tile_node.upload(tnode); lv.tile_row0.upload(trow);
}

struct KTimer {
    mra_plan* pl; int fam; hipEvent_t a = nullptr, b = nullptr;
    KTimer(mra_plan* p, int f, const Work& w) : pl(p), fam(f) {
        pl->kstat[fam].launches += 1;
        pl->kstat[fam].flops += w.alg;
        pl->kstat[fam].flops_exec += w.exec;
        pl->kstat[fam].bytes += w.bytes;
        if (pl->ktiming) {
            hipEventCreate(&a); hipEventCreate(&b);
            hipEventRecord(a, pl->stream);
        }
    }
    ~KTimer() {
        if (pl->ktiming) {
            hipEventRecord(b, pl->stream);
            pl->kev.push_back({fam, {a, b}});
        }
    }
};

template <int EPI>
static void launch_gemm(mra_plan* pl, const GemmProb* probs, size_t nprob, long maxM, long maxN, bool allow_lds = true, bool lower_tri = false) {
    mra_launch_gemm(pl, EPI, probs, nprob, maxM, maxN, allow_lds, lower_tri);
}
template <int EPI>
static void launch_leaf_gemm(mra_plan* pl, const GemmProb* probs, size_t nprob) { mra_launch_leaf_gemm(pl, EPI, probs, nprob); }
// the leaf-resident kernel pays off for leaves with many rows and a K loop of at least a few chunks; small leaves (config 5:
// 64 rows, 32 observations) keep the 64x64-tile kernel
static bool leaf_gemm_ok(const mra_plan* pl) { return pl->use_leaf_gemm && pl->leaf_max_rows >= 128 && pl->leaf_max_nop >= 64; }

static void launch_panel(mra_plan* pl, const PanelProb* probs, size_t nprob, int accumulate = 0) {
    if (!nprob) return;
    hipLaunchKernelGGL(k_panel_chol, dim3((unsigned)nprob), dim3(256), 0, pl->stream, probs, pl->dnode.p, pl->errflag.p, accumulate);
}

// row-tile triangular solve with L in LDS; returns false when nt is too large for the LDS path
static bool launch_trsm2(mra_plan* pl, const Trsm2Prob* probs, size_t nprob, int nt, long max_tiles, int tiles_per_wg) {
    if (!pl->prepare_only && (!nprob || max_tiles <= 0 || nt <= 0)) return true;
    if (nt > 12) return false;
    ensure_big_lds(pl, {(const void*)k_trsm_rows2<2>, (const void*)k_trsm_rows2<4>, (const void*)k_trsm_rows2<8>, (const void*)k_trsm_rows2<12>});
    if (pl->prepare_only) return true;
    const size_t lds = (size_t)(nt * (nt - 1) / 2 + nt) * 2048 + (size_t)nt * 16 * sizeof(int);      // L image + the Ut gather list
    const unsigned gx = (unsigned)((max_tiles + tiles_per_wg - 1) / tiles_per_wg);
    const unsigned tb = 512;
#ifdef MRA_STAMPS
    // the last big launch wins the buffer (tools/stamps_trsm.py reads it after a pass)
    unsigned long long* tst = nullptr;
    if (nprob >= 1024 && max_tiles <= 64) {
        if (pl->tstamps.n < nprob * 64 * 8) { pl->tstamps.alloc(nprob * 64 * 8); HIP_TRY(mraMemset(pl->tstamps.p, 0, pl->tstamps.n * sizeof(double))); HIP_TRY(hipDeviceSynchronize()); }
        tst = (unsigned long long*)pl->tstamps.p;
    }
#define MRA_TSTAMP_VAL , tst
#else
#define MRA_TSTAMP_VAL
#endif
    for (size_t off = 0; off < nprob; off += 65535) {
        dim3 grid(gx, (unsigned)std::min<size_t>(65535, nprob - off));
        if (nt <= 2) hipLaunchKernelGGL((k_trsm_rows2<2>), grid, dim3(tb), lds, pl->stream, probs + off, tiles_per_wg MRA_TSTAMP_VAL);
        else if (nt <= 4) hipLaunchKernelGGL((k_trsm_rows2<4>), grid, dim3(tb), lds, pl->stream, probs + off, tiles_per_wg MRA_TSTAMP_VAL);
        else if (nt <= 8) hipLaunchKernelGGL((k_trsm_rows2<8>), grid, dim3(tb), lds, pl->stream, probs + off, tiles_per_wg MRA_TSTAMP_VAL);
        else hipLaunchKernelGGL((k_trsm_rows2<12>), grid, dim3(tb), lds, pl->stream, probs + off, tiles_per_wg MRA_TSTAMP_VAL);
    }
    return true;
}

// whole prior of a regular tree: per level a tiny knot pass (knot rows cascade -> kInv -> Cholesky),
// then ONE cascade over all leaf row tiles that writes W once
static double kernel_cov0(const mra_plan* pl) { return pl->kp.amp; }   // C(x,x) of every stationary kernel: amp * 1

// tiles of the likelihood-only row cascade: 16 observed rows each (obs_idx, padded with -1), with their leaf's ancestor chain; one
// workgroup per leaf (or per family of sibling leaves, as the full cascade groups them)
static void ensure_lik_tiles(mra_plan* pl) {
    if (pl->lik_tiles_valid) return;
    const size_t nl = pl->leaf_nodes.size();
    std::vector<int> chains, tleaf, wgn;
    std::vector<long> wg0;
    for (size_t t = 0; t < nl; ++t) {
        const int i = pl->leaf_nodes[t];
        const long nt = (pl->obs_off_host[t + 1] - pl->obs_off_host[t]) / 16;
        if (!nt) continue;
        int ch[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int a = pl->parent[i]; a >= 0; a = pl->parent[a]) ch[pl->node_level[a]] = pl->node_slot[a];
        const bool join = pl->cascade_group_siblings && !wg0.empty() && t > 0 && pl->parent[i] >= 0 && pl->parent[i] == pl->parent[pl->leaf_nodes[t - 1]] &&
                          wg0.back() + wgn.back() == pl->obs_off_host[t] / 16;
        if (join) wgn.back() += (int)nt;
        else { wg0.push_back(pl->obs_off_host[t] / 16); wgn.push_back((int)nt); }
        for (long k = 0; k < nt; ++k) { tleaf.push_back((int)t); for (int c = 0; c < 8; ++c) chains.push_back(ch[c]); }
    }
    // (tile numbers are positions in obs_idx / 16: leaves without observations own no tile, so the per-tile arrays are dense)
    pl->n_lik_tiles = (long)tleaf.size(); pl->n_lik_wg = (long)wg0.size();
    pl->lik_chain.upload(chains); pl->lik_leaf.upload(tleaf); pl->lik_wg0.upload(wg0); pl->lik_wgn.upload(wgn);
    pl->lik_tiles_valid = true;
}

// Level-by-level path, likelihood-only passes: W is needed at the observed rows (Ut, C) and at the knots (the deeper levels'
// residual products gather them) - at config 5 60 % of the rows.  The sorted list of such rows, padded to 16 with -1 wherever a
// non-leaf node's row range starts or ends; a node's list is a contiguous piece of it; per level the one-launch prior problems over
// blocks of 512 entries.
static bool ensure_lik_general(mra_plan* pl) {
    if (pl->lik_general_valid) return pl->lik_general_ok;
    pl->lik_general_valid = true;
    pl->lik_general_ok = false;
    for (int m = 0; m < pl->n_levels; ++m) if (!pl->lev[m].nodes.empty() && !pl->lev[m].prior_level_ok) return false;
    std::vector<unsigned char> need(pl->y_finite_host);
    if ((long)need.size() != pl->P) return false;
    for (int i = 0; i < pl->n_nodes; ++i)
        if (!pl->leaf[i]) for (long k = pl->knot_ptr[i]; k < pl->knot_ptr[i + 1]; ++k) need[pl->knot_rows[k]] = 1;
    // the list is cut (padded to 16) wherever a non-leaf node's row range starts or ends, so that every such node's rows are a
    // contiguous, tile-aligned piece of it (regular trees: at the boundaries of the leaves' parents, every 256 rows at config 5; a
    // 64-row leaf with 38 needed rows padded on its own would cost 48; a shard's orphan knot rows above its subtree are cut the same way)
    std::vector<long> cuts;
    for (int i = 0; i < pl->n_nodes; ++i) if (!pl->leaf[i]) { cuts.push_back(pl->row0[i]); cuts.push_back(pl->row1[i]); }
    cuts.push_back(0); cuts.push_back(pl->P);
    std::sort(cuts.begin(), cuts.end());
    cuts.erase(std::unique(cuts.begin(), cuts.end()), cuts.end());
    std::vector<int> idx;
    std::vector<long> off(cuts.size(), 0);                   // position in idx of the first list entry at or behind row cuts[c]
    idx.reserve((size_t)pl->P);
    for (size_t c = 0; c + 1 < cuts.size(); ++c) {
        off[c] = (long)idx.size();
        for (long p = cuts[c]; p < cuts[c + 1]; ++p) if (need[p]) idx.push_back((int)p);
        while (idx.size() % 16) idx.push_back(-1);
    }
    off[cuts.size() - 1] = (long)idx.size();
    if (idx.empty()) return false;
    pl->need_idx.upload(idx);
    const long blk = 512;
    for (int m = 0; m < pl->n_levels; ++m) {
        LevelData& lv = pl->lev[m];
        const size_t nn = lv.nodes.size();
        if (!nn) continue;
        std::vector<GemmProb> fz;
        long rows_needed = 0, rows_all = 0;
        const int Kanc = pl->Ka - lv.a0;
        for (size_t s = 0; s < nn; ++s) {
            const int i = lv.nodes[s];
            const size_t c0 = (size_t)(std::lower_bound(cuts.begin(), cuts.end(), pl->row0[i]) - cuts.begin());
            const size_t c1 = (size_t)(std::lower_bound(cuts.begin(), cuts.end(), pl->row1[i]) - cuts.begin());
            const long o0 = off[c0], o1 = off[c1];
            rows_needed += o1 - o0; rows_all += pl->row1[i] - pl->row0[i];
            for (long b0 = o0; b0 < o1; b0 += blk) {
                GemmProb g{};
                g.A = pl->W.p + lv.a0; g.lda = pl->ldw;
                g.B = pl->W.p + lv.a0; g.ldb = pl->ldw; g.idxB = pl->knot_idx.p + pl->knot_idx_off[i];
                g.C = pl->W.p + lv.c0; g.ldc = pl->ldw;
                g.XA = pl->X.p; g.XB = pl->X.p;
                g.M = (int)std::min(blk, o1 - b0); g.N = lv.cw; g.K = Kanc; g.lower = 0;
                g.idxA = pl->need_idx.p + b0;
                g.solveL = lv.Lp.p + s * (size_t)lv.cw * lv.cw; g.solveI = lv.invP.p + s * (size_t)lv.cwt * 256; g.var = pl->var.p;
                fz.push_back(g);
            }
        }
        lv.gResidLik.upload(fz);
        lv.lik_share = rows_all > 0 ? (double)rows_needed / (double)rows_all : 1.0;
    }
    pl->lik_general_ok = true;
    return true;
}

static void run_prior_fused(mra_plan* pl) {
    const int cw = pl->cw[0];
    CascadeArgs base{};
    base.ycol = -1;
    for (int m = 0; m < pl->NL; ++m) {
        base.lev[m].kx = pl->fl[m].kx.p; base.lev[m].kvalid = pl->fl[m].kvalid.p; base.lev[m].Wk = pl->fl[m].Wk.p;
        base.lev[m].L = pl->lev[m].Lp.p; base.lev[m].invd = pl->lev[m].invP.p;
        base.coff[m] = pl->coff[m];
    }
    base.X = pl->X.p; base.W = pl->W.p; base.ldw = pl->ldw;
    const int n_chain = (pl->use_knot_chain && pl->knot_chain_ok && pl->kc_levels >= 2) ? pl->kc_levels : 0;
    if (n_chain) {
        Work fl;
        for (int m = 0; m < n_chain; ++m) fl += pl->lev[m].fl_pchol;
        KTimer kt(pl, KF_PRIOR_CHOL, fl);
        KnotChainArgs ka{};
        for (int m = 0; m < n_chain; ++m) {
            ka.lev[m] = base.lev[m];
            ka.node_base[m] = (int)pl->level_ptr[m];
        }
        ka.chain = pl->kc_chain.p; ka.ownmask = pl->kc_ownmask.p; ka.knots = pl->kc_knots.p; ka.nl = n_chain; ka.err = pl->errflag.p;
#ifdef MRA_STAMPS
        {
            const size_t need = pl->lev[n_chain - 1].nodes.size() * 64;
            if (pl->kstamps.n < need) { pl->kstamps.alloc(need); HIP_TRY(mraMemset(pl->kstamps.p, 0, need * sizeof(double))); HIP_TRY(hipDeviceSynchronize()); }
            ka.stamps = (unsigned long long*)pl->kstamps.p;
        }
#endif
        launch_knot_chain(pl, ka);
    }
    for (int m = n_chain; m < pl->NL; ++m) {
        // one launch per level: knot rows cascade -> Wk, kInv, Cholesky factor, inverted diagonal blocks
        LevelData& lv = pl->lev[m];
        KTimer kt(pl, KF_PRIOR_CHOL, lv.fl_pchol);
        CascadeArgs ar = base;
        ar.knot_mode = 1; ar.mlast = m - 1; ar.n_wg = pl->fl[m].n_kwg; ar.knot_threads = pl->fl[m].k_threads;
        ar.wg_tile0 = pl->fl[m].kt_wg0.p; ar.wg_ntiles = pl->fl[m].kt_wgn.p;
        ar.tile_rows = pl->fl[m].kt_rows.p; ar.tile_chain = pl->fl[m].kt_chain.p; ar.tile_knot0 = pl->fl[m].kt_knot0.p;
        ar.Wk_out = pl->fl[m].Wk.p;
        ar.Lp_out = lv.Lp.p; ar.invd_out = lv.invP.p; ar.err = pl->errflag.p;
        ar.node_base = (int)pl->level_ptr[m];            // regular trees: slot s of level m is node level_ptr[m] + s
        launch_cascade_any(pl, ar);
    }
    {
        Work fl;
        for (int m = 0; m < pl->NL; ++m) fl += pl->lev[m].fl_resid + pl->lev[m].fl_trsm;
        // one pass: coordinates and y in, W (all levels + y block) and the prior variance out, observed rows once more into Ut
        fl.bytes = 8.0 * pl->P * (pl->d + 1 + pl->ldw + 1) + pl->by_leaf_ut;
        // a likelihood needs W at the OBSERVED rows only (Ut and the leaves' C = v(o,o) + R I are built from them; the knots had their
        // own pass): the cascade then walks gathered tiles of observed rows - 7 of a leaf's 16 tiles at C3
        const bool lik_rows = !(pl->run_flags & MRA_RUN_PREDICT) && pl->use_lik_rows && pl->cascade_stage_all && pl->gemm_lds &&
                              pl->leaf_max_nop / 16 <= 12 && pl->leaf_max_nop > 0 && !pl->obs_off_host.empty() && pl->obs_off_host.back() > 0;
        if (lik_rows) {
            ensure_lik_tiles(pl);
            const double share = 16.0 * (double)pl->n_lik_tiles / (double)std::max<long>(pl->P, 1);
            fl.alg *= share; fl.exec *= share; fl.bytes = share * 8.0 * pl->P * (pl->d + 1 + pl->ldw) + pl->by_leaf_ut;
        }
        KTimer kt(pl, KF_PRIOR_TRSM, fl);
        CascadeArgs ar = base;
        ar.knot_mode = 0; ar.mlast = pl->NL - 1; ar.dbg = pl->dbg;
#ifdef MRA_STAMPS
        if (!pl->stamps.p) { pl->stamps.alloc((size_t)pl->n_ftiles * 16); HIP_TRY(mraMemset(pl->stamps.p, 0, pl->stamps.n * sizeof(double))); }
        ar.stamps = (unsigned long long*)pl->stamps.p;
#endif
        ar.var_out = pl->var.p; ar.cov0 = kernel_cov0(pl);
        ar.ycol = pl->Ka; ar.y = pl->y.p;
        if (pl->leaf_max_nop / 16 <= 12 && pl->leaf_max_nop > 0 && !pl->ut_gather) {
            ar.obs_pos = pl->obs_pos.p; ar.tile_leaf = pl->ft_leaf.p; ar.leaf_ut = pl->leaf_ut.p; ar.leaf_nop = pl->leaf_nop_dev.p;
            ar.y = pl->y.p;
            // Ut rows follow W's ancestor columns of a last-level leaf: a = column - asuf[NL]
            for (int m = 0; m < pl->NL; ++m) ar.ut_off[m] = pl->coff[m] - pl->asuf[pl->NL];
            ar.ut_yrow = pl->Ka - pl->asuf[pl->NL];
        }
        if (pl->cascade_stage_all) { ar.n_wg = pl->n_fwg_leaf; ar.wg_tile0 = pl->ft_wg0_leaf.p; ar.wg_ntiles = pl->ft_wgn_leaf.p; }
        else { ar.n_wg = pl->n_fwg; ar.wg_tile0 = pl->ft_wg0.p; ar.wg_ntiles = pl->ft_wgn.p; }
        ar.tile_row0 = pl->ft_row0.p; ar.tile_chain = pl->ft_chain.p;
        if (lik_rows) {
            ar.row_gather = pl->obs_idx.p; ar.tile_chain = pl->lik_chain.p; ar.tile_leaf = pl->lik_leaf.p;
            ar.n_wg = pl->n_lik_wg; ar.wg_tile0 = pl->lik_wg0.p; ar.wg_ntiles = pl->lik_wgn.p;
            ar.var_out = nullptr;                          // the prior variance is the predictive pass's (the y block stays: Ut's y row is gathered from it)
        }
        launch_cascade_any(pl, ar);
    }
}

// predictive pass of a deep 64-wide tree (regular_hi): W holds the whitened basis after the leaf update
static void run_predict_hi(mra_plan* pl) {
    const int NL = pl->NL, nlo = NL - 4;
    PredHiArgs hi{};
    Work fl_hi, fl_lo;
    for (int h = 0; h < 4; ++h) {
        const int m = nlo + h;
        hi.hi[h].F = pl->lev[m].F.p; hi.hi[h].invF = pl->lev[m].invF.p; hi.hi[h].nf = pl->lev[m].nf;
        hi.hi[h].ld = pl->lev[m].ldf; hi.hi[h].stride = (long)pl->lev[m].nf * pl->lev[m].ldf;
        hi.coff_hi[h] = pl->coff[m];
        fl_hi += pl->lev[m].fl_trsm + pl->lev[m].fl_update;
    }
    hi.W = pl->W.p; hi.var = pl->var.p; hi.ldw = pl->ldw;
    hi.col_low = pl->coff[nlo - 1]; hi.n_low = nlo * 4 + 1; hi.lev0 = nlo;
    hi.tile_row0 = pl->ft_row0.p; hi.tile_chain = pl->ft_chain.p; hi.wg_tile0 = pl->ft_wg0.p; hi.wg_ntiles = pl->ft_wgn.p;
    // W read once, its coarse columns and y block written once; var in and out
    fl_hi.bytes = 8.0 * pl->P * (pl->ldw + (pl->ldw - pl->coff[nlo - 1]) + 2);
    if (pl->hi_fold_now) {
        // the leaf update rides in k_predict_hi (Tt and Ut in, nothing out)
        hi.wg_leaf = pl->hi_wgleaf.p; hi.leaf_ut = pl->leaf_ut.p; hi.leaf_nop = pl->leaf_nop_dev.p; hi.leaf_row0 = pl->leaf_row0_dev.p; hi.na = pl->na[NL];
        fl_hi += Work(pl->fl_leaf_update.alg, pl->fl_leaf_update.exec, pl->by_leaf_tt + pl->by_leaf_ut);
    }
    PredArgs lo{};
    for (int m = 0; m < nlo; ++m) {
        lo.lev[m].F = pl->lev[m].F.p; lo.lev[m].invF = pl->lev[m].invF.p; lo.lev[m].nf = pl->lev[m].nf;
        lo.lev[m].ld = pl->lev[m].ldf; lo.lev[m].stride = (long)pl->lev[m].nf * pl->lev[m].ldf;
        lo.coff[m] = pl->coff[m];
        fl_lo += pl->lev[m].fl_trsm + pl->lev[m].fl_update;
    }
    lo.deep = lo.lev[nlo - 1];
    lo.W = pl->W.p; lo.mean = pl->mean.p; lo.var = pl->var.p; lo.ldw = pl->ldw; lo.ycol = pl->Ka;
    lo.tile_row0 = pl->ft_row0.p; lo.tile_chain = pl->ft_chain.p; lo.wg_tile0 = pl->hi_wg0_8.p; lo.wg_ntiles = pl->hi_wgn_8.p;
    lo.n_wg = pl->n_hi_wg8; lo.nl = nlo;
    fl_lo.bytes = 8.0 * pl->P * ((pl->ldw - pl->coff[nlo - 1]) + 3);
    const size_t lds_low = (size_t)(4 * 3 / 2 + 4 + ((nlo - 1) * 4 + 1) * 4) * 2048;
    // two launches; the kernel timer brackets both (the coarse share is reported with them: they are one pass over W's coarse half)
    KTimer kt(pl, KF_PRED_UPDATE, fl_hi + fl_lo);
    launch_predict_hi(pl, hi, lo, lds_low, pl->hi_fold_now);
}

static void run_predict_fused(mra_plan* pl) {
    PredArgs ar{};
    for (int m = 0; m < pl->NL; ++m) {
        ar.lev[m].F = pl->lev[m].F.p; ar.lev[m].invF = pl->lev[m].invF.p; ar.lev[m].nf = pl->lev[m].nf;
        ar.lev[m].ld = pl->lev[m].ldf; ar.lev[m].stride = (long)pl->lev[m].nf * pl->lev[m].ldf;
        ar.coff[m] = pl->coff[m];
    }
    ar.deep = ar.lev[pl->NL - 1];
    ar.W = pl->W.p; ar.mean = pl->mean.p; ar.var = pl->var.p; ar.ldw = pl->ldw; ar.ycol = pl->Ka;
    ar.tile_row0 = pl->ft_row0.p; ar.tile_chain = pl->ft_chain.p; ar.wg_tile0 = pl->ft_wg0_x.p; ar.wg_ntiles = pl->ft_wgn_x.p;
    ar.n_wg = pl->n_fwg_x; ar.nl = pl->NL;
    const int cwt = pl->CWT, mmax = pl->NL - 1;
    size_t lds = (size_t)(cwt * (cwt - 1) / 2 + cwt + (mmax * cwt + 1) * cwt) * 2048;
    Work fl;
    for (int m = 0; m < pl->NL; ++m) fl += pl->lev[m].fl_trsm + pl->lev[m].fl_update;
    fl.bytes = 8.0 * pl->P * (pl->ldw + 3);                // W once in, var in/out, mean out (the level operands stay in L2)
    if (pl->pred_update_now) {
        // the leaf update rides in this launch (two Ut chunk stages share the LDS with the level operands)
        ar.tile_leaf = pl->ft_leaf.p; ar.wg_leaf = pl->ft_wgleaf_x.p; ar.leaf_ut = pl->leaf_ut.p; ar.leaf_nop = pl->leaf_nop_dev.p; ar.leaf_row0 = pl->leaf_row0_dev.p;
        ar.leaf_upd = pl->leaf_upd_dev.p; ar.na = pl->na[pl->NL];
        // two 8-k chunks of Ut behind the deepest level's half A (which is requested while the last chunk's products issue)
        lds = std::max(lds, (size_t)(cwt * (cwt - 1) / 2 + cwt + (mmax / 2) * cwt * cwt) * 2048 + (size_t)(2 * (pl->NL * cwt + 1) * 128) * sizeof(double));
        fl += Work(pl->fl_leaf_update.alg, pl->fl_leaf_update.exec, pl->by_leaf_tt + pl->by_leaf_ut);    // Tt and Ut in, nothing out
    }
#ifdef MRA_STAMPS
    if (!pl->pstamps.p) { pl->pstamps.alloc((size_t)pl->n_ftiles * 16); HIP_TRY(mraMemset(pl->pstamps.p, 0, pl->pstamps.n * sizeof(double))); }
    ar.stamps = (unsigned long long*)pl->pstamps.p;
#endif
    KTimer kt(pl, KF_PRED_UPDATE, fl);
    if (ar.n_wg <= 0) return;
    launch_predict_any(pl, ar, lds);
}

// events 0 and 5 bracket every pass; the four inner phase boundaries are recorded only with kernel timing on
// (each hipEventRecord between dependent launches leaves a ~6 us gap on the stream)
static void phase_mark(mra_plan* pl, int k) { if (k == 0 || k == 5 || pl->ktiming) hipEventRecord(pl->ev[k], pl->stream); }

// factorise the fronts of level m (already assembled in global memory unless do_assemble): fused kernel when the
// front (or at least its panel) fits in LDS, else panel Cholesky + Schur GEMM as separate launches
static bool run_front_fused(mra_plan* pl, int m, bool do_assemble, bool add_identity) {
    LevelData& lv = pl->lev[m];
    const size_t nn = lv.nodes.size();
    if (!nn) return true;
    if (!pl->use_front_fused || lv.front_mode == 0 || (do_assemble && lv.front_mode != 2)) return false;
    ensure_big_lds(pl, {(const void*)k_front<true>, (const void*)k_front<false>});
    KTimer kt(pl, KF_FRONT_CHOL, lv.fl_fchol + lv.fl_schur);
    if (lv.front_mode == 2)
        hipLaunchKernelGGL(k_front<true>, dim3((unsigned)nn), dim3(512), lv.front_lds, pl->stream, lv.gFront.p, pl->asmKids.p,
                           pl->dnode.p, pl->errflag.p, do_assemble ? 1 : 0, add_identity ? 1 : 0);
    else
        hipLaunchKernelGGL(k_front<false>, dim3((unsigned)nn), dim3(512), lv.front_lds, pl->stream, lv.gFront.p, pl->asmKids.p,
                           pl->dnode.p, pl->errflag.p, 0, add_identity ? 1 : 0);
    return true;
}

static void run_front_level(mra_plan* pl, int m) {
    LevelData& lv = pl->lev[m];
    const size_t nn = lv.nodes.size();
    if (!nn) return;
    { KTimer kt(pl, KF_FRONT_CHOL, lv.fl_fchol); launch_panel(pl, lv.gFrontChol.p, nn); }
    { KTimer kt(pl, KF_FRONT_SCHUR, lv.fl_schur); launch_gemm<EPI_SUB>(pl, lv.gSchur.p, nn, lv.na, lv.na, false, true); }
}

// sum_in / sum_n / sum_out: also add up that many log-determinants (in order) into *sum_out, in the same launch
static bool run_assemble_level(mra_plan* pl, int m, bool with_identity, const double* sum_in = nullptr, int sum_n = 0, double* sum_out = nullptr) {
    LevelData& lv = pl->lev[m];
    const size_t nn = lv.nodes.size();
    if (!nn) return false;
    KTimer kt(pl, KF_MISC, 0);
    const long total = (long)lv.nf * lv.nf;
    dim3 grid((unsigned)std::min<long>((total + 255) / 256, 64), (unsigned)(nn + (sum_in ? 1 : 0)));
    hipLaunchKernelGGL(k_assemble, grid, dim3(256), 0, pl->stream, lv.gAsm.p, pl->asmKids.p, with_identity ? 1 : 0, sum_in, sum_n, sum_out);
    return true;
}

static void run_add_identity(mra_plan* pl, int m) {
    LevelData& lv = pl->lev[m];
    const size_t nn = lv.nodes.size();
    if (!nn) return;
    KTimer kt(pl, KF_MISC, 0);
    hipLaunchKernelGGL(k_add_identity, dim3((unsigned)((lv.cw + 63) / 64), (unsigned)nn), dim3(64), 0, pl->stream, lv.gAsm.p);
}

static void finish_run(mra_plan* pl);

static void run_fronts_and_predict(mra_plan* pl, int m_from, bool resume) {
    for (int m = m_from; m >= 0; --m) {
        const bool is_red = (m == pl->reduce_level);
        bool red_summed = false;             // the reduce level's log-det sum was written by its assembly launch
        bool need_identity = false;          // front sits in global memory without its identity block
        bool assembled = true;
        if (!(resume && m == m_from)) {
            if (pl->direct_parent && m == pl->NL - 1) {
                const LevelData& lvp = pl->lev[m];
                if (pl->use_front_fused && pl->parent_front_nacc > 0) {
                    // children's Ut -> front -> Lt, Zt, Schur block, all in one launch (the front never visits HBM unfactorised)
                    ensure_big_lds(pl, {(const void*)k_parent_front<2>, (const void*)k_parent_front<4>, (const void*)k_parent_front<8>, (const void*)k_parent_front<12>});
                    KTimer kt(pl, KF_LEAF_SYRK, (pl->fl_leaf_syrk + lvp.fl_fchol + lvp.fl_schur).with_bytes(pl->by_leaf_ut + 8.0 * lvp.nodes.size() * (0.5 * lvp.nf * (lvp.nf + 1))));
                    const dim3 grid((unsigned)lvp.nodes.size());
                    const size_t lds = pl->parent_front_lds;
                    switch (pl->parent_front_nacc) {
                        case 2: hipLaunchKernelGGL(k_parent_front<2>, grid, dim3(512), lds, pl->stream, pl->gParentFront.p, pl->parentSegs.p, pl->dnode.p, pl->errflag.p); break;
                        case 4: hipLaunchKernelGGL(k_parent_front<4>, grid, dim3(512), lds, pl->stream, pl->gParentFront.p, pl->parentSegs.p, pl->dnode.p, pl->errflag.p); break;
                        case 8: hipLaunchKernelGGL(k_parent_front<8>, grid, dim3(512), lds, pl->stream, pl->gParentFront.p, pl->parentSegs.p, pl->dnode.p, pl->errflag.p); break;
                        default: hipLaunchKernelGGL(k_parent_front<12>, grid, dim3(512), lds, pl->stream, pl->gParentFront.p, pl->parentSegs.p, pl->dnode.p, pl->errflag.p); break;
                    }
                    continue;
                }
                if (lvp.panel_only) {
                    // only the panel columns of these fronts exist: build them, factorise them, done with the level
                    const size_t nnp = lvp.nodes.size();
                    ensure_big_lds(pl, {(const void*)k_parent_front<2>, (const void*)k_parent_front<4>});
                    {
                        KTimer kt(pl, KF_LEAF_SYRK, pl->fl_parent_panel);
                        if (lvp.cwt <= 4) hipLaunchKernelGGL(k_parent_front<2>, dim3((unsigned)nnp), dim3(512), pl->parent_own_lds, pl->stream, pl->gParentOwn.p, pl->parentSegs.p, pl->dnode.p, pl->errflag.p);
                        else hipLaunchKernelGGL(k_parent_front<4>, dim3((unsigned)nnp), dim3(512), pl->parent_own_lds, pl->stream, pl->gParentOwn.p, pl->parentSegs.p, pl->dnode.p, pl->errflag.p);
                        launch_gemm<EPI_SET>(pl, pl->gParentPanel.p, nnp, lvp.na, lvp.cw, pl->seg_gemm_lds && pl->parent_panel_lds_ok, false);
                    }
                    {
                        KTimer kt(pl, KF_FRONT_CHOL, pl->lev[m].fl_fchol.with_bytes(8.0 * 2 * nnp * (double)lvp.na * lvp.cw));
                        if (!launch_trsm2(pl, pl->gParentZt.p, nnp, lvp.cwt, lvp.na / 16, lvp.na / 16))
                            throw MraError(MRA_ERR_STATE, "panel row solve: block too wide for the LDS row solve");
                    }
                    continue;
                }
                KTimer kt(pl, KF_LEAF_SYRK, pl->fl_leaf_syrk);
                launch_gemm<EPI_SET>(pl, pl->gParentSyrk.p, lvp.nodes.size(), lvp.nf, lvp.nf, false, true);
            } else if (pl->direct_parent && m == pl->NL - 2 && pl->lev[pl->NL - 1].panel_only) {
                if (is_red) throw MraError(MRA_ERR_STATE, "the reduce level cannot be the level above panel-only fronts");
                const LevelData& lg = pl->lev[m];
                KTimer kt(pl, KF_FRONT_SCHUR, pl->fl_grand_syrk);
                if (pl->use_syrk_blk && pl->grand_syrk_blk_ok) mra_launch_syrk_blk(pl, pl->gGrandSyrk.p, lg.nodes.size(), lg.nf, pl->use_syrk_blk == 1 && pl->grand_syrk_dma_ok);
                else launch_gemm<EPI_SET>(pl, pl->gGrandSyrk.p, lg.nodes.size(), lg.nf, lg.nf, false, true);      // (64 x 64 LDS-tiled: 24.2 vs 23.7 ms at config 5)
            } else if (is_red) {
                // (the rank-local log-det sum of everything below rides in the same launch, see the reduce block)
                LevelData& lvr = pl->lev[m];
                const int lo_r = (int)pl->level_ptr[m + 1];
                red_summed = run_assemble_level(pl, m, false, pl->dnode.p + lo_r, pl->n_nodes - lo_r, lvr.F.p + (lvr.F.n - 16));
                need_identity = true;
            } else {
                assembled = false;               // the fused front kernel assembles in LDS when the whole front fits
            }
            if (is_red) {
                // the reduce level's fronts are summed over ranks WITHOUT their identity blocks; the
                // 16-double tail of the buffer carries the rank-local log-det sum of everything below
                LevelData& lv = pl->lev[m];
                const int lo = (int)pl->level_ptr[m + 1];
                if (!red_summed) {
                    KTimer kt(pl, KF_MISC, 0);
                    hipLaunchKernelGGL(k_sum_dnode, dim3(1), dim3(256), 0, pl->stream, pl->dnode.p + lo, pl->n_nodes - lo,
                                       lv.F.p + (lv.F.n - 16));
                }
                if (pl->run_flags & MRA_RUN_SPLIT) { pl->split_pending = true; return; }
                // a reduce level without a transport would silently build the likelihood from this rank's partial fronts
                if (!pl->comm || !pl->allreduce)
                    throw MraError(MRA_ERR_STATE, "reduce level set but neither a communicator (mra_comm_init) nor MRA_RUN_SPLIT: "
                                                  "the fronts of the reduce level would not be summed over ranks");
                if (pl->allreduce(lv.F.p, lv.F.p, lv.F.n, ncclDouble, ncclSum, pl->comm, pl->stream) != ncclSuccess)
                    throw MraError(MRA_ERR_COMM, "ncclAllReduce failed");
            }
        } else {
            need_identity = is_red;
        }
        if (!assembled) {
            if (run_front_fused(pl, m, true, true)) continue;
            run_assemble_level(pl, m, true);
        }
        if (run_front_fused(pl, m, false, need_identity)) continue;
        if (need_identity) run_add_identity(pl, m);
        run_front_level(pl, m);
    }
    phase_mark(pl, 3);
    if (pl->side_pending) {                                 // join: the leaf update on the side stream
        HIP_TRY(hipStreamWaitEvent(pl->stream, pl->ev_join, 0));
        pl->side_pending = false;
    }
    const bool fusedp = pl->regular && pl->use_fused && !pl->host_cov;
    const bool hip_ = !fusedp && pl->regular_hi && pl->use_fused && !pl->host_cov;      // sharded plans too: the walk is per row tile, the fronts above the reduce level are complete by now
    if ((pl->run_flags & MRA_RUN_PREDICT) && fusedp) run_predict_fused(pl);
    if ((pl->run_flags & MRA_RUN_PREDICT) && hip_) run_predict_hi(pl);
    if ((pl->run_flags & MRA_RUN_PREDICT) && !fusedp && !hip_) {
        for (int m = pl->n_levels - 1; m >= 0; --m) {
            LevelData& lv = pl->lev[m];
            const size_t nn = lv.nodes.size();
            if (!nn) continue;
            {
                KTimer kt(pl, KF_PRED_TRSM, lv.fl_trsm);
                if (!launch_trsm2(pl, lv.gTrsm2Post.p, nn, lv.cwt, lv.max_tiles, 32)) {
                    ensure_tile_lists(pl, m);
                    hipLaunchKernelGGL(k_trsm_rows, dim3((unsigned)((lv.ntiles + 3) / 4)), dim3(256), 0, pl->stream,
                                       lv.gTrsmPost.p, lv.tile_node.p, lv.tile_row0.p, lv.ntiles, pl->W.p, (long)pl->ldw,
                                       lv.c0, pl->var.p);
                }
            }
            { KTimer kt(pl, KF_PRED_UPDATE, lv.fl_update); launch_gemm<EPI_SUB>(pl, lv.gUpdate.p, nn, lv.max_rows, lv.na); }
        }
    }
    phase_mark(pl, 4);
    finish_run(pl);
}

static void finish_run(mra_plan* pl) {
    {
        KTimer kt(pl, KF_MISC, 0);
        const int nsum = pl->reduce_level >= 0 ? (int)pl->level_ptr[pl->reduce_level + 1] : pl->n_nodes;
        const double* up;
        if (pl->leaf[0]) up = pl->Gt.p + pl->leaf_goff[pl->leaf_slot[0]] + (size_t)(pl->na[0] - MRA_YB) * pl->na[0] + (pl->na[0] - MRA_YB);
        else {
            const LevelData& l0 = pl->lev[0];
            up = l0.F.p + (size_t)(l0.nf - MRA_YB) * l0.nf + (l0.nf - MRA_YB);
        }
        const double* below = nullptr;
        if (pl->reduce_level >= 0) {
            const LevelData& lr = pl->lev[pl->reduce_level];
            if (lr.F.n) below = lr.F.p + (lr.F.n - 16);
        }
        // the 32-byte record {d, u, log-det carried by the reduce level, error flag} is written by the kernel straight
        // into pinned host memory (no copy command, no gap after the last launch)
        if (!pl->host_res) {
            HIP_TRY(hipHostMalloc((void**)&pl->host_res, 4 * sizeof(double), hipHostMallocMapped));
            HIP_TRY(hipHostGetDevicePointer((void**)&pl->host_res_dev, pl->host_res, 0));
        }
        hipLaunchKernelGGL(k_sum_dnode, dim3(1), dim3(256), 0, pl->stream, pl->dnode.p, nsum, pl->host_res_dev, up, below, pl->errflag.p);
        const bool hi_path = !(pl->regular && pl->use_fused && !pl->host_cov) && pl->regular_hi && pl->use_fused && !pl->host_cov;
        if ((pl->run_flags & MRA_RUN_PREDICT) && !(pl->regular && pl->use_fused && !pl->host_cov) && !hi_path)
            hipLaunchKernelGGL(k_extract_mean, dim3((unsigned)((pl->P + 255) / 256)), dim3(256), 0, pl->stream,
                               pl->W.p, (long)pl->ldw, pl->Ka, pl->mean.p, pl->P);
    }
    phase_mark(pl, 5);
    HIP_TRY(hipStreamSynchronize(pl->stream));
    HIP_TRY(hipGetLastError());
    const int errv = (int)pl->host_res[3];
    const double dsum = pl->host_res[0] + pl->host_res[2], u = pl->host_res[1];
    pl->res_d = dsum; pl->res_u = u;
    float ms;
    for (int k = 0; k < 4; ++k) {
        static const int a[4] = {0, 1, 2, 3}, b[4] = {1, 2, 3, 4};
        ms = 0.f;
        if (pl->ktiming) hipEventElapsedTime(&ms, pl->ev[a[k]], pl->ev[b[k]]);
        pl->phase_ms[k] = ms;
    }
    hipEventElapsedTime(&ms, pl->ev[0], pl->ev[5]);
    pl->phase_ms[4] = ms;
    for (auto& e : pl->kev) {
        hipEventElapsedTime(&ms, e.second.first, e.second.second);
        pl->kstat[e.first].ms += ms;
        hipEventDestroy(e.second.first); hipEventDestroy(e.second.second);
    }
    pl->kev.clear();
    pl->ran = true;
    pl->split_pending = false;
    pl->pass_open = false;
    if (errv) pl->cphantom_valid = false;
    if (errv) {
        char b[160];
        snprintf(b, sizeof b, "matrix not positive definite in node %d (Cholesky pivot <= 0 or NaN)", errv - 1);
        throw MraError(MRA_ERR_NOT_SPD, b);
    }
}

// allocate the per-leaf Schur blocks the first time a run needs them and point the descriptors at them
static void ensure_gt(mra_plan* pl) {
    if (pl->Gt.p || pl->leaf_goff.back() == 0) return;
    pl->Gt.alloc(pl->leaf_goff.back());
    for (size_t k = 0; k < pl->hKids.size(); ++k)
        if (pl->kid_leaf[k] >= 0) pl->hKids[k].G = pl->Gt.p + pl->leaf_goff[pl->kid_leaf[k]];
    pl->asmKids.upload(pl->hKids);
    for (size_t t = 0; t < pl->hLeafSyrk.size(); ++t) pl->hLeafSyrk[t].C = pl->Gt.p + pl->leaf_goff[t];
    pl->gLeafSyrk.upload(pl->hLeafSyrk);
}

static void run_all(mra_plan* pl, uint32_t flags) {
    if (g_dry) throw MraError(MRA_ERR_STATE, "MRA_HOST_DRYRUN plan: built in host memory for the sanitizers, it cannot run");
    if (!(pl->have_locs && pl->have_obs && pl->have_kernel))
        throw MraError(MRA_ERR_STATE, "mra_run needs set_locs, set_obs and set_kernel first");
    HIP_TRY(mraSetDevice(pl->device));
    if (pl->pass_open) {
        // the previous pass never reached finish_run (an error was thrown, or a split run was abandoned before
        // mra_run_resume): wait for whatever it left on the two streams, and clear the device error flag that
        // only the last kernel of a pass resets
        pl->cphantom_valid = false;
        if (pl->side_pending) HIP_TRY(hipStreamWaitEvent(pl->stream, pl->ev_join, 0));
        HIP_TRY(hipStreamSynchronize(pl->stream));
        HIP_TRY(hipStreamSynchronize(pl->stream2));
        HIP_TRY(hipMemsetAsync(pl->errflag.p, 0, sizeof(int), pl->stream));
        for (auto& e : pl->kev) { hipEventDestroy(e.second.first); hipEventDestroy(e.second.second); }
        pl->kev.clear();
    }
    pl->side_pending = false;
    pl->split_pending = false;
    pl->pred_update_now = false;
    pl->hi_fold_now = false;
    pl->pass_open = true;
    pl->run_flags = flags;
    for (int k = 0; k < KF_COUNT; ++k) pl->kstat[k] = mra_plan::KStat();
    const bool pred = flags & MRA_RUN_PREDICT;
    phase_mark(pl, 0);
    if (!(pl->regular && pl->use_fused && !pl->host_cov)) {        // the fused prior cascade writes the y block itself
        KTimer kt(pl, KF_MISC, 0);
        const long n = pl->P * MRA_YB;
        // every level's row solve takes the LDS path (block width <= 192) and the leaves' solve subtracts |Tt|^2: the variance is
        // accumulated on the way instead of by a pass over all of W at the end
        bool acc_var = pl->leaf_max_nop / 16 <= 12;
        for (int m = 0; m < pl->n_levels; ++m) if (!pl->lev[m].nodes.empty() && pl->lev[m].cwt > 12) acc_var = false;
        pl->var_accumulated = acc_var;
        hipLaunchKernelGGL(k_init_yblock, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, pl->stream, pl->W.p,
                           (long)pl->ldw, pl->Ka, pl->y.p, pl->P, acc_var ? pl->var.p : (double*)nullptr, pl->host_cov ? 0.0 : kernel_cov0(pl),
                           pl->host_cov ? pl->covdiag.p : (const double*)nullptr);
    }
    // ---- 1. prior, top-down
    const bool fused = pl->regular && pl->use_fused && !pl->host_cov;
    if (fused) run_prior_fused(pl);
    // likelihood-only: the one-launch prior levels walk the rows a likelihood needs (observed rows and knots) when every level takes
    // that path and the leaf stage builds C from the gathered product (c_only below)
    const bool lik_general = !fused && !pred && pl->use_lik_rows && pl->use_prior_level && !pl->host_cov && pl->gemm_lds &&
                             pl->leaf_max_nop / 16 <= 12 && pl->leaf_max_nop > 0 && ensure_lik_general(pl);
    for (int m = 0; m < pl->n_levels && !fused; ++m) {
        LevelData& lv = pl->lev[m];
        const size_t nn = lv.nodes.size();
        if (!nn) continue;
        if (pl->use_prior_level && lv.prior_level_ok && !pl->host_cov && pl->gemm_lds) {
            // knots' residual block -> Lp (both sides gathered), its Cholesky, then residual + kernel + row solve of every row in one
            // launch: the residual never visits HBM
            { KTimer kt(pl, KF_PRIOR_CHOL, lv.fl_knot_resid); launch_gemm<EPI_COV>(pl, lv.gKnotResid.p, nn, lv.cw, lv.cw); }
            { KTimer kt(pl, KF_PRIOR_CHOL, lv.fl_pchol); launch_panel(pl, lv.gPriorChol.p, nn); }
            if (lik_general) {
                Work w = lv.fl_resid + lv.fl_trsm.with_bytes(0.0);
                w.alg *= lv.lik_share; w.exec *= lv.lik_share; w.bytes *= lv.lik_share;
                KTimer kt(pl, KF_PRIOR_RESID, w);
                mra_launch_prior_level(pl, lv.gResidLik.p, lv.gResidLik.n);
            } else {
                KTimer kt(pl, KF_PRIOR_RESID, lv.fl_resid + lv.fl_trsm.with_bytes(0.0));
                mra_launch_prior_level(pl, lv.gResidFused.p, lv.gResidFused.n);
            }
            continue;
        }
        {
            KTimer kt(pl, KF_PRIOR_RESID, lv.fl_resid);
            if (pl->host_cov) launch_gemm<EPI_HOSTCOV>(pl, lv.gResid.p, nn, lv.max_rows, lv.cw);
            else launch_gemm<EPI_COV>(pl, lv.gResid.p, nn, lv.max_rows, lv.cw);
        }
        {
            KTimer kt(pl, KF_MISC, 0);
            dim3 grid((unsigned)((lv.cw * lv.cw + 255) / 256), (unsigned)nn);
            hipLaunchKernelGGL(k_gather_kinv, grid, dim3(256), 0, pl->stream, lv.gKinv.p, pl->W.p, (long)pl->ldw, lv.c0);
        }
        { KTimer kt(pl, KF_PRIOR_CHOL, lv.fl_pchol); launch_panel(pl, lv.gPriorChol.p, nn); }
        {
            KTimer kt(pl, KF_PRIOR_TRSM, lv.fl_trsm);
            if (!launch_trsm2(pl, lv.gTrsm2Prior.p, nn, lv.cwt, lv.max_tiles, 32)) {
                ensure_tile_lists(pl, m);
                hipLaunchKernelGGL(k_trsm_rows, dim3((unsigned)((lv.ntiles + 3) / 4)), dim3(256), 0, pl->stream,
                                   lv.gTrsmPrior.p, lv.tile_node.p, lv.tile_row0.p, lv.ntiles, pl->W.p, (long)pl->ldw, lv.c0,
                                   (double*)nullptr);
            }
        }
    }
    phase_mark(pl, 1);
    // ---- 2. leaves
    const size_t nl = pl->leaf_nodes.size();
    if (nl) {
        // likelihood-only passes never need V[S,o]: just C = v_m(o,o) + R I from a small gathered product (the row solve gathers Ut itself
        // on the level-by-level path, the row cascade has scattered it on the fused one)
        const bool c_only = !pred && !pl->host_cov && pl->gemm_lds && pl->leaf_max_nop / 16 <= 12 && pl->leaf_max_nop > 0;
        {
            KTimer kt(pl, KF_LEAF_RESID, c_only ? pl->fl_leaf_c_only : pl->fl_leaf_resid);
            if (pl->host_cov) {
                if (leaf_gemm_ok(pl)) launch_leaf_gemm<EPI_HOSTCOV>(pl, pl->gLeafResid.p, nl);
                else launch_gemm<EPI_HOSTCOV>(pl, pl->gLeafResid.p, nl, pl->leaf_max_rows, pl->leaf_max_nop);
            }
            else if (c_only) launch_gemm<EPI_COV>(pl, pl->gLeafResidLik.p, nl, pl->leaf_max_nop, pl->leaf_max_nop);
            else if (leaf_gemm_ok(pl)) launch_leaf_gemm<EPI_COV>(pl, pl->gLeafResid.p, nl);
            else launch_gemm<EPI_COV>(pl, pl->gLeafResid.p, nl, pl->leaf_max_rows, pl->leaf_max_nop);
        }
        if (pl->leaf_max_nop > 0) {
            KTimer kt(pl, KF_MISC, 0);
            if (c_only) {
                // nothing to do: the gathered COV product wrote the whole C block including phantom identities
            } else if (pl->leaf_max_nop / 16 <= 12) {
                // C comes from the COV epilogue, Ut from the gather inside k_trsm_rows2: only the phantom rows remain
                // ... once: an identity row stays an identity row under the in-place factorisation (L[p][j] = 0, L[p][p] = 1 exactly) and
                // no epilogue writes there, so later passes find them in place.  (A failed pass - NaN times 0 - a new observation pattern
                // or a change of options bring the launch back.)
                if (!pl->cphantom_valid) {
                    hipLaunchKernelGGL(k_leaf_cphantom, dim3((unsigned)nl), dim3(256), 0, pl->stream, pl->gLeaf.p, pl->leaf_nobs.p);
                    pl->cphantom_valid = true;
                }
            } else {
                const long total = (long)(pl->leaf_max_nop + pl->leaf_max_na) * pl->leaf_max_nop;
                dim3 grid((unsigned)std::min<long>((total + 255) / 256, 64), (unsigned)nl);
                hipLaunchKernelGGL(k_leaf_fill, grid, dim3(256), 0, pl->stream, pl->gLeaf.p, pl->W.p, (long)pl->ldw, pl->R);
            }
        }
        bool solve_fused = false;
        {
            KTimer kt(pl, KF_LEAF_CHOL, pred ? pl->fl_leaf_chol : pl->fl_leaf_chol_lik);
            const int ntl = pl->leaf_max_nop / 16;
            if (ntl <= 12) {
                // (round 4, with the blocked factorisation atom: one workgroup per matrix is at least as fast as one wave per matrix at
                //  every shard size - 5.507 / 3.00 / 1.601 / 0.922 ms against 5.528 / 3.00 / 1.621 / 0.959 for 1 / 2 / 4 / 8-way C3 -
                //  so option 11 = 1 (the default) takes it for matrices of five tiles and more at any count; small matrices - config 5:
                //  65536 of at most three tiles - stay on one wave each unless a CU sees at most two of them; 0 forces k_chol_wave)
                if (ntl <= 10 && (pl->use_chol_lds == 2 || (pl->use_chol_lds == 1 && (ntl >= 5 || nl <= (size_t)(2 * pl->n_cu))))) {
                    // one workgroup per matrix, tiles in registers, next diagonal block factorised beside the trailing update
                    const size_t ns = pl->n_chol_small;
                    if (ns == nl || nl > (size_t)(2 * pl->n_cu)) {
                        if (ns) hipLaunchKernelGGL((k_chol_tiles<8, 4>), dim3((unsigned)ns), dim3(256), 0, pl->stream, pl->gLeafCholSorted.p, pl->dnode.p, pl->errflag.p);
                        if (nl > ns) hipLaunchKernelGGL((k_chol_tiles<10, 4>), dim3((unsigned)(nl - ns)), dim3(256), 0, pl->stream, pl->gLeafCholSorted.p + ns, pl->dnode.p, pl->errflag.p);
                    }
                    else hipLaunchKernelGGL((k_chol_tiles<10, 4>), dim3((unsigned)nl), dim3(256), 0, pl->stream, pl->gLeafCholSorted.p, pl->dnode.p, pl->errflag.p);
                } else
                hipLaunchKernelGGL((k_chol_wave<12>), dim3((unsigned)((nl + 3) / 4)), dim3(256), 0, pl->stream, pl->gLeafCholC.p, (int)nl, pl->dnode.p, pl->errflag.p);
                const int mt = pred ? pl->leaf_max_tiles_full : pl->leaf_max_tiles_lik;
                if (fused) {
                    // with the fused row solve + update the small leaves only need their Ut rows solved here
                    // (one 8-wave workgroup per CU: a gain when a CU sees at most two leaves - 1.21 -> 1.13 ms on an eighth of C3 -
                    // and a loss from four per CU on - 1.87 -> 1.90 ms on a quarter - where the update rides in the predictive
                    // cascade at three workgroups per CU)
                    solve_fused = pred && pl->use_leaf_solve && pl->leaf_solve_ok &&
                                  (pl->leaf_solve_mode == 1 || pl->n_trsm_small <= (size_t)(2 * pl->n_cu));
                    const Trsm2Prob* likp = pl->ut_gather ? pl->gLeafTrsmLikPlainG.p : pl->gLeafTrsmLikPlain.p;
                    const Trsm2Prob* base = pred ? (pl->ut_gather ? pl->gLeafTrsmFullPlainG.p : pl->gLeafTrsmFullPlain.p) : likp;
                    const size_t ns = pl->n_trsm_small;
                    const int mts = (pred && !solve_fused) ? pl->trsm_small_tiles_full : pl->trsm_small_tiles_lik;
                    if (ns) launch_trsm2(pl, solve_fused ? likp : base, ns, pl->trsm_small_nt, mts, mts);
                    // the few leaves with more than 128 observations: several workgroups per leaf when they are few
                    // (one leaf per workgroup would put a single 65 us workgroup on the critical path)
                    if (nl > ns) launch_trsm2(pl, base + ns, nl - ns, ntl, mt, (nl - ns) < 512 ? 4 : mt);
                }
                else launch_trsm2(pl, pred ? pl->gLeafTrsmFull.p : pl->gLeafTrsmLik.p, nl, ntl, mt, mt);
            } else {
                // right-looking, 64 columns per step: the panel (factor + solve of ALL rows below, Ut and Tt included) on one
                // workgroup per leaf, the rank-64 update of everything to its right as a batched GEMM over the whole GPU
                // (a single workgroup factorising an 8560 x 8560 block column by column took 15 s: README example 1)
                const int v = pred ? 0 : 1;
                for (size_t st = 0; st < pl->gBigPanel[v].size(); ++st) {
                    launch_panel(pl, pl->gBigPanel[v][st].p, nl, st > 0 ? 1 : 0);
                    if (pl->bigM[v][st] > 0 && pl->bigN[v][st] > 0)
                        launch_gemm<EPI_SUB>(pl, pl->gBigTrail[v][st].p, nl, pl->bigM[v][st], pl->bigN[v][st]);
                }
            }
        }
        pl->direct_parent = pl->parent_syrk && pl->reduce_level != pl->NL - 1;
        if (!pl->direct_parent) ensure_gt(pl);
        if (!pl->direct_parent) { KTimer kt(pl, KF_LEAF_SYRK, pl->fl_leaf_syrk); launch_gemm<EPI_SET>(pl, pl->gLeafSyrk.p, nl, pl->leaf_max_na, pl->leaf_max_na, false, true); }
        if (pred) {
            // fork: everything below only feeds the predictive pass.  In a sharded run it goes to the side stream, so that
            // the front chain and the all-reduce do not queue behind it; with per-kernel timing on it stays on the main
            // stream so that the hipEvent brackets measure one kernel at a time.
            const bool side = !pl->ktiming && pl->reduce_level >= 0;   // sharded runs only: on one GPU it buys nothing
            hipStream_t main_stream = pl->stream;
            struct StreamGuard {                                   // a throw below must not leave later launches on the side stream
                mra_plan* p; hipStream_t s;
                ~StreamGuard() { p->stream = s; }
            } guard{pl, main_stream};
            if (side) {
                HIP_TRY(hipEventRecord(pl->ev_fork, main_stream));
                HIP_TRY(hipStreamWaitEvent(pl->stream2, pl->ev_fork, 0));
                pl->stream = pl->stream2;                    // the launch helpers below use pl->stream
            }
            if (!fused && pl->var_accumulated) {
                KTimer kt(pl, KF_MISC, 0);
                ensure_row_leaf(pl);
                hipLaunchKernelGGL(k_leaf_finish_var, dim3((unsigned)((pl->P + 255) / 256)), dim3(256), 0, pl->stream, pl->row_leaf.p, pl->W.p,
                                   (long)pl->ldw, pl->Ka, pl->var.p, pl->P);
            } else if (!fused || pl->leaf_max_nop / 16 > 12) {
                KTimer kt(pl, KF_MISC, 0);
                double cov0 = 0.0;
                if (!pl->host_cov) {
                    // stationary kernel: C(x,x) = scale * k(0)
                    KernelParams kp = pl->kp;
                    switch (kp.kind) {
                        case 0: cov0 = kp.scale * 1.0; break;
                        case 1: case 2: case 3: cov0 = kp.scale * kp.sig; break;
                        default: cov0 = kp.scale; break;
                    }
                }
                ensure_row_leaf(pl);
                hipLaunchKernelGGL(k_leaf_moments, dim3((unsigned)((pl->P + 3) / 4)), dim3(256), 0, pl->stream, pl->gLeaf.p,
                                   pl->row_leaf.p, pl->W.p, (long)pl->ldw, pl->Ka, pl->var.p, cov0,
                                   pl->host_cov ? pl->covdiag.p : (const double*)nullptr, pl->P);
            }
            pl->pred_update_now = !solve_fused && fused && pl->use_pred_update && pl->leaf_solve_ok && pl->leaf_max_nop / 16 <= 12 &&
                                  pl->na[pl->NL] == (pl->NL * pl->CWT + 1) * 16;
            // deep 64-wide trees: the update rides in k_predict_hi (leaves of at most four observation tiles, 16 x 16-padded ancestors + y)
            // (a sharded rank keeps the separate product: it runs on the side stream beside the front chain and the all-reduce, whereas
            // k_predict_hi sits behind them on the rank's critical path; option 16 = 2 folds there too, for A/B runs)
            pl->hi_fold_now = !solve_fused && !fused && pl->use_hi_fold && (pl->reduce_level < 0 || pl->use_hi_fold == 2) && pl->regular_hi && pl->use_fused &&
                              !pl->host_cov && pl->leaf_max_nop <= 64 && pl->na[pl->NL] == (pl->NL * 4 + 1) * 16;
            if (pl->hi_fold_now) {
                // nothing to launch here
            } else if (pl->pred_update_now) {
                // the small leaves (<= 8 observation tiles) take their update inside the predictive cascade; the few larger ones here
                const size_t ns = pl->n_trsm_small;
                if (nl > ns) {
                    KTimer kt(pl, KF_LEAF_UPDATE, 0);
                    launch_gemm<EPI_SUB>(pl, pl->gLeafUpdatePlain.p + ns, nl - ns, pl->leaf_max_rows, pl->leaf_max_na);
                }
            } else if (solve_fused) {
                // Tt = V Lc^-T, var -= |Tt|^2 and W -= Tt Ut^T in one launch for the leaves with <= 8 observation tiles; the few
                // larger ones went through the full row solve above and take the plain update product
                ensure_big_lds(pl, {(const void*)k_leaf_solve_update<8, 13, true>});
                KTimer kt(pl, KF_LEAF_UPDATE, pl->fl_leaf_update);
                const size_t ns = pl->n_trsm_small;
#ifdef MRA_STAMPS
                if (pl->kstamps2.n < pl->n_leaf_solve_half * 8) { pl->kstamps2.alloc(pl->n_leaf_solve_half * 8); HIP_TRY(mraMemset(pl->kstamps2.p, 0, pl->kstamps2.n * sizeof(double))); HIP_TRY(hipDeviceSynchronize()); }
#define MRA_LSTAMP_VAL , (unsigned long long*)pl->kstamps2.p
#define MRA_LSTAMP_NUL , (unsigned long long*)nullptr
#else
#define MRA_LSTAMP_VAL
#define MRA_LSTAMP_NUL
#endif
                if (pl->leaf_solve_split == 2 && pl->n_leaf_solve_half)
                    hipLaunchKernelGGL((k_leaf_solve_update<8, 13, true>), dim3((unsigned)pl->n_leaf_solve_half), dim3(512), pl->leaf_solve_lds, pl->stream, pl->gLeafSolveHalf.p MRA_LSTAMP_VAL);
                else
                    hipLaunchKernelGGL((k_leaf_solve_update<8, 13, true>), dim3((unsigned)ns), dim3(512), pl->leaf_solve_lds, pl->stream, pl->gLeafSolve.p MRA_LSTAMP_NUL);
                if (nl > ns) launch_gemm<EPI_SUB>(pl, pl->gLeafUpdatePlain.p + ns, nl - ns, pl->leaf_max_rows, pl->leaf_max_na);
            } else {
                KTimer kt(pl, KF_LEAF_UPDATE, pl->fl_leaf_update);
                // (measured: the leaf-resident form wins for the residual - 1.07 vs 1.15 ms at C3 - but not for the update,
                // 1.34-1.39 vs 1.29 ms, whatever the pass structure; MRA_OPT_LEAF_GEMM = 2 selects it for A/B runs)
                if (leaf_gemm_ok(pl) && pl->leaf_gemm_update) launch_leaf_gemm<EPI_SUB>(pl, pl->gLeafUpdate.p, nl);
                else launch_gemm<EPI_SUB>(pl, pl->gLeafUpdate.p, nl, pl->leaf_max_rows, pl->leaf_max_na);
            }
            if (side) {
                pl->stream = main_stream;
                HIP_TRY(hipEventRecord(pl->ev_join, pl->stream2));
                pl->side_pending = true;
            }
        }
    }
    phase_mark(pl, 2);
    // ---- 3./4. fronts bottom-up, then predictive moments
    run_fronts_and_predict(pl, pl->n_levels - 1, false);
}

// ---- caller-order variants: the permutation work of an end-to-end MRATree(...) call done inside the library -----------------
// A process-wide pinned staging area (grow-only): gathers land in it, the H2D / D2H copies run at the pinned rate (a pageable
// 16 MB copy costs ~5 ms, a pinned one ~0.7 ms), and a second plan in the same process does not pay for the allocation again.
static std::mutex g_stage_mutex;
static double* g_stage = nullptr;
static size_t g_stage_n = 0;
static double* stage_buffer(size_t n) {               // caller holds g_stage_mutex
    if (n > g_stage_n) {
        if (g_stage) { if (g_dry) free(g_stage); else hipHostFree(g_stage); g_stage = nullptr; g_stage_n = 0; }
        if (g_dry) g_stage = (double*)malloc(n * sizeof(double));
        else if (hipHostMalloc((void**)&g_stage, n * sizeof(double), hipHostMallocDefault) != hipSuccess) g_stage = nullptr;
        if (!g_stage) throw MraError(MRA_ERR_HIP, "pinned staging allocation failed");
        g_stage_n = n;
    }
    return g_stage;
}


// knot coordinates of the fused levels (locs: P x d in padded leaf order): per level for the cascades, and packed per workgroup
// for the knot chain
static void set_knot_coords_src(mra_plan* pl, const double* locs, const int64_t* src);
static void set_knot_coords(mra_plan* pl, const double* locs) { set_knot_coords_src(pl, locs, nullptr); }
// (src: locs is the CALLER's N x d array and padded row p holds caller row src[p]; nullptr: locs is already in padded order)
static void set_knot_coords_src(mra_plan* pl, const double* locs, const int64_t* src) {
    if (!pl->regular) return;
    const int cw = pl->cw[0];
    std::vector<std::vector<double>> kxh(pl->kc_levels);
    for (int m = 0; m < pl->NL; ++m) {
        const LevelData& lv = pl->lev[m];
        std::vector<double> kx(lv.nodes.size() * (size_t)cw * pl->d);
        for (size_t e = 0; e < kx.size(); ++e)           // phantom knots: far away and far from each other
            kx[e] = MRA_FAR_AWAY * (double)(2 + (e / pl->d) % cw);
        for (size_t sl = 0; sl < lv.nodes.size(); ++sl) {
            const int i = lv.nodes[sl];
            const long rk = pl->knot_ptr[i + 1] - pl->knot_ptr[i];
            for (long c = 0; c < rk; ++c)
                for (int k = 0; k < pl->d; ++k)
                    kx[(sl * cw + c) * pl->d + k] = locs[(src ? src[pl->knot_rows[pl->knot_ptr[i] + c]] : pl->knot_rows[pl->knot_ptr[i] + c]) * pl->d + k];
        }
        HIP_TRY(mraMemcpy(pl->fl[m].kx.p, kx.data(), kx.size() * sizeof(double), hipMemcpyHostToDevice));
        if (m < pl->kc_levels) kxh[m].swap(kx);
    }
    // k_knot_chain: every workgroup's knots of all its levels in one record (coordinates, then 1.0 / 0.0 = real / phantom)
    if (pl->kc_levels >= 2 && pl->kc_knots.n) {
        const int nlv = pl->kc_levels, d = pl->d;
        const size_t rec = (size_t)cw * (d + 1), nb = pl->lev[nlv - 1].nodes.size();
        std::vector<double> pk(nb * nlv * rec);
        for (size_t b = 0; b < nb; ++b)
            for (int m = 0; m < nlv; ++m) {
                const int sl = pl->kc_chain_host[b * 8 + m];
                const int i = pl->lev[m].nodes[sl];
                const long rk = pl->knot_ptr[i + 1] - pl->knot_ptr[i];
                double* o = pk.data() + (b * nlv + m) * rec;
                std::memcpy(o, kxh[m].data() + (size_t)sl * cw * d, (size_t)cw * d * sizeof(double));
                for (int c = 0; c < cw; ++c) o[(size_t)cw * d + c] = c < rk ? 1.0 : 0.0;
            }
        HIP_TRY(mraMemcpy(pl->kc_knots.p, pk.data(), pk.size() * sizeof(double), hipMemcpyHostToDevice));
    }
}

// ------------------------------------------------------------------------------------------------
//  C ABI
// ------------------------------------------------------------------------------------------------
extern "C" {

const char* mra_version(void) { return MRA_VERSION_STR; }

const char* mra_last_error(mra_plan* plan) { return plan ? plan->err.c_str() : g_last_error.c_str(); }

int mra_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int mra_plan_create(mra_plan** out, const mra_topology* t, int device) {
    if (!out || !t) return MRA_ERR_INVALID;
    *out = nullptr;
    mra_plan* pl = nullptr;
    try {
        if (t->P <= 0 || (t->d != 1 && t->d != 2) || t->n_levels <= 0 || t->n_nodes <= 0)
            throw MraError(MRA_ERR_INVALID, "bad topology header");
        int ndev = 0;
        if (!g_dry) {
            if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
                throw MraError(MRA_ERR_HIP, "no HIP device available: libmra_hip needs an AMD GPU (gfx950); there is no CPU fallback");
            if (device < 0 || device >= ndev) throw MraError(MRA_ERR_INVALID, "device ordinal out of range");
        }
        PlanTrace tr("mra_plan_create");
        HIP_TRY(mraSetDevice(device));
        pl = new mra_plan();
        pl->device = device;
        pl->P = t->P; pl->d = t->d; pl->n_levels = t->n_levels; pl->n_nodes = t->n_nodes;
        pl->level_ptr.assign(t->level_ptr, t->level_ptr + t->n_levels + 1);
        pl->row0.assign(t->node_row0, t->node_row0 + t->n_nodes);
        pl->row1.assign(t->node_row1, t->node_row1 + t->n_nodes);
        pl->leaf.assign(t->node_leaf, t->node_leaf + t->n_nodes);
        pl->parent.assign(t->node_parent, t->node_parent + t->n_nodes);
        pl->child_ptr.assign(t->child_ptr, t->child_ptr + t->n_nodes + 1);
        pl->child_list.assign(t->child_list, t->child_list + pl->child_ptr.back());
        pl->knot_ptr.assign(t->knot_ptr, t->knot_ptr + t->n_nodes + 1);
        pl->knot_rows.assign(t->knot_rows, t->knot_rows + pl->knot_ptr.back());
        pl->cw.assign(t->cw, t->cw + t->n_levels);
        if (pl->level_ptr[0] != 0 || pl->level_ptr.back() != t->n_nodes) throw MraError(MRA_ERR_INVALID, "level_ptr inconsistent");
        tr.mark("copies of the topology");
        if (!g_dry) {
            // The leaf update (one large GEMM) and [parent SYRK -> front Cholesky/Schur chain -> all-reduce of a sharded run]
            // only meet again in the predictive cascade: they are issued on two streams, so the collective and the
            // latency-bound chain never wait behind the update.  (On one GPU the update keeps every SIMD's register file
            // full and the pass time does not change; the point is the sharded run.)
            acquire_streams(pl);
            HIP_TRY(hipEventCreateWithFlags(&pl->ev_fork, hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&pl->ev_join, hipEventDisableTiming));
        }
        for (int k = 0; k < 6 && !g_dry; ++k) HIP_TRY(hipEventCreate(&pl->ev[k]));
        tr.mark("streams, events");
        init_arena(pl);
        build_static(pl);
        *out = pl;
        return MRA_OK;
    } catch (const MraError& e) {
        int rc = fail(nullptr, e);
        if (pl) drop_arena(pl);
        if (pl && !g_dry) return_streams(pl);
        delete pl;
        return rc;
    } catch (const std::exception& e) {
        g_last_error = e.what();
        if (pl) drop_arena(pl);
        if (pl && !g_dry) return_streams(pl);
        delete pl;
        return MRA_ERR_INVALID;
    }
}

int mra_plan_destroy(mra_plan* pl) {
    if (!pl) return MRA_OK;
    mraSetDevice(pl->device);
    if (pl->comm && pl->rccl) {
        typedef ncclResult_t (*destroy_t)(ncclComm_t);
        destroy_t fn = (destroy_t)dlsym(pl->rccl, "ncclCommDestroy");
        if (fn) fn(pl->comm);
    }
    // the plan's device blocks go back to the cache and may be handed to another plan at once: nothing of this plan may still run
    if (pl->stream) hipStreamSynchronize(pl->stream);
    if (pl->stream2) hipStreamSynchronize(pl->stream2);
    for (int k = 0; k < 6; ++k) if (pl->ev[k]) hipEventDestroy(pl->ev[k]);
    drop_arena(pl);
    return_streams(pl);                              // (with the pinned result record and the arena's pinned mirror)
    if (pl->ev_fork) hipEventDestroy(pl->ev_fork);
    if (pl->ev_join) hipEventDestroy(pl->ev_join);
    delete pl;
    return MRA_OK;
}

int mra_release_cached_memory(void) {
    mra_topo::release_scratch();                      // the tree replay's work arrays (~100 MB at 1024^2, ~0.5 GB at config 5)
    if (g_dry) return MRA_OK;
    {
        std::lock_guard<std::mutex> lock(g_streams_mu);
        for (auto& e : g_streams) { hipStreamDestroy(e.second.hi); hipStreamDestroy(e.second.lo); if (e.second.host_res) hipHostFree(e.second.host_res); if (e.second.arena_host) hipHostFree(e.second.arena_host); }
        g_streams.clear();
    }
    {
        std::lock_guard<std::mutex> lock(g_stage_mutex);
        if (g_stage) { hipHostFree(g_stage); g_stage = nullptr; g_stage_n = 0; }
    }
    std::lock_guard<std::mutex> lock(g_pool.mu);
    g_pool.flush_locked();
    return MRA_OK;
}

int mra_plan_set_locs(mra_plan* pl, const double* locs) {
    if (!pl || !locs) return MRA_ERR_INVALID;
    try {
        HIP_TRY(mraSetDevice(pl->device));
        HIP_TRY(mraMemcpy(pl->X.p, locs, (size_t)pl->P * pl->d * sizeof(double), hipMemcpyHostToDevice));
        if (!pl->knots_pending) set_knot_coords(pl, locs);
        pl->have_locs = true;
        return MRA_OK;
    } catch (const MraError& e) { return fail(pl, e); }
}

// ---- caller-order variants (helpers above the extern "C" block)
int mra_plan_set_locs_rows(mra_plan* pl, const double* locs, const int64_t* src) {
    if (!pl || !locs || !src) return MRA_ERR_INVALID;
    try {
        PlanTrace tr("set_locs_rows");
        std::lock_guard<std::mutex> lock(g_stage_mutex);
        const int d = pl->d;
        double* xp = stage_buffer((size_t)pl->P * d);
        tr.mark("staging buffer");
        parallel_rows(pl->P, [&](int64_t a, int64_t b) {
            if (d == 2) for (int64_t p = a; p < b; ++p) { const int64_t q = src[p]; xp[2 * p] = locs[2 * q]; xp[2 * p + 1] = locs[2 * q + 1]; }
            else for (int64_t p = a; p < b; ++p) xp[p] = locs[src[p]];
        });
        tr.mark("gather into leaf order");
        const int rc = mra_plan_set_locs(pl, xp);
        tr.mark("upload X, knot coordinates");
        return rc;
    } catch (const MraError& e) { return fail(pl, e); }
      catch (const std::exception& e) { return fail(pl, MraError(MRA_ERR_INVALID, e.what())); }       // bad_alloc, thread creation
}

int mra_plan_set_obs_rows(mra_plan* pl, const double* y, const int64_t* src, const int64_t* perm, double R) {
    if (!pl || !y || !src || !perm) return MRA_ERR_INVALID;
    try {
        PlanTrace tr("set_obs_rows");
        std::lock_guard<std::mutex> lock(g_stage_mutex);
        double* yp = stage_buffer((size_t)pl->P);
        const double nan = std::nan("");
        parallel_rows(pl->P, [&](int64_t a, int64_t b) { for (int64_t p = a; p < b; ++p) yp[p] = perm[p] < 0 ? nan : y[src[p]]; });
        tr.mark("gather into leaf order");
        const int rc = mra_plan_set_obs(pl, yp, R);
        tr.mark("upload y, build_leaf");
        return rc;
    } catch (const MraError& e) { return fail(pl, e); }
      catch (const std::exception& e) { return fail(pl, MraError(MRA_ERR_INVALID, e.what())); }       // bad_alloc, thread creation
}

int mra_get_predict_rows(mra_plan* pl, const int64_t* perm, const uint8_t* in_leaf, int64_t N, double* mean, double* var) {
    return mra_get_predict_rows_sd(pl, perm, in_leaf, N, mean, var, nullptr);
}

int mra_get_predict_rows_sd(mra_plan* pl, const int64_t* perm, const uint8_t* in_leaf, int64_t N, double* mean, double* var, double* sd) {
    if (!pl || !perm || !in_leaf || !mean || !var || N <= 0) return MRA_ERR_INVALID;
    try {
        PlanTrace tr("get_predict_rows");
        std::lock_guard<std::mutex> lock(g_stage_mutex);
        double* st = stage_buffer((size_t)pl->P * 2);
        tr.mark("staging buffer");
        const int rc = mra_get_predict(pl, st, st + pl->P);
        if (rc != MRA_OK) return rc;
        tr.mark("D2H mean, var");
        const double* mp = st; const double* vp = st + pl->P;
        // rows that no leaf reports (dropped by a partition, or another rank's) read 0 - written only when there are such rows
        std::atomic<int64_t> covered{0};
        parallel_rows(pl->P, [&](int64_t a, int64_t b) {
            int64_t c = 0;
            for (int64_t p = a; p < b; ++p) c += (in_leaf[p] && perm[p] >= 0 && perm[p] < N) ? 1 : 0;
            covered += c;
        });
        if (covered.load() != N) {
            parallel_rows(N, [&](int64_t a, int64_t b) { for (int64_t i = a; i < b; ++i) { mean[i] = 0.0; var[i] = 0.0; if (sd) sd[i] = 0.0; } });
            tr.mark("zero the caller's arrays");
        }
        // every caller row sits in at most one padded row, so the scatter has no write conflicts between threads
        parallel_rows(pl->P, [&](int64_t a, int64_t b) {
            for (int64_t p = a; p < b; ++p) if (in_leaf[p]) {
                const int64_t i = perm[p];
                if (i >= 0 && i < N) { mean[i] = mp[p]; var[i] = vp[p]; if (sd) sd[i] = std::sqrt(vp[p]); }
            }
        });
        return MRA_OK;
    } catch (const MraError& e) { return fail(pl, e); }
      catch (const std::exception& e) { return fail(pl, MraError(MRA_ERR_INVALID, e.what())); }       // bad_alloc, thread creation
}

int mra_plan_set_obs(mra_plan* pl, const double* y, double R) {
    if (!pl || !y) return MRA_ERR_INVALID;
    try {
        if (!(R > 0.0)) throw MraError(MRA_ERR_INVALID, "R must be a positive scalar");
        HIP_TRY(mraSetDevice(pl->device));
        HIP_TRY(mraMemcpy(pl->y.p, y, (size_t)pl->P * sizeof(double), hipMemcpyHostToDevice));
        pl->R = R;
        if (pl->host_cov) {
            // host-evaluated covariance blocks are per observed row: a new observation pattern invalidates them (and
            // build_leaf rebuilds the leaf descriptors without their Csrc pointers).  The caller has to select
            // MRA_KERNEL_HOST and upload the blocks again; until then mra_run reports MRA_ERR_STATE.
            pl->host_cov = false;
            pl->have_kernel = false;
            pl->covsrc.release();
            pl->covdiag.release();
            for (auto& lv : pl->lev)
                if (!lv.nodes.empty()) {
                    for (auto& g : lv.hResid) { g.Csrc = nullptr; g.ldcs = 0; }
                    lv.gResid.upload(lv.hResid);
                }
        }
        build_leaf(pl, y);
        pl->have_obs = true;
        return MRA_OK;
    } catch (const MraError& e) { return fail(pl, e); }
}

int mra_plan_set_kernel(mra_plan* pl, int kind, const double* params, int n) {
    if (!pl) return MRA_ERR_INVALID;
    try {
        if (kind == MRA_KERNEL_HOST) {
            if (!pl->have_obs) throw MraError(MRA_ERR_STATE, "MRA_KERNEL_HOST needs mra_plan_set_obs first (leaf blocks are per observed row)");
            HIP_TRY(mraSetDevice(pl->device));
            // one padded block per node: non-leaf N_j x cw, leaf N_j x nop; leaves also C(x,x) per row
            pl->cov_off.assign(pl->n_nodes + 1, 0);
            for (int i = 0; i < pl->n_nodes; ++i) {
                const long nr = pl->row1[i] - pl->row0[i];
                const long ld = pl->leaf[i] ? pl->leaf_nop[pl->leaf_slot[i]] : pl->cw[pl->node_level[i]];
                pl->cov_off[i + 1] = pl->cov_off[i] + nr * ld;
            }
            pl->covsrc.alloc(std::max<long>(pl->cov_off.back(), 1));
            pl->covdiag.alloc(pl->P);
            HIP_TRY(mraMemset(pl->covsrc.p, 0, pl->covsrc.n * sizeof(double)));
            HIP_TRY(mraMemset(pl->covdiag.p, 0, pl->covdiag.n * sizeof(double)));
            for (int m = 0; m < pl->n_levels; ++m) {
                LevelData& lv = pl->lev[m];
                for (size_t sl = 0; sl < lv.nodes.size(); ++sl) {
                    lv.hResid[sl].Csrc = pl->covsrc.p + pl->cov_off[lv.nodes[sl]];
                    lv.hResid[sl].ldcs = lv.cw;
                }
                if (!lv.nodes.empty()) lv.gResid.upload(lv.hResid);
            }
            for (size_t t = 0; t < pl->leaf_nodes.size(); ++t) {
                pl->hLeafResid[t].Csrc = pl->covsrc.p + pl->cov_off[pl->leaf_nodes[t]];
                pl->hLeafResid[t].ldcs = pl->leaf_nop[t];
            }
            if (!pl->leaf_nodes.empty()) pl->gLeafResid.upload(pl->hLeafResid);
            pl->kp = KernelParams{};
            pl->host_cov = true;
            pl->have_kernel = true;
            return MRA_OK;
        }
        if (kind < 0 || kind > MRA_KERNEL_KANTER || !params || n < 3) throw MraError(MRA_ERR_INVALID, "unknown kernel kind or too few parameters (need l, sig, scale)");
        if (!(params[0] > 0.0)) throw MraError(MRA_ERR_INVALID, "length scale must be positive");
        pl->kp.kind = kind; pl->kp.d = pl->d; pl->kp.l = params[0]; pl->kp.sig = params[1]; pl->kp.scale = params[2];
        pl->kp.circular = (n >= 4 && params[3] != 0.0) ? 1 : 0;
        if (pl->kp.circular && pl->d != 1) throw MraError(MRA_ERR_INVALID, "circular distances are defined for 1-D locations only");
        derive_kernel_params(pl->kp);
        pl->host_cov = false;
        pl->have_kernel = true;
        return MRA_OK;
    } catch (const MraError& e) { return fail(pl, e); }
}

int mra_plan_set_cov_block(mra_plan* pl, int32_t node, const double* C, int64_t n_rows, int64_t n_cols, const double* diag) {
    if (!pl || !C) return MRA_ERR_INVALID;
    try {
        if (!pl->host_cov) throw MraError(MRA_ERR_STATE, "select MRA_KERNEL_HOST with mra_plan_set_kernel first");
        if (node < 0 || node >= pl->n_nodes) throw MraError(MRA_ERR_INVALID, "node out of range");
        HIP_TRY(mraSetDevice(pl->device));
        const long nr = pl->row1[node] - pl->row0[node];
        long ld, want_cols;
        if (pl->leaf[node]) {
            const int t = pl->leaf_slot[node];
            ld = pl->leaf_nop[t]; want_cols = pl->leaf_nobs_host[t];
            if (!diag) throw MraError(MRA_ERR_INVALID, "leaf blocks need diag (C(x,x) per row)");
        } else {
            ld = pl->cw[pl->node_level[node]]; want_cols = pl->knot_ptr[node + 1] - pl->knot_ptr[node];
        }
        if (n_rows != nr || n_cols != want_cols) {
            char b[200];
            snprintf(b, sizeof b, "block of node %d must be %ld x %ld (got %ld x %ld)", node, nr, want_cols, (long)n_rows, (long)n_cols);
            throw MraError(MRA_ERR_INVALID, b);
        }
        if (n_cols > 0)
            HIP_TRY(mraMemcpy2D(pl->covsrc.p + pl->cov_off[node], ld * sizeof(double), C, n_cols * sizeof(double),
                                n_cols * sizeof(double), nr, hipMemcpyHostToDevice));
        if (diag) HIP_TRY(mraMemcpy(pl->covdiag.p + pl->row0[node], diag, nr * sizeof(double), hipMemcpyHostToDevice));
        return MRA_OK;
    } catch (const MraError& e) { return fail(pl, e); }
}

int mra_run(mra_plan* pl, uint32_t flags) {
    if (!pl) return MRA_ERR_INVALID;
    try {
        if (!(flags & (MRA_RUN_LIKELIHOOD | MRA_RUN_PREDICT))) throw MraError(MRA_ERR_INVALID, "flags must request likelihood and/or predict");
        run_all(pl, flags);
        return MRA_OK;
    } catch (const MraError& e) { return fail(pl, e); }
}

int mra_run_resume(mra_plan* pl) {
    if (!pl) return MRA_ERR_INVALID;
    try {
        if (!pl->split_pending) throw MraError(MRA_ERR_STATE, "no split run pending");
        HIP_TRY(mraSetDevice(pl->device));
        run_fronts_and_predict(pl, pl->reduce_level, true);
        return MRA_OK;
    } catch (const MraError& e) { return fail(pl, e); }
}

int mra_get_likelihood(mra_plan* pl, double* d, double* u) {
    if (!pl || !d || !u) return MRA_ERR_INVALID;
    if (!pl->ran) return fail(pl, MraError(MRA_ERR_STATE, "mra_run has not completed"));
    *d = pl->res_d; *u = pl->res_u;
    return MRA_OK;
}

int mra_get_predict(mra_plan* pl, double* mean, double* var) {
    if (!pl || !mean || !var) return MRA_ERR_INVALID;
    try {
        if (!pl->ran || !(pl->run_flags & MRA_RUN_PREDICT)) throw MraError(MRA_ERR_STATE, "mra_run with MRA_RUN_PREDICT has not completed");
        HIP_TRY(mraSetDevice(pl->device));
        HIP_TRY(mraMemcpy(mean, pl->mean.p, (size_t)pl->P * sizeof(double), hipMemcpyDeviceToHost));
        HIP_TRY(mraMemcpy(var, pl->var.p, (size_t)pl->P * sizeof(double), hipMemcpyDeviceToHost));
        return MRA_OK;
    } catch (const MraError& e) { return fail(pl, e); }
}

int mra_get_buffer(mra_plan* pl, int what, double* out, int64_t cap, int64_t* n_avail) {
    if (!pl || !n_avail) return MRA_ERR_INVALID;
    try {
        HIP_TRY(mraSetDevice(pl->device));
        const double* src = nullptr; int64_t n = 0;
        if (what == 0) { src = pl->W.p; n = (int64_t)pl->W.n; }
        else if (what == 1) { src = pl->dnode.p; n = (int64_t)pl->dnode.n; }
        else if (what == 2) { src = pl->stamps.p; n = (int64_t)pl->stamps.n; }     // -DMRA_STAMPS builds: raw 64-bit clock stamps
        else if (what == 3) { src = pl->pstamps.p; n = (int64_t)pl->pstamps.n; }
        else if (what == 5) { src = pl->kstamps.p; n = (int64_t)pl->kstamps.n; }   // same, knot chain
        else if (what == 6) { src = pl->kstamps2.p; n = (int64_t)pl->kstamps2.n; } // same, fused leaf solve + update
        else if (what == 4) { src = pl->tstamps.p; n = (int64_t)pl->tstamps.n; }   // same, leaf row solves   // same, predictive cascade
        else throw MraError(MRA_ERR_INVALID, "unknown buffer id");
        *n_avail = n;
        if (out && cap > 0) HIP_TRY(mraMemcpy(out, src, (size_t)std::min(cap, n) * sizeof(double), hipMemcpyDeviceToHost));
        return MRA_OK;
    } catch (const MraError& e) { return fail(pl, e); }
}

int mra_get_node_block(mra_plan* pl, int32_t node, int what, double* out, int64_t cap, int64_t* n_rows, int64_t* n_cols) {
    if (!pl || !n_rows || !n_cols) return MRA_ERR_INVALID;
    try {
        if (node < 0 || node >= pl->n_nodes) throw MraError(MRA_ERR_INVALID, "node out of range");
        if (!pl->ran) throw MraError(MRA_ERR_STATE, "mra_run has not completed");
        HIP_TRY(mraSetDevice(pl->device));
        const int m = pl->node_level[node];
        const double* src = nullptr;
        int64_t rows = 0, cols = 0, ld = 0;
        if (what == MRA_BLOCK_W_ROWS) {
            rows = pl->row1[node] - pl->row0[node]; cols = pl->ldw; ld = pl->ldw;
            src = pl->W.p + pl->row0[node] * (long)pl->ldw;
        } else if (what == MRA_BLOCK_LPRIOR || what == MRA_BLOCK_FRONT) {
            if (pl->leaf[node]) throw MraError(MRA_ERR_INVALID, "block exists for non-leaf nodes only");
            const LevelData& lv = pl->lev[m];
            const size_t sl = (size_t)pl->node_slot[node];
            if (what == MRA_BLOCK_LPRIOR) { rows = cols = ld = lv.cw; src = lv.Lp.p + sl * (size_t)lv.cw * lv.cw; }
            else {
                if (lv.panel_only) throw MraError(MRA_ERR_STATE, "the fronts of this level are kept as their panel columns only (large fronts of the leaves' parents); "
                                                                  "set MRA_NO_LOWRANK_PARENT=1 before creating the plan to get whole fronts");
                rows = cols = ld = lv.nf; src = lv.F.p + sl * (size_t)lv.nf * lv.nf;
            }
        } else if (what == MRA_BLOCK_LEAF) {
            if (!pl->leaf[node]) throw MraError(MRA_ERR_INVALID, "block exists for leaves only");
            const int t = pl->leaf_slot[node];
            const int nop = pl->leaf_nop[t];
            rows = nop + pl->na[m] + (pl->row1[node] - pl->row0[node]); cols = ld = nop;
            src = pl->panel.p + pl->leaf_poff[t];
        } else throw MraError(MRA_ERR_INVALID, "unknown block id");
        *n_rows = rows; *n_cols = cols;
        const int64_t n = std::min<int64_t>(cap, rows * cols);
        if (out && n > 0) HIP_TRY(mraMemcpy(out, src, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
        (void)ld;
        return MRA_OK;
    } catch (const MraError& e) { return fail(pl, e); }
}

int mra_get_timers(mra_plan* pl, double* out, int cap) {
    if (!pl || !out) return 0;
    int n = std::min(cap, 5);
    for (int k = 0; k < n; ++k) out[k] = pl->phase_ms[k];
    return n;
}

int mra_plan_set_option(mra_plan* pl, int option, int64_t value) {
    if (!pl) return MRA_ERR_INVALID;
    // options that change which kernel produces or factorises the leaves' C blocks bring the phantom-row launch back; the others
    // (timing, front / knot / solve / update variants) never touch C
    if (option == 2 || option == 3 || option == 6 || option == 11) pl->cphantom_valid = false;
    if (option == 1) { pl->ktiming = value != 0; return MRA_OK; }
    if (option == 2) { pl->use_fused = value != 0; return MRA_OK; }
    if (option == 3) { pl->gemm_lds = value != 0; return MRA_OK; }
    if (option == 4) { pl->use_front_fused = value != 0; return MRA_OK; }
    if (option == 5) { pl->use_knot_chain = value != 0; return MRA_OK; }
    if (option == 6) { pl->use_leaf_gemm = value != 0; pl->leaf_gemm_update = value == 2; return MRA_OK; }
    if (option == 7) { pl->use_leaf_solve = value != 0; pl->leaf_solve_mode = (int)value; return MRA_OK; }
    if (option == 8) { pl->use_pred_update = value != 0; return MRA_OK; }
    if (option == 10) { pl->leaf_solve_split = value == 2 ? 2 : 1; return MRA_OK; }
    if (option == 11) { pl->use_chol_lds = (int)value; return MRA_OK; }
    if (option == 12) { pl->seg_gemm_lds = value != 0; return MRA_OK; }
    if (option == 14) { pl->use_syrk_blk = (int)value; return MRA_OK; }
    if (option == 15) { pl->use_prior_level = value != 0; return MRA_OK; }
    if (option == 16) { pl->use_hi_fold = (int)value; return MRA_OK; }
    if (option == 17) { pl->use_lik_rows = value != 0; return MRA_OK; }
    if (option == 13) { pl->ut_gather = value != 0; return MRA_OK; }
    if (option == 99) {
        // kernel-shape switches for A/B runs.  Bits 8 and 32 keep the results (predictive cascade at two workgroups per CU, the
        // wide leaf-residual shape); bits 1, 2, 4 (no Ut scatter / no W stores / constant instead of the kernel) give WRONG results
        // and exist only in the diagnostic what-if build (`make whatif`, -DMRA_WHATIF): the product library refuses them.
#ifdef MRA_WHATIF
        pl->dbg = (int)value;
#else
        if (value & ~(int64_t)(8 | 32)) return fail(pl, MraError(MRA_ERR_INVALID, "option 99: bits other than 8 and 32 need the -DMRA_WHATIF diagnostic build"));
        pl->dbg = (int)value;
#endif
        return MRA_OK;
    }
    return fail(pl, MraError(MRA_ERR_INVALID, "unknown option"));
}

int mra_plan_get_option(mra_plan* pl, int option, int64_t* value) {
    if (!pl || !value) return MRA_ERR_INVALID;
    switch (option) {
        case 1: *value = pl->ktiming; break;
        case 2: *value = pl->use_fused; break;
        case 3: *value = pl->gemm_lds; break;
        case 4: *value = pl->use_front_fused; break;
        case 5: *value = pl->use_knot_chain; break;
        case 6: *value = pl->use_leaf_gemm ? (pl->leaf_gemm_update ? 2 : 1) : 0; break;
        case 7: *value = pl->leaf_solve_mode; break;
        case 8: *value = pl->use_pred_update; break;
        case 10: *value = pl->leaf_solve_split; break;
        case 11: *value = pl->use_chol_lds; break;
        case 12: *value = pl->seg_gemm_lds; break;
        case 14: *value = pl->use_syrk_blk; break;
        case 15: *value = pl->use_prior_level; break;
        case 16: *value = pl->use_hi_fold; break;
        case 17: *value = pl->use_lik_rows; break;
        case 13: *value = pl->ut_gather; break;
        case 99: *value = pl->dbg; break;
        default: return fail(pl, MraError(MRA_ERR_INVALID, "unknown option"));
    }
    return MRA_OK;
}

// Raise the dynamic-LDS limit of every kernel this plan can launch on ITS device now instead of at first launch (the
// attribute is per device and per kernel; plans on different devices each do it).  n_kernels: how many kernels are on record.
int mra_plan_prepare(mra_plan* pl, int64_t* n_kernels) {
    if (!pl) return MRA_ERR_INVALID;
    try {
        HIP_TRY(mraSetDevice(pl->device));
        pl->prepare_only = true;
        struct Guard { mra_plan* p; ~Guard() { p->prepare_only = false; } } guard{pl};
        launch_trsm2(pl, nullptr, 0, 1, 0, 1);
        ensure_big_lds(pl, {(const void*)k_front<true>, (const void*)k_front<false>, (const void*)k_parent_front<2>, (const void*)k_parent_front<4>,
                            (const void*)k_parent_front<8>, (const void*)k_parent_front<12>, (const void*)k_leaf_solve_update<8, 13, true>});
        if (pl->regular) {
            CascadeArgs ar{};
            ar.n_wg = 1;
            launch_cascade_any(pl, ar);
            KnotChainArgs ka{};
            ka.nl = 1;
            launch_knot_chain(pl, ka);
            PredArgs pa{};
            pa.n_wg = 1;
            launch_predict_any(pl, pa, 0);
            pa.leaf_upd = (const unsigned char*)1;       // never dereferenced: prepare_only returns before any launch
            launch_predict_any(pl, pa, 0);
        }
        if (n_kernels) *n_kernels = (int64_t)pl->big_lds_done.size();
        return MRA_OK;
    } catch (const MraError& e) { return fail(pl, e); }
}

int mra_get_kernel_stats(mra_plan* pl, int which, char* name, int name_cap, int* launches, double* ms, double* flops) {
    if (!pl || which < 0 || which >= KF_COUNT) return MRA_ERR_INVALID;
    const bool fused = pl->regular && pl->use_fused && !pl->host_cov;
    const bool hi_path = !fused && pl->regular_hi && pl->use_fused && !pl->host_cov;
    if (name && name_cap > 0) { strncpy(name, kfam_name[fused ? 0 : (hi_path ? 2 : 1)][which], name_cap - 1); name[name_cap - 1] = 0; }
    if (launches) *launches = pl->kstat[which].launches;
    if (ms) *ms = pl->kstat[which].ms;
    if (flops) *flops = pl->kstat[which].flops;
    return MRA_OK;
}

int mra_get_kernel_work(mra_plan* pl, int which, double* out, int cap) {
    if (!pl || !out || which < 0 || which >= KF_COUNT) return MRA_ERR_INVALID;
    const double v[4] = {pl->kstat[which].flops, pl->kstat[which].flops_exec, pl->kstat[which].bytes, pl->kstat[which].ms};
    for (int k = 0; k < std::min(cap, 4); ++k) out[k] = v[k];
    return MRA_OK;
}

int mra_device_synchronize(int device) {
    if (g_dry) return MRA_OK;
    if (hipSetDevice(device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { g_last_error = "hipDeviceSynchronize failed"; return MRA_ERR_HIP; }
    return MRA_OK;
}

int mra_kernel_family_count(void) { return KF_COUNT; }

int mra_plan_info(mra_plan* pl, int64_t* out, int cap) {
    if (!pl || !out) return 0;
    int64_t v[8] = {pl->P, pl->ldw, pl->Ka, (int64_t)pl->leaf_nodes.size(), (int64_t)pl->W.n * 8,
                    (int64_t)pl->panel.n * 8, (int64_t)pl->Gt.n * 8, pl->n_nodes};
    int n = std::min(cap, 8);
    for (int k = 0; k < n; ++k) out[k] = v[k];
    return n;
}

// Diagnostics: evaluate a device kernel on n distances (pyMRA/MRATools.py:265-301 on D = dist(...)).
int mra_eval_kernel(int kind, const double* params, int n_params, const double* D, int64_t n, double* out) {
    if (!params || n_params < 3 || !D || !out || n <= 0 || kind < 0 || kind > MRA_KERNEL_KANTER) return MRA_ERR_INVALID;
    KernelParams kp{};
    kp.kind = kind; kp.d = 1; kp.l = params[0]; kp.sig = params[1]; kp.scale = params[2];
    kp.circular = (n_params >= 4 && params[3] != 0.0) ? 1 : 0;
    derive_kernel_params(kp);
    if (g_dry) { g_last_error = "MRA_HOST_DRYRUN: nothing runs in this mode"; return MRA_ERR_STATE; }
    double *dD = nullptr, *dO = nullptr;
    if (mraMalloc((void**)&dD, n * sizeof(double)) != hipSuccess || mraMalloc((void**)&dO, n * sizeof(double)) != hipSuccess) {
        g_last_error = "hipMalloc failed (no GPU?)";
        if (dD) mraFree(dD);
        return MRA_ERR_HIP;
    }
    mraMemcpy(dD, D, n * sizeof(double), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_eval_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, dD, dO, (long)n, kp);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = mraMemcpy(out, dO, n * sizeof(double), hipMemcpyDeviceToHost);
    mraFree(dD); mraFree(dO);
    if (e != hipSuccess) { g_last_error = hipGetErrorString(e); return MRA_ERR_HIP; }
    return MRA_OK;
}

// ---- native tree replay (host only, no GPU needed) ---------------------------------------------------
struct mra_tree { mra_topo::Result r; };

int mra_tree_replay_2d(const double* locs, int64_t N, int32_t r, int32_t M, uint32_t* mt_key, int32_t* mt_pos, mra_tree** out) {
    if (!locs || !mt_key || !mt_pos || !out) return MRA_ERR_INVALID;
    *out = nullptr;
    mra_tree* t = new mra_tree();
    int rc;
    try { rc = mra_topo::replay_quadtree(locs, N, r, M, mt_key, mt_pos, t->r); }
    catch (const std::exception& e) { g_last_error = e.what(); delete t; return MRA_ERR_INVALID; }
    if (rc != 0) { delete t; return 1; }
    *out = t;
    return MRA_OK;
}

int mra_tree_replay_2d_into(const double* locs, int64_t N, int32_t r, int32_t M, uint32_t* mt_key, int32_t* mt_pos, int64_t cap_rows,
                            int64_t* perm, int64_t* src, uint8_t* in_leaf, int64_t* knot_rows, mra_tree** out) {
    if (!locs || !mt_key || !mt_pos || !out || !perm || !src || !in_leaf || !knot_rows) return MRA_ERR_INVALID;
    *out = nullptr;
    if (M < 0 || M > 15 || cap_rows < N + 15 * ((int64_t)1 << (2 * M))) { g_last_error = "mra_tree_replay_2d_into: cap_rows must be at least N + 15 * 4^M"; return MRA_ERR_INVALID; }
    mra_tree* t = new mra_tree();
    t->r.ext_perm = perm; t->r.ext_src = src; t->r.ext_in_leaf = in_leaf; t->r.ext_knot_rows = knot_rows;
    int rc;
    try { rc = mra_topo::replay_quadtree(locs, N, r, M, mt_key, mt_pos, t->r); }
    catch (const std::exception& e) { g_last_error = e.what(); delete t; return MRA_ERR_INVALID; }
    if (rc != 0) { delete t; return 1; }
    *out = t;
    return MRA_OK;
}

// ---- MRATree.__init__ for large 2-D trees in one call ---------------------------------------------------------------------------
// The tree replay and the plan construction of an end-to-end MRATree(...) call, overlapped: as soon as the replay has the
// partition and the row layout (everything but the knots), a helper thread sizes and allocates the plan, gathers and uploads the
// locations and the observations and builds the leaf descriptors - while the caller's thread is still drawing knots, the one
// sequential part (it has to consume NumPy's MT19937 stream in the reference's order).  The knot rows, the leaves' knot
// counts and the knot coordinates go in when both are done.
namespace {
struct ReplayPlanCtx {
    const double* locs; const double* y; double R; int device;
    mra_plan* pl = nullptr;
    int rc = MRA_OK;
    std::string err;
};
void replay_plan_hook(void* user, const mra_topo::Result& t) {
    ReplayPlanCtx& c = *(ReplayPlanCtx*)user;
    mra_plan* pl = nullptr;
    try {
        PlanTrace tr("plan beside the knot draws");
        int ndev = 0;
        if (!g_dry) {
            if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
                throw MraError(MRA_ERR_HIP, "no HIP device available: libmra_hip needs an AMD GPU (gfx950); there is no CPU fallback");
            if (c.device < 0 || c.device >= ndev) throw MraError(MRA_ERR_INVALID, "device ordinal out of range");
        }
        HIP_TRY(mraSetDevice(c.device));
        pl = new mra_plan();
        pl->device = c.device;
        pl->knots_pending = true;
        pl->P = t.P; pl->d = 2; pl->n_levels = t.n_levels; pl->n_nodes = t.n_nodes;
        pl->level_ptr.assign(t.level_ptr.begin(), t.level_ptr.end());
        pl->row0.assign(t.row0.begin(), t.row0.end());
        pl->row1.assign(t.row1.begin(), t.row1.end());
        pl->leaf.assign(t.leaf.begin(), t.leaf.end());
        pl->parent.assign(t.parent.begin(), t.parent.end());
        pl->child_ptr.assign(t.child_ptr.begin(), t.child_ptr.end());
        pl->child_list.assign(t.child_list.begin(), t.child_list.end());
        pl->knot_ptr.assign(t.knot_ptr.begin(), t.knot_ptr.end());      // non-leaf nodes final; the leaves' entries are provisional
        pl->cw.assign(t.cw.begin(), t.cw.end());
        if (!g_dry) {
            acquire_streams(pl);
            HIP_TRY(hipEventCreateWithFlags(&pl->ev_fork, hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&pl->ev_join, hipEventDisableTiming));
        }
        for (int k = 0; k < 6 && !g_dry; ++k) HIP_TRY(hipEventCreate(&pl->ev[k]));
        tr.mark("copies of the topology, streams, events");
        init_arena(pl);
        const int64_t* src = t.ext_perm ? t.ext_src : t.src.data();
        const int64_t* perm = t.ext_perm ? t.ext_perm : t.perm.data();
        std::lock_guard<std::mutex> lock(g_stage_mutex);
        double* xp = stage_buffer((size_t)pl->P * 3);
        double* yp = xp + (size_t)pl->P * 2;
        // the gather of locations and observations into leaf order runs beside build_static (neither needs the other)
        const long P = pl->P;
        bool gather_failed = false;
        std::thread gather([&, P]() {
            const double nan = std::nan("");
            const double* locs = c.locs; const double* y = c.y;
            try {
                parallel_rows(P, [&](int64_t a, int64_t b) {
                    for (int64_t p = a; p < b; ++p) {
                        const int64_t q = src[p];
                        xp[2 * p] = locs[2 * q]; xp[2 * p + 1] = locs[2 * q + 1];
                        yp[p] = perm[p] < 0 ? nan : y[q];
                    }
                });
            } catch (...) { gather_failed = true; }
        });
        struct Join { std::thread& t; ~Join() { if (t.joinable()) t.join(); } } join{gather};
        build_static(pl);
        tr.mark("build_static");
        gather.join();
        if (gather_failed) throw MraError(MRA_ERR_INVALID, "gathering locations / observations failed (thread creation)");
        tr.mark("(gather of locations and observations: beside it)");
        int rc = mra_plan_set_locs(pl, xp);
        if (rc == MRA_OK) rc = mra_plan_set_obs(pl, yp, c.R);
        if (rc != MRA_OK) throw MraError(rc, pl->err);
        tr.mark("uploads, leaf descriptors");
        c.pl = pl;
    } catch (const MraError& e) {
        c.rc = e.code; c.err = e.msg;
        if (pl) { drop_arena(pl); if (!g_dry) return_streams(pl); delete pl; }
    } catch (const std::exception& e) {
        c.rc = MRA_ERR_INVALID; c.err = e.what();
        if (pl) { drop_arena(pl); if (!g_dry) return_streams(pl); delete pl; }
    }
}
}  // namespace

int mra_plan_create_replay_2d(const double* locs, int64_t N, int32_t r, int32_t M, uint32_t* mt_key, int32_t* mt_pos,
                              const double* y, double R, int device, int64_t cap_rows,
                              int64_t* perm, int64_t* src, uint8_t* in_leaf, int64_t* knot_rows, mra_tree** tree_out, mra_plan** plan_out) {
    if (!locs || !mt_key || !mt_pos || !y || !tree_out || !plan_out || !perm || !src || !in_leaf || !knot_rows) return MRA_ERR_INVALID;
    *tree_out = nullptr; *plan_out = nullptr;
    if (M < 0 || M > 15 || cap_rows < N + 15 * ((int64_t)1 << (2 * M))) { g_last_error = "mra_plan_create_replay_2d: cap_rows must be at least N + 15 * 4^M"; return MRA_ERR_INVALID; }
    if (!(R > 0.0)) { g_last_error = "R must be a positive scalar"; return MRA_ERR_INVALID; }
    mra_tree* t = new mra_tree();
    t->r.ext_perm = perm; t->r.ext_src = src; t->r.ext_in_leaf = in_leaf; t->r.ext_knot_rows = knot_rows;
    ReplayPlanCtx ctx{locs, y, R, device};
    int rc;
    try { rc = mra_topo::replay_quadtree(locs, N, r, M, mt_key, mt_pos, t->r, replay_plan_hook, &ctx); }
    catch (const std::exception& e) { g_last_error = e.what(); rc = MRA_ERR_INVALID; }
    mra_plan* pl = ctx.pl;
    if (rc == 0 && ctx.rc != MRA_OK) { g_last_error = ctx.err; rc = ctx.rc; }
    if (rc != 0) {                                           // not a large-2-D tree (1), or an error: nothing is handed out
        if (pl) mra_plan_destroy(pl);
        delete t;
        return rc;
    }
    try {
        PlanTrace tr("knots into the plan");
        HIP_TRY(mraSetDevice(pl->device));
        pl->knot_ptr.assign(t->r.knot_ptr.begin(), t->r.knot_ptr.end());
        pl->knot_rows.assign(knot_rows, knot_rows + t->r.n_knot_rows);
        pl->knots_pending = false;
        fill_knot_arrays(pl);
        set_knot_coords_src(pl, locs, src);                  // knot coordinates straight from the caller's rows
    } catch (const MraError& e) {
        const int code = fail(nullptr, e);
        mra_plan_destroy(pl);
        delete t;
        return code;
    }
    *tree_out = t;
    *plan_out = pl;
    return MRA_OK;
}

int mra_tree_sizes(mra_tree* t, int64_t* out5) {
    if (!t || !out5) return MRA_ERR_INVALID;
    out5[0] = t->r.P; out5[1] = t->r.n_nodes; out5[2] = t->r.n_levels;
    out5[3] = (int64_t)t->r.child_list.size(); out5[4] = t->r.n_knot_rows;
    return MRA_OK;
}

int mra_tree_export(mra_tree* t, int64_t* perm, int64_t* src, uint8_t* in_leaf, int64_t* level_ptr, int32_t* node_level,
                    int64_t* row0, int64_t* row1, uint8_t* leaf, int32_t* parent, int32_t* child_ptr, int32_t* child_list,
                    int64_t* knot_ptr, int64_t* knot_rows, int32_t* cw, int32_t* preorder) {
    if (!t) return MRA_ERR_INVALID;
    const mra_topo::Result& r = t->r;
    auto cp = [](auto* dst, const auto& v) { if (dst && !v.empty()) memcpy(dst, v.data(), v.size() * sizeof(v[0])); };
    cp(perm, r.perm); cp(src, r.src); cp(in_leaf, r.in_leaf); cp(level_ptr, r.level_ptr); cp(node_level, r.level);
    cp(row0, r.row0); cp(row1, r.row1); cp(leaf, r.leaf); cp(parent, r.parent); cp(child_ptr, r.child_ptr);
    cp(child_list, r.child_list); cp(knot_ptr, r.knot_ptr); cp(knot_rows, r.knot_rows); cp(cw, r.cw); cp(preorder, r.preorder);
    return MRA_OK;
}

int mra_tree_free(mra_tree* t) { delete t; return MRA_OK; }

// ---- multi-GPU ------------------------------------------------------------------------------------
int mra_plan_set_reduce_level(mra_plan* pl, int level) {
    if (!pl) return MRA_ERR_INVALID;
    if (level >= pl->n_levels) return fail(pl, MraError(MRA_ERR_INVALID, "reduce level out of range"));
    if (pl->lowrank_parent && level >= pl->n_levels - 3)
        return fail(pl, MraError(MRA_ERR_INVALID, "reduce level too deep: the fronts of the leaves' parents are kept as panels on this plan (set MRA_NO_LOWRANK_PARENT=1 before creating it)"));
    pl->reduce_level = level;
    return MRA_OK;
}

int mra_reduce_size(mra_plan* pl, int64_t* n) {
    if (!pl || !n) return MRA_ERR_INVALID;
    if (pl->reduce_level < 0) return fail(pl, MraError(MRA_ERR_STATE, "no reduce level set"));
    *n = (int64_t)pl->lev[pl->reduce_level].F.n;
    return MRA_OK;
}

int mra_reduce_export(mra_plan* pl, double* out) {
    if (!pl || !out) return MRA_ERR_INVALID;
    try {
        if (!pl->split_pending) throw MraError(MRA_ERR_STATE, "no split run pending");
        HIP_TRY(mraSetDevice(pl->device));
        HIP_TRY(hipStreamSynchronize(pl->stream));
        LevelData& lv = pl->lev[pl->reduce_level];
        HIP_TRY(mraMemcpy(out, lv.F.p, lv.F.n * sizeof(double), hipMemcpyDeviceToHost));
        return MRA_OK;
    } catch (const MraError& e) { return fail(pl, e); }
}

int mra_reduce_import(mra_plan* pl, const double* in) {
    if (!pl || !in) return MRA_ERR_INVALID;
    try {
        if (!pl->split_pending) throw MraError(MRA_ERR_STATE, "no split run pending");
        HIP_TRY(mraSetDevice(pl->device));
        LevelData& lv = pl->lev[pl->reduce_level];
        HIP_TRY(mraMemcpy(lv.F.p, in, lv.F.n * sizeof(double), hipMemcpyHostToDevice));
        return MRA_OK;
    } catch (const MraError& e) { return fail(pl, e); }
}

int mra_comm_unique_id(char* out, int cap) {
    if (!out || cap < 128) return MRA_ERR_INVALID;
    void* h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) { g_last_error = "cannot load librccl.so"; return MRA_ERR_COMM; }
    typedef ncclResult_t (*uid_t_)(ncclUniqueId*);
    uid_t_ fn = (uid_t_)dlsym(h, "ncclGetUniqueId");
    if (!fn || fn((ncclUniqueId*)out) != ncclSuccess) { g_last_error = "ncclGetUniqueId failed"; return MRA_ERR_COMM; }
    return MRA_OK;
}

int mra_comm_init(mra_plan* pl, const char* uid, int n_ranks, int rank) {
    if (!pl || !uid) return MRA_ERR_INVALID;
    try {
        HIP_TRY(mraSetDevice(pl->device));
        pl->rccl = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!pl->rccl) pl->rccl = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!pl->rccl) throw MraError(MRA_ERR_COMM, "cannot load librccl.so");
        ncclUniqueId id;
        static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
        memcpy(&id, uid, sizeof id);
        typedef ncclResult_t (*init_t)(ncclComm_t*, int, ncclUniqueId, int);
        init_t fn = (init_t)dlsym(pl->rccl, "ncclCommInitRank");
        if (!fn) throw MraError(MRA_ERR_COMM, "ncclCommInitRank not found");
        pl->allreduce = (decltype(pl->allreduce))dlsym(pl->rccl, "ncclAllReduce");
        if (!pl->allreduce) throw MraError(MRA_ERR_COMM, "ncclAllReduce not found");
        if (fn(&pl->comm, n_ranks, id, rank) != ncclSuccess) { pl->comm = nullptr; throw MraError(MRA_ERR_COMM, "ncclCommInitRank failed"); }
        pl->n_ranks = n_ranks; pl->rank = rank;
        return MRA_OK;
    } catch (const MraError& e) { return fail(pl, e); }
}

}  // extern "C"
