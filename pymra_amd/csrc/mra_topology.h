// mra_topology.h - native replay of pyMRA's tree construction for the common large 2-D case.
//
// What it replays (same rules as pymra_amd/topology.py, which stays the general implementation and the
// fallback):
//   * quadrant partition about the coordinate means of every node with more than 100 rows
//     (pyMRA/MRANode.py:232-239; np.mean(locs, axis=0) is a sequential row accumulation - reproduced here
//     bit for bit, because rows that sit exactly on a mean change child otherwise),
//   * knots of every non-leaf node: np.random.choice(arange(n_cand), size=r, replace=False) on NumPy's
//     GLOBAL legacy RandomState, in depth-first pre-order (pyMRA/MRANode.py:191-193).  That call is
//     permutation(n)[:r], i.e. a Fisher-Yates shuffle of arange(n) driven by MT19937 through
//     random_interval() (masked rejection sampling on 32-bit draws).  The caller hands over the MT19937
//     key/pos of np.random.get_state() and writes the advanced state back with np.random.set_state(),
//   * leaves: every not-yet-used row of the leaf is a knot (pyMRA/MRANode.py:42-45),
//   * the padded, leaf-ordered flat layout of pymra_amd.topology.Topology.
// Not applicable (returns 1, state untouched): a node above the leaves with <= 100 rows or
// <= max(100, r, 4) candidates, or an empty quadrant - those follow other rules in the reference
// (KMeans knots / splits) and are handled by the Python replay.
#pragma once
#include <stdint.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

namespace mra_topo {

struct MT19937 {
    uint32_t* key;
    int pos;
    // (written as three plain loops over distinct index ranges with the conditional as a mask, so that the compiler vectorises
    //  them: the first 227 words read old words 397 ahead, the rest read new words 227 behind - dependence distances far above
    //  any vector width)
    inline void regenerate() {
        const uint32_t MATRIX_A = 0x9908b0dfU, UPPER = 0x80000000U, LOWER = 0x7fffffffU;
        uint32_t* const k = key;
#pragma clang loop vectorize(enable) interleave(enable)
        for (int kk = 0; kk < 624 - 397; ++kk) {
            const uint32_t y = (k[kk] & UPPER) | (k[kk + 1] & LOWER);
            k[kk] = k[kk + 397] ^ (y >> 1) ^ ((0U - (y & 1U)) & MATRIX_A);
        }
#pragma clang loop vectorize(enable) interleave(enable)
        for (int kk = 624 - 397; kk < 623; ++kk) {
            const uint32_t y = (k[kk] & UPPER) | (k[kk + 1] & LOWER);
            k[kk] = k[kk - (624 - 397)] ^ (y >> 1) ^ ((0U - (y & 1U)) & MATRIX_A);
        }
        const uint32_t y = (k[623] & UPPER) | (k[0] & LOWER);
        k[623] = k[396] ^ (y >> 1) ^ ((0U - (y & 1U)) & MATRIX_A);
        pos = 0;
    }
    // n tempered outputs (n <= 624 - pos) straight from the state, vectorisable
    inline void temper_out(uint32_t* __restrict__ out, int n) {
        const uint32_t* __restrict__ k = key + pos;
#pragma clang loop vectorize(enable) interleave(enable)
        for (int i = 0; i < n; ++i) {
            uint32_t y = k[i];
            y ^= (y >> 11);
            y ^= (y << 7) & 0x9d2c5680U;
            y ^= (y << 15) & 0xefc60000U;
            y ^= (y >> 18);
            out[i] = y;
        }
        pos += n;
    }
    inline uint32_t next32() {
        if (pos == 624) regenerate();
        uint32_t y = key[pos++];
        y ^= (y >> 11);
        y ^= (y << 7) & 0x9d2c5680U;
        y ^= (y << 15) & 0xefc60000U;
        y ^= (y >> 18);
        return y;
    }
    // numpy/random/src/distributions: random_interval(max), max <= 0xffffffff
    inline uint64_t interval(uint64_t max) {
        if (max == 0) return 0;
        uint64_t mask = max;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16; mask |= mask >> 32;
        uint64_t v;
        if (max <= 0xffffffffULL) {
            while ((v = (next32() & mask)) > max) {}
        } else {
            while ((v = ((((uint64_t)next32()) << 32) | next32()) & mask) > max) {}
        }
        return v;
    }
};

// The raw 32-bit output stream of an MT19937 produced by a helper thread a few blocks ahead of its one consumer (the knot
// draws of the replay, which are sequential by nature: every shuffle starts where the previous one stopped).  The stream is
// the generator's, word for word; the state after `consumed` words is rebuilt from the snapshot taken at the start of the
// block the consumer stopped in.
struct MTStream {
    static constexpr int GEN = 624, PER_BLOCK = 128, BLOCK = GEN * PER_BLOCK, NBLOCK = 8;
    std::vector<uint32_t> ring;                  // NBLOCK blocks of tempered outputs
    std::vector<uint32_t> snap;                  // key[624] at the start of each ring block
    std::vector<int> snap_pos;
    std::atomic<long> produced{0}, released{0};  // blocks written / blocks the consumer is done with
    std::atomic<bool> stop{false};
    std::thread th;
    MT19937 gen;
    uint32_t key[624], key_init[624];
    int pos_init;
    long cur_block = -1;
    const uint32_t* cur = nullptr;
    int cur_i = BLOCK;
    long consumed_blocks = 0;
    MTStream(const uint32_t* k, int pos) : ring((size_t)NBLOCK * BLOCK), snap((size_t)NBLOCK * 624), snap_pos(NBLOCK), gen{key, pos} {
        for (int i = 0; i < 624; ++i) key[i] = key_init[i] = k[i];
        pos_init = pos;
        th = std::thread([this]() {
            long b = 0;
            while (!stop.load(std::memory_order_acquire)) {
                if (b - released.load(std::memory_order_acquire) >= NBLOCK) { std::this_thread::yield(); continue; }
                uint32_t* out = ring.data() + (size_t)(b % NBLOCK) * BLOCK;
                for (int i = 0; i < 624; ++i) snap[(size_t)(b % NBLOCK) * 624 + i] = key[i];
                snap_pos[b % NBLOCK] = gen.pos;
                for (int i = 0; i < BLOCK;) {
                    if (gen.pos == 624) gen.regenerate();
                    const int n = std::min(624 - gen.pos, BLOCK - i);
                    gen.temper_out(out + i, n);
                    i += n;
                }
                ++b;
                produced.store(b, std::memory_order_release);
            }
        });
    }
    ~MTStream() { stop.store(true, std::memory_order_release); if (th.joinable()) th.join(); }
    inline uint32_t next32() {
        if (cur_i == BLOCK) {
            if (cur_block >= 0) released.store(cur_block + 1, std::memory_order_release);
            ++cur_block;
            while (produced.load(std::memory_order_acquire) <= cur_block) std::this_thread::yield();
            cur = ring.data() + (size_t)(cur_block % NBLOCK) * BLOCK;
            cur_i = 0;
        }
        return cur[cur_i++];
    }
    inline uint64_t interval(uint64_t max) {                    // numpy random_interval, as MT19937::interval
        if (max == 0) return 0;
        uint64_t mask = max;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16; mask |= mask >> 32;
        uint64_t v;
        if (max <= 0xffffffffULL) {
            while ((v = (next32() & mask)) > max) {}
        } else {
            while ((v = ((((uint64_t)next32()) << 32) | next32()) & mask) > max) {}
        }
        return v;
    }
    // Steps i = i_hi .. i_lo of the shuffle in their store-only form p[interval(i)] = p[i] (see replay_quadtree), straight off
    // the ring: random_interval(i) masks a 32-bit word with the smallest 2^k - 1 >= i and rejects values > i.  Written without a
    // data-dependent branch - a rejected word stores p[i] onto itself and does not advance i - because the rejection is a coin
    // flip right below every power of two and mispredicts cost more than the arithmetic.  On return i = i_lo - 1.
    inline void shuffle_stores(int32_t* pp, int64_t& i, int64_t i_lo) {
        while (i >= i_lo) {
            uint64_t mask = (uint64_t)i;
            mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16; mask |= mask >> 32;
            const int64_t lo = std::max<int64_t>(i_lo, (int64_t)(mask >> 1) + 1);     // the mask is the same for i in [lo, i]
            const uint32_t m32 = (uint32_t)mask;
            while (i >= lo) {
                if (cur_i == BLOCK) {
                    if (cur_block >= 0) released.store(cur_block + 1, std::memory_order_release);
                    ++cur_block;
                    while (produced.load(std::memory_order_acquire) <= cur_block) std::this_thread::yield();
                    cur = ring.data() + (size_t)(cur_block % NBLOCK) * BLOCK;
                    cur_i = 0;
                }
                const uint32_t* w = cur + cur_i;
                const int nw = BLOCK - cur_i;
                int k = 0;
                int64_t ii = i;
                for (; k < nw && ii >= lo; ++k) {
                    const int64_t v = (int64_t)(w[k] & m32);
                    const bool ok = v <= ii;
                    pp[ok ? v : ii] = pp[ii];
                    ii -= ok;
                }
                cur_i += k;
                i = ii;
            }
        }
    }
    // The targets v_i = interval(i) of steps i = i_hi .. i_lo, into vs[i] (same words, same rejection rule as shuffle_stores; a
    // rejected word writes vs[i], the accepted one overwrites it).  No random access at all: the consumer resolves the shuffle
    // backwards from these (replay_quadtree).  On return i = i_lo - 1.
    inline void draw_targets(uint32_t* vs, int64_t& i, int64_t i_lo) {
        while (i >= i_lo) {
            uint64_t mask = (uint64_t)i;
            mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16; mask |= mask >> 32;
            const int64_t lo = std::max<int64_t>(i_lo, (int64_t)(mask >> 1) + 1);
            const uint32_t m32 = (uint32_t)mask;
            while (i >= lo) {
                if (cur_i == BLOCK) {
                    if (cur_block >= 0) released.store(cur_block + 1, std::memory_order_release);
                    ++cur_block;
                    while (produced.load(std::memory_order_acquire) <= cur_block) std::this_thread::yield();
                    cur = ring.data() + (size_t)(cur_block % NBLOCK) * BLOCK;
                    cur_i = 0;
                }
                const uint32_t* w = cur + cur_i;
                const int nw = BLOCK - cur_i;
                int k = 0;
                int64_t ii = i;
                for (; k < nw && ii >= lo; ++k) {
                    const uint32_t v = w[k] & m32;
                    vs[ii] = v;
                    ii -= (int64_t)v <= ii;
                }
                cur_i += k;
                i = ii;
            }
        }
    }
    // generator state after exactly the words handed out so far
    void final_state(uint32_t* key_out, int32_t* pos_out) {
        stop.store(true, std::memory_order_release);
        if (th.joinable()) th.join();
        if (cur_block < 0) { for (int i = 0; i < 624; ++i) key_out[i] = key_init[i]; *pos_out = pos_init; return; }
        uint32_t k2[624];
        const size_t sb = (size_t)(cur_block % NBLOCK);
        for (int i = 0; i < 624; ++i) k2[i] = snap[sb * 624 + i];
        MT19937 g{k2, snap_pos[sb]};
        for (int i = 0; i < cur_i; ++i) (void)g.next32();
        for (int i = 0; i < 624; ++i) key_out[i] = k2[i];
        *pos_out = g.pos;
    }
};

// sense-reversing barrier for a small team of threads that meet every few hundred microseconds (spin, then yield)
struct SpinBarrier {
    const int n;
    std::atomic<int> count{0};
    std::atomic<int> sense{0};
    explicit SpinBarrier(int n_) : n(n_) {}
    void wait() {
        if (n <= 1) return;
        const int s0 = sense.load(std::memory_order_acquire);
        if (count.fetch_add(1, std::memory_order_acq_rel) == n - 1) {
            count.store(0, std::memory_order_relaxed);
            sense.store(s0 ^ 1, std::memory_order_release);
        } else {
            int spins = 0;
            while (sense.load(std::memory_order_acquire) == s0) { if (++spins > 2000) std::this_thread::yield(); }
        }
    }
};

struct Result {
    // optional caller-owned destinations of the four long arrays (capacity: N + 15 * 4^M rows, N knot rows): when set the replay
    // writes them in place and the vectors of the same name stay empty (no 25 MB of copies and unmapping per tree)
    int64_t* ext_perm = nullptr; int64_t* ext_src = nullptr; uint8_t* ext_in_leaf = nullptr; int64_t* ext_knot_rows = nullptr;
    int64_t n_knot_rows = 0;
    int64_t P = 0;
    int32_t n_nodes = 0, n_levels = 0;
    std::vector<int64_t> perm, src, level_ptr, row0, row1, knot_ptr, knot_rows;
    std::vector<uint8_t> in_leaf, leaf;
    std::vector<int32_t> parent, child_ptr, child_list, level, preorder, cw;
};

// Process-wide scratch area of the replay (grow-only: a fresh 100 MB of vectors per call costs more in page faults and unmapping -
// 10 + 10 ms at 1024^2 - than a partition level does, and MLE loops replay the same tree size again and again: README.md:96-104
// builds a new MRATree per objective call).  One replay at a time; release_scratch() gives the memory back
// (mra_release_cached_memory).
struct Scratch {
    std::vector<std::vector<int32_t>> orders, inv;
    std::vector<std::vector<int64_t>> starts, knots;
    std::vector<double> cur, nxt_xy;
    std::vector<uint8_t> code, used;
    std::vector<int32_t> leaf_of, perm_idx;
    std::vector<uint32_t> vs;
    std::vector<uint64_t> bits;
    std::vector<int64_t> leaf_off, pos_of;
};
inline Scratch& scratch() { static Scratch ws; return ws; }
inline std::mutex& scratch_mutex() { static std::mutex m; return m; }
inline void release_scratch() {
    std::lock_guard<std::mutex> lock(scratch_mutex());
    scratch() = Scratch();
}

// Called (on a helper thread, while the knot draws are still running on the caller's thread) as soon as everything that does not
// depend on the knots is final: P, perm / src / in_leaf, the node arrays with knot_ptr of the NON-leaf nodes (r knots each: rows
// 0 .. n_nonleaf r of knot_rows; the leaves' entries and all of knot_rows follow when replay_quadtree returns).  A plan can be
// sized, allocated and fed its locations and observations from this (mra_plan_create_replay_2d).  Must not throw.
typedef void (*LayoutHook)(void* user, const struct Result& partial);

// returns 0 on success, 1 if the tree does not follow the large-2-D rules (nothing modified)
inline int replay_quadtree(const double* xy, int64_t N, int32_t r, int32_t M, uint32_t* mt_key, int32_t* mt_pos, Result& out,
                           LayoutHook hook = nullptr, void* hook_user = nullptr) {
    if (M < 1 || N <= 0 || r <= 0 || N >= 0x7fffffffLL) return 1;
    const bool trace = getenv("MRA_TRACE_REPLAY") != nullptr;         // phase times on stderr (tools/e2e_breakdown.py)
    auto now = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_begin = now();
    double t_cand = 0, t_shuf = 0, t_gen = 0, t_wait = 0;
    std::lock_guard<std::mutex> ws_lock(scratch_mutex());
    Scratch& ws = scratch();
    // ---- all allocations up front, on this thread (a failure inside a helper thread would end the process)
    const int L = M + 1;
    const int64_t nleaf = (int64_t)1 << (2 * M);
    auto& orders = ws.orders; auto& starts = ws.starts; auto& inv = ws.inv; auto& leaf_of = ws.leaf_of;
    auto& cur = ws.cur; auto& nxt_xy = ws.nxt_xy; auto& code = ws.code;
    auto& leaf_off = ws.leaf_off; auto& pos_of = ws.pos_of;
    auto& used = ws.used; auto& knots = ws.knots; auto& perm_idx = ws.perm_idx; auto& vs = ws.vs; auto& bits = ws.bits;
    if ((int)orders.size() < M + 1) orders.resize(M + 1);
    if ((int)starts.size() < M + 1) starts.resize(M + 1);
    if ((int)inv.size() < M) inv.resize(std::max(M, 1));
    for (int m = 0; m <= M; ++m) { orders[m].resize(N); starts[m].assign(((int64_t)1 << (2 * m)) + 1, 0); }
    for (int m = 1; m < M; ++m) inv[m].resize(N);           // inv[m][x] = position of caller row x in orders[m] (level 0 is the identity)
    leaf_of.resize(N);                                     // leaf_of[x] = the leaf of caller row x
    cur.resize(2 * N); nxt_xy.resize(2 * N); code.resize(N);
    leaf_off.assign(nleaf + 1, 0);
    pos_of.resize(N);
    used.assign(N, 0);
    if ((int)knots.size() < M) knots.resize(M);                // per level: r knots per node, node-major
    for (int m = 0; m < M; ++m) knots[m].assign(((int64_t)1 << (2 * m)) * r, -1);
    vs.resize((size_t)N + 8);
    bits.resize(((size_t)N + 63) / 64);
    const int64_t cap_rows = N + 15 * nleaf;
    if (!out.ext_perm) { out.perm.resize(cap_rows); out.src.resize(cap_rows); out.in_leaf.resize(cap_rows); }
    if (!out.ext_knot_rows) out.knot_rows.resize(N);          // every caller row is the knot of exactly one node (leaves take what is left)
    out.n_levels = L;
    out.level_ptr.assign(L + 1, 0);
    for (int m = 0; m < L; ++m) out.level_ptr[m + 1] = out.level_ptr[m] + ((int64_t)1 << (2 * m));
    const int64_t nn_all = out.level_ptr[L];
    out.n_nodes = (int32_t)nn_all;
    out.level.assign(nn_all, 0); out.row0.assign(nn_all, 0); out.row1.assign(nn_all, 0); out.leaf.assign(nn_all, 0);
    out.parent.assign(nn_all, -1); out.child_ptr.assign(nn_all + 1, 0);
    out.child_list.assign(4 * (size_t)out.level_ptr[M], 0);
    out.knot_ptr.assign(nn_all + 1, 0);
    out.cw.assign(L, 0);
    out.preorder.assign(nn_all, 0);
    std::vector<int32_t> upos;
    upos.reserve((size_t)r * (size_t)std::max(M, 1));
    std::vector<int64_t> picks(r);
    std::vector<std::pair<int, int64_t>> stack;
    stack.reserve(4 * (size_t)M + 8);

    for (int64_t i = 0; i < N; ++i) orders[0][i] = (int32_t)i;
    starts[0][0] = 0; starts[0][1] = N;
    memcpy(cur.data(), xy, (size_t)2 * N * sizeof(double));

    int n_thr = (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    if (const char* e = getenv("MRA_HOST_THREADS")) n_thr = std::max(1, atoi(e));
    // the partition team runs BESIDE the knot draws (which only ever need the levels at and above the node they are at): two
    // hardware threads stay with the draws and the generator
    const int n_team = std::max(1, n_thr - 2);

    // ---- partitions of all levels (they do not depend on the knots); the coordinates travel with the row order so every pass
    // is a sequential sweep.  One team of threads walks all levels together (SPMD with spin barriers: spawning threads per level
    // and phase costs more than a deep level does).  Per level: (A) the means - one sequential accumulation per node, as np.mean
    // does it (bit for bit: rows that sit exactly on a mean change child otherwise), nodes dealt to the threads; (B) quadrant
    // codes and counts per WORK ITEM, an item being a node or, on the upper levels where a level has fewer nodes than the team
    // has threads, a slice of a node (the root alone is the whole first pass); (C) one thread turns the counts into the
    // children's start offsets per item - stable: ascending caller index inside every child -; (D) the scatter.
    struct Item { int64_t node, s, e; int64_t off[4]; };
    std::vector<Item> items;
    items.reserve((size_t)nleaf + 64 * (size_t)n_team);
    std::vector<double> mean_xy((size_t)2 * (size_t)nleaf, 0.0);
    std::vector<int64_t> item_thr((size_t)n_team + 1, 0);
    SpinBarrier bar(n_team);
    std::atomic<int> bad{0};
    // The knot draws go depth first (the reference's RNG order), so the partition does, too, at the grain of the four level-1
    // subtrees: the root, then all levels of subtree 0, then subtree 1, ... - the draws of subtree q run while the team partitions
    // subtree q + 1.  progress[q] = m: levels <= m of orders / starts / inv are final inside level-1 subtree q.
    struct Pass { int m; int64_t j0, j1; int q; };
    std::vector<Pass> passes;
    passes.push_back(Pass{0, 0, 1, -1});
    for (int q = 0; q < 4; ++q)
        for (int m = 1; m < M; ++m) passes.push_back(Pass{m, (int64_t)q << (2 * (m - 1)), (int64_t)(q + 1) << (2 * (m - 1)), q});
    std::atomic<int> progress[4];
    for (int q = 0; q < 4; ++q) progress[q].store(0);
    for (int m = 0; m <= M; ++m) starts[m][(size_t)1 << (2 * m)] = N;
    auto team = [&](int tid) {
        for (size_t ip = 0; ip < passes.size(); ++ip) {
            const Pass ps = passes[ip];
            const int m = ps.m;
            const std::vector<int32_t>& ord = orders[m];
            const std::vector<int64_t>& st = starts[m];
            std::vector<int32_t>& nxt = orders[m + 1];
            std::vector<int64_t>& nst = starts[m + 1];
            const double* const cxy = (m & 1) ? nxt_xy.data() : cur.data();       // the coordinate arrays swap roles every level
            double* const nxy = (m & 1) ? cur.data() : nxt_xy.data();
            const int64_t row_lo = st[ps.j0], row_hi = st[ps.j1], rows = row_hi - row_lo;
            if (tid == 0) {
                // work items: about four per thread on the upper levels, whole nodes below
                items.clear();
                const int64_t target = std::max<int64_t>(4096, rows / (4 * (int64_t)n_team));
                for (int64_t j = ps.j0; j < ps.j1; ++j) {
                    const int64_t s0 = st[j], e0 = st[j + 1], n = e0 - s0;
                    if (n <= 100) bad.store(1);
                    const int64_t nc = std::max<int64_t>(1, n / target);
                    for (int64_t c = 0; c < nc; ++c) items.push_back(Item{j, s0 + n * c / nc, s0 + n * (c + 1) / nc, {0, 0, 0, 0}});
                }
                item_thr.assign((size_t)n_team + 1, (int64_t)items.size());
                item_thr[0] = 0;
                size_t k = 0;
                for (int t = 1; t < n_team; ++t) {           // contiguous runs of items of about equal row counts
                    const int64_t goal = row_lo + (rows * t) / n_team;
                    while (k < items.size() && items[k].s < goal) ++k;
                    item_thr[t] = (int64_t)k;
                }
            }
            bar.wait();
            if (bad.load()) return;
            // (A) np.mean(axis=0): sequential accumulation, then / n; the two coordinates are independent sums
            for (int64_t w = 2 * ps.j0 + tid; w < 2 * ps.j1; w += n_team) {
                const int64_t j = w >> 1, s0 = st[j], e0 = st[j + 1];
                const double* cc = cxy + (w & 1);
                double sx = 0.0;
                for (int64_t t = s0; t < e0; ++t) sx += cc[2 * t];
                mean_xy[w] = sx / (double)(e0 - s0);
            }
            bar.wait();
            // (B)
            for (int64_t k = item_thr[tid]; k < item_thr[tid + 1]; ++k) {
                Item& it = items[k];
                const double mx = mean_xy[2 * it.node], my = mean_xy[2 * it.node + 1];
                int64_t cnt[4] = {0, 0, 0, 0};
                for (int64_t t = it.s; t < it.e; ++t) {
                    const uint8_t c = (uint8_t)(2 * (cxy[2 * t] > mx) + (cxy[2 * t + 1] > my));
                    code[t] = c;
                    ++cnt[c];
                }
                for (int c = 0; c < 4; ++c) it.off[c] = cnt[c];
            }
            bar.wait();
            // (C)
            if (tid == 0) {
                size_t k = 0;
                for (int64_t j = ps.j0; j < ps.j1; ++j) {
                    const size_t k0 = k;
                    int64_t tot[4] = {0, 0, 0, 0};
                    for (; k < items.size() && items[k].node == j; ++k) for (int c = 0; c < 4; ++c) tot[c] += items[k].off[c];
                    if (!tot[0] || !tot[1] || !tot[2] || !tot[3]) { bad.store(1); break; }
                    int64_t run[4];
                    run[0] = st[j]; run[1] = run[0] + tot[0]; run[2] = run[1] + tot[1]; run[3] = run[2] + tot[2];
                    for (int c = 0; c < 4; ++c) nst[4 * j + c] = run[c];
                    nst[4 * j + 4] = st[j + 1];            // (the next node writes the same value; its pass may come much later)
                    for (size_t q = k0; q < k; ++q) for (int c = 0; c < 4; ++c) { const int64_t n = items[q].off[c]; items[q].off[c] = run[c]; run[c] += n; }
                }
            }
            bar.wait();
            if (bad.load()) return;
            // (D) stable scatter
            int32_t* const inv_next = (m + 1 < M) ? inv[m + 1].data() : nullptr;
            for (int64_t k = item_thr[tid]; k < item_thr[tid + 1]; ++k) {
                Item& it = items[k];
                int64_t off[4] = {it.off[0], it.off[1], it.off[2], it.off[3]};
                const int64_t j = it.node;
                for (int64_t t = it.s; t < it.e; ++t) {
                    const int64_t d = off[code[t]]++;
                    const int32_t x = ord[t];
                    nxt[d] = x;
                    nxy[2 * d] = cxy[2 * t];
                    nxy[2 * d + 1] = cxy[2 * t + 1];
                    if (inv_next) inv_next[x] = (int32_t)d;
                    else leaf_of[x] = (int32_t)(4 * j + code[t]);
                }
            }
            bar.wait();
            if (tid == 0) {
                if (ps.q < 0) for (int q = 0; q < 4; ++q) progress[q].store(1, std::memory_order_release);
                else progress[ps.q].store(m + 1, std::memory_order_release);
            }
        }
    };
    // ---- the flat layout that does not depend on the knots, on the partition's thread 0 right behind the last level
    const std::vector<int32_t>& oM = orders[M];
    double t_part = 0, t_layout = 0;
    std::atomic<int> layout_done{0};
    // (rows: all team threads, a run of leaves each; node arrays and the hook: thread 0)
    auto layout_rows = [&](int tid) {
        int64_t* const perm_o = out.ext_perm ? out.ext_perm : out.perm.data();
        int64_t* const src_o = out.ext_perm ? out.ext_src : out.src.data();
        uint8_t* const inl_o = out.ext_perm ? out.ext_in_leaf : out.in_leaf.data();
        for (int64_t l = nleaf * tid / n_team; l < nleaf * (tid + 1) / n_team; ++l) {
            const int64_t s = starts[M][l], e = starts[M][l + 1];
            int64_t p = leaf_off[l];
            for (int64_t t = s; t < e; ++t, ++p) { perm_o[p] = oM[t]; src_o[p] = oM[t]; inl_o[p] = 1; pos_of[oM[t]] = p; }
            for (; p < leaf_off[l + 1]; ++p) { perm_o[p] = -1; src_o[p] = oM[s]; inl_o[p] = 0; }      // phantom rows copy the leaf's first location
        }
    };
    auto layout_offsets = [&]() {
        for (int64_t l = 0; l < nleaf; ++l) {
            const int64_t c = starts[M][l + 1] - starts[M][l];
            leaf_off[l + 1] = leaf_off[l] + (c + 15) / 16 * 16;
        }
        out.P = leaf_off[nleaf];
    };
    auto layout = [&]() {
        // node arrays (knot_ptr: the non-leaf nodes' r knots each; the leaves' entries follow with the knots)
        size_t nchild = 0;
        for (int m = 0; m < L; ++m) {
            const int64_t span = (int64_t)1 << (2 * (M - m));
            for (int64_t j = 0; j < ((int64_t)1 << (2 * m)); ++j) {
                const int64_t i = out.level_ptr[m] + j;
                out.level[i] = m;
                out.row0[i] = leaf_off[j * span];
                out.row1[i] = leaf_off[(j + 1) * span];
                out.leaf[i] = (m == M);
                out.parent[i] = m ? (int32_t)(out.level_ptr[m - 1] + j / 4) : -1;
                if (m < M) for (int c = 0; c < 4; ++c) out.child_list[nchild++] = (int32_t)(out.level_ptr[m + 1] + 4 * j + c);
                out.child_ptr[i + 1] = (int32_t)nchild;
                out.knot_ptr[i + 1] = m < M ? (i + 1) * (int64_t)r : out.knot_ptr[i];      // (leaves: provisional, empty)
            }
        }
        out.knot_ptr[nn_all] = N;                          // every caller row is the knot of exactly one node: the total is known
        for (int m = 0; m < M; ++m) out.cw[m] = (r + 15) / 16 * 16;
        {
            size_t np = 0;
            std::vector<int64_t> st2;
            st2.reserve(4 * (size_t)L + 8);
            st2.push_back(0);
            while (!st2.empty()) {
                const int64_t i = st2.back();
                st2.pop_back();
                out.preorder[np++] = (int32_t)i;
                if (!out.leaf[i]) {
                    const int m = out.level[i];
                    const int64_t c0 = out.level_ptr[m + 1] + 4 * (i - out.level_ptr[m]);
                    for (int c = 3; c >= 0; --c) st2.push_back(c0 + c);
                }
            }
        }
        if (hook) hook(hook_user, out);
    };
    std::vector<std::thread> pool;
    struct Joiner { std::vector<std::thread>& p; ~Joiner() { for (auto& t : p) if (t.joinable()) t.join(); } } joiner{pool};
    for (int t = 0; t < n_team; ++t)
        pool.emplace_back([&, t]() {
            team(t);
            if (bad.load()) return;                        // (set before a barrier every thread of the team passes: all of them see it)
            if (t == 0) { t_part = now(); layout_offsets(); }
            bar.wait();
            layout_rows(t);
            bar.wait();
            if (t == 0) {
                layout();
                t_layout = now();
                layout_done.store(1, std::memory_order_release);
            }
        });
    // ---- knots: depth-first pre-order, one shuffle of arange(n_cand) each; a node of level m needs levels <= m of the partition
    MTStream rng(mt_key, *mt_pos);
    stack.push_back({0, 0});
    const int64_t min_cand = std::max<int64_t>(100, std::max<int64_t>(r, 4));
    int failed = 0;
    while (!stack.empty()) {
        const int m = stack.back().first;
        const int64_t j = stack.back().second;
        stack.pop_back();
        if (m >= 1) {
            std::atomic<int>& pq = progress[j >> (2 * (m - 1))];
            if (pq.load(std::memory_order_acquire) < m) {
                const double tw0 = trace ? now() : 0.0;
                int spins = 0;
                while (pq.load(std::memory_order_acquire) < m && !bad.load()) { if (++spins > 200) std::this_thread::yield(); }
                if (trace) t_wait += now() - tw0;
            }
        }
        if (bad.load()) { failed = 1; break; }
        const int64_t s = starts[m][j], e = starts[m][j + 1];
        const double tc0 = trace ? now() : 0.0;
        // positions (inside this node's row order) of the rows its ancestors took as knots: at most r per ancestor.  A knot of the
        // level-k ancestor is followed down the path to this node: inv[l][x] is final for a row x of the level-(l-1) ancestor (that
        // node has been partitioned), and only for those - the partition of the other subtrees may still be running, and what
        // inv holds there is left over from the previous tree
        upos.clear();
        for (int k = 0; k < m; ++k) {
            const int64_t* ak = knots[k].data() + (j >> (2 * (m - k))) * r;
            for (int i = 0; i < r; ++i) {
                const int64_t x = ak[i];
                int64_t p = -1;
                for (int l = k + 1; l <= m; ++l) {
                    const int64_t a = j >> (2 * (m - l));                      // this node's ancestor on level l (itself for l = m)
                    p = inv[l][x];
                    if (p < starts[l][a] || p >= starts[l][a + 1]) { p = -1; break; }
                }
                if (p >= 0) upos.push_back((int32_t)(p - s));
            }
        }
        std::sort(upos.begin(), upos.end());
        const int64_t nc = (e - s) - (int64_t)upos.size();        // candidates = rows of the node not used by an ancestor
        if (nc <= min_cand) { failed = 1; break; }
        const double tc1 = trace ? now() : 0.0;
        // RandomState.shuffle (_shuffle_raw): for i = n-1 .. 1 swap p[i] with p[v_i], v_i = interval(i), and the knots are p[0 .. r-1].
        // Position i is never touched again after step i, so for i >= r only the move p[v_i] = p[i] matters - and of all those
        // moves only the few that end up in p[0 .. r-1].  Resolved BACKWARDS: slot k < r wants the content of position k; the
        // LAST move into a wanted position w is the step with the smallest i >= r and v_i = w, and it carried what position i held
        // then - so the slot now wants position i, and the search goes on among the steps above i.  One ascending sweep over the
        // targets with a bitmap of the (at most r) wanted positions does it for all slots at once: a sequential read and one
        // bit test per step (the map of the root is 128 KB), against a random 4-byte store into a 4 MB array per step before.
        // The targets themselves come off the generator's word stream in the generator's order (steps n-1 .. r), branch-free.
        {
            const int64_t rr = (int64_t)std::max<int32_t>(r, 1);
            int64_t i = nc - 1;
            rng.draw_targets(vs.data(), i, rr);                               // steps i = nc-1 .. r
            if (trace) t_gen += now() - tc1;
            const size_t nwords = ((size_t)nc + 63) / 64;
            std::fill(bits.begin(), bits.begin() + nwords, (uint64_t)0);
            perm_idx.resize(rr);
            for (int64_t k = 0; k < rr && k < nc; ++k) { perm_idx[k] = (int32_t)k; bits[k >> 6] |= (uint64_t)1 << (k & 63); }
            const uint32_t* const v = vs.data();
            uint64_t* const bm = bits.data();
            int32_t* const want = perm_idx.data();
            auto hit = [&](int64_t ii) {
                const uint32_t t = v[ii];
                if (!((bm[t >> 6] >> (t & 63)) & 1)) return;
                int k = 0;
                while (want[k] != (int32_t)t) ++k;                            // a set bit always has its slot
                want[k] = (int32_t)ii;
                bm[t >> 6] &= ~((uint64_t)1 << (t & 63));
                bm[ii >> 6] |= (uint64_t)1 << (ii & 63);
            };
            int64_t ii = rr;
            for (; ii + 4 <= nc; ii += 4) {
                const uint32_t a = v[ii], b = v[ii + 1], c = v[ii + 2], d = v[ii + 3];
                const uint64_t any = ((bm[a >> 6] >> (a & 63)) | (bm[b >> 6] >> (b & 63)) | (bm[c >> 6] >> (c & 63)) | (bm[d >> 6] >> (d & 63))) & 1;
                if (any) { hit(ii); hit(ii + 1); hit(ii + 2); hit(ii + 3); }  // in order: a move may be into the position just taken up
            }
            for (; ii < nc; ++ii) hit(ii);
            // p[0 .. r-1] now: the original (= identity) content of the wanted positions; the last r-1 steps are plain swaps
            int32_t* const pp = perm_idx.data();
            for (i = rr - 1; i >= 1; --i) std::swap(pp[i], pp[rng.interval((uint64_t)i)]);
        }
        // the r picks are indices into the candidate list (ascending row order): map the k-th candidate to its row by stepping
        // over the ancestors' knots; the reference re-sorts the knots into location order (MRANode.py:203-204) = ascending pick
        for (int i = 0; i < r; ++i) picks[i] = perm_idx[i];
        std::sort(picks.begin(), picks.end());
        if (trace) { const double tc2 = now(); t_cand += tc1 - tc0; t_shuf += tc2 - tc1; }
        int64_t* kn = knots[m].data() + j * r;
        {
            size_t u = 0;
            const int32_t* om = orders[m].data() + s;
            for (int i = 0; i < r; ++i) {
                int64_t pos = picks[i] + (int64_t)u;
                while (u < upos.size() && upos[u] <= pos) { ++u; ++pos; }
                kn[i] = om[pos];
            }
        }
        for (int i = 0; i < r; ++i) used[kn[i]] = 1;
        if (m + 1 < M) for (int c = 3; c >= 0; --c) stack.push_back({m + 1, 4 * j + c});
    }
    const double t_knots = now();
    // ---- join the partition / layout (and the hook), then the knot rows
    for (auto& t : pool) if (t.joinable()) t.join();
    const double t_join = now();
    if (failed || bad.load()) return 1;
    int64_t* const kr_o = out.ext_knot_rows ? out.ext_knot_rows : out.knot_rows.data();
    int64_t nkr = 0;
    for (int m = 0; m < L; ++m) {
        for (int64_t j = 0; j < ((int64_t)1 << (2 * m)); ++j) {
            const int64_t i = out.level_ptr[m] + j;
            if (m < M) {
                const int64_t* kn = knots[m].data() + j * r;
                for (int t = 0; t < r; ++t) kr_o[nkr++] = pos_of[kn[t]];
            } else {
                const int64_t s = starts[M][j], e = starts[M][j + 1];
                for (int64_t t = s; t < e; ++t) if (!used[oM[t]]) kr_o[nkr++] = pos_of[oM[t]];
            }
            out.knot_ptr[i + 1] = nkr;
        }
    }
    out.n_knot_rows = nkr;
    if (!out.ext_perm) { out.perm.resize(out.P); out.src.resize(out.P); out.in_leaf.resize(out.P); }
    if (!out.ext_knot_rows) out.knot_rows.resize(nkr);
    const double t_nodes = now();
    rng.final_state(mt_key, mt_pos);
    if (trace) fprintf(stderr, "replay_quadtree N=%lld: partition done at %.1f ms, layout (+ hook) at %.1f ms; knots done at %.1f ms (candidates %.1f, shuffles %.1f of which target draws %.1f, waiting for the partition %.1f), join %.1f ms, knot rows %.1f ms, rng state %.1f ms\n",
                       (long long)N, t_part - t_begin, t_layout - t_begin, t_knots - t_begin, t_cand, t_shuf, t_gen, t_wait, t_join - t_knots, t_nodes - t_join, now() - t_nodes);
    return 0;
}

}  // namespace mra_topo
