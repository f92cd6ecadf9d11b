// Micro-benchmarks that settle the FP64 ceiling of this MI355X and calibrate the HBM PMC counters
// (the guide lists no FP64 row and calibrates FETCH_SIZE only for 16 B/lane streams).
//
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_peak.hip -o tools/mfma_f64_peak
//   tools/mfma_f64_peak                 # MFMA / VALU rates with in-kernel clocks (s_memtime / s_memrealtime)
//   tools/mfma_f64_peak stream          # known-size streaming kernels only (run under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE)
//
// In-kernel clock = delta(s_memtime) / delta(s_memrealtime) x 100 MHz (MI355X_MICROARCH.md, DVFS give-back item 6);
// cycles per MFMA per SIMD = delta(s_memtime) / (MFMAs the wave issued x waves per SIMD).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

struct Stamp { unsigned long long t0, t1, r0, r1; };

// NACC independent accumulators per wave; VFMA extra independent v_fma_f64 per MFMA in the same wave (co-issue test)
template <int NACC, int VFMA>
__global__ __launch_bounds__(256, 2) void k_mfma(double* out, const double* __restrict__ in, int iters, Stamp* st) {
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    double a = in[gid & 4095], b = in[(gid * 7 + 13) & 4095];
    double v[VFMA > 0 ? VFMA : 1];
    for (int i = 0; i < VFMA; ++i) v[i] = in[(gid + i) & 4095];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < VFMA; ++j) v[j] = __builtin_fma(v[j], 0.999999, 1.0e-9);
        }
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < VFMA; ++i) s += v[i];
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[gid] = s;
    if ((threadIdx.x & 63) == 0) st[gid >> 6] = Stamp{t0, t1, r0, r1};
}

// as k_mfma<4,0> but every MFMA of a round reads a different (A, B) register pair, like a 2x2 register-blocked GEMM tile
// (a0 b0, a0 b1, a1 b0, a1 b1): with the SAME two source registers in every instruction the pipe issues only every ~98 cycles,
// with varying sources every ~64-70 - the register-blocked form is the one real kernels have
__global__ __launch_bounds__(256, 2) void k_mfma_blk(double* out, const double* __restrict__ in, int iters, Stamp* st) {
    d4 c00 = {0, 0, 0, 0}, c01 = c00, c10 = c00, c11 = c00;
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const double a0 = in[gid & 4095], b0 = in[(gid * 7 + 13) & 4095], a1 = in[(gid * 3 + 1) & 4095], b1 = in[(gid * 5 + 2) & 4095];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        c00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, c00, 0, 0, 0);
        c01 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, c01, 0, 0, 0);
        c10 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, c10, 0, 0, 0);
        c11 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, c11, 0, 0, 0);
    }
    const d4 s = c00 + c01 + c10 + c11;
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[gid] = s[0] + s[1] + s[2] + s[3];
    if ((threadIdx.x & 63) == 0) st[gid >> 6] = Stamp{t0, t1, r0, r1};
}
// one accumulator, alternating source registers: the dependent-chain rate of real code
__global__ __launch_bounds__(256, 2) void k_mfma_dep(double* out, const double* __restrict__ in, int iters, Stamp* st) {
    d4 c = {0, 0, 0, 0};
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const double a0 = in[gid & 4095], b0 = in[(gid * 7 + 13) & 4095], a1 = in[(gid * 3 + 1) & 4095], b1 = in[(gid * 5 + 2) & 4095];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        c = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, c, 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[gid] = c[0] + c[1] + c[2] + c[3];
    if ((threadIdx.x & 63) == 0) st[gid >> 6] = Stamp{t0, t1, r0, r1};
}
// two accumulators, alternating
__global__ __launch_bounds__(256, 2) void k_mfma_dep2(double* out, const double* __restrict__ in, int iters, Stamp* st) {
    d4 c = {0, 0, 0, 0}, e = c;
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const double a0 = in[gid & 4095], b0 = in[(gid * 7 + 13) & 4095], a1 = in[(gid * 3 + 1) & 4095], b1 = in[(gid * 5 + 2) & 4095];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        c = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, c, 0, 0, 0);
        e = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, e, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, c, 0, 0, 0);
        e = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, e, 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    const d4 s = c + e;
    out[gid] = s[0] + s[1] + s[2] + s[3];
    if ((threadIdx.x & 63) == 0) st[gid >> 6] = Stamp{t0, t1, r0, r1};
}

// v_mfma_f64_4x4x4_4b_f64: four independent 4x4x4 products per instruction (512 flop), one f64 per lane for A, B and D
template <int NACC>
__global__ __launch_bounds__(256, 2) void k_mfma4(double* out, const double* __restrict__ in, int iters, Stamp* st) {
    double acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = 0.0;
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    double a = in[gid & 4095], b = in[(gid * 7 + 13) & 4095];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i];
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[gid] = s;
    if ((threadIdx.x & 63) == 0) st[gid >> 6] = Stamp{t0, t1, r0, r1};
}

// VALU only: 16 independent v_fma_f64 chains
__global__ __launch_bounds__(256, 2) void k_vfma(double* out, const double* __restrict__ in, int iters, Stamp* st) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    double acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = in[(gid + i) & 4095];
    const double a = 0.9999999, b = 1e-9;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_fma(a, acc[i], b);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[gid] = s;
    if ((threadIdx.x & 63) == 0) st[gid >> 6] = Stamp{t0, t1, r0, r1};
}

// mixed workgroup: waves 0..3 (one per SIMD) run MFMA only, waves 4..7 run VALU fma only: do the two pipes add up?
__global__ __launch_bounds__(512, 2) void k_split(double* out, const double* __restrict__ in, int iters, Stamp* st) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int wave = threadIdx.x >> 6;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
    if (wave < 4) {
        d4 acc[4];
        for (int i = 0; i < 4; ++i) acc[i] = d4{0, 0, 0, 0};
        double a = in[gid & 4095], b = in[(gid * 7 + 13) & 4095];
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else {
        double acc[16];
        for (int i = 0; i < 16; ++i) acc[i] = in[(gid + i) & 4095];
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_fma(0.9999999, acc[i], 1e-9);
        }
        for (int i = 0; i < 16; ++i) s += acc[i];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[gid] = s;
    if ((threadIdx.x & 63) == 0) st[gid >> 6] = Stamp{t0, t1, r0, r1};
}

// ---- known-size streams (PMC calibration): every lane reads/writes BYTES contiguous bytes per step, grid-stride
template <int BYTES>
__global__ __launch_bounds__(256) void k_read(const double* __restrict__ in, double* out, size_t n_elems) {
    constexpr int E = BYTES / 8;
    typedef double vt __attribute__((ext_vector_type(E)));
    const size_t nvec = n_elems / E;
    double s = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) {
        const vt v = *(const vt __attribute__((address_space(1)))*)(in + i * E);
#pragma unroll
        for (int e = 0; e < E; ++e) s += v[e];
    }
    if (s == 1.2345e300) out[0] = s;
}
template <>
__global__ __launch_bounds__(256) void k_read<8>(const double* __restrict__ in, double* out, size_t n_elems) {
    double s = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_elems; i += (size_t)gridDim.x * blockDim.x) s += in[i];
    if (s == 1.2345e300) out[0] = s;
}
// the access shape of the row kernels: 16 rows x 4 lanes per row, 32 B per lane, row stride ld doubles (W tiles)
__global__ __launch_bounds__(256) void k_read_rowtile(const double* __restrict__ in, double* out, size_t rows, int ld, int ncol) {
    const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwave = ((size_t)gridDim.x * blockDim.x) >> 6;
    double s = 0;
    for (size_t t = wave; t < rows / 16; t += nwave) {
        const double* p = in + (t * 16 + r) * (size_t)ld + 4 * q;
        for (int c = 0; c < ncol; c += 16) {
            const d4 v = *(const d4 __attribute__((address_space(1)))*)(p + c);
            s += v[0] + v[1] + v[2] + v[3];
        }
    }
    if (s == 1.2345e300) out[0] = s;
}
template <int BYTES>
__global__ __launch_bounds__(256) void k_write(double* out, size_t n_elems) {
    constexpr int E = BYTES / 8;
    typedef double vt __attribute__((ext_vector_type(E)));
    const size_t nvec = n_elems / E;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) {
        vt v;
#pragma unroll
        for (int e = 0; e < E; ++e) v[e] = (double)(i + e);
        *(vt __attribute__((address_space(1)))*)(out + i * E) = v;
    }
}

static void report(const char* name, const std::vector<Stamp>& st, double flop_per_wave, double mfma_per_wave, int waves_per_simd, float ms, double total_flop) {
    std::vector<double> clk, cyc;
    for (const Stamp& s : st) {
        if (s.r1 <= s.r0) continue;
        clk.push_back((double)(s.t1 - s.t0) / (double)(s.r1 - s.r0) * 0.1);       // GHz
        cyc.push_back((double)(s.t1 - s.t0));
    }
    std::sort(clk.begin(), clk.end()); std::sort(cyc.begin(), cyc.end());
    const double mclk = clk[clk.size() / 2], mcyc = cyc[cyc.size() / 2];
    printf("%-64s %7.2f TFLOP/s  %.3f ms  clock %.3f GHz (min %.3f max %.3f)", name, total_flop / ms / 1e9, ms, mclk, clk.front(), clk.back());
    if (mfma_per_wave > 0) printf("  %.1f cycles/MFMA/SIMD", mcyc / (mfma_per_wave * waves_per_simd));
    printf("\n");
}

int main(int argc, char** argv) {
    const bool stream_only = argc > 1 && !strcmp(argv[1], "stream");
    double *d, *din;
    Stamp* st;
    const size_t nthreads_max = 256 * 8 * 256;
    hipMalloc(&d, nthreads_max * 8);
    hipMalloc(&din, 4096 * 8);
    hipMalloc(&st, (nthreads_max / 64) * sizeof(Stamp));
    {
        std::vector<double> h(4096);
        unsigned long long x = 88172645463325252ull;
        for (auto& v : h) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = (double)(x >> 11) / 9007199254740992.0 * 2.0 - 1.0; }
        hipMemcpy(din, h.data(), 4096 * 8, hipMemcpyHostToDevice);
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
    if (!stream_only) {
        const int iters = 40000;
        auto run = [&](auto kern, int blocks, int threads, const char* name, double mfma_per_wave, double flop_per_thread_iter, int wps, int reps) {
            for (int rep = 0; rep < reps; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, din, iters, st);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            hipEventElapsedTime(&ms, e0, e1);
            std::vector<Stamp> h((size_t)blocks * threads / 64);
            hipMemcpy(h.data(), st, h.size() * sizeof(Stamp), hipMemcpyDeviceToHost);
            report(name, h, 0, mfma_per_wave, wps, ms, (double)blocks * threads * iters * flop_per_thread_iter);
        };
        // warm the chip for ~2 s so that the clocks below are the sustained ones
        for (int i = 0; i < 40; ++i) hipLaunchKernelGGL((k_mfma<4, 0>), dim3(256 * 8), dim3(256), 0, 0, d, din, iters, st);
        hipDeviceSynchronize();
        for (int wpc = 1; wpc <= 8; wpc *= 2) {
            char nm[96]; snprintf(nm, sizeof nm, "mfma_f64_16x16x4, %d waves/SIMD, 4 accumulators", wpc);
            run(k_mfma<4, 0>, 256 * wpc, 256, nm, 4.0 * iters, 4 * 2048.0 / 64, wpc, 3);
        }
        run(k_mfma<8, 0>, 256 * 2, 256, "mfma_f64_16x16x4, 2 waves/SIMD, 8 accumulators", 8.0 * iters, 8 * 2048.0 / 64, 2, 3);
        run(k_mfma<1, 0>, 256, 256, "mfma_f64_16x16x4, 1 wave/SIMD, 1 dependent accumulator", 1.0 * iters, 2048.0 / 64, 1, 3);
        run(k_mfma<1, 0>, 256 * 4, 256, "mfma_f64_16x16x4, 4 waves/SIMD, 1 dependent accumulator", 1.0 * iters, 2048.0 / 64, 4, 3);
        run(k_mfma<2, 0>, 256, 256, "mfma_f64_16x16x4, 1 wave/SIMD, 2 accumulators", 2.0 * iters, 2 * 2048.0 / 64, 1, 3);
        run(k_mfma_blk, 256, 256, "mfma_f64_16x16x4 2x2 register block, 1 wave/SIMD", 4.0 * iters, 4 * 2048.0 / 64, 1, 3);
        run(k_mfma_blk, 256 * 2, 256, "mfma_f64_16x16x4 2x2 register block, 2 waves/SIMD", 4.0 * iters, 4 * 2048.0 / 64, 2, 3);
        run(k_mfma_blk, 256 * 4, 256, "mfma_f64_16x16x4 2x2 register block, 4 waves/SIMD", 4.0 * iters, 4 * 2048.0 / 64, 4, 3);
        run(k_mfma_dep, 256, 256, "mfma_f64_16x16x4 ONE accumulator, varying sources, 1 wave/SIMD", 4.0 * iters, 4 * 2048.0 / 64, 1, 3);
        run(k_mfma_dep, 256 * 2, 256, "mfma_f64_16x16x4 ONE accumulator, varying sources, 2 waves/SIMD", 4.0 * iters, 4 * 2048.0 / 64, 2, 3);
        run(k_mfma_dep2, 256, 256, "mfma_f64_16x16x4 TWO accumulators, varying sources, 1 wave/SIMD", 4.0 * iters, 4 * 2048.0 / 64, 1, 3);
        run(k_mfma_dep2, 256 * 2, 256, "mfma_f64_16x16x4 TWO accumulators, varying sources, 2 waves/SIMD", 4.0 * iters, 4 * 2048.0 / 64, 2, 3);
        run(k_mfma4<8>, 256 * 4, 256, "mfma_f64_4x4x4_4b, 4 waves/SIMD, 8 accumulators", 8.0 * iters, 8 * 512.0 / 64, 4, 3);
        run(k_mfma4<8>, 256, 256, "mfma_f64_4x4x4_4b, 1 wave/SIMD, 8 accumulators", 8.0 * iters, 8 * 512.0 / 64, 1, 3);
        run(k_vfma, 256 * 8, 256, "v_fma_f64 only, 8 waves/SIMD, 16 chains", 0, 32.0, 8, 3);
        run(k_vfma, 256 * 2, 256, "v_fma_f64 only, 2 waves/SIMD, 16 chains", 0, 32.0, 2, 3);
        // co-issue inside one wave: per MFMA 2 / 4 / 8 independent v_fma_f64
        run(k_mfma<4, 2>, 256 * 4, 256, "mfma + 2 v_fma_f64 per MFMA (same wave), 4 waves/SIMD", 4.0 * iters, 4 * (2048.0 / 64 + 2 * 2.0), 4, 3);
        run(k_mfma<4, 4>, 256 * 4, 256, "mfma + 4 v_fma_f64 per MFMA (same wave), 4 waves/SIMD", 4.0 * iters, 4 * (2048.0 / 64 + 4 * 2.0), 4, 3);
        run(k_mfma<4, 8>, 256 * 4, 256, "mfma + 8 v_fma_f64 per MFMA (same wave), 4 waves/SIMD", 4.0 * iters, 4 * (2048.0 / 64 + 8 * 2.0), 4, 3);
        // co-issue across waves: 4 MFMA waves + 4 VALU waves per CU (x2 workgroups per CU)
        {
            const int blocks = 256 * 2;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(k_split, dim3(blocks), dim3(512), 0, 0, d, din, iters, st);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            hipEventElapsedTime(&ms, e0, e1);
            std::vector<Stamp> h((size_t)blocks * 8);
            hipMemcpy(h.data(), st, h.size() * sizeof(Stamp), hipMemcpyDeviceToHost);
            const double fl_m = (double)blocks * 4 * iters * 4 * 2048.0, fl_v = (double)blocks * 256 * iters * 32.0;
            report("split: 2x(4 MFMA waves + 4 VALU waves)/CU  [MFMA flops only]", h, 0, 0, 1, ms, fl_m);
            printf("%-64s %7.2f TFLOP/s MFMA + %.2f TFLOP/s VALU in the same %.3f ms (both halves finish together only if balanced)\n", "", fl_m / ms / 1e9, fl_v / ms / 1e9, ms);
        }
    }
    // ---- streams: 2 GiB buffer (>> 256 MiB Infinity Cache)
    {
        const size_t n = (size_t)1 << 28;      // doubles = 2 GiB
        double* big; hipMalloc(&big, n * 8);
        hipMemset(big, 0, n * 8);
        hipDeviceSynchronize();
        const int blocks = 256 * 8;
        auto timeit = [&](auto launch, const char* name, double bytes) {
            launch(); hipDeviceSynchronize();
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            printf("%-40s %.0f bytes  %.3f ms  %.2f TB/s\n", name, bytes, ms, bytes / ms / 1e9);
        };
        timeit([&] { hipLaunchKernelGGL(k_read<8>, dim3(blocks), dim3(256), 0, 0, big, d, n); }, "k_read<8>   (8 B/lane)", n * 8.0);
        timeit([&] { hipLaunchKernelGGL(k_read<16>, dim3(blocks), dim3(256), 0, 0, big, d, n); }, "k_read<16>  (16 B/lane)", n * 8.0);
        timeit([&] { hipLaunchKernelGGL(k_read<32>, dim3(blocks), dim3(256), 0, 0, big, d, n); }, "k_read<32>  (32 B/lane)", n * 8.0);
        // W-like row tiles: ld = 208 doubles, all 208 columns read (13 x 32 B per lane)
        const size_t rows = (n / 208) / 16 * 16;
        timeit([&] { hipLaunchKernelGGL(k_read_rowtile, dim3(blocks), dim3(256), 0, 0, big, d, rows, 208, 208); }, "k_read_rowtile (32 B/lane, ld 208)", rows * 208 * 8.0);
        timeit([&] { hipLaunchKernelGGL(k_write<8>, dim3(blocks), dim3(256), 0, 0, big, n); }, "k_write<8>  (8 B/lane)", n * 8.0);
        timeit([&] { hipLaunchKernelGGL(k_write<16>, dim3(blocks), dim3(256), 0, 0, big, n); }, "k_write<16> (16 B/lane)", n * 8.0);
        timeit([&] { hipLaunchKernelGGL(k_write<32>, dim3(blocks), dim3(256), 0, 0, big, n); }, "k_write<32> (32 B/lane)", n * 8.0);
        hipFree(big);
    }
    return 0;
}
