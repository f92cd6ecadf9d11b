// Micro-benchmark of the small-factorisation latency chain: one chol16_inv call, a dependent MFMA chain, and
// the leaf-C Cholesky kernels (k_chol_wave, k_panel_chol) on 1 / 512 / 4096 matrices of 7x7 tiles.
// Measured on MI355X: chol16_inv 3.3 us per call; dependent v_mfma_f64_16x16x4 73 ns each (29 ns when
// independent); one 112x112 factorisation ~60 us whichever kernel (an LDS-resident one-workgroup variant was
// no faster: 59 us) - the floor is the serial chain 7 x (diagonal block + dependent MFMA updates).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/chol_latency.hip -o tools/chol_latency && tools/chol_latency
#include "../pymra_amd/csrc/mra_kernels.h"
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(64) void k_c16(double* out, int iters) {
    __shared__ __attribute__((aligned(16))) double sc[288];
    const int lane = threadIdx.x & 63, r = lane & 15;
    double a0[16];
    for (int k = 0; k < 16; ++k) a0[k] = (k < r) ? 1.0 / (1 + r - k) : (k == r ? 20.0 : 0.0);
    double acc = 0;
    bool bad = false;
    for (int it = 0; it < iters; ++it) {
        double a[16], m[16];
        for (int k = 0; k < 16; ++k) a[k] = a0[k];
        const double lg = chol16_inv(a, m, r, bad, sc);
        acc += lg + m[3] + a[2];
        a0[r > 15 ? 0 : 15] += 1e-30 * acc;       // keep the calls dependent
    }
    out[threadIdx.x] = acc + (bad ? 1 : 0);
}

// the blocked atom (chol16_mfma): same dependent-call timing, and a correctness dump of L and L^-1 for one tile
template <int ABL, int VER = 0>
__global__ __launch_bounds__(256) void k_c16m(double* out, int iters, const double* tile_in, double* Lres, double* Ires) {
    __shared__ __attribute__((aligned(16))) double st[256], sl[256], si[256];
    const int lane = threadIdx.x & 63;
    for (int e = lane; e < 256; e += 64) st[e] = tile_in[e];
    __syncthreads();
    double acc = 0;
    bool bad = false;
    for (int it = 0; it < iters; ++it) {
        const double lg = VER ? chol16_ldl<ABL>(st, 16, sl, 16, si, 16, lane, bad) : chol16_mfma<ABL>(st, sl, si, lane, bad);
        lds_wave_sync();
        acc += lg + si[3 * 16 + 1] + sl[2 * 16 + 1];
        if (lane == 0) st[15 * 16 + 15] += 1e-30 * acc;      // keep the calls dependent
        lds_wave_sync();
    }
    for (int e = lane; e < 256; e += 64) { Lres[e] = sl[e]; Ires[e] = si[e]; }
    out[threadIdx.x] = acc + (bad ? 1 : 0);
}

__global__ __launch_bounds__(64) void k_mfma_chain(double* out, int iters) {
    d4 acc = {0, 0, 0, 0};
    double a = 1.0 + threadIdx.x * 1e-9, b = 1.0;
    for (int it = 0; it < iters; ++it) acc = mfma16(a, b, acc);
    out[threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

// dependent-chain latencies of the instructions the atom is made of (cycles from s_memtime, one wave)
template <int WHAT>
__global__ __launch_bounds__(256) void k_lat(double* out, long long* cyc, int iters) {
    double x = 1.0 + threadIdx.x * 1e-3, y = 0.999;
    d4 acc = {x, x, x, x};
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (WHAT == 0) { x = __builtin_fma(x, y, 1e-3); x = __builtin_fma(x, y, 1e-3); x = __builtin_fma(x, y, 1e-3); x = __builtin_fma(x, y, 1e-3); }
        if (WHAT == 1) { x = __builtin_amdgcn_rcp(x); x = __builtin_amdgcn_rcp(x); x = __builtin_amdgcn_rcp(x); x = __builtin_amdgcn_rcp(x); }
        if (WHAT == 2) { x = rcp_pos(x); x = rcp_pos(x); x = rcp_pos(x); x = rcp_pos(x); }
        if (WHAT == 3) { x = rsqrt_pos(x); x = rsqrt_pos(x); x = rsqrt_pos(x); x = rsqrt_pos(x); }
        if (WHAT == 4) { for (int k = 0; k < 4; ++k) x = __builtin_fma(readlane_f64(x, 17), y, 1e-3); }
        if (WHAT == 5) { for (int k = 0; k < 4; ++k) acc = mfma16(acc[0], y, acc); }
        if (WHAT == 6) { for (int k = 0; k < 4; ++k) { acc = mfma16(x, y, acc); x = __builtin_fma(acc[0], y, 1e-3); } }
        if (WHAT == 7) { for (int k = 0; k < 4; ++k) x = __builtin_fma(row_bcast(x, 3), y, 1e-3); }
        if (WHAT == 8) { for (int k = 0; k < 4; ++k) x = __hiloint2double(__builtin_amdgcn_ds_bpermute(68, __double2hiint(x)), __builtin_amdgcn_ds_bpermute(68, __double2loint(x))) * y; }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = x + acc[0] + acc[1];
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    double* dout; hipMalloc(&dout, 1 << 20);
    float ms;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0); hipLaunchKernelGGL(k_c16, dim3(1), dim3(64), 0, 0, dout, 2000); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    printf("chol16_inv: %.3f us per call (one wave, dependent calls)\n", ms * 1e3 / 2000);
    {
        // a well-conditioned and a badly scaled SPD tile: L and L^-1 against a host factorisation
        for (int which = 0; which < 2; ++which) {
            double T[256], Lh[256] = {0}, Ih[256] = {0};
            for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
                const double sc = which ? pow(10.0, (i % 5) - 2.0) * pow(10.0, (j % 5) - 2.0) : 1.0;
                T[i * 16 + j] = sc * ((i == j ? 1.5 : 0.0) + exp(-0.3 * fabs((double)i - j)));
            }
            for (int j = 0; j < 16; ++j) {
                double d = T[j * 16 + j];
                for (int k = 0; k < j; ++k) d -= Lh[j * 16 + k] * Lh[j * 16 + k];
                Lh[j * 16 + j] = sqrt(d);
                for (int i = j + 1; i < 16; ++i) { double v = T[i * 16 + j]; for (int k = 0; k < j; ++k) v -= Lh[i * 16 + k] * Lh[j * 16 + k]; Lh[i * 16 + j] = v / Lh[j * 16 + j]; }
            }
            for (int c = 0; c < 16; ++c) for (int i = c; i < 16; ++i) {
                double v = (i == c) ? 1.0 : 0.0;
                for (int k = c; k < i; ++k) v -= Lh[i * 16 + k] * Ih[k * 16 + c];
                Ih[i * 16 + c] = v / Lh[i * 16 + i];
            }
            double *dT, *dL, *dI; hipMalloc(&dT, 2048); hipMalloc(&dL, 2048); hipMalloc(&dI, 2048);
            hipMemcpy(dT, T, 2048, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(k_c16m<0>, dim3(1), dim3(64), 0, 0, dout, 1, dT, dL, dI);
            double Lg[256], Ig[256], o0; hipMemcpy(Lg, dL, 2048, hipMemcpyDeviceToHost); hipMemcpy(Ig, dI, 2048, hipMemcpyDeviceToHost); hipMemcpy(&o0, dout, 8, hipMemcpyDeviceToHost);
            double eL = 0, eI = 0, lgh = 0;
            for (int i = 0; i < 16; ++i) lgh += log(Lh[i * 16 + i]);
            for (int e = 0; e < 256; ++e) { eL = fmax(eL, fabs(Lg[e] - Lh[e]) / (fabs(Lh[e]) + 1e-300 + (Lh[e] == 0))); eI = fmax(eI, fabs(Ig[e] - Ih[e]) / (fabs(Ih[e]) + (Ih[e] == 0))); }
            printf("chol16_mfma tile %d: max rel err L %.2e, L^-1 %.2e; sum log diag host %.15g\n", which, eL, eI, lgh);
            {
                hipLaunchKernelGGL((k_c16m<0, 1>), dim3(1), dim3(64), 0, 0, dout, 1, dT, dL, dI);
                hipMemcpy(Lg, dL, 2048, hipMemcpyDeviceToHost); hipMemcpy(Ig, dI, 2048, hipMemcpyDeviceToHost); hipMemcpy(&o0, dout, 8, hipMemcpyDeviceToHost);
                eL = eI = 0;
                for (int e = 0; e < 256; ++e) { eL = fmax(eL, fabs(Lg[e] - Lh[e]) / (fabs(Lh[e]) + 1e-300 + (Lh[e] == 0))); eI = fmax(eI, fabs(Ig[e] - Ih[e]) / (fabs(Ih[e]) + (Ih[e] == 0))); }
                // out[0] = lg + si[3*16+1] + sl[2*16+1] of the one call
                printf("chol16_ldl  tile %d: max rel err L %.2e, L^-1 %.2e; sum log diag device %.15g (host %.15g)\n", which, eL, eI, o0 - Ig[3 * 16 + 1] - Lg[2 * 16 + 1], lgh);
            }
            if (which == 0) {
#define TIME_ABL(A, label) do { for (int rep = 0; rep < 2; ++rep) { hipEventRecord(e0); hipLaunchKernelGGL(k_c16m<A>, dim3(1), dim3(64), 0, 0, dout, 2000, dT, dL, dI); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); } \
                printf("chol16_mfma<%2d> %-52s %.3f us per call\n", A, label, ms * 1e3 / 2000); } while (0)
                TIME_ABL(0, "(full; one wave, dependent calls, tile/results via LDS)");
                TIME_ABL(1, "without the inverse's MFMAs");
                TIME_ABL(8, "without the log + reduction");
                TIME_ABL(9, "without inverse and log");
                TIME_ABL(11, "... and bare v_rcp_f64 instead of rcp_pos");
                TIME_ABL(15, "... and bare v_rsq_f64 instead of rsqrt_pos");
                TIME_ABL(31, "... and no v_readlane");
#define TIME_LDL(A, label) do { for (int rep = 0; rep < 2; ++rep) { hipEventRecord(e0); hipLaunchKernelGGL((k_c16m<A, 1>), dim3(1), dim3(64), 0, 0, dout, 2000, dT, dL, dI); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); } \
                printf("chol16_ldl <%2d> %-52s %.3f us per call\n", A, label, ms * 1e3 / 2000); } while (0)
                TIME_LDL(0, "(full)");
                TIME_LDL(1, "without the inverse's MFMAs");
                TIME_LDL(9, "without inverse and log");
            }
        }
    }
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0); hipLaunchKernelGGL(k_mfma_chain, dim3(1), dim3(64), 0, 0, dout, 100000); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    printf("dependent mfma_f64_16x16x4: %.1f ns each\n", ms * 1e6 / 100000);
    {
        long long* dcyc; hipMalloc(&dcyc, 8);
        const char* names[9] = {"v_fma_f64", "v_rcp_f64", "rcp_pos", "rsqrt_pos", "v_readlane pair + v_fma_f64", "v_mfma_f64_16x16x4 (acc -> A operand -> acc)",
                                "v_mfma_f64 -> v_fma_f64 -> v_mfma_f64", "DPP row broadcast pair + v_fma_f64", "ds_bpermute pair + v_mul_f64"};
#define LAT(W) do { hipLaunchKernelGGL(k_lat<W>, dim3(1), dim3(64), 0, 0, dout, dcyc, 100000); hipEventRecord(e0); hipLaunchKernelGGL(k_lat<W>, dim3(1), dim3(64), 0, 0, dout, dcyc, 100000); hipEventRecord(e1); hipEventSynchronize(e1); \
            hipEventElapsedTime(&ms, e0, e1); long long c; hipMemcpy(&c, dcyc, 8, hipMemcpyDeviceToHost); \
            printf("dependent %-48s %7.2f ns each, %7.1f s_memtime ticks each\n", names[W], ms * 1e6 / 400000.0, (double)c / 400000.0); } while (0)
        LAT(0); LAT(1); LAT(2); LAT(3); LAT(4); LAT(5); LAT(6); LAT(7); LAT(8);
    }

    const int nt = 7, n = nt * 16, nmax = 4096;
    std::vector<double> h((size_t)n * n);
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) h[(size_t)i * n + j] = (i == j ? 3.0 : 0.0) + 1.0 / (1.0 + abs(i - j));
    double *dP, *dP0, *dinv, *dnode; int* derr; PanelProb* dprob;
    hipMalloc(&dP0, (size_t)n * n * 8); hipMemcpy(dP0, h.data(), (size_t)n * n * 8, hipMemcpyHostToDevice);
    hipMalloc(&dP, (size_t)nmax * n * n * 8); hipMalloc(&dinv, (size_t)nmax * nt * 256 * 8); hipMalloc(&dnode, nmax * 8); hipMalloc(&derr, 4);
    hipMemset(derr, 0, 4);
    std::vector<PanelProb> pr(nmax);
    for (int t = 0; t < nmax; ++t) pr[t] = PanelProb{dP + (size_t)t * n * n, dinv + (size_t)t * nt * 256, n, nt, nt, t};
    hipMalloc(&dprob, nmax * sizeof(PanelProb)); hipMemcpy(dprob, pr.data(), nmax * sizeof(PanelProb), hipMemcpyHostToDevice);
    const int counts[3] = {1, 512, 4096};
    for (int which = 1; which < 3; ++which)
        for (int ci = 0; ci < 3; ++ci) {
            const int cnt = counts[ci];
            float best = 1e9;
            for (int rep = 0; rep < 3; ++rep) {
                for (int t = 0; t < cnt; ++t) hipMemcpyAsync(dP + (size_t)t * n * n, dP0, (size_t)n * n * 8, hipMemcpyDeviceToDevice, 0);
                hipDeviceSynchronize();
                hipEventRecord(e0);
                if (which == 1) hipLaunchKernelGGL((k_chol_wave<12>), dim3((cnt + 3) / 4), dim3(256), 0, 0, dprob, cnt, dnode, derr);
                else hipLaunchKernelGGL(k_panel_chol, dim3(cnt), dim3(256), 0, 0, dprob, dnode, derr);
                hipEventRecord(e1); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            double d0; hipMemcpy(&d0, dnode, 8, hipMemcpyDeviceToHost);
            printf("%-13s %5d matrices of %dx%d: %8.1f us   (logdet %.12f)\n", which == 1 ? "k_chol_wave" : "k_panel_chol", cnt, n, n, best * 1e3, d0);
        }
    int e; hipMemcpy(&e, derr, 4, hipMemcpyDeviceToHost);
    printf("err flag %d\n", e);
    return 0;
}
