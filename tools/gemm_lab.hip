// Batched-GEMM laboratory: the production kernels (included from pymra_amd/csrc/mra_kernels.h) on synthetic batches of the
// leaf shapes, with K as a free parameter: how much of the gap to the MFMA rate is the short K loop (prologue / epilogue /
// workgroup turnover) and how much the K-step structure itself?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm_lab.hip -o tools/gemm_lab && tools/gemm_lab
#include "../pymra_amd/csrc/mra_kernels.h"
#include <cstdio>
#include <vector>

template <class T> T* dalloc(size_t n) { T* p = nullptr; if (hipMalloc((void**)&p, n * sizeof(T)) != hipSuccess) { printf("hipMalloc failed (%zu bytes)\n", n * sizeof(T)); exit(1); } return p; }

__global__ void k_fill(double* p, size_t n, unsigned long long seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long x = (i + 1) * 0x9E3779B97F4A7C15ull + seed;
        x ^= x >> 31; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 29;
        p[i] = (double)(x >> 11) / 9007199254740992.0 - 0.5;
    }
}

int main() {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    KernelParams kp{};
    struct Shape { const char* name; int nprob, M, N, K; bool sub; };
    const Shape shapes[] = {
        {"leaf update shape  M=256 N=208 K=112 (SUB)", 4096, 256, 208, 112, true},
        {"leaf update shape  M=256 N=208 K=448 (SUB)", 2048, 256, 208, 448, true},
        {"leaf update shape  M=256 N=208 K=1792 (SUB)", 512, 256, 208, 1792, true},
        {"leaf resid shape   M=256 N=112 K=192 (SET)", 4096, 256, 112, 192, false},
        {"leaf resid shape   M=256 N=128 K=192 (SET)", 4096, 256, 128, 192, false},
        {"leaf resid shape   M=256 N=128 K=768 (SET)", 2048, 256, 128, 768, false},
        {"leaf resid shape   M=256 N=128 K=3072 (SET)", 512, 256, 128, 3072, false},
        {"square             M=256 N=256 K=4096 (SET)", 256, 256, 256, 4096, false},
    };
    for (const Shape& sh : shapes) {
        const size_t na = (size_t)sh.M * sh.K, nb = (size_t)sh.N * sh.K, nc = (size_t)sh.M * sh.N;
        double* A = dalloc<double>(na * sh.nprob);
        double* B = dalloc<double>(nb * sh.nprob);
        double* C = dalloc<double>(nc * sh.nprob);
        hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, A, na * sh.nprob, 1ull);
        hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, B, nb * sh.nprob, 2ull);
        hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, C, nc * sh.nprob, 3ull);
        std::vector<GemmProb> h(sh.nprob);
        for (int p = 0; p < sh.nprob; ++p) {
            GemmProb g{};
            g.A = A + na * p; g.lda = sh.K; g.B = B + nb * p; g.ldb = sh.K; g.C = C + nc * p; g.ldc = sh.N;
            g.M = sh.M; g.N = sh.N; g.K = sh.K;
            h[p] = g;
        }
        GemmProb* dp = dalloc<GemmProb>(sh.nprob);
        hipMemcpy(dp, h.data(), sh.nprob * sizeof(GemmProb), hipMemcpyHostToDevice);
        const unsigned gx_lds = (unsigned)(((sh.M + 63) / 64) * ((sh.N + 63) / 64));
        const unsigned gx_dir = (unsigned)((((sh.M + 31) / 32) * ((sh.N + 31) / 32) + 3) / 4);
        const unsigned gy8 = ((sh.nprob + 7) / 8) * 8;
        float ms;
        for (int which = 0; which < 2; ++which) {
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                if (which == 0) {
                    if (sh.sub) hipLaunchKernelGGL((k_gemm_nt_lds<EPI_SUB, 2, 0>), dim3(gx_lds * gy8), dim3(256), 0, 0, dp, kp, gx_lds, (unsigned)sh.nprob);
                    else hipLaunchKernelGGL((k_gemm_nt_lds<EPI_SET, 2, 0>), dim3(gx_lds * gy8), dim3(256), 0, 0, dp, kp, gx_lds, (unsigned)sh.nprob);
                } else {
                    if (sh.sub) hipLaunchKernelGGL((k_gemm_nt<EPI_SUB, 2, 0>), dim3(gx_dir * gy8), dim3(256), 0, 0, dp, kp, gx_dir, (unsigned)sh.nprob);
                    else hipLaunchKernelGGL((k_gemm_nt<EPI_SET, 2, 0>), dim3(gx_dir * gy8), dim3(256), 0, 0, dp, kp, gx_dir, (unsigned)sh.nprob);
                }
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            hipEventElapsedTime(&ms, e0, e1);
            const double fl = 2.0 * sh.M * sh.N * (double)sh.K * sh.nprob;
            const double bytes = 8.0 * sh.nprob * ((double)na + nb + nc * (sh.sub ? 2.0 : 1.0));
            printf("%-46s %-14s %8.3f ms %7.2f TFLOP/s  (%.2f TB/s of compulsory traffic)\n", sh.name, which ? "k_gemm_nt" : "k_gemm_nt_lds", ms, fl / ms / 1e9, bytes / ms / 1e9);
        }
        hipFree(A); hipFree(B); hipFree(C); hipFree(dp);
    }
    // ---- the grandparents' signed SYRK of config 5: M = N = 464 (lower), K = 768 as 16 leaf segments of 32 + 4 child segments of 64,
    //      against the same K in fewer, longer segments: is the segment structure what holds it at 33 TFLOP/s?
    {
        const int nprob = 1024, M = 464, Ktot = 768;
        double* A = dalloc<double>((size_t)nprob * M * Ktot);
        double* C = dalloc<double>((size_t)nprob * M * M);
        hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, A, (size_t)nprob * M * Ktot, 5ull);
        struct Var { const char* name; std::vector<int> ks; };
        std::vector<Var> vars;
        { Var v{"20 segments (16 x 32 + 4 x 64)", {}}; for (int i = 0; i < 16; ++i) v.ks.push_back(32); for (int i = 0; i < 4; ++i) v.ks.push_back(64); vars.push_back(v); }
        { Var v{"5 segments (512 + 4 x 64)", {512, 64, 64, 64, 64}}; vars.push_back(v); }
        { Var v{"1 segment (768)", {768}}; vars.push_back(v); }
        { Var v{"48 segments of 16", {}}; for (int i = 0; i < 48; ++i) v.ks.push_back(16); vars.push_back(v); }
        for (const Var& v : vars) {
            std::vector<GemmSeg> hs;
            std::vector<GemmProb> hp(nprob);
            for (int p = 0; p < nprob; ++p) {
                double* base = A + (size_t)p * M * Ktot;
                size_t off = 0;
                const size_t first = hs.size();
                for (int k : v.ks) { hs.push_back(GemmSeg{base + off, base + off, k, k, k, 0}); off += (size_t)M * k; }      // each segment its own M x k block
                GemmProb g{};
                g.C = C + (size_t)p * M * M; g.ldc = M; g.M = M; g.N = M; g.lower = 1; g.nseg = (int)v.ks.size();
                g.A = g.B = base; g.K = 0;
                hp[p] = g;
                hp[p].segs = (const GemmSeg*)first;                 // index for now, pointer below
            }
            GemmSeg* ds = dalloc<GemmSeg>(hs.size());
            hipMemcpy(ds, hs.data(), hs.size() * sizeof(GemmSeg), hipMemcpyHostToDevice);
            for (int p = 0; p < nprob; ++p) hp[p].segs = ds + (size_t)hp[p].segs;
            GemmProb* dp = dalloc<GemmProb>(nprob);
            hipMemcpy(dp, hp.data(), nprob * sizeof(GemmProb), hipMemcpyHostToDevice);
            const long tm = (M + 31) / 32;
            const unsigned gx_dir = (unsigned)((tm * (tm + 1) / 2 + 3) / 4);
            const unsigned gx_lds = (unsigned)(((M + 63) / 64) * ((M + 63) / 64));
            for (int which = 0; which < 2; ++which) {
                float ms = 0;
                for (int rep = 0; rep < 3; ++rep) {
                    hipEventRecord(e0);
                    if (which == 0) hipLaunchKernelGGL((k_gemm_nt<EPI_SET, 2, 0>), dim3(gx_dir * nprob), dim3(256), 0, 0, dp, kp, gx_dir, (unsigned)nprob, 1u);
                    else hipLaunchKernelGGL((k_gemm_nt_lds<EPI_SET, 2, 0>), dim3(gx_lds * nprob), dim3(256), 0, 0, dp, kp, gx_lds, (unsigned)nprob, 1u);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    hipEventElapsedTime(&ms, e0, e1);
                }
                const double fl = (double)M * M * Ktot * nprob;      // lower triangle: half of 2 M N K
                printf("SYRK 464 lower, K = 768: %-34s %-14s %8.3f ms %7.2f TFLOP/s (triangle count)\n", v.name, which ? "k_gemm_nt_lds" : "k_gemm_nt", ms, fl / ms / 1e9);
            }
            hipFree(ds); hipFree(dp);
        }
        hipFree(A); hipFree(C);
    }
    return 0;
}
