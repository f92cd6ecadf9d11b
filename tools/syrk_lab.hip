// What-if lab for the blocked segmented SYRK (k_syrk_blk): config 5's grandparent fronts (464 x 464 from 16 positive
// segments of K = 32 and 4 negative ones of K = 64) on synthetic operands; the production kernel against k_gemm_nt<SET>
// (32 x 32 wave tiles) for correctness and time, then timing-only variants with one ingredient removed at a time.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/syrk_lab.hip -o tools/syrk_lab && tools/syrk_lab [problems]
#define MRA_KERNELS_TEMPLATES_ONLY
#include "../pymra_amd/csrc/mra_kernels.h"
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int VAR>
static float run_blk(const GemmProb* dp, int nprob, int M, int reps) {
    const long nbk = (M / 16 + SB_T - 1) / SB_T;
    const unsigned gx = (unsigned)(nbk * (nbk + 1) / 2);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const dim3 grid(gx * (((unsigned)nprob + 7u) / 8u) * 8u);
    hipLaunchKernelGGL((k_syrk_blk<EPI_SET, VAR>), grid, dim3(256), 0, 0, dp, gx, (unsigned)nprob, 1u);
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_syrk_blk<EPI_SET, VAR>), grid, dim3(256), 0, 0, dp, gx, (unsigned)nprob, 1u);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main(int argc, char** argv) {
    const int nprob = argc > 1 ? atoi(argv[1]) : 1024;
    const int M = 464, NPOS = 16, KPOS = 32, NNEG = 4, KNEG = 64;
    const size_t per = (size_t)M * (NPOS * KPOS + NNEG * KNEG);
    std::vector<double> h(per * nprob);
    unsigned long long st = 88172645463325252ull;
    for (auto& v : h) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; v = ((double)(st >> 11) / 9007199254740992.0 - 0.5) * 0.25; }
    double *dA, *dC0, *dC1;
    CK(hipMalloc(&dA, h.size() * 8));
    CK(hipMemcpy(dA, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    CK(hipMalloc(&dC0, (size_t)nprob * M * M * 8)); CK(hipMalloc(&dC1, (size_t)nprob * M * M * 8));
    CK(hipMemset(dC0, 0, (size_t)nprob * M * M * 8)); CK(hipMemset(dC1, 0, (size_t)nprob * M * M * 8));
    std::vector<GemmSeg> segs;
    std::vector<GemmProb> p0(nprob), p1(nprob);
    GemmSeg* dsegs;
    CK(hipMalloc(&dsegs, sizeof(GemmSeg) * (size_t)nprob * (NPOS + NNEG)));
    for (int i = 0; i < nprob; ++i) {
        double* base = dA + per * i;
        for (int s = 0; s < NPOS; ++s) { double* a = base + (size_t)s * M * KPOS; segs.push_back(GemmSeg{a, a, KPOS, KPOS, KPOS, 0}); }
        for (int s = 0; s < NNEG; ++s) { double* a = base + (size_t)NPOS * M * KPOS + (size_t)s * M * KNEG; segs.push_back(GemmSeg{a, a, KNEG, KNEG, KNEG, 1}); }
        GemmProb g{};
        g.ldc = M; g.M = M; g.N = M; g.lower = 1; g.segs = dsegs + (size_t)i * (NPOS + NNEG); g.nseg = NPOS + NNEG; g.diag_one = 64;
        g.C = dC0 + (size_t)i * M * M; p0[i] = g;
        g.C = dC1 + (size_t)i * M * M; p1[i] = g;
    }
    CK(hipMemcpy(dsegs, segs.data(), sizeof(GemmSeg) * segs.size(), hipMemcpyHostToDevice));
    GemmProb *dp0, *dp1;
    CK(hipMalloc(&dp0, sizeof(GemmProb) * nprob)); CK(hipMalloc(&dp1, sizeof(GemmProb) * nprob));
    CK(hipMemcpy(dp0, p0.data(), sizeof(GemmProb) * nprob, hipMemcpyHostToDevice));
    CK(hipMemcpy(dp1, p1.data(), sizeof(GemmProb) * nprob, hipMemcpyHostToDevice));
    // reference: 32 x 32 wave tiles
    {
        const long tm = (M + 31) / 32;
        const unsigned gx = (unsigned)((tm * (tm + 1) / 2 + 3) / 4);
        KernelParams kp{};
        const dim3 grid(gx * (((unsigned)nprob + 7u) / 8u) * 8u);
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL((k_gemm_nt<EPI_SET, 2, 0>), grid, dim3(256), 0, 0, dp0, kp, gx, (unsigned)nprob, 1u);
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_gemm_nt<EPI_SET, 2, 0>), grid, dim3(256), 0, 0, dp0, kp, gx, (unsigned)nprob, 1u);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("k_gemm_nt<SET> 32x32 wave tiles                     %8.3f ms for %d problems (%.3f ms per 4096)\n", ms / 3, nprob, ms / 3 * 4096.0 / nprob);
    }
    const double exec_flop = 435.0 * 256 * 768 * 2 * nprob;
    float t = run_blk<0>(dp1, nprob, M, 3);
    printf("k_syrk_blk production                               %8.3f ms (%.3f per 4096), %.1f TFLOP/s executed\n", t, t * 4096.0 / nprob, exec_flop / t * 1e-9);
    // compare the lower 16 x 16 sub-tiles of problem 0 and of the last one
    {
        std::vector<double> c0((size_t)M * M), c1((size_t)M * M);
        double worst = 0;
        for (int which : {0, nprob - 1}) {
            CK(hipMemcpy(c0.data(), dC0 + (size_t)which * M * M, c0.size() * 8, hipMemcpyDeviceToHost));
            CK(hipMemcpy(c1.data(), dC1 + (size_t)which * M * M, c1.size() * 8, hipMemcpyDeviceToHost));
            for (int i = 0; i < M; ++i)
                for (int j = 0; j < M; ++j)
                    if (j / 16 <= i / 16) worst = fmax(worst, fabs(c0[(size_t)i * M + j] - c1[(size_t)i * M + j]));
        }
        printf("max |k_syrk_blk - k_gemm_nt| over the lower sub-tiles: %.3e\n", worst);
    }
    {
        // the DMA-staged form: correctness against the same reference, then time
        CK(hipMemset(dC1, 0, (size_t)nprob * M * M * 8));
        const long nbk = (M / 16 + SB_T - 1) / SB_T;
        const unsigned gx = (unsigned)(nbk * (nbk + 1) / 2);
        const dim3 grid(gx * (((unsigned)nprob + 7u) / 8u) * 8u);
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL((k_syrk_dma<EPI_SET>), grid, dim3(256), 0, 0, dp1, gx, (unsigned)nprob, 1u);
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_syrk_dma<EPI_SET>), grid, dim3(256), 0, 0, dp1, gx, (unsigned)nprob, 1u);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<double> c0((size_t)M * M), c1((size_t)M * M);
        double worst = 0;
        for (int which : {0, nprob - 1}) {
            CK(hipMemcpy(c0.data(), dC0 + (size_t)which * M * M, c0.size() * 8, hipMemcpyDeviceToHost));
            CK(hipMemcpy(c1.data(), dC1 + (size_t)which * M * M, c1.size() * 8, hipMemcpyDeviceToHost));
            for (int i = 0; i < M; ++i)
                for (int j = 0; j < M; ++j)
                    if (j / 16 <= i / 16) worst = fmax(worst, fabs(c0[(size_t)i * M + j] - c1[(size_t)i * M + j]));
        }
        printf("k_syrk_dma (stage by LDS DMA, three workgroups per CU) %8.3f ms (%.3f per 4096), max |diff| %.3e\n", ms / 3, ms / 3 * 4096.0 / nprob, worst);
    }
    t = run_blk<1>(dp1, nprob, M, 3);   printf("  without barriers                                  %8.3f ms per 4096\n", t * 4096.0 / nprob);
    t = run_blk<2>(dp1, nprob, M, 3);   printf("  without the global loads of the loop              %8.3f\n", t * 4096.0 / nprob);
    t = run_blk<8>(dp1, nprob, M, 3);   printf("  without the LDS writes of the loop                %8.3f\n", t * 4096.0 / nprob);
    t = run_blk<12>(dp1, nprob, M, 3);  printf("  without LDS writes, fragments read once           %8.3f\n", t * 4096.0 / nprob);
    t = run_blk<13>(dp1, nprob, M, 3);  printf("  ... and without barriers                          %8.3f\n", t * 4096.0 / nprob);
    t = run_blk<15>(dp1, nprob, M, 3);  printf("  ... and without global loads (MFMAs only)         %8.3f\n", t * 4096.0 / nprob);
    t = run_blk<16>(dp1, nprob, M, 3);  printf("  without the MFMAs (loads, LDS writes, barriers)   %8.3f\n", t * 4096.0 / nprob);
    printf("(MFMA bound at 2.2 GHz: %.3f ms per 4096 for the 435 sub-tiles, %.3f with the 480 wave slots of the 6/5/5/5 deal)\n",
           435.0 * 768 / 4 * 64 * 4096 / 1024 / 2.2e6, 480.0 * 768 / 4 * 64 * 4096 / 1024 / 2.2e6);
    return 0;
}
