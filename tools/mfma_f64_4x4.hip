// v_mfma_f64_4x4x4_4b_f64 study (tools/mfma_f64_peak.hip found it runs at the datasheet FP64 rate, 1.5x the
// 16x16x4 form):
//   probe   the lane layout of A, B and D, found by one-hot operands
//   tile    LDS-fed 64x64-per-workgroup GEMM inner loops, 32x32 per wave, on both instruction forms
// hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_4x4.hip -o tools/mfma_f64_4x4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void k_probe(int* out) {
    // out[la * 64 + lb] = lane (or -1 / -2 for none / several) that received a non-zero D
    const int lane = threadIdx.x;
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
            const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
            const unsigned long long m = __ballot(d != 0.0);
            if (lane == 0) out[la * 64 + lb] = m == 0 ? -1 : (__popcll(m) == 1 ? __ffsll((long long)m) - 1 : -2);
        }
}

#ifndef LD
#define LD 20
#endif
// LD: doubles per staged row of 16 k: 160 B -> 32-byte fragment reads of 8 rows hit 8 disjoint bank groups */
// 16x16x4 form: the inner loop of k_gemm_nt_lds (2 A + 2 B fragment reads of 32 B, 16 MFMAs per 16 k)
__global__ __launch_bounds__(256, 2) void k_tile16(double* out, const double* __restrict__ in, int ksteps, int reps) {
    __shared__ __attribute__((aligned(16))) double sA[64 * LD], sB[64 * LD];
    for (int e = threadIdx.x; e < 64 * LD; e += 256) { sA[e] = in[e & 4095]; sB[e] = in[(e * 7 + 3) & 4095]; }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    const int wr0 = (wave >> 1) * 32, wc0 = (wave & 1) * 32;
    const d4 zero = {0, 0, 0, 0};
    d4 c00 = zero, c01 = zero, c10 = zero, c11 = zero;
    const int a0o = (wr0 + r) * LD + 4 * q, b0o = (wc0 + r) * LD + 4 * q;
    for (int rep = 0; rep < reps; ++rep)
        for (int ks = 0; ks < ksteps; ++ks) {
            const d4 a0 = *(const d4*)(&sA[a0o]), a1 = *(const d4*)(&sA[a0o + 16 * LD]);
            const d4 b0 = *(const d4*)(&sB[b0o]), b1 = *(const d4*)(&sB[b0o + 16 * LD]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                c00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[j], b0[j], c00, 0, 0, 0);
                c01 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[j], b1[j], c01, 0, 0, 0);
                c10 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[j], b0[j], c10, 0, 0, 0);
                c11 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[j], b1[j], c11, 0, 0, 0);
            }
            asm volatile("" ::: "memory");
        }
    const d4 s = c00 + c01 + c10 + c11;
    out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

// 4x4x4_4b form, same 32x32 wave tile: per 16 k, 2 A arrangements (row blocks 0-3 / 4-7 over the four 16-lane groups) and 8 B
// arrangements (column block bj broadcast to all four groups), 64 MFMAs into 16 accumulators.
// Lane (g = lane >> 4, t = lane & 15): A element = A[4 g + (t & 3)][4 (t >> 2) + s] for k-step s  (32 contiguous bytes per lane);
// B element = B[4 bj + (t & 3)][4 (t >> 2) + s]; which of (t & 3, t >> 2) is the row and which the k index is the probe's business -
// for the rate it does not matter.
__global__ __launch_bounds__(256, 2) void k_tile4(double* out, const double* __restrict__ in, int ksteps, int reps) {
    __shared__ __attribute__((aligned(16))) double sA[64 * LD], sB[64 * LD];
    for (int e = threadIdx.x; e < 64 * LD; e += 256) { sA[e] = in[e & 4095]; sB[e] = in[(e * 7 + 3) & 4095]; }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4, t = lane & 15;
    const int wr0 = (wave >> 1) * 32, wc0 = (wave & 1) * 32;
    double acc[2][8];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = 0.0;
    const int ao = (wr0 + 4 * g + (t & 3)) * LD + 4 * (t >> 2), bo = (wc0 + (t & 3)) * LD + 4 * (t >> 2);
    for (int rep = 0; rep < reps; ++rep)
        for (int ks = 0; ks < ksteps; ++ks) {
            const d4 a0 = *(const d4*)(&sA[ao]), a1 = *(const d4*)(&sA[ao + 16 * LD]);
            d4 b[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) b[j] = *(const d4*)(&sB[bo + 4 * j * LD]);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    acc[0][j] = __builtin_amdgcn_mfma_f64_4x4x4f64(a0[s], b[j][s], acc[0][j], 0, 0, 0);
                    acc[1][j] = __builtin_amdgcn_mfma_f64_4x4x4f64(a1[s], b[j][s], acc[1][j], 0, 0, 0);
                }
            asm volatile("" ::: "memory");
        }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) s += acc[i][j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// ---- what does each ingredient of the LDS-GEMM K-step cost?  Same 64x64 workgroup tile / 32x32 wave tile as k_tile16, plus
//   STAGE 1: one __syncthreads per K-step (double-buffered LDS image, nothing written)
//   STAGE 2: + every thread writes its 32-byte chunk of A and of B into the other buffer each step (ds_write)
//   STAGE 3: + those chunks come from global memory (an L2-resident panel), loaded two steps ahead
//   BK: K-values per barrier (16: one 16-MFMA group per barrier as in k_gemm_nt_lds; 32: two groups)
template <int STAGE, int BK>
__global__ __launch_bounds__(256, 4) void k_tile16s(double* out, const double* __restrict__ in, const double* __restrict__ panel, int ksteps, int reps) {
    constexpr int NG = BK / 16;
    __shared__ __attribute__((aligned(16))) double sA[2][NG][64 * LD], sB[2][NG][64 * LD];
    for (int e = threadIdx.x; e < 2 * NG * 64 * LD; e += 256) { (&sA[0][0][0])[e] = in[e & 4095]; (&sB[0][0][0])[e] = in[(e * 7 + 3) & 4095]; }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    const int wr0 = (wave >> 1) * 32, wc0 = (wave & 1) * 32;
    const d4 zero = {0, 0, 0, 0};
    d4 c00 = zero, c01 = zero, c10 = zero, c11 = zero;
    const int a0o = (wr0 + r) * LD + 4 * q, b0o = (wc0 + r) * LD + 4 * q;
    const int srow = threadIdx.x >> 2, sch = (threadIdx.x & 3) << 2;
    // global source: row (block & 7, srow) of a [rows][K] panel; 8 x 64 rows x 1024 k x 8 B = 4 MiB per operand: L2-resident, re-read
    const double* ap = panel + ((size_t)(blockIdx.x & 7) * 64 + srow) * 4096 + sch;
    const double* bp = ap + 256 * 64 * 4096;
    d4 ra[2][NG], rb[2][NG];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < NG; ++g) { ra[i][g] = zero; rb[i][g] = zero; }
    if (STAGE >= 3) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int g = 0; g < NG; ++g) { ra[i][g] = *(const d4*)(ap + (i * NG + g) * 16); rb[i][g] = *(const d4*)(bp + (i * NG + g) * 16); }
    }
    const int nst = ksteps * 16 / BK;
    for (int rep = 0; rep < reps; ++rep)
        for (int ks0 = 0; ks0 < nst; ks0 += 2) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int ks = ks0 + i, cur = i & 1;
                if (STAGE >= 3) {
                    const int ko = ((ks + 2) * BK) & 1023;
#pragma unroll
                    for (int g = 0; g < NG; ++g) { ra[i][g] = *(const d4*)(ap + ko + g * 16); rb[i][g] = *(const d4*)(bp + ko + g * 16); }
                }
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const d4 a0 = *(const d4*)(&sA[cur][g][a0o]), a1 = *(const d4*)(&sA[cur][g][a0o + 16 * LD]);
                    const d4 b0 = *(const d4*)(&sB[cur][g][b0o]), b1 = *(const d4*)(&sB[cur][g][b0o + 16 * LD]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        c00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[j], b0[j], c00, 0, 0, 0);
                        c01 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[j], b1[j], c01, 0, 0, 0);
                        c10 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[j], b0[j], c10, 0, 0, 0);
                        c11 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[j], b1[j], c11, 0, 0, 0);
                    }
                }
                if (STAGE >= 2) {
#pragma unroll
                    for (int g = 0; g < NG; ++g) {
                        *(d4*)(&sA[cur ^ 1][g][srow * LD + sch]) = ra[(i + 1) & 1][g];
                        *(d4*)(&sB[cur ^ 1][g][srow * LD + sch]) = rb[(i + 1) & 1][g];
                    }
                }
                __syncthreads();
            }
        }
    const d4 s = c00 + c01 + c10 + c11;
    out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

int main(int argc, char** argv) {
    int* dprobe; hipMalloc(&dprobe, 4096 * sizeof(int));
    hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, dprobe);
    std::vector<int> pr(4096);
    hipMemcpy(pr.data(), dprobe, 4096 * sizeof(int), hipMemcpyDeviceToHost);
    // print compactly: for each A lane the list of (B lane -> D lane) pairs that produce something
    printf("probe v_mfma_f64_4x4x4_4b_f64: A lane: B lane->D lane ...\n");
    for (int la = 0; la < 64; ++la) {
        printf("A%02d:", la);
        for (int lb = 0; lb < 64; ++lb) if (pr[la * 64 + lb] != -1) printf(" %d->%d", lb, pr[la * 64 + lb]);
        printf("\n");
    }
    double *d, *din;
    hipMalloc(&d, 256 * 8 * 256 * 8); hipMalloc(&din, 4096 * 8);
    std::vector<double> h(4096);
    unsigned long long x = 88172645463325252ull;
    for (auto& v : h) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = (double)(x >> 11) / 9007199254740992.0 * 2.0 - 1.0; }
    hipMemcpy(din, h.data(), 4096 * 8, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int ksteps = 64, reps = 200;
    for (int wpc = 1; wpc <= 2; ++wpc) {
        const int blocks = 256 * wpc;
        float ms;
        for (int which = 0; which < 2; ++which) {
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                if (which == 0) hipLaunchKernelGGL(k_tile16, dim3(blocks), dim3(256), 0, 0, d, din, ksteps, reps);
                else hipLaunchKernelGGL(k_tile4, dim3(blocks), dim3(256), 0, 0, d, din, ksteps, reps);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            hipEventElapsedTime(&ms, e0, e1);
            const double fl = (double)blocks * 4 * reps * ksteps * 16 * 2048.0;
            printf("%s LDS-fed 32x32 wave tile, %d workgroup(s)/CU: %.2f TFLOP/s (%.3f ms)\n", which ? "4x4x4_4b " : "16x16x4  ", wpc, fl / ms / 1e9, ms);
        }
    }
    {
        double* panel; hipMalloc(&panel, (size_t)2 * 256 * 64 * 4096 * 8);          // 1 GiB... 2 x 64 MiB x 8 B: per block 2 MiB of A and B rows
        hipMemset(panel, 0, (size_t)2 * 256 * 64 * 4096 * 8);
        auto run = [&](auto kern, const char* name, int wpc) {
            const int blocks = 256 * wpc;
            float ms;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, din, panel, ksteps, reps);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            hipEventElapsedTime(&ms, e0, e1);
            const double fl = (double)blocks * 4 * reps * ksteps * 16 * 2048.0;
            printf("%-70s %d workgroups/CU: %6.2f TFLOP/s\n", name, wpc, fl / ms / 1e9);
        };
        for (int wpc = 2; wpc <= 4; wpc += 2) {
            run(k_tile16s<1, 16>, "tile16 + barrier per 16 k", wpc);
            run(k_tile16s<2, 16>, "tile16 + barrier + ds_write staging per 16 k", wpc);
            run(k_tile16s<3, 16>, "tile16 + barrier + ds_write + global loads 2 steps ahead, per 16 k", wpc);
            run(k_tile16s<1, 32>, "tile16 + barrier per 32 k", wpc);
            run(k_tile16s<2, 32>, "tile16 + barrier + ds_write staging per 32 k", wpc);
            run(k_tile16s<3, 32>, "tile16 + barrier + ds_write + global loads 2 steps ahead, per 32 k", wpc);
        }
    }
    return 0;
}
