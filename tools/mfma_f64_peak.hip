// Micro-benchmark: sustained v_mfma_f64_16x16x4_f64 rate on MI355X (the guide lists no FP64 row).
// hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k(double* out, int iters, double a0, double b0) {
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-9, b = b0 - threadIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void kfma(double* out, int iters, double a0, double b0) {
    double acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = i;
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = fma(a, acc[i], b);
    }
    double s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    double* d; hipMalloc(&d, 256 * 8 * 2048 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int wpc = 1; wpc <= 8; wpc *= 2) {       // workgroups (of 4 waves) per CU
        int blocks = 256 * wpc;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0, 1.0);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double fl = (double)blocks * 4 * iters * 4 * 2048.0;
        printf("mfma_f64_16x16x4: %d waves/SIMD, 4 accumulators: %.2f TFLOP/s (%.3f ms)\n", wpc, fl / ms / 1e9, ms);
    }
    {
        int blocks = 256 * 4;
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0, 1.0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("mfma_f64_16x16x4: 4 waves/SIMD, 1 dependent accumulator: %.2f TFLOP/s\n", (double)blocks * 4 * iters * 2048.0 / ms / 1e9);
        blocks = 256;
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0, 1.0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("mfma_f64_16x16x4: 1 wave/SIMD, 1 dependent accumulator: %.2f TFLOP/s (%.1f cycles/MFMA at 2.4 GHz)\n",
               (double)blocks * 4 * iters * 2048.0 / ms / 1e9, ms * 1e-3 * 2.4e9 / iters);
        blocks = 256 * 8;
        hipEventRecord(e0);
        hipLaunchKernelGGL(kfma, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0000001, 1e-9);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("v_fma_f64 (VALU): %.2f TFLOP/s\n", (double)blocks * 256 * iters * 16 * 2.0 / ms / 1e9);
    }
    return 0;
}
