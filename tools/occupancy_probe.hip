// What the runtime says about workgroups per CU for the production kernels at their C3 launch shapes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/occupancy_probe.hip -o tools/occupancy_probe && tools/occupancy_probe
#include "../pymra_amd/csrc/mra_kernels.h"
#include <cstdio>
template <class K> static void probe(const char* name, K kern, int threads, size_t lds) {
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    int nb = -1;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, threads, lds);
    hipFuncAttributes fa{};
    hipFuncGetAttributes(&fa, (const void*)kern);
    printf("%-44s threads %4d  dyn LDS %7zu  static LDS %6zu  regs %3d  -> %d workgroups/CU (%s)\n", name, threads, lds, fa.sharedSizeBytes, fa.numRegs, nb, hipGetErrorString(e));
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("%s: %d CUs, LDS per CU %zu, per block %zu, regs per CU %d\n", p.name, p.multiProcessorCount, p.maxSharedMemoryPerMultiProcessor, p.sharedMemPerBlock, p.regsPerMultiprocessor);
    probe("k_trsm_rows2<8> nt=8", k_trsm_rows2<8>, 512, 36 * 2048);
    probe("k_trsm_rows2<8> nt=7", k_trsm_rows2<8>, 512, 28 * 2048);
    probe("k_trsm_rows2<8> 60 KB", k_trsm_rows2<8>, 512, 60 * 1024);
    probe("k_trsm_rows2<8> 40 KB", k_trsm_rows2<8>, 512, 40 * 1024);
    probe("k_predict_cascade<2,6,4,true,3>", k_predict_cascade<2, 6, 4, true, 3>, 256, 25 * 2048);
    probe("k_predict_cascade<2,6,4,true,2>", k_predict_cascade<2, 6, 4, true, 2>, 256, 25 * 2048);
    probe("k_prior_cascade<2,8,2,0,true>", k_prior_cascade<2, 8, 2, 0, true>, 512, 78 * 2048);
    probe("k_leaf_gemm<COV,2,0,1,8,512,4>", k_leaf_gemm<EPI_COV, 2, 0, 1, 8, 512, 4>, 512, 0);
    probe("k_parent_front<12>", k_parent_front<12>, 512, 100 * 1024);
    probe("k_chol_wave<12>", k_chol_wave<12>, 256, 0);
    return 0;
}
