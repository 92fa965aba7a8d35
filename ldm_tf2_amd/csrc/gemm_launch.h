// Definition of launch_cfg<T, MODE> (one switch over the tile configurations); each
// gemm_inst_*.hip includes this and explicitly instantiates one (T, MODE) pair.
#pragma once
#include "gemm_kernel.h"

namespace ldm_gemm_detail {

template <typename T, int MODE>
void launch_cfg(int cfg, const GemmArgs& a, dim3 grid, hipStream_t s) {
  if constexpr (MODE == 3) {   // halo-staged stride-1 convolution: tiles 15 (256x160) / 16 (256x128), bf16 only
    if constexpr (sizeof(T) == 2) {
      if (cfg == 15) hipLaunchKernelGGL((gemm_kernel<T, 256, 160, 4, 2, 3, 1, 1>), grid, dim3(512), 0, s, a);
      else hipLaunchKernelGGL((gemm_kernel<T, 256, 128, 4, 2, 3, 1, 1>), grid, dim3(512), 0, s, a);
    }
  } else
  switch (cfg) {
    case 1: hipLaunchKernelGGL((gemm_kernel<T, 256, 128, 4, 2, MODE>), grid, dim3(512), 0, s, a); break;
    case 2: hipLaunchKernelGGL((gemm_kernel<T, 128, 128, 2, 2, MODE>), grid, dim3(256), 0, s, a); break;
    case 3: hipLaunchKernelGGL((gemm_kernel<T, 128, 64, 2, 2, MODE>), grid, dim3(256), 0, s, a); break;
    case 5: hipLaunchKernelGGL((gemm_kernel<T, 256, 128, 2, 2, MODE>), grid, dim3(256), 0, s, a); break;   // 4 waves x (128x64)
    case 6: hipLaunchKernelGGL((gemm_kernel<T, 128, 160, 4, 1, MODE>), grid, dim3(256), 0, s, a); break;   // 4 waves x (32x160)
    case 7: hipLaunchKernelGGL((gemm_kernel<T, 256, 160, 8, 1, MODE>), grid, dim3(512), 0, s, a); break;   // 8 waves x (32x160)
    case 8: hipLaunchKernelGGL((gemm_kernel<T, 128, 320, 4, 2, MODE>), grid, dim3(512), 0, s, a); break;   // 8 waves x (32x160)
    case 9: case 10: case 11: case 12:
      if constexpr (sizeof(T) == 2) {
        if (cfg == 9) hipLaunchKernelGGL((gemm_kernel<T, 256, 160, 4, 2, MODE, 1, 1>), grid, dim3(512), 0, s, a);        // 8 waves x (64x80), 3-stage ring, ping-ponged halves
        else if (cfg == 10) hipLaunchKernelGGL((gemm_kernel<T, 128, 160, 2, 2, MODE, 1>), grid, dim3(256), 0, s, a);     // 4 waves x (64x80), 2 workgroups per CU
        else if (cfg == 11) hipLaunchKernelGGL((gemm_kernel<T, 256, 128, 4, 2, MODE, 1, 1>), grid, dim3(512), 0, s, a);  // 8 waves x (64x64), 3-stage ring, ping-ponged halves
        else hipLaunchKernelGGL((gemm_kernel<T, 128, 128, 2, 2, MODE, 1>), grid, dim3(256), 0, s, a);                    // 4 waves x (64x64), 2 workgroups per CU
      }
      break;
    // 17-19: the small tiles with a deeper LDS ring (gemm_kernel.h NS): for launches of 1-3 workgroups per CU
    case 17: hipLaunchKernelGGL((gemm_kernel<T, 64, 64, 2, 2, MODE, 0, 0, 4>), grid, dim3(256), 0, s, a); break;
    case 18: hipLaunchKernelGGL((gemm_kernel<T, 128, 64, 2, 2, MODE, 0, 0, 3>), grid, dim3(256), 0, s, a); break;
    case 19: hipLaunchKernelGGL((gemm_kernel<T, 128, 128, 2, 2, MODE, 0, 0, 3>), grid, dim3(256), 0, s, a); break;
    default: hipLaunchKernelGGL((gemm_kernel<T, 64, 64, 2, 2, MODE>), grid, dim3(256), 0, s, a); break;
  }
}

}  // namespace ldm_gemm_detail
