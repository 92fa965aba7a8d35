// gemm3_kernel instantiations (persistent ping-pong GEMM / conv, bf16), MODE 0: the epilogue
// combinations the sampling path uses (bias / per-sample addend / residual / activation).
#include "gemm3_kernel.h"
namespace ldm_gemm_detail {
template <>
bool launch_gemm3<0>(int tn, int epi, const Gemm3Args& a, dim3 grid, hipStream_t s) {
  if (epi == epi_code(false, false, false, 0)) {
    if (tn == 5) hipLaunchKernelGGL((gemm3_kernel<5, 0, epi_code(false, false, false, 0)>), grid, dim3(512), 0, s, a);
    else hipLaunchKernelGGL((gemm3_kernel<4, 0, epi_code(false, false, false, 0)>), grid, dim3(512), 0, s, a);
    return true;
  }
  if (epi == epi_code(true, false, false, 0)) {
    if (tn == 5) hipLaunchKernelGGL((gemm3_kernel<5, 0, epi_code(true, false, false, 0)>), grid, dim3(512), 0, s, a);
    else hipLaunchKernelGGL((gemm3_kernel<4, 0, epi_code(true, false, false, 0)>), grid, dim3(512), 0, s, a);
    return true;
  }
  if (epi == epi_code(true, false, true, 0)) {
    if (tn == 5) hipLaunchKernelGGL((gemm3_kernel<5, 0, epi_code(true, false, true, 0)>), grid, dim3(512), 0, s, a);
    else hipLaunchKernelGGL((gemm3_kernel<4, 0, epi_code(true, false, true, 0)>), grid, dim3(512), 0, s, a);
    return true;
  }
  if (epi == epi_code(false, false, true, 0)) {
    if (tn == 5) hipLaunchKernelGGL((gemm3_kernel<5, 0, epi_code(false, false, true, 0)>), grid, dim3(512), 0, s, a);
    else hipLaunchKernelGGL((gemm3_kernel<4, 0, epi_code(false, false, true, 0)>), grid, dim3(512), 0, s, a);
    return true;
  }
  if (epi == epi_code(true, false, false, 1)) {
    if (tn == 5) hipLaunchKernelGGL((gemm3_kernel<5, 0, epi_code(true, false, false, 1)>), grid, dim3(512), 0, s, a);
    else hipLaunchKernelGGL((gemm3_kernel<4, 0, epi_code(true, false, false, 1)>), grid, dim3(512), 0, s, a);
    return true;
  }
  if (epi == epi_code(true, false, false, 2)) { if (tn != 4) return false; hipLaunchKernelGGL((gemm3_kernel<4, 0, epi_code(true, false, false, 2)>), grid, dim3(512), 0, s, a); return true; }
  if (epi == epi_code(true, false, false, 3)) {
    if (tn == 5) hipLaunchKernelGGL((gemm3_kernel<5, 0, epi_code(true, false, false, 3)>), grid, dim3(512), 0, s, a);
    else hipLaunchKernelGGL((gemm3_kernel<4, 0, epi_code(true, false, false, 3)>), grid, dim3(512), 0, s, a);
    return true;
  }
  if (epi == kEpiTrans) {
    if (tn == 5) hipLaunchKernelGGL((gemm3_kernel<5, 0, kEpiTrans>), grid, dim3(512), 0, s, a);
    else hipLaunchKernelGGL((gemm3_kernel<4, 0, kEpiTrans>), grid, dim3(512), 0, s, a);
    return true;
  }
  return false;
}
}  // namespace ldm_gemm_detail
