// gemm3_kernel instantiations with the LayerNorm fold (EPI bit 6), MODE 0: the transformer block's
// LayerNorm -> projection pairs (unet.py:309-313): q|k and q (bias' only), V^T (transposed store),
// GEGLU (bias' + GEGLU).
#include "gemm3_kernel.h"
namespace ldm_gemm_detail {
bool launch_gemm3_ln(int tn, int epi, const Gemm3Args& a, dim3 grid, hipStream_t s) {
  constexpr int kBias = epi_code(true, false, false, 0) | kEpiLn;
  constexpr int kGeglu = epi_code(true, false, false, LDM_ACT_GEGLU) | kEpiLn;
  constexpr int kTrans = kEpiTrans | kEpiBias | kEpiLn;
  if (epi == kBias) {
    if (tn == 5) hipLaunchKernelGGL((gemm3_kernel<5, 0, kBias>), grid, dim3(512), 0, s, a);
    else hipLaunchKernelGGL((gemm3_kernel<4, 0, kBias>), grid, dim3(512), 0, s, a);
    return true;
  }
  if (epi == kGeglu) {
    if (tn != 4) return false;
    hipLaunchKernelGGL((gemm3_kernel<4, 0, kGeglu>), grid, dim3(512), 0, s, a);
    return true;
  }
  // q | k row-major and V^T transposed in ONE launch (EPI bit 7; 128-column n-tiles: n_split = 2 heads Sp)
  constexpr int kSplit = kTrans | kEpiSplit;
  if (epi == kSplit) {
    if (tn != 4) return false;
    hipLaunchKernelGGL((gemm3_kernel<4, 0, kSplit>), grid, dim3(512), 0, s, a);
    return true;
  }
  if (epi == kTrans) {
    if (tn == 5) hipLaunchKernelGGL((gemm3_kernel<5, 0, kTrans>), grid, dim3(512), 0, s, a);
    else hipLaunchKernelGGL((gemm3_kernel<4, 0, kTrans>), grid, dim3(512), 0, s, a);
    return true;
  }
  return false;
}
}  // namespace ldm_gemm_detail
