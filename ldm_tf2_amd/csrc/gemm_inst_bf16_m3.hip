// gemm_kernel instantiations: element type bf16_t, MODE 3 = halo-staged stride-1 convolution (gemm_kernel.h)
#include "gemm_launch.h"
namespace ldm_gemm_detail {
template void launch_cfg<bf16_t, 3>(int, const GemmArgs&, dim3, hipStream_t);
}
