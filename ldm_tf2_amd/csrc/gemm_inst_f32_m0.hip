// gemm_kernel instantiations: element type float, MODE 0 (gemm_kernel.h)
#include "gemm_launch.h"
namespace ldm_gemm_detail {
template void launch_cfg<float, 0>(int, const GemmArgs&, dim3, hipStream_t);
}
