// Probe: streaming write / read / copy rates on MI355X for the tensor sizes of the U-Net step
// (21 MB .. 168 MB activations), 16 bytes per lane, grid-stride.
//   hipcc --offload-arch=gfx950 -O3 -o build/hbm_probe tools/probes/hbm_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

__global__ __launch_bounds__(256) void k_write(u32x4* dst, size_t n, uint32_t v) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = u32x4{v, v, v, v};
}
__global__ __launch_bounds__(256) void k_read(const u32x4* src, size_t n, uint32_t* sink) {
  u32x4 a = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { u32x4 x = src[i]; a ^= x; }
  if ((a[0] ^ a[1] ^ a[2] ^ a[3]) == 0x12345678u) *sink = 1;
}
__global__ __launch_bounds__(256) void k_copy(u32x4* dst, const u32x4* src, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

template <typename F> double timeit(F f, int reps) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) f();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) f();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3 / reps;
}

int main() {
  const size_t maxb = 1024u << 20;
  char *a, *b; uint32_t* sink;
  hipMalloc(&a, maxb); hipMalloc(&b, maxb); hipMalloc(&sink, 4);
  hipMemset(a, 1, maxb); hipMemset(b, 2, maxb);
  const int grid = 256 * 8;
  for (size_t mb : {21, 42, 84, 168, 336, 1024}) {
    const size_t bytes = mb << 20, n = bytes / 16;
    double w = timeit([&] { hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, 0, (u32x4*)a, n, 7u); }, 20);
    double r = timeit([&] { hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, (const u32x4*)a, n, sink); }, 20);
    double c = timeit([&] { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, (u32x4*)b, (const u32x4*)a, n); }, 20);
    // producer/consumer: write a then read a (the write should still be in the Infinity Cache)
    double wr = timeit([&] { hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, 0, (u32x4*)a, n, 7u);
                             hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, (const u32x4*)a, n, sink); }, 20);
    printf("%5zu MB: write %7.1f us %5.2f TB/s | read %7.1f us %5.2f TB/s | copy %7.1f us %5.2f TB/s (r+w) | write-then-read %7.1f us\n",
           mb, w, bytes / w * 1e-6, r, bytes / r * 1e-6, c, 2.0 * bytes / c * 1e-6, wr);
  }
  return 0;
}
