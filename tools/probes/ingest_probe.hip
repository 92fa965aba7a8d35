// Probe: how fast can one CU pull GEMM operand tiles into LDS with LDS-DMA
// (buffer_load_dwordx4 ... lds) on gfx950?  No MFMA, no LDS reads: only the staging stream of
// a tiled GEMM, with the same structure the real kernels use (ring of stages, counted vmcnt,
// one barrier per stage).  Answers: per-CU ingest rate vs waves per workgroup, row granule
// (128-byte vs 64-byte rows), ring depth, and source residency (shared L2-hot weights vs
// per-CU activation panels).
//
//   hipcc --offload-arch=gfx950 -O3 -o build/ingest_probe tools/probes/ingest_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __attribute__((address_space(3))) void* lds_ptr;

struct Args {
  const char* src;
  uint32_t bytes;        // extent of src
  int row_stride;        // bytes between consecutive tile rows in memory
  int rows_per_stage;    // tile rows staged per stage
  int nstage_iters;      // stages to stream per workgroup
  int k_bytes_total;     // wrap of the per-stage byte advance
  int adv;               // bytes added to the offset per stage (0 = one row's ROWB)
  int priv_stride;       // byte offset between workgroups' private panels (0 = all share)
  unsigned long long* sink;
};

// ROWB = bytes per tile row per stage (128 or 64); DEPTH = stages in flight; NW waves
template <int ROWB, int DEPTH, int NW>
__global__ __launch_bounds__(NW * 64) void ingest_kernel(Args p) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int LPR = ROWB / 16;                   // lanes per row
  constexpr int RPI = 64 / LPR;                    // rows per LDS-DMA instruction
  const int ninst = p.rows_per_stage / RPI;        // instructions per stage (all waves)
  const int per_wave = ninst / NW;                 // host guarantees divisibility
  const int stage_bytes = p.rows_per_stage * ROWB;
  const char* base = p.src + (size_t)blockIdx.x * p.priv_stride;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(base), 0, p.bytes - (uint32_t)((size_t)blockIdx.x * p.priv_stride), 0x00020000);
  // lane -> (row within instruction, 16-byte chunk)
  const int r_in = lane / LPR, ck = lane % LPR;
  auto issue = [&](int s, int slot) {
    const int kb = (int)(((long long)s * (p.adv ? p.adv : ROWB)) % p.k_bytes_total);
    for (int i = 0; i < per_wave; ++i) {
      const int inst = i * NW + wave;
      const int row = inst * RPI + r_in;
      const uint32_t off = (uint32_t)row * (uint32_t)p.row_stride + (uint32_t)kb + (uint32_t)(ck * 16);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(smem + slot * stage_bytes + inst * 1024), 16, off, 0, 0, 0);
    }
  };
  for (int s = 0; s < DEPTH - 1 && s < p.nstage_iters; ++s) issue(s, s);
  int slot = 0;
  for (int s = 0; s < p.nstage_iters; ++s) {
    // stage s must have landed; DEPTH-2 younger stages may stay in flight
    if (per_wave == 1) { asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DEPTH - 2) * 1) : "memory"); }
    else if (per_wave == 2) { asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DEPTH - 2) * 2) : "memory"); }
    else if (per_wave == 3) { asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DEPTH - 2) * 3) : "memory"); }
    else if (per_wave == 4) { asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DEPTH - 2) * 4) : "memory"); }
    else if (per_wave == 6) { asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DEPTH - 2) * 6) : "memory"); }
    else if (per_wave == 8) { asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DEPTH - 2) * 8) : "memory"); }
    else { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    __builtin_amdgcn_s_barrier();
    const int nx = s + DEPTH - 1;
    int sn = slot + DEPTH - 1; sn = sn >= DEPTH ? sn - DEPTH : sn;
    if (nx < p.nstage_iters) issue(nx, sn);
    slot = slot + 1 == DEPTH ? 0 : slot + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0 && p.sink) p.sink[blockIdx.x] = *(volatile unsigned long long*)smem;
#endif
}

template <int ROWB, int DEPTH, int NW>
double run(const char* name, Args a, int grid, int reps) {
  const int lds = a.rows_per_stage * ROWB * DEPTH;
  if (lds > 160 * 1024) { printf("%-58s  skipped (LDS %d)\n", name, lds); return 0; }
  hipFuncSetAttribute((const void*)ingest_kernel<ROWB, DEPTH, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((ingest_kernel<ROWB, DEPTH, NW>), dim3(grid), dim3(NW * 64), lds, 0, a);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((ingest_kernel<ROWB, DEPTH, NW>), dim3(grid), dim3(NW * 64), lds, 0, a);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / reps;
  const double bytes = (double)grid * a.nstage_iters * a.rows_per_stage * ROWB;
  const double per_cu = bytes / (grid < 256 ? grid : 256) / us * 1e-3;   // GB/s per CU
  printf("%-58s  %8.1f us  %7.2f TB/s chip  %6.1f GB/s per CU  (err %s)\n", name, us, bytes / us * 1e-6, per_cu,
         hipGetErrorString(hipGetLastError()));
  return per_cu;
}

int main() {
  const size_t total = 512u << 20;
  char* buf;
  hipMalloc(&buf, total);
  hipMemset(buf, 1, total);
  unsigned long long* sink;
  hipMalloc(&sink, 4096 * 8);
  Args a;
  a.src = buf; a.bytes = (uint32_t)(total - 1); a.sink = sink; a.adv = 0;
  const int reps = 20;
  // ---- shared, L2-hot operand (weights [N][K] with K = 320 bf16: 640-byte rows; 2560 rows = 1.6 MB) ----
  // every workgroup streams the same rows -> after the first touch everything is an L2 hit
  printf("== shared L2-hot tile rows (640-byte row stride), 256 workgroups, 1 per CU ==\n");
  a.row_stride = 640; a.k_bytes_total = 640; a.priv_stride = 0; a.nstage_iters = 400;
  a.rows_per_stage = 256;
  run<128, 2, 4>("rows 128B  256 rows/stage depth 2  4 waves", a, 256, reps);
  run<128, 3, 4>("rows 128B  256 rows/stage depth 3  4 waves", a, 256, reps);
  run<128, 4, 4>("rows 128B  256 rows/stage depth 4  4 waves", a, 256, reps);
  run<128, 3, 8>("rows 128B  256 rows/stage depth 3  8 waves", a, 256, reps);
  run<128, 4, 8>("rows 128B  256 rows/stage depth 4  8 waves", a, 256, reps);
  run<64, 4, 8>("rows  64B  256 rows/stage depth 4  8 waves", a, 256, reps);
  run<64, 8, 8>("rows  64B  256 rows/stage depth 8  8 waves", a, 256, reps);
  a.rows_per_stage = 128;
  run<128, 4, 4>("rows 128B  128 rows/stage depth 4  4 waves", a, 256, reps);
  run<128, 8, 4>("rows 128B  128 rows/stage depth 8  4 waves", a, 256, reps);
  run<128, 8, 8>("rows 128B  128 rows/stage depth 8  8 waves", a, 256, reps);
  printf("== same, 2 workgroups per CU (512 workgroups) ==\n");
  a.rows_per_stage = 256;
  run<128, 2, 4>("rows 128B  256 rows/stage depth 2  4 waves x2", a, 512, reps);
  a.rows_per_stage = 128;
  run<128, 4, 4>("rows 128B  128 rows/stage depth 4  4 waves x2", a, 512, reps);
  // ---- long rows (K = 1280 bf16 = 2560-byte stride), shared: FF-out / conv weights ----
  printf("== shared rows with 2560-byte stride (K=1280), 320 rows = 0.8 MB ==\n");
  a.row_stride = 2560; a.k_bytes_total = 2560; a.rows_per_stage = 320;
  run<128, 3, 8>("rows 128B  320 rows/stage depth 3  8 waves (wrap 20 k)", a, 256, reps);   // 40 instr / 8 = 5 per wave -> vmcnt(0) path
  a.rows_per_stage = 256;
  run<128, 4, 8>("rows 128B  256 rows/stage depth 4  8 waves (wrap 20 k)", a, 256, reps);
  run<64, 8, 8>("rows  64B  256 rows/stage depth 8  8 waves (wrap 40 k)", a, 256, reps);
  // ---- private panels: each workgroup streams its own 128-row x 640-byte panel region (activations) ----
  printf("== private activation panels: workgroup i reads rows [128 i, 128 i + 128) of a [32768][320] bf16 tensor ==\n");
  a.row_stride = 640; a.k_bytes_total = 640; a.rows_per_stage = 128; a.priv_stride = 128 * 640; a.nstage_iters = 5;
  run<128, 4, 4>("A panel once: 5 stages of 128 rows, depth 4, 4 waves", a, 256, reps);
  a.nstage_iters = 400;
  run<128, 4, 4>("A panel re-streamed 80x (L1/L2-hot), depth 4, 4 waves", a, 256, reps);
  run<128, 8, 8>("A panel re-streamed 80x (L1/L2-hot), depth 8, 8 waves", a, 256, reps);
  // ---- streaming from HBM: every workgroup reads its own 2 MB slice once ----
  printf("== HBM streaming: each workgroup its own 2 MB, 128-byte rows contiguous ==\n");
  a.row_stride = 128; a.k_bytes_total = 2 << 20; a.adv = 256 * 128; a.rows_per_stage = 256; a.priv_stride = 2 << 20; a.nstage_iters = 64;
  run<128, 4, 8>("contiguous 32 KB stages, depth 4, 8 waves", a, 256, reps);
  run<128, 4, 4>("contiguous 32 KB stages, depth 4, 4 waves", a, 256, reps);
  hipFree(buf); hipFree(sink);
  return 0;
}
