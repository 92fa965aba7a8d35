// MFMA GEMM + implicit-GEMM 3x3 convolution for gfx950 (MI355X).
//
// One kernel template serves Dense / 1x1 / attention projections (plain A rows)
// and 3x3 convolutions (A rows gathered from an NHWC image: no im2col buffer).
//
//   * A [M][K] and W^T [N][K] are both K-contiguous, so every MFMA operand
//     fragment is one 16-byte LDS read.  A K-tile is 128 BYTES of K per row
//     (64 bf16 or 32 f32); the tile is kept in LDS as rows of 8 x 16-byte
//     chunks with chunk' = chunk ^ ((row >> 1) & 7), which makes the
//     ds_read_b128 of 32 consecutive rows at one chunk conflict-free.
//   * bf16: v_mfma_f32_32x32x16_bf16 (lane = row (l&31), k = 8*(l>>5)+j).
//     f32 : v_mfma_f32_32x32x2_f32 issued 4x per 16-byte fragment; lane half h
//     then covers k = 4h..4h+3 of each 8-wide k group for A and B alike, a
//     permutation of the summation order only.
//   * staging is branch-free and asynchronous: every 16-byte piece is a
//     `buffer_load_dwordx4 ... lds` (LDS-DMA, no VGPR round trip) whose offset is
//     pushed out of range when the piece is padding (image border, M/N/K tails)
//     -- the hardware range check then writes zeros, so the conv's zero padding
//     costs one v_cndmask per piece and no divergent control flow.  An LDS-DMA
//     wave-instruction writes 64 x 16 B linearly, so the XOR swizzle is applied
//     to the SOURCE chunk each lane fetches.  Two LDS stages form a ring: tile
//     t+1 is issued while tile t is multiplied; s_waitcnt vmcnt(0) plus ONE raw
//     s_barrier per K-tile orders it.  The 256x128 tile (one workgroup per CU
//     either way) runs a 3-stage ring with a counted vmcnt instead; on the
//     smaller tiles a third stage would cost a resident workgroup per CU.
//   * the epilogue goes through LDS: accumulators are dropped as an f32 tile and
//     re-read row-wise so that bias / addend / residual / output all move as
//     16-byte vectors (the MFMA accumulator layout alone would give 2-byte
//     scattered stores).
//   * workgroup ids are remapped so that the tiles sharing an A row-panel run
//     on one XCD (its L2 then serves the panel's re-reads).
//   * stride-1 convolutions at the 32x32 / 16x16 levels can run with a HALO-STAGED A operand (tiles
//     15 / 16, gemm_kernel.h MODE 3): the (lines + 2) x (W + 2) input pixels of a 64-channel chunk are
//     staged once and serve all nine taps; only the weight tiles go through the stage ring.
//   * small-M layers (4x4 / 8x8 feature maps) stream their weights with split-K
//     over all CUs; partial sums go to an f32 workspace and a second kernel
//     reduces + applies the epilogue.
#include "gemm_kernel.h"
#include "gemm3_kernel.h"

using namespace ldm_gemm_detail;

namespace {

// split-K reduce + epilogue: one thread per output element
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(GemmArgs p) {
  const int nout = p.act == LDM_ACT_GEGLU ? p.N / 2 : p.N;
  const int64_t total = (int64_t)p.M * nout;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * 256) {
    const int m = (int)(idx / nout), c = (int)(idx - (int64_t)m * nout);
    if (p.act == LDM_ACT_GEGLU) {
      const int nv = (c >> 5) * 64 + (c & 31), ng = nv + 32;
      float a = 0.f, g = 0.f;
      for (int s = 0; s < p.split_k; ++s) {
        const float* ws = p.ws + ((int64_t)s * p.M + m) * p.N;
        a += ws[nv]; g += ws[ng];
      }
      epi_store(p, 0, m, c, epi_pre(p, m, nv, a) * gelu_erf_f(epi_pre(p, m, ng, g)));
    } else {
      float a = 0.f;
      for (int s = 0; s < p.split_k; ++s) a += p.ws[((int64_t)s * p.M + m) * p.N + c];
      epi_store(p, 0, m, c, apply_act(p.act, epi_pre(p, m, c, a)));
    }
  }
}

// sum of the split slabs for 8 consecutive logical columns of row m, in split order
__device__ __forceinline__ void splitk_sum8(const GemmArgs& p, int m, int nlog, float (&v)[8]) {
  const float* src = p.ws + (int64_t)m * p.N + nlog;
  const int64_t slab = (int64_t)p.M * p.N;
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = 0.f;
  int s = 0;
  for (; s + 4 <= p.split_k; s += 4) {          // 8 x 16-byte loads in flight, then ordered adds
    f32x4 t[4][2];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      t[u][0] = *(const f32x4*)(src + (s + u) * slab);
      t[u][1] = *(const f32x4*)(src + (s + u) * slab + 4);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] += t[u][0][e]; v[4 + e] += t[u][1][e]; }
  }
  for (; s < p.split_k; ++s) {
    const f32x4 t0 = *(const f32x4*)(src + s * slab), t1 = *(const f32x4*)(src + s * slab + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] += t0[e]; v[4 + e] += t1[e]; }
  }
  if (p.bias) {
    const f32x4 b0 = *(const f32x4*)(p.bias + nlog), b1 = *(const f32x4*)(p.bias + nlog + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = v[e] * p.alpha + b0[e]; v[4 + e] = v[4 + e] * p.alpha + b1[e]; }
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] *= p.alpha;
  }
  if (p.addend) {
    const float* ad = p.addend + (int64_t)(m / p.add_rows) * p.add_ld + nlog;
    const f32x4 a0 = *(const f32x4*)ad, a1 = *(const f32x4*)(ad + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] += a0[e]; v[4 + e] += a1[e]; }
  }
}

// split-K reduce + epilogue for the row-major, 16-byte-aligned case: one thread per 8 output
// columns (the launches this serves are the 4x4 / 8x8 convolutions: it is pure slab traffic)
__global__ __launch_bounds__(256) void splitk_epilogue_vec_kernel(GemmArgs p) {
  const bool geglu = p.act == LDM_ACT_GEGLU;
  const int nout = geglu ? p.N / 2 : p.N;
  const int pcols = nout / 8;
  const int64_t total = (int64_t)p.M * pcols;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * 256) {
    const int m = (int)(idx / pcols), c = (int)(idx - (int64_t)m * pcols) * 8;
    const int nlog = geglu ? (c >> 5) * 64 + (c & 31) : c;
    float v[8];
    splitk_sum8(p, m, nlog, v);
    if (geglu) {
      float g[8];
      splitk_sum8(p, m, nlog + 32, g);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] *= gelu_erf_f(g[e]);
    } else if (p.act != LDM_ACT_NONE) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = apply_act(p.act, v[e]);
    }
    const int64_t ooff = (int64_t)m * p.ldc_m + c;
    const int64_t roff = (int64_t)m * p.ldr + c;
    if (p.out_dtype == LDM_BF16) {
      if (p.residual) {
        float rr[8];
        chunk_to_f32(*(const u32x4*)((const bf16_t*)p.residual + roff), rr, bf16_t());
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += rr[e];
      }
      *(u32x4*)((bf16_t*)p.out + ooff) = f32_to_chunk(v, bf16_t());
    } else {
      if (p.residual) {
        const f32x4 r0 = *(const f32x4*)((const float*)p.residual + roff);
        const f32x4 r1 = *(const f32x4*)((const float*)p.residual + roff + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] += r0[e]; v[4 + e] += r1[e]; }
      }
      f32x4 o0 = {v[0], v[1], v[2], v[3]}, o1 = {v[4], v[5], v[6], v[7]};
      *(f32x4*)((float*)p.out + ooff) = o0;
      *(f32x4*)((float*)p.out + ooff + 4) = o1;
    }
  }
}

struct TileCfg { int bm, bn; };
// index 1..4 (0 = auto)
constexpr int kNumTiles = 20;
constexpr TileCfg kTiles[kNumTiles] = {{0, 0}, {256, 128}, {128, 128}, {128, 64}, {64, 64}, {256, 128},
                                       {128, 160}, {256, 160}, {128, 320},    // 6-8: N = 160*k layers
                                       // 9-12: bf16 v_mfma_f32_16x16x32 path, wave tiles 64 x 80 / 64 x 64;
                                       // the 8-wave ones (9, 11) ping-pong their two wave halves
                                       {256, 160}, {128, 160}, {256, 128}, {128, 128},
                                       // 13, 14: persistent ping-pong kernel (gemm3_kernel.h): a workgroup walks
                                       // several n-tiles of its 256-row panel, register epilogue
                                       {256, 160}, {256, 128},
                                       // 15, 16: tiles 9 / 11 with the HALO-STAGED A operand (gemm_kernel.h MODE 3):
                                       // stride-1 convolutions whose M-tile is whole lines of one image
                                       {256, 160}, {256, 128},
                                       // 17-19: tiles 4 / 3 / 2 with a deeper LDS ring (4 / 3 / 3 stages): launches of
                                       // 1-3 workgroups per CU, where nothing else covers the per-K-tile round trip
                                       {64, 64}, {128, 64}, {128, 128}};
constexpr int kResident[kNumTiles] = {0, 1, 2, 3, 5, 1, 2, 1, 1, 1, 2, 1, 2, 1, 1, 1, 1, 2, 2, 1};   // workgroups per CU (LDS-limited)
constexpr bool kBf16Only[kNumTiles] = {false, false, false, false, false, false, false, false, false, true, true, true, true, true, true,
                                       true, true, false, false, false};
constexpr bool is_persistent(int c) { return c == 13 || c == 14; }
constexpr bool is_halo_ring(int c) { return c == 15 || c == 16; }
// MODE 3 geometry: stride 1, pad 1, no upsample, the 256-row M-tile = 256 / W whole lines of ONE image
bool halo_ring_ok(const ldm_gemm_params* p) {
  return p->conv && p->stride == 1 && !p->upsample && !p->no_lead_pad && p->dtype == LDM_BF16 && p->batch == 1 &&
         (p->W == 16 || p->W == 32) && 256 % p->W == 0 && p->H % (256 / p->W) == 0 && p->M % 256 == 0 &&
         p->OH == p->H && p->OW == p->W;
}

template <typename T>
void launch_mode(int cfg, const GemmArgs& a, dim3 grid, hipStream_t s) {
  if (cfg == 15 || cfg == 16) {
    if constexpr (sizeof(T) == 2) launch_cfg<T, 3>(cfg, a, grid, s);
    return;
  }
  if (!a.conv) launch_cfg<T, 0>(cfg, a, grid, s);
  else if (!a.upsample) launch_cfg<T, 1>(cfg, a, grid, s);
  else launch_cfg<T, 2>(cfg, a, grid, s);
}

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// Choose tile config + split-K jointly with a small cost model calibrated on MI355X
// (tools/gemm_bench.py): a launch runs in rounds of (256 CUs x resident workgroups)
// tiles; a round costs (K-tiles per split + a fixed prologue/epilogue overhead) x the
// measured time one CU needs for one K-tile of that configuration with its resident
// workgroups co-running.  Split-K adds the f32 partial round trip + one more launch.
void choose(const ldm_gemm_params* p, int esize, int* cfg_out, int* split_out) {
  static const double kStepUs[kNumTiles] = {0, 0.82, 0.82, 0.75, 0.87, 0.82, 1.12, 1.18, 1.15, 1.03, 1.09, 0.97, 0.80, 1.03, 0.97, 1.0, 0.94, 0.87, 0.75, 0.82};   // bf16, per round per K-tile (resident WGs co-running)
  static const double kOverheadSteps[kNumTiles] = {0, 8, 8, 7, 6, 8, 8, 8, 8, 8, 8, 8, 8, 4, 4, 9, 9, 6, 7, 8};   // launch + prologue + epilogue, in K-tiles
  static const int kSplits[] = {1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 20, 24, 32};
#ifdef LDM_TOOLS_BUILD
  static const bool no160 = getenv("LDM_GEMM_NO160") != nullptr;   // A/B switch, tools build only
  static const bool no_t2pref = getenv("LDM_GEMM_NO_T2PREF") != nullptr;
#else
  constexpr bool no160 = false, no_t2pref = false;
#endif
  const int bke = 128 / esize;
  const int ktiles = cdiv(p->K, bke);
  const bool geglu = p->act == LDM_ACT_GEGLU;
  const double f32x = esize == 4 ? 8.0 : 1.0;   // f32 MFMA: 1/16 the rate at half the K per tile
  double best = 1e30;
  int best_cfg = 2, best_split = 1;
  for (int c = 1; c < kNumTiles; ++c) {
    if (p->tile > 0 && p->tile < kNumTiles && c != p->tile) continue;
    if (c == 5 && p->tile != 5) continue;   // experimental: only when forced
    if (c >= 13 && p->tile != c) continue;   // persistent kernel, halo-staged conv tiles, deep-ring small tiles: only when forced (plan tables)
    if (c == 12 && p->tile != c) continue;   // 128x128 on the 16x16x32 path: no better than tile 2, only when forced
    if ((c == 1 || c == 7) && esize == 2 && p->tile != c) continue;   // bf16: their ping-ponged twins 11 / 9 are ~20 % faster
    if (c == 6 && esize == 2 && p->tile != 6) continue;   // bf16: tile 10 (same 128x160 tile, 64x80 wave tiles) is 10-14 % faster
    if (kBf16Only[c] && esize != 2) continue;
    if (p->a2 && (is_persistent(c) || is_halo_ring(c)) && p->tile != c) continue;   // second A operand: implicit-GEMM tiles only
    if (kTiles[c].bn % 160 == 0 && p->tile != c && (p->N % kTiles[c].bn != 0 || no160)) continue;   // 160/320-column tiles: N = 160*k layers
    if (p->out2 && p->n_split % kTiles[c].bn != 0) continue;                        // every tile on one side of n_split
    if (geglu && c > 2 && c != 5 && c != 11 && c != 12 && c != 14 && c != 19) continue;
    const TileCfg t = kTiles[c];
    const double tiles = (double)cdiv(p->M, t.bm) * cdiv(p->N, t.bn) * p->batch;
    for (int split : kSplits) {
      if (p->split_k > 0 && split != p->split_k) continue;
      if (is_persistent(c) && split > 1) continue;   // the persistent kernel does not split K
      // split-K only rescues launches that cannot fill the machine once, and must fit
      // the caller's workspace
      if (split > 1 && (p->batch != 1 || ktiles / split < 4 || tiles >= 256.0 * kResident[c] ||
                        (p->split_k <= 0 && (!p->workspace || (size_t)split * p->M * p->N * 4 > p->workspace_bytes))))
        continue;
      const int kps = cdiv(ktiles, split);
      const double rounds = (double)(int64_t)((tiles * split + 256.0 * kResident[c] - 1) / (256.0 * kResident[c]));
      double us = rounds * (kps + kOverheadSteps[c]) * kStepUs[c] * f32x;
      // calibration (tools/splitk_sweep.py): when the 256x128 tiling cannot fill the chip once
      // (small-M convolutions: 4x4 / 8x8 / 16x16 maps), two co-resident 128x128 workgroups per
      // CU measure 5-14 % faster than one 256x128 at the same split
      if (!no_t2pref && c == 2 && (double)cdiv(p->M, 256) * cdiv(p->N, 128) * p->batch < 256.0) us *= 0.9;
      if (split > 1) us += 3.0 + (double)p->M * p->N * 4.0 * (split + 1) / 3.0e6;   // bytes / (3 TB/s) in us
      if (us < best) { best = us; best_cfg = c; best_split = split; }
    }
  }
  if (p->split_k > 0 && p->batch == 1) best_split = p->split_k;
  if (p->tile > 0 && p->tile < kNumTiles) best_cfg = p->tile;
  // where the model picks the 256x160 ping-pong tile for a stride-1 convolution the halo-staged twin
  // (tile 15) measures 2-6 % faster from three channel chunks up (tools/conv_ring_probe.py)
  else if (best_cfg == 9 && halo_ring_ok(p) && p->N % 160 == 0 && !p->out2 && !p->ln_out && !p->a2 && ktiles / best_split >= 27)
    best_cfg = 15;
  *cfg_out = best_cfg;
  *split_out = best_split;
}

// The (tile, split) ldm_gemm really launches for p: the cost model / forced plan of choose() plus the
// overrides of the launch forms that pin the tile, and the split as it comes out after whole K-tile
// ranges (empty trailing splits dropped).  One function, because ldm_gemm, ldm_gemm_splits,
// ldm_gemm_reduce and ldm_groupnorm_splitk must agree on the slab count.
void final_plan(const ldm_gemm_params* p, int* cfg_out, int* split_out, int* kps_out) {
  const int esize = p->dtype == LDM_BF16 ? 2 : 4;
  int cfg, split;
  choose(p, esize, &cfg, &split);
  if (p->ln_out) { cfg = kLnTile; split = 1; }
  if (p->ln_cs && !is_persistent(cfg)) {
    // LayerNorm fold: the persistent kernel only (it derives the row statistics from its A tiles)
    cfg = (p->act == LDM_ACT_GEGLU || p->N % 160 != 0) ? 14 : (p->N % 128 == 0 && p->tile != 13 ? 14 : 13);
    split = 1;
  }
  if (p->out2 && !is_persistent(cfg)) split = 1;
  if (is_persistent(cfg)) split = 1;
  const int ktiles = cdiv(p->K, 128 / esize);
  int kps = cdiv(ktiles, split < 1 ? 1 : split);
  if (is_halo_ring(cfg)) kps = cdiv(kps, 9) * 9;   // splits at whole channel chunks (9 taps)
  split = cdiv(ktiles, kps);                        // drop empty trailing splits
  *cfg_out = cfg; *split_out = split; *kps_out = kps;
}

// kernel arguments of the non-persistent path (main kernel and split-K reduce alike)
void build_args(const ldm_gemm_params* p, int cfg, int split, int kps, GemmArgs* out) {
  const int esize = p->dtype == LDM_BF16 ? 2 : 4;
  const int bke = 128 / esize;
  int64_t a_bytes;
  if (p->conv) a_bytes = (((int64_t)p->B * p->H * p->W - 1) * p->lda + p->Cin) * esize;
  else a_bytes = (((int64_t)p->M - 1) * p->lda + (p->K - (p->a2 ? p->Cin2 : 0))) * esize;
  const int64_t w_bytes = (int64_t)p->N * p->K * esize;
  GemmArgs a;
  memset(&a, 0, sizeof(a));
  a.a = (const char*)p->a; a.w = (const char*)p->w; a.bias = p->bias; a.addend = p->addend;
  a.residual = (const char*)p->residual; a.out = (char*)p->out; a.ws = (float*)p->workspace;
  a.lda = p->lda; a.ldr = p->ldr; a.ldc_m = p->ldc_m; a.ldc_n = p->ldc_n;
  a.stride_a = p->stride_a; a.stride_w = p->stride_w; a.stride_c = p->stride_c; a.stride_r = p->stride_r;
  a.add_ld = p->add_ld; a.M = p->M; a.N = p->N; a.K = p->K; a.batch = p->batch;
  a.a_bytes = (uint32_t)a_bytes; a.w_bytes = (uint32_t)w_bytes;
  a.add_rows = p->add_rows > 0 ? p->add_rows : 1;
  a.conv = p->conv; a.H = p->H; a.W = p->W; a.Cin = p->Cin; a.OH = p->OH; a.OW = p->OW;
  a.stride = p->stride; a.upsample = p->upsample; a.pad = p->no_lead_pad ? 0 : 1; a.act = p->act; a.out_dtype = p->out_dtype;
  a.alpha = p->alpha;
  a.ln_out = (char*)p->ln_out; a.ln_gamma = p->ln_gamma; a.ln_beta = p->ln_beta; a.ld_ln = p->ld_ln;
  a.ln_eps = p->ln_eps;
  a.out2 = (char*)p->out2; a.ld2 = p->ld2; a.stride2 = p->stride2; a.n_split = p->n_split; a.rows2 = p->rows2;
  a.kt9 = cdiv(p->K, bke);                          // no second operand: every K-tile belongs to the first
  if (p->a2) {
    a.a2 = (const char*)p->a2; a.lda2 = p->lda2; a.kt9 = (p->K - p->Cin2) / bke;
    a.a2_bytes = (uint32_t)((((int64_t)p->M - 1) * p->lda2 + p->Cin2) * esize);
  }
  // vectorised epilogue: row-major output whose every 8-column piece is 16-byte addressable
  const int nout = p->act == LDM_ACT_GEGLU ? p->N / 2 : p->N;
  auto al = [](const void* q, int by) { return ((uintptr_t)q % by) == 0; };
  a.vec_epilogue =
      p->ldc_n == 1 && nout % 8 == 0 && p->ldc_m % 8 == 0 && p->stride_c % 8 == 0 && al(p->out, 16) &&
      (!p->residual || (p->ldr % 8 == 0 && p->stride_r % 8 == 0 && al(p->residual, 16))) &&
      (!p->bias || al(p->bias, 16)) &&
      (!p->addend || (al(p->addend, 16) && p->add_ld % 4 == 0));
  a.ktiles = cdiv(p->K, bke);
  a.split_k = split;
  a.ktiles_per_split = kps;
  const TileCfg t = kTiles[cfg];
  a.tiles_m = cdiv(p->M, t.bm);
  a.tiles_n = cdiv(p->N, t.bn);
  *out = a;
}

int launch_reduce(const ldm_gemm_params* p, const GemmArgs& a, hipStream_t s) {
  const int nout = p->act == LDM_ACT_GEGLU ? p->N / 2 : p->N;
  const bool vec = a.vec_epilogue && ((uintptr_t)p->workspace % 16) == 0 && p->N % 4 == 0;
  int64_t total = vec ? (int64_t)p->M * (nout / 8) : (int64_t)p->M * nout;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  if (vec) hipLaunchKernelGGL(splitk_epilogue_vec_kernel, dim3(blocks), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(splitk_epilogue_kernel, dim3(blocks), dim3(256), 0, s, a);
  return ldm_launch_status("ldm_gemm(splitk epilogue)");
}

}  // namespace

extern "C" int ldm_gemm_splits(const ldm_gemm_params* p) {
  if (!p || !(p->dtype == LDM_F32 || p->dtype == LDM_BF16) || p->K <= 0) return 0;
  int cfg, split, kps;
  final_plan(p, &cfg, &split, &kps);
  return split;
}

extern "C" int ldm_gemm_reduce(const ldm_gemm_params* p, void* stream) {
  LDM_CHECK_ARG(p && p->out && p->workspace, "ldm_gemm_reduce: null pointer");
  int cfg, split, kps;
  final_plan(p, &cfg, &split, &kps);
  LDM_CHECK_ARG(split > 1, "ldm_gemm_reduce: these parameters do not split K (nothing was deferred)");
  LDM_CHECK_ARG(p->workspace_bytes >= (size_t)split * p->M * p->N * 4, "ldm_gemm_reduce: workspace too small");
  GemmArgs a;
  build_args(p, cfg, split, kps, &a);
  return launch_reduce(p, a, (hipStream_t)stream);
}

extern "C" size_t ldm_gemm_workspace_bytes(const ldm_gemm_params* p) {
  if (!p) return 0;
  int cfg, split, kps;
  final_plan(p, &cfg, &split, &kps);
  return split > 1 ? (size_t)split * p->M * p->N * 4 : 0;
}

extern "C" int ldm_gemm_ln_supported(int N, int dtype) {
  return (N == kTiles[kLnTile].bn && (dtype == LDM_F32 || dtype == LDM_BF16)) ? 1 : 0;
}

extern "C" int ldm_gemm_plan(const ldm_gemm_params* p, int* tile, int* split_k) {
  LDM_CHECK_ARG(p && tile && split_k, "ldm_gemm_plan: null pointer");
  LDM_CHECK_ARG(p->dtype == LDM_F32 || p->dtype == LDM_BF16, "ldm_gemm_plan: bad dtype %d", p->dtype);
  int kps;
  final_plan(p, tile, split_k, &kps);
  return LDM_OK;
}

extern "C" int ldm_gemm(const ldm_gemm_params* p, void* stream) {
  LDM_CHECK_ARG(p && p->a && p->w && p->out, "ldm_gemm: null pointer");
  LDM_CHECK_ARG(p->dtype == LDM_F32 || p->dtype == LDM_BF16, "ldm_gemm: bad dtype %d", p->dtype);
  LDM_CHECK_ARG(p->out_dtype == LDM_F32 || p->out_dtype == LDM_BF16, "ldm_gemm: bad out_dtype");
  LDM_CHECK_ARG(p->M > 0 && p->N > 0 && p->K > 0 && p->batch > 0, "ldm_gemm: bad M/N/K/batch");
  const int esize = p->dtype == LDM_BF16 ? 2 : 4;
  const int osize = p->out_dtype == LDM_BF16 ? 2 : 4;
  const int epc = 16 / esize, bke = 8 * epc;
  LDM_CHECK_ARG(p->K % bke == 0, "ldm_gemm: K=%d must be a multiple of the K-tile (%d elements)", p->K, bke);
  LDM_CHECK_ARG(((uintptr_t)p->a % 16) == 0 && ((uintptr_t)p->w % 16) == 0,
                "ldm_gemm: a/w must be 16-byte aligned");
  LDM_CHECK_ARG(p->lda % epc == 0 && p->stride_a % epc == 0 && p->stride_w % epc == 0,
                "ldm_gemm: lda/stride_a/stride_w must be multiples of %d elements", epc);
  int64_t a_bytes;
  if (p->conv) {
    LDM_CHECK_ARG(p->Cin > 0 && p->Cin % bke == 0, "ldm_gemm(conv): Cin=%d must be a multiple of %d",
                  p->Cin, bke);
    LDM_CHECK_ARG(p->K == 9 * p->Cin + (p->a2 ? p->Cin2 : 0), "ldm_gemm(conv): K must be 9*Cin (+ Cin2 with a second operand)");
    if (p->a2) {
      LDM_CHECK_ARG(p->stride == 1 && !p->upsample && !p->no_lead_pad && p->OH == p->H && p->OW == p->W,
                    "ldm_gemm(conv): the second operand (a2) needs a stride-1, pad-1, non-upsampled convolution");
      LDM_CHECK_ARG(p->Cin2 > 0 && p->Cin2 % bke == 0 && p->lda2 % epc == 0 && p->lda2 >= p->Cin2 && ((uintptr_t)p->a2 % 16) == 0,
                    "ldm_gemm(conv): a2 needs Cin2 %% %d == 0, lda2 %% %d == 0 and 16-byte alignment", bke, epc);
      LDM_CHECK_ARG((((int64_t)p->M - 1) * p->lda2 + p->Cin2) * esize < (1ll << 31), "ldm_gemm(conv): a2 extent must be < 2 GiB");
    }
    LDM_CHECK_ARG(p->stride == 1 || p->stride == 2, "ldm_gemm(conv): stride must be 1 or 2");
    LDM_CHECK_ARG(p->B > 0 && p->H > 0 && p->W > 0 && p->OH > 0 && p->OW > 0, "ldm_gemm(conv): dims");
    LDM_CHECK_ARG(p->H < 32768 && p->W < 32768, "ldm_gemm(conv): H/W too large");
    LDM_CHECK_ARG(!(p->upsample && p->stride != 1), "ldm_gemm(conv): upsample needs stride 1");
    const int hs = p->upsample ? 2 * p->H : p->H, wsz = p->upsample ? 2 * p->W : p->W;
    LDM_CHECK_ARG(!p->no_lead_pad || p->stride == 2, "ldm_gemm(conv): no_lead_pad needs stride 2");
    const int padsum = p->no_lead_pad ? 1 : 2;
    LDM_CHECK_ARG(p->OH == (hs + padsum - 3) / p->stride + 1 && p->OW == (wsz + padsum - 3) / p->stride + 1,
                  "ldm_gemm(conv): OH/OW inconsistent with H/W/stride/upsample");
    LDM_CHECK_ARG(p->M == p->B * p->OH * p->OW, "ldm_gemm(conv): M != B*OH*OW");
    LDM_CHECK_ARG(p->batch == 1, "ldm_gemm(conv): batch must be 1");
    a_bytes = (((int64_t)p->B * p->H * p->W - 1) * p->lda + p->Cin) * esize;
  } else {
    a_bytes = (((int64_t)p->M - 1) * p->lda + p->K) * esize;
  }
  const int64_t w_bytes = (int64_t)p->N * p->K * esize;
  LDM_CHECK_ARG(a_bytes < (1ll << 31) && w_bytes < (1ll << 31),
                "ldm_gemm: operand extent must be < 2 GiB (a=%lld w=%lld bytes)", (long long)a_bytes,
                (long long)w_bytes);
  if (p->addend) LDM_CHECK_ARG(p->add_rows > 0, "ldm_gemm: add_rows must be > 0 with addend");
  if (p->act == LDM_ACT_GEGLU) {
    LDM_CHECK_ARG(p->N % 64 == 0, "ldm_gemm: GEGLU needs N %% 64 == 0");
    LDM_CHECK_ARG(p->ldc_n == 1, "ldm_gemm: GEGLU needs a row-major output");
  }
  LDM_CHECK_ARG(p->ldc_n == 1 || p->ldc_m == 1, "ldm_gemm: one of ldc_m / ldc_n must be 1");

  if (p->a2 && !p->conv) {
    LDM_CHECK_ARG(p->batch == 1 && p->Cin2 > 0 && p->Cin2 < p->K && p->Cin2 % bke == 0 && (p->K - p->Cin2) % bke == 0 &&
                      p->lda2 % epc == 0 && p->lda2 >= p->Cin2 && ((uintptr_t)p->a2 % 16) == 0 && !p->ln_cs && !p->ln_out && !p->out2,
                  "ldm_gemm: a2 (second A operand for the last Cin2 columns of K) needs batch 1, whole K-tiles on both "
                  "sides, lda2 %% %d == 0, 16-byte alignment and a plain epilogue", epc);
    LDM_CHECK_ARG((((int64_t)p->M - 1) * p->lda2 + p->Cin2) * esize < (1ll << 31), "ldm_gemm: a2 extent must be < 2 GiB");
    a_bytes = (((int64_t)p->M - 1) * p->lda + (p->K - p->Cin2)) * esize;
  }
  LDM_CHECK_ARG(!p->ln_cs || (p->dtype == LDM_BF16 && p->out_dtype == LDM_BF16 && !p->conv && p->batch == 1 && !p->ln_out),
                "ldm_gemm: ln_cs (LayerNorm fold) needs bf16 plain rows, batch 1 (persistent tiles 13 / 14)");

  LDM_CHECK_ARG(p->tile >= 0 && p->tile < kNumTiles, "ldm_gemm: tile %d does not exist (0 = auto, 1-%d)", p->tile, kNumTiles - 1);
  int cfg, split, kps;
  final_plan(p, &cfg, &split, &kps);
  if (p->ln_out) {
    LDM_CHECK_ARG(!p->conv && p->batch == 1 && p->act != LDM_ACT_GEGLU && p->ldc_n == 1 &&
                      ldm_gemm_ln_supported(p->N, p->out_dtype) && p->ln_gamma && p->ln_beta,
                  "ldm_gemm: ln_out needs plain rows, batch 1, no GEGLU, N == %d and gamma/beta", kTiles[kLnTile].bn);
    LDM_CHECK_ARG(p->ld_ln % 8 == 0 && ((uintptr_t)p->ln_out % 16) == 0 && ((uintptr_t)p->ln_gamma % 16) == 0 &&
                      ((uintptr_t)p->ln_beta % 16) == 0, "ldm_gemm: ln_out / gamma / beta alignment");
  }
  if (p->out2 && !is_persistent(cfg)) {
    LDM_CHECK_ARG(!p->conv && p->batch == 1 && p->act != LDM_ACT_GEGLU && p->ldc_n == 1 && !p->ln_out && !p->residual,
                  "ldm_gemm: out2 needs plain rows, batch 1, a row-major first output, no GEGLU / ln_out");
    LDM_CHECK_ARG(p->n_split > 0 && p->n_split < p->N && p->n_split % kTiles[cfg].bn == 0,
                  "ldm_gemm: n_split=%d must be a multiple of the tile width %d", p->n_split, kTiles[cfg].bn);
    LDM_CHECK_ARG(p->rows2 > 0 && p->rows2 % 4 == 0 && p->M % p->rows2 == 0 && p->ld2 % 4 == 0 && p->stride2 % 4 == 0 &&
                      ((uintptr_t)p->out2 % 16) == 0, "ldm_gemm: out2 geometry (rows2 %% 4, M %% rows2, ld2 / stride2 %% 4, alignment)");
  }
  if (p->act == LDM_ACT_GEGLU) LDM_CHECK_ARG(cfg <= 2 || cfg == 5 || cfg == 11 || cfg == 12 || cfg == 14 || cfg == 19, "ldm_gemm: GEGLU needs a tile whose width is a multiple of 64 (1, 2, 5, 11, 12)");
  LDM_CHECK_ARG(!kBf16Only[cfg] || esize == 2, "ldm_gemm: tile %d is bf16 only", cfg);
  LDM_CHECK_ARG(!p->a2 || (!is_persistent(cfg) && !is_halo_ring(cfg) && cfg < kNumTiles),
                "ldm_gemm: tile %d cannot take a second A operand (a2): implicit-GEMM tiles 1-12, 17-19 only", cfg);
  if (is_halo_ring(cfg))
    LDM_CHECK_ARG(halo_ring_ok(p) && p->N % kTiles[cfg].bn == 0 && !p->out2 && !p->ln_out,
                  "ldm_gemm: tile %d (halo-staged conv) needs a bf16 stride-1 3x3 convolution with W = 16 or 32, "
                  "H %% (256 / W) == 0, M %% 256 == 0 and N %% %d == 0", cfg, kTiles[cfg].bn);
  if (is_persistent(cfg)) {
    // persistent ping-pong kernel: bf16 in/out, row-major 16-byte-aligned output, whole n-tiles
    const int bn = kTiles[cfg].bn;
    auto al16 = [](const void* q) { return ((uintptr_t)q % 16) == 0; };
    // out2 with n_split == 0: the WHOLE product is stored transposed per group of rows2 rows (V^T)
    const bool trans = p->out2 != nullptr;
    const bool ln = p->ln_cs != nullptr;
    // out2 with n_split > 0: the n-tiles from n_split on are stored transposed, the ones below row-major
    // (q | k | V^T of the self-attention as one launch; LayerNorm-folded form only)
    const bool nsplit_out = trans && p->n_split > 0;
    if (ln)
      LDM_CHECK_ARG(!p->conv && p->bias && al16(p->ln_cs) && p->ln_eps > 0.f && !p->addend && !p->residual,
                    "ldm_gemm: ln_cs (LayerNorm fold) needs plain rows, the folded bias, ln_eps > 0 and no addend / residual");
    if (trans)
      LDM_CHECK_ARG((p->n_split == 0 || (ln && p->n_split % bn == 0 && p->n_split < p->N)) && !p->conv && (!p->bias || ln) &&
                        !p->addend && !p->residual && p->act == LDM_ACT_NONE &&
                        p->rows2 > 0 && p->rows2 % 32 == 0 && p->M % p->rows2 == 0 && p->ld2 % 8 == 0 && p->stride2 % 8 == 0 &&
                        al16(p->out2),
                    "ldm_gemm: tile %d transposed output needs n_split 0 (or, LayerNorm-folded, whole n-tiles below it), a plain "
                    "product, rows2 %% 32 == 0, ld2 / stride2 %% 8 == 0", cfg);
    LDM_CHECK_ARG(p->dtype == LDM_BF16 && p->out_dtype == LDM_BF16 && p->batch == 1 && p->ldc_n == 1 &&
                      !p->ln_out && split <= 1 && p->N % bn == 0 && ((trans && !nsplit_out) || (p->ldc_m % 8 == 0 && al16(p->out))) &&
                      (!p->residual || (p->ldr % 4 == 0 && al16(p->residual))) && (!p->bias || al16(p->bias)) &&
                      (!p->addend || (al16(p->addend) && p->add_ld % 4 == 0)),
                  "ldm_gemm: tile %d (persistent) needs bf16 in/out, batch 1, a row-major 16-byte-aligned output, "
                  "N %% %d == 0, no split-K / out2 / ln_out", cfg, bn);
    Gemm3Args g;
    memset(&g, 0, sizeof(g));
    g.a = (const char*)p->a; g.w = (const char*)p->w; g.bias = p->bias; g.addend = p->addend;
    g.residual = (const char*)p->residual; g.out = (char*)p->out;
    g.lda = p->lda; g.ldr = p->ldr; g.ldc = p->ldc_m; g.add_ld = p->add_ld;
    g.a_bytes = (uint32_t)a_bytes; g.w_bytes = (uint32_t)w_bytes;
    g.M = p->M; g.N = p->N; g.K = p->K; g.add_rows = p->add_rows > 0 ? p->add_rows : 1;
    g.conv = p->conv; g.H = p->H; g.W = p->W; g.Cin = p->Cin; g.OH = p->OH; g.OW = p->OW;
    g.stride = p->stride; g.upsample = p->upsample; g.pad = p->no_lead_pad ? 0 : 1;
    g.act = p->act;
    LDM_CHECK_ARG(p->alpha == 1.0f, "ldm_gemm: tile %d (persistent) needs alpha == 1", cfg);
#ifdef LDM_TOOLS_BUILD
    { static const int dbg = getenv("LDM_G3_DEBUG") ? atoi(getenv("LDM_G3_DEBUG")) : 0; g.dbg = dbg; }   // timing ablations
#endif
    g.ktiles = p->K / 64;
    g.panels = cdiv(p->M, 256);
    g.ntiles = p->N / bn;
    // workgroups per panel: fill the 256 CUs about once, at most one workgroup per n-tile
    int nsplit = p->split_k < 0 ? -p->split_k : (256 + g.panels / 2) / g.panels;
    if (nsplit < 1) nsplit = 1;
    if (nsplit > g.ntiles) nsplit = g.ntiles;
    g.tiles_per_wg = cdiv(g.ntiles, nsplit);
    g.nsplit = cdiv(g.ntiles, g.tiles_per_wg);
    if (nsplit_out) g.ntiles1 = p->n_split / bn;         // (a workgroup's range of n-tiles may straddle it)
    dim3 grid3((unsigned)(g.panels * g.nsplit));
    hipStream_t s3 = (hipStream_t)stream;
    g.out_t = (char*)p->out2; g.ld_t = p->ld2; g.stride_t = p->stride2; g.rows_t = p->rows2; g.n_split = p->n_split;
    g.ln_cs = p->ln_cs; g.ln_eps = p->ln_eps;
    const int epi = (trans ? (ln ? (kEpiTrans | kEpiBias | (nsplit_out ? kEpiSplit : 0)) : kEpiTrans)
                           : epi_code(p->bias != nullptr, p->addend != nullptr, p->residual != nullptr, p->act)) | (ln ? kEpiLn : 0);
    bool ok;
    if (ln) ok = launch_gemm3_ln(bn / 32, epi, g, grid3, s3);
    else if (!p->conv) ok = launch_gemm3<0>(bn / 32, epi, g, grid3, s3);
    else if (!p->upsample) ok = launch_gemm3<1>(bn / 32, epi, g, grid3, s3);
    else ok = launch_gemm3<2>(bn / 32, epi, g, grid3, s3);
    LDM_CHECK_ARG(ok, "ldm_gemm: tile %d (persistent) has no kernel for this epilogue (bias %d addend %d residual %d act %d)",
                  cfg, p->bias != nullptr, p->addend != nullptr, p->residual != nullptr, p->act);
    return ldm_launch_status("ldm_gemm(persistent)");
  }
  GemmArgs a;
  build_args(p, cfg, split, kps, &a);
  if (p->ln_out) LDM_CHECK_ARG(a.vec_epilogue, "ldm_gemm: ln_out needs the 16-byte-aligned row-major epilogue");
  if (split > 1) {
    const size_t need = (size_t)split * p->M * p->N * 4;
    if (!p->workspace || p->workspace_bytes < need) {
      ldm_set_error("ldm_gemm: split_k=%d needs %zu workspace bytes, have %zu", split, need,
                    p->workspace ? p->workspace_bytes : (size_t)0);
      return LDM_ERR_WORKSPACE;
    }
  }
  const int64_t nblk = (int64_t)a.tiles_m * a.tiles_n * (split > 1 ? split : p->batch);
  LDM_CHECK_ARG(nblk < (1ll << 31), "ldm_gemm: grid too large");
  dim3 grid((unsigned)nblk);
  hipStream_t s = (hipStream_t)stream;
  if (p->dtype == LDM_BF16) launch_mode<bf16_t>(cfg, a, grid, s);
  else launch_mode<float>(cfg, a, grid, s);
  int st = ldm_launch_status("ldm_gemm");
  if (st != LDM_OK) return st;
  // defer_reduce: the f32 slabs stay in the workspace; the caller completes the product with
  // ldm_gemm_reduce or fuses the reduction into the GroupNorm that follows (ldm_groupnorm_splitk)
  if (split > 1 && !p->defer_reduce) st = launch_reduce(p, a, s);
  return st;
}
