// Kernel template of the MFMA GEMM / implicit-GEMM convolution (see gemm.hip for the design
// notes).  Included by gemm.hip (host logic) and by the gemm_inst_*.hip translation units,
// each of which instantiates the tile configurations of one (element type, MODE) pair so the
// instantiations compile in parallel.
#pragma once
#include "common.h"
#include <stdlib.h>

namespace ldm_gemm_detail {


struct GemmArgs {
  const char* a;
  const char* w;
  const float* bias;
  const float* addend;
  const char* residual;
  char* out;
  float* ws;
  int64_t lda, ldr, ldc_m, ldc_n;
  int64_t stride_a, stride_w, stride_c, stride_r;
  int64_t add_ld;
  uint32_t a_bytes, w_bytes;   // addressable extent from the (per-batch) base pointers
  int M, N, K, batch;
  int add_rows;
  int conv, H, W, Cin, OH, OW, stride, upsample, pad;
  int act, out_dtype;
  int split_k, ktiles_per_split, ktiles;
  int tiles_m, tiles_n;
  int vec_epilogue;
  float alpha;
  char* ln_out;              // second output: LayerNorm of the stored rows (whole-row tiles only)
  const float* ln_gamma;
  const float* ln_beta;
  int64_t ld_ln;
  float ln_eps;
  char* out2;                // transposed second output for the columns >= n_split (q|k|v in one launch)
  int64_t ld2, stride2;
  int n_split, rows2;
  const char* a2;            // MODE 0 / 1 (stride 1): second A operand, plain rows (MODE 1: a 1x1 over image a2 at the
  int64_t lda2;              // output pixel) for the K-tiles kt >= kt9 (= K-tiles of the first; = ktiles when none)
  uint32_t a2_bytes;
  int kt9;
};

constexpr int kLnTile = 8;   // the tile whose BN (320) holds a whole row of the N = 320 layers

constexpr uint32_t kOOB = 0x80000000u;   // >= any num_records we accept: load returns 0

__device__ __forceinline__ void mma32(f32x16& acc, const u32x4& a, const u32x4& b, bf16_t) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a),
                                                __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}
__device__ __forceinline__ void mma32(f32x16& acc, const u32x4& a, const u32x4& b, float) {
#pragma unroll
  for (int j = 0; j < 4; ++j)
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a[j]), __uint_as_float(b[j]), acc,
                                               0, 0, 0);
}

// element j of six values held in registers (selects on VALUES: a select between captured references
// makes the compiler spill the whole closure to scratch and load through the selected pointer)
__device__ __forceinline__ uint32_t pick6(int j, uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, uint32_t a4, uint32_t a5) {
  uint32_t v = a0;
  v = j == 1 ? a1 : v;
  v = j == 2 ? a2 : v;
  v = j == 3 ? a3 : v;
  v = j == 4 ? a4 : v;
  v = j == 5 ? a5 : v;
  return v;
}

// value before activation: acc(already * alpha) + bias[n] + addend[group(m)][n]
__device__ __forceinline__ float epi_pre(const GemmArgs& p, int m, int n, float acc) {
  float v = acc * p.alpha;
  if (p.bias) v += p.bias[n];
  if (p.addend) v += p.addend[(int64_t)(m / p.add_rows) * p.add_ld + n];
  return v;
}

__device__ __forceinline__ void epi_store(const GemmArgs& p, int bz, int m, int ncol, float v) {
  const int64_t off = (int64_t)bz * p.stride_c + (int64_t)m * p.ldc_m + (int64_t)ncol * p.ldc_n;
  if (p.out_dtype == LDM_BF16) {
    if (p.residual)
      v += bf2f(((const bf16_t*)p.residual)[(int64_t)bz * p.stride_r + (int64_t)m * p.ldr + ncol]);
    ((bf16_t*)p.out)[off] = f2bf(v);
  } else {
    if (p.residual)
      v += ((const float*)p.residual)[(int64_t)bz * p.stride_r + (int64_t)m * p.ldr + ncol];
    ((float*)p.out)[off] = v;
  }
}

__device__ __forceinline__ float apply_act(int act, float v) {
  if (act == LDM_ACT_GELU) return gelu_erf_f(v);
  if (act == LDM_ACT_SILU) return silu_f(v);
  return v;
}

// MODE: 0 = plain rows, 1 = 3x3 conv (stride 1/2), 2 = 3x3 conv over the nearest-2x upsampled image
//       3 = 3x3 conv, stride 1, HALO-STAGED A operand (bf16 ping-pong tiles only): the M-tile is BM / W
//           whole lines of ONE image; per 64-channel chunk the (lines + 2) x (W + 2) input pixels are
//           staged ONCE (<= 43 LDS-DMA pieces) and all nine taps read their A fragments from that patch
//           at a row shift of kh (W + 2) + kw (the patch has its own swizzle key, row & 7, which keeps the
//           16x16x32 fragment reads conflict-free at every shift) instead of one 32-piece A tile per tap (288 per
//           chunk).  The patch of chunk c+1 lands while chunk c's nine taps are multiplied; only the
//           weight tiles run through the 3-stage ring.
// MF:   0 = v_mfma_f32_32x32x16 (wave tile = TM x TN blocks of 32x32)
//       1 = v_mfma_f32_16x16x32_bf16 (bf16 only; wave tile = blocks of 16x16, e.g. 64 x 80: the
//           N = 160*k layers get a 2.2x larger wave tile per LDS byte read than 32 x 160)
// ST:   1 = ping-pong the two halves of an 8-wave workgroup (waves w and w + 4 share a SIMD):
//           waves 0-3 run {stage next tile, read fragments of tile t, MFMA tile t} in every
//           barrier period, waves 4-7 run {MFMA tile t-1 (fragments kept in registers across the
//           barrier), stage, read fragments of tile t}: on each SIMD one wave multiplies while the
//           other stages and reads LDS, instead of both doing the same thing at the same time.
// NS:   0 = ring depth chosen below (2, or 3 on the one-workgroup-per-CU 8-wave tiles); 3 / 4 = a deeper ring on
//           the SMALL tiles (tiles 17-19).  A 64x64 / 128x64 launch with 5-20 K-tiles and a 2-deep ring pays one
//           L2 round trip per K-tile; with only 1-3 workgroups per CU (M <= 4096: the batch-8-per-GPU step) nothing
//           else covers it, so NS - 1 tiles in flight pay where residency does not (round 2 measured the deeper
//           rings slower at >= 5 workgroups per CU, where the LDS they cost evicts a resident workgroup).
template <typename T, int BM, int BN, int WM, int WN, int MODE, int MF = 0, int ST = 0, int NS = 0>
__global__ __launch_bounds__(WM* WN * 64) void gemm_kernel(GemmArgs p) {
  // The body uses LDS address-space pointers and gfx950 inline asm, which only the
  // device pass can parse; the host pass just needs the launch stub.
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int NW = WM * WN;
  constexpr int NT = NW * 64;
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int MB = MF ? 16 : 32;                 // MFMA block edge
  constexpr int TM = WTM / MB, TN = WTN / MB;
  constexpr int NR = MF ? 4 : 16;                  // accumulator registers per block
  // LDS-DMA instructions (8 rows x 128 B each) per wave per tile.  B rows need not fill whole
  // rounds of the NW waves: the last round is issued by the first waves only (wave-uniform
  // guard), so no LDS is spent on padding rows and the 256x160 tile fits a 3-stage ring.
  constexpr int NIB = BN / 8;
  constexpr int LA = BM / (8 * NW), LB = (NIB + NW - 1) / NW;
  constexpr int LBF = NIB / NW;                    // B rounds every wave takes part in
  constexpr int NL = LA + LB;
  constexpr int ES = (int)sizeof(T);
  constexpr int EPC = 16 / ES;
  constexpr int BKE = 8 * EPC;
  constexpr bool RING = MODE == 3;
  static_assert(!RING || (MF == 1 && ST == 1 && sizeof(T) == 2 && BM == 256), "halo-staged conv: bf16 ping-pong tiles");
  constexpr int PROWS = 344;                       // >= (BM/32 + 2) * 34 = 340 and (BM/16 + 2) * 18 = 324 patch rows
  constexpr int PATCH = PROWS * 128;
  constexpr int LAP = (PROWS / 8 + NW - 1) / NW;   // patch pieces (8 rows) per wave
  constexpr int STAGE = RING ? BN * 128 : (BM + BN) * 128;
  // A third stage (prefetch distance 2) only where it is free: tiles whose two stages already
  // leave room for just one workgroup per CU (160 KB LDS) and whose three stages still fit.
  constexpr int kLds = 160 * 1024;
  constexpr int NSTAGE = NS ? NS : RING ? 3 : (4 * STAGE > kLds && 3 * STAGE <= kLds && WM * WN == 8) ? 3 : 2;
  static_assert(NSTAGE >= 2 && NSTAGE <= 4 && (!NS || (!RING && !ST)), "ring depth");
  constexpr int RINGB = RING ? 2 * PATCH : 0;      // bytes of the two halo patches in front of the weight ring
  // the 160/320-column tiles stage their f32 epilogue tile in two row passes (LDS budget)
  constexpr int ESPLIT = (BN % 160 == 0 && WM >= 2) ? 2 : 1;
  constexpr int EROWS = BM / ESPLIT;
  constexpr int SMEM = RINGB + NSTAGE * STAGE > EROWS * BN * 4 ? RINGB + NSTAGE * STAGE : EROWS * BN * 4;
  static_assert(SMEM <= kLds, "LDS");
  static_assert(LA >= 1 && LB >= 1 && TM >= 1 && TN >= 1 && BM % (8 * NW) == 0 && BN % 8 == 0, "tile");
  static_assert(WTM % MB == 0 && WTN % MB == 0, "wave tile must be whole MFMA blocks");
  static_assert(!MF || (ES == 2 && WTM % 16 == 0 && (WTN / 2) % 8 == 0), "16x16x32 path: bf16, swizzle-aligned wave tiles");
  static_assert(EROWS % WTM == 0, "epilogue row pass must hold whole wave tiles");
  typedef __attribute__((address_space(3))) void* lds_ptr;

  __shared__ __attribute__((aligned(16))) char smem[SMEM];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  // ---- block -> (tile_m, tile_n, split, batch), XCD-aware -------------------
  const int ntile = p.tiles_m * p.tiles_n;
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x;
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
  }
  const int tile_id = bid % ntile;
  const int zz = bid / ntile;  // split index (split_k > 1) or batch index
  const int tile_m = tile_id / p.tiles_n, tile_n = tile_id % p.tiles_n;
  const int split = p.split_k > 1 ? zz : 0;
  const int bz = p.split_k > 1 ? 0 : zz;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const int kt_begin = split * p.ktiles_per_split;
  const int kt_end = min(p.ktiles, kt_begin + p.ktiles_per_split);
  const int nk = kt_end - kt_begin;

  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(p.a) + (int64_t)bz * p.stride_a * ES, 0, p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(p.w) + (int64_t)bz * p.stride_w * ES, 0, p.w_bytes, 0x00020000);

  // ---- per-lane staging geometry ---------------------------------------------
  // LDS-DMA instruction i of this wave fills rows g*8 .. g*8+7 (g = i*NW + wave) of the
  // tile: lane l lands at row g*8 + (l>>3), 16-byte slot l&7, so it must FETCH chunk
  // (l&7) ^ ((row>>1)&7) of that row (the read side applies the same XOR).
  // conv: a_base = byte offset of pixel (b, oy*s-1, ox*s-1) (+ chunk), a_mask = 9 tap-valid bits
  //       upsample: a_base = byte offset of image b, a_aux = ((oy-1) << 16) | ((ox-1) & 0xffff)
  // gemm: a_base = byte offset of row m (+ chunk), or kOOB for rows >= M
  int a_base[LA], a_mask[LA], a_aux[LA];
  [[maybe_unused]] int a2_base[LA];                 // MODE 0 / 1: byte offset of row / output pixel m in the second operand
#pragma unroll
  for (int i = 0; i < LA; ++i) {
    const int row = (i * NW + wave) * 8 + (lane >> 3);
    const int ck = (lane & 7) ^ ((row >> 1) & 7);
    const int m = m0 + row;
    a_base[i] = MODE == 0 ? (int)kOOB : 0; a_mask[i] = 0; a_aux[i] = 0;
    if constexpr (MODE <= 1) a2_base[i] = m < p.M ? (int)((int64_t)m * p.lda2 * ES) + ck * 16 : (int)kOOB;
    if (m < p.M) {
      if constexpr (MODE != 0) {
        const int ohw = p.OH * p.OW;
        const int b = m / ohw, rem = m - b * ohw;
        const int oy = rem / p.OW, ox = rem - oy * p.OW;
        const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
        const int Hs = MODE == 2 ? p.H * 2 : p.H, Ws = MODE == 2 ? p.W * 2 : p.W;
        int mask = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int iy = iy0 + t / 3, ix = ix0 + t % 3;
          if ((unsigned)iy < (unsigned)Hs && (unsigned)ix < (unsigned)Ws) mask |= 1 << t;
        }
        a_mask[i] = mask;
        if constexpr (MODE == 2) {
          a_base[i] = (int)((int64_t)b * p.H * p.W * p.lda * ES) + ck * 16;
          a_aux[i] = (iy0 << 16) | (ix0 & 0xffff);
        } else {
          a_base[i] = (int)(((int64_t)(b * p.H + iy0) * p.W + ix0) * p.lda * ES) + ck * 16;
        }
      } else {
        a_base[i] = (int)((int64_t)m * p.lda * ES) + ck * 16;
      }
    }
  }
  // weights: byte offset of row n (+ chunk), or kOOB for rows >= N (stays out of range for
  // every K offset we add: num_records < 2^31 and offsets are compared unsigned)
  int b_base[LB];
#pragma unroll
  for (int i = 0; i < LB; ++i) {
    const int row = (i * NW + wave) * 8 + (lane >> 3);
    const int ck = (lane & 7) ^ ((row >> 1) & 7);
    const int n = n0 + row;
    b_base[i] = (n < p.N && row < BN) ? (int)((int64_t)n * p.K * ES) + ck * 16 : (int)kOOB;
  }
  // waves that take part in the last (partial) round of B instructions
  const bool b_last = LB == LBF || (LBF * NW + wave) < NIB;
  const int row_pitch = (int)(p.lda * ES);          // bytes per pixel
  const int line_pitch = p.W * row_pitch;           // bytes per image line

  // K-tile kt -> loads into `stage`.  K is a multiple of the K-tile (checked on the host),
  // so only rows (M/N tails, conv padding) are ever masked, never K.
  [[maybe_unused]] const __amdgpu_buffer_rsrc_t rsA2 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(MODE <= 1 && p.a2 ? p.a2 : p.a), 0, MODE <= 1 && p.a2 ? p.a2_bytes : 0u, 0x00020000);
  auto issue_tile = [&](int kt, int stage) {
    char* dA = smem + stage * STAGE + wave * 1024;
    char* dB = dA + BM * 128;
    int kb;                                          // byte column of the weight matrix
    if (MODE <= 1 && kt >= p.kt9) {                  // (wave-uniform) the K-tiles of the second operand: plain rows
      const int kb2 = (kt - p.kt9) * 128;
      kb = kt * 128;                                 // (weights: [N][K of the first operand | K2], K-tile kt as ever)
#pragma unroll
      for (int i = 0; i < LA; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA2, (lds_ptr)(dA + i * NW * 1024), 16,
                                                 (uint32_t)a2_base[i] + (uint32_t)kb2, 0, 0, 0);
    } else if constexpr (MODE != 0) {
      // conv K order: channel chunk OUTER, tap INNER (kt = chunk*9 + tap).  The nine taps
      // of one 128-byte channel slice re-read the same pixels on consecutive K-tiles
      // (L1/L2 hits).  Only the summation order changes; the weight matrix keeps its
      // (tap, ci) layout.
      const int cc = kt / 9;
      const int tap = kt - cc * 9;
      const int cib = cc * 128;                      // channel byte offset
      kb = tap * p.Cin * ES + cib;
      const int kh = tap / 3, kw = tap - kh * 3;
      if constexpr (MODE == 2) {
#pragma unroll
        for (int i = 0; i < LA; ++i) {
          const int iy = ((a_aux[i] >> 16) + kh) >> 1;
          const int ix = ((int)(short)(a_aux[i] & 0xffff) + kw) >> 1;
          const uint32_t off = (uint32_t)(a_base[i] + iy * line_pitch + ix * row_pitch + cib);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(dA + i * NW * 1024), 16,
                                                   ((a_mask[i] >> tap) & 1) ? off : kOOB, 0, 0, 0);
        }
      } else {
        const int toff = kh * line_pitch + kw * row_pitch + cib;   // scalar
#pragma unroll
        for (int i = 0; i < LA; ++i) {
          const uint32_t off = (uint32_t)(a_base[i] + toff);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(dA + i * NW * 1024), 16,
                                                   ((a_mask[i] >> tap) & 1) ? off : kOOB, 0, 0, 0);
        }
      }
    } else {
      kb = kt * 128;
#pragma unroll
      for (int i = 0; i < LA; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(dA + i * NW * 1024), 16,
                                                 (uint32_t)a_base[i] + (uint32_t)kb, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < LBF; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_ptr)(dB + i * NW * 1024), 16,
                                               (uint32_t)b_base[i] + (uint32_t)kb, 0, 0, 0);
    if constexpr (LB != LBF) {
      if (b_last)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_ptr)(dB + LBF * NW * 1024), 16,
                                                 (uint32_t)b_base[LBF] + (uint32_t)kb, 0, 0, 0);
    }
  };

  // ---- MODE 3: halo patch geometry ----------------------------------------------------------
  // patch row = line * (W + 2) + px', line 0 = image line l0 - 1, px' 0 = pixel -1.  Piece slot j of
  // this wave covers patch rows (j NW + wave) 8 .. + 7; a lane fetches the chunk its LDS row's swizzle
  // asks for, or nothing (range check -> zeros) where the pixel lies outside the image.
  static_assert(!RING || LAP == 6, "patch piece slots");
  [[maybe_unused]] uint32_t ao0 = kOOB, ao1 = kOOB, ao2 = kOOB, ao3 = kOOB, ao4 = kOOB, ao5 = kOOB;   // (scalars: an array
  [[maybe_unused]] int prow0[TM];                    // indexed by the runtime tap would be demoted to LDS)
  [[maybe_unused]] int pw = 0, prows = 0;
  if constexpr (RING) {
    const int lines = BM / p.W;                     // whole image lines per M-tile (host-checked)
    pw = p.W + 2;
    prows = (lines + 2) * pw;
    const int tpi = p.H / lines;                    // M-tiles per image
    const int img = tile_m / tpi, l0 = (tile_m - img * tpi) * lines;
    auto piece_off = [&](int j) __attribute__((always_inline)) -> uint32_t {
      const int row = (j * NW + wave) * 8 + (lane >> 3);
      const int ck = (lane & 7) ^ (row & 7);          // patch swizzle: see calc_addr
      const int line = row / pw, px = row - line * pw - 1;
      const int iy = l0 - 1 + line;
      if (row < prows && (unsigned)iy < (unsigned)p.H && (unsigned)px < (unsigned)p.W && m0 < p.M)
        return (uint32_t)((((int64_t)img * p.H + iy) * p.W + px) * p.lda * ES) + ck * 16;
      return kOOB;
    };
    ao0 = piece_off(0); ao1 = piece_off(1); ao2 = piece_off(2);
    ao3 = piece_off(3); ao4 = piece_off(4); ao5 = piece_off(5);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int r = wm * WTM + i * MB + (lane & 15);          // output row of this lane in block i
      const int line = r / p.W;
      prow0[i] = line * pw + (r - line * p.W);                  // tap (0, 0): + kh pw + kw per tap
    }
  }
  // one patch piece (slot j = 0..5, a runtime value in the main loop)
  [[maybe_unused]] auto issue_patch_piece = [&](int lc, int chunk, int j) __attribute__((always_inline)) -> int {
    if constexpr (RING) {
      if ((j * NW + wave) * 8 >= prows) return 0;               // wave-uniform
      const uint32_t off = pick6(j, ao0, ao1, ao2, ao3, ao4, ao5);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(smem + (lc & 1) * PATCH + (j * NW + wave) * 1024), 16,
                                               off + (uint32_t)chunk * 128u, 0, 0, 0);
      return 1;
    }
    return 0;
  };
  // weight tile of step (chunk, tap) into ring stage `stage`
  [[maybe_unused]] auto issue_w = [&](int chunk, int tap, int stage) __attribute__((always_inline)) {
    if constexpr (RING) {
      char* dB = smem + RINGB + stage * STAGE + wave * 1024;
      const uint32_t kb = (uint32_t)(tap * p.Cin * ES + chunk * 128);
#pragma unroll
      for (int i = 0; i < LBF; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_ptr)(dB + i * NW * 1024), 16, (uint32_t)b_base[i] + kb, 0, 0, 0);
      if constexpr (LB != LBF) {
        if (b_last)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_ptr)(dB + LBF * NW * 1024), 16, (uint32_t)b_base[LBF] + kb, 0, 0, 0);
      }
    }
  };

  typedef float AccV __attribute__((ext_vector_type(NR)));
  AccV acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < NR; ++r) acc[i][j][r] = 0.f;

  // MFMA block geometry of this lane: lr = row (A) / column (B) inside a block, lh = k group.
  //   32x32x16: lr = l & 31, lh = l >> 5 (k = 8 lh + j per 16-wide step; 4 steps per K-tile)
  //   16x16x32: lr = l & 15, lh = l >> 4 (k = 8 lh + j per 32-wide step; 2 steps per K-tile)
  // accumulator element r of block (i, j): row i*MB + rowoff(r), column j*MB + lr
  constexpr int KS = MF ? 2 : 4;                   // MFMA k-steps per K-tile
  const int lr = MF ? (lane & 15) : (lane & 31), lh = MF ? (lane >> 4) : (lane >> 5);
  const int sw = (lr >> 1) & 7;
  // fragment byte offsets inside a stage: swizzled k-group columns, hoisted out of the loop
  int offA[KS], offB[KS];
#pragma unroll
  for (int kg = 0; kg < KS; ++kg) {
    const int coff = ((MF ? (kg * 4 + lh) : (kg * 2 + lh)) ^ sw) << 4;
    offA[kg] = (wm * WTM + lr) * 128 + coff;
    offB[kg] = (RING ? 0 : BM * 128) + (wn * WTN + lr) * 128 + coff;
  }

  // 2-stage ring, prefetch distance 1: at the top of K-tile t every outstanding LDS-DMA
  // belongs to tile t; after the wait + ONE barrier, tile t is visible to all waves and all
  // waves have finished reading the other stage (tile t-1), which tile t+1 may now overwrite.
  // 3-stage ring (NSTAGE == 3), prefetch distance 2: tiles t and t+1 are outstanding at the
  // top of K-tile t, so the wait is the counted vmcnt(this wave's loads per tile) -- tile t+1's
  // LDS-DMAs may still be in flight -- and tile t+2 goes to the stage tile t-1 was read from.
  if constexpr (!RING) {
    if (nk > 0) issue_tile(kt_begin, 0);
    if (NSTAGE >= 3 && nk > 1) issue_tile(kt_begin + 1, 1);
    if (NSTAGE >= 4 && nk > 2) issue_tile(kt_begin + 2, 2);
  }
  u32x4 fa[KS][TM], fb[KS][TN];
  // this wave's LDS-DMAs of tile t have landed: the younger tiles t+1 .. t+NSTAGE-2 (as far as they
  // exist) may stay in flight -- a counted wait of (tiles in flight) x (this wave's loads per tile)
  auto wait_tile = [&](int t) {
    const int ahead = min(NSTAGE - 2, nk - 1 - t);   // wave-uniform
    if (NSTAGE >= 4 && ahead == 2) {
      if (b_last) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NL) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (NL - 1)) : "memory");
    } else if (NSTAGE >= 3 && ahead == 1) {
      if (b_last) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL - 1) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  };
  auto read_frags = [&](const char* cS) {
#pragma unroll
    for (int kg = 0; kg < KS; ++kg) {
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[kg][i] = *(const u32x4*)(cS + offA[kg] + i * MB * 128);
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[kg][j] = *(const u32x4*)(cS + offB[kg] + j * MB * 128);
    }
  };
  // (issuing the next tile's LDS-DMAs between the two halves of the MFMA block instead of in
  // front of it was measured 3-5 % slower on the long-K convolutions and dropped)
  auto multiply = [&]() {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kg = 0; kg < KS; ++kg) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          if constexpr (MF) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[kg][i]),
                                                                __builtin_bit_cast(bf16x8, fb[kg][j]), acc[i][j], 0, 0, 0);
          } else {
            mma32(acc[i][j], fa[kg][i], fb[kg][j], T());
          }
        }
    }
    __builtin_amdgcn_s_setprio(0);
  };
  int st = 0;
  const bool late = ST && wave >= NW / 2;            // wave-uniform (wave is a readfirstlane)
  if constexpr (RING) {
    // Step s = (local chunk lc, tap): weight tile s in ring stage s % 3, A fragments from patch lc & 1.
    // After the barrier of step s a wave issues [patch piece `tap` of chunk lc + 1][weight tile s + 2];
    // at the top of step s the operations younger than weight tile s are therefore the previous
    // step's patch piece (if it issued one) and weight tile s + 1: the counted wait leaves exactly
    // those in flight.  The patch of chunk lc + 1 is complete long before its first use (its pieces go
    // out at taps 0-5 of chunk lc and are older than weight tile 9 (lc + 1)).
    const int nchunks = nk / 9, c_begin = kt_begin / 9;
#pragma unroll
    for (int j = 0; j < LAP; ++j) issue_patch_piece(0, c_begin, j);
    if (nk > 0) issue_w(c_begin, 0, 0);
    if (nk > 1) issue_w(c_begin, 1, 1);
    int lc = 0, tap = 0, prev_a = 0;
    auto wait_step = [&](int s) __attribute__((always_inline)) {
      if (s + 1 < nk) {
        const int extra = prev_a + (b_last ? 1 : 0);             // wave-uniform
        if (extra == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LB + 1) : "memory");
        else if (extra == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LB) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LB - 1) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    };
    auto stage_next = [&](int s) __attribute__((always_inline)) {
      prev_a = 0;
      if (tap < LAP && lc + 1 < nchunks) prev_a = issue_patch_piece(lc + 1, c_begin + lc + 1, tap);
      if (s + 2 < nk) {
        int t2 = tap + 2, c2 = lc;
        if (t2 >= 9) { t2 -= 9; ++c2; }
        int s2 = st + 2;
        s2 = s2 >= 3 ? s2 - 3 : s2;
        issue_w(c_begin + c2, t2, s2);
      }
    };
    // A fragment addresses of the CURRENT step, computed one step ahead (while the previous step's
    // fragment reads are in flight): the tap's row shift moves the swizzle key, so they are not a
    // constant plus an offset
    int adr[KS][TM];
    auto calc_addr = [&]() __attribute__((always_inline)) {
      const int kh = tap / 3, kw = tap - 3 * kh;
      const int shift = kh * pw + kw;
      const int pbase = (lc & 1) * PATCH;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        // The patch is swizzled by chunk ^ (row & 7), not by the (row >> 1) key of the tiles: a tap moves
        // the fragment rows by an arbitrary shift, and for the 16x16x32 fragment read (ds_read_b128 is
        // served in the lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ...) only the row & 7 key is
        // conflict-free at EVERY shift (the tiles' key: 2-way conflicts at odd shifts, 4-way at shifts
        // = 2 mod 4 -- measured 17x the bank-conflict cycles of tile 9 before this was changed).
        const int row = prow0[i] + shift;
        const int sw = row & 7;
#pragma unroll
        for (int kg = 0; kg < KS; ++kg) adr[kg][i] = pbase + row * 128 + (((kg * 4 + lh) ^ sw) << 4);
      }
    };
    calc_addr();
    auto read_ring = [&]() __attribute__((always_inline)) {
      const char* cB = smem + RINGB + st * STAGE;
#pragma unroll
      for (int kg = 0; kg < KS; ++kg) {
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[kg][i] = *(const u32x4*)(smem + adr[kg][i]);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[kg][j] = *(const u32x4*)(cB + offB[kg] + j * MB * 128);
      }
    };
    auto advance = [&]() __attribute__((always_inline)) {
      st = st + 1 == 3 ? 0 : st + 1;
      if (++tap == 9) { tap = 0; ++lc; }
    };
    if (!late) {
      for (int s = 0; s < nk; ++s) {
        wait_step(s);
        __builtin_amdgcn_s_barrier();
        stage_next(s);
        read_ring();
        advance();
        calc_addr();
        multiply();
      }
    } else {
      for (int s = 0; s < nk; ++s) {
        wait_step(s);
        __builtin_amdgcn_s_barrier();
        if (s > 0) multiply();
        stage_next(s);
        read_ring();
        advance();
        calc_addr();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
      }
      if (nk > 0) multiply();
    }
  } else if (!late) {
    for (int t = 0; t < nk; ++t) {
      wait_tile(t);
      __builtin_amdgcn_s_barrier();
      int sn = st + NSTAGE - 1;
      sn = sn >= NSTAGE ? sn - NSTAGE : sn;
      if (t + NSTAGE - 1 < nk) issue_tile(kt_begin + t + NSTAGE - 1, sn);
      read_frags(smem + st * STAGE);
      st = st + 1 == NSTAGE ? 0 : st + 1;
      multiply();
    }
  } else {
    // Late half.  Every wave passes the same nk barriers.  The fragments of tile t are read at
    // the END of period t and drained (lgkmcnt(0)) before the next barrier, so the stage of
    // tile t is free for the LDS-DMA of tile t + NSTAGE that any wave issues after that barrier.
    for (int t = 0; t < nk; ++t) {
      wait_tile(t);
      __builtin_amdgcn_s_barrier();
      int sn = st + NSTAGE - 1;
      sn = sn >= NSTAGE ? sn - NSTAGE : sn;
      if (t > 0) multiply();                          // tile t - 1, from registers
      if (t + NSTAGE - 1 < nk) issue_tile(kt_begin + t + NSTAGE - 1, sn);
      read_frags(smem + st * STAGE);
      st = st + 1 == NSTAGE ? 0 : st + 1;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    }
    if (nk > 0) multiply();
  }
  __syncthreads();   // all waves done with the staging LDS before the epilogue reuses it

  // ---- epilogue ---------------------------------------------------------------
  // accumulator register r of a block -> row inside the block (before the lane's 4*lh)
  auto rowoff = [](int r) { return MF ? r : (r & 3) + 8 * (r >> 2); };
  const int mb = m0 + wm * WTM + 4 * lh;
  const int nb = n0 + wn * WTN + lr;
  if (p.out2 && n0 >= p.n_split) {
    // this tile lies in the transposed part: 4 consecutive rows of one sample are contiguous
    // in out2 (rows2 % 4 == 0, so a 4-row group never straddles two samples)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = nb + j * MB;
        if (n >= p.N) continue;
#pragma unroll
        for (int r4 = 0; r4 < NR / 4; ++r4) {
          const int m = mb + i * MB + 8 * r4;
          if (m >= p.M) continue;
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e)
            v[e] = apply_act(p.act, epi_pre(p, min(m + e, p.M - 1), n, acc[i][j][r4 * 4 + e]));
          const int64_t off = (int64_t)(m / p.rows2) * p.stride2 + (int64_t)(n - p.n_split) * p.ld2 + (m % p.rows2);
          if (p.out_dtype == LDM_BF16) {
            u32x2 pk; pk[0] = pack_bf2(v[0], v[1]); pk[1] = pack_bf2(v[2], v[3]);
            *(u32x2*)((bf16_t*)p.out2 + off) = pk;
          } else {
            f32x4 pk = {v[0], v[1], v[2], v[3]};
            *(f32x4*)((float*)p.out2 + off) = pk;
          }
        }
      }
    return;
  }
  if (p.split_k > 1) {
    float* ws = p.ws + (int64_t)split * p.M * p.N;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = nb + j * MB;
        if (n >= p.N) continue;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          const int m = mb + i * MB + rowoff(r);
          if (m < p.M) ws[(int64_t)m * p.N + n] = acc[i][j][r];
        }
      }
    return;
  }

  if (p.vec_epilogue) {
    // (1) accumulators -> f32 tile [BM][BN] in LDS (the staging buffers are dead: the
    //     K loop's last barrier has been passed by every wave)
    float* sC = (float*)smem;
    const bool geglu = p.act == LDM_ACT_GEGLU;
    constexpr int PCOLS = BN / 8;                 // pieces per tile row (plain)
    const int pcols = geglu ? PCOLS / 2 : PCOLS;
    const int npieces = EROWS * pcols;
    const int nout = geglu ? p.N / 2 : (p.out2 ? p.n_split : p.N);
#pragma unroll
   for (int ep = 0; ep < ESPLIT; ++ep) {
    if (ep > 0) __syncthreads();                  // previous pass fully read
    if ((wm * WTM) / EROWS == ep) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < NR; ++r) {
            const int row = wm * WTM - ep * EROWS + i * MB + rowoff(r) + 4 * lh;
            sC[row * BN + wn * WTN + j * MB + lr] = acc[i][j][r] * p.alpha;
          }
    }
    __syncthreads();
    // (2) row-wise pieces of 8 output columns per thread
    for (int c = tid; c < npieces; c += NT) {
      const int row = c / pcols, pc = c - row * pcols;
      const int m = m0 + ep * EROWS + row;
      int ncol, lcol;                             // first output column, first LDS column (value)
      if (geglu) {
        const int oc = pc * 8;                    // within the tile's BN/2 output columns
        lcol = (oc >> 5) * 64 + (oc & 31);
        ncol = (n0 >> 1) + oc;
      } else {
        lcol = pc * 8;
        ncol = n0 + lcol;
      }
      if (m >= p.M || ncol >= nout) continue;
      float v[8];
      {
        const f32x4 x0 = *(const f32x4*)(sC + row * BN + lcol);
        const f32x4 x1 = *(const f32x4*)(sC + row * BN + lcol + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = x0[e]; v[4 + e] = x1[e]; }
      }
      const int nlog = geglu ? n0 + lcol : ncol;  // logical (pre-GEGLU) column of v[0]
      if (p.bias) {
        const f32x4 b0 = *(const f32x4*)(p.bias + nlog), b1 = *(const f32x4*)(p.bias + nlog + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] += b0[e]; v[4 + e] += b1[e]; }
      }
      if (p.addend) {
        const float* ad = p.addend + (int64_t)(m / p.add_rows) * p.add_ld + nlog;
        const f32x4 a0 = *(const f32x4*)ad, a1 = *(const f32x4*)(ad + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] += a0[e]; v[4 + e] += a1[e]; }
      }
      if (geglu) {
        float g[8];
        const f32x4 x0 = *(const f32x4*)(sC + row * BN + lcol + 32);
        const f32x4 x1 = *(const f32x4*)(sC + row * BN + lcol + 36);
#pragma unroll
        for (int e = 0; e < 4; ++e) { g[e] = x0[e]; g[4 + e] = x1[e]; }
        if (p.bias) {
          const f32x4 b0 = *(const f32x4*)(p.bias + nlog + 32), b1 = *(const f32x4*)(p.bias + nlog + 36);
#pragma unroll
          for (int e = 0; e < 4; ++e) { g[e] += b0[e]; g[4 + e] += b1[e]; }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= gelu_erf_f(g[e]);
      } else if (p.act != LDM_ACT_NONE) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = apply_act(p.act, v[e]);
      }
      const int64_t ooff = (int64_t)bz * p.stride_c + (int64_t)m * p.ldc_m + ncol;
      const int64_t roff = (int64_t)bz * p.stride_r + (int64_t)m * p.ldr + ncol;
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
      if (p.ln_out) {
        // keep the row AS STORED (rounded to the output dtype) for the LayerNorm pass below
        if (p.out_dtype == LDM_BF16) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = bf2f(f2bf(v[e]));
        }
        f32x4 o0 = {v[0], v[1], v[2], v[3]}, o1 = {v[4], v[5], v[6], v[7]};
        *(f32x4*)(sC + row * BN + lcol) = o0;
        *(f32x4*)(sC + row * BN + lcol + 4) = o1;
      }
    }
    if (p.ln_out) {
      // second output: LayerNorm of the rows of this pass.  The host only sets ln_out when the
      // tile holds whole rows (n0 == 0, N == BN); one wave per row, 8 columns per lane.
      __syncthreads();
      const bool act = lane < BN / 8;
      float gm[8], bt[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { gm[e] = act ? p.ln_gamma[lane * 8 + e] : 0.f; bt[e] = act ? p.ln_beta[lane * 8 + e] : 0.f; }
      for (int row = wave; row < EROWS; row += NW) {
        const int m = m0 + ep * EROWS + row;
        if (m >= p.M) break;                        // wave-uniform
        float f[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = 0.f;
        if (act) {
          const f32x4 x0 = *(const f32x4*)(sC + row * BN + lane * 8), x1 = *(const f32x4*)(sC + row * BN + lane * 8 + 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) { f[e] = x0[e]; f[4 + e] = x1[e]; }
        }
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) s += f[e];
        const float mean = wave_sum(s) * (1.0f / BN);
        float q = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = f[e] - mean; q += act ? d * d : 0.f; }
        const float rstd = rsqrtf(wave_sum(q) * (1.0f / BN) + p.ln_eps);
        if (act) {
          float y[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) y[e] = (f[e] - mean) * rstd * gm[e] + bt[e];
          const int64_t loff = (int64_t)m * p.ld_ln + lane * 8;
          if (p.out_dtype == LDM_BF16) {
            *(u32x4*)((bf16_t*)p.ln_out + loff) = f32_to_chunk(y, bf16_t());
          } else {
            f32x4 o0 = {y[0], y[1], y[2], y[3]}, o1 = {y[4], y[5], y[6], y[7]};
            *(f32x4*)((float*)p.ln_out + loff) = o0;
            *(f32x4*)((float*)p.ln_out + loff + 4) = o1;
          }
        }
      }
    }
   }
    return;
  }

  // ---- generic (unaligned / transposed) epilogue straight from the accumulators ----
  if (p.act == LDM_ACT_GEGLU) {
    if constexpr (!MF && (TN & 1) == 0) {   // 32-wide blocks: value block j, gate block j + 1
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; j += 2) {
          const int nv = nb + j * MB, ng = nv + 32;
          if (ng >= p.N) continue;
          const int ncol = ((n0 + wn * WTN + j * MB) >> 1) + lr;
#pragma unroll
          for (int r = 0; r < NR; ++r) {
            const int m = mb + i * MB + rowoff(r);
            if (m >= p.M) continue;
            const float a = epi_pre(p, m, nv, acc[i][j][r]);
            const float g = epi_pre(p, m, ng, acc[i][j + 1][r]);
            epi_store(p, bz, m, ncol, a * gelu_erf_f(g));
          }
        }
    }
    return;
  }
  if (p.ldc_n == 1) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = nb + j * MB;
        if (n >= p.N) continue;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          const int m = mb + i * MB + rowoff(r);
          if (m >= p.M) continue;
          epi_store(p, bz, m, n, apply_act(p.act, epi_pre(p, m, n, acc[i][j][r])));
        }
      }
  } else {
    // transposed store (ldc_m == 1): 4 consecutive m per lane are contiguous
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = nb + j * MB;
        if (n >= p.N) continue;
#pragma unroll
        for (int r4 = 0; r4 < NR / 4; ++r4) {
          const int m = mb + i * MB + 8 * r4;
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e)
            v[e] = apply_act(p.act, epi_pre(p, min(m + e, p.M - 1), n, acc[i][j][r4 * 4 + e]));
          const int64_t off = (int64_t)bz * p.stride_c + (int64_t)n * p.ldc_n + m;
          if (m + 3 < p.M && p.ldc_m == 1 && !p.residual) {
            if (p.out_dtype == LDM_BF16) {
              u32x2 pk; pk[0] = pack_bf2(v[0], v[1]); pk[1] = pack_bf2(v[2], v[3]);
              *(u32x2*)((bf16_t*)p.out + off) = pk;
            } else {
              f32x4 pk = {v[0], v[1], v[2], v[3]};
              *(f32x4*)((float*)p.out + off) = pk;
            }
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (m + e < p.M) epi_store(p, bz, m + e, n, v[e]);
          }
        }
      }
  }
#endif  // __HIP_DEVICE_COMPILE__
}


// host-side launcher of one (element type, MODE) pair; defined in gemm_inst_*.hip
template <typename T, int MODE>
void launch_cfg(int cfg, const GemmArgs& a, dim3 grid, hipStream_t s);

}  // namespace ldm_gemm_detail
