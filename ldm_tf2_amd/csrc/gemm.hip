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
//   * small-M layers (4x4 / 8x8 feature maps) stream their weights with split-K
//     over all CUs; partial sums go to an f32 workspace and a second kernel
//     reduces + applies the epilogue.
#include "common.h"
#include <stdlib.h>

namespace {

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
  int debug;   // ablation switches (LDM_GEMM_DEBUG env): 1 = no in-loop loads, 2 = no MFMA, 4 = no barrier
  float alpha;
  char* ln_out;              // second output: LayerNorm of the stored rows (whole-row tiles only)
  const float* ln_gamma;
  const float* ln_beta;
  int64_t ld_ln;
  float ln_eps;
  char* out2;                // transposed second output for the columns >= n_split (q|k|v in one launch)
  int64_t ld2, stride2;
  int n_split, rows2;
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
template <typename T, int BM, int BN, int WM, int WN, int MODE>
__global__ __launch_bounds__(WM* WN * 64) void gemm_kernel(GemmArgs p) {
  // The body uses LDS address-space pointers and gfx950 inline asm, which only the
  // device pass can parse; the host pass just needs the launch stub.
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int NW = WM * WN;
  constexpr int NT = NW * 64;
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int BNP = (BN + 8 * NW - 1) / (8 * NW) * (8 * NW);   // B rows staged (>= BN: whole LDS-DMA rounds)
  constexpr int LA = BM / (8 * NW), LB = BNP / (8 * NW);  // LDS-DMA instructions per wave per tile
  constexpr int NL = LA + LB;
  constexpr int ES = (int)sizeof(T);
  constexpr int EPC = 16 / ES;
  constexpr int BKE = 8 * EPC;
  constexpr int STAGE = (BM + BNP) * 128;
  // A third stage (prefetch distance 2) only where it is free: tiles whose two stages already
  // leave room for just one workgroup per CU (160 KB LDS) and whose three stages still fit.
  constexpr int kLds = 160 * 1024;
  constexpr int NSTAGE = (4 * STAGE > kLds && 3 * STAGE <= kLds && WM * WN == 8) ? 3 : 2;
  // the 160/320-column tiles stage their f32 epilogue tile in two row passes (LDS budget)
  constexpr int ESPLIT = (BN % 160 == 0) ? 2 : 1;
  constexpr int EROWS = BM / ESPLIT;
  constexpr int SMEM = NSTAGE * STAGE > EROWS * BN * 4 ? NSTAGE * STAGE : EROWS * BN * 4;
  static_assert(LA >= 1 && LB >= 1 && TM >= 1 && TN >= 1 && BM % (8 * NW) == 0, "tile");
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
#pragma unroll
  for (int i = 0; i < LA; ++i) {
    const int row = (i * NW + wave) * 8 + (lane >> 3);
    const int ck = (lane & 7) ^ ((row >> 1) & 7);
    const int m = m0 + row;
    a_base[i] = MODE == 0 ? (int)kOOB : 0; a_mask[i] = 0; a_aux[i] = 0;
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
  const int row_pitch = (int)(p.lda * ES);          // bytes per pixel
  const int line_pitch = p.W * row_pitch;           // bytes per image line

  // K-tile kt -> loads into `stage`.  K is a multiple of the K-tile (checked on the host),
  // so only rows (M/N tails, conv padding) are ever masked, never K.
  auto issue_tile = [&](int kt, int stage) {
    char* dA = smem + stage * STAGE + wave * 1024;
    char* dB = dA + BM * 128;
    int kb;                                          // byte column of the weight matrix
    if constexpr (MODE != 0) {
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
    for (int i = 0; i < LB; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_ptr)(dB + i * NW * 1024), 16,
                                               (uint32_t)b_base[i] + (uint32_t)kb, 0, 0, 0);
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int lr = lane & 31, lh = lane >> 5;
  const int sw = (lr >> 1) & 7;
  // fragment byte offsets inside a stage: 4 swizzled k-group columns, hoisted out of the loop
  int offA[4], offB[4];
#pragma unroll
  for (int kg = 0; kg < 4; ++kg) {
    const int coff = ((kg * 2 + lh) ^ sw) << 4;
    offA[kg] = (wm * WTM + lr) * 128 + coff;
    offB[kg] = BM * 128 + (wn * WTN + lr) * 128 + coff;
  }

  // 2-stage ring, prefetch distance 1: at the top of K-tile t every outstanding LDS-DMA
  // belongs to tile t; after the wait + ONE barrier, tile t is visible to all waves and all
  // waves have finished reading the other stage (tile t-1), which tile t+1 may now overwrite.
  // 3-stage ring (NSTAGE == 3), prefetch distance 2: tiles t and t+1 are outstanding at the
  // top of K-tile t, so the wait is the counted vmcnt(NL) -- tile t+1's NL LDS-DMAs may still
  // be in flight -- and tile t+2 goes to the stage tile t-1 was read from.
  if (nk > 0) issue_tile(kt_begin, 0);
  if (NSTAGE == 3 && nk > 1) issue_tile(kt_begin + 1, 1);
  int st = 0;
  for (int t = 0; t < nk; ++t) {
    if (NSTAGE == 3 && t + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (t + NSTAGE - 1 < nk) {
      int sn = st + NSTAGE - 1;
      sn = sn >= NSTAGE ? sn - NSTAGE : sn;
      issue_tile(kt_begin + t + NSTAGE - 1, sn);
    }
    const char* cS = smem + st * STAGE;
    st = st + 1 == NSTAGE ? 0 : st + 1;
    u32x4 fa[4][TM], fb[4][TN];
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) {
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[kg][i] = *(const u32x4*)(cS + offA[kg] + i * 32 * 128);
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[kg][j] = *(const u32x4*)(cS + offB[kg] + j * 32 * 128);
    }
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kg = 0; kg < 4; ++kg)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) mma32(acc[i][j], fa[kg][i], fb[kg][j], T());
    __builtin_amdgcn_s_setprio(0);
  }
  __syncthreads();   // all waves done with the staging LDS before the epilogue reuses it

  // ---- epilogue ---------------------------------------------------------------
  const int mb = m0 + wm * WTM + 4 * lh;
  const int nb = n0 + wn * WTN + lr;
  if (p.out2 && n0 >= p.n_split) {
    // this tile lies in the transposed part: 4 consecutive rows of one sample are contiguous
    // in out2 (rows2 % 4 == 0, so a 4-row group never straddles two samples)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = nb + j * 32;
        if (n >= p.N) continue;
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const int m = mb + i * 32 + 8 * r4;
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
        const int n = nb + j * 32;
        if (n >= p.N) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mb + i * 32 + (r & 3) + 8 * (r >> 2);
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
          for (int r = 0; r < 16; ++r) {
            const int row = wm * WTM - ep * EROWS + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            sC[row * BN + wn * WTN + j * 32 + lr] = acc[i][j][r] * p.alpha;
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
    if constexpr ((TN & 1) == 0) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; j += 2) {
          const int nv = nb + j * 32, ng = nv + 32;
          if (ng >= p.N) continue;
          const int ncol = ((n0 + wn * WTN + j * 32) >> 1) + lr;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = mb + i * 32 + (r & 3) + 8 * (r >> 2);
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
        const int n = nb + j * 32;
        if (n >= p.N) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mb + i * 32 + (r & 3) + 8 * (r >> 2);
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
        const int n = nb + j * 32;
        if (n >= p.N) continue;
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const int m = mb + i * 32 + 8 * r4;
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
constexpr int kNumTiles = 9;
constexpr TileCfg kTiles[kNumTiles] = {{0, 0}, {256, 128}, {128, 128}, {128, 64}, {64, 64}, {256, 128},
                                       {128, 160}, {256, 160}, {128, 320}};   // 6-8: N = 320*k layers
constexpr int kResident[kNumTiles] = {0, 1, 2, 3, 5, 1, 2, 1, 1};   // workgroups per CU (LDS-limited: 2-stage ring)

template <typename T, int MODE>
void launch_cfg(int cfg, const GemmArgs& a, dim3 grid, hipStream_t s) {
  switch (cfg) {
    case 1: hipLaunchKernelGGL((gemm_kernel<T, 256, 128, 4, 2, MODE>), grid, dim3(512), 0, s, a); break;
    case 2: hipLaunchKernelGGL((gemm_kernel<T, 128, 128, 2, 2, MODE>), grid, dim3(256), 0, s, a); break;
    case 3: hipLaunchKernelGGL((gemm_kernel<T, 128, 64, 2, 2, MODE>), grid, dim3(256), 0, s, a); break;
    case 5: hipLaunchKernelGGL((gemm_kernel<T, 256, 128, 2, 2, MODE>), grid, dim3(256), 0, s, a); break;   // 4 waves x (128x64)
    case 6: hipLaunchKernelGGL((gemm_kernel<T, 128, 160, 4, 1, MODE>), grid, dim3(256), 0, s, a); break;   // 4 waves x (32x160)
    case 7: hipLaunchKernelGGL((gemm_kernel<T, 256, 160, 8, 1, MODE>), grid, dim3(512), 0, s, a); break;   // 8 waves x (32x160)
    case 8: hipLaunchKernelGGL((gemm_kernel<T, 128, 320, 4, 2, MODE>), grid, dim3(512), 0, s, a); break;   // 8 waves x (32x160)
    default: hipLaunchKernelGGL((gemm_kernel<T, 64, 64, 2, 2, MODE>), grid, dim3(256), 0, s, a); break;
  }
}
template <typename T>
void launch_mode(int cfg, const GemmArgs& a, dim3 grid, hipStream_t s) {
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
  static const double kStepUs[kNumTiles] = {0, 0.82, 0.82, 0.75, 0.87, 0.82, 1.12, 1.18, 1.15};   // bf16, per round per K-tile (resident WGs co-running)
  static const double kOverheadSteps[kNumTiles] = {0, 8, 8, 7, 6, 8, 8, 8, 8};   // launch + prologue + epilogue, in K-tiles
  static const int kSplits[] = {1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 20, 24, 32};
  static const bool no160 = getenv("LDM_GEMM_NO160") != nullptr;   // A/B switch for tools/
  const int bke = 128 / esize;
  const int ktiles = cdiv(p->K, bke);
  const bool geglu = p->act == LDM_ACT_GEGLU;
  const double f32x = esize == 4 ? 8.0 : 1.0;   // f32 MFMA: 1/16 the rate at half the K per tile
  double best = 1e30;
  int best_cfg = 2, best_split = 1;
  for (int c = 1; c < kNumTiles; ++c) {
    if (p->tile > 0 && p->tile < kNumTiles && c != p->tile) continue;
    if (c == 5 && p->tile != 5) continue;   // experimental: only when forced
    if (c >= 6 && p->tile != c && (p->N % kTiles[c].bn != 0 || no160)) continue;   // 160/320-column tiles: N = 160*k layers
    if (p->out2 && p->n_split % kTiles[c].bn != 0) continue;                        // every tile on one side of n_split
    if (geglu && c > 2 && c != 5) continue;
    const TileCfg t = kTiles[c];
    const double tiles = (double)cdiv(p->M, t.bm) * cdiv(p->N, t.bn) * p->batch;
    for (int split : kSplits) {
      if (p->split_k > 0 && split != p->split_k) continue;
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
      static const bool no_t2pref = getenv("LDM_GEMM_NO_T2PREF") != nullptr;   // A/B switch
      if (!no_t2pref && c == 2 && (double)cdiv(p->M, 256) * cdiv(p->N, 128) * p->batch < 256.0) us *= 0.9;
      if (split > 1) us += 3.0 + (double)p->M * p->N * 4.0 * (split + 1) / 3.0e6;   // bytes / (3 TB/s) in us
      if (us < best) { best = us; best_cfg = c; best_split = split; }
    }
  }
  if (p->split_k > 0 && p->batch == 1) best_split = p->split_k;
  if (p->tile > 0 && p->tile < kNumTiles) best_cfg = p->tile;
  *cfg_out = best_cfg;
  *split_out = best_split;
}

}  // namespace

extern "C" size_t ldm_gemm_workspace_bytes(const ldm_gemm_params* p) {
  if (!p) return 0;
  int cfg, split;
  choose(p, p->dtype == LDM_BF16 ? 2 : 4, &cfg, &split);
  return split > 1 ? (size_t)split * p->M * p->N * 4 : 0;
}

extern "C" int ldm_gemm_ln_supported(int N, int dtype) {
  return (N == kTiles[kLnTile].bn && (dtype == LDM_F32 || dtype == LDM_BF16)) ? 1 : 0;
}

extern "C" int ldm_gemm_plan(const ldm_gemm_params* p, int* tile, int* split_k) {
  LDM_CHECK_ARG(p && tile && split_k, "ldm_gemm_plan: null pointer");
  LDM_CHECK_ARG(p->dtype == LDM_F32 || p->dtype == LDM_BF16, "ldm_gemm_plan: bad dtype %d", p->dtype);
  choose(p, p->dtype == LDM_BF16 ? 2 : 4, tile, split_k);
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
    LDM_CHECK_ARG(p->K == 9 * p->Cin, "ldm_gemm(conv): K must be 9*Cin");
    LDM_CHECK_ARG(p->stride == 1 || p->stride == 2, "ldm_gemm(conv): stride must be 1 or 2");
    LDM_CHECK_ARG(p->B > 0 && p->H > 0 && p->W > 0 && p->OH > 0 && p->OW > 0, "ldm_gemm(conv): dims");
    LDM_CHECK_ARG(p->H < 32768 && p->W < 32768, "ldm_gemm(conv): H/W too large");
    LDM_CHECK_ARG(!(p->upsample && p->stride != 1), "ldm_gemm(conv): upsample needs stride 1");
    const int hs = p->upsample ? 2 * p->H : p->H, wsz = p->upsample ? 2 * p->W : p->W;
    LDM_CHECK_ARG(!p->no_lead_pad || (p->stride == 2 && !p->a_scale), "ldm_gemm(conv): no_lead_pad needs stride 2");
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

  // stride-1 convolutions with a GroupNorm prologue (or a forced halo tile): halo-staged
  // kernel of conv_halo.hip.  Without a prologue the implicit-GEMM kernel measures equal or
  // faster, so it stays the default.
  if (p->conv && p->stride == 1 && ((p->tile == 0 && p->a_scale) || p->tile > 10) && p->split_k <= 1 &&
      !getenv("LDM_NO_HALO")) {
    const int r = ldm_conv_halo_try(p, p->tile > 10 ? p->tile - 10 : 0, stream);
    if (r == 1) return LDM_OK;
    if (r < 0) return r;
  }
  LDM_CHECK_ARG(!p->a_scale, "ldm_gemm: the a_scale/a_shift prologue needs the stride-1 halo conv path "
                             "(shape not eligible: use ldm_groupnorm_apply + a plain conv)");

  int cfg, split;
  choose(p, esize, &cfg, &split);
  if (p->ln_out) {
    LDM_CHECK_ARG(!p->conv && p->batch == 1 && p->act != LDM_ACT_GEGLU && p->ldc_n == 1 &&
                      ldm_gemm_ln_supported(p->N, p->out_dtype) && p->ln_gamma && p->ln_beta,
                  "ldm_gemm: ln_out needs plain rows, batch 1, no GEGLU, N == %d and gamma/beta", kTiles[kLnTile].bn);
    LDM_CHECK_ARG(p->ld_ln % 8 == 0 && ((uintptr_t)p->ln_out % 16) == 0 && ((uintptr_t)p->ln_gamma % 16) == 0 &&
                      ((uintptr_t)p->ln_beta % 16) == 0, "ldm_gemm: ln_out / gamma / beta alignment");
    cfg = kLnTile;
    split = 1;
  }
  if (p->out2) {
    LDM_CHECK_ARG(!p->conv && p->batch == 1 && p->act != LDM_ACT_GEGLU && p->ldc_n == 1 && !p->ln_out && !p->residual,
                  "ldm_gemm: out2 needs plain rows, batch 1, a row-major first output, no GEGLU / ln_out");
    LDM_CHECK_ARG(p->n_split > 0 && p->n_split < p->N && p->n_split % kTiles[cfg].bn == 0,
                  "ldm_gemm: n_split=%d must be a multiple of the tile width %d", p->n_split, kTiles[cfg].bn);
    LDM_CHECK_ARG(p->rows2 > 0 && p->rows2 % 4 == 0 && p->M % p->rows2 == 0 && p->ld2 % 4 == 0 && p->stride2 % 4 == 0 &&
                      ((uintptr_t)p->out2 % 16) == 0, "ldm_gemm: out2 geometry (rows2 %% 4, M %% rows2, ld2 / stride2 %% 4, alignment)");
    split = 1;
  }
  if (p->act == LDM_ACT_GEGLU) LDM_CHECK_ARG(cfg <= 2 || cfg == 5, "ldm_gemm: GEGLU needs tile 1, 2 or 5");
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
  // vectorised epilogue: row-major output whose every 8-column piece is 16-byte addressable
  const int nout = p->act == LDM_ACT_GEGLU ? p->N / 2 : p->N;
  auto al = [](const void* q, int by) { return ((uintptr_t)q % by) == 0; };
  { const char* dbg = getenv("LDM_GEMM_DEBUG"); a.debug = dbg ? atoi(dbg) : 0; }
  a.vec_epilogue =
      p->ldc_n == 1 && nout % 8 == 0 && p->ldc_m % 8 == 0 && p->stride_c % 8 == 0 && al(p->out, 16) &&
      (!p->residual || (p->ldr % 8 == 0 && p->stride_r % 8 == 0 && al(p->residual, 16))) &&
      (!p->bias || al(p->bias, 16)) &&
      (!p->addend || (al(p->addend, 16) && p->add_ld % 4 == 0));
  if (p->ln_out) LDM_CHECK_ARG(a.vec_epilogue, "ldm_gemm: ln_out needs the 16-byte-aligned row-major epilogue");
  a.ktiles = cdiv(p->K, bke);
  a.split_k = split;
  a.ktiles_per_split = cdiv(a.ktiles, split);
  // drop empty trailing splits
  a.split_k = split = cdiv(a.ktiles, a.ktiles_per_split);
  if (split > 1) {
    const size_t need = (size_t)split * p->M * p->N * 4;
    if (!p->workspace || p->workspace_bytes < need) {
      ldm_set_error("ldm_gemm: split_k=%d needs %zu workspace bytes, have %zu", split, need,
                    p->workspace ? p->workspace_bytes : (size_t)0);
      return LDM_ERR_WORKSPACE;
    }
  }
  const TileCfg t = kTiles[cfg];
  a.tiles_m = cdiv(p->M, t.bm);
  a.tiles_n = cdiv(p->N, t.bn);
  const int64_t nblk = (int64_t)a.tiles_m * a.tiles_n * (split > 1 ? split : p->batch);
  LDM_CHECK_ARG(nblk < (1ll << 31), "ldm_gemm: grid too large");
  dim3 grid((unsigned)nblk);
  hipStream_t s = (hipStream_t)stream;
  if (p->dtype == LDM_BF16) launch_mode<bf16_t>(cfg, a, grid, s);
  else launch_mode<float>(cfg, a, grid, s);
  int st = ldm_launch_status("ldm_gemm");
  if (st != LDM_OK) return st;
  if (split > 1) {
    const bool vec = a.vec_epilogue && al(p->workspace, 16) && p->N % 4 == 0;
    int64_t total = vec ? (int64_t)p->M * (nout / 8) : (int64_t)p->M * nout;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (vec) hipLaunchKernelGGL(splitk_epilogue_vec_kernel, dim3(blocks), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(splitk_epilogue_kernel, dim3(blocks), dim3(256), 0, s, a);
    st = ldm_launch_status("ldm_gemm(splitk epilogue)");
  }
  return st;
}
