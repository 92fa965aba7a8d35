// Row-panel kernel for the tail of the BasicTransformerBlock / SpatialTransformer at the 320-channel level
// (unet.py:311-313, :323-325, :335-338, :357-365), bf16:
//
//   [PRE ]  h  = r0 + bo + Wo . att                       cross-attention output projection + residual (:312)
//           y  = h + b2 + W2 . ( a * gelu(g) ),  (a | g) = W1 . LayerNorm(h) + b1       feed-forward (:313)
//   [POST]  out = r1 + bp + Wp . y                        proj_out of the SpatialTransformer + its residual (:363-365)
//
// as ONE launch per 128-row panel of the residual stream instead of up to four GEMM launches + a LayerNorm.
// The design is a row-panel kernel, not a per-layer GEMM:
//   * the panel (128 rows x 320 channels, 80 KB, the GEMM kernels' swizzled K-tile image) stays RESIDENT in
//     LDS and is rewritten in place from phase to phase: att rows -> h -> y.  It is the A operand of every
//     product, the source of the LayerNorm statistics (the LayerNorm is folded into W1 as in gemm3_kernel.h,
//     EPI bit 6) and the feed-forward's residual; h and y never reach HBM, nothing is re-staged (the
//     persistent GEMM re-stages its A K-tiles for every n-tile: 2/3 of its LDS-DMA traffic);
//   * the [M, 4C] hidden activation never leaves the CU: per chunk of 64 hidden units the first product
//     (128 x 128: 64 value + 64 gate columns, K = 320) is finished in registers, GEGLU'd, rounded to bf16 and
//     written into a 16 KB LDS tile that is exactly one A K-tile of the second product;
//   * every product with N = 320 accumulates [128 x 320] in registers as three 128-column pieces
//     (96 accumulator registers per lane);
//   * only weights stream: one 16 KB tile (128 weight rows x 64 K) per barrier period through a 3-stage ring
//     (18 periods for Wo, 8 per hidden chunk: 5 K-tiles of W1 + 3 column pieces of W2, 15 for Wp); a chunk's
//     folded bias / column sums ride in the ring as one extra 1 KB LDS-DMA, so the steady state contains no
//     ordinary global load (those would drain the DMA queue at their first use).
// Ablations (tools build, LDM_FFN_DEBUG; profiles/r03_probes.txt): of the feed-forward's 127 us, 32 are the
// skeleton (160 barriers, waits, A reads), 52 the MFMAs, 22 the GEGLU epilogue, 17 the B fragment reads, 7 the
// staging -- they add up: the two waves of a SIMD run in lockstep and nothing overlaps.  The per-layer launches
// this replaces cost more because each pays its own fill, drain and HBM round trip.
#include "common.h"
#include <stdlib.h>

namespace {

struct FfnArgs {
  const char* x;       // PRE = 0: [M][C] bf16 feed-forward input rows (row stride ldx): LayerNorm input AND residual
                       // PRE = 1: [M][K0] bf16 attention output rows (row stride ldx)
  const char* w1;      // [8C][C] bf16: gamma-folded, GEGLU-interleaved (blocks of 64 rows = 32 value + 32 gate)
  const char* aux;     // [8C / 128][256] f32: per 128 weight rows of w1: column sums (128) | folded bias (128)
  const char* w2;      // [C][4C] bf16
  const float* b2;     // [C]
  char* out;           // [M][C] bf16, row stride ldo
  int64_t ldx, ldo;
  uint32_t x_bytes, w1_bytes, w2_bytes, aux_bytes;
  int M;
  float eps;
  // PRE: o-projection
  const char* wo;      // [C][K0] bf16
  const float* bo;     // [C]
  const char* r0;      // [M][C] bf16 residual of the o-projection, row stride ldr0
  int64_t ldr0;
  uint32_t wo_bytes;
  // POST: proj_out
  const char* wp;      // [C][C] bf16
  const float* bp;     // [C]
  const char* r1;      // [M][C] bf16 residual of proj_out, row stride ldr1
  int64_t ldr1;
  uint32_t wp_bytes;
#ifdef LDM_TOOLS_BUILD
  int dbg;             // timing ablations (tools build only): 1 no MFMA, 2 no weight staging, 4 no GEGLU epilogue, 8 no B fragment reads
#endif
};

#ifdef LDM_TOOLS_BUILD
#define LDM_FFN_DBG(p) ((p).dbg)
#else
#define LDM_FFN_DBG(p) 0
#endif

constexpr uint32_t kOOBf = 0x80000000u;

// C: channels (320); KT0: K-tiles of the PRE product (attention width / 64; 0 = no PRE phase); POST: proj_out phase
template <int C, int KT0, bool POST>
__global__ __launch_bounds__(512) void st_tail_kernel(FfnArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr bool PRE = KT0 > 0;
  constexpr int BM = 128;
  constexpr int KT1 = C / 64;                  // K-tiles of a product over the channels (5)
  constexpr int HID = 4 * C, NCH = HID / 64;   // hidden width, chunks of 64 hidden units (20)
  constexpr int NP2 = (C + 127) / 128;         // 128-column pieces of an N = C product (3)
  constexpr int SPC = KT1 + NP2;               // barrier periods per hidden chunk (8)
  constexpr int S0 = PRE ? KT0 * NP2 : 0;      // periods of the PRE product
  constexpr int S1 = NCH * SPC;                // ... of the feed-forward
  constexpr int S2 = POST ? KT1 * NP2 : 0;     // ... of proj_out
  constexpr int S = S0 + S1 + S2;
  constexpr int TILE = 128 * 128;              // one staged tile: 128 rows x 128 bytes
  constexpr int STG = TILE + 1024;             // ring stage: weight tile + a chunk's (colsum | bias) KB
  constexpr int NSTAGE = 3;
  constexpr int OFF_H = KT1 * TILE, OFF_R = OFF_H + TILE;
  static_assert(C % 64 == 0 && OFF_R + NSTAGE * STG <= 160 * 1024 && KT0 <= KT1 + 1, "LDS");
  typedef __attribute__((address_space(3))) void* lds_ptr;
  __shared__ __attribute__((aligned(16))) char smem[OFF_R + NSTAGE * STG];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;     // wave tile: rows 32 wm .. +31, columns 64 wn .. +63 of a 128 x 128 tile
  const int lr = lane & 15, lh = lane >> 4;
  const int m0 = blockIdx.x * BM;

  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.x), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.w1), 0, p.w1_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.w2), 0, p.w2_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsAux = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.aux), 0, p.aux_bytes, 0x00020000);
  [[maybe_unused]] const __amdgpu_buffer_rsrc_t rsWo =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(PRE ? p.wo : p.w1), 0, PRE ? p.wo_bytes : 0u, 0x00020000);
  [[maybe_unused]] const __amdgpu_buffer_rsrc_t rsWp =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(POST ? p.wp : p.w1), 0, POST ? p.wp_bytes : 0u, 0x00020000);

  // ---- staging geometry: a tile is 16 LDS-DMA instructions of 8 rows; wave w issues instructions w and w + 8.
  // lane l lands at row 8 i + (l >> 3), 16-byte slot l & 7, and fetches chunk (l & 7) ^ ((row >> 1) & 7).
  int srow[2], sck[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    srow[h] = (wave + 8 * h) * 8 + (lane >> 3);
    sck[h] = ((lane & 7) ^ ((srow[h] >> 1) & 7)) * 16;
  }
  // the panel's input rows: KT1 tiles of x, or (PRE) KT0 tiles of the attention output (the last one may lie in
  // the hidden tile's region, which is free until the feed-forward starts)
  constexpr int KTP = PRE ? KT0 : KT1;
  {
    uint32_t xo[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int m = m0 + srow[h];
      xo[h] = m < p.M ? (uint32_t)((int64_t)m * p.ldx * 2) + sck[h] : kOOBf;
    }
#pragma unroll
    for (int kt = 0; kt < KTP; ++kt)
#pragma unroll
      for (int h = 0; h < 2; ++h)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_ptr)(smem + kt * TILE + (wave + 8 * h) * 1024), 16,
                                                 xo[h] == kOOBf ? kOOBf : xo[h] + kt * 128, 0, 0, 0);
  }
  // a weight tile whose rows are the output columns 128 pp .. of an N = C product, K bytes kbytes ..
  auto issue_rows = [&](const __amdgpu_buffer_rsrc_t& rs, char* dst, int pp, int kbytes, int row_pitch) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int n = 128 * pp + srow[h];
      const uint32_t off = n < C ? (uint32_t)(n * row_pitch + kbytes + sck[h]) : kOOBf;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(dst + (wave + 8 * h) * 1024), 16, off, 0, 0, 0);
    }
  };
  // weight tile of step s into ring slot `slot`
  auto issue = [&](int s, int slot) {
    char* dst = smem + OFF_R + slot * STG;
    if (PRE && s < S0) {
      const int kt = s / NP2, pp = s - kt * NP2;
      issue_rows(rsWo, dst, pp, kt * 128, KT0 * 128);
    } else if (s < S0 + S1) {
      const int f = s - S0;
      const int c = f / SPC, u = f - c * SPC;
      if (u < KT1) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const uint32_t off = (uint32_t)((128 * c + srow[h]) * (C * 2) + u * 128 + sck[h]);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW1, (lds_ptr)(dst + (wave + 8 * h) * 1024), 16, off, 0, 0, 0);
        }
        if (u == KT1 - 1)     // the chunk's (column sums | folded bias): 1 KB, every wave writes the same bytes
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsAux, (lds_ptr)(dst + TILE), 16, (uint32_t)(c * 1024 + lane * 16), 0, 0, 0);
      } else {
        issue_rows(rsW2, dst, u - KT1, c * 128, HID * 2);
      }
    } else {
      const int g = s - S0 - S1;
      const int kt = g / NP2, pp = g - kt * NP2;
      issue_rows(rsWp, dst, pp, kt * 128, C * 2);
    }
  };
  // DMAs a wave issues for step s (3 where the chunk's aux KB rides along)
  auto n_issued = [&](int s) {
    const int f = s - S0;
    return (f >= 0 && f < S1 && (f % SPC) == KT1 - 1) ? 3 : 2;
  };

  issue(0, 0);
  issue(1, 1);

  // ---- fragment addresses ------------------------------------------------------------------------
  int offA[2][2], offB[2][4];       // [k group][block]: byte offsets inside a 128 x 128-byte tile
#pragma unroll
  for (int kg = 0; kg < 2; ++kg) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = 32 * wm + 16 * i + lr;
      offA[kg][i] = row * 128 + (((kg * 4 + lh) ^ ((row >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = 64 * wn + 16 * j + lr;
      offB[kg][j] = row * 128 + (((kg * 4 + lh) ^ ((row >> 1) & 7)) << 4);
    }
  }

  f32x4 acc1[2][4], acc2[NP2][2][4];
  auto zero_acc2 = [&]() {
#pragma unroll
    for (int pp = 0; pp < NP2; ++pp)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc2[pp][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  zero_acc2();

  // Operands swapped (D = W_frag x A_frag): the lane owns row 16 i + lr and columns 16 j + 4 lh + r.
  // A fragments (resident panel / hidden tile) are read by read_a BEFORE the period's barrier where they do not
  // depend on it; B fragments one k group at a time (6 fragments live: 96 + 32 accumulator registers leave no more).
  u32x4 fa[2][2];
  auto read_a = [&](const char* sa) {
#pragma unroll
    for (int kg = 0; kg < 2; ++kg)
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[kg][i] = *(const u32x4*)(sa + offA[kg][i]);
  };
  auto mma_tile = [&](const char* sb, f32x4 (&acc)[2][4]) {
    if (LDM_FFN_DBG(p) & 1) return;
#pragma unroll
    for (int kg = 0; kg < 2; ++kg) {
      u32x4 fb[4];
      if (LDM_FFN_DBG(p) & 8) {
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j] = u32x4{(uint32_t)lane, 1u, 2u, 3u};
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j] = *(const u32x4*)(sb + offB[kg][j]);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fb[j]),
                                                              __builtin_bit_cast(bf16x8, fa[kg][i]), acc[i][j], 0, 0, 0);
    }
  };

  int slot = 0;
  // one barrier period: this wave's DMAs of step s have landed (those of step s + 1 may stay in flight), every
  // wave is past its reads of the slot step s + 2 goes into; `drain_lds`: this wave's LDS writes are visible
  auto begin_period = [&](int s, bool drain_lds) {
    if (s + 1 < S) {
      if (n_issued(s + 1) == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (drain_lds) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };
  auto issue_ahead = [&](int s) {
    int sn = slot + 2; sn = sn >= NSTAGE ? sn - NSTAGE : sn;
    if (s + 2 < S && !(LDM_FFN_DBG(p) & 2)) issue(s + 2, sn);
  };
  auto next_slot = [&]() { slot = slot + 1 == NSTAGE ? 0 : slot + 1; };

  // byte address of the 4 consecutive columns n .. n + 3 of `row` in the panel's K-tile image
  auto panel_cell = [&](int row, int n) {
    const int ck = ((n & 63) >> 3) ^ ((row >> 1) & 7);
    return smem + (n >> 6) * TILE + row * 128 + ck * 16 + (n & 7) * 2;
  };

  // ---- PRE: h = r0 + bo + Wo . att ------------------------------------------------------------------
  if constexpr (PRE) {
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");     // the panel's DMAs are older than the two weight tiles
    for (int kt = 0; kt < KT0; ++kt) {
#pragma unroll
      for (int pp = 0; pp < NP2; ++pp) {
        const int s = kt * NP2 + pp;
        begin_period(s, false);
        if (pp == 0) read_a(smem + kt * TILE);           // attention rows (visible after the first barrier)
        issue_ahead(s);
        if (128 * pp + 64 * wn < C) mma_tile(smem + OFF_R + slot * STG, acc2[pp]);
        next_slot();
      }
    }
    __builtin_amdgcn_s_barrier();                        // every wave is done with the attention rows
    // h -> the panel (bf16): + bias + residual rows from global memory
#pragma unroll
    for (int pp = 0; pp < NP2; ++pp) {
      if (128 * pp + 64 * wn >= C) continue;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = 32 * wm + 16 * i + lr;
        const int m = min(m0 + row, p.M - 1);
        const bf16_t* rr = (const bf16_t*)p.r0 + (int64_t)m * p.ldr0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int n = 128 * pp + 64 * wn + 16 * j + 4 * lh;
          const f32x4 bb = *(const f32x4*)(p.bo + n);
          const u32x2 rv = *(const u32x2*)(rr + n);
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = acc2[pp][i][j][r] + bb[r];
          v[0] += __uint_as_float(rv[0] << 16); v[1] += __uint_as_float(rv[0] & 0xffff0000u);
          v[2] += __uint_as_float(rv[1] << 16); v[3] += __uint_as_float(rv[1] & 0xffff0000u);
          u32x2 pk;
          pk[0] = pack_bf2(v[0], v[1]);
          pk[1] = pack_bf2(v[2], v[3]);
          *(u32x2*)panel_cell(row, n) = pk;
        }
      }
    }
    zero_acc2();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  } else {
    if (n_issued(0) + n_issued(1) == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }

  // ---- LayerNorm statistics of this wave's 32 rows from the resident panel -------------------------
  float rs_i[2], nm_i[2];
  {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    const bf2 one = __builtin_bit_cast(bf2, 0x3f803f80u);
    const int row = 32 * wm + (lane & 31), half = lane >> 5;   // lanes l and l + 32 split the row's chunks
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int kt = 0; kt < KT1; ++kt) {
      const char* base = smem + kt * TILE + row * 128 + half * 64;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const u32x4 cc = *(const u32x4*)(base + (((j + (row >> 1)) & 3) << 4));
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const uint32_t w = cc[e];
          const bf2 a = __builtin_bit_cast(bf2, w);
          s1 = __builtin_amdgcn_fdot2_f32_bf16(a, one, s1, false);
          s2 = __builtin_amdgcn_fdot2_f32_bf16(a, a, s2, false);
        }
      }
    }
    s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 32, 64);
    const float ik = 1.0f / (float)C;
    const float ln_mu = s1 * ik;
    const float ln_rs = rsqrtf(fmaxf(s2 * ik - ln_mu * ln_mu, 0.f) + p.eps);
    // statistics of the rows this lane owns in the MFMA layout: row 16 i + lr of the wave tile
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float mu = __shfl(ln_mu, 16 * i + lr, 64);
      rs_i[i] = __shfl(ln_rs, 16 * i + lr, 64);
      nm_i[i] = -rs_i[i] * mu;
    }
  }

  // ---- feed-forward: per hidden chunk 5 periods of W1 (-> GEGLU -> hidden tile) and 3 of W2 -----------
  // (the u loop is unrolled: the period's role and the accumulator piece are compile-time per instance)
  for (int c = 0; c < NCH; ++c) {
#pragma unroll
    for (int u = 0; u < SPC; ++u) {
      const int s = S0 + c * SPC + u;
      if (u < KT1) read_a(smem + u * TILE);          // panel K-tile u: independent of the barrier below
      begin_period(s, u == KT1);                     // (u == KT1: the hidden tile's writes of the period before)
      if (u >= KT1) read_a(smem + OFF_H);            // the hidden tile: visible after the barrier
      const char* sb = smem + OFF_R + slot * STG;
      issue_ahead(s);
      if (u < KT1) {
        if (u == 0) {
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc1[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        mma_tile(sb, acc1);
        if (u == KT1 - 1 && !(LDM_FFN_DBG(p) & 4)) {
          // GEGLU of the chunk: blocks 0, 1 = value, blocks 2, 3 = gate (one 64-row block of the interleaved
          // weights per wave); LayerNorm fold: rstd acc + (b' - rstd mean cs); result -> bf16 -> hidden tile
          const float* aux = (const float*)(sb + TILE);
#pragma unroll
          for (int jp = 0; jp < 2; ++jp) {
            const f32x4 csv = *(const f32x4*)(aux + 64 * wn + 16 * jp + 4 * lh);
            const f32x4 csg = *(const f32x4*)(aux + 64 * wn + 16 * (jp + 2) + 4 * lh);
            const f32x4 bv = *(const f32x4*)(aux + 128 + 64 * wn + 16 * jp + 4 * lh);
            const f32x4 bg = *(const f32x4*)(aux + 128 + 64 * wn + 16 * (jp + 2) + 4 * lh);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              float hv[4];
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const float val = __builtin_fmaf(rs_i[i], acc1[i][jp][r], __builtin_fmaf(nm_i[i], csv[r], bv[r]));
                const float gat = __builtin_fmaf(rs_i[i], acc1[i][jp + 2][r], __builtin_fmaf(nm_i[i], csg[r], bg[r]));
                hv[r] = val * gelu_erf_f(gat);
              }
              u32x2 pk;
              pk[0] = pack_bf2(hv[0], hv[1]);
              pk[1] = pack_bf2(hv[2], hv[3]);
              // hidden column (inside the chunk) 32 wn + 16 jp + 4 lh + r -> 16-byte chunk 4 wn + 2 jp + (lh >> 1)
              const int row = 32 * wm + 16 * i + lr;
              const int ck = (4 * wn + 2 * jp + (lh >> 1)) ^ ((row >> 1) & 7);
              *(u32x2*)(smem + OFF_H + row * 128 + ck * 16 + (lh & 1) * 8) = pk;
            }
          }
        }
      } else {
        const int pp = u - KT1;                    // compile-time after unrolling
        if (128 * pp + 64 * wn < C)                // (wave-uniform) the last piece is 64 columns wide
          mma_tile(sb, acc2[pp]);
      }
      next_slot();
    }
  }

  if constexpr (POST) {
    // ---- y = h + b2 + (second product) -> the panel, in place (each lane rewrites exactly the cells it
    // reads; the panel's last readers were the W1 periods of the last chunk, three barriers ago) ----------
#pragma unroll
    for (int pp = 0; pp < NP2; ++pp) {
      if (128 * pp + 64 * wn >= C) continue;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = 32 * wm + 16 * i + lr;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int n = 128 * pp + 64 * wn + 16 * j + 4 * lh;
          const f32x4 bb = *(const f32x4*)(p.b2 + n);
          u32x2* cell = (u32x2*)panel_cell(row, n);
          const u32x2 rv = *cell;
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = acc2[pp][i][j][r] + bb[r];
          v[0] += __uint_as_float(rv[0] << 16); v[1] += __uint_as_float(rv[0] & 0xffff0000u);
          v[2] += __uint_as_float(rv[1] << 16); v[3] += __uint_as_float(rv[1] & 0xffff0000u);
          u32x2 pk;
          pk[0] = pack_bf2(v[0], v[1]);
          pk[1] = pack_bf2(v[2], v[3]);
          *cell = pk;
        }
      }
    }
    zero_acc2();
    // ---- POST: out = r1 + bp + Wp . y ---------------------------------------------------------------
    for (int kt = 0; kt < KT1; ++kt) {
#pragma unroll
      for (int pp = 0; pp < NP2; ++pp) {
        const int s = S0 + S1 + kt * NP2 + pp;
        begin_period(s, kt == 0 && pp == 0);           // the first barrier publishes y
        if (pp == 0) read_a(smem + kt * TILE);
        issue_ahead(s);
        if (128 * pp + 64 * wn < C) mma_tile(smem + OFF_R + slot * STG, acc2[pp]);
        next_slot();
      }
    }
  }

  // ---- final epilogue: + bias + residual, bf16 store ---------------------------------------------------
  // POST: bias bp, residual r1 (global); otherwise the feed-forward's own bias b2 and residual = the panel's
  // rows, read back from LDS (PRE: h lives only there; PRE = 0: the kernel's input rows)
#pragma unroll
  for (int pp = 0; pp < NP2; ++pp) {
    if (128 * pp + 64 * wn >= C) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = 32 * wm + 16 * i + lr;
      const int m = m0 + row;
      if (m >= p.M) continue;
      bf16_t* orow = (bf16_t*)p.out + (int64_t)m * p.ldo;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = 128 * pp + 64 * wn + 16 * j + 4 * lh;
        const f32x4 bb = *(const f32x4*)((POST ? p.bp : p.b2) + n);
        u32x2 rv;
        if constexpr (POST) rv = *(const u32x2*)((const bf16_t*)p.r1 + (int64_t)m * p.ldr1 + n);
        else rv = *(const u32x2*)panel_cell(row, n);
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = acc2[pp][i][j][r] + bb[r];
        v[0] += __uint_as_float(rv[0] << 16); v[1] += __uint_as_float(rv[0] & 0xffff0000u);
        v[2] += __uint_as_float(rv[1] << 16); v[3] += __uint_as_float(rv[1] & 0xffff0000u);
        u32x2 pk;
        pk[0] = pack_bf2(v[0], v[1]);
        pk[1] = pack_bf2(v[2], v[3]);
        *(u32x2*)(orow + n) = pk;
      }
    }
  }
#endif
}

int check_common(const char* who, const void* x, int64_t ldx, const void* w1, const float* aux, const void* w2,
                 const float* b2, void* out, int64_t ldo, int M, int C, int xcols, float eps, int dtype) {
  LDM_CHECK_ARG(x && w1 && aux && w2 && b2 && out, "%s: null pointer", who);
  LDM_CHECK_ARG(dtype == LDM_BF16 && C == 320 && M > 0, "%s: bf16 with C = 320 only (M=%d C=%d dtype=%d)", who, M, C, dtype);
  LDM_CHECK_ARG(ldx % 8 == 0 && ldo % 4 == 0 && ldx >= xcols && ldo >= C && eps > 0.f, "%s: row strides / eps", who);
  auto al16 = [](const void* q) { return ((uintptr_t)q % 16) == 0; };
  LDM_CHECK_ARG(al16(x) && al16(w1) && al16(aux) && al16(w2) && al16(b2) && ((uintptr_t)out % 8) == 0,
                "%s: pointers must be 16-byte aligned (out: 8)", who);
  LDM_CHECK_ARG((((int64_t)M - 1) * ldx + xcols) * 2 < (1ll << 31), "%s: input extent must be < 2 GiB", who);
  return LDM_OK;
}

void fill_common(FfnArgs* a, const void* x, int64_t ldx, const void* w1, const float* aux, const void* w2,
                 const float* b2, void* out, int64_t ldo, int M, int C, int xcols, float eps) {
  memset(a, 0, sizeof(*a));
  a->x = (const char*)x; a->w1 = (const char*)w1; a->aux = (const char*)aux; a->w2 = (const char*)w2; a->b2 = b2;
  a->out = (char*)out; a->ldx = ldx; a->ldo = ldo;
  a->x_bytes = (uint32_t)((((int64_t)M - 1) * ldx + xcols) * 2);
  a->w1_bytes = (uint32_t)(8 * C * C * 2); a->w2_bytes = (uint32_t)(C * 4 * C * 2);
  a->aux_bytes = (uint32_t)(8 * C / 128 * 1024);
  a->M = M; a->eps = eps;
#ifdef LDM_TOOLS_BUILD
  { static const int dbg = getenv("LDM_FFN_DEBUG") ? atoi(getenv("LDM_FFN_DEBUG")) : 0; a->dbg = dbg; }
#endif
}

}  // namespace

extern "C" int ldm_ffn_geglu_supported(int M, int C, int dtype) {
  return (dtype == LDM_BF16 && C == 320 && M > 0) ? 1 : 0;
}

extern "C" int ldm_ffn_geglu(const void* x, int64_t ldx, const void* w1, const float* aux, const void* w2,
                             const float* b2, void* out, int64_t ldo, int M, int C, float eps, int dtype,
                             void* stream) {
  int st = check_common("ldm_ffn_geglu", x, ldx, w1, aux, w2, b2, out, ldo, M, C, C, eps, dtype);
  if (st) return st;
  FfnArgs a;
  fill_common(&a, x, ldx, w1, aux, w2, b2, out, ldo, M, C, C, eps);
  dim3 grid((M + 127) / 128);
  hipLaunchKernelGGL((st_tail_kernel<320, 0, false>), grid, dim3(512), 0, (hipStream_t)stream, a);
  return ldm_launch_status("ldm_ffn_geglu");
}

extern "C" int ldm_st_tail(const void* att, int64_t lda, int K0, const void* wo, const float* bo, const void* r0,
                           int64_t ldr0, const void* w1, const float* aux, const void* w2, const float* b2,
                           const void* wp, const float* bp, const void* r1, int64_t ldr1, void* out, int64_t ldo,
                           int M, int C, float eps, int dtype, void* stream) {
  int st = check_common("ldm_st_tail", att, lda, w1, aux, w2, b2, out, ldo, M, C, K0, eps, dtype);
  if (st) return st;
  LDM_CHECK_ARG(K0 == 384, "ldm_st_tail: attention width K0 = 384 (8 heads of 40 padded to 48) only, got %d", K0);
  LDM_CHECK_ARG(wo && bo && r0 && wp && bp && r1, "ldm_st_tail: null pointer");
  auto al16 = [](const void* q) { return ((uintptr_t)q % 16) == 0; };
  LDM_CHECK_ARG(al16(wo) && al16(bo) && al16(wp) && al16(bp) && ((uintptr_t)r0 % 8) == 0 && ((uintptr_t)r1 % 8) == 0 &&
                    ldr0 % 4 == 0 && ldr1 % 4 == 0 && ldr0 >= C && ldr1 >= C, "ldm_st_tail: alignment / residual strides");
  FfnArgs a;
  fill_common(&a, att, lda, w1, aux, w2, b2, out, ldo, M, C, K0, eps);
  a.wo = (const char*)wo; a.bo = bo; a.r0 = (const char*)r0; a.ldr0 = ldr0; a.wo_bytes = (uint32_t)(C * K0 * 2);
  a.wp = (const char*)wp; a.bp = bp; a.r1 = (const char*)r1; a.ldr1 = ldr1; a.wp_bytes = (uint32_t)(C * C * 2);
  dim3 grid((M + 127) / 128);
  hipLaunchKernelGGL((st_tail_kernel<320, 6, true>), grid, dim3(512), 0, (hipStream_t)stream, a);
  return ldm_launch_status("ldm_st_tail");
}
