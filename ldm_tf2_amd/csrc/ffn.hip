// Row-panel kernel for the BasicTransformerBlock / SpatialTransformer at the 320-channel level, from the
// self-attention's output to the block's output (unet.py:310-313, :323-325, :335-338, :357-365), bf16:
//
//   [FRONT] h1 = r0 + bo1 + Wo1 . att1                    self-attention output projection + residual (:310)
//           q  = Wq' . LayerNorm(h1)                      cross-attention query projection (:311, :262)
//   [XATT ] att = softmax(q K^T) V                        cross-attention against the <= 80 context keys (:273-291)
//   [PRE ]  h  = h1 + bo + Wo . att                       cross-attention output projection + residual (:312)
//           y  = h + b2 + W2 . ( a * gelu(g) ),  (a | g) = W1 . LayerNorm(h) + b1       feed-forward (:313)
//   [POST]  out = r1 + bp + Wp . y                        proj_out of the SpatialTransformer + its residual (:363-365)
//
// Entry points by how much of the chain they take: ldm_ffn_geglu (the feed-forward), ldm_st_tail (PRE .. POST),
// ldm_st_xtail (XATT .. POST), ldm_st_block (all of it: one launch instead of seven GEMM launches and an attention
// launch).
//
// ONE launch per 128-row panel of the residual stream.
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
// Schedules measured against this one (same probe, feed-forward alone, 108 us):
//   * two wave groups half a period apart (load phase / matrix phase, two barriers per period, s_setprio): 119.7 us;
//   * the same rotation with ONE barrier per period (both groups run the same stream L(s) M(s) L(s+1) ..., group 0
//     synchronises in front of L(s), group 1 between L(s) and M(s), so that between two barriers one executes
//     L(s) M(s) and the other M(s-1) L(s); same waits, requests and registers as in lockstep): parity green, 118.9 us
//     (ldm_st_block 176.5 vs 158) -- one wave's MFMAs beside the other's LDS reads on a SIMD is not faster than both
//     reading, then both multiplying; what the period waits for is neither of the two;
//   * 4 waves with wave tiles 64 x 64 and the whole 512-register file (1/3 less LDS read traffic): 119.7 us -- one
//     wave per SIMD has nobody to hide its LDS latency behind;
//   * all 8 B fragments read before the first MFMA: no change (109 us);
//   * B fragments read one step ahead (second register set; the reads of step s + 1 issued between the barrier and
//     the MFMAs of step s; request distance 3): parity green, 110.2 us (ldm_st_block 165.5 vs 158 us; 60 - 148 B of
//     scratch): the LDS reads are not what the period waits for;
//   * fully unrolled phase loops (the compiler's choice without `#pragma unroll 1`): the same in a hot loop, but
//     in situ, where every launch starts with a cold instruction cache, straight-line code is fetched at ~1 us
//     per period: the 33 extra periods of ldm_st_block cost 37 us unrolled, 17 us as loops.
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
  // FRONT: the self-attention's o-projection (fw [C][K0], fb, residual r0) and the cross-attention's LayerNorm-folded
  // query projection (qw [K0][C] gamma-folded, qcs its column sums, qb the folded bias) run first; the o-projection
  // of PRE then takes its residual from the rows this kernel parked in `out`
  const char* fw;
  const float* fb;
  const char* qw;
  const float* qcs;
  const float* qb;
  uint32_t fw_bytes, qw_bytes, out_bytes;
  // XATT: x = the cross-attention's QUERY rows (ldm_attention_ms layout); keys [R][Tk][K0] / values^T [R][K0][ldv]
  const char* ctx_k;
  const char* ctx_vt;
  int Tk, ldv, T;      // keys per sample, V^T row stride, query rows per sample (multiple of 128)
  // The INPUT row tensors (x, r0, r1) may hold only the first `in_rows` rows of the panel range: output row m then
  // reads input row m - in_rows (m >= in_rows).  This is the classifier-free-guidance pair of the DDIM loop
  // (model_runners.py:449-452: the U-Net runs on concat([xt, xt])): until the first cross-attention the two halves
  // of the batch are the same numbers, so everything in front of this launch ran once, on half the rows.
  int in_rows;
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
constexpr int kNW = 8;

// C: channels (320); KT0: K-tiles of the PRE product (attention width / 64; 0 = no PRE phase);
// KIND 0: feed-forward only, 1: + proj_out (POST);
// 3 (XATT): as 1, but the input rows are the cross-attention's QUERIES and the attention itself (unet.py:273-291
// against the <= 80 context keys of the panel's sample) runs first, in place in the LDS panel;
// 4 (FRONT): as 3, but the input rows are the SELF-attention's output: its o-projection + residual and the
// LayerNorm-folded query projection (unet.py:310-311) run first and leave the queries in the panel
template <int N> __device__ __forceinline__ void wait_vm() {      // s_waitcnt vmcnt(N): at most N vector-memory ops in flight
  static_assert(N >= 0 && N <= 10, "vmcnt");
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if constexpr (N == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
}

// NW: waves per workgroup, (NW / 2) x 2 over the 128 x 128 period tile: 8 (wave tile 32 x 64, 2 waves per SIMD) or
// 4 (wave tile 64 x 64, one wave per SIMD with the whole 512-register file: 1/3 less LDS read traffic per MFMA)
// BM: rows of a panel, 128 or 64.  The 64-row form keeps the LDS image of the 128-row one (tiles of 128 rows, the
// upper half unused) and halves the wave tile (16 x 64): half the MFMA work per barrier period at the same
// skeleton, for launches whose 128-row panels would leave half the chip idle (M = 16384: 128 panels on 256 CUs).
template <int C, int KT0, int KIND, int NW, int BM = 128>
__global__ __launch_bounds__(64 * NW) void st_tail_kernel(FfnArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr bool PRE = KT0 > 0, POST = KIND != 0, XATT = KIND >= 3, FRONT = KIND == 4;
  static_assert(BM == 128 || (BM == 64 && NW == 8), "panel height");
  constexpr int TM = BM / (NW / 2) / 16, RW = 16 * TM;   // 16-row blocks / rows of a wave tile
  constexpr int NH = 16 / NW;                            // LDS-DMA instructions per staged tile and wave
  static_assert(NW == 8 || NW == 4, "waves");          // (NW = 4 is not instantiated: see the file comment)
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
  static_assert(C % 64 == 0 && OFF_R + NSTAGE * STG <= 160 * 1024 && KT0 <= KT1 + 1 && KIND != 2, "LDS");
  typedef __attribute__((address_space(3))) void* lds_ptr;
  constexpr int XIMG = 80 * 768;               // XATT: the sample's keys / values^T image (80 x 768 B = 384 x 160 B)
  constexpr int SMEM = (XATT && OFF_R + XIMG > OFF_R + NSTAGE * STG) ? OFF_R + XIMG : OFF_R + NSTAGE * STG;
  static_assert(SMEM <= 160 * 1024 && (!XATT || KT0 == 6), "LDS");
  __shared__ __attribute__((aligned(16))) char smem[SMEM];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;     // wave tile: rows RW wm .., columns 64 wn .. +63 of a 128 x 128 tile
  const int lr = lane & 15, lh = lane >> 4;
  const int m0 = blockIdx.x * BM;

  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.x), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.w1), 0, p.w1_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.w2), 0, p.w2_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsAux = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.aux), 0, p.aux_bytes, 0x00020000);
  [[maybe_unused]] const __amdgpu_buffer_rsrc_t rsOut =
      __builtin_amdgcn_make_buffer_rsrc(p.out, 0, FRONT ? p.out_bytes : 0u, 0x00020000);
  [[maybe_unused]] const __amdgpu_buffer_rsrc_t rsWo =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(PRE ? p.wo : p.w1), 0, PRE ? p.wo_bytes : 0u, 0x00020000);
  [[maybe_unused]] const __amdgpu_buffer_rsrc_t rsWp =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(POST ? p.wp : p.w1), 0, POST ? p.wp_bytes : 0u, 0x00020000);

  // ---- staging geometry: a tile is 16 LDS-DMA instructions of 8 rows; wave w issues instructions w, w + NW, ...
  // lane l lands at row 8 i + (l >> 3), 16-byte slot l & 7, and fetches chunk (l & 7) ^ ((row >> 1) & 7).
  int srow[NH], sck[NH];
#pragma unroll
  for (int h = 0; h < NH; ++h) {
    srow[h] = (wave + NW * h) * 8 + (lane >> 3);
    sck[h] = ((lane & 7) ^ ((srow[h] >> 1) & 7)) * 16;
  }
  // the panel's input rows: KT1 tiles of x, or (PRE) KT0 tiles of the attention output (the last one may lie in
  // the hidden tile's region, which is free until the feed-forward starts)
  constexpr int KTP = PRE ? KT0 : KT1;
  {
    uint32_t xo[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      const int m = m0 + srow[h];
      xo[h] = (m < p.M && srow[h] < BM) ? (uint32_t)((int64_t)(m >= p.in_rows ? m - p.in_rows : m) * p.ldx * 2) + sck[h] : kOOBf;
    }
#pragma unroll
    for (int kt = 0; kt < KTP; ++kt)
#pragma unroll
      for (int h = 0; h < NH; ++h)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_ptr)(smem + kt * TILE + (wave + NW * h) * 1024), 16,
                                                 xo[h] == kOOBf ? kOOBf : xo[h] + kt * 128, 0, 0, 0);
  }
  // a weight tile whose rows are the output columns 128 pp .. of an N = C product, K bytes kbytes ..
  auto issue_rows = [&](const __amdgpu_buffer_rsrc_t& rs, char* dst, int pp, int kbytes, int row_pitch, int nmax) {
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      const int n = 128 * pp + srow[h];
      const uint32_t off = n < nmax ? (uint32_t)(n * row_pitch + kbytes + sck[h]) : kOOBf;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(dst + (wave + NW * h) * 1024), 16, off, 0, 0, 0);
    }
  };
  // weight tile of step s into ring slot `slot`
  auto issue = [&](int s, int slot) {
    char* dst = smem + OFF_R + slot * STG;
    if (PRE && s < S0) {
      const int kt = s / NP2, pp = s - kt * NP2;
      issue_rows(rsWo, dst, pp, kt * 128, KT0 * 128, C);
    } else if (s < S0 + S1) {
      const int f = s - S0;
      const int c = f / SPC, u = f - c * SPC;
      if (u < KT1) {
#pragma unroll
        for (int h = 0; h < NH; ++h) {
          const uint32_t off = (uint32_t)((128 * c + srow[h]) * (C * 2) + u * 128 + sck[h]);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW1, (lds_ptr)(dst + (wave + NW * h) * 1024), 16, off, 0, 0, 0);
        }
        if (u == KT1 - 1)     // the chunk's (column sums | folded bias): 1 KB, every wave writes the same bytes
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsAux, (lds_ptr)(dst + TILE), 16, (uint32_t)(c * 1024 + lane * 16), 0, 0, 0);
      } else {
        issue_rows(rsW2, dst, u - KT1, c * 128, HID * 2, C);
      }
    } else {
      const int g = s - S0 - S1;
      const int kt = g / NP2, pp = g - kt * NP2;
      issue_rows(rsWp, dst, pp, kt * 128, C * 2, C);
    }
  };
  // DMAs a wave issues for step s (one more where the chunk's aux KB rides along)
  auto n_issued = [&](int s) {
    const int f = s - S0;
    return (f >= 0 && f < S1 && (f % SPC) == KT1 - 1) ? NH + 1 : NH;
  };

  // ---- fragment addresses ------------------------------------------------------------------------
  int offA[2][TM], offB[2][4];       // [k group][block]: byte offsets inside a 128 x 128-byte tile
#pragma unroll
  for (int kg = 0; kg < 2; ++kg) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int row = RW * wm + 16 * i + lr;
      offA[kg][i] = row * 128 + (((kg * 4 + lh) ^ ((row >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = 64 * wn + 16 * j + lr;
      offB[kg][j] = row * 128 + (((kg * 4 + lh) ^ ((row >> 1) & 7)) << 4);
    }
  }

  f32x4 acc1[TM][4], acc2[NP2][TM][4];
  auto zero_acc2 = [&]() {
#pragma unroll
    for (int pp = 0; pp < NP2; ++pp)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc2[pp][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  };

  // Operands swapped (D = W_frag x A_frag): the lane owns row 16 i + lr and columns 16 j + 4 lh + r.
  // A fragments (resident panel / hidden tile) are read by read_a BEFORE the period's barrier where they do not
  // depend on it; B fragments one k group at a time (6 fragments live: 96 + 32 accumulator registers leave no more).
  u32x4 fa[2][TM];
  auto read_a = [&](const char* sa) {
#pragma unroll
    for (int kg = 0; kg < 2; ++kg)
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[kg][i] = *(const u32x4*)(sa + offA[kg][i]);
  };
  auto mma_tile = [&](const char* sb, f32x4 (&acc)[TM][4]) {
    if (LDM_FFN_DBG(p) & 1) return;
    u32x4 fb[2][4];                 // all 8 B fragments up front: the second k group's reads overlap the first's MFMAs
#pragma unroll
    for (int kg = 0; kg < 2; ++kg)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        fb[kg][j] = (LDM_FFN_DBG(p) & 8) ? u32x4{(uint32_t)lane, 1u, 2u, 3u} : *(const u32x4*)(sb + offB[kg][j]);
#pragma unroll
    for (int kg = 0; kg < 2; ++kg)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fb[kg][j]),
                                                              __builtin_bit_cast(bf16x8, fa[kg][i]), acc[i][j], 0, 0, 0);
  };

  int slot = 0;
  // one barrier period: this wave's DMAs of step s have landed (those of step s + 1 may stay in flight), every
  // wave is past its reads of the slot step s + 2 goes into; `drain_lds`: this wave's LDS writes are visible
  auto begin_period = [&](int s, bool drain_lds) {
    if (s + 1 < S) {
      if (n_issued(s + 1) == NH + 1) wait_vm<NH + 1>();
      else wait_vm<NH>();
    } else {
      wait_vm<0>();
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

  // ---- LayerNorm statistics of this wave's 32 rows from the resident panel -------------------------
  float rs_i[TM], nm_i[TM];
  auto ln_stats = [&]() {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    const bf2 one = __builtin_bit_cast(bf2, 0x3f803f80u);
    if constexpr (RW == 16) {
      // 16 rows per wave: lane l takes row l & 15 and the 16-byte chunks c = (l >> 4) mod 4 of every K-tile; the four
      // lanes of a row add up by two xor-shuffles, so every lane ends with ITS row's statistics (row lr = l & 15)
      const int row = RW * wm + lr;
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int kt = 0; kt < KT1; ++kt) {
        const char* base = smem + kt * TILE + row * 128;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const u32x4 cc = *(const u32x4*)(base + (((2 * lh + j + (row >> 1)) & 7) << 4));
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const uint32_t w = cc[e];
            const bf2 a = __builtin_bit_cast(bf2, w);
            s1 = __builtin_amdgcn_fdot2_f32_bf16(a, one, s1, false);
            s2 = __builtin_amdgcn_fdot2_f32_bf16(a, a, s2, false);
          }
        }
      }
      s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
      s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
      const float ik = 1.0f / (float)C;
      const float mu = s1 * ik;
      rs_i[0] = rsqrtf(fmaxf(s2 * ik - mu * mu, 0.f) + p.eps);
      nm_i[0] = -rs_i[0] * mu;
    } else {
    const int half = lane >> 5;                                // lanes l and l + 32 split the row's chunks
    float ln_mu[RW / 32], ln_rs[RW / 32];
#pragma unroll
    for (int rg = 0; rg < RW / 32; ++rg) {
      const int row = RW * wm + 32 * rg + (lane & 31);
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
      ln_mu[rg] = s1 * ik;
      ln_rs[rg] = rsqrtf(fmaxf(s2 * ik - ln_mu[rg] * ln_mu[rg], 0.f) + p.eps);
    }
    // statistics of the rows this lane owns in the MFMA layout: row 16 i + lr of the wave tile
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const float mu = __shfl(ln_mu[i >> 1], 16 * (i & 1) + lr, 64);
      rs_i[i] = __shfl(ln_rs[i >> 1], 16 * (i & 1) + lr, 64);
      nm_i[i] = -rs_i[i] * mu;
    }
    }
  };

  if constexpr (!XATT) {       // (XATT: the ring's region first holds the keys / values of the attention)
    issue(0, 0);
    issue(1, 1);
  }

  if constexpr (FRONT) {
    // ---- h1 = r0 + fb + Wf . att1 (self-attention o-projection), q = Wq' . LayerNorm(h1) ------------------
    // the same ring, its own 33 periods; h1 is parked in this panel's rows of `out` (this lane reads back in PRE
    // exactly what it stores here), the queries replace it in the panel
    constexpr int SA = KT0 * NP2, SF = SA + KT1 * NP2;
    const __amdgpu_buffer_rsrc_t rsF = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.fw), 0, p.fw_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsQ = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.qw), 0, p.qw_bytes, 0x00020000);
    auto issue_f = [&](int s, int sl) {
      char* dst = smem + OFF_R + sl * STG;
      if (s < SA) {
        const int kt = s / NP2, pp = s - kt * NP2;
        issue_rows(rsF, dst, pp, kt * 128, KT0 * 128, C);
      } else {
        const int g = s - SA;
        const int kt = g / NP2, pp = g - kt * NP2;
        issue_rows(rsQ, dst, pp, kt * 128, C * 2, KT0 * 64);
      }
    };
    int sl = 0;
    // `landed`: step s's DMAs are known to have landed (steps SA, SA + 1: the h1 epilogue's global loads are
    // younger and have returned; its stores may still be in flight and must not be waited for)
    auto period_f = [&](int s, bool landed) {
      if (s + 1 >= SF) wait_vm<0>();
      else if (!landed) wait_vm<NH>();
      __builtin_amdgcn_s_barrier();
    };
    auto ahead_f = [&](int s) {
      int sn = sl + 2; sn = sn >= NSTAGE ? sn - NSTAGE : sn;
      if (s + 2 < SF) issue_f(s + 2, sn);
    };
    issue_f(0, 0);
    issue_f(1, 1);
    zero_acc2();
    wait_vm<2 * NH>();
#pragma unroll 1
    for (int kt = 0; kt < KT0; ++kt) {
#pragma unroll
      for (int pp = 0; pp < NP2; ++pp) {
        const int s = kt * NP2 + pp;
        period_f(s, false);
        if (pp == 0) read_a(smem + kt * TILE);
        ahead_f(s);
        if (128 * pp + 64 * wn < C) mma_tile(smem + OFF_R + sl * STG, acc2[pp]);
        sl = sl + 1 == NSTAGE ? 0 : sl + 1;
      }
    }
    __builtin_amdgcn_s_barrier();                        // every wave is done with the attention rows
    // (the global loads below drain this wave's DMA queue: steps SA, SA + 1 have landed when they return)
#pragma unroll
    for (int pp = 0; pp < NP2; ++pp) {
      if (128 * pp + 64 * wn >= C) continue;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = RW * wm + 16 * i + lr;
        const int m = min(m0 + row, p.M - 1);
        const bf16_t* rr = (const bf16_t*)p.r0 + (int64_t)(m >= p.in_rows ? m - p.in_rows : m) * p.ldr0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int n = 128 * pp + 64 * wn + 16 * j + 4 * lh;
          const f32x4 bb = *(const f32x4*)(p.fb + n);
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
          // rows past M: out of range, dropped -- the instruction is issued regardless
          __builtin_amdgcn_raw_buffer_store_b64(pk, rsOut, m0 + row < p.M ? (uint32_t)(((int64_t)m * p.ldo + n) * 2) : kOOBf, 0, 0);
        }
      }
    }
    zero_acc2();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    ln_stats();
#pragma unroll 1
    for (int kt = 0; kt < KT1; ++kt) {
#pragma unroll
      for (int pp = 0; pp < NP2; ++pp) {
        const int s = SA + kt * NP2 + pp;
        period_f(s, kt == 0 && pp < 2);
        if (pp == 0) read_a(smem + kt * TILE);
        ahead_f(s);
        mma_tile(smem + OFF_R + sl * STG, acc2[pp]);
        sl = sl + 1 == NSTAGE ? 0 : sl + 1;
      }
    }
    __builtin_amdgcn_s_barrier();                        // every wave is done with h1 (and with the ring)
#pragma unroll
    for (int pp = 0; pp < NP2; ++pp)
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = RW * wm + 16 * i + lr;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int n = 128 * pp + 64 * wn + 16 * j + 4 * lh;
          const f32x4 bb = *(const f32x4*)(p.qb + n);
          const f32x4 cs = *(const f32x4*)(p.qcs + n);
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r)
            v[r] = __builtin_fmaf(rs_i[i], acc2[pp][i][j][r], __builtin_fmaf(nm_i[i], cs[r], bb[r]));
          u32x2 pk;
          pk[0] = pack_bf2(v[0], v[1]);
          pk[1] = pack_bf2(v[2], v[3]);
          *(u32x2*)panel_cell(row, n) = pk;
        }
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // published by the attention phase's first barrier
  }

  if constexpr (XATT) {
    // ---- cross-attention of the panel's 128 query rows, in place: q -> softmax(q K^T) V ----------------
    // One key tile (Tk <= 80), so a plain softmax.  Wave (wm, wn): rows 32 wm .. +31, heads 4 wn .. +3 (48 = 40 + 8
    // padded dims each).  16x16x16 MFMAs chain through registers: S^T = K q^T leaves lane (lr, lh) with the logits
    // of row lr for keys 16 jk + 4 lh + r, which (exponentiated, bf16) is the B operand of O^T = V^T P^T, which
    // leaves the lane with dims 16 jd + 4 lh + r of row lr: the very LDS cells its query fragments came from.
    // ldm_attention_ms's layout: the logits arrive in the exp2 domain, keys carry 1 and values^T a row of ones at
    // padded dim 40, so O[40] is the row sum of the ROUNDED probabilities.
    // The sample's keys (80 x 768 B), then its values^T (384 x 160 B) are LDS-DMA'd into the (still idle) ring
    // region, 16-byte chunks permuted on the source side so that the 8-byte fragment reads are conflict-free
    // (fragments straight from global memory made this phase 17 us: 16 rows x 32 B per load instruction).
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    const int sample = m0 / p.T;
    const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(p.ctx_k) + (int64_t)sample * p.Tk * (KT0 * 128), 0, (uint32_t)(p.Tk * KT0 * 128), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(p.ctx_vt) + (int64_t)sample * (KT0 * 64) * p.ldv * 2, 0, (uint32_t)(KT0 * 64 * p.ldv * 2), 0x00020000);
    char* img = smem + OFF_R;
#pragma unroll
    for (int it = 0; it < (XIMG / 1024 + NW - 1) / NW; ++it) {
      const int n = it * NW + wave;                    // (wave-uniform)
      if (n < XIMG / 1024) {
        const int g = 64 * n + lane, key = g / 48, pos = g - 48 * key;
        const int c = (pos & ~15) | ((pos ^ key) & 15);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, (lds_ptr)(img + n * 1024), 16,
                                                 key < p.Tk ? (uint32_t)(key * (KT0 * 128) + c * 16) : kOOBf, 0, 0, 0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    u32x2 pf[4][TM][5];
#pragma unroll
    for (int hh = 0; hh < 4; ++hh) {
      __builtin_amdgcn_sched_barrier(0);              // one head's fragments at a time (registers)
      const int head = 4 * wn + hh;
      u32x2 kf[5][3];
#pragma unroll
      for (int jk = 0; jk < 5; ++jk)
#pragma unroll
        for (int kd = 0; kd < 3; ++kd) {    // key 16 jk + lr (rows past Tk are zero), dims 16 kd + 4 lh ..
          const int key = 16 * jk + lr, c = 6 * head + 2 * kd + (lh >> 1);
          kf[jk][kd] = *(const u32x2*)(img + key * (KT0 * 128) + (((c & ~15) | ((c ^ key) & 15)) << 4) + (lh & 1) * 8);
        }
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = RW * wm + 16 * i + lr;
        u32x2 qf[3];
#pragma unroll
        for (int kd = 0; kd < 3; ++kd) {
          const int n = 48 * head + 16 * kd + 4 * lh;
          qf[kd] = *(const u32x2*)(smem + (n >> 6) * TILE + row * 128 + ((((n & 63) >> 3) ^ ((row >> 1) & 7)) << 4) + (n & 7) * 2);
        }
        f32x4 sc[5];
        float mx = -INFINITY;
#pragma unroll
        for (int jk = 0; jk < 5; ++jk) {
          sc[jk] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kd = 0; kd < 3; ++kd)
            sc[jk] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, kf[jk][kd]),
                                                               __builtin_bit_cast(s16x4, qf[kd]), sc[jk], 0, 0, 0);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (16 * jk + 4 * lh + r >= p.Tk) sc[jk][r] = -INFINITY;
            mx = fmaxf(mx, sc[jk][r]);
          }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
#pragma unroll
        for (int jk = 0; jk < 5; ++jk) {
          pf[hh][i][jk][0] = pack_bf2(__builtin_amdgcn_exp2f(sc[jk][0] - mx), __builtin_amdgcn_exp2f(sc[jk][1] - mx));
          pf[hh][i][jk][1] = pack_bf2(__builtin_amdgcn_exp2f(sc[jk][2] - mx), __builtin_amdgcn_exp2f(sc[jk][3] - mx));
          // (pinned here: the compiler otherwise sinks all 640 exponentials below the barrier, next to their use,
          // and keeps every logit alive until then)
          asm volatile("" : "+v"(pf[hh][i][jk][0]), "+v"(pf[hh][i][jk][1]));
        }
      }
    }
    __builtin_amdgcn_s_barrier();                     // every wave is done with the keys
#pragma unroll
    for (int it = 0; it < (XIMG / 1024 + NW - 1) / NW; ++it) {
      const int n = it * NW + wave;
      if (n < XIMG / 1024) {
        const int g = 64 * n + lane, dim = g / 10, pos = g - 10 * dim;
        const int c = pos ^ ((dim >> 3) & 1);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, (lds_ptr)(img + n * 1024), 16,
                                                 (uint32_t)(dim * p.ldv * 2 + c * 16), 0, 0, 0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int hh = 0; hh < 4; ++hh) {
      __builtin_amdgcn_sched_barrier(0);
      const int head = 4 * wn + hh;
      u32x2 vf[3][5];
#pragma unroll
      for (int jd = 0; jd < 3; ++jd)
#pragma unroll
        for (int jk = 0; jk < 5; ++jk) {    // dim 16 jd + lr, keys 16 jk + 4 lh ..; pad keys masked (may hold anything)
          const int dim = 48 * head + 16 * jd + lr, c = (2 * jk + (lh >> 1)) ^ ((dim >> 3) & 1);
          u32x2 v = *(const u32x2*)(img + dim * 160 + c * 16 + (lh & 1) * 8);
          const int k0 = 16 * jk + 4 * lh;
          if (k0 + 1 >= p.Tk) v[0] &= (k0 < p.Tk) ? 0x0000ffffu : 0u;
          if (k0 + 3 >= p.Tk) v[1] &= (k0 + 2 < p.Tk) ? 0x0000ffffu : 0u;
          vf[jd][jk] = v;
        }
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = RW * wm + 16 * i + lr;
        f32x4 o[3];
#pragma unroll
        for (int jd = 0; jd < 3; ++jd) {
          o[jd] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int jk = 0; jk < 5; ++jk)
            o[jd] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, vf[jd][jk]),
                                                              __builtin_bit_cast(s16x4, pf[hh][i][jk]), o[jd], 0, 0, 0);
        }
        // row sum: dim 40 = block 2, lh = 2, r = 0
        const float inv = __builtin_amdgcn_rcpf(__shfl(o[2][0], 32 + lr, 64));
#pragma unroll
        for (int jd = 0; jd < 3; ++jd) {
          u32x2 pk;
          pk[0] = pack_bf2(o[jd][0] * inv, o[jd][1] * inv);
          pk[1] = pack_bf2(o[jd][2] * inv, o[jd][3] * inv);
          if (jd == 2 && lh >= 2) pk = u32x2{0u, 0u};      // padded dims 40 .. 47
          const int n = 48 * head + 16 * jd + 4 * lh;
          *(u32x2*)(smem + (n >> 6) * TILE + row * 128 + ((((n & 63) >> 3) ^ ((row >> 1) & 7)) << 4) + (n & 7) * 2) = pk;
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                     // the values^T image is free: the ring starts
    issue(0, 0);
    issue(1, 1);
  }

  zero_acc2();

  // ---- PRE: h = r0 + bo + Wo . att ------------------------------------------------------------------
  if constexpr (PRE) {
    wait_vm<2 * NH>();                                   // the panel's DMAs are older than the two weight tiles
#pragma unroll 1
    for (int kt = 0; kt < KT0; ++kt) {
#pragma unroll
      for (int pp = 0; pp < NP2; ++pp) {
        const int s = kt * NP2 + pp;
        begin_period(s, false);                          // (XATT: the lgkmcnt(0) above precedes the first barrier)
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
      for (int i = 0; i < TM; ++i) {
        const int row = RW * wm + 16 * i + lr;
        const int m = min(m0 + row, p.M - 1);
        const bf16_t* rr = (const bf16_t*)p.r0 + (int64_t)(FRONT ? m : (m >= p.in_rows ? m - p.in_rows : m)) * p.ldr0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int n = 128 * pp + 64 * wn + 16 * j + 4 * lh;
          const f32x4 bb = *(const f32x4*)(p.bo + n);
          u32x2 rv;
          if constexpr (FRONT)
            rv = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rsOut, (uint32_t)(((int64_t)m * p.ldo + n) * 2), 0, 0));
          else rv = *(const u32x2*)(rr + n);
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
    if (n_issued(0) + n_issued(1) == 2 * NH) wait_vm<2 * NH>();
    else wait_vm<2 * NH + 1>();
    __builtin_amdgcn_s_barrier();
  }

  ln_stats();

  // ---- feed-forward: per hidden chunk 5 periods of W1 (-> GEGLU -> hidden tile) and 3 of W2 -----------
  // (the u loop is unrolled: the period's role and the accumulator piece are compile-time per instance)
#pragma unroll 1
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
          for (int i = 0; i < TM; ++i)
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
            for (int i = 0; i < TM; ++i) {
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
              const int row = RW * wm + 16 * i + lr;
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
      for (int i = 0; i < TM; ++i) {
        const int row = RW * wm + 16 * i + lr;
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
#pragma unroll 1
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
    for (int i = 0; i < TM; ++i) {
      const int row = RW * wm + 16 * i + lr;
      const int m = m0 + row;
      if (m >= p.M) continue;
      bf16_t* orow = (bf16_t*)p.out + (int64_t)m * p.ldo;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = 128 * pp + 64 * wn + 16 * j + 4 * lh;
        const f32x4 bb = *(const f32x4*)((POST ? p.bp : p.b2) + n);
        u32x2 rv;
        if constexpr (POST) rv = *(const u32x2*)((const bf16_t*)p.r1 + (int64_t)(m >= p.in_rows ? m - p.in_rows : m) * p.ldr1 + n);
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
  a->M = M; a->eps = eps; a->in_rows = M;
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
  hipLaunchKernelGGL((st_tail_kernel<320, 0, 0, kNW>), grid, dim3(64 * kNW), 0, (hipStream_t)stream, a);
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
  hipLaunchKernelGGL((st_tail_kernel<320, 6, 1, kNW>), grid, dim3(64 * kNW), 0, (hipStream_t)stream, a);
  return ldm_launch_status("ldm_st_tail");
}

extern "C" int ldm_st_xtail(const void* q, int64_t ldq, int K0, const void* ctx_k, const void* ctx_vt, int Tk, int ldv,
                            int T, const void* wo, const float* bo, const void* r0, int64_t ldr0, const void* w1,
                            const float* aux, const void* w2, const float* b2, const void* wp, const float* bp,
                            const void* r1, int64_t ldr1, void* out, int64_t ldo, int M, int C, float eps, int dtype,
                            void* stream) {
  int st = check_common("ldm_st_xtail", q, ldq, w1, aux, w2, b2, out, ldo, M, C, K0, eps, dtype);
  if (st) return st;
  LDM_CHECK_ARG(K0 == 384, "ldm_st_xtail: attention width K0 = 384 (8 heads of 40 padded to 48) only, got %d", K0);
  LDM_CHECK_ARG(wo && bo && r0 && wp && bp && r1 && ctx_k && ctx_vt, "ldm_st_xtail: null pointer");
  LDM_CHECK_ARG(Tk >= 1 && Tk <= 80 && ldv >= 80 && ldv % 4 == 0 && T > 0 && T % 128 == 0 && M % T == 0,
                "ldm_st_xtail: needs Tk <= 80 keys (V^T rows of >= 80), query rows per sample a multiple of 128 "
                "(Tk=%d ldv=%d T=%d M=%d)", Tk, ldv, T, M);
  auto al16 = [](const void* x) { return ((uintptr_t)x % 16) == 0; };
  LDM_CHECK_ARG(al16(wo) && al16(bo) && al16(wp) && al16(bp) && ((uintptr_t)r0 % 8) == 0 && ((uintptr_t)r1 % 8) == 0 &&
                    al16(ctx_k) && al16(ctx_vt) && ldv % 8 == 0 && ldr0 % 4 == 0 && ldr1 % 4 == 0 &&
                    ldr0 >= C && ldr1 >= C, "ldm_st_xtail: alignment / residual strides");
  FfnArgs a;
  fill_common(&a, q, ldq, w1, aux, w2, b2, out, ldo, M, C, K0, eps);
  a.wo = (const char*)wo; a.bo = bo; a.r0 = (const char*)r0; a.ldr0 = ldr0; a.wo_bytes = (uint32_t)(C * K0 * 2);
  a.wp = (const char*)wp; a.bp = bp; a.r1 = (const char*)r1; a.ldr1 = ldr1; a.wp_bytes = (uint32_t)(C * C * 2);
  a.ctx_k = (const char*)ctx_k; a.ctx_vt = (const char*)ctx_vt; a.Tk = Tk; a.ldv = ldv; a.T = T;
  dim3 grid((M + 127) / 128);
  hipLaunchKernelGGL((st_tail_kernel<320, 6, 3, kNW>), grid, dim3(64 * kNW), 0, (hipStream_t)stream, a);
  return ldm_launch_status("ldm_st_xtail");
}

extern "C" int ldm_st_block(const void* att, int64_t lda, int K0, const void* wo1, const float* bo1, const void* r0,
                            int64_t ldr0, const void* wq, const float* qcs, const float* qb, const void* ctx_k,
                            const void* ctx_vt, int Tk, int ldv, int T, const void* wo2, const float* bo2,
                            const void* w1, const float* aux, const void* w2, const float* b2, const void* wp,
                            const float* bp, const void* r1, int64_t ldr1, void* out, int64_t ldo, int M, int in_rows,
                            int C, float eps, int dtype, void* stream) {
  if (in_rows <= 0) in_rows = M;
  LDM_CHECK_ARG(in_rows == M || (in_rows % 128 == 0 && M == 2 * in_rows && T > 0 && in_rows % T == 0),
                "ldm_st_block: in_rows=%d must be M or M / 2 (whole 128-row panels, whole samples), M=%d", in_rows, M);
  int st = check_common("ldm_st_block", att, lda, w1, aux, w2, b2, out, ldo, in_rows, C, K0, eps, dtype);
  if (st) return st;
  LDM_CHECK_ARG(K0 == 384, "ldm_st_block: attention width K0 = 384 (8 heads of 40 padded to 48) only, got %d", K0);
  LDM_CHECK_ARG(wo1 && bo1 && r0 && wq && qcs && qb && wo2 && bo2 && wp && bp && r1 && ctx_k && ctx_vt, "ldm_st_block: null pointer");
  LDM_CHECK_ARG(Tk >= 1 && Tk <= 80 && ldv >= 80 && ldv % 8 == 0 && T > 0 && T % 128 == 0 && M % T == 0,
                "ldm_st_block: needs Tk <= 80 keys (V^T rows of >= 80), query rows per sample a multiple of 128 "
                "(Tk=%d ldv=%d T=%d M=%d)", Tk, ldv, T, M);
  auto al16 = [](const void* x) { return ((uintptr_t)x % 16) == 0; };
  LDM_CHECK_ARG(al16(wo1) && al16(bo1) && al16(wq) && al16(qcs) && al16(qb) && al16(wo2) && al16(bo2) && al16(wp) &&
                    al16(bp) && al16(ctx_k) && al16(ctx_vt) && ((uintptr_t)r0 % 8) == 0 && ((uintptr_t)r1 % 8) == 0 &&
                    ldr0 % 4 == 0 && ldr1 % 4 == 0 && ldr0 >= C && ldr1 >= C, "ldm_st_block: alignment / residual strides");
  LDM_CHECK_ARG((((int64_t)M - 1) * ldo + C) * 2 < (1ll << 31), "ldm_st_block: output extent must be < 2 GiB");
  LDM_CHECK_ARG(out != r0 && out != r1 && out != att, "ldm_st_block: out is also scratch, it must not alias an input");
  FfnArgs a;
  fill_common(&a, att, lda, w1, aux, w2, b2, out, ldo, in_rows, C, K0, eps);     // (x extent: in_rows rows)
  a.M = M; a.in_rows = in_rows;
  a.fw = (const char*)wo1; a.fb = bo1; a.r0 = (const char*)r0; a.ldr0 = ldr0; a.fw_bytes = (uint32_t)(C * K0 * 2);
  a.qw = (const char*)wq; a.qcs = qcs; a.qb = qb; a.qw_bytes = (uint32_t)(K0 * C * 2);
  a.out_bytes = (uint32_t)((((int64_t)M - 1) * ldo + C) * 2);
  a.wo = (const char*)wo2; a.bo = bo2; a.wo_bytes = (uint32_t)(C * K0 * 2);
  a.wp = (const char*)wp; a.bp = bp; a.r1 = (const char*)r1; a.ldr1 = ldr1; a.wp_bytes = (uint32_t)(C * C * 2);
  a.ctx_k = (const char*)ctx_k; a.ctx_vt = (const char*)ctx_vt; a.Tk = Tk; a.ldv = ldv; a.T = T;
  // 128-row panels from 192 of them on (3/4 of the CUs busy); below that 64-row panels: twice the workgroups
  if ((M + 127) / 128 >= 192) {
    dim3 grid((M + 127) / 128);
    hipLaunchKernelGGL((st_tail_kernel<320, 6, 4, kNW, 128>), grid, dim3(64 * kNW), 0, (hipStream_t)stream, a);
  } else {
    dim3 grid((M + 63) / 64);
    hipLaunchKernelGGL((st_tail_kernel<320, 6, 4, kNW, 64>), grid, dim3(64 * kNW), 0, (hipStream_t)stream, a);
  }
  return ldm_launch_status("ldm_st_block");
}
