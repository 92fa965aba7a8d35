// Persistent ping-pong GEMM / implicit-GEMM 3x3 convolution for gfx950, bf16 (tiles 13 / 14 of
// ldm_gemm).  Same operand formats and LDS image as gemm_kernel.h (A rows and W^T rows
// K-contiguous, 128-byte K-tiles, XOR-swizzled 16-byte chunks, LDS-DMA staging with the hardware
// range check as zero padding); what differs is the schedule around them:
//
//   * a workgroup = 8 waves owns a 256-row panel of the output and walks a RANGE of n-tiles of
//     it.  The 3-stage LDS ring never drains between tiles: the first K-tiles of tile n+1 are
//     already in flight while tile n's last K-tiles are multiplied (a non-persistent launch pays
//     one pipeline fill + drain per tile: ~8 K-tile times for the 5-20 K-tiles of the
//     transformer GEMMs);
//   * the two wave halves are ping-ponged (waves w and w+4 share a SIMD): waves 0-3 run
//     {stage, read fragments of step s, MFMA step s} in a barrier period, waves 4-7 run {MFMA
//     step s-1 from registers, stage, read fragments of step s};
//   * the epilogue of tile n runs right AFTER the first barrier of tile n+1, straight from the
//     accumulators (no LDS round trip, no workgroup barrier): each half's epilogue overlaps the
//     other half's MFMAs.  The MFMA is issued with the operands swapped (D = W_frag x A_frag), so
//     a lane owns ONE output row m and 4 consecutive columns n per 16x16 block; two blocks are
//     exchanged with v_permlane16_swap so that every lane stores 16 contiguous bytes.
//     bias / per-sample addend (timestep embedding) / GELU, SiLU, GEGLU / residual are applied in
//     f32 on the way.
#pragma once
#include "gemm_kernel.h"

namespace ldm_gemm_detail {

struct Gemm3Args {
  const char* a;
  const char* w;
  const float* bias;
  const float* addend;
  const char* residual;
  char* out;
  int64_t lda, ldr, ldc, add_ld;
  uint32_t a_bytes, w_bytes;
  int M, N, K;
  int add_rows;
  int conv, H, W, Cin, OH, OW, stride, upsample, pad;
  int act;
  int ktiles;         // K / 64
  int panels;         // ceil(M / 256)
  int ntiles;         // N / BN
  int nsplit;         // workgroups per panel
  int tiles_per_wg;   // ceil(ntiles / nsplit)
  char* out_t;        // transposed output (EPI bit 5): out_t[(m / rows_t) * stride_t + n * ld_t + m % rows_t]
  int64_t ld_t, stride_t;
  int rows_t;
  int n_split;        // EPI bit 7: columns >= n_split (whole n-tiles) take the transposed store, relative to n_split
  int ntiles1;        // ... n-tiles below n_split
  const float* ln_cs; // EPI bit 6: column sums of the gamma-scaled weights (LayerNorm folded into the product)
  float ln_eps;
#ifdef LDM_TOOLS_BUILD
  int dbg;            // timing ablations (tools build only): 1 = no stores, 2 = no epilogue, 4 = no MFMA, 8 = no staging
#endif
};

// Timing ablations exist only in the tools build (make tools -> libldm_hip_tools.so, -DLDM_TOOLS_BUILD);
// in the product library the tests below are the constant 0 and fold away.
#ifdef LDM_TOOLS_BUILD
#define LDM_G3_DBG(p) ((p).dbg)
#else
#define LDM_G3_DBG(p) 0
#endif

// Epilogue variant, a compile-time constant: a runtime "is there a bias / residual" test around
// each epilogue load makes the compiler branch and drain the memory pipeline per load.
//   bit 0 bias, bit 1 per-group addend, bit 2 residual, bits 3-4 activation (LDM_ACT_* code)
//   bit 5 TRANSPOSED store per group of rows_t rows (the V projection lands directly in the
//         attention kernel's V^T [sample][head dim][token]); plain product only
//   bit 6 LayerNorm of the A ROWS folded into the product (plain rows, K = the normalised width): the
//         host passes W' = bf16(gamma (.) W), bias' = bias + W beta and ln_cs[n] = sum_k W'[n][k]; since
//         LN(x) W^T = rstd (x W'^T - mean ln_cs) + bias', only the per-row mean / rstd are missing, and
//         every wave derives those of ITS 64 rows from the A tiles that pass through LDS anyway during
//         the workgroup's first n-tile (one row per lane, v_dot2c_f32_bf16 against (1, 1) and against
//         itself): no LayerNorm launch, no normalised copy of the rows in HBM.  Statistics are f32
//         sums of the bf16 values (E[x^2] - mean^2, clamped at 0)
//   bit 7 SPLIT output (with bit 5): n-tiles below n_split store row-major into `out`, the n-tiles from n_split on
//         transposed into out_t (column n - n_split): the self-attention's q | k and V^T projections -- two
//         products over the same LayerNorm'ed rows -- as ONE launch (one round of workgroups, one fill / drain
//         instead of two).  The kernel body is instantiated once per side of n_split and a workgroup runs the
//         instance(s) its range of n-tiles needs (a per-n-tile switch inside ONE body kept both epilogues'
//         temporaries alive across the pipeline loop: 256 registers + 92 B of scratch against 190)
constexpr int kEpiBias = 1, kEpiAdd = 2, kEpiRes = 4, kEpiTrans = 32, kEpiLn = 64, kEpiSplit = 128;
constexpr int epi_code(bool bias, bool add, bool res, int act) { return (bias ? 1 : 0) | (add ? 2 : 0) | (res ? 4 : 0) | (act << 3); }

// BN = 32 * TN columns per n-tile; waves 4 (M) x 2 (N); wave tile 64 x (16 TN)
#if defined(__HIP_DEVICE_COMPILE__)
// TR: this workgroup's n-tiles are stored transposed (EPI bit 5; with bit 7 the workgroups of the transposed side)
template <int TN, int MODE, int EPI, bool TR>
__device__ __forceinline__ void gemm3_body(const Gemm3Args& p, char* smem, const int panel, const int nt_begin, const int ntl) {
  constexpr int BM = 256, BN = 32 * TN, WM = 4, WN = 2, NW = 8;
  constexpr int WTM = 64, WTN = 16 * TN, TM = 4;
  constexpr int NIB = BN / 8;
  constexpr int LA = BM / (8 * NW), LB = (NIB + NW - 1) / NW, LBF = NIB / NW;
  constexpr int NL = LA + LB;
  constexpr int ES = 2;
  constexpr int STAGE = (BM + BN) * 128;
  constexpr int NSTAGE = 3;
  // LayerNorm fold: 4 KB behind the ring for the row-statistics exchange between the two waves (wn = 0 / 1)
  // that share 64 rows (ONE __shared__ array: a second object makes the compiler drain vmcnt per step)
  constexpr int LNX = (EPI & 64) ? NW * 64 * 8 : 0;
  static_assert(NSTAGE * STAGE + LNX <= 160 * 1024, "LDS");
  typedef __attribute__((address_space(3))) void* lds_ptr;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int m0 = panel * BM;
  const int nk = p.ktiles;
  const int S = ntl * nk;                                                // pipeline steps

  const __amdgpu_buffer_rsrc_t rsA =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.a), 0, p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.w), 0, p.w_bytes, 0x00020000);

  // ---- per-lane staging geometry (as gemm_kernel.h) -------------------------------------
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
  int b_base[LB];
#pragma unroll
  for (int i = 0; i < LB; ++i) {
    const int row = (i * NW + wave) * 8 + (lane >> 3);
    const int ck = (lane & 7) ^ ((row >> 1) & 7);
    const int n = nt_begin * BN + row;
    b_base[i] = (row < BN) ? (int)((int64_t)n * p.K * ES) + ck * 16 : (int)kOOB;
  }
  const bool b_last = LB == LBF || (LBF * NW + wave) < NIB;
  const int row_pitch = (int)(p.lda * ES);
  const int line_pitch = p.W * row_pitch;
  const int tile_pitch = BN * p.K * ES;             // bytes between consecutive n-tiles of W^T

  // step -> LDS-DMAs of (n-tile `tl` of this workgroup, K-tile kt) into `stage`
  auto issue_tile = [&](int tl, int kt, int stage) {
    char* dA = smem + stage * STAGE + wave * 1024;
    char* dB = dA + BM * 128;
    int kb;
    if (LDM_G3_DBG(p) & 16) {            // timing ablation: no A staging at all (what a halo-staged A would approach)
      if constexpr (MODE != 0) { const int cc = kt / 9; const int tap = kt - cc * 9; kb = tap * p.Cin * ES + cc * 128; }
      else kb = kt * 128;
    } else if constexpr (MODE != 0) {
      const int cc = kt / 9;
      const int tap = kt - cc * 9;
      const int cib = cc * 128;
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
        const int toff = kh * line_pitch + kw * row_pitch + cib;
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
    const uint32_t wb = (uint32_t)kb + (uint32_t)tl * (uint32_t)tile_pitch;
#pragma unroll
    for (int i = 0; i < LBF; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_ptr)(dB + i * NW * 1024), 16,
                                               (uint32_t)b_base[i] + wb, 0, 0, 0);
    if constexpr (LB != LBF) {
      if (b_last)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_ptr)(dB + LBF * NW * 1024), 16,
                                                 (uint32_t)b_base[LBF] + wb, 0, 0, 0);
    }
  };

  f32x4 acc[TM][TN];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  zero_acc();

  const int lr = lane & 15, lh = lane >> 4;
  const int sw = (lr >> 1) & 7;
  int offA[2], offB[2];
#pragma unroll
  for (int kg = 0; kg < 2; ++kg) {
    const int coff = ((kg * 4 + lh) ^ sw) << 4;
    offA[kg] = (wm * WTM + lr) * 128 + coff;
    offB[kg] = BM * 128 + (wn * WTN + lr) * 128 + coff;
  }

  u32x4 fa[2][TM], fb[2][TN];
  if (LDM_G3_DBG(p) & 32) {                                  // defined values for the no-read ablation
#pragma unroll
    for (int kg = 0; kg < 2; ++kg) {
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[kg][i] = u32x4{(uint32_t)lane, 1u, 2u, 3u};
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[kg][j] = u32x4{(uint32_t)lane, 5u, 6u, 7u};
    }
  }
  auto read_frags = [&](const char* cS) {
    if (LDM_G3_DBG(p) & 32) return;                          // timing ablation: no LDS fragment reads
#pragma unroll
    for (int kg = 0; kg < 2; ++kg) {
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[kg][i] = *(const u32x4*)(cS + offA[kg] + i * 16 * 128);
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[kg][j] = *(const u32x4*)(cS + offB[kg] + j * 16 * 128);
    }
  };
  // operands SWAPPED: D[n][m] = sum_k W[n][k] A[m][k]; lane l then holds, for block (i, j),
  // output row m = 16 i + (l & 15) and columns n = 16 j + 4 (l >> 4) + r, r = 0..3
  // prio: the LATE half multiplies at a higher priority than the early half.  Both halves want the
  // SIMD's matrix pipe in the middle of a period (the early half once its fragment reads are back,
  // the late half from the barrier on); sharing it evenly makes both finish together and leaves
  // the late half's fragment reads exposed at the end of the period.
  constexpr bool SPLIT = (EPI & kEpiSplit) != 0;
  auto multiply = [&](int prio) {
    if (LDM_G3_DBG(p) & 4) return;
    if (prio) __builtin_amdgcn_s_setprio(3);
    else __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kg = 0; kg < 2; ++kg)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          if constexpr (TR)                           // natural order: lane = column n, registers = 4 rows m
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[kg][i]),
                                                                __builtin_bit_cast(bf16x8, fb[kg][j]), acc[i][j], 0, 0, 0);
          else
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fb[kg][j]),
                                                                __builtin_bit_cast(bf16x8, fa[kg][i]), acc[i][j], 0, 0, 0);
        }
    __builtin_amdgcn_s_setprio(0);
  };

  // ---- LayerNorm fold: row statistics of this wave's 64 A rows (lane = row) -------------------
  // All 8 physical 16-byte chunks of the row's K-tile are read (the XOR swizzle only permutes them),
  // rotated by row >> 1 so that the 16 lanes one ds_read_b128 pass serves hit 16 different
  // (row parity, chunk) bank groups.
  [[maybe_unused]] float ln_s = 0.f, ln_q = 0.f;
  [[maybe_unused]] auto ln_accum = [&](const char* cS) {
    if constexpr ((EPI & kEpiLn) != 0) {
      typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
      const int row = wm * WTM + lane;
      const char* base = cS + row * 128;
      const int rot = row >> 1;
      const bf2 one = __builtin_bit_cast(bf2, 0x3f803f80u);
#pragma unroll
      for (int j = 0; j < 8; j += 2) {
        const u32x4 c0 = *(const u32x4*)(base + (((j + rot) & 7) << 4));
        const u32x4 c1 = *(const u32x4*)(base + (((j + 1 + rot) & 7) << 4));
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          // (copies first: __builtin_bit_cast applied to the vector ELEMENT expression c0[e] reads element 0
          // for every e with this compiler -- found by a one-hot probe of which elements the statistics see)
          const uint32_t w0 = c0[e], w1 = c1[e];
          const bf2 a0 = __builtin_bit_cast(bf2, w0), a1 = __builtin_bit_cast(bf2, w1);
          ln_s = __builtin_amdgcn_fdot2_f32_bf16(a0, one, ln_s, false);
          ln_q = __builtin_amdgcn_fdot2_f32_bf16(a0, a0, ln_q, false);
          ln_s = __builtin_amdgcn_fdot2_f32_bf16(a1, one, ln_s, false);
          ln_q = __builtin_amdgcn_fdot2_f32_bf16(a1, a1, ln_q, false);
        }
      }
    }
  };

  // The two waves that share 64 rows (wn = 0 / 1) split the K-tiles between them (even / odd) and
  // exchange their partial sums through LDS once, when the first n-tile is complete.
  [[maybe_unused]] auto ln_publish = [&]() {
    if constexpr ((EPI & kEpiLn) != 0) {
      ldm_f32x2 v = {ln_s, ln_q};
      *(ldm_f32x2*)(smem + NSTAGE * STAGE + (wave * 64 + lane) * 8) = v;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // visible to the partner after the next barrier
    }
  };
  [[maybe_unused]] auto ln_combine = [&]() {
    if constexpr ((EPI & kEpiLn) != 0) {
      const ldm_f32x2 o = *(const ldm_f32x2*)(smem + NSTAGE * STAGE + ((wave ^ 1) * 64 + lane) * 8);
      ln_s += o[0]; ln_q += o[1];
    }
  };

  // ---- register epilogue of one n-tile ----------------------------------------------------
  constexpr bool HB = (EPI & kEpiBias) != 0, HA = (EPI & kEpiAdd) != 0, HR = (EPI & kEpiRes) != 0;
  constexpr int ACT = (EPI >> 3) & 3;
  constexpr bool LN = (EPI & kEpiLn) != 0;
  static_assert(!LN || (MODE == 0 && HB), "LayerNorm fold: plain rows, with the folded bias");
  constexpr bool GEGLU = ACT == LDM_ACT_GEGLU;
  static_assert(!GEGLU || TN == 4, "GEGLU: the wave tile must be one 64-row block of the interleaved weights");
  auto epilogue = [&](int tl) {
    if (LDM_G3_DBG(p) & 2) return;
    const int n_w = (nt_begin + tl) * BN + wn * WTN;          // first column of this wave's tile
    const int g = lh;
    [[maybe_unused]] float ln_mu = 0.f, ln_r = 0.f;           // of row wm * 64 + lane (sums complete after n-tile 0)
    if constexpr (LN) {
      // The contraction is PINNED (one fmul per moment, one explicit fma for E[x^2] - mean^2): this lambda is
      // inlined at several places (in the pipeline loops and behind them) and, left to -ffp-contract=fast, the
      // copies fused `q ik - mu mu` differently -- the variance of a row then differed in its last bit between a
      // workgroup's last n-tile and its others, i.e. between two deals of n-tiles to workgroups (found in round 4:
      // 8 - 46 of 25 M outputs on the other side of a bf16 rounding tie)
      const float ik = 1.0f / (float)p.K;
      ln_mu = ln_s * ik;
      const float ex2 = ln_q * ik;
      ln_r = rsqrtf(fmaxf(__builtin_fmaf(-ln_mu, ln_mu, ex2), 0.f) + p.ln_eps);
    }
    if constexpr (TR) {
      const int n_t = SPLIT ? n_w - p.n_split : n_w;         // column of out_t
      // Transposed store: the lane owns column n = n_w + 16 j + (l & 15) and rows 16 i + 4 g + r.  Two
      // row blocks (i, i+1) are exchanged with v_permlane16_swap (8 consecutive rows = 16 bytes per
      // lane), then moved so that the four pieces of 32 consecutive rows of ONE column sit in adjacent
      // lanes: every lane quad writes 64 contiguous bytes of out_t[sample][n][token].
      const int col2 = lane >> 2, pc = lane & 3;
      const int src = (16 * ((pc >> 1) | ((pc & 1) << 1)) + col2) * 4;
#pragma unroll
      for (int ip = 0; ip < TM; ip += 2) {
        const int mrow = m0 + wm * WTM + 16 * ip + 8 * pc;      // first of this lane's 8 rows (after the move)
        const bool valid = mrow < p.M && !(LDM_G3_DBG(p) & 1);
        const int mc = valid ? mrow : 0;
        const int smp = mc / p.rows_t, tok = mc - smp * p.rows_t;
        bf16_t* obase = (bf16_t*)p.out_t + (int64_t)smp * p.stride_t + tok;
        // LayerNorm fold: before the moves the lane owns column n_w + 16 j + (l & 15) and rows
        // 16 i + 4 g + r; the rows' statistics live in the lanes of that number
        [[maybe_unused]] float rmu[2][4], rrs[2][4];
        if constexpr (LN) {
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              rmu[h][r] = __shfl(ln_mu, 16 * (ip + h) + 4 * g + r, 64);
              rrs[h][r] = __shfl(ln_r, 16 * (ip + h) + 4 * g + r, 64);
            }
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          if constexpr (LN) {
            const float cs = p.ln_cs[n_w + 16 * j + lr], bb = p.bias[n_w + 16 * j + lr];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
              for (int r = 0; r < 4; ++r)
                acc[ip + h][j][r] = __builtin_fmaf(rrs[h][r], acc[ip + h][j][r], __builtin_fmaf(-rrs[h][r] * rmu[h][r], cs, bb));
          }
          const uint32_t x0 = pack_bf2(acc[ip][j][0], acc[ip][j][1]), x1 = pack_bf2(acc[ip][j][2], acc[ip][j][3]);
          const uint32_t y0 = pack_bf2(acc[ip + 1][j][0], acc[ip + 1][j][1]), y1 = pack_bf2(acc[ip + 1][j][2], acc[ip + 1][j][3]);
          const auto s0 = __builtin_amdgcn_permlane16_swap(x0, y0, false, false);
          const auto s1 = __builtin_amdgcn_permlane16_swap(x1, y1, false, false);
          u32x4 o;
          o[0] = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)s0[0]);
          o[1] = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)s1[0]);
          o[2] = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)s0[1]);
          o[3] = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)s1[1]);
          if (valid) *(u32x4*)(obase + (int64_t)(n_t + 16 * j + col2) * p.ld_t) = o;
        }
      }
      return;
    }
    // All global loads of the epilogue are issued in batches ahead of their use (beside LDS-DMAs
    // the compiler waits vmcnt(0) at the first use of an ordinary load).
    f32x4 bv[TN];                                            // bias of this lane's 4 columns per block
    if constexpr (HB) {
#pragma unroll
      for (int j = 0; j < TN; ++j) bv[j] = *(const f32x4*)(p.bias + n_w + 16 * j + 4 * g);
    }
    [[maybe_unused]] f32x4 csv[TN];                          // LayerNorm fold: column sums of this lane's columns
    if constexpr (LN) {
#pragma unroll
      for (int j = 0; j < TN; ++j) csv[j] = *(const f32x4*)(p.ln_cs + n_w + 16 * j + 4 * g);
    }
    constexpr int NOB = GEGLU ? TN / 2 : TN;                 // output blocks per row
    const int n_o = GEGLU ? (n_w >> 1) : n_w;                // first OUTPUT column of this wave's tile
    // the residual of ALL the wave tile's row blocks up front: one memory round trip for the tile instead of
    // one per row block (a workgroup with a single n-tile has nothing to hide them behind: + 4.5 us per launch
    // at M = 8192, K = N = 640, tools/dense_probe.py)
    [[maybe_unused]] u32x2 rva[TM][NOB];
    if constexpr (HR) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm * WTM + 16 * i + lr;
        const bf16_t* rr = (const bf16_t*)p.residual + (int64_t)(m < p.M ? m : p.M - 1) * p.ldr + n_o + 4 * g;
#pragma unroll
        for (int j = 0; j < NOB; ++j) rva[i][j] = *(const u32x2*)(rr + 16 * j);
      }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int m = m0 + wm * WTM + 16 * i + lr;
      const bool valid = m < p.M;
      const int mc = valid ? m : p.M - 1;
      f32x4 av[TN];
      if constexpr (HA) {
        const float* ad = p.addend + (int64_t)(mc / p.add_rows) * p.add_ld + n_w + 4 * g;
#pragma unroll
        for (int j = 0; j < TN; ++j) av[j] = *(const f32x4*)(ad + 16 * j);
      }
      u32x2 rv[NOB];
      if constexpr (HR) {
#pragma unroll
        for (int j = 0; j < NOB; ++j) rv[j] = rva[i][j];
      }
      [[maybe_unused]] float mu_i = 0.f, rs_i = 0.f;         // statistics of row 16 i + lr: held by that lane
      [[maybe_unused]] float nm_i = 0.f;                       // -rstd * mean
      if constexpr (LN) {
        mu_i = __shfl(ln_mu, 16 * i + lr, 64);
        rs_i = __shfl(ln_r, 16 * i + lr, 64);
        nm_i = -rs_i * mu_i;
      }
      // value of block j, registers r = 0..3 (columns n_w + 16 j + 4 g + r), before the residual
      auto block = [&](int j, float (&v)[4]) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r];
        if constexpr (LN) {
          // rstd (acc - mean cs) + b' as two fmas: rstd acc + (b' - rstd mean cs)
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = __builtin_fmaf(rs_i, v[r], __builtin_fmaf(nm_i, csv[j][r], bv[j][r]));
        } else if constexpr (HB) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += bv[j][r];
        }
        if constexpr (HA) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += av[j][r];
        }
      };
      auto finish = [&](int jo, float (&v)[4], uint32_t (&pk)[2]) {   // + residual of output block jo, pack
        if constexpr (HR) {
          v[0] += __uint_as_float(rv[jo][0] << 16); v[1] += __uint_as_float(rv[jo][0] & 0xffff0000u);
          v[2] += __uint_as_float(rv[jo][1] << 16); v[3] += __uint_as_float(rv[jo][1] & 0xffff0000u);
        }
        pk[0] = pack_bf2(v[0], v[1]);
        pk[1] = pack_bf2(v[2], v[3]);
      };
      uint32_t pk[NOB][2];
      if constexpr (GEGLU) {
        // wave tile = one 64-row block of the interleaved GEGLU weights: blocks 0, 1 = value,
        // blocks 2, 3 = gate; output columns (n_w / 2) + 16 j + 4 g + r
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          float v[4], gt[4];
          block(j, v);
          block(j + 2, gt);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] *= gelu_erf_f(gt[r]);
          finish(j, v, pk[j]);
        }
      } else {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          float v[4];
          block(j, v);
          if constexpr (ACT == LDM_ACT_GELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = gelu_erf_f(v[r]);
          } else if constexpr (ACT == LDM_ACT_SILU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = silu_f(v[r]);
          }
          finish(j, v, pk[j]);
        }
      }
      // 16-lane groups g: X = block j, Y = block j+1.  After the swaps group 0 holds columns 0-7
      // of block j, group 1 columns 0-7 of block j+1, group 2 columns 8-15 of block j, group 3
      // columns 8-15 of block j+1: 16 contiguous bytes per lane, but the four 16-byte pieces of
      // one row's 64 bytes sit in lanes c, c+16, c+32, c+48.  A store instruction is coalesced
      // per 4 ADJACENT lanes, so the pieces are moved (ds_bpermute: crossbar only, no LDS memory)
      // to lane 4 * row + piece first: every quad then writes 64 contiguous bytes (16 requests
      // per instruction instead of 64).
      const int row2 = lane >> 2, pc = lane & 3;              // after the move: row inside the block, piece
      const int src = (16 * ((pc >> 1) | ((pc & 1) << 1)) + row2) * 4;   // piece 0..3 <- group 0, 2, 1, 3
      const int m2 = m0 + wm * WTM + 16 * i + row2;
      const bool valid2 = m2 < p.M && !(LDM_G3_DBG(p) & 1);
      bf16_t* orow2 = (bf16_t*)p.out + (int64_t)(valid2 ? m2 : 0) * p.ldc + n_o;
#pragma unroll
      for (int j = 0; j + 1 < NOB; j += 2) {
        const auto s0 = __builtin_amdgcn_permlane16_swap(pk[j][0], pk[j + 1][0], false, false);
        const auto s1 = __builtin_amdgcn_permlane16_swap(pk[j][1], pk[j + 1][1], false, false);
        u32x4 o;
        o[0] = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)s0[0]);
        o[1] = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)s1[0]);
        o[2] = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)s0[1]);
        o[3] = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)s1[1]);
        if (valid2) *(u32x4*)(orow2 + 16 * j + 8 * pc) = o;
      }
      if constexpr (NOB & 1) {
        // last block alone: 8 bytes per lane (columns 4 g .. 4 g + 3); same move, pieces in group order
        const int src1 = (16 * pc + row2) * 4;
        u32x2 o;
        o[0] = (uint32_t)__builtin_amdgcn_ds_bpermute(src1, (int)pk[NOB - 1][0]);
        o[1] = (uint32_t)__builtin_amdgcn_ds_bpermute(src1, (int)pk[NOB - 1][1]);
        if (valid2) *(u32x2*)(orow2 + 16 * (NOB - 1) + 4 * pc) = o;
      }
    }
  };

  // ---- pipeline ---------------------------------------------------------------------------
  int is_tl = 0, is_kt = 0;                          // next step to stage
  auto issue_next = [&](int stage) {
    if (!(LDM_G3_DBG(p) & 8)) issue_tile(is_tl, is_kt, stage);
    if (++is_kt == nk) { is_kt = 0; ++is_tl; }
  };
  issue_next(0);
  if (S > 1) issue_next(1);
  int st = 0, ck = 0, ctl = 0;                       // ring slot, K-tile and n-tile of step s
  const bool late = wave >= NW / 2;
  // One barrier period per step s.  Early half: stage step s+2, read step s, multiply step s.
  // Late half: multiply step s-1 (fragments kept in registers), stage, read step s, drain the
  // reads (the stage may be overwritten after the next barrier).  A tile that completed with
  // step s-1 gets its epilogue right after the barrier of period s in BOTH halves, so each
  // half's epilogue runs beside the other half's MFMAs.  (Two loops, not one with `late` tests
  // inside: the merged loop keeps the fragments live across the epilogue and spills.)
  auto wait_step = [&](int s) {                      // this wave's LDS-DMAs of step s have landed
    if (s + 1 < S && (LDM_G3_DBG(p) & 16)) {
      if (b_last) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LB) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LB - 1) : "memory");
    } else if (s + 1 < S) {
      if (b_last) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL - 1) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  };
  if (!late) {
    for (int s = 0; s < S; ++s) {
      wait_step(s);
      __builtin_amdgcn_s_barrier();
      if constexpr (LN) { if (s == nk) ln_combine(); }
      if (ck == 0 && s > 0) { epilogue(ctl - 1); zero_acc(); }
      int sn = st + 2; sn = sn >= NSTAGE ? sn - NSTAGE : sn;
      if (s + 2 < S) issue_next(sn);
      read_frags(smem + st * STAGE);
      if constexpr (LN) {
        if (ctl == 0) {
          if ((ck & 1) == wn) ln_accum(smem + st * STAGE);
          if (ck == nk - 1) ln_publish();
        }
      }
      multiply(0);
      st = st + 1 == NSTAGE ? 0 : st + 1;
      if (++ck == nk) { ck = 0; ++ctl; }
    }
  } else {
    for (int s = 0; s < S; ++s) {
      wait_step(s);
      __builtin_amdgcn_s_barrier();
      if (s > 0) multiply(1);                        // step s - 1, fragments kept in registers
      if constexpr (LN) { if (s == nk) ln_combine(); }
      if (ck == 0 && s > 0) { epilogue(ctl - 1); zero_acc(); }
      int sn = st + 2; sn = sn >= NSTAGE ? sn - NSTAGE : sn;
      if (s + 2 < S) issue_next(sn);
      read_frags(smem + st * STAGE);
      if constexpr (LN) {
        if (ctl == 0) {
          if ((ck & 1) == wn) ln_accum(smem + st * STAGE);
          if (ck == nk - 1) ln_publish();
        }
      }
      st = st + 1 == NSTAGE ? 0 : st + 1;
      if (++ck == nk) { ck = 0; ++ctl; }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    }
    multiply(1);
  }
  if constexpr (LN) {
    if (ntl == 1) {                                  // one n-tile: no barrier period followed the exchange yet
      __builtin_amdgcn_s_barrier();
      ln_combine();
    }
  }
  epilogue(ntl - 1);
}
#endif

template <int TN, int MODE, int EPI>
__global__ __launch_bounds__(512) void gemm3_kernel(Gemm3Args p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BM = 256, BN = 32 * TN;
  constexpr int LNX = (EPI & 64) ? 8 * 64 * 8 : 0;
  __shared__ __attribute__((aligned(16))) char smem[3 * (BM + BN) * 128 + LNX];
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x;
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
  }
  const int panel = bid / p.nsplit, sp = bid - panel * p.nsplit;
  if constexpr ((EPI & kEpiSplit) != 0) {
    // a workgroup walks a contiguous range of the panel's n-tiles; the part below ntiles1 is row-major, the part
    // from it on transposed.  A range that straddles ntiles1 runs the two instances of the body one after the other
    // (the pipeline drains in between and the second instance takes its own row statistics): the n-tiles can then
    // be dealt evenly whatever the two sides' sizes (6 + 3 n-tiles on 2 workgroups per panel at M = 32768)
    const int nt_begin = sp * p.tiles_per_wg;
    const int nt_end = min(p.ntiles, nt_begin + p.tiles_per_wg);
    const int a_end = min(nt_end, p.ntiles1);
    if (nt_begin < a_end) gemm3_body<TN, MODE, EPI, false>(p, smem, panel, nt_begin, a_end - nt_begin);
    const int b_begin = max(nt_begin, p.ntiles1);
    if (b_begin < nt_end) {
      if (nt_begin < a_end) __syncthreads();               // every wave is done with the ring of the first instance
      gemm3_body<TN, MODE, EPI, true>(p, smem, panel, b_begin, nt_end - b_begin);
    }
  } else {
    const int nt_begin = sp * p.tiles_per_wg;
    const int ntl = min(p.ntiles, nt_begin + p.tiles_per_wg) - nt_begin;   // n-tiles of this workgroup
    if (ntl > 0)                                                           // (whole workgroup: uniform)
      gemm3_body<TN, MODE, EPI, (EPI & kEpiTrans) != 0>(p, smem, panel, nt_begin, ntl);
  }
#endif
}

// returns false when the (tile, epilogue) combination is not instantiated
template <int MODE>
bool launch_gemm3(int tn, int epi, const Gemm3Args& a, dim3 grid, hipStream_t s);
// the LayerNorm-fold variants (EPI bit 6, plain rows), gemm3_inst_ln.hip
bool launch_gemm3_ln(int tn, int epi, const Gemm3Args& a, dim3 grid, hipStream_t s);

}  // namespace ldm_gemm_detail
