// Fused multi-head attention (flash style) on MFMA for gfx950.
//
// Transposed formulation so that softmax state is lane-local:
//   S^T = K . Q^T   (A operand = K tile rows (keys), B operand = Q (lane = query))
//   O^T = V^T . P^T (A operand = V^T rows (head dims), B operand = P = the S^T
//                    accumulator itself, repacked in registers -- no LDS trip)
// A 32x32 MFMA accumulator holds its COLUMN on the lane, so lane l owns query
// (l & 31): its running max / sum and the rescale of its O^T column are plain
// per-lane scalars; the only cross-lane step is one exchange with lane l^32.
// For bf16 the rows of the K tile are read in the order kappa(i) = i with bits
// 2 and 3 swapped, which makes the accumulator's register order the natural key
// order of the following P.V MFMA (16-byte V^T fragments, no permutes).
//
// One workgroup = 4 waves = 128 queries of one (batch, head); K and V^T tiles of
// KT keys are staged global -> registers -> LDS with the next tile's loads in
// flight during the current tile's MFMAs.  Logits are never written to memory.
#include "common.h"
#include <atomic>
#include <stdlib.h>

namespace {

struct AttnArgs {
  const char* q; const char* k; const char* vt; char* out;
  int64_t ldq, q_bs, ldk, k_bs, ldvt, vt_bs, ldo, o_bs;
  int heads, Tq, Tk;
  float scale;
};

__device__ __forceinline__ void mma32a(f32x16& acc, const u32x4& a, const u32x4& b, bf16_t) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a),
                                                __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}
__device__ __forceinline__ void mma32a(f32x16& acc, const u32x4& a, const u32x4& b, float) {
#pragma unroll
  for (int j = 0; j < 4; ++j)
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a[j]), __uint_as_float(b[j]), acc,
                                               0, 0, 0);
}


// Keeps the first `nvalid` elements of a 16-byte chunk of V^T and zeroes the rest: the columns
// [Tk, ldvt) of vt are padding the kernel must not depend on (P is 0 there, but 0 x NaN = NaN in the
// MFMA, so a caller's uninitialised padding would poison the output).  Only the chunk that crosses Tk
// pays for it.
template <typename T>
__device__ __forceinline__ u32x4 keep_first(u32x4 v, int nvalid) {
  constexpr int EPW = 4 / (int)sizeof(T);          // elements per 32-bit word
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const int left = nvalid - w * EPW;              // valid elements from this word on
    if (left <= 0) v[w] = 0u;
    else if (EPW == 2 && left == 1) v[w] &= 0xffffu;
  }
  return v;
}

template <typename T> struct AttnTraits;
template <> struct AttnTraits<bf16_t> {
  static constexpr int NPC = 2;   // P chunks (k-steps) per 32-key sub-tile
  __device__ static __forceinline__ int kappa(int i) {  // swap bits 2 and 3
    return (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1);
  }
  // P chunk pc of a sub-tile: registers 8pc..8pc+7 -> 8 bf16
  __device__ static __forceinline__ u32x4 pchunk(const f32x16& s, int pc) {
    u32x4 c;
#pragma unroll
    for (int e = 0; e < 4; ++e) c[e] = pack_bf2(s[8 * pc + 2 * e], s[8 * pc + 2 * e + 1]);
    return c;
  }
};
template <> struct AttnTraits<float> {
  static constexpr int NPC = 4;
  __device__ static __forceinline__ int kappa(int i) { return i; }
  __device__ static __forceinline__ u32x4 pchunk(const f32x16& s, int pc) {
    u32x4 c;
#pragma unroll
    for (int e = 0; e < 4; ++e) c[e] = __float_as_uint(s[4 * pc + e]);
    return c;
  }
};

// SP = padded head size (multiple of 16: the padding lives in zero weight rows), KT = keys
// per LDS tile.  The O^T accumulator is built from 32-row MFMA tiles, so when SP is not a
// multiple of 32 the last tile's upper rows read unstaged LDS: they only ever feed output
// rows >= SP, which are never stored.
// MS ("matrix-side softmax", bf16, SP = 48 for the 40-wide heads): the padded head dims are put to work.
//   * dim RD = 40 of K holds 1.0 and dim RD of Q holds -m (the query's current reference point, a bf16
//     value), and the logits arrive in the exp2 domain (scale * log2(e) folded into the query
//     projection's weights): the first MFMA returns  q.k * scale * log2(e) - m  directly -- no fma per
//     logit;
//   * row RD of V^T holds 1.0: row RD of O^T accumulates sum_k P[k], the softmax denominator, from the
//     same bf16 P the numerator uses -- no add per logit.
// Per logit that leaves max + exp2 (+ the bf16 pack): the d = 40 self-attention was VALU-bound with
// 704 softmax cycles against 448 MFMA cycles per 64-key tile and wave.  The ones in K / V^T come from the
// producing projections' biases (the host puts 1.0 into the padded rows), -m is written into the Q
// fragment in registers; the reference only moves when some query's tile maximum exceeds it by 2^8
// (as before), and then by an amount that keeps it a bf16 number, so numerator and denominator see
// exactly the same reference.
// NWV / PF: waves per workgroup (4 or 8: 128 or 256 queries share one K / V^T tile) and the prefetch
// distance of the register-staged tiles (1 or 2).  The 4-wave form keeps ONE tile in flight while one is
// multiplied: at d = 40 a tile is ~0.5 us of work per workgroup against 1-2 us of memory latency under
// load, and the four resident workgroups of a CU all wait the same way (the kernel ran 3x above its
// MFMA / VALU bound).  With 8 waves a tile costs each thread half the staging registers, so TWO tiles
// fit in flight within the same register budget (T = 1024, d = 40, same box: 115 -> 110 us plain, 97 -> 86 us
// with the matrix-side softmax).  A ping-ponged form on top of it -- the late half of the 8 waves defers
// O += V^T P by one barrier period (P kept in registers, three tile buffers in LDS, one barrier per tile) so
// that one wave's softmax runs beside its SIMD partner's MFMAs -- was built, passed every parity test and
// measured SLOWER (90.8 vs 84.4 us): removed again.  The PMC anatomy (tools/attn_pmc_probe.py,
// profiles/r03_probes.txt) shows the waves parked at waits 45 % of their cycles with the matrix pipe 31 %
// busy: the dependent chain LDS read -> MFMA -> exp2 -> pack -> LDS read -> MFMA inside one wave is what a
// tile costs, whatever the partner does.
template <typename T, int SP, int KT, bool MS = false, int NWV = 4, int PF = 1>
// Waves per SIMD: the softmax (VALU, exp2) and the two MFMA phases of different waves overlap,
// so residency pays: measured 130 -> 109 -> 91 us at T = 1024, Sp = 48 for 2 -> 3 -> 4 waves per
// SIMD (5 spills 144 bytes per lane and loses again); the wide heads keep the compiler's choice.
__global__ __launch_bounds__(64 * NWV, (sizeof(T) == 2 ? (SP == 32 ? 3 : SP <= 64 ? 4 : 1) : 1)) void attn_kernel(AttnArgs p) {
  constexpr int NT = 64 * NWV;
  static_assert((NWV == 4 || NWV == 8) && (PF == 1 || PF == 2), "attention workgroup shape");
  using TR = AttnTraits<T>;
  constexpr int EPC = Elem<T>::kPerChunk;
  constexpr int NPC = TR::NPC;
  constexpr int NSUB = KT / 32;
  constexpr int DCH = SP / EPC;                 // 16-byte chunks per K row
  constexpr int VCH = KT / EPC;                 // 16-byte chunks per V^T row
  constexpr int NKG = SP * (int)sizeof(T) / 32; // 32-byte k groups over the head dim
  constexpr int ND = (SP + 31) / 32;            // O^T tiles
  constexpr int KRS = SP * (int)sizeof(T) + 16; // padded LDS row strides (odd * 16 B)
  constexpr int VRS = KT * (int)sizeof(T) + 16;
  constexpr int CK = (KT * DCH + NT - 1) / NT;
  constexpr int CV = (SP * VCH + NT - 1) / NT;
  static_assert(SP % 16 == 0 && KT % 32 == 0 && (SP * (int)sizeof(T)) % 32 == 0, "attention tile");

  __shared__ __attribute__((aligned(16))) char smem[KT * KRS + ND * 32 * VRS];
  char* sK = smem;
  char* sV = smem + KT * KRS;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int head = blockIdx.y, b = blockIdx.z;
  const int q0 = blockIdx.x * (32 * NWV) + wave * 32;

  const T* Q = (const T*)p.q + (int64_t)b * p.q_bs + (int64_t)head * SP;
  const T* Kp = (const T*)p.k + (int64_t)b * p.k_bs + (int64_t)head * SP;
  const T* Vt = (const T*)p.vt + (int64_t)b * p.vt_bs + (int64_t)head * SP * p.ldvt;

  // Q fragments: B operand, lane = query
  u32x4 qf[NKG];
  {
    const int qi = min(q0 + lr, p.Tq - 1);
    const T* qr = Q + (int64_t)qi * p.ldq;
#pragma unroll
    for (int g = 0; g < NKG; ++g) qf[g] = *(const u32x4*)(qr + (2 * g + lh) * EPC);
  }

  f32x16 o[ND];
#pragma unroll
  for (int d = 0; d < ND; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
  constexpr int RD = 40;                         // MS: the padded head dim that carries -m / 1 / the row sums
  static_assert(!MS || (sizeof(T) == 2 && SP == 48), "matrix-side softmax: bf16, 40-wide heads padded to 48");
  float m_run = MS ? 0.f : -INFINITY, l_run = 0.f;   // running max (log2 domain) and sum; MS: Q[RD] = -0 initially
  const float c2 = p.scale * 1.4426950408889634f;  // scale * log2(e)

  u32x4 rkA[CK], rvA[CV];
  [[maybe_unused]] u32x4 rkB[PF == 2 ? CK : 1], rvB[PF == 2 ? CV : 1];   // second tile in flight (PF = 2)
  auto load_tile = [&](int kt0, u32x4 (&rk)[CK], u32x4 (&rv)[CV]) {
#pragma unroll
    for (int i = 0; i < CK; ++i) {
      const int id = tid + i * NT;
      const int key = id / DCH, dc = id - key * DCH;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (id < KT * DCH && kt0 + key < p.Tk)
        v = *(const u32x4*)(Kp + (int64_t)(kt0 + key) * p.ldk + dc * EPC);
      rk[i] = v;
    }
#pragma unroll
    for (int i = 0; i < CV; ++i) {
      const int id = tid + i * NT;
      const int dim = id / VCH, kc = id - dim * VCH;
      u32x4 v = {0u, 0u, 0u, 0u};
      const int left = p.Tk - (kt0 + kc * EPC);        // keys of this chunk that exist
      if (id < SP * VCH && left > 0) {
        v = *(const u32x4*)(Vt + (int64_t)dim * p.ldvt + kt0 + kc * EPC);
        if (left < EPC) v = keep_first<T>(v, left);
      }
      rv[i] = v;
    }
  };
  auto store_tile = [&](const u32x4 (&rk)[CK], const u32x4 (&rv)[CV]) {
#pragma unroll
    for (int i = 0; i < CK; ++i) {
      const int id = tid + i * NT;
      const int key = id / DCH, dc = id - key * DCH;
      if (id < KT * DCH) *(u32x4*)(sK + key * KRS + dc * 16) = rk[i];
    }
#pragma unroll
    for (int i = 0; i < CV; ++i) {
      const int id = tid + i * NT;
      const int dim = id / VCH, kc = id - dim * VCH;
      if (id < SP * VCH) *(u32x4*)(sV + dim * VRS + kc * 16) = rv[i];
    }
  };

  const int krow = TR::kappa(lr);
  const int ntiles = (p.Tk + KT - 1) / KT;
  // one staged tile in LDS -> S^T, softmax, O^T update
  auto tile_body = [&](int t) {
    const int kt0 = t * KT;

    // ---- S^T = K . Q^T ---------------------------------------------------------
    f32x16 s[NSUB];
#pragma unroll
    for (int j = 0; j < NSUB; ++j) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[j][r] = 0.f;
      const char* kr = sK + (j * 32 + krow) * KRS + lh * 16;
#pragma unroll
      for (int g = 0; g < NKG; ++g) {
        const u32x4 kf = *(const u32x4*)(kr + g * 32);
        mma32a(s[j], kf, qf[g], T());
      }
    }
    // ---- online softmax (per lane = per query) ------------------------------------
    // The softmax VALU work outweighs the MFMAs at these head sizes, so it is kept to
    // max + fma + exp2 + add per logit: the scale (applied after q.k^T, unet.py:281) is
    // folded with log2(e) into the exp2 argument (scale > 0 commutes with max); keys are
    // masked only in the tile that crosses Tk; the running max is only advanced -- and O,
    // l rescaled -- when some query's max grew by more than 2^8 (softmax is invariant to
    // the reference point; P then stays <= 256, harmless in bf16/f32).
    if (kt0 + KT > p.Tk) {   // wave-uniform
#pragma unroll
      for (int j = 0; j < NSUB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kt0 + j * 32 + TR::kappa((r & 3) + 8 * (r >> 2) + 4 * lh);
          if (key >= p.Tk) s[j][r] = -INFINITY;
        }
    }
    if constexpr (MS) {
      // s = logit - m_run already (exp2 domain)
      float mt = -INFINITY;
#pragma unroll
      for (int j = 0; j < NSUB; ++j)
#pragma unroll
        for (int r = 0; r < 16; r += 2) mt = fmaxf(fmaxf(mt, s[j][r]), s[j][r + 1]);   // v_max3_f32
      mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
      if (t == 0 || !__all(mt <= 8.0f)) {   // wave-uniform
        // move the reference: to the first tile's maximum (either direction), afterwards only up
        const float want = m_run + (t == 0 ? mt : fmaxf(mt, 0.f));
        const float m_new = bf2f(f2bf(want));                 // the reference stays a bf16 number
        const float delta = m_new - m_run;                    // exact (difference of two bf16 numbers)
        m_run = m_new;
        if (t > 0) {
          const float alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
          for (int d = 0; d < ND; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[d][r] *= alpha;    // (row RD, the denominator, included)
        }
#pragma unroll
        for (int j = 0; j < NSUB; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) s[j][r] -= delta;
        // Q fragment element of dim RD: k group RD / 16, lane half (RD % 16) / 8, element RD % 8
        if (lh == (RD % 16) / 8) {
          uint32_t w = qf[RD / 16][(RD % 8) / 2];
          const uint32_t nb = (uint32_t)f2bf(-m_new);
          w = (RD % 2) ? ((w & 0x0000ffffu) | (nb << 16)) : ((w & 0xffff0000u) | nb);
          qf[RD / 16][(RD % 8) / 2] = w;
        }
      }
#pragma unroll
      for (int j = 0; j < NSUB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[j][r] = __builtin_amdgcn_exp2f(s[j][r]);
    } else {
    float mt = -INFINITY;
#pragma unroll
    for (int j = 0; j < NSUB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) mt = fmaxf(mt, s[j][r]);
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64)) * c2;
    if (!__all(mt <= m_run + 8.0f)) {   // wave-uniform; always taken for the first tile
      const float m_new = fmaxf(m_run, mt);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int d = 0; d < ND; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
    }
    float ls = 0.f;
#pragma unroll
    for (int j = 0; j < NSUB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(s[j][r], c2, -m_run));
        s[j][r] = e;
        ls += e;
      }
    l_run += ls;
    }
    // ---- O^T += V^T . P^T -------------------------------------------------------
#pragma unroll
    for (int j = 0; j < NSUB; ++j)
#pragma unroll
      for (int pc = 0; pc < NPC; ++pc) {
        const u32x4 pf = TR::pchunk(s[j], pc);
        const int coff = (j * 2 * NPC + 2 * pc + lh) * 16;
#pragma unroll
        for (int d = 0; d < ND; ++d) {
          const u32x4 vf = *(const u32x4*)(sV + (d * 32 + lr) * VRS + coff);
          mma32a(o[d], vf, pf, T());
        }
      }
  };

  load_tile(0, rkA, rvA);
  if constexpr (PF == 2) {
    if (ntiles > 1) load_tile(KT, rkB, rvB);
    for (int t = 0; t < ntiles; t += 2) {
      __syncthreads();
      store_tile(rkA, rvA);
      __syncthreads();
      if (t + 2 < ntiles) load_tile((t + 2) * KT, rkA, rvA);
      tile_body(t);
      if (t + 1 < ntiles) {                           // (uniform)
        __syncthreads();
        store_tile(rkB, rvB);
        __syncthreads();
        if (t + 3 < ntiles) load_tile((t + 3) * KT, rkB, rvB);
        tile_body(t + 1);
      }
    }
  } else {
    for (int t = 0; t < ntiles; ++t) {
      __syncthreads();
      store_tile(rkA, rvA);
      __syncthreads();
      if (t + 1 < ntiles) load_tile((t + 1) * KT, rkA, rvA);
      tile_body(t);
    }
  }

  float l_tot;
  if constexpr (MS) {
    // row RD of O^T: tile RD / 32, tile row RD % 32 = (r & 3) + 8 (r >> 2) + 4 lh -> lanes with lh = 0
    constexpr int RR = RD % 32;
    static_assert((RR % 8) < 4, "row RD must live in the lh = 0 half");
    l_tot = __shfl(o[RD / 32][(RR % 4) + 4 * (RR / 8)], lane & 31, 64);
  } else {
    l_tot = l_run + __shfl_xor(l_run, 32, 64);
  }
  const float inv = 1.0f / l_tot;
  const int qi = q0 + lr;
  if (qi < p.Tq) {
    T* orow = (T*)p.out + (int64_t)b * p.o_bs + (int64_t)qi * p.ldo + (int64_t)head * SP;
#pragma unroll
    for (int d = 0; d < ND; ++d)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int dim = d * 32 + 8 * r4 + 4 * lh;
        if (dim >= SP) continue;                  // rows of the partial last tile
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = o[d][r4 * 4 + e] * inv;
        if constexpr (MS) {                         // the denominator's row is padding of the output: zero
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (dim + e == RD) ? 0.f : v[e];
        }
        if constexpr (sizeof(T) == 2) {
          u32x2 pk; pk[0] = pack_bf2(v[0], v[1]); pk[1] = pack_bf2(v[2], v[3]);
          *(u32x2*)(orow + dim) = pk;
        } else {
          f32x4 pk = {v[0], v[1], v[2], v[3]};
          *(f32x4*)(orow + dim) = pk;
        }
      }
  }
}



// ---- wide heads (SP = 512: the autoencoder's single-head attention, autoencoder.py:74-97) ----------
// A lane-owned O^T column of 512 dims would need 256 accumulator registers, so the head dim is
// SPLIT over the 4 waves of a workgroup: all four waves serve the SAME 32 queries, wave w owns dims
// [128 w, 128 w + 128).  Per 32-key tile: each wave forms the partial logits of its slice
// (S_w^T = K[:, slice] . Q[:, slice]^T), the four partials are summed through LDS in wave order (so every
// wave holds the identical full S^T), the online softmax runs redundantly per wave (lane-local, as
// above), and each wave accumulates its own 128 rows of O^T = V^T . P^T.  Logits never leave the chip
// (the GEMM + softmax + GEMM path materialised [B, T, T] floats: 268 MB at B = 4, T = 4096).
// This block runs once per image (0.1 % of the decoder's FLOPs): staging is the simple
// load -> barrier -> compute form, not pipelined.
template <typename T>
__global__ __launch_bounds__(256) void attn_wide_kernel(AttnArgs p) {
  using TR = AttnTraits<T>;
  constexpr int EPC = Elem<T>::kPerChunk;
  constexpr int NPC = TR::NPC;
  constexpr int SP = 512, SPW = 128, KT = 32;
  constexpr int ES = (int)sizeof(T);
  constexpr int DCH = SP / EPC;                    // 16-byte chunks per K row
  constexpr int VCH = KT / EPC;                    // 16-byte chunks per V^T row
  constexpr int NKGW = SPW * ES / 32;              // 32-byte k groups over this wave's slice
  constexpr int NDW = SPW / 32;                    // O^T tiles of this wave
  constexpr int KRS = SP * ES + 16;                // padded LDS row strides (odd * 16 B)
  constexpr int VRS = KT * ES + 16;
  constexpr int CK = KT * DCH / 256;               // staging chunks per thread
  constexpr int CV = SP * VCH / 256;
  static_assert((KT * DCH) % 256 == 0 && (SP * VCH) % 256 == 0, "staging");
  extern __shared__ __attribute__((aligned(16))) char smem_w[];
  char* sK = smem_w;
  char* sV = sK + KT * KRS;
  float* sX = (float*)(sV + SP * VRS);             // [4 waves][64 lanes][16] partial logits

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int head = blockIdx.y, b = blockIdx.z;
  const int q0 = blockIdx.x * 32;
  const T* Q = (const T*)p.q + (int64_t)b * p.q_bs + (int64_t)head * SP;
  const T* Kp = (const T*)p.k + (int64_t)b * p.k_bs + (int64_t)head * SP;
  const T* Vt = (const T*)p.vt + (int64_t)b * p.vt_bs + (int64_t)head * SP * p.ldvt;

  u32x4 qf[NKGW];                                  // this wave's slice of the query rows (B operand)
  {
    const int qi = min(q0 + lr, p.Tq - 1);
    const T* qr = Q + (int64_t)qi * p.ldq + wave * SPW;
#pragma unroll
    for (int g = 0; g < NKGW; ++g) qf[g] = *(const u32x4*)(qr + (2 * g + lh) * EPC);
  }
  f32x16 o[NDW];
#pragma unroll
  for (int d = 0; d < NDW; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const float c2 = p.scale * 1.4426950408889634f;
  const int krow = TR::kappa(lr);
  const int ntiles = (p.Tk + KT - 1) / KT;
  for (int t = 0; t < ntiles; ++t) {
    const int kt0 = t * KT;
    __syncthreads();                               // previous tile fully consumed
#pragma unroll
    for (int i = 0; i < CK; ++i) {
      const int id = tid + i * 256;
      const int key = id / DCH, dc = id - key * DCH;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (kt0 + key < p.Tk) v = *(const u32x4*)(Kp + (int64_t)(kt0 + key) * p.ldk + dc * EPC);
      *(u32x4*)(sK + key * KRS + dc * 16) = v;
    }
#pragma unroll
    for (int i = 0; i < CV; ++i) {
      const int id = tid + i * 256;
      const int dim = id / VCH, kc = id - dim * VCH;
      u32x4 v = {0u, 0u, 0u, 0u};
      const int left = p.Tk - (kt0 + kc * EPC);        // keys of this chunk that exist
      if (left > 0) {
        v = *(const u32x4*)(Vt + (int64_t)dim * p.ldvt + kt0 + kc * EPC);
        if (left < EPC) v = keep_first<T>(v, left);
      }
      *(u32x4*)(sV + dim * VRS + kc * 16) = v;
    }
    __syncthreads();
    // partial logits of this wave's 128 dims
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
    {
      const char* kr = sK + krow * KRS + wave * SPW * ES + lh * 16;
#pragma unroll
      for (int g = 0; g < NKGW; ++g) {
        const u32x4 kf = *(const u32x4*)(kr + g * 32);
        mma32a(s, kf, qf[g], T());
      }
    }
    // sum of the four partials, in wave order (identical in every wave)
    {
      float* mine = sX + (wave * 64 + lane) * 16;
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const f32x4 v = {s[4 * r4], s[4 * r4 + 1], s[4 * r4 + 2], s[4 * r4 + 3]};
        *(f32x4*)(mine + 4 * r4) = v;
      }
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const float* src = sX + (w * 64 + lane) * 16;
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const f32x4 v = *(const f32x4*)(src + 4 * r4);
#pragma unroll
          for (int e = 0; e < 4; ++e) s[4 * r4 + e] += v[e];
        }
      }
    }
    if (kt0 + KT > p.Tk) {   // wave-uniform
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kt0 + TR::kappa((r & 3) + 8 * (r >> 2) + 4 * lh);
        if (key >= p.Tk) s[r] = -INFINITY;
      }
    }
    float mt = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) mt = fmaxf(mt, s[r]);
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64)) * c2;
    if (!__all(mt <= m_run + 8.0f)) {
      const float m_new = fmaxf(m_run, mt);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int d = 0; d < NDW; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
    }
    float ls = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], c2, -m_run));
      s[r] = e;
      ls += e;
    }
    l_run += ls;
    // O^T[slice] += V^T[slice] . P^T
#pragma unroll
    for (int pc = 0; pc < NPC; ++pc) {
      const u32x4 pf = TR::pchunk(s, pc);
      const int coff = (2 * pc + lh) * 16;
#pragma unroll
      for (int d = 0; d < NDW; ++d) {
        const u32x4 vf = *(const u32x4*)(sV + (wave * SPW + d * 32 + lr) * VRS + coff);
        mma32a(o[d], vf, pf, T());
      }
    }
  }
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  const int qi = q0 + lr;
  if (qi < p.Tq) {
    T* orow = (T*)p.out + (int64_t)b * p.o_bs + (int64_t)qi * p.ldo + (int64_t)head * SP + wave * SPW;
#pragma unroll
    for (int d = 0; d < NDW; ++d)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int dim = d * 32 + 8 * r4 + 4 * lh;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = o[d][r4 * 4 + e] * inv;
        if constexpr (sizeof(T) == 2) {
          u32x2 pk; pk[0] = pack_bf2(v[0], v[1]); pk[1] = pack_bf2(v[2], v[3]);
          *(u32x2*)(orow + dim) = pk;
        } else {
          f32x4 pk = {v[0], v[1], v[2], v[3]};
          *(f32x4*)(orow + dim) = pk;
        }
      }
  }
}

// A/B of the workgroup shape (tools build only: the product library reads no environment variable)
static bool force_4_waves() {
#ifdef LDM_TOOLS_BUILD
  static const bool f = getenv("LDM_ATTN_WAVES") && atoi(getenv("LDM_ATTN_WAVES")) == 4;
  return f;
#else
  return false;
#endif
}

template <typename T>
int launch_attn_wide(const AttnArgs& a, int batch, hipStream_t s) {
  constexpr int ES = (int)sizeof(T);
  const int lds = 32 * (512 * ES + 16) + 512 * (32 * ES + 16) + 4 * 64 * 16 * 4;
  // the attribute belongs to the (kernel, DEVICE) pair: remember it per device, atomically (two host
  // threads may launch at once; setting it twice is harmless)
  static std::atomic<uint64_t> attr_set{0};      // bit d = done on device d; per element type (template instance)
  int devid = 0;
  if (hipGetDevice(&devid) != hipSuccess) return -1;
  const uint64_t bit = devid < 64 ? (1ull << devid) : 0;
  if (!(attr_set.load(std::memory_order_acquire) & bit) || !bit) {
    if (hipFuncSetAttribute((const void*)attn_wide_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
      return -1;
    attr_set.fetch_or(bit, std::memory_order_release);
  }
  dim3 grid((a.Tq + 31) / 32, a.heads, batch);
  hipLaunchKernelGGL((attn_wide_kernel<T>), grid, dim3(256), lds, s, a);
  return 0;
}

template <typename T>
int launch_attn(const AttnArgs& a, int Sp, dim3 grid, hipStream_t s) {
  // keys per tile: 128 for bf16 heads <= 64 (half the barriers), 64 above (register budget);
  // float32 tiles are half as many keys for the same LDS bytes
  constexpr bool H = sizeof(T) == 2;
#define ATTN_CASE(SPV, KTB, KTF)                                                                 \
  case SPV:                                                                                      \
    hipLaunchKernelGGL((attn_kernel<T, SPV, H ? KTB : KTF>), grid, dim3(256), 0, s, a);          \
    break;
  if constexpr (H) {
    // 40-wide heads with long query runs: 8-wave workgroups, two staged tiles in flight (NWV = 8, PF = 2)
    if (Sp == 48 && a.Tq >= 256 && !force_4_waves()) {
      dim3 g8((a.Tq + 255) / 256, grid.y, grid.z);
      hipLaunchKernelGGL((attn_kernel<T, 48, 64, false, 8, 2>), g8, dim3(512), 0, s, a);
      return 0;
    }
  }
  switch (Sp) {
    ATTN_CASE(32, 128, 64)
    ATTN_CASE(48, 64, 32)
    ATTN_CASE(64, 64, 32)
    ATTN_CASE(80, 64, 32)
    ATTN_CASE(96, 64, 32)
    ATTN_CASE(160, 64, 32)
    default: return -1;
  }
#undef ATTN_CASE
  return 0;
}

}  // namespace

extern "C" int ldm_attention_ms(const void* q, int64_t ldq, int64_t q_bs, const void* k, int64_t ldk,
                                int64_t k_bs, const void* vt, int64_t ldvt, int64_t vt_bs, void* out,
                                int64_t ldo, int64_t o_bs, int batch, int heads, int Tq, int Tk, int Sp,
                                int dtype, void* stream) {
  LDM_CHECK_ARG(q && k && vt && out, "ldm_attention_ms: null pointer");
  LDM_CHECK_ARG(dtype == LDM_BF16 && Sp == 48, "ldm_attention_ms: bf16 with 40-wide heads padded to 48 only");
  LDM_CHECK_ARG(batch > 0 && heads > 0 && Tq > 0 && Tk > 0 && batch < 65536 && heads < 65536, "ldm_attention_ms: bad dims");
  LDM_CHECK_ARG(ldq % 8 == 0 && ldk % 8 == 0 && ldvt % 8 == 0 && ldo % 8 == 0 && q_bs % 8 == 0 && k_bs % 8 == 0 &&
                    vt_bs % 8 == 0 && o_bs % 8 == 0 && ldvt >= Tk, "ldm_attention_ms: strides must be multiples of 8 elements, ldvt >= Tk");
  LDM_CHECK_ARG(((uintptr_t)q % 16) == 0 && ((uintptr_t)k % 16) == 0 && ((uintptr_t)vt % 16) == 0 &&
                    ((uintptr_t)out % 16) == 0, "ldm_attention_ms: pointers must be 16-byte aligned");
  AttnArgs a;
  a.q = (const char*)q; a.k = (const char*)k; a.vt = (const char*)vt; a.out = (char*)out;
  a.ldq = ldq; a.q_bs = q_bs; a.ldk = ldk; a.k_bs = k_bs; a.ldvt = ldvt; a.vt_bs = vt_bs;
  a.ldo = ldo; a.o_bs = o_bs; a.heads = heads; a.Tq = Tq; a.Tk = Tk; a.scale = 1.0f;
  if (Tq >= 256 && !force_4_waves()) {      // long query runs: 8-wave workgroups
    dim3 g8((Tq + 255) / 256, heads, batch);
    hipLaunchKernelGGL((attn_kernel<bf16_t, 48, 64, true, 8, 2>), g8, dim3(512), 0, (hipStream_t)stream, a);
  } else {
    dim3 grid((Tq + 127) / 128, heads, batch);
    hipLaunchKernelGGL((attn_kernel<bf16_t, 48, 64, true>), grid, dim3(256), 0, (hipStream_t)stream, a);
  }
  return ldm_launch_status("ldm_attention_ms");
}

extern "C" int ldm_attention(const void* q, int64_t ldq, int64_t q_bs, const void* k, int64_t ldk,
                             int64_t k_bs, const void* vt, int64_t ldvt, int64_t vt_bs, void* out,
                             int64_t ldo, int64_t o_bs, int batch, int heads, int Tq, int Tk, int Sp,
                             float scale, int dtype, void* stream) {
  LDM_CHECK_ARG(q && k && vt && out, "ldm_attention: null pointer");
  LDM_CHECK_ARG(dtype == LDM_F32 || dtype == LDM_BF16, "ldm_attention: bad dtype");
  LDM_CHECK_ARG(batch > 0 && heads > 0 && Tq > 0 && Tk > 0, "ldm_attention: bad dims");
  LDM_CHECK_ARG(batch < 65536 && heads < 65536, "ldm_attention: batch/heads too large");
  const int epc = dtype == LDM_BF16 ? 8 : 4;
  LDM_CHECK_ARG(ldq % epc == 0 && ldk % epc == 0 && ldvt % epc == 0 && ldo % epc == 0 &&
                    q_bs % epc == 0 && k_bs % epc == 0 && vt_bs % epc == 0 && o_bs % epc == 0,
                "ldm_attention: strides must be multiples of %d elements", epc);
  LDM_CHECK_ARG(ldvt >= Tk, "ldm_attention: ldvt=%lld < Tk=%d", (long long)ldvt, Tk);
  LDM_CHECK_ARG(((uintptr_t)q % 16) == 0 && ((uintptr_t)k % 16) == 0 && ((uintptr_t)vt % 16) == 0 &&
                    ((uintptr_t)out % 16) == 0, "ldm_attention: pointers must be 16-byte aligned");
  AttnArgs a;
  a.q = (const char*)q; a.k = (const char*)k; a.vt = (const char*)vt; a.out = (char*)out;
  a.ldq = ldq; a.q_bs = q_bs; a.ldk = ldk; a.k_bs = k_bs; a.ldvt = ldvt; a.vt_bs = vt_bs;
  a.ldo = ldo; a.o_bs = o_bs; a.heads = heads; a.Tq = Tq; a.Tk = Tk; a.scale = scale;
  dim3 grid((Tq + 127) / 128, heads, batch);
  hipStream_t s = (hipStream_t)stream;
  int r;
  if (Sp == 512) r = dtype == LDM_BF16 ? launch_attn_wide<bf16_t>(a, batch, s) : launch_attn_wide<float>(a, batch, s);
  else r = dtype == LDM_BF16 ? launch_attn<bf16_t>(a, Sp, grid, s) : launch_attn<float>(a, Sp, grid, s);
  LDM_CHECK_ARG(r == 0, "ldm_attention: unsupported padded head size Sp=%d (32/48/64/80/96/160, or 512 = head dim split over 4 waves)", Sp);
  return ldm_launch_status("ldm_attention");
}
