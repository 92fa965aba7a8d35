// Single-launch GroupNorm (+SiLU) kernel template, shared by norms.hip (plain form) and gn_splitk.hip
// (split-K source).  The two sets of instantiations live in SEPARATE translation units on purpose:
// co-compiled with the split-K variants, hipcc gave the unrelated layernorm_kernel of norms.hip 128
// VGPRs + 668 bytes of scratch per lane instead of 62 VGPRs (12x slower: 6 -> 75 us per launch).
#pragma once
#include "common.h"

namespace ldm_gn {

// ---- single-launch GroupNorm (+SiLU) for the U-Net's image sizes ---------------------------
// One workgroup owns GB whole groups of one sample: its [HW][GB*cpg] slab (S 16-byte chunks
// per pixel) is read from HBM ONCE, kept in registers between the passes (mean; centred
// sum of squares; normalise) and written once.  No partial buffer, no second launch, and the
// variance is the two-pass form.  Thread t owns chunk column cs = t % S of pixels t/S,
// t/S + P, ...; a 16-byte chunk spans at most two groups (cpg >= elements per chunk), so a
// thread carries two running sums.  Reductions are fixed-order (no float atomics): results
// do not depend on scheduling.  Blocks of one sample share i % 8, i.e. one XCD's L2, which
// also holds the other halves of the partially used cache lines.
struct GnFusedPlan { int GB, S, NT, maxch; };

// keeps the compiler from hoisting the three passes' conversions of the register-resident
// slab into one (which would hold every element as a float at once)
__device__ __forceinline__ u32x4 opaque(u32x4 c) {
  asm volatile("" : "+v"(c));
  return c;
}

inline bool gn_fused_plan(int B, int HW, int C, int G, int esize, GnFusedPlan* pl) {
  if (G <= 0 || C % G) return false;
  const int epc = 16 / esize, cpg = C / G;
  if (cpg < epc) return false;
  int GB = 0;
  for (int g = 1; g <= 8; g *= 2)
    if (G % g == 0 && (g * cpg) % epc == 0) { GB = g; break; }
  if (!GB) return false;
  const int S = GB * cpg / epc;
  if (S > 64) return false;
  // small slabs: ONE wave per workgroup (block barriers degenerate, reductions are pure
  // butterflies), up to 24 chunks per lane
  // (measured: pays when the launch still has >= 2 waves per CU at <= 16 chunks per lane, or
  // >= 4 per CU at <= 24)
  {
    const int need = (HW + 64 / S - 1) / (64 / S);
    const int64_t waves = (int64_t)B * (G / GB);
    if (need <= 8 && waves >= 512) { *pl = {GB, S, 64, 8}; return true; }
    if (need <= 16 && waves >= 512) { *pl = {GB, S, 64, 16}; return true; }
    if (need <= 24 && waves >= 1024) { *pl = {GB, S, 64, 24}; return true; }
  }
  // few chunks per thread (register budget: 4 VGPRs each), more threads before more chunks
  for (int NT : {256, 512, 1024}) {
    const int need = (HW + NT / S - 1) / (NT / S);
    if (need <= 8) { *pl = {GB, S, NT, 8}; return true; }
    if (need <= 16) { *pl = {GB, S, NT, 16}; return true; }
  }
  for (int NT : {256, 512}) {
    const int need = (HW + NT / S - 1) / (NT / S);
    if (need <= 24) { *pl = {GB, S, NT, 24}; return true; }
  }
  return false;
}

// Split-K source of the slab (SK = true, ldm_groupnorm_splitk): instead of reading x the workgroup
// sums the float32 partial slabs of the product that PRODUCES x, applies that product's epilogue in the
// order of splitk_epilogue_vec_kernel (slabs in split order, * alpha + bias, + addend, + residual, round
// to T: the value is bit-identical to the plain reduce's) and optionally stores x on the way.
struct GnSplitK {
  const float* ws;        // [split][M][N] float32
  int64_t slab;           // M * N
  int split, N;
  const float* bias;      // [N] or null
  const float* addend;    // row (b) * add_ld + n, or null
  int64_t add_ld;
  const void* residual;   // T, row stride ldr, or null
  int64_t ldr;
  void* xout;             // T, row stride ldx (the kernel's ldx argument), or null: x is not stored
  float alpha;
};

// (SK: the slab sums of all of a thread's chunks are kept in float registers at once, MAXCH * EPC of
// them, so that every slab load of a split is in flight together: 2 waves per SIMD of register budget)
template <typename T, int NT, int MAXCH, bool SK = false>
__global__ __launch_bounds__(NT, (SK ? (NT >= 1024 ? 4 : 2) : (MAXCH <= 8 || (MAXCH <= 16 && NT > 64)) ? 4 : 2)) void gn_fused_kernel(const T* __restrict__ x, int64_t ldx,
                                                      const float* __restrict__ gamma,
                                                      const float* __restrict__ beta,
                                                      T* __restrict__ out, int64_t ldo, int B, int HW,
                                                      int C, int G, int GB, int S, float eps,
                                                      int do_silu, GnSplitK sk) {
  constexpr int EPC = Elem<T>::kPerChunk;
  constexpr int NW = NT / 64;
  __shared__ float s_part[NT == 64 ? 1 : NT][2];
  __shared__ float s_tot[64][2];
  __shared__ float s_red[8], s_mean[8], s_rstd[8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nblk = G / GB;
  const int bi = blockIdx.x;
  const int b = (bi & 7) + 8 * ((bi >> 3) / nblk), blk = (bi >> 3) % nblk;
  if (b >= B) return;                               // uniform per block
  const int cpg = C / G;
  const int P = NT / S;
  const int pl = tid / S, cs = tid - pl * S;
  const bool active = pl < P;
  const int e0 = cs * EPC;                          // first element of my chunk in the segment
  const int g_lo = e0 / cpg;
  const int nlo = min(EPC, (g_lo + 1) * cpg - e0);  // my chunk's elements that belong to g_lo
  const int c0 = blk * GB * cpg + e0;               // first channel of my chunk
  const T* xb = x + (int64_t)b * HW * ldx + c0;

  u32x4 v[MAXCH];
  if constexpr (SK) {
    // this thread's chunk column is fixed: bias / addend of its EPC channels once
    float bs[EPC], ad[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      bs[e] = (active && sk.bias) ? sk.bias[c0 + e] : 0.f;
      ad[e] = (active && sk.addend) ? sk.addend[(int64_t)b * sk.add_ld + c0 + e] : 0.f;
    }
    // Slab sums in batches of KB chunks; inside a batch: split OUTER, chunk INNER, and the loads of one
    // split are UNCONDITIONAL (an inactive slot re-reads the thread's first chunk, or element 0 of the
    // workspace) and land in temporaries before any of them is consumed.  A load inside `if (valid)` next
    // to its use cannot be hoisted over the other chunks' branches: the chunks then pay one memory round
    // trip EACH (measured: the fused launch cost more than reduce + GroupNorm separately).  Summation
    // order per element is unchanged: slabs in split order, as the plain reduce.
    const int64_t row0 = (int64_t)b * HW + pl;
    const float* src0 = sk.ws + row0 * sk.N + c0;
    const int64_t kstep = (int64_t)P * sk.N;                  // floats between a thread's consecutive chunks
    // KB chunks x SU splits = 8 chunk-loads in flight per round trip, as before, but no longer 8 x 1: a one-wave
    // workgroup (the 8x8 / 4x4 levels: 2 - 6 chunks per thread, 4 - 12 splits) paid one memory round trip PER SPLIT
    // -- 16 us for a 1.3 MB tensor -- and takes 2 x 4 (13.5 us); the larger slabs (2 splits, 6 - 11 chunks) take 4 x 2.  A batch
    // of chunks none of whose pixels exist is skipped (workgroup-uniform); a split past the last one loads slab 0
    // and adds nothing.  Summation order per element: slabs in split order, as ever.
    constexpr int SU = NT == 64 ? 4 : 2;
    constexpr int KBW = 8 / SU;                      // (4 x 4 on the one-wave form: 250 VGPRs, one wave per SIMD less: 15.0 vs 13.5 us)
    constexpr int KB = MAXCH < KBW ? MAXCH : KBW;
#pragma unroll
    for (int k0 = 0; k0 < MAXCH; k0 += KB) {
      if (k0 > 0 && k0 * P >= HW) {                  // pl + k P >= HW for every thread and every k >= k0:
#pragma unroll
        for (int k = 0; k < KB; ++k) { u32x4 z = {0u, 0u, 0u, 0u}; v[k0 + k] = z; }   // nothing to sum (uniform)
        continue;
      }
      int64_t koff[KB];
      bool kval[KB];
      float f[KB][EPC];
      u32x4 rres[KB];
#pragma unroll
      for (int k = 0; k < KB; ++k) {
        kval[k] = active && (pl + (k0 + k) * P < HW);
        koff[k] = kval[k] ? (int64_t)(k0 + k) * kstep : ((active && pl < HW) ? 0 : -(row0 * sk.N + c0));   // (pl >= HW: no chunk of mine is in range)
#pragma unroll
        for (int e = 0; e < EPC; ++e) f[k][e] = 0.f;
        u32x4 z = {0u, 0u, 0u, 0u};
        rres[k] = z;
        if (sk.residual && kval[k])
          rres[k] = *(const u32x4*)((const T*)sk.residual + (row0 + (int64_t)(k0 + k) * P) * sk.ldr + c0);
      }
      for (int sp = 0; sp < sk.split; sp += SU) {    // SU slabs' loads in flight, added in split order
        f32x4 t[SU][KB][EPC / 4];
#pragma unroll
        for (int u = 0; u < SU; ++u) {
          const float* src = src0 + (sp + u < sk.split ? sp + u : 0) * sk.slab;
#pragma unroll
          for (int k = 0; k < KB; ++k)
#pragma unroll
            for (int q = 0; q < EPC / 4; ++q) t[u][k][q] = *(const f32x4*)(src + koff[k] + 4 * q);
        }
#pragma unroll
        for (int u = 0; u < SU; ++u) {
          const bool live = sp + u < sk.split;        // (uniform)
#pragma unroll
          for (int k = 0; k < KB; ++k)
#pragma unroll
            for (int q = 0; q < EPC / 4; ++q)
#pragma unroll
              for (int e = 0; e < 4; ++e) f[k][4 * q + e] += (kval[k] && live) ? t[u][k][q][e] : 0.f;
        }
      }
#pragma unroll
      for (int k = 0; k < KB; ++k) {
        u32x4 z = {0u, 0u, 0u, 0u};
        v[k0 + k] = z;
        if (kval[k]) {
          const int64_t m = row0 + (int64_t)(k0 + k) * P;
          if (sk.bias) {
#pragma unroll
            for (int e = 0; e < EPC; ++e) f[k][e] = f[k][e] * sk.alpha + bs[e];
          } else {
#pragma unroll
            for (int e = 0; e < EPC; ++e) f[k][e] *= sk.alpha;
          }
          if (sk.addend) {
#pragma unroll
            for (int e = 0; e < EPC; ++e) f[k][e] += ad[e];
          }
          if (sk.residual) {
            float rr[EPC];
            chunk_to_f32(rres[k], rr, T());
#pragma unroll
            for (int e = 0; e < EPC; ++e) f[k][e] += rr[e];
          }
          v[k0 + k] = f32_to_chunk(f[k], T());
          if (sk.xout) *(u32x4*)((T*)sk.xout + m * ldx + c0) = v[k0 + k];
        }
      }
    }
  } else {
#pragma unroll
    for (int k = 0; k < MAXCH; ++k) {
      const int p = pl + k * P;
      u32x4 z = {0u, 0u, 0u, 0u};
      v[k] = z;
      if (active && p < HW) v[k] = *(const u32x4*)(xb + (int64_t)p * ldx);
    }
  }
  float gm[EPC], bt[EPC];                           // fetched now: off the post-reduction critical path
#pragma unroll
  for (int e = 0; e < EPC; ++e) {
    gm[e] = active ? gamma[c0 + e] : 0.f;
    bt[e] = active ? beta[c0 + e] : 0.f;
  }

  // fixed-order block reduction of a (lo, hi) pair per thread -> s_red[g], g < GB
  // (a lane's lo belongs to group g_lo, its hi to g_lo + 1): per group one wave butterfly over
  // the lanes' selected contributions, then the waves' partials are added in wave order
  auto reduce_groups = [&](float lo, float hi) {
    if constexpr (NT == 64) {
      for (int g = 0; g < GB; ++g) {
        const float mine = (g == g_lo ? lo : 0.f) + (g == g_lo + 1 ? hi : 0.f);
        const float t = wave_sum(mine);
        if (lane == 0) s_red[g] = t;
      }
      __syncthreads();
    } else {
      // multi-wave workgroups: per chunk column (fixed lo/hi split) over the pixel lanes, then
      // the columns of each group
      s_part[tid][0] = lo; s_part[tid][1] = hi;
      __syncthreads();
      for (int c = wave; c < S; c += NW) {
        float a0 = 0.f, a1 = 0.f;
        for (int q = lane; q < P; q += 64) { a0 += s_part[q * S + c][0]; a1 += s_part[q * S + c][1]; }
        a0 = wave_sum(a0); a1 = wave_sum(a1);
        if (lane == 0) { s_tot[c][0] = a0; s_tot[c][1] = a1; }
      }
      __syncthreads();
      if (tid < GB) {
        float t = 0.f;
        for (int c = 0; c < S; ++c) {
          const int ec = c * EPC, gl = ec / cpg;
          const int nl = min(EPC, (gl + 1) * cpg - ec);
          if (gl == tid) t += s_tot[c][0];
          if (nl < EPC && gl + 1 == tid) t += s_tot[c][1];
        }
        s_red[tid] = t;
      }
      __syncthreads();
    }
  };

  const float n = (float)HW * (float)cpg;
  {
    float lo = 0.f, hi = 0.f;                       // padding chunks are zeros: no predicate needed
#pragma unroll
    for (int k = 0; k < MAXCH; ++k) {
      float f[EPC];
      chunk_to_f32(opaque(v[k]), f, T());
#pragma unroll
      for (int e = 0; e < EPC; ++e) { if (e < nlo) lo += f[e]; else hi += f[e]; }
    }
    reduce_groups(lo, hi);
    if (tid < GB) s_mean[tid] = s_red[tid] / n;
    __syncthreads();
  }
  const float m_lo = s_mean[g_lo], m_hi = s_mean[min(g_lo + 1, GB - 1)];
  {
    float lo = 0.f, hi = 0.f;
#pragma unroll
    for (int k = 0; k < MAXCH; ++k) {
      if (!(active && pl + k * P < HW)) continue;
      float f[EPC];
      chunk_to_f32(opaque(v[k]), f, T());
#pragma unroll
      for (int e = 0; e < EPC; ++e) {
        const float d = f[e] - (e < nlo ? m_lo : m_hi);
        if (e < nlo) lo += d * d; else hi += d * d;
      }
    }
    reduce_groups(lo, hi);
    if (tid < GB) s_rstd[tid] = rsqrtf(s_red[tid] / n + eps);
    __syncthreads();
  }
  if (!active) return;
  float mu[EPC], sc[EPC], sh[EPC];
  {
    const float r_lo = s_rstd[g_lo], r_hi = s_rstd[min(g_lo + 1, GB - 1)];
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      mu[e] = e < nlo ? m_lo : m_hi;
      sc[e] = (e < nlo ? r_lo : r_hi) * gm[e];
      sh[e] = bt[e];
    }
  }
  T* ob = out + (int64_t)b * HW * ldo + c0;
#pragma unroll
  for (int k = 0; k < MAXCH; ++k) {
    const int p = pl + k * P;
    if (p >= HW) continue;
    float f[EPC];
    chunk_to_f32(opaque(v[k]), f, T());
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      const float y = (f[e] - mu[e]) * sc[e] + sh[e];
      f[e] = do_silu ? silu_f(y) : y;
    }
    *(u32x4*)(ob + (int64_t)p * ldo) = f32_to_chunk(f, T());
  }
}

template <typename T, bool SK = false>
void gn_fused_launch(const GnFusedPlan& pl, dim3 grid, hipStream_t s, const void* x, int64_t ldx,
                     const float* gamma, const float* beta, void* out, int64_t ldo, int B, int HW, int C,
                     int G, float eps, int silu, const GnSplitK& sk = GnSplitK()) {
#define GNF(NT_, MC_)                                                                                  \
  hipLaunchKernelGGL((gn_fused_kernel<T, NT_, MC_, SK>), grid, dim3(NT_), 0, s, (const T*)x, ldx, gamma,   \
                     beta, (T*)out, ldo, B, HW, C, G, pl.GB, pl.S, eps, silu, sk)
  if (pl.NT == 64) { if (pl.maxch == 8) GNF(64, 8); else if (pl.maxch == 16) GNF(64, 16); else GNF(64, 24); }
  else if (pl.NT == 256) { if (pl.maxch == 8) GNF(256, 8); else if (pl.maxch == 16) GNF(256, 16); else GNF(256, 24); }
  else if (pl.NT == 512) { if (pl.maxch == 8) GNF(512, 8); else if (pl.maxch == 16) GNF(512, 16); else GNF(512, 24); }
  else { if (pl.maxch == 8) GNF(1024, 8); else GNF(1024, 16); }
#undef GNF
}


}  // namespace ldm_gn
