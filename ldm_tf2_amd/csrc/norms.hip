// GroupNorm (+SiLU), LayerNorm, row softmax -- HBM-bound streaming kernels.
// All loads/stores are 16 bytes per lane; statistics are float32.
#include "common.h"

namespace {

// Column tiling shared by the two GroupNorm kernels: a 256-thread block covers
// VT = min(nvec, 256) 16-byte channel vectors x P = 256/VT pixels at a time.
struct GnGeom {
  int nvec, VT, P;
};
__host__ __device__ inline GnGeom gn_geom(int C, int epc) {
  GnGeom g;
  g.nvec = C / epc;
  g.VT = g.nvec < 256 ? g.nvec : 256;
  g.P = 256 / g.VT;
  return g;
}

// partial[b][chunk][g][2] = (sum x, sum x^2) over the chunk's pixels and group g
template <typename T>
__global__ __launch_bounds__(256) void gn_partial_kernel(const T* __restrict__ x, int64_t ldx,
                                                         float* __restrict__ partial, int HW, int C,
                                                         int G, int nchunks) {
  constexpr int EPC = Elem<T>::kPerChunk;
  extern __shared__ __attribute__((aligned(16))) float sm[];  // [2][P][C]
  const GnGeom gg = gn_geom(C, EPC);
  const int tid = threadIdx.x;
  const int chunk = blockIdx.x, b = blockIdx.y;
  const int ppc = (HW + nchunks - 1) / nchunks;
  const int p_begin = chunk * ppc, p_end = min(HW, p_begin + ppc);
  const int pl = tid / gg.VT, vl = tid - pl * gg.VT;
  const T* xb = x + (int64_t)b * HW * ldx;
  float* s1 = sm;
  float* s2 = sm + gg.P * C;
  for (int v0 = 0; v0 < gg.nvec; v0 += gg.VT) {
    const int v = v0 + vl;
    float a1[EPC], a2[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) { a1[e] = 0.f; a2[e] = 0.f; }
    if (pl < gg.P && v < gg.nvec) {
      // 4 independent 16-byte loads in flight per lane (the kernel is latency-bound otherwise)
      for (int p = p_begin + pl; p < p_end; p += 4 * gg.P) {
        u32x4 c[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int pp = p + u * gg.P;
          c[u] = u32x4{0u, 0u, 0u, 0u};
          if (pp < p_end) c[u] = *(const u32x4*)(xb + (int64_t)pp * ldx + v * EPC);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          float f[EPC];
          chunk_to_f32(c[u], f, T());
#pragma unroll
          for (int e = 0; e < EPC; ++e) { a1[e] += f[e]; a2[e] += f[e] * f[e]; }
        }
      }
#pragma unroll
      for (int e = 0; e < EPC; ++e) {
        s1[pl * C + v * EPC + e] = a1[e];
        s2[pl * C + v * EPC + e] = a2[e];
      }
    }
  }
  __syncthreads();
  // 8 lanes per group split its (pixel-lane, channel) cells, then a 3-step butterfly
  // (fixed order: deterministic).  G <= 32 groups -> one pass of the 256 threads.
  const int cpg = C / G;
  for (int g0 = 0; g0 < G; g0 += 32) {
    const int g = g0 + (tid >> 3), sub = tid & 7;
    float t1 = 0.f, t2 = 0.f;
    if (g < G) {
      const int cells = cpg * gg.P;
      for (int i = sub; i < cells; i += 8) {
        const int q = i / cpg, c = g * cpg + (i - q * cpg);
        t1 += s1[q * C + c]; t2 += s2[q * C + c];
      }
    }
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) { t1 += __shfl_xor(t1, o, 64); t2 += __shfl_xor(t2, o, 64); }
    if (g < G && sub == 0) {
      float* o = partial + (((int64_t)b * nchunks + chunk) * G + g) * 2;
      o[0] = t1; o[1] = t2;
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void gn_apply_kernel(const T* __restrict__ x, int64_t ldx,
                                                       const float* __restrict__ partial,
                                                       const float* __restrict__ gamma,
                                                       const float* __restrict__ beta,
                                                       T* __restrict__ out, int64_t ldo, int HW,
                                                       int C, int G, int nchunks, int achunks,
                                                       float eps, int do_silu) {
  constexpr int EPC = Elem<T>::kPerChunk;
  __shared__ float s_mean[64], s_rstd[64];
  const GnGeom gg = gn_geom(C, EPC);
  const int tid = threadIdx.x;
  const int chunk = blockIdx.x, b = blockIdx.y;
  const int cpg = C / G;
  for (int g0 = 0; g0 < G; g0 += 32) {
    const int g = g0 + (tid >> 3), sub = tid & 7;   // 8 lanes per group over the chunks
    float t1 = 0.f, t2 = 0.f;
    if (g < G)
      for (int c = sub; c < nchunks; c += 8) {
        const float* o = partial + (((int64_t)b * nchunks + c) * G + g) * 2;
        t1 += o[0]; t2 += o[1];
      }
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) { t1 += __shfl_xor(t1, o, 64); t2 += __shfl_xor(t2, o, 64); }
    if (g < G && sub == 0) {
      const float n = (float)HW * (float)cpg;
      const float mean = t1 / n;
      float var = t2 / n - mean * mean;
      var = var < 0.f ? 0.f : var;
      s_mean[g] = mean;
      s_rstd[g] = rsqrtf(var + eps);
    }
  }
  __syncthreads();
  const int ppc = (HW + achunks - 1) / achunks;
  const int p_begin = chunk * ppc, p_end = min(HW, p_begin + ppc);
  const int pl = tid / gg.VT, vl = tid - pl * gg.VT;
  const T* xb = x + (int64_t)b * HW * ldx;
  T* ob = out + (int64_t)b * HW * ldo;
  for (int v0 = 0; v0 < gg.nvec; v0 += gg.VT) {
    const int v = v0 + vl;
    if (!(pl < gg.P && v < gg.nvec)) continue;
    float mu[EPC], sc[EPC], sh[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      const int c = v * EPC + e;
      const int g = c / cpg;
      mu[e] = s_mean[g];
      sc[e] = s_rstd[g] * gamma[c];
      sh[e] = beta[c];
    }
    for (int p = p_begin + pl; p < p_end; p += 4 * gg.P) {
      u32x4 cin[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int pp = p + u * gg.P;
        if (pp < p_end) cin[u] = *(const u32x4*)(xb + (int64_t)pp * ldx + v * EPC);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int pp = p + u * gg.P;
        if (pp < p_end) {
          float f[EPC];
          chunk_to_f32(cin[u], f, T());
#pragma unroll
          for (int e = 0; e < EPC; ++e) {
            float y = (f[e] - mu[e]) * sc[e] + sh[e];
            f[e] = do_silu ? silu_f(y) : y;
          }
          *(u32x4*)(ob + (int64_t)pp * ldo + v * EPC) = f32_to_chunk(f, T());
        }
      }
    }
  }
}

// LPR lanes share a row (64/LPR rows per wave), each lane keeps <= 8 16-byte pieces of
// its row in registers between the passes; reductions are LPR-wide butterflies.  Small
// C therefore packs several rows into a wave instead of idling most of its lanes, and
// every lane has all its loads in flight at once.
template <typename T, int LPR>
__global__ __launch_bounds__(256, 4) void layernorm_kernel(const T* __restrict__ x, int64_t ldx,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta,
                                                        T* __restrict__ out, int64_t ldo, int rows,
                                                        int C, float eps) {
  constexpr int EPC = Elem<T>::kPerChunk;
  constexpr int MAXV = 8;
  constexpr int RPW = 64 / LPR;
  const int lane = threadIdx.x & 63;
  const int sub = lane % LPR;
  const int row = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + lane / LPR;
  const bool rok = row < rows;
  const int nvec = C / EPC;
  const T* xr = x + (int64_t)(rok ? row : 0) * ldx;
  u32x4 buf[MAXV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int v = sub + i * LPR;
    if (v < nvec) buf[i] = *(const u32x4*)(xr + v * EPC);
  }
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int v = sub + i * LPR;
    if (v < nvec) {
      float f[EPC];
      chunk_to_f32(buf[i], f, T());
#pragma unroll
      for (int e = 0; e < EPC; ++e) s += f[e];
    }
  }
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  const float mean = s / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int v = sub + i * LPR;
    if (v < nvec) {
      float f[EPC];
      chunk_to_f32(buf[i], f, T());
#pragma unroll
      for (int e = 0; e < EPC; ++e) { const float d = f[e] - mean; q += d * d; }
    }
  }
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
  const float rstd = rsqrtf(q / (float)C + eps);
  if (!rok) return;
  T* orow = out + (int64_t)row * ldo;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int v = sub + i * LPR;
    if (v < nvec) {
      float f[EPC];
      chunk_to_f32(buf[i], f, T());
      const f32x4* gp = (const f32x4*)(gamma + v * EPC);
      const f32x4* bp = (const f32x4*)(beta + v * EPC);
#pragma unroll
      for (int h = 0; h < EPC / 4; ++h) {
        const f32x4 g4 = gp[h], b4 = bp[h];
#pragma unroll
        for (int e = 0; e < 4; ++e) f[h * 4 + e] = (f[h * 4 + e] - mean) * rstd * g4[e] + b4[e];
      }
      *(u32x4*)(orow + v * EPC) = f32_to_chunk(f, T());
    }
  }
}

// one block per row; out may alias x when TI == TO
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const TI* x, int64_t ldx, TO* out,
                                                           int64_t ldo, int cols, float scale) {
  __shared__ float red[4];
  const TI* r = x + (int64_t)blockIdx.x * ldx;
  TO* o = out + (int64_t)blockIdx.x * ldo;
  const int tid = threadIdx.x;
  float m = -INFINITY;
  for (int c = tid; c < cols; c += 256) m = fmaxf(m, Elem<TI>::ld(r + c) * scale);
  m = wave_max(m);
  if ((tid & 63) == 0) red[tid >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float s = 0.f;
  for (int c = tid; c < cols; c += 256) s += __expf(Elem<TI>::ld(r + c) * scale - m);
  s = wave_sum(s);
  if ((tid & 63) == 0) red[tid >> 6] = s;
  __syncthreads();
  s = red[0] + red[1] + red[2] + red[3];
  const float inv = 1.0f / s;
  // a thread only ever rewrites the columns it alone reads, so aliasing is safe
  // when sizeof(TI) == sizeof(TO); f32 -> bf16 must not alias.
  for (int c = tid; c < cols; c += 256)
    Elem<TO>::st(o + c, __expf(Elem<TI>::ld(r + c) * scale - m) * inv);
}


}  // namespace

#include "gn_fused.h"
using namespace ldm_gn;

extern "C" int ldm_groupnorm_fused_supported(int B, int HW, int C, int groups, int dtype) {
  if (!(dtype == LDM_F32 || dtype == LDM_BF16) || B <= 0 || HW <= 0 || C <= 0) return 0;
  GnFusedPlan pl;
  return gn_fused_plan(B, HW, C, groups, dtype == LDM_BF16 ? 2 : 4, &pl) ? 1 : 0;
}

extern "C" int ldm_groupnorm_fused(const void* x, int64_t ldx, const float* gamma, const float* beta,
                                   void* out, int64_t ldo, int B, int HW, int C, int groups, float eps,
                                   int silu, int dtype, void* stream) {
  LDM_CHECK_ARG(x && gamma && beta && out, "ldm_groupnorm_fused: null pointer");
  LDM_CHECK_ARG(dtype == LDM_F32 || dtype == LDM_BF16, "ldm_groupnorm_fused: bad dtype");
  const int epc = dtype == LDM_BF16 ? 8 : 4;
  LDM_CHECK_ARG(B > 0 && HW > 0 && C > 0 && groups > 0 && C % groups == 0, "ldm_groupnorm_fused: bad dims");
  LDM_CHECK_ARG(ldx % epc == 0 && ldo % epc == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)out % 16) == 0,
                "ldm_groupnorm_fused: ldx/ldo/x/out must be 16-byte aligned");
  GnFusedPlan pl;
  LDM_CHECK_ARG(gn_fused_plan(B, HW, C, groups, 16 / epc, &pl),
                "ldm_groupnorm_fused: shape B=%d HW=%d C=%d groups=%d not supported (use partial+apply)", B,
                HW, C, groups);
  dim3 grid(8 * (groups / pl.GB) * ((B + 7) / 8));
  hipStream_t s = (hipStream_t)stream;
  if (dtype == LDM_BF16)
    gn_fused_launch<bf16_t>(pl, grid, s, x, ldx, gamma, beta, out, ldo, B, HW, C, groups, eps, silu);
  else
    gn_fused_launch<float>(pl, grid, s, x, ldx, gamma, beta, out, ldo, B, HW, C, groups, eps, silu);
  return ldm_launch_status("ldm_groupnorm_fused");
}

extern "C" int ldm_groupnorm_nchunks(int B, int HW, int C) {
  (void)C;
  int n = (1024 + B - 1) / B;
  int cap = HW / 32;
  if (cap < 1) cap = 1;
  if (n > cap) n = cap;
  if (n > 128) n = 128;
  if (n < 1) n = 1;
  return n;
}

static int gn_check(const char* who, const void* x, int64_t ldx, int B, int HW, int C, int groups,
                    int nchunks, int dtype) {
  LDM_CHECK_ARG(x, "%s: null x", who);
  LDM_CHECK_ARG(dtype == LDM_F32 || dtype == LDM_BF16, "%s: bad dtype", who);
  const int epc = dtype == LDM_BF16 ? 8 : 4;
  LDM_CHECK_ARG(B > 0 && HW > 0 && C > 0 && groups > 0 && groups <= 64 && nchunks > 0,
                "%s: bad dims", who);
  LDM_CHECK_ARG(C % groups == 0 && C % epc == 0 && ldx % epc == 0 && ((uintptr_t)x % 16) == 0,
                "%s: C=%d must divide by groups=%d and %d; ldx, x 16-byte aligned", who, C, groups, epc);
  return LDM_OK;
}

extern "C" int ldm_groupnorm_partial(const void* x, int64_t ldx, float* partial, int B, int HW,
                                     int C, int groups, int nchunks, int dtype, void* stream) {
  int st = gn_check("ldm_groupnorm_partial", x, ldx, B, HW, C, groups, nchunks, dtype);
  if (st) return st;
  LDM_CHECK_ARG(partial, "ldm_groupnorm_partial: null partial");
  const int epc = dtype == LDM_BF16 ? 8 : 4;
  const GnGeom gg = gn_geom(C, epc);
  const size_t shm = (size_t)2 * gg.P * C * sizeof(float);
  LDM_CHECK_ARG(shm <= 64 * 1024, "ldm_groupnorm_partial: C=%d too large", C);
  dim3 grid(nchunks, B);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == LDM_BF16)
    hipLaunchKernelGGL(gn_partial_kernel<bf16_t>, grid, dim3(256), shm, s, (const bf16_t*)x, ldx,
                       partial, HW, C, groups, nchunks);
  else
    hipLaunchKernelGGL(gn_partial_kernel<float>, grid, dim3(256), shm, s, (const float*)x, ldx,
                       partial, HW, C, groups, nchunks);
  return ldm_launch_status("ldm_groupnorm_partial");
}

extern "C" int ldm_groupnorm_apply(const void* x, int64_t ldx, const float* partial,
                                   const float* gamma, const float* beta, void* out, int64_t ldo,
                                   int B, int HW, int C, int groups, int nchunks, float eps, int silu,
                                   int dtype, void* stream) {
  int st = gn_check("ldm_groupnorm_apply", x, ldx, B, HW, C, groups, nchunks, dtype);
  if (st) return st;
  const int epc = dtype == LDM_BF16 ? 8 : 4;
  LDM_CHECK_ARG(partial && gamma && beta && out, "ldm_groupnorm_apply: null pointer");
  LDM_CHECK_ARG(ldo % epc == 0 && ((uintptr_t)out % 16) == 0, "ldm_groupnorm_apply: out alignment");
  // fatter blocks than the statistics pass: ~512 workgroups, >= 32 pixels each
  int achunks = (512 + B - 1) / B;
  if (achunks > HW / 32) achunks = HW / 32;
  if (achunks < 1) achunks = 1;
  dim3 grid(achunks, B);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == LDM_BF16)
    hipLaunchKernelGGL(gn_apply_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)x, ldx, partial,
                       gamma, beta, (bf16_t*)out, ldo, HW, C, groups, nchunks, achunks, eps, silu);
  else
    hipLaunchKernelGGL(gn_apply_kernel<float>, grid, dim3(256), 0, s, (const float*)x, ldx, partial,
                       gamma, beta, (float*)out, ldo, HW, C, groups, nchunks, achunks, eps, silu);
  return ldm_launch_status("ldm_groupnorm_apply");
}

extern "C" int ldm_layernorm(const void* x, int64_t ldx, const float* gamma, const float* beta,
                             void* out, int64_t ldo, int rows, int C, float eps, int dtype,
                             void* stream) {
  LDM_CHECK_ARG(x && gamma && beta && out, "ldm_layernorm: null pointer");
  LDM_CHECK_ARG(dtype == LDM_F32 || dtype == LDM_BF16, "ldm_layernorm: bad dtype");
  const int epc = dtype == LDM_BF16 ? 8 : 4;
  LDM_CHECK_ARG(rows > 0 && C > 0 && C % epc == 0 && C / epc <= 8 * 64,
                "ldm_layernorm: C=%d must be a multiple of %d and <= %d", C, epc, 512 * epc);
  LDM_CHECK_ARG(ldx % epc == 0 && ldo % epc == 0 && ((uintptr_t)x % 16) == 0 &&
                    ((uintptr_t)out % 16) == 0, "ldm_layernorm: alignment");
  LDM_CHECK_ARG(((uintptr_t)gamma % 16) == 0 && ((uintptr_t)beta % 16) == 0, "ldm_layernorm: gamma/beta alignment");
  const int nvec = C / epc;
  int lpr = 4;
  while (lpr < 64 && lpr * 8 < nvec) lpr *= 2;
  const int rows_per_block = 4 * (64 / lpr);
  dim3 grid((rows + rows_per_block - 1) / rows_per_block);
  hipStream_t s = (hipStream_t)stream;
#define LN_LAUNCH(TT, L)                                                                         \
  hipLaunchKernelGGL((layernorm_kernel<TT, L>), grid, dim3(256), 0, s, (const TT*)x, ldx, gamma, \
                     beta, (TT*)out, ldo, rows, C, eps)
#define LN_SWITCH(TT)                     \
  switch (lpr) {                          \
    case 4: LN_LAUNCH(TT, 4); break;      \
    case 8: LN_LAUNCH(TT, 8); break;      \
    case 16: LN_LAUNCH(TT, 16); break;    \
    case 32: LN_LAUNCH(TT, 32); break;    \
    default: LN_LAUNCH(TT, 64); break;    \
  }
  if (dtype == LDM_BF16) { LN_SWITCH(bf16_t) } else { LN_SWITCH(float) }
#undef LN_SWITCH
#undef LN_LAUNCH
  return ldm_launch_status("ldm_layernorm");
}

extern "C" int ldm_softmax_rows(const void* x, int64_t ldx, int in_dtype, void* out, int64_t ldo,
                                int out_dtype, int rows, int cols, float scale, void* stream) {
  LDM_CHECK_ARG(x && out && rows > 0 && cols > 0, "ldm_softmax_rows: bad args");
  LDM_CHECK_ARG((in_dtype == LDM_F32 || in_dtype == LDM_BF16) &&
                    (out_dtype == LDM_F32 || out_dtype == LDM_BF16), "ldm_softmax_rows: bad dtype");
  LDM_CHECK_ARG(!(x == out && in_dtype != out_dtype), "ldm_softmax_rows: aliasing needs equal dtypes");
  hipStream_t s = (hipStream_t)stream;
  dim3 g(rows), b(256);
  if (in_dtype == LDM_F32 && out_dtype == LDM_F32)
    hipLaunchKernelGGL((softmax_rows_kernel<float, float>), g, b, 0, s, (const float*)x, ldx, (float*)out, ldo, cols, scale);
  else if (in_dtype == LDM_F32)
    hipLaunchKernelGGL((softmax_rows_kernel<float, bf16_t>), g, b, 0, s, (const float*)x, ldx, (bf16_t*)out, ldo, cols, scale);
  else if (out_dtype == LDM_F32)
    hipLaunchKernelGGL((softmax_rows_kernel<bf16_t, float>), g, b, 0, s, (const bf16_t*)x, ldx, (float*)out, ldo, cols, scale);
  else
    hipLaunchKernelGGL((softmax_rows_kernel<bf16_t, bf16_t>), g, b, 0, s, (const bf16_t*)x, ldx, (bf16_t*)out, ldo, cols, scale);
  return ldm_launch_status("ldm_softmax_rows");
}
