// Small / HBM-bound kernels of the sampling path: tiny-channel 3x3 convs, timestep
// embedding + skinny Dense, CFG + DDIM update, first-stage prologue, VQ lookup,
// token embedding, per-image min-max -> uint8, casts.
#include "common.h"

namespace {

template <typename T> __device__ __forceinline__ float ldf(const T* p);
template <> __device__ __forceinline__ float ldf<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ldf<bf16_t>(const bf16_t* p) { return bf2f(*p); }
template <typename T> __device__ __forceinline__ void stf(T* p, float v);
template <> __device__ __forceinline__ void stf<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void stf<bf16_t>(bf16_t* p, float v) { *p = f2bf(v); }

// ---- conv_in, LDS-resident weights: Cin = 4, Cout % 8 == 0, 9*Cin*Cout floats <= 64 KB --------
// Persistent workgroups keep the whole HWIO kernel in LDS; a work item is (4 consecutive pixels
// of a row, 8 consecutive couts): per input row 6 pixel loads (a 4-channel float pixel is one
// 16-byte load) serve the 3 x 4 taps, every weight pair (2 ds_read_b128) is used for 4 pixels,
// one 16-byte store per pixel.  Same accumulation order per output as the kernel below.
template <typename TI, typename TO, int CIN>
__global__ __launch_bounds__(256, 2) void conv_in_lds_kernel(const TI* __restrict__ x, int64_t ldx,
                                                          const float* __restrict__ w,
                                                          const float* __restrict__ bias,
                                                          TO* __restrict__ out, int64_t ldo, int B, int H,
                                                          int W, int Cout) {
  extern __shared__ __attribute__((aligned(16))) float sw[];      // [9*CIN][Cout]
  const int nw = 9 * CIN * Cout;
  for (int i = threadIdx.x * 4; i < nw; i += 256 * 4) *(f32x4*)(sw + i) = *(const f32x4*)(w + i);
  __syncthreads();
  const int c8n = Cout >> 3, wg4 = (W + 3) >> 2;
  const int64_t total = (int64_t)B * H * wg4 * c8n;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(idx % c8n) * 8;
    const int64_t g = idx / c8n;
    const int ox0 = (int)(g % wg4) * 4, oy = (int)((g / wg4) % H), b = (int)(g / ((int64_t)wg4 * H));
    float acc[4][8];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[j][e] = bias ? bias[c8 + e] : 0.f;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int iy = oy + kh - 1;
      if ((unsigned)iy >= (unsigned)H) continue;
      float xr[6][CIN];
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        const int ix = ox0 + q - 1;
        const bool ok = (unsigned)ix < (unsigned)W;
        const TI* xp = x + ((int64_t)(b * H + iy) * W + (ok ? ix : 0)) * ldx;
        if constexpr (CIN == 4 && sizeof(TI) == 4) {
          const f32x4 v = *(const f32x4*)xp;
#pragma unroll
          for (int ci = 0; ci < 4; ++ci) xr[q][ci] = ok ? v[ci] : 0.f;
        } else {
#pragma unroll
          for (int ci = 0; ci < CIN; ++ci) xr[q][ci] = ok ? ldf<TI>(xp + ci) : 0.f;
        }
      }
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const float* wp = sw + (kh * 3 + kw) * CIN * Cout + c8;
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) {
          const f32x4 w0 = *(const f32x4*)(wp + ci * Cout), w1 = *(const f32x4*)(wp + ci * Cout + 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float xv = xr[j + kw][ci];
#pragma unroll
            for (int e = 0; e < 4; ++e) { acc[j][e] += xv * w0[e]; acc[j][4 + e] += xv * w1[e]; }
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (ox0 + j >= W) break;
      TO* op = out + ((int64_t)(b * H + oy) * W + ox0 + j) * ldo + c8;
      if constexpr (sizeof(TO) == 2) {
        *(u32x4*)op = f32_to_chunk(acc[j], bf16_t());
      } else {
        f32x4 o0 = {acc[j][0], acc[j][1], acc[j][2], acc[j][3]}, o1 = {acc[j][4], acc[j][5], acc[j][6], acc[j][7]};
        *(f32x4*)op = o0;
        *(f32x4*)(op + 4) = o1;
      }
    }
  }
}

// ---- conv_in: Cin <= 8, any Cout.  thread = (pixel, 4 consecutive couts) -------
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void conv_small_in_kernel(const TI* __restrict__ x, int64_t ldx,
                                                            const float* __restrict__ w,
                                                            const float* __restrict__ bias,
                                                            TO* __restrict__ out, int64_t ldo, int B,
                                                            int H, int W, int Cin, int Cout) {
  const int cq = Cout >> 2;
  const int64_t total = (int64_t)B * H * W * cq;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * 256) {
    const int c4 = (int)(idx % cq) * 4;
    const int64_t pix = idx / cq;
    const int ox = (int)(pix % W), oy = (int)((pix / W) % H), b = (int)(pix / ((int64_t)W * H));
    float acc[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] = bias ? bias[c4 + e] : 0.f;
    for (int kh = 0; kh < 3; ++kh) {
      const int iy = oy + kh - 1;
      if ((unsigned)iy >= (unsigned)H) continue;
      for (int kw = 0; kw < 3; ++kw) {
        const int ix = ox + kw - 1;
        if ((unsigned)ix >= (unsigned)W) continue;
        const TI* xp = x + ((int64_t)(b * H + iy) * W + ix) * ldx;
        const float* wp = w + (int64_t)((kh * 3 + kw) * Cin) * Cout + c4;
        for (int ci = 0; ci < Cin; ++ci) {
          const float xv = ldf<TI>(xp + ci);
          const f32x4 wv = *(const f32x4*)(wp + (int64_t)ci * Cout);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[e] += xv * wv[e];
        }
      }
    }
    TO* op = out + pix * ldo + c4;
#pragma unroll
    for (int e = 0; e < 4; ++e) stf<TO>(op + e, acc[e]);
  }
}

// ---- conv_out: Cout <= 4, Cin multiple of 16-byte chunks.  One wave = 8 pixels,
// 8 lanes split the 9*Cin reduction of one pixel; weights [9*Cin][Cout] f32 in LDS.
template <typename TI, typename TO, int COUT>
__global__ __launch_bounds__(256) void conv_small_out_kernel(const TI* __restrict__ x, int64_t ldx,
                                                             const float* __restrict__ w,
                                                             const float* __restrict__ bias,
                                                             TO* __restrict__ out, int64_t ldo,
                                                             int B, int H, int W, int Cin) {
  constexpr int EPC = Elem<TI>::kPerChunk;
  extern __shared__ __attribute__((aligned(16))) float sw[];  // [9*Cin][COUT]
  const int nw = 9 * Cin * COUT;
  for (int i = threadIdx.x; i < nw; i += 256) sw[i] = w[i];
  __syncthreads();
  const int sub = threadIdx.x & 7;
  const int64_t npix = (int64_t)B * H * W;
  const int nvec = Cin / EPC;
  for (int64_t pix = (int64_t)blockIdx.x * 32 + (threadIdx.x >> 3); pix < npix;
       pix += (int64_t)gridDim.x * 32) {
    const int ox = (int)(pix % W), oy = (int)((pix / W) % H), b = (int)(pix / ((int64_t)W * H));
    float acc[COUT];
#pragma unroll
    for (int e = 0; e < COUT; ++e) acc[e] = 0.f;
    for (int kh = 0; kh < 3; ++kh) {
      const int iy = oy + kh - 1;
      if ((unsigned)iy >= (unsigned)H) continue;
      for (int kw = 0; kw < 3; ++kw) {
        const int ix = ox + kw - 1;
        if ((unsigned)ix >= (unsigned)W) continue;
        const TI* xp = x + ((int64_t)(b * H + iy) * W + ix) * ldx;
        const float* wp = sw + (kh * 3 + kw) * Cin * COUT;
        for (int v = sub; v < nvec; v += 8) {
          const u32x4 c = *(const u32x4*)(xp + v * EPC);
          float f[EPC];
          chunk_to_f32(c, f, TI());
#pragma unroll
          for (int e = 0; e < EPC; ++e) {
            const float* wr = wp + (v * EPC + e) * COUT;
#pragma unroll
            for (int co = 0; co < COUT; ++co) acc[co] += f[e] * wr[co];
          }
        }
      }
    }
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
      float v = acc[co];
      v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
      acc[co] = v;
    }
    if (sub == 0) {
      TO* op = out + pix * ldo;
#pragma unroll
      for (int co = 0; co < COUT; ++co) stf<TO>(op + co, acc[co] + (bias ? bias[co] : 0.f));
    }
  }
}

// ---- timestep embedding -------------------------------------------------------
__global__ void time_embedding_kernel(const int32_t* __restrict__ t_rows,
                                      const int32_t* __restrict__ steps,
                                      const int32_t* __restrict__ index, float* __restrict__ out,
                                      int rows, int channels) {
  const int half = channels / 2;
  const int total = rows * half;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int r = i / half, k = i - r * half;
    const int t = index ? steps[*index] : t_rows[r];
    // freqs = exp(-ln(10000) * k / half) in float32 (unet.py:413-416)
    const float f = expf(-logf(10000.0f) * (float)k / (float)half);
    const float a = (float)t * f;
    out[(int64_t)r * channels + k] = cosf(a);
    out[(int64_t)r * channels + half + k] = sinf(a);
    if ((channels & 1) && k == 0) out[(int64_t)r * channels + channels - 1] = 0.f;
  }
}

// ---- skinny Dense: one wave per output column n, all rows at once (rows <= 64) ---
template <typename T, int RT>
__global__ __launch_bounds__(256) void gemv_kernel(const float* __restrict__ x, int64_t ldx,
                                                   const T* __restrict__ wt,
                                                   const float* __restrict__ bias,
                                                   float* __restrict__ y, int64_t ldy, int rows, int N,
                                                   int K, int act_in, int act_out) {
  constexpr int EPC = Elem<T>::kPerChunk;
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const int r0 = blockIdx.y * RT;
  float acc[RT];
#pragma unroll
  for (int r = 0; r < RT; ++r) acc[r] = 0.f;
  const T* wr = wt + (int64_t)n * K;
  for (int k = lane * EPC; k < K; k += 64 * EPC) {
    const u32x4 c = *(const u32x4*)(wr + k);
    float wv[EPC];
    chunk_to_f32(c, wv, T());
#pragma unroll
    for (int r = 0; r < RT; ++r) {
      if (r0 + r < rows) {
        const float* xr = x + (int64_t)(r0 + r) * ldx + k;
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
          float xv = xr[e];
          if (act_in == LDM_ACT_SILU) xv = silu_f(xv);
          acc[r] += xv * wv[e];
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < RT; ++r) {
    const float s = wave_sum(acc[r]);
    if (lane == 0 && r0 + r < rows) {
      float v = s + (bias ? bias[n] : 0.f);
      if (act_out == LDM_ACT_SILU) v = silu_f(v);
      y[(int64_t)(r0 + r) * ldy + n] = v;
    }
  }
}

// ---- CFG + DDIM update --------------------------------------------------------------
template <typename TX>
__global__ __launch_bounds__(256) void cfg_ddim_kernel(const float* __restrict__ eps_all,
                                                       const float* __restrict__ xt,
                                                       const float* __restrict__ noise,
                                                       int64_t noise_stride,
                                                       float* __restrict__ xt_out,
                                                       float* __restrict__ pred_x0_out,
                                                       TX* __restrict__ x_unet, const float* coef,
                                                       int32_t* index, int dec_index, float gs,
                                                       int clip, int B, int64_t n) {
  const int idx = *index;
  if (noise) noise += (int64_t)idx * noise_stride;
  const float c1 = coef[idx * 4 + 0], c2 = coef[idx * 4 + 1];
  const float a_prev = coef[idx * 4 + 2], sigma = coef[idx * 4 + 3];
  const float sa = sqrtf(a_prev);
  const float sb = sqrtf(1.0f - a_prev - sigma * sigma);
  const int64_t total = (int64_t)B * n;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const float eu = eps_all[i], ec = eps_all[total + i];
    const float eps = eu + gs * (ec - eu);
    const float x = xt[i];
    float x0 = c1 * x - c2 * eps;
    if (clip) x0 = fminf(fmaxf(x0, -1.f), 1.f);
    const float mean = sa * x0 + sb * eps;
    const float o = mean + (noise ? noise[i] : 0.f) * sigma;
    xt_out[i] = o;
    if (pred_x0_out) pred_x0_out[i] = x0;
    if (x_unet) { stf<TX>(x_unet + i, o); stf<TX>(x_unet + total + i, o); }
  }
}
// decrement happens in its own 1-thread kernel AFTER the update so that every block
// of the update kernel has read *index first
__global__ void dec_index_kernel(int32_t* index) { *index = *index - 1; }

// ---- out[:] = table[i][:], i = *index (pre_decrement: i = --*index first) --------------------------
// ONE workgroup: the thread that moves the loop counter is in the same workgroup as every reader of it, so the
// decrement needs no launch of its own, and every later launch of the step sees the new value.
__global__ __launch_bounds__(1024) void select_row_kernel(const float* __restrict__ table, int64_t ld, int rows,
                                                          int cols, int32_t* __restrict__ index, int pre_decrement,
                                                          float* __restrict__ out) {
  __shared__ int si;
  if (threadIdx.x == 0) {
    int i = *index;
    if (pre_decrement) { i -= 1; *index = i; }
    si = i < 0 ? 0 : (i >= rows ? rows - 1 : i);      // (a stray index reads a valid row; the loop never leaves the table)
  }
  __syncthreads();
  const float* src = table + (int64_t)si * ld;
  for (int c = threadIdx.x * 4; c < cols; c += 1024 * 4) *(f32x4*)(out + c) = *(const f32x4*)(src + c);
}

// ---- post_quant: out = Dense(latents / sf) ----------------------------------------
template <typename TO>
__global__ __launch_bounds__(256) void post_quant_kernel(const float* __restrict__ z, float sf,
                                                         const float* __restrict__ kio,
                                                         const float* __restrict__ bias,
                                                         TO* __restrict__ out, int64_t pixels, int C) {
  for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < pixels; p += (int64_t)gridDim.x * 256) {
    float zi[8];
    for (int i = 0; i < C; ++i) zi[i] = z[p * C + i] / sf;
    for (int o = 0; o < C; ++o) {
      float a = 0.f;
      for (int i = 0; i < C; ++i) a += zi[i] * kio[i * C + o];
      stf<TO>(out + p * C + o, a + (bias ? bias[o] : 0.f));
    }
  }
}

// ---- DiagonalGaussian sample / mode ------------------------------------------------------
__global__ __launch_bounds__(256) void gaussian_sample_kernel(const float* __restrict__ mom,
                                                              const float* __restrict__ noise,
                                                              float* __restrict__ out, float scale,
                                                              int64_t total, int C) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t p = i / C;
    const int c = (int)(i - p * C);
    const float mean = mom[p * 2 * C + c];
    float v = mean;
    if (noise) v = mean + expf(0.5f * mom[p * 2 * C + C + c]) * noise[i];
    out[i] = v * scale;
  }
}

// ---- VQ nearest: one wave per row, lanes stride the codebook ------------------------
__global__ __launch_bounds__(256) void vq_nearest_kernel(const float* __restrict__ z,
                                                         const float* __restrict__ cb,
                                                         float* __restrict__ out,
                                                         int64_t* __restrict__ indices, int64_t rows,
                                                         int V, int C) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float zi[8];
  float zz = 0.f;
  for (int i = 0; i < C; ++i) { zi[i] = z[row * C + i]; zz += zi[i] * zi[i]; }
  float best = INFINITY;
  int bidx = 0x7fffffff;
  for (int v = lane; v < V; v += 64) {
    float ee = 0.f, ze = 0.f;
    for (int i = 0; i < C; ++i) { const float e = cb[(int64_t)v * C + i]; ee += e * e; ze += zi[i] * e; }
    const float d = zz + ee - 2.0f * ze;   // quantize.py:66-70
    if (d < best) { best = d; bidx = v; }
  }
  // argmin with lowest-index tie break (tf.argmin returns the first minimum)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bidx, o, 64);
    if (ob < best || (ob == best && oi < bidx)) { best = ob; bidx = oi; }
  }
  if (lane < C) {
    const float zv = z[row * C + lane];
    const float q = cb[(int64_t)bidx * C + lane];
    out[row * C + lane] = zv + (q - zv);   // quantize.py:87 straight-through form
  }
  if (lane == 0 && indices) indices[row] = bidx;
}

// ---- token + positional embedding ---------------------------------------------------
template <typename TO>
__global__ __launch_bounds__(256) void embedding_kernel(const int64_t* __restrict__ ids,
                                                        const float* __restrict__ tok,
                                                        const float* __restrict__ pos,
                                                        TO* __restrict__ out, int rows, int T, int D,
                                                        int vocab) {
  const int64_t total = (int64_t)rows * T * D;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int d = (int)(i % D);
    const int64_t rt = i / D;
    const int t = (int)(rt % T);
    int64_t id = ids[rt];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    stf<TO>(out + i, tok[id * D + d] + pos[(int64_t)t * D + d]);
  }
}

// ---- min-max -> uint8 ------------------------------------------------------------------
template <typename TI>
__global__ __launch_bounds__(256) void minmax_kernel(const TI* __restrict__ x, float* scratch,
                                                     int64_t n) {
  // grid (nblk, B); scratch[b][nblk][2]
  __shared__ float smin[4], smax[4];
  const int b = blockIdx.y;
  const TI* xb = x + (int64_t)b * n;
  float mn = INFINITY, mx = -INFINITY;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float v = ldf<TI>(xb + i);
    mn = fminf(mn, v); mx = fmaxf(mx, v);
  }
  mn = wave_min(mn); mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) { smin[threadIdx.x >> 6] = mn; smax[threadIdx.x >> 6] = mx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float* o = scratch + ((int64_t)b * gridDim.x + blockIdx.x) * 2;
    o[0] = fminf(fminf(smin[0], smin[1]), fminf(smin[2], smin[3]));
    o[1] = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
  }
}
template <typename TI>
__global__ __launch_bounds__(256) void to_u8_kernel(const TI* __restrict__ x, const float* scratch,
                                                    int nblk, uint8_t* __restrict__ out, int64_t n) {
  const int b = blockIdx.y;
  float mn = INFINITY, mx = -INFINITY;
  for (int i = 0; i < nblk; ++i) {
    mn = fminf(mn, scratch[((int64_t)b * nblk + i) * 2]);
    mx = fmaxf(mx, scratch[((int64_t)b * nblk + i) * 2 + 1]);
  }
  const float range = mx - mn;
  const TI* xb = x + (int64_t)b * n;
  uint8_t* ob = out + (int64_t)b * n;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float v = (ldf<TI>(xb + i) - mn) / range;   // run_ldm_sampler.py:21-22
    v *= 255.0f;                                  // :23
    ob[i] = (uint8_t)v;                           // :24 astype(uint8): truncation
  }
}

template <typename TI, typename TO>
__global__ __launch_bounds__(256) void cast_kernel(const TI* __restrict__ x, int64_t ldx,
                                                   TO* __restrict__ out, int64_t ldo, int64_t rows,
                                                   int cols) {
  const int64_t total = rows * cols;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / cols;
    const int c = (int)(i - r * cols);
    stf<TO>(out + r * ldo + c, ldf<TI>(x + r * ldx + c));
  }
}

inline int grid_for(int64_t total, int per_block = 256, int cap = 4096) {
  int64_t g = (total + per_block - 1) / per_block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}
constexpr int kMinmaxBlocks = 64;

}  // namespace

#define DT_OK(d) ((d) == LDM_F32 || (d) == LDM_BF16)

extern "C" int ldm_conv3x3_small(const void* x, int64_t ldx, int in_dtype, const float* kernel_hwio,
                                 const float* bias, void* out, int64_t ldo, int out_dtype, int B,
                                 int H, int W, int Cin, int Cout, void* stream) {
  LDM_CHECK_ARG(x && kernel_hwio && out, "ldm_conv3x3_small: null pointer");
  LDM_CHECK_ARG(DT_OK(in_dtype) && DT_OK(out_dtype), "ldm_conv3x3_small: bad dtype");
  LDM_CHECK_ARG(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "ldm_conv3x3_small: bad dims");
  hipStream_t s = (hipStream_t)stream;
  const int64_t npix = (int64_t)B * H * W;
  if (Cin == 4 && Cout % 8 == 0 && (size_t)9 * Cin * Cout * sizeof(float) <= 64 * 1024 &&
      ldo % 8 == 0 && ((uintptr_t)out % 16) == 0 && ((uintptr_t)kernel_hwio % 16) == 0 &&
      (in_dtype != LDM_F32 || Cin != 4 || (ldx % 4 == 0 && ((uintptr_t)x % 16) == 0))) {
    const size_t shm = (size_t)9 * Cin * Cout * sizeof(float);
    dim3 g(grid_for(((npix + 3) / 4) * (Cout / 8), 256, 768));
#define LAUNCH_INL(TI, TO, CI)                                                                     \
  hipLaunchKernelGGL((conv_in_lds_kernel<TI, TO, CI>), g, dim3(256), shm, s, (const TI*)x, ldx,    \
                     kernel_hwio, bias, (TO*)out, ldo, B, H, W, Cout)
#define LAUNCH_INL_T(TI, TO) LAUNCH_INL(TI, TO, 4)   /* CIN = 3 spills in this form: it takes the kernel below */
    if (in_dtype == LDM_F32 && out_dtype == LDM_F32) { LAUNCH_INL_T(float, float); }
    else if (in_dtype == LDM_F32) { LAUNCH_INL_T(float, bf16_t); }
    else if (out_dtype == LDM_F32) { LAUNCH_INL_T(bf16_t, float); }
    else { LAUNCH_INL_T(bf16_t, bf16_t); }
#undef LAUNCH_INL_T
#undef LAUNCH_INL
    return ldm_launch_status("ldm_conv3x3_small(in, lds)");
  }
  if (Cin <= 8) {
    LDM_CHECK_ARG(Cout % 4 == 0, "ldm_conv3x3_small: Cout must be a multiple of 4 when Cin <= 8");
    dim3 g(grid_for(npix * (Cout / 4), 256, 8192));
#define LAUNCH_IN(TI, TO)                                                                          \
  hipLaunchKernelGGL((conv_small_in_kernel<TI, TO>), g, dim3(256), 0, s, (const TI*)x, ldx,        \
                     kernel_hwio, bias, (TO*)out, ldo, B, H, W, Cin, Cout)
    if (in_dtype == LDM_F32 && out_dtype == LDM_F32) LAUNCH_IN(float, float);
    else if (in_dtype == LDM_F32) LAUNCH_IN(float, bf16_t);
    else if (out_dtype == LDM_F32) LAUNCH_IN(bf16_t, float);
    else LAUNCH_IN(bf16_t, bf16_t);
#undef LAUNCH_IN
    return ldm_launch_status("ldm_conv3x3_small(in)");
  }
  LDM_CHECK_ARG(Cout <= 4, "ldm_conv3x3_small: needs Cin <= 8 or Cout <= 4 (got %d -> %d)", Cin, Cout);
  const int epc = in_dtype == LDM_BF16 ? 8 : 4;
  LDM_CHECK_ARG(Cin % epc == 0 && ldx % epc == 0 && ((uintptr_t)x % 16) == 0,
                "ldm_conv3x3_small: Cin/ldx/x alignment");
  const size_t shm = (size_t)9 * Cin * Cout * sizeof(float);
  LDM_CHECK_ARG(shm <= 64 * 1024, "ldm_conv3x3_small: Cin*Cout too large for LDS");
  dim3 g(grid_for(npix, 32, 4096));
#define LAUNCH_OUT(TI, TO, CO)                                                                     \
  hipLaunchKernelGGL((conv_small_out_kernel<TI, TO, CO>), g, dim3(256), shm, s, (const TI*)x, ldx, \
                     kernel_hwio, bias, (TO*)out, ldo, B, H, W, Cin)
#define LAUNCH_OUT_T(TI, TO)                        \
  switch (Cout) {                                   \
    case 1: LAUNCH_OUT(TI, TO, 1); break;           \
    case 2: LAUNCH_OUT(TI, TO, 2); break;           \
    case 3: LAUNCH_OUT(TI, TO, 3); break;           \
    default: LAUNCH_OUT(TI, TO, 4); break;          \
  }
  if (in_dtype == LDM_F32 && out_dtype == LDM_F32) { LAUNCH_OUT_T(float, float) }
  else if (in_dtype == LDM_F32) { LAUNCH_OUT_T(float, bf16_t) }
  else if (out_dtype == LDM_F32) { LAUNCH_OUT_T(bf16_t, float) }
  else { LAUNCH_OUT_T(bf16_t, bf16_t) }
#undef LAUNCH_OUT_T
#undef LAUNCH_OUT
  return ldm_launch_status("ldm_conv3x3_small(out)");
}

extern "C" int ldm_time_embedding(const int32_t* t_rows, const int32_t* steps, const int32_t* index,
                                  float* out, int rows, int channels, void* stream) {
  LDM_CHECK_ARG(out && rows > 0 && channels > 1, "ldm_time_embedding: bad args");
  LDM_CHECK_ARG((index && steps) || t_rows, "ldm_time_embedding: need t_rows or (steps, index)");
  hipLaunchKernelGGL(time_embedding_kernel, dim3(grid_for((int64_t)rows * (channels / 2))), dim3(256),
                     0, (hipStream_t)stream, t_rows, steps, index, out, rows, channels);
  return ldm_launch_status("ldm_time_embedding");
}

extern "C" int ldm_gemv(const float* x, int64_t ldx, const void* wt, const float* bias, float* y,
                        int64_t ldy, int rows, int N, int K, int act_in, int act_out, int dtype,
                        void* stream) {
  LDM_CHECK_ARG(x && wt && y, "ldm_gemv: null pointer");
  LDM_CHECK_ARG(DT_OK(dtype), "ldm_gemv: bad dtype");
  const int epc = dtype == LDM_BF16 ? 8 : 4;
  LDM_CHECK_ARG(rows > 0 && N > 0 && K > 0 && K % epc == 0 && ((uintptr_t)wt % 16) == 0,
                "ldm_gemv: K=%d must be a multiple of %d", K, epc);
  hipStream_t s = (hipStream_t)stream;
  constexpr int RT = 4;
  dim3 g((N + 3) / 4, (rows + RT - 1) / RT);
  if (dtype == LDM_BF16)
    hipLaunchKernelGGL((gemv_kernel<bf16_t, RT>), g, dim3(256), 0, s, x, ldx, (const bf16_t*)wt, bias,
                       y, ldy, rows, N, K, act_in, act_out);
  else
    hipLaunchKernelGGL((gemv_kernel<float, RT>), g, dim3(256), 0, s, x, ldx, (const float*)wt, bias, y,
                       ldy, rows, N, K, act_in, act_out);
  return ldm_launch_status("ldm_gemv");
}

extern "C" int ldm_cfg_ddim_update(const float* eps_all, const float* xt, const float* noise,
                                   int64_t noise_index_stride, float* xt_out, float* pred_x0_out,
                                   void* x_unet_out, int x_dtype, const float* coef,
                                   int32_t* index, int dec_index, float guidance_scale,
                                   int clip_denoised, int B, int64_t n_per_sample, void* stream) {
  LDM_CHECK_ARG(eps_all && xt && xt_out && coef && index, "ldm_cfg_ddim_update: null pointer");
  LDM_CHECK_ARG(DT_OK(x_dtype) && B > 0 && n_per_sample > 0, "ldm_cfg_ddim_update: bad args");
  hipStream_t s = (hipStream_t)stream;
  dim3 g(grid_for((int64_t)B * n_per_sample, 256, 1024));
  if (x_dtype == LDM_BF16)
    hipLaunchKernelGGL(cfg_ddim_kernel<bf16_t>, g, dim3(256), 0, s, eps_all, xt, noise, noise_index_stride, xt_out, pred_x0_out,
                       (bf16_t*)x_unet_out, coef, index, dec_index, guidance_scale, clip_denoised, B,
                       n_per_sample);
  else
    hipLaunchKernelGGL(cfg_ddim_kernel<float>, g, dim3(256), 0, s, eps_all, xt, noise, noise_index_stride, xt_out, pred_x0_out,
                       (float*)x_unet_out, coef, index, dec_index, guidance_scale, clip_denoised, B,
                       n_per_sample);
  int st = ldm_launch_status("ldm_cfg_ddim_update");
  if (st != LDM_OK) return st;
  if (dec_index) {
    hipLaunchKernelGGL(dec_index_kernel, dim3(1), dim3(1), 0, s, index);
    st = ldm_launch_status("ldm_cfg_ddim_update(dec)");
  }
  return st;
}

extern "C" int ldm_select_row(const float* table, int64_t ld, int rows, int cols, int32_t* index, int pre_decrement,
                              float* out, void* stream) {
  LDM_CHECK_ARG(table && index && out && rows > 0 && cols > 0, "ldm_select_row: bad args");
  LDM_CHECK_ARG(cols % 4 == 0 && ld % 4 == 0 && ((uintptr_t)table % 16) == 0 && ((uintptr_t)out % 16) == 0,
                "ldm_select_row: cols / ld must be multiples of 4 floats, table / out 16-byte aligned");
  hipLaunchKernelGGL(select_row_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, table, ld, rows, cols, index,
                     pre_decrement, out);
  return ldm_launch_status("ldm_select_row");
}

extern "C" int ldm_post_quant(const float* latents, float scale_factor, const float* kernel_io,
                              const float* bias, void* out, int out_dtype, int64_t pixels, int C,
                              void* stream) {
  LDM_CHECK_ARG(latents && kernel_io && out && pixels > 0 && C > 0 && C <= 8 && DT_OK(out_dtype),
                "ldm_post_quant: bad args");
  hipStream_t s = (hipStream_t)stream;
  dim3 g(grid_for(pixels));
  if (out_dtype == LDM_BF16)
    hipLaunchKernelGGL(post_quant_kernel<bf16_t>, g, dim3(256), 0, s, latents, scale_factor, kernel_io,
                       bias, (bf16_t*)out, pixels, C);
  else
    hipLaunchKernelGGL(post_quant_kernel<float>, g, dim3(256), 0, s, latents, scale_factor, kernel_io,
                       bias, (float*)out, pixels, C);
  return ldm_launch_status("ldm_post_quant");
}

extern "C" int ldm_gaussian_sample(const float* moments, const float* noise, float* out,
                                   float out_scale, int64_t pixels, int C, void* stream) {
  LDM_CHECK_ARG(moments && out && pixels > 0 && C > 0, "ldm_gaussian_sample: bad args");
  dim3 g(grid_for(pixels * C, 256, 4096));
  hipLaunchKernelGGL(gaussian_sample_kernel, g, dim3(256), 0, (hipStream_t)stream, moments, noise, out,
                     out_scale, pixels * C, C);
  return ldm_launch_status("ldm_gaussian_sample");
}

extern "C" int ldm_vq_nearest(const float* z, const float* codebook, float* out, int64_t* indices,
                              int64_t rows, int V, int C, void* stream) {
  LDM_CHECK_ARG(z && codebook && out && rows > 0 && V > 0 && C > 0 && C <= 8, "ldm_vq_nearest: bad args");
  LDM_CHECK_ARG((rows + 3) / 4 < (1ll << 31), "ldm_vq_nearest: too many rows");
  hipLaunchKernelGGL(vq_nearest_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0,
                     (hipStream_t)stream, z, codebook, out, indices, rows, V, C);
  return ldm_launch_status("ldm_vq_nearest");
}

extern "C" int ldm_embedding(const int64_t* ids, const float* tok_emb, const float* pos_emb, void* out,
                             int rows, int T, int D, int vocab, int out_dtype, void* stream) {
  LDM_CHECK_ARG(ids && tok_emb && pos_emb && out && rows > 0 && T > 0 && D > 0 && vocab > 0 &&
                    DT_OK(out_dtype), "ldm_embedding: bad args");
  hipStream_t s = (hipStream_t)stream;
  dim3 g(grid_for((int64_t)rows * T * D));
  if (out_dtype == LDM_BF16)
    hipLaunchKernelGGL(embedding_kernel<bf16_t>, g, dim3(256), 0, s, ids, tok_emb, pos_emb,
                       (bf16_t*)out, rows, T, D, vocab);
  else
    hipLaunchKernelGGL(embedding_kernel<float>, g, dim3(256), 0, s, ids, tok_emb, pos_emb, (float*)out,
                       rows, T, D, vocab);
  return ldm_launch_status("ldm_embedding");
}

extern "C" int ldm_minmax_u8(const void* x, int in_dtype, uint8_t* out, float* scratch, int B,
                             int64_t n_per_image, void* stream) {
  LDM_CHECK_ARG(x && out && scratch && B > 0 && n_per_image > 0 && DT_OK(in_dtype),
                "ldm_minmax_u8: bad args (scratch needs %d floats per image)", 2 * kMinmaxBlocks);
  hipStream_t s = (hipStream_t)stream;
  dim3 g1(kMinmaxBlocks, B), g2(grid_for(n_per_image, 256, 256), B);
  if (in_dtype == LDM_BF16) {
    hipLaunchKernelGGL(minmax_kernel<bf16_t>, g1, dim3(256), 0, s, (const bf16_t*)x, scratch, n_per_image);
    hipLaunchKernelGGL(to_u8_kernel<bf16_t>, g2, dim3(256), 0, s, (const bf16_t*)x, scratch,
                       kMinmaxBlocks, out, n_per_image);
  } else {
    hipLaunchKernelGGL(minmax_kernel<float>, g1, dim3(256), 0, s, (const float*)x, scratch, n_per_image);
    hipLaunchKernelGGL(to_u8_kernel<float>, g2, dim3(256), 0, s, (const float*)x, scratch,
                       kMinmaxBlocks, out, n_per_image);
  }
  return ldm_launch_status("ldm_minmax_u8");
}

extern "C" int ldm_cast(const void* x, int64_t ldx, int in_dtype, void* out, int64_t ldo,
                        int out_dtype, int64_t rows, int cols, void* stream) {
  LDM_CHECK_ARG(x && out && rows > 0 && cols > 0 && DT_OK(in_dtype) && DT_OK(out_dtype),
                "ldm_cast: bad args");
  hipStream_t s = (hipStream_t)stream;
  dim3 g(grid_for(rows * cols));
  if (in_dtype == LDM_F32 && out_dtype == LDM_F32)
    hipLaunchKernelGGL((cast_kernel<float, float>), g, dim3(256), 0, s, (const float*)x, ldx, (float*)out, ldo, rows, cols);
  else if (in_dtype == LDM_F32)
    hipLaunchKernelGGL((cast_kernel<float, bf16_t>), g, dim3(256), 0, s, (const float*)x, ldx, (bf16_t*)out, ldo, rows, cols);
  else if (out_dtype == LDM_F32)
    hipLaunchKernelGGL((cast_kernel<bf16_t, float>), g, dim3(256), 0, s, (const bf16_t*)x, ldx, (float*)out, ldo, rows, cols);
  else
    hipLaunchKernelGGL((cast_kernel<bf16_t, bf16_t>), g, dim3(256), 0, s, (const bf16_t*)x, ldx, (bf16_t*)out, ldo, rows, cols);
  return ldm_launch_status("ldm_cast");
}
