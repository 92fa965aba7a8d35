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
//   * global -> register -> LDS staging, double-buffered: the loads of K-tile
//     t+1 are in flight while tile t is multiplied; one barrier per K-tile.
//   * workgroup ids are remapped so that the tiles sharing an A row-panel run
//     on one XCD (its L2 then serves the panel's re-reads).
//   * small-M layers (4x4 / 8x8 feature maps) stream their weights with split-K
//     over all CUs; partial sums go to an f32 workspace and a second kernel
//     reduces + applies the epilogue.
#include "common.h"

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
  int M, N, K, batch;
  int add_rows;
  int conv, H, W, Cin, OH, OW, stride, upsample;
  int act, out_dtype;
  int split_k, ktiles_per_split, ktiles;
  int tiles_m, tiles_n;
  float alpha;
};

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

// value before activation: alpha*acc + bias[n] + addend[group(m)][n]
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

template <typename T, int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(WM* WN * 64) void gemm_kernel(GemmArgs p) {
  constexpr int NT = WM * WN * 64;
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int CA = BM * 8 / NT, CB = BN * 8 / NT;
  constexpr int EPC = 16 / (int)sizeof(T);
  constexpr int BKE = 8 * EPC;
  constexpr int RSTEP = NT / 8;  // rows covered per staging pass
  static_assert(CA >= 1 && CB >= 1 && TM >= 1 && TN >= 1, "tile");

  __shared__ __attribute__((aligned(16))) char smem[2 * (BM + BN) * 128];
  char* sA = smem;
  char* sB = smem + 2 * BM * 128;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
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

  const T* A = (const T*)p.a + (int64_t)bz * p.stride_a;
  const T* Wt = (const T*)p.w + (int64_t)bz * p.stride_w;

  // ---- per-thread staging geometry ------------------------------------------
  const int ck = tid & 7;
  const int srow = tid >> 3;
  int a_pix[CA], a_iy[CA], a_ix[CA];
#pragma unroll
  for (int i = 0; i < CA; ++i) {
    const int m = m0 + srow + i * RSTEP;
    if (p.conv) {
      if (m < p.M) {
        const int ohw = p.OH * p.OW;
        const int b = m / ohw, rem = m - b * ohw;
        const int oy = rem / p.OW, ox = rem - oy * p.OW;
        a_pix[i] = b * p.H * p.W;
        a_iy[i] = oy * p.stride - 1;
        a_ix[i] = ox * p.stride - 1;
      } else {
        a_pix[i] = 0; a_iy[i] = -(1 << 20); a_ix[i] = 0;
      }
    } else {
      a_pix[i] = m < p.M ? m : -1;
      a_iy[i] = 0; a_ix[i] = 0;
    }
  }
  const int Hs = p.upsample ? p.H * 2 : p.H, Ws = p.upsample ? p.W * 2 : p.W;

  u32x4 ra[CA], rb[CB];
  auto load_tile = [&](int kt) {
    const int k0 = kt * BKE;
    const int kc = k0 + ck * EPC;
    if (p.conv) {
      const int tap = k0 / p.Cin;
      const int ci = k0 - tap * p.Cin + ck * EPC;
      const int kh = tap / 3, kw = tap - kh * 3;
#pragma unroll
      for (int i = 0; i < CA; ++i) {
        int iy = a_iy[i] + kh, ix = a_ix[i] + kw;
        const bool ok = (unsigned)iy < (unsigned)Hs && (unsigned)ix < (unsigned)Ws;
        if (p.upsample) { iy >>= 1; ix >>= 1; }
        u32x4 v = {0u, 0u, 0u, 0u};
        if (ok) v = *(const u32x4*)(A + (int64_t)(a_pix[i] + iy * p.W + ix) * p.lda + ci);
        ra[i] = v;
      }
    } else {
#pragma unroll
      for (int i = 0; i < CA; ++i) {
        u32x4 v = {0u, 0u, 0u, 0u};
        if (a_pix[i] >= 0 && kc < p.K) v = *(const u32x4*)(A + (int64_t)a_pix[i] * p.lda + kc);
        ra[i] = v;
      }
    }
#pragma unroll
    for (int i = 0; i < CB; ++i) {
      const int n = n0 + srow + i * RSTEP;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (n < p.N && kc < p.K) v = *(const u32x4*)(Wt + (int64_t)n * p.K + kc);
      rb[i] = v;
    }
  };
  auto store_tile = [&](int buf) {
    char* dA = sA + buf * (BM * 128);
    char* dB = sB + buf * (BN * 128);
#pragma unroll
    for (int i = 0; i < CA; ++i) {
      const int row = srow + i * RSTEP;
      *(u32x4*)(dA + row * 128 + ((ck ^ ((row >> 1) & 7)) << 4)) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < CB; ++i) {
      const int row = srow + i * RSTEP;
      *(u32x4*)(dB + row * 128 + ((ck ^ ((row >> 1) & 7)) << 4)) = rb[i];
    }
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
  const int offA = (wm * WTM + lr) * 128;
  const int offB = (wn * WTN + lr) * 128;

  if (kt_begin < kt_end) {
    load_tile(kt_begin);
    store_tile(0);
  }
  __syncthreads();
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    const int buf = (kt - kt_begin) & 1;
    if (kt + 1 < kt_end) load_tile(kt + 1);
    const char* cA = sA + buf * (BM * 128) + offA;
    const char* cB = sB + buf * (BN * 128) + offB;
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) {
      const int coff = ((kg * 2 + lh) ^ sw) << 4;
      u32x4 fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[i] = *(const u32x4*)(cA + i * 32 * 128 + coff);
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[j] = *(const u32x4*)(cB + j * 32 * 128 + coff);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) mma32(acc[i][j], fa[i], fb[j], T());
    }
    if (kt + 1 < kt_end) store_tile(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue ---------------------------------------------------------------
  const int mb = m0 + wm * WTM + 4 * lh;
  const int nb = n0 + wn * WTN + lr;
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

struct TileCfg { int bm, bn; };
// index 1..4 (0 = auto)
constexpr TileCfg kTiles[5] = {{0, 0}, {256, 64}, {128, 128}, {128, 64}, {64, 64}};

template <typename T>
void launch_cfg(int cfg, const GemmArgs& a, dim3 grid, hipStream_t s) {
  switch (cfg) {
    case 1: hipLaunchKernelGGL((gemm_kernel<T, 256, 64, 4, 1>), grid, dim3(256), 0, s, a); break;
    case 2: hipLaunchKernelGGL((gemm_kernel<T, 128, 128, 2, 2>), grid, dim3(256), 0, s, a); break;
    case 3: hipLaunchKernelGGL((gemm_kernel<T, 128, 64, 2, 2>), grid, dim3(256), 0, s, a); break;
    default: hipLaunchKernelGGL((gemm_kernel<T, 64, 64, 2, 2>), grid, dim3(256), 0, s, a); break;
  }
}

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// Choose tile config + split-K for a problem.  256 CUs, 2 resident blocks per CU.
void choose(const ldm_gemm_params* p, int esize, int* cfg_out, int* split_out) {
  const int bke = 128 / esize;
  const int ktiles = cdiv(p->K, bke);
  int cfg = p->tile;
  if (cfg <= 0 || cfg > 4) {
    const bool geglu = p->act == LDM_ACT_GEGLU;
    double best = -1;
    cfg = 2;
    for (int c = 1; c <= 4; ++c) {
      if (geglu && c > 2) continue;
      const TileCfg t = kTiles[c];
      const double tiles = (double)cdiv(p->M, t.bm) * cdiv(p->N, t.bn) * p->batch;
      const double useful = (double)p->M * p->N * p->batch / (tiles * t.bm * t.bn);
      // machine fill: tiles are executed in waves of 512 resident blocks
      const double waves = tiles / 512.0;
      const double fill = waves >= 1.0 ? waves / (double)((int64_t)(waves + 0.999999)) : 1.0;
      // arithmetic intensity preference (bigger tiles re-read less through L2/LDS)
      const double ai = (double)t.bm * t.bn / (t.bm + t.bn) / 64.0;  // 1.0 for 128x128
      double score = useful * fill * (0.75 + 0.25 * ai);
      if (tiles < 256.0 && ktiles < 8) score *= tiles / 256.0;  // cannot be rescued by split-K
      if (score > best) { best = score; cfg = c; }
    }
  }
  int split = p->split_k;
  if (split <= 0) {
    split = 1;
    if (p->batch == 1) {
      const TileCfg t = kTiles[cfg];
      const int tiles = cdiv(p->M, t.bm) * cdiv(p->N, t.bn);
      if (tiles < 192 && ktiles >= 8) {
        split = (384 + tiles - 1) / tiles;
        if (split > ktiles / 4) split = ktiles / 4;
        if (split > 32) split = 32;
        if (split < 1) split = 1;
      }
    }
  }
  if (p->batch != 1) split = 1;
  *cfg_out = cfg;
  *split_out = split;
}

}  // namespace

extern "C" size_t ldm_gemm_workspace_bytes(const ldm_gemm_params* p) {
  if (!p) return 0;
  int cfg, split;
  choose(p, p->dtype == LDM_BF16 ? 2 : 4, &cfg, &split);
  return split > 1 ? (size_t)split * p->M * p->N * 4 : 0;
}

extern "C" int ldm_gemm(const ldm_gemm_params* p, void* stream) {
  LDM_CHECK_ARG(p && p->a && p->w && p->out, "ldm_gemm: null pointer");
  LDM_CHECK_ARG(p->dtype == LDM_F32 || p->dtype == LDM_BF16, "ldm_gemm: bad dtype %d", p->dtype);
  LDM_CHECK_ARG(p->out_dtype == LDM_F32 || p->out_dtype == LDM_BF16, "ldm_gemm: bad out_dtype");
  LDM_CHECK_ARG(p->M > 0 && p->N > 0 && p->K > 0 && p->batch > 0, "ldm_gemm: bad M/N/K/batch");
  const int esize = p->dtype == LDM_BF16 ? 2 : 4;
  const int epc = 16 / esize, bke = 8 * epc;
  LDM_CHECK_ARG(p->K % epc == 0, "ldm_gemm: K=%d must be a multiple of %d", p->K, epc);
  LDM_CHECK_ARG(((uintptr_t)p->a % 16) == 0 && ((uintptr_t)p->w % 16) == 0,
                "ldm_gemm: a/w must be 16-byte aligned");
  LDM_CHECK_ARG(p->lda % epc == 0 && p->stride_a % epc == 0 && p->stride_w % epc == 0,
                "ldm_gemm: lda/stride_a/stride_w must be multiples of %d elements", epc);
  if (p->conv) {
    LDM_CHECK_ARG(p->Cin > 0 && p->Cin % bke == 0, "ldm_gemm(conv): Cin=%d must be a multiple of %d",
                  p->Cin, bke);
    LDM_CHECK_ARG(p->K == 9 * p->Cin, "ldm_gemm(conv): K must be 9*Cin");
    LDM_CHECK_ARG(p->stride == 1 || p->stride == 2, "ldm_gemm(conv): stride must be 1 or 2");
    LDM_CHECK_ARG(p->B > 0 && p->H > 0 && p->W > 0 && p->OH > 0 && p->OW > 0, "ldm_gemm(conv): dims");
    const int hs = p->upsample ? 2 * p->H : p->H, wsz = p->upsample ? 2 * p->W : p->W;
    LDM_CHECK_ARG(p->OH == (hs + 2 - 3) / p->stride + 1 && p->OW == (wsz + 2 - 3) / p->stride + 1,
                  "ldm_gemm(conv): OH/OW inconsistent with H/W/stride/upsample");
    LDM_CHECK_ARG(p->M == p->B * p->OH * p->OW, "ldm_gemm(conv): M != B*OH*OW");
    LDM_CHECK_ARG(p->batch == 1, "ldm_gemm(conv): batch must be 1");
  }
  if (p->addend) LDM_CHECK_ARG(p->add_rows > 0, "ldm_gemm: add_rows must be > 0 with addend");
  if (p->act == LDM_ACT_GEGLU) {
    LDM_CHECK_ARG(p->N % 64 == 0, "ldm_gemm: GEGLU needs N %% 64 == 0");
    LDM_CHECK_ARG(p->ldc_n == 1, "ldm_gemm: GEGLU needs a row-major output");
  }
  LDM_CHECK_ARG(p->ldc_n == 1 || p->ldc_m == 1, "ldm_gemm: one of ldc_m / ldc_n must be 1");

  int cfg, split;
  choose(p, esize, &cfg, &split);
  if (p->act == LDM_ACT_GEGLU) LDM_CHECK_ARG(cfg <= 2, "ldm_gemm: GEGLU needs tile 1 or 2");
  GemmArgs a;
  memset(&a, 0, sizeof(a));
  a.a = (const char*)p->a; a.w = (const char*)p->w; a.bias = p->bias; a.addend = p->addend;
  a.residual = (const char*)p->residual; a.out = (char*)p->out; a.ws = (float*)p->workspace;
  a.lda = p->lda; a.ldr = p->ldr; a.ldc_m = p->ldc_m; a.ldc_n = p->ldc_n;
  a.stride_a = p->stride_a; a.stride_w = p->stride_w; a.stride_c = p->stride_c; a.stride_r = p->stride_r;
  a.add_ld = p->add_ld; a.M = p->M; a.N = p->N; a.K = p->K; a.batch = p->batch;
  a.add_rows = p->add_rows > 0 ? p->add_rows : 1;
  a.conv = p->conv; a.H = p->H; a.W = p->W; a.Cin = p->Cin; a.OH = p->OH; a.OW = p->OW;
  a.stride = p->stride; a.upsample = p->upsample; a.act = p->act; a.out_dtype = p->out_dtype;
  a.alpha = p->alpha;
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
  if (p->dtype == LDM_BF16) launch_cfg<bf16_t>(cfg, a, grid, s);
  else launch_cfg<float>(cfg, a, grid, s);
  int st = ldm_launch_status("ldm_gemm");
  if (st != LDM_OK) return st;
  if (split > 1) {
    const int nout = p->act == LDM_ACT_GEGLU ? p->N / 2 : p->N;
    int64_t total = (int64_t)p->M * nout;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(splitk_epilogue_kernel, dim3(blocks), dim3(256), 0, s, a);
    st = ldm_launch_status("ldm_gemm(splitk epilogue)");
  }
  return st;
}
