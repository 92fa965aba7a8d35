// Shared device/host helpers for libldm_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/ldm_hip.h"

typedef uint16_t bf16_t;  // raw bfloat16 bits

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

// ---- error plumbing --------------------------------------------------------
void ldm_set_error(const char* fmt, ...);
#define LDM_CHECK_ARG(cond, ...)     \
  do {                               \
    if (!(cond)) {                   \
      ldm_set_error(__VA_ARGS__);    \
      return LDM_ERR_ARG;            \
    }                                \
  } while (0)

static inline int ldm_launch_status(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    ldm_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return LDM_ERR_LAUNCH;
  }
  return LDM_OK;
}

// ---- scalar conversions ----------------------------------------------------
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// f32 -> bf16, round-to-nearest-even, NaN stays NaN: gfx950's v_cvt_pk_bf16_f32 converts
// two values in ONE instruction (a bit-twiddled software rounding costs ~6 VALU ops per
// value and was the hottest part of the attention softmax and of every bf16 epilogue).
// Written as a vector convert (not inline asm) so that hipcc emits the instruction itself
// and pads the VALU-write -> MFMA-read hazard when the result feeds an MFMA operand.
typedef __attribute__((ext_vector_type(2))) __bf16 ldm_bf16x2;
typedef __attribute__((ext_vector_type(2))) float ldm_f32x2;
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
  const ldm_f32x2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, ldm_bf16x2));
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack_bf2(f, 0.0f) & 0xffffu); }

template <typename T> struct Elem;
template <> struct Elem<float> {
  static constexpr int kPerChunk = 4;  // elements per 16-byte chunk
  __device__ static __forceinline__ float ld(const float* p) { return *p; }
  __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
  static constexpr int kPerChunk = 8;
  __device__ static __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
  __device__ static __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
};

// unpack one 16-byte chunk into floats
__device__ __forceinline__ void chunk_to_f32(const u32x4& c, float (&f)[4], float) {
  f[0] = __uint_as_float(c[0]); f[1] = __uint_as_float(c[1]);
  f[2] = __uint_as_float(c[2]); f[3] = __uint_as_float(c[3]);
}
__device__ __forceinline__ void chunk_to_f32(const u32x4& c, float (&f)[8], bf16_t) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = __uint_as_float(c[i] << 16);
    f[2 * i + 1] = __uint_as_float(c[i] & 0xffff0000u);
  }
}
__device__ __forceinline__ u32x4 f32_to_chunk(const float (&f)[4], float) {
  u32x4 c;
  c[0] = __float_as_uint(f[0]); c[1] = __float_as_uint(f[1]);
  c[2] = __float_as_uint(f[2]); c[3] = __float_as_uint(f[3]);
  return c;
}
__device__ __forceinline__ u32x4 f32_to_chunk(const float (&f)[8], bf16_t) {
  u32x4 c;
#pragma unroll
  for (int i = 0; i < 4; ++i) c[i] = pack_bf2(f[2 * i], f[2 * i + 1]);
  return c;
}

// ---- math ------------------------------------------------------------------
// x * sigmoid(x) with the hardware exp2 / reciprocal (1 ulp each): 4 instructions instead
// of the ~15 of an IEEE division -- SiLU runs on every GroupNorm output element.
__device__ __forceinline__ float silu_f(float x) {
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
// Exact-erf GELU (tf.nn.gelu default).  erf by Abramowitz & Stegun 7.1.26 (|abs err| <=
// 1.5e-7, i.e. float32 rounding level) on the hardware rcp / exp2: ~14 VALU ops where the
// device library's erff costs ~40 -- the GEGLU epilogue applies it to 4C values per token.
__device__ __forceinline__ float erf_as_f(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);
  float poly = 1.061405429f;
  poly = __builtin_fmaf(poly, t, -1.453152027f);
  poly = __builtin_fmaf(poly, t, 1.421413741f);
  poly = __builtin_fmaf(poly, t, -0.284496736f);
  poly = __builtin_fmaf(poly, t, 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * ax * ax);
  const float r = 1.0f - poly * t * e;
  return copysignf(r, x);
}
// 0.5 x (1 + erf(x / sqrt 2)) with the same A&S erf, rearranged so that the sign handling and the
// "1 -" disappear: erf(z) = sign (1 - q), q = poly(t) t exp(-z^2)  =>  gelu(x) = max(x, 0) - 0.5 |x| q.
// 13 full-rate + 2 transcendental VALU operations instead of 17 + 2 (the GEGLU epilogue of the persistent
// kernel is exposed VALU time); same approximation error (1.5e-7), no cancellation on either side of 0.
__device__ __forceinline__ float gelu_erf_f(float x) {
  const float ax = fabsf(x);
  const float z = ax * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, z, 1.0f));
  float poly = 1.061405429f;
  poly = __builtin_fmaf(poly, t, -1.453152027f);
  poly = __builtin_fmaf(poly, t, 1.421413741f);
  poly = __builtin_fmaf(poly, t, -0.284496736f);
  poly = __builtin_fmaf(poly, t, 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);
  const float q = poly * t * e;                       // 1 - erf(z), in [0, 1]
  return __builtin_fmaf(-0.5f * ax, q, fmaxf(x, 0.f));
}

// ---- wave reductions (wave = 64) ------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}
