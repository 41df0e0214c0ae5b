// Shared device/host helpers for the clip_amd HIP library (gfx950 / MI355X only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;  // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

__device__ __forceinline__ float bf16_to_f32(bf16_t v) {
  return __uint_as_float(((uint32_t)v) << 16);
}
// round-to-nearest-even; a plain cast lowers to v_cvt_pk_bf16_f32 on gfx950 and keeps NaN a NaN
__device__ __forceinline__ bf16_t f32_to_bf16(float x) {
  __bf16 b = (__bf16)x;
  return *reinterpret_cast<bf16_t*>(&b);
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  return (uint32_t)f32_to_bf16(lo) | ((uint32_t)f32_to_bf16(hi) << 16);
}

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

// erf-GELU as torch's nn.GELU() default.  erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7,
// below fp32 resolution of the surrounding sum and ~5x cheaper than libm erff in a GEMM epilogue).
__device__ __forceinline__ float erf_as(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(-ax * ax * 1.4426950408889634f);
  const float r = fmaf(-p * t, e, 1.0f);
  return copysignf(r, x);
}
__device__ __forceinline__ float gelu_erf(float x) {
  return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752440f));
}

// erf-GELU for bf16 OUTPUTS: erf(x/sqrt2) = x Q(x^2) with a degree-8 weighted-minimax Q on |x| <= 4.667,
// saturated outside; no transcendental.  |gelu error| <= 6e-5 absolute over all x (fitted and
// checked in float32 Horner form), i.e. below half a bf16 ulp of any output >= 0.03 — the output is
// rounded to bf16 right after.  14 VALU ops instead of ~40 op-equivalents: the FFN-up epilogue was
// 25 % of that kernel.  f32 outputs (check mode) keep gelu_erf.
__device__ __forceinline__ float gelu_erf_bf16out(float x) {
  const float xc = __builtin_amdgcn_fmed3f(x, -4.6669048f, 4.6669048f);
  const float u = xc * xc;
  float q = fmaf(4.616259299e-11f, u, -5.357747270e-09f);
  q = fmaf(q, u, 2.722674172e-07f);
  q = fmaf(q, u, -8.016578673e-06f);
  q = fmaf(q, u, 1.529851537e-04f);
  q = fmaf(q, u, -2.011764160e-03f);
  q = fmaf(q, u, 1.894713759e-02f);
  q = fmaf(q, u, -1.315939645e-01f);
  q = fmaf(q, u, 7.972131005e-01f);
  const float e = __builtin_amdgcn_fmed3f(q * xc, -1.0f, 1.0f);
  const float hx = 0.5f * x;
  return fmaf(hx, e, hx);
}

// 16-byte chunk swizzle for LDS tiles with 128-byte rows (8 chunks per row): conflict-free for
// the ds_read_b128 operand reads of both the 16x16x32 and the 32x32x16 MFMA.
__device__ __forceinline__ int swz128(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

