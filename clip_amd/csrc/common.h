// Shared device/host helpers for the clip_amd HIP library (gfx950 / MI355X only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;  // raw 16-bit operand bits (bf16, or fp16 in the f16 kernel set)
typedef __attribute__((ext_vector_type(8))) short s16x8_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// ---- the 16-bit OPERAND type of the MFMA kernels.  Every kernel file is compiled twice: into the
// bf16 kernel set (struct k_bf16, the default numerics setting) and, with -DCLIP_OPERAND_F16=1, into
// the fp16 kernel set (struct k_f16: v_mfma_*_f16 at the same rate, 11 significant bits instead of 8,
// fp16's range — CLIP_PRECISION_F16).  Same code, same LDS images, same schedules; only the element
// type, the MFMA opcode and the conversions below differ.
#ifndef CLIP_OPERAND_F16
#define CLIP_OPERAND_F16 0
#endif
#if CLIP_OPERAND_F16
#define KSET k_f16
typedef _Float16 op_t;
#define MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)
#define MFMA_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
__device__ __forceinline__ float op_to_f32(bf16_t v) {
  return (float)*reinterpret_cast<const _Float16*>(&v);
}
// the two halves of a 32-bit word holding two operands
__device__ __forceinline__ float op2_lo(uint32_t w) { return op_to_f32((bf16_t)(w & 0xffffu)); }
__device__ __forceinline__ float op2_hi(uint32_t w) { return op_to_f32((bf16_t)(w >> 16)); }
#else
#define KSET k_bf16
typedef __bf16 op_t;
#define MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#define MFMA_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
__device__ __forceinline__ float op_to_f32(bf16_t v) {
  return __uint_as_float(((uint32_t)v) << 16);
}
__device__ __forceinline__ float op2_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float op2_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
#endif
typedef __attribute__((ext_vector_type(8))) op_t opx8_t;
// round-to-nearest-even; a plain cast lowers to v_cvt_pk_bf16_f32 / v_cvt_f16_f32 on gfx950 and keeps
// NaN a NaN (fp16: a value beyond +-65504 becomes inf)
__device__ __forceinline__ bf16_t f32_to_op(float x) {
  op_t b = (op_t)x;
  return *reinterpret_cast<bf16_t*>(&b);
}
// two values -> one packed word in ONE v_cvt_pk_{bf16,f16}_f32 (the same round-to-nearest-even as f32_to_op;
// written as two scalar casts the compiler paired the wrong elements and repaired the word with and / shift /
// sdwa-or: 6 instructions per 4 values in every 16-bit epilogue instead of 2)
__device__ __forceinline__ uint32_t pack_op2(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) float pk_f32x2_t;
  typedef __attribute__((ext_vector_type(2))) op_t pk_opx2_t;
  const pk_opx2_t r = __builtin_convertvector((pk_f32x2_t){lo, hi}, pk_opx2_t);
  return *reinterpret_cast<const uint32_t*>(&r);
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

// LayerNorm of one element from its row statistics.  One definition for the LayerNorm kernels and
// for the GEMM epilogues that RE-compute the fp32 LayerNorm output as their residual operand
// (EPI_BIAS_LNRES): the residual stream then never passes through a bf16 rounding.
__device__ __forceinline__ float ln_apply(float x, float mean, float rstd, float g, float b) {
  return fmaf((x - mean) * rstd, g, b);
}

// Element (r, c) of a 16-bit [M, C] GEMM operand kept as contiguous 128 x 64 blocks (GemmArgs::a_tiled / out_tiled;
// M a multiple of 128, C of 64): a 128-byte row segment stays whole, a block is 16 KiB.
__host__ __device__ __forceinline__ size_t operand_tiled_offset(size_t r, int c, int C) {
  return ((((r >> 7) * (size_t)(C >> 6)) + (size_t)(c >> 6)) << 13) + ((r & 127) << 6) + (size_t)(c & 63);
}

// The fp16 residual copy of the fused-LayerNorm flow is kept TILED for the accumulator layout of the
// 256x256 GEMM tile, so that the workgroup that wrote a tile reads it back (as the next residual)
// with whole 1-KiB wave accesses and no transpose.  Element (row, col) of an [M, N] tensor, N a
// multiple of 256, M padded to 256, tiles_n = N / 256:
//   tile = (row >> 8) * tiles_n + (col >> 8);  r = row & 255, c = col & 255
//   wave = ((r >> 6) & 1) * 4 + (c >> 6)          (wr, wc)
//   blk  = (r >> 7) * 4 + ((r >> 4) & 3)          (mh, m: the 16-row block)
//   lane = ((c >> 2) & 3) * 16 + (r & 15)         (fq, fr)
//   nh = (c >> 5) & 1, i = (c >> 4) & 1, e = c & 3
//   offset = ((tile * 8 + wave) * 8 + blk) * 1024 + nh * 512 + lane * 8 + i * 4 + e
__host__ __device__ __forceinline__ size_t xres_tiled_offset(int row, int col, int tiles_n) {
  const int r = row & 255, c = col & 255;
  const size_t tile = (size_t)(row >> 8) * tiles_n + (col >> 8);
  const int wave = ((r >> 6) & 1) * 4 + (c >> 6), blk = (r >> 7) * 4 + ((r >> 4) & 3);
  const int lane = ((c >> 2) & 3) * 16 + (r & 15);
  return ((tile * 8 + wave) * 8 + blk) * 1024 + ((c >> 5) & 1) * 512 + lane * 8 + ((c >> 4) & 1) * 4 + (c & 3);
}

// The same tile order for an fp32 residual stream (CLIP_PRECISION_F16R): inside a wave's (blk, nh) piece the two
// 16-column blocks i = 0, 1 come one after the other, so that a lane's four values of a block are one 16-byte access
// and a wave's access is 1 KiB contiguous:   offset = ((tile * 8 + wave) * 8 + blk) * 1024 + nh * 512 + i * 256 + lane * 4 + e
__host__ __device__ __forceinline__ size_t xres32_tiled_offset(int row, int col, int tiles_n) {
  const int r = row & 255, c = col & 255;
  const size_t tile = (size_t)(row >> 8) * tiles_n + (col >> 8);
  const int wave = ((r >> 6) & 1) * 4 + (c >> 6), blk = (r >> 7) * 4 + ((r >> 4) & 3);
  const int lane = ((c >> 2) & 3) * 16 + (r & 15);
  return ((tile * 8 + wave) * 8 + blk) * 1024 + ((c >> 5) & 1) * 512 + ((c >> 4) & 1) * 256 + lane * 4 + (c & 3);
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

// erf-GELU for bf16 OUTPUTS: gelu(x) = x (0.5 + xc R(xc^2)), xc = clamp(x, +-3.9), R a degree-6
// minimax fit (LP, weight x^2) of 0.5 erf(x / sqrt 2) / x; no transcendental, 10 VALU ops.
// |gelu error| <= 1.46e-4 absolute for |x| <= 12 (checked in float32 Horner form), i.e. below 1/6 of
// a bf16 half-ulp of any output >= 0.25 and vanishing as x^2 near 0 — the output is rounded to bf16
// right after.  The GEMM epilogue is VALU-bound on this function (it cannot overlap the MFMAs of the
// same wave pair), so ops count: 40 op-equivalents for libm erff, 14 for the degree-8 form used
// before.  f32 outputs (check mode) keep gelu_erf.
__device__ __forceinline__ float gelu_erf_bf16out(float x) {
#ifdef GELU_DEG8  // diagnostic builds (tools/build_variant.sh): the degree-8 form, |error| <= 6e-5
  const float x8 = __builtin_amdgcn_fmed3f(x, -4.6669048f, 4.6669048f);
  const float u8 = x8 * x8;
  float q8 = fmaf(4.616259299e-11f, u8, -5.357747270e-09f);
  q8 = fmaf(q8, u8, 2.722674172e-07f);
  q8 = fmaf(q8, u8, -8.016578673e-06f);
  q8 = fmaf(q8, u8, 1.529851537e-04f);
  q8 = fmaf(q8, u8, -2.011764160e-03f);
  q8 = fmaf(q8, u8, 1.894713759e-02f);
  q8 = fmaf(q8, u8, -1.315939645e-01f);
  q8 = fmaf(q8, u8, 7.972131005e-01f);
  const float e8 = __builtin_amdgcn_fmed3f(q8 * x8, -1.0f, 1.0f);
  return fmaf(0.5f * x, e8, 0.5f * x);
#endif
  const float xc = __builtin_amdgcn_fmed3f(x, -3.9f, 3.9f);
  const float u = xc * xc;
  float q = fmaf(2.627150074e-08f, u, -1.770499288e-06f);
  q = fmaf(q, u, 5.123743311e-05f);
  q = fmaf(q, u, -8.445144153e-04f);
  q = fmaf(q, u, 8.913803536e-03f);
  q = fmaf(q, u, -6.486952130e-02f);
  q = fmaf(q, u, 3.980856433e-01f);
  return x * fmaf(xc, q, 0.5f);
}

// The same polynomial on two values at a time with packed fp32 math (v_pk_mul_f32 / v_pk_fma_f32):
// 2 clamps + 9 packed ops per pair instead of 20.  Bit-identical to gelu_erf_bf16out per element.
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
__device__ __forceinline__ f32x2_t gelu_erf_bf16out_x2(f32x2_t x) {
  f32x2_t xc;
  xc[0] = __builtin_amdgcn_fmed3f(x[0], -3.9f, 3.9f);
  xc[1] = __builtin_amdgcn_fmed3f(x[1], -3.9f, 3.9f);
  const f32x2_t u = xc * xc;
  auto c = [](float v) { return (f32x2_t){v, v}; };
  f32x2_t q = __builtin_elementwise_fma(c(2.627150074e-08f), u, c(-1.770499288e-06f));
  q = __builtin_elementwise_fma(q, u, c(5.123743311e-05f));
  q = __builtin_elementwise_fma(q, u, c(-8.445144153e-04f));
  q = __builtin_elementwise_fma(q, u, c(8.913803536e-03f));
  q = __builtin_elementwise_fma(q, u, c(-6.486952130e-02f));
  q = __builtin_elementwise_fma(q, u, c(3.980856433e-01f));
  return x * __builtin_elementwise_fma(xc, q, c(0.5f));
}

// 16-byte chunk swizzle for LDS tiles with 128-byte rows (8 chunks per row): conflict-free for
// the ds_read_b128 operand reads of both the 16x16x32 and the 32x32x16 MFMA.
__device__ __forceinline__ int swz128(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

// Two pairs at once, statement by statement: the two dependent Horner chains interleave, so a v_pk_fma never
// waits on the one just issued (the one-pair form left an s_nop after every packed op once the scheduler ran
// out of independent work).  Same bits per element as gelu_erf_bf16out.
__device__ __forceinline__ void gelu_erf_bf16out_x4(f32x2_t& a, f32x2_t& b) {
  auto c = [](float v) { return (f32x2_t){v, v}; };
  f32x2_t ac, bc;
  ac[0] = __builtin_amdgcn_fmed3f(a[0], -3.9f, 3.9f); ac[1] = __builtin_amdgcn_fmed3f(a[1], -3.9f, 3.9f);
  bc[0] = __builtin_amdgcn_fmed3f(b[0], -3.9f, 3.9f); bc[1] = __builtin_amdgcn_fmed3f(b[1], -3.9f, 3.9f);
  const f32x2_t ua = ac * ac, ub = bc * bc;
  f32x2_t qa = __builtin_elementwise_fma(c(2.627150074e-08f), ua, c(-1.770499288e-06f));
  f32x2_t qb = __builtin_elementwise_fma(c(2.627150074e-08f), ub, c(-1.770499288e-06f));
  qa = __builtin_elementwise_fma(qa, ua, c(5.123743311e-05f));  qb = __builtin_elementwise_fma(qb, ub, c(5.123743311e-05f));
  qa = __builtin_elementwise_fma(qa, ua, c(-8.445144153e-04f)); qb = __builtin_elementwise_fma(qb, ub, c(-8.445144153e-04f));
  qa = __builtin_elementwise_fma(qa, ua, c(8.913803536e-03f));  qb = __builtin_elementwise_fma(qb, ub, c(8.913803536e-03f));
  qa = __builtin_elementwise_fma(qa, ua, c(-6.486952130e-02f)); qb = __builtin_elementwise_fma(qb, ub, c(-6.486952130e-02f));
  qa = __builtin_elementwise_fma(qa, ua, c(3.980856433e-01f));  qb = __builtin_elementwise_fma(qb, ub, c(3.980856433e-01f));
  a = a * __builtin_elementwise_fma(ac, qa, c(0.5f));
  b = b * __builtin_elementwise_fma(bc, qb, c(0.5f));
}
