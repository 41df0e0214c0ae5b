// Sustained MFMA peak of THIS board: a register-only v_mfma_f32_16x16x32_{bf16,f16} loop on random
// operands, every CU, 2 waves per SIMD, nothing but MFMAs in the loop (no LDS, no memory).  SURVEY.md
// §8(d) asks for this number before any roofline fraction is quoted: the 2.5 PFLOP/s nameplate assumes
// 2.4 GHz, and the chip holds less than that under a dense MFMA stream (power), by an amount that
// differs a few percent between boards.  bench.py reports roofline.frac against the nameplate AND
// frac_sustained against this measurement.  No counterpart in the reference.
#include "common.h"
#include "../../include/clip_amd.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 pk_bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 pk_f16x8_t;

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

// operands: random sign and mantissa, magnitude in [1, 2) (bf16: exponent 0x3f8; f16: 0x3c00) — data that
// toggles like real activations, never NaN / inf
template <bool F16>
__global__ __launch_bounds__(512) void mfma_peak_kernel(int iters, float* __restrict__ sink) {
  const uint32_t id = blockIdx.x * 512 + threadIdx.x;
  uint32_t w[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const uint32_t h = mix(id * 8 + i + 1);
    w[i] = F16 ? ((h & 0x83ff83ffu) | 0x3c003c00u) : ((h & 0x807f807fu) | 0x3f803f80u);
  }
  f32x4_t acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) {  // 8 independent accumulator chains: the pipe never waits for a result
        if (F16) {
          const pk_f16x8_t a = *reinterpret_cast<const pk_f16x8_t*>(&w[0]);
          const pk_f16x8_t b = *reinterpret_cast<const pk_f16x8_t*>(&w[4]);
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(r ? b : a, r ? a : b, acc[i], 0, 0, 0);
        } else {
          const pk_bf16x8_t a = *reinterpret_cast<const pk_bf16x8_t*>(&w[0]);
          const pk_bf16x8_t b = *reinterpret_cast<const pk_bf16x8_t*>(&w[4]);
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(r ? b : a, r ? a : b, acc[i], 0, 0, 0);
        }
      }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 1.2345e-30f) sink[id] = s;  // keeps the loop alive; never true
}

// the same loop on v_mfma_f32_32x32x16_bf16 (half the operand-register reads per flop): 4 chains of 16 registers
__global__ __launch_bounds__(512) void mfma_peak32_kernel(int iters, float* __restrict__ sink) {
  const uint32_t id = blockIdx.x * 512 + threadIdx.x;
  uint32_t w[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) w[i] = (mix(id * 8 + i + 1) & 0x807f807fu) | 0x3f803f80u;
  f32x16_t acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const pk_bf16x8_t a = *reinterpret_cast<const pk_bf16x8_t*>(&w[0]);
        const pk_bf16x8_t b = *reinterpret_cast<const pk_bf16x8_t*>(&w[4]);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(r ? b : a, r ? a : b, acc[i], 0, 0, 0);
      }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) s += acc[i][e];
  if (s == 1.2345e-30f) sink[id] = s;
}

}  // namespace

// diagnostic: the 32x32x16 shape (8 MFMAs of 32768 flops per iteration and wave); tools/peak_shapes.py
extern "C" int clip_debug_mfma_peak32(int32_t iters, int32_t repeats, double* tflops_mean, double* ms_mean) {
  int dev = 0, cus = 0;
  if (!tflops_mean || !ms_mean || iters < 1 || repeats < 1) return CLIP_ERR_ARG;
  if (hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
    return CLIP_ERR_HIP;
  float* sink = nullptr;
  if (hipMalloc(&sink, (size_t)cus * 512 * 4) != hipSuccess) return CLIP_ERR_HIP;
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  double sum_ms = 0.0;
  int rc = CLIP_OK;
  for (int r = 0; r < repeats + 1; ++r) {
    (void)hipEventRecord(a, nullptr);
    hipLaunchKernelGGL(mfma_peak32_kernel, dim3(cus), dim3(512), 0, nullptr, iters, sink);
    (void)hipEventRecord(b, nullptr);
    if (hipEventSynchronize(b) != hipSuccess) { rc = CLIP_ERR_HIP; break; }
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, a, b);
    if (r) sum_ms += ms;
  }
  (void)hipEventDestroy(a); (void)hipEventDestroy(b);
  (void)hipFree(sink);
  *ms_mean = sum_ms / repeats;
  *tflops_mean = (double)cus * 8 * iters * 8 * (2.0 * 32 * 32 * 16) / (*ms_mean * 1e-3) / 1e12;
  return rc;
}

extern "C" int clip_mfma_peak(int32_t operand, int32_t iters, int32_t repeats, double* tflops_best,
                              double* tflops_mean, double* ms_mean) {
  if (!tflops_best || !tflops_mean || !ms_mean || iters < 1 || repeats < 1 ||
      (operand != CLIP_PRECISION_BF16 && operand != CLIP_PRECISION_F16))
    return CLIP_ERR_ARG;
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
    return CLIP_ERR_HIP;
  float* sink = nullptr;
  if (hipMalloc(&sink, (size_t)cus * 512 * 4) != hipSuccess) return CLIP_ERR_HIP;
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  double best = 0.0, sum_ms = 0.0;
  const double flops = (double)cus * 8 /*waves*/ * iters * 16 /*MFMAs*/ * (2.0 * 16 * 16 * 32);
  int rc = CLIP_OK;
  for (int r = 0; r < repeats + 1 && rc == CLIP_OK; ++r) {  // first launch: warm-up, not counted
    (void)hipEventRecord(a, nullptr);
    if (operand == CLIP_PRECISION_F16)
      hipLaunchKernelGGL(mfma_peak_kernel<true>, dim3(cus), dim3(512), 0, nullptr, iters, sink);
    else
      hipLaunchKernelGGL(mfma_peak_kernel<false>, dim3(cus), dim3(512), 0, nullptr, iters, sink);
    (void)hipEventRecord(b, nullptr);
    if (hipEventSynchronize(b) != hipSuccess) { rc = CLIP_ERR_HIP; break; }
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, a, b);
    if (r == 0) continue;
    sum_ms += ms;
    const double tf = flops / (ms * 1e-3) / 1e12;
    if (tf > best) best = tf;
  }
  (void)hipEventDestroy(a); (void)hipEventDestroy(b);
  (void)hipFree(sink);
  *ms_mean = sum_ms / repeats;
  *tflops_mean = flops / (*ms_mean * 1e-3) / 1e12;
  *tflops_best = best;
  return rc;
}
