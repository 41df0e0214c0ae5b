// bf16 MFMA GEMM for the encoder projections:  out[M,N] = epi(A[M,K] . W[N,K]^T + bias[N])
//   A = activations (row-major, K contiguous), W = nn.Linear weight [out,in] (K contiguous).
// Restates the dense layers of transformers BertSelfAttention/BertSelfOutput/BertIntermediate/
// BertOutput (modeling_bert.py:175-177, 289-293, 334-337, 347-351) that the reference's
// sentclf/bert_model.py:79-88 runs; the epilogues fuse bias, erf-GELU, residual add and tanh.
//
// Tile: 128x128x64 per 256-thread workgroup (4 waves, 2x2, 64x64 each, 16x16x32 MFMA).
// Operands are staged global->LDS with global_load_lds_dwordx4 (no VGPR hop), double-buffered,
// one barrier per K tile.  LDS rows are 128 B; 16-B chunks are XOR-swizzled (swz128) on the
// SOURCE address (the LDS image written by the DMA is lane-linear) and on the ds_read_b128.
// MFMA orientation: a = W fragment (rows n), b = activation fragment (cols m), so each lane ends
// with 4 consecutive n of one row m -> 8/16-byte row-major stores.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = 128 * BK * 2;  // 16 KiB per operand tile

template <int EPI, bool OUT_F32>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(
    const bf16_t* __restrict__ A, const bf16_t* __restrict__ W, const float* __restrict__ bias,
    const bf16_t* __restrict__ res, void* __restrict__ out, int M, int N, int K, int lda, int ldw,
    int ldo, int ldres, int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 buf][A 16K | W 16K]

  // XCD-aware bijective remap: blocks b and b+8 share an XCD (round-robin dispatch), give each
  // XCD a contiguous run of tiles so neighbouring tiles (same A panel) hit one L2.
  const int nwg = tiles_m * tiles_n;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int tm = wg / tiles_n, tn = wg - tm * tiles_n;
  const int mb = tm * BM, nb = tn * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 1, wn = w & 1;

  // ---- DMA source pointers: instruction q of wave w fills LDS rows (q*4+w)*8 .. +8
  const bf16_t* a_src[4];
  const bf16_t* w_src[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = (q * 4 + w) * 8 + (lane >> 3);
    const int chunk = swz128(row, lane & 7);
    const int ar = min(mb + row, M - 1), wr = min(nb + row, N - 1);
    a_src[q] = A + (size_t)ar * lda + chunk * 8;
    w_src[q] = W + (size_t)wr * ldw + chunk * 8;
  }
  auto stage = [&](int kt, int buf) {
    char* base = smem + buf * (2 * TILE_BYTES);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      __builtin_amdgcn_global_load_lds(GLB_PTR(a_src[q] + kt * BK),
                                       LDS_PTR(base + (q * 4 + w) * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(GLB_PTR(w_src[q] + kt * BK),
                                       LDS_PTR(base + TILE_BYTES + (q * 4 + w) * 1024), 16, 0, 0);
    }
  };

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fq = lane >> 4;
  const int lsw = (fr >> 1) & 7;  // swizzle key of this lane's rows (row = 16*i + fr)
  const int a_off = (wm * 64 + fr) * 128;
  const int w_off = TILE_BYTES + (wn * 64 + fr) * 128;

  const int nt = K / BK;
  stage(0, 0);
  for (int t = 0; t < nt; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t + 1 < nt) stage(t + 1, (t + 1) & 1);
    const char* base = smem + (t & 1) * (2 * TILE_BYTES);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int coff = ((ks * 4 + fq) ^ lsw) * 16;
      bf16x8_t af[4], wf[4];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        af[j] = *reinterpret_cast<const bf16x8_t*>(base + a_off + j * 16 * 128 + coff);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        wf[i] = *reinterpret_cast<const bf16x8_t*>(base + w_off + i * 16 * 128 + coff);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], af[j], acc[i][j], 0, 0, 0);
    }
  }

  // ---- epilogue: lane holds n = n0..n0+3 of row m for every (i,j)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int n0 = nb + wn * 64 + i * 16 + fq * 4;
    if (n0 >= N) continue;
    const float4 bv = *reinterpret_cast<const float4*>(bias + n0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = mb + wm * 64 + j * 16 + fr;
      if (m >= M) continue;
      float v0 = acc[i][j][0] + bv.x, v1 = acc[i][j][1] + bv.y;
      float v2 = acc[i][j][2] + bv.z, v3 = acc[i][j][3] + bv.w;
      if (EPI == EPI_BIAS_GELU) {
        v0 = gelu_erf(v0); v1 = gelu_erf(v1); v2 = gelu_erf(v2); v3 = gelu_erf(v3);
      } else if (EPI == EPI_BIAS_TANH) {
        v0 = tanhf(v0); v1 = tanhf(v1); v2 = tanhf(v2); v3 = tanhf(v3);
      } else if (EPI == EPI_BIAS_RES) {
        const uint2 rv = *reinterpret_cast<const uint2*>(res + (size_t)m * ldres + n0);
        v0 += __uint_as_float(rv.x << 16); v1 += __uint_as_float(rv.x & 0xffff0000u);
        v2 += __uint_as_float(rv.y << 16); v3 += __uint_as_float(rv.y & 0xffff0000u);
      }
      if (OUT_F32) {
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + (size_t)m * ldo + n0) =
            make_float4(v0, v1, v2, v3);
      } else {
        *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(out) + (size_t)m * ldo + n0) =
            make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
      }
    }
  }
}

template <int EPI, bool OUT_F32>
hipError_t launch_one(const GemmArgs& g, hipStream_t st) {
  const int tiles_m = (g.M + BM - 1) / BM, tiles_n = (g.N + BN - 1) / BN;
  auto kfn = gemm_bf16_kernel<EPI, OUT_F32>;
  static bool attr_set = false;  // per instantiation
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILE_BYTES);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL(kfn, dim3(tiles_m * tiles_n), dim3(256), 4 * TILE_BYTES, st, g.A, g.W, g.bias,
                     g.res, g.out, g.M, g.N, g.K, g.lda, g.ldw, g.ldo, g.ldres, tiles_m, tiles_n);
  return hipGetLastError();
}

}  // namespace

const char* gemm_check(const GemmArgs& g) {
  if (g.M <= 0 || g.N <= 0 || g.K <= 0) return "gemm: empty problem";
  if (g.K % BK) return "gemm: K must be a multiple of 64";
  if (g.N % 16) return "gemm: N must be a multiple of 16";
  if (g.lda % 8 || g.ldw % 8 || g.ldo % 4) return "gemm: leading dimensions must keep 16-byte rows";
  if (g.epi == EPI_BIAS_RES && (!g.res || g.ldres % 4)) return "gemm: residual missing/misaligned";
  if (!g.A || !g.W || !g.bias || !g.out) return "gemm: null operand";
  return nullptr;
}

hipError_t launch_gemm(const GemmArgs& g, hipStream_t st) {
  switch (g.epi * 2 + (g.out_f32 ? 1 : 0)) {
    case EPI_BIAS * 2 + 0: return launch_one<EPI_BIAS, false>(g, st);
    case EPI_BIAS * 2 + 1: return launch_one<EPI_BIAS, true>(g, st);
    case EPI_BIAS_GELU * 2 + 0: return launch_one<EPI_BIAS_GELU, false>(g, st);
    case EPI_BIAS_RES * 2 + 1: return launch_one<EPI_BIAS_RES, true>(g, st);
    case EPI_BIAS_RES * 2 + 0: return launch_one<EPI_BIAS_RES, false>(g, st);
    case EPI_BIAS_TANH * 2 + 0: return launch_one<EPI_BIAS_TANH, false>(g, st);
    case EPI_BIAS_TANH * 2 + 1: return launch_one<EPI_BIAS_TANH, true>(g, st);
    default: return hipErrorInvalidValue;
  }
}
