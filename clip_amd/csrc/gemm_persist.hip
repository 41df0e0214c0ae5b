// Persistent form of the 256x256x64 bf16 MFMA GEMM (see gemm.hip for the math and the LDS image):
//   out[M,N] = epi(A[M,K] . W[N,K]^T + bias[N]),  M, N multiples of 256, K a multiple of 128.
// One workgroup per CU walks its share of the output tiles.  What the one-tile-per-workgroup
// kernel leaves serial at 1 workgroup/CU is overlapped here:
//   * the DMA of the NEXT tile's first K tile is issued before the current tile's last MFMAs, so
//     no tile but the first pays the global->LDS latency (prologue);
//   * the epilogue's global stores are left in flight: the next tile's first wait is a COUNTED
//     s_waitcnt vmcnt(#stores) (loads, stores and LDS-DMA retire in issue order), and all
//     barriers are raw s_barrier — a __syncthreads() fence would drain the stores;
//   * output rows are written with non-temporal 16-byte stores (measured -7..-13 % kernel time:
//     the streamed output no longer evicts the A/W panels from L2);
//   * the bias slice of the next tile is parked in LDS one K tile ahead (its load is older than
//     every DMA it could queue behind).
// The epilogue transposes each wave's 128x64 sub-tile through its own 8-KiB slice of staging
// buffer 1 (idle between the last K tile and the next tile's second), buffer 0 already receives the
// next tile.  Tile order: see the schedule comment in the kernel (XCD-owned row panels, column
// tiles in L2-sized groups).
#include "common.h"
#include "kernels.h"

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
constexpr int BK = 64, BM = 256, BN = 256;
#ifndef GEMM_ABLATE
#define GEMM_ABLATE 0  // diagnostic builds: 4 no global stores, 5 no epilogue
#endif
constexpr int A_BYTES = BM * BK * 2, BUF_BYTES = 2 * A_BYTES;  // 32 KiB, 64 KiB
constexpr int LDS_BYTES = 2 * BUF_BYTES + 2 * BN * 4;          // staging + bias[2][256]

template <int EPI, bool OUT_F32>
__global__ __launch_bounds__(512) void gemm_persist_kernel(
    const bf16_t* __restrict__ A, const bf16_t* __restrict__ W, const float* __restrict__ bias,
    const bf16_t* __restrict__ res, void* __restrict__ out, int K, int lda, int ldw, int ldo,
    int ldres, int tiles_m, int tiles_n, int group_n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* bias_lds = reinterpret_cast<float*>(smem + 2 * BUF_BYTES);

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 2, wn = w & 3;
  // Tile schedule.  The workgroups of one XCD (blockIdx % 8: they share an L2) walk ONE sequence,
  // workgroup `slot` taking items slot, slot + per_x, ...  group_n > 0: the XCD owns tiles_m / 8
  // consecutive row panels and sweeps them once per group of `group_n` column tiles, so the W
  // slice of a group (<= 1.6 MB) stays in its 4-MiB L2 while A panels stream (PMC: the FFN-up GEMM
  // fetched 605 MB with the plain order, W being re-fetched every round).  group_n == 0: plain
  // order, 32 consecutive tiles (n fastest) per XCD per round.
  const int G = gridDim.x, per_x = G >> 3;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int n_tiles = tiles_m * tiles_n;
  const int panels = tiles_m >> 3;              // row panels per XCD (group_n > 0)
  const int items = panels * tiles_n;      // per XCD (group_n > 0)
  auto tile_at = [&](int i, int& tm_o, int& tn_o) -> bool {
    if (group_n > 0) {
      const int q = slot + per_x * i;
      if (q >= items) return false;
      const int per_group = panels * group_n;
      const int g = q / per_group, r = q - g * per_group;
      const int m = r / group_n;
      tm_o = xcd * panels + m;
      tn_o = g * group_n + (r - m * group_n);
      return true;
    }
    const int t = i * G + xcd * per_x + slot;
    if (t >= n_tiles) return false;
    tm_o = t / tiles_n;
    tn_o = t - tm_o * tiles_n;
    return true;
  };
  int tm, tn;
  if (!tile_at(0, tm, tn)) return;

  // DMA: piece p = 8 q + w fills LDS rows 8p .. 8p+7; the source chunk carries the swizzle
  int a_off[4], w_off[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = (q * 8 + w) * 8 + (lane >> 3);
    a_off[q] = row * lda + swz128(row, lane & 7) * 8;
    w_off[q] = row * ldw + swz128(row, lane & 7) * 8;
  }
  auto stage = [&](const bf16_t* a_base, const bf16_t* w_base, int kt, int buf) {
    char* base = smem + buf * BUF_BYTES;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      __builtin_amdgcn_global_load_lds(GLB_PTR(a_base + a_off[q] + kt * BK),
                                       LDS_PTR(base + (q * 8 + w) * 1024), 16, 0, 0);
#pragma unroll
    for (int q = 0; q < 4; ++q)
      __builtin_amdgcn_global_load_lds(GLB_PTR(w_base + w_off[q] + kt * BK),
                                       LDS_PTR(base + A_BYTES + (q * 8 + w) * 1024), 16, 0, 0);
  };

  const int fr = lane & 15, fq = lane >> 4;
  const int lsw = (fr >> 1) & 7;
  const int a_rd = (wm * 128 + fr) * 128;
  const int w_rd = A_BYTES + (wn * 64 + fr) * 128;
  const int nt = K / BK;

  int it = 0;
  const bf16_t* a_base = A + (size_t)tm * BM * lda;
  const bf16_t* w_base = W + (size_t)tn * BN * ldw;
  if (w == 0) *reinterpret_cast<float4*>(bias_lds + lane * 4) =
      *reinterpret_cast<const float4*>(bias + tn * BN + lane * 4);
  stage(a_base, w_base, 0, 0);

  while (true) {
    int ntm = 0, ntn = 0;
    const bool has_next = tile_at(it + 1, ntm, ntn);
    const bf16_t* na_base = A + (size_t)ntm * BM * lda;
    const bf16_t* nw_base = W + (size_t)ntn * BN * ldw;
    float4 nbias = make_float4(0.f, 0.f, 0.f, 0.f);

    f32x4_t acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    for (int t = 0; t < nt; ++t) {
      // K tile t has landed: everything older than the previous epilogue's stores is done
      if (t == 0 && it > 0) {
        if (OUT_F32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (t + 1 < nt) {
        if (t == nt - 2 && has_next && w == 0)  // older than the DMA below: its wait costs nothing
          nbias = *reinterpret_cast<const float4*>(bias + ntn * BN + lane * 4);
        stage(a_base, w_base, t + 1, (t + 1) & 1);
      } else if (has_next) {
        if (w == 0) *reinterpret_cast<float4*>(bias_lds + ((it + 1) & 1) * BN + lane * 4) = nbias;
        stage(na_base, nw_base, 0, 0);
      }
      const char* base = smem + (t & 1) * BUF_BYTES;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int coff = ((ks * 4 + fq) ^ lsw) * 16;
        bf16x8_t af[8], wf[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
          wf[i] = *reinterpret_cast<const bf16x8_t*>(base + w_rd + i * 16 * 128 + coff);
#pragma unroll
        for (int j = 0; j < 8; ++j)
          af[j] = *reinterpret_cast<const bf16x8_t*>(base + a_rd + j * 16 * 128 + coff);
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
          for (int i = 0; i < 4; ++i)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], af[j], acc[i][j], 0, 0, 0);
      }
    }

    // ---- epilogue through this wave's 8-KiB slice of staging buffer 1
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // every wave is done reading the last K tile (buffer 1)
    char* ep = smem + BUF_BYTES + w * 8192;
    const float* bl = bias_lds + (it & 1) * BN + wn * 64;
    float4 bv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) bv[i] = *reinterpret_cast<const float4*>(bl + i * 16 + fq * 4);
    const int m_base = tm * BM + wm * 128, n_base = tn * BN + wn * 64;
    if (GEMM_ABLATE == 5) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("" ::"v"(acc[i][j]));
    } else if (!OUT_F32) {
#pragma unroll
      for (int ps = 0; ps < 2; ++ps) {  // 64 rows per pass, image [64][128 B]
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            const int j = ps * 4 + jj;
            float v0 = acc[i][j][0] + bv[i].x, v1 = acc[i][j][1] + bv[i].y;
            float v2 = acc[i][j][2] + bv[i].z, v3 = acc[i][j][3] + bv[i].w;
            if (EPI == EPI_BIAS_GELU) {
              v0 = gelu_erf_bf16out(v0); v1 = gelu_erf_bf16out(v1); v2 = gelu_erf_bf16out(v2); v3 = gelu_erf_bf16out(v3);
            } else if (EPI == EPI_BIAS_TANH) {
              v0 = tanhf(v0); v1 = tanhf(v1); v2 = tanhf(v2); v3 = tanhf(v3);
            }
            const int row = jj * 16 + fr, slot = (i * 4 + fq) ^ ((row & 7) << 1);
            *reinterpret_cast<uint2*>(ep + row * 128 + slot * 8) =
                make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
          }
#pragma unroll
        for (int p = 0; p < 8; ++p) {
          const int row = p * 8 + (lane >> 3), c16 = lane & 7;
          const size_t m = m_base + ps * 64 + row;
          const int n = n_base + c16 * 8;
          uint4 v = *reinterpret_cast<const uint4*>(ep + row * 128 + ((c16 ^ (row & 7)) * 16));
          if (EPI == EPI_BIAS_RES) {
            const uint4 r = *reinterpret_cast<const uint4*>(res + m * ldres + n);
            const uint32_t* vi = reinterpret_cast<const uint32_t*>(&v);
            const uint32_t* ri = reinterpret_cast<const uint32_t*>(&r);
            uint32_t o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
              o[k] = pack_bf16x2(__uint_as_float(vi[k] << 16) + __uint_as_float(ri[k] << 16),
                                 __uint_as_float(vi[k] & 0xffff0000u) + __uint_as_float(ri[k] & 0xffff0000u));
            v = make_uint4(o[0], o[1], o[2], o[3]);
          }
          if (GEMM_ABLATE == 4) { asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w)); } else
          // streamed once, read by the next kernel from HBM: keep it out of the L2 the operands live in
          __builtin_nontemporal_store(*reinterpret_cast<u32x4_t*>(&v), reinterpret_cast<u32x4_t*>(reinterpret_cast<bf16_t*>(out) + m * ldo + n));
        }
      }
    } else {
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) {  // 32 rows per pass, image [32][256 B]
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) {
            const int j = ps * 2 + jj;
            float v0 = acc[i][j][0] + bv[i].x, v1 = acc[i][j][1] + bv[i].y;
            float v2 = acc[i][j][2] + bv[i].z, v3 = acc[i][j][3] + bv[i].w;
            if (EPI == EPI_BIAS_GELU) {
              v0 = gelu_erf(v0); v1 = gelu_erf(v1); v2 = gelu_erf(v2); v3 = gelu_erf(v3);
            } else if (EPI == EPI_BIAS_TANH) {
              v0 = tanhf(v0); v1 = tanhf(v1); v2 = tanhf(v2); v3 = tanhf(v3);
            }
            const int row = jj * 16 + fr, slot = (i * 4 + fq) ^ (row & 15);
            *reinterpret_cast<float4*>(ep + row * 256 + slot * 16) = make_float4(v0, v1, v2, v3);
          }
#pragma unroll
        for (int p = 0; p < 8; ++p) {
          const int row = p * 4 + (lane >> 4), c16 = lane & 15;
          const size_t m = m_base + ps * 32 + row;
          const int n = n_base + c16 * 4;
          float4 v = *reinterpret_cast<const float4*>(ep + row * 256 + ((c16 ^ (row & 15)) * 16));
          if (EPI == EPI_BIAS_RES) {
            const uint2 rv = *reinterpret_cast<const uint2*>(res + m * ldres + n);
            v.x += __uint_as_float(rv.x << 16); v.y += __uint_as_float(rv.x & 0xffff0000u);
            v.z += __uint_as_float(rv.y << 16); v.w += __uint_as_float(rv.y & 0xffff0000u);
          }
          if (GEMM_ABLATE == 4) { asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w)); } else
          __builtin_nontemporal_store(*reinterpret_cast<f32x4_t*>(&v), reinterpret_cast<f32x4_t*>(reinterpret_cast<float*>(out) + m * ldo + n));
        }
      }
    }
    if (!has_next) break;
    tm = ntm; tn = ntn; a_base = na_base; w_base = nw_base;
    ++it;
  }
}

template <int EPI, bool OUT_F32>
hipError_t launch_p(const GemmArgs& g, hipStream_t st) {
  auto kfn = gemm_persist_kernel<EPI, OUT_F32>;
  static int n_cu = 0;
  if (n_cu == 0) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) return e;
    int dev = 0, cus = 0;
    if ((e = hipGetDevice(&dev)) != hipSuccess) return e;
    if ((e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return e;
    n_cu = cus & ~7;
    if (n_cu < 8) return hipErrorInvalidValue;
  }
  const int tiles_m = g.M / BM, tiles_n = g.N / BN, n_tiles = tiles_m * tiles_n;
  const int grid = n_cu < ((n_tiles + 7) & ~7) ? n_cu : ((n_tiles + 7) & ~7);
  // column-tile group whose W slice (group_n x 256 rows x K) fits an XCD's L2 beside the A stream
  int group_n = 0;
  if (tiles_m % 8 == 0) {
    group_n = tiles_n;
    if ((long)tiles_n * BN * g.K * 2 > 1700000 && g.K < 2048)
      for (int d = 1; d <= tiles_n; ++d)
        if (tiles_n % d == 0 && (long)d * BN * g.K * 2 <= 1700000) group_n = d;
  }
  hipLaunchKernelGGL(kfn, dim3(grid), dim3(512), LDS_BYTES, st, static_cast<const bf16_t*>(g.A),
                     static_cast<const bf16_t*>(g.W), g.bias, static_cast<const bf16_t*>(g.res), g.out, g.K,
                     g.lda, g.ldw, g.ldo, g.ldres, tiles_m, tiles_n, group_n);
  return hipGetLastError();
}

}  // namespace

bool gemm_persist_ok(const GemmArgs& g) {
  return !g.f32_in && g.M % BM == 0 && g.N % BN == 0 && g.K % (2 * BK) == 0 && g.M > 0 && g.N > 0 && g.K > 0;
}

hipError_t launch_gemm_persist(const GemmArgs& g, hipStream_t st) {
  if (!gemm_persist_ok(g)) return hipErrorInvalidValue;
  switch (g.epi * 2 + (g.out_f32 ? 1 : 0)) {
    case EPI_BIAS * 2 + 0: return launch_p<EPI_BIAS, false>(g, st);
    case EPI_BIAS * 2 + 1: return launch_p<EPI_BIAS, true>(g, st);
    case EPI_BIAS_GELU * 2 + 0: return launch_p<EPI_BIAS_GELU, false>(g, st);
    case EPI_BIAS_RES * 2 + 1: return launch_p<EPI_BIAS_RES, true>(g, st);
    case EPI_BIAS_RES * 2 + 0: return launch_p<EPI_BIAS_RES, false>(g, st);
    case EPI_BIAS_TANH * 2 + 0: return launch_p<EPI_BIAS_TANH, false>(g, st);
    default: return hipErrorInvalidValue;
  }
}
