// bf16 MFMA GEMM for the encoder projections:  out[M,N] = epi(A[M,K] . W[N,K]^T + bias[N])
//   A = activations (row-major, K contiguous), W = nn.Linear weight [out,in] (K contiguous).
// Restates the dense layers of transformers BertSelfAttention/BertSelfOutput/BertIntermediate/
// BertOutput/BertPooler (modeling_bert.py:175-177, 289-293, 334-337, 347-351, 457-463) that the
// reference's sentclf/bert_model.py:79-88 runs; the epilogues fuse bias, erf-GELU, residual add,
// tanh.
//
// One kernel template, two tile shapes (16x16x32 MFMA, BK = 64):
//   256x256 tile, 8 waves (2 M x 4 N, 128x64 per wave), 128 KiB LDS, 1 workgroup/CU  — the
//     encoder GEMMs (M = tokens of a step).  Per K tile a wave issues 24 ds_read_b128 for 64
//     MFMAs: half the LDS traffic per flop of the 128x128 tile.
//   128x128 tile, 4 waves (2 x 2, 64x64 per wave), 64 KiB LDS — small M (pooler, trimmed layer).
// Operands are staged global->LDS with global_load_lds_dwordx4 (no VGPR hop) through a ring of LDS buffers (2, or 4 for
// 128-row tiles on a grid of at most one workgroup per CU); DMA runs ahead of the MFMAs behind counted vmcnt, one
// barrier per K tile.  LDS
// rows are 128 B; 16-B chunks are XOR-swizzled (swz128) on the SOURCE address (the image the DMA
// writes is lane-linear) and on the ds_read_b128: conflict-free operand reads.
// MFMA orientation: a = W fragment (rows n), b = activation fragment (cols m), so each lane ends
// with 4 consecutive n of one row m -> 8/16-byte row-major stores.
#include "common.h"
#include "kernels.h"

namespace {

#ifndef GEMM_ABLATE
#define GEMM_ABLATE 0  // diagnostic builds: 1 no MFMA, 2 no DMA in loop, 3 no epilogue stores
#endif

// F32IN = the fp32 check mode of the SAME kernel: operands and residual are fp32, the K tile is 32
// elements (LDS rows stay 128 B, same swizzle, same ds_read_b128), the MFMA is the exact-fp32
// mfma_f32_16x16x4f32 (a k-ordered fmaf chain, 1/16 of the bf16 rate).  A lane's 16-byte chunk
// holds 4 consecutive k of its row; MFMA m of a k-step takes element m of every lane's chunk, so
// its k set is {4 fq + m}: any order works as long as both operands use the same one.
// NST: LDS ring depth (2: double buffer, 64 KiB for the 128 x 128 tile, two workgroups per CU; 4: three K tiles in
// flight, 128 KiB, one workgroup per CU — the launcher takes 4 when the grid has at most one workgroup per CU anyway)
template <int EPI, bool OUT_F32, int WAVES_M, int WAVES_N, int TM, int TN, bool F32IN = false, int NST = 2>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N) void gemm_bf16_kernel(
    const void* __restrict__ Av, const void* __restrict__ Wv, const float* __restrict__ bias,
    const void* __restrict__ resv, void* __restrict__ out, int M, int N, int K, int lda, int ldw,
    int ldo, int ldres, int tiles_m, int tiles_n, const float2* __restrict__ ln_stats,
    const float* __restrict__ ln_gamma, const float* __restrict__ ln_beta,
    const int32_t* __restrict__ res_index, int res_tiles_n) {
  static_assert(!F32IN || OUT_F32, "fp32 operands imply fp32 output");
  constexpr int ESZ = F32IN ? 4 : 2;       // operand element bytes
  constexpr int BKE = 128 / ESZ;           // K elements per tile (128-byte LDS rows)
  const char* A = reinterpret_cast<const char*>(Av);
  const char* W = reinterpret_cast<const char*>(Wv);
  const bf16_t* res = reinterpret_cast<const bf16_t*>(resv);
  constexpr int NW = WAVES_M * WAVES_N;
  constexpr int BM = WAVES_M * TM * 16, BN = WAVES_N * TN * 16;
  constexpr int A_BYTES = BM * 128, W_BYTES = BN * 128, BUF_BYTES = A_BYTES + W_BYTES;
  constexpr int A_PIECES = BM / 8 / NW, W_PIECES = BN / 8 / NW;  // 1-KiB DMA pieces per wave
  static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "tile rows must split over the waves");
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [NST buf][A | W]

  // XCD-aware bijective remap: blocks b and b+8 share an XCD (round-robin dispatch); give each
  // XCD a contiguous run of tiles, n fastest, so neighbouring tiles (same A panel) hit one L2.
  const int nwg = tiles_m * tiles_n;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int tm = wg / tiles_n, tn = wg - tm * tiles_n;
  const int mb = tm * BM, nb = tn * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w / WAVES_N, wn = w % WAVES_N;

  // ---- DMA source pointers: piece p = q*NW + w fills LDS rows 8p .. 8p+7 (1 KiB, lane-linear)
  const char* a_src[A_PIECES];
  const char* w_src[W_PIECES];
#pragma unroll
  for (int q = 0; q < A_PIECES; ++q) {
    const int row = (q * NW + w) * 8 + (lane >> 3);
    a_src[q] = A + ((size_t)min(mb + row, M - 1) * lda) * ESZ + swz128(row, lane & 7) * 16;
  }
#pragma unroll
  for (int q = 0; q < W_PIECES; ++q) {
    const int row = (q * NW + w) * 8 + (lane >> 3);
    w_src[q] = W + ((size_t)min(nb + row, N - 1) * ldw) * ESZ + swz128(row, lane & 7) * 16;
  }
  auto stage = [&](int kt, int buf) {
    char* base = smem + buf * BUF_BYTES;
#pragma unroll
    for (int q = 0; q < A_PIECES; ++q)
      __builtin_amdgcn_global_load_lds(GLB_PTR(a_src[q] + kt * 128),
                                       LDS_PTR(base + (q * NW + w) * 1024), 16, 0, 0);
#pragma unroll
    for (int q = 0; q < W_PIECES; ++q)
      __builtin_amdgcn_global_load_lds(GLB_PTR(w_src[q] + kt * 128),
                                       LDS_PTR(base + A_BYTES + (q * NW + w) * 1024), 16, 0, 0);
  };

  f32x4_t acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fq = lane >> 4;
  const int lsw = (fr >> 1) & 7;  // swizzle key of this lane's rows (row = 16*t + fr)
  const int a_off = (wm * TM * 16 + fr) * 128;
  const int w_off = A_BYTES + (wn * TN * 16 + fr) * 128;

  // NST - 1 K tiles in flight: with one (the double buffer) every K tile of a small call waits out a full load latency.
  constexpr int D = NST - 1, PIECES = A_PIECES + W_PIECES;
  const int nt = K / BKE;
#pragma unroll
  for (int s = 0; s < D; ++s)
    if (s < nt) stage(s, s);
  for (int t = 0; t < nt; ++t) {
    // K tile t has landed for this wave: only the tiles issued after it (PIECES DMA instructions each) may be outstanding
    const int rem = nt - 1 - t < D - 1 ? nt - 1 - t : D - 1;
    if (rem == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    else if (rem == 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(PIECES) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * PIECES) : "memory");
    __builtin_amdgcn_s_barrier();   // (not __syncthreads(): its fence would drain every DMA in flight)
    if (t + D < nt && GEMM_ABLATE != 2) stage(t + D, (t + D) % NST);
    const char* base = smem + (t % NST) * BUF_BYTES;
    if constexpr (!F32IN) {
      // all 2 x (TN + TM) fragment reads of the tile are issued up front; the compiler's counted
      // lgkmcnt waits let the first MFMAs start while the later reads are still in flight
      opx8_t af[2][TM], wf[2][TN];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int coff = ((ks * 4 + fq) ^ lsw) * 16;
#pragma unroll
        for (int i = 0; i < TN; ++i)
          wf[ks][i] = *reinterpret_cast<const opx8_t*>(base + w_off + i * 16 * 128 + coff);
#pragma unroll
        for (int j = 0; j < TM; ++j)
          af[ks][j] = *reinterpret_cast<const opx8_t*>(base + a_off + j * 16 * 128 + coff);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
          for (int i = 0; i < TN; ++i)
            if (GEMM_ABLATE == 1) {
              asm volatile("" ::"v"(wf[ks][i]), "v"(af[ks][j]));
            } else {
              acc[i][j] = MFMA_16x16x32(wf[ks][i], af[ks][j], acc[i][j]);
            }
    } else {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {  // 16 fp32 k per step = 4 MFMAs of k = 4
        const int coff = ((ks * 4 + fq) ^ lsw) * 16;
        f32x4_t af[TM], wf[TN];
#pragma unroll
        for (int i = 0; i < TN; ++i)
          wf[i] = *reinterpret_cast<const f32x4_t*>(base + w_off + i * 16 * 128 + coff);
#pragma unroll
        for (int j = 0; j < TM; ++j)
          af[j] = *reinterpret_cast<const f32x4_t*>(base + a_off + j * 16 * 128 + coff);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int i = 0; i < TN; ++i)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[i][m], af[j][m], acc[i][j], 0, 0, 0);
      }
    }
  }

  // ---- epilogue.  Lane holds n = 16 i + 4 fq .. +3 of row m = 16 j + fr for every (i, j): stored
  // straight from the accumulators that is 16 rows x 32 B per instruction (measured ~2 TB/s).
  // Instead each wave transposes its (16 TM) x 64 sub-tile through its own 16-KiB slice of the
  // (now idle) staging LDS and writes whole 128-B (bf16) / 256-B (f32) row segments, 16 B per lane.
  static_assert(TN == 4, "epilogue assumes 64 output columns per wave");
  __syncthreads();  // every wave is done reading the last K tile
  constexpr int EP_STRIDE = OUT_F32 ? 16384 : TM * 16 * 128;   // one wave's transpose image (f32: passes of 64 rows x 256 B)
  static_assert(NW * EP_STRIDE <= NST * BUF_BYTES, "the epilogue images must fit the staging ring");
  char* ep = smem + w * EP_STRIDE;
  float4 bv[TN];
#pragma unroll
  for (int i = 0; i < TN; ++i) {
    const int n0 = min(nb + (wn * TN + i) * 16 + fq * 4, N - 4);
    bv[i] = *reinterpret_cast<const float4*>(bias + n0);
  }
  const int m_base = mb + wm * TM * 16, n_base = nb + wn * 64;
  if (!OUT_F32) {
    // LDS image [16 TM rows][128 B]; 8-byte slots XOR-swizzled in pairs so 16-B reads stay whole
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        float v0 = acc[i][j][0] + bv[i].x, v1 = acc[i][j][1] + bv[i].y;
        float v2 = acc[i][j][2] + bv[i].z, v3 = acc[i][j][3] + bv[i].w;
        if (EPI == EPI_BIAS_GELU) {
          v0 = gelu_erf_bf16out(v0); v1 = gelu_erf_bf16out(v1); v2 = gelu_erf_bf16out(v2); v3 = gelu_erf_bf16out(v3);
        } else if (EPI == EPI_BIAS_TANH) {
          v0 = tanhf(v0); v1 = tanhf(v1); v2 = tanhf(v2); v3 = tanhf(v3);
        }
        const int row = j * 16 + fr, slot = (i * 4 + fq) ^ ((row & 7) << 1);
        if (GEMM_ABLATE == 3) { asm volatile("" ::"v"(v0), "v"(v1), "v"(v2), "v"(v3)); continue; }
        *reinterpret_cast<uint2*>(ep + row * 128 + slot * 8) =
            make_uint2(pack_op2(v0, v1), pack_op2(v2, v3));
      }
#pragma unroll
    for (int p = 0; p < TM * 2; ++p) {
      const int row = p * 8 + (lane >> 3), c16 = lane & 7;
      const int m = m_base + row, n = n_base + c16 * 8;
      uint4 v = *reinterpret_cast<const uint4*>(ep + row * 128 + ((c16 ^ (row & 7)) * 16));
      if (GEMM_ABLATE == 3) continue;
      if (m < M && n < N) {
        if (EPI == EPI_BIAS_RES) {
          const uint4 r = *reinterpret_cast<const uint4*>(res + (size_t)m * ldres + n);
          const uint32_t* vi = reinterpret_cast<const uint32_t*>(&v);
          const uint32_t* ri = reinterpret_cast<const uint32_t*>(&r);
          uint32_t o[4];
#pragma unroll
          for (int k = 0; k < 4; ++k)
            o[k] = pack_op2(op2_lo(vi[k]) + op2_lo(ri[k]),
                               op2_hi(vi[k]) + op2_hi(ri[k]));
          v = make_uint4(o[0], o[1], o[2], o[3]);
        }
        *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(out) + (size_t)m * ldo + n) = v;
      }
    }
  } else {
    // f32: passes of 64 rows; LDS image [64 rows][256 B], 16-byte slots XOR-swizzled by row
#pragma unroll
    for (int ps = 0; ps < TM / 4; ++ps) {
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const int j = ps * 4 + jj;
          float v0 = acc[i][j][0] + bv[i].x, v1 = acc[i][j][1] + bv[i].y;
          float v2 = acc[i][j][2] + bv[i].z, v3 = acc[i][j][3] + bv[i].w;
          if (EPI == EPI_BIAS_GELU) {
            v0 = gelu_erf(v0); v1 = gelu_erf(v1); v2 = gelu_erf(v2); v3 = gelu_erf(v3);
          } else if (EPI == EPI_BIAS_TANH) {
            v0 = tanhf(v0); v1 = tanhf(v1); v2 = tanhf(v2); v3 = tanhf(v3);
          }
          const int row = jj * 16 + fr, slot = (i * 4 + fq) ^ (row & 15);
          if (GEMM_ABLATE == 3) { asm volatile("" ::"v"(v0), "v"(v1), "v"(v2), "v"(v3)); continue; }
          *reinterpret_cast<float4*>(ep + row * 256 + slot * 16) = make_float4(v0, v1, v2, v3);
        }
#pragma unroll
      for (int p = 0; p < 16; ++p) {
        const int row = p * 4 + (lane >> 4), c16 = lane & 15;
        const int m = m_base + ps * 64 + row, n = n_base + c16 * 4;
        float4 v = *reinterpret_cast<const float4*>(ep + row * 256 + ((c16 ^ (row & 15)) * 16));
        if (GEMM_ABLATE == 3) continue;
        if (m < M && n < N) {
          if (EPI == EPI_BIAS_LNRES) {
            const int rr = res_index ? res_index[m] : m;
            const float4 rv = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(resv) + (size_t)rr * ldres + n);
            const float2 sv = ln_stats[rr];
            const float4 gv = *reinterpret_cast<const float4*>(ln_gamma + n);
            const float4 be = *reinterpret_cast<const float4*>(ln_beta + n);
            v.x += ln_apply(rv.x, sv.x, sv.y, gv.x, be.x); v.y += ln_apply(rv.y, sv.x, sv.y, gv.y, be.y);
            v.z += ln_apply(rv.z, sv.x, sv.y, gv.z, be.z); v.w += ln_apply(rv.w, sv.x, sv.y, gv.w, be.w);
          }
          if (EPI == EPI_BIAS_RES16) {
            typedef __attribute__((ext_vector_type(4))) _Float16 f16x4_t;
            const int rr = res_index ? res_index[m] : m;
            if (res_tiles_n < 0) {   // the fp32 residual stream of CLIP_PRECISION_F16R (tiled by the fused GEMMs)
              const float4 rv = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(resv) +
                                                                  xres32_tiled_offset(rr, n, -res_tiles_n));
              v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
            } else {
              const size_t ro = res_tiles_n > 0 ? xres_tiled_offset(rr, n, res_tiles_n) : (size_t)rr * ldres + n;
              const f16x4_t rv = *reinterpret_cast<const f16x4_t*>(reinterpret_cast<const uint16_t*>(resv) + ro);
              v.x += (float)rv[0]; v.y += (float)rv[1]; v.z += (float)rv[2]; v.w += (float)rv[3];
            }
          }
          if (EPI == EPI_BIAS_RES) {
            if constexpr (F32IN) {
              const float4 rv = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(resv) + (size_t)m * ldres + n);
              v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
            } else {
              const uint2 rv = *reinterpret_cast<const uint2*>(res + (size_t)m * ldres + n);
              v.x += op2_lo(rv.x); v.y += op2_hi(rv.x);
              v.z += op2_lo(rv.y); v.w += op2_hi(rv.y);
            }
          }
          *reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + (size_t)m * ldo + n) = v;
        }
      }
    }
  }
}

template <int EPI, bool OUT_F32, int WAVES_M, int WAVES_N, int TM, int TN, bool F32IN = false, int NST = 2>
hipError_t launch_cfg(const GemmArgs& g, hipStream_t st) {
  constexpr int BM = WAVES_M * TM * 16, BN = WAVES_N * TN * 16;
  constexpr int LDS = NST * (BM + BN) * 128;
  const int tiles_m = (g.M + BM - 1) / BM, tiles_n = (g.N + BN - 1) / BN;
  auto kfn = gemm_bf16_kernel<EPI, OUT_F32, WAVES_M, WAVES_N, TM, TN, F32IN, NST>;
  static DeviceOnce once;  // per instantiation and device
  int unused = 0;
  hipError_t e0 = once.get([&](int, int*) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  }, &unused);
  if (e0 != hipSuccess) return e0;
  hipLaunchKernelGGL(kfn, dim3(tiles_m * tiles_n), dim3(64 * WAVES_M * WAVES_N), LDS, st, g.A, g.W,
                     g.bias, g.res, g.out, g.M, g.N, g.K, g.lda, g.ldw, g.ldo, g.ldres, tiles_m,
                     tiles_n, reinterpret_cast<const float2*>(g.ln_stats), g.ln_gamma, g.ln_beta,
                     g.res_index, g.res_tiles_n);
  return hipGetLastError();
}

// ---- at most 32 rows: the trimmed last layer of ONE batch of the reference's loop (8 sentences -> 8 rows) ----
// With 128 x 128 tiles such a call is N / 128 workgroups that each pull 128 weight rows AND 128 (clamped, repeated)
// activation rows per K tile through one CU's LDS-DMA intake (~55 GB/s): 31 us for the FFN-down rows of a batch of 8.
// Here a workgroup owns 16 rows x 16 columns: N / 16 workgroups, 4 KiB per K tile (one 1-KiB DMA piece per wave: waves
// 0-1 the activation rows, 2-3 the weight rows) through a 16-stage ring; wave 0 alone reads the fragments and issues
// the 2 MFMAs per K tile.  Same MFMA, same K order into one accumulator, same epilogue expressions as the tile kernel
// above: the SAME BITS (tests/test_gpu_kernels.py::test_rows16_gemm_same_bits_as_tile_kernel).
constexpr int R16_NST = 16, R16_D = R16_NST - 1, R16_STAGE = 4096;

#define R16_VMCNT(n) case n: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory"); break;

template <int EPI, bool OUT_F32>
__global__ __launch_bounds__(256) void gemm_rows16_kernel(
    const bf16_t* __restrict__ A, const bf16_t* __restrict__ W, const float* __restrict__ bias,
    const void* __restrict__ resv, void* __restrict__ out, int M, int N, int K, int lda, int ldw, int ldo, int ldres,
    const float2* __restrict__ ln_stats, const float* __restrict__ ln_gamma, const float* __restrict__ ln_beta,
    const int32_t* __restrict__ res_index, int res_tiles_n, const int32_t* __restrict__ a_index) {
  __shared__ __attribute__((aligned(16))) char smem[R16_NST * R16_STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_n = N >> 4;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
  const int mb = tm * 16, nb = tn * 16;
  // this wave's DMA piece: 8 rows x 128 B of the activation (w < 2) or of the weight (w >= 2)
  const int prow = (w & 1) * 8 + (lane >> 3);
  const int arow = min(mb + prow, M - 1);
  const bf16_t* src = (w < 2 ? A + (size_t)(a_index ? a_index[arow] : arow) * lda : W + (size_t)(nb + prow) * ldw) +
                      swz128(prow, lane & 7) * 8;
  auto stage = [&](int kt, int buf) {
    __builtin_amdgcn_global_load_lds(GLB_PTR(src + kt * 64), LDS_PTR(smem + buf * R16_STAGE + w * 1024), 16, 0, 0);
  };
  const int nt = K >> 6;
  for (int s = 0; s < R16_D; ++s)
    if (s < nt) stage(s, s);
  f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
  const int fr = lane & 15, fq = lane >> 4;
  const int lsw = (fr >> 1) & 7;
  for (int t = 0; t < nt; ++t) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // wave 0 holds K tile t - 1 in registers: its buffer is free
    const int rem = nt - 1 - t < R16_D - 1 ? nt - 1 - t : R16_D - 1;     // one DMA instruction per wave and K tile
    switch (rem) {
      R16_VMCNT(0) R16_VMCNT(1) R16_VMCNT(2) R16_VMCNT(3) R16_VMCNT(4) R16_VMCNT(5) R16_VMCNT(6) R16_VMCNT(7)
      R16_VMCNT(8) R16_VMCNT(9) R16_VMCNT(10) R16_VMCNT(11) R16_VMCNT(12) R16_VMCNT(13)
      default: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
    }
    __builtin_amdgcn_s_barrier();   // (not __syncthreads(): its fence would drain every DMA in flight)
    if (t + R16_D < nt) stage(t + R16_D, (t + R16_D) % R16_NST);
    if (w == 0) {
      const char* base = smem + (t % R16_NST) * R16_STAGE;
      opx8_t af[2], wf[2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int coff = ((ks * 4 + fq) ^ lsw) * 16;
        wf[ks] = *reinterpret_cast<const opx8_t*>(base + 2048 + fr * 128 + coff);
        af[ks] = *reinterpret_cast<const opx8_t*>(base + fr * 128 + coff);
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) acc = MFMA_16x16x32(wf[ks], af[ks], acc);
    }
  }
  if (w != 0) return;
  // ---- epilogue: the lane holds n = nb + 4 fq .. +3 of row m = mb + fr; the tile kernel's expressions, in its order
  const int m = mb + fr, n = nb + fq * 4;
  if (m >= M) return;
  const float4 bv = *reinterpret_cast<const float4*>(bias + n);
  float v0 = acc[0] + bv.x, v1 = acc[1] + bv.y, v2 = acc[2] + bv.z, v3 = acc[3] + bv.w;
  if (EPI == EPI_BIAS_GELU) {
    if (OUT_F32) { v0 = gelu_erf(v0); v1 = gelu_erf(v1); v2 = gelu_erf(v2); v3 = gelu_erf(v3); }
    else { v0 = gelu_erf_bf16out(v0); v1 = gelu_erf_bf16out(v1); v2 = gelu_erf_bf16out(v2); v3 = gelu_erf_bf16out(v3); }
  } else if (EPI == EPI_BIAS_TANH) {
    v0 = tanhf(v0); v1 = tanhf(v1); v2 = tanhf(v2); v3 = tanhf(v3);
  }
  if (!OUT_F32) {
    uint2 o = make_uint2(pack_op2(v0, v1), pack_op2(v2, v3));
    if (EPI == EPI_BIAS_RES) {
      const uint2 r = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(resv) + (size_t)m * ldres + n);
      o = make_uint2(pack_op2(op2_lo(o.x) + op2_lo(r.x), op2_hi(o.x) + op2_hi(r.x)),
                     pack_op2(op2_lo(o.y) + op2_lo(r.y), op2_hi(o.y) + op2_hi(r.y)));
    }
    *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(out) + (size_t)m * ldo + n) = o;
    return;
  }
  float4 v = make_float4(v0, v1, v2, v3);
  if (EPI == EPI_BIAS_LNRES) {
    const int rr = res_index ? res_index[m] : m;
    const float4 rv = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(resv) + (size_t)rr * ldres + n);
    const float2 sv = ln_stats[rr];
    const float4 gv = *reinterpret_cast<const float4*>(ln_gamma + n);
    const float4 be = *reinterpret_cast<const float4*>(ln_beta + n);
    v.x += ln_apply(rv.x, sv.x, sv.y, gv.x, be.x); v.y += ln_apply(rv.y, sv.x, sv.y, gv.y, be.y);
    v.z += ln_apply(rv.z, sv.x, sv.y, gv.z, be.z); v.w += ln_apply(rv.w, sv.x, sv.y, gv.w, be.w);
  }
  if (EPI == EPI_BIAS_RES16) {
    typedef __attribute__((ext_vector_type(4))) _Float16 f16x4_t;
    const int rr = res_index ? res_index[m] : m;
    if (res_tiles_n < 0) {
      const float4 rv = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(resv) +
                                                          xres32_tiled_offset(rr, n, -res_tiles_n));
      v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
    } else {
      const size_t ro = res_tiles_n > 0 ? xres_tiled_offset(rr, n, res_tiles_n) : (size_t)rr * ldres + n;
      const f16x4_t rv = *reinterpret_cast<const f16x4_t*>(reinterpret_cast<const uint16_t*>(resv) + ro);
      v.x += (float)rv[0]; v.y += (float)rv[1]; v.z += (float)rv[2]; v.w += (float)rv[3];
    }
  }
  if (EPI == EPI_BIAS_RES) {
    const uint2 rv = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(resv) + (size_t)m * ldres + n);
    v.x += op2_lo(rv.x); v.y += op2_hi(rv.x);
    v.z += op2_lo(rv.y); v.w += op2_hi(rv.y);
  }
  *reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + (size_t)m * ldo + n) = v;
}
#undef R16_VMCNT

template <int EPI, bool OUT_F32>
hipError_t launch_rows16(const GemmArgs& g, hipStream_t st) {
  hipLaunchKernelGGL((gemm_rows16_kernel<EPI, OUT_F32>), dim3(((g.M + 15) / 16) * (g.N / 16)), dim3(256), 0, st,
                     static_cast<const bf16_t*>(g.A), static_cast<const bf16_t*>(g.W), g.bias, g.res, g.out, g.M, g.N,
                     g.K, g.lda, g.ldw, g.ldo, g.ldres, reinterpret_cast<const float2*>(g.ln_stats), g.ln_gamma,
                     g.ln_beta, g.res_index, g.res_tiles_n, g.a_index);
  return hipGetLastError();
}

template <int EPI, bool OUT_F32>
hipError_t launch_one(const GemmArgs& g, hipStream_t st) {
  if (g.tile == 0 && g.M <= 32 && g.lda % 8 == 0 && g.ldw % 8 == 0) return launch_rows16<EPI, OUT_F32>(g, st);   // a batch's rows
  if (g.a_index) return hipErrorInvalidValue;
  // big tile once it fills the chip about once over; the small tile otherwise
  const long big_tiles = (long)((g.M + 255) / 256) * ((g.N + 255) / 256);
  if (g.tile == 256 || (g.tile == 0 && big_tiles >= 192 && g.N % 256 == 0))
    return launch_cfg<EPI, OUT_F32, 2, 4, 8, 4>(g, st);
  // 16-bit output (QKV, FFN up of a small call): the same 128 x 128 tile on 8 waves (4 x 2, 32 x 64 per wave) — a wave's
  // LDS-DMA stream lands at ~25 GB/s, so the tile's intake grows with the number of waves that issue it
  // and through a 4-deep ring when the grid leaves one workgroup per CU anyway (a batch of 8: 0.895 -> 0.868 ms); bigger
  // grids keep the double buffer, whose 64 KiB let two workgroups share a CU (batches of 32: 1.59 -> 1.50 ms)
  static DeviceOnce cus;
  int n_cu = 0;
  const hipError_t e0 = cus.get([](int dev, int* val) {
    return hipDeviceGetAttribute(val, hipDeviceAttributeMultiprocessorCount, dev);
  }, &n_cu);
  if (e0 != hipSuccess) return e0;
  const bool deep = (long)((g.M + 127) / 128) * ((g.N + 127) / 128) <= n_cu;
  if constexpr (!OUT_F32)
    return deep ? launch_cfg<EPI, OUT_F32, 4, 2, 2, 4, false, 4>(g, st) : launch_cfg<EPI, OUT_F32, 4, 2, 2, 4>(g, st);
  return deep ? launch_cfg<EPI, OUT_F32, 2, 2, 4, 4, false, 4>(g, st) : launch_cfg<EPI, OUT_F32, 2, 2, 4, 4>(g, st);
}

}  // namespace

#if !CLIP_OPERAND_F16
bool gemm_persist_ok(const GemmArgs& g) {  // shapes the ping-pong persistent kernel (gemm8.hip) takes
  return !g.f32_in && g.M % 256 == 0 && g.N % 256 == 0 && g.K % 128 == 0 && g.M > 0 && g.N > 0 && g.K > 0;
}

const char* gemm_check(const GemmArgs& g) {
  if (g.M <= 0 || g.N <= 0 || g.K <= 0) return "gemm: empty problem";
  if (g.K % (g.f32_in ? 32 : 64)) return "gemm: K must be a multiple of 64 (32 in fp32 mode)";
  if (g.f32_in && !g.out_f32) return "gemm: fp32 operands need an fp32 output";
  if (g.N % 16) return "gemm: N must be a multiple of 16";
  if (g.lda % (g.f32_in ? 4 : 8) || g.ldw % (g.f32_in ? 4 : 8) || g.ldo % 4) return "gemm: leading dimensions must keep 16-byte rows";
  if (g.epi == EPI_BIAS_RES && (!g.res || g.ldres % 4)) return "gemm: residual missing/misaligned";
  if (g.epi == EPI_BIAS_LNRES && (!g.res || g.ldres % 4 || !g.ln_stats || !g.ln_gamma || !g.ln_beta || !g.out_f32 || g.f32_in))
    return "gemm: LayerNorm-residual epilogue needs fp32 pre-LN rows, statistics, gamma, beta and an fp32 output";
  if (g.epi == EPI_BIAS_RES16 && (!g.res || g.ldres % 4 || !g.out_f32 || g.f32_in)) return "gemm: fp16-residual epilogue needs fp16 rows and an fp32 output";
  if (g.epi == EPI_BIAS_LNOUT && (g.tile != 2 || g.out_f32 || g.f32_in || !gemm_persist_ok(g) || !g.ln_gamma || !g.ln_beta))
    return "gemm: the fused-LayerNorm epilogue runs on the ping-pong kernel only (tile 2, bf16 out, M and N multiples of 256)";
  if (g.epi == EPI_BIAS_LNOUT && (g.ldres % 8 || g.ldo % 8)) return "gemm: the fused-LayerNorm epilogue needs 16-byte rows (ldo, ldres multiples of 8)";
  if (g.res_index && g.epi != EPI_BIAS_LNRES && g.epi != EPI_BIAS_RES16) return "gemm: res_index only with the residual-row epilogues";
  if (!g.A || !g.W || !g.bias || !g.out) return "gemm: null operand";
  if (g.tile != 0 && g.tile != 128 && g.tile != 256 && g.tile != 2) return "gemm: tile must be 0 (auto), 2 (ping-pong persistent), 128 or 256";
  if (g.tile == 2 && !gemm_persist_ok(g)) return "gemm: persistent kernel needs M, N multiples of 256 and K of 128";
  return nullptr;
}

#endif

hipError_t KSET::launch_gemm(const GemmArgs& g, hipStream_t st) {
#if !CLIP_OPERAND_F16
  if (g.f32_in) {  // fp32 check mode: 128x128 tile of the same kernel, fp32 operands
    if (!g.out_f32) return hipErrorInvalidValue;
    switch (g.epi) {
      case EPI_BIAS: return launch_cfg<EPI_BIAS, true, 2, 2, 4, 4, true>(g, st);
      case EPI_BIAS_GELU: return launch_cfg<EPI_BIAS_GELU, true, 2, 2, 4, 4, true>(g, st);
      case EPI_BIAS_RES: return launch_cfg<EPI_BIAS_RES, true, 2, 2, 4, 4, true>(g, st);
      case EPI_BIAS_TANH: return launch_cfg<EPI_BIAS_TANH, true, 2, 2, 4, 4, true>(g, st);
      default: return hipErrorInvalidValue;
    }
  }
#else
  if (g.f32_in) return hipErrorInvalidValue;  // the fp32 check mode lives in the bf16 set
#endif
  if (g.res_index && g.tile == 2) return hipErrorInvalidValue;
  if ((g.a_tiled || g.out_tiled || g.w_tiled) && g.tile != 2) return hipErrorInvalidValue;  // block layouts: ping-pong kernel only
  if (g.tile == 2 || (g.tile == 0 && gemm_persist_ok(g) && (long)(g.M / 256) * (g.N / 256) >= 256 &&
                      !((g.epi == EPI_BIAS_TANH || g.epi == EPI_BIAS_GELU) && g.out_f32) && !g.res_index &&
                      g.epi != EPI_BIAS_RES16))
    return launch_gemm8(g, st);
  switch (g.epi * 2 + (g.out_f32 ? 1 : 0)) {
    case EPI_BIAS * 2 + 0: return launch_one<EPI_BIAS, false>(g, st);
    case EPI_BIAS * 2 + 1: return launch_one<EPI_BIAS, true>(g, st);
    case EPI_BIAS_GELU * 2 + 0: return launch_one<EPI_BIAS_GELU, false>(g, st);
    case EPI_BIAS_GELU * 2 + 1: return launch_one<EPI_BIAS_GELU, true>(g, st);   // LM-head transform (exact erf form)
    case EPI_BIAS_RES * 2 + 1: return launch_one<EPI_BIAS_RES, true>(g, st);
    case EPI_BIAS_RES * 2 + 0: return launch_one<EPI_BIAS_RES, false>(g, st);
    case EPI_BIAS_TANH * 2 + 0: return launch_one<EPI_BIAS_TANH, false>(g, st);
    case EPI_BIAS_TANH * 2 + 1: return launch_one<EPI_BIAS_TANH, true>(g, st);
    case EPI_BIAS_LNRES * 2 + 1: return launch_one<EPI_BIAS_LNRES, true>(g, st);
    case EPI_BIAS_RES16 * 2 + 1: return launch_one<EPI_BIAS_RES16, true>(g, st);
    default: return hipErrorInvalidValue;
  }
}
