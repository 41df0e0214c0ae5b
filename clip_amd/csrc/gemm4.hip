// Small-M form of the fused-LayerNorm GEMM (EPI_BIAS_LNOUT; gemm8.hip holds the ping-pong form):
//   v = A W^T + bias + residual;   x, xres = LayerNorm(v) over the N columns of each row
// (BertSelfOutput / BertOutput, transformers modeling_bert.py:289-293, 347-351) for calls with so few rows that
// the 256 x 256 tiles of the ping-pong kernel leave most of the chip idle — the reference's own loop feeds the model 8
// sentences (~900 tokens) at a time (apply_bert_unlabeled.py:92-110): 12 workgroups, each walking its whole K alone.
// Here a workgroup (8 waves: 4 column waves x 2 row halves) owns a 64-row x 256-column tile: 4x the workgroups, a
// quarter of the work per K tile, a 3-stage LDS-DMA ring, one barrier per K tile.
//
// SAME BITS as gemm8_kernel<EPI_BIAS_LNOUT>, by construction — a row's logits must not depend on the batch it is scored
// in (tests: test_packed_path_equals_padded_path, test_small_m_fused_gemm_same_bits):
//   * the same MFMA (16x16x32), the same fragment-to-lane map, K walked in the same order into one accumulator;
//   * bias, then residual, added in the same order; the row sums in the same order: a lane's 16 values by packed
//     adds, the 4 lanes of a row by v_permlane16/32_swap, the 4 column waves as (p0 + p1) + (p2 + p3), the column
//     tiles of the row in index order; the same mean / variance / normalise expressions.
// The N / 256 workgroups of a 64-row panel have adjacent ids, are all resident (the launcher only takes grids of at
// most one workgroup per CU) and exchange row sums exactly as the ping-pong kernel does (epoch flags, bounded spin,
// sticky error word).
#include <cstdlib>

#include "common.h"
#include "kernels.h"

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
constexpr int PM = 64, PN = 256, PK = 64;
constexpr int STAGE = PM * 128 + PN * 128;            // 8 KiB of A rows + 32 KiB of W rows (128-byte rows, swizzled)
constexpr int NSTAGE = 3;
constexpr int LDS4 = NSTAGE * STAGE + 4 * 512 + 2 * 512;   // + row partials per wave + this tile's sums + (mean, rstd)

__device__ __forceinline__ float xor16_sum4(float v) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor32_sum4(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// res_mode: 0 residual = fp16 rows (row stride ldres; may be the output buffer itself: every lane reads exactly the
//             elements it rewrites), 1 = fp16 tiled (xres_tiled_offset), 2 = fp32 tiled (xres32_tiled_offset)
// out_mode: 0 no residual copy out (fp16 operand set: x is the copy), 1 = fp16 tiled, 2 = fp32 tiled
__global__ __launch_bounds__(512) void gemm4_lnout_kernel(
    const bf16_t* __restrict__ A, const bf16_t* __restrict__ W, const float* __restrict__ bias,
    const void* resv, void* out, int K, int lda, int ldw, int ldo, int ldres, int tiles_n, int a_tiled,
    const float* __restrict__ ln_gamma, const float* __restrict__ ln_beta, LnOutArgs lo, int res_mode, int out_mode) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  // 8 waves: wc = column wave (columns 64 wc .. 64 wc + 63 of the tile), wr = row half (16-row blocks 2 wr, 2 wr + 1).
  // Rows are independent in everything below, so two row halves give the bits of one wave doing all four blocks; the
  // two waves of a SIMD are then one reading LDS while the other issues MFMAs.
  const int w8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = w8 & 3, wr = w8 >> 2;
  const int panel = blockIdx.x / tiles_n, tn = blockIdx.x - panel * tiles_n;
  const int row0 = panel * PM;                                   // first row of the panel
  const int fr = lane & 15, fq = lane >> 4;

  // ---- staging: lane-linear 1-KiB LDS-DMA images of 8 rows; the swizzle rides on the source chunk
  const int prow = lane >> 3, pslot = lane & 7;
  auto dma16 = [&](const void* g, const char* lds_dst) {
    const uint32_t m = (uint32_t)(uintptr_t)LDS_PTR(lds_dst);
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"   // "m0 is reserved": nothing else in this kernel uses it
    asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(m) : "memory", "m0");
#pragma clang diagnostic pop
  };
  const bf16_t* a_panel = a_tiled ? A + ((size_t)(row0 >> 7) * (K >> 6) << 13) + ((size_t)(row0 & 127) << 6)
                                  : A + (size_t)row0 * lda;
  const int ld_a = a_tiled ? 64 : lda, a_kstep = a_tiled ? 8192 : PK;
  const bf16_t* w_tile = W + (size_t)tn * PN * ldw;
  auto stage = [&](int kt, int buf) {
    char* base = smem + buf * STAGE;
    {
      const int g = w8, row = 8 * g + prow;                  // A: 8 row groups, one per wave
      dma16(a_panel + (size_t)kt * a_kstep + (size_t)row * ld_a + swz128(row, pslot) * 8, base + g * 1024);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {                           // W: 32 row groups, four per wave
      const int g = w8 + 8 * j, row = 8 * g + prow;
      dma16(w_tile + (size_t)row * ldw + kt * PK + swz128(row, pslot) * 8, base + PM * 128 + g * 1024);
    }
  };

  const int nt = K / PK;   // >= 2
  stage(0, 0);
  stage(1, 1);

  f32x4_t acc[2][2][2];    // [nh][mm][i]: lane holds n = 4 fq .. +3 of 16-col block (nh, i), row fr of 16-row block 2 wr + mm
#pragma unroll
  for (int nh = 0; nh < 2; ++nh)
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[nh][m][i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const int lsw = (fr >> 1) & 7;
  const int c0 = (fq ^ lsw) * 16, c1 = ((4 + fq) ^ lsw) * 16;
  const int a_rd = (wr * 32 + fr) * 128, b_rd = PM * 128 + (wc * 64 + fr) * 128;
  for (int t = 0; t < nt; ++t) {
    // this wave's pieces of K tile t have landed (5 DMA instructions per tile and wave, in order) ...
    if (t + 1 < nt) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // ... every wave's have, and K tile t - 1 has been read by all: its buffer is free
    if (t + 2 < nt) stage(t + 2, (t + 2) % NSTAGE);
    const char* buf = smem + (t % NSTAGE) * STAGE;
    opx8_t af[2][2], bf[2][2][2];
#pragma unroll
    for (int nh = 0; nh < 2; ++nh)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        bf[nh][i][0] = *reinterpret_cast<const opx8_t*>(buf + b_rd + nh * 4096 + i * 2048 + c0);
        bf[nh][i][1] = *reinterpret_cast<const opx8_t*>(buf + b_rd + nh * 4096 + i * 2048 + c1);
      }
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      af[m][0] = *reinterpret_cast<const opx8_t*>(buf + a_rd + m * 2048 + c0);
      af[m][1] = *reinterpret_cast<const opx8_t*>(buf + a_rd + m * 2048 + c1);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          acc[0][m][i] = MFMA_16x16x32(bf[0][i][ks], af[m][ks], acc[0][m][i]);
          acc[1][m][i] = MFMA_16x16x32(bf[1][i][ks], af[m][ks], acc[1][m][i]);
        }
  }

  // ---- epilogue (the steps of gemm8_kernel<EPI_BIAS_LNOUT>, one 64-row block)
  const int n_base = tn * PN + wc * 64;
  const int r256 = row0 & 255, mh = r256 >> 7, wr256 = (r256 >> 6) & 1;
  const int tm = row0 >> 8;
  const size_t wave_tile = (((size_t)tm * tiles_n + tn) * 8 + wr256 * 4 + wc) * 8192 + (size_t)mh * 4 * 1024;   // + m * 1024
  f32x4_t bv[2][2];
#pragma unroll
  for (int nh = 0; nh < 2; ++nh)
#pragma unroll
    for (int i = 0; i < 2; ++i) bv[nh][i] = *reinterpret_cast<const f32x4_t*>(bias + n_base + nh * 32 + i * 16 + fq * 4);
  // (1) v = (acc + bias) + residual
#pragma unroll
  for (int mm = 0; mm < 2; ++mm)
#pragma unroll
    for (int nh = 0; nh < 2; ++nh) {
      const int m = wr * 2 + mm;
      float r[2][4];
      if (res_mode == 2) {
        const float* rt = reinterpret_cast<const float*>(resv) + wave_tile + m * 1024 + nh * 512 + lane * 4;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const f32x4_t v = *reinterpret_cast<const f32x4_t*>(rt + i * 256);
#pragma unroll
          for (int e = 0; e < 4; ++e) r[i][e] = v[e];
        }
      } else if (res_mode == 1) {
        const f16x8_t v = *reinterpret_cast<const f16x8_t*>(reinterpret_cast<const uint16_t*>(resv) + wave_tile + m * 1024 +
                                                            nh * 512 + lane * 8);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e) r[i][e] = (float)v[i * 4 + e];
      } else {
        const uint16_t* rr = reinterpret_cast<const uint16_t*>(resv) + (size_t)(row0 + m * 16 + fr) * ldres + n_base + nh * 32 + fq * 4;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const f16x4_t v = *reinterpret_cast<const f16x4_t*>(rr + i * 16);
#pragma unroll
          for (int e = 0; e < 4; ++e) r[i][e] = (float)v[e];
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[nh][mm][i][e] = (acc[nh][mm][i][e] + bv[nh][i][e]) + r[i][e];
    }
  // (2) per-row (sum, sum of squares) over this wave's 64 columns
  float2* part_own = reinterpret_cast<float2*>(smem + NSTAGE * STAGE + wc * 512);       // [4 m][16 rows]
  float2* tot_lds = reinterpret_cast<float2*>(smem + NSTAGE * STAGE + 4 * 512);         // [64]
  float2* stat_lds = tot_lds + 64;                                                      // [64] (mean, rstd)
#pragma unroll
  for (int mm = 0; mm < 2; ++mm) {
    const int m = wr * 2 + mm;
    f32x2_t s2 = {0.f, 0.f}, q2 = {0.f, 0.f};
#pragma unroll
    for (int nh = 0; nh < 2; ++nh)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const f32x2_t lo2 = {acc[nh][mm][i][0], acc[nh][mm][i][1]};
        const f32x2_t hi2 = {acc[nh][mm][i][2], acc[nh][mm][i][3]};
        s2 += lo2; s2 += hi2;
        q2 = __builtin_elementwise_fma(lo2, lo2, q2);
        q2 = __builtin_elementwise_fma(hi2, hi2, q2);
      }
    float sm = s2[0] + s2[1], sq = q2[0] + q2[1];
    sm = xor16_sum4(sm); sq = xor16_sum4(sq);
    sm = xor32_sum4(sm); sq = xor32_sum4(sq);
    if (fq == 0) part_own[m * 16 + fr] = make_float2(sm, sq);
  }
  __syncthreads();
  // (3) the tile's 256-column partial per row, published with agent-scope stores, then the flag (ordering as gemm8.hip)
  const int flag_id = panel * tiles_n + tn;
  const size_t part_base = ((size_t)tm * tiles_n) * 256 + r256;     // + k * 256 + row
  if (tid < 64) {
    const float2* pr = reinterpret_cast<const float2*>(smem + NSTAGE * STAGE) + tid;   // row tid = m * 16 + fr
    const float2 p0 = pr[0], p1 = pr[64], p2 = pr[128], p3 = pr[192];
    const float2 P = make_float2((p0.x + p1.x) + (p2.x + p3.x), (p0.y + p1.y) + (p2.y + p3.y));
    tot_lds[tid] = P;
    unsigned long long bits;
    __builtin_memcpy(&bits, &P, 8);
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(lo.part) + part_base + (size_t)tn * 256 + tid, bits,
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0 && !(lo.test_drop && flag_id == 1))
    __hip_atomic_store(lo.flags + flag_id, lo.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // (4) wait for the other column tiles of this panel (lane 0 of wave k polls tile k), bounded and sticky
  if (lane == 0) {
    for (int k = w8; k < tiles_n; k += 8) {
      if (k == tn) continue;
      const uint32_t* f = lo.flags + panel * tiles_n + k;
      int spins = 0;
      bool dead = (__hip_atomic_load(lo.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 2) != 0;
      while (!dead && __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != lo.epoch) {
        __builtin_amdgcn_s_sleep(4);
        if (++spins > (1 << 20)) { atomicOr(lo.err, 2); break; }
        if ((spins & 255) == 0) dead = (__hip_atomic_load(lo.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 2) != 0;
      }
    }
  }
  __syncthreads();
  if (tid < 64) {
    float2 T = make_float2(0.f, 0.f);
    for (int k = 0; k < tiles_n; ++k) {  // fixed order: every tile of the panel gets the same bits
      float2 P;
      if (k == tn) {
        P = tot_lds[tid];
      } else {
        const unsigned long long bits = __hip_atomic_load(
            reinterpret_cast<const unsigned long long*>(lo.part) + part_base + (size_t)k * 256 + tid,
            __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_memcpy(&P, &bits, 8);
      }
      T.x += P.x; T.y += P.y;
    }
    const float inv_n = 1.0f / (float)(tiles_n * PN);
    const float mean = T.x * inv_n;
    const float var = fmaxf(fmaf(T.y, inv_n, -mean * mean), 0.f);
    stat_lds[tid] = make_float2(mean, 1.0f / sqrtf(var + lo.eps));
  }
  __syncthreads();
  // (5) normalise; x as 8-byte pieces (4 columns per lane), the residual copy in its tiled form
#pragma unroll
  for (int mm = 0; mm < 2; ++mm) {
    const int m = wr * 2 + mm;
    const float2 st = stat_lds[m * 16 + fr];
    uint16_t* xrow = reinterpret_cast<uint16_t*>(out) + (size_t)(row0 + m * 16 + fr) * ldo + n_base + fq * 4;
#pragma unroll
    for (int nh = 0; nh < 2; ++nh) {
      f16x8_t hv;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const f32x4_t gv = *reinterpret_cast<const f32x4_t*>(ln_gamma + n_base + nh * 32 + i * 16 + fq * 4);
        const f32x4_t be = *reinterpret_cast<const f32x4_t*>(ln_beta + n_base + nh * 32 + i * 16 + fq * 4);
        float y[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          y[e] = ln_apply(acc[nh][mm][i][e], st.x, st.y, gv[e], be[e]);
          hv[i * 4 + e] = (_Float16)y[e];
        }
        *reinterpret_cast<uint2*>(xrow + nh * 32 + i * 16) = make_uint2(pack_op2(y[0], y[1]), pack_op2(y[2], y[3]));
        if (out_mode == 2)
          *reinterpret_cast<f32x4_t*>(reinterpret_cast<float*>(lo.xres_out) + wave_tile + m * 1024 + nh * 512 + i * 256 + lane * 4) =
              (f32x4_t){y[0], y[1], y[2], y[3]};
      }
      if (out_mode == 1)
        *reinterpret_cast<u32x4_t*>(lo.xres_out + wave_tile + m * 1024 + nh * 512 + lane * 8) = *reinterpret_cast<const u32x4_t*>(&hv);
    }
  }
}


// ---- 64 x 64 tiles: the same GEMM for calls of at most a few hundred rows more (one batch of the reference's loop) ----
// A workgroup (4 waves, one 16-row block each) owns 64 rows x ONE 64-column chunk — the unit whose row sums the other two
// kernels form inside a wave — so a batch of 8 sentences spreads over (M / 64) * (N / 64) = 192 workgroups instead of 48,
// each pulling 16 KiB per K tile through an 8-stage LDS-DMA ring (7 tiles in flight).  The sums of the four chunks of a
// 256-column group, which the other kernels add across waves in LDS, are added here after the exchange, in the same
// order: ((c0 + c1) + (c2 + c3)), then the groups in index order — the SAME BITS again.  The exchange itself carries the
// launch's epoch inside every 8-byte word instead of in a separate flag (LnOutArgs::part_tag: 16 bytes per row and chunk).
// blockIdx -> (panel, chunk): workgroup b runs on XCD b % 8; an XCD takes a quarter of the panels x half of the chunks,
// so its L2 holds (P / 4) A panels and (C / 2) weight chunks instead of a bit of everything.
constexpr int QM = 64, QN = 64;
constexpr int QSTAGE = QM * 128 + QN * 128;           // 16 KiB
constexpr int QNST = 8, QD = QNST - 1;
constexpr int LDSQ = QNST * QSTAGE + 2 * 512;         // + (mean, rstd) per row, + spare

#define VMCNT_CASE(n) case n: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * n) : "memory"); break;

__global__ __launch_bounds__(256) void gemm4n_lnout_kernel(
    const bf16_t* __restrict__ A, const bf16_t* __restrict__ W, const float* __restrict__ bias,
    const void* resv, void* out, int K, int lda, int ldw, int ldo, int ldres, int tiles_n, int a_tiled,
    const float* __restrict__ ln_gamma, const float* __restrict__ ln_beta, LnOutArgs lo, int res_mode, int out_mode,
    int panels) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);        // 16-row block of the panel
  const int chunks_n = tiles_n * 4;
  int panel, chunk;
  {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int pq = panels >> 2, ch = chunks_n >> 1;              // panels per XCD row, chunks per XCD column
    panel = (xcd >> 1) * pq + j / ch;
    chunk = (xcd & 1) * ch + j % ch;
  }
  const int tn = chunk >> 2, wc = chunk & 3;
  const int row0 = panel * QM;
  const int fr = lane & 15, fq = lane >> 4;

  const int prow = lane >> 3, pslot = lane & 7;
  auto dma16 = [&](const void* g, const char* lds_dst) {
    const uint32_t m = (uint32_t)(uintptr_t)LDS_PTR(lds_dst);
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
    asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(m) : "memory", "m0");
#pragma clang diagnostic pop
  };
  const bf16_t* a_panel = a_tiled ? A + ((size_t)(row0 >> 7) * (K >> 6) << 13) + ((size_t)(row0 & 127) << 6)
                                  : A + (size_t)row0 * lda;
  const int ld_a = a_tiled ? 64 : lda, a_kstep = a_tiled ? 8192 : PK;
  const bf16_t* w_chunk = W + (size_t)chunk * QN * ldw;
  auto stage = [&](int kt, int buf) {     // 4 DMA instructions per wave: 2 row groups of A, 2 of W
    char* base = smem + buf * QSTAGE;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int g = w + 4 * j, row = 8 * g + prow;
      dma16(a_panel + (size_t)kt * a_kstep + (size_t)row * ld_a + swz128(row, pslot) * 8, base + g * 1024);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int g = w + 4 * j, row = 8 * g + prow;
      dma16(w_chunk + (size_t)row * ldw + kt * PK + swz128(row, pslot) * 8, base + QM * 128 + g * 1024);
    }
  };

  const int nt = K / PK;   // even (K % 128 == 0)
  for (int s = 0; s < QD; ++s)
    if (s < nt) stage(s, s);

  f32x4_t acc[2][2];       // [nh][i]: lane holds n = 4 fq .. +3 of 16-col block (nh, i), row fr of this wave's 16-row block
#pragma unroll
  for (int nh = 0; nh < 2; ++nh)
#pragma unroll
    for (int i = 0; i < 2; ++i) acc[nh][i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const int lsw = (fr >> 1) & 7;
  const int c0 = (fq ^ lsw) * 16, c1 = ((4 + fq) ^ lsw) * 16;
  const int a_rd = (w * 16 + fr) * 128, b_rd = QM * 128 + fr * 128;
  // One wave per SIMD: the LDS reads of K tile t + 1 are issued before the MFMAs of tile t (two register sets), so the
  // two overlap inside the wave.  Step t: the reads of tile t - 1 are complete (they ran under the MFMAs of t - 2),
  // tile t has landed, barrier: every wave holds tile t - 1 in registers, so its buffer takes tile t + 7.
  auto top = [&](int t) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int rem = nt - 1 - t < QD - 1 ? nt - 1 - t : QD - 1;   // tiles issued after tile t: 4 DMA instructions each
    switch (rem) {
      VMCNT_CASE(0) VMCNT_CASE(1) VMCNT_CASE(2) VMCNT_CASE(3) VMCNT_CASE(4) VMCNT_CASE(5)
      default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (QD - 1)) : "memory"); break;
    }
    __builtin_amdgcn_s_barrier();
    if (t + QD < nt) stage(t + QD, (t + QD) % QNST);
  };
  auto rd = [&](int t, opx8_t (&af)[2], opx8_t (&bf)[2][2][2]) {
    const char* buf = smem + (t % QNST) * QSTAGE;
#pragma unroll
    for (int nh = 0; nh < 2; ++nh)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        bf[nh][i][0] = *reinterpret_cast<const opx8_t*>(buf + b_rd + nh * 4096 + i * 2048 + c0);
        bf[nh][i][1] = *reinterpret_cast<const opx8_t*>(buf + b_rd + nh * 4096 + i * 2048 + c1);
      }
    af[0] = *reinterpret_cast<const opx8_t*>(buf + a_rd + c0);
    af[1] = *reinterpret_cast<const opx8_t*>(buf + a_rd + c1);
  };
  auto mm = [&](const opx8_t (&af)[2], const opx8_t (&bf)[2][2][2]) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int nh = 0; nh < 2; ++nh)
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[nh][i] = MFMA_16x16x32(bf[nh][i][ks], af[ks], acc[nh][i]);
  };
  opx8_t afA[2], bfA[2][2][2], afB[2], bfB[2][2][2];
  top(0);
  rd(0, afA, bfA);
  for (int t = 1; t < nt; t += 2) {
    top(t);
    rd(t, afB, bfB);
    mm(afA, bfA);               // tile t - 1
    if (t + 1 < nt) {
      top(t + 1);
      rd(t + 1, afA, bfA);
    }
    mm(afB, bfB);               // tile t
  }

  // ---- epilogue: the steps of gemm4_lnout_kernel for 16-row block m = w of column wave wc of tile tn
  const int m = w;
  const int n_base = chunk * QN;
  const int r256 = row0 & 255, mh = r256 >> 7, wr256 = (r256 >> 6) & 1;
  const int tm = row0 >> 8;
  const size_t wave_tile = (((size_t)tm * tiles_n + tn) * 8 + wr256 * 4 + wc) * 8192 + (size_t)mh * 4 * 1024 + m * 1024;
  f32x4_t bv[2][2];
#pragma unroll
  for (int nh = 0; nh < 2; ++nh)
#pragma unroll
    for (int i = 0; i < 2; ++i) bv[nh][i] = *reinterpret_cast<const f32x4_t*>(bias + n_base + nh * 32 + i * 16 + fq * 4);
  // (1) v = (acc + bias) + residual
#pragma unroll
  for (int nh = 0; nh < 2; ++nh) {
    float r[2][4];
    if (res_mode == 2) {
      const float* rt = reinterpret_cast<const float*>(resv) + wave_tile + nh * 512 + lane * 4;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const f32x4_t v = *reinterpret_cast<const f32x4_t*>(rt + i * 256);
#pragma unroll
        for (int e = 0; e < 4; ++e) r[i][e] = v[e];
      }
    } else if (res_mode == 1) {
      const f16x8_t v = *reinterpret_cast<const f16x8_t*>(reinterpret_cast<const uint16_t*>(resv) + wave_tile + nh * 512 + lane * 8);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) r[i][e] = (float)v[i * 4 + e];
    } else {
      const uint16_t* rr = reinterpret_cast<const uint16_t*>(resv) + (size_t)(row0 + m * 16 + fr) * ldres + n_base + nh * 32 + fq * 4;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const f16x4_t v = *reinterpret_cast<const f16x4_t*>(rr + i * 16);
#pragma unroll
        for (int e = 0; e < 4; ++e) r[i][e] = (float)v[e];
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[nh][i][e] = (acc[nh][i][e] + bv[nh][i][e]) + r[i][e];
  }
  // (2) per-row (sum, sum of squares) over the chunk's 64 columns: what a column wave of the other kernels forms
  f32x2_t s2 = {0.f, 0.f}, q2 = {0.f, 0.f};
#pragma unroll
  for (int nh = 0; nh < 2; ++nh)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const f32x2_t lo2 = {acc[nh][i][0], acc[nh][i][1]};
      const f32x2_t hi2 = {acc[nh][i][2], acc[nh][i][3]};
      s2 += lo2; s2 += hi2;
      q2 = __builtin_elementwise_fma(lo2, lo2, q2);
      q2 = __builtin_elementwise_fma(hi2, hi2, q2);
    }
  float sm = s2[0] + s2[1], sq = q2[0] + q2[1];
  sm = xor16_sum4(sm); sq = xor16_sum4(sq);
  sm = xor32_sum4(sm); sq = xor32_sum4(sq);
  // (3) published per chunk as two 8-byte words {value, epoch}: the tag travels WITH the data, so a reader needs one
  // round trip through memory (poll the words themselves) instead of three (data, then flag, then read) — on a call
  // this small the exchange sits on the critical path of every launch
  unsigned long long* part64 = lo.part_tag;
  const size_t part_base = (((size_t)tm * chunks_n) * 256 + r256) * 2;       // + (chunk * 256 + row of the panel) * 2
  const unsigned long long tagw = (unsigned long long)lo.epoch << 32;
  const int flag_id = panel * chunks_n + chunk;
  if (fq == 0 && !(lo.test_drop && flag_id == 1)) {
    unsigned long long* dst = part64 + part_base + ((size_t)chunk * 256 + m * 16 + fr) * 2;
    __hip_atomic_store(dst, tagw | __float_as_uint(sm), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(dst + 1, tagw | __float_as_uint(sq), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // (4)+(5) one lane per row gathers the chunks' words (its own chunk's too), bounded and sticky; the totals: chunks of
  // a 256-column group as (c0 + c1) + (c2 + c3), the groups in index order
  float2* stat_lds = reinterpret_cast<float2*>(smem + QNST * QSTAGE);     // [64] (mean, rstd)
  if (tid < 64) {
    float2 T = make_float2(0.f, 0.f);
    bool dead = (__hip_atomic_load(lo.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 2) != 0;
    int spins = 0;
    for (int k = 0; k < tiles_n; ++k) {
      float2 p[4];
      for (;;) {
        unsigned long long wv[8];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const unsigned long long* src = part64 + part_base + ((size_t)(4 * k + c) * 256 + tid) * 2;
          wv[2 * c] = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          wv[2 * c + 1] = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        bool ok = true;
#pragma unroll
        for (int c = 0; c < 8; ++c) ok = ok && (uint32_t)(wv[c] >> 32) == lo.epoch;
#pragma unroll
        for (int c = 0; c < 4; ++c) p[c] = make_float2(__uint_as_float((uint32_t)wv[2 * c]), __uint_as_float((uint32_t)wv[2 * c + 1]));
        if (ok || dead) break;
        __builtin_amdgcn_s_sleep(2);
        if (++spins > (1 << 20)) { atomicOr(lo.err, 2); dead = true; }
        else if ((spins & 255) == 0) dead = (__hip_atomic_load(lo.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 2) != 0;
      }
      const float2 P = make_float2((p[0].x + p[1].x) + (p[2].x + p[3].x), (p[0].y + p[1].y) + (p[2].y + p[3].y));
      T.x += P.x; T.y += P.y;
    }
    const float inv_n = 1.0f / (float)(tiles_n * PN);
    const float mean = T.x * inv_n;
    const float var = fmaxf(fmaf(T.y, inv_n, -mean * mean), 0.f);
    stat_lds[tid] = make_float2(mean, 1.0f / sqrtf(var + lo.eps));
  }
  __syncthreads();
  // (6) normalise and store, as gemm4_lnout_kernel
  const float2 st = stat_lds[m * 16 + fr];
  uint16_t* xrow = reinterpret_cast<uint16_t*>(out) + (size_t)(row0 + m * 16 + fr) * ldo + n_base + fq * 4;
#pragma unroll
  for (int nh = 0; nh < 2; ++nh) {
    f16x8_t hv;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const f32x4_t gv = *reinterpret_cast<const f32x4_t*>(ln_gamma + n_base + nh * 32 + i * 16 + fq * 4);
      const f32x4_t be = *reinterpret_cast<const f32x4_t*>(ln_beta + n_base + nh * 32 + i * 16 + fq * 4);
      float y[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        y[e] = ln_apply(acc[nh][i][e], st.x, st.y, gv[e], be[e]);
        hv[i * 4 + e] = (_Float16)y[e];
      }
      *reinterpret_cast<uint2*>(xrow + nh * 32 + i * 16) = make_uint2(pack_op2(y[0], y[1]), pack_op2(y[2], y[3]));
      if (out_mode == 2)
        *reinterpret_cast<f32x4_t*>(reinterpret_cast<float*>(lo.xres_out) + wave_tile + nh * 512 + i * 256 + lane * 4) =
            (f32x4_t){y[0], y[1], y[2], y[3]};
    }
    if (out_mode == 1)
      *reinterpret_cast<u32x4_t*>(lo.xres_out + wave_tile + nh * 512 + lane * 8) = *reinterpret_cast<const u32x4_t*>(&hv);
  }
}
#undef VMCNT_CASE

}  // namespace

// Takes an EPI_BIAS_LNOUT problem when the caller allows it (LnOutArgs::small_ok: 1 = flags sized for 64-row panels,
// 2 = flags and partials sized for 64 x 64 tiles too) and the whole grid is resident at once — the 64 x 64 form when
// its grid fits, else the 64 x 256 form; otherwise returns hipErrorNotReady and the caller runs the ping-pong kernel.
hipError_t KSET::launch_gemm4_lnout(const GemmArgs& g, hipStream_t st) {
  if (g.epi != EPI_BIAS_LNOUT || !g.lnout.small_ok || g.M % 256 || g.N % 256 || g.K % 128) return hipErrorNotReady;
  const int tiles_n = g.N / PN, grid = (g.M / PM) * tiles_n;
  static DeviceOnce once;
  int n_cu = 0;
  hipError_t e0 = once.get([&](int dev, int* val) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm4_lnout_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS4);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm4n_lnout_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDSQ);
    if (e != hipSuccess) return e;
    return hipDeviceGetAttribute(val, hipDeviceAttributeMultiprocessorCount, dev);
  }, &n_cu);
  if (e0 != hipSuccess) return e0;
  if (tiles_n > 8) return hipErrorNotReady;
  const int grid_n = (g.M / QM) * (g.N / QN);
  const bool narrow = g.lnout.small_ok >= 2 && g.lnout.part_tag && grid_n <= n_cu && grid_n % 8 == 0;
  if (!narrow && grid > n_cu) return hipErrorNotReady;          // partners must be co-resident: one workgroup per CU
  int res_mode, out_mode;
  const void* res = g.res;
  if (g.lnout.res_f32) {
#if CLIP_OPERAND_F16
    res_mode = g.lnout.res_tiled ? 2 : 0;
    out_mode = 2;
#else
    return hipErrorInvalidValue;
#endif
  } else if (!g.lnout.xres_out) {
#if CLIP_OPERAND_F16
    if (g.res != g.out || g.ldres != g.ldo) return hipErrorInvalidValue;
    res_mode = 0; out_mode = 0;
#else
    return hipErrorInvalidValue;
#endif
  } else {
    res_mode = g.lnout.res_tiled ? 1 : 0;
    out_mode = 1;
  }
  if (!g.lnout.part || !g.lnout.flags || !g.lnout.err || !res) return hipErrorInvalidValue;
  const void* W = g.w_tiled ? g.W_rowmajor : g.W;      // this kernel reads the row-major weight
  if (!W) return hipErrorNotReady;
  if (narrow) {
    hipLaunchKernelGGL(gemm4n_lnout_kernel, dim3(grid_n), dim3(256), LDSQ, st, static_cast<const bf16_t*>(g.A),
                       static_cast<const bf16_t*>(W), g.bias, res, g.out, g.K, g.lda, g.w_tiled ? g.K : g.ldw, g.ldo, g.ldres, tiles_n,
                       g.a_tiled ? 1 : 0, g.ln_gamma, g.ln_beta, g.lnout, res_mode, out_mode, g.M / QM);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(gemm4_lnout_kernel, dim3(grid), dim3(512), LDS4, st, static_cast<const bf16_t*>(g.A),
                     static_cast<const bf16_t*>(W), g.bias, res, g.out, g.K, g.lda, g.w_tiled ? g.K : g.ldw, g.ldo, g.ldres, tiles_n,
                     g.a_tiled ? 1 : 0, g.ln_gamma, g.ln_beta, g.lnout, res_mode, out_mode);
  return hipGetLastError();
}
