// Ping-pong ("8-phase") persistent bf16 MFMA GEMM, 256x256x64 tiles, 8 waves:
//   out[M,N] = epi(A[M,K] . W[N,K]^T + bias[N]),  M, N multiples of 256, K a multiple of 128.
// Same math, LDS row image and epilogue fusions as gemm.hip (the dense layers of
// transformers modeling_bert.py:175-177, 289-293, 334-337, 347-351 that sentclf/bert_model.py:79-88
// runs); what changes is the schedule inside the workgroup:
//
//  * A K tile (64 deep) is four 16-KiB half-tiles, staged in the order A0 (A rows 0..127),
//    B0 (W rows 0..127), B1 (W rows 128..255), A1 (A rows 128..255) into a ring of 8 slots
//    (2 K tiles), 4..6 half-tiles ahead of the one being consumed, 2 LDS-DMA pieces per wave each.
//  * A K tile is consumed in 2 phases of 32 MFMAs per wave, one 64x64 half of the wave's
//    (2 x 64) x 64 output each:  phase A reads A0 and both B n-halves -> acc[0][*]
//                                phase B reads A1 (B fragments kept)  -> acc[1][*]
//    so A0 and the B halves are dead after phase A, A1 after phase B: each slot is restaged at least
//    one phase after its last read, and every phase's ds_reads are retired (lgkmcnt(0)) BEFORE the
//    barrier that ends its load section -> no LDS-DMA can overwrite bytes still being read.
//    (A finer schedule, 4 phases of 16 MFMAs with one counted vmcnt(6) per K tile, was 4..9 % slower
//    in the main loop: git history.)
//  * Each phase is  [load section: ds_reads + 4 LDS-DMA pieces + waits] barrier [32 MFMAs] barrier.
//    Waves 4..7 run one barrier behind waves 0..3 (one extra barrier at tile start, balanced by one
//    for waves 0..3 before the epilogue), and the two waves of a SIMD are one of each group: while
//    one issues MFMAs back to back the other does its LDS reads and DMA issue (matrix beside
//    memory on every SIMD, MI355X_MICROARCH.md "Two waves per SIMD").
//  * Phase A stages (t+1).B1 and (t+1).A1, phase B stages (t+2).A0 and (t+2).B0; the VM waits are
//    counted (vmcnt(8) / vmcnt(4)): two half-tiles, or four, stay in flight across barriers, and
//    what a phase reads was retired one barrier earlier.  The stream of half-tiles runs on across
//    output tiles, so only a workgroup's first tile pays the global->LDS latency.
//  * A (and the 16-bit output) may be stored as contiguous 128 x 64 blocks (LAYOUT; GemmArgs::a_tiled / out_tiled):
//    a half-tile is then one 16-KiB block and every LDS-DMA piece 1 KiB contiguous.
//  * Epilogue (bias / polynomial erf-GELU / residual / tanh, per-wave LDS
//    transpose, whole 128-byte row segments, non-temporal stores left in flight), through a 2-KiB
//    scratch per wave beside the ring; the bias slice travels by LDS-DMA one tile ahead.
#include <cstdlib>

#include "common.h"
#include "kernels.h"

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
constexpr int BM = 256, BN = 256, BK = 64;
constexpr int HT = 16384;            // half-tile: 128 rows x 128 B
constexpr int RING = 8 * HT;         // 2 K tiles
constexpr int SCR = 2048;            // epilogue scratch per wave
constexpr int LDS8 = RING + 8 * SCR + 2 * 3072 + 4096;  // + {bias, ln gamma, ln beta}[2][256] f32 + LNOUT row sums / stats
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4_t;

#ifndef GEMM8_ABLATE
#define GEMM8_ABLATE 0  // diagnostic builds: 1 no epilogue, 2 the whole epilogue except its global stores
#endif
#ifndef GEMM8_SPLIT
#define GEMM8_SPLIT 0   // 1: the second half-tile of each phase's staging is issued in the middle of the phase's MFMA block
#endif
#ifndef LNOUT_NT_RES
#define LNOUT_NT_RES 1   // 0: diagnostic builds with cached residual loads in the fused-LayerNorm epilogue
#endif
#ifndef GEMM8_STORE
#define GEMM8_STORE 0   // diagnostic builds: 0 non-temporal output stores, 1 plain
#endif
// diagnostic builds (-DLNOUT_STAMPS, tools/lnout_stamps.py): wave 0 of every workgroup stamps its second
// tile's fused-LayerNorm epilogue with s_memtime into lo.err[64 + 8 blockIdx.x + k]; no output depends on it
#ifdef LNOUT_STAMPS
#define LN_STAMP(k)                                                                              \
  if (EPI == EPI_BIAS_LNOUT && it == 1 && tid == 0)                                              \
    reinterpret_cast<volatile uint32_t*>(lo.err)[64 + 8 * blockIdx.x + (k)] = (uint32_t)__builtin_amdgcn_s_memtime();
#else
#define LN_STAMP(k)
#endif
// diagnostic builds (-DGEMM8_STAMPS, tools/gemm8_stamps.py): waves 0 and 4 of every workgroup stamp the sections
// of ONE K tile (second output tile, K tile 4) with s_memtime into a device array read back by
// clip_debug_gemm8_stamps(); no output depends on it.  Slots: 0 phase-A start, 1 its loads retired, 2 past the
// barrier, 3 MFMAs issued, 4 past the barrier = phase-B start, 5 loads retired, 6 past the barrier, 7 MFMAs
// issued, 8 past the barrier.
#ifdef GEMM8_STAMPS
__device__ uint32_t g_gemm8_stamps[256 * 2 * 16];
#define K_STAMP(k)                                                                             \
  if (it == 1 && kk == 4 && B == 0 && lane == 0 && (w == 0 || w == 4))                          \
    g_gemm8_stamps[(blockIdx.x * 2 + (w >> 2)) * 16 + (k)] = (uint32_t)__builtin_readcyclecounter();
#else
#define K_STAMP(k)
#endif
template <typename V>
__device__ __forceinline__ void out_store(V v, V* p) {
  if (GEMM8_ABLATE == 2) { asm volatile("" ::"v"(v), "v"(p)); return; }
  if (GEMM8_STORE == 0) __builtin_nontemporal_store(v, p);
  else *p = v;
}

// v + (v of the lane 16 / 32 away): every lane of the pair ends with the sum
__device__ __forceinline__ float xor16_sum(float v) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor32_sum(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// SINGLE (EPI_BIAS_LNOUT, fp16 operands only): the LayerNorm output is written ONCE, row-major, and serves as the
// next GEMM's operand AND as the next residual (in the f16 setting the two copies of the bf16 setting hold the
// same bits); the residual is then read from the row-major buffer this tile rewrites in place (every wave reads
// exactly the 64 x 64 blocks it writes).
// LAYOUT bit 0: A, bit 1: the 16-bit output, bit 2: W, stored as contiguous 128 x 64 blocks (GemmArgs::a_tiled /
// out_tiled / w_tiled); bit 3 (EPI_BIAS_LNOUT with RES_TILED, fp16 operands): the residual read and the residual
// copy written are FP32, tiled by xres32_tiled_offset (LnOutArgs::res_f32: the residual stream of F16R)
template <int EPI, bool OUT_F32, bool RES_TILED = false, bool SINGLE = false, int LAYOUT = 0>
__global__ __launch_bounds__(512) void gemm8_kernel(
    const bf16_t* __restrict__ A, const bf16_t* __restrict__ W, const float* __restrict__ bias,
    const void* __restrict__ resv, void* __restrict__ out, int K, int lda, int ldw, int ldo,
    int ldres, int tiles_m, int tiles_n, int group_n, const float2* __restrict__ ln_stats,
    const float* __restrict__ ln_gamma, const float* __restrict__ ln_beta, LnOutArgs lo) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const bf16_t* res = reinterpret_cast<const bf16_t*>(resv);

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = w >> 2, wc = w & 3;  // wr: barrier group AND 64-row block inside each A half
  char* scr = smem + RING + w * SCR;
  float* bias_lds = reinterpret_cast<float*>(smem + RING + 8 * SCR);

  // ---- tile schedule: the workgroups of one XCD walk one sequence
  const int G = gridDim.x, per_x = G >> 3;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int n_tiles = tiles_m * tiles_n;
  // row panels are dealt to the 8 XCDs in contiguous runs (the first tiles_m % 8 XCDs take one more)
  const int panels = (tiles_m >> 3) + (xcd < (tiles_m & 7) ? 1 : 0);
  const int panel0 = xcd * (tiles_m >> 3) + min(xcd, tiles_m & 7);
  const int items = panels * tiles_n;
  auto tile_at = [&](int i, int& tm_o, int& tn_o) -> bool {
    if (EPI == EPI_BIAS_LNOUT) {
      // groups of tiles_n neighbouring workgroups (same XCD but for a few) take one row panel at a
      // time, member k always computing column tile k: they run side by side (and re-synchronise at
      // every statistics exchange), share the A panel through L2 and keep their W tile
      const int lin = xcd * per_x + slot;
      const int n_groups = G / tiles_n, grp = lin / tiles_n;
      tn_o = lin - grp * tiles_n;
      tm_o = grp + n_groups * i;
      return grp < n_groups && tm_o < tiles_m;
    }
    if (group_n > 0) {
      const int q = slot + per_x * i;
      if (q >= items) return false;
      const int per_group = panels * group_n;
      const int g = q / per_group, r = q - g * per_group;
      const int m = r / group_n;
      tm_o = panel0 + m;
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
  if (!tile_at(0, tm, tn)) return;  // workgroup-uniform

  // ---- LDS-DMA: this wave's two 1-KiB pieces of a half-tile = rows 8w..8w+7 and 64+8w..64+8w+7;
  // the source chunk carries the swizzle (the image the DMA writes is lane-linear)
  const int prow = w * 8 + (lane >> 3);
  const int pchunk = swz128(prow, lane & 7) * 8;
  // A as contiguous 128 x 64 blocks (operand_tiled_offset): a half-tile of a K tile is ONE block
  constexpr bool a_tiled = (LAYOUT & 1) != 0, out_tiled = (LAYOUT & 2) != 0;
  const int ld_a = a_tiled ? 64 : lda, a_kstep = a_tiled ? 8192 : BK;
  const size_t a_half = a_tiled ? (size_t)(K >> 6) * 8192 : (size_t)128 * lda;   // rows 128.. of the tile
  const size_t a_panel = a_tiled ? (size_t)BM * K : (size_t)BM * lda;
  constexpr bool w_tiled = (LAYOUT & 4) != 0;
  const int ld_w = w_tiled ? 64 : ldw, w_kstep = w_tiled ? 8192 : BK;
  const size_t w_half = w_tiled ? (size_t)(K >> 6) * 8192 : (size_t)128 * ldw;
  const size_t w_panel = w_tiled ? (size_t)BN * K : (size_t)BN * ldw;
  const int a_off = prow * ld_a + pchunk, w_off = prow * ld_w + pchunk;
  // The LDS-DMA loads are issued as inline assembly (m0 = LDS address of the wave's 1-KiB image).  Through the builtin
  // the compiler knows that LDS-DMA stores into `smem` are outstanding and, tracking them per array, puts
  // s_waitcnt vmcnt(0) in front of the epilogue's first LDS access (the per-wave scratch lives in the same array):
  // every epilogue then began by draining the half-tiles staged for the NEXT tile.  Hidden loads can only make the
  // compiler's own counted waits wait longer, never shorter (one in-order counter); every wait the DMA stream
  // needs is written out in this file.  (QKV - 0.7 %, FFN-up - 0.5 % in alternating bench.py processes.)
  auto dma16 = [&](const void* g, const char* lds_dst) {
    if (EPI == EPI_BIAS_LNOUT) {   // (the fused-LayerNorm kernels measured 0.5 % slower with the hidden form: builtin kept)
      __builtin_amdgcn_global_load_lds(GLB_PTR(g), LDS_PTR(lds_dst), 16, 0, 0);
      return;
    }
    const uint32_t m = (uint32_t)(uintptr_t)LDS_PTR(lds_dst);
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"   // "m0 is reserved": nothing else in this kernel uses it
    asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(m) : "memory", "m0");
#pragma clang diagnostic pop
  };
  auto stage_half = [&](const bf16_t* src, int off, int ld, int slot_idx) {
    char* dst = smem + slot_idx * HT + w * 1024;
    dma16(src + off, dst);
    dma16(src + off + 64 * ld, dst + 8192);
  };
  auto stage_bias = [&](int tn_x, int buf) {  // every wave fetches the whole 1-KiB slices: uniform counts
    char* dst = reinterpret_cast<char*>(bias_lds) + buf * 3072;
    // the lane's element offset is made opaque here, so that the three 64-bit lane addresses are formed where they
    // are used (three adds per tile) instead of being hoisted out of the tile loop: in the fused-LayerNorm kernels,
    // which sit at 256 registers, they were spilled, and every reload brought an s_waitcnt vmcnt(0) — a drain of
    // the whole LDS-DMA stream — into the K loop, three times per tile
    int l4;  // lane id re-derived on the spot (two VALU instructions, no live register across the K loop)
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l4));
    l4 *= 4;
    dma16(bias + tn_x * BN + l4, dst);
    if (EPI == EPI_BIAS_LNRES || EPI == EPI_BIAS_LNOUT) {
      dma16(ln_gamma + tn_x * BN + l4, dst + 1024);
      dma16(ln_beta + tn_x * BN + l4, dst + 2048);
    }
  };

  const int nt = K / BK;  // even
  // staging cursors: K tile t+1 (c1) and t+2 (c2) of the flattened (tile, k) stream
  const bf16_t *a1, *w1, *a2, *w2;
  bool v1, v2;
  int s_it = 0, s_kt = 0, s_tm = tm, s_tn = tn;  // position of c2
  // c1 <- c2, c2 <- the K tile after c2.  Inside a tile the cursors just step by BK (two scalar adds each);
  // the 64-bit products and tile_at's divisions run only at a tile change.  Called from phase B's load
  // section, whose end waits for the partner group anyway — never between a barrier and the next K tile's
  // first ds_read, where ~30 dependent scalar instructions used to delay the whole load section (stamps:
  // ~250 cycles per K tile on the critical path).
  auto advance = [&]() {
    a1 = a2; w1 = w2; v1 = v2;
    if (!v2) return;
    if (++s_kt == nt) {
      s_kt = 0;
      ++s_it;
      v2 = tile_at(s_it, s_tm, s_tn);
      if (!v2) return;
      a2 = A + (size_t)s_tm * a_panel;
      w2 = W + (size_t)s_tn * w_panel;
    } else {
      a2 += a_kstep;
      w2 += w_kstep;
    }
  };

  // ---- prologue: K tile 0 whole, the bias slice, three half-tiles of K tile 1
  {
    const bf16_t* a0 = A + (size_t)tm * a_panel;
    const bf16_t* w0 = W + (size_t)tn * w_panel;
    stage_half(a0, a_off, ld_a, 0);
    stage_half(w0, w_off, ld_w, 1);
    stage_half(w0 + w_half, w_off, ld_w, 2);
    stage_half(a0 + a_half, a_off, ld_a, 3);
    stage_bias(tn, 0);
    a2 = a0; w2 = w0; v2 = true;
    advance();  // c2 = K tile 1
    advance();  // c1 = K tile 1, c2 = K tile 2 (or the next tile's first)
    stage_half(a1, a_off, ld_a, 4);
    stage_half(w1, w_off, ld_w, 5);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }

  const int fr = lane & 15, fq = lane >> 4;
  const int lsw = (fr >> 1) & 7;
  const int a_rd = (wr * 64 + fr) * 128;                                  // within an A half-tile
  const int b_rd = (1 + (wc >> 1)) * HT + ((wc & 1) * 64 + fr) * 128;     // B half slot + row
  const int c0 = (fq ^ lsw) * 16, c1 = ((4 + fq) ^ lsw) * 16;             // chunk of k-step 0 / 1

  int it = 0;
  while (true) {
    int ntm = 0, ntn = 0;
    const bool has_next = tile_at(it + 1, ntm, ntn);

    // acc[mh][nh][m][i]: lane holds n = 4 fq .. +3 of 16-col block (nh, i), row fr of 16-row block
    // (mh, m).  The bias is added after the K loop, as the other GEMM kernels do: a row's result
    // is bit-identical whichever kernel (i.e. whichever batch size) computed it.
    f32x4_t acc[2][2][4][2];
#pragma unroll
    for (int mh = 0; mh < 2; ++mh)
#pragma unroll
      for (int nh = 0; nh < 2; ++nh)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int i = 0; i < 2; ++i) acc[mh][nh][m][i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    if (wr == 1) __builtin_amdgcn_s_barrier();  // waves 4..7 run one barrier behind
    LN_STAMP(0)

    opx8_t af[4][2], bf0[2][2], bf1[2][2];
    for (int kk = 0; kk < nt; kk += 2) {
      auto ktile = [&]<int B>() {
        const char* buf = smem + B * 4 * HT;
        K_STAMP(0)
#ifdef GEMM8_STAMPS
        if (it == 1 && kk == 4 && B == 1 && lane == 0 && (w == 0 || w == 4))   // start of the NEXT K tile: slot 9
          g_gemm8_stamps[(blockIdx.x * 2 + (w >> 2)) * 16 + 9] = (uint32_t)__builtin_readcyclecounter();
#endif
        // ---------------- phase A: A0, both B n-halves -> acc[0][*]; stage (t+1).B1, (t+1).A1
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          bf0[i][0] = *reinterpret_cast<const opx8_t*>(buf + b_rd + i * 2048 + c0);
          bf0[i][1] = *reinterpret_cast<const opx8_t*>(buf + b_rd + i * 2048 + c1);
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          af[m][0] = *reinterpret_cast<const opx8_t*>(buf + a_rd + m * 2048 + c0);
          af[m][1] = *reinterpret_cast<const opx8_t*>(buf + a_rd + m * 2048 + c1);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          bf1[i][0] = *reinterpret_cast<const opx8_t*>(buf + b_rd + 4096 + i * 2048 + c0);
          bf1[i][1] = *reinterpret_cast<const opx8_t*>(buf + b_rd + 4096 + i * 2048 + c1);
        }
        if (v1) {
          stage_half(w1 + w_half, w_off, ld_w, (B ^ 1) * 4 + 2);
          if (GEMM8_SPLIT == 0) stage_half(a1 + a_half, a_off, ld_a, (B ^ 1) * 4 + 3);
          // in flight behind this wait: (t+1).A0/B0 from the previous phase and the four just issued;
          // (t).A1, read next phase, is older
          if (GEMM8_SPLIT == 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        // younger than everything the wait above covers, older than phase B's DMA: retired by B's wait
        if (B == 1 && kk == 0 && has_next) stage_bias(ntn, (it + 1) & 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        K_STAMP(1)
        __builtin_amdgcn_s_barrier();
        K_STAMP(2)
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
          for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              acc[0][0][m][i] = MFMA_16x16x32(bf0[i][ks], af[m][ks], acc[0][0][m][i]);
              acc[0][1][m][i] = MFMA_16x16x32(bf1[i][ks], af[m][ks], acc[0][1][m][i]);
            }
          if (GEMM8_SPLIT == 1 && ks == 0) {
            __builtin_amdgcn_sched_barrier(0);
            if (v1) stage_half(a1 + a_half, a_off, ld_a, (B ^ 1) * 4 + 3);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        K_STAMP(3)
        __builtin_amdgcn_s_barrier();
        K_STAMP(4)
        // ---------------- phase B: A1 -> acc[1][*]; stage (t+2).A0, (t+2).B0; K tile t+1 retired
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          af[m][0] = *reinterpret_cast<const opx8_t*>(buf + 3 * HT + a_rd + m * 2048 + c0);
          af[m][1] = *reinterpret_cast<const opx8_t*>(buf + 3 * HT + a_rd + m * 2048 + c1);
        }
        const bool staged2 = v2;
        const bf16_t* w2_mid = w2;  // (GEMM8_SPLIT: the W half staged mid-block, after advance() moved the cursor)
        if (v2) {
          stage_half(a2, a_off, ld_a, B * 4 + 0);
          if (GEMM8_SPLIT == 0) stage_half(w2, w_off, ld_w, B * 4 + 1);
        }
        advance();
        if (staged2) {
          asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        K_STAMP(5)
        __builtin_amdgcn_s_barrier();
        K_STAMP(6)
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
          for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              acc[1][1][m][i] = MFMA_16x16x32(bf1[i][ks], af[m][ks], acc[1][1][m][i]);
              acc[1][0][m][i] = MFMA_16x16x32(bf0[i][ks], af[m][ks], acc[1][0][m][i]);
            }
          if (GEMM8_SPLIT == 1 && ks == 0) {
            __builtin_amdgcn_sched_barrier(0);
            if (staged2) stage_half(w2_mid, w_off, ld_w, B * 4 + 1);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        K_STAMP(7)
        __builtin_amdgcn_s_barrier();
        K_STAMP(8)
      };
      ktile.template operator()<0>();
      ktile.template operator()<1>();
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();  // balances the extra barrier of waves 4..7
    LN_STAMP(1)

    // ---- epilogue: 16-row passes through this wave's scratch, whole 128-byte row segments out
    const int n_base = tn * BN + wc * 64;
    f32x4_t bv[2][2];
#pragma unroll
    for (int nh = 0; nh < 2; ++nh)
#pragma unroll
      for (int i = 0; i < 2; ++i)
        bv[nh][i] = *reinterpret_cast<const f32x4_t*>(bias_lds + (it & 1) * 768 + wc * 64 + fq * 4 + nh * 32 + i * 16);
    if (GEMM8_ABLATE == 1) {
#pragma unroll
      for (int mh = 0; mh < 2; ++mh)
#pragma unroll
        for (int nh = 0; nh < 2; ++nh)
#pragma unroll
          for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int i = 0; i < 2; ++i) asm volatile("" ::"v"(acc[mh][nh][m][i]));
    } else if (EPI == EPI_BIAS_LNOUT) {
      // ---- fused LayerNorm.  (1) v = acc + bias + fp16 residual, in accumulator layout.  The residual
      // is the tiled copy an earlier launch of this epilogue wrote (1-KiB wave loads), or, for the first
      // layer, the row-major copy of the embedding LayerNorm.
      typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
      const size_t xt_base = (((size_t)tm * tiles_n + tn) * 8 + w) * 8192 + lane * 8;  // + blk * 1024 + nh * 512
      constexpr bool res32 = (LAYOUT & 8) != 0;
      if (RES_TILED && res32) {
        // wave-uniform base + one 32-bit lane offset: the 32 loads share ONE vector register of addressing (scalar
        // base, immediate offsets) instead of eight 64-bit lane addresses — this kernel has no registers to spare
        const float* rt_u = reinterpret_cast<const float*>(resv) + (((size_t)tm * tiles_n + tn) * 8 + w) * 8192;
        const unsigned l4 = lane * 4;
#pragma unroll
        for (int mh = 0; mh < 2; ++mh)
#pragma unroll
          for (int mp = 0; mp < 4; mp += 2) {
            __builtin_amdgcn_sched_barrier(0);  // 8 loads (32 registers) in flight at a time
            f32x4_t r[2][2][2];
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
              for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                for (int i = 0; i < 2; ++i)
                  r[k][nh][i] = __builtin_nontemporal_load(
                      reinterpret_cast<const f32x4_t*>(rt_u + ((mh * 4 + mp + k) * 1024 + nh * 512 + i * 256) + l4));
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
              for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                  for (int e = 0; e < 4; ++e)
                    acc[mh][nh][mp + k][i][e] = (acc[mh][nh][mp + k][i][e] + bv[nh][i][e]) + r[k][nh][i][e];
            // pin the sums here: left alone, the compiler issues the loads of every group first and adds later
            // (all 32 loads = 128 registers live at once -> ~60 spilled registers, some reloaded inside the K loop)
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
              for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                for (int i = 0; i < 2; ++i) asm volatile("" : "+v"(acc[mh][nh][mp + k][i]));
          }
      } else if (RES_TILED) {
        const uint16_t* rt = reinterpret_cast<const uint16_t*>(resv) + xt_base;
#pragma unroll
        for (int mh = 0; mh < 2; ++mh) {
          __builtin_amdgcn_sched_barrier(0);  // 8 loads (32 registers) in flight at a time, not 16
#pragma unroll
          for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int nh = 0; nh < 2; ++nh) {
              // read once: non-temporal, so that 128 KB per tile do not push the W tile and the A panels out of the
              // XCD's L2 (attn-out 220.6 -> 200.1 us at M = 130 560; FFN-down, whose W slice does not fit anyway, unchanged)
#if LNOUT_NT_RES
              const f16x8_t r = __builtin_nontemporal_load(reinterpret_cast<const f16x8_t*>(rt + (mh * 4 + m) * 1024 + nh * 512));
#else
              const f16x8_t r = *reinterpret_cast<const f16x8_t*>(rt + (mh * 4 + m) * 1024 + nh * 512);
#endif
#pragma unroll
              for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                  acc[mh][nh][m][i][e] = (acc[mh][nh][m][i][e] + bv[nh][i][e]) + (float)r[i * 4 + e];
            }
        }
      } else {
        // Row-major fp16 residual: the first layer's (the embedding LayerNorm's copy), or — SINGLE — the output
        // buffer's current content (read through `out`, not through the second restrict-qualified pointer; these
        // loads come before the barriers of the exchange, the stores after).  Whole 128-byte row segments (16 B
        // per lane, 8 rows per instruction) through this wave's transpose scratch, read back in accumulator
        // layout — the output path backwards; two blocks (4 loads) in flight.
        const uint16_t* xr = SINGLE ? reinterpret_cast<const uint16_t*>(out) : reinterpret_cast<const uint16_t*>(resv);
        const int ldr = SINGLE ? ldo : ldres;
        const int lrow = lane >> 3, c16 = lane & 7;
        const uint16_t* xr0 = xr + ((size_t)tm * BM + wr * 64 + lrow) * ldr + n_base + c16 * 8;
        char* scw = scr + lrow * 128 + ((c16 ^ (lrow & 7)) * 16);   // (row + 8) & 7 == row & 7: + 1024 for p = 1
#pragma unroll
        for (int mb = 0; mb < 8; mb += 2) {   // two 16-row blocks (4 loads) in flight
          __builtin_amdgcn_sched_barrier(0);
          u32x4_t rv[2][2];
#pragma unroll
          for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
              const int mh = (mb + k) >> 2, m = (mb + k) & 3;
#if LNOUT_NT_RES
              rv[k][p] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(xr0 + (size_t)(mh * 128 + m * 16 + p * 8) * ldr));
#else
              rv[k][p] = *reinterpret_cast<const u32x4_t*>(xr0 + (size_t)(mh * 128 + m * 16 + p * 8) * ldr);
#endif
            }
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            const int mh = (mb + k) >> 2, m = (mb + k) & 3;
            *reinterpret_cast<u32x4_t*>(scw) = rv[k][0];
            *reinterpret_cast<u32x4_t*>(scw + 1024) = rv[k][1];
#pragma unroll
            for (int nh = 0; nh < 2; ++nh)
#pragma unroll
              for (int i = 0; i < 2; ++i) {
                const int sl = ((nh * 2 + i) * 4 + fq) ^ ((fr & 7) << 1);
                const f16x4_t r = *reinterpret_cast<const f16x4_t*>(scr + fr * 128 + sl * 8);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[mh][nh][m][i][e] = (acc[mh][nh][m][i][e] + bv[nh][i][e]) + (float)r[e];
              }
          }
        }
      }
      LN_STAMP(2)
      // (2) per-row (sum, sum of squares) over this wave's 64 columns -> LDS, one slot per column wave
      // (each wave's 128 row partials go into ITS OWN scratch: a wave that is ahead must not write where another
      // wave is still transposing its residual rows)
      float2* part_own = reinterpret_cast<float2*>(scr);                               // [2 mh][4 m][16 rows]
      float2* tot_lds = reinterpret_cast<float2*>(smem + RING + 8 * SCR + 2 * 3072);   // [256] this tile's sums
      float2* stat_lds = tot_lds + 256;                                                // [256] (mean, rstd)
#pragma unroll
      for (int mh = 0; mh < 2; ++mh)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          // packed fp32 adds / fmas over the lane's 16 values, then the 4 lanes of a row (lane bits 4, 5) by
          // v_permlane16_swap / v_permlane32_swap (VALU; the ds_bpermute form of __shfl_xor went through the LDS)
          f32x2_t s2 = {0.f, 0.f}, q2 = {0.f, 0.f};
#pragma unroll
          for (int nh = 0; nh < 2; ++nh)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              const f32x2_t lo = {acc[mh][nh][m][i][0], acc[mh][nh][m][i][1]};
              const f32x2_t hi = {acc[mh][nh][m][i][2], acc[mh][nh][m][i][3]};
              s2 += lo; s2 += hi;
              q2 = __builtin_elementwise_fma(lo, lo, q2);
              q2 = __builtin_elementwise_fma(hi, hi, q2);
            }
          float sm = s2[0] + s2[1], sq = q2[0] + q2[1];
          sm = xor16_sum(sm); sq = xor16_sum(sq);
          sm = xor32_sum(sm); sq = xor32_sum(sq);
          if (fq == 0) part_own[(mh * 4 + m) * 16 + fr] = make_float2(sm, sq);
        }
      __syncthreads();
      LN_STAMP(3)
      // (3) the workgroup's 256-column partial per row, then the flag.  Ordering on gfx950, made explicit:
      //   writer: the partials leave as agent-scope atomic stores (sc1: written through to the memory
      //     side every XCD reads from, never parked in this XCD's L2); every storing wave waits
      //     vmcnt(0) — stores are counted in vmcnt until the memory side has acknowledged them — and only
      //     then reaches the workgroup barrier; thread 0 issues the flag store AFTER that barrier.  So
      //     the flag cannot become visible before any partial of this tile (no release fence is needed:
      //     a fence would add an L2 write-back of the whole XCD to every tile).
      //   reader: the poll is an agent-scope atomic load (served from the memory side, not from this XCD's
      //     L2); the partials are read AFTER the poller's barrier by agent-scope atomic loads as well,
      //     issued in program order behind the flag observation (a wave issues its memory instructions
      //     in order and no load is speculated past s_barrier), so they cannot return bytes older than
      //     the flag.  Epochs make a stale flag of an earlier launch unequal to the awaited value.
      const int tile_id = tm * tiles_n + tn;
      if (tid < 256) {
        // row tid = (mh, wr, m, fr): its four column waves are wr * 4 + 0..3
        const float2* pr = reinterpret_cast<const float2*>(smem + RING + ((tid >> 6) & 1) * 4 * SCR) +
                           ((tid >> 7) * 4 + ((tid >> 4) & 3)) * 16 + (tid & 15);
        const float2 p0 = pr[0], p1 = pr[SCR / 8], p2 = pr[2 * SCR / 8], p3 = pr[3 * SCR / 8];
        const float2 P = make_float2((p0.x + p1.x) + (p2.x + p3.x), (p0.y + p1.y) + (p2.y + p3.y));
        tot_lds[tid] = P;
        unsigned long long bits;
        __builtin_memcpy(&bits, &P, 8);
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(lo.part) + (size_t)tile_id * 256 + tid, bits,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0 && !(lo.test_drop && tile_id == 1))
        __hip_atomic_store(lo.flags + tile_id, lo.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      LN_STAMP(4)
      // (4) wait for the other column tiles of this row panel (lane 0 of wave k polls tile k), bounded:
      // the partners are resident and one tile-time away at most, unless the GPU is shared.  A timeout
      // is sticky (bit 1 of *err stays set until the host polls it): once any tile has timed out, no later
      // tile waits — the call is void anyway (its logits are turned into NaN) and must end quickly.
#ifndef LNOUT_NOSYNC  // diagnostic builds
      if (lane == 0 && w < tiles_n && w != tn) {
        const uint32_t* f = lo.flags + tm * tiles_n + w;
        int spins = 0;
        bool dead = (__hip_atomic_load(lo.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 2) != 0;
        while (!dead && __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != lo.epoch) {
          __builtin_amdgcn_s_sleep(4);
          if (++spins > (1 << 20)) { atomicOr(lo.err, 2); break; }
          if ((spins & 255) == 0) dead = (__hip_atomic_load(lo.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 2) != 0;
        }
      }
#endif
      __syncthreads();
      LN_STAMP(5)
      if (tid < 256) {
        float2 T = make_float2(0.f, 0.f);
        for (int k = 0; k < tiles_n; ++k) {  // fixed order: every tile of the panel gets the same bits
          float2 P;
          if (k == tn) {
            P = tot_lds[tid];
          } else {
            const unsigned long long bits = __hip_atomic_load(
                reinterpret_cast<const unsigned long long*>(lo.part) + (size_t)(tm * tiles_n + k) * 256 + tid,
                __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_memcpy(&P, &bits, 8);
          }
          T.x += P.x; T.y += P.y;
        }
        const float inv_n = 1.0f / (float)(tiles_n * BN);
        const float mean = T.x * inv_n;
        const float var = fmaxf(fmaf(T.y, inv_n, -mean * mean), 0.f);
        stat_lds[tid] = make_float2(mean, 1.0f / sqrtf(var + lo.eps));
      }
      __syncthreads();
      LN_STAMP(6)
      // (5) normalise in registers; the fp16 residual copy goes out tiled straight from the registers
      // (1-KiB wave stores), the bf16 GEMM operand through the transpose scratch as whole row segments
      const float* gl = bias_lds + (it & 1) * 768 + 256 + wc * 64 + fq * 4;
      uint16_t* xt = SINGLE ? nullptr : lo.xres_out + xt_base;
      float* xt32_u = reinterpret_cast<float*>(lo.xres_out) + (((size_t)tm * tiles_n + tn) * 8 + w) * 8192;  // wave-uniform
#pragma unroll
      for (int mh = 0; mh < 2; ++mh)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const float2 st = stat_lds[mh * 128 + wr * 64 + m * 16 + fr];
          const size_t m0 = (size_t)tm * BM + mh * 128 + wr * 64 + m * 16;
#pragma unroll
          for (int nh = 0; nh < 2; ++nh) {
            f16x8_t hv;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              const f32x4_t gv = *reinterpret_cast<const f32x4_t*>(gl + nh * 32 + i * 16);
              const f32x4_t be = *reinterpret_cast<const f32x4_t*>(gl + 256 + nh * 32 + i * 16);
              float y[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                y[e] = ln_apply(acc[mh][nh][m][i][e], st.x, st.y, gv[e], be[e]);
                hv[i * 4 + e] = (_Float16)y[e];
              }
              const int sl = ((nh * 2 + i) * 4 + fq) ^ ((fr & 7) << 1);
              *reinterpret_cast<uint2*>(scr + fr * 128 + sl * 8) =
                  make_uint2(pack_op2(y[0], y[1]), pack_op2(y[2], y[3]));
              if (res32)   // the residual stream stays fp32: 16 B per lane, 1 KiB per wave store
                out_store((f32x4_t){y[0], y[1], y[2], y[3]},
                          reinterpret_cast<f32x4_t*>(xt32_u + ((mh * 4 + m) * 1024 + nh * 512 + i * 256) + (unsigned)(lane * 4)));
            }
            if (!SINGLE && !res32)
              out_store(*reinterpret_cast<const u32x4_t*>(&hv),
                        reinterpret_cast<u32x4_t*>(xt + (mh * 4 + m) * 1024 + nh * 512));
          }
#pragma unroll
          for (int p = 0; p < 2; ++p) {
            const int row = p * 8 + (lane >> 3), c16 = lane & 7;
            const uint4 v = *reinterpret_cast<const uint4*>(scr + row * 128 + ((c16 ^ (row & 7)) * 16));
            out_store(*reinterpret_cast<const u32x4_t*>(&v),
                      reinterpret_cast<u32x4_t*>(reinterpret_cast<uint16_t*>(out) + (m0 + row) * ldo + n_base + c16 * 8));
          }
        }
      LN_STAMP(7)
    } else if (!OUT_F32) {
#pragma unroll
      for (int mh = 0; mh < 2; ++mh)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          // image [16 rows][128 B]; 8-byte slots XOR-swizzled in pairs so the 16-B reads stay whole
#pragma unroll
          for (int nh = 0; nh < 2; ++nh)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              float v0 = acc[mh][nh][m][i][0] + bv[nh][i][0], v1 = acc[mh][nh][m][i][1] + bv[nh][i][1];
              float v2f = acc[mh][nh][m][i][2] + bv[nh][i][2], v3 = acc[mh][nh][m][i][3] + bv[nh][i][3];
              if (EPI == EPI_BIAS_GELU) {
                // packed fp32 math: same bits per element as gelu_erf_bf16out, 296 -> 285 us on FFN-up
                f32x2_t g01 = {v0, v1}, g23 = {v2f, v3};
                gelu_erf_bf16out_x4(g01, g23);
                v0 = g01[0]; v1 = g01[1]; v2f = g23[0]; v3 = g23[1];
              } else if (EPI == EPI_BIAS_TANH) {
                v0 = tanhf(v0); v1 = tanhf(v1); v2f = tanhf(v2f); v3 = tanhf(v3);
              }
              const int sl = ((nh * 2 + i) * 4 + fq) ^ ((fr & 7) << 1);
              *reinterpret_cast<uint2*>(scr + fr * 128 + sl * 8) =
                  make_uint2(pack_op2(v0, v1), pack_op2(v2f, v3));
            }
          const size_t m0 = (size_t)tm * BM + mh * 128 + wr * 64 + m * 16;
#pragma unroll
          for (int p = 0; p < 2; ++p) {
            const int row = p * 8 + (lane >> 3), c16 = lane & 7;
            const size_t mg = m0 + row;
            const int n = n_base + c16 * 8;
            uint4 v = *reinterpret_cast<const uint4*>(scr + row * 128 + ((c16 ^ (row & 7)) * 16));
            if (EPI == EPI_BIAS_RES) {
              const uint4 r = *reinterpret_cast<const uint4*>(res + mg * ldres + n);
              const uint32_t* vi = reinterpret_cast<const uint32_t*>(&v);
              const uint32_t* ri = reinterpret_cast<const uint32_t*>(&r);
              uint32_t o[4];
#pragma unroll
              for (int k = 0; k < 4; ++k)
                o[k] = pack_op2(op2_lo(vi[k]) + op2_lo(ri[k]),
                                   op2_hi(vi[k]) + op2_hi(ri[k]));
              v = make_uint4(o[0], o[1], o[2], o[3]);
            }
            out_store(*reinterpret_cast<u32x4_t*>(&v),
                      reinterpret_cast<u32x4_t*>(reinterpret_cast<bf16_t*>(out) +
                                                 (out_tiled ? operand_tiled_offset(mg, n, ldo) : mg * ldo + n)));
          }
        }
    } else {
#pragma unroll
      for (int mh = 0; mh < 2; ++mh)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int nh = 0; nh < 2; ++nh) {
            // image [16 rows][32 f32 = 128 B], 16-byte slots XOR-swizzled by row
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              float v0 = acc[mh][nh][m][i][0] + bv[nh][i][0], v1 = acc[mh][nh][m][i][1] + bv[nh][i][1];
              float v2f = acc[mh][nh][m][i][2] + bv[nh][i][2], v3 = acc[mh][nh][m][i][3] + bv[nh][i][3];
              if (EPI == EPI_BIAS_GELU) {
                v0 = gelu_erf(v0); v1 = gelu_erf(v1); v2f = gelu_erf(v2f); v3 = gelu_erf(v3);
              } else if (EPI == EPI_BIAS_TANH) {
                v0 = tanhf(v0); v1 = tanhf(v1); v2f = tanhf(v2f); v3 = tanhf(v3);
              }
              *reinterpret_cast<float4*>(scr + fr * 128 + (((i * 4 + fq) ^ (fr & 7)) * 16)) =
                  make_float4(v0, v1, v2f, v3);
            }
            const size_t m0 = (size_t)tm * BM + mh * 128 + wr * 64 + m * 16;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
              const int row = p * 8 + (lane >> 3), c16 = lane & 7;
              const size_t mg = m0 + row;
              const int n = n_base + nh * 32 + c16 * 4;
              float4 v = *reinterpret_cast<const float4*>(scr + row * 128 + ((c16 ^ (row & 7)) * 16));
              if (EPI == EPI_BIAS_RES) {
                const uint2 rv = *reinterpret_cast<const uint2*>(res + mg * ldres + n);
                v.x += op2_lo(rv.x); v.y += op2_hi(rv.x);
                v.z += op2_lo(rv.y); v.w += op2_hi(rv.y);
              } else if (EPI == EPI_BIAS_LNRES) {
                const float4 rv = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(resv) + mg * ldres + n);
                const float2 sv = ln_stats[mg];
                const float* gl = bias_lds + (it & 1) * 768 + 256 + wc * 64 + nh * 32 + c16 * 4;
                const float4 gv = *reinterpret_cast<const float4*>(gl);
                const float4 be = *reinterpret_cast<const float4*>(gl + 256);
                v.x += ln_apply(rv.x, sv.x, sv.y, gv.x, be.x); v.y += ln_apply(rv.y, sv.x, sv.y, gv.y, be.y);
                v.z += ln_apply(rv.z, sv.x, sv.y, gv.z, be.z); v.w += ln_apply(rv.w, sv.x, sv.y, gv.w, be.w);
              }
              out_store(*reinterpret_cast<f32x4_t*>(&v),
                                          reinterpret_cast<f32x4_t*>(reinterpret_cast<float*>(out) + mg * ldo + n));
            }
          }
    }
    if (!has_next) break;
    tm = ntm; tn = ntn;
    ++it;
  }
}

template <int EPI, bool OUT_F32, bool RES_TILED = false, bool SINGLE = false, int LAYOUT = 0>
hipError_t launch_8(const GemmArgs& g, hipStream_t st) {
  auto kfn = gemm8_kernel<EPI, OUT_F32, RES_TILED, SINGLE, LAYOUT>;
  static DeviceOnce once;  // per instantiation and device: function attribute + workgroup count
  int n_cu = 0;
  hipError_t e0 = once.get([&](int dev, int* val) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS8);
    if (e != hipSuccess) return e;
    int cus = 0;
    if ((e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return e;
    *val = cus & ~7;
    if (*val < 8) return hipErrorInvalidValue;
    if (const char* cap = getenv("GEMM8_MAX_WGS")) {   // diagnostic: tools/dual_stream_bench.py
      const int c = atoi(cap) & ~7;
      if (c >= 8 && c < *val) *val = c;
    }
    return hipSuccess;
  }, &n_cu);
  if (e0 != hipSuccess) return e0;
  const int tiles_m = g.M / BM, tiles_n = g.N / BN, n_tiles = tiles_m * tiles_n;
  const int grid = n_cu < ((n_tiles + 7) & ~7) ? n_cu : ((n_tiles + 7) & ~7);
  // column-tile group whose W slice (group_n x 256 rows x K) fits an XCD's L2 beside the A stream.  Round 2
  // sweep (M = 130 560): slices of 1.2 / 1.6 / 2.4 / 3.5 MB run in the same time (QKV 380-382 us, FFN-up 544 us;
  // 0.8 MB and the whole 4.7 MB FFN-up weight are 1-7 % slower), so the largest that fits is taken: the A panel
  // is then fetched once (QKV) or twice (FFN-up) instead of 3 times
  int group_n = 0;
  if (tiles_m >= 8) {
    group_n = tiles_n;
    static const long group_bytes = getenv("GEMM8_GROUP_BYTES") ? atol(getenv("GEMM8_GROUP_BYTES")) : 3600000;  // diagnostic override
    if ((long)tiles_n * BN * g.K * 2 > group_bytes && g.K < 2048)
      for (int d = 1; d <= tiles_n; ++d)
        if (tiles_n % d == 0 && (long)d * BN * g.K * 2 <= group_bytes) group_n = d;
  }
  hipLaunchKernelGGL(kfn, dim3(grid), dim3(512), LDS8, st, static_cast<const bf16_t*>(g.A),
                     static_cast<const bf16_t*>(g.W), g.bias, g.res, g.out, g.K,
                     g.lda, g.ldw, g.ldo, g.ldres, tiles_m, tiles_n, group_n,
                     reinterpret_cast<const float2*>(g.ln_stats), g.ln_gamma, g.ln_beta, g.lnout);
  return hipGetLastError();
}

}  // namespace

hipError_t KSET::launch_gemm8(const GemmArgs& g, hipStream_t st) {
  if (!gemm_persist_ok(g) || g.res_index) return hipErrorInvalidValue;
  // block layouts: the engine's two single-reader operands only (attention ctx / FFN intermediate -> LayerNorm GEMMs)
  if (g.a_tiled && (g.lda != g.K || g.epi != EPI_BIAS_LNOUT || !g.w_tiled)) return hipErrorInvalidValue;
  if (g.out_tiled && (g.ldo != g.N || g.out_f32 || g.a_tiled || g.w_tiled || g.epi != EPI_BIAS_GELU)) return hipErrorInvalidValue;
  if (g.w_tiled && (g.ldw != g.K || g.out_f32 || (g.epi != EPI_BIAS_LNOUT && g.epi != EPI_BIAS) ||
                    (g.epi == EPI_BIAS_LNOUT && !g.a_tiled)))
    return hipErrorInvalidValue;   // (the instantiated forms: QKV with W in blocks, LayerNorm GEMMs with A and W in blocks)
  switch (g.epi * 2 + (g.out_f32 ? 1 : 0)) {
    case EPI_BIAS * 2 + 0:
      return g.w_tiled ? launch_8<EPI_BIAS, false, false, false, 4>(g, st) : launch_8<EPI_BIAS, false>(g, st);
    case EPI_BIAS * 2 + 1: return launch_8<EPI_BIAS, true>(g, st);
    case EPI_BIAS_GELU * 2 + 0:
      return g.out_tiled ? launch_8<EPI_BIAS_GELU, false, false, false, 2>(g, st) : launch_8<EPI_BIAS_GELU, false>(g, st);
    case EPI_BIAS_RES * 2 + 1: return launch_8<EPI_BIAS_RES, true>(g, st);
    case EPI_BIAS_RES * 2 + 0: return launch_8<EPI_BIAS_RES, false>(g, st);
    case EPI_BIAS_TANH * 2 + 0: return launch_8<EPI_BIAS_TANH, false>(g, st);
    case EPI_BIAS_LNRES * 2 + 1: return launch_8<EPI_BIAS_LNRES, true>(g, st);
    case EPI_BIAS_LNOUT * 2 + 0: {
      // few rows: 64 x 256 tiles on 4x the workgroups (gemm4.hip), the same bits
      const hipError_t e4 = launch_gemm4_lnout(g, st);
      if (e4 != hipErrorNotReady) return e4;
    }
      if (g.N / BN > 8 || !g.lnout.part || !g.lnout.flags || !g.lnout.err || !g.res) return hipErrorInvalidValue;
      if (!g.lnout.xres_out) {
#if CLIP_OPERAND_F16
        // one copy: operand = residual (fp16), read row-major from the buffer the tile rewrites
        if (g.lnout.res_tiled || g.res != g.out || g.ldres != g.ldo) return hipErrorInvalidValue;
        return g.a_tiled ? launch_8<EPI_BIAS_LNOUT, false, false, true, 5>(g, st)
                         : launch_8<EPI_BIAS_LNOUT, false, false, true>(g, st);
#else
        return hipErrorInvalidValue;   // bf16 operands need the separate fp16 residual copy
#endif
      }
      if (g.lnout.res_f32) {
#if CLIP_OPERAND_F16
        if (g.lnout.res_tiled)
          return g.a_tiled ? launch_8<EPI_BIAS_LNOUT, false, true, false, 13>(g, st)
                           : launch_8<EPI_BIAS_LNOUT, false, true, false, 8>(g, st);
        // where the fp32 stream starts (the first fused GEMM of a forward): the residual is the embedding LayerNorm's
        // fp16 output, read row-major from the operand buffer this tile rewrites (the SINGLE read path)
        if (g.res != g.out || g.ldres != g.ldo) return hipErrorInvalidValue;
        return g.a_tiled ? launch_8<EPI_BIAS_LNOUT, false, false, true, 13>(g, st)
                         : launch_8<EPI_BIAS_LNOUT, false, false, true, 8>(g, st);
#else
        return hipErrorInvalidValue;   // the fp32 residual stream belongs to the fp16 operand set (F16R)
#endif
      }
      if (g.a_tiled)
        return g.lnout.res_tiled ? launch_8<EPI_BIAS_LNOUT, false, true, false, 5>(g, st)
                                 : launch_8<EPI_BIAS_LNOUT, false, false, false, 5>(g, st);
      return g.lnout.res_tiled ? launch_8<EPI_BIAS_LNOUT, false, true>(g, st) : launch_8<EPI_BIAS_LNOUT, false, false>(g, st);
    default: return hipErrorInvalidValue;
  }
}

#ifdef GEMM8_STAMPS
#if !CLIP_OPERAND_F16
extern "C" int clip_debug_gemm8_stamps(uint32_t* host, int n_words) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_gemm8_stamps), (size_t)n_words * 4);
}
#endif
#endif
