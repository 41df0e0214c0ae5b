// Varlen (packed, cu_seqlens) bidirectional self-attention, head_dim 64, bf16 MFMA, fp32 softmax.
// Restates softmax(Q K^T / sqrt(d) + key-padding mask) V of transformers BertSelfAttention
// (modeling_bert.py:164-203, math :111-136) for the reference's sentclf/bert_model.py:79-88.
// Packing replaces the padding mask (keys past a sequence's end do not exist); 1/sqrt(64) is
// folded exactly into Wq/bq at weight load.
//
// Workgroup = 4 waves = one 128-row q tile of one (sequence, head); each wave owns 32 q rows.
// Per 64-key tile:  S^T = K Q^T  (32x32x16 MFMA, key on rows, q on the lane -> a q row's softmax
// statistics are lane-local), P = exp2(S^T - m) converted in place to the B operand of
// O^T = V^T P^T (the accumulator-as-operand k order of the guide, §3), V^T fragments come from
// the row-major V tile in LDS through ds_read_b64_tr_b16.  K swizzled for ds_read_b128, V swizzled
// for the transposed read.  K/V tiles reach LDS in one of two ways, chosen per launch:
//   attention_kernel<false> (sequences of several key tiles per q tile): by LDS-DMA into two tile
//     buffers — the next tile's loads are issued right after the barrier that hands over the current
//     one, no staging registers, no ds_write, one barrier per tile;
//   attention_kernel<true>  (max_seqlen <= 128, nothing to overlap): through registers, the next
//     tile's global loads issued before the current tile's MFMAs and written to LDS after a barrier.
#include "common.h"
#include "kernels.h"

namespace {

#ifndef ATTN_DMA
#define ATTN_DMA 1          // 0: K/V tiles through registers and ds_write (diagnostic builds)
#endif
#ifndef ATTN_PAIR
#define ATTN_PAIR 1          // 0: diagnostic builds with the 2-D grid for every launch
#endif
#ifndef ATTN_NBUF
#define ATTN_NBUF 2          // K/V tile buffers of the LDS-DMA form: NBUF - 1 tiles in flight beside the one being computed
                             // (3: 48 KB of LDS, 3 workgroups per CU instead of 4: attention 1.89 -> 2.06 ms per step, EXPERIMENTS.md E)
#endif
constexpr int HD = 64;       // head dim
constexpr int KT = 64;       // keys per tile
constexpr float LOG2E = 1.4426950408889634f;

__device__ __forceinline__ int swz_v(int row, int chunk) { return chunk ^ (((row >> 1) & 1) << 2); }

// diagnostic builds (-DATTN_STAMPS, tools/attn_stamps.py): wave 0 of the first 4096 workgroups stamps its life
// (s_memtime): 0 entry, 1 loop entry, 2..7 the sections of key tile 1 (top, tiles stored + barriers, next loads
// issued, S MFMAs, softmax, PV MFMAs), 8 loop exit, 9 outputs stored
#ifdef ATTN_STAMPS
__device__ uint32_t g_attn_stamps[4096 * 16];
#define A_STAMP(k)                                                               \
  if (tid == 0 && lin_id < 4096) {                                               \
    __builtin_amdgcn_sched_barrier(0);                                           \
    g_attn_stamps[lin_id * 16 + (k)] = (uint32_t)__builtin_readcyclecounter();   \
    __builtin_amdgcn_sched_barrier(0);                                           \
  }
#define A_STAMP_T(k) if (t == 1) { A_STAMP(k) }
#else
#define A_STAMP(k)
#define A_STAMP_T(k)
#endif

// KV_ONCE: every sequence of the launch has one q tile (max_seqlen <= 128), so each K/V row is read exactly once:
// loaded non-temporally (the kernel alone, 130 560 tokens at L = 128, same qkv every launch: 162 -> 130 us; inside
// the engine, where qkv has just been written: 1.80 -> 1.77 ms per step).  With several q tiles per sequence the
// later ones find the rows in L2 and the hint costs 12 %, so those launches keep the cached form.  (One kernel
// with a per-workgroup choice compiles to the cached form for both: the optimiser merges the two loads.)
template <bool KV_ONCE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void attention_kernel(const bf16_t* __restrict__ qkv,
                                                        const int32_t* __restrict__ cu,
                                                        bf16_t* __restrict__ ctx, int hidden,
                                                        int q_tiles, int n_heads, int n_pairs, int ctx_tiled) {
  // DMA (launches whose sequences span several key tiles per q tile, i.e. not KV_ONCE): two K/V tile buffers filled by
  // LDS-DMA, [buffer][k | v][64 keys x 128 B]; otherwise one buffer filled through registers.  Measured per length
  // (bench.py, alternating processes, DMA against registers): L = 32 + 10 %, 64 and 128 equal, 256 - 2 %, 512 - 6.5 %.
  constexpr bool DMA = ATTN_DMA && !KV_ONCE;
  constexpr int NBUF = DMA ? ATTN_NBUF : 1;
  __shared__ __attribute__((aligned(16))) char kv_lds[NBUF * 2 * KT * 128];

  int seq, head, qt;
  if (n_pairs > 0) {
    // q_tiles >= 2, 1-D grid: the q tiles of a (sequence, head) get workgroup ids 8 apart — the same XCD (ids go
    // round-robin over the 8 XCDs), dispatched together, so the later ones' K/V reads can hit that L2
    const int id = blockIdx.x, blk = id / (8 * q_tiles), r = id - blk * 8 * q_tiles;
    const int pair = blk * 8 + (r & 7);
    if (pair >= n_pairs) return;
    qt = r >> 3;
    seq = pair / n_heads;
    head = pair - seq * n_heads;
  } else {
    seq = blockIdx.x / q_tiles; head = blockIdx.y; qt = blockIdx.x - seq * q_tiles;
  }
  const int beg = cu[seq];
  const int len = cu[seq + 1] - beg;
  const int q_tile0 = qt * 128;
  if (q_tile0 >= len) return;  // workgroup-uniform

  const int tid = threadIdx.x, lane = tid & 63;
#ifdef ATTN_STAMPS
  const int lin_id = blockIdx.y * gridDim.x + blockIdx.x;
#endif
  A_STAMP(0)
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const int ld = 3 * hidden;  // qkv row stride (elements)
  const bf16_t* qbase = qkv + (size_t)beg * ld + head * HD;
  const bf16_t* kbase = qbase + hidden;
  const bf16_t* vbase = qbase + 2 * hidden;

  const int q0 = q_tile0 + w * 32;
  const bool wave_live = q0 < len;  // wave-uniform

  // Q^T fragments (B operand of S^T): lane holds Q[q0 + l31][16 s + 8 h + j]
  opx8_t qf[4];
  {
    const int qr = min(q0 + l31, len - 1);
#pragma unroll
    for (int s = 0; s < 4; ++s)
      qf[s] = *reinterpret_cast<const opx8_t*>(qbase + (size_t)qr * ld + 16 * s + 8 * h);
  }

  // K/V tile t -> buffer t & 1 by LDS-DMA: wave w writes the 1-KiB images of rows 8 w .. 8 w + 7 and 32 + 8 w .. of k and
  // of v (lane-linear: lane l lands at row l >> 3, 16-byte slot l & 7), so the swizzle is applied to the SOURCE chunk
  // each lane fetches (both swizzles are XORs: slot = chunk ^ f(row) <=> chunk = slot ^ f(row)).  Keys past the end
  // fetch the sequence's last row: finite values, masked below.  No staging registers, no ds_write, one barrier per tile.
  auto stage_tile = [&](int kv0, int b) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int rl = w * 8 + (lane >> 3) + 32 * i;     // row inside the tile
      const int r = min(kv0 + rl, len - 1);
      const int slot = lane & 7;
      // Issued as inline assembly: through the builtin the compiler tracks "an LDS-DMA store to kv_lds is outstanding"
      // per ARRAY, not per buffer, and puts s_waitcnt vmcnt(0) in front of the first transposed V read of the tile
      // being computed — i.e. makes the tile wait for the NEXT tile's loads.  Hidden loads can only make its own
      // counted waits wait longer, never shorter (in-order counter); the waits these loads need are written out below.
      const uint32_t kd = (uint32_t)(uintptr_t)LDS_PTR(kv_lds + b * (2 * KT * 128) + (w + 4 * i) * 1024);
      const bf16_t* kg = kbase + (size_t)r * ld + (slot ^ ((rl >> 1) & 7)) * 8;
      const bf16_t* vg = vbase + (size_t)r * ld + (slot ^ (((rl >> 1) & 1) << 2)) * 8;
      // (m0 = the LDS address of the wave's 1-KiB image; the compiler reserves m0 and says so — nothing else in this
      // kernel uses it: no other LDS-DMA, no GWS, no movrel)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
      if (KV_ONCE) {
        asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off nt" ::"v"(kg), "s"(kd) : "memory", "m0");
        asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off nt" ::"v"(vg), "s"(kd + KT * 128) : "memory", "m0");
      } else {
        asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(kg), "s"(kd) : "memory", "m0");
        asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(vg), "s"(kd + KT * 128) : "memory", "m0");
      }
#pragma clang diagnostic pop
    }
  };
  // (registers) staging map: thread -> (row, chunk) for two 16-B chunks of K and of V per tile
  const int st_row = tid >> 3, st_chunk = tid & 7;  // rows st_row and st_row + 32
  uint4 kreg[2], vreg[2];
  auto load_tile = [&](int kv0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = kv0 + st_row + 32 * i;
      if (r < len) {
        if (KV_ONCE) {
          typedef __attribute__((ext_vector_type(4))) unsigned int u4_t;
          const u4_t kk = __builtin_nontemporal_load(reinterpret_cast<const u4_t*>(kbase + (size_t)r * ld + st_chunk * 8));
          const u4_t vv4 = __builtin_nontemporal_load(reinterpret_cast<const u4_t*>(vbase + (size_t)r * ld + st_chunk * 8));
          kreg[i] = make_uint4(kk[0], kk[1], kk[2], kk[3]);
          vreg[i] = make_uint4(vv4[0], vv4[1], vv4[2], vv4[3]);
        } else {
          kreg[i] = *reinterpret_cast<const uint4*>(kbase + (size_t)r * ld + st_chunk * 8);
          vreg[i] = *reinterpret_cast<const uint4*>(vbase + (size_t)r * ld + st_chunk * 8);
        }
      } else {  // keys past the end: zero (masked below; V must not carry NaN into 0 * V)
        kreg[i] = make_uint4(0, 0, 0, 0);
        vreg[i] = make_uint4(0, 0, 0, 0);
      }
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = st_row + 32 * i;
      *reinterpret_cast<uint4*>(kv_lds + r * 128 + swz128(r, st_chunk) * 16) = kreg[i];
      *reinterpret_cast<uint4*>(kv_lds + KT * 128 + r * 128 + swz_v(r, st_chunk) * 16) = vreg[i];
    }
  };

  f32x16_t o[2];  // O^T[d][q]: d = 32 db + (reg&3) + 8 (reg>>2) + 4 h, q on the lane
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;  // running max (log2 domain, shared by both halves), partial sum

  const int n_tiles = (len + KT - 1) / KT;
  if constexpr (DMA) {
    stage_tile(0, 0);
    if (NBUF > 2 && n_tiles > 1) stage_tile(KT, 1);
    // the Q fragments are pinned here: left alone, the compiler waits for them at their first use INSIDE the loop with
    // counted waits that, not knowing of the hidden LDS-DMA loads, would drain the next tile's loads in every iteration
#pragma unroll
    for (int s = 0; s < 4; ++s) asm volatile("" ::"v"(qf[s]));
  } else {
    load_tile(0);
  }
  A_STAMP(1)
  for (int t = 0; t < n_tiles; ++t) {
    A_STAMP_T(2)
    const char* k_lds = kv_lds;
    if constexpr (DMA) {
      // this wave's pieces of tile t have landed (4 DMA instructions per tile and wave; with three buffers tile t + 1 may
      // still be in flight) ...
      if (NBUF > 2 && t + 1 < n_tiles) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      // ... every wave's have, and tile t - 1 has been read by all (a bare barrier: __syncthreads()'s fence would drain
      // the loads in flight; a wave's LDS reads of tile t - 1 fed MFMAs it has already issued)
      __builtin_amdgcn_s_barrier();
      A_STAMP_T(3)
      if (t + NBUF - 1 < n_tiles) stage_tile((t + NBUF - 1) * KT, (t + NBUF - 1) % NBUF);
      k_lds = kv_lds + (t % NBUF) * (2 * KT * 128);
    } else {
      __syncthreads();  // previous tile's LDS reads done
      store_tile();
      __syncthreads();
      A_STAMP_T(3)
      if (t + 1 < n_tiles) load_tile((t + 1) * KT);
    }
    const char* v_lds = k_lds + KT * 128;
    A_STAMP_T(4)
    if (!wave_live) continue;

    const int kv0 = t * KT;
    // ---- S^T = K Q^T for the two 32-key blocks (first k-step accumulates onto an inline 0)
    f32x16_t s[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const int row = kb * 32 + l31;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const opx8_t kf = *reinterpret_cast<const opx8_t*>(
            k_lds + row * 128 + swz128(row, 2 * ks + h) * 16);
        if (ks == 0) {
          const f32x16_t zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          s[kb] = MFMA_32x32x16(kf, qf[ks], zero);
        } else {
          s[kb] = MFMA_32x32x16(kf, qf[ks], s[kb]);
        }
      }
    }
    A_STAMP_T(5)
    // ---- keys past the end exist only in the last tile (wave-uniform test): mask there only
    if (kv0 + KT > len) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kv0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          s[kb][r] = key < len ? s[kb][r] : -INFINITY;
        }
    }
    // ---- tile max on the raw scores (v_max3), then into the log2 domain
    float mx = fmaxf(s[0][0], s[1][0]);
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(fmaxf(mx, s[0][r]), s[1][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx * LOG2E);  // finite: every tile holds at least one valid key
    // the running max moves in the first tile or two and then rarely: rescale only when some row's did
    if (__any(m_new > m_run)) {
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);  // 1 exactly where the max stayed
      l_run *= alpha;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
    }
    m_run = m_new;

    // ---- P = exp2(S - m), O^T += V^T P^T
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      if (kb == 0) { A_STAMP_T(6) }
      float p[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        p[r] = __builtin_amdgcn_exp2f(fmaf(s[kb][r], LOG2E, -m_new));  // v_exp_f32: inputs <= 0
        l_run += p[r];
      }
#pragma unroll
      for (int ss = 0; ss < 2; ++ss) {
        opx8_t pf;
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[j] = (op_t)p[8 * ss + j];
        // element j of half h <-> key 32 kb + 16 ss + 8 (j>>2) + 4 h + (j&3)
        const int key0 = kb * 32 + 16 * ss + 4 * h;
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          // transposed read: 16-lane group g reads rows key0.. key0+3, cols 32 db + 16 (g&1) ..+15;
          // lane i = 4 qq + pp of the group supplies &V[key0 + qq][col0 + 4 pp]
          const int i16 = lane & 15, qq = i16 >> 2, pp = i16 & 3;
          const int col0 = 32 * db + 16 * ((lane >> 4) & 1) + 4 * pp;  // element column
          const int ra = key0 + qq, rb = key0 + 8 + qq;
          const s16x4_t va = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4_t*)(v_lds + ra * 128 +
                                                           swz_v(ra, col0 >> 3) * 16 + (col0 & 7) * 2));
          const s16x4_t vb = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4_t*)(v_lds + rb * 128 +
                                                           swz_v(rb, col0 >> 3) * 16 + (col0 & 7) * 2));
          s16x8_t vv;
          vv[0] = va[0]; vv[1] = va[1]; vv[2] = va[2]; vv[3] = va[3];
          vv[4] = vb[0]; vv[5] = vb[1]; vv[6] = vb[2]; vv[7] = vb[3];
          o[db] = MFMA_32x32x16(*reinterpret_cast<opx8_t*>(&vv), pf,
                                                          o[db]);
        }
      }
    }
    A_STAMP_T(7)
  }

  A_STAMP(8)
  // ---- output: each wave's 32 x 64 block through its 4-KiB quarter of the (now idle) K/V tiles, out as whole
  // 128-byte row segments (16 B per lane, 8 rows per instruction, non-temporal: ctx is read by the next kernel,
  // not by this one) instead of 16 stores of 8 bytes per lane
  __syncthreads();  // every wave is done reading the last key tile
  if (!wave_live) return;
  l_run += __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_run;
  char* ot = kv_lds + w * 4096;   // [32 q rows][128 B], 8-byte slots XOR-swizzled in pairs (no DMA in flight any more)
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int sl = (8 * db + 2 * g + h) ^ ((l31 & 7) << 1);
      *reinterpret_cast<uint2*>(ot + l31 * 128 + sl * 8) =
          make_uint2(pack_op2(o[db][4 * g] * inv, o[db][4 * g + 1] * inv),
                     pack_op2(o[db][4 * g + 2] * inv, o[db][4 * g + 3] * inv));
    }
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int row = p * 8 + (lane >> 3), c16 = lane & 7;
    const u32x4_t v = *reinterpret_cast<const u32x4_t*>(ot + row * 128 + ((c16 ^ (row & 7)) * 16));
    if (q0 + row < len) {
      // ctx_tiled: the out-projection's A operand as contiguous 128 x 64 blocks (a head is one block column)
      const size_t r = (size_t)(beg + q0 + row);
      const size_t at = ctx_tiled ? operand_tiled_offset(r, head * HD + c16 * 8, hidden) : r * hidden + head * HD + c16 * 8;
      __builtin_nontemporal_store(v, reinterpret_cast<u32x4_t*>(ctx + at));
    }
  }
  A_STAMP(9)
}
}  // namespace
#ifdef ATTN_STAMPS
#if !CLIP_OPERAND_F16
extern "C" int clip_debug_attn_stamps(uint32_t* host, int n_words) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_attn_stamps), (size_t)n_words * 4);
}
#endif
#endif
namespace {

// Last-layer form: only the pooled token of each sequence is consumed downstream
// (ref: sentclf/bert_model.py:52-58 gathers ONE hidden row per sentence; :234 uses row 0), so its
// attention output is the only one needed: one wave per (sequence, head), q row against all keys.
// HBM-bound (reads every K and V row once); softmax weights stay fp32.
// q_rows (optional): the focus tokens' q projections as a dense [n_seqs, hidden] tensor — the last layer
// then projects q for those rows only, and its all-token GEMM stops at K and V.
__global__ __launch_bounds__(64) void attention_focus_kernel(const bf16_t* __restrict__ qkv,
                                                             const int32_t* __restrict__ cu,
                                                             const int32_t* __restrict__ focus,
                                                             bf16_t* __restrict__ ctx_rows,
                                                             int hidden, const bf16_t* __restrict__ q_rows) {
  __shared__ float q_s[HD];
  __shared__ float p_s[512];
  const int seq = blockIdx.x, head = blockIdx.y, lane = threadIdx.x;
  const int beg = cu[seq], len = min(cu[seq + 1] - beg, 512);
  const int ld = 3 * hidden;
  const bf16_t* kbase = qkv + (size_t)beg * ld + hidden + head * HD;
  const bf16_t* vbase = kbase + hidden;
  q_s[lane] = op_to_f32(q_rows ? q_rows[(size_t)seq * hidden + head * HD + lane]
                                 : qkv[(size_t)focus[seq] * ld + head * HD + lane]);
  __syncthreads();
  // 8 lanes per key row: lane (g, c) takes dims 8c .. 8c+7 of key 8 it + g, so every load instruction covers 8
  // whole 128-byte rows (one lane per key meant 64 rows x 16 B per instruction); the 8 partial dot products meet
  // by three xor-shuffles
  const int g = lane >> 3, c = lane & 7;
  float qv[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) qv[e] = q_s[c * 8 + e];
  const int n_it = (len + 7) >> 3;
  // four key rows per lane in flight (a batch of 8 sentences is 96 waves on the whole chip: the loop is a chain of
  // load latencies, not bandwidth); the arithmetic per key, and its order, are unchanged
  for (int it0 = 0; it0 < n_it; it0 += 4) {
    uint4 kv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int key = (it0 + u) * 8 + g;
      kv[u] = key < len ? *reinterpret_cast<const uint4*>(kbase + (size_t)key * ld + c * 8) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int key = (it0 + u) * 8 + g;
      float d = 0.f;
      const uint32_t wds[4] = {kv[u].x, kv[u].y, kv[u].z, kv[u].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        d = fmaf(op2_lo(wds[e]), qv[2 * e], d);
        d = fmaf(op2_hi(wds[e]), qv[2 * e + 1], d);
      }
      d += __shfl_xor(d, 1, 64);
      d += __shfl_xor(d, 2, 64);
      d += __shfl_xor(d, 4, 64);
      if (c == 0 && key < len) p_s[key] = d * LOG2E;
    }
  }
  __syncthreads();
  float sc[8];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int key = lane + 64 * i;
    sc[i] = key < len ? p_s[key] : -INFINITY;
    mx = fmaxf(mx, sc[i]);
  }
  mx = wave_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int key = lane + 64 * i;
    const float p = key < len ? exp2f(sc[i] - mx) : 0.f;
    sum += p;
    if (key < len) p_s[key] = p;
  }
  sum = wave_sum(sum);
  __syncthreads();
  // o[dims 8c .. 8c+7] over this lane group's keys, then across the 8 groups (xor 8, 16, 32)
  float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int it0 = 0; it0 < n_it; it0 += 4) {
    uint4 vv[4];
    float pw[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int key = (it0 + u) * 8 + g;
      vv[u] = key < len ? *reinterpret_cast<const uint4*>(vbase + (size_t)key * ld + c * 8) : make_uint4(0, 0, 0, 0);
      pw[u] = key < len ? p_s[key] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if ((it0 + u) * 8 + g < len) {        // (a skipped key must not touch o: fmaf(0, 0, -0.0f) would flip a sign)
        const uint32_t wds[4] = {vv[u].x, vv[u].y, vv[u].z, vv[u].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o[2 * e] = fmaf(pw[u], op2_lo(wds[e]), o[2 * e]);
          o[2 * e + 1] = fmaf(pw[u], op2_hi(wds[e]), o[2 * e + 1]);
        }
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    o[e] += __shfl_xor(o[e], 8, 64);
    o[e] += __shfl_xor(o[e], 16, 64);
    o[e] += __shfl_xor(o[e], 32, 64);
  }
  if (g == 0) {
    const float inv = 1.0f / sum;
    *reinterpret_cast<uint4*>(ctx_rows + (size_t)seq * hidden + head * HD + c * 8) =
        make_uint4(pack_op2(o[0] * inv, o[1] * inv), pack_op2(o[2] * inv, o[3] * inv),
                   pack_op2(o[4] * inv, o[5] * inv), pack_op2(o[6] * inv, o[7] * inv));
  }
}

// fp32 check mode: the same softmax(q k^T) v on fp32 q/k/v with fp32 arithmetic throughout (no
// MFMA: there is no bf16 rounding to reproduce, and check mode is not timed).  One lane per q row,
// 128 q rows per workgroup; K/V tiles of 64 keys staged in LDS and read as wave-wide broadcasts;
// online softmax over chunks of 16 keys.
__global__ __launch_bounds__(128) void attention_f32_kernel(const float* __restrict__ qkv,
                                                            const int32_t* __restrict__ cu,
                                                            float* __restrict__ ctx, int hidden,
                                                            int q_tiles) {
  __shared__ __attribute__((aligned(16))) float k_s[KT * HD];
  __shared__ __attribute__((aligned(16))) float v_s[KT * HD];
  const int seq = blockIdx.x / q_tiles, head = blockIdx.y;
  const int beg = cu[seq];
  const int len = cu[seq + 1] - beg;
  const int q_tile0 = (blockIdx.x - seq * q_tiles) * 128;
  if (q_tile0 >= len) return;  // workgroup-uniform
  const int tid = threadIdx.x;
  const int ld = 3 * hidden;
  const float* qbase = qkv + (size_t)beg * ld + head * HD;
  const float* kbase = qbase + hidden;
  const float* vbase = qbase + 2 * hidden;
  const int qi = q_tile0 + tid;

  float q[HD], o[HD];
  {
    const float4* qr = reinterpret_cast<const float4*>(qbase + (size_t)min(qi, len - 1) * ld);
#pragma unroll
    for (int c = 0; c < HD / 4; ++c) {
      const float4 v = qr[c];
      q[4 * c] = v.x; q[4 * c + 1] = v.y; q[4 * c + 2] = v.z; q[4 * c + 3] = v.w;
    }
#pragma unroll
    for (int d = 0; d < HD; ++d) o[d] = 0.f;
  }
  float m_run = -INFINITY, l_run = 0.f;

  for (int kv0 = 0; kv0 < len; kv0 += KT) {
    __syncthreads();  // previous tile's reads done
#pragma unroll
    for (int i = 0; i < KT * HD / 4 / 128; ++i) {
      const int idx = tid + 128 * i, r = idx >> 4, c = idx & 15;
      float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
      if (kv0 + r < len) {
        kv = *reinterpret_cast<const float4*>(kbase + (size_t)(kv0 + r) * ld + 4 * c);
        vv = *reinterpret_cast<const float4*>(vbase + (size_t)(kv0 + r) * ld + 4 * c);
      }
      *reinterpret_cast<float4*>(k_s + r * HD + 4 * c) = kv;
      *reinterpret_cast<float4*>(v_s + r * HD + 4 * c) = vv;
    }
    __syncthreads();
    const int nk = min(KT, len - kv0);
    for (int c0 = 0; c0 < nk; c0 += 16) {  // workgroup-uniform trip count
      float s[16];
      float mx = -INFINITY;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const float4* kr = reinterpret_cast<const float4*>(k_s + (c0 + j) * HD);
        float d = 0.f;
#pragma unroll
        for (int c = 0; c < HD / 4; ++c) {
          const float4 kv = kr[c];
          d = fmaf(q[4 * c], kv.x, d); d = fmaf(q[4 * c + 1], kv.y, d);
          d = fmaf(q[4 * c + 2], kv.z, d); d = fmaf(q[4 * c + 3], kv.w, d);
        }
        s[j] = c0 + j < nk ? d : -INFINITY;
        mx = fmaxf(mx, s[j]);
      }
      const float m_new = fmaxf(m_run, mx);  // finite: key c0 exists
      const float alpha = expf(m_run - m_new);
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int d = 0; d < HD; ++d) o[d] *= alpha;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const float p = expf(s[j] - m_new);  // 0 for masked keys
        l_run += p;
        const float4* vr = reinterpret_cast<const float4*>(v_s + (c0 + j) * HD);
#pragma unroll
        for (int c = 0; c < HD / 4; ++c) {
          const float4 vv = vr[c];
          o[4 * c] = fmaf(p, vv.x, o[4 * c]); o[4 * c + 1] = fmaf(p, vv.y, o[4 * c + 1]);
          o[4 * c + 2] = fmaf(p, vv.z, o[4 * c + 2]); o[4 * c + 3] = fmaf(p, vv.w, o[4 * c + 3]);
        }
      }
    }
  }
  if (qi < len) {
    const float inv = 1.0f / l_run;
    float4* dst = reinterpret_cast<float4*>(ctx + (size_t)(beg + qi) * hidden + head * HD);
#pragma unroll
    for (int c = 0; c < HD / 4; ++c)
      dst[c] = make_float4(o[4 * c] * inv, o[4 * c + 1] * inv, o[4 * c + 2] * inv, o[4 * c + 3] * inv);
  }
}

}  // namespace

#if !CLIP_OPERAND_F16
hipError_t launch_attention_f32(const float* qkv, const int32_t* cu_seqlens, float* ctx, int n_seqs,
                                int max_seqlen, int n_heads, int hidden, hipStream_t st) {
  if (n_seqs <= 0) return hipSuccess;
  if (hidden != n_heads * HD || max_seqlen <= 0) return hipErrorInvalidValue;
  const int q_tiles = (max_seqlen + 127) / 128;
  hipLaunchKernelGGL(attention_f32_kernel, dim3(q_tiles * n_seqs, n_heads), dim3(128), 0, st, qkv,
                     cu_seqlens, ctx, hidden, q_tiles);
  return hipGetLastError();
}

#endif

hipError_t KSET::launch_attention_focus(const bf16_t* qkv, const int32_t* cu_seqlens,
                                  const int32_t* focus, bf16_t* ctx_rows, int n_seqs, int n_heads,
                                  int hidden, hipStream_t st, const bf16_t* q_rows) {
  if (n_seqs <= 0) return hipSuccess;
  if (hidden != n_heads * HD) return hipErrorInvalidValue;
  hipLaunchKernelGGL(attention_focus_kernel, dim3(n_seqs, n_heads), dim3(64), 0, st, qkv, cu_seqlens,
                     focus, ctx_rows, hidden, q_rows);
  return hipGetLastError();
}

hipError_t KSET::launch_attention(const bf16_t* qkv, const int32_t* cu_seqlens, bf16_t* ctx, int n_seqs,
                            int max_seqlen, int n_heads, int hidden, hipStream_t st, bool ctx_tiled) {
  const int ct = ctx_tiled ? 1 : 0;
  if (n_seqs <= 0) return hipSuccess;
  if (hidden != n_heads * HD || max_seqlen <= 0) return hipErrorInvalidValue;
  const int q_tiles = (max_seqlen + 127) / 128;
#if ATTN_PAIR   // bench.py in alternating processes: attention 1.956 -> 1.903 ms per step (3 of 3 pairs of runs)
  if (q_tiles >= 2) {
    const int n_pairs = n_seqs * n_heads;
    hipLaunchKernelGGL(attention_kernel<false>, dim3((n_pairs + 7) / 8 * 8 * q_tiles), dim3(256), 0, st, qkv, cu_seqlens, ctx, hidden,
                       q_tiles, n_heads, n_pairs, ct);
    return hipGetLastError();
  }
#endif
  if (q_tiles == 1)
    hipLaunchKernelGGL(attention_kernel<true>, dim3(n_seqs, n_heads), dim3(256), 0, st, qkv, cu_seqlens, ctx, hidden,
                       q_tiles, n_heads, 0, ct);
  else
    hipLaunchKernelGGL(attention_kernel<false>, dim3(q_tiles * n_seqs, n_heads), dim3(256), 0, st, qkv,
                       cu_seqlens, ctx, hidden, q_tiles, n_heads, 0, ct);
  return hipGetLastError();
}
