// Internal launcher interface between the kernel files and the C-ABI (engine.hip).
#pragma once
#include "common.h"

// EPI_BIAS_LNRES: out_f32 = A W^T + bias + LayerNorm(pre)[m, n], the residual operand recomputed in
// fp32 from the producer's pre-LayerNorm row (`res` = pre, fp32), its row statistics and affine —
// what the reference adds is the fp32 LayerNorm output (modeling_bert.py:292,350), not a bf16 copy.
// EPI_BIAS_LNOUT (ping-pong kernel only, N = the whole row): the LayerNorm itself is fused,
//   v = A W^T + bias + res16[m, n];  x_bf16, xres_f16 = LayerNorm(v) over the N columns of row m.
// The N / 256 workgroups that compute the column tiles of one 256-row panel run side by side (a
// fixed group of neighbouring workgroups, each keeping its column tile across panels, so the A
// panel is fetched once into their XCD's L2) and exchange per-row (sum, sum of squares) through
// device memory before normalising their accumulators.  No pre-LayerNorm tensor reaches HBM and
// there is no LayerNorm kernel.
// EPI_BIAS_RES16: out_f32 = A W^T + bias + fp16 residual rows (optionally picked by res_index).
enum { EPI_BIAS = 0, EPI_BIAS_GELU = 1, EPI_BIAS_RES = 2, EPI_BIAS_TANH = 3, EPI_BIAS_LNRES = 4,
       EPI_BIAS_LNOUT = 5, EPI_BIAS_RES16 = 6 };

struct LnOutArgs {         // EPI_BIAS_LNOUT
  uint16_t* xres_out;      // [M, N] fp16 LayerNorm output, TILED (xres_tiled_offset); may alias a tiled res
  bool res_tiled;          // res is tiled like xres_out (false: row-major fp16 with row stride ldres)
  float* part;             // [tiles_m * tiles_n * 256] (sum, sumsq) pairs, scratch
  uint32_t* flags;         // [tiles_m * tiles_n] last epoch published
  uint32_t epoch;          // differs from every earlier launch on these flags
  float eps;
  int32_t* err;            // |= 2 when a partner tile's statistics did not arrive (bounded spin); sticky:
                           // while bit 1 is set no tile waits for its partners any more
  uint32_t test_drop = 0;  // diagnostic (CLIP_AMD_TEST_DROP_PUBLISH=1): tile 1 never publishes its flag
  bool res_f32 = false;    // fp16 operand set only (CLIP_PRECISION_F16R): res and xres_out are FP32, tiled by
                           // xres32_tiled_offset (res_tiled must be set) — the residual stream never leaves fp32
  int small_ok = 0;        // 1: `flags` holds (M / 64) * (N / 256) entries (4x the ping-pong kernel's need): calls whose whole
                           // grid of 64-row tiles is resident at once may run on gemm4.hip's kernels (same bits);
                           // 2: and `part_tag` is set: the 64 x 64-tile form may run too
  unsigned long long* part_tag = nullptr;   // [(M / 256) * (N / 64) * 256][2] words {sum | epoch << 32} {sumsq | epoch << 32}:
                           // gemm4.hip's 64 x 64-tile form tags its partials and uses no flags.  Written by that kernel
                           // only; no word may hold a FUTURE epoch in its upper half (zero it once, and at an epoch wrap)
};

struct GemmArgs {
  const void* A;      // [M, K] activations (bf16; fp32 when f32_in), row stride lda (elements)
  const void* W;      // [N, K] weight, row stride ldw
  const float* bias;  // [N]
  const void* res;    // [M, N] residual (EPI_BIAS_RES), row stride ldres
  void* out;          // [M, N] bf16 or f32, row stride ldo
  int M, N, K, lda, ldw, ldo, ldres;
  int epi;
  bool out_f32;
  int tile = 0;  // 0 auto; 128 / 256: one tile per workgroup; 2: ping-pong persistent (gemm8.hip)
  bool f32_in = false;  // fp32 check mode: fp32 operands/residual, exact-fp32 MFMA (after `tile`: positional inits)
  // EPI_BIAS_LNRES: res = fp32 pre-LayerNorm rows [., N] (row stride ldres), and:
  const float* ln_stats = nullptr;   // [rows] (mean, rstd) pairs
  const float* ln_gamma = nullptr;   // [N]
  const float* ln_beta = nullptr;    // [N]
  const int32_t* res_index = nullptr;  // optional: residual row of output row m is res_index[m] (one-tile kernel only)
  const int32_t* a_index = nullptr;    // optional, M <= 32 only (gemm_rows16_kernel): row m of A is A[a_index[m]] (no gather pass)
  int res_tiles_n = 0;                 // EPI_BIAS_RES16: > 0 = res is tiled (xres_tiled_offset) with that many column tiles;
                                       // < 0 = res is FP32, tiled by xres32_tiled_offset with -res_tiles_n column tiles
  LnOutArgs lnout = {nullptr, false, nullptr, nullptr, 0, 0.f, nullptr, 0};  // EPI_BIAS_LNOUT: res = fp16 rows, ln_gamma/ln_beta = this LayerNorm's affine
  // Ping-pong kernel only (engine-internal operands: attention ctx -> attn-out, FFN-up -> FFN-down): the 16-bit
  // matrix is stored as contiguous 128 x 64 blocks, element (r, c) of an [M, C] matrix at
  //   (((r >> 7) * (C >> 6) + (c >> 6)) << 13) + ((r & 127) << 6) + (c & 63)      (operand_tiled_offset)
  // so that each LDS-DMA piece (8 rows x 128 B) is 1 KiB contiguous instead of 8 segments a row apart.
  bool a_tiled = false;    // A is stored that way (lda must equal K)
  bool out_tiled = false;  // the 16-bit output is stored that way (ldo must equal N; EPI_BIAS / EPI_BIAS_GELU)
  bool w_tiled = false;    // W is stored that way (a copy made once when the weights are loaded; ldw must equal K)
  const void* W_rowmajor = nullptr;   // with w_tiled: the same weight row-major (row stride K), for the small-M kernel
};
// ---- launchers that do not depend on the 16-bit operand type (defined once, in the bf16 compile)
const char* gemm_check(const GemmArgs& g);
bool gemm_persist_ok(const GemmArgs& g);
// fp32 check mode of the attention: qkv / ctx fp32, fp32 arithmetic throughout
hipError_t launch_attention_f32(const float* qkv, const int32_t* cu_seqlens, float* ctx, int n_seqs,
                                int max_seqlen, int n_heads, int hidden, hipStream_t st);
// dst[i] = src[rows[i]]  (16-bit rows of H, or fp32)
hipError_t launch_gather_rows(const void* src, const int32_t* rows, void* dst, int n, int H,
                              hipStream_t st, bool f32 = false);
// focus[s] = packed token index of the (n_ctx+1)-th sep of sequence s (or of its first token when
// n_ctx < 0: the [CLS] row); err[0] |= 1 if any sequence does not hold exactly 2*n_ctx+1 seps
hipError_t launch_focus_index(const int32_t* ids, const int32_t* cu_seqlens, int n_seqs,
                              int sep_id, int n_ctx, int32_t* focus, int32_t* err,
                              hipStream_t st);
hipError_t launch_prepare_packed(const int32_t* ids, const int32_t* cu_seqlens, int n_seqs,
                                 int sep_id, int n_ctx, int32_t* type_ids, int32_t* pos_ids,
                                 int32_t* focus, int32_t* err, hipStream_t st);
// fused scoring epilogue (ref: apply_bert_unlabeled.py:111-124, test_load.py:37-38)
hipError_t launch_epilogue(const float* logits, const double* label_thresh, float any_thresh,
                           int n, int L, float* probs, float* max_score, uint8_t* labels,
                           uint8_t* keep, hipStream_t st);
// mean over all elements of BCE-with-logits (ref: bert_model.py:97)
hipError_t launch_bce_loss(const float* logits, const float* targets, int count, float* loss,
                           hipStream_t st);

// CrossEntropyLoss pieces (ref: bert_model.py:177-182): row_loss[i] = logsumexp(scores[i, :V]) - scores[i, labels[i]]
// (copy: optional dense [n, V] copy of the rows), and loss[0] (+)= mean(row_loss)
hipError_t launch_ce_rows(const float* scores, int ld, int V, const int32_t* labels, int n,
                          float* row_loss, float* copy, hipStream_t st);
hipError_t launch_mean_loss(const float* row_loss, int n, float* loss, int accumulate, hipStream_t st);

// ---- launchers of the kernels that touch 16-bit operands: one set per operand type.  Every kernel
// file is compiled twice (common.h: CLIP_OPERAND_F16) and defines KSET::launch_* ; the engine picks
// the set by its precision (k_bf16 also holds the fp32 check-mode instantiations).
#define CLIP_KERNEL_SET_MEMBERS                                                                      \
  static hipError_t launch_gemm(const GemmArgs& g, hipStream_t st);                                  \
  static hipError_t launch_gemm8(const GemmArgs& g, hipStream_t st); /* ping-pong schedule */        \
  /* EPI_BIAS_LNOUT on 64 x 256 tiles for small M (gemm4.hip); hipErrorNotReady = not its case */   \
  static hipError_t launch_gemm4_lnout(const GemmArgs& g, hipStream_t st);                           \
  /* varlen self-attention over packed sequences; qkv [T, 3H] (q | k | v, heads of 64), q pre-scaled  \
     by 1/sqrt(64); ctx [T, H].  One workgroup per (128-row q tile, sequence, head). */              \
  static hipError_t launch_attention(const bf16_t* qkv, const int32_t* cu_seqlens, bf16_t* ctx,      \
                                     int n_seqs, int max_seqlen, int n_heads, int hidden,            \
                                     hipStream_t st, bool ctx_tiled = false /* GemmArgs::a_tiled */); \
  /* one q row per sequence (token index focus[s], or the dense q_rows [n_seqs, hidden]):            \
     ctx_rows [n_seqs, H]; sequences up to 512 keys */                                               \
  static hipError_t launch_attention_focus(const bf16_t* qkv, const int32_t* cu_seqlens,             \
                                           const int32_t* focus, bf16_t* ctx_rows, int n_seqs,       \
                                           int n_heads, int hidden, hipStream_t st,                  \
                                           const bf16_t* q_rows = nullptr);                          \
  /* out[t] = LayerNorm(word[ids[t]] + pos[pos_ids[t]] + type[type_ids[t]]); res16: optional fp16    \
     copy of the output */                                                                           \
  static hipError_t launch_embed_ln(const int32_t* ids, const int32_t* type_ids,                     \
                                    const int32_t* pos_ids, const float* word, const float* pos,     \
                                    const float* type, const float* gamma, const float* beta,        \
                                    float eps, void* out, int T, int H, int vocab, int max_pos,      \
                                    int type_vocab, hipStream_t st, bool out_f32 = false,            \
                                    float* pre_out = nullptr, float* stats = nullptr,                \
                                    uint16_t* res16 = nullptr);                                      \
  /* out[t] = LayerNorm(x_f32[t]); stats: optional [T] (mean, rstd) pairs out */                     \
  static hipError_t launch_layernorm(const float* x, const float* gamma, const float* beta,          \
                                     float eps, void* out, int T, int H, hipStream_t st,             \
                                     bool out_f32 = false, float* stats = nullptr,                   \
                                     uint16_t* res16 = nullptr);                                     \
  /* logits[i] = Wc . x[rows ? rows[i] : i] + bc     (Wc [L,H] f32, x 16-bit or fp32) */             \
  /* last LayerNorm of the trimmed forward + classifier + the guard of launch_poison_logits, one kernel */ \
  static hipError_t launch_ln_head(const float* x, const float* gamma, const float* beta, float eps, \
                                   int n, int H, const float* Wc, const float* bc, int L,            \
                                   float* logits, int32_t* err, hipStream_t st);                     \
  static hipError_t launch_classifier(const void* x, const int32_t* rows, const float* Wc,           \
                                      const float* bc, float* logits, int n, int H, int L,           \
                                      hipStream_t st, bool f32 = false);
struct k_bf16 { CLIP_KERNEL_SET_MEMBERS };
struct k_f16 { CLIP_KERNEL_SET_MEMBERS };
// err |= 4 unless cu_seqlens[0] == 0, every row has 1..max_seqlen tokens and cu_seqlens[n] == total_tokens
hipError_t launch_check_rows(const int32_t* cu_seqlens, int n_seqs, int total_tokens, int max_seqlen,
                             int32_t* err, hipStream_t st);
// padded [B, S] int64 tensors -> packed rows (rowwise.hip): lengths + prefix sums into cu[0..B], then the compaction
// (host_out: optional mapped host words {value, tag} a spinning host reads instead of copying back)
// (max_len, err: a row of < 1 or > max_len kept tokens sets bit 2 of *err, the row check of the packed forward)
hipError_t launch_pack_lengths(const int64_t* mask, int B, int S, int max_len, int32_t* cu, int32_t* err,
                               int32_t* host_out, int32_t tag, hipStream_t st);
hipError_t launch_publish_word(const int32_t* word, int32_t* host_out, int32_t tag, hipStream_t st);
hipError_t launch_pack_tokens(const int64_t* ids, const int64_t* mask, const int64_t* tt, const int64_t* pos,
                              long pos_row_stride, int B, int S, const int32_t* cu, int32_t* out_ids,
                              int32_t* out_tt, int32_t* out_pos, int sep_id, int n_ctx, int32_t* focus, int32_t* err,
                              hipStream_t st);   // focus != nullptr: launch_focus_index's result on the same pass
// logits[0..count) = NaN when *err != 0
hipError_t launch_poison_logits(int32_t* err, float* logits, int count, hipStream_t st);

// Per-device one-time launch set-up (function attributes, CU count): a process may hold engines on
// several GPUs, and two host threads may make their first call at once.
#include <mutex>
struct DeviceOnce {
  static constexpr int MAX_DEV = 64;
  std::mutex mu;
  bool done[MAX_DEV] = {};
  int value[MAX_DEV] = {};
  // runs init(dev, &value) once per device; returns the device's value through *out
  template <typename F>
  hipError_t get(F&& init, int* out) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= MAX_DEV) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lk(mu);
    if (!done[dev]) {
      if ((e = init(dev, &value[dev])) != hipSuccess) return e;
      done[dev] = true;
    }
    *out = value[dev];
    return hipSuccess;
  }
};
