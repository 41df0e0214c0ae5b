// C ABI of libclip_amd.so (include/clip_amd.h): engine object = device weights + activation
// workspace; clip_forward_packed chains the kernels of one BERT(+Context) forward on one stream.
#include <algorithm>
#include <chrono>
#include <map>
#include <atomic>
#include <mutex>
#include <thread>
#include <type_traits>
#include <string>
#include <vector>
#include <cstdio>
#include <cstdarg>
#include <cstring>
#include <cmath>

#include "common.h"
#include "kernels.h"
#include "../../include/clip_amd.h"

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIP_TRY(expr)                                                                  \
  do {                                                                                 \
    hipError_t _e = (expr);                                                            \
    if (_e != hipSuccess)                                                              \
      return fail(CLIP_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                  __FILE__, __LINE__);                                                 \
  } while (0)

inline uint16_t host_f32_to_bf16(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // quiet NaN
  return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

struct LayerW {
  void *wqkv, *wo, *wi, *wo2;  // bf16, or fp32 in check mode
  // 16-bit settings: the same weights once more as contiguous 128 x 64 blocks (operand_tiled_offset), what the
  // ping-pong kernel reads (GemmArgs::w_tiled); the row-major ones serve every other kernel.  Null where the shape
  // does not divide (then the row-major weight is used everywhere).
  void *wqkv_blk = nullptr, *wo_blk = nullptr, *wo2_blk = nullptr;
  float *bqkv, *bo, *bi, *bo2, *ln1g, *ln1b, *ln2g, *ln2b;
};

}  // namespace

struct clip_engine {
  clip_config cfg;
  int device;
  bool finalized = false;
  std::map<std::string, std::vector<float>> host_w;  // until finalize
  std::vector<void*> allocs;
  // weights
  float *word = nullptr, *pos = nullptr, *type = nullptr, *embg = nullptr, *embb = nullptr;
  std::vector<LayerW> layers;
  void* wpool = nullptr;
  float *bpool = nullptr, *wcls = nullptr, *bcls = nullptr;
  // numerics setting of the SAME kernels (clip_set_precision): false = bf16 operands / fp32
  // accumulation, true = fp32 check mode (fp32 weights, activations and MFMA)
  bool f32 = false;
  bool f16 = false;  // CLIP_PRECISION_F16 / F16R: the k_f16 kernel set (fp16 operands, same MFMA rate, 11-bit significand)
  bool res32 = false;  // CLIP_PRECISION_F16R: fp32 residual stream (fp32 pre-LayerNorm rows + (mean, rstd); no LayerNorm fusion)
  // workspace (allocated by clip_finalize_weights, element size by precision)
  void *x = nullptr, *qkv = nullptr, *ctx = nullptr, *mid = nullptr, *rows = nullptr,
       *pooled = nullptr, *ctx_rows = nullptr, *mid_rows = nullptr, *xf_rows = nullptr, *q_rows = nullptr;
  // fp32 pre-LayerNorm rows and their (mean, rstd), double-buffered: a GEMM writes pre[b ^ 1] while
  // its epilogue re-derives the residual from pre[b] / stats[b] (EPI_BIAS_LNRES)
  float *pre[2] = {nullptr, nullptr}, *stats[2] = {nullptr, nullptr};
  float *pre_rows[2] = {nullptr, nullptr}, *stats_rows[2] = {nullptr, nullptr};
  // fused-LayerNorm data flow (EPI_BIAS_LNOUT; bf16 setting, hidden a multiple of 256): the N = hidden
  // GEMMs write the LayerNorm output themselves — x (bf16 operand) and xres (fp16 residual copy) —
  // the column-tile workgroups of a row panel exchanging per-row sums through lnpart / lnflags
  bool fuse_ln = true;
  uint16_t *xres = nullptr;     // tiled (xres_tiled_offset): written and re-read in place by the fused GEMMs
  uint16_t *xres_rm = nullptr;  // row-major: the embedding LayerNorm's copy, residual of the first fused GEMM
  float* xres32 = nullptr;      // F16R: the FP32 residual stream, tiled (xres32_tiled_offset), rewritten in place
  float* lnpart = nullptr;
  unsigned long long* lnpart_tag = nullptr;   // epoch-tagged partials of the 64 x 64-tile small-M kernel (no flags there)
  size_t lnpart_tag_bytes = 0;
  uint32_t* lnflags = nullptr;
  int32_t* dev_err = nullptr;
  // packed rows of clip_forward_padded: ids / type ids / position ids [max_tokens], cu [max_seqs + 1], focus [max_seqs]
  int32_t *pk_ids = nullptr, *pk_tt = nullptr, *pk_pos = nullptr, *pk_cu = nullptr, *pk_focus = nullptr;
  // four words of host memory mapped into the device: {token count, tag, error word, tag}, written by the last
  // kernels of clip_forward_padded's two stages; the host spins on the tags (no copy, no stream wait)
  volatile int32_t* host_words = nullptr;
  int32_t* host_words_dev = nullptr;
  int32_t host_tag = 0;
  uint32_t epoch = 0;
  // LM prediction head of BertContextMLMSwitch (bert_model.py:113,165; optional weights mlm_predictions.*)
  void *w_mlm_t = nullptr, *w_mlm_dec = nullptr;  // transform dense [H,H]; decoder [vocab_pad,H] (zero rows past vocab)
  float *b_mlm_t = nullptr, *g_mlm = nullptr, *be_mlm = nullptr, *b_mlm_dec = nullptr;  // decoder bias: -1e30 past vocab
  int vocab_pad = 0, mlm_vocab = 0;
  bool last_full = false;  // the latest forward left every token's last hidden state in x (no last-layer trimming)
  int last_tokens = 0;
  void *mlm_rows = nullptr, *mlm_t = nullptr;  // [cap, H] gathered hidden rows / transformed rows
  float *mlm_pre = nullptr, *mlm_scores = nullptr, *mlm_row_loss = nullptr;
  int mlm_cap = 0;
  hipStream_t last_stream = nullptr;  // stream of the latest forward: clip_poll_errors reads behind it
  bool test_drop = false;             // CLIP_AMD_TEST_DROP_PUBLISH=1 (tests): one fused-LayerNorm tile never publishes
  size_t ln_tiles = 0;                // entries of lnflags
  int small_m = 2;        // fused-LayerNorm GEMMs of small calls on gemm4.hip's tiles (CLIP_AMD_SMALL_M: 0 off, 1 64 x 256 only, 2 also 64 x 64)
  bool trim_last = true;  // last layer: only the pooled row of each sequence past the K/V projection
  // HIP-event profiling of kernel kinds inside clip_forward_packed (clip_profile_*)
  uint32_t prof_mask = 0;
  std::vector<hipEvent_t> ev_pool;
  size_t ev_used = 0;
  struct Span { int kind; hipEvent_t a, b; };
  std::vector<Span> spans;
};

namespace {

int dev_alloc(clip_engine* e, void** p, size_t bytes) {
  HIP_TRY(hipMalloc(p, bytes));
  e->allocs.push_back(*p);
  return CLIP_OK;
}

int upload_f32(clip_engine* e, const std::vector<float>& h, float** d) {
  int rc = dev_alloc(e, (void**)d, h.size() * 4);
  if (rc) return rc;
  HIP_TRY(hipMemcpy(*d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  return CLIP_OK;
}

// fn(begin, end) over [0, n) on up to 8 host threads (the weight roundings: 110 M elements per engine build)
template <class F>
void parallel_ranges(size_t n, F fn) {
  const unsigned hw = std::thread::hardware_concurrency();
  const size_t nt = n < (1u << 20) ? 1 : std::min<size_t>(8, hw ? hw : 1);
  if (nt <= 1) { fn((size_t)0, n); return; }
  std::vector<std::thread> th;
  const size_t per = (n + nt - 1) / nt;
  for (size_t t = 0; t < nt; ++t) {
    const size_t a = t * per, b = std::min(n, a + per);
    if (a < b) th.emplace_back([=] { fn(a, b); });
  }
  for (auto& x : th) x.join();
}

int upload_bf16(clip_engine* e, const float* h, size_t n, bf16_t** d, float scale_first = 1.f,
                size_t n_first = 0) {
  std::vector<uint16_t> tmp(n);
  parallel_ranges(n, [&](size_t a, size_t b) {
    for (size_t i = a; i < b; ++i) tmp[i] = host_f32_to_bf16(i < n_first ? h[i] * scale_first : h[i]);
  });
  int rc = dev_alloc(e, (void**)d, n * 2);
  if (rc) return rc;
  HIP_TRY(hipMemcpy(*d, tmp.data(), n * 2, hipMemcpyHostToDevice));
  return CLIP_OK;
}

int upload_f16(clip_engine* e, const float* h, size_t n, bf16_t** d, float scale_first, size_t n_first) {
  std::vector<uint16_t> tmp(n);
  std::atomic<size_t> bad{n};   // index of a weight outside fp16's range (n: none)
  parallel_ranges(n, [&](size_t a, size_t b) {
    for (size_t i = a; i < b; ++i) {
      const float v = i < n_first ? h[i] * scale_first : h[i];
      if (!(std::fabs(v) <= 65504.f)) { bad.store(i); return; }
      const _Float16 hv = (_Float16)v;  // round to nearest even
      memcpy(&tmp[i], &hv, 2);
    }
  });
  if (bad.load() < n)
    return fail(CLIP_ERR_ARG, "a weight (%g) is outside fp16's range: use CLIP_PRECISION_BF16 for this checkpoint",
                (double)h[bad.load()]);
  int rc = dev_alloc(e, (void**)d, n * 2);
  if (rc) return rc;
  HIP_TRY(hipMemcpy(*d, tmp.data(), n * 2, hipMemcpyHostToDevice));
  return CLIP_OK;
}

// a GEMM weight: bf16 / fp16 (rounded once, here) or fp32 as loaded; the first n_first elements scaled
int upload_w(clip_engine* e, const float* h, size_t n, void** d, float scale_first = 1.f,
             size_t n_first = 0) {
  if (e->f16) return upload_f16(e, h, n, reinterpret_cast<bf16_t**>(d), scale_first, n_first);
  if (!e->f32) return upload_bf16(e, h, n, reinterpret_cast<bf16_t**>(d), scale_first, n_first);
  std::vector<float> tmp(h, h + n);
  for (size_t i = 0; i < n_first; ++i) tmp[i] *= scale_first;
  return upload_f32(e, tmp, reinterpret_cast<float**>(d));
}

// dst = src ([N, K] 16-bit, row-major) as contiguous 128 x 64 blocks; once per weight, at load
__global__ void weight_blocks_kernel(const uint16_t* __restrict__ src, uint16_t* __restrict__ dst, int N, int K) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // one 16-byte chunk (8 elements) each
  const size_t n_chunks = (size_t)N * K / 8;
  if (i >= n_chunks) return;
  const size_t r = i / (K / 8);
  const int c = (int)(i - r * (K / 8)) * 8;
  *reinterpret_cast<uint4*>(dst + operand_tiled_offset(r, c, K)) = *reinterpret_cast<const uint4*>(src + r * K + c);
}

int weight_blocks(clip_engine* e, const void* row_major, size_t N, size_t K, void** blk) {
  *blk = nullptr;
  if (e->f32 || N % 128 || K % 64) return CLIP_OK;
  int rc = dev_alloc(e, blk, N * K * 2);
  if (rc) return rc;
  const size_t n_chunks = N * K / 8;
  hipLaunchKernelGGL(weight_blocks_kernel, dim3((unsigned)((n_chunks + 255) / 256)), dim3(256), 0, 0,
                     static_cast<const uint16_t*>(row_major), static_cast<uint16_t*>(*blk), (int)N, (int)K);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipDeviceSynchronize());
  return CLIP_OK;
}

int alloc_workspace(clip_engine* e) {
  if (e->stats_rows[1]) return CLIP_OK;  // a finalize that failed on a missing weight is retried
  const clip_config* c = &e->cfg;
  const size_t Tp = ((size_t)c->max_tokens + 255) / 256 * 256, H = c->hidden_size,
               I = c->intermediate_size, S = ((size_t)c->max_seqs + 127) / 128 * 128;
  const size_t es = e->f32 ? 4 : 2;
  int rc = 0;
  rc = rc ? rc : dev_alloc(e, &e->x, Tp * H * es);
  rc = rc ? rc : dev_alloc(e, &e->qkv, Tp * 3 * H * es);
  rc = rc ? rc : dev_alloc(e, &e->ctx, Tp * H * es);
  rc = rc ? rc : dev_alloc(e, &e->mid, Tp * I * es);
  for (int b = 0; b < 2; ++b) {
    rc = rc ? rc : dev_alloc(e, (void**)&e->pre[b], Tp * H * 4);
    rc = rc ? rc : dev_alloc(e, (void**)&e->stats[b], Tp * 8);
    rc = rc ? rc : dev_alloc(e, (void**)&e->pre_rows[b], S * H * 4);
  }
  rc = rc ? rc : dev_alloc(e, &e->rows, S * H * es);
  rc = rc ? rc : dev_alloc(e, &e->pooled, S * H * es);
  rc = rc ? rc : dev_alloc(e, &e->ctx_rows, S * H * es);
  rc = rc ? rc : dev_alloc(e, &e->xf_rows, S * H * es);
  rc = rc ? rc : dev_alloc(e, &e->q_rows, S * H * es);
  rc = rc ? rc : dev_alloc(e, &e->mid_rows, S * I * es);
  for (int b = 0; b < 2; ++b) rc = rc ? rc : dev_alloc(e, (void**)&e->stats_rows[b], S * 8);
  const size_t n_tiles = (Tp / 256) * ((H + 255) / 256);
  // the separate fp16 residual copies exist in the bf16 setting only (f16: x itself; fp32: not fused)
  const bool two_copies = !e->f32 && !e->f16;
  if (two_copies) {
    rc = rc ? rc : dev_alloc(e, (void**)&e->xres, Tp * H * 2);
    rc = rc ? rc : dev_alloc(e, (void**)&e->xres_rm, Tp * H * 2);
  }
  if (e->res32 && H % 256 == 0) rc = rc ? rc : dev_alloc(e, (void**)&e->xres32, Tp * H * 4);
  rc = rc ? rc : dev_alloc(e, (void**)&e->lnpart, n_tiles * 256 * 8);
  rc = rc ? rc : dev_alloc(e, (void**)&e->lnpart_tag, 4 * n_tiles * 256 * 16);   // per 64-column chunk: gemm4.hip's 64 x 64 form
  e->lnpart_tag_bytes = 4 * n_tiles * 256 * 16;
  // (one flag per 64-row tile: 4x what the ping-pong kernel needs; gemm4.hip's small-M form uses them all)
  rc = rc ? rc : dev_alloc(e, (void**)&e->lnflags, 16 * n_tiles * 4);      // per 64 x 64 tile
  e->ln_tiles = 16 * n_tiles;
  rc = rc ? rc : dev_alloc(e, (void**)&e->dev_err, 4);
  rc = rc ? rc : dev_alloc(e, (void**)&e->pk_ids, (size_t)c->max_tokens * 4);
  rc = rc ? rc : dev_alloc(e, (void**)&e->pk_tt, (size_t)c->max_tokens * 4);
  rc = rc ? rc : dev_alloc(e, (void**)&e->pk_pos, (size_t)c->max_tokens * 4);
  rc = rc ? rc : dev_alloc(e, (void**)&e->pk_cu, ((size_t)c->max_seqs + 1) * 4);
  rc = rc ? rc : dev_alloc(e, (void**)&e->pk_focus, (size_t)c->max_seqs * 4);
  if (rc) return rc;
  {
    void* hp = nullptr;
    HIP_TRY(hipHostMalloc(&hp, 64, hipHostMallocMapped));
    memset(hp, 0, 64);
    e->host_words = static_cast<volatile int32_t*>(hp);
    HIP_TRY(hipHostGetDevicePointer((void**)&e->host_words_dev, hp, 0));
  }
  if (rc) return rc;
  if (two_copies) {
    HIP_TRY(hipMemset(e->xres, 0, Tp * H * 2));
    HIP_TRY(hipMemset(e->xres_rm, 0, Tp * H * 2));
  }
  if (e->xres32) HIP_TRY(hipMemset(e->xres32, 0, Tp * H * 4));
  HIP_TRY(hipMemset(e->lnflags, 0, 16 * n_tiles * 4));
  HIP_TRY(hipMemset(e->lnpart_tag, 0, e->lnpart_tag_bytes));   // tag 0: no launch ever carries epoch 0
  HIP_TRY(hipMemset(e->dev_err, 0, 4));
  // rows between a call's token count and the next multiple of 256 are computed (never used) by the
  // persistent GEMM: keep them finite from the start
  HIP_TRY(hipMemset(e->x, 0, Tp * H * es));
  HIP_TRY(hipMemset(e->qkv, 0, Tp * 3 * H * es));
  HIP_TRY(hipMemset(e->ctx, 0, Tp * H * es));
  HIP_TRY(hipMemset(e->mid, 0, Tp * I * es));
  for (int b = 0; b < 2; ++b) {
    HIP_TRY(hipMemset(e->pre[b], 0, Tp * H * 4));
    HIP_TRY(hipMemset(e->stats[b], 0, Tp * 8));
  }
  return CLIP_OK;
}

const std::vector<float>* find(clip_engine* e, const std::string& k, size_t numel) {
  auto it = e->host_w.find(k);
  if (it == e->host_w.end()) {
    fail(CLIP_ERR_STATE, "missing weight '%s'", k.c_str());
    return nullptr;
  }
  if (it->second.size() != numel) {
    fail(CLIP_ERR_ARG, "weight '%s' has %zu elements, expected %zu", k.c_str(), it->second.size(),
         numel);
    return nullptr;
  }
  return &it->second;
}

}  // namespace

// ---- static scratch of the entry point that allocates nothing (clip_gemm_bf16_lnout_noalloc): the
// fused-LayerNorm exchange buffers ((sum, sumsq) pairs per tile row + one flag per tile) for up to
// CLIP_NOALLOC_MAX_TILES 256x256 output tiles.  Each GPU the library runs on holds its own copy of the
// array (4.2 MB) until the process ends; host code reaches it only through hipGetSymbolAddress.
constexpr size_t SCRATCH_FLAGS_OFF = (size_t)CLIP_NOALLOC_MAX_TILES * 256 * 8;
constexpr size_t SCRATCH_BYTES = SCRATCH_FLAGS_OFF + (size_t)CLIP_NOALLOC_MAX_TILES * 4;
static __device__ __align__(256) unsigned char scratch[SCRATCH_BYTES];

namespace {
struct ScratchDev { unsigned char* ptr = nullptr; };
ScratchDev g_scratch_dev[64];
std::mutex g_scratch_mu;

struct ProfScope {
  clip_engine* e; int kind; hipStream_t st; hipEvent_t a = nullptr, b = nullptr; bool on;
  ProfScope(clip_engine* e_, int kind_, hipStream_t st_) : e(e_), kind(kind_), st(st_) {
    on = (e->prof_mask >> kind) & 1u;
    if (!on) return;
    while (e->ev_pool.size() < e->ev_used + 2) {
      hipEvent_t ev;
      if (hipEventCreate(&ev) != hipSuccess) { on = false; return; }
      e->ev_pool.push_back(ev);
    }
    a = e->ev_pool[e->ev_used++];
    b = e->ev_pool[e->ev_used++];
    (void)hipEventRecord(a, st);
  }
  ~ProfScope() {
    if (!on) return;
    (void)hipEventRecord(b, st);
    e->spans.push_back({kind, a, b});
  }
};
}  // namespace

namespace {

// One forward on the kernel set K (k_bf16: bf16 operands and the fp32 check mode; k_f16: fp16 operands)
template <class KS>
int forward_packed_impl(clip_engine* e, const int32_t* ids, const int32_t* type_ids,
                        const int32_t* pos_ids, const int32_t* cu, const int32_t* focus,
                        int32_t n_seqs, int32_t T, int32_t max_seqlen, int32_t head_kind,
                        float* logits, void* stream, bool rows_checked = false) {
  if (!e || !ids || !pos_ids || !cu || !focus || !logits)
    return fail(CLIP_ERR_ARG, "clip_forward_packed: null argument");
  if (!e->finalized) return fail(CLIP_ERR_STATE, "clip_forward_packed: weights not finalized");
  if (n_seqs == 0 || T == 0) return CLIP_OK;
  if (n_seqs < 0 || T < 0 || max_seqlen < 1 || max_seqlen > e->cfg.max_position)
    return fail(CLIP_ERR_ARG, "clip_forward_packed: bad sizes (n_seqs %d, tokens %d, max_seqlen %d)",
                n_seqs, T, max_seqlen);
  if (n_seqs > e->cfg.max_seqs || T > e->cfg.max_tokens)
    return fail(CLIP_ERR_CAPACITY, "clip_forward_packed: %d seqs / %d tokens exceed the workspace (%d / %d)",
                n_seqs, T, e->cfg.max_seqs, e->cfg.max_tokens);
  if (head_kind == 1 && !e->wpool)
    return fail(CLIP_ERR_STATE, "clip_forward_packed: pooler weights were not loaded");
  if (head_kind != 0 && head_kind != 1) return fail(CLIP_ERR_ARG, "clip_forward_packed: head_kind");
  HIP_TRY(hipSetDevice(e->device));
  hipStream_t st = (hipStream_t)stream;
  e->last_stream = st;
  const clip_config& c = e->cfg;
  const int H = c.hidden_size, I = c.intermediate_size;
  // the row table is the caller's: a row beyond max_seqlen or a table that does not end at T sets bit 2
  // (clip_forward_padded built the table itself and checked the lengths while it did)
  if (!rows_checked) HIP_TRY(launch_check_rows(cu, n_seqs, T, max_seqlen, e->dev_err, st));

  const bool f32 = e->f32;
  // Residual operands.  bf16 setting: the GEMM epilogue re-derives the fp32 LayerNorm output from the
  // producer's pre-LN row, (mean, rstd) and affine (EPI_BIAS_LNRES) — the residual stream never
  // passes through bf16 (tools/rounding_study.py: rounding it was 72 % of the logit error variance).
  // fp32 check mode: the LayerNorm output itself is fp32 and is added directly (EPI_BIAS_RES).
  const bool fused = !f32 && e->fuse_ln && H % 256 == 0 && H / 256 <= 8 && I % 128 == 0;
  // F16R: fp16 operands with the residual stream in FP32 — fused flow: xres32 (tiled, read and rewritten in place by
  // the fused GEMMs, first written by the embedding kernel); unfused flow: the fp32 pre-LayerNorm rows as they are
  const bool res32 = e->res32 && fused;
  int cur = 0;
  const float *g_prev = e->embg, *b_prev = e->embb;
  // fused flow: out = x (bf16) and xres (fp16, tiled, in place: a tile reads exactly the elements it
  // rewrites); the very first residual is the embedding LayerNorm's row-major copy
  bool first_res = true;
  // f16 setting: operand copy and residual copy would hold the same fp16 bits, so there is ONE: x, row-major,
  // read as the residual and rewritten in place by the fused GEMMs (no xres / xres_rm traffic at all)
  const bool single = std::is_same<KS, k_f16>::value && !res32;
  auto ln_gemm = [&](const void* A, const void* W, const void* W_blk, const float* bias, int M, int K,
                     const float* gamma, const float* beta, bool a_tiled) {
    // (A and W in blocks together: the instantiated form; without a block copy of W, A's consumer reads rows)
    const void* W_rm = W;
    if (a_tiled && W_blk) W = W_blk;
    // (F16R: the fp32 stream starts at the first fused GEMM, whose residual is the embedding's fp16 output x itself)
    GemmArgs a{A, W, bias, res32 ? (first_res ? (const void*)e->x : (const void*)e->xres32)
                                 : single ? e->x : (first_res ? (const void*)e->xres_rm : (const void*)e->xres),
               e->x, M, H, K, K, K, H, H, EPI_BIAS_LNOUT, false, 2};
    a.ln_gamma = gamma; a.ln_beta = beta; a.a_tiled = a_tiled; a.w_tiled = a_tiled;
    a.W_rowmajor = W_rm;
    if (e->epoch == 0xffffffffu) {
      // 2^32 launches on these flags (weeks of continuous scoring): a stale flag could equal a new epoch and let a
      // tile run ahead of its partners.  Clear the flags (stream-ordered behind every earlier launch) and start over.
      hipError_t er = hipMemsetAsync(e->lnflags, 0, e->ln_tiles * 4, st);
      if (er == hipSuccess) er = hipMemsetAsync(e->lnpart_tag, 0, e->lnpart_tag_bytes, st);
      if (er != hipSuccess) return er;
      e->epoch = 0;
    }
    a.lnout = LnOutArgs{res32 ? reinterpret_cast<uint16_t*>(e->xres32) : single ? nullptr : e->xres,
                        res32 ? !first_res : (!single && !first_res), e->lnpart, e->lnflags, ++e->epoch,
                        c.layer_norm_eps, e->dev_err, e->test_drop ? 1u : 0u, res32, e->small_m, e->lnpart_tag};
    first_res = false;
    return KS::launch_gemm(a, st);
  };
  auto res_gemm = [&](const void* A, const void* W, const float* bias, int M, int K, const float* pre_in,
                      const float* stats_in, float* pre_out, const int32_t* index) {
    GemmArgs a{A, W, bias, f32 ? e->x : (const void*)pre_in, pre_out, M, H, K, K, K, H, H,
               f32 ? EPI_BIAS_RES : EPI_BIAS_LNRES, true, 0, f32};
    if (!f32) { a.ln_stats = stats_in; a.ln_gamma = g_prev; a.ln_beta = b_prev; a.res_index = index; }
    return KS::launch_gemm(a, st);
  };
  {
    ProfScope ps(e, CLIP_KIND_EMBED_LN, st);
    HIP_TRY(KS::launch_embed_ln(ids, type_ids, pos_ids, e->word, e->pos, e->type, e->embg, e->embb,
                            c.layer_norm_eps, e->x, T, H, c.vocab_size, c.max_position,
                            c.type_vocab_size, st, f32, (f32 || fused) ? nullptr : e->pre[0],
                            (f32 || fused) ? nullptr : e->stats[0], fused && !single && !res32 ? e->xres_rm : nullptr));
  }
  GemmArgs g;
  // bf16: whole 256-row tiles for the persistent GEMM (the workspace is padded); fp32: exact rows
  const int Tg = f32 ? T : (T + 255) / 256 * 256;
  // Fused flow: the two operands with ONE writer and ONE reader — attention's ctx (-> attn-out) and the FFN
  // intermediate (-> FFN-down) — are kept as contiguous 128 x 64 blocks (GemmArgs::a_tiled); the intermediate only
  // when FFN-up itself runs on the ping-pong kernel (launch_gemm's rule), which alone writes that layout.  The layout
  // moves bytes, not arithmetic: a row's result does not depend on it.
  const bool mid_tiled = fused && (long)(Tg / 256) * (I / 256) >= 256 && I % 256 == 0 && H % 128 == 0;
  const bool ctx_tiled = fused && e->layers[0].wo_blk != nullptr;
  // the QKV GEMMs read their weight's block copy when launch_gemm would pick the ping-pong kernel anyway
  const bool qkv_pp = !f32 && (long)(Tg / 256) * (3 * H / 256) >= 256 && (3 * H) % 256 == 0 && H % 128 == 0;
  const bool kv_pp = !f32 && (long)(Tg / 256) * (2 * H / 256) >= 256 && (2 * H) % 256 == 0 && H % 128 == 0;
  // the focus-row kernels of the trimmed last layer are bf16 only: check mode runs every layer whole
  // (decided by the engine's geometry, never by the batch: a row's logits must not depend on its step)
  const bool trim = e->trim_last && c.max_position <= 512 && !f32;
  for (int i = 0; i < c.num_layers; ++i) {
    const LayerW& L = e->layers[i];
    if (trim && i == c.num_layers - 1) {
      // The rest of the last layer runs on n_seqs rows: only the pooled token's q is ever read, so the
      // all-token GEMM stops at K and V (the k|v rows of the fused weight, the k|v columns of qkv) and q
      // is projected for the gathered focus rows alone (same bits: a row's result does not depend on the
      // GEMM kernel that computes it).
      ProfScope ps(e, CLIP_KIND_GEMM_QKV, st);
      g = GemmArgs{e->x, static_cast<bf16_t*>(L.wqkv) + (size_t)H * H, L.bqkv + H, nullptr,
                   static_cast<bf16_t*>(e->qkv) + H, Tg, 2 * H, H, H, H, 3 * H, 0, EPI_BIAS, false};
      // (rows H.. of the block copy start H / 128 block rows in: the same H * H elements)
      if (kv_pp && L.wqkv_blk) { g.W = static_cast<bf16_t*>(L.wqkv_blk) + (size_t)H * H; g.w_tiled = true; g.tile = 2; }
      HIP_TRY(KS::launch_gemm(g, st));
      if (n_seqs <= 32) {   // one batch of the reference's loop: the 16-row kernel reads the focus rows by index
        g = GemmArgs{e->x, L.wqkv, L.bqkv, nullptr, e->q_rows, n_seqs, H, H, H, H, H, 0, EPI_BIAS, false};
        g.a_index = focus;
      } else {
        HIP_TRY(launch_gather_rows(e->x, focus, e->xf_rows, n_seqs, H, st));
        g = GemmArgs{e->xf_rows, L.wqkv, L.bqkv, nullptr, e->q_rows, n_seqs, H, H, H, H, H, 0, EPI_BIAS, false};
      }
      HIP_TRY(KS::launch_gemm(g, st));
      break;
    }
    {
      ProfScope ps(e, CLIP_KIND_GEMM_QKV, st);
      g = GemmArgs{e->x, L.wqkv, L.bqkv, nullptr, e->qkv, Tg, 3 * H, H, H, H, 3 * H, 0, EPI_BIAS, f32, 0, f32};
      if (qkv_pp && L.wqkv_blk) { g.W = L.wqkv_blk; g.w_tiled = true; g.tile = 2; }
      HIP_TRY(KS::launch_gemm(g, st));
    }
    {
      ProfScope ps(e, CLIP_KIND_ATTENTION, st);
      if (f32)
        HIP_TRY(launch_attention_f32(static_cast<const float*>(e->qkv), cu, static_cast<float*>(e->ctx),
                                     n_seqs, max_seqlen, c.num_heads, H, st));
      else
        HIP_TRY(KS::launch_attention(static_cast<const bf16_t*>(e->qkv), cu, static_cast<bf16_t*>(e->ctx),
                                 n_seqs, max_seqlen, c.num_heads, H, st, ctx_tiled));
    }
    if (fused) {
      ProfScope ps(e, CLIP_KIND_GEMM_ATTN_OUT, st);
      HIP_TRY(ln_gemm(e->ctx, L.wo, L.wo_blk, L.bo, Tg, H, L.ln1g, L.ln1b, ctx_tiled));
    } else {
      {
        ProfScope ps(e, CLIP_KIND_GEMM_ATTN_OUT, st);
        HIP_TRY(res_gemm(e->ctx, L.wo, L.bo, Tg, H, e->pre[cur], e->stats[cur], e->pre[cur ^ 1], nullptr));
      }
      cur ^= 1;
      {
        ProfScope ps(e, CLIP_KIND_LAYERNORM, st);
        HIP_TRY(KS::launch_layernorm(e->pre[cur], L.ln1g, L.ln1b, c.layer_norm_eps, e->x, T, H, st, f32,
                                 f32 ? nullptr : e->stats[cur]));
      }
    }
    g_prev = L.ln1g; b_prev = L.ln1b;
    {
      ProfScope ps(e, CLIP_KIND_GEMM_FFN_UP, st);
      g = GemmArgs{e->x, L.wi, L.bi, nullptr, e->mid, Tg, I, H, H, H, I, 0, EPI_BIAS_GELU, f32, 0, f32};
      if (mid_tiled && L.wo2_blk) { g.tile = 2; g.out_tiled = true; }
      HIP_TRY(KS::launch_gemm(g, st));
    }
    if (fused) {
      ProfScope ps(e, CLIP_KIND_GEMM_FFN_DOWN, st);
      HIP_TRY(ln_gemm(e->mid, L.wo2, L.wo2_blk, L.bo2, Tg, I, L.ln2g, L.ln2b, mid_tiled && L.wo2_blk));
    } else {
      {
        ProfScope ps(e, CLIP_KIND_GEMM_FFN_DOWN, st);
        HIP_TRY(res_gemm(e->mid, L.wo2, L.bo2, Tg, I, e->pre[cur], e->stats[cur], e->pre[cur ^ 1], nullptr));
      }
      cur ^= 1;
      {
        ProfScope ps(e, CLIP_KIND_LAYERNORM, st);
        HIP_TRY(KS::launch_layernorm(e->pre[cur], L.ln2g, L.ln2b, c.layer_norm_eps, e->x, T, H, st, f32,
                                 f32 ? nullptr : e->stats[cur]));
      }
    }
    g_prev = L.ln2g; b_prev = L.ln2b;
  }
  if (trim) {
    // ref: sentclf/bert_model.py:52-58 / :234 consume ONE hidden row per sentence, so past the last
    // layer's K/V projection only that row is computed: M = n_seqs instead of T.
    const LayerW& L = e->layers[c.num_layers - 1];
    {
      ProfScope ps(e, CLIP_KIND_LAST_LAYER_ROWS, st);
      HIP_TRY(KS::launch_attention_focus(static_cast<const bf16_t*>(e->qkv), cu, focus,
                                     static_cast<bf16_t*>(e->ctx_rows), n_seqs, c.num_heads, H, st,
                                     static_cast<const bf16_t*>(e->q_rows)));
    }
    {
      ProfScope ps(e, CLIP_KIND_LAST_LAYER_ROWS, st);  // residual rows picked by index: no gather pass
      if (fused) {
        // (a one-layer model trims its only layer: the residual is still the embedding's row-major copy)
        GemmArgs a{e->ctx_rows, L.wo, L.bo,
                   res32 && !first_res ? (const void*)e->xres32
                                       : (single || res32) ? e->x : (first_res ? (const void*)e->xres_rm : (const void*)e->xres),
                   e->pre_rows[0], n_seqs, H, H, H, H, H, H, EPI_BIAS_RES16, true};
        a.res_index = focus;
        a.res_tiles_n = res32 && !first_res ? -(H / 256) : (single || res32 || first_res) ? 0 : H / 256;
        HIP_TRY(KS::launch_gemm(a, st));
      } else {
        HIP_TRY(res_gemm(e->ctx_rows, L.wo, L.bo, n_seqs, H, e->pre[cur], e->stats[cur], e->pre_rows[0], focus));
      }
    }
    {
      ProfScope ps(e, CLIP_KIND_LAST_LAYER_ROWS, st);
      HIP_TRY(KS::launch_layernorm(e->pre_rows[0], L.ln1g, L.ln1b, c.layer_norm_eps, e->rows, n_seqs, H, st,
                               false, e->stats_rows[0]));
    }
    g_prev = L.ln1g; b_prev = L.ln1b;
    {
      ProfScope ps(e, CLIP_KIND_LAST_LAYER_ROWS, st);
      g = GemmArgs{e->rows, L.wi, L.bi, nullptr, e->mid_rows, n_seqs, I, H, H, H, I, 0, EPI_BIAS_GELU, false};
      HIP_TRY(KS::launch_gemm(g, st));
    }
    {
      ProfScope ps(e, CLIP_KIND_LAST_LAYER_ROWS, st);
      HIP_TRY(res_gemm(e->mid_rows, L.wo2, L.bo2, n_seqs, I, e->pre_rows[0], e->stats_rows[0], e->pre_rows[1], nullptr));
    }
    if (head_kind != 0) {
      ProfScope ps(e, CLIP_KIND_LAST_LAYER_ROWS, st);
      HIP_TRY(KS::launch_layernorm(e->pre_rows[1], L.ln2g, L.ln2b, c.layer_norm_eps, e->rows, n_seqs, H, st));
    }
  }
  ProfScope ps_head(e, CLIP_KIND_HEAD, st);
  if (trim) {
    if (head_kind == 0) {
      // last LayerNorm + classifier + the guard below in one kernel (the last kernel of the call)
      const LayerW& L = e->layers[c.num_layers - 1];
      HIP_TRY(KS::launch_ln_head(e->pre_rows[1], L.ln2g, L.ln2b, c.layer_norm_eps, n_seqs, H, e->wcls, e->bcls,
                                 c.num_labels, logits, e->dev_err, st));
      e->last_full = false;
      e->last_tokens = T;
      return CLIP_OK;
    } else {
      g = GemmArgs{e->rows, e->wpool, e->bpool, nullptr, e->pooled, n_seqs, H, H, H, H, H, 0,
                   EPI_BIAS_TANH, false};
      HIP_TRY(KS::launch_gemm(g, st));
      HIP_TRY(KS::launch_classifier(e->pooled, nullptr, e->wcls, e->bcls, logits, n_seqs, H,
                                c.num_labels, st));
    }
  } else if (head_kind == 0) {
    HIP_TRY(KS::launch_classifier(e->x, focus, e->wcls, e->bcls, logits, n_seqs, H, c.num_labels, st, f32));
  } else {
    HIP_TRY(launch_gather_rows(e->x, focus, e->rows, n_seqs, H, st, f32));
    g = GemmArgs{e->rows, e->wpool, e->bpool, nullptr, e->pooled, n_seqs, H, H, H, H, H, 0,
                 EPI_BIAS_TANH, f32, 0, f32};
    HIP_TRY(KS::launch_gemm(g, st));
    HIP_TRY(KS::launch_classifier(e->pooled, nullptr, e->wcls, e->bcls, logits, n_seqs, H,
                              c.num_labels, st, f32));
  }
  e->last_full = !trim;
  e->last_tokens = T;
  // a void call (device error word set) must not look like scores: NaN logits, poll or no poll
  HIP_TRY(launch_poison_logits(e->dev_err, logits, n_seqs * c.num_labels, st));
  return CLIP_OK;
}

// The LM head + CrossEntropyLoss of BertContextMLMSwitch.forward (bert_model.py:165,175-179) on the rows
// that carry an MLM label: CrossEntropyLoss ignores label -100, so the 768 -> vocab decoder runs on those
// rows only (the reference computes all B*S rows of scores and discards ~95 % of them).
template <class KS>
int mlm_head_impl(clip_engine* e, const int32_t* rows, const int32_t* labels, int32_t n, float* scores_out,
                  float* loss, int32_t accumulate, void* stream) {
  if (!e || !loss || (n > 0 && (!rows || !labels)) || n < 0) return fail(CLIP_ERR_ARG, "clip_mlm_head: bad argument");
  if (!e->finalized || !e->w_mlm_dec) return fail(CLIP_ERR_STATE, "clip_mlm_head: no mlm_predictions weights were loaded");
  if (!e->last_full)
    return fail(CLIP_ERR_STATE, "clip_mlm_head: the latest forward trimmed its last layer (clip_set_trim_last_layer(e, 0) first)");
  HIP_TRY(hipSetDevice(e->device));
  hipStream_t st = (hipStream_t)stream;
  const clip_config& c = e->cfg;
  const int H = c.hidden_size, V = e->mlm_vocab, Vp = e->vocab_pad;
  const bool f32 = e->f32;
  const size_t es = f32 ? 4 : 2;
  if (n > e->mlm_cap) {  // grow-only workspace of the head (rows x padded vocabulary of fp32 scores)
    HIP_TRY(hipStreamSynchronize(st));
    const int cap = (n + 255) / 256 * 256;
    for (void* p : {e->mlm_rows, e->mlm_t, (void*)e->mlm_pre, (void*)e->mlm_scores, (void*)e->mlm_row_loss})
      if (p) { (void)hipFree(p); e->allocs.erase(std::find(e->allocs.begin(), e->allocs.end(), p)); }
    e->mlm_rows = e->mlm_t = nullptr; e->mlm_pre = e->mlm_scores = e->mlm_row_loss = nullptr; e->mlm_cap = 0;
    int rc = 0;
    rc = rc ? rc : dev_alloc(e, &e->mlm_rows, (size_t)cap * H * es);
    rc = rc ? rc : dev_alloc(e, &e->mlm_t, (size_t)cap * H * es);
    rc = rc ? rc : dev_alloc(e, (void**)&e->mlm_pre, (size_t)cap * H * 4);
    rc = rc ? rc : dev_alloc(e, (void**)&e->mlm_scores, (size_t)cap * Vp * 4);
    rc = rc ? rc : dev_alloc(e, (void**)&e->mlm_row_loss, (size_t)cap * 4);
    if (rc) return rc;
    HIP_TRY(hipMemset(e->mlm_rows, 0, (size_t)cap * H * es));
    HIP_TRY(hipMemset(e->mlm_t, 0, (size_t)cap * H * es));
    e->mlm_cap = cap;
  }
  if (n > 0) {
    HIP_TRY(launch_gather_rows(e->x, rows, e->mlm_rows, n, H, st, f32));
    // transform: dense + erf-GELU (fp32 out, exact form) -> LayerNorm -> operand rows
    GemmArgs g{e->mlm_rows, e->w_mlm_t, e->b_mlm_t, nullptr, e->mlm_pre, n, H, H, H, H, H, 0, EPI_BIAS_GELU, true, 0, f32};
    HIP_TRY(KS::launch_gemm(g, st));
    HIP_TRY(KS::launch_layernorm(e->mlm_pre, e->g_mlm, e->be_mlm, c.layer_norm_eps, e->mlm_t, n, H, st, f32));
    // decoder: [n, H] x [Vp, H]^T + bias -> fp32 scores (columns past V: -1e30).  Whole 256-row tiles when
    // they are there (the workspace is padded and zeroed), so the ping-pong kernel takes the big calls.
    const int Mg = (!f32 && n >= 4096) ? (n + 255) / 256 * 256 : n;
    g = GemmArgs{e->mlm_t, e->w_mlm_dec, e->b_mlm_dec, nullptr, e->mlm_scores, Mg, Vp, H, H, H, Vp, 0, EPI_BIAS, true, 0, f32};
    HIP_TRY(KS::launch_gemm(g, st));
    HIP_TRY(launch_ce_rows(e->mlm_scores, Vp, V, labels, n, e->mlm_row_loss, scores_out, st));
  }
  HIP_TRY(launch_mean_loss(e->mlm_row_loss, n, loss, accumulate, st));
  return CLIP_OK;
}

}  // namespace

extern "C" {

const char* clip_last_error(void) { return g_err.c_str(); }
int clip_abi_version(void) { return 3; }

int clip_create(const clip_config* c, int device_id, clip_engine** out) {
  if (!c || !out) return fail(CLIP_ERR_ARG, "clip_create: null argument");
  if (c->hidden_size != c->num_heads * 64)
    return fail(CLIP_ERR_ARG, "clip_create: head_dim must be 64 (hidden %d, heads %d)",
                c->hidden_size, c->num_heads);
  switch (c->hidden_size) {
    case 128: case 256: case 384: case 512: case 768: case 1024: break;
    default: return fail(CLIP_ERR_ARG, "clip_create: unsupported hidden_size %d", c->hidden_size);
  }
  if (c->intermediate_size % 64 || c->num_labels < 1 || c->num_labels > 8 || c->num_layers < 1)
    return fail(CLIP_ERR_ARG, "clip_create: unsupported geometry");
  if (c->max_tokens < 1 || c->max_seqs < 1) return fail(CLIP_ERR_ARG, "clip_create: empty workspace");
  HIP_TRY(hipSetDevice(device_id));
  clip_engine* e = new clip_engine();
  e->cfg = *c;
  e->device = device_id;
  if (const char* t = getenv("CLIP_AMD_TEST_DROP_PUBLISH")) e->test_drop = t[0] == '1';
  if (const char* t = getenv("CLIP_AMD_SMALL_M")) e->small_m = t[0] == '0' ? 0 : t[0] == '1' ? 1 : 2;   // (tests, A/B runs: the ping-pong kernel for every size)
  if (const char* t = getenv("CLIP_AMD_TEST_EPOCH_START")) e->epoch = (uint32_t)strtoul(t, nullptr, 0);  // (tests: epoch wrap)
  *out = e;
  return CLIP_OK;
}

int clip_destroy(clip_engine* e) {
  if (!e) return CLIP_OK;
  (void)hipSetDevice(e->device);
  for (void* p : e->allocs) (void)hipFree(p);
  if (e->host_words) (void)hipHostFree(const_cast<int32_t*>(e->host_words));
  for (hipEvent_t ev : e->ev_pool) (void)hipEventDestroy(ev);
  delete e;
  return CLIP_OK;
}

int clip_set_precision(clip_engine* e, int32_t precision) {
  if (!e) return fail(CLIP_ERR_ARG, "clip_set_precision: null engine");
  if (precision != CLIP_PRECISION_BF16 && precision != CLIP_PRECISION_FP32 && precision != CLIP_PRECISION_F16 &&
      precision != CLIP_PRECISION_F16R)
    return fail(CLIP_ERR_ARG, "clip_set_precision: unknown precision %d", precision);
  if (e->finalized || e->stats_rows[1])
    return fail(CLIP_ERR_STATE, "clip_set_precision: call before clip_finalize_weights");
  e->f32 = precision == CLIP_PRECISION_FP32;
  e->f16 = precision == CLIP_PRECISION_F16 || precision == CLIP_PRECISION_F16R;
  e->res32 = precision == CLIP_PRECISION_F16R;
  return CLIP_OK;
}

int clip_load_weight(clip_engine* e, const char* name, const float* host_data, int64_t numel) {
  if (!e || !name || !host_data || numel <= 0) return fail(CLIP_ERR_ARG, "clip_load_weight: bad argument");
  if (e->finalized) return fail(CLIP_ERR_STATE, "clip_load_weight: weights already finalized");
  const std::string k(name);
  const bool known = k.rfind("bert.embeddings.", 0) == 0 || k.rfind("bert.encoder.layer.", 0) == 0 ||
                     k.rfind("bert.pooler.dense.", 0) == 0 || k.rfind("classifier.", 0) == 0 ||
                     k.rfind("mlm_predictions.", 0) == 0;
  if (!known || k == "bert.embeddings.position_ids" || k == "bert.embeddings.token_type_ids")
    return fail(CLIP_ERR_ARG, "clip_load_weight: '%s' is not a parameter of this model", name);
  e->host_w[k].assign(host_data, host_data + numel);
  return CLIP_OK;
}

int clip_finalize_weights(clip_engine* e) {
  if (!e) return fail(CLIP_ERR_ARG, "clip_finalize_weights: null engine");
  if (e->finalized) return CLIP_OK;
  HIP_TRY(hipSetDevice(e->device));
  const clip_config& c = e->cfg;
  const size_t H = c.hidden_size, I = c.intermediate_size;
  int rc;
  if ((rc = alloc_workspace(e))) return rc;
#define GET(var, key, n)                       \
  const std::vector<float>* var = find(e, key, n); \
  if (!var) return g_err.rfind("missing", 0) == 0 ? CLIP_ERR_STATE : CLIP_ERR_ARG;
  {
    GET(w, "bert.embeddings.word_embeddings.weight", (size_t)c.vocab_size * H);
    GET(p, "bert.embeddings.position_embeddings.weight", (size_t)c.max_position * H);
    GET(t, "bert.embeddings.token_type_embeddings.weight", (size_t)c.type_vocab_size * H);
    GET(g, "bert.embeddings.LayerNorm.weight", H);
    GET(b, "bert.embeddings.LayerNorm.bias", H);
    if ((rc = upload_f32(e, *w, &e->word)) || (rc = upload_f32(e, *p, &e->pos)) ||
        (rc = upload_f32(e, *t, &e->type)) || (rc = upload_f32(e, *g, &e->embg)) ||
        (rc = upload_f32(e, *b, &e->embb)))
      return rc;
  }
  e->layers.resize(c.num_layers);
  for (int i = 0; i < c.num_layers; ++i) {
    const std::string p = "bert.encoder.layer." + std::to_string(i) + ".";
    LayerW& L = e->layers[i];
    GET(wq, p + "attention.self.query.weight", H * H);
    GET(wk, p + "attention.self.key.weight", H * H);
    GET(wv, p + "attention.self.value.weight", H * H);
    GET(bq, p + "attention.self.query.bias", H);
    GET(bk, p + "attention.self.key.bias", H);
    GET(bv, p + "attention.self.value.bias", H);
    // fused [3H, H] weight; 1/sqrt(64) = 0.125 folded into the query rows (exact: power of two)
    std::vector<float> wqkv(3 * H * H), bqkv(3 * H);
    memcpy(wqkv.data(), wq->data(), H * H * 4);
    memcpy(wqkv.data() + H * H, wk->data(), H * H * 4);
    memcpy(wqkv.data() + 2 * H * H, wv->data(), H * H * 4);
    for (size_t j = 0; j < H; ++j) {
      bqkv[j] = (*bq)[j] * 0.125f;
      bqkv[H + j] = (*bk)[j];
      bqkv[2 * H + j] = (*bv)[j];
    }
    if ((rc = upload_w(e, wqkv.data(), wqkv.size(), &L.wqkv, 0.125f, H * H))) return rc;
    if ((rc = weight_blocks(e, L.wqkv, 3 * H, H, &L.wqkv_blk))) return rc;
    if ((rc = upload_f32(e, bqkv, &L.bqkv))) return rc;
    GET(wo, p + "attention.output.dense.weight", H * H);
    GET(bo, p + "attention.output.dense.bias", H);
    GET(g1, p + "attention.output.LayerNorm.weight", H);
    GET(b1, p + "attention.output.LayerNorm.bias", H);
    GET(wi, p + "intermediate.dense.weight", I * H);
    GET(bi, p + "intermediate.dense.bias", I);
    GET(w2, p + "output.dense.weight", H * I);
    GET(b2, p + "output.dense.bias", H);
    GET(g2, p + "output.LayerNorm.weight", H);
    GET(bb2, p + "output.LayerNorm.bias", H);
    if ((rc = upload_w(e, wo->data(), wo->size(), &L.wo)) || (rc = upload_f32(e, *bo, &L.bo)) ||
        (rc = upload_f32(e, *g1, &L.ln1g)) || (rc = upload_f32(e, *b1, &L.ln1b)) ||
        (rc = upload_w(e, wi->data(), wi->size(), &L.wi)) || (rc = upload_f32(e, *bi, &L.bi)) ||
        (rc = upload_w(e, w2->data(), w2->size(), &L.wo2)) || (rc = upload_f32(e, *b2, &L.bo2)) ||
        (rc = upload_f32(e, *g2, &L.ln2g)) || (rc = upload_f32(e, *bb2, &L.ln2b)))
      return rc;
    if ((rc = weight_blocks(e, L.wo, H, H, &L.wo_blk)) || (rc = weight_blocks(e, L.wo2, H, I, &L.wo2_blk))) return rc;
  }
  {
    GET(wc, "classifier.weight", (size_t)c.num_labels * H);
    GET(bc, "classifier.bias", (size_t)c.num_labels);
    if ((rc = upload_f32(e, *wc, &e->wcls)) || (rc = upload_f32(e, *bc, &e->bcls))) return rc;
    // the pooler is only needed by the sentence-only head; tolerate its absence
    auto it = e->host_w.find("bert.pooler.dense.weight");
    auto ib = e->host_w.find("bert.pooler.dense.bias");
    if (it != e->host_w.end() && ib != e->host_w.end()) {
      if (it->second.size() != H * H || ib->second.size() != H)
        return fail(CLIP_ERR_ARG, "pooler weight has the wrong size");
      if ((rc = upload_w(e, it->second.data(), H * H, &e->wpool)) ||
          (rc = upload_f32(e, ib->second, &e->bpool)))
        return rc;
    }
  }
  {
    // optional LM head (BertLMPredictionHead: transform dense + GELU + LayerNorm, decoder H -> vocab + bias).
    // decoder.bias is the tied `mlm_predictions.bias` in old checkpoints, its own tensor in new ones.
    auto wt = e->host_w.find("mlm_predictions.transform.dense.weight");
    auto wd = e->host_w.find("mlm_predictions.decoder.weight");
    if (wt != e->host_w.end() && wd != e->host_w.end()) {
      GET(bt, "mlm_predictions.transform.dense.bias", H);
      GET(lg, "mlm_predictions.transform.LayerNorm.weight", H);
      GET(lb, "mlm_predictions.transform.LayerNorm.bias", H);
      auto bd = e->host_w.find("mlm_predictions.decoder.bias");
      if (bd == e->host_w.end()) bd = e->host_w.find("mlm_predictions.bias");
      const size_t V = wd->second.size() / H;
      if (wt->second.size() != H * H || wd->second.size() != V * H || V < 2 || bd == e->host_w.end() ||
          bd->second.size() != V)
        return fail(CLIP_ERR_ARG, "mlm_predictions weights have inconsistent sizes");
      // pad the vocabulary to whole 256-column tiles: zero weight rows, bias -1e30 (vanishes in logsumexp)
      const size_t Vp = (V + 255) / 256 * 256;
      std::vector<float> wpad(Vp * H, 0.f), bpad(Vp, -1e30f);
      memcpy(wpad.data(), wd->second.data(), V * H * 4);
      memcpy(bpad.data(), bd->second.data(), V * 4);
      e->vocab_pad = (int)Vp;
      if ((rc = upload_w(e, wt->second.data(), H * H, &e->w_mlm_t)) || (rc = upload_f32(e, *bt, &e->b_mlm_t)) ||
          (rc = upload_f32(e, *lg, &e->g_mlm)) || (rc = upload_f32(e, *lb, &e->be_mlm)) ||
          (rc = upload_w(e, wpad.data(), Vp * H, &e->w_mlm_dec)) || (rc = upload_f32(e, bpad, &e->b_mlm_dec)))
        return rc;
      e->mlm_vocab = (int)V;
    }
  }
#undef GET
  e->host_w.clear();
  e->finalized = true;
  return CLIP_OK;
}

int clip_focus_index(const int32_t* ids, const int32_t* cu, int32_t n_seqs, int32_t sep_id,
                     int32_t n_ctx, int32_t* focus, int32_t* err, void* stream) {
  if (!ids || !cu || !focus || !err || n_seqs < 0) return fail(CLIP_ERR_ARG, "clip_focus_index: bad argument");
  HIP_TRY(launch_focus_index(ids, cu, n_seqs, sep_id, n_ctx, focus, err, (hipStream_t)stream));
  return CLIP_OK;
}

int clip_prepare_packed(const int32_t* ids, const int32_t* cu, int32_t n_seqs, int32_t sep_id,
                        int32_t n_ctx, int32_t* type_ids, int32_t* pos_ids, int32_t* focus,
                        int32_t* err, void* stream) {
  if (!ids || !cu || !type_ids || !pos_ids || !focus || !err || n_seqs < 0)
    return fail(CLIP_ERR_ARG, "clip_prepare_packed: bad argument");
  HIP_TRY(launch_prepare_packed(ids, cu, n_seqs, sep_id, n_ctx, type_ids, pos_ids, focus, err,
                                (hipStream_t)stream));
  return CLIP_OK;
}

int clip_forward_packed(clip_engine* e, const int32_t* ids, const int32_t* type_ids,
                        const int32_t* pos_ids, const int32_t* cu, const int32_t* focus,
                        int32_t n_seqs, int32_t T, int32_t max_seqlen, int32_t head_kind,
                        float* logits, void* stream) {
  if (e && e->f16)
    return forward_packed_impl<k_f16>(e, ids, type_ids, pos_ids, cu, focus, n_seqs, T, max_seqlen, head_kind, logits, stream);
  return forward_packed_impl<k_bf16>(e, ids, type_ids, pos_ids, cu, focus, n_seqs, T, max_seqlen, head_kind, logits, stream);
}

// Spin on a tag the device writes to mapped host memory behind the kernels queued so far; falls back to a stream wait
// (which also surfaces a failed launch) when the tag does not show within ~2 s.
static int wait_host_tag(clip_engine* e, int slot, int32_t tag, hipStream_t st) {
  const auto t0 = std::chrono::steady_clock::now();
  for (long spins = 0;; ++spins) {
    if (__atomic_load_n(const_cast<const int32_t*>(e->host_words + slot + 1), __ATOMIC_ACQUIRE) == tag) return CLIP_OK;
    __builtin_ia32_pause();
    if ((spins & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) {
      HIP_TRY(hipStreamSynchronize(st));
      if (__atomic_load_n(const_cast<const int32_t*>(e->host_words + slot + 1), __ATOMIC_ACQUIRE) == tag) return CLIP_OK;
      return fail(CLIP_ERR_HIP, "clip_forward_padded: the device never published its word");
    }
  }
}

int clip_forward_padded(clip_engine* e, const int64_t* input_ids, const int64_t* attention_mask,
                        const int64_t* token_type_ids, const int64_t* position_ids, int64_t position_row_stride,
                        int32_t B, int32_t S, int32_t sep_id, int32_t n_ctx, int32_t head_kind, float* logits,
                        int32_t* total_tokens, int32_t* error_flags, void* stream) {
  if (!e || !input_ids || !logits) return fail(CLIP_ERR_ARG, "clip_forward_padded: null argument");
  if (!e->finalized) return fail(CLIP_ERR_STATE, "clip_forward_padded: weights not finalized");
  if (total_tokens) *total_tokens = 0;
  if (error_flags) *error_flags = 0;
  if (B == 0) return CLIP_OK;
  if (B < 0 || S < 1 || position_row_stride < 0)
    return fail(CLIP_ERR_ARG, "clip_forward_padded: bad sizes (batch %d, width %d)", B, S);
  if (B > e->cfg.max_seqs)
    return fail(CLIP_ERR_CAPACITY, "clip_forward_padded: batch of %d rows exceeds the engine workspace (%d rows)", B,
                e->cfg.max_seqs);
  HIP_TRY(hipSetDevice(e->device));
  hipStream_t st = (hipStream_t)stream;
  // lengths and row offsets first: the launch geometry of everything after needs the token count on the host — the one
  // wait before the encoder, behind two tiny kernels: the scan writes the count to mapped host memory and the host
  // spins on its tag (a 4-byte copy back plus a stream wait cost several times the kernels themselves)
  const int32_t tag = ++e->host_tag;
  const int max_seqlen = S < e->cfg.max_position ? S : e->cfg.max_position;   // a longer row: error word bit 2
  HIP_TRY(launch_pack_lengths(attention_mask, B, S, max_seqlen, e->pk_cu, e->dev_err, e->host_words_dev, tag, st));
  int rc = wait_host_tag(e, 0, tag, st);
  if (rc) return rc;
  const int32_t T = e->host_words[0];
  if (total_tokens) *total_tokens = T;
  if (T > e->cfg.max_tokens)
    return fail(CLIP_ERR_CAPACITY, "clip_forward_padded: batch of %d rows / %d tokens exceeds the engine workspace (%d / %d)",
                B, T, e->cfg.max_seqs, e->cfg.max_tokens);
  if (T < 1) return fail(CLIP_ERR_ARG, "clip_forward_padded: no unmasked token in the batch");
  HIP_TRY(launch_pack_tokens(input_ids, attention_mask, token_type_ids, position_ids, (long)position_row_stride, B, S,
                             e->pk_cu, e->pk_ids, token_type_ids ? e->pk_tt : nullptr, e->pk_pos, sep_id,
                             head_kind == 1 ? -1 : n_ctx, e->pk_focus, e->dev_err, st));
  // (a wrong [SEP] count, bert_model.py:51-55, is bit 0 of the ENGINE's error word: one poll for everything)
  rc = e->f16 ? forward_packed_impl<k_f16>(e, e->pk_ids, token_type_ids ? e->pk_tt : nullptr, e->pk_pos, e->pk_cu,
                                           e->pk_focus, B, T, max_seqlen, head_kind, logits, stream, true)
              : forward_packed_impl<k_bf16>(e, e->pk_ids, token_type_ids ? e->pk_tt : nullptr, e->pk_pos, e->pk_cu,
                                            e->pk_focus, B, T, max_seqlen, head_kind, logits, stream, true);
  if (rc || !error_flags) return rc;
  // the caller asked for the error word: published behind the forward the same way, cleared when set (as clip_poll_errors)
  HIP_TRY(launch_publish_word(e->dev_err, e->host_words_dev + 2, tag, st));
  if ((rc = wait_host_tag(e, 2, tag, st))) return rc;
  *error_flags = e->host_words[2];
  if (*error_flags) {
    HIP_TRY(hipMemsetAsync(e->dev_err, 0, 4, st));
    HIP_TRY(hipStreamSynchronize(st));
  }
  return CLIP_OK;
}

int clip_mlm_head(clip_engine* e, const int32_t* rows, const int32_t* labels, int32_t n, float* scores_out,
                  float* loss, int32_t accumulate, void* stream) {
  if (e && e->f16) return mlm_head_impl<k_f16>(e, rows, labels, n, scores_out, loss, accumulate, stream);
  return mlm_head_impl<k_bf16>(e, rows, labels, n, scores_out, loss, accumulate, stream);
}

int clip_ce_loss(const float* logits, const int32_t* labels, int32_t n, int32_t n_classes, float* row_loss,
                 float* loss, int32_t accumulate, void* stream) {
  if (!logits || !labels || !row_loss || !loss || n < 0 || n_classes < 1)
    return fail(CLIP_ERR_ARG, "clip_ce_loss: bad argument");
  HIP_TRY(launch_ce_rows(logits, n_classes, n_classes, labels, n, row_loss, nullptr, (hipStream_t)stream));
  HIP_TRY(launch_mean_loss(row_loss, n, loss, accumulate, (hipStream_t)stream));
  return CLIP_OK;
}

int clip_set_fused_layernorm(clip_engine* e, int32_t on) {
  if (!e) return fail(CLIP_ERR_ARG, "clip_set_fused_layernorm: null engine");
  e->fuse_ln = on != 0;
  return CLIP_OK;
}

int clip_preferred_step_tokens(clip_engine* e, int32_t* tokens) {
  if (!e || !tokens) return fail(CLIP_ERR_ARG, "clip_preferred_step_tokens: null argument");
  const clip_config& c = e->cfg;
  *tokens = c.max_tokens;
  const int H = c.hidden_size;
  if (e->f32 || !e->fuse_ln || H % 256 || H / 256 > 8 || c.intermediate_size % 128) return CLIP_OK;
  int cus = 0;
  HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, e->device));
  // the fused-LayerNorm GEMMs run (CUs / column tiles) row panels of 256 tokens side by side: whole
  // rounds of that many panels leave no CU idle (MI355X, hidden 768: 85 panels = 21 760 tokens)
  const int round_tokens = ((cus & ~7) / (H / 256)) * 256;
  if (round_tokens > 0 && c.max_tokens >= round_tokens) *tokens = c.max_tokens / round_tokens * round_tokens;
  return CLIP_OK;
}

int clip_poll_errors(clip_engine* e, int32_t* host_flags) {
  if (!e || !host_flags) return fail(CLIP_ERR_ARG, "clip_poll_errors: null argument");
  *host_flags = 0;
  if (!e->dev_err) return CLIP_OK;
  HIP_TRY(hipSetDevice(e->device));
  // behind the engine's latest forward, on ITS stream (a blocking copy on the null stream does not order
  // against a non-blocking stream)
  HIP_TRY(hipMemcpyAsync(host_flags, e->dev_err, 4, hipMemcpyDeviceToHost, e->last_stream));
  HIP_TRY(hipStreamSynchronize(e->last_stream));
  if (*host_flags) {
    HIP_TRY(hipMemsetAsync(e->dev_err, 0, 4, e->last_stream));
    HIP_TRY(hipStreamSynchronize(e->last_stream));
  }
  return CLIP_OK;
}

int clip_set_trim_last_layer(clip_engine* e, int32_t on) {
  if (!e) return fail(CLIP_ERR_ARG, "clip_set_trim_last_layer: null engine");
  e->trim_last = on != 0;
  return CLIP_OK;
}

int clip_profile_enable(clip_engine* e, uint32_t kind_mask) {
  if (!e) return fail(CLIP_ERR_ARG, "clip_profile_enable: null engine");
  e->prof_mask = kind_mask & ((1u << CLIP_N_KINDS) - 1);
  return CLIP_OK;
}

int clip_profile_read(clip_engine* e, double* ms, int64_t* launches, int32_t reset) {
  if (!e || !ms || !launches) return fail(CLIP_ERR_ARG, "clip_profile_read: null argument");
  for (int k = 0; k < CLIP_N_KINDS; ++k) { ms[k] = 0.0; launches[k] = 0; }
  for (const auto& s : e->spans) {
    HIP_TRY(hipEventSynchronize(s.b));
    float t = 0.f;
    HIP_TRY(hipEventElapsedTime(&t, s.a, s.b));
    ms[s.kind] += t;
    launches[s.kind] += 1;
  }
  if (reset) { e->spans.clear(); e->ev_used = 0; }
  return CLIP_OK;
}

int clip_epilogue(const float* logits, const double* thr, float any_thresh, int32_t n, int32_t L,
                  float* probs, float* max_score, uint8_t* labels, uint8_t* keep, void* stream) {
  if (!logits || n < 0 || L < 1 || L > 8 || (labels && !thr))
    return fail(CLIP_ERR_ARG, "clip_epilogue: bad argument");
  HIP_TRY(launch_epilogue(logits, thr, any_thresh, n, L, probs, max_score, labels, keep,
                          (hipStream_t)stream));
  return CLIP_OK;
}

int clip_bce_loss(const float* logits, const float* targets, int32_t count, float* loss,
                  void* stream) {
  if (!logits || !targets || !loss || count < 1) return fail(CLIP_ERR_ARG, "clip_bce_loss: bad argument");
  HIP_TRY(launch_bce_loss(logits, targets, count, loss, (hipStream_t)stream));
  return CLIP_OK;
}

int clip_gemm_bf16(const uint16_t* A, const uint16_t* W, const float* bias, const uint16_t* res,
                   void* out, int32_t M, int32_t N, int32_t K, int32_t lda, int32_t ldw,
                   int32_t ldo, int32_t ldres, int32_t epi, int32_t out_f32, int32_t tile,
                   void* stream) {
  GemmArgs g{A, W, bias, res, out, M, N, K, lda, ldw, ldo, ldres, epi, out_f32 != 0, tile};
  if (const char* msg = gemm_check(g)) return fail(CLIP_ERR_ARG, "%s", msg);
  HIP_TRY(k_bf16::launch_gemm(g, (hipStream_t)stream));
  return CLIP_OK;
}

int clip_gemm_f16(const uint16_t* A, const uint16_t* W, const float* bias, const uint16_t* res,
                  void* out, int32_t M, int32_t N, int32_t K, int32_t lda, int32_t ldw,
                  int32_t ldo, int32_t ldres, int32_t epi, int32_t out_f32, int32_t tile,
                  void* stream) {
  GemmArgs g{A, W, bias, res, out, M, N, K, lda, ldw, ldo, ldres, epi, out_f32 != 0, tile};
  if (const char* msg = gemm_check(g)) return fail(CLIP_ERR_ARG, "%s", msg);
  HIP_TRY(k_f16::launch_gemm(g, (hipStream_t)stream));
  return CLIP_OK;
}

int clip_gemm_bf16_lnres(const uint16_t* A, const uint16_t* W, const float* bias, const float* res_pre,
                         const float* ln_stats, const float* ln_gamma, const float* ln_beta,
                         const int32_t* res_index, float* out, int32_t M, int32_t N, int32_t K,
                         int32_t lda, int32_t ldw, int32_t ldo, int32_t ldres, int32_t tile,
                         void* stream) {
  GemmArgs g{A, W, bias, res_pre, out, M, N, K, lda, ldw, ldo, ldres, EPI_BIAS_LNRES, true, tile};
  g.ln_stats = ln_stats; g.ln_gamma = ln_gamma; g.ln_beta = ln_beta; g.res_index = res_index;
  if (const char* msg = gemm_check(g)) return fail(CLIP_ERR_ARG, "%s", msg);
  if (res_index && (tile == 1 || tile == 2)) return fail(CLIP_ERR_ARG, "gemm: res_index needs a one-tile kernel");
  if (tile == 1) return fail(CLIP_ERR_ARG, "gemm: the LayerNorm-residual epilogue is not built for tile 1");
  HIP_TRY(k_bf16::launch_gemm(g, (hipStream_t)stream));
  return CLIP_OK;
}

int clip_gemm_bf16_lnout(const uint16_t* A, const uint16_t* W, const float* bias, const uint16_t* res16,
                         int32_t res_tiled, const float* gamma, const float* beta, float eps, uint16_t* x_out,
                         uint16_t* xres_out, float* part, uint32_t* flags, uint32_t epoch, int32_t* err,
                         int32_t M, int32_t N, int32_t K, int32_t lda, int32_t ldw, int32_t ldo,
                         int32_t ldres, void* stream) {
  GemmArgs g{A, W, bias, res16, x_out, M, N, K, lda, ldw, ldo, ldres, EPI_BIAS_LNOUT, false, 2};
  g.ln_gamma = gamma; g.ln_beta = beta;
  g.lnout = LnOutArgs{xres_out, res_tiled != 0, part, flags, epoch, eps, err};
  if (const char* msg = gemm_check(g)) return fail(CLIP_ERR_ARG, "%s", msg);
  if (!res16 || !xres_out || !part || !flags || !err || N / 256 > 8)
    return fail(CLIP_ERR_ARG, "clip_gemm_bf16_lnout: null buffer, or more than 8 column tiles");
  HIP_TRY(k_bf16::launch_gemm(g, (hipStream_t)stream));
  return CLIP_OK;
}

int clip_gemm_f16_lnout_inplace(const uint16_t* A, const uint16_t* W, const float* bias, const float* gamma,
                                const float* beta, float eps, uint16_t* x_inout, float* part, uint32_t* flags,
                                uint32_t epoch, int32_t* err, int32_t M, int32_t N, int32_t K, int32_t lda,
                                int32_t ldw, int32_t ldx, void* stream) {
  GemmArgs g{A, W, bias, x_inout, x_inout, M, N, K, lda, ldw, ldx, ldx, EPI_BIAS_LNOUT, false, 2};
  g.ln_gamma = gamma; g.ln_beta = beta;
  g.lnout = LnOutArgs{nullptr, false, part, flags, epoch, eps, err};
  if (const char* msg = gemm_check(g)) return fail(CLIP_ERR_ARG, "%s", msg);
  if (!x_inout || !part || !flags || !err || N / 256 > 8 || A == x_inout)
    return fail(CLIP_ERR_ARG, "clip_gemm_f16_lnout_inplace: null buffer, A aliasing x, or more than 8 column tiles");
  HIP_TRY(k_f16::launch_gemm(g, (hipStream_t)stream));
  return CLIP_OK;
}

int clip_gemm_f16_lnout_res32(const uint16_t* A, const uint16_t* W, const float* bias, const float* gamma,
                              const float* beta, float eps, uint16_t* x_out, float* xres32_inout, float* part,
                              uint32_t* flags, uint32_t epoch, int32_t* err, int32_t M, int32_t N, int32_t K,
                              int32_t lda, int32_t ldw, int32_t ldx, void* stream) {
  GemmArgs g{A, W, bias, xres32_inout, x_out, M, N, K, lda, ldw, ldx, N, EPI_BIAS_LNOUT, false, 2};
  g.ln_gamma = gamma; g.ln_beta = beta;
  g.lnout = LnOutArgs{reinterpret_cast<uint16_t*>(xres32_inout), true, part, flags, epoch, eps, err, 0, true};
  if (const char* msg = gemm_check(g)) return fail(CLIP_ERR_ARG, "%s", msg);
  if (!x_out || !xres32_inout || !part || !flags || !err || N / 256 > 8)
    return fail(CLIP_ERR_ARG, "clip_gemm_f16_lnout_res32: null buffer, or more than 8 column tiles");
  HIP_TRY(k_f16::launch_gemm(g, (hipStream_t)stream));
  return CLIP_OK;
}

int clip_gemm_bf16_lnout_noalloc(const uint16_t* A, const uint16_t* W, const float* bias, const uint16_t* res16,
                                 int32_t res_tiled, const float* gamma, const float* beta, float eps,
                                 uint16_t* x_out, uint16_t* xres_out, int32_t* err, int32_t M, int32_t N,
                                 int32_t K, int32_t lda, int32_t ldw, int32_t ldo, int32_t ldres, void* stream) {
  GemmArgs g{A, W, bias, res16, x_out, M, N, K, lda, ldw, ldo, ldres, EPI_BIAS_LNOUT, false, 2};
  g.ln_gamma = gamma; g.ln_beta = beta;
  if (const char* msg = gemm_check(g)) return fail(CLIP_ERR_ARG, "%s", msg);
  if (!res16 || !xres_out || !err || N / 256 > 8)
    return fail(CLIP_ERR_ARG, "clip_gemm_bf16_lnout_noalloc: null buffer, or more than 8 column tiles");
  const long tiles = (long)(M / 256) * (N / 256);
  if (tiles > CLIP_NOALLOC_MAX_TILES)
    return fail(CLIP_ERR_CAPACITY, "clip_gemm_bf16_lnout_noalloc: %ld tiles of 256x256 exceed the static scratch (%d tiles)",
                tiles, CLIP_NOALLOC_MAX_TILES);
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64) return fail(CLIP_ERR_ARG, "clip_gemm_bf16_lnout_noalloc: device %d", dev);
  hipStream_t st = (hipStream_t)stream;
  unsigned char* base;
  {
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    ScratchDev& sd = g_scratch_dev[dev];
    if (!sd.ptr) {
      // once per device, with that device current: the device address of THIS GPU's copy of the array
      // (the array's name is a host-side stand-in, never a kernel argument)
      void* p = nullptr;
      HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
      sd.ptr = static_cast<unsigned char*>(p);
    }
    base = sd.ptr;
  }
  // The array keeps the last call's contents, so EVERY call clears the flags of the tiles it will use, on
  // `stream`, ahead of its kernel, and the kernel waits for the constant value 1.  No host-side counter is
  // baked into the kernel arguments: a call recorded into a HIP graph (memset node + kernel node) replays
  // correctly any number of times.
  HIP_TRY(hipMemsetAsync(base + SCRATCH_FLAGS_OFF, 0, (size_t)tiles * 4, st));
  g.lnout = LnOutArgs{xres_out, res_tiled != 0, reinterpret_cast<float*>(base),
                      reinterpret_cast<uint32_t*>(base + SCRATCH_FLAGS_OFF), 1u, eps, err, 0};
  HIP_TRY(k_bf16::launch_gemm(g, st));
  return CLIP_OK;
}

int clip_gemm_f32(const float* A, const float* W, const float* bias, const float* res, float* out,
                  int32_t M, int32_t N, int32_t K, int32_t lda, int32_t ldw, int32_t ldo,
                  int32_t ldres, int32_t epi, void* stream) {
  GemmArgs g{A, W, bias, res, out, M, N, K, lda, ldw, ldo, ldres, epi, true, 0, true};
  if (const char* msg = gemm_check(g)) return fail(CLIP_ERR_ARG, "%s", msg);
  HIP_TRY(k_bf16::launch_gemm(g, (hipStream_t)stream));
  return CLIP_OK;
}

int clip_attention_varlen(const uint16_t* qkv, const int32_t* cu, uint16_t* ctx, int32_t n_seqs,
                          int32_t max_seqlen, int32_t n_heads, int32_t hidden, void* stream) {
  if (!qkv || !cu || !ctx || n_seqs < 0) return fail(CLIP_ERR_ARG, "clip_attention_varlen: bad argument");
  HIP_TRY(k_bf16::launch_attention(qkv, cu, ctx, n_seqs, max_seqlen, n_heads, hidden, (hipStream_t)stream));
  return CLIP_OK;
}

int clip_attention_varlen_f16(const uint16_t* qkv, const int32_t* cu, uint16_t* ctx, int32_t n_seqs,
                              int32_t max_seqlen, int32_t n_heads, int32_t hidden, void* stream) {
  if (!qkv || !cu || !ctx || n_seqs < 0) return fail(CLIP_ERR_ARG, "clip_attention_varlen_f16: bad argument");
  HIP_TRY(k_f16::launch_attention(qkv, cu, ctx, n_seqs, max_seqlen, n_heads, hidden, (hipStream_t)stream));
  return CLIP_OK;
}

int clip_layernorm(const float* x, const float* gamma, const float* beta, float eps, uint16_t* out,
                   int32_t T, int32_t H, void* stream) {
  if (!x || !gamma || !beta || !out || T < 0) return fail(CLIP_ERR_ARG, "clip_layernorm: bad argument");
  HIP_TRY(k_bf16::launch_layernorm(x, gamma, beta, eps, out, T, H, (hipStream_t)stream));
  return CLIP_OK;
}

int clip_layernorm_stats(const float* x, const float* gamma, const float* beta, float eps,
                         uint16_t* out, float* stats, int32_t T, int32_t H, void* stream) {
  if (!x || !gamma || !beta || !out || !stats || T < 0)
    return fail(CLIP_ERR_ARG, "clip_layernorm_stats: bad argument");
  HIP_TRY(k_bf16::launch_layernorm(x, gamma, beta, eps, out, T, H, (hipStream_t)stream, false, stats));
  return CLIP_OK;
}

int clip_embed_layernorm(const int32_t* ids, const int32_t* type_ids, const int32_t* pos_ids,
                         const float* word, const float* pos, const float* type, const float* gamma,
                         const float* beta, float eps, uint16_t* out, int32_t T, int32_t H,
                         int32_t vocab, int32_t max_pos, int32_t type_vocab, void* stream) {
  if (!ids || !pos_ids || !word || !pos || !type || !gamma || !beta || !out || T < 0)
    return fail(CLIP_ERR_ARG, "clip_embed_layernorm: bad argument");
  HIP_TRY(k_bf16::launch_embed_ln(ids, type_ids, pos_ids, word, pos, type, gamma, beta, eps, out, T, H,
                          vocab, max_pos, type_vocab, (hipStream_t)stream));
  return CLIP_OK;
}

}  // extern "C"
