// Row-wise (HBM-bound) kernels of the scoring path: embedding gather + LayerNorm, LayerNorm,
// focus-[SEP] index, row gather, classifier head, fused sigmoid/threshold/max epilogue, BCE loss.
// One 64-lane wave per row; all reductions are wave shuffles (no LDS, no atomics).
#include "common.h"
#include "kernels.h"

namespace {

template <int VEC>
struct VecT;
template <>
struct VecT<4> { typedef float4 type; };
template <>
struct VecT<2> { typedef float2 type; };

template <int VEC>
__device__ __forceinline__ void load_vec(const float* p, float* dst) {
  typename VecT<VEC>::type v = *reinterpret_cast<const typename VecT<VEC>::type*>(p);
  const float* f = reinterpret_cast<const float*>(&v);
#pragma unroll
  for (int i = 0; i < VEC; ++i) dst[i] = f[i];
}

// normalise the NIT*VEC values a lane holds and store them as bf16
// OUT_F32 = fp32 check mode of the same kernels: the normalised row is stored as fp32
template <int VEC, int NIT, bool OUT_F32>
__device__ __forceinline__ void ln_finish(float (&x)[NIT][VEC], const float* gamma,
                                          const float* beta, float eps, void* outv, int H,
                                          int lane, float2* stats_row, uint16_t* res16 = nullptr) {
  bf16_t* out = reinterpret_cast<bf16_t*>(outv);
  float* outf = reinterpret_cast<float*>(outv);
  float s = 0.f;
#pragma unroll
  for (int it = 0; it < NIT; ++it)
#pragma unroll
    for (int v = 0; v < VEC; ++v) s += x[it][v];
  const float mean = wave_sum(s) / (float)H;
  float q = 0.f;
#pragma unroll
  for (int it = 0; it < NIT; ++it)
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      const float d = x[it][v] - mean;
      q += d * d;
    }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)H + eps);
  if (stats_row && lane == 0) *stats_row = make_float2(mean, rstd);  // for EPI_BIAS_LNRES consumers
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int e = (it * 64 + lane) * VEC;
    float g[VEC], b[VEC];
    load_vec<VEC>(gamma + e, g);
    load_vec<VEC>(beta + e, b);
    float y[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) y[v] = ln_apply(x[it][v], mean, rstd, g[v], b[v]);
    if (res16) {  // fp16 copy of the same fp32 values: the residual operand of EPI_BIAS_LNOUT / RES16
      if (VEC == 4) {  // one 8-byte store per lane, not four 2-byte ones
        typedef __attribute__((ext_vector_type(4))) _Float16 h4_t;
        const h4_t hv = {(_Float16)y[0], (_Float16)y[1], (_Float16)y[2], (_Float16)y[3]};
        *reinterpret_cast<h4_t*>(res16 + e) = hv;
      } else {
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          const _Float16 hv = (_Float16)y[v];
          res16[e + v] = *reinterpret_cast<const uint16_t*>(&hv);
        }
      }
    }
    if (OUT_F32) {
#pragma unroll
      for (int v = 0; v < VEC; ++v) outf[e + v] = y[v];
    } else if (VEC == 4) {
      *reinterpret_cast<uint2*>(out + e) = make_uint2(pack_op2(y[0], y[1]), pack_op2(y[2], y[3]));
    } else {
      *reinterpret_cast<uint32_t*>(out + e) = pack_op2(y[0], y[1]);
    }
  }
}

template <int VEC, int NIT, bool OUT_F32>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ xin,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps,
                                                        void* __restrict__ out, int T, int H,
                                                        float2* __restrict__ stats,
                                                        uint16_t* __restrict__ res16) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= T) return;
  float x[NIT][VEC];
#pragma unroll
  for (int it = 0; it < NIT; ++it) load_vec<VEC>(xin + (size_t)row * H + (it * 64 + lane) * VEC, x[it]);
  ln_finish<VEC, NIT, OUT_F32>(x, gamma, beta, eps,
                               reinterpret_cast<char*>(out) + (size_t)row * H * (OUT_F32 ? 4 : 2), H, lane,
                               stats ? stats + row : nullptr, res16 ? res16 + (size_t)row * H : nullptr);
}

// BertEmbeddings.forward (transformers modeling_bert.py:68-108): word + token_type + position
template <int VEC, int NIT, bool OUT_F32>
__global__ __launch_bounds__(256) void embed_ln_kernel(
    const int32_t* __restrict__ ids, const int32_t* __restrict__ type_ids,
    const int32_t* __restrict__ pos_ids, const float* __restrict__ word,
    const float* __restrict__ pos, const float* __restrict__ type,
    const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
    void* __restrict__ out, int T, int H, int vocab, int max_pos, int type_vocab,
    float* __restrict__ pre_out, float2* __restrict__ stats, uint16_t* __restrict__ res16) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= T) return;
  // ids are validated on the host side of the ABI; clamp anyway so a bad id cannot fault the GPU
  const int id = min(max(ids[row], 0), vocab - 1);
  const int ty = type_ids ? min(max(type_ids[row], 0), type_vocab - 1) : 0;
  const int po = min(max(pos_ids[row], 0), max_pos - 1);
  float x[NIT][VEC];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int e = (it * 64 + lane) * VEC;
    float a[VEC], b[VEC], c[VEC];
    load_vec<VEC>(word + (size_t)id * H + e, a);
    load_vec<VEC>(type + (size_t)ty * H + e, b);
    load_vec<VEC>(pos + (size_t)po * H + e, c);
#pragma unroll
    for (int v = 0; v < VEC; ++v) x[it][v] = (a[v] + b[v]) + c[v];
    if (pre_out) {  // the un-normalised sum, for the first layer's recomputed residual
#pragma unroll
      for (int v = 0; v < VEC; ++v) pre_out[(size_t)row * H + e + v] = x[it][v];
    }
  }
  ln_finish<VEC, NIT, OUT_F32>(x, gamma, beta, eps,
                               reinterpret_cast<char*>(out) + (size_t)row * H * (OUT_F32 ? 4 : 2), H, lane,
                               stats ? stats + row : nullptr, res16 ? res16 + (size_t)row * H : nullptr);
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const char* __restrict__ src,
                                                          const int32_t* __restrict__ rows,
                                                          char* __restrict__ dst, int n,
                                                          int row_bytes) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  const uint4* s = reinterpret_cast<const uint4*>(src + (size_t)rows[i] * row_bytes);
  uint4* d = reinterpret_cast<uint4*>(dst + (size_t)i * row_bytes);
  for (int c = lane; c < row_bytes / 16; c += 64) d[c] = s[c];
}

// classifier Linear(H -> L) on one row per wave (ref: sentclf/bert_model.py:91,239)
template <bool F32>
__global__ __launch_bounds__(256) void classifier_kernel(const void* __restrict__ xv,
                                                         const int32_t* __restrict__ rows,
                                                         const float* __restrict__ Wc,
                                                         const float* __restrict__ bc,
                                                         float* __restrict__ logits, int n, int H,
                                                         int L) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  const size_t r = (size_t)(rows ? rows[i] : i) * H;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int e = lane * 2; e < H; e += 128) {
    float x0, x1;
    if (F32) {
      const float2 xf = *reinterpret_cast<const float2*>(reinterpret_cast<const float*>(xv) + r + e);
      x0 = xf.x; x1 = xf.y;
    } else {
      const uint32_t xb = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const bf16_t*>(xv) + r + e);
      x0 = op2_lo(xb); x1 = op2_hi(xb);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (j < L) {
        const float2 wv = *reinterpret_cast<const float2*>(Wc + (size_t)j * H + e);
        acc[j] += x0 * wv.x + x1 * wv.y;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (j < L) {
      const float s = wave_sum(acc[j]);
      if (lane == 0) logits[(size_t)i * L + j] = s + bc[j];
    }
  }
}

// The end of the trimmed forward in ONE kernel, one wave per sentence: the last LayerNorm (arithmetic of ln_finish,
// output rounded to the 16-bit operand type as when it is stored), the classifier Linear(H -> L) on it (ref:
// sentclf/bert_model.py:91) and the guard of poison_logits_kernel.  Every load of a lane is independent of the others
// (compile-time trip counts), so the kernel costs one memory latency instead of one per 128 columns and per launch.
template <int VEC, int NIT>
__global__ __launch_bounds__(256) void ln_head_kernel(const float* __restrict__ xin, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, float eps, int n, int H,
                                                      const float* __restrict__ Wc, const float* __restrict__ bc,
                                                      int L, float* __restrict__ logits, int32_t* __restrict__ err) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  // every load of the kernel is issued up front (the row, the affine, the L classifier rows): one memory latency
  float x[NIT][VEC], g[NIT][VEC], b[NIT][VEC], w[8][NIT][VEC];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int e = (it * 64 + lane) * VEC;
    load_vec<VEC>(xin + (size_t)row * H + e, x[it]);
    load_vec<VEC>(gamma + e, g[it]);
    load_vec<VEC>(beta + e, b[it]);
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (j < L) load_vec<VEC>(Wc + (size_t)j * H + e, w[j][it]);
  }
  const bool void_call = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
  float s = 0.f;
#pragma unroll
  for (int it = 0; it < NIT; ++it)
#pragma unroll
    for (int v = 0; v < VEC; ++v) s += x[it][v];
  const float mean = wave_sum(s) / (float)H;
  float q = 0.f;
#pragma unroll
  for (int it = 0; it < NIT; ++it)
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      const float d = x[it][v] - mean;
      q += d * d;
    }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)H + eps);
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    float y[VEC];
#pragma unroll
    for (int v = 0; v < VEC; v += 2) {
      const uint32_t u = pack_op2(ln_apply(x[it][v], mean, rstd, g[it][v], b[it][v]),
                                  ln_apply(x[it][v + 1], mean, rstd, g[it][v + 1], b[it][v + 1]));
      y[v] = op2_lo(u);
      y[v + 1] = op2_hi(u);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (j < L) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[j] += y[v] * w[j][it][v];
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (j < L) {
      float v = wave_sum(acc[j]) + bc[j];
      if (lane == 0) {
        if (void_call) v = __builtin_nanf("");
        else if (!__builtin_isfinite(v)) atomicOr(err, 8);
        logits[(size_t)row * L + j] = v;
      }
    }
  }
}

// ref: sentclf/bert_model.py:52 — where(input_ids == sep)[1].reshape(-1, 2n+1)[:, n], per row
__global__ __launch_bounds__(64) void focus_index_kernel(const int32_t* __restrict__ ids,
                                                         const int32_t* __restrict__ cu, int n_seqs,
                                                         int sep_id, int n_ctx,
                                                         int32_t* __restrict__ focus,
                                                         int32_t* __restrict__ err) {
  const int s = blockIdx.x;
  const int lane = threadIdx.x;
  const int beg = cu[s], end = cu[s + 1];
  // an EMPTY row (the row check of the forward rejects it, error word bit 2) still gets an index inside the token
  // buffer: cu[s] of an empty last row is total_tokens, one row past it
  const int fallback = beg < end ? beg : (beg > 0 ? beg - 1 : 0);
  if (n_ctx < 0) {  // sentence-only model pools token 0
    if (lane == 0) focus[s] = fallback;
    return;
  }
  int count = 0, found = -1;
  for (int t0 = beg; t0 < end; t0 += 64) {
    const int t = t0 + lane;
    const bool hit = t < end && ids[t] == sep_id;
    unsigned long long m = __ballot(hit);
    const int c = __popcll(m);
    if (found < 0 && count + c > n_ctx) {
      int k = n_ctx - count;  // k-th set bit (0-based)
      unsigned long long mm = m;
      while (k-- > 0) mm &= mm - 1;
      found = t0 + __ffsll((long long)mm) - 1;
    }
    count += c;
  }
  if (lane == 0) {
    focus[s] = found < 0 ? fallback : found;
    if (count != 2 * n_ctx + 1) atomicOr(err, 1);
  }
}

// Everything the packed forward needs besides the token ids, from the ids alone, one wave per row:
//   pos_ids[t]  = t - row start                       (transformers BertEmbeddings position_ids)
//   type_ids[t] = 0 iff exactly n_ctx separators lie at or before t, else 1
//                 (ref: sentclf/bert.py:168-176: [0, sep[n-1]) -> 1, [sep[n-1], sep[n]) -> 0,
//                  [sep[n], end) -> 1; n_ctx <= 0: all 0, the sentence-only collator)
//   focus[s]    = index of the (n_ctx+1)-th separator (bert_model.py:52) or the row's first token
__global__ __launch_bounds__(64) void prepare_packed_kernel(const int32_t* __restrict__ ids,
                                                            const int32_t* __restrict__ cu,
                                                            int sep_id, int n_ctx,
                                                            int32_t* __restrict__ type_ids,
                                                            int32_t* __restrict__ pos_ids,
                                                            int32_t* __restrict__ focus,
                                                            int32_t* __restrict__ err) {
  const int s = blockIdx.x, lane = threadIdx.x;
  const int beg = cu[s], end = cu[s + 1];
  int count = 0, found = -1;
  for (int t0 = beg; t0 < end; t0 += 64) {
    const int t = t0 + lane;
    const bool hit = n_ctx > 0 && t < end && ids[t] == sep_id;
    const unsigned long long m = __ballot(hit);
    const int incl = count + __popcll(m & ((2ull << lane) - 1));  // separators at or before t
    if (t < end) {
      pos_ids[t] = t - beg;
      type_ids[t] = (n_ctx > 0 && incl != n_ctx) ? 1 : 0;
    }
    const int c = __popcll(m);
    if (found < 0 && n_ctx > 0 && count + c > n_ctx) {
      int k = n_ctx - count;
      unsigned long long mm = m;
      while (k-- > 0) mm &= mm - 1;
      found = t0 + __ffsll((long long)mm) - 1;
    }
    count += c;
  }
  if (lane == 0) {
    focus[s] = (n_ctx > 0 && found >= 0) ? found : (beg < end ? beg : (beg > 0 ? beg - 1 : 0));  // empty row: in bounds
    if (n_ctx > 0 && count != 2 * n_ctx + 1) atomicOr(err, 1);
  }
}

// 8 lanes per sentence (lane j<L holds label j), wave-shuffle max / any over the label axis.
__global__ __launch_bounds__(256) void epilogue_kernel(const float* __restrict__ logits,
                                                       const double* __restrict__ label_thresh,
                                                       float any_thresh, int n, int L,
                                                       float* __restrict__ probs,
                                                       float* __restrict__ max_score,
                                                       uint8_t* __restrict__ labels,
                                                       uint8_t* __restrict__ keep) {
  const int g = blockIdx.x * 32 + (threadIdx.x >> 3);  // sentence
  const int j = threadIdx.x & 7;                       // label slot
  const bool live = g < n && j < L;
  float p = -1.0f;  // below every probability
  if (live) {
    const float x = logits[(size_t)g * L + j];
    p = 1.0f / (1.0f + expf(-x));  // torch.sigmoid in fp32
    if (probs) probs[(size_t)g * L + j] = p;
    if (labels) labels[(size_t)g * L + j] = label_thresh ? ((double)p > label_thresh[j]) : 0;
  }
  int any = live && (p > any_thresh);
  float m = p;
#pragma unroll
  for (int o = 1; o < 8; o <<= 1) {
    m = fmaxf(m, __shfl_xor(m, o, 64));
    any |= __shfl_xor(any, o, 64);
  }
  if (g < n && j == 0) {
    if (max_score) max_score[g] = m;
    if (keep) keep[g] = (uint8_t)any;
  }
}

__global__ __launch_bounds__(256) void bce_loss_kernel(const float* __restrict__ logits,
                                                       const float* __restrict__ targets,
                                                       int count, float* __restrict__ loss) {
  __shared__ float part[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < count; i += 256) {
    const float x = logits[i], y = targets[i];
    s += fmaxf(x, 0.f) - x * y + log1pf(expf(-fabsf(x)));
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) loss[0] = (part[0] + part[1] + part[2] + part[3]) / (float)count;
}

// Guards of clip_forward_packed (no counterpart in the reference, whose padded tensors cannot be
// inconsistent): the caller's cu_seqlens / max_seqlen are trusted by the attention launch geometry, so
// a row longer than max_seqlen (or an offset table that does not end at total_tokens) would leave
// stale ctx rows behind silently.  err |= 4 instead.
__global__ __launch_bounds__(256) void check_rows_kernel(const int32_t* __restrict__ cu, int n_seqs,
                                                         int total_tokens, int max_seqlen,
                                                         int32_t* __restrict__ err) {
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s >= n_seqs) return;
  const int len = cu[s + 1] - cu[s];
  bool bad = len < 1 || len > max_seqlen;
  if (s == 0) bad |= cu[0] != 0;
  if (s == n_seqs - 1) bad |= cu[n_seqs] != total_tokens;
  if (bad) atomicOr(err, 4);
}

// ---- the reference's PADDED call shape (model(**inputs): apply_bert_unlabeled.py:110, test_load.py:36) ----
// [B, S] int64 tensors as the HF tokenizer / the collators hand them over (sentclf/bert.py:57,152) -> the packed
// int32 arrays of clip_forward_packed, on the device: tokens with attention_mask != 0 are kept in row-major order
// (the key-padding mask of modeling_bert.py:688-708 becomes the packing).  Three small kernels instead of the
// boolean-mask indexing of a tensor library (three device-to-host waits per call there).
// One wave per row: cu[b + 1] <- number of kept tokens of row b (no mask: S).
__global__ __launch_bounds__(256) void pack_lens_kernel(const int64_t* __restrict__ mask, int B, int S,
                                                        int32_t* __restrict__ cu) {
  const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  int n = S;
  if (mask) {
    n = 0;
    for (int j0 = 0; j0 < S; j0 += 64) {
      const int j = j0 + lane;
      n += __popcll(__ballot(j < S && mask[(size_t)b * S + j] != 0));
    }
  }
  if (lane == 0) cu[b + 1] = n;
}

// cu[0] = 0, cu[1..B] <- inclusive prefix sums of the lengths pack_lens_kernel left there; ONE workgroup.
// host_out (optional, host memory mapped into the device): the token count and then `tag`, for a host that spins on
// the tag instead of paying a device-to-host copy and a stream wait.
__global__ __launch_bounds__(1024) void pack_scan_kernel(int32_t* __restrict__ cu, int B, int max_len,
                                                         int32_t* __restrict__ err, int32_t* host_out, int32_t tag) {
  __shared__ int wsum[16];
  __shared__ int carry;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (tid == 0) { cu[0] = 0; carry = 0; }
  __syncthreads();
  for (int base = 0; base < B; base += 1024) {
    const int i = base + tid;
    int x = i < B ? cu[i + 1] : 0;
    if (i < B && (x < 1 || x > max_len)) atomicOr(err, 4);     // the row check of the packed forward (check_rows_kernel)
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int y = __shfl_up(x, o);
      if (lane >= o) x += y;
    }
    if (lane == 63) wsum[w] = x;
    __syncthreads();
    if (w == 0) {
      int t = lane < 16 ? wsum[lane] : 0;
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) {
        const int y = __shfl_up(t, o);
        if (lane >= o) t += y;
      }
      if (lane < 16) wsum[lane] = t;
    }
    __syncthreads();
    const int total = carry + (w > 0 ? wsum[w - 1] : 0) + x;
    if (i < B) cu[i + 1] = total;
    __syncthreads();
    if (tid == 1023) carry = total;
    __syncthreads();
  }
  if (tid == 0 && host_out) {
    __hip_atomic_store(host_out, carry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(host_out + 1, tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// Batches of at most 1 024 rows (the reference's are 8): pack_lens_kernel and pack_scan_kernel in ONE workgroup — the
// 16 waves count the rows' tokens into LDS, then the same scan.
__global__ __launch_bounds__(1024) void pack_head_kernel(const int64_t* __restrict__ mask, int B, int S, int max_len,
                                                         int32_t* __restrict__ cu, int32_t* __restrict__ err,
                                                         int32_t* host_out, int32_t tag) {
  __shared__ int lens[1024];
  __shared__ int wsum[16];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  for (int b = w; b < B; b += 16) {
    int n = S;
    if (mask) {
      n = 0;
      for (int j0 = 0; j0 < S; j0 += 64) {
        const int j = j0 + lane;
        n += __popcll(__ballot(j < S && mask[(size_t)b * S + j] != 0));
      }
    }
    if (lane == 0) lens[b] = n;
  }
  __syncthreads();
  int x = tid < B ? lens[tid] : 0;
  if (tid < B && (x < 1 || x > max_len)) atomicOr(err, 4);
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int y = __shfl_up(x, o);
    if (lane >= o) x += y;
  }
  if (lane == 63) wsum[w] = x;
  __syncthreads();
  if (w == 0) {
    int t = lane < 16 ? wsum[lane] : 0;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
      const int y = __shfl_up(t, o);
      if (lane >= o) t += y;
    }
    if (lane < 16) wsum[lane] = t;
  }
  __syncthreads();
  const int total = (w > 0 ? wsum[w - 1] : 0) + x;
  if (tid == 0) cu[0] = 0;
  if (tid < B) cu[tid + 1] = total;
  if (tid == B - 1 && host_out) {
    __hip_atomic_store(host_out, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(host_out + 1, tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// The engine's error word and then `tag` to mapped host memory, behind a forward (same purpose as above).
__global__ void publish_word_kernel(const int32_t* __restrict__ word, int32_t* host_out, int32_t tag) {
  __hip_atomic_store(host_out, __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(host_out + 1, tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// One wave per row: stable compaction of the kept tokens to [cu[b], cu[b + 1]).  position ids: the caller's
// (row stride 0 = one row broadcast, modeling_bert.py:83), else the PADDED column index (transformers' default
// arange(S): a hole in the mask does not renumber what follows).
__global__ __launch_bounds__(256) void pack_tokens_kernel(const int64_t* __restrict__ ids,
                                                          const int64_t* __restrict__ mask,
                                                          const int64_t* __restrict__ tt,
                                                          const int64_t* __restrict__ pos, long pos_row_stride,
                                                          int B, int S, const int32_t* __restrict__ cu,
                                                          int32_t* __restrict__ out_ids,
                                                          int32_t* __restrict__ out_tt,
                                                          int32_t* __restrict__ out_pos, int sep_id, int n_ctx,
                                                          int32_t* __restrict__ focus, int32_t* __restrict__ err) {
  const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const int dst0 = cu[b], end = cu[b + 1];
  int count = 0, seps = 0, found = -1;
  for (int j0 = 0; j0 < S; j0 += 64) {
    const int j = j0 + lane;
    const size_t at = (size_t)b * S + j;
    const bool keep = j < S && (!mask || mask[at] != 0);
    const long long id = keep ? ids[at] : 0;
    const unsigned long long m = __ballot(keep);
    const int d = dst0 + count + __popcll(m & ((1ull << lane) - 1));
    if (keep && d < end) {   // d < end always (same mask as the length kernel); the guard bounds a racing caller
      out_ids[d] = (int32_t)id;
      if (out_tt) out_tt[d] = tt ? (int32_t)tt[at] : 0;
      out_pos[d] = pos ? (int32_t)pos[(size_t)b * pos_row_stride + j] : j;
    }
    // focus_index_kernel's search on the same pass (focus != nullptr): the packed index of the (n_ctx+1)-th separator
    if (focus && n_ctx >= 0) {
      const unsigned long long ms = __ballot(keep && id == sep_id);
      const int c = __popcll(ms);
      if (found < 0 && seps + c > n_ctx) {
        int k = n_ctx - seps;
        unsigned long long mm = ms;
        while (k-- > 0) mm &= mm - 1;
        const int bit = __ffsll((long long)mm) - 1;
        found = dst0 + count + __popcll(m & ((1ull << bit) - 1));
      }
      seps += c;
    }
    count += __popcll(m);
  }
  if (focus && lane == 0) {
    focus[b] = found >= 0 ? found : (dst0 < end ? dst0 : (dst0 > 0 ? dst0 - 1 : 0));   // (empty row: in bounds)
    if (n_ctx >= 0 && seps != 2 * n_ctx + 1) atomicOr(err, 1);
  }
}

// Last kernel of a forward: a non-zero device error word (fused-LayerNorm exchange timed out, bad row
// table) turns every logit of the call into NaN, so that a void result cannot be mistaken for scores
// even by a caller that never polls clip_poll_errors.
// A call whose own logits are not finite (an fp16 activation beyond 65504 -> inf -> NaN, a NaN weight) sets
// bit 3 of the word: `p > threshold` is False for NaN, so such a sentence would otherwise just be MISSING from
// ttp_doc_sents.csv and the top-k file, with exit status 0.
__global__ __launch_bounds__(256) void poison_logits_kernel(int32_t* __restrict__ err,
                                                            float* __restrict__ logits, int count) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= count) return;
  if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) logits[i] = __builtin_nanf("");
  else if (!__builtin_isfinite(logits[i])) atomicOr(err, 8);
}

// CrossEntropyLoss rows (ref: sentclf/bert_model.py:177-182, torch CrossEntropyLoss = log_softmax + NLL):
// row_loss[i] = logsumexp(scores[i, 0:V]) - scores[i, labels[i]], one workgroup per row, fp32.
// copy (optional) receives the first V columns of the row, dense [n, V] (the scores may be padded).
__global__ __launch_bounds__(256) void ce_rows_kernel(const float* __restrict__ scores, int ld, int V,
                                                      const int32_t* __restrict__ labels,
                                                      float* __restrict__ row_loss,
                                                      float* __restrict__ copy) {
  __shared__ float red[4];
  const int i = blockIdx.x, tid = threadIdx.x;
  const float* row = scores + (size_t)i * ld;
  float mx = -INFINITY;
  for (int c = tid; c < V; c += 256) {
    const float v = row[c];
    mx = fmaxf(mx, v);
    if (copy) copy[(size_t)i * V + c] = v;
  }
  mx = wave_max(mx);
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float sum = 0.f;
  for (int c = tid; c < V; c += 256) sum += expf(row[c] - mx);
  sum = wave_sum(sum);
  if ((tid & 63) == 0) red[tid >> 6] = sum;
  __syncthreads();
  if (tid == 0) {
    const int lab = labels[i];
    const float lse = mx + logf((red[0] + red[1]) + (red[2] + red[3]));
    row_loss[i] = (lab >= 0 && lab < V) ? lse - row[lab] : __builtin_nanf("");
  }
}

// loss[0] = (accumulate ? loss[0] : 0) + sum(row_loss[0:n]) / n, one workgroup, fixed summation order
// (deterministic); n == 0 gives NaN as torch's mean over no elements does.
__global__ __launch_bounds__(256) void mean_loss_kernel(const float* __restrict__ row_loss, int n,
                                                        float* __restrict__ loss, int accumulate) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += row_loss[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float m = n > 0 ? ((red[0] + red[1]) + (red[2] + red[3])) / (float)n : __builtin_nanf("");
    loss[0] = (accumulate ? loss[0] : 0.f) + m;
  }
}

template <typename F>
hipError_t ln_dispatch(int H, bool f32, F&& f) {
#define LN_CASE(h, v, n) \
  case h: return f32 ? f.template operator()<v, n, true>() : f.template operator()<v, n, false>();
  switch (H) {
    LN_CASE(128, 2, 1)
    LN_CASE(256, 4, 1)
    LN_CASE(384, 2, 3)
    LN_CASE(512, 4, 2)
    LN_CASE(768, 4, 3)
    LN_CASE(1024, 4, 4)
    default: return hipErrorInvalidValue;
  }
#undef LN_CASE
}

}  // namespace

hipError_t KSET::launch_layernorm(const float* x, const float* gamma, const float* beta, float eps,
                            void* out, int T, int H, hipStream_t st, bool out_f32, float* stats,
                            uint16_t* res16) {
  if (T <= 0) return hipSuccess;
  return ln_dispatch(H, out_f32, [&]<int VEC, int NIT, bool F32>() {
    hipLaunchKernelGGL((layernorm_kernel<VEC, NIT, F32>), dim3((T + 3) / 4), dim3(256), 0, st, x,
                       gamma, beta, eps, out, T, H, reinterpret_cast<float2*>(stats), res16);
    return hipGetLastError();
  });
}

hipError_t KSET::launch_embed_ln(const int32_t* ids, const int32_t* type_ids, const int32_t* pos_ids,
                           const float* word, const float* pos, const float* type,
                           const float* gamma, const float* beta, float eps, void* out, int T,
                           int H, int vocab, int max_pos, int type_vocab, hipStream_t st,
                           bool out_f32, float* pre_out, float* stats, uint16_t* res16) {
  if (T <= 0) return hipSuccess;
  return ln_dispatch(H, out_f32, [&]<int VEC, int NIT, bool F32>() {
    hipLaunchKernelGGL((embed_ln_kernel<VEC, NIT, F32>), dim3((T + 3) / 4), dim3(256), 0, st, ids,
                       type_ids, pos_ids, word, pos, type, gamma, beta, eps, out, T, H, vocab,
                       max_pos, type_vocab, pre_out, reinterpret_cast<float2*>(stats), res16);
    return hipGetLastError();
  });
}

#if !CLIP_OPERAND_F16
hipError_t launch_gather_rows(const void* src, const int32_t* rows, void* dst, int n, int H,
                              hipStream_t st, bool f32) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(gather_rows_kernel, dim3((n + 3) / 4), dim3(256), 0, st,
                     reinterpret_cast<const char*>(src), rows, reinterpret_cast<char*>(dst), n,
                     H * (f32 ? 4 : 2));
  return hipGetLastError();
}

#endif

hipError_t KSET::launch_ln_head(const float* x, const float* gamma, const float* beta, float eps, int n, int H,
                                const float* Wc, const float* bc, int L, float* logits, int32_t* err, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  if (L > 8 || L < 1) return hipErrorInvalidValue;
  return ln_dispatch(H, false, [&]<int VEC, int NIT, bool F32>() {
    hipLaunchKernelGGL((ln_head_kernel<VEC, NIT>), dim3((n + 3) / 4), dim3(256), 0, st, x, gamma, beta, eps, n, H, Wc,
                       bc, L, logits, err);
    return hipGetLastError();
  });
}

hipError_t KSET::launch_classifier(const void* x, const int32_t* rows, const float* Wc,
                             const float* bc, float* logits, int n, int H, int L, hipStream_t st,
                             bool f32) {
  if (n <= 0) return hipSuccess;
  if (L > 8 || H % 128) return hipErrorInvalidValue;
  if (f32)
    hipLaunchKernelGGL(classifier_kernel<true>, dim3((n + 3) / 4), dim3(256), 0, st, x, rows, Wc,
                       bc, logits, n, H, L);
  else
    hipLaunchKernelGGL(classifier_kernel<false>, dim3((n + 3) / 4), dim3(256), 0, st, x, rows, Wc,
                       bc, logits, n, H, L);
  return hipGetLastError();
}

#if !CLIP_OPERAND_F16
hipError_t launch_focus_index(const int32_t* ids, const int32_t* cu_seqlens, int n_seqs,
                              int sep_id, int n_ctx, int32_t* focus, int32_t* err,
                              hipStream_t st) {
  if (n_seqs <= 0) return hipSuccess;
  hipLaunchKernelGGL(focus_index_kernel, dim3(n_seqs), dim3(64), 0, st, ids, cu_seqlens, n_seqs,
                     sep_id, n_ctx, focus, err);
  return hipGetLastError();
}

hipError_t launch_prepare_packed(const int32_t* ids, const int32_t* cu_seqlens, int n_seqs,
                                 int sep_id, int n_ctx, int32_t* type_ids, int32_t* pos_ids,
                                 int32_t* focus, int32_t* err, hipStream_t st) {
  if (n_seqs <= 0) return hipSuccess;
  hipLaunchKernelGGL(prepare_packed_kernel, dim3(n_seqs), dim3(64), 0, st, ids, cu_seqlens, sep_id,
                     n_ctx, type_ids, pos_ids, focus, err);
  return hipGetLastError();
}

hipError_t launch_epilogue(const float* logits, const double* label_thresh, float any_thresh,
                           int n, int L, float* probs, float* max_score, uint8_t* labels,
                           uint8_t* keep, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  if (L > 8 || L < 1) return hipErrorInvalidValue;
  hipLaunchKernelGGL(epilogue_kernel, dim3((n + 31) / 32), dim3(256), 0, st, logits, label_thresh,
                     any_thresh, n, L, probs, max_score, labels, keep);
  return hipGetLastError();
}

hipError_t launch_bce_loss(const float* logits, const float* targets, int count, float* loss,
                           hipStream_t st) {
  if (count <= 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(bce_loss_kernel, dim3(1), dim3(256), 0, st, logits, targets, count, loss);
  return hipGetLastError();
}

hipError_t launch_ce_rows(const float* scores, int ld, int V, const int32_t* labels, int n,
                          float* row_loss, float* copy, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(ce_rows_kernel, dim3(n), dim3(256), 0, st, scores, ld, V, labels, row_loss, copy);
  return hipGetLastError();
}

hipError_t launch_mean_loss(const float* row_loss, int n, float* loss, int accumulate, hipStream_t st) {
  hipLaunchKernelGGL(mean_loss_kernel, dim3(1), dim3(256), 0, st, row_loss, n, loss, accumulate);
  return hipGetLastError();
}

hipError_t launch_check_rows(const int32_t* cu_seqlens, int n_seqs, int total_tokens, int max_seqlen,
                             int32_t* err, hipStream_t st) {
  if (n_seqs <= 0) return hipSuccess;
  hipLaunchKernelGGL(check_rows_kernel, dim3((n_seqs + 255) / 256), dim3(256), 0, st, cu_seqlens, n_seqs,
                     total_tokens, max_seqlen, err);
  return hipGetLastError();
}

hipError_t launch_pack_lengths(const int64_t* mask, int B, int S, int max_len, int32_t* cu, int32_t* err,
                               int32_t* host_out, int32_t tag, hipStream_t st) {
  if (B <= 0) return hipSuccess;
  if (B <= 1024) {
    hipLaunchKernelGGL(pack_head_kernel, dim3(1), dim3(1024), 0, st, mask, B, S, max_len, cu, err, host_out, tag);
  } else {
    hipLaunchKernelGGL(pack_lens_kernel, dim3((B + 3) / 4), dim3(256), 0, st, mask, B, S, cu);
    hipLaunchKernelGGL(pack_scan_kernel, dim3(1), dim3(1024), 0, st, cu, B, max_len, err, host_out, tag);
  }
  return hipGetLastError();
}

hipError_t launch_publish_word(const int32_t* word, int32_t* host_out, int32_t tag, hipStream_t st) {
  hipLaunchKernelGGL(publish_word_kernel, dim3(1), dim3(1), 0, st, word, host_out, tag);
  return hipGetLastError();
}

hipError_t launch_pack_tokens(const int64_t* ids, const int64_t* mask, const int64_t* tt, const int64_t* pos,
                              long pos_row_stride, int B, int S, const int32_t* cu, int32_t* out_ids,
                              int32_t* out_tt, int32_t* out_pos, int sep_id, int n_ctx, int32_t* focus, int32_t* err,
                              hipStream_t st) {
  if (B <= 0) return hipSuccess;
  hipLaunchKernelGGL(pack_tokens_kernel, dim3((B + 3) / 4), dim3(256), 0, st, ids, mask, tt, pos, pos_row_stride,
                     B, S, cu, out_ids, out_tt, out_pos, sep_id, n_ctx, focus, err);
  return hipGetLastError();
}

hipError_t launch_poison_logits(int32_t* err, float* logits, int count, hipStream_t st) {
  if (count <= 0) return hipSuccess;
  hipLaunchKernelGGL(poison_logits_kernel, dim3((count + 255) / 256), dim3(256), 0, st, err, logits, count);
  return hipGetLastError();
}
#endif  // !CLIP_OPERAND_F16
