// libclip_tok.so — host-side WordPiece front end (include/clip_tok.h).  Plain C++20, no GPU code.
//
// Restates what the reference's tokenizer call does per sentence (/root/reference/sentclf/bert.py:57,
// 105,152 -> HF BertTokenizer): BertNormalizer (drop NUL, U+FFFD and control characters, whitespace ->
// ' ', spaces around CJK ideographs, and when lower-casing: NFD, drop nonspacing marks, lower-case),
// BertPreTokenizer (split on whitespace, every punctuation character its own word), greedy
// longest-match WordPiece ("##" continuations, words over 100 characters -> [UNK]) — and, in
// CLIP_TOK_LIST_CELL mode, the `eval(row.sentence)` + ' '.join + .lower() of
// sentclf/neural_baselines.py:197-201 / bert.py:100 on the CSV cell itself.
// Pure-ASCII strings (MIMIC notes) take a byte path; other strings take a code-point path over tables
// generated from Python's unicodedata (unicode_tables.h, tools/gen_unicode_tables.py) — the source the
// Python restatement consults.  What is still outside the scope is flagged for that restatement
// (clip_amd/wordpiece.py), never guessed at: capital sigma (its lower case depends on context), invalid
// UTF-8, list-literal syntax beyond plain quoted strings, and every non-ASCII string when the caller
// says its Unicode tables are of another version (CLIP_TOK_ASCII_ONLY).
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <string_view>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/clip_tok.h"
#include "unicode_tables.h"

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

struct SvHash {
  using is_transparent = void;
  size_t operator()(std::string_view s) const { return std::hash<std::string_view>{}(s); }
  size_t operator()(const std::string& s) const { return std::hash<std::string_view>{}(s); }
};
struct SvEq {
  using is_transparent = void;
  bool operator()(std::string_view a, std::string_view b) const { return a == b; }
};
using Vocab = std::unordered_map<std::string, int32_t, SvHash, SvEq>;

const char* const SPECIALS[5] = {"[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"};

inline bool is_dropped(unsigned char c) {   // NUL and category Cc, except \t \n \r
  return (c < 0x20 && c != '\t' && c != '\n' && c != '\r') || c == 0x7f;
}
inline bool is_space(unsigned char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r'; }
inline bool is_punct(unsigned char c) {
  return (c >= 33 && c <= 47) || (c >= 58 && c <= 64) || (c >= 91 && c <= 96) || (c >= 123 && c <= 126);
}

}  // namespace

struct clip_tokenizer {
  Vocab vocab;
  int32_t special[5];
  bool lower = true;
  int max_chars = 100;
  size_t max_piece = 0;   // longest vocabulary entry, bytes
};

namespace {

// per-thread cache of whole words -> ids (words repeat: clinical text is Zipfian)
struct WordCache {
  std::unordered_map<std::string, std::pair<uint32_t, uint32_t>, SvHash, SvEq> map;
  std::vector<int32_t> pool;
};

// `bounds` (ncp + 1 byte offsets of the code points) is null for ASCII words: byte = character
void wordpiece(const clip_tokenizer& t, std::string_view word, WordCache& cache, std::vector<int32_t>& out,
               const uint32_t* bounds = nullptr, size_t ncp = 0) {
  auto hit = cache.map.find(word);
  if (hit != cache.map.end()) {
    out.insert(out.end(), cache.pool.begin() + hit->second.first,
               cache.pool.begin() + hit->second.first + hit->second.second);
    return;
  }
  const size_t first = out.size();
  const size_t n = bounds ? ncp : word.size();
  auto at = [&](size_t i) -> size_t { return bounds ? bounds[i] : i; };
  if ((int)n > t.max_chars) {
    out.push_back(t.special[1]);
  } else {
    std::string buf = "##";
    size_t start = 0;
    bool bad = false;
    while (start < n) {
      const size_t pre = start > 0 ? 2 : 0;
      size_t end = n;
      int32_t cur = -1;
      for (; end > start; --end) {
        const size_t len = at(end) - at(start);
        if (len + pre > t.max_piece) continue;       // longer than every vocabulary entry
        std::string_view sub;
        if (pre) {
          buf.resize(2);
          buf.append(word.data() + at(start), len);
          sub = buf;
        } else {
          sub = word.substr(0, len);
        }
        auto it = t.vocab.find(sub);
        if (it != t.vocab.end()) { cur = it->second; break; }
      }
      if (cur < 0) { bad = true; break; }
      out.push_back(cur);
      start = end;
    }
    if (bad) {
      out.resize(first);
      out.push_back(t.special[1]);
    }
  }
  if (cache.map.size() > (1u << 20)) { cache.map.clear(); cache.pool.clear(); }
  const uint32_t off = (uint32_t)cache.pool.size(), len = (uint32_t)(out.size() - first);
  cache.pool.insert(cache.pool.end(), out.begin() + first, out.end());
  cache.map.emplace(std::string(word), std::make_pair(off, len));
}

// ---------------------------------------------------------------- code-point path (non-ASCII text)
bool in_ranges(const uint32_t (*r)[2], int n, uint32_t cp) {
  int lo = 0, hi = n - 1;
  while (lo <= hi) {
    const int mid = (lo + hi) >> 1;
    if (cp < r[mid][0]) hi = mid - 1;
    else if (cp > r[mid][1]) lo = mid + 1;
    else return true;
  }
  return false;
}
uint32_t ccc_of(uint32_t cp) {
  int lo = 0, hi = UC_CCC_N - 1;
  while (lo <= hi) {
    const int mid = (lo + hi) >> 1;
    if (cp < UC_CCC[mid][0]) hi = mid - 1;
    else if (cp > UC_CCC[mid][1]) lo = mid + 1;
    else return UC_CCC[mid][2];
  }
  return 0;
}
template <int W>
const uint32_t* find_cp(const uint32_t (*tab)[W], int n, uint32_t cp) {
  int lo = 0, hi = n - 1;
  while (lo <= hi) {
    const int mid = (lo + hi) >> 1;
    if (cp < tab[mid][0]) hi = mid - 1;
    else if (cp > tab[mid][0]) lo = mid + 1;
    else return tab[mid];
  }
  return nullptr;
}
inline bool is_cjk(uint32_t cp) {
  return (cp >= 0x4E00 && cp <= 0x9FFF) || (cp >= 0x3400 && cp <= 0x4DBF) || (cp >= 0x20000 && cp <= 0x2A6DF) ||
         (cp >= 0x2A700 && cp <= 0x2B73F) || (cp >= 0x2B740 && cp <= 0x2B81F) || (cp >= 0x2B820 && cp <= 0x2CEAF) ||
         (cp >= 0xF900 && cp <= 0xFAFF) || (cp >= 0x2F800 && cp <= 0x2FA1F);
}
inline bool u_dropped(uint32_t cp) { return cp < 0x80 ? is_dropped((unsigned char)cp) : in_ranges(UC_DROP, UC_DROP_N, cp); }
inline bool u_white(uint32_t cp) { return cp < 0x80 ? is_space((unsigned char)cp) : in_ranges(UC_ZS, UC_ZS_N, cp); }
inline bool u_isspace(uint32_t cp) {   // str.isspace()
  return cp < 0x80 ? (cp == ' ' || (cp >= 9 && cp <= 13) || (cp >= 0x1c && cp <= 0x1f)) : in_ranges(UC_ISSPACE, UC_ISSPACE_N, cp);
}
inline bool u_punct(uint32_t cp) { return cp < 0x80 ? is_punct((unsigned char)cp) : in_ranges(UC_PUNCT, UC_PUNCT_N, cp); }

bool decode_utf8(const char* s, size_t n, std::vector<uint32_t>& out) {
  out.clear();
  size_t i = 0;
  while (i < n) {
    const unsigned char c = (unsigned char)s[i];
    uint32_t cp;
    int extra;
    if (c < 0x80) { cp = c; extra = 0; }
    else if ((c & 0xE0) == 0xC0) { cp = c & 0x1F; extra = 1; }
    else if ((c & 0xF0) == 0xE0) { cp = c & 0x0F; extra = 2; }
    else if ((c & 0xF8) == 0xF0) { cp = c & 0x07; extra = 3; }
    else return false;
    for (int k = 1; k <= extra; ++k) {
      if (i + k >= n) return false;
      const unsigned char d = (unsigned char)s[i + k];
      if ((d & 0xC0) != 0x80) return false;
      cp = (cp << 6) | (d & 0x3F);
    }
    static const uint32_t min_cp[4] = {0, 0x80, 0x800, 0x10000};
    if (cp < min_cp[extra] || cp > 0x10FFFF || (cp >= 0xD800 && cp <= 0xDFFF)) return false;
    out.push_back(cp);
    i += extra + 1;
  }
  return true;
}
void append_utf8(std::string& s, uint32_t cp) {
  if (cp < 0x80) s.push_back((char)cp);
  else if (cp < 0x800) { s.push_back((char)(0xC0 | (cp >> 6))); s.push_back((char)(0x80 | (cp & 0x3F))); }
  else if (cp < 0x10000) {
    s.push_back((char)(0xE0 | (cp >> 12))); s.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); s.push_back((char)(0x80 | (cp & 0x3F)));
  } else {
    s.push_back((char)(0xF0 | (cp >> 18))); s.push_back((char)(0x80 | ((cp >> 12) & 0x3F)));
    s.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); s.push_back((char)(0x80 | (cp & 0x3F)));
  }
}

// str.lower() of a code-point string; false when it holds a capital sigma (context-dependent)
bool lower_cps(const std::vector<uint32_t>& in, std::vector<uint32_t>& out) {
  out.clear();
  for (uint32_t cp : in) {
    if (cp == 0x3A3) return false;
    if (cp < 0x80) { out.push_back((cp >= 'A' && cp <= 'Z') ? cp + 32 : cp); continue; }
    if (const uint32_t* m = find_cp<5>(UC_LOWER_MULTI, UC_LOWER_MULTI_N, cp)) {
      for (uint32_t k = 0; k < m[1]; ++k) out.push_back(m[2 + k]);
      continue;
    }
    const uint32_t* l = find_cp<2>(UC_LOWER, UC_LOWER_N, cp);
    out.push_back(l ? l[1] : cp);
  }
  return true;
}

struct UScratch {
  std::vector<uint32_t> a, b, word;
  std::vector<uint32_t> bounds;
  std::string bytes;
};

// normalise + pre-tokenise + WordPiece one run given as code points; false = leave to the caller
bool encode_run_u(const clip_tokenizer& t, const std::vector<uint32_t>& cps, bool lower, WordCache& cache,
                  UScratch& u, std::vector<int32_t>& out) {
  std::vector<uint32_t>& a = u.a;
  std::vector<uint32_t>& b = u.b;
  a.clear();
  for (uint32_t cp : cps) {                                  // clean text + CJK spacing
    if (cp == 0 || cp == 0xFFFD || u_dropped(cp)) continue;
    if (u_white(cp)) { a.push_back(' '); continue; }
    if (is_cjk(cp)) { a.push_back(' '); a.push_back(cp); a.push_back(' '); }
    else a.push_back(cp);
  }
  if (lower) {
    b.clear();
    for (uint32_t cp : a) {                                  // NFD: full canonical decomposition ...
      if (cp >= 0xAC00 && cp <= 0xD7A3) {                    // Hangul syllables, algorithmically
        const uint32_t si = cp - 0xAC00, tj = si % 28;
        b.push_back(0x1100 + si / 588);
        b.push_back(0x1161 + (si % 588) / 28);
        if (tj) b.push_back(0x11A7 + tj);
      } else if (cp < 0xC0) {
        b.push_back(cp);
      } else if (const uint32_t* d = find_cp<3>(UC_NFD_IDX, UC_NFD_IDX_N, cp)) {
        for (uint32_t k = 0; k < d[2]; ++k) b.push_back(UC_NFD_DATA[d[1] + k]);
      } else {
        b.push_back(cp);
      }
    }
    for (size_t i = 1; i < b.size(); ++i) {                  // ... in canonical order (stable by class)
      const uint32_t ci = b[i] < 0x300 ? 0 : ccc_of(b[i]);
      if (!ci) continue;
      size_t j = i;
      while (j > 0) {
        const uint32_t cj = b[j - 1] < 0x300 ? 0 : ccc_of(b[j - 1]);
        if (cj <= ci) break;                                 // cj == 0 is a starter: stop
        std::swap(b[j], b[j - 1]);
        --j;
      }
    }
    a.clear();
    for (uint32_t cp : b)                                    // strip accents
      if (cp < 0x300 || !in_ranges(UC_MN, UC_MN_N, cp)) a.push_back(cp);
    if (!lower_cps(a, b)) return false;
    a.swap(b);
  }
  auto flush = [&]() {
    if (u.word.empty()) return;
    u.bytes.clear();
    u.bounds.clear();
    bool ascii = true;
    for (uint32_t cp : u.word) {
      u.bounds.push_back((uint32_t)u.bytes.size());
      append_utf8(u.bytes, cp);
      ascii = ascii && cp < 0x80;
    }
    u.bounds.push_back((uint32_t)u.bytes.size());
    if (ascii) wordpiece(t, u.bytes, cache, out);
    else wordpiece(t, u.bytes, cache, out, u.bounds.data(), u.word.size());
    u.word.clear();
  };
  u.word.clear();
  for (uint32_t cp : a) {
    if (u_isspace(cp)) flush();
    else if (u_punct(cp)) { flush(); u.word.push_back(cp); flush(); }
    else u.word.push_back(cp);
  }
  flush();
  return true;
}

inline bool all_ascii(const char* s, size_t n) {
  for (size_t k = 0; k < n; ++k)
    if ((unsigned char)s[k] >= 0x80) return false;
  return true;
}

// normalise + pre-tokenise + WordPiece one run of ASCII text (no special tokens inside)
void encode_run(const clip_tokenizer& t, const char* s, size_t n, bool lower, WordCache& cache,
                std::string& word, std::vector<int32_t>& out) {
  word.clear();
  for (size_t i = 0; i < n; ++i) {
    unsigned char c = (unsigned char)s[i];
    if (is_dropped(c)) continue;
    if (is_space(c)) {
      if (!word.empty()) { wordpiece(t, word, cache, out); word.clear(); }
    } else if (is_punct(c)) {
      if (!word.empty()) { wordpiece(t, word, cache, out); word.clear(); }
      const char p = (char)c;
      wordpiece(t, std::string_view(&p, 1), cache, out);
    } else {
      word.push_back((char)((lower && c >= 'A' && c <= 'Z') ? c + 32 : c));
    }
  }
  if (!word.empty()) { wordpiece(t, word, cache, out); word.clear(); }
}

// one run of text without special tokens: byte path when it is ASCII, code-point path otherwise
bool encode_any_run(const clip_tokenizer& t, const char* s, size_t n, bool ascii_only, WordCache& cache,
                    std::string& word, UScratch& u, std::vector<uint32_t>& cps, std::vector<int32_t>& out) {
  if (all_ascii(s, n)) {
    encode_run(t, s, n, t.lower, cache, word, out);
    return true;
  }
  if (ascii_only || !decode_utf8(s, n, cps)) return false;
  return encode_run_u(t, cps, t.lower, cache, u, out);
}

// CLIP_TOK_TEXT: literal special tokens are cut out first, as HF's added-token matching does
bool encode_text(const clip_tokenizer& t, const char* s, size_t n, bool ascii_only, WordCache& cache,
                 std::string& word, UScratch& u, std::vector<uint32_t>& cps, std::vector<int32_t>& out) {
  size_t run = 0, i = 0;
  while (i < n) {
    if (s[i] == '[') {
      int which = -1;
      for (int k = 0; k < 5; ++k) {
        const size_t L = strlen(SPECIALS[k]);
        if (i + L <= n && memcmp(s + i, SPECIALS[k], L) == 0) { which = k; break; }
      }
      if (which >= 0) {
        if (!encode_any_run(t, s + run, i - run, ascii_only, cache, word, u, cps, out)) return false;
        out.push_back(t.special[which]);
        i += strlen(SPECIALS[which]);
        run = i;
        continue;
      }
    }
    ++i;
  }
  return encode_any_run(t, s + run, n - run, ascii_only, cache, word, u, cps, out);
}

// CLIP_TOK_LIST_CELL: "['Pt', 'should', \"don't\"]" -> "Pt should don't".  Returns false when the cell
// needs the caller's Python path (not a list literal, or syntax beyond plain quoted strings).
bool parse_list_cell(const char* s, size_t n, std::string& text) {
  text.clear();
  if (n < 2 || s[0] != '[' || s[n - 1] != ']') return false;
  size_t i = 1;
  const size_t end = n - 1;
  auto skip_ws = [&]() { while (i < end && (s[i] == ' ' || s[i] == '\t')) ++i; };
  bool first = true;
  skip_ws();
  if (i == end) return true;   // []
  while (true) {
    if (i >= end) return false;
    const char q = s[i];
    if (q != '\'' && q != '"') return false;                              // number, None, prefix, nesting
    if (i + 2 < end && s[i + 1] == q && s[i + 2] == q) return false;      // triple-quoted
    ++i;
    if (!first) text.push_back(' ');
    first = false;
    while (true) {
      if (i >= end) return false;                                         // unterminated
      const char c = s[i];
      if (c == q) { ++i; break; }
      if (c == '\n' || c == '\r') return false;
      if (c == '\\') {
        if (i + 1 >= end) return false;
        const char e = s[i + 1];
        if (e == '\\' || e == '\'' || e == '"') text.push_back(e);
        else if (e == 'n') text.push_back('\n');
        else if (e == 't') text.push_back('\t');
        else if (e == 'r') text.push_back('\r');
        else return false;                                                // \x \u \N octal ...: Python's job
        i += 2;
        continue;
      }
      text.push_back(c);
      ++i;
    }
    skip_ws();
    if (i == end) return true;
    if (s[i] != ',') return false;                                        // adjacent literals, operators
    ++i;
    skip_ws();
    if (i == end) return true;                                            // trailing comma
  }
}

struct Chunk {
  std::vector<int32_t> ids;
  std::vector<int64_t> lens;
};

void run_chunk(const clip_tokenizer& t, const char* blob, const int64_t* off, int64_t a, int64_t b, int mode,
               uint8_t* fallback, Chunk& ch) {
  WordCache cache;
  UScratch u;
  std::vector<uint32_t> cps, low;
  std::string word, text;
  const bool ascii_only = (mode & CLIP_TOK_ASCII_ONLY) != 0;
  const int base = mode & 0xff;
  ch.lens.resize(b - a);
  for (int64_t i = a; i < b; ++i) {
    const char* s = blob + off[i];
    const size_t n = (size_t)(off[i + 1] - off[i]);
    const size_t before = ch.ids.size();
    bool ok;
    if (base == CLIP_TOK_LIST_CELL) {
      // .lower() precedes the tokenizer call (bert.py:100), so "[SEP]" inside a sentence is "[sep]" by
      // then: plain text, no special-token matching
      ok = (!ascii_only || all_ascii(s, n)) && parse_list_cell(s, n, text);
      if (ok) {
        if (all_ascii(text.data(), text.size())) {
          encode_run(t, text.data(), text.size(), true, cache, word, ch.ids);
        } else {
          ok = decode_utf8(text.data(), text.size(), cps) && lower_cps(cps, low) &&
               encode_run_u(t, low, t.lower, cache, u, ch.ids);
        }
      }
    } else {
      ok = encode_text(t, s, n, ascii_only, cache, word, u, cps, ch.ids);
    }
    fallback[i] = ok ? 0 : 1;
    if (!ok) ch.ids.resize(before);
    ch.lens[i - a] = (int64_t)(ch.ids.size() - before);
  }
}

}  // namespace

extern "C" {

const char* clip_tok_last_error(void) { return g_err.c_str(); }
const char* clip_tok_unicode_version(void) { return UC_VERSION; }

int clip_tok_create(const char* vocab_path, int32_t do_lower_case, clip_tokenizer** out) {
  if (!vocab_path || !out) return fail(CLIP_TOK_ERR_ARG, "clip_tok_create: null argument");
  std::ifstream f(vocab_path, std::ios::binary);
  if (!f) return fail(CLIP_TOK_ERR_IO, "clip_tok_create: cannot open '%s'", vocab_path);
  auto* t = new clip_tokenizer();
  t->lower = do_lower_case != 0;
  std::string line;
  int32_t id = 0;
  while (std::getline(f, line)) {
    if (!line.empty() && line.back() == '\r') line.pop_back();   // text-mode newline translation
    t->vocab[line] = id++;                                       // a repeated token keeps its LAST id
    t->max_piece = std::max(t->max_piece, line.size());
  }
  for (int k = 0; k < 5; ++k) {
    auto it = t->vocab.find(std::string_view(SPECIALS[k]));
    if (it == t->vocab.end()) {
      delete t;
      return fail(CLIP_TOK_ERR_ARG, "clip_tok_create: vocab lacks %s", SPECIALS[k]);
    }
    t->special[k] = it->second;
  }
  *out = t;
  return CLIP_TOK_OK;
}

int clip_tok_destroy(clip_tokenizer* t) {
  delete t;
  return CLIP_TOK_OK;
}

int clip_tok_vocab_size(const clip_tokenizer* t, int32_t* n) {
  if (!t || !n) return fail(CLIP_TOK_ERR_ARG, "clip_tok_vocab_size: null argument");
  *n = (int32_t)t->vocab.size();
  return CLIP_TOK_OK;
}

int clip_tok_special_ids(const clip_tokenizer* t, int32_t* ids) {
  if (!t || !ids) return fail(CLIP_TOK_ERR_ARG, "clip_tok_special_ids: null argument");
  memcpy(ids, t->special, sizeof t->special);
  return CLIP_TOK_OK;
}

int clip_tok_encode_plain(clip_tokenizer* t, const char* blob, const int64_t* off, int64_t n, int32_t mode,
                          int32_t n_threads, int32_t* ids, int64_t ids_cap, int64_t* ids_off, uint8_t* fallback) {
  if (!t || !off || !ids_off || !fallback || n < 0 || (n > 0 && (!blob || !ids)))
    return fail(CLIP_TOK_ERR_ARG, "clip_tok_encode_plain: null argument");
  if ((mode & ~CLIP_TOK_ASCII_ONLY) != CLIP_TOK_TEXT && (mode & ~CLIP_TOK_ASCII_ONLY) != CLIP_TOK_LIST_CELL)
    return fail(CLIP_TOK_ERR_ARG, "clip_tok_encode_plain: unknown mode %d", mode);
  for (int64_t i = 0; i < n; ++i)
    if (off[i + 1] < off[i]) return fail(CLIP_TOK_ERR_ARG, "clip_tok_encode_plain: offsets decrease at %lld", (long long)i);
  if (n > 0 && ids_cap < off[n] - off[0])
    return fail(CLIP_TOK_ERR_CAPACITY, "clip_tok_encode_plain: ids_cap %lld < %lld text bytes", (long long)ids_cap,
                (long long)(off[n] - off[0]));
  const int T = (int)std::max<int64_t>(1, std::min<int64_t>(n_threads, std::max<int64_t>(1, n / 256)));
  std::vector<Chunk> chunks(T);
  std::vector<std::thread> pool;
  // contiguous chunks of about equal BYTES (sentences differ in length by 100x)
  std::vector<int64_t> cut(T + 1, n);
  cut[0] = 0;
  if (n > 0) {
    const int64_t total = off[n] - off[0];
    for (int k = 1; k < T; ++k)
      cut[k] = std::lower_bound(off, off + n, off[0] + total * k / T) - off;
  }
  for (int k = 1; k < T; ++k)
    pool.emplace_back(run_chunk, std::cref(*t), blob, off, cut[k], cut[k + 1], mode, fallback, std::ref(chunks[k]));
  run_chunk(*t, blob, off, cut[0], cut[1], mode, fallback, chunks[0]);
  for (auto& th : pool) th.join();
  // the up-front check above rests on "tokens emitted <= input bytes" (every token consumes at least one
  // byte of its sentence); do not let a violation of that invariant become a write past the caller's buffer
  int64_t need = 0;
  for (int k = 0; k < T; ++k) need += (int64_t)chunks[k].ids.size();
  if (need > ids_cap)
    return fail(CLIP_TOK_ERR_CAPACITY, "clip_tok_encode_plain: %lld tokens exceed ids_cap %lld", (long long)need,
                (long long)ids_cap);
  int64_t pos = 0, row = 0;
  for (int k = 0; k < T; ++k) {
    for (int64_t l : chunks[k].lens) { ids_off[row++] = pos; pos += l; }
    if (!chunks[k].ids.empty())
      memcpy(ids + (pos - (int64_t)chunks[k].ids.size()), chunks[k].ids.data(), chunks[k].ids.size() * sizeof(int32_t));
  }
  ids_off[n] = pos;
  return CLIP_TOK_OK;
}

}  // extern "C"
