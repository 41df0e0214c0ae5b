// libclip_tok.so — host-side WordPiece front end (include/clip_tok.h).  Plain C++20, no GPU code.
//
// Restates, for 7-bit ASCII text, what the reference's tokenizer call does per sentence
// (/root/reference/sentclf/bert.py:57,105,152 -> HF BertTokenizer): BertNormalizer (drop NUL and
// control characters, whitespace -> ' ', lower-case; accent stripping and CJK spacing cannot apply to
// ASCII), BertPreTokenizer (split on whitespace, every punctuation character its own word), greedy
// longest-match WordPiece ("##" continuations, words over 100 characters -> [UNK]) — and, in
// CLIP_TOK_LIST_CELL mode, the `eval(row.sentence)` + ' '.join + .lower() of
// sentclf/neural_baselines.py:197-201 / bert.py:100 on the CSV cell itself.  Anything outside that
// scope (a byte >= 0x80, list-literal syntax beyond plain quoted strings) is flagged for the
// caller's Python restatement (clip_amd/wordpiece.py), never guessed at.
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

void wordpiece(const clip_tokenizer& t, std::string_view word, WordCache& cache, std::vector<int32_t>& out) {
  auto hit = cache.map.find(word);
  if (hit != cache.map.end()) {
    out.insert(out.end(), cache.pool.begin() + hit->second.first,
               cache.pool.begin() + hit->second.first + hit->second.second);
    return;
  }
  const size_t first = out.size();
  const size_t n = word.size();
  if ((int)n > t.max_chars) {
    out.push_back(t.special[1]);
  } else {
    char buf[128];   // "##" + at most max_chars bytes
    buf[0] = buf[1] = '#';
    size_t start = 0;
    bool bad = false;
    while (start < n) {
      const size_t pre = start > 0 ? 2 : 0;
      size_t end = std::min(n, start + (t.max_piece > pre ? t.max_piece - pre : 0));
      int32_t cur = -1;
      for (; end > start; --end) {
        std::string_view sub;
        if (pre) {
          memcpy(buf + 2, word.data() + start, end - start);
          sub = std::string_view(buf, 2 + end - start);
        } else {
          sub = word.substr(0, end);
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

// CLIP_TOK_TEXT: literal special tokens are cut out first, as HF's added-token matching does
void encode_text(const clip_tokenizer& t, const char* s, size_t n, WordCache& cache, std::string& word,
                 std::vector<int32_t>& out) {
  size_t run = 0, i = 0;
  while (i < n) {
    if (s[i] == '[') {
      int which = -1;
      for (int k = 0; k < 5; ++k) {
        const size_t L = strlen(SPECIALS[k]);
        if (i + L <= n && memcmp(s + i, SPECIALS[k], L) == 0) { which = k; break; }
      }
      if (which >= 0) {
        encode_run(t, s + run, i - run, t.lower, cache, word, out);
        out.push_back(t.special[which]);
        i += strlen(SPECIALS[which]);
        run = i;
        continue;
      }
    }
    ++i;
  }
  encode_run(t, s + run, n - run, t.lower, cache, word, out);
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
  std::string word, text;
  ch.lens.resize(b - a);
  for (int64_t i = a; i < b; ++i) {
    const char* s = blob + off[i];
    const size_t n = (size_t)(off[i + 1] - off[i]);
    const size_t before = ch.ids.size();
    bool ascii = true;
    for (size_t k = 0; k < n; ++k)
      if ((unsigned char)s[k] >= 0x80) { ascii = false; break; }
    bool ok = ascii;
    if (ok) {
      if (mode == CLIP_TOK_LIST_CELL) {
        ok = parse_list_cell(s, n, text);
        // .lower() precedes the tokenizer call (bert.py:100), so "[SEP]" inside a sentence is
        // "[sep]" by then: plain text, no special-token matching
        if (ok) encode_run(t, text.data(), text.size(), true, cache, word, ch.ids);
      } else {
        encode_text(t, s, n, cache, word, ch.ids);
      }
    }
    fallback[i] = ok ? 0 : 1;
    if (!ok) ch.ids.resize(before);
    ch.lens[i - a] = (int64_t)(ch.ids.size() - before);
  }
}

}  // namespace

extern "C" {

const char* clip_tok_last_error(void) { return g_err.c_str(); }

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
  if (mode != CLIP_TOK_TEXT && mode != CLIP_TOK_LIST_CELL)
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
