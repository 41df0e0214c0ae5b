// Host-only sanitizer fuzz of libclip_tok's C ABI (include/clip_tok.h): built by tests/test_tokenizer_sanitizers.py
// with -fsanitize=address,undefined TOGETHER with clip_amd/csrc/tokenizer.cpp and run as a plain executable.
// Random byte strings (valid and broken UTF-8, list-literal fragments, control bytes, long runs without
// spaces) in both text modes, several thread counts, tight and generous output capacities; checks the
// documented invariants (ids_off monotone, ids_off[n] <= bytes, ids inside the vocabulary, capacity errors
// instead of overruns).  Exit code 0 = no finding; the sanitizers abort otherwise.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>

#include "../../include/clip_tok.h"

int main(int argc, char** argv) {
  if (argc < 2) { std::fprintf(stderr, "usage: tok_fuzz vocab.txt [rounds]\n"); return 2; }
  const int rounds = argc > 2 ? std::atoi(argv[2]) : 300;
  clip_tokenizer* t = nullptr;
  if (clip_tok_create(argv[1], 1, &t) != CLIP_TOK_OK) { std::fprintf(stderr, "create: %s\n", clip_tok_last_error()); return 1; }
  int32_t vocab = 0;
  clip_tok_vocab_size(t, &vocab);
  std::mt19937_64 rng(12345);
  static const char* frags[] = {"['", "', '", "']", "\\'", "\\\\", "\"", "[SEP]", "[CLS]", "[MASK]", "##", " ", "\t", "\n",
                                "\xc3\xa9", "\xe4\xb8\xad", "\xf0\x9f\x98\x80", "\xcc\x81", "\xea\xb0\x80", "\xc3", "\xe4\xb8",
                                "\xf0\x9f", "\xff", "\xc0\x80", "\xed\xa0\x80", "the", "patient", "take", "2", ".", ",", "-",
                                "aaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaa"};
  const int n_frags = sizeof(frags) / sizeof(frags[0]);
  long total_ids = 0;
  for (int r = 0; r < rounds; ++r) {
    const int n = 1 + (int)(rng() % 40);
    std::string blob;
    std::vector<int64_t> off{0};
    for (int i = 0; i < n; ++i) {
      const int parts = (int)(rng() % 30);
      for (int p = 0; p < parts; ++p) {
        const uint64_t k = rng() % 10;
        if (k < 6) blob += frags[rng() % n_frags];
        else if (k < 8) blob += (char)(rng() & 0xff);                       // any byte, NUL included
        else for (int q = 0, m = (int)(rng() % 8); q < m; ++q) blob += (char)('a' + rng() % 26);
      }
      off.push_back((int64_t)blob.size());
    }
    const int32_t mode = (int32_t)(rng() % 2) | ((rng() % 3 == 0) ? CLIP_TOK_ASCII_ONLY : 0);
    const int32_t threads = (int32_t)(rng() % 5);
    const int64_t bytes = off.back();
    const bool tight = rng() % 4 == 0;
    const int64_t cap = tight ? (int64_t)(rng() % (bytes + 1)) : bytes;
    std::vector<int32_t> ids((size_t)cap + 1, -7);       // one guard element past the capacity
    std::vector<int64_t> ids_off((size_t)n + 1, -1);
    std::vector<uint8_t> fb((size_t)n, 9);
    const int rc = clip_tok_encode_plain(t, blob.data(), off.data(), n, mode, threads, ids.data(), cap, ids_off.data(), fb.data());
    if (ids[(size_t)cap] != -7) { std::fprintf(stderr, "round %d: wrote past ids_cap\n", r); return 1; }
    if (rc == CLIP_TOK_ERR_CAPACITY && cap < bytes) continue;               // refused, nothing overrun
    if (rc != CLIP_TOK_OK) { std::fprintf(stderr, "round %d: rc %d (%s)\n", r, rc, clip_tok_last_error()); return 1; }
    if (ids_off[0] != 0) { std::fprintf(stderr, "round %d: ids_off[0] = %ld\n", r, (long)ids_off[0]); return 1; }
    for (int i = 0; i < n; ++i) {
      if (ids_off[i + 1] < ids_off[i] || fb[i] > 1) { std::fprintf(stderr, "round %d: offsets / flags\n", r); return 1; }
      if (fb[i] && ids_off[i + 1] != ids_off[i]) { std::fprintf(stderr, "round %d: ids for a flagged string\n", r); return 1; }
      if (ids_off[i + 1] - ids_off[i] > off[i + 1] - off[i]) { std::fprintf(stderr, "round %d: more ids than bytes\n", r); return 1; }
    }
    if (ids_off[n] > cap) { std::fprintf(stderr, "round %d: ids_off[n] beyond capacity\n", r); return 1; }
    for (int64_t k = 0; k < ids_off[n]; ++k)
      if (ids[(size_t)k] < 0 || ids[(size_t)k] >= vocab) { std::fprintf(stderr, "round %d: id out of range\n", r); return 1; }
    total_ids += ids_off[n];
  }
  // argument errors are errors, not crashes
  int64_t o2[2] = {0, 3};
  int32_t one[4];
  int64_t io[2];
  uint8_t f1[1];
  if (clip_tok_encode_plain(nullptr, "abc", o2, 1, 0, 1, one, 4, io, f1) == CLIP_TOK_OK) return 1;
  if (clip_tok_encode_plain(t, nullptr, o2, 1, 0, 1, one, 4, io, f1) == CLIP_TOK_OK) return 1;
  if (clip_tok_encode_plain(t, "abc", o2, 1, 0, 1, one, 0, io, f1) != CLIP_TOK_ERR_CAPACITY) return 1;
  clip_tok_destroy(t);
  std::printf("tok_fuzz: %d rounds, %ld ids, no finding\n", rounds, total_ids);
  return 0;
}
