/* clip_tok.h — C ABI of clip_amd/lib/libclip_tok.so: the host-side WordPiece front end of the bulk
 * scoring path (host C++, no GPU code).
 *
 * What it replaces in the reference (asappresearch/clip, /root/reference):
 *   - sentclf/apply_bert_unlabeled.py:43-47   BertTokenizer.from_pretrained(...)           -> clip_tok_create
 *   - sentclf/neural_baselines.py:145-153,195-201  eval(row.sentence) per row, ' '.join, .lower()
 *   - sentclf/bert.py:57,105,152   tokenizer(text, padding=True, max_length=512, truncation=True),
 *     called >= 2(2n+1) times per sentence by ctx_collator                              -> clip_tok_encode_plain
 * The reference binds HF `tokenizers` (a Rust extension) through Python; the binding a maintainer
 * adds for this library is the ctypes stub in INTEGRATION.md §4 (clip_amd/fast_tokenizer.py).
 *
 * Scope: pure-ASCII strings (MIMIC notes are ASCII) take a byte path; other strings take a code-point
 * path (NFD, nonspacing marks dropped, lower-casing, Unicode whitespace / punctuation / CJK rules) over
 * tables generated from Python's unicodedata (clip_amd/csrc/unicode_tables.h).  What stays outside is
 * never guessed at: its `fallback` flag is set and the caller runs the Python restatement
 * (clip_amd/wordpiece.py, pinned by goldens from HF's tokenizer) on that one string — capital sigma
 * (context-dependent lower case), invalid UTF-8, list literals using syntax beyond plain quoted
 * strings, and every non-ASCII string when CLIP_TOK_ASCII_ONLY is set (the caller's Unicode tables
 * are of another version than clip_tok_unicode_version()).  Results are bit-identical to that
 * restatement (tests/test_fast_tokenizer.py).
 *
 * Conventions: `int` status (0 = ok, negative = error, message from clip_tok_last_error()); no C++
 * exceptions cross the ABI; strings are UTF-8 bytes in one blob with n+1 offsets (the Arrow
 * large_string layout); a tokenizer handle may be used by one call at a time.
 */
#ifndef CLIP_TOK_H
#define CLIP_TOK_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct clip_tokenizer clip_tokenizer;

enum { CLIP_TOK_OK = 0, CLIP_TOK_ERR_ARG = -1, CLIP_TOK_ERR_IO = -2, CLIP_TOK_ERR_CAPACITY = -4 };

/* text modes of clip_tok_encode_plain */
enum {
  CLIP_TOK_TEXT = 0,      /* the bytes are the text; "[SEP]" etc. written literally map to their ids */
  CLIP_TOK_LIST_CELL = 1, /* the bytes are a Python list-of-str literal, the `sentence` cell of the
                             reference's CSVs: text = ' '.join(items).lower()  (neural_baselines.py:
                             197-201 + bert.py:100) */
  CLIP_TOK_ASCII_ONLY = 256 /* OR-ed into the mode: flag every non-ASCII string instead of using the tables */
};

/* vocab.txt: one token per line, id = line number; must hold [PAD] [UNK] [CLS] [SEP] [MASK]. */
int clip_tok_create(const char* vocab_path, int32_t do_lower_case, clip_tokenizer** out);
int clip_tok_destroy(clip_tokenizer* t);
int clip_tok_vocab_size(const clip_tokenizer* t, int32_t* n);
/* ids[5] = pad, unk, cls, sep, mask */
int clip_tok_special_ids(const clip_tokenizer* t, int32_t* ids);

/* WordPiece ids of n strings, without the [CLS]/[SEP] template and without truncation (what
 * BertWordPieceTokenizer.encode_plain returns): string i is blob[off[i] .. off[i+1]).
 * ids_off[n+1] receives the prefix offsets into ids; ids_cap must be >= off[n] - off[0] (an id
 * consumes at least one byte of text).  fallback[i] = 1: string i was left to the caller (no ids).
 * n_threads <= 0: one thread. */
int clip_tok_encode_plain(clip_tokenizer* t, const char* blob, const int64_t* off, int64_t n,
                          int32_t mode, int32_t n_threads, int32_t* ids, int64_t ids_cap,
                          int64_t* ids_off, uint8_t* fallback);

const char* clip_tok_last_error(void);
/* Unicode version of the built-in tables, e.g. "13.0.0" */
const char* clip_tok_unicode_version(void);

#ifdef __cplusplus
}
#endif
#endif
