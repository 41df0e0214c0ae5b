"""ctypes binding of libclip_tok.so (include/clip_tok.h) and the columnar row builder on top of it.

The reference's front end runs, per sentence and in Python, `eval(row.sentence)`, `' '.join`,
`.lower()` and >= 2(2n+1) HF tokenizer calls (sentclf/neural_baselines.py:193-244,
sentclf/bert.py:70-188; its author's own note on the cost: bert.py:81-82).  One MI355X scores ~50 k
context rows per second, so that loop would starve it by two to three orders of magnitude.  Here:

  * every sentence of the CSV is tokenised ONCE, by the C++ library, on all host cores, straight from
    the CSV cells (Arrow string layout: one blob + offsets, no per-row Python);
  * strings outside the library's scope (non-ASCII bytes, unusual list-literal syntax: flagged, never
    guessed) go through the Python restatement `clip_amd.wordpiece.BertWordPieceTokenizer`;
  * rows (`[CLS] s-n [SEP] .. s0 [SEP] .. s+n [SEP]`, neighbours by the rules of
    neural_baselines.py:219-235) are assembled with array operations; the rare row that reaches 512
    tokens takes the reference's trim procedure verbatim (`clip_amd.collate._fit_context`).

`build_rows_fast` returns exactly what `clip_amd.collate.build_rows` returns
(tests/test_fast_tokenizer.py), which in turn equals the reference collators' rows
(tests/test_frontend.py).  No silent fallback for a missing library: loading raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CLIP_TOK_LIB") or os.path.join(_HERE, "lib", "libclip_tok.so")
MODE_TEXT, MODE_LIST_CELL, MODE_ASCII_ONLY = 0, 1, 256

_p, _i32, _i64 = C.c_void_p, C.c_int32, C.c_int64
SIGNATURES = {
    "clip_tok_create": [C.c_char_p, _i32, C.POINTER(_p)],
    "clip_tok_destroy": [_p],
    "clip_tok_vocab_size": [_p, _p],
    "clip_tok_special_ids": [_p, _p],
    "clip_tok_encode_plain": [_p, _p, _p, _i64, _i32, _i32, _p, _i64, _p, _p],
    "clip_tok_last_error": [],
    "clip_tok_unicode_version": [],
}
_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise ImportError(f"{LIB_PATH} not found: build it with `make -C clip_amd/csrc` "
                          "(or __graft_entry__.build())")
    lib = C.CDLL(LIB_PATH)
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_char_p if name in ("clip_tok_last_error", "clip_tok_unicode_version") else C.c_int
    _lib = lib
    return lib


def _check(rc):
    if rc != 0:
        raise RuntimeError(f"libclip_tok error {rc}: {load().clip_tok_last_error().decode()}")


def _np_ptr(a):
    return C.c_void_p(a.ctypes.data)


def strings_to_blob(strings):
    """list of str -> (uint8 blob, int64 offsets[n+1]) in the Arrow large_string layout."""
    import pyarrow as pa
    arr = pa.array(strings, type=pa.large_string())
    if arr.null_count:
        raise AssertionError("sentence cell is not a string")
    n = len(arr)
    bufs = arr.buffers()
    off = np.frombuffer(bufs[1], dtype=np.int64, count=n + 1, offset=arr.offset * 8)
    blob = np.frombuffer(bufs[2], dtype=np.uint8) if bufs[2] is not None and bufs[2].size else np.zeros(1, np.uint8)
    return blob, np.ascontiguousarray(off), arr        # arr keeps the buffers alive


class FastWordPiece:
    """The C++ tokenizer + the Python one for what it hands back."""

    def __init__(self, vocab_file, do_lower_case=True, n_threads=None):
        from .wordpiece import BertWordPieceTokenizer
        self.py = BertWordPieceTokenizer(vocab_file, do_lower_case=do_lower_case)
        self.n_threads = int(n_threads or min(16, os.cpu_count() or 1))
        lib = load()
        import unicodedata
        # the library's tables and this interpreter's unicodedata must be the same Unicode version for the
        # two to agree on every code point; otherwise non-ASCII strings stay with the Python restatement
        self.mode_flags = 0 if lib.clip_tok_unicode_version().decode() == unicodedata.unidata_version else MODE_ASCII_ONLY
        h = C.c_void_p()
        _check(lib.clip_tok_create(os.fsencode(vocab_file), int(do_lower_case), C.byref(h)))
        self._h = h
        ids = (C.c_int32 * 5)()
        _check(lib.clip_tok_special_ids(h, ids))
        assert list(ids) == [self.py.pad_token_id, self.py.unk_token_id, self.py.cls_token_id,
                             self.py.sep_token_id, self.py.mask_token_id]
        for k in ("pad_token_id", "unk_token_id", "cls_token_id", "sep_token_id", "mask_token_id", "vocab"):
            setattr(self, k, getattr(self.py, k))

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                load().clip_tok_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def encode_plain_many(self, strings, mode=MODE_TEXT, py_text=None):
        """WordPiece ids (no template, no truncation) of every string: (flat int32, offsets int64[n+1],
        fallback bool[n]).  Strings the library flags are tokenised by the Python restatement;
        `py_text(i)` gives the text for that (default: the string itself)."""
        n = len(strings)
        blob, off, keep = strings_to_blob(strings)
        cap = max(int(off[-1] - off[0]), 1)
        ids = np.empty(cap, np.int32)
        ids_off = np.empty(n + 1, np.int64)
        fb = np.zeros(max(n, 1), np.uint8)
        _check(load().clip_tok_encode_plain(self._h, _np_ptr(blob), _np_ptr(off), n, mode | self.mode_flags, self.n_threads,
                                            _np_ptr(ids), cap, _np_ptr(ids_off), _np_ptr(fb)))
        del keep
        fb = fb[:n].astype(bool)
        flat = ids[:ids_off[n]]
        if fb.any():
            lens = np.diff(ids_off)
            extra = {}
            for i in np.flatnonzero(fb):
                extra[i] = np.asarray(self.py.encode_plain(strings[i] if py_text is None else py_text(i)), np.int32)
                lens[i] = len(extra[i])
            new_off = np.concatenate(([0], np.cumsum(lens))).astype(np.int64)
            out = np.empty(int(new_off[-1]), np.int32)
            okr = np.flatnonzero(~fb)
            ln = lens[okr]
            if len(okr):
                src = np.repeat(ids_off[okr] - (np.cumsum(ln) - ln), ln) + np.arange(int(ln.sum()))
                dst = np.repeat(new_off[okr] - (np.cumsum(ln) - ln), ln) + np.arange(int(ln.sum()))
                out[dst] = flat[src]
            for i, a in extra.items():
                out[new_off[i]:new_off[i + 1]] = a
            flat, ids_off = out, new_off
        return np.ascontiguousarray(flat), ids_off, fb


def _gather_segments(flat, poff, src, dst_start, out):
    """out[dst_start[k] : +len_k] = flat[poff[src[k]] : poff[src[k]+1]] for every k, vectorised."""
    ln = (poff[src + 1] - poff[src]).astype(np.int64)
    tot = int(ln.sum())
    if tot == 0:
        return
    base = np.cumsum(ln) - ln
    ar = np.arange(tot, dtype=np.int64)
    out[np.repeat(dst_start - base, ln) + ar] = flat[np.repeat(poff[src] - base, ln) + ar]


def build_rows_fast(dataset, tokenizer: FastWordPiece, n_context_sentences):
    """Same result as clip_amd.collate.build_rows(dataset, tokenizer.py, n): (PackedRows, doc_ids,
    sent_ixs), built from the CSV cells by the C++ tokenizer + array operations."""
    from .collate import _fit_context
    from .datasets import _parse_list
    from .scoring import PackedRows
    tok, N, n = tokenizer, len(dataset), int(n_context_sentences)
    cells = dataset._sentence
    for c in cells:
        if not isinstance(c, str):
            raise AssertionError("sentence cell is not a string")      # the reference asserts the same
    text = lambda i: " ".join(str(x) for x in _parse_list(cells[i], lower=False)).lower()
    flat, poff, _ = tok.encode_plain_many(cells, MODE_LIST_CELL, py_text=text)
    lens = np.diff(poff)
    cls_id, sep_id = tok.cls_token_id, tok.sep_token_id
    doc_ids = dataset.doc_ids.astype(np.int64)

    if n == 0:
        keep = np.minimum(lens, 510)                                   # truncation=True, max_length=512
        row_len = keep + 2
        off = np.concatenate(([0], np.cumsum(row_len))).astype(np.int64)
        out = np.empty(int(off[-1]), np.int32)
        out[off[:-1]] = cls_id
        out[off[1:] - 1] = sep_id
        trunc_off = np.stack((poff[:-1], poff[:-1] + keep), 1).reshape(-1)          # [start, start+keep) per row
        ln = keep
        base = np.cumsum(ln) - ln
        ar = np.arange(int(ln.sum()), dtype=np.int64)
        out[np.repeat(off[:-1] + 1 - base, ln) + ar] = flat[np.repeat(trunc_off[0::2] - base, ln) + ar]
        return PackedRows(out, off), doc_ids, np.asarray(dataset.sent_ixs, np.int64)

    ix = np.asarray(dataset.sent_ixs)
    # two virtual sentences: N = <DOC_START>, N + 1 = <DOC_END>  (neural_baselines.py:222,226)
    start_p = np.asarray(tok.py.encode_plain("<DOC_START>".lower()), np.int32)
    end_p = np.asarray(tok.py.encode_plain("<DOC_END>".lower()), np.int32)
    flat2 = np.concatenate((flat, start_p, end_p))
    poff2 = np.concatenate((poff, [poff[-1] + len(start_p), poff[-1] + len(start_p) + len(end_p)])).astype(np.int64)
    offs = np.arange(-n, n + 1)
    rows_i = np.arange(N)
    want = ix[:, None] + offs[None, :]                                 # expected sentence index
    j = rows_i[:, None] + offs[None, :]
    jw = np.where(j < 0, j + N, j)                                     # a negative position wraps, as .iloc does
    in_range = j < N
    same = np.zeros_like(in_range)
    same[in_range] = ix[np.clip(jw, 0, N - 1)][in_range] == want[in_range]
    src = np.where(want < 0, N, np.where(in_range & same, np.clip(jw, 0, N - 1), N + 1))
    src[:, n] = rows_i                                                 # offset 0: the focus sentence
    seg_len = (poff2[src + 1] - poff2[src]).astype(np.int64)
    total = 1 + (seg_len + 1).sum(1)
    long_rows = np.flatnonzero(total >= 512)
    special = {}
    for i in long_rows:                                                # rare: the reference's trim loop, verbatim
        toks_, sent_ix, annot, label, doc_id, context = dataset[int(i)]
        s, rep = _fit_context(" ".join(toks_).lower(), tuple(c.lower() for c in context), tok.py)
        special[int(i)] = np.asarray(rep if rep is not None else tok.py.encode(s), np.int32)
        total[i] = len(special[int(i)])
    off = np.concatenate(([0], np.cumsum(total))).astype(np.int64)
    out = np.empty(int(off[-1]), np.int32)
    normal = np.ones(N, bool)
    normal[long_rows] = False
    r = np.flatnonzero(normal)
    out[off[r]] = cls_id
    seg_start = off[r, None] + 1 + np.cumsum(seg_len[r] + 1, 1) - (seg_len[r] + 1)   # first token of each segment
    out[(seg_start + seg_len[r]).reshape(-1)] = sep_id
    _gather_segments(flat2, poff2, src[r].reshape(-1), seg_start.reshape(-1), out)
    for i, a in special.items():
        out[off[i]:off[i + 1]] = a
    return PackedRows(out, off), doc_ids, np.asarray(ix, np.int64)


def build_rows_chunks(dataset, tokenizer: FastWordPiece, n_context_sentences, bounds):
    """build_rows_fast in pieces: yields (PackedRows, doc_ids, sent_ixs) for each row range (a, b) of
    `bounds`, identical to slicing the whole-file result — so a consumer (the GPU) can start on the first
    chunk while the next is being tokenised.  A row's context is looked up by row OFFSET
    (neural_baselines.py:219-235), so each piece is built over its rows plus a halo of n rows on either
    side; a negative position wraps to the END of the file (pandas .iloc), so the first pieces also carry
    the file's last n rows at their end."""
    from .scoring import PackedRows
    n, N = int(n_context_sentences), len(dataset)
    for a, b in bounds:
        lo, hi = max(0, a - n), min(N, b + n)
        idx = np.arange(lo, hi)
        if n and a - n < 0:
            idx = np.concatenate((idx, np.arange(max(hi, N - n), N)))
        rows, d, s = build_rows_fast(dataset.take(idx), tokenizer, n)
        i0, i1 = a - lo, a - lo + (b - a)
        o = rows.offsets[i0:i1 + 1]
        yield PackedRows(rows.flat[o[0]:o[-1]], o - o[0]), d[i0:i1], s[i0:i1]
