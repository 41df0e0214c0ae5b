"""libclip_tok.so (C++ WordPiece front end, include/clip_tok.h) against the Python restatement
clip_amd/wordpiece.py — itself pinned by goldens from HF's BertTokenizer under the reference's
collators (tests/test_frontend.py).  Integer work: bit-exact."""
import os
import re

import numpy as np
import pandas as pd
import pytest

from clip_amd import collate, datasets, fast_tokenizer as FT
from clip_amd.wordpiece import BertWordPieceTokenizer

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VOCAB = os.path.join(G, "vocab.txt")


@pytest.fixture(scope="module")
def fast():
    return FT.FastWordPiece(VOCAB, n_threads=4)


def test_library_exports_every_declared_symbol():
    src = open(os.path.join(ROOT, "include", "clip_tok.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = sorted(set(re.findall(r"\b(clip_tok_[a-z0-9_]+)\s*\(", src)))
    lib = FT.load()
    assert names == sorted(FT.SIGNATURES) and len(names) == 7
    for n in names:
        assert hasattr(lib, n), n
    assert lib.clip_tok_create(None, 1, None) == -1 and b"null" in lib.clip_tok_last_error()
    assert lib.clip_tok_create(b"/nonexistent/vocab.txt", 1, FT.C.byref(FT.C.c_void_p())) == -2


def _rows_of(flat, off):
    return [flat[off[i]:off[i + 1]].tolist() for i in range(len(off) - 1)]


def test_text_mode_matches_python_tokenizer(fast):
    vocab_words = [w for w in fast.vocab if w.isascii() and not w.startswith("[")]
    rng = np.random.default_rng(0)
    texts = ["", " ", "[SEP] [SEP]", "x" * 101, "x" * 100, "Pt should F/U w/ Dr. Smith in 2 wks.",
             "tab\tnew\nline\rcr", "ctl\x00a\x01b\x7fc\x1fd\x0be\x0cf", "[CLS]foo[MASK]bar[PAD][UNK] [sep] [SEP]x",
             "don't  stop -- (ever)!?", "[", "[SE", "a[SEP", "MiXeD CaSe TEXT", "naïve café", "日本語 text", "Σίσυφος",
             "� replacement", "ab" * 300]
    for _ in range(300):      # sentences from vocabulary words, glued and spaced at random
        k = int(rng.integers(1, 30))
        ws = rng.choice(vocab_words, k)
        texts.append("".join(w.replace("##", "") + (" " if rng.random() < 0.8 else "") for w in ws))
    for _ in range(300):      # every ASCII byte, uniformly
        texts.append(bytes(rng.integers(0, 128, int(rng.integers(0, 80))).astype(np.uint8)).decode("ascii"))
    flat, off, fb = fast.encode_plain_many(texts)
    ref = [fast.py.encode_plain(t) for t in texts]
    assert _rows_of(flat, off) == ref
    assert [t for t, f in zip(texts, fb) if f] == ["Σίσυφος"]       # capital sigma: context-dependent lower case
    # thread count changes nothing
    one = FT.FastWordPiece(VOCAB, n_threads=1)
    big = texts * 20
    a = fast.encode_plain_many(big)
    b = one.encode_plain_many(big)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_cased_tokenizer(tmp_path):
    f = FT.FastWordPiece(VOCAB, do_lower_case=False, n_threads=2)
    texts = ["Hello World", "the Patient", "[SEP] The end."]
    flat, off, _ = f.encode_plain_many(texts)
    assert _rows_of(flat, off) == [f.py.encode_plain(t) for t in texts]


def test_list_cell_mode_matches_python_path(fast):
    cells = pd.read_csv(os.path.join(G, "notes_synth.csv"))["sentence"].tolist()
    n_csv = len(cells)
    cells += ["[]", "['a']", "['a',]", "[ 'a' , \"b\" ]", "['it\\'s', \"say \\\"hi\\\"\", 'back\\\\slash']",
              "['tab\\there', 'nl\\nthere']", "['Caps', 'AND', 'MORE']", "['[SEP]', 'x']",
              # left to Python: not plain quoted strings / not a list literal / non-ASCII
              "[1, 2]", "['a', 2.50]", "['a' 'b']", "[['a']]", "[u'a']", "['''a''']", "['\\x41']", "['\\u00e9']",
              "plain words here", "('a', 'b')", " ['a']", "['a'", "['café']", "['a', None]", "['a'] + ['b']"]
    from clip_amd.datasets import _parse_list
    text = lambda i: " ".join(str(x) for x in _parse_list(cells[i], lower=False)).lower()
    flat, off, fb = fast.encode_plain_many(cells, FT.MODE_LIST_CELL, py_text=text)
    assert _rows_of(flat, off) == [fast.py.encode_plain(text(i)) for i in range(len(cells))]
    handled = dict(zip(cells, ~fb))
    assert handled["['it\\'s', \"say \\\"hi\\\"\", 'back\\\\slash']"] and handled["['a',]"] and handled["[]"]
    assert not handled["[1, 2]"] and not handled["['\\x41']"] and not handled["plain words here"]
    assert fb[n_csv:].sum() == 14                                          # "['café']" is the library's now
    assert not fb[:n_csv].any()                                            # every cell of the CSV, non-ASCII ones included


def _synthetic_csv(path, n_docs=60, seed=3):
    """Documents of 1..80 sentences of vocabulary words; some duplicated / skipped sentence indices
    (the <DOC_END> mismatch branch), a few very long sentences (rows reaching 512 tokens)."""
    rng = np.random.default_rng(seed)
    tok = BertWordPieceTokenizer(VOCAB)
    words = [w for w in tok.vocab if w.isascii() and w.isalpha() and len(w) > 1]
    rows = []
    for d in range(n_docs):
        k = int(rng.integers(1, 80))
        ixs = list(range(k))
        if k > 5 and rng.random() < 0.3:
            ixs[int(rng.integers(1, k))] = int(rng.integers(0, k))       # duplicate / out-of-order index
        for s in ixs:
            L = int(rng.integers(1, 40)) if rng.random() > 0.02 else int(rng.integers(150, 700))
            sent = [str(w).capitalize() if rng.random() < 0.2 else str(w) for w in rng.choice(words, L)]
            if rng.random() < 0.1:
                sent.append("it's")
            if rng.random() < 0.01:
                sent.append("café")
            rows.append({"doc_id": 1000 + d, "sent_index": s, "sentence": str(sent), "labels": "[]"})
    pd.DataFrame(rows).to_csv(path, index=False)
    return len(rows)


@pytest.mark.parametrize("n", [2, 1, 0])
def test_build_rows_fast_equals_build_rows(fast, tmp_path, n):
    paths = [os.path.join(G, "notes_synth.csv"), os.path.join(G, "dummy_test_data.csv"), str(tmp_path / "big.csv")]
    _synthetic_csv(paths[2])
    for path in paths:
        ds = (datasets.SentEvalDataset(path, "multilabel", n_context_sentences=n) if n
              else datasets.SentDataset(path, "multilabel"))
        ref, d0, s0 = collate.build_rows(ds, fast.py, n)
        got, d1, s1 = FT.build_rows_fast(ds, fast, n)
        assert np.array_equal(got.offsets, ref.offsets), path
        assert np.array_equal(got.flat, ref.flat), path
        assert np.array_equal(d0, d1) and np.array_equal(s0, s1)
        if n and path == paths[2]:
            assert (ref.lens >= 500).any()          # the trim branch was exercised
            seps = np.add.reduceat((ref.flat == fast.sep_token_id).astype(np.int64), ref.offsets[:-1])
            assert (seps == 2 * n + 1).all()        # the invariant bert_model.py:52 needs


def _fuzz_strings(n, seed):
    rng = np.random.default_rng(seed)
    special = [0x130, 0x3C2, 0x3C3, 0x2028, 0x2029, 0x85, 0xA0, 0xAD, 0x200B, 0xFFFD, 0x0, 0xE000, 0x378, 0x1F600,
               0x10FFFF, 0xFB01, 0x1E9E, 0x1C5, 0x1F88, 0x0344, 0x0F73, 0x1D15E, 0x3000, 0x1680, 0xFF01, 0x2014]
    ascii_pool = [ord(c) for c in "abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789      .,;:!?()[]-'\"/\t\n"]
    out = []
    for _ in range(n):
        cps = []
        for _ in range(int(rng.integers(0, 40))):
            r = rng.random()
            if r < 0.45:
                cp = int(rng.choice(ascii_pool))
            elif r < 0.60:
                cp = int(rng.integers(0xA0, 0x250))            # Latin-1, Latin Extended
            elif r < 0.68:
                cp = int(rng.integers(0x300, 0x370))            # combining marks (reordering, stripping)
            elif r < 0.78:
                cp = int(rng.integers(0x370, 0x530))            # Greek, Cyrillic, Armenian (capital sigma among them)
            elif r < 0.81:
                cp = int(rng.integers(0xAC00, 0xD7A4))          # Hangul syllables
            elif r < 0.84:
                cp = int(rng.integers(0x4E00, 0x4F00))          # CJK
            elif r < 0.90:
                cp = int(rng.choice(special))
            else:
                cp = int(rng.integers(0x80, 0x30000))
                if 0xD800 <= cp <= 0xDFFF:
                    cp = 0x263A
            cps.append(cp)
        out.append("".join(map(chr, cps)))
    return out


@pytest.mark.parametrize("lower", [True, False])
def test_unicode_path_matches_python_tokenizer(tmp_path, lower):
    """Random strings over many scripts, marks, separators and unassigned / private-use code points: the
    code-point path of the library equals the Python restatement; only capital sigma is handed back."""
    vocab = open(VOCAB, encoding="utf-8").read().split("\n")
    vocab = [v for v in vocab if v] + ["é", "##é", "e", "##e", "caf", "ß", "##ß", "α", "##α", "σ", "ς", "##ς", "日", "本", "i", "##̇",
                                        "ᄀ", "##ᅡ", "##ᆨ", "¿", "ø", "##ø", "д", "##д", "ﬁ", "\u2028x"]
    vf = tmp_path / "vocab.txt"
    vf.write_text("\n".join(dict.fromkeys(vocab)) + "\n", encoding="utf-8")
    f = FT.FastWordPiece(str(vf), do_lower_case=lower, n_threads=4)
    assert f.mode_flags == 0
    texts = _fuzz_strings(4000, 7 + lower) + ["café ß Straße", "İstanbul", "ǅ ǆ", "각 한국어", "日本語", "a\u0301\u0323b", "¿qué?",
                                               "ςσ ΣΑΣ", "x\u2028y\u2029z", "[SEP]é[CLS]", "ﬁn"]
    flat, off, fb = f.encode_plain_many(texts)
    # the invariant the library's output capacity rests on: no sentence yields more tokens than it has bytes
    nbytes = np.array([len(t.encode("utf-8")) for t in texts])
    assert (np.diff(off)[~fb] <= nbytes[~fb]).all() and off[-1] <= nbytes.sum()
    ref = [f.py.encode_plain(t) for t in texts]
    got = _rows_of(flat, off)
    bad = [i for i in range(len(texts)) if got[i] != ref[i]]
    assert not bad, (texts[bad[0]].encode("unicode_escape"), got[bad[0]], ref[bad[0]])
    import unicodedata
    for t, flagged in zip(texts, fb):      # handed back exactly when a capital sigma reaches the lower-casing step
        if lower:
            assert flagged == ("\u03a3" in unicodedata.normalize("NFD", t)), t.encode("unicode_escape")
        else:
            assert not flagged
    # the same strings as list cells (repr() escapes some code points as \x.. / \u....: those are Python's)
    cells = [str([t[:len(t) // 2], t[len(t) // 2:]]) for t in texts]
    from clip_amd.datasets import _parse_list
    text = lambda i: " ".join(str(x) for x in _parse_list(cells[i], lower=False)).lower()
    flat, off, fb = f.encode_plain_many(cells, FT.MODE_LIST_CELL, py_text=text)
    assert _rows_of(flat, off) == [f.py.encode_plain(text(i)) for i in range(len(cells))]
    assert (~fb).sum() > len(cells) // 4


def test_ascii_only_flag_hands_back_every_non_ascii_string(fast):
    saved = fast.mode_flags
    try:
        fast.mode_flags = FT.MODE_ASCII_ONLY
        texts = ["plain", "café", "x" * 10, "日本"]
        flat, off, fb = fast.encode_plain_many(texts)
        assert list(fb) == [False, True, False, True]
        assert _rows_of(flat, off) == [fast.py.encode_plain(t) for t in texts]
    finally:
        fast.mode_flags = saved


@pytest.mark.parametrize("n,chunk", [(2, 7), (2, 1), (1, 13), (0, 5), (2, 10**6)])
def test_chunked_build_equals_whole_file_build(n, chunk):
    """build_rows_chunks (the pipelined form the CLI feeds the GPU from) == slices of build_rows_fast, incl.
    the wrap of negative row positions to the end of the file and the duplicated / gapped sentence indices of
    the fixtures."""
    import os
    from clip_amd import columnar, datasets
    from clip_amd.fast_tokenizer import FastWordPiece, build_rows_chunks, build_rows_fast
    G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    tok = FastWordPiece(os.path.join(G, "vocab.txt"))
    for fname in ("notes_synth.csv", "dummy_test_data.csv"):
        path = os.path.join(G, fname)
        ds = datasets.SentEvalDataset(path, "multilabel", n_context_sentences=n) if n else datasets.SentDataset(path, "multilabel")
        whole, d, s = build_rows_fast(ds, tok, n)
        bounds = columnar.chunk_bounds(ds.doc_ids, chunk)
        assert bounds[0][0] == 0 and bounds[-1][1] == len(ds)
        got = list(build_rows_chunks(ds, tok, n, bounds))
        flat = np.concatenate([g[0].flat for g in got])
        lens = np.concatenate([g[0].lens for g in got])
        assert np.array_equal(flat, whole.flat) and np.array_equal(lens, whole.lens)
        assert np.array_equal(np.concatenate([g[1] for g in got]), d) and np.array_equal(np.concatenate([g[2] for g in got]), s)
