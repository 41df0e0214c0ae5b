"""Synthetic pre-tokenised note sentences (MIMIC is access-controlled).

Rows are lists of token ids with the layout the reference's collators produce:
  sentence-only (ref: sentclf/bert.py:44-68):   [CLS] w.. [SEP]
  context n     (ref: sentclf/bert.py:104,152): [CLS] s-n [SEP] .. s0 [SEP] .. s+n [SEP]
with exactly 2n+1 [SEP] per row (the invariant bert_model.py:52 relies on).
Word ids are uniform in [1000, vocab); segment lengths are log-normal (median 18 word pieces).
"""
import numpy as np

from .constants import CLS_ID, SEP_ID


def _seg_lengths(rng, n, sigma=0.6, lo=1, hi=62):
    return np.clip(np.rint(rng.lognormal(np.log(18.0), sigma, n)), lo, hi).astype(np.int64)


def sentence_rows(n_rows, vocab_size=30522, max_len=64, seed=1001):
    """Config C2: L = 2 + clip(round(LogNormal(ln 18, 0.6)), 1, max_len-2)."""
    rng = np.random.default_rng(seed)
    lens = _seg_lengths(rng, n_rows, hi=max_len - 2)
    rows = []
    for L in lens:
        w = rng.integers(1000, vocab_size, L)
        rows.append(np.concatenate(([CLS_ID], w, [SEP_ID])).astype(np.int32))
    return rows


def context_rows(n_rows, n_ctx=2, vocab_size=30522, max_len=256, seed=1002, sigma=0.6,
                 fixed_len=None, seg_hi=62):
    """Config C3/C4/C5: 2n+1 segments; total clipped to max_len by dropping OUTER context tokens
    (never a [SEP]); fixed_len pads the focus segment so every row has exactly that many tokens
    (the headline roofline variant: no padding ambiguity)."""
    rng = np.random.default_rng(seed)
    nseg = 2 * n_ctx + 1
    rows = []
    for _ in range(n_rows):
        seg = _seg_lengths(rng, nseg, sigma=sigma, hi=seg_hi)
        overhead = 1 + nseg
        if fixed_len is not None:
            budget = fixed_len - overhead
            # scale segments to the exact budget, remainder into the focus
            seg = np.maximum(1, (seg * budget // max(seg.sum(), 1))).astype(np.int64)
            seg[n_ctx] += budget - seg.sum()
            if seg[n_ctx] < 1:
                raise ValueError("fixed_len too small")
        else:
            # drop outer context tokens alternately until the row fits
            order = []
            for d in range(n_ctx, 0, -1):
                order += [n_ctx - d, n_ctx + d]
            j = 0
            while overhead + seg.sum() > max_len:
                k = order[j % len(order)]
                if seg[k] > 0:
                    seg[k] -= 1
                elif all(seg[o] == 0 for o in order):
                    seg[n_ctx] -= 1
                j += 1
        toks = [CLS_ID]
        for s in seg:
            toks.extend(rng.integers(1000, vocab_size, s).tolist())
            toks.append(SEP_ID)
        rows.append(np.asarray(toks, dtype=np.int32))
    return rows


def documents(n_rows, seed=1004, lo=40, hi=400):
    """doc_id (running int64) and sent_ix for n_rows sentences grouped in docs of lo..hi sentences."""
    rng = np.random.default_rng(seed)
    doc_ids = np.empty(n_rows, np.int64)
    sent_ixs = np.empty(n_rows, np.int32)
    i, d = 0, 0
    while i < n_rows:
        n = min(int(rng.integers(lo, hi + 1)), n_rows - i)
        doc_ids[i:i + n] = d
        sent_ixs[i:i + n] = np.arange(n)
        i += n
        d += 1
    return doc_ids, sent_ixs


def packed_context_rows(n_rows, n_ctx=2, vocab_size=30522, max_len=256, seed=1004, sigma=0.6, seg_hi=62,
                        word_lo=1000, cls_id=CLS_ID, sep_id=SEP_ID):
    """Bulk form of context_rows for the full-size configs (C4: 10 M rows, C5: 50 M): the same row
    layout and segment-length law, built with array operations only, returned columnar as
    (flat int32 ids, int64 offsets) — the input of clip_amd.scoring.PackedRows.  Rows longer than
    max_len lose OUTER context tokens first (outermost pair, then the next pair, then the focus);
    a [SEP] is never dropped.  n_ctx = 0 gives sentence-only rows ([CLS] w.. [SEP]).  Word ids are uniform in
    [word_lo, vocab_size) and never equal cls_id / sep_id (small vocabularies: clip_amd.precision's sample)."""
    rng = np.random.default_rng(seed)
    nseg = 2 * n_ctx + 1
    seg = _seg_lengths(rng, n_rows * nseg, sigma=sigma, hi=seg_hi).reshape(n_rows, nseg)
    overhead = 1 + nseg
    excess = np.maximum(seg.sum(1) + overhead - max_len, 0)
    for d in range(n_ctx, 0, -1):               # outermost pair first
        lo, hi = n_ctx - d, n_ctx + d
        take_lo = np.minimum(seg[:, lo], (excess + 1) // 2)
        take_hi = np.minimum(seg[:, hi], excess - take_lo)
        take_lo = np.minimum(seg[:, lo], excess - take_hi)   # what the other side could not give
        seg[:, lo] -= take_lo
        seg[:, hi] -= take_hi
        excess -= take_lo + take_hi
    seg[:, n_ctx] -= excess
    if int(seg[:, n_ctx].min()) < 1:
        raise ValueError("max_len too small for the focus sentence")
    lens = seg.sum(1) + overhead
    offsets = np.concatenate(([0], np.cumsum(lens))).astype(np.int64)
    flat = rng.integers(word_lo, vocab_size, int(offsets[-1]), dtype=np.int32)
    if word_lo <= max(cls_id, sep_id):
        alt = next(w for w in range(word_lo, vocab_size) if w not in (cls_id, sep_id))
        flat[(flat == cls_id) | (flat == sep_id)] = alt
    flat[offsets[:-1]] = cls_id
    # the [SEP] closing segment s of row r sits at offsets[r] + 1 + sum(seg[r, :s+1]) + s
    sep_pos = offsets[:-1, None] + 1 + np.cumsum(seg, 1) + np.arange(nseg)[None, :]
    flat[sep_pos.ravel()] = sep_id
    return flat, offsets
