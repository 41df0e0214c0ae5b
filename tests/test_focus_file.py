"""The top-k file round trip: written by clip_amd.job.write_topk_csv, read the way the reference's
pre-training dataset reads it (sentclf/pretrain_tasks.py:37,57-64: pd.read_csv + first row matching
(doc_id, sent_ix))."""
import os

import numpy as np
import pandas as pd
import pytest

from clip_amd import focus_file, job, topk

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _reference_lookup(sents, doc_id, sent_ix):
    """pretrain_tasks.py:58-63 as written: boolean mask over the table, first hit's index, None if empty."""
    row = sents[(sents.doc_id == doc_id) & (sents.sent_ix == sent_ix)]
    try:
        return row.index[0]
    except IndexError:
        return None


def test_written_file_reads_back_like_the_reference_reads_it(tmp_path):
    rng = np.random.default_rng(3)
    n = 4000
    doc = np.sort(rng.integers(0, 300, n)).astype(np.int64)
    sent = np.concatenate([np.arange(c) for c in np.bincount(doc) if c]).astype(np.int64)
    score = (rng.integers(0, 60, n) / 60).astype(np.float32)           # many ties
    rec = topk.local_topk(doc, sent, score, 500)
    path = tmp_path / "ttp_top500.csv"
    job.write_topk_csv(str(path), rec)
    df = pd.read_csv(path)                                              # pretrain_tasks.py:37
    assert list(df.columns[:2]) == ["doc_id", "sent_ix"] and len(df) == 500
    d, s, c = focus_file.read_topk_csv(str(path))
    assert np.array_equal(d, df.doc_id.to_numpy()) and np.array_equal(s, df.sent_ix.to_numpy())
    assert np.array_equal(d, rec["doc_id"]) and np.array_equal(s, rec["sent_ix"])
    assert np.array_equal(c, rec["score"])                              # str(np.float32) round-trips exactly
    # every focus record finds its sentence row, by the reference's lookup and by the vectorised one
    sents = pd.DataFrame({"doc_id": doc, "sent_ix": sent})
    idx = focus_file.focus_row_indices(doc, sent, d, s)
    for i in range(0, 500, 7):
        assert idx[i] == _reference_lookup(sents, d[i], s[i])
    assert (doc[idx] == d).all() and (sent[idx] == s).all()


def test_duplicate_keys_resolve_to_the_first_row_and_missing_records_are_flagged():
    # the reference's own fixture: doc 1339 holds sent_index 1 twice (dummy_test_data.csv:7-8)
    df = pd.read_csv(os.path.join(GOLDEN, "dummy_test_data.csv")).rename(columns={"sent_index": "sent_ix"})
    keys = list(zip(df.doc_id, df.sent_ix))
    assert len(set(keys)) < len(keys)
    fd = np.array([k[0] for k in keys] + [df.doc_id.iloc[0], 424242], np.int64)
    fs = np.array([k[1] for k in keys] + [99, 0], np.int64)
    idx = focus_file.focus_row_indices(df.doc_id, df.sent_ix, fd, fs)
    for i in range(len(fd)):
        ref = _reference_lookup(df, fd[i], fs[i])
        assert idx[i] == (-1 if ref is None else ref)
    assert idx[-1] == -1 and idx[-2] == -1


def test_headerless_scores_file_is_not_a_focus_file(tmp_path):
    p = tmp_path / "ttp_doc_sents.csv"
    p.write_text("12,0,0.5\n12,1,0.25\n")
    with pytest.raises(ValueError):
        focus_file.read_topk_csv(str(p))
