"""Host logic of precision='auto' (clip_amd/precision.py) that needs no GPU: the calibration samples keep the
row invariant the model relies on (exactly 2n+1 separators, sentclf/bert_model.py:52), the spread sample is a
function of the input alone (every rank picks the same rows), decisions follow test_load.py:37-38."""
import numpy as np
import pytest

from clip_amd import precision as P
from clip_amd.bert_model import BertForSequenceMultilabelClassification, BertSequenceMultilabelClassificationContext
from clip_amd.config import BertConfig
from clip_amd.scoring import PackedRows


@pytest.mark.parametrize("vocab,sep,n_ctx", [(30522, 102, 2), (2000, 102, 2), (150, 102, 1), (64, 3, 2), (300, 102, 0)])
def test_synthetic_sample_keeps_the_separator_invariant(vocab, sep, n_ctx):
    cfg = BertConfig(vocab_size=vocab, hidden_size=128, num_hidden_layers=1, num_attention_heads=2, intermediate_size=256)
    if n_ctx:
        m = BertSequenceMultilabelClassificationContext(cfg)
        m.set_sep_token_id(sep); m.set_n_context_sentences(n_ctx)
    else:
        m = BertForSequenceMultilabelClassification(cfg)
    s = P.synthetic_sample(m, n_rows=64)
    assert len(s) == 64 and s.flat.min() >= 0 and s.flat.max() < vocab
    assert int(s.lens.max()) <= cfg.max_position_embeddings
    for i in range(len(s)):
        row = s.flat[s.offsets[i]:s.offsets[i + 1]]
        assert (row == sep).sum() == 2 * n_ctx + 1 and row[-1] == sep and row[0] != sep


def test_spread_sample_is_deterministic_and_covers_the_input():
    lens = np.arange(1, 5001) % 7 + 3
    off = np.concatenate(([0], np.cumsum(lens)))
    rows = PackedRows(np.arange(off[-1], dtype=np.int32), off)
    a, b = P.spread_sample(rows), P.spread_sample(rows)
    assert len(a) == P.SAMPLE_ROWS == 2048 and np.array_equal(a.flat, b.flat) and np.array_equal(a.offsets, b.offsets)
    assert a.flat[0] == 0 and a.flat[-1] == rows.flat[-1]          # first and last rows are in it
    small = PackedRows(rows.flat[:off[100]], off[:101])
    assert P.spread_sample(small) is small


def test_label_decisions_compare_float32_scores_with_float64_thresholds():
    p = np.array([[np.float32(0.35)] * 7], np.float32)
    thr = np.array(P.DNOTE_LABEL_THRESHS)
    d = P.label_decisions(p, thr)
    assert d.dtype == bool and d.shape == (1, 7)
    # float32(0.35) = 0.3499999940.. is NOT > 0.35000000000000003 (test_load.py:38 promotes the score to fp64)
    assert not d[0, 5] and d[0, 1] and not d[0, 0]


def test_reference_threshold_files_are_restated_exactly():
    import os
    for name, vals in (("clip_dnote_label_threshs.npy", P.DNOTE_LABEL_THRESHS), ("clip_ttp_label_threshs.npy", P.TTP_LABEL_THRESHS)):
        path = os.path.join("/root/reference", name)
        if os.path.isfile(path):                 # the reference is not on the GPU box
            assert np.array_equal(np.load(path, allow_pickle=False), np.asarray(vals, np.float64))
