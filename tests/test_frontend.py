"""Host front end (datasets, WordPiece tokenizer, collators, columnar row builder) against golden
outputs of the reference's own SentDataset / SentEvalDataset / collator / ctx_collator over HF's
BertTokenizer (tests/golden/collate_golden.json.gz, oracle/gen_golden_collate.py).  Integer work:
bit-exact."""
import gzip
import json
import os

import numpy as np
import pytest
import torch

from clip_amd import collate, datasets
from clip_amd.wordpiece import BertWordPieceTokenizer

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def gold():
    with gzip.open(os.path.join(G, "collate_golden.json.gz"), "rt") as f:
        return json.load(f)


@pytest.fixture(scope="module")
def tok():
    return BertWordPieceTokenizer(os.path.join(G, "vocab.txt"))


def _batches(n, sizes=(1, 3, 4, 2)):
    out, i, k = [], 0, 0
    while i < n:
        out.append(list(range(i, min(i + sizes[k % len(sizes)], n))))
        i += sizes[k % len(sizes)]
        k += 1
    return out


def _norm(item):
    return json.loads(json.dumps([x.item() if isinstance(x, np.generic) else x for x in item], default=list))


@pytest.mark.parametrize("csv_name", ["dummy_test_data.csv", "notes_synth.csv"])
@pytest.mark.parametrize("n", [2, 1])
def test_eval_dataset_and_ctx_collator(gold, tok, csv_name, n):
    g = gold[f"{csv_name}|ctx{n}"]
    ds = datasets.SentEvalDataset(os.path.join(G, csv_name), "multilabel", n_context_sentences=n)
    assert len(ds) == len(g["dataset"])
    items = [ds[i] for i in range(len(ds))]
    for it, ref in zip(items, g["dataset"]):
        assert _norm(it) == ref                       # tokens, sent_ix, annotator, labels, doc_id, contexts
    for b, ref in zip(_batches(len(ds)), g["batches"]):
        out = collate.ctx_collator([items[i] for i in b], "multilabel", tok, eval=True)
        for k in ("input_ids", "attention_mask", "token_type_ids", "labels"):
            assert out[k].tolist() == ref[k], k
        assert out["sentences"] == ref["sentences"]
        assert [int(x) for x in out["doc_ids"]] == ref["doc_ids"]
        assert [int(x) for x in out["sent_ixs"]] == ref["sent_ixs"]
        # the invariant bert_model.py:52 needs
        assert ((out["input_ids"] == tok.sep_token_id).sum(1) == 2 * n + 1).all()


@pytest.mark.parametrize("csv_name", ["dummy_test_data.csv", "notes_synth.csv"])
def test_sent_dataset_and_collator(gold, tok, csv_name):
    g = gold[f"{csv_name}|sent"]
    ds = datasets.SentDataset(os.path.join(G, csv_name), "multilabel")
    items = [ds[i] for i in range(len(ds))]
    for it, ref in zip(items, g["dataset"]):
        assert _norm(it) == ref
    for b, ref in zip(_batches(len(ds)), g["batches"]):
        out = collate.collator([items[i] for i in b], "multilabel", tok, eval=True)
        for k in ("input_ids", "attention_mask", "labels"):
            assert out[k].tolist() == ref[k], k
        assert "token_type_ids" not in out            # bert.py:65-68 returns none


@pytest.mark.parametrize("n", [2, 1, 0])
def test_build_rows_equals_collator_rows(tok, n):
    """The columnar fast path (each sentence tokenised once) yields exactly the collator's rows."""
    path = os.path.join(G, "notes_synth.csv")
    if n:
        ds = datasets.SentEvalDataset(path, "multilabel", n_context_sentences=n)
        full = collate.ctx_collator([ds[i] for i in range(len(ds))], "multilabel", tok)
    else:
        ds = datasets.SentDataset(path, "multilabel")
        full = collate.collator([ds[i] for i in range(len(ds))], "multilabel", tok)
    pr, doc_ids, sent_ixs = collate.build_rows(ds, tok, n)
    ids, mask = full["input_ids"].numpy(), full["attention_mask"].numpy()
    for i in range(len(ds)):
        row = pr.flat[pr.offsets[i]:pr.offsets[i + 1]]
        if n and len(row) != mask[i].sum():
            # back-filled row (focus sentence alone >= 512 tokens): the reference overwrites ids but
            # not the mask (bert.py:156-165 no-op); the row builder keeps every id of the row
            assert np.array_equal(row, ids[i][:len(row)])
        else:
            assert np.array_equal(row, ids[i][mask[i] == 1])
    assert doc_ids.tolist() == [int(x) for x in full["doc_ids"]]


def test_select_inputs():
    x = {"input_ids": 1, "attention_mask": 2, "labels": 3, "token_type_ids": 4, "doc_ids": 5}
    assert collate.select_inputs(x) == {"input_ids": 1, "attention_mask": 2, "labels": 3, "token_type_ids": 4}
    del x["token_type_ids"]
    assert "token_type_ids" not in collate.select_inputs(x)


def test_tokenizer_edge_cases(tok):
    assert tok("")["input_ids"] == [101, 102]
    assert tok("[SEP] [SEP]")["input_ids"] == [101, 102, 102, 102]
    ids = tok("x" * 101)["input_ids"]
    assert ids == [101, tok.unk_token_id, 102]
    long = tok(" ".join(["test"] * 1000))["input_ids"]
    assert len(long) == 512 and long[-1] == 102
    out = tok(["test", "test sentence"])
    assert out["attention_mask"] == [[1, 1, 1, 0], [1, 1, 1, 1]]
