"""clip_amd/pretrain_data.py against goldens from the imported reference (oracle/gen_golden_pretrain.py:
sentclf/pretrain_tasks.py SentContextDataset + switch_mlm_collate over HF's BertTokenizer): same items under the
same `random` seed, same batches — token ids after masking, MLM labels, token types, switch labels, sentences —
under the same torch seed, bit for bit."""
import gzip
import json
import os
import random

import numpy as np
import pytest
import torch

from clip_amd import pretrain_data as PD
from clip_amd.wordpiece import BertWordPieceTokenizer

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def gold():
    with gzip.open(os.path.join(G, "pretrain_golden.json.gz"), "rt") as f:
        return json.load(f)


def _norm(it):
    if it is None:
        return None
    sent, label, sent_ix, doc_id, ctx = it
    return [list(sent), int(label), int(sent_ix), int(doc_id), list(ctx)]


def test_items_and_batches_equal_the_reference(gold):
    tok = BertWordPieceTokenizer(os.path.join(G, "vocab.txt"))
    assert len(gold["cases"]) == 6
    seen_none_item = seen_label1 = seen_masked = seen_trimmed = False
    for case in gold["cases"]:
        ds = PD.SentContextDataset(os.path.join(G, "pretrain_notes.csv"), os.path.join(G, case["focus"]), "switch",
                                   case["n_ctx"], switch_prob=case["switch_prob"], same_doc=case["same_doc"])
        random.seed(case["seed"])
        items = [ds[i] for i in range(len(ds))]
        assert [_norm(it) for it in items] == case["items"]
        seen_none_item |= any(it is None for it in items)
        seen_label1 |= any(it is not None and it[1] == 1 for it in items)
        torch.manual_seed(case["seed"])
        i = k = 0
        for want in case["batches"]:
            b = case["sizes"][k % len(case["sizes"])]
            got = PD.switch_mlm_collate(items[i:i + b], tok, mlm_probability=case["mlm_probability"])
            i += b
            k += 1
            if want is None:
                assert got is None
                continue
            assert set(got) == set(want)
            for key, v in want.items():
                g = got[key].tolist() if isinstance(got[key], torch.Tensor) else got[key]
                assert g == v, (case["focus"], case["n_ctx"], key)
            if "mlm_labels" in got:
                seen_masked |= bool((got["mlm_labels"] != -100).any())
                # only focus-sentence tokens (type 0, not special) carry a label
                lab = got["mlm_labels"] != -100
                assert not (lab & (got["token_type_ids"] != 0)).any()
            seen_trimmed |= any("  " in s or s.startswith(" [SEP]") for s in got["sentences"])
    assert seen_none_item and seen_label1 and seen_masked and seen_trimmed


def test_select_inputs_keys():
    batch = {"input_ids": 1, "attention_mask": 2, "labels": 3, "mlm_labels": 4, "token_type_ids": 5, "sentences": []}
    assert set(PD.select_inputs(batch, "switch", 2)) == {"input_ids", "attention_mask", "labels", "mlm_labels", "token_type_ids"}
    assert set(PD.select_inputs(batch, "mlm", 2)) == {"input_ids", "attention_mask", "mlm_labels", "token_type_ids"}
    assert PD.select_inputs(batch, "mlm", 0)["labels"] == 4


def test_precomputed_instances_and_simple_collate_equal_the_reference(gold):
    ds = PD.PrecomputedDataset(os.path.join(G, "pretrain_insts.jsonl"))
    pre = gold["precomputed"]
    assert len(ds) == sum(pre["sizes"])
    torch.manual_seed(pre["seed"])
    i = 0
    kept = cand = 0
    for b, want in zip(pre["sizes"], pre["batches"]):
        got = PD.simple_collate([ds[j] for j in range(i, i + b)])
        cand += sum(sum(1 for t in ds[j]["mlm_labels"] if t != -100) for j in range(i, i + b))
        i += b
        assert set(got) == set(want)
        for key, v in want.items():
            g = got[key].tolist() if isinstance(got[key], torch.Tensor) else got[key]
            assert g == v, key
        kept += int((got["mlm_labels"] != -100).sum())
    assert 0 < kept < cand          # a subset of the candidate positions survives
