"""BASELINE.json config #1: the reference's test_load.py flow (examples/label_sentences.py) — local
from_pretrained directory, setters, float64 per-label thresholds from .npy, SentEvalDataset over
dummy_test_data.csv, DataLoader(batch_size=1) + ctx_collator + select_inputs, `loss, pred = model(**inputs)`,
sigmoid, strict `>` against the thresholds, label names — against the CPU oracle doing the same."""
import importlib.util
import os
import shutil

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from clip_amd import collate, datasets  # noqa: E402
from clip_amd.bert_model import BertSequenceMultilabelClassificationContext  # noqa: E402
from clip_amd.config import BertConfig  # noqa: E402
from clip_amd.constants import LABEL_TYPES  # noqa: E402
from clip_amd.weights import procedural_state_dict  # noqa: E402
from clip_amd.wordpiece import BertWordPieceTokenizer  # noqa: E402
from oracle import bert_oracle as O  # noqa: E402  (checker only)

HERE = os.path.dirname(os.path.abspath(__file__))
G = os.path.join(HERE, "golden")
DNOTE = np.array([0.44, 0.31, 0.24, 0.17, 0.46, 0.35000000000000003, 0.43])   # the values of clip_dnote_label_threshs.npy


@pytest.mark.parametrize("fmt", ["safetensors", "bin"])
def test_reference_example_flow(tmp_path, fmt, capsys):
    cfg = BertConfig()                                    # BERT-base geometry, 201-key state dict
    sd = procedural_state_dict(cfg, 17)
    ckpt = tmp_path / "CLIP_DNote_BERT_Context"
    model = BertSequenceMultilabelClassificationContext(cfg)
    model.load_state_dict(dict(sd, **{"bert.embeddings.position_ids": torch.arange(512).unsqueeze(0)}))   # old checkpoints carry it
    model.save_pretrained(str(ckpt))
    if fmt == "bin":
        os.remove(ckpt / "model.safetensors")
        torch.save(sd, ckpt / "pytorch_model.bin")
    shutil.copyfile(os.path.join(G, "vocab.txt"), ckpt / "vocab.txt")
    np.save(tmp_path / "clip_dnote_label_threshs.npy", DNOTE)
    spec = importlib.util.spec_from_file_location("example_test_load", os.path.join(HERE, "..", "examples", "label_sentences.py"))
    ex = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ex)
    csv_path = os.path.join(G, "dummy_test_data.csv")
    got = ex.main(str(ckpt), str(tmp_path / "clip_dnote_label_threshs.npy"), csv_path)
    printed = [l for l in capsys.readouterr().out.splitlines() if l.startswith("pred_labels: ")]
    assert len(got) == len(printed) == 7                  # one line per row of the fixture, as the reference prints

    tok = BertWordPieceTokenizer(os.path.join(G, "vocab.txt"))
    ds = datasets.SentEvalDataset(csv_path, "multilabel", n_context_sentences=2)
    for i, (loss, probs, names) in enumerate(got):
        x = collate.ctx_collator([ds[i]], "multilabel", tok)
        with torch.no_grad():
            ref_loss, ref_logits = O.context_forward(sd, cfg, x["input_ids"], x["attention_mask"], x["token_type_ids"],
                                                     x["labels"], tok.sep_token_id, 2)
        ref_p = torch.sigmoid(ref_logits)[0].numpy()
        assert np.abs(probs - ref_p).max() < 5e-3 and abs(loss - float(ref_loss)) < 5e-3
        clear = np.abs(ref_p - DNOTE) > 5e-3
        mine = np.isin(np.arange(7), [LABEL_TYPES.index(n) for n in names])
        assert (mine == (ref_p > DNOTE))[clear].all()
        assert printed[i] == f"pred_labels: {names}"
