"""SURVEY.md §8(f4), end to end: the top-k file the scoring path writes -> SentContextDataset -> switch_mlm_collate
-> select_inputs -> BertContextMLMSwitch.forward on the HIP engine -> (loss, logits), against the CPU oracle's
restatement of sentclf/bert_model.py:137-186 on the SAME batches; and the evaluation pass of
sentclf/pretrain_tasks.py:308-331 (summed loss over the loader).  The data side is pinned bit for bit against
the imported reference in tests/test_pretrain_data.py (CPU)."""
import os
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from clip_amd import pretrain_data as PD  # noqa: E402
from clip_amd.bert_model import BertContextMLMSwitch  # noqa: E402
from clip_amd.config import BertConfig  # noqa: E402
from clip_amd.weights import procedural_state_dict  # noqa: E402
from clip_amd.wordpiece import BertWordPieceTokenizer  # noqa: E402
from oracle import bert_oracle as O  # noqa: E402  (checker only)

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = {"bf16": 2e-2, "f16": 5e-3, "fp32": 1e-4}


def _loader(tok, n_ctx, seed, batch_size=4, mlm_probability=0.4):
    ds = PD.SentContextDataset(os.path.join(G, "pretrain_notes.csv"), os.path.join(G, "pretrain_focus.csv"), "switch",
                               n_ctx, switch_prob=0.5)
    random.seed(seed)
    torch.manual_seed(seed)
    items = [ds[i] for i in range(len(ds))]
    return [PD.switch_mlm_collate(items[i:i + batch_size], tok, mlm_probability=mlm_probability)
            for i in range(0, len(items), batch_size)]


@pytest.mark.parametrize("precision", ["bf16", "f16", "fp32"])
def test_topk_file_to_loss(precision):
    tok = BertWordPieceTokenizer(os.path.join(G, "vocab.txt"))
    n_ctx = 2
    cfg = BertConfig(vocab_size=len(tok), hidden_size=256, num_hidden_layers=2, num_attention_heads=4,
                     intermediate_size=512, num_labels=2)
    sd = procedural_state_dict(cfg, 21, mlm_head=True)
    m = BertContextMLMSwitch(cfg, max_tokens=8192, max_seqs=64)
    m.set_precision(precision)
    m.load_state_dict(sd)
    m.to("cuda").eval()
    m.set_sep_token_id(tok.sep_token_id); m.set_n_context_sentences(n_ctx); m.set_task("binary")
    loader = _loader(tok, n_ctx, seed=3)
    assert len(loader) == 4 and all(b is not None for b in loader)
    assert sum(len(b["labels"]) for b in loader) == 13          # 14 focus records, one without a row
    tol = TOL[precision]
    want_total = 0.0
    for x in loader:
        assert int((x["mlm_labels"] != -100).sum()) > 0
        inputs = PD.select_inputs({k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in x.items()}, "switch", n_ctx)
        loss, logits = m(**inputs)
        with torch.no_grad():
            ref_loss, ref_logits, _ = O.mlm_switch_forward(sd, cfg, x["input_ids"], x["attention_mask"], x["token_type_ids"],
                                                           x["labels"], x["mlm_labels"], tok.sep_token_id, n_ctx)
        assert np.abs(logits.cpu().numpy() - ref_logits.numpy()).max() < tol
        assert abs(float(loss) - float(ref_loss)) < tol
        want_total += float(ref_loss)
    total = PD.evaluate(m, loader, "switch", torch.device("cuda"), n_ctx)
    assert abs(total - want_total) < tol * len(loader)
    m._drop_engine()
