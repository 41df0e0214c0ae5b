"""the inference forward of BertContextMLMSwitch (sentclf/bert_model.py:106-186)
through the HIP engine — encoder, LM prediction head on the labelled positions (clip_mlm_head), switch
classifier, both CrossEntropy losses — against goldens from the reference class itself."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers"))
from mlm_case import batches  # noqa: E402
from clip_amd.bert_model import BertContextMLMSwitch  # noqa: E402
from clip_amd.weights import procedural_state_dict  # noqa: E402

# logits / scores: 2e-2 (bf16, f16) and 1e-4 (fp32 check mode), BASELINE.json; the mean losses average the
# per-position errors and are held to the same bounds
TOL = {"bf16": 2e-2, "f16": 5e-3, "fp32": 1e-4}


def _model(g, precision):
    m = BertContextMLMSwitch(g["cfg"], max_tokens=16384, max_seqs=256)
    m.set_precision(precision)
    m.load_state_dict(procedural_state_dict(g["cfg"], g["seed"], mlm_head=True))
    m.to("cuda").eval()
    m.set_sep_token_id(102); m.set_n_context_sentences(g["n_ctx"]); m.set_task("binary")
    return m


@pytest.mark.parametrize("precision", ["bf16", "f16", "fp32"])
@pytest.mark.parametrize("name", ["tiny_mlm", "base_mlm"])
def test_mlm_switch_forward_matches_reference(golden, name, precision):
    g = golden(name)
    m = _model(g, precision)
    tol = TOL[precision]
    r0 = 0
    cols = torch.from_numpy(g["sample_cols"]).cuda()
    for b, (ids, mask, tt, sw, mlm, sl) in enumerate(batches(g)):
        loss, logits, scores = m(input_ids=ids.cuda(), attention_mask=mask.cuda(), token_type_ids=tt.cuda(),
                                 labels=sw.cuda(), mlm_labels=mlm.cuda(), return_mlm_scores=True)
        assert scores.shape == (sl.stop - sl.start, g["cfg"].vocab_size)
        assert np.abs(logits.cpu().numpy() - g["ref_logits"][r0:r0 + len(ids)]).max() < tol
        assert np.abs(scores[:, cols].cpu().numpy() - g["ref_sampled"][sl]).max() < tol
        assert np.abs(torch.logsumexp(scores, -1).cpu().numpy() - g["ref_lse"][sl]).max() < tol
        lab = mlm[mlm != -100].cuda()
        assert np.abs(scores.gather(1, lab[:, None])[:, 0].cpu().numpy() - g["ref_label_score"][sl]).max() < tol
        assert abs(float(loss) - g["ref_batch_loss"][b]) < tol
        if precision == "fp32":
            assert np.array_equal(scores.argmax(-1).cpu().numpy(), g["ref_argmax"][sl])
        # the reference's return shapes (bert_model.py:167-186)
        only_mlm = m(input_ids=ids.cuda(), attention_mask=mask.cuda(), token_type_ids=tt.cuda(), mlm_labels=mlm.cuda())
        assert len(only_mlm) == 2 and only_mlm[1] == () and torch.isfinite(only_mlm[0])
        r0 += len(ids)


def test_mlm_head_needs_untrimmed_forward_and_weights(golden):
    from clip_amd import _lib
    from clip_amd.bert_model import BertSequenceMultilabelClassificationContext
    g = golden("tiny_mlm")
    sd = procedural_state_dict(g["cfg"], g["seed"], mlm_head=True)
    plain = BertSequenceMultilabelClassificationContext(g["cfg"], max_tokens=4096, max_seqs=64)
    plain.load_state_dict({k: v for k, v in sd.items() if not k.startswith("mlm_predictions.")})
    plain.to("cuda")
    h = plain._ensure_engine()
    z = torch.zeros(4, dtype=torch.int32, device="cuda")
    loss = torch.zeros(1, device="cuda")
    rc = _lib.load().clip_mlm_head(h, _lib.ptr(z), _lib.ptr(z), 4, None, _lib.ptr(loss), 0, None)
    assert rc == _lib.CLIP_ERR_STATE and b"mlm_predictions" in _lib.load().clip_last_error()


def test_expand_vocab_by_num(golden):
    """bert_model.py:116-135: more word embeddings (unit-norm rows), a decoder of the new size with zero bias."""
    g = golden("tiny_mlm")
    m = _model(g, "bf16")
    V = g["cfg"].vocab_size
    m.expand_vocab_by_num(3)
    sd = m.state_dict()
    assert m.config.vocab_size == V + 3 and sd["bert.embeddings.word_embeddings.weight"].shape[0] == V + 3
    assert torch.allclose(sd["bert.embeddings.word_embeddings.weight"][V:].norm(dim=1), torch.ones(3), atol=1e-5)
    assert sd["mlm_predictions.decoder.weight"].shape == (V + 3, g["cfg"].hidden_size)
    ids, mask, tt, sw, mlm, sl = next(batches(g))
    ids[0, 1] = V + 2                                         # a new token id is now a valid input
    loss, logits, scores = m(input_ids=ids.cuda(), attention_mask=mask.cuda(), token_type_ids=tt.cuda(),
                             labels=sw.cuda(), mlm_labels=mlm.cuda(), return_mlm_scores=True)
    assert scores.shape[1] == V + 3 and torch.isfinite(scores).all() and torch.isfinite(loss)
    g["cfg"].vocab_size = V                                   # the golden's config object is shared by the session


def test_mlm_labels_on_another_device_and_ignored_switch_labels(golden):
    """mlm_labels may arrive on the CPU while the other tensors are on the GPU, and a switch label of -100 is ignored
    by the reference's CrossEntropyLoss (bert_model.py:177-183) exactly as an MLM label of -100 is."""
    import torch.nn.functional as F
    g = golden("tiny_mlm")
    m = _model(g, "fp32")
    ids, mask, tt, sw, mlm, sl = next(batches(g))
    loss_gpu, logits, scores = m(input_ids=ids.cuda(), attention_mask=mask.cuda(), token_type_ids=tt.cuda(),
                                 labels=sw.cuda(), mlm_labels=mlm.cuda(), return_mlm_scores=True)
    loss_cpu_labels, _ = m(input_ids=ids.cuda(), attention_mask=mask.cuda(), token_type_ids=tt.cuda(),
                           labels=sw, mlm_labels=mlm)                       # labels left on the CPU
    assert float(loss_cpu_labels) == float(loss_gpu)
    sw2 = sw.clone()
    sw2[0] = -100
    loss_ign, _ = m(input_ids=ids.cuda(), attention_mask=mask.cuda(), token_type_ids=tt.cuda(), labels=sw2.cuda(),
                    mlm_labels=mlm.cuda())
    lab = mlm[mlm != -100].cuda()
    want = F.cross_entropy(scores, lab) + F.cross_entropy(logits, sw2.cuda(), ignore_index=-100)
    assert torch.isfinite(loss_ign) and abs(float(loss_ign) - float(want)) < 1e-5
    sw3 = torch.full_like(sw, -100)                                         # no switch label left: torch's mean over zero rows
    loss_nan, _ = m(input_ids=ids.cuda(), attention_mask=mask.cuda(), token_type_ids=tt.cuda(), labels=sw3.cuda(),
                    mlm_labels=mlm.cuda())
    assert torch.isnan(loss_nan)
