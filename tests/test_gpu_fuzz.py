"""Randomised shapes through the whole engine against the CPU oracle: odd token counts (not multiples
of the 256-row GEMM tile), rows of the minimum length, rows of 512 tokens, one row, many tiny rows,
hidden sizes with and without the fused-LayerNorm flow, both heads, both numerics settings.  Seeds are
fixed: a failure names its case."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from clip_amd import scoring  # noqa: E402
from clip_amd.bert_model import args_to_model  # noqa: E402
from clip_amd.config import BertConfig  # noqa: E402
from clip_amd.weights import procedural_state_dict  # noqa: E402
from oracle import bert_oracle as O  # noqa: E402  (checker only)


def _rows(rng, n_rows, n_ctx, vocab, max_len):
    rows = []
    for _ in range(n_rows):
        nseg = 2 * n_ctx + 1
        kind = rng.random()
        if kind < 0.15:
            seg = np.zeros(nseg, int)                        # empty contexts ...
            seg[n_ctx] = 1 if n_ctx else int(rng.integers(0, 3))   # ... and a one-token (or empty, sentence-only) focus
        elif kind < 0.25:
            seg = rng.multinomial(max_len - 1 - nseg, np.ones(nseg) / nseg)   # exactly max_len tokens
            seg[n_ctx] = max(seg[n_ctx], 1)
            while 1 + nseg + seg.sum() > max_len:
                seg[int(np.argmax(seg))] -= 1
        else:
            seg = rng.integers(0, max(2, (max_len - 1 - nseg) // nseg), nseg)
            seg[n_ctx] = max(seg[n_ctx], 1)
        toks = [101]
        for s in seg:
            toks.extend(rng.integers(1000, vocab, int(s)).tolist())
            toks.append(102)
        rows.append(np.asarray(toks, np.int32))
    return rows


CASES = [  # (seed, hidden, heads, layers, intermediate, n_ctx, n_rows, max_len)
    (1, 128, 2, 2, 512, 2, 37, 96),      # unfused flow (hidden not a multiple of 256)
    (2, 256, 4, 2, 512, 2, 61, 200),     # fused-LayerNorm flow, one column tile
    (3, 512, 8, 1, 1024, 1, 23, 512),    # fused flow, two column tiles; one layer = the trimmed layer only
    (4, 256, 4, 3, 768, 0, 150, 64),     # sentence-only head, many short rows (down to [CLS][SEP])
    (5, 384, 6, 2, 1536, 2, 1, 300),     # a single row
    (6, 768, 12, 2, 3072, 2, 29, 512),   # BERT-base width, rows up to 512 tokens
]


@pytest.mark.parametrize("precision", ["bf16", "fp32"])
@pytest.mark.parametrize("case", CASES, ids=[f"seed{c[0]}_h{c[1]}_n{c[5]}" for c in CASES])
def test_random_rows_match_oracle(case, precision):
    seed, H, heads, layers, I, n_ctx, n_rows, max_len = case
    cfg = BertConfig(vocab_size=3000, hidden_size=H, num_hidden_layers=layers, num_attention_heads=heads,
                     intermediate_size=I)
    rng = np.random.default_rng(seed)
    rows = _rows(rng, n_rows, n_ctx, cfg.vocab_size, max_len)
    sd = procedural_state_dict(cfg, 100 + seed)
    model = args_to_model(n_ctx)(cfg, max_tokens=8192, max_seqs=256)
    model.load_state_dict(sd)
    model.set_precision(precision)
    model.to("cuda").eval()
    if n_ctx:
        model.set_sep_token_id(102)
        model.set_n_context_sentences(n_ctx)
    model.set_task("multilabel")
    # through the bulk scorer: length-bucketed packed steps of at most 3000 tokens (several steps, odd sizes)
    res = scoring.BulkScorer(model, keep_probs=True, tokens_per_step=3000).score(rows)
    p_ref, keep_ref, score_ref, logit_ref = O.score_rows(sd, cfg, rows, n_ctx, batch_size=8)
    tol = 5e-3 if precision == "bf16" else 2.5e-5        # sigmoid' <= 1/4 of the 2e-2 / 1e-4 logit budgets
    err = np.abs(res.probs - p_ref).max()
    assert np.isfinite(res.probs).all() and err < tol, err
    assert np.array_equal(res.max_score, res.probs.max(1))                 # decisions: exact functions of the scores
    assert np.array_equal(res.keep, (res.probs > np.float32(0.05)).any(1))
    # and the reference's padded batches of 8 through model.forward give the packed path's results
    ids, mask = O.pad_batch(rows[:8])
    kw = {"token_type_ids": O.token_type_ids_from_seps(ids, 102, n_ctx).cuda()} if n_ctx else {}
    (logits,) = model(input_ids=ids.cuda(), attention_mask=mask.cuda(), **kw)
    # (torch's sigmoid and the epilogue kernel's may differ in the last bit: compare through the logit)
    p8 = res.probs[:len(logits)].astype(np.float64)
    assert np.abs(np.log(p8 / (1 - p8)) - logits.cpu().numpy()).max() < 1e-5
