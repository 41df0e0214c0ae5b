"""The bulk scoring driver on an MI355X: device-side row preparation, length-bucketed packed steps,
fused epilogue — checked against the reference's golden outputs and the CPU oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from clip_amd import _lib, scoring, synth, topk  # noqa: E402
from clip_amd.bert_model import args_to_model  # noqa: E402
from clip_amd.weights import procedural_state_dict  # noqa: E402
from oracle import bert_oracle as O  # noqa: E402  (checker only)

DNOTE = np.array([0.44, 0.31, 0.24, 0.17, 0.46, 0.35000000000000003, 0.43])


def _model(cfg, seed, n_ctx, max_tokens=16384, max_seqs=512):
    m = args_to_model(n_ctx)(cfg, max_tokens=max_tokens, max_seqs=max_seqs)
    m.load_state_dict(procedural_state_dict(cfg, seed))
    m.to("cuda").eval()
    if n_ctx > 0:
        m.set_sep_token_id(102)
        m.set_n_context_sentences(n_ctx)
    m.set_task("multilabel")
    return m


@pytest.mark.parametrize("n_ctx", [2, 1, 0])
def test_prepare_packed_matches_collator_token_types(n_ctx):
    lib = _lib.load()
    rows = (synth.context_rows(40, n_ctx, 5000, max_len=300, seed=5, sigma=1.0, seg_hi=150) if n_ctx
            else synth.sentence_rows(40, 5000, seed=5))
    pr = scoring.PackedRows.from_rows(rows)
    flat, cu = pr.gather(np.arange(len(rows)))
    ids, cud = torch.from_numpy(flat).cuda(), torch.from_numpy(cu).cuda()
    T, n = len(flat), len(rows)
    tt, pos = torch.empty(T, dtype=torch.int32, device="cuda"), torch.empty(T, dtype=torch.int32, device="cuda")
    focus, err = torch.empty(n, dtype=torch.int32, device="cuda"), torch.zeros(1, dtype=torch.int32, device="cuda")
    _lib.check(lib.clip_prepare_packed(_lib.ptr(ids), _lib.ptr(cud), n, 102, n_ctx if n_ctx else -1, _lib.ptr(tt),
                                       _lib.ptr(pos), _lib.ptr(focus), _lib.ptr(err), None))
    torch.cuda.synchronize()
    assert err.item() == 0
    tt, pos, focus = tt.cpu().numpy(), pos.cpu().numpy(), focus.cpu().numpy()
    ids2d, _ = O.pad_batch(rows)
    ref_tt = O.token_type_ids_from_seps(ids2d, 102, n_ctx).numpy() if n_ctx else np.zeros_like(ids2d.numpy())
    for i, r in enumerate(rows):
        a, b = cu[i], cu[i + 1]
        assert np.array_equal(tt[a:b], ref_tt[i, :len(r)])            # bit-exact integer work
        assert np.array_equal(pos[a:b], np.arange(len(r)))
        exp = a + (np.nonzero(r == 102)[0][n_ctx] if n_ctx else 0)
        assert focus[i] == exp


def test_bulk_scorer_matches_reference_goldens(golden):
    g = golden("base_ctx")
    model = _model(g["cfg"], g["seed"], 2)
    sc = scoring.BulkScorer(model, any_threshold=0.05, label_threshs=DNOTE, keep_probs=True, tokens_per_step=4096)
    res = sc.score(g["rows"], doc_ids=np.arange(len(g["rows"])) // 4, sent_ixs=np.arange(len(g["rows"])) % 4)
    assert np.abs(res.probs - g["ref_probs"]).max() < 6e-3       # sigmoid' <= 1/4 of the 2e-2 logit budget
    _, keep_ref, score_ref = O.score_epilogue(torch.from_numpy(g["ref_logits"]), 0.05)
    assert np.abs(res.max_score - score_ref).max() < 6e-3
    assert np.array_equal(res.keep, keep_ref)
    clear = np.abs(g["ref_probs"] - DNOTE) > 6e-3
    assert (res.labels == O.label_decisions(g["ref_probs"], DNOTE))[clear].all()
    # decisions are exact functions of the engine's own fp32 probabilities
    assert np.array_equal(res.labels, O.label_decisions(res.probs, DNOTE))
    assert np.array_equal(res.max_score, res.probs.max(1))


@pytest.mark.parametrize("n_ctx", [2, 0])
def test_bulk_scorer_vs_oracle_many_ragged_rows(n_ctx):
    """600 ragged rows, several bucketed steps, results returned in input order."""
    from clip_amd.config import BertConfig
    cfg = BertConfig(vocab_size=2000, hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
                     intermediate_size=512)
    rows = (synth.context_rows(600, 2, cfg.vocab_size, max_len=200, seed=9, sigma=0.9, seg_hi=90) if n_ctx
            else synth.sentence_rows(600, cfg.vocab_size, seed=9))
    model = _model(cfg, 21, n_ctx, max_tokens=8192, max_seqs=128)
    res = scoring.BulkScorer(model, keep_probs=True).score(rows)
    sd = procedural_state_dict(cfg, 21)
    p_ref, keep_ref, score_ref, _ = O.score_rows(sd, cfg, rows, n_ctx, batch_size=8)
    assert np.abs(res.probs - p_ref).max() < 2e-3
    assert np.abs(res.max_score - score_ref).max() < 2e-3
    # top-50 by max_label_score: every row the engine ranks in its top-k is, by the oracle's score,
    # within the 2e-3 score tolerance of the oracle's k-th score (exact order needs identical scores)
    k = 50
    mine = topk.local_topk(res.doc_ids, res.sent_ixs, res.max_score, k)
    theirs = topk.local_topk(res.doc_ids, res.sent_ixs, score_ref, k)
    assert score_ref[mine["sent_ix"]].min() >= theirs["score"].min() - 4e-3
    # and with identical scores the order is bit-exact
    again = topk.local_topk(res.doc_ids, res.sent_ixs, res.max_score.copy(), k)
    assert np.array_equal(mine, again)


def test_bulk_scorer_flags_wrong_sep_count(golden):
    g = golden("tiny_ctx")
    model = _model(g["cfg"], g["seed"], 2)
    rows = [np.array([101, 5, 102, 6, 102, 7, 102], np.int32)]
    with pytest.raises(TypeError):
        scoring.BulkScorer(model).score(rows)


def test_bulk_scorer_edge_inputs(golden):
    """No rows at all, one row, a row longer than the position table, a row larger than a step."""
    g = golden("tiny_ctx")
    model = _model(g["cfg"], g["seed"], 2, max_tokens=2048, max_seqs=64)
    sc = scoring.BulkScorer(model, keep_probs=True)
    res = sc.score([])
    assert len(res.max_score) == 0 and res.probs.shape == (0, 7) and res.keep.dtype == bool
    one = sc.score(g["rows"][:1])
    many = sc.score(g["rows"][:9])
    assert np.array_equal(one.probs[0], many.probs[0])                 # a row's result does not depend on its batch
    too_long = np.concatenate(([101], np.full(600, 1500), [102, 102, 102, 102, 102])).astype(np.int32)
    with pytest.raises(_lib.ClipError):
        sc.score([too_long])                                           # 606 tokens > max_position_embeddings
    small = scoring.BulkScorer(model, tokens_per_step=64)
    with pytest.raises(ValueError):
        small.score([g["rows"][-1]])                                   # a 512-token row cannot fit a 64-token step
    with pytest.raises(ValueError):
        scoring.BulkScorer(model, tokens_per_step=4096)                # beyond the engine's workspace


@pytest.mark.parametrize("precision", ["bf16", "f16"])
def test_step_size_and_operand_layouts_do_not_change_a_bit(precision):
    """BERT-base, the same 420 ragged rows scored in steps of 2 048, 6 144 and 32 768 tokens.  The engine picks its
    kernels and operand layouts by the step's size — small steps: one-tile GEMMs, row-major operands; 6 144 tokens:
    the ping-pong kernel for the FFN with the intermediate as 128 x 64 blocks; 32 768: also the QKV GEMM on the
    block copy of its weight — and a row's scores must not depend on any of it (engine.hip: "the layout moves bytes,
    not arithmetic"; the reference scores a sentence the same whatever batch it lands in)."""
    from clip_amd.config import BertConfig
    cfg = BertConfig()
    rows = synth.context_rows(420, 2, cfg.vocab_size, max_len=256, seed=77, sigma=0.7, seg_hi=80)
    assert sum(len(r) for r in rows) > 34000
    model = args_to_model(2)(cfg, max_tokens=32768, max_seqs=1024)
    model.set_precision(precision)
    model.load_state_dict(procedural_state_dict(cfg, 5))
    model.to("cuda").eval()
    model.set_sep_token_id(102)
    model.set_n_context_sentences(2)
    model.set_task("multilabel")
    outs = []
    for tokens in (2048, 6144, 32768):
        res = scoring.BulkScorer(model, keep_probs=True, tokens_per_step=tokens).score(rows)
        outs.append(res.probs.copy())
    assert np.isfinite(outs[0]).all()
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
    model._drop_engine()
