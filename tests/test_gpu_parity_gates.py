"""The north-star parity criteria as gates (BASELINE.json): per-label logits within 2e-2 (bf16) / 1e-4
(fp32 check mode) of the reference, >= 99.9 % agreement on thresholded labels — ALL decisions counted,
none excused for being close to its threshold — with both threshold files of the reference
(test_load.py:29,37-38), at the bench's size (fixed L = 256) and on C5-style ragged rows up to 512 tokens,
and on weights with trained-like statistics (tests/golden/base_*_stress.npz from the imported reference).
The fp32 check mode is the stand-in for the reference where no golden exists (it is pinned to the
reference at <= 1e-4 on every golden, below and in test_gpu_model.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from clip_amd import scoring, synth  # noqa: E402
from clip_amd.bert_model import args_to_model  # noqa: E402
from clip_amd.config import BertConfig  # noqa: E402
from clip_amd.weights import procedural_state_dict, trained_like_state_dict  # noqa: E402
from oracle import bert_oracle as O  # noqa: E402  (checker only)
from conftest import golden_state_dict  # noqa: E402

DNOTE = np.array([0.44, 0.31, 0.24, 0.17, 0.46, 0.35000000000000003, 0.43])   # clip_dnote_label_threshs.npy
TTP = np.array([0.18, 0.6, 0.11, 0.32, 0.1, 0.1, 0.18])                        # clip_ttp_label_threshs.npy


def _model(cfg, sd, n_ctx, precision, tokens=32768):
    m = args_to_model(n_ctx)(cfg, max_tokens=tokens, max_seqs=4096)
    if precision is not None:           # None: the default of the classes, 'auto'
        m.set_precision(precision)
    m.load_state_dict(sd)
    m.to("cuda").eval()
    if n_ctx:
        m.set_sep_token_id(102)
        m.set_n_context_sentences(n_ctx)
    m.set_task("multilabel")
    return m


def _probs(cfg, sd, rows, n_ctx, precision):
    m = _model(cfg, sd, n_ctx, precision)
    res = scoring.BulkScorer(m, keep_probs=True).score(rows)
    m._drop_engine()
    return res.probs


def _agreement(p_a, p_b, thr):
    a, b = O.label_decisions(p_a, thr), O.label_decisions(p_b, thr)
    return float((a == b).mean()), int((a != b).sum())


def _logits(cfg, sd, rows, n_ctx, precision, batch_size):
    """the reference's loop (batches of `batch_size` padded rows) through the nn.Module mirror"""
    import torch
    m = _model(cfg, sd, n_ctx, precision)
    out = []
    for i in range(0, len(rows), batch_size):
        ids, mask = O.pad_batch(rows[i:i + batch_size])
        kw = {"token_type_ids": O.token_type_ids_from_seps(ids, 102, n_ctx).cuda()} if n_ctx else {}
        out.append(m(input_ids=ids.cuda(), attention_mask=mask.cuda(), **kw)[0].cpu().numpy())
    m._drop_engine()
    return np.concatenate(out)


# Operand rounding is RELATIVE: the logit error of a 16-bit-operand forward scales with the logit spread.
# Measured floor of bf16 operands (tools/rounding_study.py: weights, LayerNorm outputs, GELU outputs and
# attention outputs each rounded to 8 bits): ~0.5 % rms of the logit spread — inside 2e-2 while logits stay
# within +-2 (procedural weights: test_gpu_model.py asserts 2e-2 absolute there), 3x outside it on the
# trained-like stress goldens (logits -6.7 .. +7.9).  fp16 operands (same MFMA rate) round 8x finer: 2.1e-2
# on base_ctx_stress with an fp16 residual copy ('f16'), 1.97e-2 with the residual stream kept in fp32 ('f16r').
# The criteria are therefore asserted, at north_star's own numbers, on the setting the DEFAULT (precision
# 'auto', clip_amd/precision.py) selects and on 'f16r'; the asserts on fixed 'bf16' / 'f16' where they miss
# the bar are labelled regression guards.
BAND = 2.5


@pytest.mark.parametrize("kind", ["fixed256", "c5_varlen"])
@pytest.mark.parametrize("weights,precision", [("procedural", "auto"), ("trained_like", "auto"),
                                               ("procedural", "bf16"), ("procedural", "f16"),
                                               ("trained_like", "f16"), ("trained_like", "f16r"),
                                               ("trained_like", "bf16")])
def test_thresholded_label_agreement_at_scale(kind, weights, precision):
    cfg = BertConfig()
    sd = (trained_like_state_dict if weights == "trained_like" else procedural_state_dict)(cfg, 0)
    if kind == "fixed256":
        rows = synth.context_rows(4096, 2, cfg.vocab_size, seed=77, fixed_len=256)
    else:
        flat, off = synth.packed_context_rows(4096, 2, cfg.vocab_size, 512, seed=78, sigma=0.9, seg_hi=200)
        rows = scoring.PackedRows(flat, off)
    if precision == "auto":
        # the DEFAULT path: no set_precision call at all — BulkScorer calibrates on these rows
        m = _model(cfg, sd, 2, None)
        p16 = scoring.BulkScorer(m, keep_probs=True).score(rows).probs
        print(f"{kind}/{weights}: auto selected {m.resolved_precision}: {m.precision_report['candidates']}")
        assert m.precision == "auto"
        if weights == "procedural":
            assert m.resolved_precision == "bf16"
        else:
            assert m.resolved_precision in ("f16", "f16r")      # bf16 must have been measured out
        m._drop_engine()
    else:
        p16 = _probs(cfg, sd, rows, 2, precision)
    p32 = _probs(cfg, sd, rows, 2, "fp32")
    for name, thr in (("dnote", DNOTE), ("ttp", TTP)):
        agree, n_bad = _agreement(p16, p32, thr)
        print(f"{kind}/{weights}/{precision}/{name}: agreement {100 * agree:.4f} % ({n_bad} of {p16.size} decisions differ)")
        if not (weights == "trained_like" and precision == "bf16"):
            assert agree >= 0.999
        else:
            # REGRESSION GUARD, not the criterion: fixed bf16 operands on trained-like weights measure 99.79-99.93 %
            # (tools/agreement_ab.py) — below the 99.9 % bar on the ttp thresholds, which is why the default is
            # 'auto' and measures bf16 out for such a checkpoint (asserted above)
            assert agree >= 0.997


@pytest.mark.parametrize("name", ["base_ctx_stress", "base_sent_stress"])
def test_trained_like_stress_golden(golden, name):
    """Reference logits beyond +-6 (base_ctx_stress: -6.7 .. +7.9), LayerNorm gamma outliers, outlier
    hidden dimensions, heavy-tailed embeddings; goldens from the imported reference.  north_star's 2e-2 ABSOLUTE
    is asserted for the setting the default ('auto') selects and for 'f16r'; 1e-4 for the fp32 check mode."""
    g = golden(name)
    sd = golden_state_dict(g)
    ref = g["ref_logits"]
    err = {}
    for precision in ("fp32", "auto", "f16r", "f16", "bf16"):
        logits = _logits(g["cfg"], sd, g["rows"], g["n_ctx"], None if precision == "auto" else precision, g["batch_size"])
        assert np.isfinite(logits).all()
        err[precision] = np.abs(logits - ref)
        print(f"{name}/{precision}: max |logit - ref| = {err[precision].max():.3e}, "
              f"relative to max(1, |ref|/{BAND}): {(err[precision] / np.maximum(1, np.abs(ref) / BAND)).max():.3e}")
        p = 1 / (1 + np.exp(-logits.astype(np.float64)))
        if precision != "bf16":
            for thr in (DNOTE, TTP):
                assert _agreement(p.astype(np.float32), g["ref_probs"], thr)[0] == 1.0
    # fp32 check mode: 1e-4, relative above |logit| = 1 (fp32 itself resolves 1e-6 of a logit of 8)
    assert (err["fp32"] / np.maximum(1, np.abs(ref))).max() < 1e-4
    assert err["auto"].max() < 2e-2            # the default path, north_star's tolerance as stated
    assert err["f16r"].max() < 2e-2
    # regression guards for the fixed settings that miss 2e-2 absolute on base_ctx_stress (measured 2.14e-2 / 6.4e-2)
    assert (err["f16"] / np.maximum(1, np.abs(ref) / BAND)).max() < 2.5e-2
    assert (err["bf16"] / np.maximum(1, np.abs(ref) / BAND)).max() < 6e-2
    if name == "base_sent_stress":          # logits -2.9 .. +0.4
        assert err["f16"].max() < 2e-2


def test_auto_refuses_f16_when_an_activation_leaves_its_range(monkeypatch):
    """Weights scaled so that fp16 activations overflow: auto must not select a setting whose sampled scores are
    non-finite, a fixed 'f16' must raise instead of silently dropping sentences (scoring.py), bf16 still scores."""
    cfg = BertConfig(vocab_size=2000, hidden_size=256, num_hidden_layers=2, num_attention_heads=4, intermediate_size=1024)
    sd = procedural_state_dict(cfg, 9)
    for k in ("bert.encoder.layer.0.intermediate.dense.weight", "bert.encoder.layer.0.intermediate.dense.bias"):
        sd[k] = sd[k] * 1.0e5      # FFN intermediate of ~+-3e4 rms: beyond 65504 in places; the next LayerNorm rescales
    sd["bert.encoder.layer.0.output.dense.weight"] = sd["bert.encoder.layer.0.output.dense.weight"] * 1e-4
    rows = synth.context_rows(256, 2, cfg.vocab_size, max_len=96, seed=5)
    m = _model(cfg, sd, 2, "f16", tokens=16384)
    with pytest.raises(FloatingPointError, match="non-finite"):
        scoring.BulkScorer(m).score(rows)
    ids, mask = O.pad_batch(rows[:8])          # ... and so does the nn.Module forward
    with pytest.raises(FloatingPointError, match="non-finite"):
        m(input_ids=ids.cuda(), attention_mask=mask.cuda(), token_type_ids=O.token_type_ids_from_seps(ids, 102, 2).cuda())
    m._drop_engine()
    # auto tries the candidates fastest first and would stop at bf16 here; with the fp16 settings tried first
    # (the order a caller with an fp16-only accuracy need would see) both must be refused, not selected
    from clip_amd import precision as P
    monkeypatch.setattr(P, "CANDIDATES", ("f16", "f16r", "bf16"))
    m = _model(cfg, sd, 2, None, tokens=16384)
    res = scoring.BulkScorer(m, keep_probs=True).score(rows)
    rep = m.precision_report
    print(rep)
    assert "non-finite" in rep["candidates"]["f16"]["refused"] and "non-finite" in rep["candidates"]["f16r"]["refused"]
    assert m.resolved_precision == "bf16" and np.isfinite(res.probs).all()
    m._drop_engine()


@pytest.mark.parametrize("name", ["tiny_ctx", "tiny_sent", "base_ctx", "base_sent"])
def test_f16_setting_on_the_benign_goldens(golden, name):
    g = golden(name)
    logits = _logits(g["cfg"], golden_state_dict(g), g["rows"], g["n_ctx"], "f16", g["batch_size"])
    err = np.abs(logits - g["ref_logits"]).max()
    print(f"{name}/f16: max |logit - ref| = {err:.3e}")
    assert err < 3e-3          # bf16 on the same goldens: 1.2e-3 .. 7.3e-3


@pytest.mark.parametrize("name", ["tiny_ctx", "base_ctx", "base_sent", "base_ctx_stress"])
def test_f16r_fused_flow_equals_unfused_flow(golden, name):
    """F16R keeps the residual stream in fp32 in BOTH LayerNorm flows: fused (fp32 tiled copy rewritten in place by
    clip_gemm_f16_lnout_res32; hidden sizes that are multiples of 256) and unfused (fp32 pre-LayerNorm rows +
    statistics).  Same logits to within fp16 operand rounding, both inside the budget against the reference; the
    bulk path and the batches-of-8 path give the same bits in the fused flow."""
    import os
    g = golden(name)
    sd = golden_state_dict(g)
    out = {}
    for flow in ("1", "0"):
        old = os.environ.get("CLIP_AMD_FUSED_LN")
        os.environ["CLIP_AMD_FUSED_LN"] = flow
        try:
            out[flow] = _logits(g["cfg"], sd, g["rows"], g["n_ctx"], "f16r", g["batch_size"])
            if flow == "1":
                m = _model(g["cfg"], sd, g["n_ctx"], "f16r")
                bulk = scoring.BulkScorer(m, keep_logits=True).score(g["rows"]).logits
                m._drop_engine()
                assert np.array_equal(bulk, out[flow])
        finally:
            if old is None:
                os.environ.pop("CLIP_AMD_FUSED_LN", None)
            else:
                os.environ["CLIP_AMD_FUSED_LN"] = old
    tol = 2e-2 if name.endswith("stress") else 3e-3
    for flow, lg in out.items():
        err = np.abs(lg - g["ref_logits"]).max()
        print(f"{name}/f16r/fused_ln={flow}: max |logit - ref| = {err:.3e}")
        assert err < tol
    # (on the stress golden the two flows carry independent fp16-operand errors of up to ~2e-2 each)
    assert np.abs(out["1"] - out["0"]).max() < (2.5e-2 if name.endswith("stress") else 2e-3)


@pytest.mark.parametrize("precision", ["bf16", "f16", "f16r"])
def test_small_m_fused_gemm_same_bits(precision, monkeypatch):
    """Calls of a few hundred to a few thousand tokens run their fused-LayerNorm GEMMs on 64 x 64 or 64 x 256 tiles
    (gemm4.hip) instead of the ping-pong kernel's 256 x 256: the logits must be the SAME BITS, for the reference's batches of 8
    (apply_bert_unlabeled.py:92-110), for a mid-sized packed step and against a big step that the ping-pong kernel
    takes in both runs."""
    cfg = BertConfig()
    sd = procedural_state_dict(cfg, 4)
    flat, off = synth.packed_context_rows(700, 2, cfg.vocab_size, 256, seed=31)
    rows = scoring.PackedRows(flat, off)
    small = [rows.flat[rows.offsets[i]:rows.offsets[i + 1]] for i in range(24)]
    out = {}
    for flag in ("2", "1", "0"):       # 2 (the default): 64 x 64 tiles where their grid fits (<= 1280 tokens), else 64 x 256
        monkeypatch.setenv("CLIP_AMD_SMALL_M", flag)
        m = _model(cfg, sd, 2, precision, tokens=131072)
        big = scoring.BulkScorer(m, keep_logits=True).score(rows).logits                          # one step of ~78 k tokens
        mid = scoring.BulkScorer(m, keep_logits=True, tokens_per_step=4096).score(rows).logits   # steps of <= 4096 tokens
        tiny = scoring.BulkScorer(m, keep_logits=True, tokens_per_step=1024).score(rows).logits  # steps of <= 1024 tokens
        m._drop_engine()
        out[flag] = (big, mid, tiny, _logits(cfg, sd, small, 2, precision, 8))                    # batches of 8 padded rows
    for flag in ("2", "1"):
        for a, b in zip(out[flag], out["0"]):
            assert np.array_equal(a, b), flag
    assert np.array_equal(out["2"][0], out["2"][1]) and np.array_equal(out["2"][0], out["2"][2])
    assert np.array_equal(out["2"][0][:24], out["2"][3])


@pytest.mark.parametrize("hidden,heads,inter", [(1024, 16, 4096), (512, 8, 2048), (256, 4, 1024)])
def test_small_m_kernels_same_bits_other_widths(hidden, heads, inter, monkeypatch):
    """The small-M kernels at the other hidden sizes the engine takes (BERT-large's 1024: 16 column chunks per row panel,
    4 column tiles; 512; 256: fewer K tiles than the 64 x 64 kernel's ring holds): batches of 8, small packed steps and
    one big step give the SAME BITS, with the kernels on (2), restricted to 64 x 256 tiles (1) and off (0)."""
    cfg = BertConfig(vocab_size=3000, hidden_size=hidden, num_hidden_layers=2, num_attention_heads=heads, intermediate_size=inter)
    sd = procedural_state_dict(cfg, 6)
    flat, off = synth.packed_context_rows(400, 2, cfg.vocab_size, 256, seed=32)
    rows = scoring.PackedRows(flat, off)
    small = [rows.flat[rows.offsets[i]:rows.offsets[i + 1]] for i in range(24)]
    out = {}
    for flag in ("2", "1", "0"):
        monkeypatch.setenv("CLIP_AMD_SMALL_M", flag)
        m = _model(cfg, sd, 2, "bf16", tokens=65536)
        big = scoring.BulkScorer(m, keep_logits=True).score(rows).logits
        tiny = scoring.BulkScorer(m, keep_logits=True, tokens_per_step=1024).score(rows).logits
        m._drop_engine()
        out[flag] = (big, tiny, _logits(cfg, sd, small, 2, "bf16", 8))
    assert np.isfinite(out["2"][0]).all()
    for flag in ("2", "1"):
        for a, b in zip(out[flag], out["0"]):
            assert np.array_equal(a, b), flag
    assert np.array_equal(out["2"][0], out["2"][1]) and np.array_equal(out["2"][0][:24], out["2"][2])
