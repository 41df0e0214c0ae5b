"""End-to-end parity of the HIP engine against the reference's own outputs (tests/golden/*.npz),
through the host mirror of the reference's model API (clip_amd/bert_model.py -> C ABI)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from clip_amd.bert_model import args_to_model  # noqa: E402
from clip_amd.weights import procedural_state_dict  # noqa: E402
from oracle import bert_oracle as O  # noqa: E402  (checker only)

DNOTE = np.array([0.44, 0.31, 0.24, 0.17, 0.46, 0.35000000000000003, 0.43])   # clip_dnote_label_threshs.npy
TTP = np.array([0.18, 0.6, 0.11, 0.32, 0.1, 0.1, 0.18])                        # clip_ttp_label_threshs.npy
BF16_LOGIT_TOL = 2e-2   # BASELINE.json north_star: per-label logits within 2e-2 abs in bf16


def build_model(g):
    model = args_to_model(g["n_ctx"])(g["cfg"], max_tokens=16384, max_seqs=256)
    from conftest import golden_state_dict
    model.load_state_dict(golden_state_dict(g))
    model.to("cuda").eval()
    if g["n_ctx"] > 0:
        model.set_sep_token_id(102)
        model.set_n_context_sentences(g["n_ctx"])
    model.set_task("multilabel")
    return model


def run_padded(model, g):
    """the reference's loop: DataLoader(batch_size, shuffle=False), pad to longest, forward"""
    logits, losses = [], []
    bs = g["batch_size"]
    for i in range(0, len(g["rows"]), bs):
        ids, mask = O.pad_batch(g["rows"][i:i + bs])
        inputs = {"input_ids": ids.cuda(), "attention_mask": mask.cuda(),
                  "labels": torch.from_numpy(g["labels"][i:i + bs]).cuda()}
        if g["n_ctx"] > 0:
            inputs["token_type_ids"] = O.token_type_ids_from_seps(ids, 102, g["n_ctx"]).cuda()
        loss, pred = model(**inputs)
        logits.append(pred.cpu().numpy())
        losses.append(float(loss))
    return np.concatenate(logits), np.asarray(losses)


@pytest.mark.parametrize("name", ["tiny_ctx", "tiny_ctx_n1", "tiny_sent", "base_ctx", "base_sent"])
def test_forward_matches_reference(golden, name):
    g = golden(name)
    model = build_model(g)
    logits, losses = run_padded(model, g)
    err = np.abs(logits - g["ref_logits"]).max()
    print(f"{name}: max |logit - ref| = {err:.3e}")
    assert np.isfinite(logits).all()
    assert err < BF16_LOGIT_TOL
    assert np.abs(losses - g["ref_batch_loss"]).max() < 5e-3
    # thresholded labels: every decision whose reference probability is not within 5e-3 of its
    # threshold must agree (the >= 99.9 % agreement target is measured at scale in bench/scoring tests)
    p = 1 / (1 + np.exp(-logits.astype(np.float64)))
    for thr in (DNOTE, TTP):
        ref_dec = O.label_decisions(g["ref_probs"], thr)
        dec = O.label_decisions(p.astype(np.float32), thr)
        clear = np.abs(g["ref_probs"] - thr) > 5e-3
        assert (dec == ref_dec)[clear].all()


def test_packed_path_equals_padded_path(golden):
    """All rows in ONE packed call (the scoring driver's path) vs batches of 8 padded rows."""
    g = golden("base_ctx")
    model = build_model(g)
    padded, _ = run_padded(model, g)
    rows = g["rows"]
    lens = np.asarray([len(r) for r in rows])
    ids = torch.from_numpy(np.concatenate(rows).astype(np.int32)).cuda()
    cu = torch.from_numpy(np.concatenate(([0], np.cumsum(lens))).astype(np.int32)).cuda()
    pos = torch.from_numpy(np.concatenate([np.arange(n) for n in lens]).astype(np.int32)).cuda()
    ids2d, _ = O.pad_batch(rows)
    tt2d = O.token_type_ids_from_seps(ids2d, 102, 2)
    tt = torch.cat([tt2d[i, :n] for i, n in enumerate(lens)]).to(torch.int32).cuda()
    logits = model.forward_packed(ids, tt, pos, cu, int(lens.max())).cpu().numpy()
    # same kernels, same per-row arithmetic: batching must not change a row's result at all
    assert np.array_equal(logits, padded)
    assert np.abs(logits - g["ref_logits"]).max() < BF16_LOGIT_TOL


def test_wrong_sep_count_raises_like_the_reference(golden):
    g = golden("tiny_ctx")
    model = build_model(g)
    ids = torch.tensor([[101, 5, 102, 6, 102, 7, 102, 0]], device="cuda")       # 3 seps, needs 5
    mask = (ids != 0).long()
    with pytest.raises(TypeError):
        model(input_ids=ids, attention_mask=mask, token_type_ids=torch.zeros_like(ids),
              labels=torch.zeros(1, 7, device="cuda"))


@pytest.mark.parametrize("name", ["tiny_ctx", "tiny_sent"])
@pytest.mark.parametrize("empty", [0, 3])
def test_fully_masked_row_is_rejected(golden, name, empty):
    """A padded batch with a row of no unmasked token (first or LAST row: its packed offset is total_tokens) is caught
    by the device-side row check behind the forward — the padded entry point reads no lengths on the host."""
    g = golden(name)
    model = build_model(g)
    ids, mask = O.pad_batch(g["rows"][:4])
    mask[empty] = 0
    with pytest.raises(ValueError, match="row table"):
        model(input_ids=ids.cuda(), attention_mask=mask.cuda(), token_type_ids=torch.zeros_like(ids).cuda())
    mask[empty] = (ids[empty] != 0).long()        # the engine's error word was cleared: the next call is clean
    out = model(input_ids=ids.cuda(), attention_mask=mask.cuda(),
                token_type_ids=O.token_type_ids_from_seps(ids, 102, g["n_ctx"]).cuda() if g["n_ctx"] > 0 else None)[0]
    assert torch.isfinite(out).all()
    assert np.abs(out.cpu().numpy() - g["ref_logits"][:4]).max() < BF16_LOGIT_TOL


def _packed_by_indexing(model, ids, mask, tt, pos):
    """the padded call restated with boolean-mask indexing, through the packed entry point"""
    m = mask.bool()
    lens = m.sum(1)
    cu = torch.zeros(ids.shape[0] + 1, dtype=torch.int32)
    cu[1:] = torch.cumsum(lens, 0)
    return model.forward_packed(ids[m].int().cuda(), tt[m].int().cuda(), pos.expand_as(ids)[m].int().cuda(), cu.cuda(),
                                int(ids.shape[1]))


def test_padded_call_packs_on_the_device_like_boolean_indexing(golden):
    """clip_forward_padded's three pack kernels against torch's `tensor[mask]`: right-padded rows, HOLES in the mask
    (a dropped token does not renumber the default positions after it), caller's position ids per row and as one
    broadcast row, no mask at all."""
    g = golden("tiny_ctx")
    model = build_model(g)
    ids, mask = O.pad_batch(g["rows"][:6])
    B, S = ids.shape
    tt = O.token_type_ids_from_seps(ids, 102, g["n_ctx"])
    col = torch.arange(S).unsqueeze(0)
    holes = mask.clone()
    for b in (1, 4):
        j = next(j for j in range(1, S) if ids[b, j] not in (0, 101, 102))
        holes[b, j] = 0
    gen = torch.Generator().manual_seed(3)
    rand_pos = torch.randint(0, g["cfg"].max_position_embeddings, (B, S), generator=gen)
    cases = [("right-padded", mask, None, col), ("holes", holes, None, col), ("positions per row", mask, rand_pos, rand_pos),
             ("one broadcast row", holes, rand_pos[:1], rand_pos[:1]), ("1-d positions", mask, rand_pos[0], rand_pos[:1])]
    for name, mk, pos_arg, pos_ref in cases:
        out = model(input_ids=ids.cuda(), attention_mask=mk.cuda(), token_type_ids=tt.cuda(),
                    position_ids=None if pos_arg is None else pos_arg.cuda())[0]
        assert torch.equal(out, _packed_by_indexing(model, ids, mk, tt, pos_ref)), name
    full = ids.clone()
    full[full == 0] = 7                      # no mask: every column is a token
    out = model(input_ids=full.cuda(), token_type_ids=tt.cuda())[0]
    assert torch.equal(out, _packed_by_indexing(model, full, torch.ones_like(full), tt, col))
    out = model(input_ids=ids.int().cuda(), attention_mask=mask.bool().cuda(), token_type_ids=tt.int().cuda())[0]   # other dtypes
    assert torch.equal(out, _packed_by_indexing(model, ids, mask, tt, col))


def test_padded_call_with_more_rows_than_one_scan_block(golden):
    """2 500 rows: the prefix sum of the row lengths takes three rounds of its 1 024-thread workgroup."""
    g = golden("tiny_sent")
    model = args_to_model(0)(g["cfg"], max_tokens=131072, max_seqs=4096)
    from conftest import golden_state_dict
    model.load_state_dict(golden_state_dict(g))
    model.to("cuda").eval()
    model.set_task("multilabel")
    rows = [g["rows"][i % len(g["rows"])][:1 + (i * 7) % 23] for i in range(2500)]
    ids, mask = O.pad_batch(rows)
    out = model(input_ids=ids.cuda(), attention_mask=mask.cuda())[0]
    ref = _packed_by_indexing(model, ids, mask, torch.zeros_like(ids), torch.arange(ids.shape[1]).unsqueeze(0))
    assert torch.equal(out, ref)


def test_forward_padded_through_the_c_abi(golden):
    """clip_forward_padded as a binder calls it: the queued form (error_flags NULL, then
    clip_poll_errors), the token count it hands back, the capacity errors."""
    import ctypes as C
    from clip_amd import _lib
    g = golden("tiny_ctx")
    model = build_model(g)
    lib, h = _lib.load(), model._ensure_engine()
    ids, mask = O.pad_batch(g["rows"][:5])
    tt = O.token_type_ids_from_seps(ids, 102, g["n_ctx"])
    B, S = ids.shape
    d_ids, d_mask, d_tt = ids.cuda(), mask.cuda(), tt.cuda()
    ref = model(input_ids=d_ids, attention_mask=d_mask, token_type_ids=d_tt)[0]
    logits = torch.full((B, model.num_labels), float("nan"), device="cuda")
    T = C.c_int32(-1)
    _lib.check(lib.clip_forward_padded(h, _lib.ptr(d_ids), _lib.ptr(d_mask), _lib.ptr(d_tt), None, 0, B, S, 102, g["n_ctx"], 0,
                                       _lib.ptr(logits), C.byref(T), None, _lib.stream_ptr()))
    _lib.poll_errors(h)
    assert T.value == int(mask.sum()) and torch.equal(logits, ref)
    # a wrong [SEP] count in the queued form: bit 0 at the poll, NaN logits whether or not the caller polls
    bad = d_ids.clone()
    bad[2][bad[2] == 102] = 7
    _lib.check(lib.clip_forward_padded(h, _lib.ptr(bad), _lib.ptr(d_mask), _lib.ptr(d_tt), None, 0, B, S, 102, g["n_ctx"], 0,
                                       _lib.ptr(logits), None, None, _lib.stream_ptr()))
    flags = C.c_int32(0)
    _lib.check(lib.clip_poll_errors(h, C.byref(flags)))
    assert flags.value & _lib.ERR_SEP_COUNT and torch.isnan(logits).all()
    # more rows / tokens than the engine's workspace
    big = torch.ones((model.max_seqs + 1, 4), dtype=torch.int64, device="cuda")
    out = torch.empty((model.max_seqs + 1, model.num_labels), device="cuda")
    assert lib.clip_forward_padded(h, _lib.ptr(big), None, None, None, 0, model.max_seqs + 1, 4, 102, g["n_ctx"], 0,
                                   _lib.ptr(out), None, None, _lib.stream_ptr()) == _lib.CLIP_ERR_CAPACITY
    wide = torch.ones((model.max_seqs, 512), dtype=torch.int64, device="cuda")      # 256 x 512 tokens > max_tokens
    assert model.max_seqs * 512 > model.max_tokens
    rc = lib.clip_forward_padded(h, _lib.ptr(wide), None, None, None, 0, model.max_seqs, 512, 102, g["n_ctx"], 0,
                                 _lib.ptr(out), C.byref(T), None, _lib.stream_ptr())
    assert rc == _lib.CLIP_ERR_CAPACITY and T.value == model.max_seqs * 512
    _lib.check(lib.clip_poll_errors(h, C.byref(flags)))       # (the length kernel may have flagged rows: clear)
    assert torch.equal(model(input_ids=d_ids, attention_mask=d_mask, token_type_ids=d_tt)[0], ref)


def test_no_cpu_path(golden):
    g = golden("tiny_ctx")
    model = args_to_model(2)(g["cfg"])
    model.load_state_dict(procedural_state_dict(g["cfg"], g["seed"]))
    model.set_sep_token_id(102); model.set_n_context_sentences(2)
    ids = torch.tensor([[101, 102, 102, 5, 102, 102, 102]])
    with pytest.raises(RuntimeError):
        model(input_ids=ids)


@pytest.mark.parametrize("name", ["base_ctx", "base_sent", "tiny_ctx"])
def test_last_layer_trim_equals_full_layers(golden, name):
    """Computing only the pooled row past the last K/V projection (bert_model.py:52-58 consumes one
    row per sentence) gives the same logits as pushing every token through every layer."""
    from clip_amd import _lib
    g = golden(name)
    model = build_model(g)
    trimmed, _ = run_padded(model, g)
    _lib.check(_lib.load().clip_set_trim_last_layer(model._ensure_engine(), 0))
    full, _ = run_padded(model, g)
    # same arithmetic except fp32 (trimmed) vs bf16 (full) softmax weights in the last attention:
    # one bf16 rounding of P out of ~120 roundings on the path -> well inside the 2e-2 budget
    assert np.abs(trimmed - full).max() < 6e-3
    assert np.abs(full - g["ref_logits"]).max() < BF16_LOGIT_TOL
    assert np.abs(trimmed - g["ref_logits"]).max() < BF16_LOGIT_TOL


FP32_LOGIT_TOL = 1e-4   # BASELINE.json north_star: 1e-4 abs in the fp32 check mode


@pytest.mark.parametrize("name", ["tiny_ctx", "tiny_ctx_n1", "tiny_sent", "base_ctx", "base_sent"])
def test_fp32_check_mode_matches_reference(golden, name):
    """The same kernel templates on fp32 operands (clip_set_precision): logits within 1e-4 of the
    reference's fp32 torch forward, and every thresholded label equal unless the reference
    probability sits within 1e-5 of its threshold."""
    g = golden(name)
    model = build_model(g)
    model.set_precision("fp32")
    logits, losses = run_padded(model, g)
    err = np.abs(logits - g["ref_logits"]).max()
    print(f"{name} fp32: max |logit - ref| = {err:.3e}")
    assert err < FP32_LOGIT_TOL
    assert np.abs(losses - g["ref_batch_loss"]).max() < 1e-5
    p = (1 / (1 + np.exp(-logits.astype(np.float64)))).astype(np.float32)
    for thr in (DNOTE, TTP):
        clear = np.abs(g["ref_probs"] - thr) > 1e-5
        assert (O.label_decisions(p, thr) == O.label_decisions(g["ref_probs"], thr))[clear].all()
    # and the bf16 setting of the same model object stays inside its own budget of the fp32 one
    model.set_precision("bf16")
    bf16_logits, _ = run_padded(model, g)
    assert np.abs(bf16_logits - logits).max() < BF16_LOGIT_TOL


def test_precision_is_fixed_at_finalize(golden):
    from clip_amd import _lib
    g = golden("tiny_ctx")
    model = build_model(g)
    h = model._ensure_engine()
    assert _lib.load().clip_set_precision(h, _lib.PRECISION_FP32) == _lib.CLIP_ERR_STATE
    with pytest.raises(ValueError):
        model.set_precision("fp16")


@pytest.mark.parametrize("name", ["base_ctx", "base_sent"])
def test_fused_layernorm_equals_unfused(golden, name):
    """EPI_BIAS_LNOUT (LayerNorm fused into the N = hidden GEMMs) against the GEMM -> LayerNorm-kernel
    flow: same logits to within bf16 rounding, both inside the budget against the reference."""
    import ctypes as C
    from clip_amd import _lib
    g = golden(name)
    model = build_model(g)
    h = model._ensure_engine()
    fused, _ = run_padded(model, g)
    flags = C.c_int32(-1)
    _lib.check(_lib.load().clip_poll_errors(h, C.byref(flags)))
    assert flags.value == 0     # no statistics exchange timed out
    _lib.check(_lib.load().clip_set_fused_layernorm(h, 0))
    unfused, _ = run_padded(model, g)
    assert np.abs(fused - g["ref_logits"]).max() < BF16_LOGIT_TOL
    assert np.abs(unfused - g["ref_logits"]).max() < BF16_LOGIT_TOL
    assert np.abs(fused - unfused).max() < 1e-2
    # deterministic: the exchange order is fixed, so a second run gives the same bits
    _lib.check(_lib.load().clip_set_fused_layernorm(h, 1))
    again, _ = run_padded(model, g)
    assert np.array_equal(fused, again)


def test_inconsistent_row_table_voids_the_call():
    """The packed ABI trusts the caller's cu_seqlens / max_seqlen for its launch geometry: a row longer
    than max_seqlen sets bit 2 of the engine's error word, the logits come back NaN, and the host
    mirrors raise (no counterpart in the reference: padded tensors cannot be inconsistent)."""
    import ctypes as C
    from clip_amd import _lib
    from conftest import load_golden
    g = load_golden("tiny_ctx")
    model = build_model(g)
    h = model._ensure_engine()
    lib = _lib.load()
    rows = g["rows"][:4]
    lens = np.array([len(r) for r in rows], np.int32)
    ids = torch.from_numpy(np.concatenate(rows).astype(np.int32)).cuda()
    cu = torch.from_numpy(np.concatenate(([0], np.cumsum(lens))).astype(np.int32)).cuda()
    pos = torch.cat([torch.arange(int(l), dtype=torch.int32) for l in lens]).cuda()
    focus = model.focus_index(ids, cu)
    logits = torch.zeros((4, model.num_labels), device="cuda")
    _lib.check(lib.clip_forward_packed(h, _lib.ptr(ids), None, _lib.ptr(pos), _lib.ptr(cu), _lib.ptr(focus), 4,
                                       int(lens.sum()), int(lens.max()) - 1, 0, _lib.ptr(logits), _lib.stream_ptr()))
    torch.cuda.synchronize()
    assert torch.isnan(logits).all()
    with pytest.raises(ValueError, match="row table"):
        _lib.poll_errors(h)
    # the poll cleared the word: the same call with the true maximum is good again
    _lib.check(lib.clip_forward_packed(h, _lib.ptr(ids), None, _lib.ptr(pos), _lib.ptr(cu), _lib.ptr(focus), 4,
                                       int(lens.sum()), int(lens.max()), 0, _lib.ptr(logits), _lib.stream_ptr()))
    _lib.poll_errors(h)
    assert torch.isfinite(logits).all()


def test_fused_layernorm_timeout_is_loud(monkeypatch):
    """A fused-LayerNorm tile whose partner never publishes (forced by the diagnostic switch
    CLIP_AMD_TEST_DROP_PUBLISH) must not return scores: bounded wait, sticky abort for the later tiles,
    NaN logits, and the nn.Module forward, forward_packed and BulkScorer.score all raise."""
    import time
    from clip_amd import scoring, synth
    from clip_amd.bert_model import BertSequenceMultilabelClassificationContext
    from clip_amd.config import BertConfig
    from clip_amd.weights import procedural_state_dict
    monkeypatch.setenv("CLIP_AMD_TEST_DROP_PUBLISH", "1")
    monkeypatch.setenv("CLIP_AMD_FUSED_LN", "1")
    cfg = BertConfig(vocab_size=2000, hidden_size=512, num_hidden_layers=3, num_attention_heads=8, intermediate_size=1024)
    model = BertSequenceMultilabelClassificationContext(cfg, max_tokens=16384, max_seqs=512)
    model.load_state_dict(procedural_state_dict(cfg, 3))
    model.set_precision("bf16")      # the switch acts on the fused flow of a fixed setting (auto would meet it while calibrating)
    model.to("cuda").eval()
    model.set_sep_token_id(102); model.set_n_context_sentences(2); model.set_task("multilabel")
    rows = synth.context_rows(300, 2, cfg.vocab_size, max_len=96, seed=11)       # several row panels per GEMM
    t0 = time.perf_counter()
    with pytest.raises(RuntimeError, match="fused-LayerNorm exchange timed out"):
        scoring.BulkScorer(model).score(rows)
    assert time.perf_counter() - t0 < 60          # the first tiles wait out their bound, the rest skip
    ids, mask = O.pad_batch(rows[:8])
    tt = O.token_type_ids_from_seps(ids, 102, 2)
    with pytest.raises(RuntimeError, match="timed out"):
        model(input_ids=ids.cuda(), attention_mask=mask.cuda(), token_type_ids=tt.cuda())
    model._drop_engine()


def test_fused_layernorm_epoch_wrap_is_handled(monkeypatch):
    """The fused-LayerNorm exchange recognises a partner's statistics by the launch's epoch, a 32-bit counter per
    engine.  An engine whose counter starts a few launches below 2^32 (diagnostic switch CLIP_AMD_TEST_EPOCH_START)
    must score exactly like a fresh one: at the wrap the flags are cleared and the epochs start over."""
    from clip_amd import scoring, synth
    from clip_amd.bert_model import BertSequenceMultilabelClassificationContext
    from clip_amd.config import BertConfig
    from clip_amd.weights import procedural_state_dict
    cfg = BertConfig(vocab_size=2000, hidden_size=512, num_hidden_layers=3, num_attention_heads=8, intermediate_size=1024)
    rows = synth.context_rows(300, 2, cfg.vocab_size, max_len=96, seed=12)
    outs = []
    for start in (None, str(0xffffffff - 7)):      # 6 fused GEMMs per step, a few steps: the wrap falls inside
        if start is None:
            monkeypatch.delenv("CLIP_AMD_TEST_EPOCH_START", raising=False)
        else:
            monkeypatch.setenv("CLIP_AMD_TEST_EPOCH_START", start)
        model = BertSequenceMultilabelClassificationContext(cfg, max_tokens=8192, max_seqs=512)
        model.load_state_dict(procedural_state_dict(cfg, 3))
        model.set_precision("bf16")
        model.to("cuda").eval()
        model.set_sep_token_id(102); model.set_n_context_sentences(2); model.set_task("multilabel")
        res = scoring.BulkScorer(model, keep_probs=True, tokens_per_step=4096).score(rows)
        res2 = scoring.BulkScorer(model, keep_probs=True, tokens_per_step=4096).score(rows)
        assert np.array_equal(res.probs, res2.probs)
        outs.append(res.probs.copy())
        model._drop_engine()
    assert np.isfinite(outs[0]).all() and np.array_equal(outs[0], outs[1])
