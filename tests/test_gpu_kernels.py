"""Per-kernel parity on a real MI355X, every call through the C ABI (include/clip_amd.h).
Floating-point kernels are compared with a plain PyTorch fp32 reference of the same op computed
from the SAME bf16-rounded inputs; tolerances are written at each assert."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from clip_amd import _lib  # noqa: E402


@pytest.fixture(scope="module")
def lib():
    assert torch.cuda.is_available(), "gpu tests need an MI355X"
    return _lib.load()


def _sync():
    torch.cuda.synchronize()


def _gemm(lib, A, W, bias, res, epi, out_f32, tile=0):
    M, K = A.shape
    N = W.shape[0]
    out = torch.full((M, N), float("nan"), device="cuda", dtype=torch.float32 if out_f32 else torch.bfloat16)
    _lib.check(lib.clip_gemm_bf16(_lib.ptr(A), _lib.ptr(W), _lib.ptr(bias), _lib.ptr(res), _lib.ptr(out),
                                  M, N, K, A.stride(0), W.stride(0), out.stride(0),
                                  0 if res is None else res.stride(0), epi, int(out_f32), tile, _lib.stream_ptr()))
    _sync()
    return out


@pytest.mark.parametrize("M,N,K,epi,out_f32", [
    (128, 128, 64, 0, True),          # one tile, one K step
    (300, 384, 128, 0, False),        # M and N tails
    (1024, 768, 768, 2, True),        # attention-output projection + residual
    (515, 2304, 768, 0, False),       # fused QKV
    (257, 3072, 768, 1, False),       # FFN up + erf-GELU
    (640, 768, 3072, 2, True),        # FFN down + residual
    (48, 768, 768, 3, False),         # pooler tanh, M < tile
    (1, 128, 128, 0, True),
])
@pytest.mark.parametrize("tile", [128, 256])
def test_gemm_matches_fp32_reference(lib, M, N, K, epi, out_f32, tile):
    g = torch.Generator(device="cpu").manual_seed(M * 7 + N * 3 + K + epi)
    A = (torch.randn(M, K, generator=g) * 1.0).to(torch.bfloat16).cuda()
    W = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16).cuda()
    # asymmetric structure so a transposed write cannot pass
    W[:, 0] += torch.arange(N, device="cuda").to(torch.bfloat16) * 0.01
    bias = torch.randn(N, generator=g).cuda()
    res = torch.randn(M, N, generator=g).to(torch.bfloat16).cuda() if epi == 2 else None
    out = _gemm(lib, A, W, bias, res, epi, out_f32, tile).float()
    ref = A.float() @ W.float().T + bias
    if epi == 1:
        ref = torch.nn.functional.gelu(ref)
    elif epi == 2:
        ref = ref + res.float()
    elif epi == 3:
        ref = torch.tanh(ref)
    # fp32 accumulation in a different order: 1e-3 abs on O(1..30) values for f32 out;
    # bf16 out adds one rounding: 2^-8 relative
    tol = 2e-3 + (0 if out_f32 else 4e-3 * ref.abs().max().item())
    assert torch.isfinite(out).all()
    assert (out - ref).abs().max().item() < tol


@pytest.mark.parametrize("M", [1, 8, 16, 23, 32])
@pytest.mark.parametrize("N,K,epi,out_f32", [
    (768, 768, 0, False),       # q projection of the focus rows
    (3072, 768, 1, False),      # FFN up + erf-GELU on the rows
    (768, 3072, 2, True),       # + bf16 residual, fp32 out
    (768, 768, 2, False),       # + bf16 residual, 16-bit out
    (768, 768, 3, False),       # pooler tanh
    (768, 768, 1, True),        # exact-erf GELU, fp32 out (LM-head transform)
    (256, 128, 0, True),
])
def test_rows16_gemm_same_bits_as_tile_kernel(lib, M, N, K, epi, out_f32):
    """Calls of at most 32 rows (the trimmed last layer of one batch of the reference's loop) run on 16 x 16 tiles
    (gemm_rows16_kernel): the result must be the SAME BITS as the 128 x 128 tile kernel's, which takes the same rows
    when they come in a bigger batch, and within the fp32-reference tolerance."""
    g = torch.Generator(device="cpu").manual_seed(M * 11 + N + K + epi)
    A = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
    W = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16).cuda()
    W[:, 0] += torch.arange(N, device="cuda").to(torch.bfloat16) * 0.01
    bias = torch.randn(N, generator=g).cuda()
    res = torch.randn(M, N, generator=g).to(torch.bfloat16).cuda() if epi == 2 else None
    small = _gemm(lib, A, W, bias, res, epi, out_f32, 0)          # auto: M <= 32 -> the 16-row kernel
    tile = _gemm(lib, A, W, bias, res, epi, out_f32, 128)
    assert torch.isfinite(small.float()).all()
    assert torch.equal(small, tile)
    ref = A.float() @ W.float().T + bias
    ref = torch.nn.functional.gelu(ref) if epi == 1 else ref + res.float() if epi == 2 else torch.tanh(ref) if epi == 3 else ref
    # 16-bit out: one rounding (2^-8 relative), two when the 16-bit residual is added to the rounded product
    assert (small.float() - ref).abs().max().item() < 2e-3 + (0 if out_f32 else (8e-3 if epi == 2 else 4e-3) * ref.abs().max().item())


@pytest.mark.parametrize("M", [8, 30])
def test_rows16_gemm_layernorm_residual_same_bits(lib, M):
    """the indexed LayerNorm-residual epilogue (last FFN-down of the trimmed layer) on the 16-row kernel vs the tile kernel"""
    N, K, R = 768, 3072, 500
    g = torch.Generator(device="cpu").manual_seed(M)
    A = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
    W = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16).cuda()
    bias = torch.randn(N, generator=g).cuda()
    pre = (torch.randn(R, N, generator=g) * 1.5 + 0.2).cuda()
    gamma = (1 + 0.1 * torch.randn(N, generator=g)).cuda()
    beta = (0.1 * torch.randn(N, generator=g)).cuda()
    idx = torch.randint(0, R, (M,), generator=g).to(torch.int32).cuda()
    xb = torch.empty(R, N, dtype=torch.bfloat16, device="cuda")
    stats = torch.empty(R, 2, device="cuda")
    _lib.check(lib.clip_layernorm_stats(_lib.ptr(pre), _lib.ptr(gamma), _lib.ptr(beta), 1e-12, _lib.ptr(xb),
                                        _lib.ptr(stats), R, N, _lib.stream_ptr()))
    outs = []
    for t in (0, 128):
        out = torch.full((M, N), float("nan"), device="cuda")
        _lib.check(lib.clip_gemm_bf16_lnres(_lib.ptr(A), _lib.ptr(W), _lib.ptr(bias), _lib.ptr(pre), _lib.ptr(stats),
                                            _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(idx), _lib.ptr(out), M, N, K,
                                            K, K, N, N, t, _lib.stream_ptr()))
        _sync()
        outs.append(out)
    assert torch.isfinite(outs[0]).all() and torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("M,N,K,epi,out_f32", [
    (256, 256, 128, 0, True),         # one tile, two K steps
    (2048, 768, 768, 2, True),        # fewer tiles than CUs: some workgroups idle
    (66 * 256, 2304, 768, 0, False),  # 594 tiles over 256 workgroups: 2-3 tiles each, uneven
    (40 * 256, 3072, 768, 1, False),
    (72 * 256, 768, 3072, 2, True),
    (36 * 256, 768, 768, 2, False),
    (64 * 256, 3072, 768, 1, False),  # tiles_m % 8 == 0: XCD-owned panels, column groups of 4
    (16 * 256, 2304, 768, 0, False),  # column groups of 3, fewer items than workgroup slots
    (8 * 256, 768, 3072, 2, True),
])
def test_persistent_gemm_matches_fp32_reference(lib, M, N, K, epi, out_f32, kind=2):   # 2: ping-pong persistent (gemm8.hip)
    g = torch.Generator(device="cpu").manual_seed(M + N * 3 + K + epi)
    A = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
    W = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16).cuda()
    W[:, 0] += torch.arange(N, device="cuda").to(torch.bfloat16) * 0.01
    bias = torch.randn(N, generator=g).cuda()
    res = torch.randn(M, N, generator=g).to(torch.bfloat16).cuda() if epi == 2 else None
    out = _gemm(lib, A, W, bias, res, epi, out_f32, kind).float()
    ref = A.float() @ W.float().T + bias
    if epi == 1:
        ref = torch.nn.functional.gelu(ref)
    elif epi == 2:
        ref = ref + res.float()
    tol = 2e-3 + (0 if out_f32 else 8e-3 * ref.abs().max().item())
    assert torch.isfinite(out).all()
    assert (out - ref).abs().max().item() < tol
    # run-to-run determinism (a race in the DMA / barrier protocol shows up as flicker)
    for _ in range(3):
        out2 = _gemm(lib, A, W, bias, res, epi, out_f32, kind).float()
        assert torch.equal(out, out2)
    out3 = _gemm(lib, A, W, bias, res, epi, out_f32, 256).float()
    # identical to the one-tile-per-workgroup kernel: same MFMA order per output element, so a row's
    # result does not depend on which kernel (= which batch size) computed it
    assert torch.equal(out, out3)


def test_gemm_rejects_bad_shapes(lib):
    A = torch.zeros(16, 100, dtype=torch.bfloat16, device="cuda")
    W = torch.zeros(16, 100, dtype=torch.bfloat16, device="cuda")
    b = torch.zeros(16, device="cuda")
    o = torch.zeros(16, 16, device="cuda")
    rc = lib.clip_gemm_bf16(_lib.ptr(A), _lib.ptr(W), _lib.ptr(b), None, _lib.ptr(o), 16, 16, 100, 100, 100, 16, 0, 0, 1, 0, None)
    assert rc == _lib.CLIP_ERR_ARG and b"multiple of 64" in lib.clip_last_error()


def _attn_ref(qkv, lens, heads):
    T, H3 = qkv.shape
    H = H3 // 3
    out = torch.zeros(T, H, dtype=torch.float32, device=qkv.device)
    off = 0
    x = qkv.float()
    for L in lens:
        q, k, v = (x[off:off + L, i * H:(i + 1) * H].view(L, heads, 64).transpose(0, 1) for i in range(3))
        p = torch.softmax(q @ k.transpose(1, 2), dim=-1)          # q is pre-scaled
        out[off:off + L] = (p @ v).transpose(0, 1).reshape(L, H)
        off += L
    return out


@pytest.mark.parametrize("lens,heads", [
    ([1, 5, 31, 32, 33, 64, 65, 127, 128, 129, 200, 256, 300, 512], 2),
    ([256] * 6, 12),
    ([7, 19, 23], 12),
])
def test_attention_matches_fp32_reference(lib, lens, heads):
    H = heads * 64
    T = sum(lens)
    g = torch.Generator(device="cpu").manual_seed(T + heads)
    qkv = torch.randn(T, 3 * H, generator=g)
    qkv[:, :H] *= 0.4                 # scores with a spread of a few units
    qkv = qkv.to(torch.bfloat16).cuda()
    cu = torch.tensor(np.concatenate(([0], np.cumsum(lens))), dtype=torch.int32, device="cuda")
    ctx = torch.full((T, H), float("nan"), dtype=torch.bfloat16, device="cuda")
    _lib.check(lib.clip_attention_varlen(_lib.ptr(qkv), _lib.ptr(cu), _lib.ptr(ctx), len(lens), max(lens),
                                         heads, H, _lib.stream_ptr()))
    _sync()
    ref = _attn_ref(qkv, lens, heads)
    assert torch.isfinite(ctx.float()).all()
    # P is rounded to bf16 before P.V and the output is bf16: 2e-2 abs on |out| <~ 3
    err = (ctx.float() - ref).abs().max().item()
    assert err < 2e-2, err


def test_attention_online_softmax_rescale(lib):
    """Force the running max to jump at a later key tile (guide rule 26)."""
    heads, H, L = 1, 64, 200
    qkv = torch.zeros(L, 3 * H)
    g = torch.Generator(device="cpu").manual_seed(3)
    qkv[:, :H] = torch.randn(L, H, generator=g) * 0.3
    qkv[:, H:2 * H] = torch.randn(L, H, generator=g) * 0.3
    qkv[:, 2 * H:] = torch.randn(L, H, generator=g)
    qkv[150, H:2 * H] = qkv[10, :H] * 40.0        # key 150 (tile 2) spikes against query 10
    qkv = qkv.to(torch.bfloat16).cuda()
    cu = torch.tensor([0, L], dtype=torch.int32, device="cuda")
    ctx = torch.empty((L, H), dtype=torch.bfloat16, device="cuda")
    _lib.check(lib.clip_attention_varlen(_lib.ptr(qkv), _lib.ptr(cu), _lib.ptr(ctx), 1, L, heads, H, _lib.stream_ptr()))
    _sync()
    ref = _attn_ref(qkv, [L], heads)
    assert (ctx.float() - ref).abs().max().item() < 2e-2


@pytest.mark.parametrize("T,H", [(1, 768), (1000, 768), (37, 128), (64, 1024)])
def test_layernorm(lib, T, H):
    g = torch.Generator(device="cpu").manual_seed(T + H)
    x = (torch.randn(T, H, generator=g) * 3 + 1).cuda()
    gamma, beta = torch.randn(H, generator=g).cuda(), torch.randn(H, generator=g).cuda()
    out = torch.empty(T, H, dtype=torch.bfloat16, device="cuda")
    _lib.check(lib.clip_layernorm(_lib.ptr(x), _lib.ptr(gamma), _lib.ptr(beta), 1e-12, _lib.ptr(out), T, H, _lib.stream_ptr()))
    _sync()
    ref = torch.nn.functional.layer_norm(x, (H,), gamma, beta, 1e-12)
    # output rounded to bf16: 2^-8 relative (+1e-5 abs)
    assert ((out.float() - ref).abs() <= ref.abs() * 2 ** -8 + 1e-5).all()


def test_embed_layernorm(lib):
    T, H, V, P, TY = 777, 768, 5000, 512, 2
    g = torch.Generator(device="cpu").manual_seed(5)
    word, pos, typ = (torch.randn(n, H, generator=g).cuda() for n in (V, P, TY))
    gamma, beta = torch.randn(H, generator=g).cuda(), torch.randn(H, generator=g).cuda()
    ids = torch.randint(0, V, (T,), generator=g, dtype=torch.int32).cuda()
    tt = torch.randint(0, TY, (T,), generator=g, dtype=torch.int32).cuda()
    po = torch.randint(0, P, (T,), generator=g, dtype=torch.int32).cuda()
    out = torch.empty(T, H, dtype=torch.bfloat16, device="cuda")
    _lib.check(lib.clip_embed_layernorm(_lib.ptr(ids), _lib.ptr(tt), _lib.ptr(po), _lib.ptr(word), _lib.ptr(pos),
                                        _lib.ptr(typ), _lib.ptr(gamma), _lib.ptr(beta), 1e-12, _lib.ptr(out),
                                        T, H, V, P, TY, _lib.stream_ptr()))
    _sync()
    ref = torch.nn.functional.layer_norm(word[ids.long()] + typ[tt.long()] + pos[po.long()], (H,), gamma, beta, 1e-12)
    assert ((out.float() - ref).abs() <= ref.abs() * 2 ** -8 + 1e-5).all()


def test_focus_index(lib):
    rows = [[101, 5, 102, 102, 7, 8, 102, 9, 102, 102],          # focus sep = index 6
            [101] + [9] * 70 + [102, 102] + [4] * 100 + [102, 3, 102, 102],  # beyond one 64-token sweep
            [101, 102, 102, 5, 102, 102, 102]]
    ids = torch.tensor(sum(rows, []), dtype=torch.int32, device="cuda")
    lens = [len(r) for r in rows]
    cu = torch.tensor(np.concatenate(([0], np.cumsum(lens))), dtype=torch.int32, device="cuda")
    focus = torch.empty(3, dtype=torch.int32, device="cuda")
    err = torch.zeros(1, dtype=torch.int32, device="cuda")
    _lib.check(lib.clip_focus_index(_lib.ptr(ids), _lib.ptr(cu), 3, 102, 2, _lib.ptr(focus), _lib.ptr(err), None))
    _sync()
    exp = []
    for r, o in zip(rows, cu.tolist()):
        exp.append(o + [i for i, t in enumerate(r) if t == 102][2])
    assert focus.tolist() == exp and err.item() == 0
    # wrong [SEP] count is flagged (bert_model.py:51-55)
    _lib.check(lib.clip_focus_index(_lib.ptr(ids), _lib.ptr(cu), 3, 102, 1, _lib.ptr(focus), _lib.ptr(err), None))
    _sync()
    assert err.item() == 1
    # [CLS] row for the sentence-only head
    _lib.check(lib.clip_focus_index(_lib.ptr(ids), _lib.ptr(cu), 3, 0, -1, _lib.ptr(focus), _lib.ptr(err), None))
    _sync()
    assert focus.tolist() == cu.tolist()[:3]


def test_epilogue_bit_exact_decisions(lib):
    from oracle import bert_oracle as O
    g = torch.Generator(device="cpu").manual_seed(9)
    n = 10007
    logits = torch.randn(n, 7, generator=g) * 2
    t32 = np.float32(0.05)
    logits[0] = float(np.log(t32 / (1 - t32)))            # sits on the any-threshold
    logits[1] = -20.0
    thr = np.array([0.44, 0.31, 0.24, 0.17, 0.46, 0.35000000000000003, 0.43])
    d_logits, d_thr = logits.cuda(), torch.tensor(thr, dtype=torch.float64, device="cuda")
    probs = torch.empty(n, 7, device="cuda")
    mx = torch.empty(n, device="cuda")
    labels = torch.empty(n, 7, dtype=torch.uint8, device="cuda")
    keep = torch.empty(n, dtype=torch.uint8, device="cuda")
    _lib.check(lib.clip_epilogue(_lib.ptr(d_logits), _lib.ptr(d_thr), 0.05, n, 7, _lib.ptr(probs), _lib.ptr(mx),
                                 _lib.ptr(labels), _lib.ptr(keep), None))
    _sync()
    p_ref, keep_ref, score_ref = O.score_epilogue(logits, 0.05)
    p = probs.cpu().numpy()
    # sigmoid: within 2 ulp of torch's CPU fp32 sigmoid
    assert np.abs(p - p_ref).max() <= 2 * np.finfo(np.float32).eps
    # decisions are bit-exact functions of the device's own fp32 probabilities
    assert (labels.cpu().numpy().astype(bool) == O.label_decisions(p, thr)).all()
    assert (keep.cpu().numpy().astype(bool) == np.any(p > np.float32(0.05), axis=1)).all()
    assert (mx.cpu().numpy() == p.max(axis=1)).all()
    # and agree with the oracle's end-to-end decisions except where a probability is within 2 ulp of a threshold
    agree = (labels.cpu().numpy().astype(bool) == O.label_decisions(p_ref, thr)).mean()
    assert agree > 0.9999


def test_bce_loss(lib):
    g = torch.Generator(device="cpu").manual_seed(2)
    x = (torch.randn(8, 7, generator=g) * 3).cuda()
    y = (torch.rand(8, 7, generator=g) < 0.3).float().cuda()
    loss = torch.empty(1, device="cuda")
    _lib.check(lib.clip_bce_loss(_lib.ptr(x), _lib.ptr(y), 56, _lib.ptr(loss), None))
    _sync()
    ref = torch.nn.functional.binary_cross_entropy_with_logits(x, y)
    assert abs(loss.item() - ref.item()) < 1e-6


def test_gelu_epilogue_accuracy(lib):
    """The bf16-output GELU epilogue uses a polynomial erf: check it against exact erf-GELU over the
    whole input range, through a GEMM with an identity weight (so out = gelu(A + bias))."""
    M, K = 4096, 128
    A = torch.zeros(M, K)
    xs = torch.linspace(-12, 12, M * K // 2)
    A.view(-1)[::2] = xs                      # every other column, rest zero -> gelu(0) = 0 as well
    A = A.to(torch.bfloat16).cuda()
    W = torch.eye(K).to(torch.bfloat16).cuda()
    bias = torch.zeros(K).cuda()
    for tile in (128, 256):
        out = _gemm(lib, A, W, bias, None, 1, False, tile).float()
        ref = torch.nn.functional.gelu(A.float())
        # polynomial error <= 1.46e-4 abs, then one bf16 rounding (2^-9 relative)
        assert ((out - ref).abs() <= ref.abs() * 2 ** -8 + 2e-4).all()


@pytest.mark.parametrize("M,N,K,epi", [(128, 128, 32, 0), (300, 384, 160, 2), (515, 768, 768, 1),
                                       (64, 3072, 768, 0), (257, 768, 3072, 2), (48, 768, 768, 3)])
def test_gemm_fp32_check_mode(lib, M, N, K, epi):
    """fp32 operands through the same kernel template (exact-fp32 MFMA): 1e-4-class agreement with
    an fp64 reference — the tolerance the fp32 check mode exists for."""
    g = torch.Generator(device="cpu").manual_seed(M + N + K + epi)
    A = torch.randn(M, K, generator=g).cuda()
    W = (torch.randn(N, K, generator=g) * 0.05).cuda()
    W[:, 0] += torch.arange(N, device="cuda") * 0.01
    bias = torch.randn(N, generator=g).cuda()
    res = torch.randn(M, N, generator=g).cuda() if epi == 2 else None
    out = torch.full((M, N), float("nan"), device="cuda")
    _lib.check(lib.clip_gemm_f32(_lib.ptr(A), _lib.ptr(W), _lib.ptr(bias), _lib.ptr(res), _lib.ptr(out), M, N, K,
                                 K, K, N, N if res is not None else 0, epi, _lib.stream_ptr()))
    _sync()
    ref = A.double() @ W.double().T + bias.double()
    if epi == 1:
        ref = torch.nn.functional.gelu(ref)
    elif epi == 2:
        ref = ref + res.double()
    elif epi == 3:
        ref = torch.tanh(ref)
    err = (out.double() - ref).abs().max().item()
    assert err < 2e-5 * max(1.0, ref.abs().max().item()), err


def _ln_ref(pre, gamma, beta, eps=1e-12):
    mu = pre.mean(-1, keepdim=True)
    var = ((pre - mu) ** 2).mean(-1, keepdim=True)
    return (pre - mu) / torch.sqrt(var + eps) * gamma + beta, mu.squeeze(-1), (1 / torch.sqrt(var + eps)).squeeze(-1)


def test_layernorm_stats(lib):
    g = torch.Generator(device="cpu").manual_seed(5)
    T, H = 1000, 768
    x = (torch.randn(T, H, generator=g) * 2 + 0.3).cuda()
    gamma = (1 + 0.1 * torch.randn(H, generator=g)).cuda()
    beta = (0.1 * torch.randn(H, generator=g)).cuda()
    out = torch.empty(T, H, dtype=torch.bfloat16, device="cuda")
    stats = torch.empty(T, 2, device="cuda")
    _lib.check(lib.clip_layernorm_stats(_lib.ptr(x), _lib.ptr(gamma), _lib.ptr(beta), 1e-12, _lib.ptr(out),
                                        _lib.ptr(stats), T, H, _lib.stream_ptr()))
    _sync()
    ref, mu, rstd = _ln_ref(x.double(), gamma.double(), beta.double())
    assert (stats[:, 0].double() - mu).abs().max().item() < 1e-6
    assert ((stats[:, 1].double() - rstd) / rstd).abs().max().item() < 1e-6
    assert (out.double() - ref).abs().max().item() < 2 ** -8 * ref.abs().max().item()


@pytest.mark.parametrize("M,N,K,tile,indexed", [
    (64 * 256, 768, 768, 2, False),    # ping-pong persistent, XCD-owned panels
    (36 * 256, 768, 3072, 2, False),   # tiles_m % 8 != 0: plain tile order
    (300, 768, 768, 0, False),         # one-tile kernel, ragged M
    (77, 768, 3072, 0, True),          # residual rows picked by index (trimmed last layer)
])
def test_gemm_layernorm_residual_epilogue(lib, M, N, K, tile, indexed):
    """out = A W^T + bias + LN(pre): the residual is the fp32 LayerNorm output recomputed from the
    pre-LN rows, their (mean, rstd) and the affine — compared with an fp64 reference, and bit for bit
    between the ping-pong kernel and the one-tile kernel (batch-size invariance of a row's result)."""
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
    W = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16).cuda()
    bias = torch.randn(N, generator=g).cuda()
    R = M if not indexed else 1000
    pre = (torch.randn(R, N, generator=g) * 1.5 + 0.2).cuda()
    gamma = (1 + 0.1 * torch.randn(N, generator=g)).cuda()
    beta = (0.1 * torch.randn(N, generator=g)).cuda()
    idx = torch.randint(0, R, (M,), generator=g).to(torch.int32).cuda() if indexed else None
    xb = torch.empty(R, N, dtype=torch.bfloat16, device="cuda")
    stats = torch.empty(R, 2, device="cuda")
    _lib.check(lib.clip_layernorm_stats(_lib.ptr(pre), _lib.ptr(gamma), _lib.ptr(beta), 1e-12, _lib.ptr(xb),
                                        _lib.ptr(stats), R, N, _lib.stream_ptr()))

    def run(t):
        out = torch.full((M, N), float("nan"), device="cuda")
        _lib.check(lib.clip_gemm_bf16_lnres(_lib.ptr(A), _lib.ptr(W), _lib.ptr(bias), _lib.ptr(pre), _lib.ptr(stats),
                                            _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(idx), _lib.ptr(out), M, N, K,
                                            K, K, N, N, t, _lib.stream_ptr()))
        _sync()
        return out
    out = run(tile)
    lnref, _, _ = _ln_ref(pre.double(), gamma.double(), beta.double())
    if indexed:
        lnref = lnref[idx.long()]
    ref = A.double() @ W.double().T + bias.double() + lnref
    assert torch.isfinite(out).all()
    assert (out.double() - ref).abs().max().item() < 2e-3
    if tile == 2:
        assert torch.equal(out, run(256))
        assert torch.equal(out, run(2))


def _xres_tiled_offsets(M, N):
    """include/clip_amd.h, clip_gemm_bf16_lnout: where element (row, col) of the tiled fp16 copy lives"""
    row = torch.arange(M).view(-1, 1)
    col = torch.arange(N).view(1, -1)
    r, c = row & 255, col & 255
    tile = (row >> 8) * (N // 256) + (col >> 8)
    wave = ((r >> 6) & 1) * 4 + (c >> 6)
    blk = (r >> 7) * 4 + ((r >> 4) & 3)
    lane = ((c >> 2) & 3) * 16 + (r & 15)
    return ((tile * 8 + wave) * 8 + blk) * 1024 + ((c >> 5) & 1) * 512 + lane * 8 + ((c >> 4) & 1) * 4 + (c & 3)


@pytest.mark.parametrize("M,N,K", [(255 * 256, 768, 768), (64 * 256, 768, 768), (36 * 256, 768, 3072), (256, 768, 768),
                                   (5 * 256, 256, 256), (3 * 256, 1024, 512), (100 * 256, 512, 256)])
def test_gemm_fused_layernorm_epilogue(lib, M, N, K):
    """x, xres = LayerNorm(A W^T + bias + res16): the LayerNorm runs inside the GEMM, its row sums
    exchanged between the N/256 column-tile workgroups of each row panel.  Against an fp64
    reference; with the residual row-major (first layer) and tiled + in place (every later layer),
    which must give the same bits, also across calls (new epoch on the same flags / scratch)."""
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
    W = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16).cuda()
    bias = torch.randn(N, generator=g).cuda()
    res = (torch.randn(M, N, generator=g) * 1.5 + 0.2).to(torch.float16).cuda()
    gamma = (1 + 0.1 * torch.randn(N, generator=g)).cuda()
    beta = (0.1 * torch.randn(N, generator=g)).cuda()
    nt = (M // 256) * (N // 256)
    part = torch.empty(nt * 256 * 2, device="cuda")
    flags = torch.zeros(nt, dtype=torch.int32, device="cuda")
    err = torch.zeros(1, dtype=torch.int32, device="cuda")
    off = _xres_tiled_offsets(M, N).cuda()
    assert torch.equal(off.flatten().sort().values, torch.arange(M * N, device="cuda"))   # a permutation
    res_tiled = torch.empty(M * N, dtype=torch.float16, device="cuda")
    res_tiled[off.flatten()] = res.flatten()
    v = A.double() @ W.double().T + bias.double() + res.double()
    ref, _, _ = _ln_ref(v, gamma.double(), beta.double())
    outs = []
    for epoch, tiled in ((1, False), (2, False), (3, True)):
        x = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
        if tiled:
            xres = res_tiled.clone()   # in place: the tiled residual buffer receives the tiled output
            rin = xres
        else:
            xres = torch.full((M * N,), float("nan"), dtype=torch.float16, device="cuda")
            rin = res
        _lib.check(lib.clip_gemm_bf16_lnout(_lib.ptr(A), _lib.ptr(W), _lib.ptr(bias), _lib.ptr(rin), int(tiled),
                                            _lib.ptr(gamma), _lib.ptr(beta), 1e-12, _lib.ptr(x), _lib.ptr(xres),
                                            _lib.ptr(part), _lib.ptr(flags), epoch, _lib.ptr(err), M, N, K, K, K, N, N,
                                            _lib.stream_ptr()))
        _sync()
        assert int(err.item()) == 0
        xr = xres[off]                 # back to [M, N]
        assert torch.isfinite(x.float()).all() and torch.isfinite(xr.float()).all()
        assert (xr.double() - ref).abs().max().item() < 4e-3 + 2 ** -10 * ref.abs().max().item()
        assert (x.double() - ref).abs().max().item() < 4e-3 + 2 ** -8 * ref.abs().max().item()
        outs.append((x.clone(), xres.clone()))
    for o in outs[1:]:
        assert torch.equal(outs[0][0], o[0]) and torch.equal(outs[0][1], o[1])
    # The same launch many times over: a wave that runs ahead into the row sums must not disturb a wave still
    # reading its residual rows through the LDS scratch (row-major residual; seen once as one wrong row in a few
    # launches before each wave kept its partials in its own scratch).
    for epoch in range(4, 4 + (24 if M <= 100 * 256 else 6)):
        x = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
        xres = torch.empty(M * N, dtype=torch.float16, device="cuda")
        _lib.check(lib.clip_gemm_bf16_lnout(_lib.ptr(A), _lib.ptr(W), _lib.ptr(bias), _lib.ptr(res), 0,
                                            _lib.ptr(gamma), _lib.ptr(beta), 1e-12, _lib.ptr(x), _lib.ptr(xres),
                                            _lib.ptr(part), _lib.ptr(flags), epoch, _lib.ptr(err), M, N, K, K, K, N, N,
                                            _lib.stream_ptr()))
        _sync()
        assert torch.equal(outs[0][0], x) and torch.equal(outs[0][1], xres), f"launch {epoch} differs"


@pytest.mark.parametrize("M,N,K,epi,out_f32,tile", [
    (300, 768, 768, 0, False, 128), (48, 768, 768, 3, False, 0), (66 * 256, 2304, 768, 0, False, 2),
    (40 * 256, 3072, 768, 1, False, 2), (8 * 256, 768, 3072, 2, True, 2), (520, 3072, 768, 1, False, 256)])
def test_gemm_f16_operand_set(lib, M, N, K, epi, out_f32, tile):
    """The k_f16 kernel set (CLIP_PRECISION_F16): the same GEMM kernels compiled for fp16 operands.
    Against an fp32 reference; the output rounding is fp16's (2^-11 relative), 8x finer than bf16's."""
    g = torch.Generator(device="cpu").manual_seed(M + N + K + epi)
    A = torch.randn(M, K, generator=g).to(torch.float16).cuda()
    W = (torch.randn(N, K, generator=g) * 0.05).to(torch.float16).cuda()
    bias = torch.randn(N, generator=g).cuda()
    res = torch.randn(M, N, generator=g).to(torch.float16).cuda() if epi == 2 else None
    out = torch.empty((M, N), dtype=torch.float32 if out_f32 else torch.float16, device="cuda")
    _lib.check(lib.clip_gemm_f16(_lib.ptr(A), _lib.ptr(W), _lib.ptr(bias), _lib.ptr(res), _lib.ptr(out), M, N, K, K, K, N,
                                 0 if res is None else N, epi, int(out_f32), tile, _lib.stream_ptr()))
    _sync()
    ref = A.float() @ W.float().T + bias
    if epi == 1:
        ref = torch.nn.functional.gelu(ref)
    elif epi == 2:
        ref = ref + res.float()
    elif epi == 3:
        ref = torch.tanh(ref)
    tol = 2e-3 + (0 if out_f32 else 1e-3 * ref.abs().max().item())
    if epi == 1 and not out_f32:
        tol += 1.5e-4                 # the degree-6 GELU polynomial of the 16-bit-output epilogue
    assert torch.isfinite(out.float()).all()
    assert (out.float() - ref).abs().max().item() < tol


@pytest.mark.parametrize("M,N,K", [(255 * 256, 768, 768), (36 * 256, 768, 3072), (256, 768, 768), (3 * 256, 1024, 512)])
def test_gemm_f16_fused_layernorm_in_place(lib, M, N, K):
    """The f16 setting's fused-LayerNorm GEMM: ONE fp16 buffer holds the residual on entry and
    LayerNorm(A W^T + bias + residual) on return (operand copy = residual copy).  Against an fp64 reference, same
    bits across calls (new epoch, same scratch), and the buffer's other content (a wider row stride) untouched."""
    g = torch.Generator(device="cpu").manual_seed(M + N + K + 1)
    A = torch.randn(M, K, generator=g).to(torch.float16).cuda()
    W = (torch.randn(N, K, generator=g) * 0.05).to(torch.float16).cuda()
    bias = torch.randn(N, generator=g).cuda()
    res = (torch.randn(M, N, generator=g) * 1.5 + 0.2).to(torch.float16).cuda()
    gamma = (1 + 0.1 * torch.randn(N, generator=g)).cuda()
    beta = (0.1 * torch.randn(N, generator=g)).cuda()
    nt = (M // 256) * (N // 256)
    part = torch.empty(nt * 256 * 2, device="cuda")
    flags = torch.zeros(nt, dtype=torch.int32, device="cuda")
    err = torch.zeros(1, dtype=torch.int32, device="cuda")
    v = A.double() @ W.double().T + bias.double() + res.double()
    ref, _, _ = _ln_ref(v, gamma.double(), beta.double())
    outs = []
    for epoch, ldx in ((1, N), (2, N), (3, N + 64)):
        buf = torch.full((M, ldx), 7.0, dtype=torch.float16, device="cuda")
        buf[:, :N] = res
        _lib.check(lib.clip_gemm_f16_lnout_inplace(_lib.ptr(A), _lib.ptr(W), _lib.ptr(bias), _lib.ptr(gamma), _lib.ptr(beta),
                                                   1e-12, _lib.ptr(buf), _lib.ptr(part), _lib.ptr(flags), epoch,
                                                   _lib.ptr(err), M, N, K, K, K, ldx, _lib.stream_ptr()))
        _sync()
        assert int(err.item()) == 0
        x = buf[:, :N]
        assert torch.isfinite(x.float()).all()
        assert (x.double() - ref).abs().max().item() < 4e-3 + 2 ** -10 * ref.abs().max().item()
        if ldx > N:
            assert (buf[:, N:] == 7.0).all()
        outs.append(x.clone())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    for epoch in range(10, 10 + (24 if M <= 100 * 256 else 6)):      # repeated launches: same bits every time
        buf = res.clone()
        _lib.check(lib.clip_gemm_f16_lnout_inplace(_lib.ptr(A), _lib.ptr(W), _lib.ptr(bias), _lib.ptr(gamma), _lib.ptr(beta),
                                                   1e-12, _lib.ptr(buf), _lib.ptr(part), _lib.ptr(flags), epoch,
                                                   _lib.ptr(err), M, N, K, K, K, N, _lib.stream_ptr()))
        _sync()
        assert torch.equal(outs[0], buf), f"launch {epoch} differs"
    if K == N:      # A must not alias the in-place buffer
        assert lib.clip_gemm_f16_lnout_inplace(_lib.ptr(res), _lib.ptr(W), _lib.ptr(bias), _lib.ptr(gamma),
                                               _lib.ptr(beta), 1e-12, _lib.ptr(res), _lib.ptr(part), _lib.ptr(flags), 9,
                                               _lib.ptr(err), M, N, K, K, K, N, _lib.stream_ptr()) != 0


def _xres32_tiled_index(M, N):
    """element (row, col) -> offset in the fp32 tiled residual (include/clip_amd.h, clip_gemm_f16_lnout_res32)"""
    r = torch.arange(M)[:, None].expand(M, N)
    c = torch.arange(N)[None, :].expand(M, N)
    tile = (r >> 8) * (N // 256) + (c >> 8)
    rr, cc = r & 255, c & 255
    wave = ((rr >> 6) & 1) * 4 + (cc >> 6)
    blk = (rr >> 7) * 4 + ((rr >> 4) & 3)
    lane = ((cc >> 2) & 3) * 16 + (rr & 15)
    return ((tile * 8 + wave) * 8 + blk) * 1024 + ((cc >> 5) & 1) * 512 + ((cc >> 4) & 1) * 256 + lane * 4 + (cc & 3)


@pytest.mark.parametrize("M,N,K", [(255 * 256, 768, 768), (36 * 256, 768, 3072), (256, 768, 768), (3 * 256, 1024, 512)])
def test_gemm_f16_fused_layernorm_fp32_residual(lib, M, N, K):
    """The F16R setting's fused-LayerNorm GEMM: fp16 operands, the residual read from and the LayerNorm output
    written back to an FP32 tiled buffer (in place), the fp16 operand copy beside it.  Against an fp64 reference:
    the fp32 copy to fp32 accuracy of the sum, the fp16 copy = that copy rounded once; same bits on every launch."""
    g = torch.Generator(device="cpu").manual_seed(M + N + K + 2)
    A = torch.randn(M, K, generator=g).to(torch.float16).cuda()
    W = (torch.randn(N, K, generator=g) * 0.05).to(torch.float16).cuda()
    bias = torch.randn(N, generator=g).cuda()
    res = torch.randn(M, N, generator=g) * 1.5 + 0.2                      # fp32 residual rows
    gamma = (1 + 0.1 * torch.randn(N, generator=g)).cuda()
    beta = (0.1 * torch.randn(N, generator=g)).cuda()
    idx = _xres32_tiled_index(M, N).reshape(-1)
    assert torch.equal(torch.sort(idx).values, torch.arange(M * N))       # the layout is a permutation
    tiled = torch.empty(M * N)
    tiled[idx] = res.reshape(-1)
    nt = (M // 256) * (N // 256)
    part = torch.empty(nt * 256 * 2, device="cuda")
    flags = torch.zeros(nt, dtype=torch.int32, device="cuda")
    err = torch.zeros(1, dtype=torch.int32, device="cuda")
    v = A.double() @ W.double().T + bias.double() + res.cuda().double()
    ref, _, _ = _ln_ref(v, gamma.double(), beta.double())
    first = None
    for epoch in range(1, 8):
        xres = tiled.clone().cuda()
        x = torch.full((M, N), float("nan"), dtype=torch.float16, device="cuda")
        _lib.check(lib.clip_gemm_f16_lnout_res32(_lib.ptr(A), _lib.ptr(W), _lib.ptr(bias), _lib.ptr(gamma), _lib.ptr(beta),
                                                 1e-12, _lib.ptr(x), _lib.ptr(xres), _lib.ptr(part), _lib.ptr(flags), epoch,
                                                 _lib.ptr(err), M, N, K, K, K, N, _lib.stream_ptr()))
        _sync()
        assert int(err.item()) == 0
        y32 = xres[idx.cuda()].reshape(M, N)
        if first is None:
            assert torch.isfinite(y32).all() and torch.isfinite(x.float()).all()
            # fp16 operands, fp32 accumulation over K: the fp32 copy carries no 16-bit rounding of its own
            assert (y32.double() - ref).abs().max().item() < 2e-4 * max(1.0, (K / 768) ** 0.5)
            assert torch.equal(x, y32.to(torch.float16))                   # the operand copy = the same values rounded once
            first = (x.clone(), xres.clone())
        else:
            assert torch.equal(first[0], x) and torch.equal(first[1], xres), f"launch {epoch} differs"


def test_attention_f16_operand_set(lib):
    heads, H = 12, 768
    lens = [1, 7, 64, 65, 128, 129, 200, 256, 300, 512]
    T = sum(lens)
    g = torch.Generator(device="cpu").manual_seed(9)
    qkv = (torch.randn(T, 3 * H, generator=g) * 0.7).to(torch.float16).cuda()
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device="cuda")
    ctx = torch.full((T, H), float("nan"), dtype=torch.float16, device="cuda")
    _lib.check(lib.clip_attention_varlen_f16(_lib.ptr(qkv), _lib.ptr(cu), _lib.ptr(ctx), len(lens), max(lens), heads, H,
                                             _lib.stream_ptr()))
    _sync()
    ref = _attn_ref(qkv, lens, heads)
    assert torch.isfinite(ctx.float()).all()
    assert (ctx.float() - ref).abs().max().item() < 3e-3      # bf16 set: 2e-2 on the same inputs
