"""clip_gemm_bf16_lnout_noalloc (include/clip_amd.h): the entry point that allocates nothing — the
fused-LayerNorm GEMM with its exchange buffers in a static device array inside the library.  Checked
against clip_gemm_bf16_lnout on the same operands: identical bits (bf16 x and tiled fp16 xres), on two
calls in a row, the second on a smaller input (the array keeps the first call's contents), then a third
larger one again; and the size limit is an error, not a fault."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from clip_amd import _lib  # noqa: E402


def _operands(M, N, K, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    A = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
    W = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16).cuda()
    bias = torch.randn(N, generator=g).cuda()
    res = (torch.randn(M, N, generator=g) * 1.5 + 0.2).to(torch.float16).cuda()
    gamma = (1 + 0.1 * torch.randn(N, generator=g)).cuda()
    beta = (0.1 * torch.randn(N, generator=g)).cuda()
    return A, W, bias, res, gamma, beta


def _existing(lib, ops, M, N, K, epoch):
    A, W, bias, res, gamma, beta = ops
    nt = (M // 256) * (N // 256)
    part = torch.empty(nt * 256 * 2, device="cuda")
    flags = torch.zeros(nt, dtype=torch.int32, device="cuda")
    err = torch.zeros(1, dtype=torch.int32, device="cuda")
    x = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
    xres = torch.full((M * N,), float("nan"), dtype=torch.float16, device="cuda")
    _lib.check(lib.clip_gemm_bf16_lnout(_lib.ptr(A), _lib.ptr(W), _lib.ptr(bias), _lib.ptr(res), 0, _lib.ptr(gamma),
                                        _lib.ptr(beta), 1e-12, _lib.ptr(x), _lib.ptr(xres), _lib.ptr(part),
                                        _lib.ptr(flags), epoch, _lib.ptr(err), M, N, K, K, K, N, N, _lib.stream_ptr()))
    torch.cuda.synchronize()
    assert int(err.item()) == 0
    return x, xres


def _noalloc(lib, ops, M, N, K):
    A, W, bias, res, gamma, beta = ops
    err = torch.zeros(1, dtype=torch.int32, device="cuda")
    x = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
    xres = torch.full((M * N,), float("nan"), dtype=torch.float16, device="cuda")
    torch.cuda.synchronize()
    before = torch.cuda.memory_allocated()
    rc = lib.clip_gemm_bf16_lnout_noalloc(_lib.ptr(A), _lib.ptr(W), _lib.ptr(bias), _lib.ptr(res), 0, _lib.ptr(gamma),
                                          _lib.ptr(beta), 1e-12, _lib.ptr(x), _lib.ptr(xres), _lib.ptr(err),
                                          M, N, K, K, K, N, N, _lib.stream_ptr())
    _lib.check(rc)
    torch.cuda.synchronize()
    assert torch.cuda.memory_allocated() == before
    assert int(err.item()) == 0
    return x, xres


def test_noalloc_entry_matches_existing_entry_on_consecutive_calls():
    lib = _lib.load()
    # (M, N, K): a full-chip call, then a SMALLER one on the same static scratch, then a larger again
    for i, (M, N, K) in enumerate([(96 * 256, 768, 768), (5 * 256, 768, 3072), (40 * 256, 512, 256)]):
        ops = _operands(M, N, K, 100 + i)
        x_ref, xres_ref = _existing(lib, ops, M, N, K, epoch=7 + i)
        x, xres = _noalloc(lib, ops, M, N, K)
        assert torch.isfinite(x.float()).all() and torch.isfinite(xres.float()).all()
        assert torch.equal(x.view(torch.int16), x_ref.view(torch.int16))           # bit-exact
        assert torch.equal(xres.view(torch.int16), xres_ref.view(torch.int16))


def test_noalloc_entry_rejects_sizes_beyond_its_scratch():
    lib = _lib.load()
    M, N, K = 2049 * 256, 256, 128          # 2049 tiles > CLIP_NOALLOC_MAX_TILES; no buffer is touched
    d = torch.zeros(16, device="cuda")
    rc = lib.clip_gemm_bf16_lnout_noalloc(_lib.ptr(d), _lib.ptr(d), _lib.ptr(d), _lib.ptr(d), 0, _lib.ptr(d), _lib.ptr(d),
                                          1e-12, _lib.ptr(d), _lib.ptr(d), _lib.ptr(d), M, N, K, K, K, N, N, None)
    assert rc == _lib.CLIP_ERR_CAPACITY and b"static scratch" in lib.clip_last_error()
