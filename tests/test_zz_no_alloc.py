"""clip_gemm_bf16_lnout_noalloc (include/clip_amd.h): the entry point that allocates nothing — the
fused-LayerNorm GEMM with its exchange buffers in a static device array inside the library.  Checked
against clip_gemm_bf16_lnout on the same operands: identical bits (bf16 x and tiled fp16 xres), on two
calls in a row, the second on a smaller input (the array keeps the first call's contents), then a third
larger one again; a call captured into a HIP graph and replayed on changed inputs; and the size limit is an
error, not a fault.  "Allocates nothing" is checked at the DEVICE level (hipMemGetInfo free bytes around
the call: a hipMalloc inside the library would show, torch's own allocator counters would not see it).
Misuse the header rules out and this file does not try: two calls in flight at once on one GPU (a second
stream or thread) share the static array and race — serialise them."""
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


def _noalloc(lib, ops, M, N, K, check_device_free=True):
    A, W, bias, res, gamma, beta = ops
    err = torch.zeros(1, dtype=torch.int32, device="cuda")
    x = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
    xres = torch.full((M * N,), float("nan"), dtype=torch.float16, device="cuda")
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]      # hipMemGetInfo: sees any hipMalloc / hipFree in the library
    before = torch.cuda.memory_allocated()
    rc = lib.clip_gemm_bf16_lnout_noalloc(_lib.ptr(A), _lib.ptr(W), _lib.ptr(bias), _lib.ptr(res), 0, _lib.ptr(gamma),
                                          _lib.ptr(beta), 1e-12, _lib.ptr(x), _lib.ptr(xres), _lib.ptr(err),
                                          M, N, K, K, K, N, N, _lib.stream_ptr())
    _lib.check(rc)
    torch.cuda.synchronize()
    assert torch.cuda.memory_allocated() == before
    if check_device_free:
        assert torch.cuda.mem_get_info()[0] == free_before
    assert int(err.item()) == 0
    return x, xres


def test_noalloc_entry_matches_existing_entry_on_consecutive_calls():
    lib = _lib.load()
    # (M, N, K): a full-chip call, then a SMALLER one on the same static scratch, then a larger again
    # warm-up: the first call on a device loads the code object and resolves the array's address (the
    # driver may map memory for that once); the device-level check starts after it
    _noalloc(lib, _operands(256, 256, 128, 99), 256, 256, 128, check_device_free=False)
    for i, (M, N, K) in enumerate([(96 * 256, 768, 768), (5 * 256, 768, 3072), (40 * 256, 512, 256)]):
        ops = _operands(M, N, K, 100 + i)
        x_ref, xres_ref = _existing(lib, ops, M, N, K, epoch=7 + i)
        x, xres = _noalloc(lib, ops, M, N, K)
        assert torch.isfinite(x.float()).all() and torch.isfinite(xres.float()).all()
        assert torch.equal(x.view(torch.int16), x_ref.view(torch.int16))           # bit-exact
        assert torch.equal(xres.view(torch.int16), xres_ref.view(torch.int16))


def test_noalloc_entry_replays_in_a_hip_graph():
    """One captured call (flag clear + kernel), replayed twice on changed inputs, after a LARGER eager call has left
    its flags behind: each replay must equal the existing entry point on the inputs of that replay."""
    lib = _lib.load()
    M, N, K = 24 * 256, 768, 768
    _noalloc(lib, _operands(48 * 256, 768, 768, 300), 48 * 256, 768, 768, check_device_free=False)   # eager, larger
    ops = [t.clone() for t in _operands(M, N, K, 301)]
    A, W, bias, res, gamma, beta = ops
    err = torch.zeros(1, dtype=torch.int32, device="cuda")
    x = torch.zeros((M, N), dtype=torch.bfloat16, device="cuda")
    xres = torch.zeros((M * N,), dtype=torch.float16, device="cuda")
    graph = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(graph):
        rc = lib.clip_gemm_bf16_lnout_noalloc(_lib.ptr(A), _lib.ptr(W), _lib.ptr(bias), _lib.ptr(res), 0, _lib.ptr(gamma),
                                              _lib.ptr(beta), 1e-12, _lib.ptr(x), _lib.ptr(xres), _lib.ptr(err),
                                              M, N, K, K, K, N, N, _lib.stream_ptr())
    _lib.check(rc)
    for seed in (302, 303, 304):
        new = _operands(M, N, K, seed)
        for dst, src in zip(ops, new):
            dst.copy_(src)
        x.fill_(float("nan")); xres.fill_(float("nan"))
        graph.replay()
        torch.cuda.synchronize()
        assert int(err.item()) == 0
        x_ref, xres_ref = _existing(lib, new, M, N, K, epoch=seed)
        assert torch.equal(x.view(torch.int16), x_ref.view(torch.int16))
        assert torch.equal(xres.view(torch.int16), xres_ref.view(torch.int16))


def test_noalloc_entry_rejects_sizes_beyond_its_scratch():
    lib = _lib.load()
    M, N, K = 2049 * 256, 256, 128          # 2049 tiles > CLIP_NOALLOC_MAX_TILES; no buffer is touched
    d = torch.zeros(16, device="cuda")
    rc = lib.clip_gemm_bf16_lnout_noalloc(_lib.ptr(d), _lib.ptr(d), _lib.ptr(d), _lib.ptr(d), 0, _lib.ptr(d), _lib.ptr(d),
                                          1e-12, _lib.ptr(d), _lib.ptr(d), _lib.ptr(d), M, N, K, K, K, N, N, None)
    assert rc == _lib.CLIP_ERR_CAPACITY and b"static scratch" in lib.clip_last_error()
