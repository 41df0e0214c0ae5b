"""Where do the hand-written GEMMs stand against the vendor library?  torch.matmul / F.linear on the
encoder's four projection shapes (bf16, hipBLASLt / rocBLAS underneath), plain GEMM + bias only — no
GELU, residual or LayerNorm, which the hand-written kernels fuse — beside clip_gemm_bf16 (tile 2,
bias epilogue) on the same buffers, interleaved.   python tools/vendor_gemm_bench.py [M]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from clip_amd import _lib  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 65280
lib = _lib.load()
g = torch.Generator(device="cpu").manual_seed(0)
for name, N, K in (("qkv", 2304, 768), ("attn_out", 768, 768), ("ffn_up", 3072, 768), ("ffn_down", 768, 3072)):
    A = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
    W = (torch.randn(N, K, generator=g) * 0.02).to(torch.bfloat16).cuda()
    bias = torch.randn(N, generator=g).cuda()
    bias16 = bias.to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")

    def ours():
        rc = lib.clip_gemm_bf16(_lib.ptr(A), _lib.ptr(W), _lib.ptr(bias), None, _lib.ptr(out), M, N, K, K, K, N, 0, 0, 0, 2,
                                _lib.stream_ptr())
        assert rc == 0

    runs = {"ours (bias, bf16 out)": ours, "torch F.linear (bias)": lambda: F.linear(A, W, bias16),
            "torch matmul (no bias)": lambda: torch.matmul(A, W.t())}
    best = {k: 1e9 for k in runs}
    for f in runs.values():
        f()
    torch.cuda.synchronize()
    for rnd in range(5):
        for k, f in runs.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                f()
            e1.record()
            torch.cuda.synchronize()
            best[k] = min(best[k], e0.elapsed_time(e1) / 10)
    fl = 2.0 * M * N * K
    print(f"{name} M={M} N={N} K={K}: " + "  ".join(f"{k}: {v * 1e3:.1f} us {fl / v / 1e9:.0f} TF" for k, v in best.items()), flush=True)
