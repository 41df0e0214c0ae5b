"""A/B the GEMM tile shapes on the encoder's four projection shapes (interleaved rounds in one
process, random data; guide rule 24/25). Usage: python tools/gemm_bench.py [M]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clip_amd import _lib

lib = _lib.load()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
shapes = [("qkv", 2304, 768, 0, False), ("attn_out", 768, 768, 2, True), ("ffn_up", 3072, 768, 1, False),
          ("ffn_down", 768, 3072, 2, True)]
tiles = [int(t) for t in os.environ.get("TILES", "1,2").split(",")]
g = torch.Generator(device="cpu").manual_seed(0)
for name, N, K, epi, f32 in shapes:
    A = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
    W = (torch.randn(N, K, generator=g) * 0.02).to(torch.bfloat16).cuda()
    bias = torch.randn(N, generator=g).cuda()
    res = torch.randn(M, N, generator=g).to(torch.bfloat16).cuda() if epi == 2 else None
    out = torch.empty(M, N, dtype=torch.float32 if f32 else torch.bfloat16, device="cuda")
    def run(tile):
        _lib.check(lib.clip_gemm_bf16(_lib.ptr(A), _lib.ptr(W), _lib.ptr(bias), _lib.ptr(res), _lib.ptr(out), M, N, K,
                                      K, K, N, N if res is not None else 0, epi, int(f32), tile, _lib.stream_ptr()))
    best = {t: 1e9 for t in tiles}
    for t in tiles:
        run(t)
    torch.cuda.synchronize()
    for rnd in range(5):
        for t in tiles:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                run(t)
            e1.record(); torch.cuda.synchronize()
            best[t] = min(best[t], e0.elapsed_time(e1) / 10)
    fl = 2.0 * M * N * K
    print(name, f"M={M} N={N} K={K}", "  ".join(f"tile{t}: {best[t]*1e3:.1f} us {fl/best[t]/1e9:.0f} TF" for t in tiles), flush=True)
