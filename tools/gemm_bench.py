"""A/B GEMM kernels / builds on the encoder's four projection shapes (interleaved rounds in one
process, random data; guide rules 24/25).

    python tools/gemm_bench.py [M]
    VARIANTS="base:1,base:2,noepi:2" python tools/gemm_bench.py     # lib:tile pairs

`base` is the product library; any other name is gpurun_tmp/lib<name>.so (tools/build_variant.sh).
tile: 1 persistent, 2 ping-pong persistent, 128 / 256 one tile per workgroup."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from clip_amd import _lib  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
M = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
variants = [v.split(":") for v in os.environ.get("VARIANTS", "base:1,base:2").split(",")]
libs = {}
for name, _ in variants:
    if name in libs:
        continue
    if name == "base":
        libs[name] = _lib.load()
    else:
        lib = C.CDLL(os.path.join(ROOT, "gpurun_tmp", f"lib{name}.so"))
        lib.clip_gemm_bf16.argtypes = _lib.SIGNATURES["clip_gemm_bf16"]
        lib.clip_gemm_bf16.restype = C.c_int
        lib.clip_gemm_bf16_lnres.argtypes = _lib.SIGNATURES["clip_gemm_bf16_lnres"]
        lib.clip_gemm_bf16_lnres.restype = C.c_int
        lib.clip_gemm_bf16_lnout.argtypes = _lib.SIGNATURES["clip_gemm_bf16_lnout"]
        lib.clip_gemm_bf16_lnout.restype = C.c_int
        libs[name] = lib
# epi 4 = the engine's residual form (fp32 LayerNorm output recomputed in the epilogue)
shapes = [("qkv", 2304, 768, 0, False), ("attn_out", 768, 768, 4, True), ("ffn_up", 3072, 768, 1, False),
          ("ffn_down", 768, 3072, 4, True)]
if os.environ.get("SHAPES") == "ln":    # fused LayerNorm (epi 5) against GEMM + residual recompute (epi 4; + a 46 us LayerNorm kernel)
    shapes = [("attn_out lnres", 768, 768, 4, True), ("attn_out lnout", 768, 768, 5, False),
              ("ffn_down lnres", 768, 3072, 4, True), ("ffn_down lnout", 768, 3072, 5, False)]
if os.environ.get("SHAPES") == "bf16out":   # the two shapes gemm9.hip (tile 3) takes
    shapes = [("qkv", 2304, 768, 0, False), ("ffn_up", 3072, 768, 1, False)]
if os.environ.get("SHAPES") == "res":   # what does the residual read cost the fp32-output GEMMs?
    shapes = [("attn_out nores", 768, 768, 0, True), ("attn_out bf16res", 768, 768, 2, True), ("attn_out lnres", 768, 768, 4, True),
              ("ffn_down nores", 768, 3072, 0, True), ("ffn_down bf16res", 768, 3072, 2, True), ("ffn_down lnres", 768, 3072, 4, True),
              ("attn_out bf16out nores", 768, 768, 0, False)]
g = torch.Generator(device="cpu").manual_seed(0)
for sname, N, K, epi, f32 in shapes:
    A = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
    # ROTATE=n: n copies of A used in turn, so that the operand does not stay in the 256-MiB Infinity Cache
    # between launches (in the engine every GEMM reads an operand the kernel before it has just written)
    A_all = [A] + [A.clone() for _ in range(int(os.environ.get("ROTATE", "1")) - 1)]
    calls = [0]
    W = (torch.randn(N, K, generator=g) * 0.02).to(torch.bfloat16).cuda()
    bias = torch.randn(N, generator=g).cuda()
    res = torch.randn(M, N, generator=g).to(torch.bfloat16).cuda() if epi == 2 else None
    out = torch.empty(M, N, dtype=torch.float32 if f32 else torch.bfloat16, device="cuda")
    if epi == 4:
        pre = torch.randn(M, N, generator=g).cuda()
        stats = torch.stack([pre.mean(1), 1 / pre.std(1)], 1).contiguous()
        gamma, beta = torch.ones(N, device="cuda"), torch.zeros(N, device="cuda")

    if epi == 5:
        res16 = torch.randn(M, N, generator=g).to(torch.float16).cuda()
        xres = res16.clone()    # tiled residual, rewritten in place as in the engine
        part = torch.empty((M // 256) * (N // 256) * 512, device="cuda")
        flags = torch.zeros((M // 256) * (N // 256), dtype=torch.int32, device="cuda")
        errw = torch.zeros(1, dtype=torch.int32, device="cuda")
        gamma, beta = torch.ones(N, device="cuda"), torch.zeros(N, device="cuda")
        epoch = [0]

    def run(v):
        A = A_all[calls[0] % len(A_all)]
        calls[0] += 1
        if epi == 5:
            epoch[0] += 1
            rc = libs[v[0]].clip_gemm_bf16_lnout(_lib.ptr(A), _lib.ptr(W), _lib.ptr(bias), _lib.ptr(xres), int(os.environ.get("RES_TILED", "1")), _lib.ptr(gamma),
                                                 _lib.ptr(beta), 1e-12, _lib.ptr(out), _lib.ptr(xres), _lib.ptr(part),
                                                 _lib.ptr(flags), epoch[0], _lib.ptr(errw), M, N, K, K, K, N, N,
                                                 _lib.stream_ptr())
            assert rc == 0, rc
            return
        if epi == 4:
            rc = libs[v[0]].clip_gemm_bf16_lnres(_lib.ptr(A), _lib.ptr(W), _lib.ptr(bias), _lib.ptr(pre), _lib.ptr(stats),
                                                 _lib.ptr(gamma), _lib.ptr(beta), None, _lib.ptr(out), M, N, K, K, K, N, N,
                                                 int(v[1]), _lib.stream_ptr())
            assert rc == 0, rc
            return
        rc = libs[v[0]].clip_gemm_bf16(_lib.ptr(A), _lib.ptr(W), _lib.ptr(bias), _lib.ptr(res), _lib.ptr(out), M, N, K,
                                       K, K, N, N if res is not None else 0, epi, int(f32), int(v[1]),
                                       _lib.stream_ptr())
        assert rc == 0, rc
    keys = [":".join(v) for v in variants]
    best = {k: 1e9 for k in keys}
    for v in variants:
        run(v)
    torch.cuda.synchronize()
    for rnd in range(5):
        for v, k in zip(variants, keys):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                run(v)
            e1.record()
            torch.cuda.synchronize()
            best[k] = min(best[k], e0.elapsed_time(e1) / 10)
    fl = 2.0 * M * N * K
    print(sname, f"M={M} N={N} K={K}", "  ".join(f"{k}: {best[k]*1e3:.1f} us {fl/best[k]/1e9:.0f} TF" for k in keys),
          flush=True)
