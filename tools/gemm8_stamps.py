"""Where one K tile of the ping-pong GEMM's main loop spends its cycles: s_memtime stamps of waves 0 and 4
(-DGEMM8_STAMPS build: tools/build_variant.sh stamps -DGEMM8_STAMPS).   python tools/gemm8_stamps.py [N K]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from clip_amd import _lib  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(ROOT, "gpurun_tmp", "libstamps.so"))
lib.clip_gemm_bf16.argtypes = _lib.SIGNATURES["clip_gemm_bf16"]
N, K = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2304, 768)
M = 130560
g = torch.Generator(device="cpu").manual_seed(0)
A = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
W = (torch.randn(N, K, generator=g) * 0.02).to(torch.bfloat16).cuda()
bias = torch.randn(N, generator=g).cuda()
out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
for _ in range(20):
    assert lib.clip_gemm_bf16(_lib.ptr(A), _lib.ptr(W), _lib.ptr(bias), None, _lib.ptr(out), M, N, K, K, K, N, 0, 0, 0, 2, None) == 0
torch.cuda.synchronize()
buf = np.zeros(256 * 2 * 16, np.uint32)
assert lib.clip_debug_gemm8_stamps(buf.ctypes.data_as(C.c_void_p), buf.size) == 0
st = buf.reshape(256, 2, 16)[:, :, :10].astype(np.int64)
d = np.diff(st, axis=2) & 0xffffffff
names = ["A: reads+DMA+waits", "A: barrier", "A: 32 MFMAs issued", "A: barrier", "B: reads+DMA+waits", "B: barrier",
         "B: 32 MFMAs issued", "B: barrier"]
print(f"N={N} K={K}: median cycles over 256 workgroups (wave 0 | wave 4); K tile total "
      f"{np.median((st[:, 0, 8] - st[:, 0, 0]) & 0xffffffff):.0f} | {np.median((st[:, 1, 8] - st[:, 1, 0]) & 0xffffffff):.0f}")
names.append("to next K tile's start")
for k, n in enumerate(names):
    print(f"  {n:22s} {np.median(d[:, 0, k]):7.0f} | {np.median(d[:, 1, k]):7.0f}")
rel = (st - st[:, :1, :1]) & 0xffffffff
rel = np.where(rel > 1 << 31, rel - (1 << 32), rel)
print("  timeline (median cycles since wave 0's stamp 0):")
print("   wave 0:", " ".join(f"{np.median(rel[:, 0, k]):6.0f}" for k in range(10)))
print("   wave 4:", " ".join(f"{np.median(rel[:, 1, k]):6.0f}" for k in range(10)))
for wg in (3, 100, 200):
    print(f"   WG {wg}: wave0", " ".join(f"{int(x):6d}" for x in rel[wg, 0, :10]), "| wave4", " ".join(f"{int(x):6d}" for x in rel[wg, 1, :10]))
