"""Where does the fused-LayerNorm GEMM's epilogue spend its time?  Runs the -DLNOUT_STAMPS build
(tools/build_variant.sh stamps -DLNOUT_STAMPS) and prints median cycle counts between stamps over
workgroups (second tile of each).   python tools/lnout_stamps.py [M] [K]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from clip_amd import _lib  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
M = int(sys.argv[1]) if len(sys.argv) > 1 else 65280
K = int(sys.argv[2]) if len(sys.argv) > 2 else 768
N = 768
lib = C.CDLL(os.path.join(ROOT, "gpurun_tmp", os.environ.get("STAMPS_LIB", "libstamps.so")))
lib.clip_gemm_bf16_lnout.argtypes = _lib.SIGNATURES["clip_gemm_bf16_lnout"]
lib.clip_gemm_bf16_lnout.restype = C.c_int
g = torch.Generator(device="cpu").manual_seed(0)
A = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
W = (torch.randn(N, K, generator=g) * 0.02).to(torch.bfloat16).cuda()
bias = torch.randn(N, generator=g).cuda()
xres = torch.randn(M, N, generator=g).to(torch.float16).cuda()
out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
part = torch.empty((M // 256) * (N // 256) * 512, device="cuda")
flags = torch.zeros((M // 256) * (N // 256), dtype=torch.int32, device="cuda")
errw = torch.zeros(64 + 8 * 256, dtype=torch.int32, device="cuda")
gamma, beta = torch.ones(N, device="cuda"), torch.zeros(N, device="cuda")
for epoch in range(1, 30):
    rc = lib.clip_gemm_bf16_lnout(_lib.ptr(A), _lib.ptr(W), _lib.ptr(bias), _lib.ptr(xres), 1, _lib.ptr(gamma),
                                  _lib.ptr(beta), 1e-12, _lib.ptr(out), _lib.ptr(xres), _lib.ptr(part),
                                  _lib.ptr(flags), epoch, _lib.ptr(errw), M, N, K, K, K, N, N, _lib.stream_ptr())
    assert rc == 0
torch.cuda.synchronize()
st = errw.cpu().numpy()[64:].astype(np.uint32).reshape(256, 8)
st = st[st[:, 1] != 0]
d = np.diff(st.astype(np.int64), axis=1) & 0xffffffff
names = ["main loop", "residual add", "row sums -> LDS", "publish + flag", "partner wait", "stats", "normalise + store issue"]
print(f"M={M} K={K}: {len(st)} workgroups, median cycles (p10..p90)")
for k, n in enumerate(names):
    print(f"  {n:26s} {int(np.median(d[:, k])):8d}  ({int(np.percentile(d[:, k], 10))} .. {int(np.percentile(d[:, k], 90))})")
