"""Where an attention workgroup's time goes: s_memtime stamps of wave 0 (-DATTN_STAMPS build:
tools/build_variant.sh astamps -DATTN_STAMPS).   python tools/attn_stamps.py [L] [n_seqs]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from clip_amd import _lib  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(ROOT, "gpurun_tmp", "libastamps.so"))
lib.clip_attention_varlen.argtypes = _lib.SIGNATURES["clip_attention_varlen"]
L = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = int(sys.argv[2]) if len(sys.argv) > 2 else 510
H, nh = 768, 12
g = torch.Generator(device="cpu").manual_seed(0)
qkv = torch.randn(n * L, 3 * H, generator=g).to(torch.bfloat16).cuda()
cu = (torch.arange(n + 1, dtype=torch.int32) * L).cuda()
ctx = torch.empty(n * L, H, dtype=torch.bfloat16, device="cuda")
for _ in range(5):
    assert lib.clip_attention_varlen(_lib.ptr(qkv), _lib.ptr(cu), _lib.ptr(ctx), n, L, nh, H, None) == 0
torch.cuda.synchronize()
buf = np.zeros(4096 * 16, np.uint32)
assert lib.clip_debug_attn_stamps(buf.ctypes.data_as(C.c_void_p), buf.size) == 0
st = buf.reshape(4096, 16)[:, :10].astype(np.int64)
d = np.diff(st, axis=1) & 0xffffffff
names = ["entry -> loop (cu, Q, tile 0 loads issued)", "key tile 0 (whole)", "tile 1: barrier, store K/V, barrier", "tile 1: next loads issued",
         "tile 1: S = K Q^T (8 MFMAs)", "tile 1: mask / max / rescale", "tile 1: exp, P, PV (8 MFMAs)", "tiles 2.. to loop exit", "normalise + store"]
print(f"L={L} n={n}: median cycles of wave 0 over 4096 workgroups; workgroup life {np.median((st[:, 9] - st[:, 0]) & 0xffffffff):.0f}")
for k, nm in enumerate(names):
    print(f"  {nm:45s} {np.median(d[:, k]):8.0f}   (p10 {np.percentile(d[:, k], 10):.0f}, p90 {np.percentile(d[:, k], 90):.0f})")
