"""Sustained MFMA rate of the two bf16 shapes, register-only loops, interleaved:  python tools/peak_shapes.py"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401  (device context)
from clip_amd import _lib  # noqa: E402

lib = _lib.load()
torch.zeros(1, device="cuda")
f = lib.clip_debug_mfma_peak32
f.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
for rnd in range(3):
    best, mean, ms = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
    lib.clip_mfma_peak(0, 12000, 5, ctypes.byref(best), ctypes.byref(mean), ctypes.byref(ms))
    print(f"16x16x32 bf16: mean {mean.value:7.1f} TF  ({ms.value:.2f} ms)  clock {mean.value / 2500 * 2.4:.2f} GHz", flush=True)
    m32, ms32 = ctypes.c_double(), ctypes.c_double()
    f(12000, 5, ctypes.byref(m32), ctypes.byref(ms32))
    print(f"32x32x16 bf16: mean {m32.value:7.1f} TF  ({ms32.value:.2f} ms)  clock {m32.value / 2500 * 2.4:.2f} GHz", flush=True)
