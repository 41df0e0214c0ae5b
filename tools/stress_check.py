"""Max |logit - reference| on the trained-like stress goldens (tests/golden/base_*_stress.npz, from the
imported reference) for each numerics setting and LayerNorm flow:   python tools/stress_check.py
Settings: bf16 / f16 (fused LayerNorm flow), f16 with CLIP_AMD_FUSED_LN=0 (fp32 pre-LayerNorm rows: the
residual stream never leaves fp32), fp32 check mode."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden, golden_state_dict  # noqa: E402
from test_gpu_parity_gates import _logits, DNOTE, TTP, _agreement  # noqa: E402

for name in ("base_ctx_stress", "base_sent_stress"):
    g = load_golden(name)
    sd = golden_state_dict(g)
    ref = g["ref_logits"]
    for precision, fused in (("bf16", "1"), ("f16", "1"), ("f16", "0"), ("bf16", "0"), ("fp32", "1")):
        os.environ["CLIP_AMD_FUSED_LN"] = fused
        lg = _logits(g["cfg"], sd, g["rows"], g["n_ctx"], precision, g["batch_size"])
        err = np.abs(lg - ref)
        p = (1 / (1 + np.exp(-lg.astype(np.float64)))).astype(np.float32)
        ag = [_agreement(p, g["ref_probs"], thr)[1] for thr in (DNOTE, TTP)]
        print(f"{name:18s} {precision:5s} fused_ln={fused}: max {err.max():.3e} rms {np.sqrt((err ** 2).mean()):.3e} "
              f"p99 {np.quantile(err, 0.99):.3e}  n={err.size}  decisions differing dnote/ttp {ag}", flush=True)
