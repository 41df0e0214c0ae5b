"""Which bf16 rounding points of the engine's data flow cost how much logit error?  A plain-torch
fp32 emulation of the flow (fixed-length rows, no padding) with each rounding switchable:
  W weights | X LayerNorm output as GEMM operand | XR LayerNorm output as residual | QKV | P softmax
  weights | CTX attention output | MID GELU output.
Prints the rms / max logit error against the all-fp32 run.   python tools/rounding_study.py [rows]"""
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_amd import synth  # noqa: E402
from clip_amd.config import BertConfig  # noqa: E402
from clip_amd.weights import procedural_state_dict  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
KIND = sys.argv[2] if len(sys.argv) > 2 else "procedural"       # or "trained": clip_amd.weights.trained_like_state_dict
DEV = "cuda" if torch.cuda.is_available() else "cpu"
L = 256
cfg = BertConfig()
if KIND == "trained":
    from clip_amd.weights import trained_like_state_dict
    sd = {k: v.to(DEV) for k, v in trained_like_state_dict(cfg, 0).items()}
else:
    sd = {k: v.to(DEV) for k, v in procedural_state_dict(cfg, 0).items()}
rows = synth.context_rows(n, 2, cfg.vocab_size, seed=1002, fixed_len=L)
ids = torch.from_numpy(np.stack(rows).astype(np.int64)).to(DEV)
sep = (ids == 102)
seps = torch.where(sep)[1].reshape(-1, 5)
tt = torch.zeros_like(ids)
for t, s in zip(tt, seps):
    t[:s[1]] = 1
    t[s[2]:] = 1
focus = seps[:, 2]
ALL = ("W", "X", "XR16", "QKV", "P", "CTX", "MID")      # XR16: the residual copy is fp16 (the engine); "XR": bf16


OPERAND = os.environ.get("OPERAND", "bf16")       # "f16": the k_f16 kernel set


def r(x, on):
    if not on:
        return x
    return x.half().float() if OPERAND == "f16" else x.bfloat16().float()


def rr(x, f):     # the residual copy
    return x.half().float() if f["XR16"] else r(x, f.get("XR", False))


def ln(x, w, b):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + cfg.layer_norm_eps) * w + b


@torch.no_grad()
def forward(flags, rows_idx):
    f = {k: (k in flags) for k in ALL + ("XR",)}
    i_ = ids[rows_idx]
    B = i_.shape[0]
    H, nh = cfg.hidden_size, cfg.num_attention_heads
    pos = torch.arange(L, device=DEV).unsqueeze(0).expand(B, L)
    x = ln(sd["bert.embeddings.word_embeddings.weight"][i_] + sd["bert.embeddings.token_type_embeddings.weight"][tt[rows_idx]]
           + sd["bert.embeddings.position_embeddings.weight"][pos],
           sd["bert.embeddings.LayerNorm.weight"], sd["bert.embeddings.LayerNorm.bias"])
    for i in range(cfg.num_hidden_layers):
        p = f"bert.encoder.layer.{i}."
        W = lambda k: r(sd[p + k + ".weight"], f["W"])  # noqa: E731
        bq = sd[p + "attention.self.query.bias"]
        xo, xr = r(x, f["X"]), rr(x, f)
        q = r((xo @ W("attention.self.query").T + bq) * 0.125, f["QKV"]).view(B, L, nh, 64).transpose(1, 2)
        k = r(xo @ W("attention.self.key").T + sd[p + "attention.self.key.bias"], f["QKV"]).view(B, L, nh, 64).transpose(1, 2)
        v = r(xo @ W("attention.self.value").T + sd[p + "attention.self.value.bias"], f["QKV"]).view(B, L, nh, 64).transpose(1, 2)
        s = q @ k.transpose(2, 3)
        pu = torch.exp(s - s.max(-1, keepdim=True).values)
        o = (r(pu, f["P"]) @ v) / pu.sum(-1, keepdim=True)
        ctx = r(o.transpose(1, 2).reshape(B, L, H), f["CTX"])
        x1 = ln(ctx @ W("attention.output.dense").T + sd[p + "attention.output.dense.bias"] + xr,
                sd[p + "attention.output.LayerNorm.weight"], sd[p + "attention.output.LayerNorm.bias"])
        x1o, x1r = r(x1, f["X"]), rr(x1, f)
        h = x1o @ W("intermediate.dense").T + sd[p + "intermediate.dense.bias"]
        mid = r(0.5 * h * (1 + torch.erf(h / math.sqrt(2.0))), f["MID"])
        x = ln(mid @ W("output.dense").T + sd[p + "output.dense.bias"] + x1r,
               sd[p + "output.LayerNorm.weight"], sd[p + "output.LayerNorm.bias"])
    xf = r(x, f["X"])[torch.arange(B), focus[rows_idx]]
    return xf @ sd["classifier.weight"].T + sd["classifier.bias"]


def run(flags):
    out = []
    for a in range(0, n, 64):
        out.append(forward(flags, torch.arange(a, min(n, a + 64), device=DEV)).double().cpu().numpy())
    return np.concatenate(out)


ref = run(())
print(f"weights {KIND}: fp32 logits min {ref.min():.2f} max {ref.max():.2f} std {ref.std():.2f}", flush=True)
def report(name, flags):
    d = run(flags) - ref
    print(f"{name:28s} rms {np.sqrt((d * d).mean()):.3e}  max {np.abs(d).max():.3e}", flush=True)
report("all (engine today)", ALL)
for k in ALL:
    report("only " + k, (k,))
for k in ALL:
    report("all but " + k, tuple(x for x in ALL if x != k))
report("all, residual bf16", tuple(x for x in ALL if x != "XR16") + ("XR",))
