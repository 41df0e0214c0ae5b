"""End-to-end BulkScorer throughput on the synthetic configs of BASELINE.json (host packing, H2D and
D2H included; one GPU):  C2 sentence-only seq<=64, C3 context n=2 varlen <=256, C5-like varlen <=512.
    python tools/scorer_bench.py [rows_c2] [rows_c3] [rows_c5]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_amd import synth  # noqa: E402
from clip_amd.bert_model import args_to_model  # noqa: E402
from clip_amd.config import BertConfig  # noqa: E402
from clip_amd.scoring import BulkScorer, PackedRows  # noqa: E402
from clip_amd.weights import procedural_state_dict  # noqa: E402

n2 = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
n3 = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
n5 = int(sys.argv[3]) if len(sys.argv) > 3 else 50000
cfg = BertConfig()
sd = procedural_state_dict(cfg, 0)
H, I = cfg.hidden_size, cfg.intermediate_size


def flops(lens):
    lens = lens.astype(np.float64)
    return float((12 * (2 * lens * (4 * H * H + 2 * H * I) + 4 * lens * lens * H) + 2 * H * 7).sum())


def bench(name, n_ctx, rows):
    model = args_to_model(n_ctx)(cfg, max_tokens=65536, max_seqs=8192)
    model.load_state_dict(sd)
    model.to("cuda").eval()
    if n_ctx:
        model.set_sep_token_id(102)
        model.set_n_context_sentences(n_ctx)
    packed = PackedRows.from_rows(rows)
    scorer = BulkScorer(model, any_threshold=0.05)
    scorer.score(PackedRows.from_rows(rows[:4096]))   # warm-up (engine build, kernels)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = scorer.score(packed)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{name}: {len(rows)} rows, mean len {packed.lens.mean():.1f}, {packed.lens.sum()} tokens: "
          f"{len(rows) / dt:,.0f} sent/s, {packed.lens.sum() / dt / 1e6:.2f} Mtok/s, "
          f"{flops(packed.lens) / dt / 1e12:.0f} model TFLOP/s, kept {res.keep.mean() * 100:.1f} %", flush=True)
    del model


bench("C2 sentence-only (<=64)", 0, synth.sentence_rows(n2, cfg.vocab_size, 64, seed=1001))
bench("C3 context n=2 varlen (<=256)", 2, synth.context_rows(n3, 2, cfg.vocab_size, 256, seed=1002))
bench("C5 context n=2 varlen (<=512, sigma 0.9)", 2,
      synth.context_rows(n5, 2, cfg.vocab_size, 512, seed=1005, sigma=0.9, seg_hi=200))
