"""Logit error and thresholded-label agreement of one precision against the fp32 check mode, on the rows of
tests/test_gpu_parity_gates.py — for comparing library builds:
    CLIP_AMD_LIB=gpurun_tmp/libold.so python tools/agreement_ab.py [bf16|f16] [trained_like|procedural]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_amd import scoring, synth  # noqa: E402
from clip_amd.bert_model import args_to_model  # noqa: E402
from clip_amd.config import BertConfig  # noqa: E402
from clip_amd.weights import procedural_state_dict, trained_like_state_dict  # noqa: E402

precision = sys.argv[1] if len(sys.argv) > 1 else "bf16"
weights = sys.argv[2] if len(sys.argv) > 2 else "trained_like"
DNOTE = np.array([0.44, 0.31, 0.24, 0.17, 0.46, 0.35000000000000003, 0.43])
TTP = np.array([0.18, 0.6, 0.11, 0.32, 0.1, 0.1, 0.18])
cfg = BertConfig()
sd = (trained_like_state_dict if weights == "trained_like" else procedural_state_dict)(cfg, 0)


def probs(rows, prec):
    m = args_to_model(2)(cfg, max_tokens=32768, max_seqs=4096)
    m.set_precision(prec)
    m.load_state_dict(sd)
    m.to("cuda").eval()
    m.set_sep_token_id(102)
    m.set_n_context_sentences(2)
    m.set_task("multilabel")
    res = scoring.BulkScorer(m, keep_probs=True).score(rows)
    m._drop_engine()
    return res.probs.astype(np.float64)


for seed in (77, 177, 277):
    rows = synth.context_rows(4096, 2, cfg.vocab_size, seed=seed, fixed_len=256)
    p, q = probs(rows, precision), probs(rows, "fp32")
    lp, lq = np.log(p / (1 - p)), np.log(q / (1 - q))
    ok = np.isfinite(lp) & np.isfinite(lq)
    d = (lp - lq)[ok]
    line = f"seed {seed} {precision}/{weights}: logit err rms {np.sqrt((d * d).mean()):.4e} max {np.abs(d).max():.3e}"
    for name, thr in (("dnote", DNOTE), ("ttp", TTP)):
        bad = int(((p > thr) != (q > thr)).sum())
        line += f" | {name} {100 * (1 - bad / p.size):.4f} % ({bad})"
    print(line, flush=True)
