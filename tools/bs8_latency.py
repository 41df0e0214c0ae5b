"""The reference's own loop shape (apply_bert_unlabeled.py:92-124: batches of 8 padded rows through the nn.Module,
sigmoid, D2H per batch) on the drop-in classes, beside the bulk scorer on the same rows.
    python tools/bs8_latency.py [rows] [batch]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from clip_amd import scoring, synth  # noqa: E402
from clip_amd.bert_model import args_to_model  # noqa: E402
from clip_amd.config import BertConfig  # noqa: E402
from clip_amd.weights import procedural_state_dict  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
cfg = BertConfig()
m = args_to_model(2)(cfg, max_tokens=131072, max_seqs=16384)
m.load_state_dict(procedural_state_dict(cfg, 0))
m.set_precision(os.environ.get("PRECISION", "bf16"))
m.to("cuda").eval()
m.set_sep_token_id(102); m.set_n_context_sentences(2); m.set_task("multilabel")
flat, off = synth.packed_context_rows(N, 2, cfg.vocab_size, 256, seed=5)
rows = [flat[off[i]:off[i + 1]] for i in range(N)]
batches = []
for i in range(0, N, B):
    rs = rows[i:i + B]
    S = max(len(r) for r in rs)
    ids = np.zeros((len(rs), S), np.int64); mask = np.zeros((len(rs), S), np.int64); tt = np.zeros((len(rs), S), np.int64)
    for j, r in enumerate(rs):
        ids[j, :len(r)] = r; mask[j, :len(r)] = 1
        seps = np.flatnonzero(r == 102)
        tt[j, :seps[1]] = 1; tt[j, seps[2]:] = 1
    batches.append(tuple(torch.from_numpy(x) for x in (ids, mask, tt)))
lab = torch.zeros((B, 7))


def loop():
    out = []
    for ids, mask, tt in batches:
        loss, logits = m(input_ids=ids.cuda(), attention_mask=mask.cuda(), token_type_ids=tt.cuda(),
                         labels=lab[:ids.shape[0]].cuda())
        out.append(torch.sigmoid(logits).cpu().numpy())
    return np.concatenate(out)


loop()
torch.cuda.synchronize()
t0 = time.perf_counter(); p8 = loop(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"nn.Module loop, batches of {B} (mean {np.mean([len(r) for r in rows]):.0f} tokens/row): {dt / len(batches) * 1e3:.3f} ms per batch, "
      f"{N / dt:,.0f} sentences/s", flush=True)
sc = scoring.BulkScorer(m, keep_probs=True)
sc.score(scoring.PackedRows(flat, off))
t0 = time.perf_counter(); r = sc.score(scoring.PackedRows(flat, off)); dt2 = time.perf_counter() - t0
print(f"bulk scorer, same rows: {N / dt2:,.0f} sentences/s   max |p_bulk - p_loop| = {np.abs(r.probs - p8).max():.1e}", flush=True)
