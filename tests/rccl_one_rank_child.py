"""Child process of tests/test_gpu_multirank.py::test_rccl_group_and_fused_flow_in_one_process (not collected by pytest).

What a one-GPU box can show of the N > 1 leg: an RCCL ("nccl") process group — of one rank, the only size one card
allows — initialised in the SAME process that scores with the production kernel flow (fused-LayerNorm GEMMs, which spin
across workgroups), followed by the job's collectives on device memory: the top-k all-gather, the counters' all-reduce,
the barrier.  Prints one JSON line."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_amd import synth, topk                                   # noqa: E402
from clip_amd.bert_model import args_to_model                      # noqa: E402
from clip_amd.config import BertConfig                             # noqa: E402
from clip_amd.scoring import BulkScorer                            # noqa: E402
from clip_amd.weights import procedural_state_dict                 # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    cfg = BertConfig(vocab_size=3000, hidden_size=768, num_hidden_layers=3, num_attention_heads=12, intermediate_size=3072)
    model = args_to_model(2)(cfg, max_tokens=32768, max_seqs=512)
    model.set_precision("bf16")
    model.load_state_dict(procedural_state_dict(cfg, 3))
    model.to(dev).eval()
    model.set_sep_token_id(102)
    model.set_n_context_sentences(2)
    rows = synth.context_rows(3000, 2, cfg.vocab_size, max_len=120, seed=4)
    scorer = BulkScorer(model, any_threshold=0.3, tokens_per_step=32768)
    doc_ids, sent_ixs = np.arange(len(rows)) // 40, np.arange(len(rows)) % 40
    before = scorer.score(rows, doc_ids, sent_ixs)
    # collectives between two scoring passes: RCCL's kernels and the engine's share the card and the process
    k = 500
    local = topk.local_topk(before.doc_ids, before.sent_ixs, before.max_score, k, keep=before.keep)
    merged = topk.global_topk(local, k, device=dev, always_collective=True)
    tot = torch.tensor([len(rows), int(before.keep.sum())], dtype=torch.int64, device=dev)
    dist.all_reduce(tot)
    dist.barrier()
    after = scorer.score(rows, doc_ids, sent_ixs)
    print(json.dumps({
        "backend": dist.get_backend(), "world": dist.get_world_size(),
        "fused_layernorm": bool(getattr(model, "fused_layernorm", True)) and os.environ.get("CLIP_AMD_FUSED_LN", "1") != "0",
        "gather_equals_local_merge": bool(np.array_equal(merged, topk.merge_topk([local], k))),
        "records": int(len(merged)), "totals": tot.tolist(), "expected_totals": [len(rows), int(before.keep.sum())],
        "same_bits_after_collectives": bool(np.array_equal(before.max_score, after.max_score)),
        "finite": bool(np.isfinite(after.max_score).all())}))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
