"""Run a fine-tuned BERT(+Context) classifier over unlabeled note sentences — the MI355X drop-in
for /root/reference/sentclf/apply_bert_unlabeled.py.

    python -m clip_amd.apply_bert_unlabeled UNLABELED.csv MODEL_DIR LOCAL_WEIGHTS \
        [--threshold 0.05] [--n_context_sentences 2] [--task multilabel] [--topk 250000]

Same positional arguments and flags as the reference (:18-27); MODEL_DIR is a LOCAL directory with
config.json + vocab.txt (hub short names 'bert' / 'clinicalbert_disch' cannot resolve offline).
Writes `ttp_doc_sents.csv` next to LOCAL_WEIGHTS exactly as the reference does (:90,97-98,124: no
header, `doc_id,sent_ix,max_label_score` for rows with any label score > threshold, input order).
With --topk K it also writes `ttp_top{K}.csv` WITH the header `doc_id,sent_ix,max_label_score`, the
file sentclf/pretrain_tasks.py:37,57-58 reads as `focus_fname` (README.md:43: "sorted by decreasing
max_label_score ... take the top 250k").

One process per GPU: under `torch.distributed.run` the documents are sharded over ranks
(topk.shard_documents), every rank scores its shard with no communication, parts are concatenated in
rank order (= input order) and the top-k uses one all-gather.
"""
import argparse
import csv
import os
import sys

import numpy as np
import torch

from . import topk as topk_mod
from .bert_model import args_to_model
from .fast_tokenizer import FastWordPiece, build_rows_fast
from .config import BertConfig
from .constants import LABEL_TYPES
from .datasets import SentDataset, SentEvalDataset
from .scoring import BulkScorer, PackedRows, ScoreResult, write_ttp_csv


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("unlabeled_fname", type=str)
    parser.add_argument("model", type=str, help="local directory with config.json and vocab.txt")
    parser.add_argument("local_weights", type=str,
                        help="state-dict file (.pth / .bin / .safetensors) of the fine-tuned model")
    parser.add_argument("--threshold", type=float, default=0.05)
    parser.add_argument("--n_context_sentences", type=int, default=0)
    parser.add_argument("--task", choices=["binary", "multilabel"], default="multilabel")
    parser.add_argument("--topk", type=int, default=0, help="also write the global top-k file")
    parser.add_argument("--tokens_per_step", type=int, default=65536)
    args = parser.parse_args(argv)
    if args.task != "multilabel":
        raise NotImplementedError("only --task multilabel is on the accelerated path")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    # C++ WordPiece on all host cores (libclip_tok.so); the Python restatement behind it takes what it flags
    tokenizer = FastWordPiece(os.path.join(args.model, "vocab.txt"))                   # :43-47
    config = BertConfig.from_pretrained(args.model, num_labels=len(LABEL_TYPES))       # :55-62
    model = args_to_model(args.n_context_sentences)(config, max_tokens=args.tokens_per_step,
                                                    max_seqs=max(64, args.tokens_per_step // 8))  # :64-70
    if args.local_weights.endswith(".safetensors"):
        from safetensors.torch import load_file
        sd = load_file(args.local_weights)
    else:
        sd = torch.load(args.local_weights, map_location="cpu", weights_only=True)       # :78
    model.load_state_dict(sd)                                                          # :79
    model.to(device).eval()                                                            # :72,100
    if args.n_context_sentences > 0:
        model.set_sep_token_id(tokenizer.sep_token_id)                                 # :74
        model.set_n_context_sentences(args.n_context_sentences)                        # :75
    model.set_task(args.task)                                                          # :80

    if args.n_context_sentences == 0:                                                  # :83-88
        dataset = SentDataset(args.unlabeled_fname, args.task)
    else:
        dataset = SentEvalDataset(args.unlabeled_fname, args.task,
                                  n_context_sentences=args.n_context_sentences)
    rows, doc_ids, sent_ixs = build_rows_fast(dataset, tokenizer, args.n_context_sentences)
    a, b = topk_mod.shard_documents(doc_ids, rows.lens, world, rank)
    shard = PackedRows(rows.flat[rows.offsets[a]:rows.offsets[b]], rows.offsets[a:b + 1] - rows.offsets[a])
    res = BulkScorer(model, any_threshold=args.threshold).score(shard, doc_ids[a:b], sent_ixs[a:b])

    out_dir = os.path.dirname(args.local_weights) or "."                               # :90
    out = os.path.join(out_dir, "ttp_doc_sents.csv")
    part = out if world == 1 else f"{out}.part{rank}"
    write_ttp_csv(part, res)
    if dist is not None:
        dist.barrier()
        if rank == 0:
            with open(out, "w") as f:
                for r in range(world):
                    with open(f"{out}.part{r}") as g:
                        f.write(g.read())
                    os.remove(f"{out}.part{r}")
    if args.topk > 0:
        local = topk_mod.local_topk(res.doc_ids, res.sent_ixs, res.max_score, args.topk, keep=res.keep)
        top = topk_mod.global_topk(local, args.topk)
        if rank == 0:
            with open(os.path.join(out_dir, f"ttp_top{args.topk}.csv"), "w", newline="") as f:
                w = csv.writer(f)
                w.writerow(["doc_id", "sent_ix", "max_label_score"])
                for r in top:
                    w.writerow([r["doc_id"], r["sent_ix"], r["score"]])
    if rank == 0:
        print(f"done! results in {out_dir}  ({int(res.keep.sum())} of {len(res.keep)} sentences selected on rank 0)")
    if dist is not None:
        dist.destroy_process_group()
    return res


if __name__ == "__main__":
    main()
