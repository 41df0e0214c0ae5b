"""Full-size synthetic scoring jobs (BASELINE.json configs C2 / C4 / C5) end to end on one node:

    python -m clip_amd.job --config C4 --rows 10000000 --topk 250000 [--gpus 8]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        -m clip_amd.job --config C5 --rows 50000000 --topk 250000 --out ttp_top250000.csv

The loop of /root/reference/sentclf/apply_bert_unlabeled.py:92-124 at NOTEEVENTS scale: rows arrive
in CHUNKS of whole documents (here generated, MIMIC being access-controlled; a real run streams the
columnar file of clip_amd.collate.build_rows), chunk c belongs to rank c % world (documents are the
sharding unit), a rank scores its chunks with no communication and keeps only a
RUNNING top-k (k records, 16 B each: memory does not grow with the corpus), and the job ends with
the one all-gather + deterministic merge of clip_amd.topk.  Chunk contents depend on (seed, chunk
index) only, so the global top-k is the same for every world size — the property
tests/test_gpu_job.py checks.

Checkpoint / resume (the reference has none: apply_bert_unlabeled.py:97 reopens its output with 'w' and
starts over): `--state FILE` keeps (next chunk, running top-k, counters, seconds so far) in one .npz per
rank, rewritten after every chunk; `--max-seconds S` ends an invocation cleanly after the chunk that crosses
S.  The next invocation with the same arguments continues from the file; the JSON line of the invocation
that finishes the job reports the totals over all of them ("invocations", "seconds" = scoring time summed).
A job split this way selects the same top-k, bit for bit, as one uninterrupted run (tests/test_gpu_job.py).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

from . import synth, topk
from .scoring import BulkScorer, PackedRows

CONFIGS = {   # name -> (n_ctx, max_len, sigma, seg_hi, seed)      SURVEY.md §8d
    "C2": (0, 64, 0.6, 62, 1001),
    "C3": (2, 256, 0.6, 62, 1002),
    "C4": (2, 256, 0.6, 62, 1004),
    "C5": (2, 512, 0.9, 200, 1005),
}
DOC_STRIDE = 1 << 24      # doc ids of chunk c start at c * DOC_STRIDE: unique without coordination


def chunk_rows(config, chunk_index, n_rows, vocab_size):
    """Rows of one chunk: (PackedRows, doc_ids int64, sent_ixs int64); documents of 40..400 sentences."""
    n_ctx, max_len, sigma, seg_hi, seed = CONFIGS[config]
    flat, off = synth.packed_context_rows(n_rows, n_ctx, vocab_size, max_len, seed=seed * 1000003 + chunk_index,
                                          sigma=sigma, seg_hi=seg_hi)
    d, s = synth.documents(n_rows, seed=seed * 7919 + chunk_index)
    return PackedRows(flat, off), d + chunk_index * DOC_STRIDE, s.astype(np.int64)


class RunningTopK:
    """Best k records seen so far under the total order of clip_amd.topk (bounded memory)."""

    def __init__(self, k):
        self.k = int(k)
        self.records = np.empty(0, topk.RECORD)

    def update(self, doc_ids, sent_ixs, scores, keep=None):
        new = topk.local_topk(doc_ids, sent_ixs, scores, self.k, keep=keep)
        self.records = topk.merge_topk([self.records, new], self.k)
        return self


class JobState:
    """What outlives a chunk on one rank: position in the rank's chunk list, the running top-k, counters and the
    scoring seconds of earlier invocations.  One .npz, replaced atomically after every chunk."""

    def __init__(self, k, signature):
        self.next_i, self.top = 0, RunningTopK(k)
        self.rows = self.tokens = self.kept = 0
        self.seconds, self.invocations = 0.0, 1
        self.signature = signature

    @classmethod
    def load_or_new(cls, path, k, signature):
        st = cls(k, signature)
        if path and os.path.isfile(path):
            z = np.load(path, allow_pickle=False)
            if str(z["signature"]) != signature:
                raise ValueError(f"{path} belongs to another job ({z['signature']}), not {signature}")
            st.next_i, st.rows, st.tokens, st.kept = (int(z[k_]) for k_ in ("next_i", "rows", "tokens", "kept"))
            st.seconds, st.invocations = float(z["seconds"]), int(z["invocations"]) + 1
            st.top.records = z["records"].view(topk.RECORD).reshape(-1).copy()
        return st

    def save(self, path, seconds_now):
        tmp = path + ".tmp.npz"
        np.savez(tmp, signature=np.array(self.signature), next_i=self.next_i, rows=self.rows, tokens=self.tokens,
                 kept=self.kept, seconds=self.seconds + seconds_now, invocations=self.invocations,
                 records=self.top.records.view(np.uint8))
        os.replace(tmp, path)


def run_rank(scorer, config, n_rows, chunk_rows_n, k, rank, world, vocab_size, on_chunk=None, state=None,
             state_path=None, max_seconds=None):
    """Score this rank's chunks (from state.next_i on); returns (JobState, finished).  With state_path the state
    file is rewritten after every chunk; with max_seconds the loop ends after the chunk that crosses it."""
    from concurrent.futures import ThreadPoolExecutor
    n_chunks = (n_rows + chunk_rows_n - 1) // chunk_rows_n
    st = state or JobState(k, "")
    top = st.top
    mine = list(range(rank, n_chunks, world))
    size = lambda c: min(chunk_rows_n, n_rows - c * chunk_rows_n)
    pool = ThreadPoolExecutor(max_workers=1)      # the next chunk's rows are made while this one is scored
    post = ThreadPoolExecutor(max_workers=1)      # ... and the previous chunk's records are folded into the top-k
    i0 = st.next_i
    nxt = pool.submit(chunk_rows, config, mine[i0], size(mine[i0]), vocab_size) if i0 < len(mine) else None
    fold = None
    t0 = time.perf_counter()
    finished = True
    for i in range(i0, len(mine)):
        c = mine[i]
        n = size(c)
        rows, doc_ids, sent_ixs = nxt.result()
        stop = max_seconds is not None and time.perf_counter() - t0 > max_seconds and i > i0
        if stop:
            finished = False
            break
        if i + 1 < len(mine):
            nxt = pool.submit(chunk_rows, config, mine[i + 1], size(mine[i + 1]), vocab_size)
        res = scorer.score(rows, doc_ids, sent_ixs)
        if fold is not None:
            fold.result()
        fold = post.submit(top.update, res.doc_ids, res.sent_ixs, res.max_score, res.keep)   # under the next chunk
        st.rows += n
        st.tokens += int(rows.lens.sum())
        st.kept += int(res.keep.sum())
        st.next_i = i + 1
        if state_path:
            fold.result()                          # the file must hold the top-k INCLUDING this chunk
            st.save(state_path, time.perf_counter() - t0)
        if on_chunk is not None:
            on_chunk(c, n, res)
    if fold is not None:
        fold.result()
    pool.shutdown(wait=False)
    post.shutdown(wait=False)
    return st, finished


def write_topk_csv(path, records):
    """The focus file of task-targeted pre-training: header doc_id,sent_ix as
    sentclf/pretrain_tasks.py:37,57-58 reads it, score last."""
    from .scoring import format_score_rows
    with open(path, "w", newline="") as f:      # csv.writer's bytes (header, '\r\n' row ends), formatted by arrays
        f.write("doc_id,sent_ix,max_label_score\r\n")
        f.write(format_score_rows(records["doc_id"], records["sent_ix"], records["score"]))


def main(argv=None):
    from .bert_model import args_to_model
    from .config import BertConfig
    from .weights import procedural_state_dict
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", choices=sorted(CONFIGS), default="C4")
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--chunk-rows", type=int, default=250_000)
    ap.add_argument("--topk", type=int, default=250_000)
    ap.add_argument("--threshold", type=float, default=0.05)
    ap.add_argument("--tokens-per-step", type=int, default=131072)
    ap.add_argument("--weights", type=str, default=None, help="state dict (.safetensors); default: procedural")
    ap.add_argument("--precision", choices=["auto", "bf16", "f16", "f16r", "fp32"], default="auto",
                    help="numerics setting (clip_set_precision); auto: measured on a sample (clip_amd/precision.py)")
    ap.add_argument("--state", type=str, default=None,
                    help="checkpoint file (one per rank: FILE.rankR.npz), rewritten after every chunk; an existing one is resumed")
    ap.add_argument("--max-seconds", type=float, default=None,
                    help="end this invocation after the chunk that crosses this many seconds of scoring (needs --state)")
    ap.add_argument("--gpus", type=int, default=1, help="ranks to start, one per MI355X (ignored under torchrun)")
    ap.add_argument("--out", type=str, default=None, help="write the global top-k CSV here (rank 0)")
    ap.add_argument("--progress", action="store_true", help="one line per scored chunk on stderr")
    args = ap.parse_args(argv)

    # --gpus N: this process starts the N ranks itself (clip_amd/launch.py); under torchrun they exist already
    from . import launch
    rc = launch.spawn_ranks_if_needed(args.gpus, module="clip_amd.job") if argv is None else None
    if rc is not None:
        raise SystemExit(rc)
    dist, rank, world, device = launch.init_distributed()

    cfg = BertConfig()
    n_ctx = CONFIGS[args.config][0]
    model = args_to_model(n_ctx)(cfg, max_tokens=args.tokens_per_step, max_seqs=max(64, args.tokens_per_step // 4))
    if args.weights:
        from safetensors.torch import load_file
        model.load_state_dict(load_file(args.weights))
    else:
        model.load_state_dict(procedural_state_dict(cfg, 0))
    model.set_precision(args.precision)
    model.to(device).eval()
    if n_ctx > 0:
        model.set_sep_token_id(102)
        model.set_n_context_sentences(n_ctx)
    model.set_task("multilabel")
    scorer = BulkScorer(model, any_threshold=args.threshold)
    warm, _, _ = chunk_rows(args.config, 0, 2048, cfg.vocab_size)
    scorer.score(warm)                                    # engine build + first launches, untimed

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    barrier()
    t0 = time.perf_counter()
    def progress(c, n, res):
        print(f"[rank {rank}] chunk {c}: {n} rows, {time.perf_counter() - t0:.1f} s", file=sys.stderr, flush=True)

    if args.max_seconds is not None and not args.state:
        raise SystemExit("--max-seconds needs --state")
    signature = f"{args.config}/{args.rows}/{args.chunk_rows}/{args.topk}/{args.threshold}/{world}/{model.resolved_precision}"
    state_path = f"{args.state}.rank{rank}.npz" if args.state else None
    state = JobState.load_or_new(state_path, args.topk, signature)
    state, finished = run_rank(scorer, args.config, args.rows, args.chunk_rows, args.topk, rank, world,
                               cfg.vocab_size, on_chunk=progress if args.progress else None, state=state,
                               state_path=state_path, max_seconds=args.max_seconds)
    torch.cuda.synchronize()
    t_score = time.perf_counter() - t0
    fin = torch.tensor([int(finished)], dtype=torch.int64,
                       device=device if dist is None or dist.get_backend() == "nccl" else "cpu")
    if dist is not None:
        dist.all_reduce(fin, op=dist.ReduceOp.MIN)
    if not int(fin.item()):
        if rank == 0:
            import resource
            print(json.dumps({"job": args.config, "finished": False, "state": args.state,
                              "rows_so_far_rank0": state.rows, "seconds_this_invocation": round(t_score, 3),
                              "seconds_so_far": round(t_score + state.seconds, 3), "invocations": state.invocations,
                              "peak_host_rss_mb_rank0": round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0, 1)}),
                  flush=True)
        if dist is not None:
            dist.destroy_process_group()
        return None
    top, rows_done, tokens, kept = state.top, state.rows, state.tokens, state.kept
    glob = topk.global_topk(top.records, args.topk)
    barrier()
    t_all = time.perf_counter() - t0 + state.seconds
    tot = torch.tensor([rows_done, tokens, kept], dtype=torch.int64,
                       device=device if dist is None or dist.get_backend() == "nccl" else "cpu")
    if dist is not None:
        dist.all_reduce(tot)
    if rank == 0:
        import resource
        if args.out:
            write_topk_csv(args.out, glob)
        print(json.dumps({
            "job": args.config, "rows": int(tot[0]), "tokens": int(tot[1]), "kept": int(tot[2]), "n_gpus": world,
            "backend": dist.get_backend() if dist is not None else None,
            "precision": model.resolved_precision, "invocations": state.invocations,
            "peak_host_rss_mb_rank0": round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0, 1),
            "seconds": round(t_all, 3), "rank0_scoring_seconds": round(t_score + state.seconds, 3),
            "sentences_per_s": round(int(tot[0]) / t_all, 1), "tokens_per_s": round(int(tot[1]) / t_all, 1),
            "topk": int(len(glob)), "topk_first": [int(glob[0]["doc_id"]), int(glob[0]["sent_ix"]), float(glob[0]["score"])]
            if len(glob) else None,
            "topk_last_score": float(glob[-1]["score"]) if len(glob) else None,
            "data": "synthetic (clip_amd.synth.packed_context_rows), procedural weights, host generation + "
                    "bucketing + H2D + D2H inside the timed region"}), flush=True)
    if dist is not None:
        dist.destroy_process_group()
    return glob


if __name__ == "__main__":
    main()
