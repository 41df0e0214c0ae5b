"""Run a fine-tuned BERT(+Context) classifier over unlabeled note sentences — the MI355X drop-in
for /root/reference/sentclf/apply_bert_unlabeled.py.

    python -m clip_amd.apply_bert_unlabeled UNLABELED.csv MODEL_DIR LOCAL_WEIGHTS \
        [--threshold 0.05] [--n_context_sentences 2] [--task multilabel] [--topk 250000]

Same positional arguments and flags as the reference (:18-27); MODEL_DIR is a LOCAL directory with
config.json + vocab.txt (hub short names 'bert' / 'clinicalbert_disch' cannot resolve offline).
Writes `ttp_doc_sents.csv` next to LOCAL_WEIGHTS exactly as the reference does (:90,97-98,124: no
header, `doc_id,sent_ix,max_label_score` for rows with any label score > threshold, input order).
UNLABELED may also be a columnar directory written by an earlier run with `--save_rows DIR` (tokenise
once — `--tokenize_only` needs no GPU — and score many times; clip_amd/columnar.py).
With --topk K it also writes `ttp_top{K}.csv` WITH the header `doc_id,sent_ix,max_label_score`, the
file sentclf/pretrain_tasks.py:37,57-58 reads as `focus_fname` (README.md:43: "sorted by decreasing
max_label_score ... take the top 250k").

One process per GPU: under `torch.distributed.run` the rows are cut into chunks of whole documents,
chunk c belongs to rank c mod world, every rank scores its chunks with no communication and keeps a
running top-k; CSV parts are concatenated in chunk order (= input order) and the top-k uses one
all-gather.
"""
import argparse
import os
import queue
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np

import torch

from . import launch, topk as topk_mod
from .bert_model import args_to_model
from .fast_tokenizer import FastWordPiece, build_rows_chunks, build_rows_fast
from .config import BertConfig
from .constants import LABEL_TYPES
from .datasets import SentDataset, SentEvalDataset
from .columnar import ColumnarCorpus, graded_chunk_bounds, save_rows
from .job import RunningTopK, write_topk_csv
from .scoring import BulkScorer, PackedRows, write_ttp_csv


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("unlabeled_fname", type=str,
                        help="notes CSV (doc_id, sent_index, sentence[, labels]) or a columnar directory "
                             "written by --save_rows")
    parser.add_argument("model", type=str, help="local directory with config.json and vocab.txt")
    parser.add_argument("local_weights", type=str,
                        help="state-dict file (.pth / .bin / .safetensors) of the fine-tuned model")
    parser.add_argument("--threshold", type=float, default=0.05)
    parser.add_argument("--n_context_sentences", type=int, default=0)
    parser.add_argument("--task", choices=["binary", "multilabel"], default="multilabel")
    parser.add_argument("--topk", type=int, default=0, help="also write the global top-k file")
    parser.add_argument("--tokens_per_step", type=int, default=131072)
    parser.add_argument("--chunk_rows", type=int, default=1_000_000,
                        help="rows scored per chunk of whole documents (bounds host memory; chunk c -> rank c mod world)")
    parser.add_argument("--save_rows", type=str, default=None,
                        help="write the tokenised rows as a columnar directory (tokenise once, score many times)")
    parser.add_argument("--precision", choices=["bf16", "f16", "fp32"], default="bf16",
                        help="operand type of the GEMM / attention kernels (clip_set_precision)")
    parser.add_argument("--gpus", type=int, default=1,
                        help="ranks to start, one per MI355X (documents are sharded; ignored under torchrun)")
    parser.add_argument("--tokenize_only", action="store_true", help="stop after --save_rows (needs no GPU)")
    args = parser.parse_args(argv)
    import sys
    import time
    t_start = time.perf_counter()
    timing = os.environ.get("CLIP_AMD_TIMING") == "1"      # stage times on stderr (tools/cli_bench.py)

    def stamp(what):
        if timing:
            print(f"[clip_amd timing] {time.perf_counter() - t_start:7.3f} s  {what}", file=sys.stderr, flush=True)

    if args.task != "multilabel":
        raise NotImplementedError("only --task multilabel is on the accelerated path")

    # --gpus N: this process starts the N ranks itself (clip_amd/launch.py); under torchrun they exist already
    rc = launch.spawn_ranks_if_needed(args.gpus, module="clip_amd.apply_bert_unlabeled") if argv is None else None
    if rc is not None:
        raise SystemExit(rc)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    dist = device = None
    if not args.tokenize_only:
        dist, rank, world, device = launch.init_distributed()

    # ---- model + engine (:55-80), built on a helper thread while this one tokenises: reading the weights,
    # rounding them and allocating the workspace (HIP calls, which release the GIL) overlap the C++ tokenizer
    def build_model():
        config = BertConfig.from_pretrained(args.model, num_labels=len(LABEL_TYPES))       # :55-62
        model = args_to_model(args.n_context_sentences)(config, max_tokens=args.tokens_per_step,
                                                        max_seqs=max(64, args.tokens_per_step // 8))  # :64-70
        if args.local_weights.endswith(".safetensors"):
            from safetensors.torch import load_file
            sd = load_file(args.local_weights)
        else:
            sd = torch.load(args.local_weights, map_location="cpu", weights_only=True)       # :78
        stamp("weights file read")
        model.load_state_dict(sd)                                                          # :79
        model.set_precision(args.precision)
        model.to(device).eval()                                                            # :72,100
        stamp("state dict loaded")
        with torch.cuda.device(device):
            model._ensure_engine()
        stamp("engine ready")
        return model

    pool = ThreadPoolExecutor(max_workers=3)
    build = pool.submit(build_model) if not args.tokenize_only else None

    # ---- rows: a columnar directory as it is, or the CSV through the C++ front end — ONCE per job:
    # rank 0 tokenises on all host cores and writes the columnar directory, every rank memory-maps it
    # (N ranks tokenising the same CSV would be N x the work on the same cores)
    out_dir = os.path.dirname(args.local_weights) or "."                               # :90
    stream = None            # single rank, CSV input: (dataset, tokenizer, bounds) of the pipelined tokenizer
    if os.path.isdir(args.unlabeled_fname):
        corpus_dir = args.unlabeled_fname
        corpus = None
    else:
        corpus_dir = args.save_rows or (os.path.join(out_dir, "ttp_rows.columnar") if world > 1 else None)
        corpus = None
        if rank == 0:
            # C++ WordPiece (libclip_tok.so); the Python restatement behind it takes what it flags
            tokenizer = FastWordPiece(os.path.join(args.model, "vocab.txt"))                # :43-47
            if args.n_context_sentences == 0:                                              # :83-88
                dataset = SentDataset(args.unlabeled_fname, args.task)
            else:
                dataset = SentEvalDataset(args.unlabeled_fname, args.task,
                                          n_context_sentences=args.n_context_sentences)
            stamp(f"CSV read ({len(dataset)} rows)")
            meta = dict(n_context_sentences=args.n_context_sentences, sep_token_id=tokenizer.sep_token_id,
                        cls_token_id=tokenizer.cls_token_id, vocab_size=len(tokenizer.vocab),
                        source=os.path.basename(args.unlabeled_fname))
            if world == 1 and not args.tokenize_only:
                # one rank: tokenise in pieces of whole documents on a helper thread WHILE the GPU scores the
                # piece before (any chunking gives the same files: parts are concatenated in order and the
                # top-k is a running merge); the first pieces are small so that the GPU starts early
                stream = (dataset, tokenizer, graded_chunk_bounds(dataset.doc_ids, min(args.chunk_rows, 131_072)), meta)
            else:
                rows, doc_ids, sent_ixs = build_rows_fast(dataset, tokenizer, args.n_context_sentences)
                if corpus_dir:
                    save_rows(corpus_dir, rows, doc_ids, sent_ixs, **meta)
                if world == 1:
                    corpus = ColumnarCorpus.from_arrays(rows, doc_ids, sent_ixs, **meta)
        if dist is not None:
            dist.barrier()
    if corpus is None and stream is None:
        corpus = ColumnarCorpus(corpus_dir)
        if corpus.meta.get("n_context_sentences") != args.n_context_sentences:
            raise ValueError(f"{corpus_dir} was tokenised with n_context_sentences="
                             f"{corpus.meta.get('n_context_sentences')}, not {args.n_context_sentences}")
    sep_id = int((corpus.meta if corpus is not None else stream[3])["sep_token_id"])
    if args.tokenize_only:
        if not args.save_rows:
            raise ValueError("--tokenize_only needs --save_rows")
        print(f"tokenised {len(corpus)} rows / {corpus.meta['tokens']} tokens -> {args.save_rows}")
        return None

    model = build.result()
    if args.n_context_sentences > 0:
        model.set_sep_token_id(sep_id)                                                 # :74
        model.set_n_context_sentences(args.n_context_sentences)                        # :75
    model.set_task(args.task)                                                          # :80

    # ---- the hot loop (:92-124), chunk by chunk: score, append this chunk's CSV part, fold into the
    # running top-k.  Chunks are whole documents; chunk c belongs to rank c mod world.
    scorer = BulkScorer(model, any_threshold=args.threshold)
    out = os.path.join(out_dir, "ttp_doc_sents.csv")
    def chunk_source():
        if stream is None:
            for c, a, b, item in corpus.chunks(args.chunk_rows, rank, world):
                yield (c,) + item
            return
        dataset, tokenizer, bounds, meta = stream
        q = queue.Queue(maxsize=2)
        stop = threading.Event()        # the consumer gave up (an exception in scoring): the producer must not block forever

        def put(item):
            while not stop.is_set():
                try:
                    q.put(item, timeout=0.2)
                    return True
                except queue.Full:
                    pass
            return False

        def produce():
            try:
                for c, item in enumerate(build_rows_chunks(dataset, tokenizer, args.n_context_sentences, bounds)):
                    if not put((c,) + item):
                        return
                put(None)
            except BaseException as exc:      # surfaces in the consumer
                put(exc)

        pool.submit(produce)
        kept = []
        try:
            while True:
                item = q.get()
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                if args.save_rows:
                    kept.append(item[1:])
                yield item
        finally:
            stop.set()
        if args.save_rows:
            lens = np.concatenate([k[0].lens for k in kept]) if kept else np.zeros(0, np.int64)
            rows = PackedRows(np.concatenate([k[0].flat for k in kept]) if kept else np.zeros(0, np.int32),
                              np.concatenate(([0], np.cumsum(lens, dtype=np.int64))))
            save_rows(args.save_rows, rows, np.concatenate([k[1] for k in kept]), np.concatenate([k[2] for k in kept]), **meta)

    n_chunks = len(stream[2]) if stream is not None else len(corpus.chunk_bounds(args.chunk_rows))
    top = RunningTopK(args.topk) if args.topk > 0 else None
    counts = [0, 0]          # rows, kept
    res = None
    writes = []
    post = ThreadPoolExecutor(max_workers=1)       # per-chunk host work, in chunk order, off the scoring thread

    def fold(r):
        if top is not None:
            top.update(r.doc_ids, r.sent_ixs, r.max_score, keep=r.keep)
        counts[0] += len(r.keep)
        counts[1] += int(r.keep.sum())

    for c, rows_c, d_c, s_c in chunk_source():
        stamp(f"chunk {c} tokenised ({len(d_c)} rows)")
        res = scorer.score(rows_c, d_c, s_c)
        stamp(f"chunk {c} scored")
        # formatted / folded into the running top-k while the next chunk is scored
        writes.append(pool.submit(write_ttp_csv, f"{out}.chunk{c}", res))
        writes.append(post.submit(fold, res))
    for w in writes:
        w.result()
    pool.shutdown()
    post.shutdown()
    n_rows, n_kept = counts
    stamp("CSV parts written")
    if dist is not None:
        dist.barrier()
    if rank == 0:                                       # parts in chunk order = input order (:97-98,124)
        with open(out, "wb") as f:                      # bytes: csv.writer's '\r\n' row ends stay as the reference writes them
            for c in range(n_chunks):
                with open(f"{out}.chunk{c}", "rb") as g:
                    f.write(g.read())
                os.remove(f"{out}.chunk{c}")
    if top is not None:
        glob = topk_mod.global_topk(top.records, args.topk)
        if rank == 0:
            write_topk_csv(os.path.join(out_dir, f"ttp_top{args.topk}.csv"), glob)
    tot = torch.tensor([n_kept, n_rows], dtype=torch.int64,
                       device=device if dist is None or dist.get_backend() == "nccl" else "cpu")
    if dist is not None:
        dist.all_reduce(tot)                            # counts over all ranks
    stamp("files complete")
    if rank == 0:
        print(f"done! results in {out_dir}  ({int(tot[0])} of {int(tot[1])} sentences selected, {world} rank(s))")
    if dist is not None:
        dist.destroy_process_group()
    return res


if __name__ == "__main__":
    main()
