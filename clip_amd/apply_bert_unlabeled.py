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

One process per GPU (`--gpus N`, or under `torch.distributed.run`): EVERY rank reads the CSV's document index,
takes a contiguous range of whole documents balanced by tokens (exact counts for a columnar directory; for CSV
text the cells' byte lengths over each row's context window stand in for them), and runs the same pipeline
on its own range — tokenise a piece on a helper thread while the GPU scores the piece before — so no rank
waits for another before its first step and none tokenises for another.  CSV parts are concatenated in
(rank, piece) order = input order and the top-k uses one all-gather.
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
from .columnar import ColumnarCorpus, chunk_bounds, graded_chunk_bounds, merge_parts, save_rows
from .job import RunningTopK, write_topk_csv
from .scoring import BulkScorer, PackedRows, write_ttp_csv


def token_proxy(dataset, n):
    """A weight per row that follows its token count without tokenising: the bytes of the sentence cells of the
    row's context window — the +-n rows that belong to the same document (neural_baselines.py:219-235), a
    constant for a <DOC_START> / <DOC_END> stand-in — plus a constant for [CLS] and the [SEP]s."""
    N = len(dataset)
    cell = dataset.sents["sentence"].str.len().to_numpy(dtype=np.int64, na_value=0)
    doc = np.asarray(dataset.doc_ids)
    i = np.arange(N)
    w = cell + 8 * (2 * n + 2)
    for o in range(-n, n + 1):
        if o == 0:
            continue
        j = np.clip(i + o, 0, N - 1)
        same = (i + o >= 0) & (i + o < N) & (doc[j] == doc)
        w = w + np.where(same, cell[j], 40)
    return w


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
    parser.add_argument("--precision", choices=["auto", "bf16", "f16", "f16r", "fp32"], default="auto",
                        help="numerics setting of the kernels (clip_set_precision); auto = the fastest 16-bit setting that "
                             "measures inside 2e-2 on the logits and 99.9 %% label agreement against the fp32 check mode on a "
                             "sample of this input (clip_amd/precision.py)")
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
            print(f"[clip_amd timing] rank {os.environ.get('RANK', '0')} {time.perf_counter() - t_start:7.3f} s  {what}",
                  file=sys.stderr, flush=True)

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
        if args.precision != "auto":      # (auto: the engine is built after the calibration on this input's rows)
            with torch.cuda.device(device):
                model._ensure_engine()
            stamp("engine ready")
        return model

    pool = ThreadPoolExecutor(max_workers=3)
    build = pool.submit(build_model) if not args.tokenize_only else None

    # ---- rows.  A columnar directory as it is; a CSV through the C++ front end.  Every rank reads the document
    # index (doc_id per row) and cuts the SAME rank ranges from it — contiguous, whole documents, balanced by
    # tokens — so each rank knows its own range and every other rank's piece count without a message.
    out_dir = os.path.dirname(args.local_weights) or "."                               # :90
    n = args.n_context_sentences
    corpus = dataset = tokenizer = meta = None
    if os.path.isdir(args.unlabeled_fname):
        corpus = ColumnarCorpus(args.unlabeled_fname)
        if corpus.meta.get("n_context_sentences") != n:
            raise ValueError(f"{args.unlabeled_fname} was tokenised with n_context_sentences="
                             f"{corpus.meta.get('n_context_sentences')}, not {n}")
        doc_index, weights = corpus.doc_ids, np.diff(corpus.offsets)                       # exact token counts
        sep_id = int(corpus.meta["sep_token_id"])
        piece_rows = args.chunk_rows
        cut = chunk_bounds
    else:
        # C++ WordPiece (libclip_tok.so); the Python restatement behind it takes what it flags
        tokenizer = FastWordPiece(os.path.join(args.model, "vocab.txt"))                    # :43-47
        if n == 0:                                                                         # :83-88
            dataset = SentDataset(args.unlabeled_fname, args.task)
        else:
            dataset = SentEvalDataset(args.unlabeled_fname, args.task, n_context_sentences=n)
        stamp(f"CSV read ({len(dataset)} rows)")
        meta = dict(n_context_sentences=n, sep_token_id=tokenizer.sep_token_id, cls_token_id=tokenizer.cls_token_id,
                    vocab_size=len(tokenizer.vocab), source=os.path.basename(args.unlabeled_fname))
        doc_index, weights = dataset.doc_ids, token_proxy(dataset, n)
        sep_id = int(tokenizer.sep_token_id)
        # pieces of whole documents tokenised on a helper thread WHILE the GPU scores the piece before (any
        # chunking gives the same files: parts are concatenated in order and the top-k is a running merge);
        # the first pieces are small so that the GPU starts early
        piece_rows = min(args.chunk_rows, 131_072)
        cut = graded_chunk_bounds
    N = len(doc_index)

    def rank_bounds(r):
        a, b = topk_mod.shard_documents(doc_index, weights, world, r)
        return [(a + x, a + y) for x, y in cut(doc_index[a:b], piece_rows)]
    bounds = rank_bounds(rank)
    stamp(f"rows {bounds[0][0] if bounds else 0}..{bounds[-1][1] if bounds else 0} of {N} in {len(bounds)} pieces")

    if args.tokenize_only:
        if not args.save_rows or corpus is not None:
            raise ValueError("--tokenize_only needs a CSV and --save_rows")
        rows, doc_ids, sent_ixs = build_rows_fast(dataset, tokenizer, n)
        save_rows(args.save_rows, rows, doc_ids, sent_ixs, **meta)
        print(f"tokenised {len(rows)} rows / {int(rows.offsets[-1])} tokens -> {args.save_rows}")
        return None

    def rows_of(block_bounds):
        """(PackedRows, doc_ids, sent_ixs) for each (a, b) row range, from either kind of input"""
        if corpus is not None:
            return (corpus.slice(a, b) for a, b in block_bounds)
        return build_rows_chunks(dataset, tokenizer, n, block_bounds)

    model = build.result()
    if args.n_context_sentences > 0:
        model.set_sep_token_id(sep_id)                                                 # :74
        model.set_n_context_sentences(args.n_context_sentences)                        # :75
    model.set_task(args.task)                                                          # :80
    if args.precision == "auto" and N:
        # measured on 8 runs of 256 rows spread over the WHOLE input — the same sample on every rank, so every
        # rank reaches the same setting without a message
        per = min(256, max(N // 8, 1))
        starts = sorted({int(x) for x in (np.arange(8) * (N - per) // 7)})
        blocks = [r for r, _, _ in rows_of([(a, min(a + per, N)) for a in starts])]
        lens = np.concatenate([r.lens for r in blocks]).astype(np.int64)
        sample = PackedRows(np.concatenate([r.flat for r in blocks]), np.concatenate(([0], np.cumsum(lens))))
        model.calibrate_precision(sample, log=sys.stderr if rank == 0 else False)
        stamp(f"precision auto -> {model.resolved_precision}")

    # ---- the hot loop (:92-124), piece by piece over this rank's range: score, write the piece's CSV part,
    # fold into the running top-k.  Pieces are whole documents.
    scorer = BulkScorer(model, any_threshold=args.threshold)
    out = os.path.join(out_dir, "ttp_doc_sents.csv")
    kept = []                # --save_rows: this rank's pieces

    def chunk_source():
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
                for c, item in enumerate(rows_of(bounds)):
                    if not put((c,) + item):
                        return
                put(None)
            except BaseException as exc:      # surfaces in the consumer
                put(exc)

        pool.submit(produce)
        try:
            while True:
                item = q.get()
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                if args.save_rows and corpus is None:
                    kept.append(item[1:])
                yield item
        finally:
            stop.set()

    top = RunningTopK(args.topk) if args.topk > 0 else None
    counts = [0, 0, 0]       # rows, kept, tokens
    res = None
    writes = []
    post = ThreadPoolExecutor(max_workers=1)       # per-chunk host work, in chunk order, off the scoring thread

    def fold(r):
        if top is not None:
            top.update(r.doc_ids, r.sent_ixs, r.max_score, keep=r.keep)
        counts[0] += len(r.keep)
        counts[1] += int(r.keep.sum())

    part = lambda r, c: f"{out}.r{r:04d}.chunk{c:06d}"
    for c, rows_c, d_c, s_c in chunk_source():
        stamp(f"piece {c} ready ({len(d_c)} rows)")
        res = scorer.score(rows_c, d_c, s_c)
        counts[2] += int(rows_c.offsets[-1])
        stamp(f"piece {c} scored")
        # formatted / folded into the running top-k while the next piece is scored
        writes.append(pool.submit(write_ttp_csv, part(rank, c), res))
        writes.append(post.submit(fold, res))
    for w in writes:
        w.result()
    pool.shutdown()
    post.shutdown()
    n_rows, n_kept, n_tokens = counts
    stamp(f"CSV parts written ({n_rows} rows, {n_tokens} tokens on this rank)")
    if args.save_rows and corpus is None:               # this rank's rows; rank 0 joins the parts below
        lens = np.concatenate([k[0].lens for k in kept]).astype(np.int64) if kept else np.zeros(0, np.int64)
        mine = PackedRows(np.concatenate([k[0].flat for k in kept]) if kept else np.zeros(0, np.int32),
                          np.concatenate(([0], np.cumsum(lens))))
        save_rows(args.save_rows if world == 1 else f"{args.save_rows}.part{rank:04d}", mine,
                  np.concatenate([k[1] for k in kept]) if kept else np.zeros(0, np.int64),
                  np.concatenate([k[2] for k in kept]) if kept else np.zeros(0, np.int64), **meta)
    if dist is not None:
        dist.barrier()
    if rank == 0:                                       # parts in (rank, piece) order = input order (:97-98,124)
        with open(out, "wb") as f:                      # bytes: csv.writer's '\r\n' row ends stay as the reference writes them
            for r in range(world):
                for c in range(len(bounds if r == rank else rank_bounds(r))):
                    with open(part(r, c), "rb") as g:
                        f.write(g.read())
                    os.remove(part(r, c))
        if args.save_rows and corpus is None and world > 1:
            merge_parts([f"{args.save_rows}.part{r:04d}" for r in range(world)], args.save_rows, **meta)
    if top is not None:
        glob = topk_mod.global_topk(top.records, args.topk)
        if rank == 0:
            write_topk_csv(os.path.join(out_dir, f"ttp_top{args.topk}.csv"), glob)
    tot = torch.tensor([n_kept, n_rows, n_tokens], dtype=torch.int64,
                       device=device if dist is None or dist.get_backend() == "nccl" else "cpu")
    if dist is not None:
        dist.all_reduce(tot)                            # counts over all ranks
    stamp("files complete")
    if timing:
        print(f"[clip_amd timing] rank {rank}: {n_tokens} tokens = {100.0 * n_tokens * world / max(int(tot[2]), 1):.2f} % "
              "of an even share", file=sys.stderr, flush=True)
    if rank == 0:
        print(f"done! results in {out_dir}  ({int(tot[0])} of {int(tot[1])} sentences selected, {world} rank(s), "
              f"precision {model.resolved_precision})")
    if dist is not None:
        dist.destroy_process_group()
    return res


if __name__ == "__main__":
    main()
