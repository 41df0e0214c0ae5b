"""Deterministic top-k of (doc_id, sent_ix, max_label_score) records and its multi-GPU merge.

The reference has no top-k code: README.md:43 says the scored sentences are "sorted by decreasing
max_label_score" and the first k (250k / 500k / 1M / 2M, sentclf/train_pretrained_*.sh) feed
task-targeted pre-training.  Build-defined total order, so ties cannot depend on rank count or
arrival order:  score descending, then doc_id ascending, then sent_ix ascending.

Multi-GPU: documents are sharded over ranks, scoring exchanges nothing; every
rank reduces its shard to a local top-k (any member of the global top-k is in its rank's local
top-k), ONE all_gather (RCCL over xGMI with backend "nccl"; gloo in the CPU tests) moves the
fixed-size 16-byte records (int64 doc_id, int32 sent_ix, float32 score), and every rank runs the
same merge.
"""
import numpy as np
import torch

RECORD = np.dtype([("doc_id", "<i8"), ("sent_ix", "<i4"), ("score", "<f4")])   # 16 bytes


def _order(doc_ids, sent_ixs, scores):
    # lexsort: last key is primary. -score in float64 is exact for float32 inputs.
    return np.lexsort((sent_ixs, doc_ids, -scores.astype(np.float64)))


def local_topk(doc_ids, sent_ixs, scores, k, keep=None):
    """Top-k records of one shard, sorted by the total order. O(N) selection + O(k log k) sort."""
    doc_ids = np.asarray(doc_ids, np.int64)
    sent_ixs = np.asarray(sent_ixs, np.int64)
    scores = np.asarray(scores, np.float32)
    if keep is not None:
        sel = np.nonzero(keep)[0]
        doc_ids, sent_ixs, scores = doc_ids[sel], sent_ixs[sel], scores[sel]
    n = len(scores)
    if n > k > 0:
        kth = np.partition(scores, n - k)[n - k]          # k-th largest score
        cand = np.nonzero(scores >= kth)[0]               # every tie of the k-th score stays a candidate
        doc_ids, sent_ixs, scores = doc_ids[cand], sent_ixs[cand], scores[cand]
    o = _order(doc_ids, sent_ixs, scores)[:max(k, 0)]
    rec = np.empty(len(o), RECORD)
    rec["doc_id"], rec["sent_ix"], rec["score"] = doc_ids[o], sent_ixs[o], scores[o]
    return rec


def merge_topk(record_lists, k):
    rec = np.concatenate(record_lists) if len(record_lists) else np.empty(0, RECORD)
    o = _order(rec["doc_id"], rec["sent_ix"].astype(np.int64), rec["score"])[:k]
    return rec[o]


def global_topk(local_records, k, group=None, device=None, always_collective=False):
    """All-gather every rank's local top-k (padded to k records) and merge; identical on all ranks.
    Without an initialised process group this is the single-rank identity; so is a group of ONE rank unless
    `always_collective` (the one-GPU rehearsal of the RCCL leg: tests/test_gpu_multirank.py)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not always_collective):
        return merge_topk([local_records], k)
    world = dist.get_world_size(group)
    backend = dist.get_backend(group)
    dev = device if device is not None else (torch.device("cuda", torch.cuda.current_device())
                                             if backend == "nccl" else torch.device("cpu"))
    # ONE collective: k records + one trailer record whose doc_id field carries this rank's count
    buf = np.zeros(k + 1, RECORD)
    n = min(len(local_records), k)
    buf[:n] = local_records[:n]
    buf[k]["doc_id"] = n
    payload = torch.from_numpy(buf.view(np.uint8).reshape(k + 1, RECORD.itemsize).copy()).to(dev)
    payloads = torch.empty((world * (k + 1), RECORD.itemsize), dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(payloads, payload, group=group)
    recs = payloads.cpu().numpy().reshape(-1).view(RECORD).reshape(world, k + 1)
    parts = [recs[r, :int(recs[r, k]["doc_id"])].copy() for r in range(world)]
    return merge_topk(parts, k)


def shard_documents(doc_ids, lens, world, rank):
    """Contiguous ranges of WHOLE documents, balanced by token count, so a sentence's +-n context
    (sentclf/neural_baselines.py:219-235) never crosses a shard. Returns (start, stop) row range.
    Rows of one document must be contiguous in the input (they are in the reference's CSVs)."""
    doc_ids = np.asarray(doc_ids)
    n = len(doc_ids)
    if n == 0:
        return 0, 0
    starts = np.concatenate(([0], np.nonzero(doc_ids[1:] != doc_ids[:-1])[0] + 1))   # first row of each doc
    ctok = np.concatenate(([0], np.cumsum(np.asarray(lens, np.int64))))
    total = ctok[-1]
    bounds = [0]
    for r in range(1, world):
        target = total * r / world
        j = int(np.argmin(np.abs(ctok[starts] - target)))
        bounds.append(max(int(starts[j]), bounds[-1]))
    bounds.append(n)
    return bounds[rank], bounds[rank + 1]
