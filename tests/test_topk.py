"""Deterministic top-k: ordering with forced ties, shard-merge equivalence, and the N>1 path over
gloo with world_size 2 (the GPU path uses the same code over RCCL)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from clip_amd import topk
from oracle import bert_oracle as O


def _records(n, seed, n_scores=50):
    rng = np.random.default_rng(seed)
    doc = np.sort(rng.integers(0, n // 8 + 1, n)).astype(np.int64)
    sent = np.zeros(n, np.int64)
    for d in np.unique(doc):
        m = doc == d
        sent[m] = np.arange(m.sum())
    # few distinct scores -> many ties, as bf16-ish scores near 0.5 produce
    score = (rng.integers(0, n_scores, n) / n_scores).astype(np.float32)
    return doc, sent, score


@pytest.mark.parametrize("n,k", [(1000, 100), (1000, 1000), (1000, 5000), (37, 1), (0, 10)])
def test_local_topk_equals_oracle_sort(n, k):
    doc, sent, score = _records(n, n + k)
    rec = topk.local_topk(doc, sent, score, k)
    d, s, c = O.topk_records(doc, sent, score, k)
    assert rec["doc_id"].tolist() == d.tolist()
    assert rec["sent_ix"].tolist() == s.tolist()
    assert np.array_equal(rec["score"], c)


def test_keep_mask_filters_before_ranking():
    doc, sent, score = _records(500, 3)
    keep = score > 0.3
    rec = topk.local_topk(doc, sent, score, 50, keep=keep)
    d, s, c = O.topk_records(doc[keep], sent[keep], score[keep], 50)
    assert rec["doc_id"].tolist() == d.tolist() and rec["sent_ix"].tolist() == s.tolist()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_merge_of_shard_topk_is_global_topk(world):
    doc, sent, score = _records(4000, world)
    lens = np.full(len(doc), 20)
    k = 300
    parts = []
    covered = 0
    for r in range(world):
        a, b = topk.shard_documents(doc, lens, world, r)
        assert a == covered
        covered = b
        if a > 0 and a < len(doc):
            assert doc[a] != doc[a - 1]          # whole documents only
        parts.append(topk.local_topk(doc[a:b], sent[a:b], score[a:b], k))
    assert covered == len(doc)
    merged = topk.merge_topk(parts, k)
    ref = topk.local_topk(doc, sent, score, k)
    assert np.array_equal(merged, ref)


def test_record_is_16_bytes():
    assert topk.RECORD.itemsize == 16


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, k, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    doc, sent, score = _records(3000, 99)
    lens = np.full(len(doc), 17)
    a, b = topk.shard_documents(doc, lens, world, rank)
    local = topk.local_topk(doc[a:b], sent[a:b], score[a:b], k)
    glob = topk.global_topk(local, k)
    np.save(os.path.join(out_dir, f"r{rank}.npy"), glob)
    dist.destroy_process_group()


def test_global_topk_over_gloo_world2(tmp_path):
    k, world = 250, 2
    mp.spawn(_worker, args=(world, _free_port(), k, str(tmp_path)), nprocs=world, join=True)
    doc, sent, score = _records(3000, 99)
    ref = topk.local_topk(doc, sent, score, k)
    for r in range(world):
        got = np.load(tmp_path / f"r{r}.npy")
        assert np.array_equal(got, ref)          # bit-exact, identical on every rank


def test_write_topk_csv_is_csv_writer_bytes(tmp_path):
    """the focus file (pretrain_tasks.py:37,57-58 reads doc_id / sent_ix columns): same bytes as csv.writer
    over [doc_id, sent_ix, np.float32 score] rows"""
    import csv
    from clip_amd.job import write_topk_csv
    rng = np.random.default_rng(3)
    rec = np.empty(2000, topk.RECORD)
    rec["doc_id"] = rng.integers(-5, 10**12, 2000)
    rec["sent_ix"] = rng.integers(0, 4000, 2000)
    rec["score"] = np.concatenate((rng.random(1990, dtype=np.float32),
                                   np.asarray([0, 1, 1e-8, 0.5, 0.1, 0.30000001, 1e-45, 0.999999, 3e-5, 0.05], np.float32)))
    write_topk_csv(str(tmp_path / "a.csv"), rec)
    with open(tmp_path / "b.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["doc_id", "sent_ix", "max_label_score"])
        for r in rec:
            w.writerow([r["doc_id"], r["sent_ix"], r["score"]])
    assert (tmp_path / "a.csv").read_bytes() == (tmp_path / "b.csv").read_bytes()
    write_topk_csv(str(tmp_path / "e.csv"), rec[:0])
    assert (tmp_path / "e.csv").read_bytes() == b"doc_id,sent_ix,max_label_score\r\n"
