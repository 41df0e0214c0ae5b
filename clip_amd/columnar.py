"""Columnar pre-tokenised corpus on disk: tokenise a notes CSV once, score it many times.

The reference re-reads and re-tokenises the CSV on every run, row by row
(/root/reference/sentclf/apply_bert_unlabeled.py:83-92 -> neural_baselines.py:193-244 -> bert.py:70-188).
At NOTEEVENTS scale (50 M context rows, ~7 G tokens) the rows do not fit the host's memory as Python
objects at all.  This is the format the bulk scorer streams instead: a directory of plain .npy files

    flat.npy      int32 [tokens]   every row's ids, concatenated ([CLS] .. [SEP] layout, no padding)
    offsets.npy   int64 [rows+1]   row i = flat[offsets[i] : offsets[i+1]]
    doc_ids.npy   int64 [rows]
    sent_ixs.npy  int64 [rows]
    meta.json     n_context_sentences, sep/cls ids, vocabulary size, row / token counts

opened with numpy memory maps (nothing is loaded until a chunk is scored) and cut into chunks of WHOLE
documents — the sharding unit of SURVEY.md §8e: chunk c belongs to rank c mod world, so a sentence's
context never crosses a rank and the set of chunks does not depend on the world size.
"""
import json
import os

import numpy as np

from .scoring import PackedRows

FILES = ("flat", "offsets", "doc_ids", "sent_ixs")


def save_rows(path, rows: PackedRows, doc_ids, sent_ixs, **meta):
    os.makedirs(path, exist_ok=True)
    doc_ids = np.asarray(doc_ids, np.int64)
    sent_ixs = np.asarray(sent_ixs, np.int64)
    if not (len(rows) == len(doc_ids) == len(sent_ixs)):
        raise ValueError("rows, doc_ids and sent_ixs differ in length")
    np.save(os.path.join(path, "flat.npy"), np.ascontiguousarray(rows.flat, np.int32))
    np.save(os.path.join(path, "offsets.npy"), np.ascontiguousarray(rows.offsets, np.int64))
    np.save(os.path.join(path, "doc_ids.npy"), doc_ids)
    np.save(os.path.join(path, "sent_ixs.npy"), sent_ixs)
    meta = dict(meta, rows=int(len(rows)), tokens=int(rows.offsets[-1]), format="clip_amd.columnar/1")
    with open(os.path.join(path, "meta.json"), "w") as f:
        json.dump(meta, f, indent=1)


def merge_parts(part_dirs, path, **meta):
    """Join columnar directories that hold consecutive row ranges of one corpus (the ranks' --save_rows parts,
    in rank order) into `path`, streaming through memory maps, and remove the parts."""
    import shutil
    parts = [ColumnarCorpus(d) for d in part_dirs]
    n_rows = sum(len(p) for p in parts)
    n_tok = sum(int(p.offsets[-1]) for p in parts)
    os.makedirs(path, exist_ok=True)
    open_out = lambda name, dtype, n: np.lib.format.open_memmap(os.path.join(path, name + ".npy"), mode="w+",
                                                                dtype=dtype, shape=(n,))
    flat, offsets = open_out("flat", np.int32, n_tok), open_out("offsets", np.int64, n_rows + 1)
    doc_ids, sent_ixs = open_out("doc_ids", np.int64, n_rows), open_out("sent_ixs", np.int64, n_rows)
    r = t = 0
    offsets[0] = 0
    for p in parts:
        k, m = len(p), int(p.offsets[-1])
        flat[t:t + m] = p.flat
        offsets[r + 1:r + k + 1] = np.asarray(p.offsets[1:]) + t
        doc_ids[r:r + k], sent_ixs[r:r + k] = p.doc_ids, p.sent_ixs
        r, t = r + k, t + m
    for a in (flat, offsets, doc_ids, sent_ixs):
        a.flush()
    del flat, offsets, doc_ids, sent_ixs, parts
    meta = dict(meta, rows=int(n_rows), tokens=int(n_tok), format="clip_amd.columnar/1")
    with open(os.path.join(path, "meta.json"), "w") as f:
        json.dump(meta, f, indent=1)
    for d in part_dirs:
        shutil.rmtree(d, ignore_errors=True)


def chunk_bounds(doc_ids, chunk_rows):
    """Row ranges of about chunk_rows rows that end on document boundaries (rows of a document are
    contiguous, as in the reference's CSVs).  A document longer than chunk_rows is one chunk."""
    d = np.asarray(doc_ids)
    n = len(d)
    if n == 0:
        return []
    starts = np.concatenate(([0], np.flatnonzero(d[1:] != d[:-1]) + 1, [n]))     # first row of each document, then n
    bounds, a = [], 0
    while a < n:
        k = int(np.searchsorted(starts, a + chunk_rows, side="right")) - 1      # last boundary <= a + chunk_rows
        b = int(starts[k])
        if b <= a:                                                               # one long document
            b = int(starts[int(np.searchsorted(starts, a, side="right"))])
        bounds.append((a, b))
        a = b
    return bounds


def graded_chunk_bounds(doc_ids, chunk_rows, first_rows=16384):
    """chunk_bounds whose first pieces are small and double up to chunk_rows: a pipelined consumer (the GPU)
    starts after first_rows rows are ready instead of chunk_rows.  Any chunking gives the same output files."""
    fine = chunk_bounds(doc_ids, min(first_rows, chunk_rows))
    out, want, i = [], min(first_rows, chunk_rows), 0
    while i < len(fine):
        a, b = fine[i]
        i += 1
        while i < len(fine) and fine[i][1] - a <= want:
            b = fine[i][1]
            i += 1
        out.append((a, b))
        want = min(chunk_rows, want * 2)
    return out


class ColumnarCorpus:
    @classmethod
    def from_arrays(cls, rows: PackedRows, doc_ids, sent_ixs, **meta):
        """The same interface over rows already in memory (a CSV tokenised in this process)."""
        self = cls.__new__(cls)
        self.flat, self.offsets = rows.flat, rows.offsets
        self.doc_ids, self.sent_ixs = np.asarray(doc_ids, np.int64), np.asarray(sent_ixs, np.int64)
        self.meta = dict(meta, rows=len(rows), tokens=int(rows.offsets[-1]), format="clip_amd.columnar/1")
        return self

    def __init__(self, path, mmap=True):
        mode = "r" if mmap else None
        for k in FILES:
            setattr(self, k, np.load(os.path.join(path, k + ".npy"), mmap_mode=mode))
        with open(os.path.join(path, "meta.json")) as f:
            self.meta = json.load(f)
        n = len(self.doc_ids)
        if self.meta.get("format") != "clip_amd.columnar/1" or len(self.offsets) != n + 1 or len(self.sent_ixs) != n \
                or int(self.offsets[-1]) != len(self.flat):
            raise ValueError(f"{path}: not a consistent clip_amd columnar corpus")

    def __len__(self):
        return len(self.doc_ids)

    def slice(self, a, b):
        """rows [a, b) as (PackedRows, doc_ids, sent_ixs), copied out of the memory map."""
        o = np.asarray(self.offsets[a:b + 1])
        return (PackedRows(np.asarray(self.flat[o[0]:o[-1]]), o - o[0]), np.asarray(self.doc_ids[a:b]),
                np.asarray(self.sent_ixs[a:b]))

    def chunk_bounds(self, chunk_rows):
        return chunk_bounds(self.doc_ids, chunk_rows)
    def chunks(self, chunk_rows, rank=0, world=1):
        for c, (a, b) in enumerate(self.chunk_bounds(chunk_rows)):
            if c % world == rank:
                yield c, a, b, self.slice(a, b)
