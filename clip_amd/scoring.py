"""Bulk scoring driver: the hot loop of /root/reference/sentclf/apply_bert_unlabeled.py:92-124,
re-designed for one MI355X per process.

Reference loop: DataLoader(batch_size=8, shuffle=False) -> pad to the longest of 8 -> H2D ->
forward -> sigmoid -> D2H + sync -> numpy threshold -> csv row, 8 sentences at a time.
Here: rows are LENGTH-BUCKETED (stable sort by token count) and PACKED without padding into steps
of up to `tokens_per_step` tokens; per step the host ships only token ids + row offsets; token
types, positions and the focus-[SEP] index are derived on the device (clip_prepare_packed), the
encoder runs once over the packed tokens and the fused epilogue leaves keep / max_label_score /
per-label decisions in device memory.  Results come back once, in input order.  A row's result does
not depend on which rows share its step (tests/test_gpu_model.py::test_packed_path_equals_padded_path),
so bucketing changes nothing but speed.
"""
import csv
import ctypes as C
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib


@dataclass
class ScoreResult:
    doc_ids: np.ndarray      # int64 [N]
    sent_ixs: np.ndarray     # int64 [N]
    max_score: np.ndarray    # float32 [N]  max_j sigmoid(logit_j)     (apply_bert_unlabeled.py:124)
    keep: np.ndarray         # bool [N]     any_j(p_j > threshold)      (apply_bert_unlabeled.py:117)
    labels: np.ndarray       # bool [N, L]  p_j > label_threshs[j]      (test_load.py:38), or None
    probs: np.ndarray        # float32 [N, L] or None
    logits: np.ndarray = None   # float32 [N, L] (keep_logits=True)


class PackedRows:
    """Pre-tokenised rows in columnar form: one flat int32 id array + int64 offsets."""

    def __init__(self, flat_ids, offsets):
        self.flat = np.ascontiguousarray(flat_ids, dtype=np.int32)
        self.offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        self.lens = np.diff(self.offsets).astype(np.int32)

    @classmethod
    def from_rows(cls, rows):
        lens = np.fromiter((len(r) for r in rows), dtype=np.int64, count=len(rows))
        off = np.concatenate(([0], np.cumsum(lens)))
        flat = np.concatenate([np.asarray(r, np.int32) for r in rows]) if len(rows) else np.zeros(0, np.int32)
        return cls(flat, off)

    def __len__(self):
        return len(self.lens)

    def gather(self, idx):
        """flat ids and cu_seqlens of the rows idx (in that order)."""
        lens = self.lens[idx].astype(np.int64)
        cu = np.concatenate(([0], np.cumsum(lens)))
        # token t of output row i comes from offsets[idx[i]] + (t - cu[i])
        src = np.repeat(self.offsets[idx] - cu[:-1], lens) + np.arange(cu[-1])
        return self.flat[src], cu.astype(np.int32)

    def gather_all(self, idx):
        """The rows idx as one flat array + int64 offsets (one vectorised gather per chunk: the steps of
        a length-sorted chunk are then contiguous slices)."""
        lens = self.lens[idx].astype(np.int64)
        cu = np.concatenate(([0], np.cumsum(lens)))
        src = np.repeat(self.offsets[idx] - cu[:-1], lens) + np.arange(cu[-1])
        return np.take(self.flat, src), cu


def plan_steps(lens, tokens_per_step, max_rows_per_step):
    """Length bucketing: stable argsort by token count, then greedy steps bounded by tokens and rows.
    Returns (order, [(start, stop), ...]) over the sorted order."""
    order = np.argsort(lens, kind="stable")
    sl = lens[order].astype(np.int64)
    if len(sl) and sl[-1] > tokens_per_step:
        raise ValueError(f"a row of {sl[-1]} tokens exceeds tokens_per_step={tokens_per_step}")
    steps, i, n = [], 0, len(sl)
    csum = np.concatenate(([0], np.cumsum(sl)))
    while i < n:
        j = int(np.searchsorted(csum, csum[i] + tokens_per_step, side="right")) - 1
        j = min(max(j, i + 1), i + max_rows_per_step, n)
        steps.append((i, j))
        i = j
    return order, steps


class BulkScorer:
    def __init__(self, model, any_threshold=0.05, label_threshs=None, keep_probs=False,
                 tokens_per_step=None, max_rows_per_step=None, keep_logits=False):
        self.model = model
        self.n_ctx = getattr(model, "n_context_sentences", None) or 0
        self.sep_id = getattr(model, "sep_token_id", None) or 0
        self.any_threshold = float(any_threshold)
        self.label_threshs = None if label_threshs is None else np.asarray(label_threshs, np.float64)
        self.keep_probs, self.keep_logits = keep_probs, keep_logits
        # the engine's preferred step (clip_preferred_step_tokens) is asked at the first score(): a model with
        # precision='auto' is calibrated on that call's rows first, and only then is there an engine to ask
        self._tokens_per_step_arg = tokens_per_step
        self.tokens_per_step = int(tokens_per_step or model.max_tokens)
        self.max_rows_per_step = int(max_rows_per_step or model.max_seqs)
        if self.tokens_per_step > model.max_tokens or self.max_rows_per_step > model.max_seqs:
            raise ValueError("step size exceeds the model's engine workspace")
        self._ws = None

    def _resolve_step(self, rows):
        model = self.model
        # precision='auto': measured on (a spread sample of) the caller's own rows, once per model
        model.calibrate_precision(rows, self.label_threshs)
        if self._tokens_per_step_arg is None:
            t = C.c_int32(0)     # whole rounds of the fused-LayerNorm GEMMs (clip_preferred_step_tokens)
            _lib.check(_lib.load().clip_preferred_step_tokens(model._ensure_engine(), C.byref(t)))
            self.tokens_per_step = int(t.value or model.max_tokens)

    def _workspace(self, dev, L):
        """Per-scorer step buffers, allocated ONCE (the step loop makes no device or pinned-host allocation; its only temporaries are the numpy index arrays of the per-step gather): device
        tensors for one step (steps are serialised on the stream) and two pinned host staging buffers."""
        if self._ws is not None and self._ws["dev"] == dev:
            return self._ws
        T, R = self.tokens_per_step, self.max_rows_per_step
        i32 = dict(dtype=torch.int32, device=dev)
        ws = {"dev": dev, "ids": torch.empty(T, **i32), "cu": torch.empty(R + 1, **i32), "tt": torch.empty(T, **i32),
              "pos": torch.empty(T, **i32), "focus": torch.empty(R, **i32),
              "logits": torch.empty((R, L), dtype=torch.float32, device=dev), "stage": []}
        for _ in range(2):
            ids_h = torch.empty(T, dtype=torch.int32).pin_memory()
            cu_h = torch.empty(R + 1, dtype=torch.int32).pin_memory()
            ws["stage"].append((ids_h, cu_h, ids_h.numpy(), cu_h.numpy(), torch.cuda.Event()))
        self._ws = ws
        return ws

    def score(self, rows, doc_ids=None, sent_ixs=None):
        """rows: PackedRows or a list of int token-id arrays ([CLS] .. [SEP] layout of the
        collators). Returns a ScoreResult in INPUT order."""
        if not isinstance(rows, PackedRows):
            rows = PackedRows.from_rows(rows)
        N = len(rows)
        model, lib = self.model, _lib.load()
        dev = model.device
        if dev.type != "cuda":
            raise RuntimeError("BulkScorer needs the model on an MI355X (model.to('cuda')); no CPU path")
        L = model.num_labels
        doc_ids = np.zeros(N, np.int64) if doc_ids is None else np.asarray(doc_ids, np.int64)
        sent_ixs = np.arange(N, dtype=np.int64) if sent_ixs is None else np.asarray(sent_ixs, np.int64)
        if int(rows.lens.min(initial=1)) < 1:
            raise ValueError("empty row")
        if N:
            self._resolve_step(rows)
        order, steps = plan_steps(rows.lens, self.tokens_per_step, self.max_rows_per_step)

        out_max = torch.empty(N, dtype=torch.float32, device=dev)
        out_keep = torch.empty(N, dtype=torch.uint8, device=dev)
        out_lab = torch.empty((N, L), dtype=torch.uint8, device=dev) if self.label_threshs is not None else None
        out_p = torch.empty((N, L), dtype=torch.float32, device=dev) if self.keep_probs else None
        out_lg = torch.empty((N, L), dtype=torch.float32, device=dev) if self.keep_logits else None
        thr = None if self.label_threshs is None else torch.tensor(self.label_threshs, dtype=torch.float64, device=dev)
        err = torch.zeros(1, dtype=torch.int32, device=dev)
        n_ctx_arg = self.n_ctx if self.n_ctx > 0 else -1
        lens_s = rows.lens[order].astype(np.int64)      # the chunk in step order
        start_s = rows.offsets[order]

        with torch.cuda.device(dev):
            stream = _lib.stream_ptr()
            engine = model._ensure_engine()
            w = self._workspace(dev, L)
            ids_d, cu_d, tt_d, pos_d, focus_d, logits = (w[k] for k in ("ids", "cu", "tt", "pos", "focus", "logits"))
            for k, (a, b) in enumerate(steps):
                n = b - a
                # two pinned staging buffers, reused: wait for the copy that last read this one
                ids_h, cu_h, ids_hn, cu_hn, ev = w["stage"][k & 1]
                ev.synchronize()
                # this step's rows gathered straight into the staging buffer, while the GPU runs the steps
                # queued before (token t of row i comes from start[i] + (t - cu[i]))
                cu_hn[0] = 0
                np.cumsum(lens_s[a:b], out=cu_hn[1:n + 1])
                T = int(cu_hn[n])
                src = np.repeat(start_s[a:b] - cu_hn[:n], lens_s[a:b])
                src += np.arange(T)
                np.take(rows.flat, src, out=ids_hn[:T])
                ids_d[:T].copy_(ids_h[:T], non_blocking=True)
                cu_d[:n + 1].copy_(cu_h[:n + 1], non_blocking=True)
                ev.record()
                _lib.check(lib.clip_prepare_packed(_lib.ptr(ids_d), _lib.ptr(cu_d), n, self.sep_id, n_ctx_arg,
                                                   _lib.ptr(tt_d), _lib.ptr(pos_d), _lib.ptr(focus_d),
                                                   _lib.ptr(err), stream))
                _lib.check(lib.clip_forward_packed(engine, _lib.ptr(ids_d), _lib.ptr(tt_d), _lib.ptr(pos_d),
                                                   _lib.ptr(cu_d), _lib.ptr(focus_d), n, T,
                                                   int(rows.lens[order[b - 1]]), model.head_kind, _lib.ptr(logits), stream))
                _lib.check(lib.clip_epilogue(
                    _lib.ptr(logits), _lib.ptr(thr), self.any_threshold, n, L,
                    _lib.ptr(out_p[a:b]) if out_p is not None else None, _lib.ptr(out_max[a:b]),
                    _lib.ptr(out_lab[a:b]) if out_lab is not None else None, _lib.ptr(out_keep[a:b]), stream))
                if out_lg is not None:
                    out_lg[a:b].copy_(logits[:n])
            # one 4-byte read per chunk: an exchange time-out, a bad row table, NON-FINITE logits (fp16 overflow:
            # `p > thr` is False for NaN, the sentence would silently drop out of both output files) all raise
            _lib.poll_errors(engine)
            if int(err.item()) != 0:
                raise TypeError(f"a row does not hold exactly {2 * self.n_ctx + 1} [SEP] tokens "
                                "(sentclf/bert_model.py:51-55)")

        def unsort(t):
            a = t.cpu().numpy()
            res = np.empty_like(a)
            res[order] = a
            return res

        max_score = unsort(out_max)
        return ScoreResult(doc_ids=doc_ids, sent_ixs=sent_ixs, max_score=max_score,
                           keep=unsort(out_keep).astype(bool),
                           labels=None if out_lab is None else unsort(out_lab).astype(bool),
                           probs=None if out_p is None else unsort(out_p),
                           logits=None if out_lg is None else unsort(out_lg))


def write_ttp_csv(path, result: ScoreResult):
    """ttp_doc_sents.csv as the reference writes it (apply_bert_unlabeled.py:97-98,124): no header,
    rows `doc_id,sent_ix,max_label_score` for the kept sentences in input order, the score printed
    as str(numpy.float32)."""
    # csv.writer's format (fields joined by ',', rows ended by '\r\n'; ints and str(np.float32) need no
    # quoting), built with array operations: numpy's float32 -> str is the same shortest-repr text
    sel = np.nonzero(result.keep)[0]
    with open(path, "w", newline="") as f:
        f.write(format_score_rows(result.doc_ids[sel], np.asarray(result.sent_ixs)[sel], result.max_score[sel]))


def format_score_rows(doc_ids, sent_ixs, scores):
    """csv.writer's text for rows [int doc_id, int sent_ix, np.float32 score], as one string."""
    if len(doc_ids) == 0:
        return ""
    d = np.asarray(doc_ids).astype(str)
    s = np.asarray(sent_ixs).astype(str)
    m = np.asarray(scores, np.float32).astype(str)
    lines = np.char.add(np.char.add(np.char.add(np.char.add(d, ","), s), ","), m)
    return "\r\n".join(lines.tolist()) + "\r\n"


def read_ttp_csv(path):
    d, s, m = [], [], []
    with open(path, newline="") as f:
        for row in csv.reader(f):
            d.append(int(row[0])); s.append(int(row[1])); m.append(np.float32(row[2]))
    return np.asarray(d, np.int64), np.asarray(s, np.int64), np.asarray(m, np.float32)
