"""`precision="auto"` — the default numerics setting of the model classes, BulkScorer, the CLI and the job.

BASELINE.json states the parity criteria of the scoring path: per-label logits within 2e-2 of the
reference and >= 99.9 % agreement on thresholded labels (test_load.py:37-38: `sigmoid(pred) >
label_threshs`, every decision counted).  16-bit operand rounding is RELATIVE, so whether a 16-bit
setting meets them depends on the checkpoint (logits of +-2: bf16 does; logits of +-8 with LayerNorm
gamma outliers: bf16 misses both, tests/test_gpu_parity_gates.py).  Nothing in the reference chooses a
precision (apply_bert_unlabeled.py:72 runs fp32 torch), so the build measures instead of assuming:

  1. a sample (up to SAMPLE_ROWS = 2048 rows of the caller's own input, else synthetic rows of the model's
     layout) is scored in the fp32 check mode (pinned to the reference at <= 1e-4 on every golden);
  2. the candidates are tried fastest first — bf16, f16, f16r (fp16 operands, fp32 residual stream) —
     and the first one with   max |logit - fp32| <= 2e-2   AND   agreement >= 99.9 % on BOTH threshold
     files the reference ships (and on the caller's own thresholds, when given) is selected;
  3. a candidate is refused outright if a weight leaves its operand range (fp16: +-65504, raised by
     clip_finalize_weights) or any sampled score is non-finite (an fp16 activation overflow);
  4. if no candidate passes, the most precise finite one (f16r, else f16, else bf16) is selected and the
     report says so — `precision="fp32"` is the exact (16x slower) setting for such a checkpoint.

The choice and the measured numbers are printed once and kept in `model.precision_report`.  The
decision is a pure function of (weights, sample), so every rank of a multi-GPU job reaches the same one
without talking to the others.
"""
import hashlib
import sys

import numpy as np

from .constants import DNOTE_LABEL_THRESHS, TTP_LABEL_THRESHS

CANDIDATES = ("bf16", "f16", "f16r")
MAX_ABS_LOGIT = 2e-2          # BASELINE.json north_star
MIN_AGREEMENT = 0.999         # BASELINE.json north_star
SAMPLE_ROWS = 2048         # 14 336 decisions with 7 labels: 99.9 % leaves room for 14 flips (512 rows left room for 3: noise)
_CACHE = {}                   # (weights fingerprint, model kind, sample fingerprint) -> report


def label_decisions(probs, thr):
    """test_load.py:37-38: float32 sigmoid > float64 threshold, strict."""
    return np.asarray(probs, np.float32).astype(np.float64) > np.asarray(thr, np.float64)[None, :]


def synthetic_sample(model, n_rows=SAMPLE_ROWS, seed=20211):
    """Rows of the model's own layout ([CLS] ctx [SEP] .. focus [SEP] .. ctx [SEP], 2n+1 separators; or
    [CLS] w.. [SEP]) with uniform word ids, as clip_amd.synth builds them for the benchmarks."""
    from . import synth
    from .constants import CLS_ID, SEP_ID
    from .scoring import PackedRows
    cfg = model.config
    n_ctx = int(getattr(model, "n_context_sentences", None) or 0)
    sep = getattr(model, "sep_token_id", None)
    sep = SEP_ID if sep is None else int(sep)
    max_len = min(256 if n_ctx else 64, cfg.max_position_embeddings)
    lo = 1000 if cfg.vocab_size > 2000 else max(cfg.vocab_size // 2, 1)
    cls = CLS_ID if CLS_ID < cfg.vocab_size and CLS_ID != sep else (1 if sep != 1 else 2)
    flat, off = synth.packed_context_rows(n_rows, n_ctx, cfg.vocab_size, max_len, seed=seed, word_lo=lo,
                                          cls_id=cls, sep_id=sep)
    return PackedRows(flat, off)


def spread_sample(rows, n_rows=SAMPLE_ROWS, blocks=8):
    """Up to n_rows rows of a PackedRows: `blocks` contiguous runs spread evenly over the input."""
    from .scoring import PackedRows
    N = len(rows)
    if N <= n_rows:
        return rows
    per = n_rows // blocks
    starts = (np.arange(blocks) * (N - per) // max(blocks - 1, 1)).astype(np.int64)
    idx = (starts[:, None] + np.arange(per)[None, :]).reshape(-1)
    flat, cu = rows.gather_all(idx)
    return PackedRows(flat, cu)


def _fingerprint(model):
    h = hashlib.blake2b(digest_size=16)
    sd = model._sd
    for k in sorted(sd):
        t = sd[k]
        h.update(k.encode())
        h.update(str(tuple(t.shape)).encode())
        a = t.reshape(-1)
        step = max(1, a.numel() // 4096)          # a strided sample of every tensor: cheap, and any real change shows
        h.update(a[::step].numpy().tobytes())
    return h.hexdigest()


def _clone(model, precision, max_tokens, max_seqs):
    m = model.__class__(model.config, max_tokens=max_tokens, max_seqs=max_seqs)
    m._sd = model._sd                               # shared, read-only
    m.precision = m.resolved_precision = precision
    for k in ("sep_token_id", "n_context_sentences", "task"):
        if hasattr(model, k):
            setattr(m, k, getattr(model, k))
    m.to(model.device)
    return m


def _score(model, precision, sample, max_tokens, max_seqs, keep=False):
    """(logits, probs, clone): the clone still holds its engine when keep (the caller adopts or drops it)"""
    from .scoring import BulkScorer
    m = _clone(model, precision, max_tokens, max_seqs)
    try:
        res = BulkScorer(m, keep_probs=True, keep_logits=True).score(sample)
    except BaseException:
        m._drop_engine()
        raise
    if not keep:
        m._drop_engine()
    return res.logits, res.probs, (m if keep else None)


def choose_precision(model, rows=None, label_threshs=None, log=None):
    """Measure the candidates on a sample.  Returns (report dict with key 'selected', clone or None): the clone is
    the candidate model that passed, built with the caller's own workspace size and still holding its engine, so
    that the caller adopts that engine instead of building the same one again."""
    from . import _lib
    from .scoring import PackedRows
    if rows is not None and not isinstance(rows, PackedRows):
        rows = PackedRows.from_rows(rows)
    sample = spread_sample(rows) if rows is not None and len(rows) else synthetic_sample(model)
    source = "caller's input" if rows is not None and len(rows) else "synthetic rows"
    key = (_fingerprint(model), model.__class__.__name__, getattr(model, "n_context_sentences", None),
           getattr(model, "sep_token_id", None), hashlib.blake2b(sample.flat.tobytes(), digest_size=16).hexdigest(),
           None if label_threshs is None else tuple(np.asarray(label_threshs, np.float64).tolist()))
    if key in _CACHE:
        return _CACHE[key], None
    max_tokens = max(32768, int(sample.lens.max()))
    max_seqs = max(1024, min(len(sample), 4096))
    L = model.num_labels
    thr_sets = {}
    if L == len(DNOTE_LABEL_THRESHS):
        thr_sets = {"dnote": DNOTE_LABEL_THRESHS, "ttp": TTP_LABEL_THRESHS}
    if label_threshs is not None:
        thr_sets["caller"] = np.asarray(label_threshs, np.float64)
    import time
    t0 = time.perf_counter()
    ref_logits, ref_probs, _ = _score(model, "fp32", sample, max_tokens, max_seqs)
    report = {"sample_rows": int(len(sample)), "sample_tokens": int(sample.offsets[-1]), "sample": source,
              "criteria": {"max_abs_logit": MAX_ABS_LOGIT, "min_agreement": MIN_AGREEMENT},
              "candidates": {}, "selected": None, "meets_criteria": False}
    finite = []
    adopted = None
    for cand in CANDIDATES:
        entry = {}
        report["candidates"][cand] = entry
        try:
            logits, probs, clone = _score(model, cand, sample, max(model.max_tokens, int(sample.lens.max())),
                                          max(model.max_seqs, 1), keep=True)
        except _lib.ClipError as exc:               # a weight outside the operand range (clip_finalize_weights)
            if "range" not in str(exc):
                raise
            entry["refused"] = str(exc)
            continue
        except FloatingPointError as exc:           # non-finite logits on the sample (device error word, bit 3)
            entry["refused"] = str(exc)
            continue
        finite.append(cand)
        entry["max_abs_logit_err"] = float(np.abs(logits - ref_logits).max())
        entry["rms_logit_err"] = float(np.sqrt(np.mean((logits - ref_logits).astype(np.float64) ** 2)))
        entry["agreement"] = {k: float((label_decisions(probs, t) == label_decisions(ref_probs, t)).mean())
                              for k, t in thr_sets.items()}
        entry["passes"] = bool(entry["max_abs_logit_err"] <= MAX_ABS_LOGIT and
                               all(a >= MIN_AGREEMENT for a in entry["agreement"].values()))
        if entry["passes"]:
            report["selected"], report["meets_criteria"] = cand, True
            adopted = clone
            break
        clone._drop_engine()
    if report["selected"] is None:
        if not finite:
            raise FloatingPointError("precision='auto': no 16-bit setting gives finite scores on the sample; "
                                     "use precision='fp32' for this checkpoint")
        report["selected"] = finite[-1]
    report["seconds"] = round(time.perf_counter() - t0, 3)
    _CACHE[key] = report
    if log is not False:
        print(format_report(report), file=log or sys.stderr, flush=True)
    return report, adopted


def format_report(report):
    parts = []
    for cand, e in report["candidates"].items():
        if "refused" in e:
            parts.append(f"{cand}: refused ({e['refused']})")
        else:
            ag = ", ".join(f"{k} {100 * v:.3f} %" for k, v in e["agreement"].items())
            parts.append(f"{cand}: max |dlogit| {e['max_abs_logit_err']:.2e}, agreement {ag or 'n/a'}"
                         f" -> {'ok' if e['passes'] else 'no'}")
    tail = "" if report["meets_criteria"] else \
        " (NO 16-bit setting meets 2e-2 / 99.9 % on this checkpoint: the most precise one is used; precision='fp32' is exact)"
    return (f"[clip_amd] precision auto -> {report['selected']}{tail}; {report['sample_rows']} {report['sample']} "
            f"vs the fp32 check mode ({report.get('seconds', 0):.1f} s): " + "; ".join(parts))
