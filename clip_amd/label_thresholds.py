"""Labelled-data consumer of the scoring path: where clip_*_label_threshs.npy come from.

The reference's fine-tuning CLI ends with `evaluate(...)` on the validation split
(/root/reference/sentclf/bert.py:273-359): run the forward over every labelled sentence
(`gather_predictions`, bert.py:197-232, batches of ONE), then sweep thresholds 0.01..0.99 per label
(`multilabel_eval.balance_each_label`, multilabel_eval.py:98-122 -> `binary_eval.balanced_f1`,
binary_eval.py:8-32) and save the per-label F1-maximising thresholds (bert.py:615-618) — the
`float64[7]` files the scoring path applies (test_load.py:29,38).

Here the forward is the bulk scorer (one packed pass on the MI355X) and the sweep is vectorised
numpy over all 99 thresholds at once (the reference calls four sklearn scorers per threshold per
label).  Semantics kept, including sklearn's conventions:
  * prediction at threshold t is `score >= t` in the per-label sweep (binary_eval.py:12) but
    `score > t` in the micro sweep and when thresholds are applied (multilabel_eval.py:59,80);
  * precision / recall / F1 with an empty denominator are 0 (sklearn zero_division default);
  * the best threshold is the FIRST maximum of F1 (np.nanargmax, binary_eval.py:24);
  * AUC is the ROC area (ties share rank), 0 when only one class is present (binary_eval.py:27-31).
float32 scores are compared against float64 thresholds after promotion to float64 (numpy >= 2).
"""
import numpy as np

from .constants import LABEL_TYPES, label2abbrev

THRESHOLDS = np.arange(0, 1, .01)[1:]          # multilabel_eval.py:56,112; binary_eval.py:22


def _safe_div(num, den):
    num, den = np.asarray(num, np.float64), np.asarray(den, np.float64)
    return np.divide(num, den, out=np.zeros(np.broadcast(num, den).shape, np.float64), where=den != 0)


def _confusion(pred, y):
    """pred bool [..., N], y bool [N] -> tp, fp, fn, tn along the last axis."""
    y = y.astype(bool)
    tp = (pred & y).sum(-1)
    fp = (pred & ~y).sum(-1)
    fn = (~pred & y).sum(-1)
    return tp, fp, fn, y.size - tp - fp - fn


def threshold_metrics(scores, y, thresholds=THRESHOLDS):
    """ref: binary_eval.py:8-19 on column 1 of its two-column input (= scores).
    Returns dict of float64 arrays 'acc', 'prec', 'rec', 'f1', one value per threshold."""
    s = np.asarray(scores).astype(np.float64)
    pred = s[None, :] >= np.asarray(thresholds, np.float64)[:, None]
    tp, fp, fn, tn = _confusion(pred, np.asarray(y))
    return {"acc": (tp + tn) / float(s.size), "prec": _safe_div(tp, tp + fp), "rec": _safe_div(tp, tp + fn),
            "f1": _safe_div(2 * tp, 2 * tp + fp + fn)}


def roc_auc(y, scores):
    """Area under the ROC curve (sklearn roc_auc_score for a binary target): the Mann-Whitney
    statistic with average ranks for ties.  Raises ValueError when only one class is present."""
    y = np.asarray(y).astype(bool)
    s = np.asarray(scores).astype(np.float64)
    n_pos, n_neg = int(y.sum()), int((~y).sum())
    if n_pos == 0 or n_neg == 0:
        raise ValueError("Only one class present in y_true. ROC AUC score is not defined in that case.")
    order = np.argsort(s, kind="stable")
    ss = s[order]
    # average rank of each tie group
    starts = np.flatnonzero(np.concatenate(([True], ss[1:] != ss[:-1])))
    ends = np.concatenate((starts[1:], [s.size]))
    grp_rank = (starts + ends + 1) / 2.0                    # mean of ranks start+1 .. end
    ranks = np.empty(s.size, np.float64)
    ranks[order] = np.repeat(grp_rank, ends - starts)
    return float((ranks[y].sum() - n_pos * (n_pos + 1) / 2.0) / (n_pos * float(n_neg)))


def balanced_f1_binary(scores, y):
    """ref: binary_eval.py:21-32 -> (best_thresh, {'acc','prec','rec','f1','auc'})."""
    m = threshold_metrics(scores, y)
    best_ix = int(np.nanargmax(m["f1"]))
    out = {k: float(v[best_ix]) for k, v in m.items()}
    try:
        out["auc"] = roc_auc(y, scores)
    except ValueError:
        out["auc"] = 0
    return float(THRESHOLDS[best_ix]), out


def balance_each_label(yhat_raw, y):
    """ref: multilabel_eval.py:98-122 (get_high_prec_thresh=False).
    Returns (metrics, label_balanced_threshs, label_rec_values)."""
    yhat_raw, y = np.asarray(yhat_raw), np.asarray(y)
    metrics, threshs, rec_values = {}, [], {}
    for ix, label in enumerate(LABEL_TYPES):
        t, m = balanced_f1_binary(yhat_raw[:, ix], y[:, ix])
        rec_values[label] = threshold_metrics(yhat_raw[:, ix], y[:, ix])["rec"]
        threshs.append(t)
        for name, val in m.items():
            metrics[f"balanced-{label2abbrev[label]}-{name}"] = val
    for name in ("acc", "prec", "rec", "f1", "auc"):
        metrics[f"balanced-macro-{name}"] = float(np.mean(
            [metrics[f"balanced-{label2abbrev[label]}-{name}"] for label in LABEL_TYPES]))
    return metrics, threshs, rec_values


def _macro_micro(pred, y, f32_pred=False):
    """ref: multilabel_eval.py:209-260 (all_macro / all_micro) for pred, y bool [N, L].
    f32_pred: the caller's predictions were a float32 array (np.round of float32 scores, bert.py:225):
    `yhat.sum(axis=0) + 1e-10` is then a float32 sum, which swallows the 1e-10 unless the sum is 0."""
    inter = (pred & y).sum(0).astype(np.float64)
    union = (pred | y).sum(0).astype(np.float64)
    pden = (pred.sum(0).astype(np.float32) + np.float32(1e-10)).astype(np.float64) if f32_pred else pred.sum(0) + 1e-10
    mp = float(np.mean(inter / pden))
    mr = float(np.mean(inter / (y.sum(0) + 1e-10)))
    macro = (float(np.mean(inter / (union + 1e-10))), mp, mr, 0. if mp + mr == 0 else 2 * mp * mr / (mp + mr))
    ti, tu = float(inter.sum()), float(union.sum())
    with np.errstate(divide="ignore", invalid="ignore"):
        up = np.float64(ti) / np.float64(pred.sum())
        ur = np.float64(ti) / np.float64(y.sum())
        ua = np.float64(ti) / np.float64(tu)
    uf = 0. if up + ur == 0 else 2 * (up * ur) / (up + ur)
    return macro, (float(ua), float(up), float(ur), float(uf))


def balanced_f1_micro(yhat_raw, y):
    """ref: multilabel_eval.py:54-73 — ONE threshold for all labels, maximising micro F1
    (strict `>`).  Returns (best_thresh, metrics) or (None, None) when every F1 is NaN."""
    yhat_raw, yb = np.asarray(yhat_raw).astype(np.float64), np.asarray(y).astype(bool)
    names = ("acc", "prec", "rec", "f1")
    cols = {f"{n}_{k}": [] for k in ("micro", "macro") for n in names}
    for t in THRESHOLDS:
        macro, micro = _macro_micro(yhat_raw > t, yb)
        for n, v in zip(names, micro):
            cols[f"{n}_micro"].append(v)
        for n, v in zip(names, macro):
            cols[f"{n}_macro"].append(v)
    if np.all(np.isnan(cols["f1_micro"])):
        return None, None
    best_ix = int(np.nanargmax(cols["f1_micro"]))
    return float(THRESHOLDS[best_ix]), {k: v[best_ix] for k, v in cols.items()}


def macro_metrics_given_thresholds(yhat_raw, y, label_threshs):
    """ref: multilabel_eval.py:75-95 — apply saved thresholds (strict `>`), per-label + macro."""
    yhat_raw, yb = np.asarray(yhat_raw), np.asarray(y).astype(bool)
    thr = np.asarray(label_threshs, np.float64)
    metrics, get_auc = {}, True
    for ix, label in enumerate(LABEL_TYPES):
        ln = label2abbrev[label]
        pred = yhat_raw[:, ix].astype(np.float64) > thr[ix]
        tp, fp, fn, tn = _confusion(pred, yb[:, ix])
        vals = {"acc": (tp + tn) / float(pred.size), "prec": _safe_div(tp, tp + fp), "rec": _safe_div(tp, tp + fn),
                "f1": _safe_div(2 * tp, 2 * tp + fp + fn)}
        for n, v in vals.items():
            metrics[f"balanced-{ln}-{n}"] = float(v)
        try:
            metrics[f"balanced-{ln}-auc"] = roc_auc(yb[:, ix], yhat_raw[:, ix])
        except ValueError:
            get_auc = False
    for n in ["acc", "prec", "rec", "f1"] + (["auc"] if get_auc else []):
        m = float(np.mean([metrics[f"balanced-{label2abbrev[label]}-{n}"] for label in LABEL_TYPES]))
        metrics[f"balanced-macro-{n}"] = m
        metrics[f"{n}_macro"] = m
    return metrics


def gather_predictions(scorer, rows, labels):
    """ref: bert.py:197-232 — (yhat_raw, yhat, y) for a labelled set; the forward is one bulk pass
    (`scorer` = clip_amd.scoring.BulkScorer built with keep_probs=True) instead of batches of one."""
    res = scorer.score(rows)
    if res.probs is None:
        raise ValueError("gather_predictions needs BulkScorer(keep_probs=True)")
    yhat_raw = res.probs
    return yhat_raw, np.round(yhat_raw), np.asarray(labels)


def fit_label_thresholds(scorer, rows, labels, out_path=None):
    """The validation step of the reference's CLI (bert.py:289,615-618): scores -> per-label
    thresholds, optionally saved as the float64 .npy the scoring path loads."""
    yhat_raw, _, y = gather_predictions(scorer, rows, labels)
    metrics, threshs, _ = balance_each_label(yhat_raw, y)
    threshs = np.asarray(threshs, np.float64)
    if out_path is not None:
        np.save(out_path, threshs)
    return threshs, metrics


# ------------------------------------------------------------------ the rest of bert.py:284-315
def roc_auc_curve(y, scores):
    """sklearn.metrics.auc(*roc_curve(y, scores)[:2]): trapezoid area over the ROC points at the
    distinct score values (no collinear-point dropping needed for the area).  nan when y has no
    positive or no negative (sklearn warns and returns nan there)."""
    y = np.asarray(y).astype(bool)
    s = np.asarray(scores).astype(np.float64)
    n_pos, n_neg = int(y.sum()), int((~y).sum())
    if n_pos == 0 or n_neg == 0:
        return float("nan")
    order = np.argsort(-s, kind="stable")
    ys, ss = y[order], s[order]
    last = np.flatnonzero(np.concatenate((ss[1:] != ss[:-1], [True])))      # last index of each distinct score
    tps = np.concatenate(([0], np.cumsum(ys)[last]))
    fps = np.concatenate(([0], np.cumsum(~ys)[last]))
    return float(np.trapezoid(tps / n_pos, fps / n_neg))


def all_metrics(yhat, y, yhat_raw=None, calc_auc=True, label_order=LABEL_TYPES):
    """ref: multilabel_eval.py:18-52 (+ auc_metrics :262-292): per-label F1 at the 0.5 rounding,
    macro / micro accuracy, precision, recall, F1, macro / micro AUC."""
    f32_pred = np.asarray(yhat).dtype == np.float32
    yhat, yb = np.asarray(yhat).astype(bool), np.asarray(y).astype(bool)
    metrics = {}
    for ix, label in enumerate(label_order):
        tp, fp, fn, _ = _confusion(yhat[:, ix], yb[:, ix])
        metrics[f"{label2abbrev[label]}-f1"] = float(_safe_div(2 * tp, 2 * tp + fp + fn))
    macro, micro = _macro_micro(yhat, yb, f32_pred)
    for n, v in zip(("acc", "prec", "rec", "f1"), macro):
        metrics[f"{n}_macro"] = v
    for n, v in zip(("acc", "prec", "rec", "f1"), micro):
        metrics[f"{n}_micro"] = v
    if yhat_raw is not None and calc_auc and len(yb) > 1:
        raw = np.asarray(yhat_raw)
        aucs = []
        for i in range(yb.shape[1]):
            if yb[:, i].sum() > 0:                        # only labels with true positives (:273)
                a = roc_auc_curve(yb[:, i], raw[:, i])
                if not np.isnan(a):
                    aucs.append(a)
        metrics["auc_macro"] = float(np.mean(aucs)) if aucs else float("nan")
        metrics["auc_micro"] = roc_auc_curve(yb.ravel(), raw.ravel())
    return metrics


def precision_at_fixed_recall(scores, y, fixed_metric):
    """ref: binary_eval.py:34-42 — threshold that keeps `fixed_metric` of the true positives, and the
    precision there (`>=`)."""
    s = np.asarray(scores)
    yb = np.asarray(y).astype(bool)
    tmp = s * yb
    true_scores = np.sort(tmp[tmp > 0])
    thresh = true_scores[round(len(true_scores) * (1 - fixed_metric)) - 1]
    pred = s >= thresh
    tp, fp, _, _ = _confusion(pred, yb)
    return thresh, float(_safe_div(tp, tp + fp))


def evaluate_scores(yhat_raw, y):
    """The multilabel branch of the reference's `evaluate` (bert.py:284-315) on scores already gathered:
    unbalanced metrics at 0.5, per-label balanced thresholds, the micro-F1 threshold, the binary
    (any-label) view at the per-label thresholds, precision at 90 / 95 / 99 % recall of the binary view.
    Returns (metrics, label_threshs, micro_thresh, rec_90_thresh, rec_95_thresh, rec_99_thresh)."""
    yhat_raw, y = np.asarray(yhat_raw), np.asarray(y)
    metrics = all_metrics(np.round(yhat_raw), y, yhat_raw=yhat_raw)
    label_metrics, label_threshs, _ = balance_each_label(yhat_raw, y)
    metrics.update(label_metrics)
    micro_thresh, micro_metrics = balanced_f1_micro(yhat_raw, y)
    metrics.update({f"balanced_{k}": v for k, v in micro_metrics.items()})
    binary_yhat = np.any(yhat_raw >= np.asarray(label_threshs), axis=1)          # :300, `>=` as written
    binary_y = np.any(y, axis=1)
    tp, fp, fn, _ = _confusion(binary_yhat, binary_y)
    metrics["binary_prec"] = float(_safe_div(tp, tp + fp))
    metrics["binary_rec"] = float(_safe_div(tp, tp + fn))
    metrics["binary_f1"] = float(_safe_div(2 * tp, 2 * tp + fp + fn))
    binary_raw = np.max(yhat_raw, axis=1)
    out = []
    for r in (0.90, 0.95, 0.99):
        t, p = precision_at_fixed_recall(binary_raw, binary_y, r)
        metrics[f"prec@rec={int(round(r * 100))}"] = p
        out.append(float(t))
    return (metrics, label_threshs, micro_thresh) + tuple(out)
