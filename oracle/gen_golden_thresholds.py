"""Generate tests/golden/thresholds.json by running the REFERENCE's threshold search here.

Run in the build container only (needs /root/reference and scikit-learn):
    python oracle/gen_golden_thresholds.py

Imports sentclf/multilabel_eval.py and sentclf/binary_eval.py from where they lie and calls
balance_each_label (multilabel_eval.py:98-122), balanced_f1 (:54-73) and
macro_metrics_given_thresholds (:75-95) on synthetic score matrices; what is committed is data only:
the generator seeds and the reference's outputs.
"""
import contextlib
import io
import json
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/sentclf")
OUT = os.path.join(ROOT, "tests", "golden", "thresholds.json")


def synth_scores(n, seed, dead_label=None, quantize=None):
    """Scores that look like a trained 7-label classifier's sigmoid outputs: positives shifted up,
    per-label prevalence 3..25 %, float32.  `quantize` forces ties; `dead_label` has no positives."""
    rng = np.random.default_rng(seed)
    prev = np.array([0.05, 0.25, 0.2, 0.03, 0.04, 0.15, 0.08])
    y = (rng.random((n, 7)) < prev).astype(np.int64)
    if dead_label is not None:
        y[:, dead_label] = 0
    z = rng.normal(-2.0, 1.2, (n, 7)) + y * rng.normal(3.0, 1.0, (n, 7))
    p = (1.0 / (1.0 + np.exp(-z))).astype(np.float32)
    if quantize:
        p = (np.round(p * quantize) / quantize).astype(np.float32)
    return p, y


def main():
    import multilabel_eval as ME     # the reference
    cases = {"plain": dict(n=600, seed=31), "ties": dict(n=400, seed=32, quantize=50),
             "dead_label": dict(n=300, seed=33, dead_label=3)}
    out = {}
    for name, kw in cases.items():
        p, y = synth_scores(**kw)
        with warnings.catch_warnings(), contextlib.redirect_stdout(io.StringIO()):
            warnings.simplefilter("ignore")
            metrics, threshs, _, _, rec_values = ME.balance_each_label(p, y)
            micro_t, micro_m = ME.balanced_f1(p, y)
            given = ME.macro_metrics_given_thresholds(p, y, threshs)
            allm = ME.all_metrics(np.round(p), y, yhat_raw=p, calc_auc=True, label_order=ME.LABEL_TYPES)
            import binary_eval as BE
            from sklearn.metrics import precision_score, recall_score, f1_score
            binary_yhat = np.any(p >= np.array(threshs), axis=1)          # bert.py:300-304
            binary_y = np.any(y, axis=1)
            binary = {"binary_prec": precision_score(binary_y, binary_yhat), "binary_rec": recall_score(binary_y, binary_yhat),
                      "binary_f1": f1_score(binary_y, binary_yhat)}
            braw = np.max(p, axis=1)
            braw2 = np.stack((1 - braw, braw), axis=1)                    # bert.py:308-315
            fixed = {}
            for r in (0.90, 0.95, 0.99):
                t, pr = BE.precision_at_fixed_recall(braw2, binary_y, r)
                fixed[str(int(round(r * 100)))] = [float(t), float(pr)]
        out[name] = {
            "gen": kw,
            "label_threshs": [float(t) for t in threshs],
            "balance_each_label": {k: float(v) for k, v in metrics.items()},
            "rec_values": {k: [float(x) for x in v] for k, v in rec_values.items()},
            "micro_thresh": float(micro_t), "micro_metrics": {k: float(v) for k, v in micro_m.items()},
            "given_thresholds": {k: float(v) for k, v in given.items()},
            "all_metrics": {k: float(v) for k, v in allm.items()},
            "binary": {k: float(v) for k, v in binary.items()}, "prec_at_fixed_recall": fixed,
            "checksum_scores": float(p.astype(np.float64).sum()), "checksum_y": int(y.sum()),
        }
        print(name, "thresholds", out[name]["label_threshs"], "micro", micro_t)
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
