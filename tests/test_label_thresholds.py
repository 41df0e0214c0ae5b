"""Threshold search (the labelled-data consumer of the scoring path) against
goldens produced by the reference's own multilabel_eval.py / binary_eval.py
(oracle/gen_golden_thresholds.py, run in the build container)."""
import importlib.util
import json
import os

import numpy as np
import pytest

from clip_amd import label_thresholds as LT
from clip_amd.constants import LABEL_TYPES

HERE = os.path.dirname(os.path.abspath(__file__))


def _gen():
    spec = importlib.util.spec_from_file_location("gen_thr", os.path.join(HERE, "..", "oracle", "gen_golden_thresholds.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


GOLD = json.load(open(os.path.join(HERE, "golden", "thresholds.json")))


@pytest.mark.parametrize("case", sorted(GOLD))
def test_threshold_search_matches_reference(case):
    g = GOLD[case]
    p, y = _gen().synth_scores(**g["gen"])
    assert float(p.astype(np.float64).sum()) == g["checksum_scores"] and int(y.sum()) == g["checksum_y"]
    metrics, threshs, rec_values = LT.balance_each_label(p, y)
    assert threshs == g["label_threshs"]                       # exact: same grid point, first maximum
    assert set(metrics) == set(g["balance_each_label"])
    for k, v in g["balance_each_label"].items():
        if np.isnan(v):
            # label without positives: the reference means AUC = 0 there (binary_eval.py:27-31 catches
            # roc_auc_score's ValueError); the scikit-learn in the build container returns nan with a
            # warning instead of raising, so the reference run here recorded nan.  The written intent is kept.
            assert k.endswith("-auc") and (metrics[k] == 0 or "macro" in k), k
            continue
        assert metrics[k] == pytest.approx(v, abs=1e-12), k
    for label in LABEL_TYPES:
        np.testing.assert_allclose(rec_values[label], g["rec_values"][label], atol=1e-12)
    t, m = LT.balanced_f1_micro(p, y)
    assert t == g["micro_thresh"]
    for k, v in g["micro_metrics"].items():
        assert m[k] == pytest.approx(v, abs=1e-12), k
    given = LT.macro_metrics_given_thresholds(p, y, threshs)
    assert set(given) <= set(g["given_thresholds"])
    for k, v in g["given_thresholds"].items():
        if np.isnan(v):      # same scikit-learn drift: the reference drops the AUC rows when one label has none
            continue
        assert given[k] == pytest.approx(v, abs=1e-12), k


@pytest.mark.parametrize("case", sorted(GOLD))
def test_evaluate_scores_matches_reference(case):
    """The multilabel branch of bert.py `evaluate` (:284-315): all_metrics + AUCs, the binary (any-label)
    view at the per-label thresholds, precision at fixed recall."""
    g = GOLD[case]
    p, y = _gen().synth_scores(**g["gen"])
    metrics, label_threshs, micro_thresh, t90, t95, t99 = LT.evaluate_scores(p, y)
    assert label_threshs == g["label_threshs"] and micro_thresh == g["micro_thresh"]
    for k, v in {**g["all_metrics"], **g["binary"]}.items():
        assert (np.isnan(v) and np.isnan(metrics[k])) or metrics[k] == pytest.approx(v, abs=1e-13), k
    for r, t in (("90", t90), ("95", t95), ("99", t99)):
        assert t == g["prec_at_fixed_recall"][r][0]
        assert metrics[f"prec@rec={r}"] == pytest.approx(g["prec_at_fixed_recall"][r][1], abs=1e-13)


def test_roc_auc_ties_and_single_class():
    y = np.array([0, 0, 1, 1])
    assert LT.roc_auc(y, np.array([0.1, 0.4, 0.35, 0.8])) == pytest.approx(0.75)
    assert LT.roc_auc(y, np.array([0.5, 0.5, 0.5, 0.5])) == pytest.approx(0.5)
    with pytest.raises(ValueError):
        LT.roc_auc(np.zeros(4, int), np.array([0.1, 0.2, 0.3, 0.4]))


def test_saved_thresholds_round_trip(tmp_path):
    class FakeScorer:          # host logic only: the GPU forward is covered by tests/test_gpu_scoring.py
        def __init__(self, p):
            self.p = p

        def score(self, rows):
            from clip_amd.scoring import ScoreResult
            n = len(rows)
            return ScoreResult(np.zeros(n, np.int64), np.arange(n), self.p.max(1), np.ones(n, bool), None, self.p)

    g = GOLD["plain"]
    p, y = _gen().synth_scores(**g["gen"])
    out = tmp_path / "label_threshs.npy"
    thr, _ = LT.fit_label_thresholds(FakeScorer(p), [None] * len(p), y, out_path=str(out))
    back = np.load(out)
    assert back.dtype == np.float64 and back.shape == (7,) and list(back) == g["label_threshs"]
    assert np.array_equal(thr, back)
