"""SURVEY.md §8(f) rank 3 on the GPU: the labelled-eval consumer of the forward —
sentclf/bert.py:197-232 (gather_predictions) -> :289,615-618 -> multilabel_eval.py:98-122
(balance_each_label) — fed by the REAL BulkScorer (HIP engine), against the same threshold sweep
over the CPU oracle's probabilities."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from clip_amd import label_thresholds as LT, scoring, synth  # noqa: E402
from clip_amd.bert_model import args_to_model  # noqa: E402
from clip_amd.config import BertConfig  # noqa: E402
from clip_amd.constants import LABEL_TYPES  # noqa: E402
from clip_amd.weights import procedural_state_dict  # noqa: E402
from oracle import bert_oracle as O  # noqa: E402  (checker only)


@pytest.mark.parametrize("precision", ["bf16", "f16"])
def test_fit_label_thresholds_through_the_engine(tmp_path, precision):
    cfg = BertConfig(vocab_size=3000, hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
                     intermediate_size=512)
    sd = procedural_state_dict(cfg, 21)
    sd["classifier.weight"] = sd["classifier.weight"] * 6.0           # spread the probabilities over (0, 1)
    rows = synth.context_rows(1500, 2, cfg.vocab_size, max_len=128, seed=61)
    # CPU oracle probabilities, and labels correlated with them (a labelled validation split)
    p_ref, _, _, _ = O.score_rows(sd, cfg, rows, 2, batch_size=50)
    rng = np.random.default_rng(5)
    y = ((p_ref + 0.25 * rng.standard_normal(p_ref.shape)) > np.quantile(p_ref, 0.7, axis=0)).astype(np.int64)
    assert 0.05 < y.mean() < 0.6 and p_ref.std() > 0.1

    model = args_to_model(2)(cfg, max_tokens=32768, max_seqs=2048)
    model.set_precision(precision)
    model.load_state_dict(sd)
    model.to("cuda").eval()
    model.set_sep_token_id(102); model.set_n_context_sentences(2); model.set_task("multilabel")
    scorer = scoring.BulkScorer(model, keep_probs=True)
    out = tmp_path / "label_threshs.npy"
    thr_gpu, metrics_gpu = LT.fit_label_thresholds(scorer, rows, y, out_path=str(out))

    yhat_raw, yhat, y_back = LT.gather_predictions(scorer, rows, y)
    assert np.abs(yhat_raw - p_ref).max() < (1e-2 if precision == "bf16" else 3e-3)    # probabilities; classifier gain 6
    assert np.array_equal(yhat, np.round(yhat_raw)) and np.array_equal(y_back, y)

    metrics_cpu, thr_cpu, _ = LT.balance_each_label(p_ref, y)
    thr_cpu = np.asarray(thr_cpu, np.float64)
    # thresholds equal, except where the oracle's own F1 at the two candidates is within 1e-3 (f16; bf16,
    # whose probabilities move by up to 1e-2 with this x6 classifier: 3e-3 = 2 of 1500 rows changing side) —
    # a plateau of the sweep, where which of two equally good thresholds wins flips with one score
    for j, label in enumerate(LABEL_TYPES):
        if thr_gpu[j] != thr_cpu[j]:
            f1 = LT.threshold_metrics(p_ref[:, j], y[:, j], thresholds=np.array([thr_gpu[j], thr_cpu[j]]))["f1"]
            assert abs(f1[0] - f1[1]) < (3e-3 if precision == "bf16" else 1e-3), (label, thr_gpu[j], thr_cpu[j], f1)
    assert abs(metrics_gpu["balanced-macro-f1"] - metrics_cpu["balanced-macro-f1"]) < 5e-3
    assert abs(metrics_gpu["balanced-macro-auc"] - metrics_cpu["balanced-macro-auc"]) < 5e-3
    # the artefact the scoring path loads (test_load.py:29): float64[7], round trip
    back = np.load(out)
    assert back.dtype == np.float64 and back.shape == (7,) and np.array_equal(back, thr_gpu)
    # ... and the scorer applies it with the reference's strict fp64 compare (test_load.py:37-38)
    res = scoring.BulkScorer(model, label_threshs=back, keep_probs=True).score(rows)
    assert np.array_equal(res.labels, O.label_decisions(res.probs, back))
