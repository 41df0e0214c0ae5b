"""Host logic of the bulk scorer (no GPU): columnar rows, length bucketing, CSV format."""
import numpy as np

from clip_amd import scoring, synth


def test_packed_rows_gather_roundtrip():
    rows = synth.context_rows(50, 2, 5000, max_len=120, seed=3)
    pr = scoring.PackedRows.from_rows(rows)
    idx = np.array([7, 0, 49, 7, 13])
    flat, cu = pr.gather(idx)
    assert cu[0] == 0 and cu[-1] == len(flat)
    for i, r in enumerate(idx):
        assert np.array_equal(flat[cu[i]:cu[i + 1]], rows[r])


def test_plan_steps_covers_everything_once_within_bounds():
    rng = np.random.default_rng(0)
    lens = rng.integers(7, 513, 5000).astype(np.int32)
    order, steps = scoring.plan_steps(lens, 16384, 100)
    assert sorted(order.tolist()) == list(range(5000))
    assert steps[0][0] == 0 and steps[-1][1] == 5000
    for (a, b), (c, d) in zip(steps, steps[1:]):
        assert b == c
    for a, b in steps:
        assert 0 < b - a <= 100 and lens[order[a:b]].sum() <= 16384
    # bucketing: lengths inside a step are sorted, so padding-free attention grids stay tight
    sl = lens[order]
    assert (np.diff(sl) >= 0).all()


def test_ttp_csv_format_matches_reference_writer(tmp_path):
    """apply_bert_unlabeled.py:97-98,124: no header, only kept rows, str(np.float32) score."""
    res = scoring.ScoreResult(doc_ids=np.array([10, 10, 11]), sent_ixs=np.array([0, 1, 0]),
                              max_score=np.array([0.5, 0.01, 0.123456789], np.float32),
                              keep=np.array([True, False, True]), labels=None, probs=None)
    p = tmp_path / "ttp_doc_sents.csv"
    scoring.write_ttp_csv(p, res)
    lines = open(p).read().splitlines()
    assert lines == ["10,0,0.5", "11,0," + str(np.float32(0.123456789))] == ["10,0,0.5", "11,0,0.12345679"]
    d, s, m = scoring.read_ttp_csv(p)
    assert d.tolist() == [10, 11] and s.tolist() == [0, 0] and m[1] == np.float32(0.123456789)


def test_ttp_csv_bytes_equal_the_csv_writer_form(tmp_path):
    """write_ttp_csv builds the file with array operations; byte for byte what the reference's
    csv.writer.writerow([doc_id, sent_ix, np.float32]) loop writes (apply_bert_unlabeled.py:97-98,124)."""
    import csv
    from clip_amd.scoring import ScoreResult, read_ttp_csv, write_ttp_csv
    rng = np.random.default_rng(1)
    n = 5000
    score = np.concatenate([rng.random(n - 8).astype(np.float32),
                            np.float32([1e-5, 1e-7, 0.5, 1.0, 0.05, 3.4e-8, 0.99999994, 1e-10])])
    res = ScoreResult(rng.integers(0, 10**9, n), rng.integers(0, 400, n), score, rng.random(n) < 0.7, None, None)
    write_ttp_csv(tmp_path / "a.csv", res)
    with open(tmp_path / "b.csv", "w", newline="") as f:
        w = csv.writer(f)
        for i in np.nonzero(res.keep)[0]:
            w.writerow([res.doc_ids[i], res.sent_ixs[i], res.max_score[i]])
    assert (tmp_path / "a.csv").read_bytes() == (tmp_path / "b.csv").read_bytes()
    d, s, m = read_ttp_csv(tmp_path / "a.csv")
    assert np.array_equal(m, score[res.keep]) and np.array_equal(d, res.doc_ids[res.keep])
    res.keep[:] = False
    write_ttp_csv(tmp_path / "c.csv", res)
    assert (tmp_path / "c.csv").read_bytes() == b""
