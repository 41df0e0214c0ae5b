"""End to end on an MI355X: CSV text -> datasets -> WordPiece -> packed rows -> HIP engine ->
ttp_doc_sents.csv / top-k file, against the CPU oracle driven by the reference-semantics collators."""
import csv
import os
import shutil

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from clip_amd import apply_bert_unlabeled, collate, datasets  # noqa: E402
from clip_amd.config import BertConfig  # noqa: E402
from clip_amd.weights import procedural_state_dict  # noqa: E402
from clip_amd.wordpiece import BertWordPieceTokenizer  # noqa: E402
from oracle import bert_oracle as O  # noqa: E402  (checker only)

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("n_ctx", [2, 0])
def test_cli_matches_oracle(tmp_path, n_ctx):
    cfg = BertConfig(vocab_size=2000, hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
                     intermediate_size=512)
    model_dir = tmp_path / "model"
    cfg.save_pretrained(str(model_dir))
    shutil.copyfile(os.path.join(G, "vocab.txt"), model_dir / "vocab.txt")
    sd = procedural_state_dict(cfg, 31)
    run_dir = tmp_path / "run"
    run_dir.mkdir()
    torch.save(sd, run_dir / "model_best_f1.pth")
    path = os.path.join(G, "notes_synth.csv")
    apply_bert_unlabeled.main([path, str(model_dir), str(run_dir / "model_best_f1.pth"),
                               "--n_context_sentences", str(n_ctx), "--threshold", "0.5", "--topk", "10",
                               "--tokens_per_step", "4096"])
    got = [(int(r[0]), int(r[1]), np.float32(r[2])) for r in csv.reader(open(run_dir / "ttp_doc_sents.csv"))]

    # oracle: the reference's loop (batches of 8, padded) with the reference-semantics collators
    tok = BertWordPieceTokenizer(os.path.join(G, "vocab.txt"))
    if n_ctx:
        ds = datasets.SentEvalDataset(path, "multilabel", n_context_sentences=n_ctx)
    else:
        ds = datasets.SentDataset(path, "multilabel")
    exp = []
    probs_all = []
    with torch.no_grad():
        for i in range(0, len(ds), 8):
            items = [ds[j] for j in range(i, min(i + 8, len(ds)))]
            if n_ctx:
                x = collate.ctx_collator(items, "multilabel", tok)
                _, logits = O.context_forward(sd, cfg, x["input_ids"], x["attention_mask"], x["token_type_ids"],
                                              x["labels"], tok.sep_token_id, n_ctx)
            else:
                x = collate.collator(items, "multilabel", tok)
                _, logits = O.sentence_forward(sd, cfg, x["input_ids"], x["attention_mask"], None, x["labels"])
            p, keep, score = O.score_epilogue(logits, 0.5)
            probs_all.append(p)
            for k in range(len(items)):
                exp.append((int(x["doc_ids"][k]), int(x["sent_ixs"][k]), score[k], bool(keep[k]), p[k]))
    # every row the oracle keeps with margin must be there, in input order, with the same score +-2e-3
    got_keys = [(d, s) for d, s, _ in got]
    pos = {}
    for i, key in enumerate(got_keys):
        pos.setdefault(key, []).append(i)
    last = -1
    for d, s, score, keep, p in exp:
        margin = np.abs(p - np.float32(0.5)).min() > 2e-3
        if keep and margin:
            assert (d, s) in pos, (d, s)
            i = pos[(d, s)].pop(0)
            assert i > last
            last = i
            assert abs(float(got[i][2]) - float(score)) < 2e-3
        if (not keep) and margin and (d, s) in pos:
            # duplicated (doc, sent_ix) keys exist in the fixture; only flag a miss when none remains
            pass
    # top-k file: header the pre-training reader expects (pretrain_tasks.py:57-58), k rows, sorted
    rows = list(csv.reader(open(run_dir / "ttp_top10.csv")))
    assert rows[0] == ["doc_id", "sent_ix", "max_label_score"] and len(rows) == 11
    sc = [float(r[2]) for r in rows[1:]]
    assert sc == sorted(sc, reverse=True)


def test_cli_columnar_round_trip_and_chunking(tmp_path):
    """Tokenise once (--save_rows, --tokenize_only), score from the columnar directory in small chunks:
    byte-identical files to the one-shot CSV run."""
    cfg = BertConfig(vocab_size=2000, hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
                     intermediate_size=512)
    model_dir = tmp_path / "model"
    cfg.save_pretrained(str(model_dir))
    shutil.copyfile(os.path.join(G, "vocab.txt"), model_dir / "vocab.txt")
    sd = procedural_state_dict(cfg, 31)
    outs = []
    path = os.path.join(G, "notes_synth.csv")
    rows_dir = tmp_path / "rows"
    apply_bert_unlabeled.main([path, str(model_dir), "unused.pth", "--n_context_sentences", "2",
                               "--save_rows", str(rows_dir), "--tokenize_only"])
    for name, src, extra in (("csv", path, []), ("col", str(rows_dir), ["--chunk_rows", "7"])):
        run_dir = tmp_path / name
        run_dir.mkdir()
        torch.save(sd, run_dir / "w.pth")
        apply_bert_unlabeled.main([src, str(model_dir), str(run_dir / "w.pth"), "--n_context_sentences", "2",
                                   "--threshold", "0.5", "--topk", "10", "--tokens_per_step", "4096"] + extra)
        outs.append((open(run_dir / "ttp_doc_sents.csv").read(), open(run_dir / "ttp_top10.csv").read()))
        assert not [f for f in os.listdir(run_dir) if ".chunk" in f]
    assert outs[0] == outs[1] and len(outs[0][0]) > 0
    with pytest.raises(ValueError):
        apply_bert_unlabeled.main([str(rows_dir), str(model_dir), "w.pth", "--n_context_sentences", "1"])


def test_cli_scores_every_row_including_trimmed_and_backfilled(tmp_path):
    """--threshold -1 keeps every row: one score per input row, in input order, each within 2e-3 of the
    oracle on the reference-semantics collator tensors — including the rows that went through the trim loop
    and the back-fill of bert.py:114-145 (focus sentence alone >= 512 tokens)."""
    cfg = BertConfig(vocab_size=2000, hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
                     intermediate_size=512)
    model_dir = tmp_path / "model"
    cfg.save_pretrained(str(model_dir))
    shutil.copyfile(os.path.join(G, "vocab.txt"), model_dir / "vocab.txt")
    sd = procedural_state_dict(cfg, 33)
    run_dir = tmp_path / "run"
    run_dir.mkdir()
    torch.save(sd, run_dir / "w.pth")
    path = os.path.join(G, "notes_synth.csv")
    apply_bert_unlabeled.main([path, str(model_dir), str(run_dir / "w.pth"), "--n_context_sentences", "2",
                               "--threshold", "-1", "--tokens_per_step", "4096"])
    got = [(int(r[0]), int(r[1]), np.float32(r[2])) for r in csv.reader(open(run_dir / "ttp_doc_sents.csv"))]
    tok = BertWordPieceTokenizer(os.path.join(G, "vocab.txt"))
    ds = datasets.SentEvalDataset(path, "multilabel", n_context_sentences=2)
    assert len(got) == len(ds)
    n_long = n_backfilled = 0
    k = 0
    with torch.no_grad():
        for i in range(0, len(ds), 8):
            items = [ds[j] for j in range(i, min(i + 8, len(ds)))]
            x = collate.ctx_collator(items, "multilabel", tok)
            for it in items:
                s, rep = collate._fit_context(" ".join(it[0]).lower(), tuple(c.lower() for c in it[5]), tok)
                n_backfilled += rep is not None
            n_long += int((x["attention_mask"].sum(1) >= 511).sum())
            _, logits = O.context_forward(sd, cfg, x["input_ids"], x["attention_mask"], x["token_type_ids"],
                                          x["labels"], tok.sep_token_id, 2)
            _, _, score = O.score_epilogue(logits, -1.0)
            for j in range(len(items)):
                assert got[k][:2] == (int(x["doc_ids"][j]), int(x["sent_ixs"][j]))
                assert abs(float(got[k][2]) - float(score[j])) < 2e-3, (k, got[k], score[j])
                k += 1
    assert n_long >= 1 and n_backfilled >= 1      # the fixture does exercise those branches


def test_cli_failure_in_scoring_ends_the_job(tmp_path, monkeypatch):
    """An exception while a chunk is scored must end the CLI with that exception — not leave the tokeniser thread
    blocked on its queue (and the process alive) behind it."""
    import threading
    from clip_amd import scoring
    cfg = BertConfig(vocab_size=2000, hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
                     intermediate_size=512)
    model_dir = tmp_path / "model"
    cfg.save_pretrained(str(model_dir))
    shutil.copyfile(os.path.join(G, "vocab.txt"), model_dir / "vocab.txt")
    run_dir = tmp_path / "run"
    run_dir.mkdir()
    torch.save(procedural_state_dict(cfg, 31), run_dir / "w.pth")
    calls = []
    real = scoring.BulkScorer.score

    def flaky(self, *a, **k):
        calls.append(1)
        if len(calls) == 2:
            raise RuntimeError("injected")
        return real(self, *a, **k)

    monkeypatch.setattr(scoring.BulkScorer, "score", flaky)
    before = threading.active_count()
    with pytest.raises(RuntimeError, match="injected"):
        apply_bert_unlabeled.main([os.path.join(G, "notes_synth.csv"), str(model_dir), str(run_dir / "w.pth"),
                                   "--n_context_sentences", "2", "--chunk_rows", "5", "--tokens_per_step", "4096"])
    import time
    for _ in range(50):                 # the producer notices within its 0.2 s put timeout
        if threading.active_count() <= before:
            break
        time.sleep(0.1)
    assert threading.active_count() <= before
