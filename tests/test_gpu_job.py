"""Full-size job driver (clip_amd/job.py: configs C2/C4/C5 of BASELINE.json) on one MI355X, through
size-independent properties: chunking and sharding change nothing, the running top-k equals the
top-k of all scores, ordering is the total order of clip_amd.topk."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from clip_amd import job, scoring, topk  # noqa: E402
from clip_amd.bert_model import args_to_model  # noqa: E402
from clip_amd.config import BertConfig  # noqa: E402
from clip_amd.weights import procedural_state_dict  # noqa: E402


def _scorer(n_ctx, cfg):
    m = args_to_model(n_ctx)(cfg, max_tokens=32768, max_seqs=8192)
    m.load_state_dict(procedural_state_dict(cfg, 5))
    m.to("cuda").eval()
    if n_ctx:
        m.set_sep_token_id(102)
        m.set_n_context_sentences(n_ctx)
    m.set_task("multilabel")
    return scoring.BulkScorer(m, any_threshold=0.05)


@pytest.mark.parametrize("config", ["C4", "C2", "C5"])
def test_job_topk_independent_of_chunking_and_world_size(config):
    cfg = BertConfig(vocab_size=3000, hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
                     intermediate_size=512)
    n_ctx = job.CONFIGS[config][0]
    sc = _scorer(n_ctx, cfg)
    N, chunk, k = 30000, 5000, 700
    everything = []
    st, finished = job.run_rank(sc, config, N, chunk, k, 0, 1, cfg.vocab_size,
                                on_chunk=lambda c, n, res: everything.append(res))
    top1, rows, kept = st.top, st.rows, st.kept
    assert finished and rows == N and len(top1.records) == k
    # the running top-k equals the top-k over all scores at once
    d = np.concatenate([r.doc_ids for r in everything]); s = np.concatenate([r.sent_ixs for r in everything])
    m = np.concatenate([r.max_score for r in everything]); kp = np.concatenate([r.keep for r in everything])
    assert kept == int(kp.sum())
    assert np.array_equal(top1.records, topk.local_topk(d, s, m, k, keep=kp))
    # sorted by (score desc, doc_id asc, sent_ix asc); every record is a kept row
    r = top1.records
    key = list(zip(-r["score"].astype(np.float64), r["doc_id"], r["sent_ix"]))
    assert key == sorted(key)
    # two ranks (run one after the other on this GPU), merged: the same records, bit for bit
    parts = [job.run_rank(sc, config, N, chunk, k, rk, 2, cfg.vocab_size)[0].top.records for rk in (0, 1)]
    assert np.array_equal(topk.merge_topk(parts, k), r)
    # a chunk's rows scored inside a different step mix give identical bits (bucketing-invariance)
    rows_c, dd, ss = job.chunk_rows(config, 1, chunk, cfg.vocab_size)
    half = scoring.PackedRows(rows_c.flat[:rows_c.offsets[chunk // 2]], rows_c.offsets[:chunk // 2 + 1])
    a = sc.score(half).max_score
    assert np.array_equal(a, everything[1].max_score[:chunk // 2])


def test_job_cli_single_gpu(tmp_path, capsys):
    """The CLI itself at BERT-base size on a small corpus: JSON summary + top-k CSV."""
    import json
    out = tmp_path / "top.csv"
    glob = job.main(["--config", "C4", "--rows", "6000", "--chunk-rows", "2500", "--topk", "100", "--out", str(out)])
    line = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert line["rows"] == 6000 and line["topk"] == 100 and line["sentences_per_s"] > 0
    txt = out.read_text().splitlines()
    assert txt[0] == "doc_id,sent_ix,max_label_score" and len(txt) == 101
    first = txt[1].split(",")
    assert int(first[0]) == glob[0]["doc_id"] and np.float32(first[2]) == glob[0]["score"]


def test_job_interrupted_and_resumed_selects_the_same_topk(tmp_path, capsys):
    """--state / --max-seconds: three invocations that each stop after one or two chunks, then the finishing one,
    write the same top-k file, byte for byte, as one uninterrupted run (the reference has no resume:
    apply_bert_unlabeled.py:97 rewrites its output from scratch)."""
    import json
    common = ["--config", "C4", "--rows", "9000", "--chunk-rows", "1500", "--topk", "200", "--precision", "bf16"]
    job.main(common + ["--out", str(tmp_path / "one.csv")])
    whole = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    lines = []
    for _ in range(8):
        res = job.main(common + ["--out", str(tmp_path / "split.csv"), "--state", str(tmp_path / "st"), "--max-seconds", "0"])
        lines.append(json.loads(capsys.readouterr().out.strip().splitlines()[-1]))
        if res is not None:
            break
    assert [l.get("finished", True) for l in lines] == [False] * (len(lines) - 1) + [True] and len(lines) >= 3
    last = lines[-1]
    assert last["invocations"] == len(lines) and last["rows"] == whole["rows"] == 9000
    assert last["kept"] == whole["kept"] and last["tokens"] == whole["tokens"]
    assert (tmp_path / "split.csv").read_bytes() == (tmp_path / "one.csv").read_bytes()
    # a state file of another job is refused
    with pytest.raises(ValueError):
        job.main(["--config", "C4", "--rows", "9000", "--chunk-rows", "3000", "--topk", "200", "--precision", "bf16",
                  "--state", str(tmp_path / "st")])
