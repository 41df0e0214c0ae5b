"""The N > 1 path as the driver starts it: `bench.py --gpus 2` and `python -m clip_amd.job --gpus 2`
launch their own ranks (clip_amd/launch.py).  With two visible MI355X the ranks use RCCL (backend
"nccl"), one card each; on a one-GPU box both ranks share the card, so the process group is gloo and
the fused-LayerNorm flow (which assumes the card is not shared) is off — the same for the one-rank
run it is compared with.  Replaces the single device of sentclf/apply_bert_unlabeled.py:71."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    if torch.cuda.device_count() < 2:
        env["CLIP_AMD_FUSED_LN"] = "0"
    return env


def _run(args):
    p = subprocess.run([sys.executable, "-u", *args], cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    return json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])


def test_bench_gpus_2_starts_two_ranks():
    line = _run(["bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--rows-per-step", "85",
                 "--no-cpu-baseline"])
    assert line["n_gpus"] == 2 and line["dist"]["world_size"] == 2
    assert line["dist"]["backend"] == ("nccl" if torch.cuda.device_count() >= 2 else "gloo")
    assert line["value"] > 0 and line["scaling"] == "weak"
    # the end-of-job stage at configuration C4's size: one all-gather of 250 001 records per rank, checked in
    # bench.py itself against one sort over both ranks' records
    assert line["topk_c4"]["k"] == 250000 and line["topk_c4"]["equals_one_sort_over_all_records"] is True
    # whole-job value = rows of BOTH ranks over the slowest rank's time
    assert abs(line["value"] - 2 * 85 * 2 / (line["ms_per_step"] * 2e-3)) / line["value"] < 1e-3


def test_job_two_ranks_topk_file_identical_to_one_rank(tmp_path):
    common = ["-m", "clip_amd.job", "--config", "C4", "--rows", "6000", "--chunk-rows", "1500", "--topk", "300",
              "--tokens-per-step", "32768"]
    one = _run(common + ["--out", str(tmp_path / "one.csv")])
    two = _run(common + ["--gpus", "2", "--out", str(tmp_path / "two.csv")])
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert one["rows"] == two["rows"] == 6000 and one["kept"] == two["kept"] and one["tokens"] == two["tokens"]
    a, b = (tmp_path / "one.csv").read_bytes(), (tmp_path / "two.csv").read_bytes()
    assert a == b and a.count(b"\n") == 301


def test_cli_two_ranks_files_identical_to_one_rank(tmp_path):
    """python -m clip_amd.apply_bert_unlabeled --gpus 2: each rank reads the document index, takes its own contiguous
    range of whole documents and tokenises + scores it; CSV parts are concatenated in (rank, piece) order and the
    top-k is one all-gather: ttp_doc_sents.csv and the top-k file must be the one-rank files byte for byte — and so
    must the columnar directory --save_rows leaves (the ranks' parts joined by rank 0)."""
    import shutil
    from clip_amd.config import BertConfig
    from clip_amd.weights import procedural_state_dict
    G = os.path.join(ROOT, "tests", "golden")
    cfg = BertConfig(vocab_size=2000, hidden_size=256, num_hidden_layers=2, num_attention_heads=4, intermediate_size=512)
    model_dir = tmp_path / "model"
    cfg.save_pretrained(str(model_dir))
    shutil.copyfile(os.path.join(G, "vocab.txt"), model_dir / "vocab.txt")
    sd = procedural_state_dict(cfg, 31)
    outs = []
    for name, extra in (("one", []), ("two", ["--gpus", "2"])):
        run = tmp_path / name
        run.mkdir()
        torch.save(sd, run / "w.pth")
        p = subprocess.run([sys.executable, "-u", "-m", "clip_amd.apply_bert_unlabeled", os.path.join(G, "notes_synth.csv"),
                            str(model_dir), str(run / "w.pth"), "--n_context_sentences", "2", "--threshold", "0.3",
                            "--topk", "25", "--tokens_per_step", "4096", "--chunk_rows", "8", "--save_rows", str(run / "rows"),
                            *extra],
                           cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stderr[-3000:]
        assert f"{2 if extra else 1} rank(s)" in p.stdout
        outs.append(((run / "ttp_doc_sents.csv").read_bytes(), (run / "ttp_top25.csv").read_bytes()))
        assert not [f for f in os.listdir(run) if ".chunk" in f or ".part" in f]          # parts merged and removed
        import numpy as np
        outs[-1] += tuple(np.load(run / "rows" / f"{k}.npy").tobytes() for k in ("flat", "offsets", "doc_ids", "sent_ixs"))
    assert outs[0] == outs[1]
    assert outs[0][0].count(b"\n") == 33 and outs[0][1].count(b"\n") == 26     # 33 rows in >= 4 chunks of whole documents


def test_rccl_group_and_fused_flow_in_one_process():
    """One card allows one RCCL rank: the "nccl" process group (size 1), the job's collectives on device memory and the
    production kernel flow (fused-LayerNorm GEMMs ON — the two-rank rehearsals above share a card and must switch them
    off) in the same process.  The N >= 2 RCCL runs themselves are the driver's, on a whole node."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "CLIP_AMD_FUSED_LN")}
    import socket
    with socket.socket() as sk:                      # a free port for the one-rank rendezvous
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    p = subprocess.run([sys.executable, "-u", os.path.join(ROOT, "tests", "rccl_one_rank_child.py")], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["backend"] == "nccl" and line["world"] == 1 and line["fused_layernorm"] is True
    assert line["gather_equals_local_merge"] is True and line["records"] > 0
    assert line["totals"] == line["expected_totals"]
    assert line["same_bits_after_collectives"] is True and line["finite"] is True
