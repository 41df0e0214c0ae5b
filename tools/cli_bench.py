"""End to end from CSV TEXT on one MI355X: a synthetic notes CSV (documents of 40..400 sentences drawn
from the test vocabulary) -> pandas -> C++ WordPiece -> packed rows -> HIP engine -> ttp_doc_sents.csv +
top-k file, through the CLI itself (clip_amd.apply_bert_unlabeled).   python tools/cli_bench.py [rows]"""
import os
import shutil
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import pandas as pd  # noqa: E402
import torch  # noqa: E402
from clip_amd import apply_bert_unlabeled, synth  # noqa: E402
from clip_amd.config import BertConfig  # noqa: E402
from clip_amd.weights import procedural_state_dict  # noqa: E402
from clip_amd.wordpiece import BertWordPieceTokenizer  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
V = os.path.join(ROOT, "tests", "golden", "vocab.txt")
tmp = tempfile.mkdtemp(prefix="clip_cli_bench_")
try:
    tok = BertWordPieceTokenizer(V)
    words = np.array([w for w in tok.vocab if w.isascii() and w.isalpha() and len(w) > 1], dtype=object)
    rng = np.random.default_rng(0)
    lens = np.clip(np.rint(rng.lognormal(np.log(16), 0.6, N)), 1, 60).astype(int)
    wi = rng.integers(0, len(words), lens.sum())
    off = np.concatenate(([0], np.cumsum(lens)))
    sent = [str(list(words[wi[off[i]:off[i + 1]]])) for i in range(N)]
    doc, six = synth.documents(N, seed=5)
    csv_path = os.path.join(tmp, "notes.csv")
    pd.DataFrame({"doc_id": doc, "sent_index": six, "sentence": sent, "labels": "[]"}).to_csv(csv_path, index=False)
    cfg = BertConfig()
    model_dir = os.path.join(tmp, "model")
    cfg.save_pretrained(model_dir)
    shutil.copyfile(V, os.path.join(model_dir, "vocab.txt"))
    run = os.path.join(tmp, "run")
    os.makedirs(run)
    from safetensors.torch import save_file
    save_file(procedural_state_dict(cfg, 0), os.path.join(run, "w.safetensors"))
    print(f"{N} rows, CSV {os.path.getsize(csv_path) / 1e6:.0f} MB", flush=True)
    for name, args in (("tokenise only (CSV -> columnar dir)", [csv_path, model_dir, "x", "--n_context_sentences", "2",
                                                                "--save_rows", os.path.join(tmp, "rows"), "--tokenize_only"]),
                       ("CSV -> scores + top-k", [csv_path, model_dir, os.path.join(run, "w.safetensors"),
                                                  "--n_context_sentences", "2", "--topk", "50000"]),
                       ("columnar dir -> scores + top-k", [os.path.join(tmp, "rows"), model_dir, os.path.join(run, "w.safetensors"),
                                                           "--n_context_sentences", "2", "--topk", "50000"])):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        apply_bert_unlabeled.main(args)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"{name}: {dt:.2f} s, {N / dt:,.0f} sentences/s", flush=True)
    print("kept rows:", sum(1 for _ in open(os.path.join(run, "ttp_doc_sents.csv"))))
    gpus = int(os.environ.get("CLI_GPUS", "0"))
    if gpus > 1:
        # the multi-rank command line as a user starts it (fresh processes; ranks share the card over gloo on a
        # one-GPU box — a rehearsal of the code path and of the token balance, not a throughput measurement)
        import hashlib
        import subprocess
        one = hashlib.sha256(open(os.path.join(run, "ttp_doc_sents.csv"), "rb").read()).hexdigest()
        one_top = hashlib.sha256(open(os.path.join(run, "ttp_top50000.csv"), "rb").read()).hexdigest()
        env = dict(os.environ, CLIP_AMD_TIMING="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
        if torch.cuda.device_count() < gpus:
            env["CLIP_AMD_FUSED_LN"] = "0"
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
            env.pop(k, None)
        t0 = time.perf_counter()
        p = subprocess.run([sys.executable, "-u", "-m", "clip_amd.apply_bert_unlabeled", csv_path, model_dir,
                            os.path.join(run, "w.safetensors"), "--n_context_sentences", "2", "--topk", "50000",
                            "--gpus", str(gpus)], cwd=ROOT, env=env, capture_output=True, text=True)
        dt = time.perf_counter() - t0
        print(p.stderr[-12000:])
        print(p.stdout[-500:])
        assert p.returncode == 0
        two = hashlib.sha256(open(os.path.join(run, "ttp_doc_sents.csv"), "rb").read()).hexdigest()
        two_top = hashlib.sha256(open(os.path.join(run, "ttp_top50000.csv"), "rb").read()).hexdigest()
        print(f"--gpus {gpus} (CSV -> scores + top-k, process start to exit): {dt:.2f} s; files identical to one rank's "
              f"(fused-LayerNorm flow {'off in the ranks: bits may differ' if env.get('CLIP_AMD_FUSED_LN') == '0' else 'on'}): "
              f"ttp_doc_sents {one == two}, top-k {one_top == two_top}", flush=True)
finally:
    shutil.rmtree(tmp, ignore_errors=True)
