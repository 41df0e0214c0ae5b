"""Do two independent batches on two HIP streams overlap each other's memory-bound and MFMA-bound phases?
Two engines (own weights + workspace), two streams, alternate steps; the persistent GEMMs take
GEMM8_MAX_WGS workgroups (half the CUs) so that the two streams' kernels sit side by side.
    GEMM8_MAX_WGS=128 python tools/dual_stream_bench.py [rows_per_step] [steps] [n_streams]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from clip_amd import _lib, synth  # noqa: E402
from clip_amd.bert_model import BertSequenceMultilabelClassificationContext  # noqa: E402
from clip_amd.config import BertConfig  # noqa: E402
from clip_amd.weights import procedural_state_dict  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 504
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 24
n_streams = int(sys.argv[3]) if len(sys.argv) > 3 else 2
L = 256
cfg = BertConfig()
sd = procedural_state_dict(cfg, 0)
lib = _lib.load()
dev = torch.device("cuda", 0)
streams = [torch.cuda.Stream() for _ in range(n_streams)]
models, batches = [], []
for k in range(n_streams):
    m = BertSequenceMultilabelClassificationContext(cfg, max_tokens=rows * L, max_seqs=rows)
    m.load_state_dict(sd)
    m.to(dev).eval()
    m.set_sep_token_id(102); m.set_n_context_sentences(2); m.set_task("multilabel")
    models.append(m)
    rs = synth.context_rows(rows, 2, cfg.vocab_size, seed=1002 + k, fixed_len=L)
    lens = np.asarray([len(r) for r in rs], np.int32)
    t = lambda a: torch.from_numpy(a).to(dev)
    T = int(lens.sum())
    batches.append({"ids": t(np.concatenate(rs).astype(np.int32)), "cu": t(np.concatenate(([0], np.cumsum(lens))).astype(np.int32)),
                    "n": rows, "T": T, "tt": torch.empty(T, dtype=torch.int32, device=dev),
                    "pos": torch.empty(T, dtype=torch.int32, device=dev), "focus": torch.empty(rows, dtype=torch.int32, device=dev),
                    "logits": torch.empty((rows, 7), dtype=torch.float32, device=dev)})
err = torch.zeros(1, dtype=torch.int32, device=dev)
engines = [m._ensure_engine() for m in models]


def step(k):
    b = batches[k]
    with torch.cuda.stream(streams[k]):
        st = _lib.stream_ptr()
        _lib.check(lib.clip_prepare_packed(_lib.ptr(b["ids"]), _lib.ptr(b["cu"]), b["n"], 102, 2, _lib.ptr(b["tt"]),
                                           _lib.ptr(b["pos"]), _lib.ptr(b["focus"]), _lib.ptr(err), st))
        _lib.check(lib.clip_forward_packed(engines[k], _lib.ptr(b["ids"]), _lib.ptr(b["tt"]), _lib.ptr(b["pos"]),
                                           _lib.ptr(b["cu"]), _lib.ptr(b["focus"]), b["n"], b["T"], L, 0,
                                           _lib.ptr(b["logits"]), st))


for i in range(2 * n_streams):
    step(i % n_streams)
torch.cuda.synchronize()
ref = [b["logits"].clone() for b in batches]
t0 = time.perf_counter()
for i in range(steps):
    step(i % n_streams)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
flags = []
for k in range(n_streams):
    f = __import__("ctypes").c_int32(0)
    _lib.check(lib.clip_poll_errors(engines[k], __import__("ctypes").byref(f)))
    flags.append(f.value)
same = all(torch.equal(r, b["logits"]) for r, b in zip(ref, batches))
print(f"streams {n_streams} GEMM8_MAX_WGS={os.environ.get('GEMM8_MAX_WGS')} rows/step {rows}: {steps * rows / dt:,.0f} sent/s, "
      f"{dt / steps * 1e3:.2f} ms/step, error flags {flags}, results stable {same}", flush=True)
