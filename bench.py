"""bench.py — sentences/sec of the bulk BERT-base+Context scoring path on MI355X.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one resident batch of synthetic context-packed rows
(config 3 of BASELINE.json, fixed-length variant: every row has exactly --seq-len tokens):
focus-[SEP] index -> 12-layer encoder -> classifier -> fused sigmoid/threshold/max epilogue.
Inputs are already in HBM when the timed region starts.  Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from clip_amd import _lib, synth  # noqa: E402
from clip_amd.bert_model import BertSequenceMultilabelClassificationContext  # noqa: E402
from clip_amd.config import BertConfig  # noqa: E402
from clip_amd.weights import procedural_state_dict  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0   # dense bf16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_GBPS = 8000.0      # HBM3E, same guide
PMC_FILE = "r03_pmc_traffic.json"
DNOTE_THRESHS = np.array([0.44, 0.31, 0.24, 0.17, 0.46, 0.35000000000000003, 0.43])


def flops_per_row(L, cfg):
    """F(L) = layers*[2L(4H^2 + 2HI) + 4L^2 H] + 2*H*labels (real tokens only)."""
    H, I = cfg.hidden_size, cfg.intermediate_size
    return cfg.num_hidden_layers * (2 * L * (4 * H * H + 2 * H * I) + 4 * L * L * H) + 2 * H * cfg.num_labels


def flops_saved_per_row(L, cfg):
    """Last-layer trimming: only the pooled row's q, attention, output projection and FFN
    are computed there; K and V still need every token."""
    H, I = cfg.hidden_size, cfg.intermediate_size
    return 2 * (L - 1) * (2 * H * H + 2 * H * I) + 4 * L * H * (L - 1)


def roofline_by_kernel(kernel_ms, lens, cfg, precision, steps):
    """Every kernel kind of one step against ITS bound (north_star: MFMA fraction for the encoder GEMMs, HBM GB/s
    for the memory-bound kernels): algorithmic FLOPs and bytes per step (DESIGN.md §4's table: operands read
    once, outputs written once, real tokens only), the HIP-event time of the kind per step, and the fraction of
    whichever roofline takes longer for that work (2.5 PF dense MFMA vs 8 TB/s HBM)."""
    H, I, NL = cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers
    T, n = int(lens.sum()), len(lens)
    full = NL - 1                                        # layers run on every token (the last one is trimmed)
    ln_copies = {"bf16": 2, "f16": 1}.get(precision, 0)  # LayerNorm output copies a fused GEMM writes (16-bit each)
    unfused = ln_copies == 0                             # f16r: fp32 pre-LayerNorm rows out, fp32 rows + stats in
    ln_io = (T * H * 2 * (1 + ln_copies)) if not unfused else (T * H * 4 * 2)
    work = {
        "embed_ln": (0.0, T * H * 4 + T * H * 2 * max(ln_copies, 1)),
        "gemm_qkv": (full * 2.0 * T * H * 3 * H + 2.0 * T * H * 2 * H,
                     full * (T * H * 2 + 3 * H * H * 2 + T * 3 * H * 2) + (T * H * 2 + 2 * H * H * 2 + T * 2 * H * 2)),
        "attention": (full * float(sum(4.0 * int(L) * int(L) * H for L in lens)), full * (T * 3 * H * 2 + T * H * 2)),
        "gemm_attn_out": (full * 2.0 * T * H * H, full * (T * H * 2 + H * H * 2 + ln_io)),
        "gemm_ffn_up": (full * 2.0 * T * H * I, full * (T * H * 2 + I * H * 2 + T * I * 2)),
        "gemm_ffn_down": (full * 2.0 * T * I * H, full * (T * I * 2 + I * H * 2 + ln_io)),
        "layernorm": (0.0, 2 * full * (T * H * 4 + T * H * 2) if unfused else 0),
        "last_layer_rows": (2.0 * n * (2 * H * H + 2 * H * I) + float(sum(4.0 * int(L) * H for L in lens)),
                            T * 2 * H * 2 + (2 * H * H + 2 * H * I) * 2),
        "head": (2.0 * n * H * cfg.num_labels, n * H * 2 + n * cfg.num_labels * 4),
    }
    out = {}
    for kind, (flops, nbytes) in work.items():
        ms = kernel_ms.get(kind, 0.0)
        if ms <= 0:
            continue
        tf, gbps = flops / (ms * 1e-3) / 1e12, nbytes / (ms * 1e-3) / 1e9
        bound = "mfma" if flops / (PEAK_BF16_TFLOPS * 1e12) > nbytes / (PEAK_HBM_GBPS * 1e9) else "hbm"
        out[kind] = {"ms_per_step": round(ms, 4), "gflop_per_step": round(flops / 1e9, 2), "mb_per_step": round(nbytes / 1e6, 1),
                     "tflops": round(tf, 1), "gbps": round(gbps, 1), "bound": bound,
                     "frac": round(tf / PEAK_BF16_TFLOPS if bound == "mfma" else gbps / PEAK_HBM_GBPS, 4)}
    return out


def make_batch(rows, device):
    """What the host ships per step: packed token ids + row offsets (everything else is derived on
    the device by clip_prepare_packed)."""
    lens = np.asarray([len(r) for r in rows], np.int32)
    ids = np.concatenate(rows).astype(np.int32)
    cu = np.concatenate(([0], np.cumsum(lens))).astype(np.int32)
    t = lambda a: torch.from_numpy(a).to(device)
    T, n = int(lens.sum()), len(rows)
    return {"ids": t(ids), "cu": t(cu), "max_len": int(lens.max()), "n": n, "tokens": T, "lens": lens,
            "tt": torch.empty(T, dtype=torch.int32, device=device),
            "pos": torch.empty(T, dtype=torch.int32, device=device),
            "focus": torch.empty(n, dtype=torch.int32, device=device),
            "logits": torch.empty((n, 7), dtype=torch.float32, device=device)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows-per-step", type=int, default=510)
    ap.add_argument("--seq-len", type=int, default=256)
    ap.add_argument("--precision", choices=["bf16", "f16", "f16r"], default="bf16",
                    help="operand type of the kernels (clip_set_precision); the headline metric is bf16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--breakdown", action="store_true", help="also time every kernel kind (extra events)")
    args = ap.parse_args()

    # --gpus N without torchrun: this process starts the N ranks itself (fresh processes, before any
    # GPU call here) and only relays their exit status; rank 0 of the children prints the JSON line
    from clip_amd import launch
    rc = launch.spawn_ranks_if_needed(args.gpus)
    if rc is not None:
        sys.exit(rc)
    dist, rank, world, device = launch.init_distributed()

    cfg = BertConfig()
    n_ctx = 2
    sd = procedural_state_dict(cfg, 0)
    tokens = args.rows_per_step * args.seq_len
    model = BertSequenceMultilabelClassificationContext(cfg, max_tokens=tokens, max_seqs=args.rows_per_step)
    model.load_state_dict(sd)
    model.set_precision(args.precision)
    model.to(device).eval()
    model.set_sep_token_id(102)
    model.set_n_context_sentences(n_ctx)
    model.set_task("multilabel")
    lib = _lib.load()
    engine = model._ensure_engine()

    # two resident batches per rank (documents are sharded by rank: different seeds = different rows)
    all_rows = synth.context_rows(2 * args.rows_per_step, n_ctx, cfg.vocab_size, seed=1002 + 17 * rank,
                                  fixed_len=args.seq_len)
    batches = [make_batch(all_rows[i * args.rows_per_step:(i + 1) * args.rows_per_step], device)
               for i in range(2)]
    err = torch.zeros(1, dtype=torch.int32, device=device)
    thr = torch.tensor(DNOTE_THRESHS, dtype=torch.float64, device=device)
    n = args.rows_per_step
    out = {"max": torch.empty(n, device=device), "keep": torch.empty(n, dtype=torch.uint8, device=device),
           "labels": torch.empty((n, cfg.num_labels), dtype=torch.uint8, device=device)}
    stream = _lib.stream_ptr()

    def step(b):
        # the hot path of one step, all on the current stream, nothing allocated:
        # ids -> (type ids, positions, focus [SEP]) -> encoder + classifier -> sigmoid/threshold/max
        _lib.check(lib.clip_prepare_packed(_lib.ptr(b["ids"]), _lib.ptr(b["cu"]), b["n"], 102, n_ctx,
                                           _lib.ptr(b["tt"]), _lib.ptr(b["pos"]), _lib.ptr(b["focus"]),
                                           _lib.ptr(err), stream))
        _lib.check(lib.clip_forward_packed(engine, _lib.ptr(b["ids"]), _lib.ptr(b["tt"]), _lib.ptr(b["pos"]),
                                           _lib.ptr(b["cu"]), _lib.ptr(b["focus"]), b["n"], b["tokens"],
                                           b["max_len"], 0, _lib.ptr(b["logits"]), stream))
        _lib.check(lib.clip_epilogue(_lib.ptr(b["logits"]), _lib.ptr(thr), 0.05, b["n"], cfg.num_labels, None,
                                     _lib.ptr(out["max"]), _lib.ptr(out["labels"]), _lib.ptr(out["keep"]), stream))

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for i in range(args.warmup):
        step(batches[i % 2])
    barrier()
    dominant = _lib.KINDS.index("gemm_ffn_up")
    mask = (1 << len(_lib.KINDS)) - 1 if args.breakdown else (1 << dominant)
    _lib.check(lib.clip_profile_enable(engine, mask))
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(batches[i % 2])
    barrier()
    elapsed = time.perf_counter() - t0
    ms = (C.c_double * len(_lib.KINDS))()
    cnt = (C.c_int64 * len(_lib.KINDS))()
    _lib.check(lib.clip_profile_read(engine, ms, cnt, 1))
    _lib.check(lib.clip_profile_enable(engine, 0))
    # device-side error words after the timed loop, on every rank: a non-zero word voids the run
    flags = C.c_int32(0)
    _lib.check(lib.clip_poll_errors(engine, C.byref(flags)))
    if flags.value or int(err.item()):
        raise RuntimeError(f"rank {rank}: engine error word {flags.value}, [SEP]-count word {int(err.item())} "
                           "after the timed loop: results void")
    # sustained MFMA peak of THIS board (register-only MFMA loop, clip_mfma_peak), measured right after the
    # timed steps: the nameplate 2.5 PF assumes 2.4 GHz, which no board holds under a dense MFMA stream
    peak_best, peak_mean, peak_ms = C.c_double(0), C.c_double(0), C.c_double(0)
    if rank == 0:
        _lib.check(lib.clip_mfma_peak(_lib.PRECISION_F16 if args.precision.startswith("f16") else _lib.PRECISION_BF16, 12000, 5, C.byref(peak_best), C.byref(peak_mean),
                                      C.byref(peak_ms)))
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # end-of-job stage (outside the timed steps): local top-k + ONE all-gather + deterministic merge
    from clip_amd import topk
    step(batches[0])
    torch.cuda.synchronize()
    scores = out["max"].cpu().numpy()
    t_k = time.perf_counter()
    local = topk.local_topk(np.full(n, rank, np.int64), np.arange(n), scores, 64, keep=out["keep"].cpu().numpy() != 0)
    glob = topk.global_topk(local, 64)
    topk_ms = (time.perf_counter() - t_k) * 1e3

    # the same stage at configuration C4's size (BASELINE.json configs[3]: global top-250k): 400 k synthetic
    # records per rank with many tied scores -> local top-250k -> ONE all-gather of 250 001 16-byte records per
    # rank (4 MB each: RCCL over xGMI when N GPUs are visible) -> the same merge on every rank; rank 0 checks
    # the result against one sort over every rank's records.
    K4, N4 = 250_000, 400_000

    def c4_records(r):
        g4 = np.random.default_rng(4242 + r)
        return (r * 10_000_000 + np.arange(N4, dtype=np.int64) // 40, np.arange(N4, dtype=np.int64) % 40,
                np.round(g4.random(N4, dtype=np.float32), 3).astype(np.float32))       # ~1000 distinct scores: ties
    d4, s4, m4 = c4_records(rank)
    if dist is not None:
        dist.barrier()
    t_k = time.perf_counter()
    glob4 = topk.global_topk(topk.local_topk(d4, s4, m4, K4), K4)
    topk4_ms = (time.perf_counter() - t_k) * 1e3

    if rank == 0:
        rows_total = world * args.rows_per_step * args.steps
        value = rows_total / elapsed
        T = batches[0]["tokens"]
        H, I = cfg.hidden_size, cfg.intermediate_size
        gemm_flops = 2.0 * T * H * I                       # FFN-up GEMM, one launch
        avg_ms = ms[dominant] / max(cnt[dominant], 1)
        achieved = gemm_flops / (avg_ms * 1e-3) / 1e12
        model_flops = sum(flops_per_row(int(L), cfg) for L in batches[0]["lens"])
        # HBM bytes per launch of the dominant kernel: from separate rocprofv3 --pmc passes
        # (FETCH_SIZE x2 + WRITE_SIZE, tools/pmc_summary.py), kept under profiles/; null if absent or
        # collected at another step size
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", PMC_FILE)) as fh:
                pmc = json.load(fh)
            if pmc.get("rows_per_step") == args.rows_per_step and pmc.get("seq_len") == args.seq_len:
                traffic = pmc["kernels"]["gemm8_kernel<1, false, false>"]["hbm_bytes_per_launch"]
        except (OSError, KeyError, ValueError):
            pass
        mfma_busy = None      # MFMA busy cycles / kernel cycles of the dominant kernel, from an SQ PMC pass kept under profiles/
        try:
            with open(os.path.join(ROOT, "profiles", "r03_pmc_mfma_util.json")) as fh:
                mfma_busy = json.load(fh)["gemm8_kernel<1, false, false>"]["mfma_utilisation"]
        except (OSError, KeyError, ValueError):
            pass
        res = {
            "metric": "sentences/sec scored (whole node) BERT-base+Context seq256 " + args.precision,
            "value": round(value, 2), "unit": "sentences/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.precision,
            "data": "synthetic token ids (uniform in [1000, vocab)), procedural N(0,0.02) weights",
            "config": {"workload": f"C3 BERT-base+Context n_ctx=2, fixed L={args.seq_len}, "
                                   f"{args.rows_per_step} rows ({T} tokens) per step per GPU",
                       "rows_per_step": args.rows_per_step, "seq_len": args.seq_len,
                       "sharding": f"documents over {world} rank(s), no collective while scoring"},
            "model_tflops_per_gpu": round(model_flops * args.steps / elapsed / 1e12, 2),
            "executed_tflops_per_gpu": round((model_flops - sum(flops_saved_per_row(int(L), cfg) for L in batches[0]["lens"]))
                                             * args.steps / elapsed / 1e12, 2),
            "roofline": {"bound": "mfma", "kernel": "gemm8_kernel<EPI_BIAS_GELU,bf16> (FFN up, M=%d N=%d K=%d; layers 1-11, the trimmed last layer is timed apart)" % (T, I, H),
                         "achieved": round(achieved, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK_BF16_TFLOPS, 4), "traffic": traffic,
                         "peak_sustained": round(peak_mean.value, 1),
                         "frac_sustained": round(achieved / peak_mean.value, 4),
                         "peak_sustained_note": "clip_mfma_peak: register-only v_mfma_f32_16x16x32_bf16 loop, random operands, "
                                                "all CUs x 2 waves/SIMD, mean of 5 launches of %.2f ms (best %.1f TF); implied "
                                                "clock %.2f GHz at 16 cycles per MFMA" % (
                                                    peak_ms.value, peak_best.value,
                                                    peak_mean.value * 1e12 / (torch.cuda.get_device_properties(device).multi_processor_count * 4 * 1024.0) / 1e9),
                         "mfma_busy_frac_pmc": mfma_busy,
                         "traffic_note": "HBM-side bytes/launch from PMC passes (profiles/%s); algorithmic operand+output bytes = %d" % (PMC_FILE, 2 * T * H + 2 * I * H + 2 * T * I),
                         "avg_launch_ms": round(avg_ms, 5), "launches": int(cnt[dominant]),
                         "flops_per_launch": gemm_flops},
        }
        res["dist"] = {"world_size": dist.get_world_size() if dist is not None else 1,
                       "backend": dist.get_backend() if dist is not None else None,
                       "launcher": "torchrun" if "TORCHELASTIC_RUN_ID" in os.environ else ("self" if world > 1 else None),
                       "gpus_visible": torch.cuda.device_count()}
        res["topk"] = {"k": 64, "records": int(len(glob)), "ms_local_select_allgather_merge": round(topk_ms, 3)}
        every = [c4_records(r) for r in range(world)]
        want4 = topk.local_topk(np.concatenate([e[0] for e in every]), np.concatenate([e[1] for e in every]),
                                np.concatenate([e[2] for e in every]), K4)
        res["topk_c4"] = {"k": K4, "records_per_rank": N4, "allgather_bytes_per_rank": (K4 + 1) * 16,
                          "ms_local_select_allgather_merge": round(topk4_ms, 1),
                          "equals_one_sort_over_all_records": bool(glob4.tobytes() == want4.tobytes())}
        if not res["topk_c4"]["equals_one_sort_over_all_records"]:
            raise RuntimeError("global top-k over %d rank(s) differs from one sort over all records" % world)
        if args.breakdown:
            res["kernel_ms_per_step"] = {k: round(ms[i] / args.steps, 4) for i, k in enumerate(_lib.KINDS)}
            res["roofline_by_kernel"] = roofline_by_kernel({k: ms[i] / args.steps for i, k in enumerate(_lib.KINDS)},
                                                           batches[0]["lens"], cfg, args.precision, args.steps)
        if world == 1 and not args.no_cpu_baseline:
            # (skipped with --no-cpu-baseline: profile runs must see the benchmark's own launches only)
            # what precision='auto' (the default everywhere but here: the headline metric is bf16) selects for these
            # weights, with the numbers it measured (clip_amd/precision.py), outside the timed region
            probe = BertSequenceMultilabelClassificationContext(cfg, max_tokens=32768, max_seqs=1024)
            probe._sd = model._sd
            probe.to(device)
            probe.set_sep_token_id(102); probe.set_n_context_sentences(n_ctx)
            probe.calibrate_precision(all_rows, log=False)
            rep = probe.precision_report
            res["precision_auto"] = {"selected": rep["selected"], "meets_criteria": rep["meets_criteria"],
                                     "sample_rows": rep["sample_rows"], "seconds": rep["seconds"],
                                     "candidates": {k: {kk: vv for kk, vv in v.items() if kk != "rms_logit_err"}
                                                    for k, v in rep["candidates"].items()}}
            probe._drop_engine()
        if world == 1 and not args.no_cpu_baseline:
            from oracle import bert_oracle as O    # CPU baseline leg only
            sample = all_rows[:args.rows_per_step]
            t1 = time.perf_counter()
            _, keep_ref, score_ref, logits_ref = O.score_rows(sd, cfg, sample, n_ctx, batch_size=8)
            dt = time.perf_counter() - t1
            res["cpu_baseline"] = {
                "value": round(len(sample) / dt, 3), "unit": "sentences/s", "cores": torch.get_num_threads(),
                "kind": "port", "sample": f"first {len(sample)} rows of the same workload, batch 8 padded "
                f"(apply_bert_unlabeled.py:92 semantics), torch {torch.__version__} fp32 CPU oracle, {dt:.1f} s"}
            # the same rows through the GPU path, as a live parity check beside the numbers
            step(batches[0])
            torch.cuda.synchronize()
            assert int(err.item()) == 0
            g = batches[0]["logits"].cpu().numpy()[:len(sample)]
            res["parity_vs_cpu_oracle_max_abs_logit"] = float(np.abs(g - logits_ref).max())
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
