"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes per kernel (MB per launch).
FETCH_SIZE on gfx950 reports half the bytes of wide (16 B/lane) coalesced reads
(MI355X_MICROARCH.md §HBM; confirmed here on layernorm_kernel: 100.7 MB reported for 201.3 MB read),
so fetch is doubled; WRITE_SIZE is exact for 16-B stores.  Usage:
    python tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> out.json [rows_per_step seq_len]"""
import collections, csv, json, re, sys

PAT = re.compile(r"(gemm8_kernel<[^>]*>|gemm_bf16_kernel<[^>]*>|attention_kernel|attention_focus_kernel|"
                 r"layernorm_kernel|embed_ln_kernel)")


def key(name):
    """gemm8_kernel<EPI, OUT_F32, RES_TILED, SINGLE>: the default SINGLE = false is left out, so the keys of
    earlier summaries (three arguments) stay valid"""
    m = re.fullmatch(r"(gemm8_kernel<\d+, \w+, \w+), false(?:, \d+)?>", name)
    return m.group(1) + ">" if m else name


def load(name, f):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        m = PAT.search(r["Kernel_Name"])
        if r["Counter_Name"] == name and m and int(r["Grid_Size"]) >= 65536:
            agg[key(m.group(1))].append(float(r["Counter_Value"]) * 1024)      # KB -> bytes
    return {k: sum(v) / len(v) for k, v in agg.items()}, {k: len(v) for k, v in agg.items()}


F, n = load("FETCH_SIZE", sys.argv[1])
W, _ = load("WRITE_SIZE", sys.argv[2])
out = {k: {"launches": n[k], "fetch_bytes_reported": F[k], "fetch_bytes_corrected_x2": 2 * F[k],
           "write_bytes": W.get(k, 0.0), "hbm_bytes_per_launch": 2 * F[k] + W.get(k, 0.0)} for k in F}
meta = {"rows_per_step": int(sys.argv[4]), "seq_len": int(sys.argv[5])} if len(sys.argv) > 5 else {}
json.dump({**meta, "kernels": out}, open(sys.argv[3], "w"), indent=1)
for k, v in out.items():
    print(f"{k:40s} fetch {v['fetch_bytes_corrected_x2']/1e6:8.1f} MB  write {v['write_bytes']/1e6:8.1f} MB")
