"""Per-batch timeline of the batches-of-8 loop from a rocprofv3 kernel trace:
    rocprofv3 --kernel-trace --stats -d DIR -o bs8 --output-format csv -- python3 tools/bs8_latency.py 2048 8
    python tools/bs8_trace_report.py DIR/bs8_kernel_trace.csv [--timeline]"""
import re
import sys

import numpy as np
import pandas as pd

t = pd.read_csv(sys.argv[1]).sort_values("Start_Timestamp").reset_index(drop=True)
t["k"] = t.Kernel_Name.map(lambda n: re.sub(r"\(anonymous namespace\)::|void ", "", n)[:64])
t["dur"] = (t.End_Timestamp - t.Start_Timestamp) / 1e3
pz = t.index[t.k.str.startswith("ln_head") | t.k.str.startswith("poison")].tolist()      # last kernel of a forward
lo, hi = len(pz) // 2 + 20, len(pz) - 10          # batches of the timed (second) pass
if "--timeline" in sys.argv:
    w = t.iloc[pz[lo] + 1:pz[lo + 1] + 3]
    base = w.Start_Timestamp.iloc[0]
    for _, r in w.iterrows():
        print(f"{(r.Start_Timestamp - base) / 1e3:9.1f} {r.dur:7.1f}  {r.k}")
n = hi - lo
w = t.iloc[pz[lo] + 1:pz[hi] + 1]
first = t.index[t.k.str.startswith("pack_lens") | t.k.str.startswith("focus_index")]
print(f"period {np.mean(np.diff(t.End_Timestamp.values[pz[lo:hi]])) / 1e3:.1f} us per batch under the profiler; "
      f"GPU busy {w.dur.sum() / n:.1f} us; {len(w) / n:.0f} kernels per batch")
g = w.groupby("k").dur.agg(["count", "sum", "mean"])
g["per_batch_us"] = g["sum"] / n
print(g.sort_values("sum", ascending=False).head(24)[["count", "mean", "per_batch_us"]].to_string())
