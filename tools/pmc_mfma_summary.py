"""MFMA utilisation and wave-state shares per kernel from one rocprofv3 --pmc pass:
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY \
              SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv ...
    python tools/pmc_mfma_summary.py run_counter_collection.csv out.json
MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles x SIMDs), kernel cycles = GRBM_GUI_ACTIVE / 8
(the counter sums the 8 XCDs; MI355X_MICROARCH.md, DVFS), SIMDs = 4 x CUs.  Against the 2.5 PF nameplate the
fraction is that utilisation x (clock held / 2.4 GHz)."""
import collections
import csv
import json
import re
import sys

CUS = 256
PAT = re.compile(r"(gemm8_kernel<[^>]*>|gemm_bf16_kernel<[^>]*>|attention_kernel|attention_focus_kernel|layernorm_kernel|embed_ln_kernel)")


def key(name):     # as tools/pmc_summary.py: the default SINGLE = false template argument is left out of the key
    m = re.fullmatch(r"(gemm8_kernel<\d+, \w+, \w+), false(?:, \d+)?>", name)
    return m.group(1) + ">" if m else name


agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    m = PAT.search(r["Kernel_Name"])
    if not m or int(r["Grid_Size"]) < 65536:
        continue
    agg[key(m.group(1))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur[key(m.group(1))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
out = {}
for k, v in agg.items():
    med = {c: sorted(x)[len(x) // 2] for c, x in v.items()}
    cyc = med["GRBM_GUI_ACTIVE"] / 8.0
    wc = max(med["SQ_WAVE_CYCLES"], 1.0)
    ns = sorted(dur[k])[len(dur[k]) // 2]
    out[k] = {"launches": len(next(iter(v.values()))), "kernel_cycles": cyc, "median_ns_under_profiler": ns,
              "clock_ghz": round(cyc / ns, 3),
              "mfma_utilisation": round(med["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 4 * CUS), 4),
              "waves_parked_share": round(med["SQ_WAIT_ANY"] / wc, 3),
              "waves_issue_stalled_share": round(med["SQ_WAIT_INST_ANY"] / wc, 3),
              "waves_issuing_share": round(med["SQ_ACTIVE_INST_ANY"] / wc, 3),
              "lds_conflict_cycles_per_lds_cycle": round(med["SQ_LDS_BANK_CONFLICT"] / max(med["SQ_LDS_IDX_ACTIVE"], 1.0), 4)}
    o = out[k]
    print(f"{k:44s} MFMA util {o['mfma_utilisation']:.3f}  clock {o['clock_ghz']:.2f} GHz  parked {o['waves_parked_share']:.2f} "
          f"stalled {o['waves_issue_stalled_share']:.2f} issuing {o['waves_issuing_share']:.2f}  LDS conflicts {o['lds_conflict_cycles_per_lds_cycle']:.3f}")
json.dump(out, open(sys.argv[2], "w"), indent=1)
