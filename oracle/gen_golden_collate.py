"""Golden vectors for the host front end, produced by the REFERENCE's own datasets and collators
(sentclf/neural_baselines.py SentDataset/SentEvalDataset, sentclf/bert.py collator/ctx_collator)
over HF's BertTokenizer on tests/golden/vocab.txt.  Run in the build container only:
    python oracle/make_vocab.py && python oracle/gen_golden_collate.py
Writes tests/golden/notes_synth.csv (synthetic input rows), tests/golden/dummy_test_data.csv (the
reference's 7-row data fixture, data only) and tests/golden/collate_*.json (expected outputs)."""
import json
import os
import shutil
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.gen_golden import install_shims, REF  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")


def synth_csv(path):
    rng = np.random.default_rng(77)
    vocab_words = [f"w{i:03d}" for i in range(400)] + "patient should follow up with cardiology in 2 weeks".split()
    rows = []

    def sent(n, extra=()):
        w = [vocab_words[int(i)] for i in rng.integers(0, len(vocab_words), n)]
        for pos, tok in extra:
            w.insert(pos, tok)
        return w

    # doc 1: ordinary short sentences with mixed case, punctuation, accents, OOV, digits
    d = 9001
    specials = [["Pt", "should", "F/U", "w/", "Dr.", "Müller", "in", "2", "wks", "."],
                ["Vancomycin", "1g", "IV", "q12h", "x", "14", "days", ";", "check", "LABS", "(", "CBC", ")"],
                ["zzzunknownzzz", "x" * 120, "naïve", "café", "MRI", "pending"],
                ["no", "heavy", "lifting", "[sep]", "literal", "brackets"]]
    for i, s in enumerate(specials + [sent(int(n)) for n in rng.integers(3, 30, 8)]):
        rows.append((d, i, s, ["I-Appointment-related followup"] if i % 3 == 0 else []))
    # doc 2: long context sentences so that focus + 4 contexts >= 512 tokens (drop-outer branch)
    d = 9002
    for i, n in enumerate([150, 140, 30, 160, 170, 20, 200, 210, 25, 15]):
        rows.append((d, i, sent(n), []))
    # doc 3: a focus sentence that alone exceeds 512 tokens (split-in-half branch), and one > 1100
    d = 9003
    for i, n in enumerate([10, 600, 12, 1200, 8]):
        rows.append((d, i, sent(n), ["I-Lab-related followup"] if i == 1 else []))
    # doc 4: gap in sentence indices (-> <DOC_END> inside a document) and a duplicated index
    d = 9004
    for i in [0, 1, 3, 4, 4, 5]:
        rows.append((d, i, sent(6), []))
    import csv
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["doc_id", "sent_index", "sentence", "labels"])
        for d, i, s, l in rows:
            w.writerow([d, i, str(s), str(l)])
    return len(rows)


def batches(n, sizes=(1, 3, 4, 2)):
    out, i, k = [], 0, 0
    while i < n:
        b = sizes[k % len(sizes)]
        out.append(list(range(i, min(i + b, n))))
        i += b
        k += 1
    return out


def tolist(v):
    import torch
    if isinstance(v, torch.Tensor):
        return v.tolist()
    if isinstance(v, (list, tuple)):
        return [tolist(x) for x in v]
    if isinstance(v, np.generic):
        return v.item()
    return v


def main():
    install_shims()
    import neural_baselines as NB
    import bert as RB
    from transformers import BertTokenizer
    shutil.copyfile(os.path.join(REF, "dummy_test_data.csv"), os.path.join(G, "dummy_test_data.csv"))
    n_rows = synth_csv(os.path.join(G, "notes_synth.csv"))
    tok = BertTokenizer(os.path.join(G, "vocab.txt"), do_lower_case=True)
    print("rows", n_rows, "sep", tok.sep_token_id)

    out = {}
    for csv_name in ("dummy_test_data.csv", "notes_synth.csv"):
        path = os.path.join(G, csv_name)
        for n in (2, 1):
            ds = NB.SentEvalDataset(path, "multilabel", n_context_sentences=n)
            items = [ds[i] for i in range(len(ds))]
            res = {"dataset": [tolist(list(it)) for it in items], "batches": []}
            for b in batches(len(ds)):
                r = RB.ctx_collator([items[i] for i in b], "multilabel", tok, eval=True)
                res["batches"].append({k: tolist(v) for k, v in r.items()})
            out[f"{csv_name}|ctx{n}"] = res
        ds = NB.SentDataset(path, "multilabel")
        items = [ds[i] for i in range(len(ds))]
        res = {"dataset": [tolist(list(it)) for it in items], "batches": []}
        for b in batches(len(ds)):
            r = RB.collator([items[i] for i in b], "multilabel", tok, eval=True)
            res["batches"].append({k: tolist(v) for k, v in r.items()})
        out[f"{csv_name}|sent"] = res
    import gzip
    with gzip.open(os.path.join(G, "collate_golden.json.gz"), "wt") as f:
        json.dump(out, f)
    print("wrote", os.path.getsize(os.path.join(G, "collate_golden.json.gz")) // 1024, "KB")


if __name__ == "__main__":
    main()
