"""Golden vectors for the data side of task-targeted pre-training, produced by the REFERENCE's own
`SentContextDataset` and `switch_mlm_collate` (sentclf/pretrain_tasks.py:34-98, 136-279) over HF's BertTokenizer
on tests/golden/vocab.txt, under seeded `random` / torch generators.  Run in the build container only:
    python oracle/gen_golden_pretrain.py
Writes tests/golden/pretrain_notes.csv (synthetic notes table: doc_id, sent_ix, sentence as plain text),
tests/golden/pretrain_focus.csv (a top-k file in clip_amd.job.write_topk_csv's format, incl. one record with no
row) and tests/golden/pretrain_golden.json.gz (expected items and batches).

Harness-side shims, none touching reference files: those of oracle/gen_golden.py (nltk / gensim stubs:
word_tokenize = str.split) plus ONE more — sentclf/pretrain_tasks.py:16 imports `BertContextCNNMLMSwitch` from
bert_model, a name the reference never defines, so the module cannot import as shipped; the
harness sets that attribute to None on the imported bert_model module first — and `tokenizer._pad_token`
(:232), a private attribute transformers 5.x no longer has, is set on the tokenizer instance."""
import csv
import gzip
import json
import os
import random
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.gen_golden import install_shims  # noqa: E402
from oracle.gen_golden_collate import tolist  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")
CASES = [   # (n_ctx, switch_prob, same_doc, mlm_probability, seed, batch sizes)
    (2, 0.5, True, 0.15, 11, (4, 3, 5)),
    (1, 0.3, False, 0.4, 12, (6, 2)),
    (2, 0.0, True, 0.0, 13, (5,)),
]


def get_len(ids):
    return ids.index(0) if 0 in ids else len(ids)


def synth_tables():
    rng = np.random.default_rng(99)
    words = [f"w{i:03d}" for i in range(400)] + "patient should follow up with cardiology in 2 weeks".split()

    def sent(n):
        return " ".join(words[int(i)] for i in rng.integers(0, len(words), n))

    rows = []
    for i, s in enumerate(["Pt should F/U w/ Dr. Müller in 2 wks .", "Vancomycin 1g IV q12h x 14 days ; check LABS ( CBC )",
                           "no heavy lifting"] + [sent(int(n)) for n in rng.integers(3, 30, 9)]):
        rows.append((7001, i, s))
    for i, n in enumerate([150, 140, 30, 160, 170, 20, 200, 210, 25, 15]):     # contexts dropped to fit 512
        rows.append((7002, i, sent(n)))
    for i, n in enumerate([10, 600, 12, 8]):                                   # a focus sentence over 512 tokens
        rows.append((7003, i, sent(n)))
    for i in [0, 1, 3, 4, 4, 5]:                                               # a gap and a duplicated index
        rows.append((7004, i, sent(6)))
    with open(os.path.join(G, "pretrain_notes.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["doc_id", "sent_ix", "sentence"])
        w.writerows(rows)
    focus = [(7001, 0), (7001, 5), (7002, 2), (7002, 5), (7001, 11), (7004, 4), (7002, 8), (7009, 1),   # 7009: no such row
             (7004, 1), (7001, 2), (7002, 0), (7004, 5), (7003, 0), (7001, 7)]
    with open(os.path.join(G, "pretrain_focus.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["doc_id", "sent_ix", "max_label_score"])
        for k, (d, s) in enumerate(focus):
            w.writerow([d, s, np.float32(0.99 - 0.01 * k)])
    # a second focus file whose records include the over-long focus sentence (the separator back-fill branch)
    with open(os.path.join(G, "pretrain_focus_long.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["doc_id", "sent_ix", "max_label_score"])
        for k, (d, s) in enumerate([(7003, 1), (7002, 6), (7003, 2)]):
            w.writerow([d, s, np.float32(0.5 - 0.01 * k)])


def main():
    install_shims()
    import bert_model as RBM
    RBM.BertContextCNNMLMSwitch = None          # the name pretrain_tasks.py:16 imports and the reference never defines
    import pretrain_tasks as PT
    from transformers import BertTokenizer
    synth_tables()
    tok = BertTokenizer(os.path.join(G, "vocab.txt"), do_lower_case=True)
    # transformers 5.x API drift (an ordinary AttributeError, as gen_golden.py's post_init shim): the private
    # attribute pretrain_tasks.py:232 reads is gone; the harness puts the public value where the reference looks
    tok.__dict__["_pad_token"] = tok.pad_token
    notes = os.path.join(G, "pretrain_notes.csv")
    out = {"cases": []}
    for focus_name in ("pretrain_focus.csv", "pretrain_focus_long.csv"):
        for n_ctx, switch_prob, same_doc, mlm_p, seed, sizes in CASES:
            ds = PT.SentContextDataset(notes, os.path.join(G, focus_name), "switch", n_ctx, switch_prob=switch_prob,
                                       same_doc=same_doc)
            random.seed(seed)
            items = [ds[i] for i in range(len(ds))]
            torch.manual_seed(seed)
            batches, i, k = [], 0, 0
            while i < len(items):
                b = sizes[k % len(sizes)]
                r = PT.switch_mlm_collate(items[i:i + b], tok, mlm_probability=mlm_p)
                batches.append(None if r is None else {kk: tolist(v) for kk, v in r.items()})
                i += b
                k += 1
            out["cases"].append({"focus": focus_name, "n_ctx": n_ctx, "switch_prob": switch_prob, "same_doc": same_doc,
                                 "mlm_probability": mlm_p, "seed": seed, "sizes": list(sizes),
                                 "items": [None if it is None else tolist(list(it)) for it in items],
                                 "batches": batches})
            print(focus_name, n_ctx, "items", len(items), "None batches", sum(b is None for b in batches))
    # PrecomputedDataset + simple_collate (pretrain_tasks.py:19-32, 100-134) over a small JSONL of instances
    rng = np.random.default_rng(5)
    insts = []
    for k in range(9):
        n, pad = int(rng.integers(6, 40)), int(rng.integers(0, 12))
        ids = [101] + [int(x) for x in rng.integers(1000, 3000, n)] + [102]
        cand = [int(t) if rng.random() < 0.5 else -100 for t in ids]
        insts.append({"input_ids": ids + [0] * pad, "token_type_ids": [0] * len(ids) + [0] * pad,
                      "mlm_labels": cand + [-100] * pad, "sentences": f"instance {k}"})
    insts[3]["input_ids"] = insts[3]["input_ids"][:get_len(insts[3]["input_ids"])]      # one without any padding
    for key in ("token_type_ids", "mlm_labels"):
        insts[3][key] = insts[3][key][:len(insts[3]["input_ids"])]
    with open(os.path.join(G, "pretrain_insts.jsonl"), "w") as f:
        for inst in insts:
            f.write(json.dumps(inst) + "\n")
    ds = PT.PrecomputedDataset(os.path.join(G, "pretrain_insts.jsonl"))
    torch.manual_seed(21)
    out["precomputed"] = {"seed": 21, "sizes": [4, 3, 2], "batches": []}
    i = 0
    for b in out["precomputed"]["sizes"]:
        r = PT.simple_collate([ds[j] for j in range(i, i + b)])
        out["precomputed"]["batches"].append({kk: tolist(v) for kk, v in r.items()})
        i += b
    with gzip.open(os.path.join(G, "pretrain_golden.json.gz"), "wt") as f:
        json.dump(out, f)
    print("wrote", os.path.getsize(os.path.join(G, "pretrain_golden.json.gz")) // 1024, "KB")


if __name__ == "__main__":
    main()
