"""Shared by the CPU (oracle) and GPU (engine) tests of BertContextMLMSwitch: rebuild the padded batches of a
tests/golden/*_mlm.npz case (oracle/gen_golden.py::make_mlm_case)."""
import numpy as np
import torch


def batches(g):
    """yields (input_ids, attention_mask, token_type_ids, switch_labels, mlm_labels [B,S], slice of the
    labelled positions in the case's flat per-position arrays)."""
    from oracle import bert_oracle as O
    rows, bs = g["rows"], g["batch_size"]
    off = np.concatenate(([0], np.cumsum(g["lens"])))
    flat = g["mlm_labels"]
    done = 0
    for i in range(0, len(rows), bs):
        ids, mask = O.pad_batch(rows[i:i + bs])
        tt = O.token_type_ids_from_seps(ids, 102, g["n_ctx"])
        mlm = torch.full(ids.shape, -100, dtype=torch.long)
        for k in range(ids.shape[0]):
            lab = flat[off[i + k]:off[i + k + 1]]
            mlm[k, :len(lab)] = torch.from_numpy(lab.astype(np.int64))
        n = int((mlm != -100).sum())
        yield ids, mask, tt, torch.from_numpy(g["switch_labels"][i:i + bs]), mlm, slice(done, done + n)
        done += n
