"""Data side of task-targeted pre-training's evaluation pass: from the top-k file the
scoring path wrote to the batches `BertContextMLMSwitch.forward` consumes.  Restates
/root/reference/sentclf/pretrain_tasks.py:

  * `SentContextDataset` (:34-98): focus record -> (tokens, switch label, sent_ix, doc_id, contexts).  The
    reference finds the sentence row with a boolean mask over the whole table per item (:58) and the rows of a
    document with another (:69); here both are looked up once (clip_amd.focus_file.focus_row_indices, one
    group-by) — same row, same candidates, same `random` draws in the same order.
  * `switch_mlm_collate` (:136-279): context join, the trim loop (shared with ctx_collator:
    clip_amd.collate._fit_context, bert.py:99-147 == pretrain_tasks.py:156-200 except that nothing is
    lower-cased here — the dataset did that), token types, BERT-style MLM masking restricted to the focus
    sentence (:224-251), switch labels.  The masking draws from torch's global generator with the reference's
    own sequence of calls (bernoulli p, bernoulli 0.8, bernoulli 0.5, randint), so a seeded run produces the
    reference's batches bit for bit (tests/test_pretrain_data.py against goldens from the imported reference).
  * `select_inputs` (:293-306), `evaluate` (:308-331), and the other input form of the same loop: `PrecomputedDataset`
    (:19-32) + `simple_collate` (:100-134).

Reference quirks kept: `if eval:` (:277) tests the BUILTIN, so 'sentences' is always returned; rows that needed
back-filled separators are written back AFTER token types and masking were computed from the tokenizer's row
(:254-263), with `len(replace_toks)` (the list) where the row length was meant, and the attention-mask patch
is a no-op (`replace_masks` is never filled); a batch whose separator count does not reshape returns None
(:212-215).  nltk's word_tokenize (:50) is not installed: `str.split` stands in, as in clip_amd.datasets.
Not kept: doc_position, abc, the CNN variant (a name the reference imports but never defines, :16).
"""
import random as _random

import numpy as np
import torch

from .collate import _fit_context
from .datasets import SentDataset
from .focus_file import focus_row_indices


class SentContextDataset(SentDataset):
    """pretrain_tasks.py:34-98.  `fname`: notes CSV with columns doc_id, sent_ix, sentence (plain text);
    `focus_fname`: the top-k file (header doc_id,sent_ix[,max_label_score])."""

    def __init__(self, fname, focus_fname, task, n_context_sentences, switch_prob=0.0, same_doc=True,
                 doc_position=False, start_at=0, rng=None):
        super().__init__(fname, task)
        import pandas as pd
        if doc_position:
            raise NotImplementedError("doc_position is not on the scoring path")
        self.doc_sents = pd.read_csv(focus_fname)                                 # :37
        self.n_context_sentences = n_context_sentences
        self.switch_prob = switch_prob
        self.same_doc = same_doc
        self.start_at = start_at
        self.rng = rng or _random                   # the reference draws from the `random` module's global state
        if self.sent_ixs is None or "sent_ix" not in self.sents.columns:
            raise ValueError("the notes table needs a sent_ix column (pretrain_tasks.py:58)")
        self._rows = focus_row_indices(self.doc_ids, self.sent_ixs, self.doc_sents["doc_id"].to_numpy(),
                                       self.doc_sents["sent_ix"].to_numpy())
        self._doc_rows = None

    def tokenize_sentence(self, cell):                                            # :45-50
        if not isinstance(cell, str):
            return [""]
        return [x.lower() for x in cell.split()]        # reference: nltk.word_tokenize (absent here)

    def __len__(self):
        return len(self.doc_sents)

    def _rows_of_doc(self, doc_id):
        if self._doc_rows is None:                       # row positions of every document, ascending, once
            order = np.argsort(self.doc_ids, kind="stable")
            d = self.doc_ids[order]
            cut = np.flatnonzero(d[1:] != d[:-1]) + 1
            self._doc_rows = {int(k[0]): v for k, v in zip(np.split(d, cut), np.split(order, cut))}
        return self._doc_rows[int(doc_id)]

    def __getitem__(self, idx):
        idx = int(self._rows[idx + self.start_at])                                # :57-60
        if idx < 0:
            print(f"issue getting index: no row for focus record {idx}")
            return None                                                           # :61-63
        sent_ix, doc_id = self.sent_ixs[idx], self.doc_ids[idx]
        if self.rng.random() < self.switch_prob:                                  # :67-75
            if self.same_doc:
                idxs = set(self._rows_of_doc(doc_id).tolist())
                idxs.remove(idx)
                switch_ix = self.rng.choice(list(idxs))
            else:
                switch_ix = self.rng.choice(range(len(self.sents)))
            sent = self.tokenize_sentence(self._sentence[switch_ix])
            label = 1
        else:
            sent = self.tokenize_sentence(self._sentence[idx])
            label = 0
        n, N = self.n_context_sentences, len(self.sents)
        ctx_sents = []
        for offset in range(-n, n + 1):                                           # :82-91
            if offset == 0:
                continue
            if sent_ix + offset < 0:
                ctx_sents.append(["<DOC_START>"])
            elif idx + offset >= N or self.sent_ixs[idx + offset] != sent_ix + offset:   # idx + offset < 0 wraps, as .iloc
                ctx_sents.append(["<DOC_END>"])
            else:
                ctx_sents.append(self.tokenize_sentence(self._sentence[idx + offset]))
        contexts = tuple(" ".join(c) for c in ctx_sents)
        return sent, label, sent_ix, doc_id, contexts                             # :98


class PrecomputedDataset:
    """pretrain_tasks.py:19-32: instances tokenised ahead of time, one JSON object per line with the keys
    simple_collate reads (input_ids, token_type_ids, mlm_labels, sentences; id lists right-padded with 0)."""

    def __init__(self, fname):
        import json
        self.insts = []
        with open(fname) as f:
            for line in f:
                self.insts.append(json.loads(line.strip()))

    def __len__(self):
        return len(self.insts)

    def __getitem__(self, idx):
        return self.insts[idx]


def get_padded_length(lst, el):
    """pretrain_tasks.py:100-104: position of the first `el` (the pad id), else the full length"""
    try:
        return lst.index(el)
    except ValueError:
        return len(lst)


def simple_collate(batch):
    """pretrain_tasks.py:106-134: re-pad precomputed instances to the longest of the batch (token types padded with
    1, MLM labels with -100), then keep a random 25 % of the candidate MLM positions (one torch.bernoulli draw over
    the whole [B, S] matrix, as the reference does: a seeded run gives its batches bit for bit)."""
    batch_len = max(get_padded_length(inst["input_ids"], 0) for inst in batch)
    input_ids, sentences, attention_mask, token_type_ids, mlm_labels = [], [], [], [], []
    for inst in batch:
        n = get_padded_length(inst["input_ids"], 0)
        input_ids.append(inst["input_ids"][:n] + [0] * (batch_len - n))
        sentences.append(inst["sentences"])
        attention_mask.append([1] * n + [0] * (batch_len - n))
        token_type_ids.append(inst["token_type_ids"][:n] + [1] * (batch_len - n))
        mlm_labels.append(inst["mlm_labels"][:n] + [-100] * (batch_len - n))
    mlm_labels = torch.LongTensor(mlm_labels)
    new_mlm_labels = mlm_labels.clone()
    prob_mask = torch.zeros(new_mlm_labels.size())
    prob_mask.masked_fill_(new_mlm_labels != -100, 0.25)
    new_mask = torch.bernoulli(prob_mask).bool()
    new_mlm_labels[~new_mask] = -100
    return {"input_ids": torch.LongTensor(input_ids), "attention_mask": torch.LongTensor(attention_mask),
            "sentences": sentences, "token_type_ids": torch.LongTensor(token_type_ids), "mlm_labels": new_mlm_labels}


def special_ids(tokenizer):
    """ids `tokenizer.get_special_tokens_mask(..., already_has_special_tokens=True)` marks (:228-230)"""
    return [tokenizer.unk_token_id, tokenizer.sep_token_id, tokenizer.pad_token_id, tokenizer.cls_token_id,
            tokenizer.mask_token_id]


def switch_mlm_collate(batch, tokenizer, mlm_probability=0, doc_position=False, abc=False):
    """pretrain_tasks.py:136-279"""
    if doc_position or abc:
        raise NotImplementedError("doc_position / abc are not on the scoring path")
    sents, labels, replace = [], [], []
    n = 0
    for ix, inst in enumerate(batch):
        if inst is None:                                                          # :147-148
            continue
        toks, label, sent_ix, doc_id, context = inst
        n = len(context) // 2
        sent = " ".join(toks)
        if len(context) >= 2:                                                     # :163-170
            s, rep = _fit_context(sent, context, tokenizer)
        else:
            s, rep = sent, None
        if rep is not None:
            replace.append((ix, rep))
        sents.append(s)
        labels.append(label)
    if len(sents) == 0:
        return None                                                               # :204-205
    tokd = tokenizer(sents, padding=True, max_length=512, truncation=True)
    toks = torch.LongTensor(tokd["input_ids"])
    tt = torch.zeros(toks.shape).long()                                           # :211-221
    try:
        seps = torch.where(toks == tokenizer.sep_token_id)[1].reshape(-1, 2 * n + 1)
    except RuntimeError:
        return None
    for t, sep in zip(tt, seps):
        t[:sep[n - 1]] = 1
        t[sep[n]:] = 1
    mlm_labels = None
    if mlm_probability > 0:                                                       # :224-251
        mlm_labels = toks.clone()
        probability_matrix = torch.full(mlm_labels.shape, mlm_probability)
        special = torch.isin(mlm_labels, torch.tensor(special_ids(tokenizer)))
        probability_matrix.masked_fill_(special, value=0.0)
        probability_matrix.masked_fill_(mlm_labels.eq(tokenizer.pad_token_id), value=0.0)
        if n > 0:                                   # only the focus sentence is masked (token type 0)
            probability_matrix.masked_fill_(~tt.eq(0), value=0.0)
        masked_indices = torch.bernoulli(probability_matrix).bool()
        mlm_labels[~masked_indices] = -100
        indices_replaced = torch.bernoulli(torch.full(mlm_labels.shape, 0.8)).bool() & masked_indices
        toks[indices_replaced] = tokenizer.mask_token_id
        indices_random = torch.bernoulli(torch.full(mlm_labels.shape, 0.5)).bool() & masked_indices & ~indices_replaced
        random_words = torch.randint(len(tokenizer), mlm_labels.shape, dtype=torch.long)
        toks[indices_random] = random_words[indices_random]
    for ix, rep in replace:                                                       # :254-263
        if len(replace) > toks.size(1):             # (sic: the LIST of replacements, not the row)
            print("replace toks too long... ")
            return None
        row = rep + [tokenizer.pad_token_id] * (toks.size(1) - len(rep)) if len(replace) < toks.size(1) else rep
        toks[ix] = torch.LongTensor(row)
    mask = torch.LongTensor(tokd["attention_mask"])       # never patched (reference no-op, :264-266)
    output = {"input_ids": toks, "attention_mask": mask, "labels": torch.Tensor(labels).long(),
              "token_type_ids": tt, "doc_positions": []}
    if mlm_probability > 0:
        output["mlm_labels"] = mlm_labels
    output["sentences"] = sents                           # `if eval:` on the builtin (:277)
    return output


def select_inputs(batch, task, n_context_sentences, doc_position=False):
    """pretrain_tasks.py:293-306"""
    if doc_position:
        raise NotImplementedError("doc_position is not on the scoring path")
    if task == "mlm":
        key = "labels" if n_context_sentences == 0 else "mlm_labels"
        inputs = {"input_ids": batch["input_ids"], "attention_mask": batch["attention_mask"], key: batch["mlm_labels"]}
    else:
        inputs = {"input_ids": batch["input_ids"], "attention_mask": batch["attention_mask"],
                  "labels": batch["labels"], "mlm_labels": batch["mlm_labels"]}
    if "token_type_ids" in batch:
        inputs["token_type_ids"] = batch["token_type_ids"]
    return inputs


def evaluate(model, loader, task, device, n_context_sentences, doc_position=False, max_batches=1000):
    """pretrain_tasks.py:308-331: the summed loss of the first 1000 batches (None batches and batches the model
    refuses are skipped, as there)."""
    total_loss = 0.0
    model.eval()
    for ix, x in enumerate(loader):
        if ix >= max_batches:
            break
        if x is None:
            continue
        x = {k: (v.to(device) if isinstance(v, torch.Tensor) else v) for k, v in x.items()}
        inputs = select_inputs(x, task, n_context_sentences, doc_position)
        try:
            output = model(**inputs)
        except Exception as exp:
            print(f"error putting example through model: {exp}")
            continue
        if output is None:
            continue
        loss, _ = output
        total_loss += loss.item()
    return total_loss
