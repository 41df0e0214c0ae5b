"""Collators of the scoring path, restating /root/reference/sentclf/bert.py: multilabel_labeler
(:35-42), collator (:44-68), ctx_collator (:70-188), select_inputs (:191-195) — same dict keys, same
tensors — plus `build_rows`, the columnar form the bulk scorer eats: the very token ids ctx_collator
produces per row, without padding and with each sentence tokenised once.

Reference quirks kept: the trim loop measures the TRUNCATED tokenisation (`len(test) >= 512`);
contexts are dropped outermost-first, alternating, starting with the longer of the two outer ones
(by characters of context[0] vs context[1], sic); dropped contexts leave an empty segment so the
row keeps 2n+1 [SEP]; the attention-mask patch for back-filled rows is a no-op (`replace_masks` is
never filled, bert.py:156-165).  Not kept: doc_position, abc (not on the scoring path).
"""
import numpy as np
import torch

from .constants import LABEL_TYPES


def multilabel_labeler(label):
    """list of label names -> multi-hot float array (bert.py:35-42)"""
    if isinstance(label, str):
        import ast
        label = ast.literal_eval(label)
    out = np.zeros(len(LABEL_TYPES))
    out[[LABEL_TYPES.index(l) for l in label]] = 1
    return out


def collator(batch, task, tokenizer, eval=False, doc_position=False):
    """bert.py:44-68"""
    sents, labels, doc_ids, sent_ixs = [], [], [], []
    for sent, label, doc_id, sent_ix in batch:
        sents.append(" ".join(sent))
        labels.append(multilabel_labeler(label) if task == "multilabel" else int(label))
        doc_ids.append(doc_id)
        sent_ixs.append(sent_ix)
    tokd = tokenizer(sents, padding=True, max_length=512, truncation=True)
    out = {"input_ids": torch.LongTensor(tokd["input_ids"]),
           "attention_mask": torch.LongTensor(tokd["attention_mask"]),
           "labels": torch.Tensor(np.asarray(labels)), "doc_ids": doc_ids, "sent_ixs": sent_ixs}
    if task == "binary":
        out["labels"] = out["labels"].long()
    if eval:
        out["sentences"] = sents
    return out


def _fit_context(sent, context, tokenizer):
    """The per-row body of ctx_collator (bert.py:99-147): returns (sent_w_context string,
    replacement token list or None)."""
    n = len(context) // 2
    sep_id = tokenizer.sep_token_id
    before, after = context[:n], context[n:]
    s = " [SEP] ".join(before + (sent,) + after)
    test = tokenizer(s, padding=True, max_length=512, truncation=True)["input_ids"]
    first_longer = len(context[0]) > len(context[1])
    trim_start = first_longer
    start_ix = 1 if first_longer else 0
    end_ix = 0 if first_longer else 1
    replace = None
    while len(test) >= 512:
        before, after = context[start_ix:n], context[n:len(context) - end_ix]
        s = " [SEP] ".join(("",) * start_ix + before + (sent,) + after + ("",) * end_ix)
        test = tokenizer(s, padding=True, max_length=512, truncation=True)["input_ids"]
        if len(test) >= 512 and (end_ix > n or start_ix > n):
            # the focus sentence alone is too long: keep its second half (bert.py:120-126)
            toks = s.split()
            s = " ".join(toks[:n] + toks[len(toks) // 2:-1])
            test = tokenizer(s, padding=True, max_length=512, truncation=True)["input_ids"]
            if len(test) >= 512:
                for i in range(n + 1):
                    test[len(test) - 1 - i] = sep_id
                replace = list(test)
            if sum(1 for t in test if t == sep_id) < 2 * n + 1:
                for i in range(n + 1):
                    test[len(test) - 1 - i] = sep_id
                replace = list(test)
            break
        trim_start = not trim_start
        if trim_start:
            start_ix += 1
        else:
            end_ix += 1
    return s, replace


def ctx_collator(batch, task, tokenizer, eval=False, doc_position=False, abc=False):
    """bert.py:70-188"""
    if doc_position or abc:
        raise NotImplementedError("doc_position / abc are not on the scoring path")
    sents, labels, doc_ids, sent_ixs, replace = [], [], [], [], {}
    n = 0
    for ix, (toks, sent_ix, annotator, label, doc_id, context) in enumerate(batch):
        doc_ids.append(doc_id)
        sent_ixs.append(sent_ix)
        n = len(context) // 2
        sent = " ".join(toks).lower()
        context = tuple(c.lower() for c in context)
        s, rep = _fit_context(sent, context, tokenizer)
        if rep is not None:
            replace[ix] = rep
        sents.append(s)
        labels.append(multilabel_labeler(label) if task == "multilabel" else int(label))
    tokd = tokenizer(sents, padding=True, max_length=512, truncation=True)
    toks = torch.LongTensor(tokd["input_ids"])
    for ix, rep in replace.items():
        row = rep + [tokenizer.pad_token_id] * (toks.size(1) - len(rep))
        toks[ix] = torch.LongTensor(row)
    mask = torch.LongTensor(tokd["attention_mask"])       # never patched (reference no-op)
    tt = torch.zeros(toks.shape).long()
    seps = torch.where(toks == tokenizer.sep_token_id)[1].reshape(-1, 2 * n + 1)
    for t, sep in zip(tt, seps):
        t[:sep[n - 1]] = 1
        t[sep[n]:] = 1
    lab = torch.Tensor(np.asarray(labels))
    if task == "binary":
        lab = lab.long()
    out = {"input_ids": toks, "attention_mask": mask, "labels": lab, "token_type_ids": tt,
           "doc_positions": [], "doc_ids": doc_ids, "sent_ixs": sent_ixs}
    if eval:
        out["sentences"] = sents
    return out


def select_inputs(x):
    """bert.py:191-195"""
    inputs = {"input_ids": x["input_ids"], "attention_mask": x["attention_mask"], "labels": x["labels"]}
    if "token_type_ids" in x:
        inputs["token_type_ids"] = x["token_type_ids"]
    return inputs


# --------------------------------------------------------------------------- columnar fast path
def build_rows(dataset, tokenizer, n_context_sentences):
    """All rows of `dataset` as the token-id arrays the collators would produce (no padding):
    sentence-only -> [CLS] w.. [SEP] truncated to 512; context -> ctx_collator's row.
    Every sentence is tokenised ONCE (the reference tokenises each >= 2(2n+1) times); rows whose
    untruncated length reaches 512 go through the reference's trim procedure verbatim."""
    from .scoring import PackedRows
    tok = tokenizer
    N = len(dataset)
    pieces = [None] * N

    def sent_pieces(i, lower_join):
        if pieces[i] is None:
            pieces[i] = tok.encode_plain(lower_join(i))
        return pieces[i]

    rows = []
    if n_context_sentences == 0:
        for i in range(N):
            sent = dataset[i][0]                      # already lower-cased (neural_baselines.py:153)
            rows.append(np.asarray(tok.encode(" ".join(sent)), np.int32))
        return PackedRows.from_rows(rows), dataset.doc_ids.astype(np.int64), \
            np.asarray(dataset.sent_ixs, np.int64)

    n = n_context_sentences
    ix = dataset.sent_ixs
    start_p = tok.encode_plain("<DOC_START>".lower())
    end_p = tok.encode_plain("<DOC_END>".lower())
    from .datasets import _parse_list
    text = lambda i: " ".join(str(x) for x in _parse_list(dataset._sentence[i], lower=False)).lower()
    for i in range(N):
        segs = []
        for offset in range(-n, n + 1):
            if offset == 0:
                segs.append(sent_pieces(i, text))
                continue
            want, j = ix[i] + offset, i + offset
            if want < 0:
                segs.append(start_p)
            elif j >= N or ix[j] != want:
                segs.append(end_p)
            else:
                segs.append(sent_pieces(j, text))
        total = 1 + sum(len(s) + 1 for s in segs)
        if total < 512:
            row = [tok.cls_token_id]
            for s in segs:
                row.extend(s)
                row.append(tok.sep_token_id)
            rows.append(np.asarray(row, np.int32))
        else:       # rare: the reference's trim loop, on strings, verbatim
            toks_, sent_ix, annot, label, doc_id, context = dataset[i]
            s, rep = _fit_context(" ".join(toks_).lower(), tuple(c.lower() for c in context), tok)
            rows.append(np.asarray(rep if rep is not None else tok.encode(s), np.int32))
    return PackedRows.from_rows(rows), dataset.doc_ids.astype(np.int64), np.asarray(ix, np.int64)
