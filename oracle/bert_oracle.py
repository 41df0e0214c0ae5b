"""CPU ORACLE — TEST INFRASTRUCTURE ONLY.

A plain PyTorch-fp32 CPU restatement of the reference's bulk BERT(+Context) scoring path.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
the product (clip_amd/) never does and fails loudly when its HIP library is missing.

Pinning: the reference holds no tests/golden vectors for this path, so the oracle
is pinned by outputs of the reference ITSELF, imported in the build container by
oracle/gen_golden.py (fixtures under tests/golden/, checked by tests/test_oracle_golden.py).

Each function cites the reference (or the third-party transformers 5.15.0 file the reference
delegates to) that it restates.  `tf:` = transformers/models/bert/modeling_bert.py,
`ref:` = /root/reference/.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- encoder pieces
def layer_norm(x, w, b, eps):
    """nn.LayerNorm over the last dim, biased variance (tf:62, tf:292, tf:350)."""
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def gelu_erf(x):
    """hidden_act='gelu' = exact erf form (tf:334-337 via ACT2FN['gelu'])."""
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def embeddings(sd, cfg, input_ids, token_type_ids):
    """word + token_type + position(0..S) -> LayerNorm; dropout is identity in eval (tf:68-108)."""
    B, S = input_ids.shape
    if token_type_ids is None:
        token_type_ids = torch.zeros_like(input_ids)
    pos = torch.arange(S).unsqueeze(0).expand(B, S)
    e = (sd["bert.embeddings.word_embeddings.weight"][input_ids]
         + sd["bert.embeddings.token_type_embeddings.weight"][token_type_ids]
         + sd["bert.embeddings.position_embeddings.weight"][pos])
    return layer_norm(e, sd["bert.embeddings.LayerNorm.weight"],
                      sd["bert.embeddings.LayerNorm.bias"], cfg.layer_norm_eps)


def self_attention(sd, cfg, p, x, add_mask):
    """Q,K,V = Linear(x); per head softmax(QK^T/sqrt(d) + mask)·V; heads concatenated
    (tf:164-203, math of tf:111-136)."""
    B, S, H = x.shape
    nh, d = cfg.num_attention_heads, cfg.hidden_size // cfg.num_attention_heads

    def proj(n):
        y = x @ sd[p + f"attention.self.{n}.weight"].T + sd[p + f"attention.self.{n}.bias"]
        return y.view(B, S, nh, d).transpose(1, 2)

    q, k, v = proj("query"), proj("key"), proj("value")
    s = (q @ k.transpose(2, 3)) * (d ** -0.5)
    if add_mask is not None:
        s = s + add_mask
    a = torch.softmax(s, dim=-1)
    return (a @ v).transpose(1, 2).reshape(B, S, H)


def encoder_layer(sd, cfg, i, x, add_mask):
    """BertLayer: attention -> SelfOutput(dense + residual + LN) -> Intermediate(dense+GELU)
    -> Output(dense + residual + LN) (tf:374-416, 289-293, 334-337, 347-351)."""
    p = f"bert.encoder.layer.{i}."
    eps = cfg.layer_norm_eps
    a = self_attention(sd, cfg, p, x, add_mask)
    a = a @ sd[p + "attention.output.dense.weight"].T + sd[p + "attention.output.dense.bias"]
    x1 = layer_norm(a + x, sd[p + "attention.output.LayerNorm.weight"],
                    sd[p + "attention.output.LayerNorm.bias"], eps)
    h = gelu_erf(x1 @ sd[p + "intermediate.dense.weight"].T + sd[p + "intermediate.dense.bias"])
    o = h @ sd[p + "output.dense.weight"].T + sd[p + "output.dense.bias"]
    return layer_norm(o + x1, sd[p + "output.LayerNorm.weight"],
                      sd[p + "output.LayerNorm.bias"], eps)


def bert_model(sd, cfg, input_ids, attention_mask=None, token_type_ids=None, want_pooler=False):
    """BertModel.forward: embeddings -> key-padding additive mask -> L layers (-> pooler)
    (tf:623-718; mask tf:688-708; pooler tanh(W h[:,0] + b) tf:457-463)."""
    x = embeddings(sd, cfg, input_ids, token_type_ids)
    add_mask = None
    if attention_mask is not None:
        neg = torch.finfo(torch.float32).min
        add_mask = torch.zeros(attention_mask.shape, dtype=torch.float32)
        add_mask = add_mask.masked_fill(attention_mask == 0, neg)[:, None, None, :]
    for i in range(cfg.num_hidden_layers):
        x = encoder_layer(sd, cfg, i, x, add_mask)
    pooled = None
    if want_pooler:
        pooled = torch.tanh(x[:, 0] @ sd["bert.pooler.dense.weight"].T + sd["bert.pooler.dense.bias"])
    return x, pooled


# --------------------------------------------------------------------------- classifier heads
def collect_focus_repr(last_hidden, input_ids, sep_token_id, n_context_sentences):
    """ref: sentclf/bert_model.py:49-64.  Column index of every [SEP] in row-major order,
    reshaped to [-1, 2n+1], column n = the focus sentence's closing [SEP]; gather that row."""
    cols = torch.where(input_ids == sep_token_id)[1]
    end = cols.reshape(-1, 2 * n_context_sentences + 1)[:, n_context_sentences]
    idx = end[:, None, None].repeat(1, 1, last_hidden.shape[2])
    return torch.gather(last_hidden, 1, idx).squeeze(1)


def context_forward(sd, cfg, input_ids, attention_mask=None, token_type_ids=None, labels=None,
                    sep_token_id=102, n_context_sentences=2):
    """BertSequenceMultilabelClassificationContext.forward, task='multilabel'
    (ref: sentclf/bert_model.py:66-103). Returns (loss, logits) if labels given else (logits,)."""
    x, _ = bert_model(sd, cfg, input_ids, attention_mask, token_type_ids)
    pooled = collect_focus_repr(x, input_ids, sep_token_id, n_context_sentences)
    logits = pooled @ sd["classifier.weight"].T + sd["classifier.bias"]
    if labels is None:
        return (logits,)
    return F.binary_cross_entropy_with_logits(logits, labels), logits


def sentence_forward(sd, cfg, input_ids, attention_mask=None, token_type_ids=None, labels=None):
    """BertForSequenceMultilabelClassification.forward (ref: sentclf/bert_model.py:210-251):
    pooler output -> dropout(identity) -> classifier."""
    _, pooled = bert_model(sd, cfg, input_ids, attention_mask, token_type_ids, want_pooler=True)
    logits = pooled @ sd["classifier.weight"].T + sd["classifier.bias"]
    if labels is None:
        return (logits,)
    return F.binary_cross_entropy_with_logits(logits, labels), logits


def mlm_switch_forward(sd, cfg, input_ids, attention_mask=None, token_type_ids=None, labels=None,
                       mlm_labels=None, sep_token_id=102, n_context_sentences=2):
    """BertContextMLMSwitch.forward (ref: sentclf/bert_model.py:137-186).  mlm_scores =
    BertLMPredictionHead(last_hidden): dense + gelu + LayerNorm, decoder H -> vocab + bias (tf
    BertPredictionHeadTransform / BertLMPredictionHead); switch logits = classifier(focus [SEP] row);
    total_loss = CrossEntropy(mlm_scores, mlm_labels) (ignore_index -100) + CrossEntropy(logits, labels).
    Returns (total_loss, logits, mlm_scores) — the reference returns the first two."""
    x, _ = bert_model(sd, cfg, input_ids, attention_mask, token_type_ids)
    pooled = collect_focus_repr(x, input_ids, sep_token_id, n_context_sentences)
    t = x @ sd["mlm_predictions.transform.dense.weight"].T + sd["mlm_predictions.transform.dense.bias"]
    t = layer_norm(gelu_erf(t), sd["mlm_predictions.transform.LayerNorm.weight"],
                   sd["mlm_predictions.transform.LayerNorm.bias"], cfg.layer_norm_eps)
    dec_bias = sd.get("mlm_predictions.decoder.bias", sd.get("mlm_predictions.bias"))
    mlm_scores = t @ sd["mlm_predictions.decoder.weight"].T + dec_bias
    logits = pooled @ sd["classifier.weight"].T + sd["classifier.bias"] if labels is not None else None
    total = None
    if mlm_labels is not None:
        total = F.cross_entropy(mlm_scores.view(-1, mlm_scores.shape[-1]), mlm_labels.view(-1))
        if labels is not None:
            total = total + F.cross_entropy(logits.view(-1, 2), labels.view(-1))
    return total, logits, mlm_scores


# --------------------------------------------------------------------------- scoring epilogue
def score_epilogue(logits, any_threshold=0.05):
    """ref: sentclf/apply_bert_unlabeled.py:111-124. p = sigmoid(logits) in fp32;
    keep = any_j(p_j > threshold): a float32 array against a Python float compares in float32
    (the scalar is weak-typed in numpy 1.x value-based casting and in numpy 2 NEP 50 alike), i.e.
    against float32(threshold); score = max_j p_j (fp32)."""
    p = torch.sigmoid(logits.float()).numpy()
    keep = np.any(p > np.float32(any_threshold), axis=1)
    return p, keep, p.max(axis=1)


def label_decisions(probs, label_threshs):
    """ref: test_load.py:37-38. float32 sigmoid > float64[7] thresholds, strict '>' in fp64."""
    return np.asarray(probs, dtype=np.float32).astype(np.float64) > np.asarray(label_threshs, np.float64)


def topk_records(doc_ids, sent_ixs, scores, k):
    """ref: README.md:43 ("sorted by decreasing max_label_score ... take the top k"); there is no
    code in the reference. Build-defined total order: (score desc, doc_id asc, sent_ix asc)."""
    doc_ids = np.asarray(doc_ids, np.int64)
    sent_ixs = np.asarray(sent_ixs, np.int64)
    scores = np.asarray(scores, np.float32)
    order = np.lexsort((sent_ixs, doc_ids, -scores.astype(np.float64)))
    order = order[:k]
    return doc_ids[order], sent_ixs[order], scores[order]


# --------------------------------------------------------------------------- the scoring loop
def pad_batch(rows, pad_id=0):
    """Pad-to-longest of one DataLoader batch, as the HF tokenizer's padding=True does
    (ref: sentclf/bert.py:57,152): ids right-padded with [PAD], mask 1 on real tokens."""
    S = max(len(r) for r in rows)
    ids = torch.full((len(rows), S), pad_id, dtype=torch.long)
    mask = torch.zeros((len(rows), S), dtype=torch.long)
    for i, r in enumerate(rows):
        ids[i, :len(r)] = torch.as_tensor(r, dtype=torch.long)
        mask[i, :len(r)] = 1
    return ids, mask


def token_type_ids_from_seps(ids, sep_token_id, n_ctx):
    """ref: sentclf/bert.py:168-176. type 1 on [0, sep[n-1]) and on [sep[n], end) — padding
    included; type 0 on the [SEP] before the focus and on the focus tokens."""
    tt = torch.zeros_like(ids)
    seps = torch.where(ids == sep_token_id)[1].reshape(-1, 2 * n_ctx + 1)
    for t, sep in zip(tt, seps):
        t[:sep[n_ctx - 1]] = 1
        t[sep[n_ctx]:] = 1
    return tt


def score_rows(sd, cfg, rows, n_ctx, sep_token_id=102, batch_size=8, any_threshold=0.05):
    """The hot loop of ref: sentclf/apply_bert_unlabeled.py:92-124 over pre-tokenised rows
    (lists of token ids), in input order, DataLoader(batch_size=8, shuffle=False) semantics.
    Returns (probs[N,7], keep[N], score[N], logits[N,7])."""
    P, K, Sc, L = [], [], [], []
    with torch.no_grad():
        for i in range(0, len(rows), batch_size):
            ids, mask = pad_batch(rows[i:i + batch_size], cfg.pad_token_id)
            labels = torch.zeros((ids.shape[0], cfg.num_labels))
            if n_ctx > 0:
                tt = token_type_ids_from_seps(ids, sep_token_id, n_ctx)
                _, logits = context_forward(sd, cfg, ids, mask, tt, labels, sep_token_id, n_ctx)
            else:
                _, logits = sentence_forward(sd, cfg, ids, mask, None, labels)
            p, keep, score = score_epilogue(logits, any_threshold)
            P.append(p); K.append(keep); Sc.append(score); L.append(logits.numpy())
    return np.concatenate(P), np.concatenate(K), np.concatenate(Sc), np.concatenate(L)
