"""Generate tests/golden/*.npz by running the REFERENCE ITSELF in the build container.

Run here only (needs /root/reference, never exists on the GPU box):
    python oracle/gen_golden.py            # writes tests/golden/

What travels is data only: seeds, token ids, and the reference's fp32 outputs.  The reference's
source is imported from where it lies; nothing of it is copied.

Two harness-side shims, neither touches reference files:
  1. transformers 5.x wants post_init() before init_weights(); the reference's __init__
     (sentclf/bert_model.py:21,202) calls init_weights() directly.
  2. nltk / gensim are not installed; sentclf/neural_baselines.py:14,16 imports them at module
     top.  Stub modules are registered (word_tokenize = str.split is only the reference's
     fallback when eval(row.sentence) fails, never hit for list-literal CSVs).
Hub assets (model names, tokenizer vocab) are unreachable offline: weights are the procedural
generator of clip_amd/weights.py loaded through load_state_dict().
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")


def install_shims():
    from transformers import PreTrainedModel
    orig = PreTrainedModel.init_weights

    def init_weights(self):
        if "all_tied_weights_keys" not in self.__dict__:
            return self.post_init()
        return orig(self)

    PreTrainedModel.init_weights = init_weights
    for name in ("nltk", "nltk.tokenize", "gensim", "gensim.models"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["nltk"].word_tokenize = str.split
    sys.modules["nltk.tokenize"].word_tokenize = str.split
    sys.modules["gensim.models"].KeyedVectors = object
    sys.modules["gensim"].models = sys.modules["gensim.models"]
    sys.path.insert(0, os.path.join(REF, "sentclf"))


def hf_config(cfg):
    from transformers import BertConfig as HFConfig
    return HFConfig(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size,
                    num_hidden_layers=cfg.num_hidden_layers,
                    num_attention_heads=cfg.num_attention_heads,
                    intermediate_size=cfg.intermediate_size,
                    max_position_embeddings=cfg.max_position_embeddings,
                    type_vocab_size=cfg.type_vocab_size, layer_norm_eps=cfg.layer_norm_eps,
                    num_labels=cfg.num_labels)


def pad(rows, pad_id=0):
    S = max(len(r) for r in rows)
    ids = torch.full((len(rows), S), pad_id, dtype=torch.long)
    mask = torch.zeros((len(rows), S), dtype=torch.long)
    for i, r in enumerate(rows):
        ids[i, :len(r)] = torch.as_tensor(np.asarray(r), dtype=torch.long)
        mask[i, :len(r)] = 1
    return ids, mask


def ref_token_types(ids, sep_id, n):
    """token types exactly as the reference builds them: run the lines of
    sentclf/bert.py:168-176 through the reference's own ctx_collator is not possible without a
    tokenizer, so this harness applies the same slice assignments to the same tensors."""
    tt = torch.zeros(ids.shape).long()
    seps = torch.where(ids == sep_id)[1].reshape(-1, 2 * n + 1)
    for t, sep in zip(tt, seps):
        t[:sep[n - 1]] = 1
        t[sep[n]:] = 1
    return tt


def run_reference(model, rows, n_ctx, sep_id, batch_size, num_labels, label_seed):
    """DataLoader(batch_size, shuffle=False) + pad-to-longest + forward + sigmoid, the loop of
    sentclf/apply_bert_unlabeled.py:92-124 with the reference MODEL doing the arithmetic."""
    rng = np.random.default_rng(label_seed)
    logits, losses, labels_all = [], [], []
    with torch.no_grad():
        model.eval()
        for i in range(0, len(rows), batch_size):
            ids, mask = pad(rows[i:i + batch_size])
            labels = torch.from_numpy((rng.random((ids.shape[0], num_labels)) < 0.2).astype(np.float32))
            inputs = {"input_ids": ids, "attention_mask": mask, "labels": labels}
            if n_ctx > 0:
                inputs["token_type_ids"] = ref_token_types(ids, sep_id, n_ctx)
            loss, pred = model(**inputs)
            logits.append(pred.numpy().copy())
            losses.append(float(loss))
            labels_all.append(labels.numpy())
            print(f"  ref batch {i // batch_size}: S={ids.shape[1]}", flush=True)
    return np.concatenate(logits), np.asarray(losses, np.float32), np.concatenate(labels_all)


def pack(rows):
    lens = np.asarray([len(r) for r in rows], np.int32)
    return np.concatenate([np.asarray(r, np.int32) for r in rows]), lens


def make_case(name, cfg, seed, rows, n_ctx, batch_size=8, weights="procedural"):
    import bert_model as ref_model
    from clip_amd.weights import procedural_state_dict, trained_like_state_dict
    from clip_amd.constants import SEP_ID
    sd = trained_like_state_dict(cfg, seed) if weights == "trained_like" else procedural_state_dict(cfg, seed)
    cls = (ref_model.BertSequenceMultilabelClassificationContext if n_ctx > 0
           else ref_model.BertForSequenceMultilabelClassification)
    model = cls(hf_config(cfg))
    missing = model.load_state_dict(sd, strict=True)
    print(name, "load_state_dict:", missing, "keys:", len(model.state_dict()))
    if n_ctx > 0:
        model.set_sep_token_id(SEP_ID)
        model.set_n_context_sentences(n_ctx)
    model.set_task("multilabel")
    logits, losses, labels = run_reference(model, rows, n_ctx, SEP_ID, batch_size, cfg.num_labels, seed + 1)
    ids, lens = pack(rows)
    probs = torch.sigmoid(torch.from_numpy(logits)).numpy()
    from dataclasses import asdict
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"),
        cfg=np.asarray([str(asdict(cfg))]), seed=np.int64(seed), n_ctx=np.int64(n_ctx),
        weights=np.asarray([weights]), batch_size=np.int64(batch_size), ids=ids, lens=lens, labels=labels,
        ref_logits=logits.astype(np.float32), ref_probs=probs.astype(np.float32),
        ref_batch_loss=losses)
    print(name, "done", logits.shape, "logit range", logits.min(), logits.max())


def main():
    os.makedirs(OUT, exist_ok=True)
    install_shims()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    from clip_amd.config import BertConfig
    from clip_amd import synth
    only = sys.argv[1:]          # e.g. `python oracle/gen_golden.py base_ctx_stress`: just these cases
    global make_case
    _make = make_case

    def make_case(name, *a, **kw):
        if not only or name in only:
            _make(name, *a, **kw)

    tiny = BertConfig(vocab_size=2000, hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
                      intermediate_size=512, max_position_embeddings=512)
    base = BertConfig()

    # tiny: structure / key-layout parity, many ragged lengths incl. minimum rows
    rows = synth.context_rows(29, 2, tiny.vocab_size, max_len=96, seed=11)
    rows += [np.asarray([101, 102, 102, 1500, 102, 102, 102], np.int32)]          # empty contexts
    rows += synth.context_rows(2, 2, tiny.vocab_size, max_len=512, seed=12, fixed_len=512)
    make_case("tiny_ctx", tiny, 3, rows, 2)
    rows = synth.context_rows(16, 1, tiny.vocab_size, max_len=64, seed=13)
    make_case("tiny_ctx_n1", tiny, 4, rows, 1)
    rows = synth.sentence_rows(24, tiny.vocab_size, max_len=64, seed=14)
    make_case("tiny_sent", tiny, 5, rows, 0)

    # full-size BERT-base
    rows = synth.context_rows(40, 2, base.vocab_size, max_len=256, seed=1002)
    rows += synth.context_rows(8, 2, base.vocab_size, max_len=256, seed=21, fixed_len=256)
    rows += synth.context_rows(6, 2, base.vocab_size, max_len=512, seed=22, sigma=1.0, seg_hi=200)
    rows += synth.context_rows(2, 2, base.vocab_size, max_len=512, seed=23, fixed_len=512)
    rows += [np.asarray([101, 102, 102, 2023, 102, 102, 102], np.int32),
             np.asarray([101, 5000, 102, 102, 2023, 2003, 102, 6000, 102, 102], np.int32)]
    make_case("base_ctx", base, 7, rows, 2)
    rows = synth.sentence_rows(48, base.vocab_size, max_len=64, seed=1001)
    make_case("base_sent", base, 8, rows, 0)

    # trained-like statistics (clip_amd.weights.trained_like_state_dict): logits beyond +-6, LayerNorm gamma
    # outliers, outlier hidden dimensions, heavy-tailed embeddings — the stress case of the bf16 budget
    rows = synth.context_rows(40, 2, base.vocab_size, max_len=256, seed=31)
    rows += synth.context_rows(8, 2, base.vocab_size, max_len=256, seed=32, fixed_len=256)
    rows += synth.context_rows(6, 2, base.vocab_size, max_len=512, seed=33, sigma=1.0, seg_hi=200)
    rows += synth.context_rows(2, 2, base.vocab_size, max_len=512, seed=34, fixed_len=512)
    make_case("base_ctx_stress", base, 9, rows, 2, weights="trained_like")
    rows = synth.sentence_rows(32, base.vocab_size, max_len=64, seed=35)
    make_case("base_sent_stress", base, 10, rows, 0, weights="trained_like")


if __name__ == "__main__":
    main()
