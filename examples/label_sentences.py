"""Label the sentences of a notes CSV with a fine-tuned BERT+Context checkpoint on an MI355X.

    python examples/label_sentences.py CHECKPOINT_DIR clip_dnote_label_threshs.npy dummy_test_data.csv

What /root/reference/test_load.py demonstrates (hub checkpoint -> setters -> per-label thresholds ->
one sentence at a time -> label names), through the same model API (from_pretrained / set_* / forward ->
(loss, logits)), from a LOCAL checkpoint directory: config.json, model.safetensors | pytorch_model.bin,
vocab.txt.  For millions of sentences use `python -m clip_amd.apply_bert_unlabeled` instead.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_amd import collate, datasets
from clip_amd.bert_model import BertSequenceMultilabelClassificationContext
from clip_amd.constants import LABEL_TYPES
from clip_amd.wordpiece import BertWordPieceTokenizer


def label_sentences(checkpoint_dir, threshs_npy, csv_path, n_context=2, device="cuda"):
    """Yields (bce_loss, probabilities float32[7], label names) per CSV row, in file order."""
    tokenizer = BertWordPieceTokenizer(os.path.join(checkpoint_dir, "vocab.txt"))
    model = BertSequenceMultilabelClassificationContext.from_pretrained(checkpoint_dir)
    model.set_sep_token_id(tokenizer.sep_token_id)
    model.set_n_context_sentences(n_context)
    model.set_task("multilabel")
    model.to(device).eval()                      # no CPU execution path: the model must sit on the GPU
    thresholds = np.load(threshs_npy)            # float64[7]; a label fires when its probability is strictly above
    rows = datasets.SentEvalDataset(csv_path, "multilabel", n_context_sentences=n_context)
    for i in range(len(rows)):
        batch = collate.ctx_collator([rows[i]], "multilabel", tokenizer)
        tensors = {name: t.to(device) for name, t in collate.select_inputs(batch).items()}
        loss, logits = model(**tensors)
        probs = torch.sigmoid(logits)[0].cpu().numpy()
        yield float(loss), probs, [LABEL_TYPES[j] for j in np.flatnonzero(probs > thresholds)]


def main(checkpoint_dir, threshs_npy, csv_path):
    out = []
    for loss, probs, names in label_sentences(checkpoint_dir, threshs_npy, csv_path):
        print(f"pred_labels: {names}")
        out.append((loss, probs, names))
    return out


if __name__ == "__main__":
    main(*sys.argv[1:4])
