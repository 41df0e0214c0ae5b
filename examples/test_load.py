"""The reference's load-and-score example (/root/reference/test_load.py) on an MI355X, line for line,
with the two hub names replaced by a LOCAL checkpoint directory (config.json, model.safetensors |
pytorch_model.bin, vocab.txt) and local paths for the thresholds and the CSV:

    python examples/test_load.py CHECKPOINT_DIR clip_dnote_label_threshs.npy dummy_test_data.csv
"""
import os
import sys
from functools import partial

import numpy as np
import torch
from torch.utils.data import DataLoader

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_amd.datasets import SentEvalDataset                                   # sentclf/neural_baselines.py
from clip_amd.bert_model import BertSequenceMultilabelClassificationContext    # sentclf/bert_model.py
from clip_amd.collate import ctx_collator, select_inputs                        # sentclf/bert.py
from clip_amd.constants import LABEL_TYPES                                      # sentclf/constants.py
from clip_amd.wordpiece import BertWordPieceTokenizer                           # AutoTokenizer, from vocab.txt

N_CONTEXT_SENTENCES = 2
TASK = 'multilabel'

label2id = {label: ix for ix, label in enumerate(LABEL_TYPES)}
id2label = {ix: label for label, ix in label2id.items()}


def main(checkpoint_dir, threshs_npy, csv_path, device="cuda"):
    model = BertSequenceMultilabelClassificationContext.from_pretrained(checkpoint_dir)
    tokenizer = BertWordPieceTokenizer(os.path.join(checkpoint_dir, "vocab.txt"))

    model.set_sep_token_id(tokenizer.sep_token_id)
    model.set_n_context_sentences(N_CONTEXT_SENTENCES)
    model.set_task(TASK)
    model.to(device).eval()                       # the reference example stays on the CPU; this engine has no CPU path

    # load model thresholds
    label_threshs = np.load(threshs_npy)

    dataset = SentEvalDataset(csv_path, TASK, n_context_sentences=N_CONTEXT_SENTENCES)
    loader = DataLoader(dataset, batch_size=1, shuffle=False,
                        collate_fn=partial(ctx_collator, task=TASK, tokenizer=tokenizer))

    results = []
    for x in loader:
        inputs = {k: v.to(device) for k, v in select_inputs(x).items()}
        loss, pred = model(**inputs)
        pred = torch.sigmoid(pred)
        pred_label_ids = np.where(pred.data.squeeze().cpu().numpy() > label_threshs)[0]
        pred_labels = [id2label[lid] for lid in pred_label_ids]
        print(f"pred_labels: {pred_labels}")
        results.append((float(loss), pred.squeeze().cpu().numpy(), pred_labels))
    return results


if __name__ == "__main__":
    main(*sys.argv[1:4])
