"""Procedural (counter-based) weights for offline parity and benchmarking.

No pretrained checkpoint is reachable offline, so parity vectors and bench runs use weights
drawn from a generator DEFINED HERE: numpy Philox keyed by (seed, crc32(tensor name)).  The same
state dict is loaded into the reference model (oracle/gen_golden.py) and into the HIP engine, so
only seeds + inputs + outputs need to be committed as fixtures.

Key layout is the reference's: 201 keys for the context model
(/root/reference/sentclf/bert_model.py:13-21 + transformers BertModel), see state_dict_keys().
"""
import zlib

import numpy as np
import torch

from .config import BertConfig


def state_dict_shapes(cfg: BertConfig, mlm_head: bool = False):
    """mlm_head: also the BertLMPredictionHead of BertContextMLMSwitch (bert_model.py:113), 207 keys."""
    H, I = cfg.hidden_size, cfg.intermediate_size
    s = {
        "bert.embeddings.word_embeddings.weight": (cfg.vocab_size, H),
        "bert.embeddings.position_embeddings.weight": (cfg.max_position_embeddings, H),
        "bert.embeddings.token_type_embeddings.weight": (cfg.type_vocab_size, H),
        "bert.embeddings.LayerNorm.weight": (H,),
        "bert.embeddings.LayerNorm.bias": (H,),
    }
    for i in range(cfg.num_hidden_layers):
        p = f"bert.encoder.layer.{i}."
        for n in ("query", "key", "value"):
            s[p + f"attention.self.{n}.weight"] = (H, H)
            s[p + f"attention.self.{n}.bias"] = (H,)
        s[p + "attention.output.dense.weight"] = (H, H)
        s[p + "attention.output.dense.bias"] = (H,)
        s[p + "attention.output.LayerNorm.weight"] = (H,)
        s[p + "attention.output.LayerNorm.bias"] = (H,)
        s[p + "intermediate.dense.weight"] = (I, H)
        s[p + "intermediate.dense.bias"] = (I,)
        s[p + "output.dense.weight"] = (H, I)
        s[p + "output.dense.bias"] = (H,)
        s[p + "output.LayerNorm.weight"] = (H,)
        s[p + "output.LayerNorm.bias"] = (H,)
    s["bert.pooler.dense.weight"] = (H, H)
    s["bert.pooler.dense.bias"] = (H,)
    s["classifier.weight"] = (cfg.num_labels, H)
    s["classifier.bias"] = (cfg.num_labels,)
    if mlm_head:
        s["mlm_predictions.bias"] = (cfg.vocab_size,)
        s["mlm_predictions.transform.dense.weight"] = (H, H)
        s["mlm_predictions.transform.dense.bias"] = (H,)
        s["mlm_predictions.transform.LayerNorm.weight"] = (H,)
        s["mlm_predictions.transform.LayerNorm.bias"] = (H,)
        s["mlm_predictions.decoder.weight"] = (cfg.vocab_size, H)
        s["mlm_predictions.decoder.bias"] = (cfg.vocab_size,)
    return s


def state_dict_keys(cfg: BertConfig):
    return list(state_dict_shapes(cfg).keys())


def procedural_state_dict(cfg: BertConfig, seed: int = 0, weight_std: float = 0.02, mlm_head: bool = False):
    """name -> fp32 torch tensor. Linear/embedding weights and biases ~ N(0, weight_std);
    LayerNorm gamma = 1 + 0.1 N(0,1), beta = 0.1 N(0,1) so the affine part is exercised."""
    out = {}
    for name, shape in state_dict_shapes(cfg, mlm_head).items():
        g = np.random.Generator(np.random.Philox(key=[seed, zlib.crc32(name.encode())]))
        x = g.standard_normal(shape, dtype=np.float32)
        if "LayerNorm.weight" in name:
            x = 1.0 + 0.1 * x
        elif "LayerNorm.bias" in name:
            x = 0.1 * x
        else:
            x = weight_std * x
        out[name] = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
    return out


def trained_like_state_dict(cfg: BertConfig, seed: int = 0):
    """Weights with the STATISTICS of a fine-tuned BERT checkpoint, for stress-testing the bf16 budget
    (no checkpoint is reachable offline; procedural_state_dict is benign: logits within +-1.3, LayerNorm
    gamma 1 +- 0.1, no outliers).  Same Philox keying, plus:
      * dense weights N(0, 0.03), embeddings heavy-tailed (Student t, 3 dof, clipped to +-1);
      * 3 OUTLIER hidden dimensions: every block's output projection writes large values there
        (rows x4, bias +-1) and every LayerNorm carries gamma x4 and beta +-1 there — the
        residual-stream outliers of trained transformers;
      * 6 more LayerNorm gamma entries x5..x10 per LayerNorm;
      * classifier N(0, 0.045), bias N(0, 1): reference logits span beyond +-6."""
    sd = procedural_state_dict(cfg, seed, weight_std=0.03)
    H = cfg.hidden_size
    g = np.random.Generator(np.random.Philox(key=[seed, 0x74726e64]))
    outl = g.choice(H, 3, replace=False)
    sign = g.choice([-1.0, 1.0], 3).astype(np.float32)
    for name in list(sd):
        x = sd[name].numpy().copy()
        gn = np.random.Generator(np.random.Philox(key=[seed + 1, zlib.crc32(name.encode())]))
        if name.endswith("word_embeddings.weight") or name.endswith("position_embeddings.weight"):
            x = np.clip(0.03 * gn.standard_t(3, x.shape), -1.0, 1.0).astype(np.float32)
        elif "LayerNorm.weight" in name:
            big = gn.choice(H, 6, replace=False)
            x[big] *= gn.uniform(5.0, 10.0, 6).astype(np.float32)
            x[outl] = 4.0 * np.abs(x[outl])
        elif "LayerNorm.bias" in name:
            x[outl] += 1.0 * sign
        elif name.endswith("output.dense.weight"):          # attention.output.dense and output.dense
            x[outl, :] *= 4.0
        elif name.endswith("output.dense.bias"):
            x[outl] += 1.0 * sign
        elif name == "classifier.weight":
            x = (0.045 * gn.standard_normal(x.shape)).astype(np.float32)
        elif name == "classifier.bias":
            x = gn.standard_normal(x.shape).astype(np.float32)
        sd[name] = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
    return sd
