"""Encoder geometry. Mirrors the fields of transformers.BertConfig the hot path reads
(/root/reference/sentclf/apply_bert_unlabeled.py:55-62, bert_model.py:13-19)."""
import json
import os
from dataclasses import dataclass, asdict


@dataclass
class BertConfig:
    vocab_size: int = 30522
    hidden_size: int = 768
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    intermediate_size: int = 3072
    max_position_embeddings: int = 512
    type_vocab_size: int = 2
    layer_norm_eps: float = 1e-12
    hidden_act: str = "gelu"
    num_labels: int = 7
    pad_token_id: int = 0

    @property
    def head_dim(self):
        return self.hidden_size // self.num_attention_heads

    @classmethod
    def from_pretrained(cls, path, **overrides):
        """`path` is a local directory holding config.json (or the json file itself).
        Hub names are not resolvable offline; they raise."""
        f = os.path.join(path, "config.json") if os.path.isdir(path) else path
        if not os.path.isfile(f):
            raise OSError(f"{path!r} is not a local checkpoint directory / config.json "
                          "(hub names are not supported: no network)")
        with open(f) as fh:
            raw = json.load(fh)
        fields = {k: raw[k] for k in cls.__dataclass_fields__ if k in raw}
        if "num_labels" not in fields and "id2label" in raw:
            fields["num_labels"] = len(raw["id2label"])
        fields.update({k: v for k, v in overrides.items() if k in cls.__dataclass_fields__})
        return cls(**fields)

    def save_pretrained(self, path):
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, "config.json"), "w") as fh:
            json.dump(asdict(self), fh, indent=1)

    def validate_for_device(self):
        if self.hidden_act != "gelu":
            raise ValueError("only erf-GELU ('gelu') is implemented")
        if self.head_dim != 64:
            raise ValueError("HIP attention kernel is built for head_dim 64")
        if self.hidden_size % 64 or self.intermediate_size % 64:
            raise ValueError("hidden/intermediate size must be multiples of 64")
