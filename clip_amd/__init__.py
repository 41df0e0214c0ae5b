"""clip_amd — MI355X-native bulk BERT(+Context) sentence scoring (the one hot path of
asappresearch/clip: sentclf/apply_bert_unlabeled.py -> bert_model.py -> sigmoid/threshold/max
-> top-k)."""
from .config import BertConfig  # noqa: F401
from .constants import LABEL_TYPES, label2abbrev  # noqa: F401
