"""Local-file BERT WordPiece tokenizer (no hub, no `transformers` import).

The reference tokenises with HF `BertTokenizer.from_pretrained(name)` (sentclf/apply_bert_unlabeled.py:
43-47, test_load.py:22) and calls it as `tokenizer(text_or_list, padding=True, max_length=512,
truncation=True)` (sentclf/bert.py:57,105,152).  This class restates that call for an uncased BERT
vocabulary read from a local vocab.txt: BertNormalizer (clean text, space CJK, strip accents,
lower-case) -> BertPreTokenizer (whitespace + isolated punctuation) -> greedy longest-match WordPiece
("##" continuation, words over 100 chars -> [UNK]) -> [CLS] .. [SEP] template, longest-first
truncation of a single sequence, right padding to the longest row of the call.  Special tokens
written literally in the text ("[SEP]", which ctx_collator joins segments with, sentclf/bert.py:104)
are cut out before normalisation and mapped to their ids, as HF's added-token matching does.
Pinned by tests/golden/collate_*.json, generated with the reference's collators over HF's tokenizer.
"""
import re
import unicodedata


def _is_whitespace(ch):
    return ch in " \t\n\r" or unicodedata.category(ch) == "Zs"


def _is_control(ch):
    if ch in "\t\n\r":
        return False
    return unicodedata.category(ch).startswith("C")


def _is_punctuation(ch):
    cp = ord(ch)
    if 33 <= cp <= 47 or 58 <= cp <= 64 or 91 <= cp <= 96 or 123 <= cp <= 126:
        return True
    return unicodedata.category(ch).startswith("P")


def _is_cjk(cp):
    return (0x4E00 <= cp <= 0x9FFF or 0x3400 <= cp <= 0x4DBF or 0x20000 <= cp <= 0x2A6DF or
            0x2A700 <= cp <= 0x2B73F or 0x2B740 <= cp <= 0x2B81F or 0x2B820 <= cp <= 0x2CEAF or
            0xF900 <= cp <= 0xFAFF or 0x2F800 <= cp <= 0x2FA1F)


class BertWordPieceTokenizer:
    SPECIALS = ("[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]")

    def __init__(self, vocab_file, do_lower_case=True, max_input_chars_per_word=100):
        with open(vocab_file, encoding="utf-8") as f:
            self.vocab = {line.rstrip("\n"): i for i, line in enumerate(f)}
        self.do_lower_case = do_lower_case
        self.max_chars = max_input_chars_per_word
        for t in self.SPECIALS:
            if t not in self.vocab:
                raise ValueError(f"vocab lacks {t}")
        self.pad_token_id = self.vocab["[PAD]"]
        self.unk_token_id = self.vocab["[UNK]"]
        self.cls_token_id = self.vocab["[CLS]"]
        self.sep_token_id = self.vocab["[SEP]"]
        self.mask_token_id = self.vocab["[MASK]"]
        self._special_re = re.compile("(" + "|".join(re.escape(t) for t in self.SPECIALS) + ")")
        self._word_cache = {}

    def __len__(self):
        """len(tokenizer) of HF's tokenizer: the vocabulary size (pretrain_tasks.py:250 draws random words from it)"""
        return len(self.vocab)

    # ---- normaliser + pre-tokeniser
    def _normalize(self, text):
        out = []
        for ch in text:
            cp = ord(ch)
            if cp == 0 or cp == 0xFFFD or _is_control(ch):
                continue
            out.append(" " if _is_whitespace(ch) else ch)
        text = "".join(out)
        text = "".join(f" {ch} " if _is_cjk(ord(ch)) else ch for ch in text)
        if self.do_lower_case:     # strip_accents=None follows lowercase
            text = "".join(ch for ch in unicodedata.normalize("NFD", text) if unicodedata.category(ch) != "Mn")
            text = text.lower()
        return text

    @staticmethod
    def _split_words(text):
        words, cur = [], []
        for ch in text:
            if ch.isspace():
                if cur:
                    words.append("".join(cur)); cur = []
            elif _is_punctuation(ch):
                if cur:
                    words.append("".join(cur)); cur = []
                words.append(ch)
            else:
                cur.append(ch)
        if cur:
            words.append("".join(cur))
        return words

    def _wordpiece(self, word):
        ids = self._word_cache.get(word)
        if ids is not None:
            return ids
        if len(word) > self.max_chars:
            ids = (self.unk_token_id,)
        else:
            ids, start, n = [], 0, len(word)
            while start < n:
                end, cur = n, None
                while start < end:
                    sub = word[start:end]
                    if start > 0:
                        sub = "##" + sub
                    cur = self.vocab.get(sub)
                    if cur is not None:
                        break
                    end -= 1
                if cur is None:
                    ids = [self.unk_token_id]
                    break
                ids.append(cur)
                start = end
            ids = tuple(ids)
        self._word_cache[word] = ids
        return ids

    def encode_plain(self, text):
        """token ids of `text` without [CLS]/[SEP] template and without truncation."""
        ids = []
        for part in self._special_re.split(text):
            if not part:
                continue
            if part in self.SPECIALS:
                ids.append(self.vocab[part])
                continue
            for w in self._split_words(self._normalize(part)):
                ids.extend(self._wordpiece(w))
        return ids

    def encode(self, text, max_length=512, truncation=True):
        ids = self.encode_plain(text)
        if truncation and max_length is not None and len(ids) > max_length - 2:
            ids = ids[:max_length - 2]
        return [self.cls_token_id] + ids + [self.sep_token_id]

    def __call__(self, text, padding=True, max_length=512, truncation=True):
        """Mirrors the HF call of the collators: a str gives flat lists, a list of str gives lists
        of rows right-padded to the longest row (padding=True)."""
        if isinstance(text, str):
            ids = self.encode(text, max_length, truncation)
            return {"input_ids": ids, "token_type_ids": [0] * len(ids), "attention_mask": [1] * len(ids)}
        rows = [self.encode(t, max_length, truncation) for t in text]
        S = max((len(r) for r in rows), default=0) if padding else None
        out = {"input_ids": [], "token_type_ids": [], "attention_mask": []}
        for r in rows:
            pad = (S - len(r)) if padding else 0
            out["input_ids"].append(r + [self.pad_token_id] * pad)
            out["token_type_ids"].append([0] * (len(r) + pad))
            out["attention_mask"].append([1] * len(r) + [0] * pad)
        return out
