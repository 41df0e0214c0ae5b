"""Sentence datasets of the scoring path, restating /root/reference/sentclf/neural_baselines.py:
SentDataset (:122-167) and SentEvalDataset (:187-244).  Same tuples, same quirks; the per-row
pandas `.iloc` + `eval()` of the reference is replaced by one columnar pass at load time
(`ast.literal_eval`, never `eval`).  nltk's word_tokenize — the reference's fallback for a
`sentence` cell that is not a list literal — is not installed here; the fallback is `str.split`,
said where it applies.
"""
import ast

import numpy as np
import pandas as pd


def _parse_list(cell, lower):
    """'["a", "b"]' -> ['a', 'b'] (neural_baselines.py:145-153 / :195-201)."""
    if isinstance(cell, str) and cell.startswith("[") and cell.endswith("]"):
        try:
            v = ast.literal_eval(cell)
            if isinstance(v, (list, tuple)):
                return [str(x).lower() for x in v] if lower else list(v)
        except (ValueError, SyntaxError):
            pass
    if not isinstance(cell, str):
        raise AssertionError("sentence cell is not a string")      # the reference asserts the same
    toks = cell.split()          # reference: nltk.word_tokenize (absent here)
    return [x.lower() for x in toks] if lower else toks


def _parse_labels(cell):
    try:
        v = ast.literal_eval(cell) if isinstance(cell, str) else cell
        return list(v) if isinstance(v, (list, tuple)) else []
    except (ValueError, SyntaxError):
        return []


def read_notes_csv(fname):
    """pd.read_csv(fname) (neural_baselines.py:127), through pyarrow's multi-threaded CSV reader when it is
    installed and accepts the file (3x faster on a 16-core host; same frame: tests/test_datasets.py checks the
    two engines against each other on quoted, empty, NA-spelled and multi-line cells), else pandas' C parser."""
    try:
        import pyarrow  # noqa: F401
    except ImportError:
        return pd.read_csv(fname)
    try:
        return pd.read_csv(fname, engine="pyarrow")
    except Exception:           # anything the arrow reader refuses: the C parser decides
        return pd.read_csv(fname)


class SentDataset:
    """Rows -> (lower-cased tokens, labels, doc_id, sent_ix)   (neural_baselines.py:143-167)."""

    def __init__(self, fname, task, word2ix=None, df=None):
        sents = read_notes_csv(fname) if df is None else df
        self.sents = sents.dropna(subset=["sentence"]).reset_index()      # :129
        self.task = task
        self._ix_col = "sent_index" if "sent_index" in self.sents.columns else \
            ("sent_ix" if "sent_ix" in self.sents.columns else None)
        self.doc_ids = self.sents["doc_id"].to_numpy()
        self.sent_ixs = self.sents[self._ix_col].to_numpy() if self._ix_col else None
        self._sentence = self.sents["sentence"].tolist()
        self._labels = self.sents["labels"].tolist() if "labels" in self.sents.columns else None

    def __len__(self):
        return len(self.sents)

    def take(self, idx):
        """A dataset over the rows idx (in that order) of this one: the same object shape over sliced
        columns, nothing re-read.  A row's neighbours are then its neighbours IN idx (the reference looks
        them up by row offset, neural_baselines.py:219-235), which is what the chunked tokenizer relies on."""
        import copy
        sub = copy.copy(self)
        idx = np.asarray(idx, np.int64)
        sub.sents = self.sents.iloc[idx]
        sub.doc_ids = self.doc_ids[idx]
        sub.sent_ixs = None if self.sent_ixs is None else self.sent_ixs[idx]
        sub._sentence = [self._sentence[i] for i in idx]
        sub._labels = None if self._labels is None else [self._labels[i] for i in idx]
        if getattr(self, "_annot", None) is not None:
            sub._annot = [self._annot[i] for i in idx]
        return sub

    def _row_labels(self, idx):
        if self.task == "multilabel":
            return _parse_labels(self._labels[idx]) if self._labels is not None else []
        return self._labels[idx]

    def __getitem__(self, idx):
        sent = _parse_list(self._sentence[idx], lower=True)
        sent_ix = None if self.sent_ixs is None else self.sent_ixs[idx]
        return sent, self._row_labels(idx), self.doc_ids[idx], sent_ix


class SentEvalDataset(SentDataset):
    """Adds the +-n neighbour sentences of the same document (neural_baselines.py:193-244):
    offset loop -n..n skipping 0; expected index < 0 -> '<DOC_START>'; row idx+offset past the end
    of the file, or holding a different sentence index than expected -> '<DOC_END>' (so a negative
    offset can yield '<DOC_END>' too, e.g. after a duplicated sent_index: dummy_test_data.csv:7-8).
    Returns (tokens, sent_ix, annotator, labels, doc_id, contexts[2n])."""

    def __init__(self, fname, task, word2ix=None, n_context_sentences=0, doc_position=False, df=None):
        super().__init__(fname, task, word2ix, df=df)
        self.n_context_sentences = n_context_sentences
        if doc_position:
            raise NotImplementedError("doc_position is not on the scoring path")
        self._annot = self.sents["annotator"].tolist() if "annotator" in self.sents.columns else None

    def context_of(self, idx):
        n, ix = self.n_context_sentences, self.sent_ixs
        out = []
        for offset in range(-n, n + 1):
            if offset == 0:
                continue
            want = ix[idx] + offset
            j = idx + offset
            if want < 0:
                out.append("<DOC_START>")
            elif j >= len(self) or ix[j] != want:       # j < 0 wraps like pandas .iloc does
                out.append("<DOC_END>")
            else:
                out.append(" ".join(str(x) for x in _parse_list(self._sentence[j], lower=False)))
        return tuple(out)

    def __getitem__(self, idx):
        sent = _parse_list(self._sentence[idx], lower=False)       # :197 — not lower-cased here
        try:
            labels = self._row_labels(idx)
        except Exception:
            labels = []
        annotator = self._annot[idx] if self._annot is not None else ""
        return sent, self.sent_ixs[idx], annotator, labels, self.doc_ids[idx], self.context_of(idx)
