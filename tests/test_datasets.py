"""read_notes_csv (clip_amd/datasets.py): the arrow-backed read gives the frame pandas' C parser gives
(neural_baselines.py:127 is a plain pd.read_csv), on the cells that differ between CSV dialects."""
import pandas as pd
import pytest

from clip_amd import datasets

EDGE = '''doc_id,sent_index,sentence,labels
1,0,"['a', 'b, c']","[]"
1,1,,"[]"
1,2,NA,"[]"
1,3,"['he said ""hi""', 'x']","['Patient instructions']"
2,0,"line one
line two","[]"
2,1,null,
2,2,"  spaced  ",[]
3,0,nan,[]
3,1,"[]",[]
3,2,None,[]
3,3,"",[]
4,0,"['café', '中文']",[]
'''


def test_read_notes_csv_matches_c_parser(tmp_path):
    pytest.importorskip("pyarrow")
    p = tmp_path / "e.csv"
    p.write_text(EDGE, encoding="utf-8")
    a, b = pd.read_csv(p), datasets.read_notes_csv(p)
    assert a.equals(b) and list(a.dtypes) == list(b.dtypes)
    ds_a = datasets.SentEvalDataset(None, "multilabel", n_context_sentences=1, df=a)
    ds_b = datasets.SentEvalDataset(str(p), "multilabel", n_context_sentences=1)
    assert len(ds_a) == len(ds_b) == 6      # the six NA-spelled / empty sentences are dropped (:129)
    for i in range(len(ds_a)):
        ta, tb = ds_a[i], ds_b[i]
        assert ta[0] == tb[0] and ta[3] == tb[3] and ta[5] == tb[5] and int(ta[4]) == int(tb[4])


def test_read_notes_csv_falls_back(tmp_path):
    p = tmp_path / "ragged.csv"        # a row with an extra field: the arrow reader refuses it, the C parser raises too
    p.write_text("doc_id,sent_index,sentence\n1,0,a\n1,1,b,EXTRA\n")
    with pytest.raises(Exception):
        datasets.read_notes_csv(p)
    q = tmp_path / "ok.csv"
    q.write_text("doc_id,sent_index,sentence\n1,0,\"['a']\"\n")
    assert len(datasets.read_notes_csv(q)) == 1
