"""Reader side of the top-k file: how task-targeted pre-training consumes what the scoring path wrote.

/root/reference/sentclf/pretrain_tasks.py:34-98 (`SentContextDataset`): `pd.read_csv(focus_fname)` (:37, so
the file needs the header row `doc_id,sent_ix,...` that clip_amd.job.write_topk_csv writes), and for focus
record i the sentence row is the FIRST row of the notes table with that (doc_id, sent_ix) (:57-64:
boolean mask, `row.index[0]`); a record without a match is reported and skipped (:59-63).  Its context
window is then built by row offsets exactly as SentEvalDataset does (:82-91 == neural_baselines.py:219-235),
which clip_amd.datasets already restates.

Here: the same file read without pandas' per-row masks — one sort + searchsorted over the whole table
(50 M rows x 250 k focus records would be 250 k full-table scans in the reference).
"""
import csv

import numpy as np


def read_topk_csv(path):
    """(doc_ids int64, sent_ixs int64, scores float32 or None) from a file with a header naming at least
    doc_id and sent_ix, in file order (= rank order: score desc, doc_id asc, sent_ix asc)."""
    with open(path, newline="") as f:
        r = csv.reader(f)
        header = next(r)
        if "doc_id" not in header or "sent_ix" not in header:
            raise ValueError(f"{path}: header {header} lacks doc_id / sent_ix (sentclf/pretrain_tasks.py:57-58 "
                             "reads those columns; ttp_doc_sents.csv has no header and is not a focus file)")
        di, si = header.index("doc_id"), header.index("sent_ix")
        ci = header.index("max_label_score") if "max_label_score" in header else None
        d, s, c = [], [], []
        for row in r:
            d.append(int(row[di])); s.append(int(row[si]))
            if ci is not None:
                c.append(np.float32(row[ci]))
    return (np.asarray(d, np.int64), np.asarray(s, np.int64),
            np.asarray(c, np.float32) if ci is not None else None)


def focus_row_indices(table_doc_ids, table_sent_ixs, focus_doc_ids, focus_sent_ixs):
    """Row index in the notes table of each focus record: the first row whose (doc_id, sent_ix) equals
    the record's, -1 where there is none (pretrain_tasks.py:58-63).  Duplicate keys in the table (the
    reference's own fixture has one, dummy_test_data.csv:7-8) resolve to the first row, as `row.index[0]`."""
    td = np.asarray(table_doc_ids, np.int64); ts = np.asarray(table_sent_ixs, np.int64)
    fd = np.asarray(focus_doc_ids, np.int64); fs = np.asarray(focus_sent_ixs, np.int64)
    if len(td) == 0:
        return np.full(len(fd), -1, np.int64)
    order = np.lexsort((np.arange(len(td)), ts, td))          # by (doc_id, sent_ix), ties by row index
    sd, ss = td[order], ts[order]
    # position of the first sorted entry >= the key, on the pair (doc_id, sent_ix)
    lo = np.searchsorted(sd, fd, side="left")
    hi = np.searchsorted(sd, fd, side="right")
    out = np.full(len(fd), -1, np.int64)
    for i in range(len(fd)):          # per-record search inside its document's run (runs are short)
        a, b = lo[i], hi[i]
        if a == b:
            continue
        j = a + np.searchsorted(ss[a:b], fs[i], side="left")
        if j < b and ss[j] == fs[i]:
            out[i] = order[j]
    return out
