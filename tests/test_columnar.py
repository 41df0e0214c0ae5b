"""Columnar pre-tokenised corpus (clip_amd/columnar.py): round trip through memory-mapped .npy files
and chunking at document boundaries."""
import numpy as np
import pytest

from clip_amd import columnar, synth
from clip_amd.scoring import PackedRows


def _corpus(n=5000, seed=4):
    flat, off = synth.packed_context_rows(n, 2, 3000, 128, seed=seed)
    d, s = synth.documents(n, seed=seed, lo=1, hi=60)
    return PackedRows(flat, off), d, s.astype(np.int64)


def test_round_trip_and_chunks(tmp_path):
    rows, d, s = _corpus()
    columnar.save_rows(str(tmp_path / "c"), rows, d, s, n_context_sentences=2, sep_token_id=102)
    c = columnar.ColumnarCorpus(str(tmp_path / "c"))
    assert len(c) == len(rows) and c.meta["tokens"] == int(rows.offsets[-1]) and c.meta["n_context_sentences"] == 2
    assert isinstance(c.flat, np.memmap)
    bounds = c.chunk_bounds(300)
    assert bounds[0][0] == 0 and bounds[-1][1] == len(rows)
    assert all(a1 == b0 for (_, b0), (a1, _) in zip(bounds, bounds[1:]))            # a partition
    for a, b in bounds:
        assert b > a and (b == len(rows) or d[b] != d[b - 1])                        # ends on a document boundary
        assert b - a <= 300 or len(set(d[a:b])) == 1                                 # only a long document exceeds it
    # every rank's chunks together = the corpus, each row once, contents intact
    seen = np.zeros(len(rows), int)
    for world in (1, 3):
        seen[:] = 0
        for rank in range(world):
            for ci, a, b, (r, dd, ss) in c.chunks(300, rank, world):
                assert ci % world == rank
                seen[a:b] += 1
                assert np.array_equal(r.flat, rows.flat[rows.offsets[a]:rows.offsets[b]])
                assert np.array_equal(r.offsets, rows.offsets[a:b + 1] - rows.offsets[a])
                assert np.array_equal(dd, d[a:b]) and np.array_equal(ss, s[a:b])
        assert (seen == 1).all()
    mem = columnar.ColumnarCorpus.from_arrays(rows, d, s)
    assert mem.chunk_bounds(300) == bounds


def test_one_long_document_and_bad_directory(tmp_path):
    rows, d, s = _corpus(400)
    d[:] = 7
    c = columnar.ColumnarCorpus.from_arrays(rows, d, s)
    assert c.chunk_bounds(50) == [(0, 400)]
    columnar.save_rows(str(tmp_path / "c"), rows, d, s)
    np.save(str(tmp_path / "c" / "sent_ixs.npy"), s[:-1])
    with pytest.raises(ValueError):
        columnar.ColumnarCorpus(str(tmp_path / "c"))
    with pytest.raises(ValueError):
        columnar.save_rows(str(tmp_path / "d"), rows, d[:-1], s)


def test_graded_chunk_bounds():
    _, d, _ = _corpus(20000, seed=9)
    b = columnar.graded_chunk_bounds(d, 4096, first_rows=256)
    assert b[0][0] == 0 and b[-1][1] == len(d)
    assert all(a1 == b0 for (_, b0), (a1, _) in zip(b, b[1:]))                       # a partition
    assert all(e == len(d) or d[e] != d[e - 1] for _, e in b)                        # whole documents
    sizes = [e - a for a, e in b]
    assert sizes[0] <= 256 and max(sizes) <= 4096 and max(sizes) > 2048              # small first, grows to chunk_rows
    assert columnar.graded_chunk_bounds(d, 100, first_rows=256) == columnar.chunk_bounds(d, 100)
    assert columnar.graded_chunk_bounds(d[:0], 100) == []


def test_merge_parts_equals_the_whole(tmp_path):
    """The ranks' --save_rows parts (consecutive row ranges) joined by merge_parts = the corpus saved in one go."""
    rows, d, s = _corpus(3000, seed=9)
    columnar.save_rows(str(tmp_path / "whole"), rows, d, s, n_context_sentences=2, sep_token_id=102)
    cuts = [0, 700, 700, 2100, 3000]                      # an empty part too (a rank without documents)
    parts = []
    for k, (a, b) in enumerate(zip(cuts, cuts[1:])):
        o = rows.offsets[a:b + 1]
        columnar.save_rows(str(tmp_path / f"p{k}"), PackedRows(rows.flat[o[0]:o[-1]], o - o[0]), d[a:b], s[a:b],
                           n_context_sentences=2, sep_token_id=102)
        parts.append(str(tmp_path / f"p{k}"))
    columnar.merge_parts(parts, str(tmp_path / "joined"), n_context_sentences=2, sep_token_id=102)
    w, j = columnar.ColumnarCorpus(str(tmp_path / "whole")), columnar.ColumnarCorpus(str(tmp_path / "joined"))
    for k in columnar.FILES:
        assert np.array_equal(getattr(w, k), getattr(j, k)) and getattr(w, k).dtype == getattr(j, k).dtype
    assert w.meta == j.meta
    import os
    assert not any(os.path.exists(p) for p in parts)      # the parts are gone


def test_rank_ranges_are_whole_documents_balanced_by_weight():
    """What every rank of the CLI computes from the document index alone (clip_amd/apply_bert_unlabeled.py): a
    partition into contiguous ranges that end on document boundaries, within one document of an even token share."""
    from clip_amd import topk
    rows, d, s = _corpus(20000, seed=3)
    w = rows.lens.astype(np.int64)
    for world in (2, 3, 8):
        r = [topk.shard_documents(d, w, world, k) for k in range(world)]
        assert r[0][0] == 0 and r[-1][1] == len(d) and all(a[1] == b[0] for a, b in zip(r, r[1:]))
        for a, b in r:
            assert b == len(d) or a == b or d[b] != d[b - 1]
        share = np.array([w[a:b].sum() for a, b in r]) * world / w.sum()
        assert np.abs(share - 1).max() < 0.02
