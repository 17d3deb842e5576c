"""The numpy all-mapping driver over the oracle (tests/oracle_driver.all_mapping, the checker of tests/test_all_mapping_gpu.py)
against first principles on a small case: what it reports is what an exhaustive scan of the genome finds at the seeded placements."""
import numpy as np
import torch

import nvbio_amd as nvb
from nvbio_amd import aligner as A, workloads as W
from oracle import pyoracle as O
from tests import oracle_driver as OD


def test_all_mapping_oracle_reports_every_copy():
    rng = np.random.default_rng(77)
    text = rng.integers(0, 4, 1 << 14, dtype=np.uint8)
    unit = text[1000:1100].copy()
    copies = [1000, 5000, 9000, 13000]
    for c in copies[1:]:
        text[c:c + 100] = unit
    text[9050] = (text[9050] + 1) & 3                                   # the third copy differs by one base
    host, rhost = O.FMIndex(text), O.FMIndex(text[::-1].copy())
    reads = [unit.copy(), (3 - unit)[::-1].copy(), text[3000:3100].copy(), rng.integers(0, 4, 100, dtype=np.uint8)]
    params = A.Params()
    scheme = nvb.SmithWatermanScoringScheme()
    gw = W._pack_chunked(torch.from_numpy(text), 2, True).numpy().view(np.uint32)
    e = OD.all_mapping(host, rhost, reads, gw, text.size, params, scheme, 2)
    rid = e["read_id"]; pos = (e["alignments_scored"] >> np.uint64(32)).astype(np.int64); rc = ((e["alignments_scored"] >> np.uint64(28)) & np.uint64(1)).astype(np.int64)
    w = e["alignments_scored"] & np.uint64(0xFFFFFFFF)
    score = np.where(w & np.uint64(1), -1, 1) * ((w >> np.uint64(1)) & np.uint64(0x1FFFF)).astype(np.int64)
    for r, strand in ((0, 0), (1, 1)):
        k = rid == r
        assert (rc[k] == strand).all() and sorted(pos[k].tolist()) == copies              # all four copies, once each
        assert sorted(score[k].tolist()) == [-5, 0, 0, 0]                                 # Q30 mismatch = -(2 + 4 * 30 / 40) at the third copy
    assert pos[rid == 2].tolist() == [3000] and score[rid == 2].tolist() == [0]
    assert (rid != 3).all()                                                                # the random read: nothing reaches min_score
    assert (score >= scheme.min_score(100)).all()
    # finished words: window begin = read start - band/2, edit distance 0/1, final score = extension score
    fin = e["alignments"]
    fpos = (fin >> np.uint64(32)).astype(np.int64); fw = fin & np.uint64(0xFFFFFFFF)
    assert (fpos == np.maximum(pos - 15, 0)).all()
    assert (((fw >> np.uint64(18)) & np.uint64(0x3FF)).astype(np.int64) == (score < 0)).all()
    fscore = np.where(fw & np.uint64(1), -1, 1) * ((fw >> np.uint64(1)) & np.uint64(0x1FFFF)).astype(np.int64)
    assert (fscore == score).all()
    assert e["stats"]["hits"] >= e["stats"]["unique"] >= rid.size == 9


def test_all_mapping_oracle_matches_committed_vectors():
    """The checker itself is pinned: tests/golden/all_mapping_vectors.npz (written by tests/golden/make_all_mapping_vectors.py) holds its
    output on a seeded case -- one batch and 257-row batches, a two-sequence reference, ragged reads with qualities."""
    import os
    from tests.golden import make_all_mapping_vectors as G
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "all_mapping_vectors.npz"))
    text, reads, quals = G.case()
    for bs in (1 << 20, 257):
        e = G.run(text, reads, quals, bs)
        m = e["read_id"].size
        assert m == g["read_id_%d" % bs].size and m > 200
        assert (e["read_id"] == g["read_id_%d" % bs]).all() and (e["alignments_scored"] == g["scored_%d" % bs]).all() and (e["alignments"] == g["finished_%d" % bs]).all()
        assert (e["tb"]["cigar_len"] == g["cigar_len_%d" % bs]).all() and (e["tb"]["cigar"][:m, :12] == g["cigar_%d" % bs]).all() and (e["mds_len"] == g["mds_len_%d" % bs]).all()
        assert [e["stats"]["hits"], e["stats"]["ranges"], e["stats"]["unique"]] == g["stats_%d" % bs].tolist()
    assert g["read_id_257"].size > g["read_id_%d" % (1 << 20)].size                      # de-duplication is per batch
