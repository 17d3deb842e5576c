"""nvBowtie score reduction (best / second best per read) and mapping quality through the C-ABI vs the
oracle's restatement of reduce_inl.h:71-160 and mapq.h:42-330: bit-exact io::Alignment words and MAPQ bytes."""
import numpy as np
import pytest
import torch

import nvbio_amd as nvb
from oracle import pyoracle as O


def make_hits(rng, n_reads, n_active, ragged):
    read_len = (rng.integers(30, 251, n_reads) if ragged else np.full(n_reads, 100)).astype(np.uint32)
    read_ids = rng.permutation(n_reads)[:n_active].astype(np.uint32)
    counts = rng.integers(0, 9, n_active)
    hb = np.zeros(n_active + 1, dtype=np.uint64); hb[1:] = np.cumsum(counts)
    total = int(hb[-1])
    # locations cluster around a few loci per read so that "same location", "within read_len/2" and
    # "distinct" all occur; scores repeat so that ties occur
    locus = rng.integers(0, 1 << 30, n_active)
    owner = np.repeat(np.arange(n_active), counts)
    loc = (locus[owner] + rng.choice([0, 0, 1, 7, 40, 60, 130, 5000, 100000], total)).astype(np.uint32)
    loc[rng.random(total) < 0.02] = 3                      # near zero: pos2 - min(pos2, dist)
    score = rng.integers(-60, 1, total).astype(np.int32) * rng.choice([1, 1, 3], total).astype(np.int32)
    rc = (rng.random(total) < 0.3).astype(np.uint8)
    return read_len, read_ids, hb, score, loc, rc


def test_alignment_invalid_word():
    assert int(nvb.lib().nvbio_hip_alignment_invalid()) == O.alignment_invalid()
    w = O.alignment_invalid()
    assert (w >> 32) == 0xFFFFFFFF and ((w >> 1) & 0x1FFFF) == (1 << 17) - 1 and ((w >> 18) & 0x3FF) == 255


@pytest.mark.gpu
@pytest.mark.parametrize("ragged", [False, True])
def test_score_reduce_and_mapq(cuda, ragged):
    rng = np.random.default_rng(77 + ragged)
    n_reads, n_active = 20000, 15000
    read_len, read_ids, hb, score, loc, rc = make_hits(rng, n_reads, n_active, ragged)
    d = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).view(dt)).to(cuda)
    e2e = nvb.SmithWatermanScoringScheme()
    dl = dict(read_len=d(read_len, np.int32) if ragged else None, fixed_read_len=0 if ragged else 100, max_read_len=250)
    best = nvb.BestAlignments(n_reads, e2e, device=cuda, **dl)
    exp = O.init_alignments(read_len, e2e.m_score_min)
    assert (best.data.cpu().numpy().view(np.uint64) == exp).all()
    assert not bool(best.is_aligned(0).any()) and int(best.score(0)[0]) == e2e.min_score(int(read_len[0]))
    for rnd in range(3):                                   # several extension rounds accumulate into the same best pair
        if rnd:
            read_len2, read_ids, hb, score, loc, rc = make_hits(rng, n_reads, n_active, ragged)
            score = score + 5 * rnd
        O.score_reduce(exp, hb, score, loc, rc, read_len, read_ids)
        nvb.score_reduce(best, d(hb, np.int64), d(score, np.int32), d(loc, np.int32), d(rc, np.uint8),
                         read_len=d(read_len, np.int32) if ragged else None, fixed_read_len=0 if ragged else 100, read_ids=d(read_ids, np.int32))
        torch.cuda.synchronize()
        got = best.data.cpu().numpy().view(np.uint64)
        bad = np.nonzero((got != exp).any(0))[0]
        assert bad.size == 0, (rnd, bad[:5], [hex(x) for x in got[:, bad[0]]], [hex(x) for x in exp[:, bad[0]]])
    assert bool(best.is_aligned(1).any()) and bool((~best.is_aligned(0)).any()) and bool((best.is_aligned(0) & ~best.is_aligned(1)).any())
    # mapping quality from the reduced pairs, both calculators, end-to-end and local schemes
    for scheme in (nvb.SmithWatermanScoringScheme(), nvb.SmithWatermanScoringScheme.local()):
        if scheme.m_match:                                 # local scheme: make the scores look like local scores
            b2 = nvb.BestAlignments(n_reads, scheme, device=cuda, **dl)
            s2 = (-score * 2).astype(np.int32)
            e2 = O.init_alignments(read_len, scheme.m_score_min)
            O.score_reduce(e2, hb, s2, loc, rc, read_len, read_ids)
            nvb.score_reduce(b2, d(hb, np.int64), d(s2, np.int32), d(loc, np.int32), d(rc, np.uint8),
                             read_len=d(read_len, np.int32) if ragged else None, fixed_read_len=0 if ragged else 100, read_ids=d(read_ids, np.int32))
            use, euse = b2, e2
        else:
            use, euse = best, exp
        for version in (2, 3):
            em = O.mapq(version, scheme.m_match, scheme.m_score_min, scheme.m_monotone, euse, read_len)
            gm = nvb.mapq(use, scheme, read_len=d(read_len, np.int32) if ragged else None, fixed_read_len=0 if ragged else 100,
                          version=version, max_read_len=250)
            torch.cuda.synchronize()
            gm = gm.cpu().numpy()
            bad = np.nonzero(gm != em)[0]
            assert bad.size == 0, (version, scheme.m_match, bad[:5], gm[bad[:5]], em[bad[:5]])
            assert len(set(em.tolist())) > 5               # a spread of qualities, not one constant
