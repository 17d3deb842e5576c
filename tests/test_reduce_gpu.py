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


@pytest.mark.gpu
@pytest.mark.parametrize("anchor", [0, 1])
@pytest.mark.parametrize("policy", [0, 1, 2, 3])
def test_score_reduce_paired(cuda, anchor, policy):
    """score_reduce_paired_kernel (reduce_inl.h:355-500): paired / unpaired updates of the four best slots, both anchors, all
    pairing policies; several rounds accumulate; bit-exact io::Alignment words vs the oracle."""
    rng = np.random.default_rng(900 + 10 * anchor + policy)
    n_reads, n_active = 6000, 5000
    sch = nvb.SmithWatermanScoringScheme()
    read_len = rng.integers(50, 251, n_reads).astype(np.uint32)
    d = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).view(dt)).to(cuda)
    rl = d(read_len, np.int32)
    best = nvb.BestAlignments(n_reads, sch, read_len=rl, max_read_len=250, device=cuda, mate=0)
    best_o = nvb.BestAlignments(n_reads, sch, read_len=rl, max_read_len=250, device=cuda, mate=1)
    e = O.init_alignments(read_len, sch.m_score_min, 0)
    eo = O.init_alignments(read_len, sch.m_score_min, 1)
    assert (best_o.data.cpu().numpy().view(np.uint64) == eo).all()
    score_limit = -200
    for rnd in range(3):
        a = anchor if rnd < 2 else 1 - anchor                 # the second anchor pass of the reference comes after the first
        read_ids = rng.permutation(n_reads)[:n_active].astype(np.uint32)
        counts = rng.integers(0, 6, n_active)
        hb = np.zeros(n_active + 1, np.uint64); hb[1:] = np.cumsum(counts)
        tot = int(hb[-1])
        owner = np.repeat(np.arange(n_active), counts)
        locus = rng.integers(1000, 1 << 28, n_active)
        loc = (locus[owner] + rng.choice([0, 0, 2, 30, 70, 900, 50000], tot)).astype(np.uint32)
        sink = (loc + rng.integers(40, 260, tot)).astype(np.uint32)
        score = (-rng.integers(0, 60, tot)).astype(np.int32)
        rc = (rng.random(tot) < 0.4).astype(np.uint8)
        o_loc = (loc + rng.integers(-400, 400, tot)).astype(np.uint32)
        o_sink = (o_loc + rng.integers(40, 260, tot)).astype(np.uint32)
        o_sink2 = (o_loc + rng.integers(40, 260, tot)).astype(np.uint32)
        o_score = np.where(rng.random(tot) < 0.6, -rng.integers(0, 60, tot), -100000).astype(np.int32)     # -100000 = scheme_type::worst_score stand-in
        o_score2 = np.where(rng.random(tot) < 0.2, -rng.integers(0, 60, tot), -100000).astype(np.int32)
        O.score_reduce_paired(e, eo, hb, loc, sink, score, rc, o_loc, o_sink, o_sink2, o_score, o_score2, read_len, a, policy, True, score_limit, read_ids)
        nvb.score_reduce_paired(best, best_o, d(hb, np.int64), d(loc, np.int32), d(sink, np.int32), d(score, np.int32), d(rc, np.uint8),
                                d(o_loc, np.int32), d(o_sink, np.int32), d(o_sink2, np.int32), d(o_score, np.int32), d(o_score2, np.int32),
                                anchor=a, pe_policy=policy, pe_unpaired=True, score_limit=score_limit, read_len=rl, read_ids=d(read_ids, np.int32))
        torch.cuda.synchronize()
        g, go = best.data.cpu().numpy().view(np.uint64), best_o.data.cpu().numpy().view(np.uint64)
        bad = np.nonzero((g != e).any(0) | (go != eo).any(0))[0]
        assert bad.size == 0, (rnd, bad[:5], [hex(x) for x in g[:, bad[0]]], [hex(x) for x in e[:, bad[0]]], [hex(x) for x in go[:, bad[0]]], [hex(x) for x in eo[:, bad[0]]])
    paired = (e[0] >> np.uint64(30)) & np.uint64(1)
    assert 0 < int(paired.sum()) < n_reads                   # both paired and unpaired outcomes occur
    o_len = rng.integers(50, 251, n_reads).astype(np.uint32)
    for version in (2, 3):
        em = O.mapq_paired(version, sch.m_match, sch.m_score_min, sch.m_monotone, e, eo, read_len, o_len)
        gm = nvb.mapq_paired(best, best_o, sch, read_len=rl, o_read_len=d(o_len, np.int32), version=version, max_read_len=250).cpu().numpy()
        assert (gm == em).all(), (version, np.nonzero(gm != em)[0][:5])
        assert len(set(em.tolist())) > 3


@pytest.mark.gpu
@pytest.mark.parametrize("anchor", [0, 1])
def test_opposite_mate_windows(cuda, anchor):
    """BestOppositeScoreStream::init_context: thresholds, strands and genome windows of the opposite mates, all pairing policies,
    fragment-length limits incl. the genome ends, best pairs with and without a paired second best"""
    rng = np.random.default_rng(950 + anchor)
    n_reads, n_hits, G = 3000, 20000, 1_000_000
    d = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).view(dt)).to(cuda)
    a_len = rng.integers(50, 251, n_reads).astype(np.uint32); o_len = rng.integers(50, 251, n_reads).astype(np.uint32)
    for sch in (nvb.SmithWatermanScoringScheme(), nvb.SmithWatermanScoringScheme.local()):
        # some best pairs from a reduce round, so that compute_target_score sees paired seconds and the skip rule fires
        e, eo = O.init_alignments(a_len, sch.m_score_min, 0), O.init_alignments(o_len, sch.m_score_min, 1)
        counts = rng.integers(0, 4, n_reads); hb = np.zeros(n_reads + 1, np.uint64); hb[1:] = np.cumsum(counts); tot = int(hb[-1])
        loc = rng.integers(0, G - 300, tot).astype(np.uint32); sink = (loc + 100).astype(np.uint32)
        sgn = 1 if sch.m_match else -1
        sc = (sgn * rng.integers(0, 60, tot)).astype(np.int32); rc = (rng.random(tot) < 0.5).astype(np.uint8)
        ol = (loc + 200).astype(np.uint32); osk = (ol + 100).astype(np.uint32)
        os1 = np.where(rng.random(tot) < 0.7, sgn * rng.integers(0, 60, tot), -100000).astype(np.int32)
        O.score_reduce_paired(e, eo, hb, loc, sink, sc, rc, ol, osk, osk, os1, np.full(tot, -100000, np.int32), a_len, anchor, 1, True, -1000)
        best = nvb.BestAlignments(n_reads, sch, read_len=d(a_len, np.int32), max_read_len=250, device=cuda)
        best_o = nvb.BestAlignments(n_reads, sch, read_len=d(o_len, np.int32), max_read_len=250, device=cuda, mate=1)
        best.data.copy_(d(e, np.int64)); best_o.data.copy_(d(eo, np.int64))
        h_read = rng.integers(0, n_reads, n_hits).astype(np.uint32); h_rc = (rng.random(n_hits) < 0.5).astype(np.uint8)
        h_loc = rng.choice([0, 5, 300, G - 10, G - 400], n_hits).astype(np.uint32) + rng.integers(0, 5, n_hits).astype(np.uint32) * 1000
        h_loc = np.where(rng.random(n_hits) < 0.7, rng.integers(0, G - 1, n_hits), h_loc).astype(np.uint32)
        revisit = rng.random(n_hits) < 0.05                   # hits at an already recorded location
        h_loc[revisit] = (e[0, h_read[revisit]] >> np.uint64(32)).astype(np.uint32)
        h_score = (sgn * rng.integers(0, 80, n_hits)).astype(np.int32)
        for policy in (0, 1, 2, 3):
            for (mn, mx, ov) in ((0, 500, True), (150, 400, False), (0, 1 << 20, True)):
                exp = O.opposite_windows(h_read, h_rc, h_loc, h_score, a_len, o_len, e, eo, sch.m_match, sch.m_score_min, sch.text_gap_open(), sch.text_gap_extension(),
                                         policy, mn, mx, ov, -1000, anchor, G)
                got = nvb.opposite_mate_windows(d(h_read, np.int32), d(h_rc, np.uint8), d(h_loc, np.int32), d(h_score, np.int32), best, best_o, sch, anchor, G,
                                                a_read_len=d(a_len, np.int32), o_read_len=d(o_len, np.int32), pe_policy=policy, min_frag_len=mn, max_frag_len=mx,
                                                pe_overlap=ov, score_limit=-1000, max_read_len=250)
                torch.cuda.synchronize()
                for k in exp:
                    g = got[k].cpu().numpy().view(exp[k].dtype)
                    bad = np.nonzero(g != exp[k])[0]
                    assert bad.size == 0, (sch.m_match, policy, mn, mx, ov, k, bad[:5], g[bad[:5]], exp[k][bad[:5]])
                assert 0 < exp["valid"].sum() < n_hits
