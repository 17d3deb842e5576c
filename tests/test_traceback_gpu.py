"""Parity of the HIP banded Gotoh traceback (through the C-ABI) with the CPU oracle: bit-exact score,
sink, source and CIGAR (nvBowtie's io::Cigar encoding, end of the alignment first)."""
import json
import os

import numpy as np
import pytest
import torch

import nvbio_amd as nvb
from oracle import pyoracle as O
from test_banded_gpu import random_pairs, dna

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
KAT = json.load(open(os.path.join(HERE, "golden", "kat.json")))


def to_dev(hs, dev):
    return nvb.PackedStringSet.from_host(hs.words, hs.bits, hs.big_endian, hs.begin, hs.length, device=dev)


def compare(exp, got, tag, rows=None):
    """every field of the first exp["score"].size jobs (rows: a boolean mask of the jobs to compare)"""
    g = {k: v.cpu().numpy() for k, v in got.items()}
    n = exp["score"].size
    r = np.ones(n, bool) if rows is None else rows
    assert (exp["score"][r] == g["score"][:n][r]).all(), tag
    assert (exp["sink"][r] == g["sink"].view(np.uint32)[:n][r]).all(), tag
    bad = np.nonzero((exp["source"] != g["source"].view(np.uint32)[:n]).any(1) & r)[0]
    assert bad.size == 0, (tag, bad[:5], exp["source"][bad[:3]], g["source"].view(np.uint32)[bad[:3]])
    assert (exp["cigar_len"][r] == g["cigar_len"].view(np.uint32)[:n][r]).all(), tag
    gc = g["cigar"].view(np.uint16)[:n]
    stride = gc.shape[1]
    mask = np.arange(stride)[None, :] < np.minimum(exp["cigar_len"], stride)[:, None]
    bad = np.nonzero(((exp["cigar"][:n] != gc) & mask).any(1) & r)[0]
    assert bad.size == 0, (tag, bad[:5], exp["cigar"][bad[0]], gc[bad[0]])


def test_cigar_kats_on_gpu(cuda):
    """the CIGAR literals of the reference's own test (alignment_test.cu:793, :825)"""
    for case in (c for c in KAT["gotoh"] if "cigar" in c):
        p, t = dna(KAT["strings"][case["p"]]), dna(KAT["strings"][case["t"]])
        hp, ht = O.StringSet.from_lists([p], 4, True), O.StringSet.from_lists([t], 2, False)
        aligner = nvb.make_gotoh_aligner(case["type"], nvb.SimpleGotohScheme(*case["scheme"]))
        got = nvb.batch_banded_alignment_traceback(case["band"], aligner, to_dev(hp, cuda), to_dev(ht, cuda), max_pattern_length=len(p))
        torch.cuda.synchronize()
        k = int(got["cigar_len"][0])
        cig = got["cigar"][0, :k].cpu().numpy().view(np.uint16)
        assert "".join("%d%s" % (c >> 2, "MIDS"[c & 3]) for c in cig) == case["cigar"]
        assert int(got["score"][0]) == case["score"]
        assert got["sink"][0].tolist() == case["sink"]


@pytest.mark.parametrize("band", [3, 5, 7, 15, 31])
@pytest.mark.parametrize("ty", [nvb.GLOBAL, nvb.LOCAL, nvb.SEMI_GLOBAL])
def test_random_ragged_pairs(cuda, band, ty):
    rng = np.random.default_rng(7000 + band * 3 + ty)
    pats, txts = random_pairs(rng, 1500, band)
    longest = 0
    for pbits, pbe, tbe in ((4, True, True), (2, False, False)):
        pp = [np.minimum(p, 3) for p in pats] if pbits == 2 else pats
        hp, ht = O.StringSet.from_lists(pp, pbits, pbe), O.StringSet.from_lists(txts + [np.zeros(64, np.uint8)], 2, tbe)
        ht = O.StringSet(ht.words, 2, tbe, ht.begin[:-1], ht.length[:-1])       # defined padding after the last text
        maxM = int(hp.length.max())
        for scheme in ((2, -1, -2, -1), (0, -5, -8, -3)):
            stride = 24 if scheme[0] else 40          # small enough that some CIGARs overflow and are only counted
            exp = O.batch_banded_gotoh_traceback(band, ty, scheme, hp, ht, stride)
            got = nvb.batch_banded_alignment_traceback(band, nvb.make_gotoh_aligner(ty, nvb.SimpleGotohScheme(*scheme)),
                                                       to_dev(hp, cuda), to_dev(ht, cuda), max_pattern_length=maxM, cigar_stride=stride)
            torch.cuda.synchronize()
            compare(exp, got, (band, ty, scheme, pbits))
            longest = max(longest, int(exp["cigar_len"].max()))
    assert longest > 3


@pytest.mark.parametrize("ty", [nvb.LOCAL, nvb.SEMI_GLOBAL])
def test_quality_aware_scheme(cuda, ty):
    """nvBowtie's traceback instantiation: SmithWatermanScoringScheme + read qualities (traceback_inl.h:205-262)"""
    band = 15
    rng = np.random.default_rng(7100 + ty)
    pats, txts = random_pairs(rng, 1500, band)
    hp, ht = O.StringSet.from_lists(pats, 4, True), O.StringSet.from_lists(txts + [np.zeros(64, np.uint8)], 2, True)
    ht = O.StringSet(ht.words, 2, True, ht.begin[:-1], ht.length[:-1])
    total = int(hp.begin[-1] + hp.length[-1])
    quals = rng.integers(0, 60, total + 3, dtype=np.uint8)
    quals[::97] = 255
    for scheme in (nvb.SmithWatermanScoringScheme(), nvb.SmithWatermanScoringScheme.local()):
        st = scheme.struct()
        lut = np.array([st.mismatch[q] for q in range(256)], dtype=np.int32)
        s5 = (st.match, st.pattern_gap_open, st.pattern_gap_ext, st.text_gap_open, st.text_gap_ext)
        exp = O.batch_banded_gotoh_traceback(band, ty, s5, hp, ht, 48, lut, quals)
        got = nvb.batch_banded_alignment_traceback(band, nvb.make_gotoh_aligner(ty, scheme), to_dev(hp, cuda), to_dev(ht, cuda),
                                                   max_pattern_length=int(hp.length.max()), quals=torch.from_numpy(quals).to(cuda), cigar_stride=48)
        torch.cuda.synchronize()
        compare(exp, got, (ty, "qual"))


def test_fixed_length_batch_and_score_agreement(cuda):
    """100 bp reads against band-15 windows (the extension stage's shape): traceback score/sink equal the
    score kernel's, every CIGAR consumes exactly the read (nvBowtie's assert, traceback_inl.h:168-171)."""
    from nvbio_amd import workloads as W
    n, M, band = 200_000, 100, 15
    p, t = W.make_sw_batch(n, M, M + band, device=cuda, seed=5)
    aligner = nvb.make_gotoh_aligner(nvb.LOCAL, nvb.SimpleGotohScheme(2, -1, -2, -1))
    score, sink = nvb.batch_banded_alignment_score(band, aligner, p, t)
    got = nvb.batch_banded_alignment_traceback(band, aligner, p, t, cigar_stride=32)
    torch.cuda.synchronize()
    assert torch.equal(score, got["score"]) and torch.equal(sink, got["sink"])
    cig = got["cigar"].to(torch.int32) & 0xFFFF
    k = torch.arange(cig.shape[1], device=cuda)[None, :] < got["cigar_len"][:, None]
    consumed = (((cig >> 2) * ((cig & 3) != 2)) * k).sum(1)
    assert int(got["cigar_len"].max()) <= 32
    assert bool((consumed == M).all())
    # spot-check a sample against the oracle
    hp, ht = O.StringSet.from_device(p), O.StringSet.from_device(t)
    idx = np.arange(0, n, 997)
    sub = lambda s: O.StringSet(s.words, s.bits, s.big_endian, s.begin[idx], s.length[idx])
    exp = O.batch_banded_gotoh_traceback(band, nvb.LOCAL, (2, -1, -2, -1), sub(hp), sub(ht), 32)
    compare(exp, {k2: v[torch.from_numpy(idx).to(cuda)] for k2, v in got.items()}, "fixed")


def test_refusals(cuda):
    hp, ht = O.StringSet.from_lists([np.zeros(10, np.uint8)], 4, True), O.StringSet.from_lists([np.zeros(30, np.uint8)], 2, True)
    big = nvb.make_gotoh_aligner(nvb.LOCAL, nvb.SimpleGotohScheme(2000, -1000, -2000, -1000))
    with pytest.raises(RuntimeError):          # outside the int16 checkpoint range of the reference: refused, not approximated
        nvb.batch_banded_alignment_traceback(15, big, to_dev(hp, cuda), to_dev(ht, cuda), max_pattern_length=10)
    with pytest.raises(ValueError):
        nvb.BatchedBandedAlignmentTraceback(9)


# ---------------------------------------------------------------------------- full-matrix traceback
@pytest.fixture(params=["wave", "lanes"])
def tb_kernel(request, monkeypatch):
    """Both executions of the full-matrix traceback: one job per wave segment (the default where a job's pattern blocks fit a wave)
    and one job per lane (NVBIO_HIP_TRACEBACK_LANES=1; what longer patterns get)."""
    if request.param == "lanes":
        nvb.set_test_switch("NVBIO_HIP_TRACEBACK_LANES", "1")
    else:
        nvb.set_test_switch("NVBIO_HIP_TRACEBACK_LANES", 0)
    return request.param


def test_full_matrix_cigar_kats_on_gpu(cuda, tb_kernel):
    """alignment_test.cu:788-792: the full-matrix Gotoh CIGARs"""
    p, t = dna(KAT["strings"]["short_p"]), dna(KAT["strings"]["short_t"])
    hp, ht = O.StringSet.from_lists([p], 4, True), O.StringSet.from_lists([t], 2, False)
    for ty, lit in ((nvb.GLOBAL, "1M2D3M1D3M10D"), (nvb.LOCAL, "4M1D3M"), (nvb.SEMI_GLOBAL, "4M1D3M")):
        got = nvb.batch_alignment_traceback(nvb.make_gotoh_aligner(ty, nvb.SimpleGotohScheme(2, -1, -1, -1)), to_dev(hp, cuda), to_dev(ht, cuda), len(p), len(t))
        torch.cuda.synchronize()
        k = int(got["cigar_len"][0])
        cig = got["cigar"][0, :k].cpu().numpy().view(np.uint16)
        assert "".join("%d%s" % (c >> 2, "MIDS"[c & 3]) for c in cig) == lit


@pytest.mark.parametrize("ty", [nvb.GLOBAL, nvb.LOCAL, nvb.SEMI_GLOBAL])
def test_full_matrix_traceback(cuda, ty, tb_kernel):
    """random pairs incl. the opposite-mate shape (150 bp in a 650-bp window): score, sink (pattern-blocking order),
    source and CIGAR equal to the oracle's restatement of alignment_traceback"""
    rng = np.random.default_rng(8100 + ty)
    pats, txts = [], []
    for i in range(600):
        if i % 4 == 0:
            M, N = 150, 650
        else:
            M, N = int(rng.integers(1, 140)), int(rng.integers(1, 300))
        t = rng.integers(0, 4, N).astype(np.uint8)
        off = int(rng.integers(0, max(1, N - M)))
        p = np.resize(t[off:off + M] if N > off + 3 else rng.integers(0, 4, M).astype(np.uint8), M).copy()
        mut = rng.random(M) < 0.06
        p[mut] = rng.integers(0, 5, int(mut.sum()))
        if M > 30 and i % 3 == 0:
            cut = int(rng.integers(5, M - 5))
            p = np.concatenate([p[:cut], p[cut + 3:], rng.integers(0, 4, 3).astype(np.uint8)])
        pats.append(p); txts.append(t)
    hp, ht = O.StringSet.from_lists(pats, 4, True), O.StringSet.from_lists(txts, 2, True)
    for scheme in ((2, -1, -2, -1), (0, -6, -8, -3), (2, -6, -8, -3)):
        stride = 40
        exp = O.batch_gotoh_traceback(ty, scheme, hp, ht, stride)
        got = nvb.batch_alignment_traceback(nvb.make_gotoh_aligner(ty, nvb.SimpleGotohScheme(*scheme)), to_dev(hp, cuda), to_dev(ht, cuda), 150, 650, cigar_stride=stride)
        torch.cuda.synchronize()
        compare(exp, got, (ty, scheme))
        # the score pass of the same aligner (pattern blocking) agrees
        gs, gk, _ = nvb.batch_alignment_score(nvb.make_gotoh_aligner(ty, nvb.SimpleGotohScheme(*scheme), nvb.PATTERN_BLOCKING), to_dev(hp, cuda), to_dev(ht, cuda), 150, 650)
        assert torch.equal(gs, got["score"]) and torch.equal(gk, got["sink"])


# ---------------------------------------------------------------------------- SW / edit-distance tracebacks
@pytest.mark.parametrize("ty", [nvb.GLOBAL, nvb.LOCAL, nvb.SEMI_GLOBAL])
def test_sw_and_ed_tracebacks(cuda, ty, tb_kernel):
    """SmithWatermanAligner / EditDistanceAligner tracebacks, banded (incl. the reference's LOCAL walk that never stops at a zero
    cell) and full matrix (sink of the 16-column pattern-blocking pass), vs the oracle's restatement of sw_banded_inl.h / sw_inl.h;
    and the reference's three full-matrix SW CIGAR literals (alignment_test.cu:776-780)."""
    rng = np.random.default_rng(8300 + ty)
    pats, txts = random_pairs(rng, 800, 15)
    hp, ht = O.StringSet.from_lists(pats, 4, True), O.StringSet.from_lists(txts + [np.zeros(64, np.uint8)], 2, True)
    ht = O.StringSet(ht.words, 2, True, ht.begin[:-1], ht.length[:-1])
    maxM = int(hp.length.max())
    for scheme in ((2, -1, -1, -1), (0, -1, -1, -1)):
        al = nvb.make_edit_distance_aligner(ty) if scheme[0] == 0 else nvb.make_smith_waterman_aligner(ty, nvb.SimpleSmithWatermanScheme(*scheme))
        for band in (7, 15, 31):
            exp = O.batch_sw_traceback(band, ty, scheme, hp, ht, 40)
            got = nvb.batch_banded_alignment_traceback(band, al, to_dev(hp, cuda), to_dev(ht, cuda), max_pattern_length=maxM, cigar_stride=40)
            torch.cuda.synchronize()
            compare(exp, got, (ty, scheme, band))
    fp, ft = [], []
    for i in range(400):
        M, N = int(rng.integers(1, 120)), int(rng.integers(1, 260))
        t = rng.integers(0, 4, N).astype(np.uint8)
        p = np.resize(t[int(rng.integers(0, N)):], M).copy()
        mut = rng.random(M) < 0.08
        p[mut] = rng.integers(0, 4, int(mut.sum()))
        fp.append(p); ft.append(t)
    hp, ht = O.StringSet.from_lists(fp, 4, True), O.StringSet.from_lists(ft, 2, False)
    for scheme in ((2, -1, -1, -1), (0, -1, -1, -1)):
        al = nvb.make_edit_distance_aligner(ty) if scheme[0] == 0 else nvb.make_smith_waterman_aligner(ty, nvb.SimpleSmithWatermanScheme(*scheme))
        exp = O.batch_sw_traceback(0, ty, scheme, hp, ht, 48)
        got = nvb.batch_alignment_traceback(al, to_dev(hp, cuda), to_dev(ht, cuda), 120, 260, cigar_stride=48)
        torch.cuda.synchronize()
        compare(exp, got, (ty, scheme, "full"))
    p, t = dna(KAT["strings"]["short_p"]), dna(KAT["strings"]["short_t"])
    h1, h2 = O.StringSet.from_lists([p], 4, True), O.StringSet.from_lists([t], 2, False)
    got = nvb.batch_alignment_traceback(nvb.make_smith_waterman_aligner(ty, nvb.SimpleSmithWatermanScheme(2, -1, -1, -1)), to_dev(h1, cuda), to_dev(h2, cuda), 7, 20)
    cig = got["cigar"][0, :int(got["cigar_len"][0])].cpu().numpy().view(np.uint16)
    assert "".join("%d%s" % (c >> 2, "MIDS"[c & 3]) for c in cig) == {nvb.GLOBAL: "1M2D3M1D3M10D", nvb.LOCAL: "4M1D3M", nvb.SEMI_GLOBAL: "4M1D3M"}[ty]


@pytest.mark.parametrize("ty", [nvb.LOCAL, nvb.SEMI_GLOBAL, nvb.GLOBAL])
def test_full_matrix_traceback_quality_aware(cuda, ty, tb_kernel):
    """nvBowtie's opposite-mate traceback: quality-aware scheme over the full matrix (asymmetric gap costs included)"""
    rng = np.random.default_rng(8500 + ty)
    pats, txts = [], []
    for i in range(300):
        M, N = int(rng.integers(1, 150)), int(rng.integers(1, 400))
        t = rng.integers(0, 4, N).astype(np.uint8)
        p = np.resize(t[int(rng.integers(0, N)):], M).copy()
        mut = rng.random(M) < 0.06
        p[mut] = rng.integers(0, 5, int(mut.sum()))
        pats.append(p); txts.append(t)
    hp, ht = O.StringSet.from_lists(pats, 4, True), O.StringSet.from_lists(txts, 2, True)
    quals = rng.integers(0, 60, int(hp.begin[-1] + hp.length[-1]) + 3, dtype=np.uint8)
    for scheme in (nvb.SmithWatermanScoringScheme.local(),
                   nvb.SmithWatermanScoringScheme(match=1, mmp_min=1, mmp_max=9, read_gap_const=4, read_gap_coeff=2, ref_gap_const=7, ref_gap_coeff=1)):
        st = scheme.struct()
        lut = np.array([st.mismatch[q] for q in range(256)], dtype=np.int32)
        s5 = (st.match, st.pattern_gap_open, st.pattern_gap_ext, st.text_gap_open, st.text_gap_ext)
        exp = O.batch_gotoh_traceback(ty, s5, hp, ht, 48, lut, quals)
        got = nvb.batch_alignment_traceback(nvb.make_gotoh_aligner(ty, scheme), to_dev(hp, cuda), to_dev(ht, cuda), 150, 400, cigar_stride=48,
                                            quals=torch.from_numpy(quals).to(cuda))
        torch.cuda.synchronize()
        compare(exp, got, (ty, st.match))


@pytest.mark.parametrize("band", [3, 5, 7, 15, 31])
@pytest.mark.parametrize("ty", [nvb.GLOBAL, nvb.LOCAL, nvb.SEMI_GLOBAL])
def test_ungapped_alignments_every_band_offset(cuda, band, ty):
    """The bulk case of real reads -- no gap, a few substitutions -- at every band offset, with windows that end inside the band
    (where the reference's text cache turns out-of-range symbols into 3 for bands other than 3,5,7,15), N's in the reads, leading /
    trailing junk (LOCAL clipping) and a minority of gapped reads mixed in: the diagonal fast path and the full kernel must agree with
    the oracle job by job."""
    rng = np.random.default_rng(7300 + band * 3 + ty)
    pats, txts = [], []
    for i in range(900):
        M = int(rng.integers(4, 120))
        off = int(rng.integers(0, band))
        tail = [band + 3, band - 1 - off, 0, int(rng.integers(0, band))][i % 4]          # text beyond the diagonal's end: plenty / flush with the band / none / random
        t = rng.integers(0, 4, off + M + tail, dtype=np.uint8)
        p = t[off:off + M].copy()
        for j in rng.integers(0, M, [0, 0, 1, 2, 4][i % 5]):
            p[j] = (p[j] + 1 + rng.integers(0, 3)) & 3
        if i % 11 == 0:
            p[int(rng.integers(0, M))] = 4
        if i % 7 == 0 and M > 20:                                                        # junk at both ends: LOCAL clips it
            p[:3] = rng.integers(0, 4, 3); p[-4:] = rng.integers(0, 4, 4)
        if i % 13 == 0 and M > 12:                                                       # a deletion in the read: not ungapped
            d = int(rng.integers(4, M - 4)); p = np.concatenate([p[:d], p[d + 1:]])
        if i % 17 == 0:
            p[:] = 3                                                                     # poly-T reads meet the cache quirk's 3s
        pats.append(p); txts.append(t)
    hp, ht = O.StringSet.from_lists(pats, 4, True), O.StringSet.from_lists(txts + [np.full(64, 3, np.uint8)], 2, True)
    ht = O.StringSet(ht.words, 2, True, ht.begin[:-1], ht.length[:-1])                   # what follows the last text is defined (all T)
    maxM = int(hp.length.max())
    ungapped = 0
    for scheme in ((2, -3, -5, -3), (0, -6, -8, -3), (1, -1, -1, -1)):
        exp = O.batch_banded_gotoh_traceback(band, ty, scheme, hp, ht, 40)
        got = nvb.batch_banded_alignment_traceback(band, nvb.make_gotoh_aligner(ty, nvb.SimpleGotohScheme(*scheme)), to_dev(hp, cuda), to_dev(ht, cuda),
                                                   max_pattern_length=maxM, cigar_stride=40)
        torch.cuda.synchronize()
        compare(exp, got, (band, ty, scheme))
        ok = exp["cigar_len"] > 0
        ungapped += int((((exp["cigar"] & 3) != 1) & ((exp["cigar"] & 3) != 2) | (np.arange(40)[None, :] >= exp["cigar_len"][:, None])).all(1)[ok].sum())
    assert ungapped > (10 if ty == nvb.GLOBAL else 900)
    if band == 31 or band == 15:
        total = int(hp.begin[-1] + hp.length[-1])
        quals = rng.integers(0, 50, total + 3, dtype=np.uint8)
        for sch in (nvb.SmithWatermanScoringScheme(), nvb.SmithWatermanScoringScheme.local()):
            st = sch.struct()
            lut = np.array([st.mismatch[q] for q in range(256)], dtype=np.int32)
            s5 = (st.match, st.pattern_gap_open, st.pattern_gap_ext, st.text_gap_open, st.text_gap_ext)
            exp = O.batch_banded_gotoh_traceback(band, ty, s5, hp, ht, 40, lut, quals)
            got = nvb.batch_banded_alignment_traceback(band, nvb.make_gotoh_aligner(ty, sch), to_dev(hp, cuda), to_dev(ht, cuda), max_pattern_length=maxM,
                                                       quals=torch.from_numpy(quals).to(cuda), cigar_stride=40)
            torch.cuda.synchronize()
            compare(exp, got, (band, ty, "qual"))


def test_full_matrix_traceback_on_tie_heavy_batches(cuda, tb_kernel):
    """Binary-alphabet texts (many equal-scoring placements): the traceback -- whose ungapped jobs take their sink from the pattern-blocking
    score kernel -- still equals the oracle's job by job, and score kernel and traceback agree on every sink."""
    rng = np.random.default_rng(8700)
    pats, txts = [], []
    for i in range(1200):
        M = int(rng.integers(5, 150)); N = int(rng.integers(M, 500))
        t = rng.integers(0, 2 if i % 2 else 4, N, dtype=np.uint8)
        o = int(rng.integers(0, N - M + 1))
        p = t[o:o + M].copy()
        for j in rng.integers(0, M, int(rng.integers(0, 5))):
            p[j] = (p[j] + 1) & 3
        if i % 6 == 0:
            p = rng.integers(0, 2, M, dtype=np.uint8)
        pats.append(p); txts.append(t)
    hp, ht = O.StringSet.from_lists(pats, 4, True), O.StringSet.from_lists(txts + [np.zeros(64, np.uint8)], 2, True)
    ht = O.StringSet(ht.words, 2, True, ht.begin[:-1], ht.length[:-1])
    dp, dt = to_dev(hp, cuda), to_dev(ht, cuda)
    for ty in (nvb.LOCAL, nvb.SEMI_GLOBAL):
        for scheme in ((2, -6, -8, -3), (1, -1, -2, -1), (0, -6, -8, -3)):
            al = nvb.make_gotoh_aligner(ty, nvb.SimpleGotohScheme(*scheme), nvb.PATTERN_BLOCKING)
            exp = O.batch_gotoh_traceback(ty, scheme, hp, ht, 48)
            got = nvb.batch_alignment_traceback(al, dp, dt, 150, 500, cigar_stride=48)
            s, k, _ = nvb.batch_alignment_score(al, dp, dt, 150, 500)
            torch.cuda.synchronize()
            compare(exp, got, (ty, scheme, "ties"))
            assert torch.equal(s, got["score"]) and torch.equal(k, got["sink"])


def test_full_matrix_traceback_long_left_context(cuda, tb_kernel):
    """The opposite-mate shape at its extreme: the alignment ends at the end of a text several times the read (the queued, gapped jobs
    drop the text columns no alignment with their score can reach).  Gapped reads, tandem-repeat texts (equal-scoring paths that
    slide along the repeat and long gap runs are common there), cheap and expensive gaps, both types, with and without qualities."""
    rng = np.random.default_rng(8900)
    pats, txts = [], []
    for i in range(1500):
        M = int(rng.integers(30, 151)); N = int(rng.integers(max(M + 40, 200), 651))
        if i % 3 == 0:
            unit = rng.integers(0, 4, int(rng.integers(1, 7)), dtype=np.uint8)
            t = np.resize(unit, N).copy()
            t[rng.integers(0, N, N // 25)] = rng.integers(0, 4, N // 25)
        else:
            t = rng.integers(0, 4, N, dtype=np.uint8)
        L = M + int(rng.integers(0, 4))
        p = t[N - L - int(rng.integers(0, 3)):][:L].copy()
        for j in rng.integers(0, p.size, int(rng.integers(0, 6))):
            p[j] = (p[j] + 1 + rng.integers(0, 3)) & 3
        k = i % 5
        if k == 1:
            c = int(rng.integers(5, p.size - 5)); p = np.delete(p, slice(c, c + int(rng.integers(1, 9))))
        elif k == 2:
            c = int(rng.integers(5, p.size - 5)); p = np.insert(p, c, rng.integers(0, 4, int(rng.integers(1, 6))))
        elif k == 3:
            c, d = sorted(rng.integers(5, p.size - 5, 2)); p = np.insert(np.delete(p, slice(c, c + 2)), max(d - 2, 1), rng.integers(0, 4, 3))
        pats.append(p[:150].astype(np.uint8)); txts.append(t)
    hp, ht = O.StringSet.from_lists(pats, 4, True), O.StringSet.from_lists(txts + [np.zeros(64, np.uint8)], 2, True)
    ht = O.StringSet(ht.words, 2, True, ht.begin[:-1], ht.length[:-1])
    dp, dt = to_dev(hp, cuda), to_dev(ht, cuda)
    gapped = 0
    for ty in (nvb.LOCAL, nvb.SEMI_GLOBAL):
        for scheme in ((2, -6, -8, -3), (1, -2, -1, -1), (0, -6, -5, -3), (3, -1, -4, -1)):
            exp = O.batch_gotoh_traceback(ty, scheme, hp, ht, 64)
            got = nvb.batch_alignment_traceback(nvb.make_gotoh_aligner(ty, nvb.SimpleGotohScheme(*scheme), nvb.PATTERN_BLOCKING), dp, dt, 150, 650, cigar_stride=64)
            torch.cuda.synchronize()
            compare(exp, got, (ty, scheme, "left context"))
            gapped += int(((exp["cigar"][:, :8] & 3) % 3 != 0).any(axis=1).sum())
    assert gapped > 2000
    quals = rng.integers(0, 45, int(hp.begin[-1] + hp.length[-1]) + 3, dtype=np.uint8)
    for ty, scheme in ((nvb.LOCAL, nvb.SmithWatermanScoringScheme.local()), (nvb.SEMI_GLOBAL, nvb.SmithWatermanScoringScheme())):
        st = scheme.struct()
        lut = np.array([st.mismatch[q] for q in range(256)], dtype=np.int32)
        s5 = (st.match, st.pattern_gap_open, st.pattern_gap_ext, st.text_gap_open, st.text_gap_ext)
        exp = O.batch_gotoh_traceback(ty, s5, hp, ht, 64, lut, quals)
        got = nvb.batch_alignment_traceback(nvb.make_gotoh_aligner(ty, scheme), dp, dt, 150, 650, cigar_stride=64, quals=torch.from_numpy(quals).to(cuda))
        torch.cuda.synchronize()
        compare(exp, got, (ty, "qual", "left context"))


@pytest.mark.parametrize("ty", [nvb.GLOBAL, nvb.LOCAL, nvb.SEMI_GLOBAL])
def test_full_matrix_traceback_block_counts(cuda, ty):
    """Patterns from 1 symbol to 700: every lanes-per-job width of the wave kernel (1 ... 64 blocks, several jobs per wave and one),
    and past 512 symbols (Gotoh) the one-job-per-lane kernel; Gotoh (8-symbol blocks) and SW (16-symbol blocks)."""
    rng = np.random.default_rng(9100 + ty)
    for maxM, n in ((7, 200), (16, 200), (33, 150), (100, 120), (257, 60), (512, 40), (700, 24)):
        pats, txts = [], []
        for i in range(n):
            M = maxM if i % 3 == 0 else int(rng.integers(1, maxM + 1))
            N = int(rng.integers(max(1, M // 2), M + 120))
            t = rng.integers(0, 4, N).astype(np.uint8)
            p = np.resize(t[int(rng.integers(0, max(1, N - M))):], M).copy()
            mut = rng.random(M) < 0.05
            p[mut] = rng.integers(0, 4, int(mut.sum()))
            if M > 20 and i % 2:
                c = int(rng.integers(3, M - 8)); p = np.concatenate([p[:c], p[c + 4:], rng.integers(0, 4, 4).astype(np.uint8)])
            pats.append(p.astype(np.uint8)); txts.append(t)
        hp, ht = O.StringSet.from_lists(pats, 4, True), O.StringSet.from_lists(txts + [np.zeros(64, np.uint8)], 2, True)
        ht = O.StringSet(ht.words, 2, True, ht.begin[:-1], ht.length[:-1])
        dp, dt = to_dev(hp, cuda), to_dev(ht, cuda)
        maxN = int(ht.length.max())
        stride = 96
        exp = O.batch_gotoh_traceback(ty, (2, -3, -5, -2), hp, ht, stride)
        got = nvb.batch_alignment_traceback(nvb.make_gotoh_aligner(ty, nvb.SimpleGotohScheme(2, -3, -5, -2)), dp, dt, maxM, maxN, cigar_stride=stride)
        torch.cuda.synchronize()
        compare(exp, got, (ty, maxM, "gotoh"))
        exp = O.batch_sw_traceback(0, ty, (2, -1, -1, -1), hp, ht, stride)
        got = nvb.batch_alignment_traceback(nvb.make_smith_waterman_aligner(ty, nvb.SimpleSmithWatermanScheme(2, -1, -1, -1)), dp, dt, maxM, maxN, cigar_stride=stride)
        torch.cuda.synchronize()
        compare(exp, got, (ty, maxM, "sw"))


@pytest.mark.parametrize("ty", [nvb.LOCAL, nvb.SEMI_GLOBAL])
def test_full_matrix_traceback_known_score_windows(cuda, ty, tb_kernel):
    """nvBowtie's opposite-mate traceback windows: [window begin, sink of the scoring pass), the score known.  The _known_score forms
    (which drop the text rows no alignment of that score can reach before any DP runs) must return what the plain traceback -- and the
    oracle -- return over the whole window: gapped reads at the far end of windows several reads long, repeats, with and without
    qualities."""
    rng = np.random.default_rng(9300 + ty)
    pats, txts = [], []
    for i in range(1200):
        M = int(rng.integers(30, 151)); N = int(rng.integers(max(M + 40, 200), 651))
        if i % 3 == 0:
            unit = rng.integers(0, 4, int(rng.integers(1, 7)), dtype=np.uint8)
            t = np.resize(unit, N).copy()
            t[rng.integers(0, N, N // 25)] = rng.integers(0, 4, N // 25)
        else:
            t = rng.integers(0, 4, N, dtype=np.uint8)
        o = int(rng.integers(0, N - M))
        p = t[o:o + M + 3].copy()
        for j in rng.integers(0, p.size, int(rng.integers(0, 6))):
            p[j] = (p[j] + 1 + rng.integers(0, 3)) & 3
        k = i % 4
        if k == 1:
            c = int(rng.integers(5, p.size - 5)); p = np.delete(p, slice(c, c + int(rng.integers(1, 9))))
        elif k == 2:
            c = int(rng.integers(5, p.size - 5)); p = np.insert(p, c, rng.integers(0, 4, int(rng.integers(1, 6))))
        pats.append(p[:150].astype(np.uint8)); txts.append(t)
    hp, ht = O.StringSet.from_lists(pats, 4, True), O.StringSet.from_lists(txts + [np.zeros(64, np.uint8)], 2, True)
    ht = O.StringSet(ht.words, 2, True, ht.begin[:-1], ht.length[:-1])
    quals = rng.integers(0, 60, int(hp.begin[-1] + hp.length[-1]) + 3, dtype=np.uint8)
    dq = torch.from_numpy(quals).to(cuda)
    checked = cropped = 0
    for scheme in ((2, -6, -8, -3), (1, -2, -1, -1), "qual"):
        if scheme == "qual":
            sch = nvb.SmithWatermanScoringScheme.local() if ty == nvb.LOCAL else nvb.SmithWatermanScoringScheme()
            st = sch.struct()
            lut = np.array([st.mismatch[q] for q in range(256)], dtype=np.int32)
            okw = dict(lut=lut, quals=quals, s=(st.match, st.pattern_gap_open, st.pattern_gap_ext, st.text_gap_open, st.text_gap_ext))
            al, kw = nvb.make_gotoh_aligner(ty, sch), dict(quals=dq)
            first = O.batch_gotoh_traceback(ty, okw["s"], hp, ht, 64, lut, quals)
        else:
            al, kw = nvb.make_gotoh_aligner(ty, nvb.SimpleGotohScheme(*scheme)), {}
            first = O.batch_gotoh_traceback(ty, scheme, hp, ht, 64)
        # the scoring pass's verdict: windows end at its sinks
        ok = first["sink"][:, 0].view(np.int32) > 0
        keep = np.nonzero(ok)[0]
        wl = first["sink"][keep, 0].astype(np.uint32)
        sub_p = O.StringSet(hp.words, 4, True, hp.begin[keep], hp.length[keep])
        sub_t = O.StringSet(ht.words, 2, True, ht.begin[keep], wl)
        exp = O.batch_gotoh_traceback(ty, okw["s"], sub_p, sub_t, 64, lut, quals) if scheme == "qual" else O.batch_gotoh_traceback(ty, scheme, sub_p, sub_t, 64)
        assert (exp["score"] == first["score"][keep]).all() and (exp["sink"][:, 0] == wl).all()        # the premise
        known = torch.from_numpy(first["score"][keep].astype(np.int32)).to(cuda)
        got = nvb.batch_alignment_traceback(al, to_dev(sub_p, cuda), to_dev(sub_t, cuda), 150, 650, cigar_stride=64, known_score=known, **kw)
        torch.cuda.synchronize()
        compare(exp, got, (ty, scheme, "known score"))
        # a caller whose premise is WRONG still gets the plain traceback: scores announced too high (the window is cut too short for
        # the real alignment's gaps) and windows that do not end at their alignment's sink are found by the check on the cropped DP and
        # traced again over the whole window (full_traceback.hip, verify_known_kernel)
        before = int(nvb.lib().nvbio_hip_known_score_redone())
        wrong = first["score"][keep].astype(np.int32).copy()
        wrong[::3] += 40; wrong[1::3] -= 7
        got = nvb.batch_alignment_traceback(al, to_dev(sub_p, cuda), to_dev(sub_t, cuda), 150, 650, cigar_stride=64,
                                            known_score=torch.from_numpy(wrong).to(cuda), **kw)
        torch.cuda.synchronize()
        compare(exp, got, (ty, scheme, "wrong known score"))
        assert int(nvb.lib().nvbio_hip_known_score_redone()) > before
        longer = O.StringSet(ht.words, 2, True, ht.begin[keep], np.minimum(wl + 9, ht.length[keep]).astype(np.uint32))
        exp_l = O.batch_gotoh_traceback(ty, okw["s"], sub_p, longer, 64, lut, quals) if scheme == "qual" else O.batch_gotoh_traceback(ty, scheme, sub_p, longer, 64)
        got = nvb.batch_alignment_traceback(al, to_dev(sub_p, cuda), to_dev(longer, cuda), 150, 650, cigar_stride=64, known_score=known, **kw)
        torch.cuda.synchronize()
        # (here the premise is false AND the score is right: the check on the cropped DP cannot tell a second alignment of the same score
        # that happens to end at the window's last row from the one the whole-window order picks -- cropping shifts the column blocks the
        # LOCAL tie order is made of.  Such a tie is allowed, nothing else: same score, the other alignment ends at the last text row.)
        g = {k: v.cpu().numpy() for k, v in got.items()}
        n_l = exp_l["score"].size
        assert (exp_l["score"] == g["score"][:n_l]).all()
        g_sink = g["sink"].view(np.uint32)[:n_l]
        tie = (exp_l["sink"] != g_sink).any(1)
        assert int(tie.sum()) <= max(2, n_l // 200), int(tie.sum())
        assert (g_sink[tie, 0] == longer.length[tie]).all()
        compare(exp_l, got, (ty, scheme, "window past the sink"), rows=~tie)
        checked += keep.size
        cropped += int((wl > sub_p.length + 40).sum())
    assert checked > 2500 and cropped > 1000
