"""Pins the CPU oracle (oracle/nvbio_oracle.c) to the known answers the reference's own
tests hold for this path, and to the reference tests' own property checks.  CPU only."""
import json
import os

import numpy as np
import pytest

from oracle import pyoracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
KAT = json.load(open(os.path.join(HERE, "golden", "kat.json")))


def dna(s):
    return np.array(["ACGT".index(c) for c in s], dtype=np.uint8)


def rescore_cigar(cigar, pattern, text, text_start, scheme):
    """What the reference's TestBacktracker::score does (alignment_test_utils.h:628-): walk the
    alignment and apply the Gotoh scheme (first gap symbol gap_open, further ones gap_ext)."""
    match, mismatch, gap_open, gap_ext = scheme
    import re
    i, j, score = 0, text_start, 0
    # the test's backtracker pushes ops end-to-start and scores them reversed
    # (alignment_test_utils.h:656-658), so its RLE literal reads right-to-left
    for cnt, op in reversed(re.findall(r"(\d+)([MID])", cigar)):
        cnt = int(cnt)
        if op == "M":
            for _ in range(cnt):
                score += match if pattern[i] == text[j] else mismatch
                i += 1; j += 1
        elif op == "D":      # consumes text
            score += gap_open + (cnt - 1) * gap_ext
            j += cnt
        else:                # I: consumes pattern
            score += gap_open + (cnt - 1) * gap_ext
            i += cnt
    return score, i, j


@pytest.mark.parametrize("case", KAT["gotoh"], ids=lambda c: "%s-b%d-t%d" % (c["p"], c["band"], c["type"]))
def test_gotoh_kat(case):
    p, t = dna(KAT["strings"][case["p"]]), dna(KAT["strings"][case["t"]])
    for pb, tb in ((8, 8), (4, 2), (2, 2)):
        for be in (False, True):
            ok, score, sx, sy = O.banded_gotoh_score(case["band"], case["type"], case["scheme"], p, t,
                                                     pat_bits=pb, txt_bits=tb, pat_be=be, txt_be=not be)
            assert ok
            assert (score, [sx, sy]) == (case["score"], case["sink"])
    # the reference test's differential oracle (alignment_test.cu:310-326)
    if len(t) >= len(p) + case["band"] - 1:
        assert O.ref_banded_sw(case["band"], case["type"], case["scheme"], p, t) == case["score"]
    if "cigar" in case:
        # the CIGAR literal held by the reference test, re-scored as the test does
        text_used = sum(int(n) for n, op in __import__("re").findall(r"(\d+)([MID])", case["cigar"]) if op in "MD")
        start = case["sink"][0] - text_used
        s, i, j = rescore_cigar(case["cigar"], p, t, start, case["scheme"])
        assert (i, j) == (len(p), case["sink"][0])
        assert s == case["score"]


@pytest.mark.parametrize("case", [c for c in KAT["gotoh"] if "cigar" in c], ids=lambda c: "%s-b%d" % (c["cigar"], c["band"]))
def test_traceback_cigar_kat(case):
    """banded_alignment_traceback against the CIGAR literals of the reference's own test
    (alignment_test.cu:793, :825; rle(backtracker.aln) at :280)."""
    p, t = dna(KAT["strings"][case["p"]]), dna(KAT["strings"][case["t"]])
    for pb, be in ((4, True), (2, False)):
        hp = O.StringSet.from_lists([p], pb, be)
        ht = O.StringSet.from_lists([t], 2, not be)
        r = O.banded_gotoh_traceback(case["band"], case["type"], case["scheme"], hp, ht)
        assert O.cigar_rle(r["ops"]) == case["cigar"]
        assert r["score"] == case["score"] and list(r["sink"]) == case["sink"]
        assert r["source"][1] == 0 and r["clip_end"] == 0 and r["clip_begin"] == 0


def _first(ss):
    """the first string of a StringSet, over the same words"""
    return O.StringSet(ss.words, ss.bits, ss.big_endian, ss.begin[:1], ss.length[:1])


def walk_alignment(r, pattern, text, scheme, mm_lut=None, quals=None):
    """TestBacktracker::score generalised: replay the ops (stored end first) from the source cell."""
    match, mismatch, go, ge = scheme[:4]
    i, j = r["source"][1], r["source"][0]
    score, prev = 0, None
    for op in r["ops"][::-1]:
        if op == 0:
            mm = mismatch if mm_lut is None else int(mm_lut[quals[i]])
            score += match if pattern[i] == text[j] else mm
            i += 1; j += 1
        elif op == 2:
            score += ge if prev == 2 else go
            j += 1
        else:
            score += ge if prev == 1 else go
            i += 1
        prev = op
    return score, i, j


@pytest.mark.parametrize("band", [3, 5, 7, 15, 31])
@pytest.mark.parametrize("aln_type", [0, 1, 2])
def test_traceback_properties(band, aln_type):
    """What the reference test asserts of every traceback (alignment_test.cu:277-355): the replayed
    alignment scores exactly the banded score, starts at `source` and ends at `sink`."""
    rng = np.random.default_rng(100 * band + aln_type)
    scheme = (2, -1, -2, -1)
    for it in range(60):
        M = int(rng.integers(1, 60))
        N = M + band - 1 + int(rng.integers(0, 4)) if it % 3 else M + int(rng.integers(0, band))
        t = rng.integers(0, 4, N).astype(np.uint8)
        p = t[band // 2: band // 2 + M].copy() if N >= band // 2 + M else rng.integers(0, 4, M).astype(np.uint8)
        p = p[:M] if len(p) == M else rng.integers(0, 4, M).astype(np.uint8)
        for _ in range(int(rng.integers(0, 4))):       # a few substitutions / indels
            k = int(rng.integers(0, M))
            kind = rng.integers(0, 3)
            if kind == 0:
                p[k] = rng.integers(0, 4)
            elif kind == 1 and M > 2:
                p = np.concatenate([p[:k], p[k + 1:], rng.integers(0, 4, 1).astype(np.uint8)])
            else:
                p = np.concatenate([p[:k], rng.integers(0, 4, 1).astype(np.uint8), p[k:]])[:M]
        hp = O.StringSet.from_lists([p], 4, True)
        ht = O.StringSet.from_lists([t, np.zeros(64, np.uint8)], 2, True)     # a second string = defined padding
        r = O.banded_gotoh_traceback(band, aln_type, scheme, hp, _first(ht))
        sc_b, sk_b = O.batch_banded_gotoh_score(band, aln_type, scheme, hp, _first(ht))
        assert (r["score"], r["sink"]) == (int(sc_b[0]), (int(sk_b[0, 0]), int(sk_b[0, 1])))
        score, (sx, sy) = r["score"], r["sink"]
        if band == 31:
            continue          # the 2-bit text cache of this band (see test_band31_cache_truncation_quirk) breaks replay near the text end
        # what the DP saw past the end of the text: the band's preload reads the stream unchecked
        # (gotoh_banded_inl.h:437-441; zero padding here), later fetches see 255 (:580-582)
        seen = np.concatenate([t, np.zeros(max(0, band - 1 - N), np.uint8), np.full(band, 255, np.uint8)])
        s, i, j = walk_alignment(r, p, seen, scheme)
        assert (j, i) == r["sink"]
        if aln_type == 0 and r["source"][0] > 0:
            # GLOBAL: row zero of the band is initialised with the cost of the leading text gap
            # (gotoh_banded_inl.h:46-77); the walk ends at row 0 without pushing those deletions
            s += scheme[2] + (r["source"][0] - 1) * scheme[3]
        assert s == score
        assert r["clip_end"] == M - sy and r["clip_begin"] == r["source"][1]
        if aln_type != 1:
            assert r["source"][1] == 0 and sy == M
        # the packed CIGAR consumes exactly the pattern (nvBowtie's assert, traceback_inl.h:168-171)
        cig = r["cigar"]
        assert sum(int(c) >> 2 for c in cig if (int(c) & 3) != 2) == M


def test_text_shorter_than_pattern():
    exp = KAT["text_shorter_than_pattern"]
    p, t = dna(KAT["strings"]["real_p"])[:100], dna(KAT["strings"]["real_t"])[:50]
    for ty in (0, 1, 2):
        ok, score, sx, sy = O.banded_gotoh_score(15, ty, (2, -1, -2, -1), p, t)
        assert ok == exp["returns"] and score == exp["score"] and [sx, sy] == exp["sink"]


def test_banded_edit_distance_kats():
    grp = KAT["banded_edit_distance_band5_semi_global"]
    for c in grp["cases"]:
        p = np.frombuffer(c["pattern"].encode(), dtype=np.uint8)
        t = np.frombuffer(c["text"].encode(), dtype=np.uint8)
        ok, score, _, _ = O.banded_gotoh_score(5, O.SEMI_GLOBAL, grp["scheme"], p, t)
        assert ok and score == c["score"], c
        # ... and on the restatement of the code path the reference test really runs: the banded
        # Smith-Waterman recurrence with EditDistanceSWScheme (ed_banded_inl.h -> sw_banded_inl.h)
        hp, ht = O.StringSet.from_lists([p], 8, False), O.StringSet.from_lists([t], 8, False)
        s2, _ = O.batch_sw_score(5, O.SEMI_GLOBAL, (0, -1, -1, -1), hp, ht)
        assert int(s2[0]) == c["score"], c


@pytest.mark.parametrize("band", [3, 5, 7, 15, 31])
@pytest.mark.parametrize("aln_type", [O.GLOBAL, O.LOCAL, O.SEMI_GLOBAL])
def test_differential_vs_ref_banded_sw(band, aln_type):
    """banded_alignment_score == ref_banded_sw on random pairs, as alignment_test.cu:310-326
    asserts (valid where the text covers the whole band: N >= M + BAND - 1)."""
    rng = np.random.default_rng(band * 10 + aln_type)
    for scheme in [(2, -1, -2, -1), (0, -5, -8, -3), (2, -1, -1, -1), (1, -3, -5, -2)]:
        for _ in range(40):
            M = int(rng.integers(1, 120))
            N = M + band - 1 + int(rng.integers(0, 5))
            t = rng.integers(0, 4, N, dtype=np.uint8)
            p = t[band // 2: band // 2 + M].copy()
            mut = rng.random(M) < 0.1
            p[mut] = rng.integers(0, 4, int(mut.sum()), dtype=np.uint8)
            ok, score, sx, sy = O.banded_gotoh_score(band, aln_type, scheme, p, t, pat_bits=4, txt_bits=2)
            assert ok
            assert score == O.ref_banded_sw(band, aln_type, scheme, p, t)


def test_band31_cache_truncation_quirk():
    """Reference_cache<31> is a 2-bit PackedStream (alignment_base_inl.h:75-98): an out-of-range
    text symbol (255) re-read from the cache is 3 ('T').  With a pattern of T's and a text that
    ends early the score therefore differs from a 'true mismatch' model."""
    p = np.full(40, 3, dtype=np.uint8)
    t = np.full(45, 3, dtype=np.uint8)       # N < M + BAND - 1 -> 255s enter the band
    ok, s31, _, _ = O.banded_gotoh_score(31, O.LOCAL, (2, -1, -2, -1), p, t, pat_bits=4, txt_bits=2)
    ok, s15, _, _ = O.banded_gotoh_score(15, O.LOCAL, (2, -1, -2, -1), p, t, pat_bits=4, txt_bits=2)
    assert s31 == 80 and s15 == 80


# ---------------------------------------------------------------------------- FM-index
@pytest.fixture(scope="module")
def fmi():
    rng = np.random.default_rng(7)
    n = 50021
    text = rng.integers(0, 4, n, dtype=np.uint8)
    return text, O.FMIndex(text)


def test_suffix_array_is_sorted(fmi):
    text, f = fmi
    s = bytes(text)
    sa = f.sa
    assert sa[0] == f.length and sorted(sa.tolist()) == list(range(f.length + 1))
    rng = np.random.default_rng(1)
    for i in rng.integers(1, f.length, 3000):
        assert s[int(sa[i]):] < s[int(sa[i + 1]):]


def test_rank_equals_naive_count(fmi):
    """nvbio-test/rank_test.cu:55-86 and fmindex_test.cu: rank == running naive count."""
    text, f = fmi
    n = f.length
    sa = f.sa.astype(np.int64)
    col = np.where(sa == 0, 9, text[(sa - 1) % n])      # the BWT column with '$' at `primary`
    cum = np.stack([np.cumsum(col == c) for c in range(4)], 1)
    k = np.arange(n + 1, dtype=np.uint32)
    for c in range(4):
        assert (f.rank(k, np.full(n + 1, c, np.uint8)) == cum[:, c]).all()
    assert (f.rank4(k) == cum).all()
    assert (f.rank(np.array([0xFFFFFFFF], np.uint32), np.array([2], np.uint8)) == 0).all()
    # range form == two point queries
    rng = np.random.default_rng(3)
    x = rng.integers(0, n + 1, 5000).astype(np.int64) - 1
    y = np.minimum(x + rng.integers(0, 300, 5000), n)
    c = rng.integers(0, 4, 5000).astype(np.uint8)
    rr = f.rank_range(np.stack([x.astype(np.uint32), y.astype(np.uint32)], 1), c)
    ex = np.where(x < 0, 0, cum[np.maximum(x, 0), c])
    assert (rr[:, 0] == ex).all() and (rr[:, 1] == cum[y, c]).all()


def test_match_and_locate_find_the_pattern(fmi):
    """fmindex_test.cu:610-664: every located position of a matched pattern holds the pattern."""
    text, f = fmi
    s = bytes(text)
    rng = np.random.default_rng(5)
    seeds = [text[p:p + 8].copy() for p in rng.integers(0, f.length - 8, 300)]
    seeds += [rng.integers(0, 4, 11, dtype=np.uint8) for _ in range(100)]
    seeds += [np.array([0, 1, 4, 2], np.uint8)]                    # holds an N
    ss = O.StringSet.from_lists(seeds, 4, True)
    rg = f.match(ss)
    assert rg[-1].tolist() == [1, 0]
    for i, sd in enumerate(seeds[:-1]):
        pat = bytes(sd)
        cnt = sum(1 for j in range(len(s) - len(pat) + 1) if s.startswith(pat, j)) if i % 25 == 0 else None
        x, y = int(rg[i, 0]), int(rg[i, 1])
        if cnt is not None:
            assert (y - x + 1 if x <= y else 0) == cnt
        if x <= y:
            pos = f.locate(np.arange(x, min(y, x + 10) + 1, dtype=np.uint32))
            for q in pos:
                assert s[int(q):int(q) + len(pat)] == pat
    # locate of every row reproduces the suffix array (ssa check, fmindex_test.cu:582-592)
    rows = np.arange(1, f.length + 1, dtype=np.uint32)
    pos, steps = f.locate(rows, want_steps=True)
    assert (pos == f.sa[1:]).all() and steps > 0
    it = f.locate_ssa_iterator(rows)
    assert (f.lookup_ssa_iterator(it) == pos).all()


def test_filter_rank_locate(fmi):
    text, f = fmi
    rng = np.random.default_rng(9)
    seeds = [text[p:p + 10].copy() for p in rng.integers(0, f.length - 10, 200)]
    ss = O.StringSet.from_lists(seeds, 2, True)
    total, ranges, slots = f.filter_rank(ss)
    sizes = np.where(ranges[:, 0] <= ranges[:, 1], ranges[:, 1].astype(np.int64) - ranges[:, 0] + 1, 0)
    assert total == sizes.sum() and (slots == np.cumsum(sizes)).all()
    hits = f.filter_locate(ranges, slots, 0, total)
    s = bytes(text)
    for pos, sid in hits:
        pat = bytes(seeds[int(sid)])
        assert s[int(pos):int(pos) + len(pat)] == pat


# ---------------------------------------------------------------------------- one-mismatch seed mapping
def _brute_hits(text, scan, len1, find_exact):
    """Text positions whose substring equals the scan sequence (consumed back to front by the FM-index:
    text[p + L-1-t] == scan[t]) with the mismatch pattern map<find_exact> accepts (mapping_inl.h:124-223):
    exact in scan[0,len1), exactly one substitution in scan[len1,L) (or none, if find_exact);
    a single N in [len1,L) is the forced mismatch position."""
    L = len(scan)
    n_pos = [i for i in range(L) if scan[i] > 3]
    if len(n_pos) > 1 or (n_pos and n_pos[0] < len1):
        return set()
    if n_pos:
        len1 = n_pos[0]
    want = np.array(scan[::-1], dtype=np.int16)          # text order
    win = np.lib.stride_tricks.sliding_window_view(text, L).astype(np.int16)
    mism = win != want[None, :]
    n_mism = mism.sum(1)
    in_exact = mism[:, L - len1:].any(1) if len1 else np.zeros(len(win), bool)     # scan index t <-> text offset L-1-t
    ok = ((n_mism == 1) & ~in_exact)
    if n_pos:
        ok = (n_mism == 1) & ~in_exact                   # the N position always mismatches
    elif find_exact:
        ok |= (n_mism == 0)
    return set(np.nonzero(ok)[0].tolist())


@pytest.mark.parametrize("algorithm,subseed", [(1, 8), (1, 14), (2, 0)])
def test_one_mismatch_mappers_find_exactly_the_allowed_occurrences(algorithm, subseed):
    rng = np.random.default_rng(31 + algorithm + subseed)
    text = rng.integers(0, 4, 6000, dtype=np.uint8)
    text[1000:1200] = np.tile(np.array([0, 1, 2], dtype=np.uint8), 67)[:200]
    f, rf = O.FMIndex(text), O.FMIndex(text[::-1].copy())
    L = 20
    reads = []
    for i in range(150):
        p = int(rng.integers(0, text.size - L))
        r = text[p:p + L].copy()
        if i % 2:
            r = (3 - r)[::-1].copy()
        if i % 3 == 0:
            r[int(rng.integers(0, L))] = rng.integers(0, 4)
        if i % 11 == 0:
            r[int(rng.integers(0, L))] = 4
        if i % 37 == 0:
            r[[2, 15]] = 4
        reads.append(r[::-1].copy())                     # stored reversed (io::REVERSE)
    hr = O.StringSet.from_lists(reads, 4, True)
    sf = np.full(L + 1, 7, np.uint32)
    pd = dict(seed_len=L, min_read_len=12, max_hits=1000, max_reseed=2, retry=0, rep_seeds=1000, fw=1, rc=1)
    hits, counts, reseed = O.map_seeds(algorithm, subseed, f, rf, hr, pd, sf, 400)
    rtext = text[::-1].copy()
    n_with_mismatch_hits = 0
    for i, stored in enumerate(reads):
        got = {}
        for h in hits[i, :counts[i]]:
            begin, w1 = int(h & 0xFFFFFFFF), int(h >> 32)
            delta, pos, rc, idir = w1 & 0xFFFFF, (w1 >> 20) & 0x3FF, (w1 >> 30) & 1, (w1 >> 31) & 1
            sa = (rf if idir else f).sa
            got.setdefault((rc, idir), set()).update(int(x) for x in sa[begin:begin + delta])
            assert pos == {(0, 0): 0, (1, 0): 0, (0, 1): L - 1, (1, 1): L - 1}[(rc, idir)]     # one seed per read at pos 0 here
        fwd = list(stored)                               # f_reader: scan t = stored[t]
        rev = fwd[::-1]
        comp = lambda q: [c if c > 3 else 3 - c for c in q]
        if algorithm == 1:
            nN = sum(c == 4 for c in fwd)
            exp = {} if nN >= 2 else {(0, 0): _brute_hits(text, fwd, subseed, True), (1, 0): _brute_hits(text, comp(rev), subseed, False)}
        else:
            exp = {(0, 0): _brute_hits(text, fwd, L // 2, True), (0, 1): _brute_hits(rtext, rev, (L + 1) // 2, False),
                   (1, 1): _brute_hits(rtext, comp(fwd), L // 2, True), (1, 0): _brute_hits(text, comp(rev), (L + 1) // 2, False)}
        exp = {k: v for k, v in exp.items() if v}
        assert got == exp, (i, got, exp)
        n_with_mismatch_hits += (i % 3 == 0 and i % 11 != 0 and len(got) > 0)      # substituted reads that were still found
    assert n_with_mismatch_hits > 5
    # algorithm 0 through the same entry point == the exact mapper
    h0, c0, r0 = O.map_seeds(0, 0, f, None, hr, pd, sf, 400)
    h1, c1, r1 = O.map_exact(f, hr, pd, sf, 400)
    assert (h0 == h1).all() and (c0 == c1).all() and (r0 == r1).all()


# ---------------------------------------------------------------------------- SW / edit-distance aligners
@pytest.mark.parametrize("scheme", [(0, -1, -1, -1), (2, -1, -1, -1), (1, -3, -2, -2)])
def test_linear_gap_aligners_equal_gotoh_with_open_eq_ext(scheme):
    """SmithWatermanAligner / EditDistanceAligner (linear gaps, deletion == insertion) compute the same H
    matrix as GotohAligner with gap_open == gap_ext -- the reference's own functional test expects the
    same alignments from both (alignment_test.cu:769-793).  Checked restatement against restatement:
    banded for every band and type (scores and sinks); full matrix: scores for every type, sinks for
    GLOBAL / SEMI_GLOBAL (LOCAL ties resolve by the visiting order, 16-column blocks vs 8)."""
    rng = np.random.default_rng(sum(scheme) + 50)
    g = (scheme[0], scheme[1], scheme[2], scheme[2])
    pats, txts = [], []
    for i in range(400):
        M = int(rng.integers(1, 70))
        N = M + int(rng.integers(0, 60)) if i % 5 else max(1, M - int(rng.integers(0, 3)))
        t = rng.integers(0, 4, N).astype(np.uint8)
        off = int(rng.integers(0, max(1, N - M + 1)))
        p = np.resize(t[off:off + M], M).copy()
        mut = rng.random(M) < 0.1
        p[mut] = rng.integers(0, 5, int(mut.sum()))
        pats.append(p); txts.append(t)
    hp, ht = O.StringSet.from_lists(pats, 4, True), O.StringSet.from_lists(txts + [np.zeros(64, np.uint8)], 2, True)
    ht = O.StringSet(ht.words, 2, True, ht.begin[:-1], ht.length[:-1])
    for ty in (O.GLOBAL, O.LOCAL, O.SEMI_GLOBAL):
        for band in (3, 5, 7, 15, 31):
            a, ak = O.batch_sw_score(band, ty, scheme, hp, ht)
            b, bk = O.batch_banded_gotoh_score(band, ty, g, hp, ht)
            assert (a == b).all() and (ak == bk).all(), (ty, band)
        a, ak = O.batch_sw_score(0, ty, scheme, hp, ht)
        b, bk, ok = O.batch_gotoh_score(ty, g, hp, ht)
        assert (a == b).all(), ty
        if ty != O.LOCAL:
            assert (ak == bk).all(), ty
    # the full-matrix SW / Gotoh KAT strings of the functional test: same scores from both aligners
    p, t = dna(KAT["strings"]["short_p"]), dna(KAT["strings"]["short_t"])
    hp1, ht1 = O.StringSet.from_lists([p], 4, True), O.StringSet.from_lists([t], 2, False)
    for ty in (O.GLOBAL, O.LOCAL, O.SEMI_GLOBAL):
        a, ak = O.batch_sw_score(0, ty, (2, -1, -1, -1), hp1, ht1)
        assert int(a[0]) == O.ref_sw_gotoh(ty, (2, -1, -1, -1), p, t)


def test_pattern_blocking_and_text_blocking_restatements_agree():
    """Two independent restatements (gotoh_inl.h:459-900 vs :969-1489, sw_inl.h:417-760 vs :881-1222) of the same
    matrix: scores agree everywhere, sinks agree for GLOBAL / SEMI_GLOBAL; LOCAL sinks differ only where several
    cells hold the maximum (the visiting order picks the survivor)."""
    rng = np.random.default_rng(8)
    pats, txts = [], []
    for i in range(500):
        M, N = int(rng.integers(1, 90)), int(rng.integers(1, 200))
        t = rng.integers(0, 4, N).astype(np.uint8)
        p = t[:M].copy() if N >= M else rng.integers(0, 4, M).astype(np.uint8)
        p = np.resize(p, M); mut = rng.random(M) < 0.1; p[mut] = rng.integers(0, 4, int(mut.sum()))
        pats.append(p); txts.append(t)
    hp, ht = O.StringSet.from_lists(pats, 4, True), O.StringSet.from_lists(txts, 2, False)
    differs = 0
    for ty in (O.GLOBAL, O.LOCAL, O.SEMI_GLOBAL):
        for kind, scheme in ((0, (2, -1, -2, -1)), (1, (2, -1, -1, -1)), (1, (0, -1, -1, -1))):
            a, ak, aok = O.batch_score_pattern_blocking(kind, ty, scheme, hp, ht)
            b, bk = (O.batch_gotoh_score(ty, scheme, hp, ht)[:2]) if kind == 0 else O.batch_sw_score(0, ty, scheme, hp, ht)
            assert (a == b).all() and aok.all()
            if ty != O.LOCAL:
                assert (ak == bk).all()
            else:
                differs += int((ak != bk).any(1).sum())
    assert differs > 0


def test_full_matrix_traceback_kats_and_properties():
    """alignment_traceback (full matrix, Gotoh): the three CIGAR literals of the reference's functional test
    (alignment_test.cu:788-792), and on random pairs the replayed alignment re-scores to the reported score."""
    p, t = dna(KAT["strings"]["short_p"]), dna(KAT["strings"]["short_t"])
    hp, ht = O.StringSet.from_lists([p], 4, True), O.StringSet.from_lists([t], 2, False)
    for ty, lit in ((O.GLOBAL, "1M2D3M1D3M10D"), (O.LOCAL, "4M1D3M"), (O.SEMI_GLOBAL, "4M1D3M")):
        r = O.gotoh_traceback(ty, (2, -1, -1, -1), hp, ht)
        assert O.cigar_rle(r["ops"]) == lit
        assert r["score"] == O.ref_sw_gotoh(ty, (2, -1, -1, -1), p, t)
    rng = np.random.default_rng(12)
    scheme = (2, -1, -2, -1)
    for it in range(150):
        M, N = int(rng.integers(1, 50)), int(rng.integers(1, 90))
        t = rng.integers(0, 4, N).astype(np.uint8)
        p = np.resize(t[int(rng.integers(0, N)):], M).copy()
        mut = rng.random(M) < 0.15
        p[mut] = rng.integers(0, 4, int(mut.sum()))
        hp, ht = O.StringSet.from_lists([p], 4, True), O.StringSet.from_lists([t], 2, True)
        for ty in (O.GLOBAL, O.LOCAL, O.SEMI_GLOBAL):
            r = O.gotoh_traceback(ty, scheme, hp, ht)
            sc_pb, sk_pb, _ = O.batch_score_pattern_blocking(0, ty, scheme, hp, ht)
            assert r["score"] == int(sc_pb[0]) and r["sink"] == (int(sk_pb[0, 0]), int(sk_pb[0, 1]))
            # replay from the source: x walks the text, y the pattern
            x, y = r["source"]
            s, prev = 0, None
            for op in r["ops"][::-1]:
                if op == 0:
                    s += scheme[0] if p[y] == t[x] else scheme[1]; x += 1; y += 1
                elif op == 2:
                    s += scheme[3] if prev == 2 else scheme[2]; x += 1
                else:
                    s += scheme[3] if prev == 1 else scheme[2]; y += 1
                prev = op
            assert (x, y) == r["sink"] and s == r["score"], (ty, it)
            if ty == O.GLOBAL:
                assert r["source"] == (0, 0) and r["sink"] == (N, M)
            if ty != O.LOCAL:
                assert r["source"][1] == 0 and r["sink"][1] == M


def test_sw_tracebacks_kats_and_relation_to_gotoh():
    """SW / ED tracebacks: the reference's three full-matrix SW CIGAR literals (alignment_test.cu:776-780); and against the Gotoh
    restatement with gap_open == gap_ext: identical in the band for GLOBAL / SEMI_GLOBAL and in the full matrix whenever the sinks
    coincide, different for banded LOCAL (the SW band context never marks a sink, sw_banded_inl.h:269-279)."""
    p, t = dna(KAT["strings"]["short_p"]), dna(KAT["strings"]["short_t"])
    hp, ht = O.StringSet.from_lists([p], 4, True), O.StringSet.from_lists([t], 2, False)
    for ty, lit in ((O.GLOBAL, "1M2D3M1D3M10D"), (O.LOCAL, "4M1D3M"), (O.SEMI_GLOBAL, "4M1D3M")):
        assert O.cigar_rle(O.sw_traceback(0, ty, (2, -1, -1, -1), hp, ht)["ops"]) == lit
    rng = np.random.default_rng(3)
    key = lambda r: (r["score"], r["sink"], r["source"], r["cigar"].tolist())
    local_differs = 0
    for it in range(200):
        M = int(rng.integers(1, 50)); N = M + int(rng.integers(0, 40))
        t = rng.integers(0, 4, N).astype(np.uint8)
        p = np.resize(t[int(rng.integers(0, max(1, N - M + 1))):], M).copy()
        mut = rng.random(M) < 0.12
        p[mut] = rng.integers(0, 4, int(mut.sum()))
        hp = O.StringSet.from_lists([p], 4, True)
        ht = O.StringSet.from_lists([t, np.zeros(64, np.uint8)], 2, True); ht = O.StringSet(ht.words, 2, True, ht.begin[:1], ht.length[:1])
        for ty in (O.GLOBAL, O.LOCAL, O.SEMI_GLOBAL):
            a, b = O.sw_traceback(15, ty, (2, -1, -1, -1), hp, ht), O.banded_gotoh_traceback(15, ty, (2, -1, -1, -1), hp, ht)
            if ty == O.LOCAL:
                local_differs += key(a) != key(b)
                assert a["source"][1] == 0 and (a["score"], a["sink"]) == (b["score"], b["sink"])
            else:
                assert key(a) == key(b)
            a, b = O.sw_traceback(0, ty, (2, -1, -1, -1), hp, ht), O.gotoh_traceback(ty, (2, -1, -1, -1), hp, ht)
            assert a["score"] == b["score"]
            if a["sink"] == b["sink"]:
                assert key(a) == key(b)
    assert local_differs > 0


def test_banana_suffix_array_and_bwt_literals():
    """The literals of nvbio-test/packedstream_test.cpp:161-202: the suffix array of BANANA (B=1, A=0, N=2) is 5 3 1 0 4 2 and its BWT
    (the sentinel's row dropped, as saisxx_bwt writes it) reads ANNBAA -- the oracle's index construction on the same string."""
    t = np.array([1, 0, 2, 0, 2, 0], dtype=np.uint8)
    f = O.FMIndex(t)
    assert f.sa[0] == 6 and f.sa[1:].tolist() == [5, 3, 1, 0, 4, 2]
    assert "".join("ABNT"[c] for c in f.bwt) == "ANNBAA"
    assert f.primary == 4                       # the row of the whole string: where the dropped sentinel stood
