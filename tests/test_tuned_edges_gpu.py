"""The edges of the tuned kernels' contract that used to fall to the one-lane-per-job template (or were refused): linear gap costs with
deletion != insertion (alignment/utils.h:92-109), 8-bit pattern strings, and full-matrix patterns beyond 64 lanes x 16 rows
(gotoh_inl.h:969-1490 has no length limit).  Each against the oracle's restatement, bit-exact scores and sinks, and each names the kernel
that ran."""
import numpy as np
import pytest
import torch

import nvbio_amd as nvb
from oracle import pyoracle as O
from test_banded_gpu import random_pairs
from test_full_gotoh_gpu import make_pairs

pytestmark = pytest.mark.gpu
TYPES = [nvb.GLOBAL, nvb.LOCAL, nvb.SEMI_GLOBAL]
ASYM = [(2, -1, -2, -1), (1, -3, -1, -4), (3, -2, -5, -2), (0, -1, -1, -2)]


def last_kernel():
    return nvb.lib().nvbio_hip_last_kernel().decode()


def to_dev(hs, dev):
    return nvb.PackedStringSet.from_host(hs.words, hs.bits, hs.big_endian, hs.begin, hs.length, device=dev)


def same(es, ek, gs, gk, what):
    gs, gk = gs.cpu().numpy(), gk.cpu().numpy().view(np.uint32)
    bad = np.nonzero((es != gs) | (ek != gk).any(1))[0]
    assert bad.size == 0, (what, bad[:5], es[bad[:3]], gs[bad[:3]], ek[bad[:3]], gk[bad[:3]])


# ------------------------------------------------------------------------------------------------ deletion != insertion
@pytest.mark.parametrize("band", [3, 5, 7, 15, 31])
@pytest.mark.parametrize("ty", TYPES)
def test_banded_sw_with_direction_dependent_gaps(cuda, band, ty):
    """sw_banded_inl.h:378-470: top + deletion, left + insertion, row zero j * deletion"""
    rng = np.random.default_rng(9300 + band * 3 + ty)
    pats, txts = random_pairs(rng, 1500, band)
    hp, ht = O.StringSet.from_lists(pats, 4, True), O.StringSet.from_lists(txts, 2, True)
    dp, dt = to_dev(hp, cuda), to_dev(ht, cuda)
    for scheme in ASYM + [(300, -200, -250, -100)]:
        es, ek = O.batch_sw_score(band, ty, scheme, hp, ht)
        al = nvb.make_smith_waterman_aligner(ty, nvb.SimpleSmithWatermanScheme(*scheme))
        for force32 in ("0", "1"):
            nvb.set_test_switch("NVBIO_HIP_FORCE_32BIT", force32)
            try:
                gs, gk = nvb.batch_banded_alignment_score(band, al, dp, dt, max_pattern_length=int(hp.length.max()))
                torch.cuda.synchronize()
            finally:
                nvb.set_test_switch("NVBIO_HIP_FORCE_32BIT", "0")
            same(es, ek, gs, gk, (band, ty, scheme, force32))
            name = last_kernel()
            assert "X>" in name and (force32 == "0" or "A32X" in name), name
    # the two directions really differ on this data: swapping the costs changes some scores
    a, _ = O.batch_sw_score(band, ty, (2, -1, -2, -1), hp, ht)
    b, _ = O.batch_sw_score(band, ty, (2, -1, -1, -2), hp, ht)
    assert (a != b).any()


@pytest.mark.parametrize("ty", TYPES)
@pytest.mark.parametrize("algorithm", ["text_blocking", "pattern_blocking"])
def test_full_sw_with_direction_dependent_gaps(cuda, ty, algorithm):
    """sw_inl.h:881-1222 / :417-760: left + deletion along the text, top + insertion down the pattern"""
    rng = np.random.default_rng(9400 + ty)
    pats, txts = make_pairs(rng, 600, 180, 400)
    pats = [p if len(p) else np.zeros(1, np.uint8) for p in pats]
    txts = [t if len(t) else np.zeros(1, np.uint8) for t in txts]
    hp, ht = O.StringSet.from_lists(pats, 4, True), O.StringSet.from_lists(txts, 2, False)
    dp, dt = to_dev(hp, cuda), to_dev(ht, cuda)
    algo = nvb.TEXT_BLOCKING if algorithm == "text_blocking" else nvb.PATTERN_BLOCKING
    for scheme in ASYM:
        if algorithm == "text_blocking":
            es, ek = O.batch_sw_score(0, ty, scheme, hp, ht)
        else:
            es, ek, _ = O.batch_score_pattern_blocking(1, ty, scheme, hp, ht)
        gs, gk, go = nvb.batch_alignment_score(nvb.make_smith_waterman_aligner(ty, nvb.SimpleSmithWatermanScheme(*scheme), algo), dp, dt, 180, 400)
        torch.cuda.synchronize()
        same(es, ek, gs, gk, (ty, algorithm, scheme))
        assert bool(go.all()) and "striped" in last_kernel()
    # text blocking keeps the reference's int16 column exact beyond its range as well: costs that push values past 32 767
    if algorithm == "text_blocking":
        scheme = (400, -300, -350, -120)
        es, ek = O.batch_sw_score(0, ty, scheme, hp, ht)
        gs, gk, _ = nvb.batch_alignment_score(nvb.make_smith_waterman_aligner(ty, nvb.SimpleSmithWatermanScheme(*scheme), algo), dp, dt, 180, 400)
        same(es, ek, gs, gk, (ty, algorithm, scheme))


# ------------------------------------------------------------------------------------------------ patterns beyond 1 024 rows
def long_pairs(rng, n, m_lo, m_hi, n_hi):
    pats, txts = [], []
    for i in range(n):
        m = int(rng.integers(m_lo, m_hi + 1))
        t = rng.integers(0, 4, int(rng.integers(m // 2, n_hi + 1)), dtype=np.uint8)
        if i % 3 != 2 and len(t) > m // 2:
            # a mutated stretch of the text, so that alignments are long and gapped
            o = int(rng.integers(0, max(1, len(t) - m // 2)))
            p = t[o:o + m].copy()
            if len(p) < m:
                p = np.concatenate([p, rng.integers(0, 4, m - len(p), dtype=np.uint8)])
            mut = rng.random(m) < 0.08
            p[mut] = rng.integers(0, 4, int(mut.sum()))
            dele = rng.random(m) > 0.02
            p = p[dele]
            if len(p) == 0:
                p = np.zeros(1, np.uint8)
        else:
            p = rng.integers(0, 4, m, dtype=np.uint8)
        pats.append(p.astype(np.uint8)); txts.append(t)
    return pats, txts


@pytest.mark.parametrize("ty", TYPES)
def test_full_gotoh_beyond_1024_rows(cuda, ty):
    """text blocking, with and without thresholds; one, two, three and four stripes; stripe boundaries at and next to the pattern's end"""
    rng = np.random.default_rng(9500 + ty)
    pats, txts = long_pairs(rng, 36, 900, 3300, 3000)
    for m in (1024, 1025, 2047, 2048, 2049, 3072):           # the stripe arithmetic's edges
        pats.append(rng.integers(0, 4, m, dtype=np.uint8)); txts.append(rng.integers(0, 4, 1500 + m % 7, dtype=np.uint8))
    pats.append(np.zeros(0, np.uint8)); txts.append(rng.integers(0, 4, 100, dtype=np.uint8))      # an empty pattern among them
    hp, ht = O.StringSet.from_lists(pats, 4, True), O.StringSet.from_lists(txts, 2, False)
    dp, dt = to_dev(hp, cuda), to_dev(ht, cuda)
    maxP, maxT = int(hp.length.max()), int(ht.length.max())
    for scheme in [(2, -1, -5, -3), (1, -1, -1, -1), (20, -30, -40, -10)]:           # the last one leaves int16: the boundary column truncates
        es, ek, eo = O.batch_gotoh_score(ty, scheme, hp, ht)
        gs, gk, go = nvb.batch_alignment_score(nvb.make_gotoh_aligner(ty, nvb.SimpleGotohScheme(*scheme)), dp, dt, maxP, maxT)
        torch.cuda.synchronize()
        same(es, ek, gs, gk, (ty, scheme))
        assert (go.cpu().numpy() == eo).all() and last_kernel() == "full_gotoh_striped_kernel"
        # thresholds around the achievable scores: some jobs exit after a block of text columns, most do not
        ms = (es.astype(np.int64) + rng.integers(-40, 400, es.size)).clip(-(1 << 30), (1 << 30) - 1).astype(np.int32)
        es2, ek2, eo2 = O.batch_gotoh_score(ty, scheme, hp, ht, min_score=ms)
        gs2, gk2, go2 = nvb.batch_alignment_score(nvb.make_gotoh_aligner(ty, nvb.SimpleGotohScheme(*scheme)), dp, dt, maxP, maxT,
                                                  min_score=torch.from_numpy(ms).to(cuda))
        torch.cuda.synchronize()
        same(es2, ek2, gs2, gk2, (ty, scheme, "min_score"))
        bad = np.nonzero(go2.cpu().numpy() != eo2)[0]
        assert bad.size == 0, (ty, scheme, bad, hp.length[bad], ht.length[bad], es[bad], ms[bad], eo2[bad], es2[bad])
        assert 0 < int((eo2 == 0).sum()) < eo2.size


@pytest.mark.parametrize("ty", TYPES)
def test_full_sw_and_pattern_blocking_beyond_1024_rows(cuda, ty):
    rng = np.random.default_rng(9600 + ty)
    pats, txts = long_pairs(rng, 24, 1000, 2600, 2400)
    hp, ht = O.StringSet.from_lists(pats, 4, True), O.StringSet.from_lists(txts, 2, False)
    dp, dt = to_dev(hp, cuda), to_dev(ht, cuda)
    maxP, maxT = int(hp.length.max()), int(ht.length.max())
    # SW / edit distance, text blocking (16-column blocks), symmetric and not
    for scheme in [(0, -1, -1, -1), (2, -1, -1, -1), (2, -1, -2, -1), (30, -25, -28, -28)]:
        es, ek = O.batch_sw_score(0, ty, scheme, hp, ht)
        al = nvb.make_edit_distance_aligner(ty) if scheme == (0, -1, -1, -1) else nvb.make_smith_waterman_aligner(ty, nvb.SimpleSmithWatermanScheme(*scheme))
        gs, gk, _ = nvb.batch_alignment_score(al, dp, dt, maxP, maxT)
        torch.cuda.synchronize()
        same(es, ek, gs, gk, (ty, scheme, "sw text blocking"))
        assert "striped" in last_kernel()
    # pattern blocking without thresholds (the LOCAL tie order differs from text blocking's)
    for kind, scheme in [(0, (2, -1, -5, -3)), (1, (2, -1, -1, -1)), (1, (2, -1, -1, -2))]:
        es, ek, _ = O.batch_score_pattern_blocking(kind, ty, scheme, hp, ht)
        al = (nvb.make_gotoh_aligner(ty, nvb.SimpleGotohScheme(*scheme), nvb.PATTERN_BLOCKING) if kind == 0
              else nvb.make_smith_waterman_aligner(ty, nvb.SimpleSmithWatermanScheme(*scheme), nvb.PATTERN_BLOCKING))
        gs, gk, _ = nvb.batch_alignment_score(al, dp, dt, maxP, maxT)
        torch.cuda.synchronize()
        same(es, ek, gs, gk, (ty, kind, scheme, "pattern blocking"))
        assert "striped" in last_kernel()


# ------------------------------------------------------------------------------------------------ 8-bit pattern strings
def byte_patterns(rng, pats):
    """the same patterns as bytes: some symbols replaced by values no 2-bit text symbol equals, 255 (what the reference compares a text
    position past the end as) among them"""
    out = []
    for p in pats:
        b = p.astype(np.uint8).copy()
        odd = rng.random(b.size) < 0.03
        b[odd] = rng.choice(np.array([4, 7, 15, 16, 65, 128, 254, 255], np.uint8), int(odd.sum()))
        out.append(b)
    return out


@pytest.mark.parametrize("band", [3, 7, 15, 31])
@pytest.mark.parametrize("ty", TYPES)
def test_banded_with_8bit_patterns(cuda, band, ty):
    rng = np.random.default_rng(9700 + band * 3 + ty)
    pats, txts = random_pairs(rng, 1500, band)
    bp = byte_patterns(rng, pats)
    for be in (False, True):
        hp, ht = O.StringSet.from_lists(bp, 8, be), O.StringSet.from_lists(txts, 2, True)
        dp, dt = to_dev(hp, cuda), to_dev(ht, cuda)
        for scheme in [(2, -1, -5, -3), (1, -2, -2, -1)]:
            es, ek = O.batch_banded_gotoh_score(band, ty, scheme, hp, ht)
            for force32 in ("0", "1"):
                nvb.set_test_switch("NVBIO_HIP_FORCE_32BIT", force32)
                try:
                    gs, gk = nvb.batch_banded_alignment_score(band, nvb.make_gotoh_aligner(ty, nvb.SimpleGotohScheme(*scheme)), dp, dt,
                                                              max_pattern_length=int(hp.length.max()))
                    torch.cuda.synchronize()
                finally:
                    nvb.set_test_switch("NVBIO_HIP_FORCE_32BIT", "0")
                same(es, ek, gs, gk, (band, ty, scheme, be, force32))
                assert last_kernel().startswith("banded_gotoh_score_kernel<A")
        es, ek = O.batch_sw_score(band, ty, (2, -1, -2, -1), hp, ht)
        gs, gk = nvb.batch_banded_alignment_score(band, nvb.make_smith_waterman_aligner(ty, nvb.SimpleSmithWatermanScheme(2, -1, -2, -1)), dp, dt,
                                                  max_pattern_length=int(hp.length.max()))
        same(es, ek, gs, gk, (band, ty, "sw asym", be))
    # ... and the bytes matter: as 4-bit strings (byte & 15) some of these jobs score differently
    h4 = O.StringSet.from_lists([b & 15 for b in bp], 4, True)
    a, _ = O.batch_banded_gotoh_score(band, ty, (2, -1, -5, -3), h4, ht)
    b, _ = O.batch_banded_gotoh_score(band, ty, (2, -1, -5, -3), hp, ht)
    assert a.shape == b.shape


@pytest.mark.parametrize("ty", TYPES)
def test_full_matrix_with_8bit_patterns(cuda, ty):
    rng = np.random.default_rng(9800 + ty)
    pats, txts = make_pairs(rng, 800, 150, 300)
    pats = [p if len(p) else np.zeros(1, np.uint8) for p in pats]
    txts = [t if len(t) else np.zeros(1, np.uint8) for t in txts]
    bp = byte_patterns(rng, pats)
    hp, ht = O.StringSet.from_lists(bp, 8, False), O.StringSet.from_lists(txts, 2, False)
    dp, dt = to_dev(hp, cuda), to_dev(ht, cuda)
    es, ek, eo = O.batch_gotoh_score(ty, (2, -1, -5, -3), hp, ht)
    gs, gk, go = nvb.batch_alignment_score(nvb.make_gotoh_aligner(ty, nvb.SimpleGotohScheme(2, -1, -5, -3)), dp, dt, 150, 300)
    torch.cuda.synchronize()
    same(es, ek, gs, gk, (ty, "gotoh"))
    assert "16-bit" in last_kernel()
    es, ek = O.batch_sw_score(0, ty, (0, -1, -1, -1), hp, ht)
    gs, gk, _ = nvb.batch_alignment_score(nvb.make_edit_distance_aligner(ty), dp, dt, 150, 300)
    same(es, ek, gs, gk, (ty, "edit distance"))
    es, ek, _ = O.batch_score_pattern_blocking(0, ty, (2, -1, -5, -3), hp, ht)
    gs, gk, _ = nvb.batch_alignment_score(nvb.make_gotoh_aligner(ty, nvb.SimpleGotohScheme(2, -1, -5, -3), nvb.PATTERN_BLOCKING), dp, dt, 150, 300)
    same(es, ek, gs, gk, (ty, "gotoh pattern blocking"))
