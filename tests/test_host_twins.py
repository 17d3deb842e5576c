"""The `_host` twins of the C-ABI (nvbio_amd/csrc/host_twins.hip): the reference's HostThreadScheduler / host fm_index paths
behind the device entry points' argument lists, with host pointers.  CPU suite: against the oracle, bit for bit."""
import ctypes as C

import numpy as np
import pytest

import nvbio_amd
from nvbio_amd import _lib
from oracle import pyoracle as O
from tests.test_compat_alignment_gpu import make_jobs

GLOBAL, LOCAL, SEMI = 0, 1, 2


def sset(hs):
    s = _lib.StringSetStruct()
    s.words, s.n_words, s.bits, s.big_endian = hs.words.ctypes.data, hs.words.size, hs.bits, hs.big_endian
    s.begin, s.length, s.fixed_length = hs.begin.ctypes.data, hs.length.ctypes.data, 0
    return s


@pytest.mark.parametrize("typ", [GLOBAL, LOCAL, SEMI])
@pytest.mark.parametrize("band", [7, 15, 31])
def test_banded_host_twins(typ, band):
    L = nvbio_amd.lib()
    reads, _, wins = make_jobs(3 + band + typ, 2500, band=band)
    for pb, tb, pbe, tbe in ((4, 2, True, False), (2, 2, True, True), (8, 8, False, False)):
        rr = [np.minimum(r, 3) for r in reads] if pb == 2 else reads
        hp, ht = O.StringSet.from_lists(rr, pb, pbe), O.StringSet.from_lists(wins, tb, tbe)
        sp, st = sset(hp), sset(ht)
        score = np.zeros(len(reads), np.int32); sink = np.zeros((len(reads), 2), np.uint32)
        g = _lib.GotohSchemeStruct(2, -1, -2, -1)
        assert L.nvbio_hip_banded_gotoh_score_host(C.byref(g), typ, band, C.byref(sp), C.byref(st), len(reads), score.ctypes.data, sink.ctypes.data, 0) == 0
        es, ek = O.batch_banded_gotoh_score(band, typ, (2, -1, -2, -1), hp, ht)
        assert (score == es).all() and (sink == ek).all()
        w = _lib.GotohSchemeStruct(1, -2, -3, -1)              # {match, mismatch, deletion, insertion}: asymmetric linear gaps
        assert L.nvbio_hip_banded_sw_score_host(C.byref(w), typ, band, C.byref(sp), C.byref(st), len(reads), score.ctypes.data, sink.ctypes.data, 2) == 0
        es, ek = O.batch_sw_score(band, typ, (1, -2, -3, -1), hp, ht)
        assert (score == es).all() and (sink == ek).all()
    assert L.nvbio_hip_banded_gotoh_score_host(C.byref(g), typ, 9, C.byref(sp), C.byref(st), 1, score.ctypes.data, sink.ctypes.data, 0) == 801


@pytest.mark.parametrize("typ", [GLOBAL, LOCAL, SEMI])
@pytest.mark.parametrize("algorithm", [0, 1])
def test_full_matrix_host_twin(typ, algorithm):
    L = nvbio_amd.lib()
    reads, _, wins = make_jobs(90 + typ + algorithm, 600, max_read=110, full=True, short_text_every=10 ** 9)
    reads = [r if len(r) else np.array([1], np.uint8) for r in reads]
    hp, ht = O.StringSet.from_lists(reads, 4, True), O.StringSet.from_lists(wins, 2, False)
    sp, st = sset(hp), sset(ht)
    n = len(reads)
    rng = np.random.default_rng(1)
    th = np.where(rng.random(n) < 0.5, -(1 << 30), rng.integers(-40, 160, n)).astype(np.int32)
    score = np.zeros(n, np.int32); sink = np.zeros((n, 2), np.uint32); ok = np.zeros(n, np.uint8)
    sc = np.array((2, -1, -2, -1), np.int32)
    assert L.nvbio_hip_alignment_score_host(0, algorithm, sc.ctypes.data, typ, C.byref(sp), C.byref(st), th.ctypes.data, n,
                                            score.ctypes.data, sink.ctypes.data, ok.ctypes.data, 0) == 0
    if algorithm == 0:
        es, ek, eo = O.batch_score_pattern_blocking(0, typ, (2, -1, -2, -1), hp, ht, min_score=th)
    else:
        es, ek, eo = O.batch_gotoh_score(typ, (2, -1, -2, -1), hp, ht, min_score=th)
    assert (score == es).all() and (sink == ek).all() and (ok == eo).all()
    assert 0 < int(eo.sum()) < n                       # both outcomes of the early exit occur
    sw = np.array((2, -2, -4, -1), np.int32)
    assert L.nvbio_hip_alignment_score_host(1, algorithm, sw.ctypes.data, typ, C.byref(sp), C.byref(st), None, n,
                                            score.ctypes.data, sink.ctypes.data, None, 0) == 0
    if algorithm == 0:
        es, ek, _ = O.batch_score_pattern_blocking(1, typ, (2, -2, -4, -1), hp, ht)
    else:
        es, ek = O.batch_sw_score(0, typ, (2, -2, -4, -1), hp, ht)
    assert (score == es).all() and (sink == ek).all()


@pytest.mark.parametrize("sa_int", [16, 4])
def test_fm_host_twins(sa_int):
    L = nvbio_amd.lib()
    rng = np.random.default_rng(12)
    n = 70001
    text = rng.integers(0, 4, n, dtype=np.uint8)
    host = O.FMIndex(text, sa_int=sa_int)
    f = _lib.FMIndexStruct()
    f.length, f.primary, f.sa_int = host.length, host.primary, sa_int
    for i in range(5):
        f.L2[i] = int(host.L2[i])
    f.bwt_occ, f.ssa = host.bwt_occ.ctypes.data, host.ssa.ctypes.data
    k = np.concatenate([rng.integers(0, n + 1, 20000), [0xFFFFFFFF, n, host.primary]]).astype(np.uint32)
    c = rng.integers(0, 4, k.size).astype(np.uint8)
    out = np.zeros(k.size, np.uint32)
    assert L.nvbio_hip_fm_rank_host(C.byref(f), k.ctypes.data, c.ctypes.data, k.size, out.ctypes.data, 0) == 0
    assert (out == host.rank(k, c)).all()
    seeds = []
    for i in range(8000):
        ln = int(rng.integers(1, 40))
        s = text[(p := int(rng.integers(0, n - ln))):p + ln].copy() if i % 5 else rng.integers(0, 4, ln, dtype=np.uint8)
        if i % 40 == 3:
            s[int(rng.integers(0, ln))] = 4
        seeds.append(s)
    hs = O.StringSet.from_lists(seeds, 4, True)
    ss = sset(hs)
    ranges = np.zeros((len(seeds), 2), np.uint32)
    assert L.nvbio_hip_fm_match_host(C.byref(f), C.byref(ss), len(seeds), ranges.ctypes.data, 0) == 0
    assert (ranges == host.match(hs)).all()
    rows = rng.integers(0, n + 1, 20000).astype(np.uint32)
    pos = np.zeros(rows.size, np.uint32)
    assert L.nvbio_hip_fm_locate_host(C.byref(f), rows.ctypes.data, rows.size, pos.ctypes.data, 0) == 0
    assert (pos == host.locate(rows)).all()
