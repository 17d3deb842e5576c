"""The drop-in template layer (include/nvbio_hip/compat): a caller TU written against the reference's stream concept
(tests/compat/aln_callers.hip -- user stream classes with init_context / load_strings / output functors, byte strings,
a user scoring scheme, Best2Sink, HostThreadScheduler) compiled by hipcc with `-I include/nvbio_hip/compat`, its
results compared bit for bit with the CPU oracle.  Recognised packed streams must run on the tuned kernels, everything
else on the generic per-lane templates."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from oracle import pyoracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GLOBAL, LOCAL, SEMI = 0, 1, 2


@pytest.fixture(scope="module")
def callers():
    path = os.path.join(ROOT, "tests", "compat", "libaln_callers.so")
    assert os.path.exists(path), "build with python -c 'import __graft_entry__ as g; g.build()'"
    L = C.CDLL(path)
    vp, u32, i32 = C.c_void_p, C.c_uint32, C.c_int
    L.compat_banded_score.argtypes = [i32, i32, i32, i32, i32, vp, u32, vp, vp, vp, u32, vp, vp, u32, vp, vp, C.c_char_p]
    L.compat_full_score.argtypes = [i32, i32, i32, i32, i32, vp, u32, vp, vp, vp, u32, vp, vp, u32, vp, vp, vp, C.c_char_p]
    L.compat_per_thread_score.argtypes = [vp, u32, vp, vp, vp, vp, vp]
    return L


def make_jobs(seed, n, max_read=150, band=31, max_sym=4, short_text_every=53, full=False):
    """ragged reads (empty ones included) next to reference windows they mostly derive from (substitutions, indels, Ns)."""
    rng = np.random.default_rng(seed)
    reads, quals, wins = [], [], []
    for i in range(n):
        L = int(rng.integers(0, max_read + 1)) if i % 7 else int(rng.integers(0, 4))
        W = L + band + int(rng.integers(0, 12)) if not full else int(rng.integers(L, 3 * L + 40))
        w = rng.integers(0, 4, W, dtype=np.uint8)
        off = int(rng.integers(0, band // 2 + 1)) if W >= L + band // 2 else 0
        r = w[off:off + L].copy()
        if r.size < L:
            r = np.concatenate([r, rng.integers(0, 4, L - r.size, dtype=np.uint8)])
        mut = rng.random(L) < 0.06
        r[mut] = rng.integers(0, 4, int(mut.sum()), dtype=np.uint8)
        if L > 20 and i % 3 == 0:
            p = int(rng.integers(5, L - 5))
            r = np.concatenate([r[:p], r[p + 2:], rng.integers(0, 4, 2, dtype=np.uint8)]) if i % 2 else np.concatenate([r[:p], rng.integers(0, 4, 2, dtype=np.uint8), r[p:-2]])
        if max_sym > 3 and L and i % 11 == 0:
            r[int(rng.integers(0, L))] = max_sym
        if i % short_text_every == short_text_every - 1 and L > 3:
            w = w[:L - 2]                                     # a text shorter than its pattern: the job is refused
        reads.append(r.astype(np.uint8)); quals.append(rng.integers(0, 60, L, dtype=np.uint8)); wins.append(w)
    return reads, quals, wins


def offsets(strings):
    o = np.zeros(len(strings) + 1, dtype=np.uint32)
    o[1:] = np.cumsum([len(s) for s in strings])
    return o


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


class Batch:
    def __init__(self, reads, quals, wins, packed, on_device):
        self.n = len(reads)
        self.ro, self.wo = offsets(reads), offsets(wins)
        cat_r = np.concatenate(reads) if reads else np.zeros(0, np.uint8)
        cat_w = np.concatenate(wins) if wins else np.zeros(0, np.uint8)
        self.cat_q = np.concatenate(quals)
        self.longest_read = int(max(len(r) for r in reads))
        self.longest_win = int(max(len(w) for w in wins))
        if packed:
            self.hr = O.StringSet(O.pack(cat_r, 4, True), 4, True, self.ro[:-1].astype(np.uint64), np.diff(self.ro))
            self.hw = O.StringSet(O.pack(cat_w, 2, False), 2, False, self.wo[:-1].astype(np.uint64), np.diff(self.wo))
            rd, wd = self.hr.words, self.hw.words
        else:
            pad = np.zeros(8, np.uint8)
            rb, wb = np.concatenate([cat_r, pad]), np.concatenate([cat_w, pad])
            self.hr = O.StringSet(np.frombuffer(np.concatenate([rb, np.zeros(-rb.size % 4, np.uint8)]).tobytes(), dtype=np.uint32), 8, False,
                                  self.ro[:-1].astype(np.uint64), np.diff(self.ro))
            self.hw = O.StringSet(np.frombuffer(np.concatenate([wb, np.zeros(-wb.size % 4, np.uint8)]).tobytes(), dtype=np.uint32), 8, False,
                                  self.wo[:-1].astype(np.uint64), np.diff(self.wo))
            rd, wd = rb, wb
        mk = dev if on_device else (lambda a: torch.from_numpy(np.ascontiguousarray(a).copy()))
        self.t = dict(ro=mk(self.ro.view(np.int32)), wo=mk(self.wo.view(np.int32)), r=mk(rd.view(np.int32) if packed else rd), w=mk(wd.view(np.int32) if packed else wd),
                      q=mk(np.concatenate([self.cat_q, np.zeros(8, np.uint8)])))
        self.score = mk(np.full(self.n, 12345, np.int32))
        self.sink = mk(np.full((self.n, 2), 777, np.int32))

    def ptr(self, k):
        return C.c_void_p(self.t[k].data_ptr())

    def results(self):
        return self.score.cpu().numpy(), self.sink.cpu().numpy().view(np.uint32)


def run_banded(L, b, strings, where, kind, typ, band, scheme):
    sc = np.array(scheme, dtype=np.int32)
    path = C.create_string_buffer(16)
    rc = L.compat_banded_score(strings, where, kind, typ, band, sc.ctypes.data, b.n, b.ptr("ro"), b.ptr("r"), b.ptr("q"), b.longest_read,
                               b.ptr("wo"), b.ptr("w"), b.longest_win, C.c_void_p(b.score.data_ptr()), C.c_void_p(b.sink.data_ptr()), path)
    assert rc == 0, (rc, path.value)
    return path.value.decode()


def expect_banded(b, kind, typ, band, scheme):
    if kind == 0:
        return O.batch_banded_gotoh_score(band, typ, scheme, b.hr, b.hw)
    if kind in (1, 2):
        return O.batch_sw_score(band, typ, scheme if kind == 1 else (0, -1, -1, -1), b.hr, b.hw)
    lut = np.array([-(2 + min(q, 40) // 10) for q in range(256)], dtype=np.int32)
    return O.batch_banded_gotoh_score_qual(band, typ, (2, -8, -3, -6, -2, 0), lut, np.concatenate([b.cat_q, np.zeros(8, np.uint8)]), b.hr, b.hw)


@pytest.mark.parametrize("typ", [GLOBAL, LOCAL, SEMI])
@pytest.mark.parametrize("kind,scheme", [(0, (2, -1, -2, -1)), (0, (0, -5, -8, -3)), (1, (2, -2, -3, -3)), (2, (0, -1, -1, -1))])
@pytest.mark.parametrize("band", [15, 31])
def test_packed_stream_runs_on_the_tuned_kernels(callers, typ, kind, scheme, band):
    reads, quals, wins = make_jobs(100 + band + typ, 3000, band=band)
    b = Batch(reads, quals, wins, packed=True, on_device=True)
    path = run_banded(callers, b, 0, 0, kind, typ, band, scheme)
    assert path == "tuned"
    es, ek = expect_banded(b, kind, typ, band, scheme)
    gs, gk = b.results()
    assert (gs == es).all() and (gk == ek).all()
    assert (es == -(1 << 30)).sum() > 10          # refused jobs reach output() with the untouched sink


@pytest.mark.parametrize("typ", [GLOBAL, SEMI])
@pytest.mark.parametrize("packed", [True, False], ids=["packed", "bytes"])
@pytest.mark.parametrize("band", [15, 31])
def test_banded_bitvector_edit_distance_on_the_device(callers, typ, packed, band):
    """EditDistanceAligner<TYPE, MyersTag<5>> through BatchedBandedAlignmentScore on the device (what examples/fmmap/fmmap.cu:359-367 calls): the
    generic lanes run the banded bit-vector algorithm of alignment.h, against the oracle's restatement of myers_banded_inl.h.  The stream's
    'no threshold' -2^30 narrows to an int16 0: only windows that hold the read exactly report -- reads with an N never do."""
    rng = np.random.default_rng(40 + band + typ)
    reads, quals, wins = [], [], []
    for i in range(3000):
        L = int(rng.integers(1, 140))
        W = L + band if typ == GLOBAL else L + band + int(rng.integers(0, 9))
        w = rng.integers(0, 4, W, dtype=np.uint8)
        off = int(rng.integers(0, band))
        r = w[off:off + L].copy()
        if i % 3 == 0:
            m = rng.random(L) < 0.03; r[m] = (r[m] + 1) & 3
        if i % 5 == 0:
            r[int(rng.integers(0, L))] = 4
        if i % 41 == 40 and L > 3:
            w = w[:L - 1]
        reads.append(r); quals.append(np.zeros(L, np.uint8)); wins.append(w)
    b = Batch(reads, quals, wins, packed=packed, on_device=True)
    path = run_banded(callers, b, 0 if packed else 1, 0, 4, typ, band, (0, -1, -1, -1))
    assert path == "generic"
    es, ek = O.batch_banded_myers_score(band, typ, 5, b.hr, b.hw)
    if not packed:                                   # (the byte-string stream of the test declines every 97th job: it is output as its context was built)
        declined = (np.arange(b.n) % 97) == 96
        es[declined] = -(1 << 30); ek[declined] = 0xFFFFFFFF
    gs, gk = b.results()
    assert (gs == es).all() and (gk == ek).all()
    reported = es > -(1 << 30)
    assert reported.sum() > (300 if typ == SEMI else 20) and (es[reported] == 0).all()          # (GLOBAL: only a read sitting on the band's last diagonal ends at distance 0)
    has_n = np.array([bool((r == 4).any()) for r in reads])
    assert not reported[has_n].any()


def test_asymmetric_linear_gaps_run_on_the_tuned_kernels(callers):
    """deletion != insertion (alignment/utils.h:92-109): banded and full matrix, both tags -- refused by the tuned kernels until round 6"""
    reads, quals, wins = make_jobs(7, 2000, band=15)
    b = Batch(reads, quals, wins, packed=True, on_device=True)
    for typ in (GLOBAL, LOCAL, SEMI):
        assert run_banded(callers, b, 0, 0, 1, typ, 15, (2, -2, -4, -1)) == "tuned"
        es, ek = O.batch_sw_score(15, typ, (2, -2, -4, -1), b.hr, b.hw)
        gs, gk = b.results()
        assert (gs == es).all() and (gk == ek).all()
    reads, quals, wins = make_jobs(8, 900, max_read=120, full=True, short_text_every=10 ** 9)
    reads = [r if len(r) else np.array([1], np.uint8) for r in reads]
    quals = [q if len(q) else np.array([0], np.uint8) for q in quals]
    b = Batch(reads, quals, wins, packed=True, on_device=True)
    for typ in (GLOBAL, LOCAL, SEMI):
        for tag in (0, 1):
            assert run_full(callers, b, 0, 0, 1, typ, tag, (2, -2, -4, -1), None) == "tuned"
            es, ek = expect_full(b, 1, typ, tag, (2, -2, -4, -1), None)
            gs, gk = b.results()
            assert (gs == es).all() and (gk == ek).all(), (typ, tag)


class BytePatternBatch(Batch):
    """patterns as bytes (symbols no 2-bit text symbol equals among them, 255 included), windows packed 2 bits per base"""
    def __init__(self, reads, quals, wins):
        Batch.__init__(self, [r & 3 for r in reads], quals, wins, packed=True, on_device=True)
        cat_r = np.concatenate(reads + [np.zeros(8, np.uint8)])
        self.hr = O.StringSet(np.frombuffer(np.concatenate([cat_r, np.zeros(-cat_r.size % 4, np.uint8)]).tobytes(), dtype=np.uint32), 8, False,
                              self.ro[:-1].astype(np.uint64), np.diff(self.ro))
        self.t["r"] = dev(cat_r)


@pytest.mark.parametrize("typ", [GLOBAL, LOCAL, SEMI])
def test_byte_patterns_over_a_packed_reference_run_in_place(callers, typ):
    """vector_view<const uint8*> patterns: an 8-bit string set behind the C-ABI, nothing staged"""
    rng = np.random.default_rng(900 + typ)
    for band in (15, 31):
        reads, quals, wins = make_jobs(910 + band + typ, 2500, band=band)
        for r in reads:
            odd = rng.random(r.size) < 0.03
            r[odd] = rng.choice(np.array([4, 9, 16, 77, 200, 255], np.uint8), int(odd.sum()))
        b = BytePatternBatch(reads, quals, wins)
        for kind, scheme in ((0, (2, -1, -2, -1)), (1, (2, -2, -4, -1))):
            assert run_banded(callers, b, 2, 0, kind, typ, band, scheme) == "tuned"
            es, ek = expect_banded(b, kind, typ, band, scheme)
            gs, gk = b.results()
            assert (gs == es).all() and (gk == ek).all(), (band, kind)
    reads, quals, wins = make_jobs(950 + typ, 900, max_read=120, full=True, short_text_every=10 ** 9)
    reads = [r if len(r) else np.array([1], np.uint8) for r in reads]
    quals = [q if len(q) else np.array([0], np.uint8) for q in quals]
    for r in reads:
        odd = rng.random(r.size) < 0.03
        r[odd] = rng.choice(np.array([4, 9, 16, 77, 200, 255], np.uint8), int(odd.sum()))
    b = BytePatternBatch(reads, quals, wins)
    for tag in (0, 1):
        for kind, scheme in ((0, (2, -1, -2, -1)), (1, (2, -2, -3, -3))):
            assert run_full(callers, b, 2, 0, kind, typ, tag, scheme, None) == "tuned"
            es, ek = expect_full(b, kind, typ, tag, scheme, None)
            gs, gk = b.results()
            assert (gs == es).all() and (gk == ek).all(), (tag, kind)


@pytest.mark.parametrize("typ", [GLOBAL, LOCAL, SEMI])
def test_full_matrix_patterns_beyond_1024_rows_stay_on_the_tuned_route(callers, typ):
    rng = np.random.default_rng(990 + typ)
    reads, quals, wins = [], [], []
    for i in range(40):
        L = int(rng.integers(800, 2600))
        w = rng.integers(0, 4, int(rng.integers(L, L + 900)), dtype=np.uint8)
        r = w[:L].copy()
        m = rng.random(L) < 0.05; r[m] = rng.integers(0, 4, int(m.sum()))
        reads.append(r); quals.append(np.zeros(L, np.uint8)); wins.append(w)
    b = Batch(reads, quals, wins, packed=True, on_device=True)
    th = np.where(rng.random(b.n) < 0.5, -(1 << 30), rng.integers(-40, 3000, b.n)).astype(np.int32)
    for kind, scheme, tag, thresholds in ((0, (2, -1, -2, -1), 1, th), (0, (2, -1, -2, -1), 0, None), (1, (2, -2, -3, -3), 1, None), (2, None, 1, None)):
        assert run_full(callers, b, 0, 0, kind, typ, tag, scheme or (0, -1, -1, -1), thresholds) == "tuned"
        es, ek = expect_full(b, kind, typ, tag, scheme, thresholds)
        gs, gk = b.results()
        assert (gs == es).all() and (gk == ek).all(), (kind, tag)


def test_packed_stream_outside_the_tuned_contract_falls_to_the_generic_lanes(callers):
    # a band the tuned kernels are not instantiated for
    for typ in (GLOBAL, LOCAL, SEMI):
        reads, quals, wins = make_jobs(9 + typ, 1500, band=9)
        b = Batch(reads, quals, wins, packed=True, on_device=True)
        assert run_banded(callers, b, 0, 0, 0, typ, 9, (2, -1, -2, -1)) == "generic"
        es, ek = O.batch_banded_gotoh_score(9, typ, (2, -1, -2, -1), b.hr, b.hw)
        gs, gk = b.results()
        assert (gs == es).all() and (gk == ek).all()


@pytest.mark.parametrize("typ", [GLOBAL, LOCAL, SEMI])
@pytest.mark.parametrize("kind,scheme", [(0, (2, -1, -2, -1)), (1, (2, -2, -4, -1)), (2, None), (3, None)])
@pytest.mark.parametrize("band", [15, 31])
def test_byte_strings_user_scheme_and_declined_jobs(callers, typ, kind, scheme, band):
    """uint8 strings (symbols up to 5: band 31's 2-bit window cache truncates them, as in the reference), per-base
    qualities, a user-defined scheme, Best2Sink, jobs whose init_context declines."""
    reads, quals, wins = make_jobs(300 + band + typ + kind, 2500, band=band, max_sym=5)
    b = Batch(reads, quals, wins, packed=False, on_device=True)
    assert run_banded(callers, b, 1, 0, kind, typ, band, scheme or (0, 0, 0, 0)) == "generic"
    es, ek = expect_banded(b, kind, typ, band, scheme)
    gs, gk = b.results()
    # a job whose init_context declines is still handed to output(), with the sink its context was built with -- as every per-job body of
    # the reference does (batched_banded_inl.h:53-75, batched_inl.h:58-63).  nvBowtie depends on it: a skipped anchor hit gets its worst
    # score this way (score_paired_inl.h:147-172), where leaving the output alone kept a stale score of the slot's previous hit
    declined = (np.arange(b.n) % 97) == 96
    assert (gs[declined] == -(1 << 30)).all() and (gk[declined] == 0xFFFFFFFF).all()          # BestSink() / Best2Sink(): Field_traits<int32>::min(), (-1, -1)
    assert (gs[~declined] == es[~declined]).all() and (gk[~declined] == ek[~declined]).all()


def run_full(L, b, strings, where, kind, typ, tag, scheme, thresholds):
    sc = np.array(scheme, dtype=np.int32)
    path = C.create_string_buffer(16)
    th = None
    if thresholds is not None:
        th = dev(thresholds) if where == 0 else torch.from_numpy(np.ascontiguousarray(thresholds).copy())
    rc = L.compat_full_score(strings, where, kind, typ, tag, sc.ctypes.data, b.n, b.ptr("ro"), b.ptr("r"), b.ptr("q"), b.longest_read,
                             b.ptr("wo"), b.ptr("w"), b.longest_win, C.c_void_p(th.data_ptr()) if th is not None else None,
                             C.c_void_p(b.score.data_ptr()), C.c_void_p(b.sink.data_ptr()), path)
    assert rc == 0, (rc, path.value)
    return path.value.decode()


def expect_full(b, kind, typ, tag, scheme, thresholds):
    if kind == 3:
        lut = np.array([-(2 + min(q, 40) // 10) for q in range(256)], dtype=np.int32)
        s, k, _ = O.batch_gotoh_score_qual(tag, typ, (2, -8, -3, -6, -2), lut, np.concatenate([b.cat_q, np.zeros(8, np.uint8)]), b.hr, b.hw, min_score=thresholds)
        return s, k
    if tag == 0:
        s, k, _ = O.batch_score_pattern_blocking(0 if kind == 0 else 1, typ, scheme if kind != 2 else (0, -1, -1, -1), b.hr, b.hw, min_score=thresholds)
        return s, k
    if kind == 0:
        s, k, _ = O.batch_gotoh_score(typ, scheme, b.hr, b.hw, min_score=thresholds)
        return s, k
    return O.batch_sw_score(0, typ, scheme if kind == 1 else (0, -1, -1, -1), b.hr, b.hw)


@pytest.mark.parametrize("typ", [GLOBAL, LOCAL, SEMI])
@pytest.mark.parametrize("tag", [0, 1])
@pytest.mark.parametrize("kind,scheme", [(0, (2, -1, -2, -1)), (1, (2, -2, -3, -3)), (2, None)])
def test_full_matrix_packed_stream(callers, typ, tag, kind, scheme):
    reads, quals, wins = make_jobs(500 + typ + 3 * tag + kind, 1200, max_read=120, full=True, short_text_every=10 ** 9)
    reads = [r if len(r) else np.array([1], np.uint8) for r in reads]          # the reference reads uninitialised cells for M == 0
    quals = [q if len(q) else np.array([0], np.uint8) for q in quals]
    b = Batch(reads, quals, wins, packed=True, on_device=True)
    th = None
    if kind == 0:
        rng = np.random.default_rng(5)
        th = np.where(rng.random(b.n) < 0.5, -(1 << 30), rng.integers(-40, 160, b.n)).astype(np.int32)
    assert run_full(callers, b, 0, 0, kind, typ, tag, scheme or (0, -1, -1, -1), th) == "tuned"
    es, ek = expect_full(b, kind, typ, tag, scheme, th)
    gs, gk = b.results()
    assert (gs == es).all() and (gk == ek).all()


@pytest.mark.parametrize("typ", [GLOBAL, LOCAL, SEMI])
@pytest.mark.parametrize("tag", [0, 1])
@pytest.mark.parametrize("kind,scheme", [(0, (2, -1, -2, -1)), (1, (2, -2, -4, -1)), (3, None)])
def test_full_matrix_generic_lanes(callers, typ, tag, kind, scheme):
    """byte strings, asymmetric linear gaps, the user scheme, thresholds with early exits: the generic full-matrix templates"""
    reads, quals, wins = make_jobs(700 + typ + 3 * tag + kind, 800, max_read=90, full=True, short_text_every=10 ** 9, max_sym=5)
    reads = [r if len(r) else np.array([1], np.uint8) for r in reads]
    quals = [q if len(q) else np.array([0], np.uint8) for q in quals]
    b = Batch(reads, quals, wins, packed=False, on_device=True)
    rng = np.random.default_rng(6)
    th = np.where(rng.random(b.n) < 0.5, -(1 << 30), rng.integers(-40, 120, b.n)).astype(np.int32) if kind != 1 else None
    assert run_full(callers, b, 1, 0, kind, typ, tag, scheme or (0, 0, 0, 0), th) == "generic"
    es, ek = expect_full(b, kind, typ, tag, scheme, th)
    gs, gk = b.results()
    declined = (np.arange(b.n) % 97) == 96
    assert (gs[~declined] == es[~declined]).all() and (gk[~declined] == ek[~declined]).all()


def test_full_matrix_long_patterns_and_host(callers):
    """patterns to 1024 symbols run on the register-resident sweep (16 rows per lane), longer ones on the striped one; HostThreadScheduler on the host"""
    rng = np.random.default_rng(8)
    for lo, hi, path in ((400, 900, "tuned"), (1030, 1300, "tuned")):
        reads = [rng.integers(0, 4, int(rng.integers(lo, hi)), dtype=np.uint8) for _ in range(60)]
        wins = [np.concatenate([rng.integers(0, 4, 30, dtype=np.uint8), r, rng.integers(0, 4, 30, dtype=np.uint8)]) for r in reads]
        quals = [np.zeros(len(r), np.uint8) for r in reads]
        b = Batch(reads, quals, wins, packed=True, on_device=True)
        assert run_full(callers, b, 0, 0, 0, LOCAL, 0, (2, -1, -2, -1), None) == path
        es, ek, _ = O.batch_score_pattern_blocking(0, LOCAL, (2, -1, -2, -1), b.hr, b.hw)
        gs, gk = b.results()
        assert (gs == es).all() and (gk == ek).all()
    bh = Batch(reads, quals, wins, packed=True, on_device=False)
    assert run_full(callers, bh, 0, 1, 0, SEMI, 1, (2, -1, -2, -1), None) == "host"
    es, ek, _ = O.batch_gotoh_score(SEMI, (2, -1, -2, -1), bh.hr, bh.hw)
    gs, gk = bh.results()
    assert (gs == es).all() and (gk == ek).all()


def test_per_thread_function_in_a_user_kernel(callers):
    """aln::banded_alignment_score<7>(aligner, pattern, text, min_score) called from a user kernel on byte strings"""
    reads, quals, wins = make_jobs(77, 3000, max_read=60, band=7, max_sym=5)
    b = Batch(reads, quals, wins, packed=False, on_device=True)
    sc = np.array((2, -1, -1, -1), dtype=np.int32)
    assert callers.compat_per_thread_score(sc.ctypes.data, b.n, b.ptr("ro"), b.ptr("r"), b.ptr("wo"), b.ptr("w"), C.c_void_p(b.score.data_ptr())) == 0
    es, _ = O.batch_banded_gotoh_score(7, SEMI, (2, -1, -1, -1), b.hr, b.hw)
    assert (b.score.cpu().numpy() == es).all()


# ---------------------------------------------------------------------------- traceback streams
@pytest.mark.parametrize("typ", [GLOBAL, LOCAL, SEMI])
@pytest.mark.parametrize("band", [15, 31, 0])
def test_traceback_stream_with_a_user_backtracer(callers, typ, band):
    """BatchedBandedAlignmentTraceback / BatchedAlignmentTraceback over a stream written to the reference's traceback concept
    (context {min_score, backtracer, alignment}): the caller's own backtracer must receive clip / push / clip exactly as the
    reference's per-job body issues them, so the run-length CIGAR it forms, and the Alignment handed to output(), equal the
    oracle's; declined jobs are output untouched; jobs whose text is shorter than the pattern get no backtracer call."""
    callers.compat_traceback.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    reads, quals, wins = make_jobs(4100 + 10 * typ + band, 900, max_read=120, band=band or 31, max_sym=3, full=(band == 0))
    b = Batch(reads, quals, wins, packed=True, on_device=True)
    stride = 200
    for kind, scheme in ((0, (2, -2, -4, -1)), (1, (2, -1, -1, -1)), (2, (0, -1, -1, -1))):
        sc = np.array(scheme, dtype=np.int32)
        score = dev(np.full(b.n, 12345, np.int32)); sink = dev(np.full((b.n, 2), 777, np.int32)); source = dev(np.full((b.n, 2), 777, np.int32))
        cigar = dev(np.zeros((b.n, stride), np.int16)); clen = dev(np.full(b.n, 999, np.int32))
        rc = callers.compat_traceback(kind, typ, band, sc.ctypes.data, b.n, b.ptr("ro"), b.ptr("r"), b.longest_read, b.ptr("wo"), b.ptr("w"), b.longest_win,
                                      C.c_void_p(score.data_ptr()), C.c_void_p(sink.data_ptr()), C.c_void_p(source.data_ptr()), C.c_void_p(cigar.data_ptr()), stride,
                                      C.c_void_p(clen.data_ptr()))
        assert rc == 0
        if kind == 0:
            exp = O.batch_banded_gotoh_traceback(band, typ, scheme, b.hr, b.hw, stride) if band else O.batch_gotoh_traceback(typ, scheme, b.hr, b.hw, stride)
        else:
            exp = O.batch_sw_traceback(band, typ, scheme, b.hr, b.hw, stride)
        gs, gk, gsrc = score.cpu().numpy(), sink.cpu().numpy().view(np.uint32), source.cpu().numpy().view(np.uint32)
        gc, gl = cigar.cpu().numpy().view(np.uint16), clen.cpu().numpy().view(np.uint32)
        declined = (np.arange(b.n) % 53) == 52
        # (an empty pattern makes the reference's full-matrix pattern-blocking pass read uninitialised cells -- oracle/nvbio_oracle.c,
        # score_pattern_blocking: nothing to compare there)
        live = ~declined & ((np.diff(b.ro) > 0) | (band != 0))
        assert (gs[declined] == -77).all() and (gk[declined] == 7).all() and (gl[declined] == 0).all()
        assert (gs[live] == exp["score"][live]).all(), (typ, band, kind)
        assert (gk[live] == exp["sink"][live]).all() and (gsrc[live] == exp["source"][live]).all(), (typ, band, kind)
        assert (gl[live] == exp["cigar_len"][live]).all(), (typ, band, kind)
        mask = (np.arange(stride)[None, :] < exp["cigar_len"][:, None]) & live[:, None]
        assert ((gc == exp["cigar"]) | ~mask).all(), (typ, band, kind)
        assert int(exp["cigar_len"][live].max()) < stride


# ---------------------------------------------------------------------------------------------------------------------
# generic tracebacks: byte strings, user schemes, asymmetric linear gaps -- streams the tuned kernels do not take
# ---------------------------------------------------------------------------------------------------------------------
PHRED_LUT = np.array([-(2 + min(q, 40) // 10) for q in range(256)], dtype=np.int32)


def run_byte_tracebacks(L, where, typ, band, n=400, seed=0):
    """ByteTracebackStream of tests/compat/aln_callers.hip through Batched(Banded)AlignmentTraceback; where 0 = device scheduler,
    1 = HostThreadScheduler.  Returns the number of jobs compared."""
    L.compat_traceback_bytes.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p,
                                         C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_char_p]
    reads, quals, wins = make_jobs(8100 + 10 * typ + band + seed, n, max_read=90, band=band or 31, max_sym=5, full=(band == 0))
    b = Batch(reads, quals, wins, packed=False, on_device=(where == 0))
    mk = dev if where == 0 else (lambda a: torch.from_numpy(np.ascontiguousarray(a).copy()))
    stride = 200
    compared = 0
    for kind, scheme in ((0, (2, -2, -4, -1)), (1, (2, -1, -2, -3)), (1, (1, -1, -3, -1)), (2, (0, -1, -1, -1)), (3, None)):
        sc = np.array(scheme if scheme else (0, 0, 0, 0), dtype=np.int32)
        score = mk(np.full(b.n, 12345, np.int32)); sink = mk(np.full((b.n, 2), 777, np.int32)); source = mk(np.full((b.n, 2), 777, np.int32))
        cigar = mk(np.zeros((b.n, stride), np.int16)); clen = mk(np.full(b.n, 999, np.int32))
        path = C.create_string_buffer(16)
        rc = L.compat_traceback_bytes(where, kind, typ, band, sc.ctypes.data, b.n, b.ptr("ro"), b.ptr("r"), b.ptr("q"), b.longest_read, b.ptr("wo"), b.ptr("w"), b.longest_win,
                                      C.c_void_p(score.data_ptr()), C.c_void_p(sink.data_ptr()), C.c_void_p(source.data_ptr()), C.c_void_p(cigar.data_ptr()), stride,
                                      C.c_void_p(clen.data_ptr()), path)
        assert rc == 0
        assert path.value == (b"generic" if where == 0 else b"host")
        qbuf = np.concatenate([b.cat_q, np.zeros(8, np.uint8)])
        if kind == 0:
            exp = O.batch_banded_gotoh_traceback(band, typ, scheme, b.hr, b.hw, stride) if band else O.batch_gotoh_traceback(typ, scheme, b.hr, b.hw, stride)
        elif kind == 3:
            s5 = (2, -8, -3, -6, -2)
            exp = O.batch_banded_gotoh_traceback(band, typ, s5, b.hr, b.hw, stride, PHRED_LUT, qbuf) if band else O.batch_gotoh_traceback(typ, s5, b.hr, b.hw, stride, PHRED_LUT, qbuf)
        else:
            exp = O.batch_sw_traceback(band, typ, scheme, b.hr, b.hw, stride)
        gs, gk, gsrc = score.cpu().numpy(), sink.cpu().numpy().view(np.uint32), source.cpu().numpy().view(np.uint32)
        gc, gl = cigar.cpu().numpy().view(np.uint16), clen.cpu().numpy().view(np.uint32)
        declined = (np.arange(b.n) % 53) == 52
        live = ~declined & ((np.diff(b.ro) > 0) | (band != 0))
        assert (gs[declined] == -77).all() and (gk[declined] == 7).all() and (gl[declined] == 0).all()
        assert (gs[live] == exp["score"][live]).all(), (typ, band, kind)
        assert (gk[live] == exp["sink"][live]).all() and (gsrc[live] == exp["source"][live]).all(), (typ, band, kind)
        assert (gl[live] == exp["cigar_len"][live]).all(), (typ, band, kind)
        mask = (np.arange(stride)[None, :] < exp["cigar_len"][:, None]) & live[:, None]
        assert ((gc == exp["cigar"]) | ~mask).all(), (typ, band, kind)
        compared += int(live.sum())
    return compared


@pytest.mark.parametrize("typ", [GLOBAL, LOCAL, SEMI])
@pytest.mark.parametrize("band", [15, 31, 0])
def test_generic_traceback_streams_on_the_device(callers, typ, band):
    assert run_byte_tracebacks(callers, 0, typ, band, n=600) > 2000
