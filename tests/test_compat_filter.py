"""FMIndexFilter behind the reference's own template: tests/compat/filter_callers.hip is written to the shape
of examples/fmmap/fmmap.cu:92-97, 293-367 -- FMIndexFilterDevice<fm_index_type> over the production interleaved index, rank the
seed string-set, locate batches of hits, hits -> diagonals, read / genome infix sets, batch_banded_alignment_score<31> with the
edit-distance aligner into BestSink<int16> -- and must agree with the oracle bit for bit on the tuned route (reference layout and
line-native index), on the generic route (64-bit coordinates, separate arrays) and on the host filter."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import pyoracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "tests", "compat", "libfilter_callers.so")


class FmmapArgs(C.Structure):
    _fields_ = [("n", C.c_uint32), ("primary", C.c_uint32), ("L2", C.c_void_p), ("bwt_occ", C.c_void_p), ("ssa", C.c_void_p), ("count_table", C.c_void_p),
                ("genome_words", C.c_void_p), ("genome_len", C.c_uint32),
                ("read_words", C.c_void_p), ("read_index", C.c_void_p), ("n_reads", C.c_uint32), ("max_read_len", C.c_uint32),
                ("seeds", C.c_void_p), ("n_seeds", C.c_uint32), ("batch_size", C.c_uint32), ("line_native", C.c_uint32),
                ("out_n_hits", C.c_void_p), ("out_diagonals", C.c_void_p), ("out_scores", C.c_void_p), ("out_sinks", C.c_void_p), ("out_capacity", C.c_uint32),
                ("out_ranges", C.c_void_p), ("out_ranks", C.c_void_p)]


@pytest.fixture(scope="module")
def lib():
    assert os.path.exists(LIB), "build with python -c 'import __graft_entry__ as g; g.build()'"
    L = C.CDLL(LIB)
    vp, u32, u64 = C.c_void_p, C.c_uint32, C.c_uint64
    L.compat_fmmap.argtypes = [C.POINTER(FmmapArgs), C.c_char_p, C.c_char_p, C.c_char_p]
    L.compat_filter64.argtypes = [u64, u64, vp, vp, vp, vp, vp, vp, vp, u32, vp, vp, vp, vp, u32, C.c_char_p]
    L.compat_filter_host.argtypes = [u32, u32, vp, vp, vp, vp, vp, vp, u32, vp, vp, vp, vp, u32]
    return L


@pytest.fixture(scope="module")
def world():
    rng = np.random.default_rng(23)
    n = 80003
    text = rng.integers(0, 4, n, dtype=np.uint8)
    text[5000:5900] = np.tile(np.array([0, 1, 2, 2], np.uint8), 225)          # a repeat: seeds with many occurrences
    host = O.FMIndex(text)
    reads, origin = [], []
    for _ in range(900):
        L = int(rng.integers(50, 101))
        o = int(rng.integers(0, n - L))
        r = text[o:o + L].copy()
        for j in rng.integers(0, L, int(rng.integers(0, 4))):
            r[j] = (r[j] + 1 + rng.integers(0, 3)) & 3
        reads.append(r); origin.append(o)
    for k in range(30):                                                          # reads inside the repeat
        o = 5000 + int(rng.integers(0, 700)); reads.append(text[o:o + 80].copy()); origin.append(o)
    rs = O.StringSet.from_lists(reads, 2, True)
    index = np.concatenate([rs.begin, [rs.begin[-1] + rs.length[-1]]]).astype(np.uint32)
    seeds = []
    for i, r in enumerate(reads):
        for b in range(0, len(r) - 20 + 1, 10):
            seeds.append((i, b, b + 20))
    seeds = np.array(seeds, dtype=np.uint32)
    seed_set = O.StringSet(rs.words, 2, True, (index[seeds[:, 0]] + seeds[:, 1]).astype(np.uint64), (seeds[:, 2] - seeds[:, 1]).astype(np.uint32))
    return dict(text=text, n=n, host=host, reads=reads, rs=rs, index=index, seeds=seeds, seed_set=seed_set, gw=O.pack(text, 2, True, pad_words=4))


def expected(w, capacity):
    host = w["host"]
    total, ranges, slots = host.filter_rank(w["seed_set"])
    cap = min(total, capacity)
    hits = host.filter_locate(ranges, slots, 0, cap)                               # (text pos, seed id)
    sd = w["seeds"][hits[:, 1]]
    diag = np.stack([(hits[:, 0] - sd[:, 1]).astype(np.uint32), sd[:, 0]], axis=1).astype(np.uint32)
    return total, ranges, slots, hits, diag


def test_host_filter_matches_oracle(lib, world):
    w = world
    host = w["host"]
    n = w["n"]
    nb = (n + 63) // 64 + 1
    occ = np.zeros(nb * 4, dtype=np.uint32)
    cum = np.zeros((n + 1, 4), dtype=np.int64)
    for c in range(4):
        cum[1:, c] = np.cumsum(host.bwt == c)
    for k in range(nb):
        occ[4 * k: 4 * k + 4] = cum[min(64 * k, n)]
    bw = np.concatenate([O.pack(host.bwt, 2, True, pad_words=0), np.zeros(8, np.uint32)])
    total, ranges, slots, hits, _ = expected(w, 60000)
    sr = np.stack([w["seed_set"].begin.astype(np.uint32), (w["seed_set"].begin + w["seed_set"].length).astype(np.uint32)], axis=1).astype(np.uint32)
    ns = len(sr)
    o_n = np.zeros(1, np.uint64); o_ranges = np.zeros((ns, 2), np.uint32); o_ranks = np.zeros(ns, np.uint64); o_hits = np.zeros((60000, 2), np.uint32)
    p = lambda a: a.ctypes.data
    L2 = host.L2.astype(np.uint32)
    assert lib.compat_filter_host(n, host.primary, p(L2), p(bw), p(occ), p(host.ssa), p(w["rs"].words), p(sr), ns, p(o_n), p(o_ranges), p(o_ranks), p(o_hits), 60000) == 0
    assert int(o_n[0]) == total
    assert (o_ranges == ranges).all() and (o_ranks == slots).all()
    assert (o_hits[:len(hits)] == hits).all()


@pytest.mark.gpu
@pytest.mark.parametrize("line_native", [0, 1])
def test_fmmap_shaped_caller_runs_tuned_and_matches_oracle(lib, world, line_native):
    import torch
    w = world
    host = w["host"]
    dev = torch.device("cuda:0")
    keep = []

    def d(x):
        t = torch.from_numpy(np.ascontiguousarray(x).view(np.int32 if x.dtype == np.uint32 else x.dtype)).to(dev); keep.append(t); return t.data_ptr()

    cap = 70000
    total, ranges, slots, hits, diag = expected(w, cap)
    assert total > 20000
    a = FmmapArgs()
    a.n, a.primary = w["n"], host.primary
    a.L2, a.bwt_occ, a.ssa, a.count_table = d(host.L2.astype(np.uint32)), d(host.bwt_occ), d(host.ssa), d(np.zeros(256, np.uint32))
    a.genome_words, a.genome_len = d(w["gw"]), w["n"]
    a.read_words, a.read_index, a.n_reads, a.max_read_len = d(w["rs"].words), d(w["index"]), len(w["reads"]), 100
    a.seeds, a.n_seeds = d(w["seeds"]), len(w["seeds"])
    a.batch_size, a.line_native = 16 * 1024, line_native
    n_hits = np.zeros(1, np.uint64)
    a.out_n_hits = n_hits.ctypes.data
    o_diag = torch.zeros((cap, 2), dtype=torch.int32, device=dev); o_sc = torch.zeros(cap, dtype=torch.int16, device=dev); o_sk = torch.zeros((cap, 2), dtype=torch.int32, device=dev)
    o_ranges = torch.zeros((a.n_seeds, 2), dtype=torch.int32, device=dev); o_ranks = torch.zeros(a.n_seeds, dtype=torch.int64, device=dev)
    a.out_diagonals, a.out_scores, a.out_sinks, a.out_capacity = o_diag.data_ptr(), o_sc.data_ptr(), o_sk.data_ptr(), cap
    a.out_ranges, a.out_ranks = o_ranges.data_ptr(), o_ranks.data_ptr()
    rp, lp, sp = C.create_string_buffer(16), C.create_string_buffer(16), C.create_string_buffer(16)
    assert lib.compat_fmmap(C.byref(a), rp, lp, sp) == 0
    # rank / locate on the gfx950 kernels; the score stream asks for the bit-vector edit distance (MyersTag), its own algorithm: generic lanes
    assert rp.value == b"tuned" and lp.value == b"tuned" and sp.value == b"generic"
    assert int(n_hits[0]) == total
    assert (o_ranges.cpu().numpy().view(np.uint32) == ranges).all()
    assert (o_ranks.cpu().numpy().view(np.uint64) == slots).all()
    m = len(hits)
    assert (o_diag.cpu().numpy().view(np.uint32)[:m] == diag).all()
    # the banded edit distance of every hit's read against its genome window
    rl = (w["index"][diag[:, 1] + 1] - w["index"][diag[:, 1]]).astype(np.uint32)
    gb = np.where(diag[:, 0].astype(np.int64) > 15, diag[:, 0].astype(np.int64) - 15, 0)
    gb = np.where(diag[:, 0] > np.uint32(0x7FFFFFFF), diag[:, 0].astype(np.int64) - 15, gb)     # wrapped diagonals (seed at a read offset beyond its text position)
    gb = gb.astype(np.uint32)
    ge = np.minimum(gb.astype(np.int64) + rl + 31, w["n"])
    tl = np.maximum(ge - gb.astype(np.int64), 0).astype(np.uint32)
    ps = O.StringSet(w["rs"].words, 2, True, w["index"][diag[:, 1]].astype(np.uint64), rl)
    ts = O.StringSet(w["gw"], 2, True, gb.astype(np.uint64), tl)
    sane = diag[:, 0] < np.uint32(w["n"])
    # (the batch function's "no threshold" narrows to an int16 0 inside the bit-vector scorer: exact occurrences report 0, everything else keeps
    # the BestSink<int16> it started with -- myers_banded_inl.h:243, batched_inl.h:946)
    es, ek = O.batch_banded_myers_score(31, O.SEMI_GLOBAL, 5, ps, ts, sink_bits=16)
    gs = o_sc.cpu().numpy()[:m].astype(np.int32)
    gk = o_sk.cpu().numpy().view(np.uint32)[:m]
    ok = ek[:, 0] != 0xFFFFFFFF
    assert (gs[sane] == es[sane]).all() and (gk[sane] == ek[sane]).all()
    assert (gs[sane & ~ok] == -32768).all()
    assert (sane & ok).sum() > 1000 and (es[sane & ok] == 0).all()


@pytest.mark.gpu
def test_filter_generic_route_64bit(lib, world):
    import torch
    w = world
    host = w["host"]
    n = w["n"]
    dev = torch.device("cuda:0")
    nb = (n + 63) // 64 + 1
    occ = np.zeros(nb * 4, dtype=np.uint64)
    cum = np.zeros((n + 1, 4), dtype=np.int64)
    for c in range(4):
        cum[1:, c] = np.cumsum(host.bwt == c)
    for k in range(nb):
        occ[4 * k: 4 * k + 4] = cum[min(64 * k, n)]
    bw = np.concatenate([O.pack(host.bwt, 2, True, pad_words=0), np.zeros(8, np.uint32)])
    bw = bw[: (bw.size // 2) * 2]
    bw64 = (bw[0::2].astype(np.uint64) << np.uint64(32)) | bw[1::2].astype(np.uint64)
    ssa64 = host.ssa.astype(np.uint64); ssa64[0] = np.uint64(0xFFFFFFFFFFFFFFFF)
    keep = []

    def d(x):
        t = torch.from_numpy(np.ascontiguousarray(x).view(np.int64 if x.dtype == np.uint64 else np.int32 if x.dtype == np.uint32 else x.dtype)).to(dev); keep.append(t); return t.data_ptr()

    total, ranges, slots, hits, _ = expected(w, 50000)
    sr = np.stack([w["seed_set"].begin.astype(np.uint32), (w["seed_set"].begin + w["seed_set"].length).astype(np.uint32)], axis=1).astype(np.uint32)
    ns = len(sr)
    o_n = np.zeros(1, np.uint64)
    o_ranges = torch.zeros((ns, 2), dtype=torch.int64, device=dev); o_ranks = torch.zeros(ns, dtype=torch.int64, device=dev); o_hits = torch.zeros((50000, 2), dtype=torch.int64, device=dev)
    path = C.create_string_buffer(16)
    assert lib.compat_filter64(n, host.primary, d(host.L2.astype(np.uint64)), d(bw64), d(occ), d(np.zeros(256, np.uint32)), d(ssa64), d(w["rs"].words), d(sr), ns,
                               o_n.ctypes.data, o_ranges.data_ptr(), o_ranks.data_ptr(), o_hits.data_ptr(), 50000, path) == 0
    assert path.value == b"generic"
    assert int(o_n[0]) == total
    got = o_ranges.cpu().numpy().view(np.uint64)
    nonempty = ranges[:, 0] <= ranges[:, 1]
    assert (got[nonempty] == ranges[nonempty].astype(np.uint64)).all()
    assert (o_ranks.cpu().numpy().view(np.uint64) == slots).all()
    assert (o_hits.cpu().numpy().view(np.uint64)[:len(hits)] == hits.astype(np.uint64)).all()
