"""nvBowtie's all-mapping mode (Aligner::all / score_all, aligner_all.h) through the C-ABI stages vs the independent numpy driver over
the oracle: the accepted alignments of every read (all rows of all its seed hit ranges), their tracebacks, MD strings and finished
alignment words, in the same (batch, read, strand, position) order."""
import numpy as np
import pytest
import torch

import nvbio_amd as nvb
from nvbio_amd import aligner as A, select as S, workloads as W
from oracle import pyoracle as O
from tests import oracle_driver as OD

pytestmark = pytest.mark.gpu


def _genome(rng, n=1 << 16):
    text = rng.integers(0, 4, n, dtype=np.uint8)
    text[7000:7800] = np.tile(np.array([0, 1, 2], dtype=np.uint8), 267)[:800]       # a tandem repeat: wide SA ranges
    for k in range(4):                                                               # a 300 bp element, 5 copies with a few differences
        c = text[20000:20300].copy()
        c[rng.integers(0, 300, 3)] = rng.integers(0, 4, 3)
        text[30000 + 7000 * k: 30300 + 7000 * k] = c
    return text


def _reads(rng, text, n, ragged):
    reads, quals = [], []
    for i in range(n):
        L = int(rng.integers(30, 131)) if ragged else 100
        where = i % 6
        p = int(rng.integers(20000, 20300 - 30)) if where == 0 else int(rng.integers(7000, 7700)) if where == 1 else \
            (0 if i % 31 == 2 else text.size - L if i % 31 == 3 else int(rng.integers(0, text.size - L)))
        p = min(p, text.size - L)
        r = text[p:p + L].copy()
        for j in rng.integers(0, L, [0, 1, 2, 4][i % 4]):
            r[j] = (r[j] + 1 + rng.integers(0, 3)) & 3
        if i % 10 == 0:
            d = int(rng.integers(8, L - 8)); r = np.concatenate([r[:d], r[d + 1:], rng.integers(0, 4, 1, dtype=np.uint8)])
        if i % 13 == 0:
            r[int(rng.integers(0, L))] = 4
        if i % 2:
            r = np.where(r > 3, r, 3 - r)[::-1].copy()
        if i % 23 == 0:
            r = rng.integers(0, 4, L, dtype=np.uint8)
        reads.append(r); quals.append(rng.integers(2, 42, r.size).astype(np.uint8))
    return reads, quals


CONFIGS = {
    "default": (dict(), False),
    "small_batches": (dict(batch_size=1500), False),            # the same placement in two batches is reported twice, as in the reference
    "tiny_batches": (dict(batch_size=97), False),
    "local": (dict(local=True, seed_len=20, seed_freq=(2, 1.0, 0.75)), False),
    "one_mismatch_seeds": (dict(allow_sub=1, seed_len=20, max_hits=40, batch_size=4000), False),
    "ragged": (dict(batch_size=3000), True),
    "few_hits": (dict(max_hits=4), False),
    "sequences": (dict(batch_size=2500), False),                # a multi-sequence reference: seeds straddling a boundary are dropped
}


@pytest.mark.parametrize("config", sorted(CONFIGS))
def test_all_mapping_matches_oracle(cuda, config):
    rng = np.random.default_rng(4242)
    text = _genome(rng)
    host, rhost = O.FMIndex(text), O.FMIndex(text[::-1].copy())
    fmi, rfmi = nvb.FMIndexDevice.from_host(host, cuda), nvb.FMIndexDevice.from_host(rhost, cuda)
    kw, ragged = CONFIGS[config]
    params = A.Params(**kw)
    n = 600
    reads, quals = _reads(rng, text, n, ragged)
    scheme = nvb.SmithWatermanScoringScheme.local() if params.local else nvb.SmithWatermanScoringScheme()
    gw = W._pack_chunked(torch.from_numpy(text), 2, True)
    seq_index = [0, 9000, 20100, 41000, text.size] if config == "sequences" else None
    e = OD.all_mapping(host, rhost, reads, gw.numpy().view(np.uint32), text.size, params, scheme, 1 if params.local else 2, read_quals=quals,
                       cigar_stride=96, sequence_index=seq_index)
    if ragged:
        index = np.zeros(n + 1, np.int64); index[1:] = np.cumsum([r.size for r in reads])
        batch = A.ReadBatch.from_ragged(torch.from_numpy(np.concatenate(reads)).to(cuda), torch.from_numpy(index).to(cuda), torch.from_numpy(np.concatenate(quals)).to(cuda))
    else:
        batch = A.ReadBatch.from_matrix(torch.from_numpy(np.stack(reads)).to(cuda), quals=torch.from_numpy(np.stack(quals)))
    r = A.all_mapping(fmi, rfmi, batch, gw.to(cuda), text.size, params, scheme, cigar_stride=96, sequence_index=seq_index)
    torch.cuda.synchronize()
    assert r["stats"] == e["stats"], (r["stats"], e["stats"])
    m = e["read_id"].size
    assert m > n and r["read_id"].numel() == m
    assert (r["read_id"].cpu().numpy().view(np.uint32) == e["read_id"]).all()
    assert (r["alignments_scored"].cpu().numpy().view(np.uint64) == e["alignments_scored"]).all()
    assert (r["alignments"].cpu().numpy().view(np.uint64) == e["alignments"]).all()
    tb = e["tb"]
    assert (r["cigar_len"].cpu().numpy().view(np.uint32) == tb["cigar_len"]).all()
    assert (r["cigar"].cpu().numpy().view(np.uint16) == tb["cigar"][:m]).all()
    assert (r["source"].cpu().numpy().view(np.uint32) == tb["source"]).all() and (r["sink"].cpu().numpy().view(np.uint32) == tb["sink"]).all()
    assert (r["mds_len"].cpu().numpy().view(np.uint32) == e["mds_len"]).all()
    mk = np.arange(256)[None, :] < np.minimum(e["mds_len"], 256)[:, None]
    assert ((r["mds"].cpu().numpy() == e["mds"]) | ~mk).all()
    # what the mode is for: reads from the 5-copy element report several placements; within a batch no placement twice
    per_read = np.bincount(e["read_id"], minlength=n)
    assert per_read.max() >= 4
    if config == "default":
        key = (e["read_id"].astype(np.uint64) << np.uint64(34)) | (e["alignments_scored"] >> np.uint64(32)) | (((e["alignments_scored"] >> np.uint64(28)) & np.uint64(1)) << np.uint64(33))
        assert np.unique(key).size == m
    if config == "sequences":
        assert e["stats"]["unique"] < OD.all_mapping(host, rhost, reads, gw.numpy().view(np.uint32), text.size, params, scheme, 2, read_quals=quals,
                                                     cigar_stride=96)["stats"]["unique"]


def test_all_mapping_stage_kernels(cuda):
    """gather_ranges / select_all decode the global hit numbering of random deques exactly like a direct enumeration."""
    rng = np.random.default_rng(5)
    n, stride = 300, 12
    counts = rng.integers(0, stride + 1, n).astype(np.uint32)
    counts[::17] = 0
    hits = np.zeros((n, stride), np.uint64)
    exp = []
    for r in range(n):
        for k in range(int(counts[r])):
            begin, delta, pos, rc, idir = int(rng.integers(0, 1 << 30)), int(rng.integers(1, 40)), int(rng.integers(0, 1000)), int(rng.integers(0, 2)), int(rng.integers(0, 2))
            hits[r, k] = begin | ((delta | (pos << 20) | (rc << 30) | (idir << 31)) << 32)
            for j in range(delta):
                exp.append((begin + j, pos | (idir << 12) | (rc << 13), r))
    exp = np.array(exp, np.int64)
    d_hits = torch.from_numpy(hits.view(np.int64)).to(cuda)
    d_counts = torch.from_numpy(counts.view(np.int32)).to(cuda)
    scan = torch.cumsum(d_counts.to(torch.int64), 0).to(torch.int32)
    ranges = S.gather_ranges(d_hits, d_counts, scan, int(counts.sum()))
    rs = torch.cumsum(ranges, 0)
    assert int(rs[-1]) == exp.shape[0]
    for off, cnt in ((0, exp.shape[0]), (1234, 777), (exp.shape[0] - 5, 5)):
        loc, seed, rid = S.select_all(off, cnt, d_hits, scan, rs)
        torch.cuda.synchronize()
        assert (loc.cpu().numpy().view(np.uint32) == exp[off:off + cnt, 0]).all()
        assert (seed.cpu().numpy() == exp[off:off + cnt, 1]).all() and (rid.cpu().numpy() == exp[off:off + cnt, 2]).all()


@pytest.mark.parametrize("n", [1, 2, 257, 5000, 120_000, 1_048_576])
def test_sort_hits_pingpong(cuda, n):
    """nvbio_hip_sort_hits_pingpong: the (read, strand, position) index and the first-of-run flags of sort_hits -- against numpy's stable sort of
    SortingKeys (aligner_all.h:229-247) --, and the stale index the reference's mark_straddling reads (aligner_all.h:520): whatever half of the
    ping-pong buffer it is, it holds every hit exactly once."""
    rng = np.random.default_rng(9000 + n)
    rid = rng.integers(0, max(n // 7, 1), n).astype(np.uint32)
    loc = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    if n > 16:                       # runs of equal keys: the dedup flags have something to drop
        loc[n // 2:] = loc[:n - n // 2]; rid[n // 2:] = rid[:n - n // 2]
    seed = (rng.integers(0, 2, n).astype(np.uint32) << 13) | rng.integers(0, 100, n).astype(np.uint32)
    if n > 16:
        seed[n // 2:] = (seed[:n - n // 2] & (1 << 13)) | rng.integers(0, 100, n - n // 2).astype(np.uint32)
    key = loc.astype(np.uint64) + (rid.astype(np.uint64) << np.uint64(33)) + (((seed >> 13) & 1).astype(np.uint64) << np.uint64(32))
    exp_idx = np.argsort(key, kind="stable")
    sk = key[exp_idx]
    exp_first = np.ones(n, bool); exp_first[1:] = sk[1:] != sk[:-1]
    t = lambda a: torch.from_numpy(a.view(np.int32)).to(cuda)
    idx, first, stale = S.sort_hits_pingpong(t(rid), t(loc), t(seed))
    idx0, first0 = S.sort_hits(t(rid), t(loc), t(seed))
    torch.cuda.synchronize()
    assert (idx.cpu().numpy() == exp_idx).all() and (first.cpu().numpy().astype(bool) == exp_first).all()
    assert torch.equal(idx, idx0) and torch.equal(first, first0)
    st = stale.cpu().numpy()
    assert (np.sort(st) == np.arange(n)).all()
    # which half: up to ~10^5 hits both sorts of this image's radix sort end in the same one (the stale pointer sees the final index); at a full
    # batch of 2^20 hits they do not, and the half holds the last pass but one -- still every hit once, and the same in both programs
    if n <= 5000:
        assert (st == exp_idx).all()


@pytest.mark.parametrize("config", ["default", "small_batches", "local", "one_mismatch_seeds", "sequences"])
def test_cxx_all_mapping_driver_matches_oracle(cuda, config):
    """The C++ host driver (include/nvbio_hip/aligner.h: Aligner::all), called through tests/cxx/aligner_shim.cpp on device-resident
    inputs, vs the numpy driver over the oracle (cigar stride 64, constant Q30)."""
    import ctypes as C, os
    from nvbio_amd import pipeline as P
    from tests.test_select_gpu import _ShimParams
    shim_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cxx", "libaligner_shim.so")
    if not os.path.exists(shim_path):
        pytest.fail("tests/cxx/libaligner_shim.so is missing: run `python __graft_entry__.py`")
    shim = C.CDLL(shim_path)
    rng = np.random.default_rng(4242)
    text = _genome(rng)
    host, rhost = O.FMIndex(text), O.FMIndex(text[::-1].copy())
    fmi, rfmi = nvb.FMIndexDevice.from_host(host, cuda), nvb.FMIndexDevice.from_host(rhost, cuda)
    kw, _ = CONFIGS[config]
    params = A.Params(**kw)
    n, L = 600, 100
    reads, _ = _reads(rng, text, n, False)
    sym = np.stack(reads)
    scheme = nvb.SmithWatermanScoringScheme.local() if params.local else nvb.SmithWatermanScoringScheme()
    gw = W._pack_chunked(torch.from_numpy(text), 2, True)
    seq_index = [0, 9000, 20100, 41000, text.size] if config == "sequences" else [0, text.size]
    e = OD.all_mapping(host, rhost, reads, gw.numpy().view(np.uint32), text.size, params, scheme, 1 if params.local else 2, cigar_stride=64, sequence_index=seq_index)

    reads_rev, fwrc = P.pack_read_streams(torch.from_numpy(sym).to(cuda))
    quals = torch.full((2 * n * L + 8,), 30, dtype=torch.uint8, device=cuda)
    d_gw = gw.to(cuda)
    sp = _ShimParams(int(params.local), 0, 0, params.max_effort_init, params.max_effort, params.min_ext, params.max_ext,
                     params.max_reseed, params.rep_seeds, params.max_hits, params.allow_sub, params.subseed_len, params.seed_len, params.seed_freq[0],
                     params.min_read_len, params.max_dist, 0, params.batch_size, params.hits_stride or 0,
                     params.seed_freq[1], params.seed_freq[2], scheme.m_match, scheme.m_score_min[0], scheme.m_score_min[1], scheme.m_score_min[2], 1)
    m = e["read_id"].size
    cap = m + 16
    rid = np.zeros(cap, np.uint32); aln = np.zeros(cap, np.uint64); scored = np.zeros(cap, np.uint64); cigar = np.zeros((cap, 64), np.uint16)
    cigar_len = np.zeros(cap, np.uint32); source = np.zeros((cap, 2), np.uint32); sink = np.zeros((cap, 2), np.uint32)
    mds = np.zeros((cap, 256), np.uint8); mds_len = np.zeros(cap, np.uint32); stats = np.zeros(3, np.uint64); count = np.zeros(1, np.uint64)
    si = np.asarray(seq_index, np.uint32)
    fs, rs = fmi.struct(), rfmi.struct()
    vp = lambda t: C.c_void_p(t.data_ptr())
    hp = lambda a: a.ctypes.data_as(C.c_void_p)
    torch.cuda.synchronize()
    rc = shim.nvbio_aligner_all(C.byref(fs), C.byref(rs), C.c_uint32(n), C.c_uint32(L), vp(reads_rev.words), C.c_uint64(reads_rev.words.numel()), vp(reads_rev.begin),
                                vp(fwrc), C.c_uint64(fwrc.numel()), vp(quals), C.c_uint64(quals.numel()), vp(d_gw), C.c_uint64(d_gw.numel()), C.c_uint32(text.size),
                                hp(si), C.c_uint32(si.size), C.byref(sp), C.c_uint64(cap), hp(count), hp(rid), hp(aln), hp(scored), hp(cigar), hp(cigar_len), hp(source),
                                hp(sink), hp(mds), hp(mds_len), hp(stats))
    assert rc == 0
    assert (int(stats[0]), int(stats[1]), int(stats[2])) == (e["stats"]["hits"], e["stats"]["ranges"], e["stats"]["unique"])
    assert int(count[0]) == m
    assert (rid[:m] == e["read_id"]).all() and (scored[:m] == e["alignments_scored"]).all() and (aln[:m] == e["alignments"]).all()
    tb = e["tb"]
    assert (cigar_len[:m] == tb["cigar_len"]).all() and (cigar[:m] == tb["cigar"][:m]).all()
    assert (source[:m] == tb["source"]).all() and (sink[:m] == tb["sink"]).all()
    assert (mds_len[:m] == e["mds_len"]).all()
    mk = np.arange(256)[None, :] < np.minimum(e["mds_len"], 256)[:, None]
    assert ((mds[:m] == e["mds"]) | ~mk).all()


def test_cxx_all_mapping_driver_in_edit_distance_mode_equals_the_python_driver(cuda):
    """--all --scoring ed (compute_thread.cu:265-278): the C++ Aligner::all against the Python all_mapping, which tests/test_ref_tests_gpu.py holds
    to the unchanged nvBowtie in this mode"""
    import ctypes as C, os
    from nvbio_amd import pipeline as P
    from tests.test_select_gpu import _ShimParams
    shim = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "cxx", "libaligner_shim.so"))
    rng = np.random.default_rng(4343)
    text = _genome(rng)
    host, rhost = O.FMIndex(text), O.FMIndex(text[::-1].copy())
    fmi, rfmi = nvb.FMIndexDevice.from_host(host, cuda), nvb.FMIndexDevice.from_host(rhost, cuda)
    params = A.Params(scoring_mode="ed")
    n, L = 600, 100
    reads, _ = _reads(rng, text, n, False)
    sym = np.stack(reads)
    scheme = nvb.SmithWatermanScoringScheme()
    gw = W._pack_chunked(torch.from_numpy(text), 2, True)
    d_sym, d_gw = torch.from_numpy(sym).to(cuda), gw.to(cuda)
    e = A.all_mapping(fmi, rfmi, d_sym, d_gw, text.size, params, scheme, cigar_stride=64)
    sw = A.all_mapping(fmi, rfmi, d_sym, d_gw, text.size, A.Params(), scheme, cigar_stride=64)
    reads_rev, fwrc = P.pack_read_streams(d_sym)
    quals = torch.full((2 * n * L + 8,), 30, dtype=torch.uint8, device=cuda)
    sp = _ShimParams(int(params.local), 0, 0, params.max_effort_init, params.max_effort, params.min_ext, params.max_ext,
                     params.max_reseed, params.rep_seeds, params.max_hits, params.allow_sub, params.subseed_len, params.seed_len, params.seed_freq[0],
                     params.min_read_len, params.max_dist, 0, params.batch_size, params.hits_stride or 0,
                     params.seed_freq[1], params.seed_freq[2], scheme.m_match, scheme.m_score_min[0], scheme.m_score_min[1], scheme.m_score_min[2], 1, 1)
    m = int(e["read_id"].numel())
    assert m > n // 2
    cap = m + 16
    rid = np.zeros(cap, np.uint32); aln = np.zeros(cap, np.uint64); scored = np.zeros(cap, np.uint64); cigar = np.zeros((cap, 64), np.uint16)
    cigar_len = np.zeros(cap, np.uint32); source = np.zeros((cap, 2), np.uint32); sink = np.zeros((cap, 2), np.uint32)
    mds = np.zeros((cap, 256), np.uint8); mds_len = np.zeros(cap, np.uint32); stats = np.zeros(3, np.uint64); count = np.zeros(1, np.uint64)
    si = np.asarray([0, text.size], np.uint32)
    fs, rs = fmi.struct(), rfmi.struct()
    vp = lambda t: C.c_void_p(t.data_ptr())
    hp = lambda a: a.ctypes.data_as(C.c_void_p)
    torch.cuda.synchronize()
    rc = shim.nvbio_aligner_all(C.byref(fs), C.byref(rs), C.c_uint32(n), C.c_uint32(L), vp(reads_rev.words), C.c_uint64(reads_rev.words.numel()), vp(reads_rev.begin),
                                vp(fwrc), C.c_uint64(fwrc.numel()), vp(quals), C.c_uint64(quals.numel()), vp(d_gw), C.c_uint64(d_gw.numel()), C.c_uint32(text.size),
                                hp(si), C.c_uint32(si.size), C.byref(sp), C.c_uint64(cap), hp(count), hp(rid), hp(aln), hp(scored), hp(cigar), hp(cigar_len), hp(source),
                                hp(sink), hp(mds), hp(mds_len), hp(stats))
    assert rc == 0 and int(count[0]) == m
    u64 = lambda t: t.cpu().numpy().view(np.uint64)
    assert (rid[:m] == e["read_id"].cpu().numpy().view(np.uint32)).all() and (scored[:m] == u64(e["alignments_scored"])).all() and (aln[:m] == u64(e["alignments"])).all()
    assert (cigar_len[:m] == e["cigar_len"].cpu().numpy().view(np.uint32)).all() and (cigar[:m] == e["cigar"].cpu().numpy().view(np.uint16)[:m]).all()
    assert (source[:m] == e["source"].cpu().numpy().view(np.uint32)).all() and (mds_len[:m] == e["mds_len"].cpu().numpy().view(np.uint32)).all()
    assert int(sw["read_id"].numel()) != m or not torch.equal(sw["alignments"], e["alignments"])          # the mode is not the Smith-Waterman one


def test_cxx_all_mapping_driver_takes_reads_of_their_own_lengths(cuda):
    """ReadBatch::read_begin / read_len through the C++ Aligner::all, against the Python all_mapping on the same ragged batch"""
    import ctypes as C, os
    from tests.test_select_gpu import _ShimParams, _ragged_reads
    shim = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "cxx", "libaligner_shim.so"))
    rng = np.random.default_rng(4444)
    text = _genome(rng)
    host, rhost = O.FMIndex(text), O.FMIndex(text[::-1].copy())
    fmi, rfmi = nvb.FMIndexDevice.from_host(host, cuda), nvb.FMIndexDevice.from_host(rhost, cuda)
    params = A.Params()
    n = 500
    flat, index, fq = _ragged_reads(rng, text, n, 60, 140)
    scheme = nvb.SmithWatermanScoringScheme()
    d_gw = W._pack_chunked(torch.from_numpy(text), 2, True).to(cuda)
    rb = A.ReadBatch.from_ragged(torch.from_numpy(flat).to(cuda), torch.from_numpy(index).to(cuda), torch.from_numpy(fq).to(cuda))
    e = A.all_mapping(fmi, rfmi, rb, d_gw, text.size, params, scheme, cigar_stride=64)
    sp = _ShimParams(int(params.local), 0, 0, params.max_effort_init, params.max_effort, params.min_ext, params.max_ext,
                     params.max_reseed, params.rep_seeds, params.max_hits, params.allow_sub, params.subseed_len, params.seed_len, params.seed_freq[0],
                     params.min_read_len, params.max_dist, 0, params.batch_size, params.hits_stride or 0,
                     params.seed_freq[1], params.seed_freq[2], scheme.m_match, scheme.m_score_min[0], scheme.m_score_min[1], scheme.m_score_min[2], 1, 0)
    m = int(e["read_id"].numel())
    assert m > n // 2
    cap = m + 16
    rid = np.zeros(cap, np.uint32); aln = np.zeros(cap, np.uint64); scored = np.zeros(cap, np.uint64); cigar = np.zeros((cap, 64), np.uint16)
    cigar_len = np.zeros(cap, np.uint32); source = np.zeros((cap, 2), np.uint32); sink = np.zeros((cap, 2), np.uint32)
    mds = np.zeros((cap, 256), np.uint8); mds_len = np.zeros(cap, np.uint32); stats = np.zeros(3, np.uint64); count = np.zeros(1, np.uint64)
    si = np.asarray([0, text.size], np.uint32)
    fs, rs = fmi.struct(), rfmi.struct()
    vp = lambda t: C.c_void_p(t.data_ptr())
    hp = lambda a: a.ctypes.data_as(C.c_void_p)
    shim.nvbio_aligner_set_ragged.restype = None
    torch.cuda.synchronize()
    shim.nvbio_aligner_set_ragged(0, vp(rb.read_begin), vp(rb.read_len), C.c_uint64(int(rb.rc_offset)))
    try:
        rc = shim.nvbio_aligner_all(C.byref(fs), C.byref(rs), C.c_uint32(n), C.c_uint32(rb.max_len), vp(rb.reversed.words), C.c_uint64(rb.reversed.words.numel()), vp(rb.reversed.begin),
                                    vp(rb.fw_rc_words), C.c_uint64(rb.fw_rc_words.numel()), vp(rb.quals), C.c_uint64(rb.quals.numel()), vp(d_gw), C.c_uint64(d_gw.numel()), C.c_uint32(text.size),
                                    hp(si), C.c_uint32(si.size), C.byref(sp), C.c_uint64(cap), hp(count), hp(rid), hp(aln), hp(scored), hp(cigar), hp(cigar_len), hp(source),
                                    hp(sink), hp(mds), hp(mds_len), hp(stats))
    finally:
        shim.nvbio_aligner_set_ragged(0, None, None, C.c_uint64(0))
    assert rc == 0 and int(count[0]) == m
    u64 = lambda t: t.cpu().numpy().view(np.uint64)
    assert (rid[:m] == e["read_id"].cpu().numpy().view(np.uint32)).all() and (scored[:m] == u64(e["alignments_scored"])).all() and (aln[:m] == u64(e["alignments"])).all()
    assert (cigar_len[:m] == e["cigar_len"].cpu().numpy().view(np.uint32)).all() and (cigar[:m] == e["cigar"].cpu().numpy().view(np.uint16)[:m]).all()
    assert (mds_len[:m] == e["mds_len"].cpu().numpy().view(np.uint32)).all()


@pytest.mark.parametrize("bs", [1 << 20, 257])
def test_all_mapping_matches_committed_vectors(cuda, bs):
    """The HIP path against the committed fixture (tests/golden/all_mapping_vectors.npz): no oracle call at run time."""
    import os
    from tests.golden import make_all_mapping_vectors as G
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "all_mapping_vectors.npz"))
    text, reads, quals = G.case()
    host, rhost = O.FMIndex(text), O.FMIndex(text[::-1].copy())                     # (index construction only)
    fmi, rfmi = nvb.FMIndexDevice.from_host(host, cuda), nvb.FMIndexDevice.from_host(rhost, cuda)
    gw = W._pack_chunked(torch.from_numpy(text), 2, True).to(cuda)
    index = np.zeros(len(reads) + 1, np.int64); index[1:] = np.cumsum([r.size for r in reads])
    batch = A.ReadBatch.from_ragged(torch.from_numpy(np.concatenate(reads)).to(cuda), torch.from_numpy(index).to(cuda), torch.from_numpy(np.concatenate(quals)).to(cuda))
    r = A.all_mapping(fmi, rfmi, batch, gw, text.size, A.Params(batch_size=bs), nvb.SmithWatermanScoringScheme(), cigar_stride=64, sequence_index=[0, 5000, text.size])
    torch.cuda.synchronize()
    m = g["read_id_%d" % bs].size
    assert r["read_id"].numel() == m
    assert (r["read_id"].cpu().numpy().view(np.uint32) == g["read_id_%d" % bs]).all()
    assert (r["alignments_scored"].cpu().numpy().view(np.uint64) == g["scored_%d" % bs]).all() and (r["alignments"].cpu().numpy().view(np.uint64) == g["finished_%d" % bs]).all()
    assert (r["cigar_len"].cpu().numpy().view(np.uint32) == g["cigar_len_%d" % bs]).all()
    assert (r["cigar"].cpu().numpy().view(np.uint16)[:, :12] == g["cigar_%d" % bs]).all() and (r["mds_len"].cpu().numpy().view(np.uint32) == g["mds_len_%d" % bs]).all()
    assert [r["stats"]["hits"], r["stats"]["ranges"], r["stats"]["unique"]] == g["stats_%d" % bs].tolist()
