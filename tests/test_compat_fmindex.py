"""FM-index half of the drop-in template layer: caller kernels written against the reference's fm_index<> / rank_dictionary<>
templates (tests/compat/fm_callers.hip, in the shape of nvbio-test/fmindex_test.cu:63-92 and rank_test.cu) compiled with
`hipcc -I include/nvbio_hip/compat`: match / match_reverse / locate / ssa iterators / rank / rank4 per thread, over separate
bwt + occ arrays with 32- and 64-bit indices and over the interleaved uint4 production layout -- against the oracle.
The host instantiation of the same templates runs on the CPU suite."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import pyoracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "tests", "compat", "libfm_callers.so")


@pytest.fixture(scope="module")
def fm():
    assert os.path.exists(LIB), "build with python -c 'import __graft_entry__ as g; g.build()'"
    L = C.CDLL(LIB)
    vp, u32, u64, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int
    L.compat_fm_search.argtypes = [i32, u64, u64, vp, vp, vp, vp, vp, u32, u32, vp, vp, vp, vp, vp]
    L.compat_fm_rank.argtypes = [i32, u64, u64, vp, vp, vp, vp, u32, vp, vp, vp, vp]
    L.compat_fm_search_host.argtypes = [u32, u32, vp, vp, vp, vp, u32, u32, vp, vp, vp, vp]
    L.compat_fm_rank4_range.argtypes = [i32, u64, u64, vp, vp, vp, vp, u32, vp, vp, vp, vp, vp]
    L.compat_fm_one_mismatch.argtypes = [i32, u32, u32, vp, vp, vp, vp, u32, u32, u32, vp, vp, vp]
    return L


@pytest.fixture(scope="module")
def index():
    rng = np.random.default_rng(17)
    n = 60001
    text = rng.integers(0, 4, n, dtype=np.uint8)
    text[2000:2600] = np.tile(np.array([0, 1, 1], np.uint8), 200)
    host = O.FMIndex(text)
    nb = (n + 63) // 64 + 1
    occ = np.zeros(nb * 4, dtype=np.uint32)
    cum = np.zeros((n + 1, 4), dtype=np.int64)
    for c in range(4):
        cum[1:, c] = np.cumsum(host.bwt == c)
    for k in range(nb):
        occ[4 * k: 4 * k + 4] = cum[min(64 * k, n)]
    bw = np.concatenate([O.pack(host.bwt, 2, True, pad_words=0), np.zeros(8, np.uint32)])
    bw = bw[: (bw.size // 2) * 2]
    gw = np.concatenate([O.pack(text, 2, True, pad_words=0), np.zeros(8, np.uint32)])
    gw = gw[: (gw.size // 2) * 2]
    to64 = lambda w: (w[0::2].astype(np.uint64) << np.uint64(32)) | w[1::2].astype(np.uint64)
    ssa64 = host.ssa.astype(np.uint64)
    ssa64[0] = np.uint64(0xFFFFFFFFFFFFFFFF)
    d = dict(text=text, host=host, n=n, occ32=occ, bwt32=bw, genome32=gw, L2_32=host.L2.astype(np.uint32), ssa32=host.ssa,
             occ64=occ.astype(np.uint64), bwt64=to64(bw), genome64=to64(gw), L2_64=host.L2.astype(np.uint64), ssa64=ssa64)
    ct = np.zeros(256, np.uint32)
    d["count_table"] = ct
    return d


def queries(d, nq, qlen, seed):
    rng = np.random.default_rng(seed)
    starts = rng.integers(0, d["n"] - qlen + 1, nq).astype(np.uint32)
    starts[:4] = [0, d["n"] - qlen, 2000, 2301]
    pats = [d["text"][s:s + qlen] for s in starts]
    exp = d["host"].match(O.StringSet.from_lists(pats, 2, True))
    rexp = d["host"].match(O.StringSet.from_lists([p[::-1] for p in pats], 2, True))
    pos = d["host"].locate(exp[:, 0])
    return starts, exp, rexp, pos


def test_host_instantiation_matches_oracle(fm, index):
    d = index
    starts, exp, _, pos = queries(d, 3000, 22, 1)
    ranges = np.zeros((starts.size, 2), np.uint32)
    positions = np.zeros(starts.size, np.uint32)
    p = lambda a: a.ctypes.data
    assert fm.compat_fm_search_host(d["n"], d["host"].primary, p(d["L2_32"]), p(d["bwt32"]), p(d["occ32"]), p(d["ssa32"]),
                                    starts.size, 22, p(d["genome32"]), p(starts), p(ranges), p(positions)) == 0
    assert (ranges == exp).all()
    assert (positions == pos).all()
    assert (d["text"][positions[0]:positions[0] + 22] == d["text"][starts[0]:starts[0] + 22]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("layout", [0, 1, 2], ids=["separate32", "separate64", "interleaved_uint4"])
@pytest.mark.parametrize("qlen", [22, 7, 33])
def test_caller_kernel_match_locate(fm, index, layout, qlen):
    import torch
    d = index
    starts, exp, rexp, pos = queries(d, 20000, qlen, 2 + qlen)
    dt = np.uint64 if layout == 1 else np.uint32
    sfx = "64" if layout == 1 else "32"
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64 if a.dtype == np.uint64 else np.int32 if a.dtype == np.uint32 else a.dtype)).cuda()
    L2, ssa, genome = dev(d["L2_" + sfx]), dev(d["ssa" + sfx]), dev(d["genome" + sfx])
    if layout == 2:
        bwt, occ = dev(d["host"].bwt_occ), None
    else:
        bwt, occ = dev(d["bwt" + sfx]), dev(d["occ" + sfx])
    ct, st = dev(d["count_table"]), dev(starts)
    tdt = torch.int64 if layout == 1 else torch.int32
    ranges = torch.zeros((starts.size, 2), dtype=tdt, device="cuda")
    rranges = torch.zeros((starts.size, 2), dtype=tdt, device="cuda")
    positions = torch.zeros(starts.size, dtype=tdt, device="cuda")
    vp = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    assert fm.compat_fm_search(layout, d["n"], d["host"].primary, vp(L2), vp(bwt), vp(occ), vp(ct), vp(ssa), starts.size, qlen, vp(genome), vp(st),
                               vp(ranges), vp(positions), vp(rranges)) == 0
    got = ranges.cpu().numpy().view(dt)
    assert (got == exp.astype(dt)).all()
    assert (positions.cpu().numpy().view(dt) == pos.astype(dt)).all()
    grr = rranges.cpu().numpy().view(dt)
    empty = rexp[:, 0] > rexp[:, 1]
    assert (grr[~empty] == rexp[~empty].astype(dt)).all()
    # an empty range is the raw (l, r) of the step that emptied it: the same 32-bit values, computed in the index's own width
    if layout != 1:
        assert (grr[empty] == rexp[empty]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("layout", [0, 1, 2], ids=["separate32", "separate64", "interleaved_uint4"])
def test_caller_kernel_rank(fm, index, layout):
    import torch
    d = index
    rng = np.random.default_rng(9)
    n = d["n"]
    rows = np.concatenate([rng.integers(0, n + 1, 30000), [0, n, d["host"].primary, d["host"].primary - 1, d["host"].primary + 1, 63, 64, 65]]).astype(np.uint32)
    syms = rng.integers(0, 4, rows.size).astype(np.uint8)
    exp, exp4 = d["host"].rank(rows, syms), d["host"].rank4(rows)
    dt = np.uint64 if layout == 1 else np.uint32
    sfx = "64" if layout == 1 else "32"
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64 if a.dtype == np.uint64 else np.int32 if a.dtype == np.uint32 else a.dtype)).cuda()
    L2 = dev(d["L2_" + sfx])
    if layout == 2:
        bwt, occ = dev(d["host"].bwt_occ), None
    else:
        bwt, occ = dev(d["bwt" + sfx]), dev(d["occ" + sfx])
    ct, r, s = dev(d["count_table"]), dev(rows.astype(dt)), dev(syms)
    tdt = torch.int64 if layout == 1 else torch.int32
    out = torch.zeros(rows.size, dtype=tdt, device="cuda")
    out4 = torch.zeros((rows.size, 4), dtype=tdt, device="cuda")
    vp = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    assert fm.compat_fm_rank(layout, n, d["host"].primary, vp(L2), vp(bwt), vp(occ), vp(ct), rows.size, vp(r), vp(s), vp(out), vp(out4)) == 0
    assert (out.cpu().numpy().view(dt) == exp.astype(dt)).all()
    assert (out4.cpu().numpy().view(dt) == exp4.astype(dt)).all()


@pytest.mark.gpu
@pytest.mark.parametrize("layout", [0, 1, 2], ids=["separate32", "separate64", "interleaved_uint4"])
def test_caller_kernel_range_rank4_and_rank_all(fm, index, layout):
    """rank4(fmi, (l, r), &lo, &hi) / rank_all(fmi, (l, r), ...) / comp() -- the forms nvBowtie's map<> steps with
    (fmindex.h:451-500, numbers.h:239-290) -- per end against the oracle's rank4, on ranges inside one block, across blocks, around
    `primary`, and with the l == -1 / r == n ends of the first search step."""
    import torch
    d = index
    rng = np.random.default_rng(19)
    n, pr = d["n"], d["host"].primary
    lo = rng.integers(0, n + 1, 40000).astype(np.int64)
    hi = np.minimum(n, lo + rng.choice([0, 1, 5, 40, 63, 64, 200, 5000], lo.size))
    extra_lo = np.array([-1, -1, 0, pr - 1, pr, pr - 1, n - 1, n, 62, 63, -1], np.int64)
    extra_hi = np.array([n, 0, 0, pr, pr, pr + 1, n, n, 63, 64, pr], np.int64)
    lo, hi = np.concatenate([lo, extra_lo]), np.concatenate([hi, extra_hi])
    dt = np.uint64 if layout == 1 else np.uint32
    exp_lo, exp_hi = d["host"].rank4(lo.astype(np.uint32)), d["host"].rank4(hi.astype(np.uint32))
    sfx = "64" if layout == 1 else "32"
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64 if a.dtype == np.uint64 else np.int32 if a.dtype == np.uint32 else a.dtype)).cuda()
    L2 = dev(d["L2_" + sfx])
    if layout == 2:
        bwt, occ = dev(d["host"].bwt_occ), None
    else:
        bwt, occ = dev(d["bwt" + sfx]), dev(d["occ" + sfx])
    ct = dev(d["count_table"])
    lo_d, hi_d = dev(lo.astype(np.int64).astype(dt) if layout == 1 else lo.astype(np.uint32)), dev(hi.astype(dt))
    tdt = torch.int64 if layout == 1 else torch.int32
    out_lo = torch.zeros((lo.size, 4), dtype=tdt, device="cuda")
    out_hi = torch.zeros((lo.size, 4), dtype=tdt, device="cuda")
    agree = torch.zeros(lo.size, dtype=torch.int32, device="cuda")
    vp = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    assert fm.compat_fm_rank4_range(layout, n, pr, vp(L2), vp(bwt), vp(occ), vp(ct), lo.size, vp(lo_d), vp(hi_d), vp(out_lo), vp(out_hi), vp(agree)) == 0
    assert (out_lo.cpu().numpy().view(dt) == exp_lo.astype(dt)).all()
    assert (out_hi.cpu().numpy().view(dt) == exp_hi.astype(dt)).all()
    assert int(agree.sum()) == lo.size


@pytest.mark.gpu
@pytest.mark.parametrize("layout", [0, 2], ids=["separate32", "interleaved_uint4"])
@pytest.mark.parametrize("len1,len2", [(10, 22), (0, 8), (4, 9)])
def test_caller_kernel_one_mismatch_search(fm, index, layout, len1, len2):
    """A caller of the shape of nvBowtie's map<> (mapping_inl.h:128-220) through the drop-in templates: exact over the first len1
    symbols, one substitution in the rest, stepping with the range rank4 + comp().  Expected: the oracle's match() of every
    substituted query, which reaches the same SA range by plain backward search."""
    import torch
    d = index
    rng = np.random.default_rng(100 * len1 + len2 + layout)
    nq = 1500
    starts = rng.integers(0, d["n"] - len2 + 1, nq).astype(np.uint32)
    starts[:3] = [0, 2000, 2301]
    slots = 1 + 3 * (len2 - len1)
    # query[i] is prepended at step i: the string searched for is the query reversed
    pats, where = [], []
    for q, s in enumerate(starts):
        qu = d["text"][s:s + len2]
        pats.append(qu[::-1].copy()); where.append((q, 0))
        for i in range(len1, len2):
            k = 0
            for sub in range(4):
                if sub == qu[i]:
                    continue
                v = qu.copy(); v[i] = sub
                pats.append(v[::-1].copy()); where.append((q, 1 + 3 * (i - len1) + k))
                k += 1
    got_exp = d["host"].match(O.StringSet.from_lists(pats, 2, True))
    exp = np.zeros((nq, slots, 2), np.uint32)
    exp[:, :, 0] = 1
    for (q, s), r in zip(where, got_exp):
        if r[0] <= r[1]:
            exp[q, s] = r
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int32 if a.dtype == np.uint32 else a.dtype)).cuda()
    L2 = dev(d["L2_32"])
    if layout == 2:
        bwt, occ = dev(d["host"].bwt_occ), None
    else:
        bwt, occ = dev(d["bwt32"]), dev(d["occ32"])
    ct, st, genome = dev(d["count_table"]), dev(starts), dev(d["genome32"])
    out = torch.zeros((nq, slots, 2), dtype=torch.int32, device="cuda")
    vp = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    assert fm.compat_fm_one_mismatch(layout, d["n"], d["host"].primary, vp(L2), vp(bwt), vp(occ), vp(ct), nq, len1, len2, vp(genome), vp(st), vp(out)) == 0
    got = out.cpu().numpy().view(np.uint32)
    assert (got == exp).all()
    if len2 < 10:
        assert (exp[:, 1:, 0] <= exp[:, 1:, 1]).sum() > nq                 # the substitution branch really fired (8- and 9-mers of a 60 k text)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 1, 2, 3, 4], ids=["one_walk", "two_walks_interleaved", "repeated_queries", "one_index_in_shared_memory", "one_index_behind_a_pointer"])
def test_line_native_records_under_the_per_thread_functions(fm, mode):
    """fmindex/line_native.h: the same caller kernel -- a walk in the shape of nvBowtie's match_range (one rank(index, (x - 1, y), c) per symbol on an
    index held by value) and locate_ssa_iterator / lookup_ssa_iterator over random rows -- on the production-layout fm_index WITH the line-native
    records attached and WITHOUT them: every raw pair of counts of every step, every final range (emptied ones included), every iterator and every
    position must be identical, and equal the oracle's.  Mode 1 interleaves two searches on one index object and mode 2 repeats queries, so that the
    step a call keeps for its successor is found by calls it was not meant for.  Modes 3 and 4 put ONE fm_index object in __shared__ / global memory and
    let every lane of the block / grid walk it at once -- the reference's fm_index is a read-only view, so a shared object must answer exactly as a
    private copy does (the kept step is only ever used on an object in the lane's private memory, line_native.h: thread_private()).
    The genome has exact and diverged repeats (long non-empty walks)."""
    import torch
    import nvbio_amd as nvb
    fm.compat_fm_native_walk.argtypes = [C.c_uint32, C.c_uint32] + [C.c_void_p] * 5 + [C.c_uint32] * 3 + [C.c_void_p] * 4 + [C.c_uint32] + [C.c_void_p] * 3
    rng = np.random.default_rng(91)
    n = 300_007
    text = rng.integers(0, 4, n, dtype=np.uint8)
    fam = rng.integers(0, 4, 700, dtype=np.uint8)
    for k in range(120):
        at = int(rng.integers(0, n - 800)); c = fam.copy(); m = rng.random(c.size) < 0.01; c[m] = (c[m] + 1) & 3
        text[at:at + c.size] = c
    text[5000:5900] = np.tile(np.array([2, 0, 0, 3], np.uint8), 225)
    host = O.FMIndex(text)
    dev = torch.device("cuda:0")
    fmi = nvb.FMIndexDevice.from_host(host, dev).with_dimer()
    qlen, nq, nrows = 40, 30000, 50000
    # the walk consumes query[0] first and PREPENDS every symbol: it searches for the reversed query.  Queries are therefore slices of the reversed
    # genome (their reversals occur: walks that stay non-empty), followed by a stretch of unrelated random symbols (walks that run empty)
    source = np.concatenate([text[::-1], rng.integers(0, 4, 60_000, dtype=np.uint8)])
    starts = rng.integers(0, source.size - qlen, nq).astype(np.uint32)
    starts[:3] = [0, n - qlen, n - 5900]
    gw = np.concatenate([O.pack(source, 2, True, pad_words=0), np.zeros(8, np.uint32)])
    rows = rng.integers(0, n + 1, nrows).astype(np.uint32)
    rows[:4] = [0, n, host.primary, max(host.primary - 1, 0)]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).to(dev)
    d_L2, d_ct, d_gw, d_st, d_rows = t(host.L2.astype(np.uint32)), t(np.zeros(256, np.uint32)), t(gw), t(starts), t(rows)
    vp = lambda x: C.c_void_p(x.data_ptr()) if x is not None else None
    outs = []
    for native in (None, fmi.dimer):
        ranges = torch.zeros((nq, 2), dtype=torch.int32, device=dev)
        trace = torch.zeros((nq, 2 * qlen), dtype=torch.int32, device=dev)
        its = torch.zeros((nrows, 2), dtype=torch.int32, device=dev)
        posn = torch.zeros(nrows, dtype=torch.int32, device=dev)
        assert fm.compat_fm_native_walk(n, host.primary, vp(d_L2), vp(fmi.bwt_occ), vp(d_ct), vp(fmi.ssa), vp(native), mode, nq, qlen, vp(d_gw), vp(d_st),
                                        vp(ranges), vp(trace), nrows, vp(d_rows), vp(its), vp(posn)) == 0
        outs.append([x.cpu().numpy().view(np.uint32) for x in (ranges, trace, its, posn)])
    for a, b, what in zip(outs[0], outs[1], ("ranges", "step counts", "ssa iterators", "positions")):
        assert (a == b).all(), "%s differ between the reference layout and the line-native records (%d of %d)" % (what, int((a != b).sum()), a.size)
    assert not (outs[1][1] == 0xDEADBEEF).any()
    # against the oracle: the walk consumes query[0] first, i.e. it is a backward search for the reversed slice
    pats = [source[s:s + qlen][::-1] for s in starts]
    exp = host.match(O.StringSet.from_lists(pats, 2, True))
    assert (outs[1][0] == exp).all()
    assert (outs[1][3] == host.locate(rows)).all()
    emptied = int((exp[:, 0] > exp[:, 1]).sum())
    assert 0 < emptied < nq
