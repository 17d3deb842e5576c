"""Parity of the HIP FM-index kernels (through the C-ABI) with the CPU oracle: bit-exact
ranks, SA ranges, located positions and filter hits.  Mirrors nvbio-test/rank_test.cu and
fmindex_test.cu (host vs device equality on synthetic texts, sorted and shuffled queries)."""
import numpy as np
import pytest
import torch

import nvbio_amd as nvb
from nvbio_amd import workloads as W
from oracle import pyoracle as O

pytestmark = pytest.mark.gpu


def i32(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).to(dev)


def u32(t):
    return t.cpu().numpy().view(np.uint32)


@pytest.fixture(scope="module", params=[1 << 20, 100003, 77], ids=["1M", "100003", "77"])
def index(request):
    n = request.param
    rng = np.random.default_rng(n)
    text = rng.integers(0, 4, n, dtype=np.uint8)
    if n > 1000:
        text[n // 3: n // 3 + 2000] = 0          # a long run: exercises wide SA ranges and c==0 counting
    host = O.FMIndex(text)
    dev = nvb.FMIndexDevice.from_host(host, "cuda")
    return text, host, dev


def test_rank_point_queries(cuda, index):
    text, host, dev = index
    n = host.length
    rng = np.random.default_rng(1)
    k = np.concatenate([np.arange(min(n + 1, 5000)), rng.integers(0, n + 1, 50000),
                        [0xFFFFFFFF, n, host.primary, max(host.primary - 1, 0), min(host.primary + 1, n)]]).astype(np.uint32)
    c = rng.integers(0, 4, k.size).astype(np.uint8)
    exp = host.rank(k, c)
    got = u32(nvb.rank(dev, i32(k, cuda), torch.from_numpy(c).to(cuda)))
    assert (got == exp).all()
    assert (u32(nvb.rank4(dev, i32(k, cuda))) == host.rank4(k)).all()


def test_rank_range(cuda, index):
    text, host, dev = index
    n = host.length
    rng = np.random.default_rng(2)
    x = rng.integers(-1, n + 1, 60000)
    y = np.minimum(x + rng.integers(0, 200, x.size), n)
    x[:1000] = -1
    y[1000:2000] = n
    x[2000:3000] = y[2000:3000]
    r = np.stack([x.astype(np.int64) & 0xFFFFFFFF, y.astype(np.int64)], 1).astype(np.uint32)
    c = rng.integers(0, 4, x.size).astype(np.uint8)
    exp = host.rank_range(r, c)
    got = u32(nvb.rank_range(dev, i32(r, cuda), torch.from_numpy(c).to(cuda)))
    assert (got == exp).all()


def make_seeds(rng, text, n_seeds, length, bits, be, with_n=False):
    n = text.size
    seeds = []
    for i in range(n_seeds):
        L = length if length else int(rng.integers(1, 40))
        if i % 10 != 9 and n > L:
            p = int(rng.integers(0, n - L))
            s = text[p:p + L].copy()
        else:
            s = rng.integers(0, 4, L, dtype=np.uint8)
        if with_n and bits == 4 and i % 50 == 7:
            s[int(rng.integers(0, L))] = 4
        seeds.append(s)
    return O.StringSet.from_lists(seeds, bits, be)


@pytest.mark.parametrize("bits,be", [(2, True), (2, False), (4, True), (4, False)])
def test_match_ranges(cuda, index, bits, be):
    text, host, dev = index
    rng = np.random.default_rng(bits * 2 + be)
    for length in (22, 0):
        hs = make_seeds(rng, text, 20000, length, bits, be, with_n=True)
        exp = host.match(hs)
        ds = nvb.PackedStringSet.from_host(hs.words, bits, be, hs.begin, hs.length, device=cuda)
        got = u32(nvb.match(dev, ds))
        assert (got == exp).all()
    # fixed-length form (length == NULL)
    hs = make_seeds(rng, text, 5000, 22, bits, be)
    ds = nvb.PackedStringSet.from_host(hs.words, bits, be, hs.begin, None, 22, device=cuda)
    assert (u32(nvb.match(dev, ds)) == host.match(hs)).all()


def test_locate_sorted_and_shuffled(cuda, index):
    """fmindex_test.cu:666-716: device == host for sorted and shuffled query orders."""
    text, host, dev = index
    n = host.length
    rng = np.random.default_rng(4)
    rows = rng.integers(0, n + 1, 100000).astype(np.uint32)
    for order in (np.sort(rows), rows):
        exp = host.locate(order)
        assert (u32(nvb.locate(dev, i32(order, cuda))) == exp).all()
        it = nvb.locate_ssa_iterator(dev, i32(order, cuda))
        assert (u32(it) == host.locate_ssa_iterator(order)).all()
        assert (u32(nvb.lookup_ssa_iterator(dev, it)) == exp).all()
    if n > 1000:   # every row: locate reproduces the suffix array
        allrows = np.arange(1, n + 1, dtype=np.uint32)
        assert (u32(nvb.locate(dev, i32(allrows, cuda))) == host.sa[1:]).all()


def test_filter_rank_and_locate(cuda, index):
    text, host, dev = index
    rng = np.random.default_rng(6)
    hs = make_seeds(rng, text, 4000, 12 if host.length > 1000 else 3, 2, True)
    total, eranges, eslots = host.filter_rank(hs)
    ds = nvb.PackedStringSet.from_host(hs.words, 2, True, hs.begin, hs.length, device=cuda)
    flt = nvb.FMIndexFilter()
    assert flt.rank(dev, ds) == total
    assert (u32(flt.ranges) == eranges).all()
    assert (flt.slots.cpu().numpy().view(np.uint64) == eslots).all()
    for (b, e) in ((0, min(total, 50000)), (total // 3, min(total, total // 3 + 10000))):
        if e > b:
            assert (u32(flt.locate(b, e)) == host.filter_locate(eranges, eslots, b, e)).all()


def test_build_bwt_occ_matches_host(cuda, index):
    text, host, dev = index
    n = host.length
    nb = (n + 63) // 64
    bw = np.zeros(nb * 4, dtype=np.uint32)
    pw = O.pack(host.bwt, 2, True, pad_words=0)
    bw[:pw.size] = pw
    out, L2 = nvb.build_bwt_occ(n, i32(bw, cuda))
    assert (u32(out) == host.bwt_occ[:nb * 8]).all()
    assert L2 == [int(x) for x in host.L2]


def test_random_bwt_rank_property_large(cuda):
    """Config-3 shaped index (random BWT, device-built occ table) at 2^28 symbols: rank of the
    last row equals the symbol totals, rank is monotone, and a sample agrees with a direct count."""
    n = (1 << 28) + 12345
    words = W.make_random_bwt(n, device=cuda)
    bwt_occ, L2 = nvb.build_bwt_occ(n, words)
    f = nvb.FMIndexDevice(n, n, L2, bwt_occ)          # primary = n: no '$' shift inside [0,n)
    k = torch.randint(0, n, (1 << 20,), device=cuda, dtype=torch.int64)
    k, _ = torch.sort(k)
    k32 = k.to(torch.int32)
    r4 = nvb.rank4(f, k32).to(torch.int64) & 0xFFFFFFFF
    assert bool((r4.sum(1) == k + 1).all())                          # all symbols accounted for
    assert bool((r4[1:] >= r4[:-1]).all())                           # monotone in k
    last = nvb.rank4(f, torch.tensor([n - 1], device=cuda, dtype=torch.int32)).to(torch.int64)[0]
    assert [int(x) for x in last] == [L2[i + 1] - L2[i] for i in range(4)]
    # direct count of a prefix on the host for a few queries
    kk = k[:: 1 << 14][:32].cpu().numpy()
    w = words.cpu().numpy().view(np.uint32)
    for q in kk[:8]:
        sym = O.unpack(w, 0, int(q) + 1, 2, True)
        assert np.bincount(sym, minlength=4).tolist() == [int(x) for x in nvb.rank4(f, torch.tensor([int(q)], device=cuda, dtype=torch.int32)).cpu()[0]]
        if q > 3_000_000:
            break


def test_device_built_index_equals_host_index(cuda):
    """The device index builder used by the FM-index bench configs (torch prefix-doubling SA +
    the build_occurrence_table kernel) reproduces the host oracle's index bit for bit."""
    rng = np.random.default_rng(21)
    n = 300007
    text = rng.integers(0, 4, n, dtype=np.uint8)
    text[1000:3000] = 2
    host = O.FMIndex(text)
    fmi, sa = W.build_fm_index(torch.from_numpy(text).to(cuda), keep_sa=True)
    assert fmi.length == n and fmi.primary == host.primary and fmi.L2 == [int(x) for x in host.L2]
    assert (sa.cpu().numpy().astype(np.uint32) == host.sa).all()
    assert (u32(fmi.bwt_occ) == host.bwt_occ[: fmi.bwt_occ.numel()]).all()
    assert (u32(fmi.ssa) == host.ssa).all()
    seeds = W.make_seeds(torch.from_numpy(text).to(cuda), 20000, 22)
    assert (u32(nvb.match(fmi, seeds)) == host.match(O.StringSet.from_device(seeds))).all()


@pytest.mark.parametrize("k", [1, 5, 8, 12])
def test_ktab_accelerated_match_is_identical(cuda, index, k):
    """The optional k-mer table must not change a single range: seeds shorter than k, seeds with an
    N inside / outside the tabulated tail, absent seeds (empty ranges frozen at their stop state)."""
    text, host, dev = index
    rng = np.random.default_rng(100 + k)
    fk = dev.with_ktab(k)
    tab = u32(fk.ktab)
    # the table itself == match of every k-mer (code: symbol t at bits 2t)
    codes = rng.integers(0, 4 ** k, 2000)
    kmers = [np.array([(c >> (2 * t)) & 3 for t in range(k)], dtype=np.uint8) for c in codes]
    assert (tab[codes] == host.match(O.StringSet.from_lists(kmers, 2, True))).all()
    for bits, be in ((2, True), (4, True), (4, False), (2, False)):
        for length in (22, 0):
            hs = make_seeds(rng, text, 20000, length, bits, be, with_n=True)
            ds = nvb.PackedStringSet.from_host(hs.words, bits, be, hs.begin, hs.length, device=cuda)
            exp = host.match(hs)
            assert (u32(nvb.match(fk, ds)) == exp).all()
            assert (u32(nvb.match(dev, ds)) == exp).all()


@pytest.mark.parametrize("sa_int", [1, 4, 64])
def test_dense_ssa_locate_is_identical(cuda, index, sa_int):
    text, host, dev = index
    n = host.length
    fd = dev.with_dense_ssa(sa_int)
    rng = np.random.default_rng(sa_int)
    rows = rng.integers(0, n + 1, 100000).astype(np.uint32)
    exp = host.locate(rows)
    assert (u32(nvb.locate(fd, i32(rows, cuda))) == exp).all()
    it = nvb.locate_ssa_iterator(fd, i32(rows, cuda))
    assert (u32(nvb.lookup_ssa_iterator(fd, it)) == exp).all()
    if sa_int == 1:
        assert (u32(fd.ssa)[1:] == host.sa[1:]).all()


# ------------------------------------------------------------------ the line-native two-symbol index
def test_dimer_index_layout_equals_model(cuda, index):
    """The device builder writes exactly the buffer the layout model describes (header constants, folded
    counters, bit-planes, fillers at primary and at the SA = 1 row)."""
    from tests import dimer_model as DM
    text, host, dev = index
    fd = dev.with_dimer()
    got = u32(fd.dimer)
    exp = DM.build(text, host.sa, host.L2)
    assert got.size == exp.size
    assert (got[:32] == exp[:32]).all(), (got[:32], exp[:32])
    assert (got == exp).all()
    p1, fill1, S, T = fd.dimer_consts
    assert [p1, fill1] == [int(exp[3]), int(exp[4])] and S == [int(x) for x in exp[8:12]] and T == [int(x) for x in exp[12:16]]


@pytest.mark.parametrize("bits,be", [(2, True), (2, False), (4, True), (4, False)])
def test_dimer_match_is_identical(cuda, index, bits, be):
    """Two symbols per step must not change a single range: odd and even lengths, seeds shorter than a pair,
    N inside a pair (either half), absent seeds (the raw (x,y) of the emptying step), with and without the k-mer table."""
    text, host, dev = index
    rng = np.random.default_rng(40 + bits * 2 + be)
    fd = dev.with_dimer()
    variants = [fd, fd.with_ktab(5), fd.with_ktab(8)]
    for length in (22, 21, 0, 1, 2, 33):
        hs = make_seeds(rng, text, 20000, length, bits, be, with_n=True)
        exp = host.match(hs)
        ds = nvb.PackedStringSet.from_host(hs.words, bits, be, hs.begin, hs.length, device=cuda)
        for f in variants:
            got = u32(nvb.match(f, ds))
            bad = np.nonzero((got != exp).any(1))[0]
            assert bad.size == 0, (length, f.ktab_k, bad[:5], got[bad[:5]], exp[bad[:5]])
    hs = make_seeds(rng, text, 5000, 22, bits, be)
    ds = nvb.PackedStringSet.from_host(hs.words, bits, be, hs.begin, None, 22, device=cuda)
    assert (u32(nvb.match(fd, ds)) == host.match(hs)).all()


@pytest.mark.parametrize("sa_int", [16, 1, 4, 64])
def test_dimer_locate_is_identical(cuda, index, sa_int):
    """Two LF steps per record: same iterators (row, steps) and positions for every SA row, the rows around
    primary and the SA = 1 row included."""
    text, host, dev = index
    n = host.length
    base = dev if sa_int == 16 else dev.with_dense_ssa(sa_int)
    fd = base.with_dimer()
    rows = np.arange(0, n + 1, dtype=np.uint32)
    if n > 200000:
        rng = np.random.default_rng(5)
        rows = np.concatenate([rng.integers(0, n + 1, 200000).astype(np.uint32),
                               np.arange(max(host.primary, 5) - 5, min(host.primary + 5, n) + 1, dtype=np.uint32)])
    exp_it = u32(nvb.locate_ssa_iterator(base, i32(rows, cuda)))
    if sa_int == 16:
        assert (exp_it == host.locate_ssa_iterator(rows)).all()
    got_it = nvb.locate_ssa_iterator(fd, i32(rows, cuda))
    bad = np.nonzero((u32(got_it) != exp_it).any(1))[0]
    assert bad.size == 0, (bad[:5], rows[bad[:5]], u32(got_it)[bad[:5]], exp_it[bad[:5]], host.primary)
    exp = host.locate(rows)
    assert (u32(nvb.locate(fd, i32(rows, cuda))) == exp).all()
    assert (u32(nvb.lookup_ssa_iterator(fd, got_it)) == exp).all()


def test_dimer_filter_is_identical(cuda, index):
    text, host, dev = index
    rng = np.random.default_rng(8)
    fd = dev.with_dimer()
    hs = make_seeds(rng, text, 4000, 12 if host.length > 1000 else 3, 2, True)
    total, eranges, eslots = host.filter_rank(hs)
    ds = nvb.PackedStringSet.from_host(hs.words, 2, True, hs.begin, hs.length, device=cuda)
    flt = nvb.FMIndexFilter()
    assert flt.rank(fd, ds) == total
    assert (u32(flt.ranges) == eranges).all()
    if total:
        e = min(total, 50000)
        assert (u32(flt.locate(0, e)) == host.filter_locate(eranges, eslots, 0, e)).all()


def test_dimer_attach_rejects_a_foreign_buffer(cuda, index):
    import ctypes as C
    from nvbio_amd._lib import lib
    text, host, dev = index
    fd = dev.with_dimer()
    other = nvb.FMIndexDevice(dev.length, dev.primary ^ 1, dev.L2, dev.bwt_occ, dev.ssa, dev.sa_int)
    s = other.struct()
    assert lib().nvbio_hip_fm_attach_dimer_index(C.byref(s), C.c_void_p(fd.dimer.data_ptr()), None) == 1      # hipErrorInvalidValue
    s = dev.struct()
    assert lib().nvbio_hip_fm_attach_dimer_index(C.byref(s), C.c_void_p(fd.dimer.data_ptr()), None) == 0 and s.dimer == fd.dimer.data_ptr()


# ------------------------------------------------------------------ BASELINE configs 3 / 4 at their own size
@pytest.fixture(scope="module")
def genome_3gbp(cuda):
    """A true FM-index of a 3 * 10^9-symbol i.i.d. genome built on the device (row indices above 2^31, the regime the
    3 Gbp configurations live in), its host copy for the oracle, and the line-native index next to it."""
    ng = 3_000_000_000
    g = torch.Generator(device=cuda)
    g.manual_seed(0x5EED0003)
    text = torch.randint(0, 4, (ng,), dtype=torch.uint8, generator=g, device=cuda)
    fmi = W.build_fm_index(text)
    host = O.FMIndex(parts=(fmi.length, fmi.primary, np.array(fmi.L2, dtype=np.uint32),
                            fmi.bwt_occ.cpu().numpy().view(np.uint32), fmi.ssa.cpu().numpy().view(np.uint32), fmi.sa_int))
    yield text, fmi, host
    del text, fmi, host
    torch.cuda.empty_cache()


def test_full_size_rank_properties(cuda, genome_3gbp):
    text, fmi, host = genome_3gbp
    n = fmi.length
    assert n == 3_000_000_000 and fmi.L2[4] == n
    # sorted rows spread over the whole index, most of them above 2^31: all symbols accounted for, monotone, ends exact
    k = torch.sort(torch.randint(0, n, (1 << 21,), device=cuda, dtype=torch.int64)).values
    k[-1] = n - 1
    k32 = k.to(torch.int32)
    r4 = nvb.rank4(fmi, k32).to(torch.int64) & 0xFFFFFFFF
    below_primary = (k < fmi.primary).to(torch.int64)
    assert bool((r4.sum(1) == k + below_primary).all())                  # rows [0,k] hold k+1 entries, one of them '$' once k >= primary
    assert bool((r4[1:] >= r4[:-1]).all())
    assert int((k >= (1 << 31)).sum()) > (1 << 19)
    # point ranks == the matching rank4 component, and a sample against the oracle on the host copy
    c = torch.randint(0, 4, (k.numel(),), device=cuda, dtype=torch.uint8)
    r = nvb.rank(fmi, k32, c).to(torch.int64) & 0xFFFFFFFF
    assert bool((r == r4.gather(1, c.to(torch.int64).unsqueeze(1)).squeeze(1)).all())
    m = 200000
    kk, cc = k32[:: k.numel() // m][:m].contiguous(), c[:: k.numel() // m][:m].contiguous()
    assert (u32(nvb.rank(fmi, kk, cc)) == host.rank(kk.cpu().numpy().view(np.uint32), cc.cpu().numpy())).all()
    last = nvb.rank4(fmi, torch.tensor([n], device=cuda, dtype=torch.int64).to(torch.int32)).to(torch.int64)[0] & 0xFFFFFFFF
    assert [int(x) for x in last] == [fmi.L2[i + 1] - fmi.L2[i] for i in range(4)]


@pytest.mark.parametrize("flavour", ["reference_layout", "line_native", "line_native_ktab12", "line_native_ktab16"])
def test_full_size_match_and_locate(cuda, genome_3gbp, flavour):
    """BASELINE config 3-ii at its own size: >= 1 M 22-bp seeds (90 % drawn from the genome) on the 3 Gbp index, match ranges
    and located positions against the oracle on a host copy -- for the reference layout and for the line-native index."""
    text, fmi, host = genome_3gbp
    idx = fmi if flavour == "reference_layout" else fmi.with_dimer()
    if flavour.endswith("ktab12"):
        idx = idx.with_ktab(12)
    if flavour.endswith("ktab16"):                       # 2^32 codes, 34 GB: the largest table (a code is a uint32)
        if torch.cuda.mem_get_info(cuda)[0] < (48 << 30):
            pytest.skip("needs 34 GB for the table")
        idx = idx.with_ktab(16)
    seeds = W.make_seeds(text, 1_200_000, 22)
    ranges = nvb.match(idx, seeds)
    exp = host.match(O.StringSet.from_device(seeds), n_threads=0)
    got = u32(ranges)
    assert (got == exp).all()
    ok = exp[:, 0] <= exp[:, 1]
    assert 0.85 < ok.mean() < 0.95
    assert int((exp[ok, 0] >= np.uint32(1 << 31)).sum()) > 100000           # rows beyond 2^31 are exercised
    rows = ranges[:, 0][torch.from_numpy(ok).to(cuda)].contiguous()
    pos = nvb.locate(idx, rows)
    epos = host.locate(exp[ok, 0], n_threads=0)
    assert (u32(pos) == epos).all()
    # and the property the reference's own test checks (fmindex_test.cu:636-657): the text at a located position is the seed
    sample = torch.arange(0, rows.numel(), max(rows.numel() // 4096, 1), device=cuda)
    p = pos[sample].to(torch.int64) & 0xFFFFFFFF
    hs = O.StringSet.from_device(seeds)
    okidx = np.nonzero(ok)[0][sample.cpu().numpy()]
    for j, q in zip(okidx[:512], p[:512].cpu().numpy()):
        assert (text[int(q):int(q) + 22].cpu().numpy() == O.unpack(hs.words, int(hs.begin[j]), 22, hs.bits, hs.big_endian)).all()
    del idx
    torch.cuda.empty_cache()


def test_config3_at_50m_seeds(cuda, genome_3gbp):
    """BASELINE config 3 as written: 50 M 22-bp exact seeds on the 3 Gbp index -- match + locate on the index the loaders build by default
    (FMIndexDevice.hbm_default) and on the reference layout.  A sample of 1.2 M seeds spread over the 50 M goes against the oracle on a host copy;
    all 50 M are held to the properties the domain offers: the two layouts agree range for range and position for position; a seed drawn from the
    genome is found (non-empty range) and the text at EVERY located position spells the seed (what the reference's own test checks on a sample,
    fmindex_test.cu:636-657, here on all ~45 M located rows by a packed compare on the device); a range's first row locates to a position whose
    22-mer is the smallest suffix carrying the seed, so locating the LAST row of the range must spell the seed too."""
    text, fmi, host = genome_3gbp
    n = 50_000_000
    seeds = W.make_seeds(text, n, 22)
    rich, desc = fmi.hbm_default()
    assert desc["line_native"] and desc["sa_int"] < 16
    r_lean = nvb.match(fmi, seeds)
    r_rich = nvb.match(rich, seeds)
    assert torch.equal(r_lean, r_rich)
    lo, hi = r_lean[:, 0].to(torch.int64) & 0xFFFFFFFF, r_lean[:, 1].to(torch.int64) & 0xFFFFFFFF
    found = lo <= hi
    share = float(found.float().mean().item())
    assert 0.85 < share < 0.95, share                                   # 90 % of the seeds are drawn from the genome, the rest are random 22-mers
    rows_first = r_lean[:, 0][found].contiguous()
    rows_last = r_lean[:, 1][found].contiguous()
    p_lean = nvb.locate(fmi, rows_first)
    p_rich = nvb.locate(rich, rows_first)
    assert torch.equal(p_lean, p_rich)
    p_last = nvb.locate(rich, rows_last)
    # the text at every located position is the seed: compare 22 symbols as a packed 44-bit integer, in chunks
    sym = seeds.words          # packed 2-bit big-endian words; decode the seeds on the device the same way the text is packed
    idx_found = torch.nonzero(found).squeeze(1)
    begin = seeds.begin[idx_found]
    def spell(words, big_endian_bits2, start):           # 22 symbols from symbol offset `start` of a 2-bit big-endian word stream -> int64
        out = torch.zeros_like(start)
        for k in range(22):
            s = start + k
            w = words[(s >> 4)].to(torch.int64) & 0xFFFFFFFF
            out = (out << 2) | ((w >> (30 - 2 * (s & 15))) & 3)
        return out
    assert seeds.bits == 2 and seeds.big_endian
    chunk = 1 << 23
    ar = torch.arange(22, device=cuda)
    for s0 in range(0, idx_found.numel(), chunk):
        e0 = min(idx_found.numel(), s0 + chunk)
        want = spell(seeds.words, True, begin[s0:e0])
        for p in (p_lean[s0:e0], p_last[s0:e0]):
            q = (p.to(torch.int64) & 0xFFFFFFFF)
            t = text[q.unsqueeze(1) + ar.unsqueeze(0)].to(torch.int64)
            got = torch.zeros(e0 - s0, dtype=torch.int64, device=cuda)
            for k in range(22):
                got = (got << 2) | t[:, k]
            assert torch.equal(got, want)
    # the sample against the oracle
    step = n // 1_200_000
    pick = torch.arange(0, n, step, device=cuda)[:1_200_000]
    sub = nvb.PackedStringSet(seeds.words, 2, True, seeds.begin[pick].contiguous(), None, 22)
    exp = host.match(O.StringSet.from_device(sub), n_threads=0)
    got = u32(r_lean[pick])
    assert (got == exp).all()
    ok = exp[:, 0] <= exp[:, 1]
    epos = host.locate(exp[ok, 0], n_threads=0)
    pos_all = torch.full((n,), -1, dtype=torch.int32, device=cuda)
    pos_all[idx_found] = p_lean
    assert (u32(pos_all[pick][torch.from_numpy(ok).to(cuda)]) == epos).all()
    del rich
    torch.cuda.empty_cache()


# ------------------------------------------------------------------ three symbols per step
def test_trimer_arrays_equal_model(cuda, index):
    from tests import dimer_model as DM
    text, host, dev = index
    if host.length > 200000:
        pytest.skip("the model enumerates suffixes in Python")
    ft = dev.with_trimer()
    got = u32(ft.trimer)
    exp = DM.build_trimer(text, host.sa, host.L2)
    assert got.size == exp.size
    assert (got[:4] == exp[:4]).all() and (got[64:128] == exp[64:128]).all(), (got[:4], exp[:4])
    assert (got[128:] == exp[128:]).all()


@pytest.mark.parametrize("bits,be", [(2, True), (4, True), (4, False)])
def test_trimer_match_is_identical(cuda, index, bits, be):
    """three symbols per step: every length mod 3, seeds shorter than a triple, an N in any of the three places, absent seeds
    (the raw (x,y) of the emptying step through the smaller-step replay), with and without the k-mer table"""
    text, host, dev = index
    rng = np.random.default_rng(60 + bits * 2 + be)
    ft = dev.with_trimer()
    variants = [ft, ft.with_ktab(5), ft.with_ktab(8)]
    for length in (22, 21, 20, 0, 1, 2, 3, 33):
        hs = make_seeds(rng, text, 20000, length, bits, be, with_n=True)
        exp = host.match(hs)
        ds = nvb.PackedStringSet.from_host(hs.words, bits, be, hs.begin, hs.length, device=cuda)
        for f in variants:
            got = u32(nvb.match(f, ds))
            bad = np.nonzero((got != exp).any(1))[0]
            assert bad.size == 0, (length, f.ktab_k, bad[:5], got[bad[:5]], exp[bad[:5]])
