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
