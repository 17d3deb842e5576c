"""Checks against REFERENCE code compiled from its own sources (oracle/_ref/libref_basic.so, built by `make -C oracle ref`
where /root/reference exists; the built object travels to the GPU box): nvbio/basic/popcount.h (the counting inside rank),
nvbio/fmindex/bwt.h (gen_sa / gen_bwt_from_sa: SA, BWT and primary conventions), nvbio/basic/priority_deque.h (the hit
deque), nvbio/basic/algorithms.h (upper_bound), nvbio/io/bam_format.h (the BAM record header) and nvbio/basic/bnt.cpp
(.ann / .amb).  They pin the oracle's restatements -- and the product's file writers -- to code the reference ships."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import pyoracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "_ref", "libref_basic.so")
pytestmark = pytest.mark.skipif(not os.path.exists(LIB), reason="oracle/_ref/libref_basic.so not built (needs /root/reference at build time)")


@pytest.fixture(scope="module")
def ref():
    return C.CDLL(LIB)


def p(a):
    return a.ctypes.data_as(C.c_void_p)


def test_rank_counting_equals_the_reference_popcounts(ref):
    """rank on the interleaved layout composed from the reference's popc_2bit / hibits_2bit (rank_dictionary_inl.h:502-513)
    == the oracle's rank == a direct count of the BWT."""
    rng = np.random.default_rng(3)
    n = 40013
    text = rng.integers(0, 4, n, dtype=np.uint8)
    text[500:1500] = 0                                     # c == 0 against zero padding: the correction of popcount_inl.h:343-350
    host = O.FMIndex(text)
    idx = np.concatenate([rng.integers(0, n, 30000), np.arange(0, 200), [n - 1, 0xFFFFFFFF]]).astype(np.uint32)
    c = rng.integers(0, 4, idx.size).astype(np.uint8)
    out = np.zeros(idx.size, np.uint32)
    ref.ref_dict_rank(p(host.bwt_occ), p(idx), p(c), idx.size, p(out))
    # the same dictionary index seen through fm_index::rank: rows at or after primary are shifted by the '$' row
    k = np.where(idx == 0xFFFFFFFF, idx, np.where(idx >= host.primary, idx + 1, idx)).astype(np.uint32)
    assert (out == host.rank(k, c)).all()
    cum = np.stack([np.concatenate([[0], np.cumsum(host.bwt == s)]) for s in range(4)], 1)
    direct = np.where(idx == 0xFFFFFFFF, 0, cum[np.minimum(idx.astype(np.int64) + 1, n), c])
    assert (out == direct).all()
    # the word-level functions themselves
    x = rng.integers(0, 1 << 32, 50000, dtype=np.uint64).astype(np.uint32)
    cc = rng.integers(0, 4, x.size).astype(np.uint8)
    im = rng.integers(0, 16, x.size).astype(np.uint32)
    sym = ((x[:, None] >> (30 - 2 * np.arange(16, dtype=np.uint32))[None, :]) & 3).astype(np.uint8)          # big-endian symbols of each word
    got = np.zeros(x.size, np.uint32)
    ref.ref_popc_2bit(p(x), p(cc), x.size, p(got))
    assert (got == (sym == cc[:, None]).sum(1)).all()
    ref.ref_popc_2bit_prefix(p(x), p(cc), p(im), x.size, p(got))          # symbols 0 .. 15 - i_mod of the word
    keep = np.arange(16)[None, :] <= (15 - im)[:, None]
    assert (got == ((sym == cc[:, None]) & keep).sum(1)).all()
    ref.ref_popc_2bit_all(p(x), x.size, p(got))
    packed = sum(((sym == s).sum(1).astype(np.uint32)) << np.uint32(8 * s) for s in range(4))
    assert (got == packed).all()


def test_sa_bwt_primary_conventions_equal_the_reference(ref):
    """gen_sa (sais) + gen_bwt_from_sa (bwt.h:36-60): SA has n+1 rows with SA[0] = n, the '$' row is dropped from the BWT and
    its index is `primary` -- what the oracle's own index construction (numpy prefix doubling) must reproduce."""
    for n, seed in ((1, 0), (2, 1), (77, 2), (5000, 3), (30001, 4)):
        rng = np.random.default_rng(seed)
        text = rng.integers(0, 4, n, dtype=np.uint8)
        if n > 1000:
            text[100:400] = 2
        sa = np.zeros(n + 1, np.int32)
        bwt = np.zeros(n + 1, np.uint8)
        primary = ref.ref_gen_sa_bwt(n, p(text), p(sa), p(bwt))
        host = O.FMIndex(text)
        assert (sa.astype(np.uint32) == host.sa).all()
        assert primary == host.primary
        assert (bwt[:n] == host.bwt).all()
    t1, t2 = np.zeros(256, np.uint32), np.zeros(256, np.uint32)
    ref.ref_gen_bwt_count_table(p(t1))
    for b in range(256):
        t2[b] = sum(1 << (8 * ((b >> (2 * s)) & 3)) for s in range(4))
    assert (t1 == t2).all()


def test_hit_deque_equals_the_reference_priority_deque(ref):
    """priority_deque<SeedHit, vector_view, hit_compare> (priority_deque.h:329-421) replayed next to the oracle's restated
    interval heap: same array after every push / pop_top / pop_bottom, top() = the smallest range (array slot 1, or slot 0
    when alone), bottom() = array slot 0 (a largest range) -- the conventions the selection stage relies on."""
    rng = np.random.default_rng(11)
    push, pop_bottom, pop_top = O.hit_deque_ops()
    for trial in range(40):
        n_ops = 400
        ops = rng.choice([0, 0, 0, 1, 2], n_ops).astype(np.uint8)
        delta = rng.integers(1, 40 if trial % 2 else 1 << 20, n_ops).astype(np.uint64)         # many ties in odd trials
        values = (delta << np.uint64(32)) | rng.integers(0, 1 << 31, n_ops).astype(np.uint64)
        storage = np.zeros(n_ops + 1, np.uint64)
        sizes, tops, bottoms = np.zeros(n_ops, np.uint32), np.zeros(n_ops, np.uint64), np.zeros(n_ops, np.uint64)
        ref.ref_priority_deque_replay(p(storage), n_ops, p(ops), p(values), p(sizes), p(tops), p(bottoms))
        a = np.zeros(n_ops + 1, np.uint64)
        size = 0
        for i in range(n_ops):
            if ops[i] == 0:
                a[size] = values[i]; size += 1; push(a, size)
            elif ops[i] == 1 and size:
                pop_top(a, size); size -= 1
            elif ops[i] == 2 and size:
                pop_bottom(a, size); size -= 1
            assert size == sizes[i]
            if size:
                assert bottoms[i] == a[0] and tops[i] == (a[1] if size > 1 else a[0])
                d = (a[:size] >> np.uint64(32)) & np.uint64(0xFFFFF)
                assert ((tops[i] >> np.uint64(32)) & np.uint64(0xFFFFF)) == d.min() and ((bottoms[i] >> np.uint64(32)) & np.uint64(0xFFFFF)) == d.max()
        assert (storage[:size] == a[:size]).all()


def test_upper_bound_equals_the_reference(ref):
    rng = np.random.default_rng(5)
    slots = np.cumsum(rng.integers(0, 5, 5000)).astype(np.uint64)
    keys = rng.integers(0, int(slots[-1]) + 3, 20000).astype(np.uint64)
    out = np.zeros(keys.size, np.uint32)
    ref.ref_upper_bound_u64(p(slots), slots.size, p(keys), keys.size, p(out))
    assert (out == np.searchsorted(slots, keys, side="right")).all()


def test_bam_records_parse_with_the_reference_struct(ref, tmp_path):
    """the fixed part of every record nvbio_amd.io.sam_to_bam writes, read back through io::BAM_alignment (bam_format.h:60-71)"""
    import gzip
    import struct
    from nvbio_amd import io as nio
    rng = np.random.default_rng(2)
    lines = ["@HD\tVN:1.0\tSO:unsorted", "@SQ\tSN:chrA\tLN:500000", "@SQ\tSN:chrB\tLN:90000"]
    want = []
    for i in range(300):
        L = int(rng.integers(20, 120))
        seq = "".join("ACGT"[c] for c in rng.integers(0, 4, L)); qual = "".join(chr(33 + int(q)) for q in rng.integers(0, 42, L))
        if i % 9 == 0:
            lines.append("u%d\t4\t*\t0\t0\t*\t*\t0\t0\t%s\t*" % (i, seq)); want.append((-1, -1, 4680, 0, len("u%d" % i) + 1, 4, 0, L, -1, -1, 0))
            continue
        pos, mq, fl = int(rng.integers(1, 80000)), int(rng.integers(0, 43)), (99 if i % 2 else 147)
        rname = "chrB" if i % 4 == 0 else "chrA"
        cig = "5S%dM2D%dM" % (L - 15, 10) if i % 3 == 0 else "%dM" % L
        span = (L - 15 + 2 + 10) if i % 3 == 0 else L
        lines.append("r%d\t%d\t%s\t%d\t%d\t%s\t=\t%d\t%d\t%s\t%s\tNM:i:1" % (i, fl, rname, pos, mq, cig, pos + 50, 150, seq, qual))
        b0, e0 = pos - 1, pos - 1 + span - 1
        bin_ = next((off + (b0 >> sh) for sh, off in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)) if b0 >> sh == e0 >> sh), 0)
        rid = 1 if rname == "chrB" else 0
        want.append((rid, pos - 1, bin_, mq, len("r%d" % i) + 1, fl, 4 if i % 3 == 0 else 1, L, rid, pos + 49, 150))
    path = str(tmp_path / "x.bam")
    nio.sam_to_bam("\n".join(lines) + "\n", path, spec_bins=True)
    raw = gzip.open(path, "rb").read()
    o = 12 + struct.unpack_from("<i", raw, 4)[0]
    for _ in range(struct.unpack_from("<i", raw, o - 4)[0]):
        o += 8 + struct.unpack_from("<i", raw, o)[0]
    fields = np.zeros(12, np.int32)
    for w in want:
        rec = np.frombuffer(raw[o:o + 36], dtype=np.uint8).copy()
        assert ref.ref_bam_alignment_fields(p(rec), p(fields)) == 36
        got = tuple(int(x) for x in fields[1:])
        assert got == w, (got, w)
        o += 4 + int(fields[0])
    assert o == len(raw)


def test_bns_files_round_trip_through_the_reference(ref, tmp_path):
    """.ann / .amb written by nvbio_amd.io.write_bns load with the reference's load_bns (bnt.cpp:83-163), and files written by its
    save_bns load with nvbio_amd.io.read_bns"""
    from nvbio_amd import io as nio
    names, lengths = ["chr1", "chr2 extra", "scaffold_77"], [1000, 250, 64]
    names_only = ["chr1", "chr2", "scaffold_77"]
    annos = ["", "some annotation", ""]
    holes = [(10, 5, "N"), (1100, 2, "N")]
    prefix = str(tmp_path / "mine")
    nio.write_bns(prefix, names_only, lengths, annos=annos, holes=holes)
    l_pac, n_seqs, seed, n_holes = C.c_int64(), C.c_int32(), C.c_uint32(), C.c_int32()
    off, ln, na, gi = np.zeros(8, np.int64), np.zeros(8, np.int32), np.zeros(8, np.int32), np.zeros(8, np.uint32)
    nm = C.create_string_buffer(4096)
    ho, hl, hc = np.zeros(8, np.int64), np.zeros(8, np.int32), C.create_string_buffer(8)
    assert ref.ref_load_bns(prefix.encode(), C.byref(l_pac), C.byref(n_seqs), C.byref(seed), C.byref(n_holes), p(off), p(ln), p(na), p(gi),
                            nm, 4096, p(ho), p(hl), hc, 8, 8) == 0
    assert (l_pac.value, n_seqs.value, n_holes.value) == (sum(lengths), 3, 2)
    assert list(off[:3]) == [0, 1000, 1250] and list(ln[:3]) == lengths and list(na[:3]) == [1, 1, 0]
    got_names = [ln_.split("\t") for ln_ in nm.value.decode().strip("\n").split("\n")]
    assert [g[0] for g in got_names] == names_only and [g[1].strip() for g in got_names] == annos
    # the .amb body: the reference's reader scans "%lld%d%c" against lines its own writer formats as "%lld %d %c", so %c takes
    # the blank and every later record fails to parse (bnt.cpp:80 vs :152) -- only the first hole's numbers are comparable
    assert int(ho[0]) == 10 and int(hl[0]) == 5
    # the other direction
    prefix2 = str(tmp_path / "theirs")
    arr = (C.c_char_p * 3)(*[s.encode() for s in names_only]); ann = (C.c_char_p * 3)(*[s.encode() for s in annos])
    offs, lens, nambs = np.array([0, 1000, 1250], np.int64), np.array(lengths, np.int32), np.array([1, 1, 0], np.int32)
    hoff, hlen = np.array([10, 1100], np.int64), np.array([5, 2], np.int32)
    assert ref.ref_save_bns(prefix2.encode(), C.c_int64(sum(lengths)), 11, 3, p(offs), p(lens), p(nambs), arr, ann, 2, p(hoff), p(hlen), b"NN") == 0
    b = nio.read_bns(prefix2)
    assert list(b.names) == names_only and [int(x) for x in b.offsets] == [0, 1000, 1250] and [int(x) for x in b.lengths] == lengths
    assert [a.strip() for a in b.annos] == annos and [(int(o), int(l), c) for o, l, c in b.holes] == holes
