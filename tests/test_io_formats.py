"""On-disk formats: byte layouts of .bwt/.sa/.wpac/.pac as the reference's loaders and
nvBWT's writers define them (fmindex_impl.cu:119-262, nvBWT.cu:222-353, sequence_pac.cpp:94-190)."""
import os
import struct

import numpy as np
import pytest
import torch

from nvbio_amd import io as nio
from oracle import pyoracle as O


def test_bwt_and_sa_byte_layout(tmp_path):
    rng = np.random.default_rng(1)
    text = rng.integers(0, 4, 1000, dtype=np.uint8)
    h = O.FMIndex(text)
    prefix = str(tmp_path / "g")
    nio.save_fmindex(prefix, h)
    raw = open(prefix + ".bwt", "rb").read()
    primary, c0, c1, c2, c3 = struct.unpack("<5I", raw[:20])
    assert primary == h.primary and c3 == 1000                          # the last cumulative frequency is the length
    assert [c0, c1, c2, c3] == np.cumsum(np.bincount(h.bwt, minlength=4)).tolist()
    assert len(raw) == 20 + 4 * ((1000 + 15) // 16)                      # BWT without '$', 16 symbols per word
    w0 = struct.unpack("<I", raw[20:24])[0]
    assert [(w0 >> (30 - 2 * k)) & 3 for k in range(16)] == h.bwt[:16].tolist()      # big-endian 2-bit symbols
    raw = open(prefix + ".sa", "rb").read()
    f = struct.unpack("<7I", raw[:28])
    assert f[0] == h.primary and f[5] == 16 and f[6] == 1000
    n_ssa = (1000 + 16) // 16
    assert len(raw) == 28 + 4 * (n_ssa - 1)                              # ssa[0] (= -1) is not stored
    assert struct.unpack("<I", raw[28:32])[0] == h.sa[16]
    # readers
    p2, cum, n, words = nio.read_bwt(prefix + ".bwt")
    assert (p2, n) == (h.primary, 1000) and words.size % 4 == 0 and words.size >= 4 * ((n + 63) // 64)
    assert (nio.read_sa(prefix + ".sa", n, p2) == h.ssa).all()
    with pytest.raises(nio.FileMismatch):
        nio.read_sa(prefix + ".sa", n, p2 + 1)
    with pytest.raises(nio.FileMismatch):
        nio.read_sa(prefix + ".sa", n, p2, sa_int=32)


@pytest.mark.parametrize("n", [1, 3, 4, 16, 17, 64, 1001, 1004])
def test_pac_and_wpac(tmp_path, n):
    rng = np.random.default_rng(n)
    sym = rng.integers(0, 4, n, dtype=np.uint8)
    words = O.pack(sym, 2, True)
    p, w = str(tmp_path / "x.pac"), str(tmp_path / "x.wpac")
    nio.write_pac(p, n, words)
    nio.write_wpac(w, n, words)
    assert os.path.getsize(p) == n // 4 + 2 if n % 4 == 0 else os.path.getsize(p) == (n + 3) // 4 + 1      # BWA's size rule
    raw = np.fromfile(p, dtype=np.uint8)
    assert raw[-1] == n % 4
    assert [(int(raw[i // 4]) >> (6 - 2 * (i % 4))) & 3 for i in range(n)] == sym.tolist()
    assert struct.unpack("<Q", open(w, "rb").read(8))[0] == n
    for loader in (nio.read_pac, nio.read_wpac):
        n2, w2 = loader(p if loader is nio.read_pac else w)
        assert n2 == n
        assert [(int(w2[i // 16]) >> (30 - 2 * (i % 16))) & 3 for i in range(n)] == sym.tolist()
    os.remove(w)
    assert nio.load_genome(str(tmp_path / "x"))[0] == n                  # falls back to .pac


@pytest.mark.gpu
def test_index_round_trip_through_files(tmp_path, cuda):
    """write forward + reverse index files, load them with FMIndexDataDevice (occurrence table built on the
    device), and search: identical to the index uploaded directly."""
    import nvbio_amd as nvb
    rng = np.random.default_rng(2)
    text = rng.integers(0, 4, 50_000, dtype=np.uint8)
    h, rh = O.FMIndex(text), O.FMIndex(text[::-1].copy())
    prefix = str(tmp_path / "genome")
    nio.save_fmindex(prefix, h)
    nio.save_fmindex(prefix, rh, reverse=True)
    data = nio.FMIndexDataDevice(prefix, device=cuda)
    # the arrays exactly as the files hold them (lean_index), and the index the loader hands out by default on this device: the same arrays plus the
    # line-native records, a k-mer table and the whole suffix array (every SA row, checked against the oracle's)
    for loaded, rich, host in ((data.lean_index(), data.index(), h), (data.lean_rindex(), data.rindex(), rh)):
        assert (loaded.length, loaded.primary) == (host.length, host.primary)
        assert loaded.L2 == [int(x) for x in host.L2]
        assert (loaded.bwt_occ.cpu().numpy().view(np.uint32) == host.bwt_occ).all()
        assert (loaded.ssa.cpu().numpy().view(np.uint32) == host.ssa).all()
        assert rich.dimer is not None and rich.ktab is not None and rich.sa_int == 1
        assert (rich.ssa.cpu().numpy().view(np.uint32) == host.locate(np.arange(host.length + 1, dtype=np.uint32))).all()
    assert data.description["sa_int"] == 1 and data.description["line_native"]
    seeds = [text[i:i + 20] for i in rng.integers(0, text.size - 20, 500)]
    hs = O.StringSet.from_lists(seeds, 2, True)
    ds = nvb.PackedStringSet.from_host(hs.words, 2, True, hs.begin, hs.length, device=cuda)
    r = nvb.match(data.index(), ds).cpu().numpy().view(np.uint32)
    assert (r == h.match(hs)).all()
    pos = nvb.locate(data.index(), torch.from_numpy(r[:, 0].astype(np.int64)).to(cuda).to(torch.int32)).cpu().numpy().view(np.uint32)
    assert all((text[p:p + 20] == s).all() for p, s in zip(pos, seeds))
    # SA-less load
    d2 = nio.FMIndexDataDevice(prefix, flags=nio.FORWARD, device=cuda)
    assert d2.index().ssa is None and d2.rindex() is None


def test_fastq_to_sequence_data(tmp_path):
    """FASTQ -> SequenceData (sequence_encoder.cpp): base codes, phred conversion, strand encodings and 4-bit BE packing"""
    fq = tmp_path / "r.fastq"
    fq.write_text("@r1 extra\nACGTNacgtn\n+\nIIII#5555!\n@r2\nTTGCA\n+r2\nABCDE\n")
    d = nio.read_fastq(str(fq))
    assert d.size() == 2 and d.names == ["r1", "r2"] and d.sequence_index.tolist() == [0, 10, 15]
    assert d.symbols.tolist() == [0, 1, 2, 3, 4, 0, 1, 2, 3, 4, 3, 3, 2, 1, 0]
    assert d.quals[:10].tolist() == [40, 40, 40, 40, 2, 20, 20, 20, 20, 0] and d.quals[10:].tolist() == [32, 33, 34, 35, 36]
    r = nio.read_fastq(str(fq), flags=nio.SEQ_REVERSE)                       # what nvBowtie loads
    assert r.symbols[:10].tolist() == [4, 3, 2, 1, 0, 4, 3, 2, 1, 0] and r.quals[:10].tolist() == [0, 20, 20, 20, 20, 2, 40, 40, 40, 40]
    both = nio.read_fastq(str(fq), flags=nio.SEQ_FORWARD | nio.SEQ_REVERSE_COMPLEMENT, max_reads=1)
    assert both.size() == 2 and both.symbols[10:].tolist() == [4, 0, 1, 2, 3, 4, 0, 1, 2, 3]     # rc: N stays N
    p64 = nio.read_fastq(str(fq), quality_encoding=nio.PHRED64)
    assert p64.quals[10] == (ord("A") - 64)
    # the packed stream is what the kernels read: symbol i of the set at bits 28 - 4*(i % 8) of word i / 8
    import torch
    from nvbio_amd.strings import pack_symbols
    w = pack_symbols(torch.from_numpy(d.symbols), 4, True).numpy().view(np.uint32)
    assert [(int(w[i // 8]) >> (28 - 4 * (i % 8))) & 15 for i in range(15)] == d.symbols.tolist()
    with pytest.raises(IOError):
        bad = tmp_path / "bad.fastq"; bad.write_text("@x\nACGT\n+\nII\n"); nio.read_fastq(str(bad))


@pytest.mark.gpu
def test_files_to_sam_example(tmp_path, cuda):
    """index files + genome + FASTQ -> SAM through tools/align_fastq.py: reads come back at the positions they were drawn
    from, on the right strand, with CIGARs that consume the read"""
    import io as _io, sys, re
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import align_fastq
    rng = np.random.default_rng(3)
    text = rng.integers(0, 4, 200_000, dtype=np.uint8)
    prefix = str(tmp_path / "g")
    nio.save_fmindex(prefix, O.FMIndex(text))
    nio.write_wpac(prefix + ".wpac", text.size, O.pack(text, 2, True))
    n = 300
    lens = np.where(np.arange(n) % 3 == 0, 76, 100)                      # reads of two lengths in one file
    pos = rng.integers(0, text.size - 100, n)
    with open(prefix + ".fastq", "w") as f:
        for i, p in enumerate(pos):
            L = int(lens[i])
            r = text[p:p + L].copy()
            mut = rng.random(L) < 0.03
            r[mut] = (r[mut] + 1) & 3
            if i % 2:
                r = (3 - r)[::-1]
            f.write("@read%d\n%s\n+\n%s\n" % (i, "".join("ACGT"[c] for c in r), "I" * L))
    buf = _io.StringIO()
    align_fastq.main(prefix, prefix + ".fastq", buf, device=cuda)
    lines = [ln.split("\t") for ln in buf.getvalue().splitlines() if not ln.startswith("@")]
    assert len(lines) == n
    aligned = [ln for ln in lines if ln[1] != "4"]
    assert len(aligned) > 0.9 * n
    good = 0
    for ln in aligned:
        i = int(ln[0][4:])
        L = int(lens[i])
        consumed = sum(int(k) for k, op in re.findall(r"(\d+)([MIDS])", ln[5]) if op in "MIS")
        assert consumed == L == len(ln[9])
        ok = (int(ln[3]) - 1 == pos[i]) and (ln[1] == ("16" if i % 2 else "0"))
        good += ok
        tags = dict((t.split(":")[0], t.split(":", 2)[2]) for t in ln[11:])
        assert set(tags) == {"NM", "AS", "XM", "XO", "XG", "MD"}          # SamOutput's tag set (output_sam.cpp:354-365)
        if ok and ln[5] == "%dM" % L:
            # an ungapped placement at the origin: NM = XM = the Hamming distance to the genome, AS = -(sum of the mismatch penalties) = -6 each at Q40
            seq = np.array(["ACGT".index(c) for c in ln[9]], dtype=np.uint8)
            ham = int((seq != text[pos[i]:pos[i] + L]).sum())
            assert int(tags["NM"]) == ham == int(tags["XM"]) and tags["XO"] == tags["XG"] == "0" and int(tags["AS"]) == -6 * ham
            assert sum(int(x) for x in re.findall(r"\d+", tags["MD"])) + ham == L
    assert good > 0.95 * len(aligned)


def test_bns_files_round_trip(tmp_path):
    """.ann / .amb (BWA 0.6.1's text format, as bnt.cpp:83-163 reads it): names with and without comments, offsets, holes"""
    prefix = str(tmp_path / "ref")
    nio.write_bns(prefix, ["chr1", "chr2", "chrM"], [1000, 2500, 300], annos=["first one", "", "mito genome"], holes=[(10, 5, "N"), (1200, 3, "R")])
    assert open(prefix + ".ann").read().splitlines()[:3] == ["3800 3 11", "0 chr1 first one", "0 1000 1"]
    b = nio.read_bns(prefix)
    assert b.names == ["chr1", "chr2", "chrM"] and b.annos == ["first one", "", "mito genome"] and b.l_pac == 3800
    assert b.offsets == [0, 1000, 3500] and b.lengths == [1000, 2500, 300] and b.n_ambs == [1, 1, 0] and b.holes == [(10, 5, "N"), (1200, 3, "R")]
    assert b.sequence_index() == [0, 1000, 3500, 3800]                      # SequenceData's sequence index (sequence_pac.cpp:214-226)
    assert b.locate(999) == (0, 999) and b.locate(1000) == (1, 0) and b.locate(3799) == (2, 299)
    with open(prefix + ".amb", "w") as f:
        f.write("3801 3 0\n")
    with pytest.raises(nio.FileMismatch):
        nio.read_bns(prefix)


@pytest.mark.gpu
def test_files_to_sam_multi_sequence_reference(tmp_path, cuda):
    """a reference of three sequences (.ann / .amb next to the index): @SQ per sequence, RNAME and POS relative to the read's sequence,
    in the best-mapping and the all-mapping flows; all-mapping drops seeds that straddle two sequences"""
    import io as _io, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import align_fastq
    rng = np.random.default_rng(41)
    lens = [30_000, 50_000, 20_000]
    text = rng.integers(0, 4, sum(lens), dtype=np.uint8)
    prefix = str(tmp_path / "g")
    nio.save_fmindex(prefix, O.FMIndex(text))
    nio.write_wpac(prefix + ".wpac", text.size, O.pack(text, 2, True))
    nio.write_bns(prefix, ["chrA", "chrB", "chrC"], lens)
    starts = [0, 30_000, 80_000]
    n = 90
    where = []
    with open(prefix + ".fastq", "w") as f:
        for i in range(n):
            k = i % 3
            p = int(rng.integers(0, lens[k] - 100))
            r = text[starts[k] + p: starts[k] + p + 100].copy()
            if i % 2:
                r = (3 - r)[::-1]
            where.append((k, p))
            f.write("@read%d\n%s\n+\n%s\n" % (i, "".join("ACGT"[c] for c in r), "I" * 100))
    for flow in (align_fastq.main, align_fastq.main_all):
        buf = _io.StringIO()
        flow(prefix, prefix + ".fastq", buf, device=cuda)
        text_out = buf.getvalue().splitlines()
        assert [ln for ln in text_out if ln.startswith("@SQ")] == ["@SQ\tSN:chrA\tLN:30000", "@SQ\tSN:chrB\tLN:50000", "@SQ\tSN:chrC\tLN:20000"]
        recs = [ln.split("\t") for ln in text_out if not ln.startswith("@")]
        assert len(recs) == n
        for ln in recs:
            k, p = where[int(ln[0][4:])]
            assert ln[2] == "chr" + "ABC"[k] and int(ln[3]) - 1 == p and ln[5] == "100M", ln[:6]


@pytest.mark.gpu
def test_files_to_sam_all_mapping_example(tmp_path, cuda):
    """tools/align_fastq.py --all: a genome with a 3-copy element; reads from the element come back once per copy (MAPQ 255), unique reads once"""
    import io as _io, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import align_fastq
    rng = np.random.default_rng(31)
    text = rng.integers(0, 4, 120_000, dtype=np.uint8)
    copies = [10_000, 50_000, 90_000]
    for c in copies[1:]:
        text[c:c + 500] = text[10_000:10_500]
    prefix = str(tmp_path / "g")
    nio.save_fmindex(prefix, O.FMIndex(text))
    nio.write_wpac(prefix + ".wpac", text.size, O.pack(text, 2, True))
    n = 60
    pos = [int(rng.integers(10_000, 10_400)) if i % 2 == 0 else int(rng.integers(20_000, 40_000)) for i in range(n)]
    with open(prefix + ".fastq", "w") as f:
        for i, p in enumerate(pos):
            r = text[p:p + 100].copy()
            if i % 4 >= 2:
                r = (3 - r)[::-1]
            f.write("@read%d\n%s\n+\n%s\n" % (i, "".join("ACGT"[c] for c in r), "I" * 100))
    buf = _io.StringIO()
    align_fastq.main_all(prefix, prefix + ".fastq", buf, device=cuda)
    lines = [ln.split("\t") for ln in buf.getvalue().splitlines() if not ln.startswith("@")]
    by_read = {}
    for ln in lines:
        assert ln[4] == "255" and ln[5] == "100M" and ln[1] == ("16" if int(ln[0][4:]) % 4 >= 2 else "0")
        by_read.setdefault(int(ln[0][4:]), []).append(int(ln[3]) - 1)
    for i, p in enumerate(pos):
        expect = sorted(c + p - 10_000 for c in copies) if i % 2 == 0 else [p]
        assert sorted(by_read[i]) == expect, (i, by_read[i], expect)


@pytest.mark.gpu
def test_paired_files_to_sam_example(tmp_path, cuda):
    """index files + two FASTQ files of FR mates -> SAM through the paired-end driver: both records of a pair carry the paired flags, point at each
    other (PNEXT / TLEN) and sit at the fragment's two ends; a pair whose second mate is junk keeps mate 1 as an unpaired alignment"""
    import io as _io, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import align_fastq
    rng = np.random.default_rng(5)
    text = rng.integers(0, 4, 200_000, dtype=np.uint8)
    prefix = str(tmp_path / "g")
    nio.save_fmindex(prefix, O.FMIndex(text))
    nio.write_wpac(prefix + ".wpac", text.size, O.pack(text, 2, True))
    L, n = 100, 120
    pos = rng.integers(0, text.size - 600, n); flen = rng.integers(220, 420, n)
    with open(prefix + "_1.fastq", "w") as f1, open(prefix + "_2.fastq", "w") as f2:
        for i in range(n):
            a = text[pos[i]:pos[i] + L].copy(); b = text[pos[i] + flen[i] - L:pos[i] + flen[i]].copy()
            a[rng.integers(0, L, 2)] ^= 1; b[rng.integers(0, L, 1)] ^= 2
            b = (3 - b)[::-1]
            if i % 10 == 0:
                b = rng.integers(0, 4, L, dtype=np.uint8)                     # junk mate 2
            for f, r in ((f1, a), (f2, b)):
                f.write("@frag%d\n%s\n+\n%s\n" % (i, "".join("ACGT"[c] for c in r), "I" * L))
    buf = _io.StringIO()
    align_fastq.main_paired(prefix, prefix + "_1.fastq", prefix + "_2.fastq", buf, device=cuda)
    lines = [ln.split("\t") for ln in buf.getvalue().splitlines() if not ln.startswith("@")]
    assert len(lines) == 2 * n
    proper = 0
    for i in range(n):
        a, b = lines[2 * i], lines[2 * i + 1]
        assert a[0] == b[0] == "frag%d" % i
        fa, fb = int(a[1]), int(b[1])
        if i % 10 == 0:
            assert fb == 4 and (fa & 0x1) and (fa & 0x8) and not (fa & 0x2) and abs(int(a[3]) - 1 - pos[i]) <= 2          # mate 1 alone, mate flagged unmapped (two substitutions on the first two bases cost more than a 2-base gap: the alignment then starts 2 later)
            continue
        if (fa & 0x2) and (fb & 0x2):
            proper += 1
            assert (fa & 0x41) == 0x41 and (fb & 0x81) == 0x81 and bool(fa & 0x10) != bool(fb & 0x10) and bool(fa & 0x20) == bool(fb & 0x10)
            assert int(a[3]) - 1 == pos[i] and int(b[3]) - 1 == pos[i] + flen[i] - L
            assert int(a[7]) == int(b[3]) and int(b[7]) == int(a[3]) and int(a[8]) == flen[i] == -int(b[8])
    assert proper >= 0.9 * (n - n // 10)


def test_bam_round_trip(tmp_path):
    """SAM records -> BAM (BGZF blocks, binary records with BamOutput's tag types) -> parsed back: every field survives"""
    rng = np.random.default_rng(9)
    lines = ["@HD\tVN:1.0\tSO:unsorted", "@SQ\tSN:ref\tLN:200000", "@PG\tID:nvbio_amd\tPN:nvbio_amd"]
    recs = []
    for i in range(700):                                       # enough for more than one BGZF block
        L = int(rng.integers(30, 151))
        seq = "".join("ACGTN"[c] for c in rng.integers(0, 5, L)); qual = "".join(chr(33 + int(q)) for q in rng.integers(0, 42, L))
        if i % 7 == 0:
            # unmapped records, some without stored sequence ('*') or qualities ('*' = 0xFF bytes): the forms secondary /
            # unmapped lines of tools/align_fastq.py may take
            if i % 21 == 0:
                seq, qual = "*", "*"
            elif i % 14 == 0:
                qual = "*"
            recs.append(("u%d" % i, 4, "*", 0, 0, "*", "*", 0, 0, seq, qual, {}))
            lines.append("u%d\t4\t*\t0\t0\t*\t*\t0\t0\t%s\t%s" % (i, seq, qual))
            continue
        cig = "%dS%dM1D%dM" % (3, L - 10, 7) if i % 3 == 0 else "%dM" % L
        tags = dict(NM=int(rng.integers(0, 20)), AS=-int(rng.integers(0, 300)), XM=int(rng.integers(0, 9)), XO=int(rng.integers(0, 3)), XG=int(rng.integers(0, 3)), MD="12A7^CG030")
        pos, pn, tl = int(rng.integers(1, 190000)), int(rng.integers(1, 190000)), int(rng.integers(-500, 500))
        recs.append(("r%d" % i, 99 if i % 2 else 147, "ref", pos, int(rng.integers(0, 43)), cig, "=", pn, tl, seq, qual, tags))
        r = recs[-1]
        lines.append("\t".join(str(x) for x in r[:11]) + "\tNM:i:%d\tAS:i:%d\tXM:i:%d\tXO:i:%d\tXG:i:%d\tMD:Z:%s" % tuple(tags[k] for k in ("NM", "AS", "XM", "XO", "XG", "MD")))
    path = str(tmp_path / "t.bam")
    nio.sam_to_bam("\n".join(lines) + "\n", path, spec_bins=True)
    text, refs, got = nio.read_bam(path)
    assert refs == [("ref", 200000)] and text.splitlines() == lines[:3] and len(got) == len(recs)
    for g, r in zip(got, recs):
        want_seq = "" if r[9] == "*" else r[9]
        want_qual = "" if r[9] == "*" else (chr(255 + 33) * len(r[9]) if r[10] == "*" else r[10])
        assert (g["name"], g["flag"], g["pos"], g["mapq"], g["cigar"], g["pnext"], g["tlen"], g["seq"], g["qual"]) == (r[0], r[1], r[3], r[4], r[5], r[7], r[8], want_seq, want_qual)
        assert g["ref"] == (0 if r[2] == "ref" else -1) and g["next_ref"] == (0 if r[6] == "=" else -1)
        if r[11]:
            assert g["tags"]["MD"] == r[11]["MD"] and g["tags"]["NM"] == r[11]["NM"] and g["tags"]["AS"] == r[11]["AS"] & 0xFFFFFFFF
    # the record layout again, by an independent walk over the raw bytes: block sizes chain exactly to the end of the
    # stream, l_seq / n_cigar / l_read_name locate every field, and the bin is reg2bin of the aligned interval
    import gzip, struct
    raw = gzip.open(path, "rb").read()
    o = 12 + struct.unpack_from("<i", raw, 4)[0]
    for _ in range(struct.unpack_from("<i", raw, o - 4)[0]):
        o += 8 + struct.unpack_from("<i", raw, o)[0]
    for r in recs:
        bs = struct.unpack_from("<i", raw, o)[0]
        rid, pos0, l_name, mapq, bin_, n_cig, flag, l_seq = struct.unpack_from("<iiBBHHHI", raw, o + 4)
        assert raw[o + 36:o + 36 + l_name] == r[0].encode() + b"\0" and l_seq == (0 if r[9] == "*" else len(r[9]))
        fixed = 32 + l_name + 4 * n_cig + (l_seq + 1) // 2 + l_seq
        assert fixed <= bs
        if r[5] != "*":
            import re
            span = sum(int(l) for l, op in re.findall(r"(\d+)([MIDNSHP=X])", r[5]) if op in "MDN=X")
            b0, e0 = pos0, pos0 + span - 1
            want = next((off + (b0 >> sh) for sh, off in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)) if b0 >> sh == e0 >> sh), 0)
            assert bin_ == want
        o += 4 + bs
    assert o == len(raw)
    assert open(path, "rb").read()[-28:] == bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")      # the BGZF end-of-file block


def test_native_sam_writer_equals_the_python_one(tmp_path):
    """include/nvbio_hip/sam.h (the C++ host layer's SAM records: what a run of millions of reads is written with) against the Python
    formatter of tools/align_fastq.py on synthetic driver output: unaligned reads, both strands, reads with N, CIGARs with every operation,
    MDS streams with mismatches / insertions / deletions / 255+ match runs, alignments that bridge two reference sequences, READ_1 as an
    extra flag.  (The Python formatter is the one the nvBowtie-equality tests validated record by record.)"""
    import io as _io
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import align_fastq as AF
    rng = np.random.default_rng(5)
    n = 3000
    lens = rng.integers(30, 160, n)
    index = np.zeros(n + 1, np.int64); index[1:] = np.cumsum(lens)
    symbols = rng.integers(0, 4, int(index[-1])).astype(np.uint8)
    symbols[rng.random(symbols.size) < 0.01] = 4
    quals = rng.integers(2, 41, int(index[-1])).astype(np.uint8)
    names = ["read_%d/x" % i for i in range(n)]
    nio.write_bns(str(tmp_path / "g"), ["chrA", "chrB", "chrC"], [5000, 300, 4700])

    class Ref(AF.Reference):
        pass
    ref = Ref(str(tmp_path / "g"), 10_000, "ref")
    best = np.zeros((2, n), np.uint64)
    mapq = rng.integers(0, 43, n).astype(np.uint8)
    cig = np.zeros((n, 64), np.uint16); clen = np.zeros(n, np.uint32)
    source = np.zeros((n, 2), np.uint32); mds = np.zeros((n, 256), np.uint8)
    for i in range(n):
        if i % 11 == 0:
            best[0, i] = (0xFFFFFFFF << 32) | (int(rng.integers(0, 4)) << 28)
            continue
        pos = int(rng.integers(0, 9_900))
        score = int(rng.integers(-300, 301))
        w = (1 if score < 0 else 0) | (abs(score) << 1) | (int(rng.integers(0, 40)) << 18) | (int(rng.integers(0, 2)) << 28)
        best[0, i] = (pos << 32) | w
        k = int(rng.integers(1, 9))
        ops = rng.integers(0, 4, k); ops[0] = 0
        for j in range(k):
            cig[i, j] = int(ops[j]) | (int(rng.integers(1, 200)) << 2)
        clen[i] = k
        source[i] = (int(rng.integers(0, 20)), int(rng.integers(0, 5)))
        toks = []
        for _ in range(int(rng.integers(1, 12))):
            op = int(rng.choice([0, 0, 0, 1, 2, 3]))
            if op == 0:
                toks += [0, int(rng.choice([1, 17, 200, 255]))]
            elif op == 1:
                toks += [1, int(rng.integers(0, 5))]
            else:
                l = int(rng.integers(1, 5)); toks += [op, l] + [int(x) for x in rng.integers(0, 5, l)]
        total = 2 + len(toks)
        mds[i, 0], mds[i, 1] = total & 0xFF, total >> 8
        mds[i, 2:total] = toks
    for flags in (0, 64):
        buf = _io.StringIO()
        AF.write_records_se(buf, ref, names, symbols, index, quals, best, mapq, cig, clen, source, mds, extra_flags=flags)
        path = str(tmp_path / ("native_%d.sam" % flags))
        AF.write_records_se_native(path, ref, names, symbols, index, quals, best, mapq, cig, clen, source, mds, extra_flags=flags)
        text = open(path).read()
        assert text == ref.header() + buf.getvalue()
    assert "\t4\t" in text or "\t68\t" in text


def test_native_sam_writer_stays_inside_short_rows(tmp_path):
    """A CIGAR or MD string longer than its row (cigar_len > cigar_stride, the MDS length field > mds_stride: a long read with many edits against short
    strides) was cut at the row's end by the stage that wrote it; the writer must not walk into the next read's row (include/nvbio_hip/sam.h): the
    CIGAR is printed up to the row's end, the MD field of a cut string is '*', and the neighbours' records are untouched."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import align_fastq as AF
    nio.write_bns(str(tmp_path / "g"), ["chrA"], [10_000])
    ref = AF.Reference(str(tmp_path / "g"), 10_000, "ref")
    n, L = 3, 20
    index = np.arange(0, (n + 1) * L, L, dtype=np.int64)
    symbols = np.tile(np.array([0, 1, 2, 3], np.uint8), n * L // 4)
    quals = np.full(n * L, 30, np.uint8)
    names = ["a", "b", "c"]
    best = np.zeros((2, n), np.uint64)
    for i in range(n):
        best[0, i] = ((100 + 50 * i) << 32) | (10 << 1)
    mapq = np.full(n, 42, np.uint8)
    cig = np.zeros((n, 4), np.uint16); clen = np.array([2, 9, 2], np.uint32)            # read b claims 9 ops in a row of 4
    for i in range(n):
        cig[i] = [0 | (5 << 2), 0 | (7 << 2), 0 | (3 << 2), 0 | (5 << 2)]
    source = np.zeros((n, 2), np.uint32)
    mds = np.zeros((n, 8), np.uint8)
    mds[0, :4] = [4, 0, 0, 20]                                                              # "20"
    mds[1, :8] = [44, 1, 0, 9, 1, 2, 0, 9]                                                  # claims 300 bytes in a row of 8
    mds[2, :6] = [6, 0, 0, 12, 1, 3]                                                        # "12T"
    path = str(tmp_path / "short_rows.sam")
    AF.write_records_se_native(path, ref, names, symbols, index, quals, best, mapq, cig, clen, source, mds)
    recs = [l.split("\t") for l in open(path).read().splitlines() if not l.startswith("@")]
    assert [r[0] for r in recs] == names
    assert recs[0][5] == "7M5M" and recs[0][-1] == "MD:Z:20"
    assert recs[1][5] == "5M3M7M5M" and recs[1][-1] == "MD:Z:*"                           # the four ops of its own row, no more; a cut MD string is not printed
    assert recs[2][5] == "7M5M" and recs[2][-1] == "MD:Z:12T"
