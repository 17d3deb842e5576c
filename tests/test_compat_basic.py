"""CPU suite for the small building blocks the round-4 boundary work added to the drop-in template layer
(include/nvbio_hip/compat): priority_deque, the range forms of rank4 / rank_all with comp(), max_text_gaps -- host-compiled
(tests/compat/host_basic.cpp, g++) -- and the reference's own rank test (nvbio-test/rank_test.cu compiled as it lies against the
layer, oracle/_ref/ref_rank_test, built where /root/reference exists)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import pyoracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "tests", "compat", "libhost_basic.so")


@pytest.fixture(scope="module")
def hb():
    assert os.path.exists(LIB), "build with python -c 'import __graft_entry__ as g; g.build()'"
    return C.CDLL(LIB)


def p(a):
    return a.ctypes.data_as(C.c_void_p)


def test_priority_deque_replays_the_reference_heap(hb):
    """33 k push / pop_top / pop_bottom operations recorded from the reference's compiled interval_heap.h: the drop-in
    priority_deque holds the same array after every one (which of several equal-sized hits surfaces first included)."""
    z = np.load(os.path.join(ROOT, "tests", "golden", "hit_deque_vectors.npz"))
    arrs = [np.ascontiguousarray(z[k]) for k in ("ops", "vals", "caps", "sizes", "states", "case_start")]
    assert hb.replay(*[p(a) for a in arrs], C.c_int(len(arrs[5]) - 1)) == 0


def test_priority_deque_builds_the_reference_heap_over_any_array(hb):
    """priority_deque(seq, constructed = false) -- what nvBowtie's selection kernels construct at every round over hits whose ranges shrank in
    place -- leaves the arrangement the reference's make_interval_heap leaves: 4 000 arrays of 0..63 words, few distinct range sizes (ties decide
    everything), against the oracle's restatement and, where oracle/_ref is built, the reference header itself."""
    rng = np.random.default_rng(911)
    n_cases, stride = 4000, 64
    sizes = rng.integers(0, 64, n_cases).astype(np.uint32)
    maxsz = rng.choice([2, 3, 5, 16, 1 << 19], n_cases)
    data = ((rng.integers(0, 1 << 30, (n_cases, stride)) % maxsz[:, None]).astype(np.uint64) << np.uint64(32)) | rng.integers(0, 1 << 32, (n_cases, stride)).astype(np.uint64)
    data = np.ascontiguousarray(data)
    want = data.copy()
    for c in range(n_cases):
        O.hit_deque_make(want[c], int(sizes[c]))
    ref_so = os.path.join(ROOT, "oracle", "_ref", "libref_hit_deque.so")
    if os.path.exists(ref_so):
        ref = C.CDLL(ref_so)
        again = data.copy()
        for c in range(n_cases):
            ref.ref_hit_deque_make(again[c].ctypes.data_as(C.POINTER(C.c_uint64)), C.c_uint32(int(sizes[c])))
        assert (again == want).all()
    hb.heapify_arrays(p(data), p(sizes), C.c_int(n_cases), C.c_uint32(stride))
    assert (data == want).all()


def test_range_rank4_rank_all_on_host(hb):
    rng = np.random.default_rng(23)
    n = 20011
    text = rng.integers(0, 4, n, dtype=np.uint8)
    host = O.FMIndex(text)
    nb = (n + 63) // 64 + 1
    occ = np.zeros(nb * 4, dtype=np.uint32)
    cum = np.zeros((n + 1, 4), dtype=np.int64)
    for c in range(4):
        cum[1:, c] = np.cumsum(host.bwt == c)
    for k in range(nb):
        occ[4 * k: 4 * k + 4] = cum[min(64 * k, n)]
    bw = np.concatenate([O.pack(host.bwt, 2, True, pad_words=0), np.zeros(8, np.uint32)])
    pr = host.primary
    lo = rng.integers(0, n + 1, 20000).astype(np.int64)
    hi = np.minimum(n, lo + rng.choice([0, 1, 5, 40, 63, 64, 200, 5000], lo.size))
    lo = np.concatenate([lo, [-1, -1, 0, pr - 1, pr, pr - 1, n - 1, n, 62, 63, -1]]).astype(np.uint32)
    hi = np.concatenate([hi, [n, 0, 0, pr, pr, pr + 1, n, n, 63, 64, pr]]).astype(np.uint32)
    out_lo, out_hi = np.zeros((lo.size, 4), np.uint32), np.zeros((lo.size, 4), np.uint32)
    L2 = host.L2.astype(np.uint32)
    bad = hb.rank_ranges(C.c_uint32(n), C.c_uint32(pr), p(L2), p(bw), p(occ), C.c_uint32(lo.size), p(lo), p(hi), p(out_lo), p(out_hi))
    assert bad == 0
    assert (out_lo == host.rank4(lo)).all() and (out_hi == host.rank4(hi)).all()


def test_max_gaps():
    hb = C.CDLL(LIB)
    out = np.zeros(3, np.uint32)
    # nvBowtie local defaults: match 2, gap open 5+3, ext 3 (stored negative); 150-bp mate, min_score 20 + 8 ln(150) = 60
    hb.gaps(2, -8, -3, 60, 150, p(out))
    # 300 - 8 = 292; extensions while score >= 60: 292, 289, ... -> n = 78 steps, result n - 1
    score, k = 300 - 8, 0
    while score >= 60 and k < 150:
        score -= 3; k += 1
    assert out[0] == k - 1 and out[1] == k - 1
    hb.gaps(2, -8, -3, 295, 150, p(out))
    assert out[0] == 0xFFFFFFFF            # the opening alone sinks the score: "steps - 1" wraps (utils_inl.h:176-200)
    hb.gaps(2, -8, -3, 301, 150, p(out))
    assert out[0] == 0
    hb.gaps(2, -8, -3, -7, 150, p(out))
    assert out[2] == 7                      # edit distance: -min_score


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "ref_rank_test")), reason="oracle/_ref/ref_rank_test not built (needs /root/reference)")
def test_reference_rank_test_passes_on_the_drop_in_layer():
    """nvbio-test/rank_test.cu (whole TU, compiled as it lies with -I include/nvbio_hip/compat): rank() and rank_all() at every
    position of a random text against running counts, for uint32 / uint4 / uint64 dictionaries (rank_test.cu:55-232)."""
    r = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "ref_rank_test"), "-length", "200"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-400:]
    assert "rank test... done" in r.stderr and "mismatch" not in r.stderr


def _load_reads(hb, path, flags, qenc=1, max_len=0xFFFFFFFF, trim3=0, trim5=0, batch=1 << 20, recordwise=0, cap=1 << 22):
    index = np.zeros(200001, np.uint32); syms = np.zeros(cap, np.uint8); quals = np.zeros(cap, np.uint8)
    names = np.zeros(cap, np.uint8); info = np.zeros(8, np.uint32)
    n = hb.load_reads(str(path).encode(), C.c_uint32(flags), C.c_uint32(qenc), C.c_uint32(max_len), C.c_uint32(trim3), C.c_uint32(trim5),
                      C.c_uint32(batch), C.c_int(recordwise), p(index), p(syms), p(quals), p(names), C.c_uint32(index.size - 1), C.c_uint32(cap),
                      C.c_uint32(cap), p(info))
    assert n >= 0, n
    total = int(index[n])
    nm = bytes(names[:int(info[3])]).split(b"\0")[:-1] if info[3] else []
    return n, index[:n + 1].copy(), syms[:total].copy(), quals[:total].copy(), nm, info


@pytest.mark.parametrize("flags", [1, 2, 1 | 8, 1 | 2 | 4 | 8])
def test_text_sequence_loader_batch_and_record_forms(hb, tmp_path, flags):
    """The drop-in FASTQ loader (compat/nvbio/io/sequence/sequence.h): its batch form -- lines parsed into flat arrays, symbols packed
    by all OpenMP threads -- and its record-at-a-time form give the same SequenceData, which is the one nvbio_amd.io.read_fastq
    (the restatement of sequence_encoder.cpp checked in test_io_formats.py) builds; several batches, ragged lengths, N, lower case."""
    from nvbio_amd import io as nio
    rng = np.random.default_rng(77 + flags)
    n = 3000
    lens = rng.integers(1, 160, n)
    fq = tmp_path / "r.fastq"
    with open(fq, "wb") as f:
        for i in range(n):
            s = rng.choice(np.frombuffer(b"ACGTNacgt", np.uint8), lens[i]).tobytes()
            q = rng.integers(33, 74, lens[i]).astype(np.uint8).tobytes()
            f.write(b"@read%d\n%s\n+\n%s\n" % (i, s, q))
    want = nio.read_fastq(str(fq), flags=flags)
    for recordwise in (0, 1):
        for batch in (1 << 20, 257):
            got_n, index, syms, quals, names, info = _load_reads(hb, fq, flags, batch=batch, recordwise=recordwise)
            assert got_n == want.size() and (index == want.sequence_index).all()
            assert (syms == want.symbols).all() and (quals == want.quals).all()
            per = bin(flags).count("1")
            assert names == [("read%d" % (i // per)).encode() for i in range(got_n)]
            assert info[4] == 1
    # trimming and truncation are applied per strand copy, before the strand operation
    t_n, t_index, t_syms, t_quals, _, _ = _load_reads(hb, fq, flags, trim3=3, trim5=2, max_len=50)
    r_n, r_index, r_syms, r_quals, _, _ = _load_reads(hb, fq, flags, trim3=3, trim5=2, max_len=50, recordwise=1)
    assert t_n == r_n and (t_index == r_index).all() and (t_syms == r_syms).all() and (t_quals == r_quals).all()
    fwd = nio.read_fastq(str(fq), flags=1)
    if flags == 1:
        for i in (0, 1, 2, n - 1):
            b, e = int(fwd.sequence_index[i]), int(fwd.sequence_index[i + 1])
            keep = min(max(e - b - 5, 0), 50)
            assert (t_syms[t_index[i]:t_index[i + 1]] == fwd.symbols[b + 2:b + 2 + keep]).all()


def test_text_sequence_loader_tolerates_line_breaks_and_blank_lines(hb, tmp_path):
    """CR LF line ends, records wrapped over several lines, blank lines between records, a FASTA record: both forms agree with the
    plain 4-line text"""
    plain = tmp_path / "p.fastq"; odd = tmp_path / "o.fastq"
    plain.write_bytes(b"@a x\nACGTACGTAC\n+\nIIIIIIIIII\n@b\nTTNNA\n+\n#####\n@c\nG\n+\n5\n")
    odd.write_bytes(b"@a x\r\nACGTA\r\nCGTAC\r\n+a x\r\nIIIII\r\nIIIII\r\n\r\n\n@b\nTTNNA\n+\n#####\n\n@c\nG\n+\n5")
    ref = _load_reads(hb, plain, 2)
    for recordwise in (0, 1):
        got = _load_reads(hb, odd, 2, recordwise=recordwise)
        assert got[0] == ref[0] == 3 and all((a == b).all() for a, b in zip(got[1:4], ref[1:4])) and got[4] == ref[4] == [b"a x", b"b", b"c"]
    bad = tmp_path / "bad.fastq"; bad.write_bytes(b"@x\nACGT\n+\nII\n")
    for recordwise in (0, 1):
        assert _load_reads(hb, bad, 1, recordwise=recordwise)[5][4] == 0          # is_ok() turns false: incomplete read


@pytest.mark.parametrize("flags", [1, 2, 1 | 2 | 4 | 8])
def test_reads_out_of_sam_and_bam_files(hb, tmp_path, flags):
    """open_sequence_file on a .sam / .bam name (compat AlignmentSequenceFile; sequence_sam.cpp:405-495, sequence_bam.cpp:232-388): header lines
    and secondary alignments are skipped, a record flagged reverse-complemented is turned back per requested strand, SAM qualities are
    phred + 33 and BAM qualities plain phred, a missing QUAL reads as phred 0.  The BAM file is the SAM text through nvbio_amd.io.sam_to_bam."""
    from nvbio_amd import io as nio
    rng = np.random.default_rng(5 + flags)
    n = 500
    recs, lines = [], ["@HD\tVN:1.3", "@SQ\tSN:chr1\tLN:100000", "@PG\tID:x"]
    for i in range(n):
        ln = int(rng.integers(1, 120))
        seq = bytes(rng.choice(np.frombuffer(b"ACGTN", np.uint8), ln, p=[0.24, 0.24, 0.24, 0.24, 0.04]))
        star = rng.random() < 0.05
        qual = b"*" if star else bytes((33 + rng.integers(0, 41, ln)).astype(np.uint8))
        fl = int(rng.choice([0, 16, 4, 256, 256 | 16, 1 | 64, 1 | 128 | 16]))
        lines.append("r%d\t%d\tchr1\t%d\t30\t%dM\t*\t0\t0\t%s\t%s\tNM:i:0" % (i, fl, 1 + int(rng.integers(0, 90000)), ln, seq.decode(), qual.decode()))
        if not fl & 256:
            recs.append((("r%d" % i).encode(), seq, np.zeros(ln, np.uint8) if star else np.frombuffer(qual, np.uint8) - 33, bool(fl & 16)))
    sam = tmp_path / "reads.sam"
    sam.write_text("\n".join(lines) + "\n")
    bam = tmp_path / "reads.bam"
    nio.sam_to_bam(sam.read_text(), str(bam))
    code = np.full(256, 4, np.uint8)
    for k, c in enumerate(b"ACGT"):
        code[c] = k; code[c + 32] = k
    names, syms, quals, lens = [], [], [], []
    for name, seq, q, rc in recs:
        s = code[np.frombuffer(seq, np.uint8)]
        comp = np.where(s < 4, 3 - s, 4).astype(np.uint8)
        # (requested strand) -> what is stored for a forward record / for a reverse-complemented one
        for flag, fwd, rev in ((1, (s, q), (comp[::-1], q[::-1])), (2, (s[::-1], q[::-1]), (comp, q)), (4, (comp, q), (s[::-1], q[::-1])), (8, (comp[::-1], q[::-1]), (s, q))):
            if flags & flag:
                a, b = rev if rc else fwd
                names.append(name); syms.append(a); quals.append(b); lens.append(len(a))
    for path in (sam, bam):
        for batch in (1 << 20, 97):
            got_n, index, gs, gq, gnames, info = _load_reads(hb, path, flags, batch=batch)
            assert got_n == len(lens) and (np.diff(index.astype(np.int64)) == np.array(lens)).all()
            assert (gs == np.concatenate(syms)).all() and (gq == np.concatenate(quals)).all()
            assert gnames == names and info[4] == 1


def test_reads_out_of_a_txt_file(hb, tmp_path):
    """one read per line (sequence_txt.cpp): no names, every base with the best quality '~' (phred 93 under Phred33), empty lines skipped; plain or gzip"""
    import gzip
    rng = np.random.default_rng(12)
    reads = [bytes(rng.choice(np.frombuffer(b"ACGTNacgt", np.uint8), int(rng.integers(1, 90)))) for _ in range(300)]
    text = b"\n".join(r if i % 17 else r + b"\n" for i, r in enumerate(reads)) + b"\n"          # (some blank lines in between)
    plain = tmp_path / "reads.txt"; plain.write_bytes(text)
    packed = tmp_path / "reads2.txt.gz"; packed.write_bytes(gzip.compress(text))
    code = np.full(256, 4, np.uint8)
    for k, c in enumerate(b"ACGT"):
        code[c] = k; code[c + 32] = k
    want = np.concatenate([code[np.frombuffer(r, np.uint8)][::-1] for r in reads])               # io::REVERSE, as nvBowtie loads reads
    for path in (plain, packed):
        got_n, index, gs, gq, gnames, info = _load_reads(hb, path, 2, batch=64)
        assert got_n == len(reads) and (np.diff(index.astype(np.int64)) == np.array([len(r) for r in reads])).all()
        assert (gs == want).all() and (gq == 93).all() and gnames == [b""] * len(reads)


MYERS_CASES = [(31, O.SEMI_GLOBAL, 5, 16, 0), (31, O.SEMI_GLOBAL, 5, 16, -(1 << 30)), (31, O.SEMI_GLOBAL, 5, 32, -12), (31, O.GLOBAL, 5, 32, -40),
               (15, O.SEMI_GLOBAL, 4, 32, -9), (15, O.GLOBAL, 4, 32, -100), (7, O.SEMI_GLOBAL, 2, 32, -6)]


@pytest.mark.parametrize("band,aln_type,alphabet,sink_bits,min_score", MYERS_CASES)
def test_banded_bitvector_edit_distance(hb, band, aln_type, alphabet, sink_bits, min_score):
    """EditDistanceAligner<TYPE, MyersTag<A>> under banded_alignment_score: the drop-in layer's bit-vector band against the oracle's restatement
    of myers_banded_inl.h:236-291 -- distances, sink positions, the int16 threshold (a -2^30 'none' becomes 0), texts no longer than the
    pattern, texts shorter (declined), patterns shorter than the band.  Strings are read through InfixSets over one byte string."""
    rng = np.random.default_rng(band * 131 + alphabet)
    pats, txts = [], []
    for k in range(3000):
        pl = int(rng.integers(1, 90))
        tl = pl + int(rng.integers(-2, band + 6))
        hi = 2 if alphabet == 2 else 4
        t = rng.integers(0, hi, max(tl, 0)).astype(np.uint8)
        off = int(rng.integers(0, max(1, min(band, tl - pl + 1)))) if tl >= pl else 0
        q = list(t[off:off + pl]) if tl >= pl else list(rng.integers(0, hi, pl))
        q += list(rng.integers(0, hi, pl - len(q)))
        for _ in range(int(rng.integers(0, 6))):
            at, r = int(rng.integers(0, len(q))), rng.random()
            if r < 0.5:
                q[at] = (q[at] + 1) % hi
            elif r < 0.75 and len(q) > 1:
                del q[at]
            else:
                q.insert(at, int(rng.integers(0, hi)))
        q = np.array(q, dtype=np.uint8)
        if alphabet == 5 and rng.random() < 0.2:
            q[int(rng.integers(0, q.size))] = 4                    # an N in the read
        pats.append(q); txts.append(t)
    ps, ts = O.StringSet.from_lists(pats, 8, False), O.StringSet.from_lists(txts, 8, False)
    es, ek = O.batch_banded_myers_score(band, aln_type, alphabet, ps, ts, min_score=min_score, sink_bits=sink_bits)
    pat, txt = np.concatenate(pats), np.concatenate(txts + [np.zeros(1, np.uint8)])
    pc = np.stack([ps.begin, ps.begin + ps.length], 1).astype(np.uint32)
    tc = np.stack([ts.begin, ts.begin + ts.length], 1).astype(np.uint32)
    score, sink = np.zeros(len(pats), np.int32), np.zeros((len(pats), 2), np.uint32)
    assert hb.banded_myers(C.c_uint32(band), C.c_int(aln_type), C.c_uint32(alphabet), C.c_int(sink_bits), p(pat), p(txt), p(pc), p(tc), C.c_uint32(len(pats)),
                           C.c_int32(min_score), p(score), p(sink)) == 0
    assert (score == es).all() and (sink == ek).all()
    reported = (ek[:, 0] != 0xFFFFFFFF)
    assert reported.any() and (~reported).any()
    if min_score <= -(1 << 30):                                      # the threshold the batch functions pass: exact occurrences only
        assert (es[reported] == 0).all()


def test_infix_set_over_a_packed_string_set(hb):
    rng = np.random.default_rng(77)
    reads = [rng.integers(0, 5, int(rng.integers(30, 120))).astype(np.uint8) for _ in range(200)]
    hs = O.StringSet.from_lists(reads, 4, True)
    offsets = np.concatenate([hs.begin, [hs.begin[-1] + hs.length[-1]]]).astype(np.uint32)
    coords, want = [], []
    for k in range(1000):
        r = int(rng.integers(0, len(reads))); b = int(rng.integers(0, len(reads[r]) - 22)); e = b + int(rng.integers(1, 23))
        coords.append((r, b, e, 0)); want.append(reads[r][b:e])
    coords = np.array(coords, dtype=np.uint32)
    out, ids = np.zeros(sum(len(w) for w in want), np.uint8), np.zeros(len(want), np.uint32)
    n = hb.read_set_infixes(p(hs.words), p(offsets), C.c_uint32(len(reads)), p(coords), C.c_uint32(len(want)), p(out), p(ids))
    assert n == out.size and (out == np.concatenate(want)).all() and (ids == coords[:, 0]).all()
