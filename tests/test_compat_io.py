"""CPU suite for the drop-in layer's alignment writers (include/nvbio_hip/compat/nvbio/io/output: SamOutput, BamOutput behind
OutputFile::open): tests/compat/io_callers.hip builds a HostOutputBatchSE / PE from flat arrays -- host code only -- and writes it as
SAM and as BAM; the SAM text is compared with lines formed here the way nvbio/io/output/output_sam.cpp:372-520 forms them, the BAM
records (read back with nvbio_amd.io.read_bam) with the SAM text."""
import ctypes as C
import os

import numpy as np
import pytest

from nvbio_amd import io as nio

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "tests", "compat", "libio_callers.so")


class SlotSet(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("words", "aligns", "cigar_offsets", "cigar_ops", "cigar_source", "mds_offsets", "mds_bytes", "mapq")]


class WriteArgs(C.Structure):
    _fields_ = [("path", C.c_char_p), ("paired", C.c_uint32), ("n_ref", C.c_uint32), ("ref_names", C.c_char_p), ("ref_lengths", C.c_void_p),
                ("n", C.c_uint32), ("read_len", C.c_uint32), ("names", C.c_char_p * 2), ("bases", C.c_void_p * 2), ("quals", C.c_void_p * 2),
                ("slots", SlotSet * 2)]


@pytest.fixture(scope="module")
def lib():
    assert os.path.exists(LIB), "build with python -c 'import __graft_entry__ as g; g.build()'"
    return C.CDLL(LIB)


REF = [("chrA", 5000), ("chrB", 3000)]
L = 60
SHAPES = [[(0, 60)], [(3, 5), (0, 55)], [(0, 20), (1, 2), (0, 38)], [(0, 30), (2, 3), (0, 30)], [(0, 10), (1, 1), (0, 20), (2, 2), (0, 25), (3, 4)]]     # (op, len): M I D S = 0 1 2 3


def make_slot(rng, n, mate_bit, paired):
    """random alignments: (arrays for the caller, per-read dicts for the expectation)"""
    words, aligns, src, mapq, recs = [], [], [], [], []
    cig_off, cig_ops, mds_off, mds = [0], [], [0], []
    starts = np.cumsum([0] + [l for _, l in REF])
    for i in range(n):
        ops = SHAPES[int(rng.integers(0, len(SHAPES)))]
        span = sum(l for t, l in ops if t in (0, 2))
        k = int(rng.integers(0, len(REF)))
        aligned = rng.random() < 0.9
        over = rng.random() < 0.05                                        # an alignment that runs over the end of its sequence
        pos = int(starts[k + 1] - span + 3) if over else int(starts[k] + rng.integers(16, REF[k][1] - span))
        s = int(rng.integers(0, 16))
        rc, score, ed = int(rng.integers(0, 2)), int(rng.integers(-200, 120)), int(rng.integers(0, 40))
        pair_bit, disc = (int(rng.integers(0, 2)), int(rng.integers(0, 2))) if paired else (0, 0)
        w = (1 if score < 0 else 0) | (abs(score) << 1) | (ed << 18) | (rc << 28) | (mate_bit[i] << 29) | (pair_bit << 30) | (disc << 31)
        words.append(w); aligns.append(pos - s if aligned else 0xFFFFFFFF); src.append(s); mapq.append(int(rng.integers(0, 43)))
        cig_ops += [(t | (l << 2)) for t, l in ops[::-1]]; cig_off.append(len(cig_ops))
        # an MD program: matches, a mismatch, an insertion and a deletion in random order (output_sam.cpp:233-314 reads it token by token)
        body = []
        for tok in rng.permutation(5):
            body += [[0, int(rng.integers(1, 60))], [1, int(rng.integers(0, 4))], [2, 2, 0, 1], [3, 2, 2, 3], [0, 200, 0, 100]][tok]
        prog = [(len(body) + 2) & 255, (len(body) + 2) >> 8] + body
        mds += prog; mds_off.append(len(mds))
        recs.append(dict(aligned=aligned, pos=pos, rc=rc, score=score, ed=ed, ops=ops, span=span, mapq=mapq[-1], mds=np.array(prog, np.uint8), seq=k,
                         mate=mate_bit[i], concordant=bool(pair_bit and not disc), over=aligned and pos + span > starts[k + 1]))
    arr = dict(words=np.array(words, np.uint32), aligns=np.array(aligns, np.uint32), cigar_offsets=np.array(cig_off, np.uint32),
               cigar_ops=np.array(cig_ops, np.uint16), cigar_source=np.array(src, np.uint32), mds_offsets=np.array(mds_off, np.uint32),
               mds_bytes=np.array(mds, np.uint8), mapq=np.array(mapq, np.uint8))
    return arr, recs


def run(lib, path, paired, n, reads, slots):
    a = WriteArgs()
    a.path, a.paired, a.n_ref = str(path).encode(), int(paired), len(REF)
    a.ref_names = b"\0".join(nm.encode() for nm, _ in REF) + b"\0"
    lens = np.array([l for _, l in REF], np.uint32); a.ref_lengths = lens.ctypes.data
    a.n, a.read_len = n, L
    keep = [lens]
    for m, (names, bases, quals) in enumerate(reads):
        nb = b"\0".join(names) + b"\0"
        a.names[m] = nb; a.bases[m] = bases.ctypes.data; a.quals[m] = quals.ctypes.data; keep += [nb, bases, quals]
    for s, arr in enumerate(slots):
        for k, v in arr.items():
            setattr(a.slots[s], k, v.ctypes.data)
        keep.append(arr)
    assert lib.write_alignments(C.byref(a)) == 0


def make_reads(rng, n, tag):
    names = [("%s%d" % (tag, i)).encode() for i in range(n)]
    bases = np.frombuffer(b"ACGTN", np.uint8)[rng.choice(5, (n, L), p=[0.24, 0.24, 0.24, 0.24, 0.04])].copy()
    quals = (33 + rng.integers(2, 41, (n, L))).astype(np.uint8)
    return names, bases, quals


def text_of(rec, read, strand_of_word):
    names, bases, quals = read
    comp = bytes.maketrans(b"ACGTN", b"TGCAN")
    def seq_qual(i):
        s, q = bases[i].tobytes(), quals[i].tobytes()
        return (s.translate(comp)[::-1], q[::-1]) if strand_of_word else (s, q)
    return seq_qual


def cigar_text(ops):
    return "".join("%d%s" % (l, "MIDS"[t]) for t, l in ops)


def test_single_end_sam_lines_and_bam_records(lib, tmp_path):
    rng = np.random.default_rng(90)
    n = 400
    reads = make_reads(rng, n, "read")
    arr, recs = make_slot(rng, n, [0] * n, paired=False)
    sam, bam = tmp_path / "o.sam", tmp_path / "o.bam"
    run(lib, sam, False, n, [reads], [arr]); run(lib, bam, False, n, [reads], [arr])
    starts = np.cumsum([0] + [l for _, l in REF])
    comp = bytes.maketrans(b"ACGTN", b"TGCAN")
    lines = [ln.rstrip("\n") for ln in open(sam) if not ln.startswith("@")]
    header = [ln for ln in open(sam) if ln.startswith("@")]
    assert header[0] == "@HD\tVN:1.3\n" and "@SQ\tSN:chrA\tLN:5000\n" in header and "@SQ\tSN:chrB\tLN:3000\n" in header
    assert len(lines) == n
    n_over = 0
    for i, (ln, r) in enumerate(zip(lines, recs)):
        s, q = reads[1][i].tobytes(), reads[2][i].tobytes()
        if r["rc"]:
            s, q = s.translate(comp)[::-1], q[::-1]
        if not r["aligned"]:
            want = "%s\t4\t*\t0\t0\t*\t*\t0\t0\t%s\t%s" % (reads[0][i].decode(), s.decode(), q.decode())
        else:
            md, mm, gapo, gape = nio.sam_md_string(r["mds"])
            flags = 64 | (16 if r["rc"] else 0) | (4 if r["over"] else 0)
            n_over += r["over"]
            want = "%s\t%d\t%s\t%d\t%d\t%s\t*\t0\t0\t%s\t%s\tNM:i:%d\tAS:i:%d\tXM:i:%d\tXO:i:%d\tXG:i:%d\tMD:Z:%s" % (
                reads[0][i].decode(), flags, REF[r["seq"]][0], r["pos"] - starts[r["seq"]] + 1, 0 if r["over"] else r["mapq"], cigar_text(r["ops"]),
                s.decode(), q.decode(), r["ed"], r["score"], mm, gapo, gape, md or "*")
        assert ln == want, (i, ln, want)
    assert n_over > 5
    text_h, refs, brecs = nio.read_bam(str(bam))
    assert refs == REF and len(brecs) == n and text_h.startswith("@HD\tVN:1.3\n@PG\tID:id\tPN:test\tVN:0.1\n")
    for ln, b, r in zip(lines, brecs, recs):
        f = ln.split("\t")
        assert b["name"] == f[0] and b["seq"] == f[9] and b["qual"] == f[10]
        if not r["aligned"] or r["over"]:
            # BamOutput: no reference, no position, no CIGAR, no tags for an unmapped record (output_bam.cpp:296-340); the other flags stay
            assert b["ref"] == -1 and b["pos"] == 0 and b["cigar"] == "*" and not b["tags"] and (b["flag"] & 4)
            continue
        assert b["flag"] == int(f[1]) and refs[b["ref"]][0] == f[2] and b["pos"] == int(f[3]) and b["mapq"] == int(f[4]) and b["cigar"] == f[5]
        assert {k: str(v) for k, v in b["tags"].items()} == dict((t.split(":")[0], t.split(":", 2)[2]) for t in f[11:])


def test_paired_end_mate_fields_sam_and_bam(lib, tmp_path):
    rng = np.random.default_rng(91)
    n = 300
    reads = [make_reads(rng, n, "pair"), make_reads(rng, n, "pair")]
    anchor_mate = rng.integers(0, 2, n)
    a0, r0 = make_slot(rng, n, list(anchor_mate), paired=True)
    a1, r1 = make_slot(rng, n, list(1 - anchor_mate), paired=True)
    sam, bam = tmp_path / "p.sam", tmp_path / "p.bam"
    run(lib, sam, True, n, reads, [a0, a1]); run(lib, bam, True, n, reads, [a0, a1])
    starts = np.cumsum([0] + [l for _, l in REF])
    lines = [ln.rstrip("\n").split("\t") for ln in open(sam) if not ln.startswith("@")]
    assert len(lines) == 2 * n
    for i in range(n):
        for k, (me, other) in enumerate(((r0[i], r1[i]), (r1[i], r0[i]))):
            f = lines[2 * i + k]
            assert f[0] == "pair%d" % i
            if not me["aligned"]:
                assert f[1] == "4" and f[2] == "*"
                continue
            flags = (128 if me["mate"] else 64) | (16 if me["rc"] else 0) | 1 | (2 if other["concordant"] else 0) | (0 if other["aligned"] else 8) | \
                    (32 if other["rc"] else 0) | (4 if me["over"] else 0)
            assert int(f[1]) == flags, (i, k, f[1], flags)
            assert f[2] == REF[me["seq"]][0] and int(f[3]) == me["pos"] - starts[me["seq"]] + 1
            if other["aligned"]:
                assert f[6] == ("=" if other["seq"] == me["seq"] else REF[other["seq"]][0]) and int(f[7]) == other["pos"] - starts[other["seq"]] + 1
                if other["seq"] == me["seq"]:
                    tlen = max(other["pos"] + other["span"], me["pos"] + me["span"]) - min(other["pos"], me["pos"])
                    assert int(f[8]) == (-tlen if other["pos"] < me["pos"] else tlen)
                else:
                    assert int(f[8]) == 0
            else:
                assert f[6] == "=" and int(f[7]) == int(f[3]) and int(f[8]) == 0
    _, refs, brecs = nio.read_bam(str(bam))
    assert len(brecs) == 2 * n
    names = [r[0] for r in refs]
    for f, b in zip(lines, brecs):
        assert b["name"] == f[0] and b["seq"] == f[9] and b["qual"] == f[10]
        if int(f[1]) & 4:
            assert b["ref"] == -1 and (b["flag"] & 4)
            continue
        assert b["flag"] == int(f[1]) and names[b["ref"]] == f[2] and b["pos"] == int(f[3]) and b["cigar"] == f[5]
        assert names[b["next_ref"]] == (f[2] if f[6] == "=" else f[6]) and b["pnext"] == int(f[7]) and b["tlen"] == int(f[8])


def test_sam_writer_thread_and_recycled_buffers(lib, tmp_path, monkeypatch):
    """SamOutput hands a batch's text to a writer thread and formats the next batch into the buffers the previous one was written from
    (compat/nvbio/io/output/output_writer.h): several batches through one file come out in order, byte for byte what the writes inside
    process() (NVBIO_HIP_SYNC_OUTPUT=1) give, single-end and paired, also when a batch is larger than the buffers the first one left."""
    rng = np.random.default_rng(92)
    n = 700
    reads = make_reads(rng, n, "read")
    arr, _ = make_slot(rng, n, [0] * n, paired=False)
    preads = [make_reads(rng, n, "pair"), make_reads(rng, n, "pair")]
    anchor_mate = rng.integers(0, 2, n)
    a0, _ = make_slot(rng, n, list(anchor_mate), paired=True)
    a1, _ = make_slot(rng, n, list(1 - anchor_mate), paired=True)

    def body(path):
        text = open(path, "rb").read().split(b"\n")
        return [ln for ln in text if ln and not ln.startswith(b"@")]

    for paired, rd, slots in ((False, [reads], [arr]), (True, preads, [a0, a1])):
        per_batch = n * (2 if paired else 1)
        monkeypatch.delenv("NVBIO_IO_CALLERS_REPEAT", raising=False)
        monkeypatch.setenv("NVBIO_HIP_SYNC_OUTPUT", "1")
        one = tmp_path / ("one%d.sam" % paired)
        run(lib, one, paired, n, rd, slots)
        want = body(one)
        assert len(want) == per_batch
        for sync in ("1", "0"):
            monkeypatch.setenv("NVBIO_HIP_SYNC_OUTPUT", sync)
            monkeypatch.setenv("NVBIO_IO_CALLERS_REPEAT", "5")
            many = tmp_path / ("many%d_%s.sam" % (paired, sync))
            run(lib, many, paired, n, rd, slots)
            got = body(many)
            assert len(got) == 5 * per_batch
            for k in range(5):
                assert got[k * per_batch:(k + 1) * per_batch] == want, (paired, sync, k)
