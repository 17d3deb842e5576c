"""The reference's OWN test programs, compiled as they lie against the drop-in template layer (`-I include/nvbio_hip/compat`,
tools/ref_bind_check.py --install-ref-tests -> oracle/_ref/ref_*_test; built in the development container, where
/root/reference exists, and carried to the GPU box as binaries), run on the MI355X:

  ref_alignment_test   nvbio-test/alignment_test.cu, whole TU + nvbio-test/alignment_test_utils.h: the banded edit-distance
                       literals, Gotoh / SW / ED score + traceback checked against the test's own reference DP (ref_sw,
                       ref_banded_sw) and CIGAR literals, then every Batched*AlignmentScore scheduler and the per-thread kernel
  ref_sw_benchmark     sw-benchmark/sw-benchmark.cu, whole TU: the reference's headline benchmark program on synthetic FASTQ / FASTA input
  ref_nvBowtie         the whole nvBowtie application (29 TUs as they lie + contrib/crc), FASTQ + index files -> SAM
  ref_fmmap            examples/fmmap/fmmap.cu, whole TU: FM-index filter over infix seed sets + banded bit-vector edit distance, against the oracle
  ref_fmindex_test     nvbio-test/fmindex_test.cu:56-717: SA -> BWT -> occurrence table -> SSA (host, and built on the device from the
                       FM-index alone), match + locate on host and in its device kernel, 32- and 64-bit, separate and interleaved

Each program exits non-zero (exit(1) at the first mismatch) or prints "error" when a check fails."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")
pytestmark = pytest.mark.gpu


def run(name, args, timeout=900):
    exe = os.path.join(REF, name)
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/%s not built (needs /root/reference in the build container)" % name)
    r = subprocess.run([exe] + args, capture_output=True, text=True, timeout=timeout)
    text = (r.stdout + r.stderr).replace("\r", "\n")
    assert r.returncode == 0, text[-1500:]
    assert "error" not in text.lower() and "mismatch" not in text.lower(), text[-1500:]
    return text


def test_reference_alignment_test_passes():
    text = run("ref_alignment_test", ["-N-thread-tasks", "16384", "-N-warp-tasks", "1024"])
    assert "synthetic Edit Distance test 6... passed!" in text
    assert "testing alignment... done" in text
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    open(os.path.join(ROOT, "gpurun_out", "ref_alignment_test.log"), "w").write(text)


def test_reference_fmindex_test_passes():
    text = run("ref_fmindex_test", ["200000", "20000"])
    assert text.count("gpu alignment... done") >= 6          # (sorted + shuffled) x (separate + interleaved) for 32-bit, separate for 64-bit
    assert "fmindex synthetic test... done" in text
    open(os.path.join(ROOT, "gpurun_out", "ref_fmindex_test.log"), "w").write(text)


def test_reference_sw_benchmark_runs(tmp_path):
    """sw-benchmark/sw-benchmark.cu, the program BASELINE's headline numbers come from, compiled as it lies: FASTQ reads and a FASTA
    reference through the drop-in io::open_sequence_file / FASTA_inc_reader, every read against the whole reference with full-matrix
    Gotoh (global / semi-global / local) and edit distance -- BatchedAlignmentScore<AlignmentStream, DeviceThreadScheduler> (tuned
    kernels) and the program's own one-thread-per-read kernel (generic lane code) -- timed and printed by the program itself."""
    import random
    import re
    rnd = random.Random(11)
    n_reads, read_len, ref_len = 20000, 150, 16384
    ref = "".join(rnd.choice("ACGT") for _ in range(ref_len))
    ref_name, reads_name = str(tmp_path / "ref.fa"), str(tmp_path / "reads.fq")
    with open(ref_name, "w") as f:
        f.write(">chr1 synthetic\n" + "\n".join(ref[i:i + 70] for i in range(0, ref_len, 70)) + "\n")
    with open(reads_name, "w") as f:
        for i in range(n_reads):
            p = rnd.randrange(0, ref_len - read_len)
            r = list(ref[p:p + read_len])
            for _ in range(4):
                r[rnd.randrange(read_len)] = rnd.choice("ACGT")
            f.write("@read%d\n%s\n+\n%s\n" % (i, "".join(r), "I" * read_len))
    exe = os.path.join(REF, "ref_sw_benchmark")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/ref_sw_benchmark not built (needs /root/reference in the build container)")
    r = subprocess.run([exe, reads_name, ref_name], capture_output=True, text=True, timeout=900)
    text = (r.stdout + r.stderr).replace("\r", "\n")
    assert r.returncode == 0, text[-1500:]
    assert "%u reads, avg: %u bps, max: %u bps" % (n_reads, read_len, read_len) in text, text[-1500:]
    assert "done (%u bps)" % ref_len in text and "sw-benchmark... done" in text, text[-1500:]
    rows = re.findall(r"(global|semi-global|local)\s*:\s*([0-9.]+)\s+([0-9.]+) GCUPS", text)
    assert len(rows) == 4, text[-1500:]                      # Gotoh x 3 + edit distance semi-global
    assert all(float(a) > 0 and float(b) > 0 for _, a, b in rows), rows
    open(os.path.join(ROOT, "gpurun_out", "ref_sw_benchmark.log"), "w").write(text)


def _write_reference(tmp_path, rng, n_genome, names_and_lengths):
    """index + genome + annotation files of a synthetic reference, the way nvBWT leaves them: <prefix>.bwt / .sa / .rbwt / .rsa /
    .wpac / .ann / .amb (made with the oracle's suffix sorting: test infrastructure)"""
    import numpy as np
    from nvbio_amd import io as nio
    from oracle import pyoracle as O
    text = rng.integers(0, 4, n_genome, dtype=np.uint8)
    prefix = str(tmp_path / "genome")
    nio.save_fmindex(prefix, O.FMIndex(text))
    nio.save_fmindex(prefix, O.FMIndex(text[::-1].copy()), reverse=True)
    nio.write_wpac(prefix + ".wpac", text.size, O.pack(text, 2, True))
    nio.write_bns(prefix, [n for n, _ in names_and_lengths], [l for _, l in names_and_lengths])
    return prefix, text



def test_reference_fmmap_runs_and_matches_the_oracle(tmp_path):
    """examples/fmmap/fmmap.cu, whole TU as it lies (SURVEY 8(b)'s caller list): reads in both strands -> seeds every 10 bp as an InfixSet over the
    packed read set -> FMIndexFilterDevice::rank / locate -> diagonals -> windows -> batch_banded_alignment_score<31> with the bit-vector
    edit-distance aligner into BestSink<int16> -> best per read.  Its own 'aligned % reads' line must equal the same pipeline run through the CPU
    oracle (FM-index match + locate, banded bit-vector distance with the reference's int16 threshold: the batch function's 'no threshold' -2^30
    narrows to 0, so only windows holding the read exactly report a score -- myers_banded_inl.h:243, batched_inl.h:946)."""
    import re
    import numpy as np
    from oracle import pyoracle as O
    exe = os.path.join(REF, "ref_fmmap")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/ref_fmmap not built (needs /root/reference in the build container)")
    rng = np.random.default_rng(41)
    n_genome, n, L = 200_000, 4000, 100
    prefix, text = _write_reference(tmp_path, rng, n_genome, [("chrA", 120_000), ("chrB", 80_000)])
    pos = rng.integers(300, n_genome - L - 300, n)
    reads = []
    for i, q in enumerate(pos):
        r = text[q:q + L].copy()
        if i % 3:                                                  # a third of the reads are exact copies
            m = rng.random(L) < 0.01; r[m] = (r[m] + 1) & 3
        if i % 7 == 0:
            r[int(rng.integers(0, L))] = 4                         # an N
        reads.append((3 - r)[::-1] if i % 2 and r.max() < 4 else r)
    fastq = str(tmp_path / "reads.fastq")
    with open(fastq, "w") as f:
        for i, r in enumerate(reads):
            f.write("@read%d\n%s\n+\n%s\n" % (i, "".join("ACGTN"[c] for c in r), "I" * L))
    r = subprocess.run([exe, prefix, fastq], capture_output=True, text=True, timeout=900)
    out = (r.stdout + r.stderr).replace("\r", "\n")
    assert r.returncode == 0, out[-2000:]
    m = re.findall(r"aligned\s+([0-9.]+) % reads", out)
    assert m, out[-2000:]
    occ = re.findall(r"occurrences\s+:\s+([0-9.]+) B", out)
    # ---- the same pipeline on the CPU oracle
    host = O.FMIndex(text)
    both = []
    for r_ in reads:                                               # io::FORWARD | io::REVERSE_COMPLEMENT: strings 2i and 2i + 1 (N stays N)
        both.append(r_); both.append(np.where(r_ < 4, 3 - r_, 4)[::-1].astype(np.uint8))
    seeds, owner, offs = [], [], []
    for sid, s in enumerate(both):
        for b in range(0, L - 22 + 1, 10):
            seeds.append(s[b:b + 22]); owner.append(sid); offs.append(b)
    ranges = host.match(O.StringSet.from_lists(seeds, 4, True))
    pats, txts, who = [], [], []
    n_hits = 0
    for k, (lo, hi) in enumerate(ranges):
        if lo > hi:
            continue
        rows = np.arange(lo, hi + 1, dtype=np.uint32)
        n_hits += rows.size
        for tp in host.locate(rows):
            diag = int(tp) - offs[k]
            gb = diag - 15 if diag > 15 else 0
            ge = min(gb + L + 31, n_genome)
            pats.append(both[owner[k]]); txts.append(text[gb:ge]); who.append(owner[k] // 2)
    score, _ = O.batch_banded_myers_score(31, O.SEMI_GLOBAL, 5, O.StringSet.from_lists(pats, 4, True), O.StringSet.from_lists(txts, 2, True), sink_bits=16)
    best = np.full(n, -32768, np.int64)
    np.maximum.at(best, np.array(who), score)
    # (the program hands update_scores the HIT count, not the segment count reduce_by_key returns (fmmap.cu:376-393): the tail of its zero-initialised
    # out_reads / out_scores vectors is folded in too, i.e. read 0 of a batch always ends with a score of at least 0 once a read has two hits)
    if n_hits > len(set(who)):
        best[0] = max(best[0], 0)
    expected = 100.0 * float((best >= -20).sum()) / n
    assert 20.0 < expected < 60.0
    assert abs(float(m[-1]) - expected) < 0.006, (m[-1], expected)
    if occ:
        assert abs(float(occ[-1]) - n_hits * 1e-9) < 1.1e-3
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    open(os.path.join(ROOT, "gpurun_out", "ref_fmmap.log"), "w").write(out + "\noracle: %.4f %% of %d reads aligned, %d seed hits\n" % (expected, n, n_hits))

def _simulated_run(tmp_path, seed, n, L, indel_rate=0.0):
    """a two-sequence 200 kbp reference on disk and a FASTQ file of n reads of L bases drawn from it: 3 % substitutions, every other read
    reverse-complemented, and with indel_rate a one- or two-base insertion or deletion in the middle of a read"""
    import numpy as np
    rng = np.random.default_rng(seed)
    n_genome = 200_000
    prefix, text = _write_reference(tmp_path, rng, n_genome, [("chrA", 120_000), ("chrB", 80_000)])
    pos = rng.integers(0, n_genome - L - 4, n)
    pos = np.where((pos < 120_000) & (pos + L + 4 > 120_000), pos - L - 4, pos)   # no read across the two sequences
    fastq = str(tmp_path / "reads.fastq")
    with open(fastq, "w") as f:
        for i, p in enumerate(pos):
            r = text[p:p + L].copy()
            if rng.random() < indel_rate:
                k, g = int(rng.integers(30, 70)), int(rng.integers(1, 3))
                if rng.random() < 0.5: r = np.concatenate([r[:k], rng.integers(0, 4, g).astype(np.uint8), r[k:]])[:L]      # insertion in the read
                else:                  r = np.concatenate([text[p:p + k], text[p + k + g:p + L + g]])                    # deletion from the read
            mut = rng.random(L) < 0.03
            r[mut] = (r[mut] + 1) & 3
            if i % 2:
                r = (3 - r)[::-1]
            f.write("@read%d\n%s\n+\n%s\n" % (i, "".join("ACGT"[c] for c in r), "I" * L))
    return prefix, fastq, pos


def test_reference_nvbowtie_runs_end_to_end(tmp_path):
    """nvBowtie ITSELF -- all 29 translation units of the reference's application compiled as they lie against the drop-in layer and linked
    with libnvbio_hip.so (tools/nvbowtie_tu_check.py --link) -- aligning a FASTQ file against index files on the MI355X and writing SAM:
    its own drivers, queues, selection, reduction and reporting code on top of this repository's templates and kernels.  Reads drawn from
    a two-sequence synthetic reference with 3 % substitutions must come back at their origin, on the right strand, with CIGARs that consume
    the read, in the reference's tag set."""
    import re
    import numpy as np
    exe = os.path.join(REF, "ref_nvBowtie")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/ref_nvBowtie not built (needs /root/reference in the build container)")
    n, L = 4000, 100
    prefix, fastq, pos = _simulated_run(tmp_path, 3, n, L)
    sam = str(tmp_path / "out.sam")
    r = subprocess.run([exe, "--file-ref", "-x", prefix, "-U", fastq, "-S", sam] + os.environ.get("NVBOWTIE_EXTRA_ARGS", "").split(), capture_output=True, text=True, timeout=900)
    log = (r.stdout + r.stderr).replace("\r", "\n")
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    open(os.path.join(ROOT, "gpurun_out", "ref_nvbowtie.log"), "w").write(log[-20000:])
    assert r.returncode == 0, log[-3000:]
    lines = [ln.rstrip("\n").split("\t") for ln in open(sam) if not ln.startswith("@")]
    header = [ln for ln in open(sam) if ln.startswith("@")]
    assert any(h.startswith("@SQ\tSN:chrA\tLN:120000") for h in header) and any(h.startswith("@SQ\tSN:chrB\tLN:80000") for h in header)
    assert len(lines) == n, (len(lines), log[-2000:])
    aligned = [ln for ln in lines if not (int(ln[1]) & 4)]
    assert len(aligned) > 0.95 * n, len(aligned)
    good = 0
    for ln in aligned:
        i = int(ln[0][4:])
        consumed = sum(int(k) for k, op in re.findall(r"(\d+)([MIDS])", ln[5]) if op in "MIS")
        assert consumed == L == len(ln[9])
        origin = int(pos[i])
        chrom, off = ("chrA", origin) if origin < 120_000 else ("chrB", origin - 120_000)
        good += (ln[2] == chrom and int(ln[3]) - 1 == off and bool(int(ln[1]) & 16) == bool(i % 2))
        tags = set(t.split(":")[0] for t in ln[11:])
        assert tags == {"NM", "AS", "XM", "XO", "XG", "MD"}
    assert good > 0.97 * len(aligned), (good, len(aligned))
    open(os.path.join(ROOT, "gpurun_out", "ref_nvbowtie_sam_head.txt"), "w").write("".join(header) + "\n".join("\t".join(ln) for ln in lines[:20]) + "\n")


def test_own_pipeline_equals_reference_nvbowtie(tmp_path, cuda):
    """The strongest parity statement this repository can make: the reference's nvBowtie APPLICATION -- its own drivers, queues, selection,
    reduction, MAPQ, traceback and finishing code, compiled unchanged on top of the drop-in layer -- and this repository's from-scratch
    driver (nvbio_amd.aligner.best_approx over the C-ABI, tools/align_fastq.py) align the same FASTQ file against the same index files, and
    every SAM record agrees: position, strand, MAPQ, CIGAR, sequence, and the NM / AS / XM / XO / XG / MD tags.  A quarter of the reads carry an
    indel so that gapped alignments (where the two once disagreed on AS: the gap costs missing from finish_alignment's restatement) are compared."""
    import io as _io
    import sys
    exe = os.path.join(REF, "ref_nvBowtie")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/ref_nvBowtie not built (needs /root/reference in the build container)")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import align_fastq
    n, L = 6000, 100
    prefix, fastq, pos = _simulated_run(tmp_path, 17, n, L, indel_rate=0.25)
    sam = str(tmp_path / "ref.sam")
    r = subprocess.run([exe, "--file-ref", "-x", prefix, "-U", fastq, "-S", sam], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    ref = [ln.rstrip("\n").split("\t") for ln in open(sam) if not ln.startswith("@")]
    buf = _io.StringIO()
    align_fastq.main(prefix, fastq, buf, device=cuda)
    own = [ln.split("\t") for ln in buf.getvalue().splitlines() if not ln.startswith("@")]
    assert len(ref) == len(own) == n
    differ = []
    gapped = 0
    for a, b in zip(ref, own):
        a = list(a); a[1] = str(int(a[1]) & ~64)              # nvBowtie flags single-end reads READ_1 (output_sam.cpp:431); the example writer does not
        gapped += ("I" in a[5] or "D" in a[5])
        if a != b:
            differ.append((a[:6] + a[11:], b[:6] + b[11:]))
    assert gapped > 0.15 * n, gapped
    assert not differ, (len(differ), differ[:3])


@pytest.mark.parametrize("mode", ["local", "all", "paired"])
def test_own_drivers_equal_reference_nvbowtie_in_every_mode(mode, cuda):
    """The same comparison for nvBowtie --local, --all and paired-end (-1 / -2, FR, 200-400 bp fragments): tools/nvbowtie_compare.py runs the
    reference's application and this repository's driver of that mode (best_approx with local=True, all_mapping, best_approx_paired) on
    one simulated input and compares the SAM records; all of them must be identical."""
    import argparse
    import sys
    if not os.path.exists(os.path.join(REF, "ref_nvBowtie")):
        pytest.skip("oracle/_ref/ref_nvBowtie not built (needs /root/reference in the build container)")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import nvbowtie_compare
    same, n_ref, n_own = nvbowtie_compare.compare(argparse.Namespace(mode=mode, reads=3000, seed=23, indels=0.2, show=3))
    assert n_ref == n_own and n_ref >= 3000
    assert same == n_ref, (mode, same, n_ref)


@pytest.mark.parametrize("paired", [False, True], ids=["single-end", "paired-end"])
def test_reference_nvbowtie_writes_bam(tmp_path, paired):
    """`-S out.bam`: the drop-in layer's BamOutput (compat/nvbio/io/output/output_bam.h: BGZF blocks of binary records in BamOutput's
    layout, output_bam.cpp:234-519) against the SAM text of the same run, field by field -- name, flags, reference, position, MAPQ,
    CIGAR, mate fields, TLEN, bases, qualities, the NM / AS / XM / XO / XG / MD tags; the file must read back as a gzip stream that
    ends with the BGZF end-of-file block."""
    import numpy as np
    from nvbio_amd import io as nio
    exe = os.path.join(REF, "ref_nvBowtie")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/ref_nvBowtie not built (needs /root/reference in the build container)")
    n, L = 3000, 100
    if paired:
        rng = np.random.default_rng(12)
        prefix, text = _write_reference(tmp_path, rng, 200_000, [("chrA", 120_000), ("chrB", 80_000)])
        frag = rng.integers(200, 400, n); pos = rng.integers(120_000, 200_000 - 420, n)          # on the SECOND sequence: next_refID must name it
        files = []
        for k in (0, 1):
            path = str(tmp_path / ("m%d.fastq" % k))
            with open(path, "w") as f:
                for i in range(n):
                    r = text[pos[i]:pos[i] + L] if k == 0 else (3 - text[pos[i] + frag[i] - L:pos[i] + frag[i]])[::-1]
                    r = r.copy(); mut = rng.random(L) < 0.02; r[mut] = (r[mut] + 1) & 3
                    if i % 50 == 7 and k == 1:
                        r = rng.integers(0, 4, L).astype(np.uint8)                                 # a mate that aligns nowhere
                    f.write("@pair%d\n%s\n+\n%s\n" % (i, "".join("ACGT"[c] for c in r), "".join(chr(33 + int(q)) for q in rng.integers(2, 41, L))))
            files.append(path)
        inputs = ["-1", files[0], "-2", files[1]]
    else:
        prefix, fastq, _ = _simulated_run(tmp_path, 9, n, L, indel_rate=0.3)
        inputs = ["-U", fastq]
    outs = {}
    for ext in ("sam", "bam"):
        out = str(tmp_path / ("out." + ext))
        r = subprocess.run([exe, "--file-ref", "-x", prefix] + inputs + ["-S", out], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
        outs[ext] = out
    sam = [ln.rstrip("\n").split("\t") for ln in open(outs["sam"]) if not ln.startswith("@")]
    raw = open(outs["bam"], "rb").read()
    assert raw[:4] == b"\x1f\x8b\x08\x04" and raw[-28:] == bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0])
    text_h, refs, recs = nio.read_bam(outs["bam"])
    assert refs == [("chrA", 120_000), ("chrB", 80_000)] and text_h.startswith("@HD\tVN:1.3\n") and "@PG\tID:" in text_h
    assert len(recs) == len(sam) == (2 * n if paired else n)
    names = [r_[0] for r_ in refs]
    n_mapped = n_gapped = 0
    for s_, b in zip(sam, recs):
        flag = int(s_[1])
        assert b["name"] == s_[0] and b["seq"] == s_[9] and b["qual"] == s_[10]
        if flag & 4:
            # SamOutput prints an unmapped read with the flag 4 alone; BamOutput likewise
            assert b["flag"] == 4 and b["ref"] == -1 and b["pos"] == 0 and b["cigar"] == "*" and b["next_ref"] == -1 and not b["tags"]
            continue
        n_mapped += 1
        assert b["flag"] == flag and names[b["ref"]] == s_[2] and b["pos"] == int(s_[3]) and b["mapq"] == int(s_[4]) and b["cigar"] == s_[5]
        n_gapped += ("I" in s_[5]) or ("D" in s_[5])
        if paired:
            assert (names[b["next_ref"]] == s_[2] if s_[6] == "=" else names[b["next_ref"]] == s_[6]) and b["pnext"] == int(s_[7]) and b["tlen"] == int(s_[8])
        else:
            assert b["next_ref"] == -1 and b["pnext"] == 0 and b["tlen"] == 0          # SamOutput prints '*' 0 0
        tags = dict((t.split(":")[0], t.split(":", 2)[2]) for t in s_[11:])
        assert {k: str(v) for k, v in b["tags"].items()} == tags
    assert n_mapped > 0.95 * len(sam) * (0.98 if paired else 1.0)
    if not paired:
        assert n_gapped > 100


@pytest.mark.parametrize("case", [dict(mode="se", quals="random", seed=41), dict(mode="local", quals="random", len=250, seed=42),
                                  dict(mode="se", ns=0.01, len=150, seed=43), dict(mode="se", len=50, seed=45),
                                  dict(mode="all", quals="random", ns=0.005, seed=46), dict(mode="paired", quals="random", seed=51),
                                  dict(mode="se", repeats=0.6, quals="random", seed=109, reads=10000), dict(mode="paired", repeats=0.6, seed=110, reads=5000),
                                  dict(mode="all", repeats=0.6, seed=107), dict(mode="all", repeats=0.6, seed=108, extra="-N 1", own="allow_sub=1"),
                                  dict(mode="local", repeats=0.6, seed=111, extra="--no-rand", own="randomized=False", reads=10000)],
                         ids=["se-random-quals", "local-250bp-random-quals", "se-150bp-with-N", "se-50bp", "all-random-quals-with-N", "paired-random-quals",
                              "se-repeats", "paired-repeats", "all-repeats", "all-repeats-N1", "local-repeats-no-rand"])
def test_own_drivers_equal_reference_nvbowtie_on_varied_reads(case, cuda):
    """Per-base qualities drawn from phred 2 .. 40 (nvBowtie's mismatch penalty depends on them, scoring.h:206-356), reads with N, read
    lengths 50 / 150 / 250, and genomes 60 % covered by 1 %-diverged copies of six repeat families (several placements per read: second-best
    scores, low MAPQ, randomized choice among equals, 5 placements per read on average in --all): still every SAM record of the reference's
    application equals the from-scratch driver's."""
    import argparse
    import sys
    if not os.path.exists(os.path.join(REF, "ref_nvBowtie")):
        pytest.skip("oracle/_ref/ref_nvBowtie not built (needs /root/reference in the build container)")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import nvbowtie_compare
    args = dict(mode="se", reads=3000, seed=5, indels=0.2, show=3, len=100, ns=0.0, quals="I", repeats=0.0, extra="", own=""); args.update(case)
    same, n_ref, n_own = nvbowtie_compare.compare(argparse.Namespace(**args))
    assert n_ref == n_own and n_ref >= args["reads"]
    assert same == n_ref, (case, same, n_ref)


@pytest.mark.parametrize("reads", [4000, 150_000])
def test_own_all_mapping_equals_reference_nvbowtie_with_repeats_across_sequence_boundaries(reads, cuda, monkeypatch):
    """--all on a genome whose repeat copies may lie ACROSS sequence boundaries: the reference drops hits whose seed straddles two sequences through an
    index it reads from a stale pointer into its sort's ping-pong buffer (aligner_all.h:520) -- the final index for small batches, the last pass
    but one for a full batch of 2^20 hits (the 150 000-read case).  The drivers replay the two sorts on one pair of halves
    (nvbio_hip_sort_hits_pingpong): every record the same."""
    import argparse
    import sys
    if not os.path.exists(os.path.join(REF, "ref_nvBowtie")):
        pytest.skip("oracle/_ref/ref_nvBowtie not built (needs /root/reference in the build container)")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import nvbowtie_compare
    monkeypatch.setenv("NVBOWTIE_COMPARE_STRADDLE", "1")
    args = dict(mode="all", reads=reads, seed=107, indels=0.2, show=3, len=100, ns=0.0, quals="I", repeats=0.6, extra="", own="")
    same, n_ref, n_own = nvbowtie_compare.compare(argparse.Namespace(**args))
    assert n_ref == n_own and n_ref >= reads
    assert same == n_ref, (reads, same, n_ref)


@pytest.mark.parametrize("case", [dict(mode="se", extra="-N 1", own="allow_sub=1", seed=61),
                                  dict(mode="se", extra="-L 18 -D 20 -R 3", own="seed_len=18,max_effort=20,max_reseed=3", seed=62),
                                  dict(mode="local", extra="-N 1 -L 16", own="allow_sub=1,seed_len=16", seed=63),
                                  dict(mode="paired", extra="-I 250 -X 380", own="min_frag_len=250,max_frag_len=380", seed=64, reads=10000),
                                  dict(mode="paired", extra="--no-mixed", own="pe_unpaired=False", seed=65),
                                  dict(mode="paired", extra="--no-discordant", own="pe_discordant=False", seed=66),
                                  dict(mode="se", extra="--no-rand", own="randomized=False", seed=67),
                                  dict(mode="se", extra="--no-rand --no-multi-hits 1", own="randomized=False,no_multi_hits=True", seed=67),
                                  dict(mode="paired", extra="--no-rand", own="randomized=False", seed=70, reads=5000),
                                  dict(mode="se", extra="--nofw", own="fw=False", seed=68),
                                  dict(mode="all", extra="-N 1", own="allow_sub=1", seed=69),
                                  dict(mode="se", extra="--top 1", own="top_seed=1", seed=81),
                                  dict(mode="se", extra="--max-dist 7", own="max_dist=7", seed=82),
                                  dict(mode="se", extra="--max-hits 20 --min-ext 10 --max-ext 40", own="max_hits=20,min_ext=10,max_ext=40", seed=83),
                                  dict(mode="se", extra="-N 1 --subseed-len 10", own="allow_sub=1,subseed_len=10", seed=84),
                                  dict(mode="paired", extra="--no-overlap", own="pe_overlap=False", seed=85),
                                  dict(mode="paired", extra="--ff", own="pe_policy=0", seed=86),
                                  dict(mode="paired", extra="--top 1 -N 1", own="top_seed=1,allow_sub=1", seed=88),
                                  dict(mode="se", extra="--rep-seeds 5 -R 4", own="rep_seeds=5,max_reseed=4", seed=89),
                                  dict(mode="se", extra="--scoring ed", own="scoring_mode=ed", seed=91),
                                  dict(mode="se", extra="--scoring ed --max-dist 7", own="scoring_mode=ed,max_dist=7", seed=92, indels=0.5),
                                  dict(mode="paired", extra="--scoring ed", own="scoring_mode=ed", seed=93, reads=6000),
                                  dict(mode="all", extra="--scoring ed", own="scoring_mode=ed", seed=94),
                                  dict(mode="paired", mixed=True, seed=95, reads=8000),
                                  dict(mode="paired", mixed=True, extra="--local", own="local=True", seed=96, reads=5000, quals="random")],
                         ids=["se-N1", "se-L18-D20-R3", "local-N1-L16", "paired-I250-X380", "paired-no-mixed", "paired-no-discordant", "se-no-rand",
                              "se-no-rand-single-hit", "paired-no-rand", "se-nofw", "all-N1", "se-top", "se-max-dist-7", "se-max-hits-ext", "se-N1-subseed",
                              "paired-no-overlap", "paired-ff", "paired-top-N1", "se-rep-seeds", "se-edit-distance", "se-edit-distance-max-dist-7", "paired-edit-distance", "all-edit-distance", "paired-mixed-lengths", "paired-mixed-lengths-local"])
def test_own_drivers_equal_reference_nvbowtie_under_its_options(case, cuda):
    """nvBowtie's command-line options against the same settings of this repository's drivers: one mismatch in the seed (-N 1: the
    case-pruning mapper in the best modes, the approximate mapper in --all, aligner_all.h:177-212), seed length / effort / re-seeding,
    insert-size limits (-I / -X: here a mate starts exactly at the edge of the opposite-mate window for 1 fragment length in 200, which is
    where a declined anchor job must still be OUTPUT -- batched_banded_inl.h:53-75 -- or the skipped hit keeps a stale score),
    --no-mixed, --no-discordant, the non-randomized selection (--no-rand: nvBowtie pops SA rows off `&deque.top()` in place, so a const
    vector_view must hand out references, vector_view.h:96), --nofw.  Every SAM record identical."""
    import argparse
    import sys
    if not os.path.exists(os.path.join(REF, "ref_nvBowtie")):
        pytest.skip("oracle/_ref/ref_nvBowtie not built (needs /root/reference in the build container)")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import nvbowtie_compare
    args = dict(mode="se", reads=3000, seed=5, indels=0.2, show=3, len=100, ns=0.0, quals="I", repeats=0.0, extra="", own=""); args.update(case)
    same, n_ref, n_own = nvbowtie_compare.compare(argparse.Namespace(**args))
    assert n_ref == n_own and n_ref >= args["reads"]
    assert same == n_ref, (case, same, n_ref)


def test_own_driver_equals_reference_nvbowtie_above_half_a_batch(cuda):
    """More than BATCH_SIZE / 2 reads in flight (and, with this seed, reads with an insertion over the first bases of the genome: their
    seed hits locate below zero, `SA position - offset in the read` wraps, and nvBowtie loads a scoring window ~1 GiB past the reference
    stream -- io::SequenceDataDevice covers every 32-bit coordinate for that, compat/nvbio/io/sequence/sequence.h: coordinate_cover):
    nvBowtie then selects ONE hit per read and round through its warp-aggregated queue allocation
    (`alloc()`, utils.h:58-71: a ballot, a leader elected by `mask << (32 - warp_tid())` -- a shift by 32 for lane 0, which CUDA defines as 0
    and the drop-in layer's warp_tid() type reproduces --, a broadcast through a per-warp shared slot), on 32-lane virtual warps.  600 000
    reads, end to end; every SAM record equal to the from-scratch driver's."""
    import argparse
    import sys
    if not os.path.exists(os.path.join(REF, "ref_nvBowtie")):
        pytest.skip("oracle/_ref/ref_nvBowtie not built (needs /root/reference in the build container)")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import nvbowtie_compare
    same, n_ref, n_own = nvbowtie_compare.compare(argparse.Namespace(mode="se", reads=600_000, seed=31, indels=0.1, show=3))
    assert n_ref == n_own == 600_000
    assert same == n_ref, (same, n_ref)


@pytest.mark.parametrize("mode,reads", [("se", 300_000), ("paired", 120_000)], ids=["single-end", "paired-end"])
def test_reference_nvbowtie_with_two_compute_threads(mode, reads, cuda):
    """nvBowtie's multi-device mode pointed twice at the one GPU (`--device 0 --device 0`: two compute threads, each with its own Aligner, one
    input thread, one mutexed output file; nvBowtie.cpp:809-864, paired-end :638-703), batches of 16 K reads so that both threads get many:
    every read's record must be the one the single-thread run prints (itself held to the from-scratch drivers by the tests above), on each of
    three runs.  (Rounds 1-4 served the drop-in layer's vectors from a hipMemPool; under two threads live blocks lost their contents and two
    runs in three printed a few hundred different records -- profiles/r05/two_threads_pool.txt.)"""
    import argparse
    import sys
    if not os.path.exists(os.path.join(REF, "ref_nvBowtie")):
        pytest.skip("oracle/_ref/ref_nvBowtie not built (needs /root/reference in the build container)")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import nvbowtie_compare
    args = dict(mode=mode, reads=reads, seed=71, indels=0.3, show=3, len=100, ns=0.0, quals="random", repeats=0.0, extra="--batch-size 16", own="")
    n, same = nvbowtie_compare.two_threads(argparse.Namespace(**args), repeats=3)
    assert n >= reads
    assert same == [n, n, n], (mode, n, same)


def test_reference_nvbowtie_equals_own_driver_at_3gbp():
    """BASELINE config 4 as written -- a 3 Gbp index -- through the reference's own application: a repeat-rich synthetic genome (60 % of it diverged
    copies of three repeat families, the largest with three million copies: SA ranges beyond 2^20 rows, rows above 2^31, MAPQ across its whole
    range), its forward and reverse FM-indices written as the files nvBWT leaves, 5 M reads as FASTQ; the unchanged nvBowtie binary and this
    repository's driver (same batches of 1024 K reads) must print the same SAM record for every read (tools/nvbowtie_3gbp.py)."""
    import sys
    exe = os.path.join(REF, "ref_nvBowtie")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/ref_nvBowtie not built (needs /root/reference in the build container)")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import json
    import nvbowtie_3gbp as T
    out, log = T.run(3_000_000_000, 5_000_000, 0.6)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "nvbowtie_3gbp.json"), "w"), indent=1, default=str)
    open(os.path.join(ROOT, "gpurun_out", "nvbowtie_3gbp.log"), "w").write(log)
    assert out["nvbowtie_exit"] == 0, log[-2000:]
    assert out["records_ref"] == out["records_own"] == 5_000_000
    assert out["identical"] == 5_000_000, (out["difference_categories"], out["first_differences"][:3])
    # the C++ host driver on the HBM-rich index the device gets by default: the same 5 M records
    assert out["records_cxx"] == 5_000_000 and out["cxx_identical"] == 5_000_000, (out["cxx_index"], out["cxx_difference_categories"], out["cxx_first_differences"][:3])
    assert out["aligned_share_first_200k"] > 0.9


def test_reference_nvbowtie_equals_own_drivers_at_3gbp_in_edit_distance_mode():
    """--scoring ed at BASELINE config 4's index size: the unchanged nvBowtie and both from-scratch drivers (Params.scoring_mode = "ed" /
    Params::scoring_mode = EditDistanceMode) on the same 3 Gbp repeat-rich files, 2 M reads: every SAM record identical"""
    import sys
    exe = os.path.join(REF, "ref_nvBowtie")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/ref_nvBowtie not built (needs /root/reference in the build container)")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import nvbowtie_3gbp as T
    out, log = T.run(3_000_000_000, 2_000_000, 0.6, extra=("--scoring", "ed"), own_overrides=dict(scoring_mode="ed"))
    assert out["nvbowtie_exit"] == 0, log[-2000:]
    assert out["records_ref"] == out["records_own"] == out["records_cxx"] == 2_000_000
    assert out["identical"] == 2_000_000, (out["difference_categories"], out["first_differences"][:3])
    assert out["cxx_identical"] == 2_000_000, (out["cxx_difference_categories"], out["cxx_first_differences"][:3])


def test_reference_nvbowtie_equals_own_paired_driver_at_3gbp():
    """BASELINE config 5 as written, at one device and one batch: paired-end 2 x 150 bp, --local (LOCAL Gotoh, band 31), on the 3 Gbp repeat-rich index
    files, 1024 K pairs (nvBowtie's batch) with per-base qualities, indels and Ns: the unchanged nvBowtie (-1 / -2) and the C++ paired-end driver
    (Aligner::best_approx over a PairedReadBatch, on the HBM-rich index the loaders build by default) must print the same two SAM records for every
    pair -- flags, positions, MAPQ, CIGAR, RNEXT / PNEXT / TLEN, NM / AS / XM / XO / XG / MD (tools/nvbowtie_3gbp.py run_paired; the records come out of
    the host layer's paired SAM writer, include/nvbio_hip/sam.h)."""
    import sys
    exe = os.path.join(REF, "ref_nvBowtie")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/ref_nvBowtie not built (needs /root/reference in the build container)")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import json
    import nvbowtie_3gbp as T
    pairs = 1 << 20
    out, log = T.run_paired(3_000_000_000, pairs, 0.6)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "nvbowtie_3gbp_paired.json"), "w"), indent=1, default=str)
    open(os.path.join(ROOT, "gpurun_out", "nvbowtie_3gbp_paired.log"), "w").write(log)
    assert out["nvbowtie_exit"] == 0, log[-2000:]
    assert out["records_ref"] == out["records_own"] == 2 * pairs
    assert out["identical"] == 2 * pairs, (out["difference_categories"], out["first_differences"][:3])
