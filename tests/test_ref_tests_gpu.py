"""The reference's OWN test programs, compiled as they lie against the drop-in template layer (`-I include/nvbio_hip/compat`,
tools/ref_bind_check.py --install-ref-tests -> oracle/_ref/ref_*_test; built in the development container, where
/root/reference exists, and carried to the GPU box as binaries), run on the MI355X:

  ref_alignment_test   nvbio-test/alignment_test.cu, whole TU + nvbio-test/alignment_test_utils.h: the banded edit-distance
                       literals, Gotoh / SW / ED score + traceback checked against the test's own reference DP (ref_sw,
                       ref_banded_sw) and CIGAR literals, then every Batched*AlignmentScore scheduler and the per-thread kernel
  ref_sw_benchmark     sw-benchmark/sw-benchmark.cu, whole TU: the reference's headline benchmark program on synthetic FASTQ / FASTA input
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
