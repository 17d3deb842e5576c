"""The reference's OWN test programs, compiled as they lie against the drop-in template layer (`-I include/nvbio_hip/compat`,
tools/ref_bind_check.py --install-ref-tests -> oracle/_ref/ref_*_test; built in the development container, where
/root/reference exists, and carried to the GPU box as binaries), run on the MI355X:

  ref_alignment_test   nvbio-test/alignment_test.cu, whole TU + nvbio-test/alignment_test_utils.h: the banded edit-distance
                       literals, Gotoh / SW / ED score + traceback checked against the test's own reference DP (ref_sw,
                       ref_banded_sw) and CIGAR literals, then every Batched*AlignmentScore scheduler and the per-thread kernel
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
