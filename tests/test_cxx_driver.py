"""The C++ host layer (include/nvbio_hip/*.h) and its parity driver tests/cxx/nvbio_hip_test.cpp,
which is written the way the reference's nvbio-test suites are (-aln / -rank / -fm-index)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cxx", "nvbio_hip_test")


def test_cxx_driver_compiles_against_the_host_layer():
    import __graft_entry__ as g
    out = g.build_cxx_tests()
    assert os.path.exists(out)


@pytest.mark.gpu
@pytest.mark.parametrize("suite", ["-aln", "-rank", "-fm-index"])
def test_cxx_driver_suites(cuda, suite):
    assert os.path.exists(BIN), "run __graft_entry__.build() first"
    r = subprocess.run([BIN, suite], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "all passed" in r.stderr
