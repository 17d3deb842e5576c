"""HostThreadScheduler through the drop-in template layer, on the CPU suite: the caller TU of tests/compat/aln_callers.hip
(the reference's stream concept) run over host pointers with OpenMP -- the generic banded and full-matrix templates of
include/nvbio_hip/compat/nvbio/alignment/alignment.h against the oracle.  (BASELINE config 1: "sw-benchmark host path".)"""
import os

import numpy as np
import pytest

from oracle import pyoracle as O
from tests.test_compat_alignment_gpu import Batch, make_jobs, run_banded, run_full, callers, GLOBAL, LOCAL, SEMI   # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "tests", "compat", "libaln_callers.so")),
                                reason="tests/compat/libaln_callers.so not built")


@pytest.mark.parametrize("typ", [GLOBAL, LOCAL, SEMI])
def test_host_banded_gotoh_packed(callers, typ):
    reads, quals, wins = make_jobs(41 + typ, 3000, band=15)
    b = Batch(reads, quals, wins, packed=True, on_device=False)
    assert run_banded(callers, b, 0, 1, 0, typ, 15, (2, -1, -2, -1)) == "host"
    es, ek = O.batch_banded_gotoh_score(15, typ, (2, -1, -2, -1), b.hr, b.hw)
    gs, gk = b.results()
    assert (gs == es).all() and (gk == ek).all()
    # band 31: the 2-bit window cache
    reads, quals, wins = make_jobs(61 + typ, 1500, band=31)
    b = Batch(reads, quals, wins, packed=True, on_device=False)
    assert run_banded(callers, b, 0, 1, 0, typ, 31, (0, -5, -8, -3)) == "host"
    es, ek = O.batch_banded_gotoh_score(31, typ, (0, -5, -8, -3), b.hr, b.hw)
    gs, gk = b.results()
    assert (gs == es).all() and (gk == ek).all()


@pytest.mark.parametrize("typ", [GLOBAL, LOCAL, SEMI])
def test_host_banded_sw_bytes_asymmetric_gaps(callers, typ):
    reads, quals, wins = make_jobs(42 + typ, 2000, band=31, max_sym=5)
    b = Batch(reads, quals, wins, packed=False, on_device=False)
    for band in (9, 15, 31):
        assert run_banded(callers, b, 1, 1, 1, typ, band, (1, -1, -2, -3)) == "host"
        es, ek = O.batch_sw_score(band, typ, (1, -1, -2, -3), b.hr, b.hw)
        gs, gk = b.results()
        declined = (np.arange(b.n) % 97) == 96
        assert (gs[declined] == -(1 << 30)).all()                # a declined job is still output, with its fresh sink (batched_banded_inl.h:53-75)
        assert (gs[~declined] == es[~declined]).all() and (gk[~declined] == ek[~declined]).all()


@pytest.mark.parametrize("typ", [GLOBAL, LOCAL, SEMI])
@pytest.mark.parametrize("tag", [0, 1])
def test_host_full_matrix(callers, typ, tag):
    reads, quals, wins = make_jobs(500 + typ + 3 * tag, 500, max_read=100, full=True, short_text_every=10 ** 9)
    reads = [r if len(r) else np.array([1], np.uint8) for r in reads]
    quals = [q if len(q) else np.array([0], np.uint8) for q in quals]
    b = Batch(reads, quals, wins, packed=True, on_device=False)
    rng = np.random.default_rng(5)
    th = np.where(rng.random(b.n) < 0.5, -(1 << 30), rng.integers(-40, 160, b.n)).astype(np.int32)
    assert run_full(callers, b, 0, 1, 0, typ, tag, (2, -1, -2, -1), th) == "host"
    if tag == 0:
        es, ek, _ = O.batch_score_pattern_blocking(0, typ, (2, -1, -2, -1), b.hr, b.hw, min_score=th)
    else:
        es, ek, _ = O.batch_gotoh_score(typ, (2, -1, -2, -1), b.hr, b.hw, min_score=th)
    gs, gk = b.results()
    assert (gs == es).all() and (gk == ek).all()
    # Smith-Waterman with asymmetric gap costs over byte strings
    bb = Batch(reads, quals, wins, packed=False, on_device=False)
    assert run_full(callers, bb, 1, 1, 1, typ, tag, (2, -2, -4, -1), None) == "host"
    if tag == 0:
        es, ek, _ = O.batch_score_pattern_blocking(1, typ, (2, -2, -4, -1), bb.hr, bb.hw)
    else:
        es, ek = O.batch_sw_score(0, typ, (2, -2, -4, -1), bb.hr, bb.hw)
    gs, gk = bb.results()
    declined = (np.arange(bb.n) % 97) == 96
    assert (gs[~declined] == es[~declined]).all() and (gk[~declined] == ek[~declined]).all()


@pytest.mark.parametrize("typ", [GLOBAL, LOCAL, SEMI])
@pytest.mark.parametrize("band", [15, 31, 0])
def test_host_generic_tracebacks(callers, typ, band):
    """BatchedBandedAlignmentTraceback / BatchedAlignmentTraceback<..., HostThreadScheduler> over byte strings with Gotoh, asymmetric
    Smith-Waterman, edit-distance and a user-defined quality scheme: the per-lane templates of compat/nvbio/alignment/traceback.h
    (batched_banded_inl.h:299-326 takes any stream) against the oracle's tracebacks."""
    from tests.test_compat_alignment_gpu import run_byte_tracebacks
    assert run_byte_tracebacks(callers, 1, typ, band, n=250) > 800
