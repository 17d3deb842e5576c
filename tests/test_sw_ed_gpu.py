"""SmithWatermanAligner / EditDistanceAligner (linear gaps) through the C-ABI vs the oracle's restatement of
sw_banded_inl.h and of the text-blocking sw_inl.h (16-column blocks): bit-exact scores and sinks, banded
and full matrix -- sw-benchmark's edit-distance leg and nvbio-test's ed / sw cases."""
import json
import os

import numpy as np
import pytest
import torch

import nvbio_amd as nvb
from oracle import pyoracle as O
from test_banded_gpu import random_pairs
from test_full_gotoh_gpu import make_pairs, dna

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
KAT = json.load(open(os.path.join(HERE, "golden", "kat.json")))
SCHEMES = [(0, -1, -1, -1), (2, -1, -1, -1), (1, -3, -2, -2)]


def to_dev(hs, dev):
    return nvb.PackedStringSet.from_host(hs.words, hs.bits, hs.big_endian, hs.begin, hs.length, device=dev)


def aligner_for(ty, scheme):
    return nvb.make_edit_distance_aligner(ty) if scheme == (0, -1, -1, -1) else nvb.make_smith_waterman_aligner(ty, nvb.SimpleSmithWatermanScheme(*scheme))


@pytest.mark.parametrize("band", [3, 5, 7, 15, 31])
@pytest.mark.parametrize("ty", [nvb.GLOBAL, nvb.LOCAL, nvb.SEMI_GLOBAL])
def test_banded(cuda, band, ty):
    rng = np.random.default_rng(9000 + band * 3 + ty)
    pats, txts = random_pairs(rng, 2000, band)
    hp, ht = O.StringSet.from_lists(pats, 4, True), O.StringSet.from_lists(txts, 2, True)
    for scheme in SCHEMES:
        es, ek = O.batch_sw_score(band, ty, scheme, hp, ht)
        gs, gk = nvb.batch_banded_alignment_score(band, aligner_for(ty, scheme), to_dev(hp, cuda), to_dev(ht, cuda),
                                                  max_pattern_length=int(hp.length.max()))
        torch.cuda.synchronize()
        gs, gk = gs.cpu().numpy(), gk.cpu().numpy().view(np.uint32)
        bad = np.nonzero((es != gs) | (ek != gk).any(1))[0]
        assert bad.size == 0, (band, ty, scheme, bad[:5], es[bad[:3]], gs[bad[:3]])


def test_banded_edit_distance_kats(cuda):
    """alignment_test.cu:677-747: banded_alignment_score<5>(make_edit_distance_aligner<SEMI_GLOBAL>()) literals;
    the test's strings are ASCII, mapped to DNA codes here (only equality of symbols matters)."""
    grp = KAT["banded_edit_distance_band5_semi_global"]
    code = {c: i for i, c in enumerate("ACGT")}
    pats = [np.array([code[c] for c in k["pattern"]], np.uint8) for k in grp["cases"]]
    txts = [np.array([code[c] for c in k["text"]], np.uint8) for k in grp["cases"]]
    hp, ht = O.StringSet.from_lists(pats, 4, True), O.StringSet.from_lists(txts + [np.zeros(64, np.uint8)], 2, False)
    ht = O.StringSet(ht.words, 2, False, ht.begin[:-1], ht.length[:-1])
    gs, _ = nvb.batch_banded_alignment_score(5, nvb.make_edit_distance_aligner(nvb.SEMI_GLOBAL), to_dev(hp, cuda), to_dev(ht, cuda), max_pattern_length=16)
    assert gs.cpu().tolist() == [k["score"] for k in grp["cases"]]


@pytest.mark.parametrize("ty", [nvb.GLOBAL, nvb.LOCAL, nvb.SEMI_GLOBAL])
def test_full_matrix(cuda, ty):
    rng = np.random.default_rng(9100 + ty)
    pats, txts = make_pairs(rng, 1500, 200, 400)
    pats = [p if len(p) else np.zeros(1, np.uint8) for p in pats]
    txts = [t if len(t) else np.zeros(1, np.uint8) for t in txts]
    hp, ht = O.StringSet.from_lists(pats, 4, True), O.StringSet.from_lists(txts, 2, False)
    for scheme in SCHEMES:
        es, ek = O.batch_sw_score(0, ty, scheme, hp, ht)
        for generic in ("0", "1"):
            nvb.set_test_switch("NVBIO_HIP_FULL_GENERIC", generic)
            try:
                gs, gk, go = nvb.batch_alignment_score(aligner_for(ty, scheme), to_dev(hp, cuda), to_dev(ht, cuda), 200, 400)
                torch.cuda.synchronize()
            finally:
                nvb.set_test_switch("NVBIO_HIP_FULL_GENERIC", "0")
            gs, gk = gs.cpu().numpy(), gk.cpu().numpy().view(np.uint32)
            bad = np.nonzero((es != gs) | (ek != gk).any(1))[0]
            assert bad.size == 0, (ty, scheme, generic, bad[:5], es[bad[:3]], gs[bad[:3]], ek[bad[:3]], gk[bad[:3]])
            assert bool(go.all())
    # (with LOCAL the 16-column visiting order is observable: on most data sets some ties resolve differently from the Gotoh
    # (8-column) form -- tests/test_oracle_kat.py::test_pattern_blocking_and_text_blocking_restatements_agree counts them)


def test_sw_benchmark_edit_distance_leg(cuda):
    """sw-benchmark.cu:641-657: every read against the whole reference, edit distance, SEMI_GLOBAL"""
    rng = np.random.default_rng(5)
    ref = rng.integers(0, 4, 4096, dtype=np.uint8)
    reads = []
    for i in range(512):
        p = int(rng.integers(0, ref.size - 150))
        r = ref[p:p + 150].copy()
        mut = rng.random(150) < 0.04
        r[mut] = rng.integers(0, 4, int(mut.sum()))
        reads.append(r)
    hp = O.StringSet.from_lists(reads, 4, True)
    hr = O.StringSet.from_lists([ref], 2, False)
    ht = O.StringSet(hr.words, 2, False, np.zeros(512, np.uint64), np.full(512, ref.size, np.uint32))      # every job: the same text
    es, ek = O.batch_sw_score(0, nvb.SEMI_GLOBAL, (0, -1, -1, -1), hp, ht)
    gs, gk, _ = nvb.batch_alignment_score(nvb.make_edit_distance_aligner(nvb.SEMI_GLOBAL), to_dev(hp, cuda), to_dev(ht, cuda), 150, ref.size)
    torch.cuda.synchronize()
    assert (gs.cpu().numpy() == es).all() and (gk.cpu().numpy().view(np.uint32) == ek).all()
    assert es.max() <= 0 and es.min() < -2


def test_asymmetric_costs_run_on_the_tuned_kernels(cuda):
    """deletion != insertion was refused until round 6 (tests/test_tuned_edges_gpu.py holds the parity tests)"""
    hp, ht = O.StringSet.from_lists([np.zeros(10, np.uint8)], 4, True), O.StringSet.from_lists([np.zeros(30, np.uint8)], 2, True)
    al = nvb.make_smith_waterman_aligner(nvb.LOCAL, nvb.SimpleSmithWatermanScheme(2, -1, -2, -1))
    gs, _ = nvb.batch_banded_alignment_score(15, al, to_dev(hp, cuda), to_dev(ht, cuda))
    assert gs.cpu().tolist() == [20]
    gs, _, _ = nvb.batch_alignment_score(al, to_dev(hp, cuda), to_dev(ht, cuda), 10, 30)
    assert gs.cpu().tolist() == [20]


@pytest.mark.parametrize("ty", [nvb.GLOBAL, nvb.SEMI_GLOBAL])
@pytest.mark.parametrize("algorithm", ["text_blocking", "pattern_blocking"])
def test_edit_distance_bitvector_kernel(cuda, ty, algorithm):
    """The edit-distance aligner without a min_score runs on the bit-vector kernel (one lane per job, 64 pattern rows per word):
    every word count (patterns to 512), ragged and empty strings, N symbols, texts shorter and much longer than the pattern, both
    algorithm tags -- scores and sinks equal the oracle's and the sweep kernel's (NVBIO_HIP_ED_SWEEP=1)."""
    from nvbio_amd._lib import lib
    rng = np.random.default_rng(9500 + ty)
    tag = nvb.PATTERN_BLOCKING if algorithm == "pattern_blocking" else nvb.TEXT_BLOCKING
    for max_m, max_n, n in ((64, 300, 800), (100, 260, 800), (150, 900, 500), (192, 400, 400), (256, 700, 300), (400, 1000, 200), (512, 1500, 160)):
        pats, txts = [], []
        for i in range(n):
            M = max_m if i % 5 == 0 else int(rng.integers(0 if i % 31 == 0 else 1, max_m + 1))
            N = int(rng.integers(0 if i % 29 == 0 else 1, max_n + 1))
            t = rng.integers(0, 4, N, dtype=np.uint8)
            if N > M + 4 and M > 0 and i % 3:
                o = int(rng.integers(0, N - M)); q = t[o:o + M].copy()
                mut = rng.random(M) < 0.08
                q[mut] = rng.integers(0, 5, int(mut.sum()), dtype=np.uint8)               # 4 = N: matches nothing
                if M > 20 and i % 2:
                    c = int(rng.integers(3, M - 8)); q = np.concatenate([q[:c], q[c + 3:], rng.integers(0, 4, 3, dtype=np.uint8)])
            else:
                q = rng.integers(0, 4, M, dtype=np.uint8)
            pats.append(q.astype(np.uint8)); txts.append(t)
        hp, ht = O.StringSet.from_lists(pats, 4, True), O.StringSet.from_lists(txts + [np.zeros(64, np.uint8)], 2, bool(ty == nvb.GLOBAL))
        ht = O.StringSet(ht.words, 2, ht.big_endian, ht.begin[:-1], ht.length[:-1])
        if algorithm == "pattern_blocking":
            es, ek, _ = O.batch_score_pattern_blocking(1, ty, (0, -1, -1, -1), hp, ht)
        else:
            es, ek = O.batch_sw_score(0, ty, (0, -1, -1, -1), hp, ht)
        al = nvb.make_edit_distance_aligner(ty, tag)
        dp, dt = to_dev(hp, cuda), to_dev(ht, cuda)
        for sweep in ("0", "1"):
            nvb.set_test_switch("NVBIO_HIP_ED_SWEEP", sweep)
            try:
                gs, gk, go = nvb.batch_alignment_score(al, dp, dt, max_m, max_n)
                torch.cuda.synchronize()
                kernel = lib().nvbio_hip_last_kernel()
            finally:
                nvb.set_test_switch("NVBIO_HIP_ED_SWEEP", "0")
            assert (b"edit_distance_bitvector" in kernel) == (sweep == "0"), kernel
            gs, gk = gs.cpu().numpy(), gk.cpu().numpy().view(np.uint32)
            # (an empty pattern makes the reference's pattern-blocking pass read uninitialised cells: there the two kernels are only
            # compared with each other)
            defined = (hp.length > 0) | (algorithm == "text_blocking")
            bad = np.nonzero(((es != gs) | (ek != gk).any(1)) & defined)[0]
            assert bad.size == 0, (ty, algorithm, max_m, sweep, bad[:5], [(len(pats[b]), len(txts[b])) for b in bad[:3]], es[bad[:3]], gs[bad[:3]], ek[bad[:3]], gk[bad[:3]])
            assert bool(go.all())
            if sweep == "0":
                first = (gs, gk)
            else:
                assert (first[0] == gs).all() and (first[1] == gk).all()
