"""Full-matrix Gotoh score (text-blocking form, the aligner sw-benchmark instantiates) through the
C-ABI vs the oracle's restatement of gotoh_inl.h:969-1489: bit-exact score, sink and ok flag,
including the LOCAL tie order, the int16 boundary column and the early exit."""
import os

import numpy as np
import pytest
import torch

import nvbio_amd as nvb
from oracle import pyoracle as O

pytestmark = pytest.mark.gpu


def dna(s):
    return np.array(["ACGT".index(c) for c in s], dtype=np.uint8)


def run(ty, scheme, pats, txts, dev, min_score=None, maxM=None, maxN=None, pbits=4, pbe=True, tbe=False):
    hp, ht = O.StringSet.from_lists(pats, pbits, pbe), O.StringSet.from_lists(txts, 2, tbe)
    es, ek, eo = O.batch_gotoh_score(ty, scheme, hp, ht, min_score=min_score)
    p = nvb.PackedStringSet.from_host(hp.words, pbits, pbe, hp.begin, hp.length, device=dev)
    t = nvb.PackedStringSet.from_host(ht.words, 2, tbe, ht.begin, ht.length, device=dev)
    ms = torch.from_numpy(np.ascontiguousarray(min_score, dtype=np.int32)).to(dev) if min_score is not None else None
    maxM = maxM or max(1, max(len(x) for x in pats))
    maxN = maxN or max(1, max(len(x) for x in txts))
    for generic in ("0", "1"):          # the 16-bit systolic sweep (when eligible) and the generic 32-bit one
        nvb.set_test_switch("NVBIO_HIP_FULL_GENERIC", generic)
        try:
            gs, gk, go = nvb.batch_alignment_score(nvb.make_gotoh_aligner(ty, nvb.SimpleGotohScheme(*scheme)), p, t, maxM, maxN, ms)
            torch.cuda.synchronize()
        finally:
            nvb.set_test_switch("NVBIO_HIP_FULL_GENERIC", "0")
        gs, gk, go = gs.cpu().numpy(), gk.cpu().numpy().view(np.uint32), go.cpu().numpy()
        bad = np.nonzero((es != gs) | (ek != gk).any(1) | (eo != go))[0]
        assert bad.size == 0, "type %d scheme %s generic=%s: %d mismatches; first %d: M=%d N=%d cpu (%d,%s,%d) gpu (%d,%s,%d)" % (
            ty, scheme, generic, bad.size, bad[0], len(pats[bad[0]]), len(txts[bad[0]]), es[bad[0]], ek[bad[0]], eo[bad[0]], gs[bad[0]], gk[bad[0]], go[bad[0]])
    return es, ek, eo


def test_kats(cuda):
    """alignment_test.cu:749-795: ACAACTA vs AAACACCCTAACACACTAAA, Gotoh (2,-1,-1,-1); the scores are what
    the test's CIGARs (1M2D3M1D3M10D, 4M1D3M, 4M1D3M) re-score to."""
    p, t = dna("ACAACTA"), dna("AAACACCCTAACACACTAAA")
    for ty, score in ((nvb.GLOBAL, 1), (nvb.LOCAL, 13), (nvb.SEMI_GLOBAL, 13)):
        es, ek, _ = run(ty, (2, -1, -1, -1), [p], [t], cuda)
        assert es[0] == score and O.ref_sw_gotoh(ty, (2, -1, -1, -1), p, t) == score


def make_pairs(rng, n, max_m, max_n):
    pats, txts = [], []
    for i in range(n):
        M = int(rng.integers(0 if i % 50 == 0 else 1, max_m + 1))
        N = int(rng.integers(0 if i % 40 == 0 else 1, max_n + 1))
        t = rng.integers(0, 4, N, dtype=np.uint8)
        if N > M + 4 and M > 0:
            off = int(rng.integers(0, N - M))
            p = t[off:off + M].copy()
            mut = rng.random(M) < 0.08
            p[mut] = rng.integers(0, 5, int(mut.sum()), dtype=np.uint8)
            if M > 12 and i % 3 == 0:
                cut = int(rng.integers(3, M - 3)); p = np.concatenate([p[:cut], p[cut + 2:]])
        else:
            p = rng.integers(0, 4, M, dtype=np.uint8)
        if i % 7 == 0 and N > 20:        # low-complexity text: many tied maxima
            t[:] = np.tile(np.array([0, 1], dtype=np.uint8), N)[:N]
            p = np.tile(np.array([0, 1], dtype=np.uint8), M + 1)[:len(p)]
        pats.append(p); txts.append(t)
    return pats, txts


@pytest.mark.parametrize("ty", [nvb.GLOBAL, nvb.LOCAL, nvb.SEMI_GLOBAL])
@pytest.mark.parametrize("max_m", [64, 128, 192, 256, 400, 512, 700, 1024])       # one lane holds 1, 2, 3, 4, 8 or 16 pattern rows
def test_random_pairs(cuda, ty, max_m):
    rng = np.random.default_rng(ty * 10 + max_m)
    pats, txts = make_pairs(rng, 1500 if max_m <= 512 else 500, max_m, 400)
    for scheme in ((2, -1, -2, -1), (0, -5, -8, -3), (2, -1, -1, -1)):
        es, ek, eo = run(ty, scheme, pats, txts, cuda, maxM=max_m, maxN=400)
    for i in range(0, len(pats), 97):       # and the oracle agrees with the reference test's own checker
        if len(txts[i]) > 0 and len(pats[i]) > 0:
            s, _, _ = O.batch_gotoh_score(ty, (2, -1, -1, -1), O.StringSet.from_lists([pats[i]], 4, True), O.StringSet.from_lists([txts[i]], 2, False))
            assert s[0] == O.ref_sw_gotoh(ty, (2, -1, -1, -1), np.minimum(pats[i], 9), txts[i])


@pytest.mark.parametrize("ty", [nvb.GLOBAL, nvb.LOCAL, nvb.SEMI_GLOBAL])
def test_early_exit_against_min_score(cuda, ty):
    rng = np.random.default_rng(77 + ty)
    pats, txts = make_pairs(rng, 1200, 100, 300)
    ms = rng.integers(-60, 220, 1200).astype(np.int32)
    for scheme in ((2, -1, -2, -1), (0, -5, -8, -3)):
        es, ek, eo = run(ty, scheme, pats, txts, cuda, min_score=ms, maxM=100, maxN=300)
    assert 0 < eo.sum() < 1200          # both outcomes occur


@pytest.mark.parametrize("ty", [nvb.GLOBAL, nvb.LOCAL, nvb.SEMI_GLOBAL])
def test_int16_boundary_column_truncation(cuda, ty):
    """Costs large enough that H / E crossing a block boundary do not fit the reference's short2
    boundary column (gotoh_inl.h:1065): the wrap-around is part of the reference's results."""
    rng = np.random.default_rng(5 + ty)
    pats, txts = make_pairs(rng, 600, 100, 300)
    run(ty, (700, -900, -1100, -800), pats, txts, cuda, maxM=100, maxN=300)
    run(ty, (2, -1, -2, -1), pats, txts, cuda, maxM=100, maxN=40000)     # bound unknown -> TRUNC variant, same results


def test_global_long_texts_on_the_16bit_sweep(cuda):
    """GLOBAL against texts of thousands of symbols (sw-benchmark's shape: 150 x 16384): (M + N) * max|cost| exceeds int16 there, but
    the values the sweep holds do not when the expensive costs are not the gap extensions -- the host's tighter proof puts these
    batches on the 16-bit sweep (checked through nvbio_hip_last_kernel), up to where the boundary row itself nears -32000."""
    from nvbio_amd._lib import lib
    rng = np.random.default_rng(611)
    for maxN, scheme, sweep16 in ((16384, (2, -1, -2, -1), True), (16384, (1, -3, -5, -1), True), (31000, (2, -1, -2, -1), True),
                                  (16384, (2, -3, -5, -2), False), (33000, (2, -1, -2, -1), False)):
        pats, txts = [], []
        for i in range(48):
            M = int(rng.integers(1, 151)) if i % 3 else 150
            N = maxN if i % 4 == 0 else int(rng.integers(1, maxN + 1))
            t = rng.integers(0, 4, N, dtype=np.uint8)
            o = int(rng.integers(0, max(1, N - M)))
            pp = np.resize(t[o:o + M], M).copy()
            mut = rng.random(M) < 0.08
            pp[mut] = rng.integers(0, 5, int(mut.sum()), dtype=np.uint8)
            pats.append(pp); txts.append(t)
        run(nvb.GLOBAL, scheme, pats, txts, cuda, maxM=150, maxN=maxN)
        hp, ht = O.StringSet.from_lists(pats, 4, True), O.StringSet.from_lists(txts, 2, False)
        p = nvb.PackedStringSet.from_host(hp.words, 4, True, hp.begin, hp.length, device=cuda)
        t = nvb.PackedStringSet.from_host(ht.words, 2, False, ht.begin, ht.length, device=cuda)
        nvb.batch_alignment_score(nvb.make_gotoh_aligner(nvb.GLOBAL, nvb.SimpleGotohScheme(*scheme)), p, t, 150, maxN, None)
        torch.cuda.synchronize()
        assert (b"16-bit" in lib().nvbio_hip_last_kernel()) == sweep16, (maxN, scheme, lib().nvbio_hip_last_kernel())


def test_packings(cuda):
    rng = np.random.default_rng(9)
    pats, txts = make_pairs(rng, 400, 90, 200)
    p2 = [np.minimum(p, 3) for p in pats]
    for (pb, pbe, tbe, pp) in ((4, False, True, pats), (2, True, True, p2), (2, False, False, p2)):
        run(nvb.LOCAL, (2, -1, -2, -1), pp, txts, cuda, pbits=pb, pbe=pbe, tbe=tbe)


# ---------------------------------------------------------------------------- PatternBlockingTag (the default algorithm tag)
@pytest.mark.parametrize("ty", [nvb.GLOBAL, nvb.LOCAL, nvb.SEMI_GLOBAL])
@pytest.mark.parametrize("kind", [0, 1])
def test_pattern_blocking(cuda, ty, kind):
    """Gotoh (8-symbol blocks) and SW / ED (16-symbol blocks) in the pattern-blocking form, vs the oracle's restatement of
    gotoh_inl.h:459-900 / sw_inl.h:417-760: scores, sinks (LOCAL ties follow the pattern-block order) and the per-block
    early exit against min_score."""
    rng = np.random.default_rng(300 + 10 * kind + ty)
    for max_m in (60, 130, 200, 256, 400):
        pats, txts = make_pairs(rng, 800, max_m, 350)
        pats = [p if len(p) else np.zeros(1, np.uint8) for p in pats]            # M >= 1 (M == 0 is undefined in the reference)
        hp, ht = O.StringSet.from_lists(pats, 4, True), O.StringSet.from_lists(txts, 2, False)
        dp = nvb.PackedStringSet.from_host(hp.words, 4, True, hp.begin, hp.length, device=cuda)
        dt = nvb.PackedStringSet.from_host(ht.words, 2, False, ht.begin, ht.length, device=cuda)
        ms = rng.integers(-80, 260, 800).astype(np.int32)
        for scheme in (((2, -1, -2, -1), (0, -5, -8, -3)) if kind == 0 else ((2, -1, -1, -1), (0, -1, -1, -1))):
            mk = (lambda s_: nvb.make_gotoh_aligner(ty, nvb.SimpleGotohScheme(*s_), nvb.PATTERN_BLOCKING)) if kind == 0 else \
                 (lambda s_: nvb.make_smith_waterman_aligner(ty, nvb.SimpleSmithWatermanScheme(*s_), nvb.PATTERN_BLOCKING))
            for min_score in (None, ms):
                es, ek, eo = O.batch_score_pattern_blocking(kind, ty, scheme, hp, ht, min_score=min_score)
                gs, gk, go = nvb.batch_alignment_score(mk(scheme), dp, dt, max_m, 350,
                                                       torch.from_numpy(min_score).to(cuda) if min_score is not None else None)
                torch.cuda.synchronize()
                gs, gk, go = gs.cpu().numpy(), gk.cpu().numpy().view(np.uint32), go.cpu().numpy()
                bad = np.nonzero((es != gs) | (ek != gk).any(1) | (eo != go))[0]
                assert bad.size == 0, (kind, ty, scheme, max_m, min_score is not None, bad[:5], len(pats[bad[0]]), len(txts[bad[0]]),
                                       (es[bad[0]], ek[bad[0]], eo[bad[0]]), (gs[bad[0]], gk[bad[0]], go[bad[0]]))
                if min_score is not None and scheme[0] > 0:
                    assert 0 < eo.sum() < 800          # both outcomes occur


@pytest.mark.parametrize("ty", [nvb.GLOBAL, nvb.LOCAL, nvb.SEMI_GLOBAL])
@pytest.mark.parametrize("algo", [nvb.PATTERN_BLOCKING, nvb.TEXT_BLOCKING])
def test_quality_aware_scheme_full_matrix(cuda, ty, algo):
    """nvBowtie's opposite-mate scoring: GotohAligner<TYPE, SmithWatermanScoringScheme> over the full matrix, both tags, with
    min_score; asymmetric read / reference gap costs exercise the tag-dependent boundary initialisation."""
    rng = np.random.default_rng(600 + 10 * algo + ty)
    pats, txts = make_pairs(rng, 900, 150, 400)
    pats = [p if len(p) else np.zeros(1, np.uint8) for p in pats]
    hp, ht = O.StringSet.from_lists(pats, 4, True), O.StringSet.from_lists(txts, 2, True)
    total = int(hp.begin[-1] + hp.length[-1])
    quals = rng.integers(0, 60, total + 3, dtype=np.uint8); quals[::89] = 255
    dp = nvb.PackedStringSet.from_host(hp.words, 4, True, hp.begin, hp.length, device=cuda)
    dt = nvb.PackedStringSet.from_host(ht.words, 2, True, ht.begin, ht.length, device=cuda)
    dq = torch.from_numpy(quals).to(cuda)
    ms = rng.integers(-60, 200, 900).astype(np.int32)
    for scheme in (nvb.SmithWatermanScoringScheme(), nvb.SmithWatermanScoringScheme.local(),
                   nvb.SmithWatermanScoringScheme(match=1, mmp_min=1, mmp_max=9, read_gap_const=4, read_gap_coeff=2, ref_gap_const=7, ref_gap_coeff=1)):
        st = scheme.struct()
        lut = np.array([st.mismatch[q] for q in range(256)], dtype=np.int32)
        s5 = (st.match, st.pattern_gap_open, st.pattern_gap_ext, st.text_gap_open, st.text_gap_ext)
        for min_score in (None, ms):
            es, ek, eo = O.batch_gotoh_score_qual(algo, ty, s5, lut, quals, hp, ht, min_score=min_score)
            gs, gk, go = nvb.batch_alignment_score(nvb.make_gotoh_aligner(ty, scheme, algo), dp, dt, 150, 400,
                                                   torch.from_numpy(min_score).to(cuda) if min_score is not None else None, quals=dq)
            torch.cuda.synchronize()
            gs, gk, go = gs.cpu().numpy(), gk.cpu().numpy().view(np.uint32), go.cpu().numpy()
            bad = np.nonzero((es != gs) | (ek != gk).any(1) | (eo != go))[0]
            assert bad.size == 0, (ty, algo, st.match, min_score is not None, bad[:5], len(pats[bad[0]]), len(txts[bad[0]]),
                                   (es[bad[0]], ek[bad[0]], eo[bad[0]]), (gs[bad[0]], gk[bad[0]], go[bad[0]]))


@pytest.mark.parametrize("ty", [nvb.LOCAL, nvb.SEMI_GLOBAL, nvb.GLOBAL])
@pytest.mark.parametrize("max_m,max_n", [(150, 520), (100, 300), (75, 200), (170, 400), (40, 90)])
def test_several_jobs_per_wave(cuda, ty, max_m, max_n, monkeypatch):
    """Short patterns run two to four jobs per wave (full_gotoh_score_multi_kernel: segments of 32 / 21 / 16 lanes, a segment's first lane
    taking the row above its own matrix).  Ragged M and N inside one wave, job counts that do not fill the last wave, empty patterns / texts,
    tie-heavy texts, min_score early exits of both blocking orders (the second sweep of an exited job), qualities: the multi-job
    kernel, the single-job kernel (NVBIO_HIP_FULL_SINGLE_JOB=1) and the oracle must agree bit for bit."""
    rng = np.random.default_rng(77 + ty * 1000 + max_m)
    for n in (1, 2, 3, 5, 1501):
        pats, txts = make_pairs(rng, n, max_m, max_n)
        hp, ht = O.StringSet.from_lists(pats, 4, True), O.StringSet.from_lists(txts, 2, False)
        p = nvb.PackedStringSet.from_host(hp.words, 4, True, hp.begin, hp.length, device=cuda)
        t = nvb.PackedStringSet.from_host(ht.words, 2, False, ht.begin, ht.length, device=cuda)
        ms_host = np.where(rng.random(n) < 0.4, -(1 << 30), rng.integers(-30, 120, n)).astype(np.int32)
        for scheme in ((2, -1, -2, -1), (1, -3, -5, -2)):
            for algo, min_score in ((nvb.TEXT_BLOCKING, None), (nvb.TEXT_BLOCKING, ms_host), (nvb.PATTERN_BLOCKING, None), (nvb.PATTERN_BLOCKING, ms_host)):
                if algo == nvb.TEXT_BLOCKING:
                    es, ek, eo = O.batch_gotoh_score(ty, scheme, hp, ht, min_score=min_score)
                else:
                    live = hp.length > 0                                     # (the reference reads uninitialised cells for M == 0 under pattern blocking)
                    es, ek, eo = O.batch_score_pattern_blocking(0, ty, scheme, hp, ht, min_score=min_score)
                ms = torch.from_numpy(min_score).to(cuda) if min_score is not None else None
                kernels = []
                # "0": the depth the host picks; "r8" / "r10": four jobs per wave at 8 / 10 rows per lane where the pattern fits; "1": one job per wave
                for single in ("0", "r8", "r10", "1"):
                    nvb.set_test_switch("NVBIO_HIP_FULL_SINGLE_JOB", "1" if single == "1" else "0")
                    if single.startswith("r"):
                        if max_m > 16 * int(single[1:]):
                            continue
                        nvb.set_test_switch("NVBIO_HIP_FULL_ROWS", single[1:])
                    else:
                        nvb.set_test_switch("NVBIO_HIP_FULL_ROWS", 0)
                    al = nvb.make_gotoh_aligner(ty, nvb.SimpleGotohScheme(*scheme), algo)
                    gs, gk, go = nvb.batch_alignment_score(al, p, t, max_m, max_n, ms)
                    torch.cuda.synchronize()
                    kernels.append(nvb.lib().nvbio_hip_last_kernel().decode())
                    gs, gk, go = gs.cpu().numpy(), gk.cpu().numpy().view(np.uint32), go.cpu().numpy()
                    sel = np.ones(n, bool) if algo == nvb.TEXT_BLOCKING else live
                    bad = np.nonzero(((es != gs) | (ek != gk).any(1) | (eo != go)) & sel)[0]
                    assert bad.size == 0, "type %d scheme %s algo %d single=%s n=%d: %d mismatches; first %d: M=%d N=%d cpu (%d,%s,%d) gpu (%d,%s,%d)" % (
                        ty, scheme, algo, single, n, bad.size, bad[0], len(pats[bad[0]]), len(txts[bad[0]]), es[bad[0]], ek[bad[0]], eo[bad[0]], gs[bad[0]], gk[bad[0]], go[bad[0]])
                assert "multi" in kernels[0] and "multi" not in kernels[-1], kernels
    nvb.set_test_switch("NVBIO_HIP_FULL_SINGLE_JOB", 0)
    nvb.set_test_switch("NVBIO_HIP_FULL_ROWS", 0)
