"""The policy stages pinned to REFERENCE code.  oracle/build_ref_policy.py compiles nvBowtie's own mapq.h (BowtieMapq2 / 3 behind
MapqFunctorSE / PE), reduce_inl.h (score_reduce_kernel, score_reduce_paired_kernel, try_update ..., both reduce contexts),
aligner.h's init_alignments_kernel and nvbio/io/alignments.h -- verbatim line ranges, read in place -- into
oracle/_ref/libref_policy.so; tests/golden/make_ref_policy_vectors.py recorded its outputs in tests/golden/ref_policy_vectors.npz.
Here: the oracle's restatements against the recording (always), against the library itself on fresh seeds (where it is built), and
the HIP kernels against the recording through the product's Python layer (GPU)."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import pyoracle as O

import ref_policy_cases as K

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_policy_vectors.npz"))
SMIN = (0, -0.6, -0.6)
needs_lib = pytest.mark.skipif(not os.path.exists(K.LIB), reason="oracle/_ref/libref_policy.so not built (needs /root/reference at build time)")


def functor_lengths(first, case):
    """the read lengths MapqFunctorPE hands the calculator: the first slot's mate bit picks read_len2 or read_len1
    (aligner_best_approx_paired.h:83-86: best_paired.anchor_mate<0>() ? read_len2 : read_len1, and the other way round)"""
    m = ((first[0] >> np.uint64(29)) & np.uint64(1)).astype(bool)
    return np.where(m, case["L2"], case["L1"]).astype(np.uint32), np.where(m, case["L1"], case["L2"]).astype(np.uint32)


def oracle_mapq_pe(version, mate, match, smin, case):
    """what MapqFunctorPE(mate) evaluates, through the oracle's entry: (anchor, opposite) slots for mate 0, swapped for mate 1"""
    first, second = (case["anchor"], case["opposite"]) if mate == 0 else (case["opposite"], case["anchor"])
    a_len, o_len = functor_lengths(first, case)
    return O.mapq_paired(version, match, smin, match == 0, first, second, a_len, o_len)


def golden_case(prefix, keys):
    return {k: G["%s_%s" % (prefix, k)] for k in keys}


def test_oracle_mapq_equals_the_recorded_reference():
    seen = set()
    for si, (match, smin) in enumerate(K.SCHEMES):
        se = golden_case("mapq_se%d" % si, ("read_len", "best"))
        pe = golden_case("mapq_pe%d" % si, ("L1", "L2", "a_len", "o_len", "anchor", "opposite"))
        for version in (2, 3):
            exp = G["mapq_se%d_v%d" % (si, version)]
            got = O.mapq(version, match, smin, match == 0, se["best"], se["read_len"])
            assert (got == exp).all(), (si, version, np.nonzero(got != exp)[0][:5])
            seen |= set(exp.tolist())
            for mate in (0, 1):
                exp = G["mapq_pe%d_v%d_mate%d" % (si, version, mate)]
                got = oracle_mapq_pe(version, mate, match, smin, pe)
                assert (got == exp).all(), (si, version, mate, np.nonzero(got != exp)[0][:5])
    assert len(seen) > 40                                   # nearly every value either calculator can return


def replay_reduce(tag, paired, context, step):
    """drive `step(round dict, best, best_o, trys) -> erased flags` over the recorded rounds and compare after each"""
    L = G[tag + "_read_len"]; n = L.size
    best = O.init_alignments(L, SMIN, 0)
    assert (best == G[tag + "_init"]).all()                 # init_alignments_kernel (aligner.h:323-346)
    best_o = None
    if paired:
        best_o = O.init_alignments(L, SMIN, 1)
        assert (best_o == G[tag + "_init_o"]).all()
    trys = G[tag + "_trys0"].copy()
    keys = ["active", "hit_begin", "loc", "rc", "top_flag", "score", "n_ext", "min_ext", "max_ext", "max_effort"] + \
           (["sink", "o_loc", "o_sink", "o_sink2", "o_score", "o_score2", "anchor", "pe_policy", "pe_unpaired", "score_limit"] if paired else [])
    for i in range(5):
        r = {k: G["%s_r%d_%s" % (tag, i, k)] for k in keys}
        erased = step(r, L, best, best_o, trys)
        assert (best == G["%s_r%d_best" % (tag, i)]).all(), (tag, i)
        if paired:
            assert (best_o == G["%s_r%d_best_o" % (tag, i)]).all(), (tag, i)
        if context:
            assert (trys == G["%s_r%d_trys" % (tag, i)]).all(), (tag, i)
            assert (erased == G["%s_r%d_erased" % (tag, i)]).all(), (tag, i)
        else:
            assert not G["%s_r%d_erased" % (tag, i)].any()  # ReduceBestExactContext never stops a traversal
    return best, best_o


def oracle_step(context, paired):
    def step(r, L, best, best_o, trys):
        counts = np.ones(L.size, np.uint32)
        if not paired and context == 0:
            O.score_reduce(best, r["hit_begin"], r["score"], r["loc"], r["rc"], L, r["active"])
        elif not paired:
            O.score_reduce_best_approx(best, r["active"], r["hit_begin"], r["score"], r["loc"], K.seed_words(r), L, -(1 << 16), trys, counts,
                                       int(r["n_ext"]), int(r["min_ext"]), int(r["max_ext"]), int(r["max_effort"]))
        elif context == 0:
            O.score_reduce_paired(best, best_o, r["hit_begin"], r["loc"], r["sink"], r["score"], r["rc"], r["o_loc"], r["o_sink"], r["o_sink2"], r["o_score"], r["o_score2"],
                                  L, int(r["anchor"]), int(r["pe_policy"]), bool(r["pe_unpaired"]), int(r["score_limit"]), r["active"])
        else:
            O.score_reduce_paired_best_approx(best, best_o, r["active"], r["hit_begin"], r["loc"], r["sink"], r["score"], K.seed_words(r), r["o_loc"], r["o_sink"], r["o_sink2"],
                                              r["o_score"], r["o_score2"], L, int(r["anchor"]), int(r["pe_policy"]), bool(r["pe_unpaired"]), int(r["score_limit"]),
                                              trys, counts, int(r["n_ext"]), int(r["min_ext"]), int(r["max_ext"]), int(r["max_effort"]))
        return (counts == 0).astype(np.uint8)
    return step


@pytest.mark.parametrize("context", [0, 1])
@pytest.mark.parametrize("paired", [0, 1])
def test_oracle_reduce_equals_the_recorded_reference(paired, context):
    tag = "red%s_c%d" % ("pe" if paired else "se", context)
    best, best_o = replay_reduce(tag, paired, context, oracle_step(context, paired))
    second = (best[1] >> np.uint64(32)) != np.uint64(K.INV)
    assert 0.2 < second.mean() < 1.0                        # second-best slots filled for many reads, not all
    if paired:
        is_paired = ((best[0] >> np.uint64(30)) & np.uint64(1)).astype(bool)
        assert 0.2 < is_paired.mean() < 0.98                # both paired and unpaired outcomes


def test_alignment_words_equal_the_reference_constructor():
    """io::Alignment(pos, ed, score, rc, mate, paired, discordant) and its accessors (alignments.h:82-129), as compiled"""
    f = {k: G["aln_" + k] for k in ("pos", "ed", "score", "rc", "mate", "paired", "disc")}
    assert (K.pack_words(f["pos"], f["ed"], f["score"], f["rc"], f["mate"], f["paired"], f["disc"]) == G["aln_words"]).all()
    assert O.alignment_invalid() == int(G["aln_invalid"][0])
    acc = G["aln_accessors"]; w = G["aln_words"]
    aligned = f["pos"] != K.INV
    assert (acc[:, 0] == f["score"]).all() and (acc[:, 1] == aligned).all() and (acc[:, 3] == f["rc"]).all() and (acc[:, 5] == f["mate"]).all()
    assert (acc[:, 4] == (f["ed"] & 0x3FF)).all()
    assert (acc[:, 6] == ((f["paired"] == 1) & aligned)).all() and (acc[:, 7] == ((f["paired"] == 0) & aligned)).all()
    assert (acc[:, 8] == ((f["paired"] == 1) & (f["disc"] == 0))).all() and (acc[:, 9] == ((f["paired"] == 1) & (f["disc"] == 1))).all()


@needs_lib
@pytest.mark.parametrize("seed", [11, 12, 13])
def test_oracle_equals_the_compiled_reference_on_fresh_seeds(seed):
    lib = K.ref_lib()
    for si, (match, smin) in enumerate(K.SCHEMES):
        se = K.mapq_se_case(seed * 100 + si, 3000, match, smin)
        pe = K.mapq_pe_case(seed * 100 + 50 + si, 3000, match, smin)
        for version in (2, 3):
            assert (O.mapq(version, match, smin, match == 0, se["best"], se["read_len"]) == K.ref_mapq_se(lib, version, match, smin, se)).all()
            for mate in (0, 1):
                assert (oracle_mapq_pe(version, mate, match, smin, pe) == K.ref_mapq_pe(lib, version, mate, match, smin, pe)).all()
    for paired in (0, 1):
        for context in (0, 1):
            rr = K.reduce_rounds(seed * 1000 + 10 * paired + context, 1500, bool(paired))
            L = rr["read_len"]; n = L.size
            best = O.init_alignments(L, SMIN, 0); best_o = O.init_alignments(L, SMIN, 1) if paired else None
            rbest = np.zeros((2, n), np.uint64); rbest_o = np.zeros((2, n), np.uint64) if paired else None
            lib.ref_init_alignments(SMIN[0], C.c_float(SMIN[1]), C.c_float(SMIN[2]), C.c_uint32(n), K.P(K.seq_index(L)), C.c_uint32(0), K.P(rbest), C.c_uint32(n))
            if paired:
                lib.ref_init_alignments(SMIN[0], C.c_float(SMIN[1]), C.c_float(SMIN[2]), C.c_uint32(n), K.P(K.seq_index(L)), C.c_uint32(1), K.P(rbest_o), C.c_uint32(n))
            trys, rtrys = rr["trys0"].copy(), rr["trys0"].copy()
            step = oracle_step(context, paired)
            for r in rr["rounds"]:
                erased = step(r, L, best, best_o, trys)
                rerased = K.ref_reduce_round(lib, context, r, L, rtrys, rbest, rbest_o)
                assert (best == rbest).all() and (not paired or (best_o == rbest_o).all())
                if context:
                    assert (trys == rtrys).all() and (erased == rerased).all()


# ------------------------------------------------------------------------------------------------------------------------------
# the HIP kernels against the recording
# ------------------------------------------------------------------------------------------------------------------------------
def _scheme(nvb, match, smin):
    s = nvb.SmithWatermanScoringScheme.local() if match else nvb.SmithWatermanScoringScheme()
    s.m_match = match; s.m_monotone = (match == 0); s.m_score_min = smin
    return s


@pytest.mark.gpu
def test_hip_mapq_equals_the_recorded_reference(cuda):
    import torch
    import nvbio_amd as nvb
    d = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).view(dt)).to(cuda)

    def slots(words, sch, L):
        b = nvb.BestAlignments(words.shape[1], sch, read_len=d(L, np.int32), max_read_len=250, device=cuda)
        b.data.copy_(d(words, np.int64))
        return b
    for si, (match, smin) in enumerate(K.SCHEMES):
        sch = _scheme(nvb, match, smin)
        se = golden_case("mapq_se%d" % si, ("read_len", "best"))
        pe = golden_case("mapq_pe%d" % si, ("L1", "L2", "a_len", "o_len", "anchor", "opposite"))
        b = slots(se["best"], sch, se["read_len"]); a = slots(pe["anchor"], sch, pe["a_len"]); o = slots(pe["opposite"], sch, pe["o_len"])
        for version in (2, 3):
            got = nvb.mapq(b, sch, read_len=d(se["read_len"], np.int32), version=version, max_read_len=250).cpu().numpy()
            assert (got == G["mapq_se%d_v%d" % (si, version)]).all(), (si, version)
            for mate, (first, second, words) in enumerate(((a, o, pe["anchor"]), (o, a, pe["opposite"]))):
                a_len, o_len = functor_lengths(words, pe)
                got = nvb.mapq_paired(first, second, sch, read_len=d(a_len, np.int32), o_read_len=d(o_len, np.int32), version=version, max_read_len=250).cpu().numpy()
                assert (got == G["mapq_pe%d_v%d_mate%d" % (si, version, mate)]).all(), (si, version, mate)


@pytest.mark.gpu
@pytest.mark.parametrize("paired", [0, 1])
def test_hip_reduce_equals_the_recorded_reference(cuda, paired):
    """score_reduce / score_reduce_paired kernels (exact context: the update rules; the best-approx counters run inside the C++
    driver and are compared with the oracle's driver in tests/test_select_gpu.py) replayed over the recorded rounds"""
    import torch
    import nvbio_amd as nvb
    d = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).view(dt)).to(cuda)
    tag = "red%s_c0" % ("pe" if paired else "se")
    L = G[tag + "_read_len"]; n = L.size
    sch = nvb.SmithWatermanScoringScheme()
    rl = d(L, np.int32)
    best = nvb.BestAlignments(n, sch, read_len=rl, max_read_len=250, device=cuda, mate=0)
    assert (best.data.cpu().numpy().view(np.uint64) == G[tag + "_init"]).all()
    best_o = nvb.BestAlignments(n, sch, read_len=rl, max_read_len=250, device=cuda, mate=1) if paired else None
    for i in range(5):
        r = lambda k: G["%s_r%d_%s" % (tag, i, k)]
        if paired:
            nvb.score_reduce_paired(best, best_o, d(r("hit_begin"), np.int64), d(r("loc"), np.int32), d(r("sink"), np.int32), d(r("score"), np.int32), d(r("rc"), np.uint8),
                                    d(r("o_loc"), np.int32), d(r("o_sink"), np.int32), d(r("o_sink2"), np.int32), d(r("o_score"), np.int32), d(r("o_score2"), np.int32),
                                    anchor=int(r("anchor")), pe_policy=int(r("pe_policy")), pe_unpaired=bool(r("pe_unpaired")), score_limit=int(r("score_limit")),
                                    read_len=rl, read_ids=d(r("active"), np.int32))
        else:
            nvb.score_reduce(best, d(r("hit_begin"), np.int64), d(r("score"), np.int32), d(r("loc"), np.int32), d(r("rc"), np.uint8), read_len=rl, read_ids=d(r("active"), np.int32))
        torch.cuda.synchronize()
        assert (best.data.cpu().numpy().view(np.uint64) == r("best")).all(), (tag, i)
        if paired:
            assert (best_o.data.cpu().numpy().view(np.uint64) == r("best_o")).all(), (tag, i)
