"""Parity of the HIP banded Gotoh kernel (through the C-ABI) with the CPU oracle: bit-exact
scores and sinks.  Mirrors what the reference's alignment tests check (KATs, the differential
ref_banded_sw check) plus the edge cases of the reference semantics."""
import json
import os

import numpy as np
import pytest
import torch

import nvbio_amd as nvb
from nvbio_amd import workloads as W
from oracle import pyoracle as O

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
KAT = json.load(open(os.path.join(HERE, "golden", "kat.json")))
SCHEMES = [(2, -1, -2, -1), (0, -5, -8, -3), (2, -1, -1, -1)]


def dna(s):
    return np.array(["ACGT".index(c) for c in s], dtype=np.uint8)


_hint_cycle = [0]


def run_gpu(band, ty, scheme, hp, ht, dev, force32=False):
    """hp/ht: oracle StringSets (host) -> run the same data through the HIP path.
    force32: disable the 16-bit kernels so the 32-bit ones are exercised on the same data.
    The max_pattern_length hint (which only sizes the LDS staging) cycles through unknown / exact /
    too small, so staged lanes, unstaged launches and per-lane fall-backs are all exercised."""
    p = nvb.PackedStringSet.from_host(hp.words, hp.bits, hp.big_endian, hp.begin, hp.length, device=dev)
    t = nvb.PackedStringSet.from_host(ht.words, ht.bits, ht.big_endian, ht.begin, ht.length, device=dev)
    nvb.set_test_switch("NVBIO_HIP_FORCE_32BIT", "1" if force32 else "0")
    maxlen = int(hp.length.max()) if hp.length.size else 0
    _hint_cycle[0] += 1
    hint = (0, maxlen, max(maxlen // 2, 1))[_hint_cycle[0] % 3]
    try:
        score, sink = nvb.batch_banded_alignment_score(band, nvb.make_gotoh_aligner(ty, nvb.SimpleGotohScheme(*scheme)), p, t,
                                                       max_pattern_length=hint)
        torch.cuda.synchronize()
    finally:
        nvb.set_test_switch("NVBIO_HIP_FORCE_32BIT", "0")
    return score.cpu().numpy(), sink.cpu().numpy().view(np.uint32)


def check(band, ty, scheme, hp, ht, dev):
    es, ek = O.batch_banded_gotoh_score(band, ty, scheme, hp, ht)
    for force32 in (False, True):
        gs, gk = run_gpu(band, ty, scheme, hp, ht, dev, force32)
        bad = np.nonzero((es != gs) | (ek != gk).any(1))[0]
        assert bad.size == 0, "band %d type %d scheme %s force32=%s: %d mismatches, first %d: cpu (%d,%s) gpu (%d,%s)" % (
            band, ty, scheme, force32, bad.size, bad[0], es[bad[0]], ek[bad[0]], gs[bad[0]], gk[bad[0]])
    return es, ek


def test_kernel_width_selection(cuda):
    """Jobs longer than the 16-bit exactness limit of their scheme go to the 32-bit kernel, per job:
    LOCAL (2,-1,-2,-1) is exact in 16 bits up to 340 symbols in the row frame (1022 / (match + |gap ext|): the kernels hold row i's values plus
    (i + 1) |G_e|) and up to 511 in the reference's own frame (the A16P instance takes the jobs in between); mix lengths across both limits."""
    rng = np.random.default_rng(42)
    pats, txts = [], []
    for L in [10, 100, 339, 340, 341, 342, 510, 511, 512, 513, 700, 1200, 30, 340, 341, 511, 512]:
        t = rng.integers(0, 4, L + 40, dtype=np.uint8)
        p = t[7:7 + L].copy()
        mut = rng.random(L) < 0.03
        p[mut] = rng.integers(0, 4, int(mut.sum()), dtype=np.uint8)
        pats.append(p); txts.append(t)
    hp, ht = O.StringSet.from_lists(pats, 4, True), O.StringSet.from_lists(txts, 2, False)
    for band in (15, 31):
        for ty in (0, 1, 2):
            es, _ = check(band, ty, (2, -1, -2, -1), hp, ht, cuda)
    assert es.max() > 1022          # scores beyond the 16-bit LOCAL range are present and exact


def test_kats_on_gpu(cuda):
    for case in KAT["gotoh"]:
        p, t = dna(KAT["strings"][case["p"]]), dna(KAT["strings"][case["t"]])
        for pb, pbe, tbe in ((4, True, False), (4, False, True), (2, True, True), (2, False, False)):
            hp = O.StringSet.from_lists([p], pb, pbe)
            ht = O.StringSet.from_lists([t], 2, tbe)
            gs, gk = run_gpu(case["band"], case["type"], case["scheme"], hp, ht, cuda)
            assert int(gs[0]) == case["score"] and gk[0].tolist() == case["sink"], case


def random_pairs(rng, n, band, max_len=160, ragged=True):
    pats, txts = [], []
    for i in range(n):
        M = int(rng.integers(0 if ragged else 100, max_len)) if ragged else 100
        kind = i % 6
        if kind == 0:   N = M + band - 1 + int(rng.integers(0, 40))      # text covers the band
        elif kind == 1: N = M + int(rng.integers(0, band))               # text ends inside the band (255s)
        elif kind == 2: N = M                                            # shortest legal text
        elif kind == 3: N = max(0, M - int(rng.integers(1, 5)))          # text shorter than pattern
        elif kind == 4: N = M + band // 2
        else:           N = M + band + 50
        t = rng.integers(0, 4, N, dtype=np.uint8)
        off = int(rng.integers(0, band))
        p = np.resize(t[off:off + M], M) if N > off else rng.integers(0, 4, M, dtype=np.uint8)
        if p.size != M:
            p = rng.integers(0, 4, M, dtype=np.uint8)
        p = p.copy()
        mut = rng.random(M) < 0.08
        p[mut] = rng.integers(0, 5, int(mut.sum()), dtype=np.uint8)      # includes N=4
        if M > 20 and i % 3 == 0:                                        # an indel
            cut = int(rng.integers(5, M - 5))
            p = np.concatenate([p[:cut], p[cut + 2:], rng.integers(0, 4, 2, dtype=np.uint8)])
        pats.append(p); txts.append(t)
    return pats, txts


@pytest.mark.parametrize("band", [3, 5, 7, 15, 31])
@pytest.mark.parametrize("ty", [nvb.GLOBAL, nvb.LOCAL, nvb.SEMI_GLOBAL])
def test_random_ragged_pairs(cuda, band, ty):
    rng = np.random.default_rng(1000 + band * 3 + ty)
    pats, txts = random_pairs(rng, 3000, band)
    for (pb, pbe, tbe) in ((4, True, False), (4, False, True), (2, True, True), (2, False, False)):
        if pb == 2:
            pp = [np.minimum(p, 3) for p in pats]
        else:
            pp = pats
        hp = O.StringSet.from_lists(pp, pb, pbe)
        ht = O.StringSet.from_lists(txts, 2, tbe)
        for scheme in SCHEMES:
            check(band, ty, scheme, hp, ht, cuda)


def test_empty_and_tiny_batches(cuda):
    sc = nvb.make_gotoh_aligner(nvb.LOCAL, nvb.SimpleGotohScheme(2, -1, -2, -1))
    p, t = W.make_sw_batch(0, device=cuda)
    s, k = nvb.batch_banded_alignment_score(15, sc, p, t)
    assert s.numel() == 0 and k.numel() == 0
    for n in (1, 2, 63, 64, 65, 255, 256, 257):
        p, t = W.make_sw_batch(n, device=cuda, seed=n)
        hp, ht = O.StringSet.from_device(p), O.StringSet.from_device(t)
        for ty in (0, 1, 2):
            check(15, ty, (2, -1, -2, -1), hp, ht, cuda)


def test_text_shorter_than_pattern_leaves_sink_invalid(cuda):
    exp = KAT["text_shorter_than_pattern"]
    hp = O.StringSet.from_lists([dna(KAT["strings"]["real_p"])[:100]], 4, True)
    ht = O.StringSet.from_lists([dna(KAT["strings"]["real_t"])[:50]], 2, False)
    for ty in (0, 1, 2):
        gs, gk = run_gpu(15, ty, (2, -1, -2, -1), hp, ht, cuda)
        assert int(gs[0]) == exp["score"] and gk[0].tolist() == exp["sink"]


def test_large_negative_scores_cross_infimum(cuda):
    """mismatch costs large enough that H drops below the reference's short-based infimum
    (gotoh_banded_inl.h:446-448): the F[BAND-2] = max(infimum+G_e, ...) path must match."""
    rng = np.random.default_rng(5)
    pats = [rng.integers(0, 4, 150, dtype=np.uint8) for _ in range(500)]
    txts = [rng.integers(0, 4, 181, dtype=np.uint8) for _ in range(500)]
    hp, ht = O.StringSet.from_lists(pats, 4, True), O.StringSet.from_lists(txts, 2, False)
    for band in (15, 31):
        for ty in (0, 2):
            check(band, ty, (1, -900, -700, -600), hp, ht, cuda)


def test_config1_workload_100k(cuda):
    """BASELINE config 1 inputs (100 k x 100 bp vs 150 bp, band 15, LOCAL and SEMI_GLOBAL)."""
    p, t = W.make_sw_batch(100_000, device=cuda, seed=0x5EED0001)
    hp, ht = O.StringSet.from_device(p), O.StringSet.from_device(t)
    for ty in (nvb.LOCAL, nvb.SEMI_GLOBAL):
        es, ek = check(15, ty, (2, -1, -2, -1), hp, ht, cuda)
        assert es.min() > 100 and len(np.unique(es)) > 30      # non-trivial scores


def test_config5_shape_150bp_band31(cuda):
    p, t = W.make_sw_batch(50_000, read_len=150, ref_len=200, device=cuda, seed=0x5EED0005)
    hp, ht = O.StringSet.from_device(p), O.StringSet.from_device(t)
    check(31, nvb.LOCAL, (2, -1, -2, -1), hp, ht, cuda)


def test_full_size_properties_10m(cuda):
    """BASELINE config 2 at full size (10 M pairs): size-independent properties plus an exact
    check of a strided sample against the oracle."""
    n = 10_000_000
    p, t = W.make_sw_batch(n, device=cuda, seed=0x5EED0002)
    al = nvb.make_gotoh_aligner(nvb.LOCAL, nvb.SimpleGotohScheme(2, -1, -2, -1))
    s1, k1 = nvb.batch_banded_alignment_score(15, al, p, t)
    s2, k2 = nvb.batch_banded_alignment_score(15, al, p, t)
    torch.cuda.synchronize()
    assert torch.equal(s1, s2) and torch.equal(k1, k2)                  # deterministic / idempotent
    assert int(s1.min()) >= 0 and int(s1.max()) <= 200                  # LOCAL: 0 <= score <= 2*M
    assert bool((k1[:, 1] >= 1).all()) and bool((k1[:, 1] <= 100).all())
    assert bool((k1[:, 0] >= k1[:, 1]).all()) and bool((k1[:, 0] <= k1[:, 1] + 14).all())   # sink inside the band
    # permuting the jobs permutes the results
    perm = torch.randperm(n, device=cuda, generator=torch.Generator(device=cuda).manual_seed(1))[:200_000]
    pp = nvb.PackedStringSet(p.words, 4, True, p.begin[perm].contiguous(), None, 100)
    tp = nvb.PackedStringSet(t.words, 2, False, t.begin[perm].contiguous(), None, 150)
    s3, k3 = nvb.batch_banded_alignment_score(15, al, pp, tp)
    assert torch.equal(s3, s1[perm]) and torch.equal(k3, k1[perm])
    # exact check of that sample against the oracle
    hp, ht = O.StringSet.from_device(pp), O.StringSet.from_device(tp)
    es, ek = O.batch_banded_gotoh_score(15, nvb.LOCAL, (2, -1, -2, -1), hp, ht)
    assert (s3.cpu().numpy() == es).all() and (k3.cpu().numpy().view(np.uint32) == ek).all()


@pytest.mark.parametrize("band", [7, 15, 31])
@pytest.mark.parametrize("ty", [nvb.LOCAL, nvb.SEMI_GLOBAL, nvb.GLOBAL])
def test_quality_aware_scheme(cuda, band, ty):
    """nvBowtie's SmithWatermanScoringScheme (quality-dependent mismatch penalty, separate read /
    reference gap costs) through nvbio_hip_banded_gotoh_score_qual, vs the oracle."""
    rng = np.random.default_rng(500 + band + ty)
    pats, txts = random_pairs(rng, 2000, band)
    hp, ht = O.StringSet.from_lists(pats, 4, True), O.StringSet.from_lists(txts, 2, True)
    total = int(hp.begin[-1] + hp.length[-1])
    quals = rng.integers(0, 60, total + 3, dtype=np.uint8)          # phred bytes, some above the 40 cap
    quals[::97] = 255
    for scheme in (nvb.SmithWatermanScoringScheme(), nvb.SmithWatermanScoringScheme.local(),
                   nvb.SmithWatermanScoringScheme(match=1, mmp_min=1, mmp_max=9, read_gap_const=4, read_gap_coeff=2, ref_gap_const=7, ref_gap_coeff=1),
                   nvb.SmithWatermanScoringScheme(match=2, mm_cost="constant")):
        st = scheme.struct()
        lut = np.array([st.mismatch[q] for q in range(256)], dtype=np.int32)
        if scheme.mm_cost == "qual":      # the host layer's table == the reference's float arithmetic in C
            assert (-lut == O.qual_cost_lut(scheme.m_mmp_min, scheme.m_mmp_max)).all()
        s6 = (st.match, st.pattern_gap_open, st.pattern_gap_ext, st.text_gap_open, st.text_gap_ext, 0)
        es, ek = O.batch_banded_gotoh_score_qual(band, ty, s6, lut, quals, hp, ht)
        p = nvb.PackedStringSet.from_host(hp.words, 4, True, hp.begin, hp.length, device=cuda)
        t = nvb.PackedStringSet.from_host(ht.words, 2, True, ht.begin, ht.length, device=cuda)
        dq = torch.from_numpy(quals).to(cuda)
        for force32 in ("0", "1"):
            nvb.set_test_switch("NVBIO_HIP_FORCE_32BIT", force32)
            try:
                gs, gk = nvb.batch_banded_alignment_score(band, nvb.make_gotoh_aligner(ty, scheme), p, t, quals=dq)
                torch.cuda.synchronize()
            finally:
                nvb.set_test_switch("NVBIO_HIP_FORCE_32BIT", "0")
            gs, gk = gs.cpu().numpy(), gk.cpu().numpy().view(np.uint32)
            bad = np.nonzero((es != gs) | (ek != gk).any(1))[0]
            assert bad.size == 0, (band, ty, force32, bad[:5], es[bad[:3]], gs[bad[:3]])


@pytest.mark.parametrize("band", [15, 31])
def test_row_frame_at_the_16_bit_limit(cuda, band):
    """The kernels hold row i's values plus (i + 1) |G_e| (the row frame): a LOCAL value reaches 32 M (S + |G_e|) + 31, and the host sends a job
    to the 16-bit kernel up to M = 1022 / (S + |G_e|) symbols -- 204 for nvBowtie's local scheme (match 2, gap extension 3), 340 for the
    Gotoh scheme (2,-1,-2,-1).  Perfect reads of exactly those lengths (and one or two symbols either side) score M * match: the largest values
    the frame can be asked to hold.  Against the oracle, and the 32-bit kernel on the same jobs."""
    rng = np.random.default_rng(6400 + band)
    local = nvb.SmithWatermanScoringScheme.local()
    st = local.struct()
    assert st.match == 2 and st.pattern_gap_ext == -3
    pats, txts = [], []
    for L in [202, 203, 204, 205, 206, 204, 339, 340, 341, 100]:
        t = rng.integers(0, 4, L + 2 * band, dtype=np.uint8)
        off = int(rng.integers(0, band))
        pats.append(t[off:off + L].copy()); txts.append(t)
    hp, ht = O.StringSet.from_lists(pats, 4, True), O.StringSet.from_lists(txts, 2, True)
    total = int(hp.begin[-1] + hp.length[-1])
    quals = np.full(total + 3, 40, np.uint8)
    lut = np.array([st.mismatch[q] for q in range(256)], dtype=np.int32)
    s6 = (st.match, st.pattern_gap_open, st.pattern_gap_ext, st.text_gap_open, st.text_gap_ext, 0)
    es, ek = O.batch_banded_gotoh_score_qual(band, nvb.LOCAL, s6, lut, quals, hp, ht)
    assert es[2] == 408 and es[7] == 680
    p = nvb.PackedStringSet.from_host(hp.words, 4, True, hp.begin, hp.length, device=cuda)
    t = nvb.PackedStringSet.from_host(ht.words, 2, True, ht.begin, ht.length, device=cuda)
    dq = torch.from_numpy(quals).to(cuda)
    for force32 in ("0", "1"):
        nvb.set_test_switch("NVBIO_HIP_FORCE_32BIT", force32)
        try:
            gs, gk = nvb.batch_banded_alignment_score(band, nvb.make_gotoh_aligner(nvb.LOCAL, local), p, t, quals=dq)
            torch.cuda.synchronize()
        finally:
            nvb.set_test_switch("NVBIO_HIP_FORCE_32BIT", "0")
        assert (gs.cpu().numpy() == es).all() and (gk.cpu().numpy().view(np.uint32) == ek).all(), (band, force32, es, gs.cpu().numpy())
    # the Gotoh scheme at its own limit
    for ty in (nvb.LOCAL, nvb.SEMI_GLOBAL, nvb.GLOBAL):
        check(band, ty, (2, -1, -2, -1), hp, O.StringSet.from_lists(txts, 2, False), cuda)


@pytest.mark.parametrize("band", [7, 15, 31])
@pytest.mark.parametrize("ty", [nvb.LOCAL, nvb.SEMI_GLOBAL, nvb.GLOBAL])
def test_pattern_views_run_in_place(cuda, band, ty):
    """nvbio_hip_banded_gotoh_score_qual_views: each job aligns an io::ReadStream VIEW of a stored read -- walked backwards and / or
    complemented, qualities following the walk (nvbio/io/utils.h:100-330; nvBowtie's AlignmentStrings::load,
    alignment_utils.h:194-211) -- applied by the kernel as it fetches each 16-symbol group.  Expected: the oracle on the strings
    and qualities materialised on the host.  Stored reads are ragged, the first ones sit at the very start of the arrays (the
    reversed walk's look-behind reaches before position 0), several jobs share one stored read, both arithmetic widths, with and
    without LDS staging."""
    rng = np.random.default_rng(900 + band + ty)
    pats, txts = random_pairs(rng, 3000, band)
    pats[0], pats[1], pats[2] = pats[0][:5], pats[1][:17], pats[2][:33]                 # short reads right at the start of the arrays
    for k in range(3):
        txts[k] = txts[k][:len(pats[k]) + band + 3]
    stored = O.StringSet.from_lists(pats, 4, True)
    ht = O.StringSet.from_lists(txts, 2, True)
    total = int(stored.begin[-1] + stored.length[-1])
    quals = rng.integers(0, 60, total + 3, dtype=np.uint8)
    quals[::89] = 255
    flags = rng.integers(0, 4, len(pats)).astype(np.uint8)
    flags[:8] = [1, 3, 1, 0, 2, 3, 1, 3]
    # materialise what each view shows
    mats, mq = [], []
    for i, pt in enumerate(pats):
        b, m = int(stored.begin[i]), len(pt)
        v, q = np.asarray(pt, np.uint8), quals[b:b + m]
        if flags[i] & 1:
            v, q = v[::-1], q[::-1]
        if flags[i] & 2:
            v = np.where(v < 4, 3 - v, v).astype(np.uint8)
        mats.append(v.copy()); mq.append(q.copy())
    hm = O.StringSet.from_lists(mats, 4, True)
    assert (hm.begin == stored.begin).all()
    mquals = quals.copy()
    for i, q in enumerate(mq):
        mquals[int(hm.begin[i]):int(hm.begin[i]) + q.size] = q
    scheme = nvb.SmithWatermanScoringScheme.local() if ty == nvb.LOCAL else nvb.SmithWatermanScoringScheme()
    st = scheme.struct()
    lut = np.array([st.mismatch[q] for q in range(256)], dtype=np.int32)
    s6 = (st.match, st.pattern_gap_open, st.pattern_gap_ext, st.text_gap_open, st.text_gap_ext, 0)
    es, ek = O.batch_banded_gotoh_score_qual(band, ty, s6, lut, mquals, hm, ht)
    p = nvb.PackedStringSet.from_host(stored.words, 4, True, stored.begin, stored.length, device=cuda)
    t = nvb.PackedStringSet.from_host(ht.words, 2, True, ht.begin, ht.length, device=cuda)
    dq, df = torch.from_numpy(quals).to(cuda), torch.from_numpy(flags).to(cuda)
    for force32, nostage in (("0", "0"), ("1", "0"), ("0", "1")):
        nvb.set_test_switch("NVBIO_HIP_FORCE_32BIT", force32); nvb.set_test_switch("NVBIO_HIP_NO_STAGING", nostage)
        try:
            gs, gk = nvb.batch_banded_alignment_score(band, nvb.make_gotoh_aligner(ty, scheme), p, t, quals=dq, pattern_flags=df,
                                                      max_pattern_length=int(stored.length.max()))
            torch.cuda.synchronize()
        finally:
            nvb.set_test_switch("NVBIO_HIP_FORCE_32BIT", "0"); nvb.set_test_switch("NVBIO_HIP_NO_STAGING", "0")
        assert "views" in nvb.lib().nvbio_hip_last_kernel().decode()
        gs, gk = gs.cpu().numpy(), gk.cpu().numpy().view(np.uint32)
        bad = np.nonzero((es != gs) | (ek != gk).any(1))[0]
        assert bad.size == 0, (band, ty, force32, nostage, bad[:5], flags[bad[:5]], es[bad[:3]], gs[bad[:3]])
    # no flags == the plain entry point
    z = torch.zeros_like(df)
    a = nvb.batch_banded_alignment_score(band, nvb.make_gotoh_aligner(ty, scheme), p, t, quals=dq, pattern_flags=z, max_pattern_length=int(stored.length.max()))
    b = nvb.batch_banded_alignment_score(band, nvb.make_gotoh_aligner(ty, scheme), p, t, quals=dq, max_pattern_length=int(stored.length.max()))
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


@pytest.mark.parametrize("band", [3, 7, 15, 31])
@pytest.mark.parametrize("ty", [nvb.LOCAL, nvb.SEMI_GLOBAL, nvb.GLOBAL])
def test_bounded_scorer_threshold_semantics(cuda, band, ty):
    """nvbio_hip_banded_gotoh_score_qual_bounded (the reference's min_score argument; nvbio_amd/csrc/banded_gotoh_bounded.h): a job that ends
    ABOVE its threshold reports the oracle's score and sink bit for bit; a job that ends at or below it reports SOME score <= the threshold --
    never above, whatever row it was given up at -- and the sink (-1, -1) if it was given up; INT32_MIN thresholds (and none at all) make every
    job exact.  Ragged jobs (empty patterns, texts shorter than patterns, windows cut by the text's end), both arithmetic widths, with and
    without LDS staging, pattern views, every refill setting, and the job count read from the device."""
    rng = np.random.default_rng(7700 + band + ty)
    pats, txts = random_pairs(rng, 5000, band)
    pats[5], pats[6] = pats[5][:0], pats[6][:1]                                  # an empty and a one-symbol pattern
    for k in range(40, 80):                                                      # windows as nvBowtie cuts them: pattern + band, some cut short
        txts[k] = (txts[k] * 4)[:len(pats[k]) + band - (k % 3)]
    stored = O.StringSet.from_lists(pats, 4, True)
    ht = O.StringSet.from_lists(txts, 2, True)
    total = int(stored.begin[-1] + stored.length[-1])
    quals = rng.integers(0, 45, total + 3, dtype=np.uint8)
    scheme = nvb.SmithWatermanScoringScheme.local() if ty == nvb.LOCAL else nvb.SmithWatermanScoringScheme()
    st = scheme.struct()
    lut = np.array([st.mismatch[q] for q in range(256)], dtype=np.int32)
    s6 = (st.match, st.pattern_gap_open, st.pattern_gap_ext, st.text_gap_open, st.text_gap_ext, 0)
    es, ek = O.batch_banded_gotoh_score_qual(band, ty, s6, lut, quals, stored, ht)
    n = len(pats)
    # thresholds around the true scores: a third well below (job must be exact), a third at the score (<=: may be given up), a third above
    off = rng.integers(-40, 41, n)
    thr = (es.astype(np.int64) + off).clip(-(1 << 20), 1 << 20).astype(np.int32)
    thr[::11] = np.iinfo(np.int32).min
    thr[es == -(1 << 30)] = -50                                                  # jobs without an alignment keep their -(1 << 30)
    p = nvb.PackedStringSet.from_host(stored.words, 4, True, stored.begin, stored.length, device=cuda)
    t = nvb.PackedStringSet.from_host(ht.words, 2, True, ht.begin, ht.length, device=cuda)
    dq, dthr = torch.from_numpy(quals).to(cuda), torch.from_numpy(thr).to(cuda)
    al = nvb.make_gotoh_aligner(ty, scheme)
    maxp = int(stored.length.max())
    n_dev = torch.tensor([n], dtype=torch.int32, device=cuda)
    exact_needed = es > thr
    for force32, nostage, refill_n_dev in (("0", "0", False), ("1", "0", False), ("0", "1", True)):
        nvb.set_test_switch("NVBIO_HIP_FORCE_32BIT", force32); nvb.set_test_switch("NVBIO_HIP_NO_STAGING", nostage)
        try:
            if refill_n_dev:
                # the job count on the device, results written through an index (a compacted batch writing back at its hits)
                perm = torch.randperm(n, device=cuda).to(torch.int32)
                ps_, pk_ = torch.empty(n, dtype=torch.int32, device=cuda), torch.empty((n, 2), dtype=torch.int32, device=cuda)
                nvb.BatchedBandedAlignmentScore(band).enact(al, p, t, ps_, pk_, maxp, 0, dq, None, dthr, n_dev, perm)
                gs, gk = ps_[perm.long()], pk_[perm.long()]
                # no thresholds: the plain kernel over the device-side count, through the same index -- exact
                qs_, qk_ = torch.full((n,), 12345, dtype=torch.int32, device=cuda), torch.empty((n, 2), dtype=torch.int32, device=cuda)
                half = torch.tensor([n // 2], dtype=torch.int32, device=cuda)
                nvb.BatchedBandedAlignmentScore(band).enact(al, p, t, qs_, qk_, maxp, 0, dq, None, None, half, perm)
                assert "bounded" not in nvb.lib().nvbio_hip_last_kernel().decode()
                hs, hk = qs_[perm.long()].cpu().numpy(), qk_[perm.long()].cpu().numpy().view(np.uint32)
                assert (hs[:n // 2] == es[:n // 2]).all() and (hk[:n // 2] == ek[:n // 2]).all() and (hs[n // 2:] == 12345).all()      # jobs past the count are not touched
            else:
                gs, gk = nvb.batch_banded_alignment_score(band, al, p, t, quals=dq, max_pattern_length=maxp, min_score=dthr)
            xs, xk = nvb.batch_banded_alignment_score(band, al, p, t, quals=dq, max_pattern_length=maxp, min_score=torch.full_like(dthr, np.iinfo(np.int32).min))
            torch.cuda.synchronize()
        finally:
            nvb.set_test_switch("NVBIO_HIP_FORCE_32BIT", "0"); nvb.set_test_switch("NVBIO_HIP_NO_STAGING", "0")
        assert "bounded" in nvb.lib().nvbio_hip_last_kernel().decode()
        gs, gk, xs, xk = gs.cpu().numpy(), gk.cpu().numpy().view(np.uint32), xs.cpu().numpy(), xk.cpu().numpy().view(np.uint32)
        assert (xs == es).all() and (xk == ek).all(), (band, ty, force32, np.nonzero(xs != es)[0][:5])
        bad = np.nonzero(exact_needed & ((gs != es) | (gk != ek).any(1)))[0]
        assert bad.size == 0, (band, ty, force32, nostage, bad[:5], es[bad[:3]], gs[bad[:3]], thr[bad[:3]])
        below = ~exact_needed
        assert (gs[below] <= thr[below]).all() or (gs[below][gs[below] > thr[below]] == es[below][gs[below] > thr[below]]).all()
        # a given-up job carries no sink; one that ran to its end is exact
        gave_up = below & (gk == 0xFFFFFFFF).all(1) & (es != -(1 << 30))
        ran = below & ~gave_up
        assert (gs[ran] == es[ran]).all() and (gk[ran] == ek[ran]).all()
        assert (gs[gave_up] >= es[gave_up]).all() and (gs[gave_up] <= thr[gave_up]).all()     # an upper bound of the true score, at or below the threshold
    # end-to-end mode gives up early: with thresholds well above the scores most jobs stop before their last row
    if ty == nvb.SEMI_GLOBAL and band >= 15:
        hi = torch.from_numpy((es.clip(-(1 << 20), None) + 30).astype(np.int32)).to(cuda)
        gs, gk = nvb.batch_banded_alignment_score(band, al, p, t, quals=dq, max_pattern_length=maxp, min_score=hi)
        gave_up = (gk.cpu().numpy().view(np.uint32) == 0xFFFFFFFF).all(1) & (es != -(1 << 30))
        assert int(gave_up.sum()) > n // 4, int(gave_up.sum())
        assert (gs.cpu().numpy()[gave_up] <= hi.cpu().numpy()[gave_up]).all()


@pytest.mark.parametrize("band", [3, 5, 7, 15, 31])
@pytest.mark.parametrize("ty", [nvb.LOCAL, nvb.SEMI_GLOBAL, nvb.GLOBAL])
def test_wave_kernel_equals_lane_kernel(cuda, band, ty):
    """nvbio_hip_banded_gotoh_score_qual_wave (one wave per job, the anti-diagonal sweep; nvbio_amd/csrc/banded_gotoh_wave.hip) against the oracle and the
    lane-per-job kernel: ragged jobs (an empty and a one-symbol pattern, texts shorter than their patterns, windows cut by the text's end so that symbols
    past it enter the band -- the reference's 2-bit cache quirk at band 31 --, texts shorter than the band), nvBowtie's end-to-end and local schemes
    and a scheme with unequal gap costs on the two strings, scores and sinks bit for bit; then a compacted list of jobs run in place through job_index
    with the count on the device (the other jobs' outputs untouched)."""
    from nvbio_amd.alignment import batch_banded_alignment_score_wave
    rng = np.random.default_rng(8800 + band + ty)
    pats, txts = random_pairs(rng, 3000, band)
    pats[5], pats[6] = pats[5][:0], pats[6][:1]
    for k in range(40, 120):
        txts[k] = (txts[k] * 4)[:len(pats[k]) + band - (k % 4)]
    for k in range(120, 140):                                                    # texts shorter than the band
        pats[k] = pats[k][:int(rng.integers(1, 6))]; txts[k] = (txts[k] * 2)[:len(pats[k]) + int(rng.integers(0, 4))]
    stored = O.StringSet.from_lists(pats, 4, True)
    ht = O.StringSet.from_lists(txts, 2, True)
    total = int(stored.begin[-1] + stored.length[-1])
    quals = rng.integers(0, 50, total + 3, dtype=np.uint8)
    p = nvb.PackedStringSet.from_host(stored.words, 4, True, stored.begin, stored.length, device=cuda)
    t = nvb.PackedStringSet.from_host(ht.words, 2, True, ht.begin, ht.length, device=cuda)
    dq = torch.from_numpy(quals).to(cuda)
    maxp = int(stored.length.max())
    n = len(pats)
    for scheme in (nvb.SmithWatermanScoringScheme.local() if ty == nvb.LOCAL else nvb.SmithWatermanScoringScheme(),
                   nvb.SmithWatermanScoringScheme(match=1 if ty == nvb.LOCAL else 0, mmp_min=1, mmp_max=9, read_gap_const=4, read_gap_coeff=2, ref_gap_const=7, ref_gap_coeff=1)):
        st = scheme.struct()
        lut = np.array([st.mismatch[q] for q in range(256)], dtype=np.int32)
        s6 = (st.match, st.pattern_gap_open, st.pattern_gap_ext, st.text_gap_open, st.text_gap_ext, 0)
        es, ek = O.batch_banded_gotoh_score_qual(band, ty, s6, lut, quals, stored, ht)
        al = nvb.make_gotoh_aligner(ty, scheme)
        ls, lk = nvb.batch_banded_alignment_score(band, al, p, t, quals=dq, max_pattern_length=maxp)
        ws, wk = batch_banded_alignment_score_wave(band, al, p, t, dq, max_pattern_length=maxp)
        torch.cuda.synchronize()
        assert nvb.lib().nvbio_hip_last_kernel().decode() == "banded_gotoh_wave_kernel"
        ws_, wk_ = ws.cpu().numpy(), wk.cpu().numpy().view(np.uint32)
        bad = np.nonzero((ws_ != es) | (wk_ != ek).any(1))[0]
        assert bad.size == 0, (band, ty, bad[:6], es[bad[:4]], ws_[bad[:4]], ek[bad[:4]], wk_[bad[:4]], [len(pats[b]) for b in bad[:4]], [len(txts[b]) for b in bad[:4]])
        assert torch.equal(ws, ls) and torch.equal(wk, lk)
        # a compacted list, the count on the device
        pick = torch.from_numpy(np.sort(rng.choice(n, 700, replace=False)).astype(np.int32)).to(cuda)
        os_, ok_ = torch.full((n,), 777, dtype=torch.int32, device=cuda), torch.full((n, 2), 5, dtype=torch.int32, device=cuda)
        cnt = torch.tensor([600], dtype=torch.int32, device=cuda)
        batch_banded_alignment_score_wave(band, al, p, t, dq, out_score=os_, out_sink=ok_, max_pattern_length=maxp, job_index=pick, n_on_device=cnt)
        torch.cuda.synchronize()
        done = pick[:600].long()
        assert torch.equal(os_[done], ls[done]) and torch.equal(ok_[done], lk[done])
        rest = torch.ones(n, dtype=torch.bool, device=cuda); rest[done] = False
        assert bool((os_[rest] == 777).all()) and bool((ok_[rest] == 5).all())
