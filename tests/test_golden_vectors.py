"""Committed golden vectors (tests/golden/hot_path_vectors.npz, made by tests/golden/make_fixtures.py):
the oracle must still reproduce them (CPU), and the HIP path must reproduce them (GPU)."""
import os

import numpy as np
import pytest

from oracle import pyoracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, "golden", "hot_path_vectors.npz"))
SCHEMES = ((2, -1, -2, -1), (0, -5, -8, -3))
MAP_PARAMS = dict(seed_len=22, min_read_len=12, max_hits=100, max_reseed=2, retry=0, rep_seeds=300, fw=1, rc=1)


def host_index():
    return O.FMIndex(parts=(int(G["fm_text"].size), int(G["fm_primary"][0]), G["fm_L2"], G["fm_bwt_occ"], G["fm_ssa"], 16))


def test_oracle_reproduces_golden_vectors():
    for band in (3, 5, 7, 15, 31):
        hp = O.StringSet(G["b%d_pw" % band], 4, True, G["b%d_pb" % band], G["b%d_pl" % band])
        ht = O.StringSet(G["b%d_tw" % band], 2, False, G["b%d_tb" % band], G["b%d_tl" % band])
        for ty in (0, 1, 2):
            for si, sc in enumerate(SCHEMES):
                s, k = O.batch_banded_gotoh_score(band, ty, sc, hp, ht)
                assert (s == G["b%d_t%d_s%d_score" % (band, ty, si)]).all() and (k == G["b%d_t%d_s%d_sink" % (band, ty, si)]).all()
    f = O.FMIndex(G["fm_text"])
    assert (f.bwt_occ == G["fm_bwt_occ"]).all() and (f.ssa == G["fm_ssa"]).all() and f.primary == int(G["fm_primary"][0])
    assert (f.rank(G["fm_k"], G["fm_c"]) == G["fm_rank"]).all() and (f.rank4(G["fm_k"]) == G["fm_rank4"]).all()
    assert (f.match(O.StringSet(G["fm_sw"], 2, True, G["fm_sb"], G["fm_sl"])) == G["fm_ranges"]).all()
    assert (f.locate(G["fm_rows"]) == G["fm_pos"]).all()
    h, c, r = O.map_exact(f, O.StringSet(G["map_rw"], 4, True, G["map_rb"], G["map_rl"]), MAP_PARAMS, G["map_sf"], 16)
    assert (h == G["map_hits"]).all() and (c == G["map_counts"]).all() and (r == G["map_reseed"]).all()


@pytest.mark.gpu
def test_hip_path_reproduces_golden_vectors(cuda):
    import torch
    import nvbio_amd as nvb

    def dev_set(w, bits, be, b, ln):
        return nvb.PackedStringSet.from_host(w, bits, be, b, ln, device=cuda)

    for band in (3, 5, 7, 15, 31):
        p = dev_set(G["b%d_pw" % band], 4, True, G["b%d_pb" % band], G["b%d_pl" % band])
        t = dev_set(G["b%d_tw" % band], 2, False, G["b%d_tb" % band], G["b%d_tl" % band])
        for ty in (0, 1, 2):
            for si, sc in enumerate(SCHEMES):
                s, k = nvb.batch_banded_alignment_score(band, nvb.make_gotoh_aligner(ty, nvb.SimpleGotohScheme(*sc)), p, t)
                assert (s.cpu().numpy() == G["b%d_t%d_s%d_score" % (band, ty, si)]).all()
                assert (k.cpu().numpy().view(np.uint32) == G["b%d_t%d_s%d_sink" % (band, ty, si)]).all()
    fmi = nvb.FMIndexDevice.from_host(host_index(), cuda)
    i32 = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).to(cuda)
    u32 = lambda x: x.cpu().numpy().view(np.uint32)
    assert (u32(nvb.rank(fmi, i32(G["fm_k"]), torch.from_numpy(G["fm_c"]).to(cuda))) == G["fm_rank"]).all()
    assert (u32(nvb.rank4(fmi, i32(G["fm_k"]))) == G["fm_rank4"]).all()
    assert (u32(nvb.match(fmi, dev_set(G["fm_sw"], 2, True, G["fm_sb"], G["fm_sl"]))) == G["fm_ranges"]).all()
    assert (u32(nvb.locate(fmi, i32(G["fm_rows"]))) == G["fm_pos"]).all()
    h, c, r = nvb.map_exact(fmi, dev_set(G["map_rw"], 4, True, G["map_rb"], G["map_rl"]), nvb.MappingParams(), 63, hits_stride=16)
    hh, cc = h.cpu().numpy().view(np.uint64), c.cpu().numpy().view(np.uint32)
    assert (cc == G["map_counts"]).all() and (r.cpu().numpy() == G["map_reseed"]).all()
    for i in range(cc.size):
        assert (np.sort(hh[i, :cc[i]]) == np.sort(G["map_hits"][i, :cc[i]])).all()


# ---------------------------------------------------------------------------- the rows built around the path
E = np.load(os.path.join(HERE, "golden", "extension_vectors.npz"))
MM_PARAMS = dict(seed_len=22, min_read_len=12, max_hits=100, max_reseed=2, retry=0, rep_seeds=300, fw=1, rc=1)
TB_KEYS = ("score", "sink", "source", "cigar", "cigar_len")


def _sorted_hits(h, c, stride):
    return np.sort(np.where(np.arange(stride)[None, :] < c[:, None], h, np.uint64(2**64 - 1)), axis=1)


def test_oracle_reproduces_extension_vectors():
    hp = O.StringSet(E["tb_pw"], 4, True, E["tb_pb"], E["tb_pl"]); ht = O.StringSet(E["tb_tw"], 2, True, E["tb_tb"], E["tb_tl"])
    for ty in (0, 1, 2):
        r = O.batch_banded_gotoh_traceback(15, ty, (2, -1, -2, -1), hp, ht, 48)
        for k in TB_KEYS:
            assert (r[k] == E["tb_t%d_%s" % (ty, k)]).all(), (ty, k)
    hp = O.StringSet(E["fu_pw"], 4, True, E["fu_pb"], E["fu_pl"]); ht = O.StringSet(E["fu_tw"], 2, False, E["fu_tb"], E["fu_tl"])
    ms = E["fu_min_score"]
    for ty in (0, 1, 2):
        for tag, fn in (("tb", lambda: O.batch_gotoh_score(ty, (2, -1, -2, -1), hp, ht, min_score=ms)),
                        ("pb", lambda: O.batch_score_pattern_blocking(0, ty, (2, -1, -2, -1), hp, ht, min_score=ms))):
            s, k, ok = fn()
            assert (s == E["fu_%s_t%d_score" % (tag, ty)]).all() and (k == E["fu_%s_t%d_sink" % (tag, ty)]).all() and (ok == E["fu_%s_t%d_ok" % (tag, ty)]).all()
        s, k = O.batch_sw_score(0, ty, (0, -1, -1, -1), hp, ht)
        assert (s == E["fu_ed_t%d_score" % ty]).all() and (k == E["fu_ed_t%d_sink" % ty]).all()
        r = O.batch_gotoh_traceback(ty, (2, -1, -2, -1), hp, ht, 48)
        for kk in TB_KEYS:
            assert (r[kk] == E["fu_tr_t%d_%s" % (ty, kk)]).all(), (ty, kk)
    f, rf = O.FMIndex(E["mm_text"]), O.FMIndex(E["mm_text"][::-1].copy())
    hr = O.StringSet(E["mm_rw"], 4, True, E["mm_rb"], E["mm_rl"])
    for algo, sub in ((1, 12), (2, 0)):
        h, c, rs = O.map_seeds(algo, sub, f, rf, hr, MM_PARAMS, E["mm_sf"], 96)
        assert (c == E["mm_a%d_counts" % algo]).all() and (rs == E["mm_a%d_reseed" % algo]).all() and (_sorted_hits(h, c, 96) == E["mm_a%d_hits" % algo]).all()
    best = O.init_alignments(E["rd_read_len"], (0, -0.6, -0.6))
    O.score_reduce(best, E["rd_hit_begin"], E["rd_score"], E["rd_loc"], E["rd_rc"], E["rd_read_len"])
    assert (best == E["rd_best"]).all()
    assert (O.mapq(2, 0, (0, -0.6, -0.6), True, best, E["rd_read_len"]) == E["rd_mapq2"]).all()
    assert (O.mapq(3, 0, (0, -0.6, -0.6), True, best, E["rd_read_len"]) == E["rd_mapq3"]).all()


@pytest.mark.gpu
def test_hip_path_reproduces_extension_vectors(cuda):
    import torch
    import nvbio_amd as nvb

    def dev_set(w, bits, be, b, ln):
        return nvb.PackedStringSet.from_host(w, bits, be, b, ln, device=cuda)

    def same(got, key, exp):
        g = got[key].cpu().numpy()
        n = exp.shape[0]
        return (g.view(exp.dtype)[:n] == exp).all()

    p, t = dev_set(E["tb_pw"], 4, True, E["tb_pb"], E["tb_pl"]), dev_set(E["tb_tw"], 2, True, E["tb_tb"], E["tb_tl"])
    gotoh = lambda ty, algo=nvb.TEXT_BLOCKING: nvb.make_gotoh_aligner(ty, nvb.SimpleGotohScheme(2, -1, -2, -1), algo)
    for ty in (0, 1, 2):
        got = nvb.batch_banded_alignment_traceback(15, gotoh(ty), p, t, max_pattern_length=int(E["tb_pl"].max()), cigar_stride=48)
        for k in TB_KEYS:
            assert same(got, k, E["tb_t%d_%s" % (ty, k)]), (ty, k)
    p, t = dev_set(E["fu_pw"], 4, True, E["fu_pb"], E["fu_pl"]), dev_set(E["fu_tw"], 2, False, E["fu_tb"], E["fu_tl"])
    maxM, maxN = int(E["fu_pl"].max()), int(E["fu_tl"].max())
    ms = torch.from_numpy(E["fu_min_score"]).to(cuda)
    for ty in (0, 1, 2):
        for tag, algo in (("tb", nvb.TEXT_BLOCKING), ("pb", nvb.PATTERN_BLOCKING)):
            s, k, ok = nvb.batch_alignment_score(gotoh(ty, algo), p, t, maxM, maxN, ms)
            assert (s.cpu().numpy() == E["fu_%s_t%d_score" % (tag, ty)]).all() and (k.cpu().numpy().view(np.uint32) == E["fu_%s_t%d_sink" % (tag, ty)]).all()
            assert (ok.cpu().numpy() == E["fu_%s_t%d_ok" % (tag, ty)]).all()
        s, k, _ = nvb.batch_alignment_score(nvb.make_edit_distance_aligner(ty), p, t, maxM, maxN)
        assert (s.cpu().numpy() == E["fu_ed_t%d_score" % ty]).all() and (k.cpu().numpy().view(np.uint32) == E["fu_ed_t%d_sink" % ty]).all()
        got = nvb.batch_alignment_traceback(gotoh(ty), p, t, maxM, maxN, cigar_stride=48)
        for kk in TB_KEYS:
            assert same(got, kk, E["fu_tr_t%d_%s" % (ty, kk)]), (ty, kk)
    fmi = nvb.FMIndexDevice.from_host(O.FMIndex(E["mm_text"]), cuda)
    rfmi = nvb.FMIndexDevice.from_host(O.FMIndex(E["mm_text"][::-1].copy()), cuda)
    reads = dev_set(E["mm_rw"], 4, True, E["mm_rb"], E["mm_rl"])
    for algo, sub in ((1, 12), (2, 0)):
        h, c, rs = nvb.map_seeds(fmi, rfmi, reads, nvb.MappingParams(), 79, allow_sub=1, subseed_len=sub, hits_stride=96)
        cc = c.cpu().numpy().view(np.uint32)
        assert (cc == E["mm_a%d_counts" % algo]).all() and (rs.cpu().numpy() == E["mm_a%d_reseed" % algo]).all()
        assert (_sorted_hits(h.cpu().numpy().view(np.uint64), cc, 96) == E["mm_a%d_hits" % algo]).all()
    i32 = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).to(cuda)
    sch = nvb.SmithWatermanScoringScheme()
    rl = i32(E["rd_read_len"])
    best = nvb.BestAlignments(E["rd_read_len"].size, sch, read_len=rl, max_read_len=200, device=cuda)
    nvb.score_reduce(best, torch.from_numpy(E["rd_hit_begin"].view(np.int64)).to(cuda), i32(E["rd_score"]), i32(E["rd_loc"]),
                     torch.from_numpy(E["rd_rc"]).to(cuda), read_len=rl)
    assert (best.data.cpu().numpy().view(np.uint64) == E["rd_best"]).all()
    assert (nvb.mapq(best, sch, read_len=rl, version=2, max_read_len=200).cpu().numpy() == E["rd_mapq2"]).all()
    assert (nvb.mapq(best, sch, read_len=rl, version=3, max_read_len=200).cpu().numpy() == E["rd_mapq3"]).all()


def test_reference_compiled_vectors_pin_the_oracle():
    """tests/golden/ref_basic_vectors.npz holds outputs of REFERENCE code compiled from its own sources (popcount.h, bwt.h +
    sais, priority_deque.h; generator: tests/golden/make_ref_basic_vectors.py).  The oracle's index construction, rank and
    hit deque must reproduce them."""
    v = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_basic_vectors.npz"))
    host = O.FMIndex(v["text"])
    assert (host.sa == v["sa"].astype(np.uint32)).all() and host.primary == int(v["primary"]) and (host.bwt == v["bwt"]).all()
    assert (host.bwt_occ[: v["bwt_occ"].size] == v["bwt_occ"]).all() and (host.L2 == v["L2"]).all()
    idx, sym = v["rank_idx"], v["rank_sym"]
    k = np.where(idx == 0xFFFFFFFF, idx, np.where(idx >= host.primary, idx + 1, idx)).astype(np.uint32)
    assert (host.rank(k, sym) == v["rank"]).all()
    push, pop_bottom, pop_top = O.hit_deque_ops()
    a = np.zeros(v["deque_ops"].size + 1, np.uint64)
    size = 0
    for i, op in enumerate(v["deque_ops"]):
        if op == 0:
            a[size] = v["deque_values"][i]; size += 1; push(a, size)
        elif op == 1 and size:
            pop_top(a, size); size -= 1
        elif op == 2 and size:
            pop_bottom(a, size); size -= 1
        assert size == v["deque_sizes"][i]
        if size:
            assert a[0] == v["deque_bottoms"][i] and (a[1] if size > 1 else a[0]) == v["deque_tops"][i]
    assert (a[:size] == v["deque_final"]).all()


@pytest.mark.gpu
def test_hip_rank_and_hit_deque_reproduce_reference_compiled_vectors(cuda):
    """the HIP rank kernels (reference layout and through the line-native index's plane records) and the device hit deque against
    outputs of the reference's own compiled code -- no oracle call in between"""
    import ctypes as C
    import torch
    import nvbio_amd as nvb
    from nvbio_amd._lib import lib
    v = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_basic_vectors.npz"))
    n, primary = int(v["text"].size), int(v["primary"])
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int32 if a.dtype == np.uint32 else a.dtype)).to(cuda)
    fmi = nvb.FMIndexDevice(n, primary, [int(x) for x in v["L2"]], dev(np.concatenate([v["bwt_occ"], np.zeros(8, np.uint32)])))
    idx, sym = v["rank_idx"], v["rank_sym"]
    k = np.where(idx == 0xFFFFFFFF, idx, np.where(idx >= primary, idx + 1, idx)).astype(np.uint32)
    got = nvb.rank(fmi, dev(k), dev(sym)).cpu().numpy().view(np.uint32)
    assert (got == v["rank"]).all()
    r4 = nvb.rank4(fmi, dev(k)).cpu().numpy().view(np.uint32)
    assert (r4[np.arange(k.size), sym] == v["rank"]).all()
    # one backward-search step per symbol on the line-native index = L2 + rank + 1 of the same counts
    fd = fmi.with_dimer()
    one = [np.array([c], np.uint8) for c in range(4)]
    from oracle import pyoracle as O
    hs = O.StringSet.from_lists(one, 2, True)
    ds = nvb.PackedStringSet.from_host(hs.words, 2, True, hs.begin, hs.length, device=cuda)
    rg = nvb.match(fd, ds).cpu().numpy().view(np.uint32)
    for c in range(4):
        assert int(rg[c, 0]) == int(v["L2"][c]) + 1 and int(rg[c, 1]) == int(v["L2"][c + 1])
    # the device hit deque replays the reference priority_deque's program: the same array after every operation
    from tests.test_select_gpu import device_hit_deque_replay
    states = device_hit_deque_replay(cuda, v["deque_ops"], v["deque_values"], v["deque_sizes"])
    assert (states == v["deque_states"]).all()
