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
