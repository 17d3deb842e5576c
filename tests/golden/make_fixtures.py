#!/usr/bin/env python3
"""Regenerates tests/golden/hot_path_vectors.npz: seeded inputs and the CPU oracle's outputs for the
hot path (banded Gotoh scores + sinks, FM-index ranks / ranges / located positions / seed hits).

The oracle is pinned to the reference by tests/golden/kat.json (tests/test_oracle_kat.py); these
vectors freeze its behaviour on wider random inputs so that a change in the oracle OR in the HIP
path shows up against committed data.  Run from the repo root:  python tests/golden/make_fixtures.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as O  # noqa: E402


def banded_cases(rng, n, band):
    pats, txts = [], []
    for i in range(n):
        M = int(rng.integers(1, 130))
        N = [M + band - 1 + int(rng.integers(0, 20)), M + int(rng.integers(0, band)), M, max(0, M - 2)][i % 4]
        t = rng.integers(0, 4, N, dtype=np.uint8)
        p = np.resize(t[band // 2: band // 2 + M], M).copy() if N > band // 2 else rng.integers(0, 4, M, dtype=np.uint8)
        mut = rng.random(M) < 0.07
        p[mut] = rng.integers(0, 5, int(mut.sum()), dtype=np.uint8)
        pats.append(p); txts.append(t)
    return pats, txts


def main():
    rng = np.random.default_rng(0x5EED)
    out = {}
    for band in (3, 5, 7, 15, 31):
        pats, txts = banded_cases(rng, 300, band)
        hp, ht = O.StringSet.from_lists(pats, 4, True), O.StringSet.from_lists(txts, 2, False)
        out["b%d_pw" % band], out["b%d_pb" % band], out["b%d_pl" % band] = hp.words, hp.begin, hp.length
        out["b%d_tw" % band], out["b%d_tb" % band], out["b%d_tl" % band] = ht.words, ht.begin, ht.length
        for ty in (0, 1, 2):
            for si, sc in enumerate(((2, -1, -2, -1), (0, -5, -8, -3))):
                s, k = O.batch_banded_gotoh_score(band, ty, sc, hp, ht)
                out["b%d_t%d_s%d_score" % (band, ty, si)] = s
                out["b%d_t%d_s%d_sink" % (band, ty, si)] = k
    text = rng.integers(0, 4, 40000, dtype=np.uint8)
    text[3000:3400] = 1
    f = O.FMIndex(text)
    out["fm_text"], out["fm_bwt_occ"], out["fm_ssa"], out["fm_L2"] = text, f.bwt_occ, f.ssa, f.L2
    out["fm_primary"] = np.array([f.primary], dtype=np.uint32)
    k = rng.integers(0, 40001, 3000).astype(np.uint32)
    c = rng.integers(0, 4, 3000).astype(np.uint8)
    out["fm_k"], out["fm_c"], out["fm_rank"], out["fm_rank4"] = k, c, f.rank(k, c), f.rank4(k)
    seeds = [text[p:p + 14].copy() if i % 5 else rng.integers(0, 4, 14, dtype=np.uint8) for i, p in enumerate(rng.integers(0, 39000, 1500))]
    ss = O.StringSet.from_lists(seeds, 2, True)
    rg = f.match(ss)
    out["fm_sw"], out["fm_sb"], out["fm_sl"], out["fm_ranges"] = ss.words, ss.begin, ss.length, rg
    rows = rng.integers(0, 40001, 3000).astype(np.uint32)
    out["fm_rows"], out["fm_pos"] = rows, f.locate(rows)
    reads = [text[p:p + 60][::-1].copy() for p in rng.integers(0, 39000, 300)]
    hr = O.StringSet.from_lists(reads, 4, True)
    sf = O.simple_func_table(2, 1.0, 1.15, 64)
    pd = dict(seed_len=22, min_read_len=12, max_hits=100, max_reseed=2, retry=0, rep_seeds=300, fw=1, rc=1)
    h, cnt, rs = O.map_exact(f, hr, pd, sf, 16)
    out["map_rw"], out["map_rb"], out["map_rl"], out["map_sf"] = hr.words, hr.begin, hr.length, sf
    out["map_hits"], out["map_counts"], out["map_reseed"] = h, cnt, rs
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "hot_path_vectors.npz"), **out)
    print("wrote", len(out), "arrays")


if __name__ == "__main__":
    main()
