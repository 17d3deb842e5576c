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


def extension_vectors():
    """tests/golden/extension_vectors.npz: the rows built around the path -- tracebacks, full-matrix scores in
    both algorithm tags, SW / ED, the one-mismatch seed mappers, score reduction and MAPQ."""
    rng = np.random.default_rng(0xE87)
    out = {}
    pats, txts = banded_cases(rng, 200, 15)
    hp, ht = O.StringSet.from_lists(pats, 4, True), O.StringSet.from_lists(txts + [np.zeros(64, np.uint8)], 2, True)
    ht = O.StringSet(ht.words, 2, True, ht.begin[:-1], ht.length[:-1])
    out["tb_pw"], out["tb_pb"], out["tb_pl"], out["tb_tw"], out["tb_tb"], out["tb_tl"] = hp.words, hp.begin, hp.length, ht.words, ht.begin, ht.length
    for ty in (0, 1, 2):
        r = O.batch_banded_gotoh_traceback(15, ty, (2, -1, -2, -1), hp, ht, 48)
        for k, v in r.items():
            out["tb_t%d_%s" % (ty, k)] = v
    # full matrix
    fp, ft = [], []
    for i in range(200):
        M, N = int(rng.integers(1, 120)), int(rng.integers(1, 260))
        t = rng.integers(0, 4, N, dtype=np.uint8)
        p = np.resize(t[int(rng.integers(0, N)):], M).copy()
        mut = rng.random(M) < 0.08
        p[mut] = rng.integers(0, 4, int(mut.sum()), dtype=np.uint8)
        fp.append(p); ft.append(t)
    hp, ht = O.StringSet.from_lists(fp, 4, True), O.StringSet.from_lists(ft, 2, False)
    out["fu_pw"], out["fu_pb"], out["fu_pl"], out["fu_tw"], out["fu_tb"], out["fu_tl"] = hp.words, hp.begin, hp.length, ht.words, ht.begin, ht.length
    ms = rng.integers(-40, 160, 200).astype(np.int32)
    out["fu_min_score"] = ms
    for ty in (0, 1, 2):
        s, k, ok = O.batch_gotoh_score(ty, (2, -1, -2, -1), hp, ht, min_score=ms)
        out["fu_tb_t%d_score" % ty], out["fu_tb_t%d_sink" % ty], out["fu_tb_t%d_ok" % ty] = s, k, ok
        s, k, ok = O.batch_score_pattern_blocking(0, ty, (2, -1, -2, -1), hp, ht, min_score=ms)
        out["fu_pb_t%d_score" % ty], out["fu_pb_t%d_sink" % ty], out["fu_pb_t%d_ok" % ty] = s, k, ok
        s, k = O.batch_sw_score(0, ty, (0, -1, -1, -1), hp, ht)
        out["fu_ed_t%d_score" % ty], out["fu_ed_t%d_sink" % ty] = s, k
        r = O.batch_gotoh_traceback(ty, (2, -1, -2, -1), hp, ht, 48)
        for kk, v in r.items():
            out["fu_tr_t%d_%s" % (ty, kk)] = v
    # one-mismatch seed mappers on a forward + reverse index
    text = rng.integers(0, 4, 30000, dtype=np.uint8)
    out["mm_text"] = text
    f, rf = O.FMIndex(text), O.FMIndex(text[::-1].copy())
    reads = []
    for i, p0 in enumerate(rng.integers(0, 29000, 250)):
        r = text[p0:p0 + 70].copy()
        if i % 2:
            r = (3 - r)[::-1].copy()
        mut = rng.random(70) < 0.03
        r[mut] = rng.integers(0, 4, int(mut.sum()), dtype=np.uint8)
        if i % 13 == 0:
            r[int(rng.integers(0, 70))] = 4
        reads.append(r[::-1].copy())
    hr = O.StringSet.from_lists(reads, 4, True)
    out["mm_rw"], out["mm_rb"], out["mm_rl"] = hr.words, hr.begin, hr.length
    sf = O.simple_func_table(2, 1.0, 1.15, 80)
    out["mm_sf"] = sf
    pd = dict(seed_len=22, min_read_len=12, max_hits=100, max_reseed=2, retry=0, rep_seeds=300, fw=1, rc=1)
    for algo, sub in ((1, 12), (2, 0)):
        h, c, rs = O.map_seeds(algo, sub, f, rf, hr, pd, sf, 96)
        out["mm_a%d_hits" % algo], out["mm_a%d_counts" % algo], out["mm_a%d_reseed" % algo] = np.sort(np.where(np.arange(96)[None, :] < c[:, None], h, np.uint64(2**64 - 1)), axis=1), c, rs
    # score reduction + MAPQ
    n_reads = 2000
    read_len = rng.integers(40, 200, n_reads).astype(np.uint32)
    counts = rng.integers(0, 7, n_reads)
    hb = np.zeros(n_reads + 1, np.uint64); hb[1:] = np.cumsum(counts)
    tot = int(hb[-1])
    owner = np.repeat(np.arange(n_reads), counts)
    locus = rng.integers(0, 1 << 28, n_reads)
    loc = (locus[owner] + rng.choice([0, 0, 3, 50, 90, 4000], tot)).astype(np.uint32)
    score = (-rng.integers(0, 70, tot)).astype(np.int32)
    rc = (rng.random(tot) < 0.3).astype(np.uint8)
    out["rd_read_len"], out["rd_hit_begin"], out["rd_score"], out["rd_loc"], out["rd_rc"] = read_len, hb, score, loc, rc
    best = O.init_alignments(read_len, (0, -0.6, -0.6))
    O.score_reduce(best, hb, score, loc, rc, read_len)
    out["rd_best"] = best
    out["rd_mapq2"] = O.mapq(2, 0, (0, -0.6, -0.6), True, best, read_len)
    out["rd_mapq3"] = O.mapq(3, 0, (0, -0.6, -0.6), True, best, read_len)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "extension_vectors.npz"), **out)
    print("wrote", len(out), "arrays (extension)")


if __name__ == "__main__":
    if "--extension-only" not in sys.argv:
        main()
    extension_vectors()
