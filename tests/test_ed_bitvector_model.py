"""The edit-distance bit-vector kernel's algorithm (nvbio_amd/csrc/full_gotoh.hip: edit_distance_bitvector_kernel -- Myers' column of
vertical differences in 64-row words with Hyyro's carries) restated with Python integers and checked on the CPU against the oracle's
matrix-filling edit distance: same scores and sinks for SEMI_GLOBAL (last best column of the last row) and GLOBAL, patterns over one
to nine words, N symbols, texts shorter and longer than the pattern.  The GPU test compares the kernel itself with the same oracle."""
import numpy as np
import pytest

from oracle import pyoracle as O

GLOBAL, SEMI = 0, 2
WORD = 64
MASK = (1 << WORD) - 1


def advance_block(pv, mv, eq, hin):
    """one 64-row word, one text symbol: returns (pv, mv, hout, ph, mh before the shift)"""
    xv = eq | mv
    if hin < 0:
        eq |= 1
    xh = ((((eq & pv) + pv) & MASK) ^ pv) | eq
    ph = (mv | ~(xh | pv)) & MASK
    mh = pv & xh
    hout = (ph >> (WORD - 1)) - (mh >> (WORD - 1))
    ph0, mh0 = ph, mh
    ph = (ph << 1) & MASK
    mh = (mh << 1) & MASK
    if hin < 0:
        mh |= 1
    elif hin > 0:
        ph |= 1
    return (mh | ~(xv | ph)) & MASK, ph & xv, hout, ph0, mh0


def edit_distance_bitvector(pattern, text, aln_type):
    M, N = len(pattern), len(text)
    if M == 0 or N == 0:
        return None
    W = (M + WORD - 1) // WORD
    peq = [[0] * W for _ in range(4)]
    for r, c in enumerate(pattern):
        if c < 4:
            peq[c][r // WORD] |= 1 << (r % WORD)
    pv, mv = [MASK] * W, [0] * W
    lw, top = (M - 1) // WORD, (M - 1) % WORD
    d, best_d, best_i = M, None, 0
    for i, c in enumerate(text):
        hin = 1 if aln_type == GLOBAL else 0
        dd = 0
        for w in range(W):
            pv[w], mv[w], hout, ph, mh = advance_block(pv[w], mv[w], peq[c][w], hin)
            if w == lw:
                dd = ((ph >> top) & 1) - ((mh >> top) & 1)
            hin = hout
        d += dd
        if best_d is None or d <= best_d:
            best_d, best_i = d, i + 1
    return (-best_d, best_i, M) if aln_type == SEMI else (-d, N, M)


@pytest.mark.parametrize("aln_type", [GLOBAL, SEMI])
def test_bitvector_edit_distance_equals_the_matrix(aln_type):
    rng = np.random.default_rng(70 + aln_type)
    pats, txts = [], []
    for i in range(260):
        M = int(rng.choice([1, 5, 63, 64, 65, 100, 128, 129, 150, 200, 320, 513])) if i % 2 else int(rng.integers(1, 400))
        N = int(rng.integers(1, 2 * M + 60))
        t = rng.integers(0, 4, N, dtype=np.uint8)
        if N > M + 2 and i % 3:
            o = int(rng.integers(0, N - M)); p = t[o:o + M].copy()
            mut = rng.random(M) < 0.1
            p[mut] = rng.integers(0, 5, int(mut.sum()), dtype=np.uint8)
        else:
            p = rng.integers(0, 5, M, dtype=np.uint8)
        pats.append(p.astype(np.uint8)); txts.append(t)
    hp, ht = O.StringSet.from_lists(pats, 4, True), O.StringSet.from_lists(txts, 2, False)
    es, ek = O.batch_sw_score(0, aln_type, (0, -1, -1, -1), hp, ht)
    for i, (p, t) in enumerate(zip(pats, txts)):
        s, x, y = edit_distance_bitvector(p.tolist(), t.tolist(), aln_type)
        assert (s, x, y) == (int(es[i]), int(ek[i, 0]), int(ek[i, 1])), (i, len(p), len(t))
