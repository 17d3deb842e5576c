"""The line-native two-symbol index on the CPU suite: the model of its layout and query formulas
(tests/dimer_model.py, a transcription of nvbio_amd/csrc/fmindex_dimer.h) must return the reference's
match ranges -- including the raw (x,y) of the step that empties a range and (1,0) on an N
(nvbio/fmindex/fmindex_inl.h:307-341) -- and the reference's locate iterators (fmindex_inl.h:511-545),
as the oracle computes them on the reference layout."""
import numpy as np
import pytest

from oracle import pyoracle as O
from tests import dimer_model as DM


@pytest.mark.parametrize("n,sa_int", [(5000, 16), (127, 4), (128, 16), (255, 8), (1, 16), (2, 2), (3000, 64)])
def test_model_matches_oracle(n, sa_int):
    rng = np.random.default_rng(n)
    text = rng.integers(0, 4, n, dtype=np.uint8)
    if n > 1000:
        text[n // 3: n // 3 + 300] = 0           # long A run: wide ranges, the AA filler case
        text[10:40] = np.tile([0, 1], 15)
    host = O.FMIndex(text, sa_int=sa_int)
    m = DM.Model(DM.build(text, host.sa, host.L2))
    m3 = DM.Model(DM.build(text, host.sa, host.L2), DM.build_trimer(text, host.sa, host.L2)) if n <= 5000 else None
    seeds = []
    for i in range(600):
        L = int(rng.integers(1, 40))
        if i % 4 != 3 and n > L:
            p = int(rng.integers(0, n - L + 1))
            s = text[p:p + L].copy()
            if i % 8 == 1 and L > 2:
                s[int(rng.integers(0, L))] ^= 1              # a near miss: empties late
        else:
            s = rng.integers(0, 4, L, dtype=np.uint8)
        if i % 25 == 7:
            s[int(rng.integers(0, L))] = 4
        seeds.append(s)
    ss = O.StringSet.from_lists(seeds, 4, True)
    exp = host.match(ss)
    for s, e in zip(seeds, exp):
        assert m.match(s) == (int(e[0]), int(e[1])), (s, e)
        if m3 is not None:
            assert m3.match(s) == (int(e[0]), int(e[1])), (s, e)        # three symbols per step: same ranges
    rows = np.arange(n + 1, dtype=np.uint32)
    its = host.locate_ssa_iterator(rows)
    for j in range(n + 1):
        assert m.locate_it(j, sa_int) == (int(its[j, 0]), int(its[j, 1])), j


def test_traffic_model_halves_the_lines():
    """Records touched per 22-mer on a 64 k text: 11 dimer steps, the first four (ranges wider than a record) with
    two records -- against 22 steps and ~30 records on the reference layout."""
    rng = np.random.default_rng(7)
    n = 1 << 16
    text = rng.integers(0, 4, n, dtype=np.uint8)
    host = O.FMIndex(text)
    m = DM.Model(DM.build(text, host.sa, host.L2))
    seeds = [text[p:p + 22] for p in rng.integers(0, n - 22, 300)]
    for s in seeds:
        m.match(s)
    assert m.lines / len(seeds) < 16.0
