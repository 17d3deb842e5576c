"""CPU tests of the oracle's restatement of nvBowtie's hit deque, probability tree and hit-selection stage
(oracle/nvbio_oracle.c; reference: nvbio/basic/interval_heap.h, priority_deque.h, sum_tree_inl.h,
nvBowtie/bowtie2/cuda/select_inl.h, select.cu, reduce.h)."""
import ctypes
import os

import numpy as np
import pytest

from oracle import pyoracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SO = os.path.join(os.path.dirname(HERE), "oracle", "_ref", "libref_hit_deque.so")


def replay(push, pop_bottom, pop_top, G, check, make=O.hit_deque_make):
    ops, vals, caps, sizes, states, cs = (G[k] for k in ("ops", "vals", "caps", "sizes", "states", "case_start"))
    off = 0
    for c in range(cs.size - 1):
        a = np.zeros(64, np.uint64)
        n = 0
        for i in range(int(cs[c]), int(cs[c + 1])):
            if ops[i] == 0:
                if n == caps[i]:
                    pop_bottom(a, n); n -= 1
                a[n] = vals[i]; n += 1
                push(a, n)
            elif ops[i] == 1:
                pop_top(a, n); n -= 1
            elif ops[i] == 2:
                pop_bottom(a, n); n -= 1
            elif ops[i] == 3:                      # a selection shrinks the range of the hit in one slot, in place
                slot, size = int(vals[i]) & 0xFFFFFFFF, int(vals[i]) >> 32
                a[slot] = np.uint64((int(a[slot]) & ~(0xFFFFF << 32)) | (size << 32))
            else:                                  # hits[read_id] of the next selection round: the heap is rebuilt
                make(a, n)
            assert n == sizes[i]
            check(a[:n], states[off:off + n], c, i)
            off += n
    assert off == states.size


def test_hit_deque_matches_reference_vectors():
    """The golden programs were run through the reference's interval_heap.h (tests/golden/make_hit_deque_vectors.py): after
    every push / pop_top / pop_bottom -- and every rebuild over hits whose ranges shrank in place, the selection kernels'
    make_interval_heap -- the restated heap holds the same array, ties included."""
    G = np.load(os.path.join(HERE, "golden", "hit_deque_vectors.npz"))
    push, pop_bottom, pop_top = O.hit_deque_ops()

    def check(a, want, c, i):
        assert (a == want).all(), (c, i)
    replay(push, pop_bottom, pop_top, G, check)


@pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref not built (needs /root/reference: `make -C oracle ref`)")
def test_hit_deque_matches_reference_build_live():
    """Fresh random programs against the compiled reference header itself."""
    ref = ctypes.CDLL(REF_SO)
    P = ctypes.POINTER(ctypes.c_uint64)
    push, pop_bottom, pop_top = O.hit_deque_ops()
    rng = np.random.default_rng(77)
    for trial in range(600):
        cap = int(rng.integers(1, 60)); maxsz = int(rng.choice([2, 5, 1000]))
        a = np.zeros(64, np.uint64); b = np.zeros(64, np.uint64); n = 0
        for step in range(int(rng.integers(1, 150))):
            u = rng.random()
            if u < 0.6 or n == 0:
                if n == cap:
                    pop_bottom(a, n); ref.ref_hit_deque_pop_bottom(b.ctypes.data_as(P), n); n -= 1
                v = (int(rng.integers(1, maxsz)) << 32) | int(rng.integers(0, 1 << 32))
                a[n] = v; b[n] = v; n += 1
                push(a, n); ref.ref_hit_deque_push(b.ctypes.data_as(P), n)
            elif u < 0.7:
                pop_top(a, n); ref.ref_hit_deque_pop_top(b.ctypes.data_as(P), n); n -= 1
            elif u < 0.8:
                pop_bottom(a, n); ref.ref_hit_deque_pop_bottom(b.ctypes.data_as(P), n); n -= 1
            else:                                  # a selection round: some ranges shrink in place, then the heap is rebuilt
                for slot in rng.integers(0, n, int(rng.integers(0, 4))):
                    size = (int(a[slot]) >> 32) & 0xFFFFF
                    size = int(rng.integers(0, size + 1))
                    a[slot] = b[slot] = np.uint64((int(a[slot]) & ~(0xFFFFF << 32)) | (size << 32))
                O.hit_deque_make(a, n); ref.ref_hit_deque_make(b.ctypes.data_as(P), n)
            assert (a[:n] == b[:n]).all() and ref.ref_hit_deque_is_heap(b.ctypes.data_as(P), n)
    # and over arbitrary arrays
    for trial in range(3000):
        n = int(rng.integers(0, 64)); maxsz = int(rng.choice([2, 3, 8, 1000]))
        a = ((rng.integers(0, maxsz, 64).astype(np.uint64) << np.uint64(32)) | rng.integers(0, 1 << 32, 64).astype(np.uint64))
        b = a.copy()
        O.hit_deque_make(a, n); ref.ref_hit_deque_make(b.ctypes.data_as(P), n)
        assert (a == b).all() and ref.ref_hit_deque_is_heap(b.ctypes.data_as(P), n)


def test_hit_deque_order_properties():
    """top() is a smallest range, slot 0 a largest; popping tops drains in non-decreasing size order."""
    push, pop_bottom, pop_top = O.hit_deque_ops()
    rng = np.random.default_rng(5)
    for _ in range(200):
        n_in = int(rng.integers(1, 60))
        sz = rng.integers(1, 30, n_in)
        a = np.zeros(64, np.uint64); n = 0
        for s in sz:
            a[n] = np.uint64(int(s) << 32 | int(rng.integers(0, 1 << 32))); n += 1; push(a, n)
        size_of = lambda w: (int(w) >> 32) & 0xFFFFF
        assert size_of(a[0]) == sz.max()
        drained = []
        while n:
            top = 0 if n == 1 else 1
            drained.append(size_of(a[top]))
            pop_top(a, n); n -= 1
        assert drained == sorted(sz.tolist())


def test_sum_tree():
    """setup sums level by level; set() keeps every ancestor equal to the sum of its children; sample() inverts the
    cumulative distribution (checked in float64 away from bin edges) and never lands on a zero-probability leaf."""
    rng = np.random.default_rng(3)
    for size in (1, 2, 3, 4, 5, 7, 8, 9, 16, 33, 100):
        nodes = O.sum_tree_node_count(size)
        padded = (nodes + 1) // 2
        assert padded >= size and padded & (padded - 1) == 0 and (padded == 1 or padded // 2 < size)
        cells = np.zeros(nodes, np.float32)
        cells[:size] = (1.0 / rng.integers(1, 40, size).astype(np.float32) ** 2).astype(np.float32)
        cells[rng.integers(0, size)] = 0.0
        O.sum_tree_setup(cells, size)

        def check_tree():
            src, n = 0, padded
            while n >= 2:
                for i in range(n // 2):
                    assert cells[src + n + i] == np.float32(cells[src + 2 * i] + cells[src + 2 * i + 1])
                src += n; n //= 2
        check_tree()
        for _ in range(5):
            O.sum_tree_set(cells, size, int(rng.integers(0, size)), float(np.float32(rng.random())) if rng.random() < 0.7 else 0.0)
            check_tree()
        total = float(cells[nodes - 1])
        if total <= 0:
            continue
        cdf = np.cumsum(cells[:size].astype(np.float64)) / cells[:size].astype(np.float64).sum()
        for v in rng.random(300):
            k = O.sum_tree_sample(cells, size, float(np.float32(v)))
            assert 0 <= k < size and cells[k] > 0
            lo = cdf[k - 1] if k else 0.0
            assert lo - 1e-4 <= np.float32(v) <= cdf[k] + 1e-4
        assert cells[O.sum_tree_sample(cells, size, 0.0)] > 0 and cells[O.sum_tree_sample(cells, size, 1.0)] > 0


def test_sum_tree_reference_literals():
    """The known answers the reference's own test holds for SumTree (nvbio-test/sum_tree_test.cpp:41-193: int32 trees of 128 and 80
    leaves, sample() at 0, 0.5 and 1, leaves removed with add() / set()): the oracle's restatement of setup / set / sample
    (sum_tree_inl.h:38-178) must give them.  sample() casts the cells to float before every operation, and these cell values are small
    integers, so a float tree holds the same numbers exactly.  (add(i, d) is set(i, cell + d) on exact values.)"""
    def tree(values):
        size = len(values)
        cells = np.zeros(O.sum_tree_node_count(size), np.float32)
        cells[:size] = np.asarray(values, np.float32)
        O.sum_tree_setup(cells, size)
        return cells, size
    s3 = lambda cells, size: tuple(O.sum_tree_sample(cells, size, v) for v in (0.0, 0.5, 1.0))
    n = 128
    assert O.sum_tree_node_count(n) == 255
    assert s3(*tree(np.arange(n))) == (1, 90, 127)                                  # test 1 (:50-62)
    assert s3(*tree([0 if i < n // 2 else 1 for i in range(n)])) == (64, 96, 127)    # test 2 (:66-79)
    assert s3(*tree([1 if i < n // 2 else 0 for i in range(n)])) == (0, 32, 63)      # test 3 (:82-95)
    n = 80
    assert (O.sum_tree_node_count(n) + 1) // 2 == 128                                # padded_size() (:106-110)
    assert s3(*tree([0 if i < n // 2 else 1 for i in range(n)])) == (40, 60, 79)     # test 4 (:113-126)
    cells, size = tree([1 if i < n // 2 else 0 for i in range(n)])
    assert s3(cells, size) == (0, 20, 39)                                            # test 5 (:129-146)
    O.sum_tree_set(cells, size, 39, float(cells[39]) - 1.0)                          # add(39, -1)
    assert O.sum_tree_sample(cells, size, 1.0) == 38                                 # test 6 (:149-157)
    O.sum_tree_set(cells, size, 38, float(cells[38]) - 1.0)                          # add(38, -1)
    assert O.sum_tree_sample(cells, size, 1.0) == 37                                 # test 7 (:160-167)
    O.sum_tree_set(cells, size, 38, float(cells[38]) + 1.0)                          # add(38, 1)
    O.sum_tree_set(cells, size, 38, 0.0)                                             # set(38, 0)
    assert O.sum_tree_sample(cells, size, 1.0) == 37                                 # test 8 (:170-180)
    for i in range(10):                                                              # test 9 (:183-193)
        O.sum_tree_set(cells, size, i, 0.0)
        assert O.sum_tree_sample(cells, size, 0.0) == i + 1


def _random_deques(rng, n_reads, stride, max_size=40):
    push, _, _ = O.hit_deque_ops()
    hits = np.zeros((n_reads, stride), np.uint64)
    counts = np.zeros(n_reads, np.uint32)
    for r in range(n_reads):
        k = int(rng.integers(0, stride + 1)) if r % 7 else 0
        for j in range(k):
            size = int(rng.integers(1, max_size))
            begin = int(rng.integers(0, 1 << 30))
            flags = (int(rng.integers(0, 1 << 10)) << 20) | (int(rng.integers(0, 4)) << 30)
            hits[r, j] = np.uint64(((size | flags) << 32) | begin)
            push(hits[r], j + 1)
        counts[r] = k
    return hits, counts


@pytest.mark.parametrize("randomized", [False, True])
@pytest.mark.parametrize("n_multi", [1, 3, 16])
def test_select_hands_out_every_row_once(randomized, n_multi):
    """Run selection rounds (never giving up) to exhaustion: every SA row of every hit comes out exactly once, tagged with
    its hit's flags; deterministic selection drains ranges in non-decreasing size order."""
    rng = np.random.default_rng(11 + n_multi)
    n, stride = 120, 12
    hits, counts = _random_deques(rng, n, stride)
    h0, c0 = hits.copy(), counts.copy()
    names, idx = O.pack_names(["read/%d" % i for i in range(n)])
    probs, trys, rseeds = O.select_init(hits, counts, names, idx, 15, randomized, 0)
    assert (trys == 15).all()
    want = {}
    for r in range(n):
        for j in range(int(c0[r])):
            w = int(h0[r, j]); hi = w >> 32
            seed = ((hi >> 20) & 0x3FF) | (((hi >> 31) & 1) << 12) | (((hi >> 30) & 1) << 13)
            for k in range(hi & 0xFFFFF):
                want.setdefault(r, []).append(((w & 0xFFFFFFFF) + k, seed))
    got = {}
    active = np.arange(n, dtype=np.uint32)
    rounds = 0
    while active.size:
        prev = active
        active, hb, rid, loc, seed = O.select(randomized, n_multi, active, hits, counts, probs, rseeds, trys)
        assert np.isin(active & 0x7FFFFFFF, prev & 0x7FFFFFFF).all() and (np.diff(hb.astype(np.int64)) >= 1).all() and (np.diff(hb.astype(np.int64)) <= n_multi).all()
        for t in range(active.size):
            for i in range(int(hb[t]), int(hb[t + 1])):
                assert rid[i] == active[t] & 0x7FFFFFFF
                got.setdefault(int(rid[i]), []).append((int(loc[i]), int(seed[i]) & 0x3FFF))
        rounds += 1
        assert rounds < 5000
    # randomized selection may leave rows behind only through its 10-draw bail-out; with these sizes it does not
    for r in want:
        assert sorted(got.get(r, [])) == sorted(want[r]), r
    assert set(got) == set(want)
    if not randomized:
        for r in want:
            sizes = {}
            for j in range(int(c0[r])):
                hi = int(h0[r, j]) >> 32
                sizes[int(h0[r, j]) & 0xFFFFFFFF] = hi & 0xFFFFF
            # the first row of each range appears in non-decreasing range-size order
            firsts = [sizes[l] for (l, s) in got[r] if l in sizes]
            assert firsts == sorted(firsts)


def test_select_respects_try_counters_and_top_flag():
    rng = np.random.default_rng(2)
    hits, counts = _random_deques(rng, 50, 8)
    probs, trys, rseeds = O.select_init(hits, counts, None, None, 3, False, 1)
    trys[::2] = 0
    active = np.arange(50, dtype=np.uint32) | np.uint32(1 << 31)
    out, hb, rid, loc, seed = O.select(False, 1, active, hits, counts, probs, rseeds, trys)
    kept = out & 0x7FFFFFFF
    assert (kept % 2 == 1).all() and set(kept.tolist()) == {r for r in range(1, 50, 2) if counts[r] or r in kept}
    assert ((out >> 31) == 1).all() and ((seed >> 14) & 1).all()       # first round: still on the top range


def test_reduce_best_approx_counters():
    """An improving score refills the tries; a non-improving one past min_ext burns one; the deque is erased when they run out."""
    read_len = np.full(4, 100, np.uint32)
    best = O.init_alignments(read_len, (0, -0.6, -0.6))
    trys = np.array([2, 2, 1, 5], np.uint32); counts = np.array([9, 9, 9, 9], np.uint32)
    active = np.arange(4, dtype=np.uint32)
    hb = np.array([0, 1, 2, 3, 4], np.uint64)
    score = np.array([-10, -500, -500, -(1 << 30)], np.int32)          # read 0 improves; 1..3 fall below the threshold
    loc = np.array([1000, 2000, 3000, 4000], np.uint32); seed = np.zeros(4, np.uint32)
    O.score_reduce_best_approx(best, active, hb, score, loc, seed, read_len, -(1 << 16), trys, counts, 40, 30, 400, 15)
    assert trys.tolist() == [15, 1, 0, 4] and counts.tolist() == [9, 9, 0, 9]
    assert (best[0, 0] >> np.uint64(32)) == 1000 and (best[0, 1] >> np.uint64(32)) == 0xFFFFFFFF
    # below min_ext nothing is burnt; a top-seed hit never burns; reaching max_ext erases
    trys[:] = 3; counts[:] = 9
    seed2 = np.array([0, 1 << 14, 0, 0], np.uint32)
    O.score_reduce_best_approx(best, active, hb, np.full(4, -900, np.int32), loc + 7, seed2, read_len, -(1 << 16), trys, counts, 10, 30, 400, 15)
    assert trys.tolist() == [3, 3, 3, 3] and counts.tolist() == [9, 9, 9, 9]
    O.score_reduce_best_approx(best, active, hb, np.full(4, -900, np.int32), loc + 9, seed2, read_len, -(1 << 16), trys, counts, 400, 30, 400, 15)
    assert trys.tolist() == [2, 3, 2, 2] and counts.tolist() == [0, 0, 0, 0]


def test_sam_md_string_rendering():
    """SamOutput::generate_md_string over nvbio's byte-coded MDS (output_sam.cpp:233-314), as restated in nvbio_amd.io.sam_md_string"""
    from nvbio_amd.io import sam_md_string
    def mds(*tokens):
        body = [b for t in tokens for b in t]
        n = len(body) + 2
        return np.array([n & 0xFF, n >> 8] + body, dtype=np.uint8)
    assert sam_md_string(mds((0, 60))) == ("60", 0, 0, 0)
    assert sam_md_string(mds((0, 10), (1, 3), (0, 49))) == ("10T49", 1, 0, 0)                      # the symbol stored with a MISMATCH is printed
    assert sam_md_string(mds((0, 30), (3, 2, 1, 2), (0, 30))) == ("30^CG030", 0, 1, 1)               # deletion: '^' + symbols + '0'
    assert sam_md_string(mds((2, 3, 0, 0, 0), (0, 57))) == ("57", 0, 1, 2)                           # insertion (or soft clip): counted, not printed
    assert sam_md_string(mds((1, 4), (0, 5)))[0] == "N5"


def test_default_hit_rows_hold_every_range_a_read_can_yield():
    """Params.resolved_hits_stride (nvbio_amd/aligner.py; Params::resolved_hits_stride in include/nvbio_hip/aligner.h): when the caller names no
    row size, a read's deque gets the reference's capacity min(max_hits, 128) -- or, under exact seeding, the 16 / 32 slots that hold one range per
    seed and strand of the longest read, counted the way the mapping kernel walks the seeds (mapping_inl.h:236-262: pos += seed_freq(len) while
    the seed fits)."""
    from nvbio_amd import aligner as A, mapping as M

    def ranges_at_most(p, length):
        if length < p.min_read_len:
            return 0
        f = max(M.simple_func(*p.seed_freq, length), 0)
        if f == 0:
            return 0
        sl, n, pos = min(p.seed_len, length), 0, 0
        while pos + sl <= length:
            n += 1; pos += f
        return 2 * n

    assert A.Params().resolved_hits_stride(100) == 16                    # 7 seeds x 2 strands
    assert A.Params(local=True).resolved_hits_stride(150) == 32          # 14 seeds x 2 strands
    assert A.Params(allow_sub=1).resolved_hits_stride(100) == 100        # one-mismatch seeding: several ranges per seed, the reference's capacity
    assert A.Params(hits_stride=64).resolved_hits_stride(100) == 64      # a named size is taken as it is
    assert A.Params(max_hits=8).resolved_hits_stride(100) == 8           # never above the reference's own cap
    for kw in ({}, {"local": True}, {"seed_len": 12, "seed_freq": (M.LINEAR_FUNC if hasattr(M, "LINEAR_FUNC") else 0, 3.0, 0.0)}, {"min_read_len": 30}):
        p = A.Params(**kw)
        for longest in (20, 50, 100, 151, 250, 400):
            rows = p.resolved_hits_stride(longest)
            need = max(ranges_at_most(p, length) for length in range(1, longest + 1))
            assert rows >= min(need, min(p.max_hits, 128)), (kw, longest, rows, need)
            assert rows % 2 == 0 or rows == min(p.max_hits, 128)
