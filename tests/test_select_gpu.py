"""nvBowtie's hit-selection stage, the per-round stages of its best-approx loop and the composed single-end driver
through the C-ABI vs the oracle (reference: nvBowtie/bowtie2/cuda/select_inl.h, select.cu, locate_inl.h,
score_best_inl.h, reduce.h, aligner_best_approx.h)."""
import os
import sys

import numpy as np
import pytest
import torch

import nvbio_amd as nvb
from nvbio_amd import select as S, aligner as A, workloads as W, pipeline as P
from nvbio_amd._lib import lib, check
from oracle import pyoracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)
import oracle_driver as OD  # noqa: E402
from test_select_oracle import _random_deques  # noqa: E402

pytestmark = pytest.mark.gpu


def dev_i32(a, cuda):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).to(cuda)


def test_device_hit_deque_replays_reference_vectors(cuda):
    """The programs recorded from the reference's interval_heap.h, through the device deque: same array after every operation."""
    import ctypes as C
    G = np.load(os.path.join(HERE, "golden", "hit_deque_vectors.npz"))
    cs, sizes = G["case_start"], G["sizes"].astype(np.uint64)
    per_case = np.add.reduceat(sizes, cs[:-1].astype(np.int64))
    state_start = np.zeros(cs.size - 1, np.uint64); state_start[1:] = np.cumsum(per_case)[:-1]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8)).to(cuda)
    d_cs, d_ops, d_vals, d_caps, d_ss = t(cs), t(G["ops"]), t(G["vals"]), t(G["caps"]), t(state_start)
    scratch = torch.zeros((cs.size - 1) * 64, dtype=torch.int64, device=cuda)
    out = torch.zeros(G["states"].size, dtype=torch.int64, device=cuda)
    vp = lambda x: C.c_void_p(x.data_ptr())
    check(lib().nvbio_hip_hit_deque_replay(cs.size - 1, vp(d_cs), vp(d_ops), vp(d_vals), vp(d_caps), vp(d_ss), vp(scratch), 64, vp(out), None), "replay")
    torch.cuda.synchronize()
    assert (out.cpu().numpy().view(np.uint64) == G["states"]).all()


def device_hit_deque_replay(cuda, ops, vals, sizes):
    """One program through the device deque: the array after every operation, concatenated (sizes = the expected sizes)."""
    import ctypes as C
    n_ops = int(ops.size)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8)).to(cuda)
    stride = int(sizes.max()) + 2
    d_cs, d_ops, d_vals = t(np.array([0, n_ops], np.uint32)), t(ops.astype(np.uint8)), t(vals.astype(np.uint64))
    d_caps, d_ss = t(np.full(n_ops, stride - 1, np.uint32)), t(np.zeros(1, np.uint64))
    scratch = torch.zeros(stride, dtype=torch.int64, device=cuda)
    out = torch.zeros(int(sizes.astype(np.int64).sum()), dtype=torch.int64, device=cuda)
    vp = lambda x: C.c_void_p(x.data_ptr())
    check(lib().nvbio_hip_hit_deque_replay(1, vp(d_cs), vp(d_ops), vp(d_vals), vp(d_caps), vp(d_ss), vp(scratch), stride, vp(out), None), "replay")
    torch.cuda.synchronize()
    return out.cpu().numpy().view(np.uint64)


@pytest.mark.parametrize("randomized", [False, True])
@pytest.mark.parametrize("n_multi", [1, 4, 32])
@pytest.mark.parametrize("top_seed", [0, 1])
@pytest.mark.parametrize("stride", [12, 16, 20, 40])
def test_select_rounds_match_oracle(cuda, randomized, n_multi, top_seed, stride):
    """select_init + repeated select rounds on random deques: queues, selected hits, and the whole mutable state
    (arena, counts, probability trees bit for bit, LCG states) after every round; some reads give up on the way.
    select_init builds the trees with 4 lanes per read (stride <= 16), 8 lanes per read (<= 32) or one lane per read (wider rows)."""
    _select_rounds(cuda, randomized, n_multi, top_seed, stride)


@pytest.mark.parametrize("n_multi", [1, 4])
@pytest.mark.parametrize("stride", [16, 32, 40])
def test_select_rounds_over_wide_ranges_match_oracle(cuda, n_multi, stride):
    """The same rounds over deques as a repeat-rich 3 Gbp index fills them: ranges from one row to 2^20 - 1 rows in one deque, i.e. leaves
    twelve orders of magnitude apart -- where the float arithmetic of sample() decides among the wide ranges only after the narrow ones ran out."""
    _select_rounds(cuda, True, n_multi, 0, stride, wide=True, max_rounds=60)


@pytest.mark.parametrize("lanes", ["1", "2", "4"])
@pytest.mark.parametrize("n_multi", [1, 4])
@pytest.mark.parametrize("stride", [8, 16, 32])
def test_select_lane_forms_match_oracle(cuda, monkeypatch, lanes, n_multi, stride):
    """NVBIO_HIP_SELECT_LANES=1: select_init and select with one lane per read; =4: the randomized select with a read's row and tree
    in 4 / 8 lanes, four leaves each; =2: one lane per read WITHOUT the table of make() arrangements (every row rebuilds its heap exchange by
    exchange; the default looks the arrangement of rows with at most two distinct range sizes up).  Same picks and state as the oracle every way."""
    nvb.set_test_switch("NVBIO_HIP_SELECT_LANES", lanes)
    _select_rounds(cuda, True, n_multi, 1, stride)


def _two_size_deques(rng, n_reads, stride, sizes):
    """deques as the mappers push them, every hit's range size drawn from `sizes` (one or two values): the rows the make() table serves"""
    push, _, _ = O.hit_deque_ops()
    hits = np.zeros((n_reads, stride), np.uint64)
    counts = np.zeros(n_reads, np.uint32)
    for r in range(n_reads):
        k = int(rng.integers(0, min(stride, 16) + 1)) if r % 9 else 0
        for j in range(k):
            size = int(sizes[int(rng.integers(0, len(sizes)))])
            begin = int(rng.integers(0, 1 << 30))
            flags = (int(rng.integers(0, 1 << 10)) << 20) | (int(rng.integers(0, 4)) << 30)
            hits[r, j] = np.uint64(((size | flags) << 32) | begin)
            push(hits[r], j + 1)
        counts[r] = k
    return hits, counts


@pytest.mark.parametrize("sizes", [(1,), (1, 2), (1, 3), (2, 900_000), (7, 7)], ids=["unique_seeds", "one_two", "one_three", "far_apart", "all_equal"])
@pytest.mark.parametrize("n_multi", [1, 4])
@pytest.mark.parametrize("stride", [16, 32])
def test_select_rounds_on_rows_the_make_table_serves(cuda, sizes, n_multi, stride):
    """Rows whose range sizes take one or two values -- a read's seeds unique in the genome, then some used up -- have their heap arrangement looked up
    (select_make_table_kernel) instead of rebuilt exchange by exchange: selection rounds to exhaustion against the oracle's replay of the
    reference's interval heap, every round's picks and every hit row slot for slot (rows drift out of the table's reach as ranges shrink to a
    third value and back in as they run out; top_seed on, so the first pick is the row's slot 0)."""
    _select_rounds(cuda, True, n_multi, 1, stride, two_sizes=sizes)
    _select_rounds(cuda, True, n_multi, 0, stride, two_sizes=sizes)


def _leaves_match(g_probs, e_probs, counts):
    """the library keeps a tree's LEAVES in memory (leaf i < the read's hit count), the oracle -- like the reference -- all its nodes:
    compare leaf for leaf, bit for bit"""
    g, e = g_probs.cpu().numpy().view(np.uint32), e_probs.view(np.uint32)
    w = min(g.shape[1], e.shape[1])
    mask = np.arange(w)[None, :] < np.asarray(counts)[:, None]
    return bool((g[:, :w][mask] == e[:, :w][mask]).all())


def _wide_deques(rng, n_reads, stride):
    """deques the way a repeat-rich 3 Gbp index fills them: SA ranges from 1 row to the 20 bits a SeedHit keeps (leaves 1 / delta^2 from 1 down to 1e-12)"""
    push, _, _ = O.hit_deque_ops()
    hits = np.zeros((n_reads, stride), np.uint64)
    counts = np.zeros(n_reads, np.uint32)
    for r in range(n_reads):
        k = int(rng.integers(1, min(stride, 14) + 1))
        for j in range(k):
            kind = int(rng.integers(0, 4))
            size = int(rng.integers(1, 40)) if kind == 0 else int(rng.integers(300, 5000)) if kind == 1 else int(rng.integers(100_000, 1 << 20))
            begin = int(rng.integers(0, 3_000_000_000 - (1 << 21)))
            flags = (int(rng.integers(0, 1 << 10)) << 20) | (int(rng.integers(0, 4)) << 30)
            hits[r, j] = np.uint64(((size | flags) << 32) | begin)
            push(hits[r], j + 1)
        counts[r] = k
    return hits, counts


def _select_rounds(cuda, randomized, n_multi, top_seed, stride, wide=False, max_rounds=2000, two_sizes=None):
    rng = np.random.default_rng(40 + n_multi + 2 * top_seed + 100 * stride)
    n = 3000
    if two_sizes is not None:
        hits, counts = _two_size_deques(rng, n, stride, two_sizes)
    else:
        hits, counts = _wide_deques(rng, n, stride) if wide else _random_deques(rng, n, stride, max_size=min(stride, 16))
    names = ["r%d/%d" % (i, i * 7919 % 13) for i in range(n)]
    arena, idx = O.pack_names(names)
    e_probs, e_trys, e_rseeds = O.select_init(hits, counts, arena, idx, 15, randomized, top_seed)
    d_hits, d_counts = torch.from_numpy(hits.view(np.int64).copy()).to(cuda), dev_i32(counts, cuda)
    st = S.SelectState(d_hits, d_counts, S.pack_names(names, cuda), 15, randomized, top_seed)
    torch.cuda.synchronize()
    assert (st.trys.cpu().numpy().view(np.uint32) == e_trys).all()
    if randomized:
        assert (st.rseeds.cpu().numpy().view(np.uint32) == e_rseeds).all()
        assert _leaves_match(st.probs, e_probs, counts)
    e_active = (np.arange(n, dtype=np.uint32) | np.uint32(top_seed << 31))[rng.permutation(n)]
    d_active = dev_i32(e_active, cuda)
    rounds = total = 0
    while e_active.size:
        if rounds % 3 == 2:                                    # some reads run out of tries between rounds
            give_up = rng.integers(0, n, 40)
            e_trys[give_up] = 0
            st.trys[torch.from_numpy(give_up).to(cuda)] = 0
        e_active, e_hb, e_rid, e_loc, e_seed = O.select(randomized, n_multi, e_active, hits, counts, e_probs, e_rseeds, e_trys)
        d_active, d_hb, d_rid, d_loc, d_seed = S.select(st, d_active, n_multi)
        torch.cuda.synchronize()
        u = lambda x: x.cpu().numpy().view(np.uint32)
        assert (u(d_active) == e_active).all() and (d_hb.cpu().numpy().view(np.uint64) == e_hb).all()
        assert (u(d_rid) == e_rid).all() and (u(d_loc) == e_loc).all() and (u(d_seed) == e_seed).all()
        assert (st.hits.cpu().numpy().view(np.uint64) == hits).all() and (u(st.counts) == counts).all()
        if randomized:
            assert (u(st.rseeds) == e_rseeds).all() and _leaves_match(st.probs, e_probs, counts)
        rounds += 1; total += e_loc.size
        if (wide or two_sizes is not None) and rounds >= max_rounds:
            break
        assert rounds < 2000
    assert rounds > 3 and total > n


@pytest.mark.parametrize("stride", [16, 24, 40])
def test_select_init_of_a_queue_only(cuda, stride):
    """nvbio_hip_select_init_queued: the reads of a queue get exactly what the whole-batch select_init gives them (tries, LCG seed from
    the name, probability tree), every other read's state is left as it was."""
    import ctypes as C
    from nvbio_amd._lib import check
    rng = np.random.default_rng(4100 + stride)
    n = 5000
    hits, counts = _random_deques(rng, n, stride, max_size=min(stride, 16))
    names = ["q%d/%d" % (i * 31 % 977, i) for i in range(n)]
    arena, idx = O.pack_names(names)
    e_probs, e_trys, e_rseeds = O.select_init(hits, counts, arena, idx, 15, True, 1)
    queue = np.sort(rng.choice(n, 700, replace=False)).astype(np.uint32)
    d_hits, d_counts = torch.from_numpy(hits.view(np.int64).copy()).to(cuda), dev_i32(counts, cuda)
    d_arena, d_idx = S.pack_names(names, cuda)
    ps = S.sum_tree_node_count(stride)
    probs = torch.full((n, ps), -7.0, dtype=torch.float32, device=cuda)
    trys = torch.full((n,), 99, dtype=torch.int32, device=cuda)
    rseeds = torch.full((n,), 12345, dtype=torch.int32, device=cuda)
    d_queue = dev_i32(queue, cuda)
    vp = lambda x: C.c_void_p(x.data_ptr())
    check(lib().nvbio_hip_select_init_queued(queue.size, vp(d_queue), vp(d_arena), vp(d_idx), vp(d_hits), stride, vp(d_counts), vp(probs), ps,
                                             vp(trys), vp(rseeds), 15, 1, 1, None), "nvbio_hip_select_init_queued")
    torch.cuda.synchronize()
    inq = np.zeros(n, bool); inq[queue] = True
    g_trys, g_rseeds, g_probs = trys.cpu().numpy().view(np.uint32), rseeds.cpu().numpy().view(np.uint32), probs.cpu().numpy()
    assert (g_trys[inq] == e_trys[inq]).all() and (g_trys[~inq] == 99).all()
    assert (g_rseeds[inq] == e_rseeds[inq]).all() and (g_rseeds[~inq] == 12345).all()
    assert (g_probs[~inq] == -7.0).all()
    # a queued read's tree: its leaves as the oracle builds them; empty deques are not touched
    has = inq & (counts > 0)
    for r in np.nonzero(has)[0][:400]:
        m = int(counts[r])
        assert (g_probs[r, :m].view(np.uint32) == e_probs[r, :m].view(np.uint32)).all(), (r, m)
    assert (g_probs[inq & (counts == 0)] == -7.0).all()


def _small_index(rng, n_genome=1 << 17):
    text = rng.integers(0, 4, n_genome, dtype=np.uint8)
    text[7000:7800] = np.tile(np.array([0, 1, 2], dtype=np.uint8), 267)[:800]
    text[30000:30400] = text[90000:90400]                       # a 400 bp duplication: two equally good placements
    return text


def test_locate_setup_reduce_match_oracle(cuda):
    """locate_hits (both index directions), score_best_setup, score_reduce_best_approx on random hits."""
    rng = np.random.default_rng(9)
    text = _small_index(rng)
    host, rhost = O.FMIndex(text), O.FMIndex(text[::-1].copy())
    fmi, rfmi = nvb.FMIndexDevice.from_host(host, cuda), nvb.FMIndexDevice.from_host(rhost, cuda)
    n_reads, L, n_hits = 500, 100, 24000                        # enough random rows that some read starts wrap below zero under any seed
    rows = rng.integers(0, text.size + 1, n_hits).astype(np.uint32)
    seed = (rng.integers(0, 80, n_hits) | (rng.integers(0, 2, n_hits) << 12) | (rng.integers(0, 2, n_hits) << 13) | (rng.integers(0, 2, n_hits) << 14)).astype(np.uint32)
    e_loc = O.locate_hits(host, rhost, rows, seed)
    d_loc, d_seed = dev_i32(rows, cuda), dev_i32(seed, cuda)
    S.locate_hits(fmi, rfmi, d_loc, d_seed)
    torch.cuda.synchronize()
    assert (d_loc.cpu().numpy().view(np.uint32) == e_loc).all()
    assert (e_loc > 0xFFFF0000).any()                           # some read starts wrap below zero

    rid = np.sort(rng.integers(0, n_reads, n_hits)).astype(np.uint32)
    read_len = np.full(n_reads, L, np.uint32)
    best = O.init_alignments(read_len, (0, -0.6, -0.6))
    best[1, ::3] = best[1, ::3] & ~np.uint64(0x3FFFF) | np.uint64((20 << 1) | 1)      # some second-best scores: -20
    e_tb, e_tl, e_ms = O.score_best_setup(rid, e_loc, read_len, 31, text.size, best, -(1 << 16))
    d_best = torch.from_numpy(best.view(np.int64).copy()).to(cuda)
    pb, pl, tb, tl, ms = S.score_best_setup(dev_i32(rid, cuda), d_loc, d_seed, d_best, 31, text.size, -(1 << 16), fixed_read_len=L, rc_offset=n_reads * L)
    torch.cuda.synchronize()
    assert (tb.cpu().numpy().view(np.uint64) == e_tb).all() and (tl.cpu().numpy().view(np.uint32) == e_tl).all() and (ms.cpu().numpy() == e_ms).all()
    assert (pb.cpu().numpy() == rid.astype(np.int64) * L + ((seed >> 13) & 1).astype(np.int64) * n_reads * L).all()
    assert (e_tl == 0).any() and (e_tl == L + 31).any() and (e_ms == -20).any()

    # reduce with the give-up counters: groups of consecutive hits per active read
    active = np.unique(rid).astype(np.uint32)
    hb = np.searchsorted(rid, np.append(active, n_reads)).astype(np.uint64)
    for n_ext in (0, 25, 395):
        score = rng.integers(-80, 1, n_hits).astype(np.int32)
        score[rng.random(n_hits) < 0.2] = -(1 << 30)
        loc2 = (e_loc // 50 * 50).astype(np.uint32)             # collisions with recorded locations
        trys = rng.integers(0, 4, n_reads).astype(np.uint32); counts = rng.integers(0, 9, n_reads).astype(np.uint32)
        e_best, e_trys, e_counts = best.copy(), trys.copy(), counts.copy()
        O.score_reduce_best_approx(e_best, active, hb, score, loc2, seed, read_len, -(1 << 16), e_trys, e_counts, n_ext, 30, 400, 15)
        st = S.SelectState.__new__(S.SelectState)
        st.trys, st.counts = dev_i32(trys, cuda), dev_i32(counts, cuda)
        g_best = torch.from_numpy(best.view(np.int64).copy()).to(cuda)
        S.score_reduce_best_approx(g_best, st, dev_i32(active, cuda), torch.from_numpy(hb.view(np.int64)).to(cuda), torch.from_numpy(score).to(cuda),
                                   dev_i32(loc2, cuda), d_seed, -(1 << 16), n_ext, 30, 400, 15, fixed_read_len=L)
        torch.cuda.synchronize()
        assert (g_best.cpu().numpy().view(np.uint64) == e_best).all()
        assert (st.trys.cpu().numpy().view(np.uint32) == e_trys).all() and (st.counts.cpu().numpy().view(np.uint32) == e_counts).all()
        assert n_ext == 0 or ((e_counts != counts).any() and (e_trys != trys).any())
        best = e_best


def _reads(rng, text, n, L):
    sym = np.zeros((n, L), np.uint8)
    pos = rng.integers(0, text.size - L, n)
    for i in range(n):
        r = text[pos[i]:pos[i] + L].copy()
        k = [0, 1, 3, 6, 12][i % 5]
        for j in rng.integers(0, L, k):
            r[j] = (r[j] + 1 + rng.integers(0, 3)) & 3
        if i % 11 == 0:
            d = int(rng.integers(10, L - 10)); r = np.concatenate([r[:d], r[d + 2:], rng.integers(0, 4, 2, dtype=np.uint8)])      # a 2-base deletion
        if i % 13 == 0:
            r[int(rng.integers(0, L))] = 4
        if i % 2:
            r = np.where(r > 3, r, 3 - r)[::-1]
        if i % 19 == 0:
            r = rng.integers(0, 4, L, dtype=np.uint8)             # unalignable
        sym[i] = r
    return sym, pos


CONFIGS = {
    "default": dict(),
    "no_rand": dict(randomized=False),
    "top_seed": dict(top_seed=1),
    "one_hit_rounds": dict(batch_size=1000),                     # n_reads > BATCH_SIZE/2: one hit per read per round
    "multi_rounds": dict(batch_size=6000),                       # 5 hits per read per round, more as the queue drains
    "low_effort": dict(max_effort=2, max_effort_init=2, min_ext=3, max_ext=20),
    "one_mismatch_seeds": dict(allow_sub=1, seed_len=20, max_hits=30),
    "subseed": dict(allow_sub=1, subseed_len=12, max_reseed=1),
    "local": dict(local=True, seed_len=20, seed_freq=(2, 1.0, 0.75)),
    "read_quals": dict(),                                        # per-base Phred qualities instead of a constant
    "finish": dict(),                                            # + finish_alignment: MD strings, edit distances, final scores
}


@pytest.mark.parametrize("config", sorted(CONFIGS))
def test_best_approx_driver_matches_oracle(cuda, config):
    """Aligner::best_approx end to end -- seeding passes, selection rounds, locate, extension, reduction with give-up
    counters, re-seeding, MAPQ, traceback -- vs the independent numpy driver over the oracle: identical best / second-best
    alignments, MAPQs, CIGARs and per-pass queue sizes; and reads land where they were sampled."""
    rng = np.random.default_rng(123)
    text = _small_index(rng)
    host, rhost = O.FMIndex(text), O.FMIndex(text[::-1].copy())
    fmi, rfmi = nvb.FMIndexDevice.from_host(host, cuda), nvb.FMIndexDevice.from_host(rhost, cuda)
    n, L = 1200, 100
    sym, pos = _reads(rng, text, n, L)
    names = ["sim.%d" % i for i in range(n)]
    params = A.Params(**CONFIGS[config])
    scheme = nvb.SmithWatermanScoringScheme.local() if params.local else nvb.SmithWatermanScoringScheme()
    gw = W._pack_chunked(torch.from_numpy(text), 2, True)
    rq = rng.integers(2, 42, (n, L)).astype(np.uint8) if config == "read_quals" else None
    fin = config == "finish"
    e = OD.best_approx(host, rhost, sym, gw.numpy().view(np.uint32), text.size, params, scheme, names, 1 if params.local else 2, read_quals=rq, finish=fin)
    r = A.best_approx(fmi, rfmi, torch.from_numpy(sym).to(cuda), gw.to(cuda), text.size, params, scheme, names, cigar_stride=64,
                      quals=torch.from_numpy(rq) if rq is not None else None, finish=fin)
    if fin:
        assert (r["mds_len"].cpu().numpy().view(np.uint32) == e["mds_len"]).all()
        m = np.arange(256)[None, :] < np.minimum(e["mds_len"], 256)[:, None]
        assert ((r["mds"].cpu().numpy() == e["mds"]) | ~m).all()
        assert (r["best_scored"].cpu().numpy().view(np.uint64) == e["best_scored"]).all()
        # the finished words: edit distance = mismatches + gap symbols of the CIGAR / MD, score <= 0 end-to-end, position = the window's begin
        fw = e["best"][0][e["aligned_ids"]]
        ed = ((fw >> np.uint64(18)) & np.uint64(0x3FF)).astype(np.int64)
        assert ed.max() > 3 and (ed >= 0).all()
        k = int(e["aligned_ids"][int(np.argmax(ed))])
        assert len(O.mds_to_string(e["mds"][k])) > 0
    torch.cuda.synchronize()
    assert r["stats"] == e["stats"], (r["stats"], e["stats"])
    assert (r["best"].cpu().numpy().view(np.uint64) == e["best"]).all()
    assert (r["mapq"].cpu().numpy() == e["mapq"]).all()
    ids = r["aligned_ids"].cpu().numpy()
    assert (ids == e["aligned_ids"]).all()
    tb = e["tb"]
    assert (r["cigar_len"].cpu().numpy()[ids].view(np.uint32) == tb["cigar_len"]).all()
    assert (r["cigar"].cpu().numpy()[ids].view(np.uint16) == tb["cigar"][: ids.size]).all()
    assert (r["tb_score"].cpu().numpy()[ids] == tb["score"]).all()
    assert (r["cigar_len"].cpu().numpy().sum() == tb["cigar_len"].sum()) and (r["source"].cpu().numpy()[~np.isin(np.arange(n), ids)] == -1).all()
    assert (r["sink"].cpu().numpy()[ids].view(np.uint32) == tb["sink"]).all() and (r["source"].cpu().numpy()[ids].view(np.uint32) == tb["source"]).all()
    # sanity of the result itself: simulated reads (not the random ones) come back at their origin
    best0 = (e["best_scored"] if fin else e["best"])[0]
    loc = (best0 >> np.uint64(32)).astype(np.int64)
    aligned = loc != 0xFFFFFFFF
    sim = np.arange(n) % 19 != 0
    ok = aligned & sim & (np.abs(loc - pos) <= 4)
    assert ok.sum() >= 0.7 * sim.sum(), (ok.sum(), sim.sum(), aligned.sum())
    assert e["stats"]["seeding_passes"] >= 2 and e["stats"]["rounds"] >= 2
    # the traceback re-derives the score the extension stage recorded
    sc = np.where(best0 & np.uint64(1), -((best0 >> np.uint64(1)) & np.uint64(0x1FFFF)).astype(np.int64), ((best0 >> np.uint64(1)) & np.uint64(0x1FFFF)).astype(np.int64))
    assert (tb["score"] == sc[ids]).all()


def _pairs(rng, text, n, L, frag=(200, 420)):
    """FR pairs: mate 1 forward at the fragment's left end, mate 2 reverse-complement at its right end; substitutions, some indels,
    some pairs with an unalignable or a far-away (discordant) mate"""
    s1 = np.zeros((n, L), np.uint8); s2 = np.zeros((n, L), np.uint8)
    pos = np.zeros(n, np.int64); flen = np.zeros(n, np.int64)
    for i in range(n):
        f = int(rng.integers(frag[0], frag[1]))
        p = int(rng.integers(0, text.size - f - 2000))
        a = text[p:p + L].copy()
        q = p + f - L if i % 23 else p + 1500                       # every 23rd pair: mate 2 far away (no concordant placement)
        b = text[q:q + L].copy()
        for r, k in ((a, [0, 1, 2, 5][i % 4]), (b, [1, 0, 4, 2][i % 4])):
            for j in rng.integers(0, L, k):
                r[j] = (r[j] + 1 + rng.integers(0, 3)) & 3
        if i % 9 == 0:
            d = int(rng.integers(20, L - 20)); b = np.concatenate([b[:d], b[d + 1:], rng.integers(0, 4, 1, dtype=np.uint8)])
        if i % 17 == 0:
            b = rng.integers(0, 4, L, dtype=np.uint8)                  # mate 2 unalignable: mate 1 must still be reported (mixed mode)
        if i % 31 == 0:
            a = rng.integers(0, 4, L, dtype=np.uint8); b = rng.integers(0, 4, L, dtype=np.uint8)
        s1[i] = a; s2[i] = np.where(b > 3, b, 3 - b)[::-1]
        pos[i] = p; flen[i] = f
    return s1, s2, pos, flen


PAIRED_CONFIGS = {
    "default": dict(),
    "no_rand_local": dict(randomized=False, local=True, seed_len=20, seed_freq=(2, 1.0, 0.75)),
    "one_hit_rounds": dict(batch_size=600),
    "multi_rounds_no_mixed": dict(batch_size=3000, pe_unpaired=False, pe_discordant=False),
    "low_effort_subseed": dict(max_effort=3, max_effort_init=3, min_ext=4, max_ext=40, allow_sub=1, subseed_len=12, max_reseed=1),
    "ff_policy": dict(pe_policy=0, max_frag_len=450),
    "finish": dict(),                                            # + finish_alignment on both slot sets (MD strings, edit distances, final scores)
}


@pytest.mark.parametrize("config", sorted(PAIRED_CONFIGS))
def test_best_approx_paired_driver_matches_oracle(cuda, config):
    """Aligner::best_approx for pairs -- per anchor mate: seeding passes, selection rounds, anchor extension with pair-aware thresholds,
    opposite-mate windows + full-matrix scoring, paired reduction with give-up counters; then discordant marking, both MAPQs and the
    anchor / opposite tracebacks -- vs the independent numpy driver over the oracle."""
    rng = np.random.default_rng(321)
    text = _small_index(rng, 1 << 17)
    host, rhost = O.FMIndex(text), O.FMIndex(text[::-1].copy())
    fmi, rfmi = nvb.FMIndexDevice.from_host(host, cuda), nvb.FMIndexDevice.from_host(rhost, cuda)
    n, L = 700, 100
    s1, s2, pos, flen = _pairs(rng, text, n, L)
    if PAIRED_CONFIGS[config].get("pe_policy") == 0:                   # FF: both mates forward
        s2 = np.where(s2 > 3, s2, 3 - s2)[:, ::-1].copy()
    names = ["pair.%d" % i for i in range(n)]
    params = A.Params(**PAIRED_CONFIGS[config])
    scheme = nvb.SmithWatermanScoringScheme.local() if params.local else nvb.SmithWatermanScoringScheme()
    gw = W._pack_chunked(torch.from_numpy(text), 2, True)
    fin = config == "finish"
    e = OD.best_approx_paired(host, rhost, s1, s2, gw.numpy().view(np.uint32), text.size, params, scheme, names, 1 if params.local else 2, finish=fin)
    r = A.best_approx_paired(fmi, rfmi, torch.from_numpy(s1).to(cuda), torch.from_numpy(s2).to(cuda), gw.to(cuda), text.size, params, scheme, names, finish=fin)
    torch.cuda.synchronize()
    assert r["stats"] == e["stats"], (r["stats"], e["stats"])
    for key in ("best", "best_o") + (("best_scored", "best_o_scored") if fin else ()):
        assert (r[key].cpu().numpy().view(np.uint64) == e[key]).all(), key
    assert (r["mapq1"].cpu().numpy() == e["mapq1"]).all() and (r["mapq2"].cpu().numpy() == e["mapq2"]).all()
    for slot, data in (("tb1", "best_scored" if fin else "best"), ("tb2", "best_o_scored" if fin else "best_o")):
        al = (e[data][0] >> np.uint64(32)) != np.uint64(0xFFFFFFFF)
        for key in ("cigar_len", "cigar", "source", "sink"):
            got = r[slot][key].cpu().numpy()
            assert (got.view(e[slot][key].dtype) == e[slot][key]).all(), (slot, key)
        assert (r[slot]["score"].cpu().numpy()[al] == e[slot]["score"][al]).all(), slot
        if fin:
            k = "mds1" if slot == "tb1" else "mds2"
            assert (r[k + "_len"].cpu().numpy().view(np.uint32) == e[slot]["mds_len"]).all(), k
            m = np.arange(256)[None, :] < np.minimum(e[slot]["mds_len"], 256)[:, None]
            assert ((r[k].cpu().numpy() == e[slot]["mds"]) | ~m).all(), k
    if fin:
        e = dict(e, best=e["best_scored"], best_o=e["best_o_scored"])          # the sanity checks below look at the extension-stage words
    # the result itself: most pairs come back concordant at their fragment's two ends
    b, bo = e["best"][0], e["best_o"][0]
    paired = ((b >> np.uint64(30)) & np.uint64(1)) != 0
    disc = ((b >> np.uint64(31)) & np.uint64(1)) != 0
    conc = paired & ~disc
    a_pos = (b >> np.uint64(32)).astype(np.int64)
    good = conc & ((np.abs(a_pos - pos) <= 4) | (np.abs(a_pos - (pos + flen - L)) <= 4))
    expect = (np.arange(n) % 23 != 0) & (np.arange(n) % 17 != 0) & (np.arange(n) % 31 != 0)
    assert good[expect].mean() >= 0.8, (good[expect].mean(), conc.mean())
    assert e["stats"]["opposite_extensions"] > n // 2
    if params.pe_unpaired:
        lone = (np.arange(n) % 17 == 0) & (np.arange(n) % 31 != 0)
        assert (((b[lone] >> np.uint64(32)) != np.uint64(0xFFFFFFFF)) & ~paired[lone]).mean() >= 0.5     # mate 1 reported alone
    if params.pe_discordant and params.pe_unpaired:
        assert disc.sum() >= 1


import ctypes as _C  # noqa: E402


class _ShimParams(_C.Structure):
    _fields_ = [(k, _C.c_uint32) for k in ("local", "randomized", "top_seed", "max_effort_init", "max_effort", "min_ext", "max_ext", "max_reseed", "rep_seeds",
                                           "max_hits", "allow_sub", "subseed_len", "seed_len", "seed_freq_type", "min_read_len", "max_dist", "no_multi_hits",
                                           "batch_size", "hits_stride")] + \
               [("seed_freq_k", _C.c_float), ("seed_freq_m", _C.c_float), ("match", _C.c_int32), ("score_min_type", _C.c_int32),
                ("score_min_k", _C.c_float), ("score_min_m", _C.c_float), ("finish", _C.c_uint32), ("edit_distance", _C.c_uint32)]


@pytest.mark.parametrize("config", ["default", "no_rand", "one_hit_rounds", "multi_rounds", "one_mismatch_seeds", "local", "low_effort", "finish"])
def test_cxx_aligner_driver_matches_oracle(cuda, config):
    """The C++ host driver (include/nvbio_hip/aligner.h: nvbio::bowtie2::cuda::Aligner::best_approx), called through tests/cxx/aligner_shim.cpp
    on device-resident inputs: identical to the numpy driver over the oracle (and so to the Python driver)."""
    import ctypes as C
    shim_path = os.path.join(HERE, "cxx", "libaligner_shim.so")
    if not os.path.exists(shim_path):
        pytest.fail("tests/cxx/libaligner_shim.so is missing: run `python __graft_entry__.py`")
    shim = C.CDLL(shim_path)
    rng = np.random.default_rng(123)
    text = _small_index(rng)
    host, rhost = O.FMIndex(text), O.FMIndex(text[::-1].copy())
    fmi, rfmi = nvb.FMIndexDevice.from_host(host, cuda), nvb.FMIndexDevice.from_host(rhost, cuda)
    n, L = 1200, 100
    sym, pos = _reads(rng, text, n, L)
    names = ["sim.%d" % i for i in range(n)]
    params = A.Params(**CONFIGS[config])
    scheme = nvb.SmithWatermanScoringScheme.local() if params.local else nvb.SmithWatermanScoringScheme()
    gw = W._pack_chunked(torch.from_numpy(text), 2, True)
    fin = config == "finish"
    e = OD.best_approx(host, rhost, sym, gw.numpy().view(np.uint32), text.size, params, scheme, names, 1 if params.local else 2, finish=fin)

    d_sym = torch.from_numpy(sym).to(cuda)
    reads_rev, fwrc = P.pack_read_streams(d_sym)
    quals = torch.full((2 * n * L + 8,), 30, dtype=torch.uint8, device=cuda)
    arena, idx = S.pack_names(names, cuda)
    d_gw = gw.to(cuda)
    sp = _ShimParams(int(params.local), int(params.randomized), params.top_seed, params.max_effort_init, params.max_effort, params.min_ext, params.max_ext,
                     params.max_reseed, params.rep_seeds, params.max_hits, params.allow_sub, params.subseed_len, params.seed_len, params.seed_freq[0],
                     params.min_read_len, params.max_dist, int(params.no_multi_hits), params.batch_size, params.hits_stride or 0,
                     params.seed_freq[1], params.seed_freq[2], scheme.m_match, scheme.m_score_min[0], scheme.m_score_min[1], scheme.m_score_min[2], int(fin))
    mds = np.zeros((n, 256), np.uint8); mds_len = np.zeros(n, np.uint32)
    best = np.zeros((2, n), np.uint64); mapq = np.zeros(n, np.uint8); cigar = np.zeros((n, 64), np.uint16); cigar_len = np.zeros(n, np.uint32)
    source = np.zeros((n, 2), np.uint32); sink = np.zeros((n, 2), np.uint32); tb_score = np.zeros(n, np.int32); stats = np.zeros(12, np.uint64)
    fs, rs = fmi.struct(), rfmi.struct()
    vp = lambda t: C.c_void_p(t.data_ptr())
    hp = lambda a: a.ctypes.data_as(C.c_void_p)
    torch.cuda.synchronize()
    rc = shim.nvbio_aligner_best_approx(C.byref(fs), C.byref(rs), C.c_uint32(n), C.c_uint32(L), vp(reads_rev.words), C.c_uint64(reads_rev.words.numel()),
                                        vp(reads_rev.begin), vp(fwrc), C.c_uint64(fwrc.numel()), vp(quals), C.c_uint64(quals.numel()), vp(arena), vp(idx),
                                        vp(d_gw), C.c_uint64(d_gw.numel()), C.c_uint32(text.size), C.byref(sp),
                                        hp(best), hp(mapq), hp(cigar), hp(cigar_len), hp(source), hp(sink), hp(tb_score), hp(stats), hp(mds), hp(mds_len))
    assert rc == 0
    if fin:
        assert (mds_len == e["mds_len"]).all()
        m = np.arange(256)[None, :] < np.minimum(e["mds_len"], 256)[:, None]
        assert ((mds == e["mds"]) | ~m).all()
    assert (best == e["best"]).all() and (mapq == e["mapq"]).all()
    st = e["stats"]
    assert (int(stats[0]), int(stats[1]), int(stats[2])) == (st["extensions"], st["rounds"], st["seeding_passes"])
    assert [int(x) for x in stats[4:4 + int(stats[3])]] == st["queue"]
    ids, tb = e["aligned_ids"], e["tb"]
    assert (cigar_len[ids] == tb["cigar_len"]).all() and (cigar[ids] == tb["cigar"][: ids.size]).all()
    assert (source[ids] == tb["source"]).all() and (sink[ids] == tb["sink"]).all() and (tb_score[ids] == tb["score"]).all()
    rest = ~np.isin(np.arange(n), ids)
    assert (cigar_len[rest] == 0).all() and (source[rest] == 0xFFFFFFFF).all()


@pytest.mark.parametrize("max_dist", [15, 7])
def test_cxx_aligner_driver_in_edit_distance_mode_equals_the_python_driver(cuda, max_dist):
    """--scoring ed (params.h:47-51, compute_thread.cu:296): both from-scratch drivers extend, reduce and trace with the edit-distance
    aligner against score-min = -max_dist, and give MAPQ and the final scores from the Smith-Waterman scheme.  The Python driver is held
    to the unchanged nvBowtie by tests/test_ref_tests_gpu.py (--scoring ed); here the C++ driver is held to the Python one."""
    import ctypes as C
    shim = C.CDLL(os.path.join(HERE, "cxx", "libaligner_shim.so"))
    rng = np.random.default_rng(321 + max_dist)
    text = _small_index(rng)
    host, rhost = O.FMIndex(text), O.FMIndex(text[::-1].copy())
    fmi, rfmi = nvb.FMIndexDevice.from_host(host, cuda), nvb.FMIndexDevice.from_host(rhost, cuda)
    n, L = 1500, 100
    sym, pos = _reads(rng, text, n, L)
    names = ["sim.%d" % i for i in range(n)]
    params = A.Params(scoring_mode="ed", max_dist=max_dist)
    scheme = nvb.SmithWatermanScoringScheme()
    gw = W._pack_chunked(torch.from_numpy(text), 2, True)
    d_sym = torch.from_numpy(sym).to(cuda)
    d_gw = gw.to(cuda)
    e = A.best_approx(fmi, rfmi, d_sym, d_gw, text.size, params, scheme, names, finish=True, cigar_stride=64)
    sw = A.best_approx(fmi, rfmi, d_sym, d_gw, text.size, A.Params(max_dist=max_dist), scheme, names, finish=True, cigar_stride=64)
    reads_rev, fwrc = P.pack_read_streams(d_sym)
    quals = torch.full((2 * n * L + 8,), 30, dtype=torch.uint8, device=cuda)
    arena, idx = S.pack_names(names, cuda)
    sp = _ShimParams(int(params.local), int(params.randomized), params.top_seed, params.max_effort_init, params.max_effort, params.min_ext, params.max_ext,
                     params.max_reseed, params.rep_seeds, params.max_hits, params.allow_sub, params.subseed_len, params.seed_len, params.seed_freq[0],
                     params.min_read_len, params.max_dist, int(params.no_multi_hits), params.batch_size, params.hits_stride or 0,
                     params.seed_freq[1], params.seed_freq[2], scheme.m_match, scheme.m_score_min[0], scheme.m_score_min[1], scheme.m_score_min[2], 1, 1)
    mds = np.zeros((n, 256), np.uint8); mds_len = np.zeros(n, np.uint32)
    best = np.zeros((2, n), np.uint64); mapq = np.zeros(n, np.uint8); cigar = np.zeros((n, 64), np.uint16); cigar_len = np.zeros(n, np.uint32)
    source = np.zeros((n, 2), np.uint32); sink = np.zeros((n, 2), np.uint32); tb_score = np.zeros(n, np.int32); stats = np.zeros(12, np.uint64)
    fs, rs = fmi.struct(), rfmi.struct()
    vp = lambda t: C.c_void_p(t.data_ptr())
    hp = lambda a: a.ctypes.data_as(C.c_void_p)
    torch.cuda.synchronize()
    rc = shim.nvbio_aligner_best_approx(C.byref(fs), C.byref(rs), C.c_uint32(n), C.c_uint32(L), vp(reads_rev.words), C.c_uint64(reads_rev.words.numel()),
                                        vp(reads_rev.begin), vp(fwrc), C.c_uint64(fwrc.numel()), vp(quals), C.c_uint64(quals.numel()), vp(arena), vp(idx),
                                        vp(d_gw), C.c_uint64(d_gw.numel()), C.c_uint32(text.size), C.byref(sp),
                                        hp(best), hp(mapq), hp(cigar), hp(cigar_len), hp(source), hp(sink), hp(tb_score), hp(stats), hp(mds), hp(mds_len))
    assert rc == 0
    eb = e["best"].cpu().numpy().view(np.uint64)
    assert (best == eb).all() and (mapq == e["mapq"].cpu().numpy()).all()
    ids = e["aligned_ids"].cpu().numpy()
    assert ids.size > n // 2
    assert (cigar_len[ids] == e["cigar_len"].cpu().numpy().view(np.uint32)[ids]).all()
    assert (cigar[ids] == e["cigar"].cpu().numpy().view(np.uint16)[ids]).all()
    assert (source[ids] == e["source"].cpu().numpy().view(np.uint32)[ids]).all()
    assert (mds_len[ids] == e["mds_len"].cpu().numpy().view(np.uint32)[ids]).all()
    # the mode matters: the two modes accept different reads (max_dist edits against score-min = -0.6 - 0.6 L)
    sb = sw["best"].cpu().numpy().view(np.uint64)
    aligned_ed = (eb[0] >> np.uint64(32)) != np.uint64(0xFFFFFFFF)
    aligned_sw = (sb[0] >> np.uint64(32)) != np.uint64(0xFFFFFFFF)
    assert (aligned_ed != aligned_sw).any() and (eb != sb).any()


@pytest.mark.parametrize("config", ["default", "local", "one_mismatch_seeds"])
def test_best_approx_ragged_reads_matches_oracle(cuda, config):
    """Reads of different lengths (14 .. 160 bp, per-base qualities) through the driver: per-read seed intervals, thresholds, windows,
    MAPQ scales, tracebacks and MD strings all follow each read's own length; vs the numpy driver over the oracle."""
    rng = np.random.default_rng(777)
    text = _small_index(rng)
    host, rhost = O.FMIndex(text), O.FMIndex(text[::-1].copy())
    fmi, rfmi = nvb.FMIndexDevice.from_host(host, cuda), nvb.FMIndexDevice.from_host(rhost, cuda)
    n = 900
    reads, quals, pos = [], [], []
    for i in range(n):
        L = int(rng.integers(14, 161))
        p = int(rng.integers(0, text.size - L))
        r = text[p:p + L].copy()
        for j in rng.integers(0, L, [0, 1, 2, 3][i % 4] * L // 60):
            r[j] = (r[j] + 1 + rng.integers(0, 3)) & 3
        if i % 9 == 0 and L > 40:
            d = int(rng.integers(10, L - 10)); r = np.concatenate([r[:d], r[d + 1:]])
        if i % 14 == 0:
            r[int(rng.integers(0, r.size))] = 4
        if i % 2:
            r = np.where(r > 3, r, 3 - r)[::-1].copy()
        reads.append(r); quals.append(rng.integers(2, 42, r.size).astype(np.uint8)); pos.append(p)
    names = ["rag.%d" % i for i in range(n)]
    params = A.Params(**CONFIGS[config])
    scheme = nvb.SmithWatermanScoringScheme.local() if params.local else nvb.SmithWatermanScoringScheme()
    gw = W._pack_chunked(torch.from_numpy(text), 2, True)
    e = OD.best_approx(host, rhost, reads, gw.numpy().view(np.uint32), text.size, params, scheme, names, 1 if params.local else 2, read_quals=quals, finish=True,
                       cigar_stride=96)
    index = np.zeros(n + 1, np.int64); index[1:] = np.cumsum([r.size for r in reads])
    batch = A.ReadBatch.from_ragged(torch.from_numpy(np.concatenate(reads)).to(cuda), torch.from_numpy(index).to(cuda), torch.from_numpy(np.concatenate(quals)).to(cuda))
    r = A.best_approx(fmi, rfmi, batch, gw.to(cuda), text.size, params, scheme, names, cigar_stride=96, finish=True)
    torch.cuda.synchronize()
    assert r["stats"] == e["stats"], (r["stats"], e["stats"])
    assert (r["best_scored"].cpu().numpy().view(np.uint64) == e["best_scored"]).all() and (r["best"].cpu().numpy().view(np.uint64) == e["best"]).all()
    assert (r["mapq"].cpu().numpy() == e["mapq"]).all()
    ids, tb = e["aligned_ids"], e["tb"]
    assert (r["aligned_ids"].cpu().numpy() == ids).all()
    assert (r["cigar_len"].cpu().numpy()[ids].view(np.uint32) == tb["cigar_len"]).all()
    assert (r["cigar"].cpu().numpy()[ids].view(np.uint16) == tb["cigar"][: ids.size]).all()
    assert (r["source"].cpu().numpy()[ids].view(np.uint32) == tb["source"]).all() and (r["sink"].cpu().numpy()[ids].view(np.uint32) == tb["sink"]).all()
    assert (r["mds_len"].cpu().numpy().view(np.uint32) == e["mds_len"]).all()
    m = np.arange(256)[None, :] < np.minimum(e["mds_len"], 256)[:, None]
    assert ((r["mds"].cpu().numpy() == e["mds"]) | ~m).all()
    loc = (e["best_scored"][0] >> np.uint64(32)).astype(np.int64)
    lens = np.array([x.size for x in reads])
    ok = (loc != 0xFFFFFFFF) & (np.abs(loc - np.array(pos)) <= 4)
    assert ok[lens >= 40].mean() > 0.75 and (loc != 0xFFFFFFFF)[lens < 22].sum() >= 1          # long reads recovered; some shorter than a seed still align


def test_best_approx_edge_cases_match_oracle(cuda):
    """Reads hanging over both genome ends (the located read start wraps below zero / the window is clipped), all-N and half-N reads,
    exact duplicates, reads from the repeat, a single-read batch and a batch where nothing aligns: no crash, and the oracle driver's
    results."""
    rng = np.random.default_rng(4242)
    text = _small_index(rng, 1 << 17)
    host, rhost = O.FMIndex(text), O.FMIndex(text[::-1].copy())
    fmi, rfmi = nvb.FMIndexDevice.from_host(host, cuda), nvb.FMIndexDevice.from_host(rhost, cuda)
    L = 100
    gw = W._pack_chunked(torch.from_numpy(text), 2, True)
    rc = lambda r: np.where(r > 3, r, 3 - r)[::-1]
    reads = []
    for k in (0, 1, 5, 14, 40):                                   # k junk bases hang over the genome start / end
        reads.append(np.concatenate([rng.integers(0, 4, k, dtype=np.uint8), text[:L - k]]))
        reads.append(np.concatenate([text[text.size - (L - k):], rng.integers(0, 4, k, dtype=np.uint8)]))
        reads.append(rc(reads[-2])); reads.append(rc(reads[-2]))
    reads.append(np.full(L, 4, np.uint8))
    reads.append(np.concatenate([text[3000:3050], np.full(50, 4, np.uint8)]))
    reads += [text[20000:20100].copy()] * 3
    reads += [text[7100:7200].copy(), rc(text[7300:7400])]          # inside the period-3 repeat
    reads += [rng.integers(0, 4, L, dtype=np.uint8) for _ in range(5)]
    sym = np.stack(reads).astype(np.uint8)
    for tag, batch, kw in (("mixed", sym, {}), ("single", sym[:1], {}), ("no_rand", sym, dict(randomized=False)), ("local", sym, dict(local=True, seed_len=20)),
                           ("unalignable", np.stack([rng.integers(0, 4, L, dtype=np.uint8) for _ in range(8)]).astype(np.uint8), {})):
        params = A.Params(**kw)
        scheme = nvb.SmithWatermanScoringScheme.local() if params.local else nvb.SmithWatermanScoringScheme()
        names = ["e%d" % i for i in range(batch.shape[0])]
        e = OD.best_approx(host, rhost, batch, gw.numpy().view(np.uint32), text.size, params, scheme, names, 1 if params.local else 2, finish=True)
        r = A.best_approx(fmi, rfmi, torch.from_numpy(batch).to(cuda), gw.to(cuda), text.size, params, scheme, names, cigar_stride=64, finish=True)
        torch.cuda.synchronize()
        assert r["stats"] == e["stats"], tag
        assert (r["best"].cpu().numpy().view(np.uint64) == e["best"]).all() and (r["best_scored"].cpu().numpy().view(np.uint64) == e["best_scored"]).all(), tag
        assert (r["mapq"].cpu().numpy() == e["mapq"]).all(), tag
        ids = e["aligned_ids"]
        assert (r["cigar"].cpu().numpy()[ids].view(np.uint16) == e["tb"]["cigar"][: ids.size]).all() and (r["mds_len"].cpu().numpy().view(np.uint32) == e["mds_len"]).all(), tag
        if tag == "mixed":
            assert ids.size >= 15
        if tag == "unalignable":
            assert ids.size <= 1


class _ShimPeParams(_C.Structure):
    _fields_ = [("pe_policy", _C.c_int32)] + [(k, _C.c_uint32) for k in ("pe_overlap", "pe_unpaired", "pe_discordant", "min_frag_len", "max_frag_len")]


@pytest.mark.parametrize("config", ["default", "no_rand_local", "one_hit_rounds", "multi_rounds_no_mixed", "ff_policy", "finish"])
def test_cxx_paired_aligner_driver_matches_oracle(cuda, config):
    """The C++ paired-end driver (Aligner::best_approx over a PairedReadBatch, include/nvbio_hip/aligner.h) through the shim: both slot sets, both MAPQs,
    all CIGARs (and MD strings) equal the numpy driver's over the oracle."""
    import ctypes as C
    shim_path = os.path.join(HERE, "cxx", "libaligner_shim.so")
    if not os.path.exists(shim_path):
        pytest.fail("tests/cxx/libaligner_shim.so is missing: run `python __graft_entry__.py`")
    shim = C.CDLL(shim_path)
    rng = np.random.default_rng(321)
    text = _small_index(rng, 1 << 17)
    host, rhost = O.FMIndex(text), O.FMIndex(text[::-1].copy())
    fmi, rfmi = nvb.FMIndexDevice.from_host(host, cuda), nvb.FMIndexDevice.from_host(rhost, cuda)
    n, L = 700, 100
    s1, s2, pos, flen = _pairs(rng, text, n, L)
    if PAIRED_CONFIGS[config].get("pe_policy") == 0:
        s2 = np.where(s2 > 3, s2, 3 - s2)[:, ::-1].copy()
    names = ["pair.%d" % i for i in range(n)]
    params = A.Params(**PAIRED_CONFIGS[config])
    scheme = nvb.SmithWatermanScoringScheme.local() if params.local else nvb.SmithWatermanScoringScheme()
    gw = W._pack_chunked(torch.from_numpy(text), 2, True)
    fin = config == "finish"
    e = OD.best_approx_paired(host, rhost, s1, s2, gw.numpy().view(np.uint32), text.size, params, scheme, names, 1 if params.local else 2, finish=fin)

    packed = [P.pack_read_streams(torch.from_numpy(s).to(cuda)) for s in (s1, s2)]
    quals = torch.full((2 * n * L + 8,), 30, dtype=torch.uint8, device=cuda)
    both = torch.cat([packed[0][1], packed[1][1]]); mate_offset = packed[0][1].numel() * 8
    both_q = torch.full((mate_offset + 2 * n * L + 8,), 30, dtype=torch.uint8, device=cuda)
    arena, idx = S.pack_names(names, cuda)
    d_gw = gw.to(cuda)
    sp = _ShimParams(int(params.local), int(params.randomized), params.top_seed, params.max_effort_init, params.max_effort, params.min_ext, params.max_ext,
                     params.max_reseed, params.rep_seeds, params.max_hits, params.allow_sub, params.subseed_len, params.seed_len, params.seed_freq[0],
                     params.min_read_len, params.max_dist, int(params.no_multi_hits), params.batch_size, params.hits_stride or 0,
                     params.seed_freq[1], params.seed_freq[2], scheme.m_match, scheme.m_score_min[0], scheme.m_score_min[1], scheme.m_score_min[2], int(fin))
    pp = _ShimPeParams(params.pe_policy, int(params.pe_overlap), int(params.pe_unpaired), int(params.pe_discordant), params.min_frag_len, params.max_frag_len)
    out = dict(best=[np.zeros((2, n), np.uint64) for _ in range(2)], mapq=[np.zeros(n, np.uint8) for _ in range(2)], cigar=[np.zeros((n, 64), np.uint16) for _ in range(2)],
               cigar_len=[np.zeros(n, np.uint32) for _ in range(2)], source=[np.zeros((n, 2), np.uint32) for _ in range(2)], sink=[np.zeros((n, 2), np.uint32) for _ in range(2)],
               mds=[np.zeros((n, 256), np.uint8) for _ in range(2)], mds_len=[np.zeros(n, np.uint32) for _ in range(2)])
    stats = np.zeros(12, np.uint64)
    pair_ptrs = lambda ts: (C.c_void_p * 2)(*[t.data_ptr() for t in ts])
    pair_host = lambda arrs: (C.c_void_p * 2)(*[a.ctypes.data for a in arrs])
    u64x2 = lambda v: (C.c_uint64 * 2)(*v)
    fs, rs = fmi.struct(), rfmi.struct()
    vp = lambda t: C.c_void_p(t.data_ptr())
    torch.cuda.synchronize()
    rc = shim.nvbio_aligner_best_approx_paired(
        C.byref(fs), C.byref(rs), C.c_uint32(n), C.c_uint32(L),
        pair_ptrs([packed[0][0].words, packed[1][0].words]), u64x2([packed[0][0].words.numel(), packed[1][0].words.numel()]), pair_ptrs([packed[0][0].begin, packed[1][0].begin]),
        pair_ptrs([packed[0][1], packed[1][1]]), u64x2([packed[0][1].numel(), packed[1][1].numel()]), vp(quals), C.c_uint64(quals.numel()), vp(arena), vp(idx),
        vp(both), C.c_uint64(both.numel()), C.c_uint64(mate_offset), vp(both_q), C.c_uint64(both_q.numel()),
        vp(d_gw), C.c_uint64(d_gw.numel()), C.c_uint32(text.size), C.byref(sp), C.byref(pp),
        pair_host(out["best"]), pair_host(out["mapq"]), pair_host(out["cigar"]), pair_host(out["cigar_len"]), pair_host(out["source"]), pair_host(out["sink"]),
        pair_host(out["mds"]), pair_host(out["mds_len"]), stats.ctypes.data_as(C.c_void_p))
    assert rc == 0
    st = e["stats"]
    assert (int(stats[0]), int(stats[1]), int(stats[2])) == (st["extensions"], st["rounds"], st["seeding_passes"])
    assert [int(x) for x in stats[4:4 + int(stats[3])]] == st["queue"]
    assert (out["best"][0] == e["best"]).all() and (out["best"][1] == e["best_o"]).all()
    assert (out["mapq"][0] == e["mapq1"]).all() and (out["mapq"][1] == e["mapq2"]).all()
    for slot, key in ((0, "tb1"), (1, "tb2")):
        for f in ("cigar_len", "cigar", "source", "sink"):
            assert (out[f][slot] == e[key][f]).all(), (key, f)
        if fin:
            assert (out["mds_len"][slot] == e[key]["mds_len"]).all(), key
            m = np.arange(256)[None, :] < np.minimum(e[key]["mds_len"], 256)[:, None]
            assert ((out["mds"][slot] == e[key]["mds"]) | ~m).all(), key


def test_cxx_paired_driver_in_edit_distance_mode_equals_the_python_driver(cuda):
    """--scoring ed for read pairs: anchor and opposite mates extended against the edit-distance costs and score-min = -max_dist, traced with
    the edit-distance aligner's banded and full-matrix walks, MAPQ and final scores from the Smith-Waterman scheme -- the C++ driver against
    the Python one (which tests/test_ref_tests_gpu.py holds to the unchanged nvBowtie --scoring ed)."""
    import ctypes as C
    shim = C.CDLL(os.path.join(HERE, "cxx", "libaligner_shim.so"))
    rng = np.random.default_rng(4321)
    text = _small_index(rng, 1 << 17)
    host, rhost = O.FMIndex(text), O.FMIndex(text[::-1].copy())
    fmi, rfmi = nvb.FMIndexDevice.from_host(host, cuda), nvb.FMIndexDevice.from_host(rhost, cuda)
    n, L = 900, 100
    s1, s2, pos, flen = _pairs(rng, text, n, L)
    names = ["pair.%d" % i for i in range(n)]
    params = A.Params(scoring_mode="ed")
    scheme = nvb.SmithWatermanScoringScheme()
    gw = W._pack_chunked(torch.from_numpy(text), 2, True)
    d1, d2, d_gw = torch.from_numpy(s1).to(cuda), torch.from_numpy(s2).to(cuda), gw.to(cuda)
    e = A.best_approx_paired(fmi, rfmi, d1, d2, d_gw, text.size, params, scheme, names, finish=True)
    sw = A.best_approx_paired(fmi, rfmi, d1, d2, d_gw, text.size, A.Params(), scheme, names, finish=True)
    packed = [P.pack_read_streams(x) for x in (d1, d2)]
    quals = torch.full((2 * n * L + 8,), 30, dtype=torch.uint8, device=cuda)
    both = torch.cat([packed[0][1], packed[1][1]]); mate_offset = packed[0][1].numel() * 8
    both_q = torch.full((mate_offset + 2 * n * L + 8,), 30, dtype=torch.uint8, device=cuda)
    arena, idx = S.pack_names(names, cuda)
    sp = _ShimParams(int(params.local), int(params.randomized), params.top_seed, params.max_effort_init, params.max_effort, params.min_ext, params.max_ext,
                     params.max_reseed, params.rep_seeds, params.max_hits, params.allow_sub, params.subseed_len, params.seed_len, params.seed_freq[0],
                     params.min_read_len, params.max_dist, int(params.no_multi_hits), params.batch_size, params.hits_stride or 0,
                     params.seed_freq[1], params.seed_freq[2], scheme.m_match, scheme.m_score_min[0], scheme.m_score_min[1], scheme.m_score_min[2], 1, 1)
    pp = _ShimPeParams(params.pe_policy, int(params.pe_overlap), int(params.pe_unpaired), int(params.pe_discordant), params.min_frag_len, params.max_frag_len)
    out = dict(best=[np.zeros((2, n), np.uint64) for _ in range(2)], mapq=[np.zeros(n, np.uint8) for _ in range(2)], cigar=[np.zeros((n, 64), np.uint16) for _ in range(2)],
               cigar_len=[np.zeros(n, np.uint32) for _ in range(2)], source=[np.zeros((n, 2), np.uint32) for _ in range(2)], sink=[np.zeros((n, 2), np.uint32) for _ in range(2)],
               mds=[np.zeros((n, 256), np.uint8) for _ in range(2)], mds_len=[np.zeros(n, np.uint32) for _ in range(2)])
    stats = np.zeros(12, np.uint64)
    pair_ptrs = lambda ts: (C.c_void_p * 2)(*[t.data_ptr() for t in ts])
    pair_host = lambda arrs: (C.c_void_p * 2)(*[a.ctypes.data for a in arrs])
    u64x2 = lambda v: (C.c_uint64 * 2)(*v)
    fs, rs = fmi.struct(), rfmi.struct()
    vp = lambda t: C.c_void_p(t.data_ptr())
    torch.cuda.synchronize()
    rc = shim.nvbio_aligner_best_approx_paired(
        C.byref(fs), C.byref(rs), C.c_uint32(n), C.c_uint32(L),
        pair_ptrs([packed[0][0].words, packed[1][0].words]), u64x2([packed[0][0].words.numel(), packed[1][0].words.numel()]), pair_ptrs([packed[0][0].begin, packed[1][0].begin]),
        pair_ptrs([packed[0][1], packed[1][1]]), u64x2([packed[0][1].numel(), packed[1][1].numel()]), vp(quals), C.c_uint64(quals.numel()), vp(arena), vp(idx),
        vp(both), C.c_uint64(both.numel()), C.c_uint64(mate_offset), vp(both_q), C.c_uint64(both_q.numel()),
        vp(d_gw), C.c_uint64(d_gw.numel()), C.c_uint32(text.size), C.byref(sp), C.byref(pp),
        pair_host(out["best"]), pair_host(out["mapq"]), pair_host(out["cigar"]), pair_host(out["cigar_len"]), pair_host(out["source"]), pair_host(out["sink"]),
        pair_host(out["mds"]), pair_host(out["mds_len"]), stats.ctypes.data_as(C.c_void_p))
    assert rc == 0
    u64 = lambda t: t.cpu().numpy().view(np.uint64)
    assert (out["best"][0] == u64(e["best"])).all() and (out["best"][1] == u64(e["best_o"])).all()
    assert (out["mapq"][0] == e["mapq1"].cpu().numpy()).all() and (out["mapq"][1] == e["mapq2"].cpu().numpy()).all()
    for slot, key, md in ((0, "tb1", "mds1"), (1, "tb2", "mds2")):
        assert (out["cigar_len"][slot] == e[key]["cigar_len"].cpu().numpy().view(np.uint32)).all(), key
        assert (out["cigar"][slot] == e[key]["cigar"].cpu().numpy().view(np.uint16)).all(), key
        assert (out["source"][slot] == e[key]["source"].cpu().numpy().view(np.uint32)).all(), key
        assert (out["mds_len"][slot] == e[md + "_len"].cpu().numpy().view(np.uint32)).all(), md
    b = u64(e["best"])[0]
    paired = ((b >> np.uint64(30)) & np.uint64(1)) != 0
    assert paired.mean() > 0.5                                        # most pairs come back paired in this mode too
    assert (u64(e["best"]) != u64(sw["best"])).any()                  # ... and the mode is not the Smith-Waterman one


def _ragged_reads(rng, text, n, lo, hi):
    """reads of lengths lo .. hi (a few substitutions, one in nine a deletion, every other one reverse-complemented) -> (flat symbols, index, flat quals)"""
    reads, quals = [], []
    for i in range(n):
        L = int(rng.integers(lo, hi + 1))
        p = int(rng.integers(0, text.size - L - 2))
        r = text[p:p + L].copy()
        for j in rng.integers(0, L, [0, 1, 2, 3][i % 4] * L // 60):
            r[j] = (r[j] + 1 + rng.integers(0, 3)) & 3
        if i % 9 == 0 and L > 40:
            d = int(rng.integers(10, L - 10)); r = np.concatenate([r[:d], r[d + 1:], text[p + L:p + L + 1]])
        if i % 2:
            r = (3 - r)[::-1].copy()
        reads.append(r.astype(np.uint8)); quals.append(rng.integers(2, 42, r.size).astype(np.uint8))
    index = np.zeros(n + 1, np.int64); index[1:] = np.cumsum([r.size for r in reads])
    return np.concatenate(reads), index, np.concatenate(quals)


def test_cxx_aligner_driver_takes_reads_of_their_own_lengths(cuda):
    """ReadBatch::read_begin / read_len (include/nvbio_hip/aligner.h): 60 .. 140-bp reads with per-base qualities through the C++ single-end driver,
    against the Python driver on the same ragged batch (which test_best_approx_ragged_reads_matches_oracle holds to the oracle)"""
    import ctypes as C
    shim = C.CDLL(os.path.join(HERE, "cxx", "libaligner_shim.so"))
    rng = np.random.default_rng(515)
    text = _small_index(rng)
    host, rhost = O.FMIndex(text), O.FMIndex(text[::-1].copy())
    fmi, rfmi = nvb.FMIndexDevice.from_host(host, cuda), nvb.FMIndexDevice.from_host(rhost, cuda)
    n = 1300
    flat, index, fq = _ragged_reads(rng, text, n, 60, 140)
    names = ["rag.%d" % i for i in range(n)]
    params = A.Params()
    scheme = nvb.SmithWatermanScoringScheme()
    d_gw = W._pack_chunked(torch.from_numpy(text), 2, True).to(cuda)
    rb = A.ReadBatch.from_ragged(torch.from_numpy(flat).to(cuda), torch.from_numpy(index).to(cuda), torch.from_numpy(fq).to(cuda))
    e = A.best_approx(fmi, rfmi, rb, d_gw, text.size, params, scheme, names, finish=True, cigar_stride=64)
    L = rb.max_len
    arena, idx = S.pack_names(names, cuda)
    sp = _ShimParams(int(params.local), int(params.randomized), params.top_seed, params.max_effort_init, params.max_effort, params.min_ext, params.max_ext,
                     params.max_reseed, params.rep_seeds, params.max_hits, params.allow_sub, params.subseed_len, params.seed_len, params.seed_freq[0],
                     params.min_read_len, params.max_dist, int(params.no_multi_hits), params.batch_size, params.hits_stride or 0,
                     params.seed_freq[1], params.seed_freq[2], scheme.m_match, scheme.m_score_min[0], scheme.m_score_min[1], scheme.m_score_min[2], 1, 0)
    mds = np.zeros((n, 256), np.uint8); mds_len = np.zeros(n, np.uint32)
    best = np.zeros((2, n), np.uint64); mapq = np.zeros(n, np.uint8); cigar = np.zeros((n, 64), np.uint16); cigar_len = np.zeros(n, np.uint32)
    source = np.zeros((n, 2), np.uint32); sink = np.zeros((n, 2), np.uint32); tb_score = np.zeros(n, np.int32); stats = np.zeros(12, np.uint64)
    fs, rs = fmi.struct(), rfmi.struct()
    vp = lambda t: C.c_void_p(t.data_ptr())
    hp = lambda a: a.ctypes.data_as(C.c_void_p)
    shim.nvbio_aligner_set_ragged.restype = None
    torch.cuda.synchronize()
    shim.nvbio_aligner_set_ragged(0, vp(rb.read_begin), vp(rb.read_len), C.c_uint64(int(rb.rc_offset)))
    try:
        rc = shim.nvbio_aligner_best_approx(C.byref(fs), C.byref(rs), C.c_uint32(n), C.c_uint32(L), vp(rb.reversed.words), C.c_uint64(rb.reversed.words.numel()),
                                            vp(rb.reversed.begin), vp(rb.fw_rc_words), C.c_uint64(rb.fw_rc_words.numel()), vp(rb.quals), C.c_uint64(rb.quals.numel()), vp(arena), vp(idx),
                                            vp(d_gw), C.c_uint64(d_gw.numel()), C.c_uint32(text.size), C.byref(sp),
                                            hp(best), hp(mapq), hp(cigar), hp(cigar_len), hp(source), hp(sink), hp(tb_score), hp(stats), hp(mds), hp(mds_len))
    finally:
        shim.nvbio_aligner_set_ragged(0, None, None, C.c_uint64(0))
    assert rc == 0
    assert (best == e["best"].cpu().numpy().view(np.uint64)).all() and (mapq == e["mapq"].cpu().numpy()).all()
    ids = e["aligned_ids"].cpu().numpy()
    assert ids.size > n * 3 // 4
    assert (cigar_len[ids] == e["cigar_len"].cpu().numpy().view(np.uint32)[ids]).all() and (cigar[ids] == e["cigar"].cpu().numpy().view(np.uint16)[ids]).all()
    assert (source[ids] == e["source"].cpu().numpy().view(np.uint32)[ids]).all() and (mds_len[ids] == e["mds_len"].cpu().numpy().view(np.uint32)[ids]).all()
    st = e["stats"]
    assert (int(stats[0]), int(stats[1]), int(stats[2])) == (st["extensions"], st["rounds"], st["seeding_passes"])


def test_cxx_paired_driver_takes_mates_of_their_own_lengths(cuda):
    """PairedReadBatch with ragged mates (70 .. 130 bp each) through the C++ paired driver, against the Python paired driver on the same batches
    (which tests/test_ref_tests_gpu.py holds to the unchanged nvBowtie on mixed-length pairs)"""
    import ctypes as C
    shim = C.CDLL(os.path.join(HERE, "cxx", "libaligner_shim.so"))
    rng = np.random.default_rng(616)
    text = _small_index(rng, 1 << 17)
    host, rhost = O.FMIndex(text), O.FMIndex(text[::-1].copy())
    fmi, rfmi = nvb.FMIndexDevice.from_host(host, cuda), nvb.FMIndexDevice.from_host(rhost, cuda)
    n = 800
    m1, m2 = [], []
    for i in range(n):
        f = int(rng.integers(220, 420)); p = int(rng.integers(0, text.size - f - 2))
        a, b = int(rng.integers(70, 131)), int(rng.integers(70, 131))
        r1 = text[p:p + a].copy(); r2 = (3 - text[p + f - b:p + f])[::-1].copy()
        for r in (r1, r2):
            mut = rng.random(r.size) < 0.02; r[mut] = (r[mut] + 1) & 3
        m1.append(r1.astype(np.uint8)); m2.append(r2.astype(np.uint8))
    names = ["pair.%d" % i for i in range(n)]

    def batch(reads):
        index = np.zeros(n + 1, np.int64); index[1:] = np.cumsum([r.size for r in reads])
        return A.ReadBatch.from_ragged(torch.from_numpy(np.concatenate(reads)).to(cuda), torch.from_numpy(index).to(cuda),
                                       torch.from_numpy(rng.integers(2, 42, int(index[-1])).astype(np.uint8)).to(cuda))
    b = [batch(m1), batch(m2)]
    params = A.Params()
    scheme = nvb.SmithWatermanScoringScheme()
    d_gw = W._pack_chunked(torch.from_numpy(text), 2, True).to(cuda)
    e = A.best_approx_paired(fmi, rfmi, b[0], b[1], d_gw, text.size, params, scheme, names, finish=True)
    L = max(b[0].max_len, b[1].max_len)
    both = torch.cat([b[0].fw_rc_words, b[1].fw_rc_words]); mate_offset = b[0].fw_rc_words.numel() * 8
    both_q = torch.zeros(mate_offset + 2 * int(b[1].rc_offset) + 8, dtype=torch.uint8, device=cuda)
    both_q[: 2 * int(b[0].rc_offset)] = b[0].quals[: 2 * int(b[0].rc_offset)]
    both_q[mate_offset: mate_offset + 2 * int(b[1].rc_offset)] = b[1].quals[: 2 * int(b[1].rc_offset)]
    arena, idx = S.pack_names(names, cuda)
    # (the shim takes one size for both mates' quality streams: pad them to the same length)
    qn = max(b[0].quals.numel(), b[1].quals.numel())
    mate_q = [torch.cat([x.quals, torch.zeros(qn - x.quals.numel(), dtype=torch.uint8, device=cuda)]) for x in b]
    sp = _ShimParams(int(params.local), int(params.randomized), params.top_seed, params.max_effort_init, params.max_effort, params.min_ext, params.max_ext,
                     params.max_reseed, params.rep_seeds, params.max_hits, params.allow_sub, params.subseed_len, params.seed_len, params.seed_freq[0],
                     params.min_read_len, params.max_dist, int(params.no_multi_hits), params.batch_size, params.hits_stride or 0,
                     params.seed_freq[1], params.seed_freq[2], scheme.m_match, scheme.m_score_min[0], scheme.m_score_min[1], scheme.m_score_min[2], 1, 0)
    pp = _ShimPeParams(params.pe_policy, int(params.pe_overlap), int(params.pe_unpaired), int(params.pe_discordant), params.min_frag_len, params.max_frag_len)
    out = dict(best=[np.zeros((2, n), np.uint64) for _ in range(2)], mapq=[np.zeros(n, np.uint8) for _ in range(2)], cigar=[np.zeros((n, 64), np.uint16) for _ in range(2)],
               cigar_len=[np.zeros(n, np.uint32) for _ in range(2)], source=[np.zeros((n, 2), np.uint32) for _ in range(2)], sink=[np.zeros((n, 2), np.uint32) for _ in range(2)],
               mds=[np.zeros((n, 256), np.uint8) for _ in range(2)], mds_len=[np.zeros(n, np.uint32) for _ in range(2)])
    stats = np.zeros(12, np.uint64)
    pair_ptrs = lambda ts: (C.c_void_p * 2)(*[t.data_ptr() for t in ts])
    pair_host = lambda arrs: (C.c_void_p * 2)(*[a.ctypes.data for a in arrs])
    u64x2 = lambda v: (C.c_uint64 * 2)(*v)
    fs, rs = fmi.struct(), rfmi.struct()
    vp = lambda t: C.c_void_p(t.data_ptr())
    shim.nvbio_aligner_set_ragged.restype = None
    torch.cuda.synchronize()
    for m in (0, 1):
        shim.nvbio_aligner_set_ragged(m, vp(b[m].read_begin), vp(b[m].read_len), C.c_uint64(int(b[m].rc_offset)))
    try:
        rc = shim.nvbio_aligner_best_approx_paired_quals(
            C.byref(fs), C.byref(rs), C.c_uint32(n), C.c_uint32(L),
            pair_ptrs([b[0].reversed.words, b[1].reversed.words]), u64x2([b[0].reversed.words.numel(), b[1].reversed.words.numel()]), pair_ptrs([b[0].reversed.begin, b[1].reversed.begin]),
            pair_ptrs([b[0].fw_rc_words, b[1].fw_rc_words]), u64x2([b[0].fw_rc_words.numel(), b[1].fw_rc_words.numel()]), pair_ptrs(mate_q),
            C.c_uint64(mate_q[0].numel()), vp(arena), vp(idx),
            vp(both), C.c_uint64(both.numel()), C.c_uint64(mate_offset), vp(both_q), C.c_uint64(both_q.numel()),
            vp(d_gw), C.c_uint64(d_gw.numel()), C.c_uint32(text.size), C.byref(sp), C.byref(pp),
            pair_host(out["best"]), pair_host(out["mapq"]), pair_host(out["cigar"]), pair_host(out["cigar_len"]), pair_host(out["source"]), pair_host(out["sink"]),
            pair_host(out["mds"]), pair_host(out["mds_len"]), stats.ctypes.data_as(C.c_void_p), None)
    finally:
        for m in (0, 1):
            shim.nvbio_aligner_set_ragged(m, None, None, C.c_uint64(0))
    assert rc == 0
    u64 = lambda t: t.cpu().numpy().view(np.uint64)
    assert (out["best"][0] == u64(e["best"])).all() and (out["best"][1] == u64(e["best_o"])).all()
    for slot, key in ((0, "mapq1"), (1, "mapq2")):
        bad = np.nonzero(out["mapq"][slot] != e[key].cpu().numpy())[0]
        assert bad.size == 0, (key, bad[:8], out["mapq"][slot][bad[:8]], e[key].cpu().numpy()[bad[:8]], [hex(int(x)) for x in out["best"][0][:, bad[:4]].ravel()],
                               [hex(int(x)) for x in out["best"][1][:, bad[:4]].ravel()], b[0].read_len.cpu().numpy()[bad[:4]], b[1].read_len.cpu().numpy()[bad[:4]])
    for slot, key, md in ((0, "tb1", "mds1"), (1, "tb2", "mds2")):
        assert (out["cigar_len"][slot] == e[key]["cigar_len"].cpu().numpy().view(np.uint32)).all(), key
        assert (out["cigar"][slot] == e[key]["cigar"].cpu().numpy().view(np.uint16)).all(), key
        assert (out["source"][slot] == e[key]["source"].cpu().numpy().view(np.uint32)).all(), key
        assert (out["mds_len"][slot] == e[md + "_len"].cpu().numpy().view(np.uint32)).all(), md
    paired = ((u64(e["best"])[0] >> np.uint64(30)) & np.uint64(1)) != 0
    assert paired.mean() > 0.7


def test_cxx_drivers_with_the_round_sizes_copied_instead_of_read_as_they_land(cuda):
    """select() (include/nvbio_hip/select.h) has the device leave a round's queue sizes in pinned host memory and reads them there; the switch is
    read once per process, so the synchronise-and-copy form (NVBIO_HIP_POLL_SIZES=0) runs the C++ drivers' oracle tests in a process of its own."""
    import subprocess
    env = dict(os.environ, NVBIO_HIP_POLL_SIZES="0")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                        "-k", "test_cxx_aligner_driver_matches_oracle or test_cxx_paired_aligner_driver_matches_oracle"],
                       env=env, capture_output=True, text=True, timeout=1500, cwd=os.path.dirname(HERE))
    assert r.returncode == 0 and " passed" in r.stdout, (r.stdout[-3000:], r.stderr[-1000:])
