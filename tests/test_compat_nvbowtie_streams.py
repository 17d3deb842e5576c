"""Streams of READ VIEWS through the drop-in template layer -- the concept nvBowtie's extension streams are built on (reads stored
back to front and looked at through io::ReadLoader as REVERSE / STANDARD or FORWARD / COMPLEMENT views, pattern.qualities(), genome
windows through PackedStringLoader, a quality-aware scheme with the seven accessors, a CIGAR-forming backtracer).  The caller,
tests/compat/nvbowtie_streams.hip, is a client written against the layer's documented concepts with its own types; that the
reference's stream classes themselves (alignment_utils.h:170-340, score_best_inl.h:54-148, score_paired / score_all /
score_opposite_inl.h, traceback_inl.h:53-189, scoring.h:206-356) bind and take the same routes is proved verbatim by
tools/ref_bind_check.py in the build container.
Banded score batches must run on the views IN PLACE (last_path() == "tuned-views": nvbio_hip_banded_gotoh_score_qual_views, nothing
staged); whole-window scores and the tracebacks run the tuned kernels on staged copies ("tuned"); every route reproduces the oracle
bit for bit."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import pyoracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "tests", "compat", "libnvbowtie_streams.so")

pytestmark = pytest.mark.gpu


class Args(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ("rdg_c", "rdg_k", "rfg_c", "rfg_k", "match", "mmp_min", "mmp_max", "local")] + [
        ("band_len", C.c_uint32),
        ("read_words", C.c_void_p), ("read_quals", C.c_void_p), ("read_index", C.c_void_p), ("longest", C.c_uint32),
        ("mate_words", C.c_void_p), ("mate_quals", C.c_void_p), ("mate_index", C.c_void_p), ("mate_longest", C.c_uint32),
        ("genome_words", C.c_void_p), ("genome_length", C.c_uint32),
        ("idx_queue", C.c_void_p), ("hits", C.c_void_p), ("n_hits", C.c_uint32),
        ("second_best", C.c_void_p), ("score_limit", C.c_int32),
        ("hit_score", C.c_void_p), ("hit_sink", C.c_void_p), ("raw_score", C.c_void_p), ("raw_sink", C.c_void_p),
        ("o_windows", C.c_void_p), ("max_window", C.c_uint32),
        ("cigar", C.c_void_p), ("cigar_stride", C.c_uint32), ("cigar_len", C.c_void_p), ("aln_score", C.c_void_p),
        ("aln_source", C.c_void_p), ("aln_sink", C.c_void_p)]


@pytest.fixture(scope="module")
def lib():
    assert os.path.exists(LIB), "build with python -c 'import __graft_entry__ as g; g.build()'"
    L = C.CDLL(LIB)
    L.bt2_banded_score.argtypes = [C.POINTER(Args), C.c_char_p]
    L.bt2_banded_score_generic.argtypes = [C.POINTER(Args)]
    L.bt2_opposite_score.argtypes = [C.POINTER(Args), C.c_char_p]
    L.bt2_traceback.argtypes = [C.POINTER(Args), C.c_int, C.c_char_p]
    return L


COMP = np.array([3, 2, 1, 0, 4], dtype=np.uint8)


def make_world(seed, n_reads, n_hits, min_len=40, max_len=100):
    """A genome, reads (and mates) sampled from both strands with substitutions / indels / Ns, stored REVERSED as nvBowtie stores them
    (io::REVERSE, nvBowtie.cpp:579-597), qualities alike; hits pointing at (or near) the reads' origins."""
    rng = np.random.default_rng(seed)
    G = 150000
    genome = rng.integers(0, 4, G, dtype=np.uint8)

    def sample(n):
        fw, qs, origin, strand = [], [], [], []
        for _ in range(n):
            L = int(rng.integers(min_len, max_len + 1))
            o = int(rng.integers(0, G - L - 8))
            s = genome[o:o + L + 6].copy()
            for j in rng.integers(0, s.size, int(rng.integers(0, 5))):
                s[j] = (s[j] + 1 + rng.integers(0, 3)) & 3
            k = int(rng.integers(0, 4))
            if k == 1:
                c = int(rng.integers(5, s.size - 5)); s = np.delete(s, slice(c, c + int(rng.integers(1, 4))))
            elif k == 2:
                c = int(rng.integers(5, s.size - 5)); s = np.insert(s, c, rng.integers(0, 4, int(rng.integers(1, 4))))
            s = s[:L]
            if rng.random() < 0.05:
                s[int(rng.integers(0, L))] = 4
            rc = bool(rng.integers(0, 2))
            read = COMP[s][::-1].copy() if rc else s            # the sequenced read: the sampled strand
            fw.append(read.astype(np.uint8)); qs.append(rng.integers(0, 60, L, dtype=np.uint8)); origin.append(o); strand.append(rc)
        return fw, qs, np.array(origin), np.array(strand)

    reads, quals, origin, strand = sample(n_reads)
    mates, mquals, morigin, mstrand = sample(n_reads)

    def store(rd, qs):
        rev = [r[::-1] for r in rd]
        ss = O.StringSet.from_lists(rev, 4, True)
        index = np.concatenate([ss.begin, [ss.begin[-1] + ss.length[-1]]]).astype(np.uint32)
        q = np.concatenate([x[::-1] for x in qs] + [np.zeros(8, np.uint8)])
        return ss, index, q

    rs, rindex, rq = store(reads, quals)
    ms, mindex, mq = store(mates, mquals)
    hit_read = rng.integers(0, n_reads, n_hits).astype(np.uint32)
    hit_rc = strand[hit_read].astype(np.uint32)
    flip = rng.random(n_hits) < 0.1
    hit_rc[flip] ^= 1
    hit_loc = (origin[hit_read] + rng.integers(-4, 5, n_hits)).clip(0, G - 1).astype(np.uint32)
    hit_loc[:8] = [0, 1, 6, 7, G - 1, G - 30, G - 101, G - 120][:8]                  # windows clamped at both genome ends
    far = rng.random(n_hits) < 0.05
    hit_loc[far] = rng.integers(0, G - 1, int(far.sum()))
    hits = np.stack([hit_read, hit_loc, hit_rc], axis=1).astype(np.uint32)
    idx_queue = rng.permutation(n_hits).astype(np.uint32)
    gw = O.pack(genome, 2, True, pad_words=4)
    return dict(genome=genome, G=G, gw=gw, reads=reads, quals=quals, mates=mates, mquals=mquals, rs=rs, rindex=rindex, rq=rq, ms=ms, mindex=mindex, mq=mq,
                hits=hits, idx_queue=idx_queue, second_best=rng.integers(-60, 40, n_reads).astype(np.int32), morigin=morigin, mstrand=mstrand,
                longest=max(len(r) for r in reads), mate_longest=max(len(r) for r in mates))


def view(read, qual, rc):
    """the pattern and quality string nvBowtie's AlignmentStrings::load hands the aligner for a hit on the given strand"""
    return (COMP[read][::-1].copy(), qual[::-1].copy()) if rc else (read, qual)


def jobs(w, band, mate=False, windows=None):
    """per hit: (pattern, quals) as aligned, and the genome window"""
    pats, qs, wb, wl = [], [], [], []
    for h, (rid, loc, rc) in enumerate(w["hits"]):
        if mate:
            p, q = view(w["mates"][rid], w["mquals"][rid], not rc)
            gb, ge = windows[h]
        else:
            p, q = view(w["reads"][rid], w["quals"][rid], bool(rc))
            gb = loc - band // 2 if loc > band // 2 else 0
            ge = min(gb + band + len(p), w["G"])
        pats.append(p); qs.append(q); wb.append(gb); wl.append(max(int(ge) - int(gb), 0))
    ps = O.StringSet.from_lists(pats, 4, True)
    qbuf = np.zeros(int(ps.begin[-1] + ps.length[-1]) + 8, np.uint8)
    for b, q in zip(ps.begin, qs):
        qbuf[int(b):int(b) + len(q)] = q
    ts = O.StringSet(w["gw"], 2, True, np.array(wb, np.uint64), np.array(wl, np.uint32))
    return ps, qbuf, ts


SCHEMES = {   # rdg const/coeff, rfg const/coeff, match, mmp min/max, local
    "local": (5, 3, 5, 3, 2, 2, 6, 1),
    "end_to_end": (5, 3, 6, 2, 0, 2, 6, 0),
}


def fill(a, w, dev, scheme, band):
    import torch
    keep = []

    def d(x):
        t = torch.from_numpy(np.ascontiguousarray(x).view(np.int32 if x.dtype == np.uint32 else np.int16 if x.dtype == np.uint16 else x.dtype)).to(dev)
        keep.append(t)
        return t.data_ptr()

    (a.rdg_c, a.rdg_k, a.rfg_c, a.rfg_k, a.match, a.mmp_min, a.mmp_max, a.local) = scheme
    a.band_len = band
    a.read_words, a.read_quals, a.read_index, a.longest = d(w["rs"].words), d(w["rq"]), d(w["rindex"]), w["longest"]
    a.mate_words, a.mate_quals, a.mate_index, a.mate_longest = d(w["ms"].words), d(w["mq"]), d(w["mindex"]), w["mate_longest"]
    a.genome_words, a.genome_length = d(w["gw"]), w["G"]
    a.idx_queue, a.hits, a.n_hits = d(w["idx_queue"]), d(w["hits"]), len(w["hits"])
    a.second_best, a.score_limit = d(w["second_best"]), -40
    n = len(w["hits"])
    out = dict(hit_score=torch.full((n,), 12345, dtype=torch.int32, device=dev), hit_sink=torch.zeros(n, dtype=torch.int32, device=dev),
               raw_score=torch.full((n,), 12345, dtype=torch.int32, device=dev), raw_sink=torch.zeros((n, 2), dtype=torch.int32, device=dev),
               cigar=torch.zeros((n, 64), dtype=torch.int16, device=dev), cigar_len=torch.zeros(n, dtype=torch.int32, device=dev),
               aln_score=torch.zeros(n, dtype=torch.int32, device=dev), aln_source=torch.zeros((n, 2), dtype=torch.int32, device=dev),
               aln_sink=torch.zeros((n, 2), dtype=torch.int32, device=dev))
    for k, t in out.items():
        setattr(a, k, t.data_ptr())
    a.cigar_stride = 64
    return keep, out


def scheme_tables(scheme):
    rdg_c, rdg_k, rfg_c, rfg_k, match, mmin, mmax, local = scheme
    lut = -O.qual_cost_lut(mmin, mmax)
    s5 = (match, -rdg_c - rdg_k, -rdg_k, -rfg_c - rfg_k, -rfg_k)
    return lut, s5, (O.LOCAL if local else O.SEMI_GLOBAL)


@pytest.fixture(scope="module")
def cuda():
    import torch
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.mark.parametrize("kind", ["local", "end_to_end"])
@pytest.mark.parametrize("band", [15, 31])
def test_best_score_stream_runs_tuned_and_matches_oracle(lib, cuda, kind, band):
    import torch
    w = make_world(100 + band, 1500, 12000)
    a = Args()
    keep, out = fill(a, w, cuda, SCHEMES[kind], band)
    path = C.create_string_buffer(16)
    assert lib.bt2_banded_score(C.byref(a), path) == 0
    assert path.value == b"tuned-views"
    lut, s5, ty = scheme_tables(SCHEMES[kind])
    ps, qbuf, ts = jobs(w, band)
    es, ek = O.batch_banded_gotoh_score_qual(band, ty, s5 + (0,), lut, qbuf, ps, ts)
    got_s = out["raw_score"].cpu().numpy()
    got_k = out["raw_sink"].cpu().numpy().view(np.uint32)
    assert (got_s == es).all()
    assert (got_k == ek).all()
    gb = np.array(ts.begin, dtype=np.uint32)
    assert (out["hit_score"].cpu().numpy() == np.maximum(es, -(1 << 16))).all()
    assert (out["hit_sink"].cpu().numpy().view(np.uint32) == (gb + ek[:, 0]).astype(np.uint32)).all()
    assert (es > (30 if kind == "local" else -30)).sum() > 3000
    # the staged tuned route (what the in-place one replaced) still agrees
    out["raw_score"].fill_(777); out["raw_sink"].fill_(5)
    assert lib.bt2_banded_score_staged(C.byref(a), path) == 0 and path.value == b"tuned"
    assert (out["raw_score"].cpu().numpy() == es).all() and (out["raw_sink"].cpu().numpy().view(np.uint32) == ek).all()


def test_generic_lane_gives_the_same_results(lib, cuda):
    """the one-lane-per-job template over the same stream (what an unrecognised stream runs): identical outputs"""
    w = make_world(7, 500, 3000)
    a = Args()
    keep, out = fill(a, w, cuda, SCHEMES["local"], 15)
    path = C.create_string_buffer(16)
    assert lib.bt2_banded_score(C.byref(a), path) == 0 and path.value == b"tuned-views"
    tuned = (out["raw_score"].cpu().numpy().copy(), out["raw_sink"].cpu().numpy().copy())
    out["raw_score"].fill_(777); out["raw_sink"].fill_(5)
    assert lib.bt2_banded_score_generic(C.byref(a)) == 0
    assert (out["raw_score"].cpu().numpy() == tuned[0]).all() and (out["raw_sink"].cpu().numpy() == tuned[1]).all()


def mate_windows(w, rng, max_window):
    """opposite-mate windows: around the mate's origin, a few empty (skipped by init_context)"""
    n = len(w["hits"])
    win = np.zeros((n, 2), np.uint32)
    for h, (rid, loc, rc) in enumerate(w["hits"]):
        o = int(w["morigin"][rid])
        b = max(0, o - int(rng.integers(0, 150)))
        e = min(w["G"], b + int(rng.integers(len(w["mates"][rid]) + 10, max_window)))
        win[h] = (b, e) if h % 37 != 5 else (b, b)
    return win


@pytest.mark.parametrize("kind", ["local", "end_to_end"])
def test_opposite_score_stream_runs_tuned_and_matches_oracle(lib, cuda, kind):
    import torch
    w = make_world(300, 800, 4000)
    rng = np.random.default_rng(5)
    win = mate_windows(w, rng, 400)
    # the mate is aligned on the strand opposite to the hit's: make that the mate's true strand for most hits
    w["hits"][:, 2] = np.where(rng.random(len(w["hits"])) < 0.9, 1 - w["mstrand"][w["hits"][:, 0]], w["hits"][:, 2])
    a = Args()
    keep, out = fill(a, w, cuda, SCHEMES[kind], 15)
    dw = torch.from_numpy(win.view(np.int32)).to(cuda)
    a.o_windows, a.max_window = dw.data_ptr(), 400
    path = C.create_string_buffer(16)
    assert lib.bt2_opposite_score(C.byref(a), path) == 0
    assert path.value == b"tuned"
    lut, s5, ty = scheme_tables(SCHEMES[kind])
    ps, qbuf, ts = jobs(w, 0, mate=True, windows=win)
    ms = np.full(len(ps), -40, np.int32)
    es, ek, ok = O.batch_gotoh_score_qual(0, ty, s5, lut, qbuf, ps, ts, min_score=ms)
    valid = win[:, 1] > win[:, 0]
    got_s = out["raw_score"].cpu().numpy()
    got_k = out["raw_sink"].cpu().numpy().view(np.uint32)
    assert (got_s[valid] == es[valid]).all() and (got_k[valid] == ek[valid]).all()
    assert (got_s[~valid] == -(1 << 30)).all()             # declined jobs are still output, with the sink init_context left (batched_inl.h:58-63): a fresh BestSink
    assert (es[valid] > (30 if kind == "local" else -30)).sum() > 1000


@pytest.mark.parametrize("kind", ["local", "end_to_end"])
@pytest.mark.parametrize("full", [0, 1])
def test_traceback_streams_run_tuned_and_match_oracle(lib, cuda, kind, full):
    import torch
    w = make_world(500 + full, 300, 700)
    rng = np.random.default_rng(11)
    a = Args()
    keep, out = fill(a, w, cuda, SCHEMES[kind], 15)
    lut, s5, ty = scheme_tables(SCHEMES[kind])
    if full:
        w["hits"][:, 2] = 1 - w["mstrand"][w["hits"][:, 0]]
        keep2, out = fill(a, w, cuda, SCHEMES[kind], 15)
        win = mate_windows(w, rng, 300)
        win[win[:, 1] == win[:, 0], 1] += 120
        win[:, 1] = np.minimum(win[:, 1], w["G"])
        dw = torch.from_numpy(win.view(np.int32)).to(cuda)
        a.o_windows, a.max_window = dw.data_ptr(), 300
        ps, qbuf, ts = jobs(w, 0, mate=True, windows=win)
        exp = O.batch_gotoh_traceback(ty, s5, ps, ts, 64, lut, qbuf)
    else:
        ps, qbuf, ts = jobs(w, 15)
        exp = O.batch_banded_gotoh_traceback(15, ty, s5 + (0,), ps, ts, 64, lut, qbuf)
    path = C.create_string_buffer(16)
    assert lib.bt2_traceback(C.byref(a), full, path) == 0
    assert path.value == b"tuned"
    assert (out["aln_score"].cpu().numpy() == exp["score"]).all()
    assert (out["aln_sink"].cpu().numpy().view(np.uint32) == exp["sink"]).all()
    assert (out["aln_source"].cpu().numpy().view(np.uint32) == exp["source"]).all()
    gl = out["cigar_len"].cpu().numpy().view(np.uint32)
    assert (gl == exp["cigar_len"]).all()
    gc = out["cigar"].cpu().numpy().view(np.uint16)
    for i in range(len(gl)):
        assert (gc[i, :gl[i]] == exp["cigar"][i, :gl[i]]).all(), i
    assert (exp["cigar_len"] > 1).sum() > 50
