"""ctypes/numpy front-end of the CPU oracle (oracle/nvbio_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package (nvbio_amd) never imports it.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "nvbio_oracle.c")
_LIB = os.path.join(_HERE, "libnvbio_oracle.so")

GLOBAL, LOCAL, SEMI_GLOBAL = 0, 1, 2   # nvbio/alignment/alignment_base.h:54

_CFLAGS = ["-O3", "-ffp-contract=off", "-fopenmp", "-fPIC", "-std=c99", "-fvisibility=hidden"]


def build(native=False, out=None):
    """Compile the oracle.  native=True adds -march=native (used for the
    cpu_baseline timing on the box it is timed on) and writes next to `out`."""
    out = out or _LIB
    flags = list(_CFLAGS) + (["-march=native"] if native else ["-march=x86-64-v2"])
    subprocess.check_call(["gcc"] + flags + ["-shared", "-o", out, _SRC, "-lm"])
    return out


class _Fmi(C.Structure):
    _fields_ = [("length", C.c_uint32), ("primary", C.c_uint32), ("L2", C.c_uint32 * 5),
                ("bwt_occ", C.c_void_p), ("ssa", C.c_void_p), ("sa_int", C.c_uint32)]


def _load(path):
    lib = C.CDLL(path)
    u32p, u64p, i32p, u8p = (C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_int32), C.POINTER(C.c_uint8))
    lib.oracle_banded_gotoh_score.restype = C.c_int
    lib.oracle_ref_banded_sw.restype = C.c_int32
    lib.oracle_ref_sw_gotoh.restype = C.c_int32
    lib.oracle_bwt_from_sa.restype = C.c_uint32
    lib.oracle_filter_rank.restype = C.c_uint64
    lib.oracle_num_threads.restype = C.c_int
    return lib


_lib = None
_lib_native = None


def lib(native=False):
    global _lib, _lib_native
    if native:
        if _lib_native is not None:
            return _lib_native
        import tempfile
        try:
            path = build(native=True, out=os.path.join(tempfile.gettempdir(), "libnvbio_oracle_native_%d.so" % os.getpid()))
            _lib_native = _load(path)
            return _lib_native
        except Exception as e:  # fall back to the portable build
            print("oracle: native rebuild failed (%s), using portable build" % e, file=sys.stderr)
    if _lib is None:
        if not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(_SRC):
            build()
        _lib = _load(_LIB)
    return _lib


def _p(a, ty=None):
    if a is None:
        return None
    return a.ctypes.data_as(C.c_void_p)


def _u32(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


def _u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


# ----------------------------------------------------------------------------
# packed streams
# ----------------------------------------------------------------------------
def words_for(n_symbols, bits):
    per = 32 // bits
    return (int(n_symbols) + per - 1) // per


def pack(sym, bits, big_endian, pad_words=4):
    """Pack uint8 symbols into uint32 words (PackedStream layout)."""
    sym = np.ascontiguousarray(sym, dtype=np.uint8)
    n = sym.size
    per = 32 // bits
    nw = words_for(n, bits) + pad_words
    s = np.zeros(nw * per, dtype=np.uint32)
    s[:n] = sym & ((1 << bits) - 1)
    s = s.reshape(nw, per)
    k = np.arange(per, dtype=np.uint32)
    sh = (32 - bits - k * bits) if big_endian else (k * bits)
    return np.bitwise_or.reduce(s << sh.astype(np.uint32), axis=1).astype(np.uint32)


def unpack(words, begin, n, bits, big_endian):
    out = np.zeros(n, dtype=np.uint8)
    w = _u32(words)
    lib().oracle_unpack(_p(w), C.c_uint64(begin), C.c_uint64(n), C.c_uint32(bits), C.c_uint32(big_endian), _p(out))
    return out


class StringSet:
    """A set of strings inside one packed word stream:
    string i = symbols [begin[i], begin[i]+length[i])."""

    def __init__(self, words, bits, big_endian, begin, length):
        self.words = _u32(words)
        self.bits = int(bits)
        self.big_endian = int(bool(big_endian))
        self.begin = _u64(begin)
        self.length = _u32(length)
        assert self.begin.size == self.length.size

    def __len__(self):
        return self.begin.size

    @staticmethod
    def from_device(ps):
        """Copy a device-side packed string set (any object with words/bits/big_endian/begin/
        length/fixed_length tensors, e.g. nvbio_amd.PackedStringSet) to the host."""
        n = len(ps)
        length = (ps.length.cpu().numpy().view(np.uint32) if ps.length is not None
                  else np.full(n, ps.fixed_length, dtype=np.uint32))
        return StringSet(ps.words.cpu().numpy().view(np.uint32), ps.bits, ps.big_endian,
                         ps.begin.cpu().numpy().view(np.uint64), length)

    @staticmethod
    def from_lists(strings, bits, big_endian):
        lens = np.array([len(s) for s in strings], dtype=np.uint32)
        begin = np.zeros(len(strings), dtype=np.uint64)
        if len(strings):
            begin[1:] = np.cumsum(lens[:-1], dtype=np.uint64)
        cat = np.concatenate([np.asarray(s, dtype=np.uint8) for s in strings]) if len(strings) else np.zeros(0, np.uint8)
        return StringSet(pack(cat, bits, big_endian), bits, big_endian, begin, lens)


# ----------------------------------------------------------------------------
# alignment
# ----------------------------------------------------------------------------
def _scheme(s):
    return np.ascontiguousarray(s, dtype=np.int32)


def banded_gotoh_score(band, aln_type, scheme, pattern, text, pat_bits=8, txt_bits=8, pat_be=False, txt_be=False):
    """Single alignment of two uint8 symbol arrays -> (ok, score, sink_x, sink_y)."""
    pw = pack(pattern, pat_bits, pat_be)
    tw = pack(text, txt_bits, txt_be)
    out = np.zeros(3, dtype=np.int32)
    sc = _scheme(scheme)
    ok = lib().oracle_banded_gotoh_score(
        C.c_uint32(band), C.c_int(aln_type), _p(sc),
        _p(pw), C.c_uint32(pat_bits), C.c_uint32(pat_be), C.c_uint64(0), C.c_uint32(len(pattern)),
        _p(tw), C.c_uint32(txt_bits), C.c_uint32(txt_be), C.c_uint64(0), C.c_uint32(len(text)),
        _p(out))
    return bool(ok), int(out[0]), int(np.uint32(out[1])), int(np.uint32(out[2]))


def batch_banded_gotoh_score(band, aln_type, scheme, patterns, texts, n_threads=0, native=False):
    """HostThreadScheduler semantics over two StringSets -> (score[n] int32, sink[n,2] uint32)."""
    n = len(patterns)
    assert len(texts) == n
    score = np.empty(n, dtype=np.int32)
    sink = np.empty((n, 2), dtype=np.uint32)
    sc = _scheme(scheme)
    lib(native).oracle_batch_banded_gotoh_score(
        C.c_uint32(band), C.c_int(aln_type), _p(sc),
        _p(patterns.words), C.c_uint32(patterns.bits), C.c_uint32(patterns.big_endian), _p(patterns.begin), _p(patterns.length),
        _p(texts.words), C.c_uint32(texts.bits), C.c_uint32(texts.big_endian), _p(texts.begin), _p(texts.length),
        C.c_uint32(n), _p(score), _p(sink), C.c_int(n_threads))
    return score, sink


def batch_gotoh_score(aln_type, scheme, patterns, texts, min_score=None, n_threads=0, native=False):
    """Full-matrix Gotoh score, text-blocking form (gotoh_inl.h:969-1489), HostThreadScheduler semantics
    -> (score[n], sink[n,2], ok[n])."""
    n = len(patterns)
    score = np.empty(n, dtype=np.int32)
    sink = np.empty((n, 2), dtype=np.uint32)
    ok = np.empty(n, dtype=np.uint8)
    sc = _scheme(scheme)
    ms = np.ascontiguousarray(min_score, dtype=np.int32) if min_score is not None else None
    lib(native).oracle_batch_gotoh_score(
        C.c_int(aln_type), _p(sc),
        _p(patterns.words), C.c_uint32(patterns.bits), C.c_uint32(patterns.big_endian), _p(patterns.begin), _p(patterns.length),
        _p(texts.words), C.c_uint32(texts.bits), C.c_uint32(texts.big_endian), _p(texts.begin), _p(texts.length),
        _p(ms), C.c_uint32(n), _p(score), _p(sink), _p(ok), C.c_int(n_threads))
    return score, sink, ok


def banded_gotoh_traceback(band, aln_type, scheme, patterns, texts, i=0, mm_lut=None, quals=None):
    """banded_alignment_traceback of job i of two StringSets -> dict(score, source, sink, ops (end first:
    0=M 1=I 2=D), clip_end, clip_begin, cigar).  With mm_lut/quals: scheme = (match, pattern_gap_open,
    pattern_gap_ext, text_gap_open, text_gap_ext) (nvBowtie's quality-aware scheme).
    cigar = what nvBowtie's Backtracker (alignment_utils.h:125-168) leaves in its io::Cigar vector:
    uint16 = type | len << 2 (type:2, len:14 bit-field), stored end of the alignment first,
    soft clips (type 3) at both ends."""
    M, N = int(patterns.length[i]), int(texts.length[i])
    res = np.zeros(8, dtype=np.int32)
    cap = 2 * M + band + 8        # deletions <= band - 1 + insertions
    ops = np.zeros(cap, dtype=np.uint8)
    flags = np.zeros(max(1, M * band), dtype=np.uint8)
    tail = (_p(patterns.words), C.c_uint32(patterns.bits), C.c_uint32(patterns.big_endian), C.c_uint64(int(patterns.begin[i])), C.c_uint32(M),
            _p(texts.words), C.c_uint32(texts.bits), C.c_uint32(texts.big_endian), C.c_uint64(int(texts.begin[i])), C.c_uint32(N),
            _p(res), _p(ops), C.c_uint32(cap), _p(flags))
    if mm_lut is None:
        sc = _scheme(scheme)
        lib().oracle_banded_gotoh_traceback(C.c_uint32(band), C.c_int(aln_type), _p(sc), *tail)
    else:
        sc = np.ascontiguousarray(scheme, dtype=np.int32)
        lut = np.ascontiguousarray(mm_lut, dtype=np.int32)
        q = np.ascontiguousarray(quals, dtype=np.uint8)
        lib().oracle_banded_gotoh_traceback_qual(C.c_uint32(band), C.c_int(aln_type), _p(sc), _p(lut), _p(q), *tail)
    n = int(res[5])
    o = ops[:n].copy()
    cig = []
    if res[6]:
        cig.append(3 | (int(res[6]) << 2))
    k = 0
    while k < n:
        e = k
        while e < n and o[e] == o[k]:
            e += 1
        cig.append(int(o[k]) | ((e - k) << 2))
        k = e
    if res[7]:
        cig.append(3 | (int(res[7]) << 2))
    return dict(score=int(res[0]), source=(int(np.uint32(res[1])), int(np.uint32(res[2]))), sink=(int(np.uint32(res[3])), int(np.uint32(res[4]))),
                ops=o, clip_end=int(res[6]), clip_begin=int(res[7]), cigar=np.array(cig, dtype=np.uint16))


def batch_banded_gotoh_traceback(band, aln_type, scheme, patterns, texts, cigar_stride, mm_lut=None, quals=None):
    """banded_gotoh_traceback over two StringSets -> dict of arrays laid out like the C-ABI's outputs."""
    n = len(patterns)
    out = dict(score=np.empty(n, np.int32), sink=np.empty((n, 2), np.uint32), source=np.empty((n, 2), np.uint32),
               cigar=np.zeros((max(n, 1), cigar_stride), np.uint16), cigar_len=np.empty(n, np.uint32))
    for i in range(n):
        r = banded_gotoh_traceback(band, aln_type, scheme, patterns, texts, i, mm_lut, quals)
        out["score"][i] = r["score"]; out["sink"][i] = r["sink"]; out["source"][i] = r["source"]
        c = r["cigar"]
        out["cigar_len"][i] = c.size
        out["cigar"][i, :min(c.size, cigar_stride)] = c[:cigar_stride]
    return out


def gotoh_traceback(aln_type, scheme, patterns, texts, i=0, mm_lut=None, quals=None):
    """alignment_traceback (full matrix, Gotoh) of job i -> dict like banded_gotoh_traceback.  With mm_lut / quals:
    scheme = (match, pattern_gap_open, pattern_gap_ext, text_gap_open, text_gap_ext)."""
    M, N = int(patterns.length[i]), int(texts.length[i])
    res = np.zeros(8, dtype=np.int32)
    cap = M + N + 8
    ops = np.zeros(cap, dtype=np.uint8)
    flags = np.zeros(max(1, M * N), dtype=np.uint8)
    hrow, frow = np.zeros(M + 1, dtype=np.int32), np.zeros(M + 1, dtype=np.int32)
    sc = _scheme(scheme)
    tail = (_p(patterns.words), C.c_uint32(patterns.bits), C.c_uint32(patterns.big_endian), C.c_uint64(int(patterns.begin[i])), C.c_uint32(M),
            _p(texts.words), C.c_uint32(texts.bits), C.c_uint32(texts.big_endian), C.c_uint64(int(texts.begin[i])), C.c_uint32(N),
            _p(res), _p(ops), C.c_uint32(cap), _p(flags), _p(hrow), _p(frow))
    if mm_lut is None:
        lib().oracle_gotoh_traceback(C.c_int(aln_type), _p(sc), *tail)
    else:
        lib().oracle_gotoh_traceback_qual(C.c_int(aln_type), _p(sc), _p(np.ascontiguousarray(mm_lut, dtype=np.int32)), _p(np.ascontiguousarray(quals, dtype=np.uint8)), *tail)
    n = int(res[5])
    o = ops[:n].copy()
    cig = []
    if res[6]:
        cig.append(3 | (int(res[6]) << 2))
    k = 0
    while k < n:
        e = k
        while e < n and o[e] == o[k]:
            e += 1
        cig.append(int(o[k]) | ((e - k) << 2))
        k = e
    if res[7]:
        cig.append(3 | (int(res[7]) << 2))
    return dict(score=int(res[0]), source=(int(np.uint32(res[1])), int(np.uint32(res[2]))), sink=(int(np.uint32(res[3])), int(np.uint32(res[4]))),
                ops=o, clip_end=int(res[6]), clip_begin=int(res[7]), cigar=np.array(cig, dtype=np.uint16))


def sw_traceback(band, aln_type, scheme, patterns, texts, i=0):
    """SmithWatermanAligner / EditDistanceAligner traceback of job i (band > 0: banded; 0: full matrix)."""
    M, N = int(patterns.length[i]), int(texts.length[i])
    res = np.zeros(8, dtype=np.int32)
    cap = 2 * M + N + band + 8
    ops = np.zeros(cap, dtype=np.uint8)
    flags = np.zeros(max(1, M * (band if band else N)), dtype=np.uint8)
    row = np.zeros(M + 1, dtype=np.int32)
    sc = _scheme(scheme)
    lib().oracle_sw_traceback(
        C.c_uint32(band), C.c_int(aln_type), _p(sc),
        _p(patterns.words), C.c_uint32(patterns.bits), C.c_uint32(patterns.big_endian), C.c_uint64(int(patterns.begin[i])), C.c_uint32(M),
        _p(texts.words), C.c_uint32(texts.bits), C.c_uint32(texts.big_endian), C.c_uint64(int(texts.begin[i])), C.c_uint32(N),
        _p(res), _p(ops), C.c_uint32(cap), _p(flags), _p(row))
    n = int(res[5])
    o = ops[:n].copy()
    cig = []
    if res[6]:
        cig.append(3 | (int(res[6]) << 2))
    k = 0
    while k < n:
        e = k
        while e < n and o[e] == o[k]:
            e += 1
        cig.append(int(o[k]) | ((e - k) << 2))
        k = e
    if res[7]:
        cig.append(3 | (int(res[7]) << 2))
    return dict(score=int(res[0]), source=(int(np.uint32(res[1])), int(np.uint32(res[2]))), sink=(int(np.uint32(res[3])), int(np.uint32(res[4]))),
                ops=o, clip_end=int(res[6]), clip_begin=int(res[7]), cigar=np.array(cig, dtype=np.uint16))


def batch_sw_traceback(band, aln_type, scheme, patterns, texts, cigar_stride):
    n = len(patterns)
    out = dict(score=np.empty(n, np.int32), sink=np.empty((n, 2), np.uint32), source=np.empty((n, 2), np.uint32),
               cigar=np.zeros((max(n, 1), cigar_stride), np.uint16), cigar_len=np.empty(n, np.uint32))
    for i in range(n):
        r = sw_traceback(band, aln_type, scheme, patterns, texts, i)
        out["score"][i] = r["score"]; out["sink"][i] = r["sink"]; out["source"][i] = r["source"]
        c = r["cigar"]
        out["cigar_len"][i] = c.size
        out["cigar"][i, :min(c.size, cigar_stride)] = c[:cigar_stride]
    return out


def batch_gotoh_traceback(aln_type, scheme, patterns, texts, cigar_stride, mm_lut=None, quals=None):
    n = len(patterns)
    out = dict(score=np.empty(n, np.int32), sink=np.empty((n, 2), np.uint32), source=np.empty((n, 2), np.uint32),
               cigar=np.zeros((max(n, 1), cigar_stride), np.uint16), cigar_len=np.empty(n, np.uint32))
    for i in range(n):
        r = gotoh_traceback(aln_type, scheme, patterns, texts, i, mm_lut, quals)
        out["score"][i] = r["score"]; out["sink"][i] = r["sink"]; out["source"][i] = r["source"]
        c = r["cigar"]
        out["cigar_len"][i] = c.size
        out["cigar"][i, :min(c.size, cigar_stride)] = c[:cigar_stride]
    return out


def cigar_rle(ops_end_first):
    """Run-length encode the backtracer's pushes in push order, as the reference test prints them
    (alignment_test_utils.h: rle(backtracker.aln))."""
    out, prev, cnt = [], None, 0
    for o in ops_end_first:
        if o == prev:
            cnt += 1
        else:
            if prev is not None:
                out.append("%d%s" % (cnt, "MID"[prev]))
            prev, cnt = int(o), 1
    if prev is not None:
        out.append("%d%s" % (cnt, "MID"[prev]))
    return "".join(out)


def batch_sw_score(band, aln_type, scheme, patterns, texts, n_threads=0):
    """SmithWatermanAligner (scheme = match, mismatch, deletion, insertion; edit distance = (0,-1,-1,-1)):
    banded score for band > 0, full-matrix text-blocking score for band == 0."""
    n = len(patterns)
    score = np.empty(n, dtype=np.int32)
    sink = np.empty((n, 2), dtype=np.uint32)
    sc = _scheme(scheme)
    lib().oracle_batch_sw_score(
        C.c_uint32(band), C.c_int(aln_type), _p(sc),
        _p(patterns.words), C.c_uint32(patterns.bits), C.c_uint32(patterns.big_endian), _p(patterns.begin), _p(patterns.length),
        _p(texts.words), C.c_uint32(texts.bits), C.c_uint32(texts.big_endian), _p(texts.begin), _p(texts.length),
        C.c_uint32(n), _p(score), _p(sink), C.c_int(n_threads))
    return score, sink


def batch_banded_myers_score(band, aln_type, alphabet, patterns, texts, min_score=-(1 << 30), sink_bits=32, n_threads=0):
    """EditDistanceAligner<TYPE, MyersTag<alphabet>> through BatchedBandedAlignmentScore<band>: the banded bit-vector edit distance
    (myers_banded_inl.h:236-291).  min_score is narrowed to int16 as the reference's signature does; sink_bits = 16 = BestSink<int16>."""
    n = len(patterns)
    score = np.empty(n, dtype=np.int32)
    sink = np.empty((n, 2), dtype=np.uint32)
    lib().oracle_batch_banded_myers_score(
        C.c_uint32(band), C.c_int(aln_type), C.c_uint32(alphabet), C.c_int32(min_score), C.c_uint32(sink_bits),
        _p(patterns.words), C.c_uint32(patterns.bits), C.c_uint32(patterns.big_endian), _p(patterns.begin), _p(patterns.length),
        _p(texts.words), C.c_uint32(texts.bits), C.c_uint32(texts.big_endian), _p(texts.begin), _p(texts.length),
        C.c_uint32(n), _p(score), _p(sink), C.c_int(n_threads))
    return score, sink


def batch_gotoh_score_qual(algorithm, aln_type, scheme5, mm_lut, quals, patterns, texts, min_score=None, n_threads=0):
    """Full-matrix Gotoh score with nvBowtie's quality-aware scheme: scheme5 = (match, pattern_gap_open, pattern_gap_ext,
    text_gap_open, text_gap_ext); algorithm 0 = pattern blocking, 1 = text blocking -> (score, sink, ok)."""
    n = len(patterns)
    score = np.empty(n, dtype=np.int32); sink = np.empty((n, 2), dtype=np.uint32); ok = np.empty(n, dtype=np.uint8)
    sc = np.ascontiguousarray(scheme5, dtype=np.int32); lut = np.ascontiguousarray(mm_lut, dtype=np.int32); q = np.ascontiguousarray(quals, dtype=np.uint8)
    ms = np.ascontiguousarray(min_score, dtype=np.int32) if min_score is not None else None
    lib().oracle_batch_gotoh_score_qual(
        C.c_int(algorithm), C.c_int(aln_type), _p(sc), _p(lut), _p(q),
        _p(patterns.words), C.c_uint32(patterns.bits), C.c_uint32(patterns.big_endian), _p(patterns.begin), _p(patterns.length),
        _p(texts.words), C.c_uint32(texts.bits), C.c_uint32(texts.big_endian), _p(texts.begin), _p(texts.length),
        _p(ms), C.c_uint32(n), _p(score), _p(sink), _p(ok), C.c_int(n_threads))
    return score, sink, ok


def batch_score_pattern_blocking(kind, aln_type, scheme, patterns, texts, min_score=None, n_threads=0):
    """Full-matrix score, pattern-blocking form (the default algorithm tag): kind 0 = Gotoh (gotoh_inl.h:459-900),
    kind 1 = SW / edit distance (sw_inl.h:417-760) -> (score[n], sink[n,2], ok[n])."""
    n = len(patterns)
    score = np.empty(n, dtype=np.int32)
    sink = np.empty((n, 2), dtype=np.uint32)
    ok = np.empty(n, dtype=np.uint8)
    sc = _scheme(scheme)
    ms = np.ascontiguousarray(min_score, dtype=np.int32) if min_score is not None else None
    lib().oracle_batch_score_pattern_blocking(
        C.c_int(kind), C.c_int(aln_type), _p(sc),
        _p(patterns.words), C.c_uint32(patterns.bits), C.c_uint32(patterns.big_endian), _p(patterns.begin), _p(patterns.length),
        _p(texts.words), C.c_uint32(texts.bits), C.c_uint32(texts.big_endian), _p(texts.begin), _p(texts.length),
        _p(ms), C.c_uint32(n), _p(score), _p(sink), _p(ok), C.c_int(n_threads))
    return score, sink, ok


def ref_sw_gotoh(aln_type, scheme, pattern, text):
    p = np.ascontiguousarray(pattern, dtype=np.uint8)
    t = np.ascontiguousarray(text, dtype=np.uint8)
    sc = _scheme(scheme)
    return int(lib().oracle_ref_sw_gotoh(C.c_int(aln_type), _p(sc), _p(p), C.c_uint32(p.size), _p(t), C.c_uint32(t.size)))


def qual_cost_lut(min_val, max_val):
    """QualCost<int>(min,max)(q) for q in 0..255 (nvBowtie scoring.h:86-104), computed in C floats."""
    lut = np.zeros(256, dtype=np.int32)
    lib().oracle_qual_cost_lut(C.c_int32(min_val), C.c_int32(max_val), _p(lut))
    return lut


def batch_banded_gotoh_score_qual(band, aln_type, scheme6, mm_lut, quals, patterns, texts, n_threads=0):
    """As batch_banded_gotoh_score with nvBowtie's quality-aware scheme: scheme6 = (match,
    pattern_gap_open, pattern_gap_ext, text_gap_open, text_gap_ext, 0), mm_lut[256] = mismatch(q)."""
    n = len(patterns)
    score = np.empty(n, dtype=np.int32)
    sink = np.empty((n, 2), dtype=np.uint32)
    sc = np.ascontiguousarray(scheme6, dtype=np.int32)
    lut = np.ascontiguousarray(mm_lut, dtype=np.int32)
    q = np.ascontiguousarray(quals, dtype=np.uint8)
    lib().oracle_batch_banded_gotoh_score_qual(
        C.c_uint32(band), C.c_int(aln_type), _p(sc), _p(lut), _p(q),
        _p(patterns.words), C.c_uint32(patterns.bits), C.c_uint32(patterns.big_endian), _p(patterns.begin), _p(patterns.length),
        _p(texts.words), C.c_uint32(texts.bits), C.c_uint32(texts.big_endian), _p(texts.begin), _p(texts.length),
        C.c_uint32(n), _p(score), _p(sink), C.c_int(n_threads))
    return score, sink


def ref_banded_sw(band, aln_type, scheme, pattern, text, pos=0):
    p = np.ascontiguousarray(pattern, dtype=np.uint8)
    t = np.ascontiguousarray(text, dtype=np.uint8)
    assert len(t) >= pos + len(p) + band - 1
    sc = _scheme(scheme)
    return int(lib().oracle_ref_banded_sw(C.c_uint32(band), C.c_int(aln_type), _p(sc), _p(p), C.c_uint32(len(p)), _p(t), C.c_uint32(pos)))


# ----------------------------------------------------------------------------
# FM-index
# ----------------------------------------------------------------------------
def suffix_array(text):
    """SA of text (symbols 0..3) with the reference's padding convention
    (nvbio/fmindex/bwt.h:36-45): n+1 rows, SA[0] = n (the empty '$' suffix).
    Prefix doubling in numpy -- test tooling for small n."""
    t = np.ascontiguousarray(text, dtype=np.uint8)
    n = t.size
    K = 12
    s = np.zeros(n + 1 + K, dtype=np.int64)
    s[:n] = t.astype(np.int64) + 1
    key = np.zeros(n + 1, dtype=np.int64)
    for k in range(K):
        key = key * 5 + s[k:k + n + 1]
    h = K
    while True:
        order = np.argsort(key, kind="stable")
        sk = key[order]
        newr = np.zeros(n + 1, dtype=np.int64)
        newr[1:] = np.cumsum(sk[1:] != sk[:-1])
        rank = np.empty(n + 1, dtype=np.int64)
        rank[order] = newr
        if newr[-1] == n:
            return order.astype(np.uint32)
        nxt = np.zeros(n + 1, dtype=np.int64)
        if h <= n:
            nxt[:n + 1 - h] = rank[h:] + 1
        key = rank * (n + 2) + nxt
        h *= 2


class FMIndex:
    """Host FM-index in the reference's production layout (interleaved bwt|occ
    records, nvbio/io/fmindex/fmindex_impl.cu:305-327; SSA every sa_int rows)."""

    def __init__(self, text=None, sa_int=16, parts=None):
        if parts is not None:
            self.length, self.primary, self.L2, self.bwt_occ, self.ssa, self.sa_int = parts
        else:
            t = np.ascontiguousarray(text, dtype=np.uint8)
            n = t.size
            sa = suffix_array(t)
            bwt = np.zeros(n + 1, dtype=np.uint8)
            primary = lib().oracle_bwt_from_sa(C.c_uint32(n), _p(t), _p(sa), _p(bwt))
            n_blocks = (n + 63) // 64
            bw = np.zeros(n_blocks * 4 + 4, dtype=np.uint32)
            pw = pack(bwt[:n], 2, True, pad_words=0)
            bw[:pw.size] = pw
            bwt_occ = np.zeros(max(n_blocks, 1) * 8, dtype=np.uint32)
            L2 = np.zeros(5, dtype=np.uint32)
            lib().oracle_build_bwt_occ(C.c_uint32(n), _p(bw), _p(bwt_occ), _p(L2))
            ssa = np.zeros((n + 1 + sa_int - 1) // sa_int, dtype=np.uint32)
            lib().oracle_build_ssa(C.c_uint32(n), _p(sa), C.c_uint32(sa_int), _p(ssa))
            self.length, self.primary, self.L2, self.bwt_occ, self.ssa, self.sa_int = n, int(primary), L2, bwt_occ, ssa, sa_int
            self.sa = sa
            self.bwt = bwt[:n]
        self._c = _Fmi()
        self._c.length = self.length
        self._c.primary = self.primary
        for i in range(5):
            self._c.L2[i] = int(self.L2[i])
        self._c.bwt_occ = self.bwt_occ.ctypes.data
        self._c.ssa = self.ssa.ctypes.data if self.ssa is not None else None
        self._c.sa_int = self.sa_int

    def _ref(self):
        return C.byref(self._c)

    def rank(self, k, c):
        k = _u32(k); c = np.ascontiguousarray(c, dtype=np.uint8)
        out = np.empty(k.size, dtype=np.uint32)
        lib().oracle_fm_rank(self._ref(), _p(k), _p(c), C.c_uint32(k.size), _p(out))
        return out

    def rank4(self, k):
        k = _u32(k)
        out = np.empty((k.size, 4), dtype=np.uint32)
        lib().oracle_fm_rank4(self._ref(), _p(k), C.c_uint32(k.size), _p(out))
        return out

    def rank_range(self, ranges, c):
        r = _u32(ranges).reshape(-1, 2); c = np.ascontiguousarray(c, dtype=np.uint8)
        out = np.empty_like(r)
        lib().oracle_fm_rank_range(self._ref(), _p(r), _p(c), C.c_uint32(r.shape[0]), _p(out))
        return out

    def match(self, seeds, n_threads=0, native=False, want_bytes=False):
        n = len(seeds)
        out = np.empty((n, 2), dtype=np.uint32)
        nbytes = C.c_uint64(0)
        lib(native).oracle_fm_match(self._ref(), _p(seeds.words), C.c_uint32(seeds.bits), C.c_uint32(seeds.big_endian),
                                    _p(seeds.begin), _p(seeds.length), C.c_uint32(n), _p(out), C.byref(nbytes), C.c_int(n_threads))
        return (out, int(nbytes.value)) if want_bytes else out

    def locate(self, rows, n_threads=0, native=False, want_steps=False):
        rows = _u32(rows)
        out = np.empty(rows.size, dtype=np.uint32)
        steps = C.c_uint64(0)
        lib(native).oracle_fm_locate(self._ref(), _p(rows), C.c_uint32(rows.size), _p(out), C.byref(steps), C.c_int(n_threads))
        return (out, int(steps.value)) if want_steps else out

    def locate_ssa_iterator(self, rows):
        rows = _u32(rows)
        out = np.empty((rows.size, 2), dtype=np.uint32)
        lib().oracle_fm_locate_ssa_iterator(self._ref(), _p(rows), C.c_uint32(rows.size), _p(out))
        return out

    def lookup_ssa_iterator(self, its):
        its = _u32(its).reshape(-1, 2)
        out = np.empty(its.shape[0], dtype=np.uint32)
        lib().oracle_fm_lookup_ssa_iterator(self._ref(), _p(its), C.c_uint32(its.shape[0]), _p(out))
        return out

    def filter_rank(self, seeds):
        n = len(seeds)
        ranges = np.empty((n, 2), dtype=np.uint32)
        slots = np.empty(n, dtype=np.uint64)
        total = lib().oracle_filter_rank(self._ref(), _p(seeds.words), C.c_uint32(seeds.bits), C.c_uint32(seeds.big_endian),
                                         _p(seeds.begin), _p(seeds.length), C.c_uint32(n), _p(ranges), _p(slots))
        return int(total), ranges, slots

    def filter_locate(self, ranges, slots, begin, end):
        hits = np.empty((end - begin, 2), dtype=np.uint32)
        lib().oracle_filter_locate(self._ref(), _p(_u32(ranges)), _p(_u64(slots)), C.c_uint32(len(slots)),
                                   C.c_uint64(begin), C.c_uint64(end), _p(hits))
        return hits


class MapParams(C.Structure):
    _fields_ = [(k, C.c_uint32) for k in ("seed_len", "min_read_len", "max_hits", "max_reseed", "retry", "rep_seeds", "fw", "rc")]


def simple_func_table(ftype, k, m, n):
    """nvBowtie's SimpleFunc(type,k,m)(x) for x in [0,n) (func.h:39-70), in C floats."""
    out = np.zeros(n, dtype=np.uint32)
    lib().oracle_simple_func_table(C.c_int(ftype), C.c_float(k), C.c_float(m), C.c_uint32(n), _p(out))
    return out


def map_exact(fmi, reads, params, seed_freq_by_len, hits_stride, in_queue=None):
    """nvBowtie exact seed mapping of a read StringSet -> (hits uint64[n_reads,stride], counts, reseed)."""
    n_reads = len(reads)
    q = _u32(in_queue) if in_queue is not None else None
    n = q.size if q is not None else n_reads
    hits = np.zeros((n_reads, hits_stride), dtype=np.uint64)
    counts = np.zeros(n_reads, dtype=np.uint32)
    reseed = np.zeros(n, dtype=np.uint8)
    sf = _u32(seed_freq_by_len)
    mp = MapParams(**params)
    lib().oracle_map_exact(fmi._ref(), _p(reads.words), C.c_uint32(reads.bits), C.c_uint32(reads.big_endian), _p(reads.begin), _p(reads.length),
                           _p(q), C.c_uint32(n), C.byref(mp), _p(sf), _p(hits), C.c_uint32(hits_stride), _p(counts), _p(reseed))
    return hits, counts, reseed


def map_seeds(algorithm, subseed_len, fmi, rfmi, reads, params, seed_freq_by_len, hits_stride, in_queue=None):
    """nvBowtie seed mapping (map_queues_kernel<ALGO>): algorithm 0 exact, 1 approx (exact subseed + one
    mismatch in the rest, forward index), 2 case pruning (needs rfmi = the index of the reversed text)."""
    n_reads = len(reads)
    q = _u32(in_queue) if in_queue is not None else None
    n = q.size if q is not None else n_reads
    hits = np.zeros((n_reads, hits_stride), dtype=np.uint64)
    counts = np.zeros(n_reads, dtype=np.uint32)
    reseed = np.zeros(n, dtype=np.uint8)
    sf = _u32(seed_freq_by_len)
    mp = MapParams(**params)
    lib().oracle_map(C.c_int(algorithm), C.c_uint32(subseed_len), fmi._ref(), rfmi._ref() if rfmi is not None else None,
                     _p(reads.words), C.c_uint32(reads.bits), C.c_uint32(reads.big_endian), _p(reads.begin), _p(reads.length),
                     _p(q), C.c_uint32(n), C.byref(mp), _p(sf), _p(hits), C.c_uint32(hits_stride), _p(counts), _p(reseed))
    return hits, counts, reseed


def alignment_invalid():
    lib().oracle_alignment_invalid.restype = C.c_uint64
    return int(lib().oracle_alignment_invalid())


def init_alignments(read_len, score_min, mate=0):
    rl = _u32(read_len)
    best = np.zeros((2, rl.size), dtype=np.uint64)
    lib().oracle_init_alignments(C.c_uint32(rl.size), _p(rl), C.c_int(score_min[0]), C.c_float(score_min[1]), C.c_float(score_min[2]), C.c_uint32(mate),
                                 _p(best), C.c_uint32(rl.size))
    return best


def score_reduce(best, hit_begin, hit_score, hit_loc, hit_rc, read_len, read_ids=None):
    """score_reduce_kernel over uint64 best[2, stride] (in place)."""
    hb = _u64(hit_begin); n = hb.size - 1
    rid = _u32(read_ids) if read_ids is not None else None
    lib().oracle_score_reduce(C.c_uint32(n), _p(rid), _p(hb), _p(np.ascontiguousarray(hit_score, dtype=np.int32)), _p(_u32(hit_loc)),
                              _p(np.ascontiguousarray(hit_rc, dtype=np.uint8)), _p(_u32(read_len)), _p(best), C.c_uint32(best.shape[1]))
    return best


def hit_deque_ops(lib_handle=None):
    """(push, pop_bottom, pop_top) over uint64 arrays: the restated interval heap of the hit deque."""
    L = lib()
    mk = lambda f: (lambda a, n: f(_p(a), C.c_uint32(n)))
    return mk(L.oracle_hit_deque_push), mk(L.oracle_hit_deque_pop_bottom), mk(L.oracle_hit_deque_pop_top)


def hit_deque_make(a, n):
    """make_interval_heap over a[:n] in place (what priority_deque(seq, constructed=False) runs)."""
    lib().oracle_hit_deque_make(_p(a), C.c_uint32(n))


def sum_tree_node_count(size):
    lib().oracle_sum_tree_node_count.restype = C.c_uint32
    return int(lib().oracle_sum_tree_node_count(C.c_uint32(size)))


def sum_tree_setup(cells, size):
    lib().oracle_sum_tree_setup(_p(cells), C.c_uint32(size))


def sum_tree_set(cells, size, i, v):
    lib().oracle_sum_tree_set(_p(cells), C.c_uint32(size), C.c_uint32(i), C.c_float(v))


def sum_tree_sample(cells, size, v):
    lib().oracle_sum_tree_sample.restype = C.c_uint32
    return int(lib().oracle_sum_tree_sample(_p(cells), C.c_uint32(size), C.c_float(v)))


def pack_names(names):
    """(char arena uint8[], index uint32[n+1]) of NUL-terminated read names, as SequenceData keeps them."""
    blob = b"".join(nm.encode() + b"\0" for nm in names)
    idx = np.zeros(len(names) + 1, dtype=np.uint32)
    idx[1:] = np.cumsum([len(nm.encode()) + 1 for nm in names])
    return np.frombuffer(blob, dtype=np.uint8).copy(), idx


def select_init(hits, counts, names, names_idx, max_effort_init, randomized=True, top_seed=0, rseeds=None):
    """select_init_kernel: returns (probs float32[n, node_count(stride)], trys uint32[n], rseeds uint32[n])."""
    n, stride = hits.shape
    ps = sum_tree_node_count(stride)
    probs = np.zeros((n, ps), dtype=np.float32)
    trys = np.zeros(n, dtype=np.uint32)
    rs = np.zeros(n, dtype=np.uint32) if rseeds is None else _u32(rseeds).copy()
    lib().oracle_select_init(C.c_uint32(n), _p(names), _p(_u32(names_idx)) if names_idx is not None else None, _p(hits), C.c_uint32(stride),
                             _p(_u32(counts)), _p(probs), C.c_uint32(ps), _p(trys), _p(rs), C.c_uint32(max_effort_init),
                             C.c_int(int(randomized)), C.c_int(int(top_seed)))
    return probs, trys, rs


def select(randomized, n_multi, active_in, hits, counts, probs, rseeds, trys):
    """One selection round (in place on hits / counts / probs / rseeds).  Returns (active_out, hit_begin uint64[n_out+1],
    hit_read_id, hit_loc (SA rows), hit_seed (packed_seed words))."""
    q = _u32(active_in); n = q.size
    active_out = np.zeros(n, dtype=np.uint32)
    hit_begin = np.zeros(n + 1, dtype=np.uint64)
    cap = max(n * n_multi, 1)
    rid = np.zeros(cap, dtype=np.uint32); loc = np.zeros(cap, dtype=np.uint32); seed = np.zeros(cap, dtype=np.uint32)
    sizes = np.zeros(2, dtype=np.uint32)
    lib().oracle_select(C.c_int(int(randomized)), C.c_uint32(n_multi), _p(q), C.c_uint32(n), _p(hits), C.c_uint32(hits.shape[1]), _p(counts),
                        _p(probs), C.c_uint32(probs.shape[1]), _p(rseeds), _p(trys), _p(active_out), _p(hit_begin), _p(rid), _p(loc), _p(seed), _p(sizes))
    no, nh = int(sizes[0]), int(sizes[1])
    return active_out[:no].copy(), hit_begin[:no + 1].copy(), rid[:nh].copy(), loc[:nh].copy(), seed[:nh].copy()


def locate_hits(fmi, rfmi, hit_loc, hit_seed):
    """locate_kernel: SA rows -> read-start genome coordinates (in a copy)."""
    loc = _u32(hit_loc).copy()
    lib().oracle_locate_hits(fmi._ref(), rfmi._ref() if rfmi is not None else fmi._ref(), C.c_uint32(loc.size), _p(loc), _p(_u32(hit_seed)))
    return loc


def score_best_setup(hit_read_id, hit_loc, read_len, band_len, genome_len, best, score_limit):
    n = _u32(hit_read_id).size
    tb = np.zeros(n, dtype=np.uint64); tl = np.zeros(n, dtype=np.uint32); ms = np.zeros(n, dtype=np.int32)
    lib().oracle_score_best_setup(C.c_uint32(n), _p(_u32(hit_read_id)), _p(_u32(hit_loc)), _p(_u32(read_len)), C.c_uint32(band_len),
                                  C.c_uint32(genome_len), _p(best), C.c_uint32(best.shape[1]), C.c_int32(score_limit), _p(tb), _p(tl), _p(ms))
    return tb, tl, ms


def score_reduce_best_approx(best, active, hit_begin, hit_score, hit_loc, hit_seed, read_len, worst_score, trys, counts,
                             n_ext, min_ext, max_ext, max_effort):
    """score_reduce_kernel + ReduceBestApproxContext, in place on best / trys / counts."""
    hb = _u64(hit_begin); n = hb.size - 1
    lib().oracle_score_reduce_best_approx(C.c_uint32(n), _p(_u32(active)), _p(hb), _p(np.ascontiguousarray(hit_score, dtype=np.int32)),
                                          _p(_u32(hit_loc)), _p(_u32(hit_seed)), _p(_u32(read_len)), _p(best), C.c_uint32(best.shape[1]),
                                          C.c_int32(worst_score), _p(trys), _p(counts), C.c_uint32(n_ext), C.c_uint32(min_ext),
                                          C.c_uint32(max_ext), C.c_uint32(max_effort))
    return best


def anchor_score_setup(hit_read_id, hit_loc, hit_seed, a_read_len, o_read_len, band_len, genome_len, best, best_o, match, score_min, score_limit, anchor):
    n = _u32(hit_read_id).size
    tb = np.zeros(n, dtype=np.uint64); tl = np.zeros(n, dtype=np.uint32); ms = np.zeros(n, dtype=np.int32)
    lib().oracle_anchor_score_setup(C.c_uint32(n), _p(_u32(hit_read_id)), _p(_u32(hit_loc)), _p(_u32(hit_seed)), _p(_u32(a_read_len)), _p(_u32(o_read_len)),
                                    C.c_uint32(band_len), C.c_uint32(genome_len), _p(best), _p(best_o), C.c_uint32(best.shape[1]), C.c_int32(match),
                                    C.c_int(score_min[0]), C.c_float(score_min[1]), C.c_float(score_min[2]), C.c_int32(score_limit), C.c_uint32(anchor),
                                    _p(tb), _p(tl), _p(ms))
    return tb, tl, ms


def anchor_score_finish(raw_score, raw_sink, text_begin, min_score, worst_score):
    n = raw_score.size
    hs = np.zeros(n, dtype=np.int32); hk = np.zeros(n, dtype=np.uint32)
    lib().oracle_anchor_score_finish(C.c_uint32(n), _p(np.ascontiguousarray(raw_score, dtype=np.int32)), _p(_u32(raw_sink)), _p(_u64(text_begin)),
                                     _p(np.ascontiguousarray(min_score, dtype=np.int32)), C.c_int32(worst_score), _p(hs), _p(hk))
    return hs, hk


def score_reduce_paired_best_approx(best, best_o, active, hit_begin, hit_loc, hit_sink, hit_score, hit_seed, o_loc, o_sink, o_sink2, o_score, o_score2,
                                    read_len, anchor, pe_policy, pe_unpaired, score_limit, trys, counts, n_ext, min_ext, max_ext, max_effort):
    hb = _u64(hit_begin); n = hb.size - 1
    i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
    lib().oracle_score_reduce_paired_best_approx(C.c_uint32(n), _p(_u32(active)), _p(hb), _p(_u32(hit_loc)), _p(_u32(hit_sink)), _p(i32(hit_score)), _p(_u32(hit_seed)),
                                                 _p(_u32(o_loc)), _p(_u32(o_sink)), _p(_u32(o_sink2)), _p(i32(o_score)), _p(i32(o_score2)), _p(_u32(read_len)),
                                                 C.c_uint32(anchor), C.c_int(pe_policy), C.c_int(int(pe_unpaired)), C.c_int32(score_limit), _p(best), _p(best_o),
                                                 C.c_uint32(best.shape[1]), _p(trys), _p(counts), C.c_uint32(n_ext), C.c_uint32(min_ext), C.c_uint32(max_ext),
                                                 C.c_uint32(max_effort))
    return best, best_o


def mark_discordant(best, best_o):
    lib().oracle_mark_discordant(C.c_uint32(best.shape[1]), _p(best), _p(best_o), C.c_uint32(best.shape[1]))


def finish_alignment(valid, patterns, quals, texts, cigar, cigar_len, source, match, mismatch_lut, n_penalty, best_row, idx=None, mds_stride=256, gap_costs=(-8, -3, -8, -3)):
    """finish_alignment_kernel over n jobs: returns (mds uint8[n, stride], mds_len uint32[n]) and rewrites best_row[idx[i] | i] in place.
    gap_costs = (pattern_gap_open, pattern_gap_ext, text_gap_open, text_gap_ext) of the scheme; the default is nvBowtie's (5 + 3, 3) twice."""
    n = len(patterns)
    mds = np.zeros((max(n, 1), mds_stride), dtype=np.uint8); mds_len = np.zeros(n, dtype=np.uint32)
    cg = np.ascontiguousarray(cigar, dtype=np.uint16)
    q = np.ascontiguousarray(quals, dtype=np.uint8) if quals is not None else None
    lib().oracle_finish_alignment(C.c_uint32(n), _p(np.ascontiguousarray(valid, dtype=np.uint8)), _p(patterns.words), C.c_uint32(patterns.bits), C.c_uint32(patterns.big_endian),
                                  _p(patterns.begin), _p(patterns.length), _p(q), C.c_uint64(q.size if q is not None else 0),
                                  _p(texts.words), C.c_uint32(texts.big_endian), _p(texts.begin), _p(texts.length),
                                  _p(cg), C.c_uint32(cg.shape[1]), _p(_u32(cigar_len)), _p(_u32(source)),
                                  C.c_int32(match), _p(np.ascontiguousarray(mismatch_lut, dtype=np.int32)), C.c_int32(n_penalty), _p(np.ascontiguousarray(gap_costs, dtype=np.int32)),
                                  _p(_u32(idx)) if idx is not None else None, _p(best_row), _p(mds), C.c_uint32(mds_stride), _p(mds_len))
    return mds, mds_len


def mds_to_string(mds):
    """nvbio's byte-coded MDS -> the SAM MD:Z string (match counts, mismatched reference bases are not kept by nvbio: it stores the READ symbol
    of a mismatch, so the SAM writer re-reads the reference; here the read symbol is shown lower-case) -- a debugging aid for the tests."""
    n = int(mds[0]) | (int(mds[1]) << 8)
    out, i = [], 2
    while i < n:
        op = int(mds[i])
        if op == 0: out.append(str(int(mds[i + 1]))); i += 2
        elif op == 1: out.append("acgtn"[min(int(mds[i + 1]), 4)]); i += 2
        else:
            l = int(mds[i + 1]); out.append(("^" if op == 3 else "+") + "".join("ACGTN"[min(int(c), 4)] for c in mds[i + 2:i + 2 + l])); i += 2 + l
    return "".join(out)


def score_reduce_paired(best, best_o, hit_begin, hit_loc, hit_sink, hit_score, hit_rc, o_loc, o_sink, o_sink2, o_score, o_score2,
                        read_len, anchor, pe_policy, pe_unpaired, score_limit, read_ids=None):
    hb = _u64(hit_begin); n = hb.size - 1
    rid = _u32(read_ids) if read_ids is not None else None
    i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
    lib().oracle_score_reduce_paired(C.c_uint32(n), _p(rid), _p(hb), _p(_u32(hit_loc)), _p(_u32(hit_sink)), _p(i32(hit_score)), _p(np.ascontiguousarray(hit_rc, dtype=np.uint8)),
                                     _p(_u32(o_loc)), _p(_u32(o_sink)), _p(_u32(o_sink2)), _p(i32(o_score)), _p(i32(o_score2)), _p(_u32(read_len)),
                                     C.c_uint32(anchor), C.c_int(pe_policy), C.c_int(int(pe_unpaired)), C.c_int32(score_limit), _p(best), _p(best_o), C.c_uint32(best.shape[1]))
    return best, best_o


def mapq(version, match, score_min, monotone, best, read_len):
    n = best.shape[1]
    out = np.zeros(n, dtype=np.uint8)
    lib().oracle_mapq(C.c_int(version), C.c_int32(match), C.c_int(score_min[0]), C.c_float(score_min[1]), C.c_float(score_min[2]), C.c_int(int(monotone)),
                      C.c_uint32(n), _p(best), C.c_uint32(n), _p(_u32(read_len)), _p(out))
    return out


class _PeParams(C.Structure):
    _fields_ = [("pe_policy", C.c_int32), ("min_frag_len", C.c_int32), ("max_frag_len", C.c_int32), ("pe_overlap", C.c_int32),
                ("score_limit", C.c_int32), ("anchor", C.c_uint32), ("genome_length", C.c_uint32)]


def opposite_windows(hit_read_id, hit_rc, hit_loc, hit_score, a_read_len, o_read_len, best, best_o, match, score_min, text_gap_open, text_gap_ext,
                     pe_policy, min_frag_len, max_frag_len, pe_overlap, score_limit, anchor, genome_length):
    n = len(hit_loc)
    out = dict(valid=np.zeros(n, np.uint8), min_score=np.zeros(n, np.int32), read_rc=np.zeros(n, np.uint8), genome_begin=np.zeros(n, np.uint32), genome_end=np.zeros(n, np.uint32))
    pp = _PeParams(pe_policy, min_frag_len, max_frag_len, int(pe_overlap), score_limit, anchor, genome_length)
    lib().oracle_opposite_windows(C.c_uint32(n), _p(_u32(hit_read_id)), _p(np.ascontiguousarray(hit_rc, dtype=np.uint8)), _p(_u32(hit_loc)), _p(np.ascontiguousarray(hit_score, dtype=np.int32)),
                                  _p(_u32(a_read_len)), _p(_u32(o_read_len)), _p(best), _p(best_o), C.c_uint32(best.shape[1]),
                                  C.c_int32(match), C.c_int(score_min[0]), C.c_float(score_min[1]), C.c_float(score_min[2]), C.c_int32(text_gap_open), C.c_int32(text_gap_ext), C.byref(pp),
                                  _p(out["valid"]), _p(out["min_score"]), _p(out["read_rc"]), _p(out["genome_begin"]), _p(out["genome_end"]))
    return out


def mapq_paired(version, match, score_min, monotone, best, best_o, read_len, o_read_len):
    n = best.shape[1]
    out = np.zeros(n, dtype=np.uint8)
    lib().oracle_mapq_paired(C.c_int(version), C.c_int32(match), C.c_int(score_min[0]), C.c_float(score_min[1]), C.c_float(score_min[2]), C.c_int(int(monotone)),
                             C.c_uint32(n), _p(best), _p(best_o), C.c_uint32(n), _p(_u32(read_len)), _p(_u32(o_read_len)), _p(out))
    return out


def num_threads():
    return int(lib().oracle_num_threads())
