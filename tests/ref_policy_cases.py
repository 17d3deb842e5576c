"""Seeded inputs for the MAPQ / score-reduction pin (tests/test_ref_policy.py, tests/golden/make_ref_policy_vectors.py), and the
ctypes front of oracle/_ref/libref_policy.so -- nvBowtie's own mapq.h / reduce_inl.h / alignments.h compiled by
oracle/build_ref_policy.py.  Plain data plumbing: nothing here computes an expected value."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "_ref", "libref_policy.so")
INV = 0xFFFFFFFF
SCHEMES = ((0, (0, -0.6, -0.6)), (2, (1, 20.0, 8.0)), (0, (2, -3.0, -1.5)))      # (match bonus, --score-min type / const / coeff)


def P(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def u32(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


def ref_lib():
    lib = C.CDLL(LIB)
    lib.ref_alignment_pack.restype = C.c_uint64
    lib.ref_alignment_invalid.restype = C.c_uint64
    return lib


def pack_words(pos, ed, score, rc, mate, paired, disc):
    """io::Alignment words from fields: plain bit packing of the layout `alignments.h:128-129` declares (checked against the
    reference's own constructor in test_alignment_words_equal_the_reference_constructor)."""
    pos, ed, score, rc, mate, paired, disc = [np.asarray(x).astype(np.int64) for x in (pos, ed, score, rc, mate, paired, disc)]
    w = (score < 0).astype(np.uint64) | ((np.abs(score).astype(np.uint64) & np.uint64(0x1FFFF)) << np.uint64(1)) | ((ed.astype(np.uint64) & np.uint64(0x3FF)) << np.uint64(18)) \
        | (rc.astype(np.uint64) << np.uint64(28)) | (mate.astype(np.uint64) << np.uint64(29)) | (paired.astype(np.uint64) << np.uint64(30)) | (disc.astype(np.uint64) << np.uint64(31))
    return w | (pos.astype(np.uint64) << np.uint64(32))


def simple_func(t, k, m, x):
    """SimpleFunc in C floats (func.h:47-52): linear / log / sqrt"""
    x = np.asarray(x, dtype=np.float32)
    f = np.log(x) if t == 1 else np.sqrt(x) if t == 2 else x
    return (np.float32(k) + np.float32(m) * f.astype(np.float32)).astype(np.float32).astype(np.int32)


def mapq_se_case(seed, n, match, smin):
    """single-end best / second-best slots inside the calculators' defined domain (second-best >= min score, as nvBowtie's score
    limit guarantees: BowtieMapq3's tables are 11 x 11), 5 % unaligned reads"""
    rng = np.random.default_rng(seed)
    L = rng.integers(30, 251, n).astype(np.uint32)
    mn = simple_func(smin[0], smin[1], smin[2], L).astype(np.int64); perfect = L.astype(np.int64) * match
    s1 = rng.integers(mn - 3, perfect + 1); s1 = np.where(rng.random(n) < 0.1, perfect, s1)
    s2 = np.maximum(mn, s1 - rng.integers(0, 40, n)); s2 = np.where(rng.random(n) < 0.2, np.maximum(mn, s1), s2)
    s2 = np.minimum(s1, s2)
    pos1 = rng.integers(0, 1 << 30, n); pos2 = np.where(rng.random(n) < 0.4, INV, rng.integers(0, 1 << 30, n))
    z = np.zeros(n, int)
    a1 = pack_words(pos1, rng.integers(0, 50, n), s1, rng.integers(0, 2, n), z, z, z)
    a2 = pack_words(pos2, rng.integers(0, 50, n), s2, rng.integers(0, 2, n), z, z, z)
    un = rng.random(n) < 0.05                                   # unaligned reads keep what init_alignments_kernel wrote (aligner.h:323-346)
    init = pack_words(np.full(n, INV), np.full(n, 255), mn, z, z, z, z)
    a1[un] = init[un]; a2[un] = init[un]
    return dict(read_len=L, best=np.ascontiguousarray(np.stack([a1, a2])))


def mapq_pe_case(seed, n, match, smin):
    rng = np.random.default_rng(seed)
    L1 = rng.integers(30, 251, n).astype(np.uint32); L2 = rng.integers(30, 251, n).astype(np.uint32)
    m1 = rng.integers(0, 2, n)                                  # the mate the anchor's best alignment belongs to
    La = np.where(m1 == 1, L2, L1).astype(np.uint32); Lo = np.where(m1 == 1, L1, L2).astype(np.uint32)
    mna, mno = [simple_func(smin[0], smin[1], smin[2], x).astype(np.int64) for x in (La, Lo)]
    pa, po = La.astype(np.int64) * match, Lo.astype(np.int64) * match
    paired1 = rng.random(n) < 0.6; paired2 = paired1 & (rng.random(n) < 0.5)
    sa1 = rng.integers(mna, pa + 1); so1 = rng.integers(mno, po + 1)
    sa1 = np.where(rng.random(n) < 0.1, pa, sa1); so1 = np.where(rng.random(n) < 0.1, po, so1)
    sa2 = np.maximum(mna, sa1 - rng.integers(0, 30, n)); so2 = np.maximum(mno, so1 - rng.integers(0, 30, n))
    posa1 = rng.integers(0, 1 << 30, n)
    posa2 = np.where(paired2 | (rng.random(n) < 0.5), rng.integers(0, 1 << 30, n), INV)
    poso1 = np.where(paired1 | (rng.random(n) < 0.3), rng.integers(0, 1 << 30, n), INV)
    poso2 = np.where(paired2 | (rng.random(n) < 0.3), rng.integers(0, 1 << 30, n), INV)
    m2 = np.where(paired2, m1, rng.integers(0, 2, n))
    z = np.zeros(n, int)
    a1 = pack_words(posa1, rng.integers(0, 50, n), sa1, rng.integers(0, 2, n), m1, paired1, z)
    a2 = pack_words(posa2, rng.integers(0, 50, n), sa2, rng.integers(0, 2, n), m2, paired2, z)
    o1 = pack_words(poso1, rng.integers(0, 50, n), so1, rng.integers(0, 2, n), 1 - m1, paired1, z)
    o2 = pack_words(poso2, rng.integers(0, 50, n), so2, rng.integers(0, 2, n), 1 - m2, paired2, z)
    un = rng.random(n) < 0.05                                   # unaligned pairs keep what init_alignments_kernel wrote (mate in the rc slot)
    inita = pack_words(np.full(n, INV), np.full(n, 255), mna, m1, z, z, z); inito = pack_words(np.full(n, INV), np.full(n, 255), mno, 1 - m1, z, z, z)
    a1[un] = inita[un]; a2[un] = inita[un]; o1[un] = inito[un]; o2[un] = inito[un]
    return dict(L1=L1, L2=L2, a_len=La, o_len=Lo, anchor=np.ascontiguousarray(np.stack([a1, a2])), opposite=np.ascontiguousarray(np.stack([o1, o2])))


def seq_index(L):
    return np.concatenate([[0], np.cumsum(L)]).astype(np.uint32)


def ref_mapq_se(lib, version, match, smin, case):
    n = case["read_len"].size
    out = np.zeros(n, np.uint8)
    lib.ref_mapq_se(version, match, smin[0], C.c_float(smin[1]), C.c_float(smin[2]), int(match == 0), C.c_uint32(n), P(case["best"]), C.c_uint32(n), P(seq_index(case["read_len"])), P(out))
    return out


def ref_mapq_pe(lib, version, mate, match, smin, case):
    """MapqFunctorPE(mate, ..., best_anchor, best_opposite, ...) as aligner_best_approx_paired.h:297,394 construct it: mate 1 swaps
    the two slot sets inside the functor"""
    n = case["L1"].size
    out = np.zeros(n, np.uint8)
    b, bo = case["anchor"], case["opposite"]
    lib.ref_mapq_pe(version, mate, match, smin[0], C.c_float(smin[1]), C.c_float(smin[2]), int(match == 0), C.c_uint32(n), P(b), P(bo), C.c_uint32(n),
                    P(seq_index(case["L1"])), P(seq_index(case["L2"])), P(out))
    return out


def reduce_rounds(seed, n, paired, rounds=5):
    """extension results for `rounds` successive score_reduce calls over the same reads: hit positions cluster around one locus per
    read so that repeats, near-repeats (inside / outside read_len / 2 and / 4) and distinct loci all occur"""
    rng = np.random.default_rng(seed)
    L = rng.integers(40, 251, n).astype(np.uint32)
    base = rng.integers(1000, 1 << 28, n)
    out = dict(read_len=L, trys0=rng.integers(0, 4, n).astype(np.uint32), rounds=[])
    for rnd in range(rounds):
        act = np.sort(rng.choice(n, size=int(rng.integers(n // 2, n)), replace=False)).astype(np.uint32)
        cnt = rng.integers(0, 5, act.size)
        hb = np.concatenate([[0], np.cumsum(cnt)]).astype(np.uint64)
        m = int(hb[-1]); rid = np.repeat(act, cnt)
        loc = u32(base[rid] + rng.integers(0, 3, m) * 1000 + rng.integers(0, 80, m))
        rc = rng.integers(0, 2, m).astype(np.uint8); top = rng.integers(0, 2, m).astype(np.uint8)
        mn = (-0.6 - 0.6 * L[rid]).astype(np.int64)
        r = dict(active=act, hit_begin=hb, loc=loc, rc=rc, top_flag=top, score=rng.integers(mn - 5, 1).astype(np.int32),
                 n_ext=np.uint32(rng.integers(0, 6)), min_ext=np.uint32(3), max_ext=np.uint32(rng.integers(5, 12)), max_effort=np.uint32(3))
        if paired:
            oloc = u32(loc.astype(np.int64) + rng.integers(-400, 400, m))
            r.update(sink=u32(loc + rng.integers(L[rid] - 5, L[rid] + 6)), o_loc=oloc, o_sink=u32(oloc + rng.integers(30, 260, m)), o_sink2=u32(oloc + rng.integers(30, 260, m)),
                     o_score=np.where(rng.random(m) < 0.6, rng.integers(-150, 1, m), -400).astype(np.int32),
                     o_score2=np.where(rng.random(m) < 0.3, rng.integers(-150, 1, m), -400).astype(np.int32),
                     anchor=np.uint32(rnd & 1), pe_policy=np.int32(rng.integers(0, 4)), pe_unpaired=np.int32(rng.integers(0, 2)), score_limit=np.int32(-200))
        out["rounds"].append(r)
    return out


def seed_words(r):
    """packed_seed words (defs.h:171-183): rc at bit 13, top_flag at bit 14"""
    return u32((r["rc"].astype(np.uint32) << 13) | (r["top_flag"].astype(np.uint32) << 14))


def ref_reduce_round(lib, context, r, read_len, trys, best, best_o=None):
    """one launch of the reference's score_reduce_kernel / score_reduce_paired_kernel, in place; returns the erased flags"""
    n = read_len.size
    erased = np.zeros(n, np.uint8)
    idx = seq_index(read_len)
    if best_o is None:
        lib.ref_score_reduce(context, C.c_uint32(r["active"].size), P(r["active"]), P(r["hit_begin"]), P(r["loc"]), P(r["score"]), P(r["rc"]), P(r["top_flag"]), P(idx),
                             P(trys), C.c_uint32(int(r["n_ext"])), C.c_uint32(int(r["max_effort"])), C.c_uint32(int(r["min_ext"])), C.c_uint32(int(r["max_ext"])),
                             P(best), C.c_uint32(n), P(erased))
    else:
        lib.ref_score_reduce_paired(context, C.c_uint32(r["active"].size), P(r["active"]), P(r["hit_begin"]), P(r["loc"]), P(r["sink"]), P(r["score"]), P(r["rc"]), P(r["top_flag"]),
                                    P(r["o_loc"]), P(r["o_sink"]), P(r["o_sink2"]), P(r["o_score"]), P(r["o_score2"]), P(idx),
                                    C.c_uint32(int(r["anchor"])), int(r["pe_policy"]), int(r["pe_unpaired"]), C.c_int32(int(r["score_limit"])),
                                    P(trys), C.c_uint32(int(r["n_ext"])), C.c_uint32(int(r["max_effort"])), C.c_uint32(int(r["min_ext"])), C.c_uint32(int(r["max_ext"])),
                                    P(best), P(best_o), C.c_uint32(n), P(erased))
    return erased
