"""Records outputs of REFERENCE code -- oracle/_ref/libref_policy.so: nvBowtie's mapq.h (BowtieMapq2 / 3 behind MapqFunctorSE / PE),
reduce_inl.h (score_reduce_kernel, score_reduce_paired_kernel with both contexts), aligner.h's init_alignments_kernel, and
io::Alignment's own constructor, compiled by oracle/build_ref_policy.py -- as a fixture, so that the oracle and the HIP kernels can be
checked against them where /root/reference does not exist.  Run in the dev container: python tests/golden/make_ref_policy_vectors.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes as C

import ref_policy_cases as K

lib = K.ref_lib()
out = {}

# 1. MAPQ: every scheme x both calculators, single-end and paired-end (both mates' functors)
N = 700
for si, (match, smin) in enumerate(K.SCHEMES):
    se = K.mapq_se_case(1000 + si, N, match, smin)
    pe = K.mapq_pe_case(2000 + si, N, match, smin)
    for k, v in se.items():
        out["mapq_se%d_%s" % (si, k)] = v
    for k, v in pe.items():
        out["mapq_pe%d_%s" % (si, k)] = v
    for version in (2, 3):
        out["mapq_se%d_v%d" % (si, version)] = K.ref_mapq_se(lib, version, match, smin, se)
        for mate in (0, 1):
            out["mapq_pe%d_v%d_mate%d" % (si, version, mate)] = K.ref_mapq_pe(lib, version, mate, match, smin, pe)

# 2. init_alignments_kernel + five rounds of score_reduce_kernel / score_reduce_paired_kernel, exact and best-approx contexts
SMIN = (0, -0.6, -0.6)
for paired in (0, 1):
    for context in (0, 1):
        tag = "red%s_c%d" % ("pe" if paired else "se", context)
        n = 900
        rr = K.reduce_rounds(3000 + 10 * paired + context, n, bool(paired))
        L = rr["read_len"]; idx = K.seq_index(L)
        best = np.zeros((2, n), np.uint64); best_o = np.zeros((2, n), np.uint64) if paired else None
        lib.ref_init_alignments(SMIN[0], C.c_float(SMIN[1]), C.c_float(SMIN[2]), C.c_uint32(n), K.P(idx), C.c_uint32(0), K.P(best), C.c_uint32(n))
        out[tag + "_read_len"] = L; out[tag + "_trys0"] = rr["trys0"]; out[tag + "_init"] = best.copy()
        if paired:
            lib.ref_init_alignments(SMIN[0], C.c_float(SMIN[1]), C.c_float(SMIN[2]), C.c_uint32(n), K.P(idx), C.c_uint32(1), K.P(best_o), C.c_uint32(n))
            out[tag + "_init_o"] = best_o.copy()
        trys = rr["trys0"].copy()
        for i, r in enumerate(rr["rounds"]):
            erased = K.ref_reduce_round(lib, context, r, L, trys, best, best_o)
            for k, v in r.items():
                out["%s_r%d_%s" % (tag, i, k)] = np.asarray(v)
            out["%s_r%d_best" % (tag, i)] = best.copy(); out["%s_r%d_trys" % (tag, i)] = trys.copy(); out["%s_r%d_erased" % (tag, i)] = erased
            if paired:
                out["%s_r%d_best_o" % (tag, i)] = best_o.copy()

# 3. io::Alignment's constructor / accessors on random fields
rng = np.random.default_rng(4000)
m = 400
f = dict(pos=np.where(rng.random(m) < 0.1, K.INV, rng.integers(0, 1 << 32, m)).astype(np.uint32), ed=rng.integers(0, 1024, m).astype(np.uint32),
         score=rng.integers(-(1 << 17) + 1, 1 << 17, m).astype(np.int32), rc=rng.integers(0, 2, m).astype(np.uint32), mate=rng.integers(0, 2, m).astype(np.uint32),
         paired=rng.integers(0, 2, m).astype(np.uint32), disc=rng.integers(0, 2, m).astype(np.uint32))
words = np.array([lib.ref_alignment_pack(C.c_uint32(int(f["pos"][i])), C.c_uint32(int(f["ed"][i])), C.c_int32(int(f["score"][i])), C.c_uint32(int(f["rc"][i])),
                                         C.c_uint32(int(f["mate"][i])), int(f["paired"][i]), int(f["disc"][i])) for i in range(m)], dtype=np.uint64)
acc = np.zeros((m, 10), np.int32)
for i in range(m):
    lib.ref_alignment_unpack(C.c_uint64(int(words[i])), K.P(acc[i]))
for k, v in f.items():
    out["aln_" + k] = v
out["aln_words"] = words; out["aln_accessors"] = acc
out["aln_invalid"] = np.array([lib.ref_alignment_invalid()], dtype=np.uint64)

path = os.path.join(ROOT, "tests", "golden", "ref_policy_vectors.npz")
np.savez_compressed(path, **out)
print(path, os.path.getsize(path), "bytes,", len(out), "arrays")
