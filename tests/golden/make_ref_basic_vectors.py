"""Records outputs of REFERENCE code (oracle/_ref/libref_basic.so: nvbio/basic/popcount.h, nvbio/fmindex/bwt.h + contrib/sais.h,
nvbio/basic/priority_deque.h compiled from their own sources) as a fixture, so that the oracle and the HIP kernels can be
checked against them where /root/reference does not exist.  Run in the dev container: python tests/golden/make_ref_basic_vectors.py"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as O      # only for packing the BWT words into the interleaved layout (data plumbing)

ref = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_basic.so"))
p = lambda a: a.ctypes.data_as(C.c_void_p)
rng = np.random.default_rng(20260927)

# 1. SA / BWT / primary of a text, by the reference's gen_sa + gen_bwt_from_sa
n = 6000
text = rng.integers(0, 4, n, dtype=np.uint8)
text[1000:1400] = 0
sa = np.zeros(n + 1, np.int32); bwt = np.zeros(n + 1, np.uint8)
primary = ref.ref_gen_sa_bwt(n, p(text), p(sa), p(bwt))

# 2. rank on the interleaved layout, counted by the reference's popc_2bit / hibits_2bit.  The layout itself (words + running
#    counters) is plain data derived from that BWT.
nb = (n + 63) // 64
words = np.concatenate([O.pack(bwt[:n], 2, True, pad_words=0), np.zeros(4 * nb, np.uint32)])[:4 * nb]
bwt_occ = np.zeros(8 * nb, np.uint32)
run = np.zeros(4, np.uint32)
for k in range(nb):
    bwt_occ[8 * k: 8 * k + 4] = words[4 * k: 4 * k + 4]
    bwt_occ[8 * k + 4: 8 * k + 8] = run
    blk = bwt[64 * k: min(64 * k + 64, n)]
    run = run + np.bincount(blk, minlength=4).astype(np.uint32)
L2 = np.concatenate([[0], np.cumsum(run)]).astype(np.uint32)
idx = np.concatenate([rng.integers(0, n, 4000), [0, n - 1, 63, 64, 65, 0xFFFFFFFF]]).astype(np.uint32)
sym = rng.integers(0, 4, idx.size).astype(np.uint8)
rank = np.zeros(idx.size, np.uint32)
ref.ref_dict_rank(p(bwt_occ), p(idx), p(sym), idx.size, p(rank))

# 3. priority_deque traces (nvBowtie's hit deque): a program that never pops an empty deque, and the array after every operation
n_ops = 1200
ops = np.zeros(n_ops, np.uint8)
size = 0
for i in range(n_ops):
    u = rng.random()
    ops[i] = 0 if (u < 0.58 or size == 0) else (1 if u < 0.8 else 2)
    size += 1 if ops[i] == 0 else -1
delta = rng.integers(1, 64, n_ops).astype(np.uint64)
values = (delta << np.uint64(32)) | rng.integers(0, 1 << 31, n_ops).astype(np.uint64)
sizes, tops, bottoms = np.zeros(n_ops, np.uint32), np.zeros(n_ops, np.uint64), np.zeros(n_ops, np.uint64)
states = []
storage = np.zeros(n_ops + 1, np.uint64)
for i in range(n_ops):          # replay prefix by prefix to capture the array after each operation
    st = np.zeros(n_ops + 1, np.uint64)
    s_, t_, b_ = np.zeros(i + 1, np.uint32), np.zeros(i + 1, np.uint64), np.zeros(i + 1, np.uint64)
    f = ref.ref_priority_deque_replay(p(st), i + 1, p(ops), p(values), p(s_), p(t_), p(b_))
    sizes[i], tops[i], bottoms[i] = s_[i], t_[i], b_[i]
    states.append(st[:f].copy())
    storage, final = st, f
states = np.concatenate(states)

np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ref_basic_vectors.npz"),
                    text=text, sa=sa, bwt=bwt[:n], primary=np.uint32(primary), bwt_occ=bwt_occ, L2=L2, rank_idx=idx, rank_sym=sym, rank=rank,
                    deque_ops=ops, deque_values=values, deque_sizes=sizes, deque_tops=tops, deque_bottoms=bottoms, deque_final=storage[:final], deque_states=states)
print("wrote ref_basic_vectors.npz: n=%d primary=%d, %d rank queries, %d deque ops" % (n, primary, idx.size, n_ops))
