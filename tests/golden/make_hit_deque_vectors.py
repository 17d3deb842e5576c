#!/usr/bin/env python3
"""Regenerates tests/golden/hit_deque_vectors.npz from the REFERENCE's own interval heap
(oracle/_ref/libref_hit_deque.so = nvbio/basic/interval_heap.h compiled where it lies, `make -C oracle ref`;
needs /root/reference, i.e. the development container).

Each case is a random program of deque operations on SeedHit words -- push (with the mapper's "full: pop_bottom
first" rule), pop_top, pop_bottom -- with many equal range sizes, and the array the reference heap holds after
every operation.  tests/test_select_oracle.py replays the programs through the oracle's restatement (CPU) and
tests/test_select_gpu.py through the device one."""
import ctypes
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def main():
    ref = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_hit_deque.so"))
    P = ctypes.POINTER(ctypes.c_uint64)
    rng = np.random.default_rng(0xDE0E)
    ops, vals, caps, states, sizes, starts = [], [], [], [], [], [0]
    for case in range(400):
        cap = int(rng.integers(1, 48))
        maxsz = int(rng.choice([2, 3, 6, 40, 1 << 19]))
        a = np.zeros(64, np.uint64)
        n = 0
        for step in range(int(rng.integers(4, 160))):
            u = rng.random()
            op = 0 if (u < 0.62 or n == 0) else (1 if u < 0.82 else 2)
            v = 0
            if op == 0:
                if n == cap:
                    ref.ref_hit_deque_pop_bottom(a.ctypes.data_as(P), n); n -= 1
                v = (int(rng.integers(1, maxsz)) << 32) | (int(rng.integers(0, 1 << 12)) << 52) | int(rng.integers(0, 1 << 32))
                a[n] = v; n += 1
                ref.ref_hit_deque_push(a.ctypes.data_as(P), n)
            elif op == 1:
                ref.ref_hit_deque_pop_top(a.ctypes.data_as(P), n); n -= 1
            else:
                ref.ref_hit_deque_pop_bottom(a.ctypes.data_as(P), n); n -= 1
            assert ref.ref_hit_deque_is_heap(a.ctypes.data_as(P), n)
            ops.append(op); vals.append(v); caps.append(cap); sizes.append(n); states.append(a[:n].copy())
        starts.append(len(ops))
    flat = np.concatenate(states) if states else np.zeros(0, np.uint64)
    np.savez_compressed(os.path.join(HERE, "hit_deque_vectors.npz"), ops=np.array(ops, np.uint8), vals=np.array(vals, np.uint64),
                        caps=np.array(caps, np.uint32), sizes=np.array(sizes, np.uint32), states=flat, case_start=np.array(starts, np.uint32))
    print("cases", len(starts) - 1, "ops", len(ops), "state words", flat.size)


if __name__ == "__main__":
    sys.exit(main())
