#!/usr/bin/env python3
"""Regenerates tests/golden/hit_deque_vectors.npz from the REFERENCE's own interval heap
(oracle/_ref/libref_hit_deque.so = nvbio/basic/interval_heap.h compiled where it lies, `make -C oracle ref`;
needs /root/reference, i.e. the development container).

Each case is a random program of deque operations on SeedHit words -- 0 push (with the mapper's "full: pop_bottom
first" rule), 1 pop_top, 2 pop_bottom, and what the selection stage does to a stored deque: 3 shrink the range of the hit
in one slot in place (val = slot | new size << 32; no heap operation), 4 rebuild the heap over the array as it stands
(make_interval_heap = priority_deque(seq, constructed = false), what every hits[read_id] of the selection kernels runs) --
with many equal range sizes, and the array the reference heap holds after every operation.  tests/test_select_oracle.py replays the programs through the oracle's restatement (CPU) and
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
        dirty = False
        for step in range(int(rng.integers(4, 160))):
            u = rng.random()
            op = 0 if (u < 0.50 or n == 0) else (1 if u < 0.62 else (2 if u < 0.72 else (3 if u < 0.88 else 4)))
            v = 0
            if dirty and op in (0, 1, 2):        # the heap operations expect a heap: rebuild first, as the selection kernels do
                op = 4
            if op == 0:
                if n == cap:
                    ref.ref_hit_deque_pop_bottom(a.ctypes.data_as(P), n); n -= 1
                v = (int(rng.integers(1, maxsz)) << 32) | (int(rng.integers(0, 1 << 12)) << 52) | int(rng.integers(0, 1 << 32))
                a[n] = v; n += 1
                ref.ref_hit_deque_push(a.ctypes.data_as(P), n)
            elif op == 1:
                ref.ref_hit_deque_pop_top(a.ctypes.data_as(P), n); n -= 1
            elif op == 2:
                ref.ref_hit_deque_pop_bottom(a.ctypes.data_as(P), n); n -= 1
            elif op == 3:
                slot = int(rng.integers(0, n))
                size = (int(a[slot]) >> 32) & 0xFFFFF
                size = int(rng.integers(0, size + 1)) if rng.random() < 0.7 else max(size - 1, 0)
                a[slot] = np.uint64((int(a[slot]) & ~(0xFFFFF << 32)) | (size << 32))
                v = slot | (size << 32)
                dirty = True
            else:
                ref.ref_hit_deque_make(a.ctypes.data_as(P), n)
                dirty = False
            assert dirty or ref.ref_hit_deque_is_heap(a.ctypes.data_as(P), n)
            ops.append(op); vals.append(v); caps.append(cap); sizes.append(n); states.append(a[:n].copy())
        starts.append(len(ops))
    flat = np.concatenate(states) if states else np.zeros(0, np.uint64)
    np.savez_compressed(os.path.join(HERE, "hit_deque_vectors.npz"), ops=np.array(ops, np.uint8), vals=np.array(vals, np.uint64),
                        caps=np.array(caps, np.uint32), sizes=np.array(sizes, np.uint32), states=flat, case_start=np.array(starts, np.uint32))
    print("cases", len(starts) - 1, "ops", len(ops), "state words", flat.size)


if __name__ == "__main__":
    sys.exit(main())
