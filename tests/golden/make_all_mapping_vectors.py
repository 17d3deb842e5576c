"""Writes tests/golden/all_mapping_vectors.npz: the output of the numpy all-mapping driver over the oracle (tests/oracle_driver.all_mapping,
the restatement of nvBowtie/bowtie2/cuda/aligner_all.h) on a seeded case, so that the CPU suite notices any change in the checker
that the GPU parity tests compare the HIP path with.  Inputs are regenerated from the seed by the test (same generator as here).

    python tests/golden/make_all_mapping_vectors.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import nvbio_amd as nvb                                    # noqa: E402
from nvbio_amd import aligner as A, workloads as W          # noqa: E402
from oracle import pyoracle as O                            # noqa: E402
from tests import oracle_driver as OD                       # noqa: E402


def case(seed=20260927):
    rng = np.random.default_rng(np.random.SeedSequence(seed))      # (a SeedSequence, not an int: the fuzz campaigns' seed shift must not move a fixture's inputs)
    text = rng.integers(0, 4, 1 << 14, dtype=np.uint8)
    text[3000:3300] = np.tile(np.array([0, 1, 3], dtype=np.uint8), 100)
    for k in range(3):
        text[6000 + 2500 * k: 6200 + 2500 * k] = text[1000:1200]
    reads = []
    for i in range(150):
        L = int(rng.integers(40, 121))
        p = int(rng.integers(1000, 1200 - 40)) if i % 4 == 0 else int(rng.integers(3000, 3200)) if i % 4 == 1 else int(rng.integers(0, text.size - L))
        r = text[p:p + L].copy()
        for j in rng.integers(0, L, i % 3):
            r[j] = (r[j] + 1) & 3
        if i % 7 == 0:
            r = np.delete(r, int(rng.integers(5, L - 5)))
        if i % 2:
            r = (3 - r)[::-1].copy()
        reads.append(r)
    quals = [rng.integers(2, 42, r.size).astype(np.uint8) for r in reads]
    return text, reads, quals


def run(text, reads, quals, batch_size):
    host, rhost = O.FMIndex(text), O.FMIndex(text[::-1].copy())
    gw = W._pack_chunked(torch.from_numpy(text), 2, True).numpy().view(np.uint32)
    params = A.Params(batch_size=batch_size)
    return OD.all_mapping(host, rhost, reads, gw, text.size, params, nvb.SmithWatermanScoringScheme(), 2, read_quals=quals, cigar_stride=64,
                          sequence_index=[0, 5000, text.size])


if __name__ == "__main__":
    text, reads, quals = case()
    out = {}
    for bs in (1 << 20, 257):
        e = run(text, reads, quals, bs)
        m = e["read_id"].size
        out.update({"read_id_%d" % bs: e["read_id"], "scored_%d" % bs: e["alignments_scored"], "finished_%d" % bs: e["alignments"],
                    "cigar_len_%d" % bs: e["tb"]["cigar_len"], "cigar_%d" % bs: e["tb"]["cigar"][:m, :12], "mds_len_%d" % bs: e["mds_len"],
                    "stats_%d" % bs: np.array([e["stats"]["hits"], e["stats"]["ranges"], e["stats"]["unique"]], np.int64)})
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "all_mapping_vectors.npz"), **out)
    print({k: v.shape for k, v in out.items()})
