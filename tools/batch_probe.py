#!/usr/bin/env python3
"""Batch probe: BASELINE config 4's batches through the C++ single-end driver with 1..3 host threads / HIP streams per device, on the lean
(line-native) and the HBM-rich index flavours.  The command rocprofv3 traces / counter passes of the end-to-end stages are taken under
(`rocprofv3 --kernel-trace --stats -- python tools/batch_probe.py ...`).  (Round 3's tools/cosched_probe.py, minus the CU-mask / grid-limit /
seeding-token knobs that were removed from the library.)

    python tools/batch_probe.py [--genome 3e9] [--reads 10000000] [--batches 6] [--workers 1,2]

Prints one JSON line: per index flavour and worker count the wall time of all batches, M reads/s, and whether every batch's bests / MAPQs
equal the serial run's."""
import argparse
import ctypes as C
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nvbio_amd as nvb                                   # noqa: E402
from nvbio_amd import workloads as W, pipeline as P, aligner as AL, select as SEL   # noqa: E402
import bench                                              # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome", type=float, default=3.0e9)
    ap.add_argument("--reads", type=int, default=10_000_000)
    ap.add_argument("--batches", type=int, default=6)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--workers", default="1,2", help="host threads / streams sharing the device, one run per entry")
    ap.add_argument("--indices", default="line_native,hbm_rich")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    ng, n, L = int(a.genome), a.reads, 100
    g = torch.Generator(device=dev)
    g.manual_seed(0x5EED0003)
    text = torch.randint(0, 4, (ng,), dtype=torch.uint8, generator=g, device=dev)
    fmi = W.build_fm_index(text)
    genome_words = W._pack_chunked(text, 2, True)
    batches = []
    for b in range(a.batches):
        sym, _, _ = P.make_reads(text, n, L, seed=0x5EED0040 + b)
        batches.append(P.pack_read_streams(sym))
        del sym
    del text
    torch.cuda.empty_cache()
    names = SEL.pack_names(["r%d" % i for i in range(n)], dev)
    prm = AL.Params(hits_stride=16, batch_size=n)
    scheme = nvb.SmithWatermanScoringScheme()
    sp = bench._shim_params(prm, scheme)
    shim = C.CDLL(os.path.join(ROOT, "tests", "cxx", "libaligner_shim.so"))
    quals = torch.full((2 * n * L + 8,), 30, dtype=torch.uint8, device=dev)
    arena, nidx = names
    vp = lambda t: C.c_void_p(t.data_ptr())
    ptrs = lambda ts: (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    out = {"genome_symbols": ng, "reads_per_batch": n, "batches": a.batches}
    for flavour in a.indices.split(","):
        idx = fmi.with_dimer()
        if flavour == "hbm_rich":
            idx = idx.with_ktab(15 if ng > (1 << 28) else 12).with_dense_ssa(1)
        if flavour == "hbm_rich_k16":                        # the match range of every 16-mer (34 GB)
            idx = idx.with_ktab(16).with_dense_ssa(1)
        if flavour == "hbm_rich_k16_trimer":
            idx = idx.with_trimer().with_ktab(16).with_dense_ssa(1)
        if flavour == "hbm_rich_trimer":                     # + the three-symbol rank arrays (32 GB at 3 Gbp)
            idx = idx.with_trimer().with_ktab(15 if ng > (1 << 28) else 12).with_dense_ssa(1)
        fs = idx.struct()
        best = [torch.zeros((2, n), dtype=torch.int64, device=dev) for _ in range(a.batches)]
        mapq = [torch.zeros(n, dtype=torch.uint8, device=dev) for _ in range(a.batches)]
        ref = None
        res = {}
        for cfg in a.workers.split(","):
            workers = int(cfg)
            for t in best:
                t.zero_()
            ms = (C.c_double * 1)()
            torch.cuda.synchronize()
            rc = shim.nvbio_aligner_best_approx_pipelined(
                C.byref(fs), None, C.c_uint32(n), C.c_uint32(L), C.c_uint32(a.batches),
                ptrs([b[0].words for b in batches]), C.c_uint64(batches[0][0].words.numel()), ptrs([b[0].begin for b in batches]),
                ptrs([b[1] for b in batches]), C.c_uint64(batches[0][1].numel()), vp(quals), C.c_uint64(quals.numel()), vp(arena), vp(nidx),
                vp(genome_words), C.c_uint64(genome_words.numel()), C.c_uint32(ng), C.byref(sp),
                C.c_uint32(workers), C.c_uint32(a.reps), ms, ptrs(best), ptrs(mapq))
            torch.cuda.synchronize()
            if rc != 0:
                res[cfg] = {"error": rc}
                continue
            if ref is None:
                ref = ([t.clone() for t in best], [t.clone() for t in mapq])
            same = all(torch.equal(x, y) for x, y in zip(best, ref[0])) and all(torch.equal(x, y) for x, y in zip(mapq, ref[1]))
            res[cfg] = {"workers": workers, "ms_total": ms[0], "Mreads_per_s": n * a.batches / ms[0] / 1e3, "identical_to_serial": bool(same)}
            sys.stderr.write("%s %s %s\n" % (flavour, cfg, json.dumps(res[cfg])))
        out[flavour] = res
        del idx, best, mapq, ref
        torch.cuda.empty_cache()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
