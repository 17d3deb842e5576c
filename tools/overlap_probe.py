"""Config 4's batches through the single-end driver one after the other, and two at a time on two HIP streams (one host thread each,
as the reference runs one host thread per device): does a fabric-bound stage of one batch overlap the issue-bound stages of another?
usage: python tools/overlap_probe.py [genome_symbols] [reads_per_batch] [batches]"""
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nvbio_amd as nvb
from nvbio_amd import aligner as AL, pipeline as P, select as SEL, workloads as W


def main():
    ng = int(float(sys.argv[1])) if len(sys.argv) > 1 else 3_000_000_000
    n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 10_000_000
    nb = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev); g.manual_seed(0x5EED0003)
    text = torch.randint(0, 4, (ng,), dtype=torch.uint8, generator=g, device=dev)
    idx = W.build_fm_index(text).with_dimer()
    gw = W._pack_chunked(text, 2, True)
    names = SEL.pack_names(["r%d" % i for i in range(n)], dev)
    prm = AL.Params(hits_stride=16, batch_size=n)
    batches = []
    for b in range(nb):
        sym, pos, _ = P.make_reads(text, n, 100, seed=0x5EED0040 + b)
        batches.append((sym, P.pack_read_streams(sym)))
    run = lambda b: AL.best_approx(idx, None, batches[b][0], gw, ng, prm, names=names, packed=batches[b][1])
    ref = [run(b)["best"].clone() for b in range(nb)]           # warm-up + the answers
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in range(nb):
        run(b)
    torch.cuda.synchronize()
    serial = time.perf_counter() - t0
    out = [None] * nb

    def worker(k, streams):
        with torch.cuda.stream(streams[k]):
            for b in range(k, nb, len(streams)):
                out[b] = run(b)["best"]
            streams[k].synchronize()

    for nt in (2, 3):
        streams = [torch.cuda.Stream() for _ in range(nt)]
        for rep in range(2):                                     # the first pass grows each stream's allocator pool
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            th = [threading.Thread(target=worker, args=(k, streams)) for k in range(nt)]
            [t.start() for t in th]; [t.join() for t in th]
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        same = all(torch.equal(out[b], ref[b]) for b in range(nb))
        print("%d batches of %d reads: serial %.1f ms (%.1f M reads/s) | %d streams %.1f ms (%.1f M reads/s) identical %s" %
              (nb, n, serial * 1e3, nb * n / serial / 1e6, nt, dt * 1e3, nb * n / dt / 1e6, same), flush=True)


if __name__ == "__main__":
    main()
