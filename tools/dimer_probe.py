"""Times match / locate / map_exact on the reference layout vs the line-native two-symbol index
(nvbio_amd/csrc/fmindex_dimer.h) on a synthetic genome, and checks the results are identical.
usage: python tools/dimer_probe.py [genome_symbols] [seeds] [reads]"""
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import nvbio_amd as nvb
from nvbio_amd import workloads as W
from nvbio_amd import pipeline as P


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for e0, e1 in ev:
        e0.record(); fn(); e1.record()
    torch.cuda.synchronize()
    return sum(e0.elapsed_time(e1) for e0, e1 in ev) / reps


def main():
    ng = int(float(sys.argv[1])) if len(sys.argv) > 1 else 3_000_000_000
    ns = int(float(sys.argv[2])) if len(sys.argv) > 2 else 50_000_000
    nr = int(float(sys.argv[3])) if len(sys.argv) > 3 else 10_000_000
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev); g.manual_seed(0x5EED0003)
    text = torch.randint(0, 4, (ng,), dtype=torch.uint8, generator=g, device=dev)
    t0 = time.perf_counter(); fmi = W.build_fm_index(text); torch.cuda.synchronize()
    print("index build %.2f s" % (time.perf_counter() - t0), flush=True)
    t0 = time.perf_counter(); fd = fmi.with_dimer(); torch.cuda.synchronize()
    print("dimer build %.3f s, %.2f GB" % (time.perf_counter() - t0, fd.dimer.numel() * 4 / 1e9), flush=True)
    seeds = W.make_seeds(text, ns, 22)
    r0 = nvb.match(fmi, seeds); r1 = nvb.match(fd, seeds)
    print("match identical:", bool(torch.equal(r0, r1)), flush=True)
    m0 = timed(lambda: nvb.match(fmi, seeds, out=r0)); m1 = timed(lambda: nvb.match(fd, seeds, out=r1))
    print("match  %d seeds: reference layout %.2f ms (%.2f G/s) | dimer %.2f ms (%.2f G/s)" % (ns, m0, ns / m0 / 1e6, m1, ns / m1 / 1e6), flush=True)
    fk = fd.with_ktab(12)
    r2 = nvb.match(fk, seeds)
    m2 = timed(lambda: nvb.match(fk, seeds, out=r2))
    print("match  dimer + ktab12 %.2f ms (%.2f G/s) identical %s" % (m2, ns / m2 / 1e6, bool(torch.equal(r0, r2))), flush=True)
    del fk, r2
    t0 = time.perf_counter(); ft = fd.with_trimer(); torch.cuda.synchronize()
    print("trimer build %.3f s, %.2f GB" % (time.perf_counter() - t0, ft.trimer.numel() * 4 / 1e9), flush=True)
    r3 = nvb.match(ft, seeds)
    m3 = timed(lambda: nvb.match(ft, seeds, out=r3))
    print("match  trimer %.2f ms (%.2f G/s) identical %s" % (m3, ns / m3 / 1e6, bool(torch.equal(r0, r3))), flush=True)
    fk3 = ft.with_ktab(12)
    m4 = timed(lambda: nvb.match(fk3, seeds, out=r3))
    print("match  trimer + ktab12 %.2f ms (%.2f G/s) identical %s" % (m4, ns / m4 / 1e6, bool(torch.equal(r0, r3))), flush=True)
    del fk3, r3
    ok = (r0[:, 0].to(torch.int64) & 0xFFFFFFFF) <= (r0[:, 1].to(torch.int64) & 0xFFFFFFFF)
    rows = r0[:, 0][ok].contiguous()
    p0 = nvb.locate(fmi, rows); p1 = nvb.locate(fd, rows)
    print("locate identical:", bool(torch.equal(p0, p1)), flush=True)
    l0 = timed(lambda: nvb.locate(fmi, rows, out=p0)); l1 = timed(lambda: nvb.locate(fd, rows, out=p1))
    srt, _ = torch.sort(rows.to(torch.int64) & 0xFFFFFFFF); srt = srt.to(torch.int32)
    l0s = timed(lambda: nvb.locate(fmi, srt, out=p0)); l1s = timed(lambda: nvb.locate(fd, srt, out=p1))
    nrow = rows.numel()
    print("locate %d rows: reference %.2f ms (%.2f G/s; sorted %.2f) | dimer %.2f ms (%.2f G/s; sorted %.2f)" %
          (nrow, l0, nrow / l0 / 1e6, l0s, l1, nrow / l1 / 1e6, l1s), flush=True)
    del p0, p1, r0, r1, rows, srt, seeds
    sym, pos, _ = P.make_reads(text, nr, 100, seed=0x5EED0004)
    reads, _ = P.pack_read_streams(sym)
    mp = nvb.MappingParams()
    h0, c0, q0 = nvb.map_exact(fmi, reads, mp, 100)
    h1, c1, q1 = nvb.map_exact(fd, reads, mp, 100)
    print("map_exact identical:", bool(torch.equal(h0, h1) and torch.equal(c0, c1) and torch.equal(q0, q1)), flush=True)
    h2, c2, q2 = nvb.map_exact(ft, reads, mp, 100)
    print("map_exact trimer identical:", bool(torch.equal(h0, h2) and torch.equal(c0, c2) and torch.equal(q0, q2)), flush=True)
    t0 = timed(lambda: nvb.map_exact(fmi, reads, mp, 100), 3); t1 = timed(lambda: nvb.map_exact(fd, reads, mp, 100), 3)
    t2 = timed(lambda: nvb.map_exact(ft, reads, mp, 100), 3)
    print("map_exact %d reads: reference %.2f ms | dimer %.2f ms | trimer %.2f ms" % (nr, t0, t1, t2), flush=True)


if __name__ == "__main__":
    main()
