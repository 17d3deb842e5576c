#!/usr/bin/env python3
"""banded_gotoh_score_bounded_kernel next to the plain kernel on nvBowtie-shaped jobs (100 bp reads with per-base qualities against 131-symbol
windows, band 31, end-to-end scheme): kernel time with every job exact (thresholds off: what the persistent / refill structure costs by itself)
and with thresholds drawn so that a given share of the jobs is given up at rows spread over the read.  GPU box only.
    python tools/bounded_probe.py [--jobs 4000000] [--band 31]"""
import argparse, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nvbio_amd as nvb
from nvbio_amd import workloads as W


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in e:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in e)[reps // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--jobs", type=int, default=4_000_000)
    ap.add_argument("--band", type=int, default=31)
    ap.add_argument("--len", type=int, default=100)
    ap.add_argument("--wave", action="store_true", help="instead: the one-wave-per-job anti-diagonal kernel next to the lane-per-job kernel, from 256 jobs to --jobs")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    n, L, B = a.jobs, a.len, a.band
    g = torch.Generator(device=dev); g.manual_seed(5)
    # windows: random text; reads: the window's middle with 6 % substitutions
    txt = torch.randint(0, 4, (n, L + B), dtype=torch.uint8, generator=g, device=dev)
    rd = txt[:, B // 2:B // 2 + L].clone()
    mut = torch.rand((n, L), generator=g, device=dev) < 0.06
    rd = torch.where(mut, (rd + 1 + torch.randint(0, 3, (n, L), dtype=torch.uint8, generator=g, device=dev)) & 3, rd)
    pw = W._pack_chunked(rd.reshape(-1), 4, True); tw = W._pack_chunked(txt.reshape(-1), 2, True)
    p = nvb.PackedStringSet(pw, 4, True, torch.arange(0, n * L, L, dtype=torch.int64, device=dev), None, L)
    t = nvb.PackedStringSet(tw, 2, True, torch.arange(0, n * (L + B), L + B, dtype=torch.int64, device=dev), None, L + B)
    q = torch.randint(2, 41, (n * L + 8,), dtype=torch.uint8, generator=g, device=dev)
    al = nvb.make_gotoh_aligner(nvb.SEMI_GLOBAL, nvb.SmithWatermanScoringScheme())
    s0, k0 = nvb.batch_banded_alignment_score(B, al, p, t, quals=q)
    if a.wave:
        # same jobs, the first m of them: latency of a small launch (what a tail round of the paired driver pays) and throughput of a large one
        from nvbio_amd.alignment import batch_banded_alignment_score_wave
        rows = []
        m = 256
        while m <= n:
            pm = nvb.PackedStringSet(pw, 4, True, p.begin[:m].contiguous(), None, L); tm = nvb.PackedStringSet(tw, 2, True, t.begin[:m].contiguous(), None, L + B)
            sl, kl = torch.empty(m, dtype=torch.int32, device=dev), torch.empty((m, 2), dtype=torch.int32, device=dev)
            sw, kw = torch.empty_like(sl), torch.empty_like(kl)
            lane = timed(lambda: nvb.batch_banded_alignment_score(B, al, pm, tm, quals=q, out_score=sl, out_sink=kl))
            wave = timed(lambda: batch_banded_alignment_score_wave(B, al, pm, tm, q, out_score=sw, out_sink=kw))
            rows.append(dict(jobs=m, lane_per_job_us=round(lane * 1e3, 1), wave_per_job_us=round(wave * 1e3, 1), identical=bool(torch.equal(sl, sw) and torch.equal(kl, kw)),
                             lane_GCUPS=round(m * L * B / lane / 1e6, 1), wave_GCUPS=round(m * L * B / wave / 1e6, 1)))
            m *= 4
        print(json.dumps({"band": B, "len": L, "rows": rows}))
        return
    out = {"jobs": n, "band": B, "len": L, "mean_score": float(s0.float().mean().item())}
    out["plain_ms"] = timed(lambda: nvb.batch_banded_alignment_score(B, al, p, t, quals=q, out_score=s0, out_sink=k0))
    s1, k1 = torch.empty_like(s0), torch.empty_like(k0)
    out["bounded_exact_ms"] = timed(lambda: nvb.batch_banded_alignment_score(B, al, p, t, quals=q, out_score=s1, out_sink=k1, min_score=None, n_on_device=torch.tensor([n], dtype=torch.int32, device=dev)))
    out["bounded_exact_identical"] = bool(torch.equal(s0, s1) and torch.equal(k0, k1))
    for frac in (0.5, 0.95):
        # thresholds: for `frac` of the jobs, a value the job's score falls through at a uniformly drawn fraction of its rows (score * u), else far below
        u = torch.rand(n, generator=g, device=dev)
        thr = torch.where(torch.rand(n, generator=g, device=dev) < frac, (s0.float() * u).to(torch.int32), torch.full_like(s0, -100000))
        out["bounded_%d_ms" % int(frac * 100)] = timed(lambda: nvb.batch_banded_alignment_score(B, al, p, t, quals=q, out_score=s1, out_sink=k1, min_score=thr))
        ok = s0 > thr
        out["bounded_%d_ok" % int(frac * 100)] = bool(torch.equal(s0[ok], s1[ok]) and bool((s1[~ok] <= thr[~ok]).all()))
        out["bounded_%d_given_up" % int(frac * 100)] = float(((k1 == -1).all(1)).float().mean().item())
    print(json.dumps(out))


if __name__ == "__main__":
    main()
