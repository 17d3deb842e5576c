"""Throughput of the banded score kernel across bands / schemes / read lengths (GCUPS)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, nvbio_amd as nvb
from nvbio_amd import workloads as W
dev = "cuda"
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
import sys
CASES = ((100, 31), (150, 31)) if "b31" in sys.argv[1:] else ((100, 15), (150, 15)) if "b15" in sys.argv[1:] else ((100, 7), (100, 5), (100, 3), (250, 31))
for L, band in CASES:
    n = 4_000_000
    p, t = W.make_sw_batch(n, L, L + max(band, 15), seed=3, device=dev)
    sc = torch.empty(n, dtype=torch.int32, device=dev); sk = torch.empty((n, 2), dtype=torch.int32, device=dev)
    b = nvb.BatchedBandedAlignmentScore(band)
    q = torch.randint(2, 41, (n * L + 8,), dtype=torch.uint8, device=dev)
    for name, al, kw in (("simple LOCAL", nvb.make_gotoh_aligner(nvb.LOCAL, nvb.SimpleGotohScheme(2, -1, -2, -1)), {}),
                         ("simple SEMI ", nvb.make_gotoh_aligner(nvb.SEMI_GLOBAL, nvb.SimpleGotohScheme(0, -6, -8, -3)), {}),
                         ("qual LOCAL  ", nvb.make_gotoh_aligner(nvb.LOCAL, nvb.SmithWatermanScoringScheme.local()), dict(quals=q)),
                         ("qual SEMI   ", nvb.make_gotoh_aligner(nvb.SEMI_GLOBAL, nvb.SmithWatermanScoringScheme()), dict(quals=q))):
        ms = timed(lambda: b.enact(al, p, t, sc, sk, **kw))
        print("L %3d band %2d %s : %7.2f ms  %6.1f M aln/s  %7.0f GCUPS  [%s]" % (L, band, name, ms, n / ms / 1e3, n * L * band / ms / 1e6, nvb.lib().nvbio_hip_last_kernel().decode()))
    del p, t, q
