"""sw-benchmark's batch (20 000 x 150 bp against one 16 384-bp text) on the full-matrix kernel: python tools/full_dp_small_probe.py [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, nvbio_amd as nvb
from nvbio_amd import workloads as W
dev = "cuda"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
L, N = 150, 16384
g = torch.Generator(device=dev); g.manual_seed(8)
text = torch.randint(0, 4, (N,), dtype=torch.uint8, generator=g, device=dev)
reads = torch.randint(0, 4, (n, L), dtype=torch.uint8, generator=g, device=dev)
mp_ = nvb.PackedStringSet(W._pack_chunked(reads.reshape(-1), 4, True), 4, True, torch.arange(n, dtype=torch.int64, device=dev) * L, None, L)
wt = nvb.PackedStringSet(W._pack_chunked(text, 2, True), 2, True, torch.zeros(n, dtype=torch.int64, device=dev), None, N)
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for name, ty in (("LOCAL", nvb.LOCAL), ("SEMI_GLOBAL", nvb.SEMI_GLOBAL), ("GLOBAL", nvb.GLOBAL)):
    al = nvb.make_gotoh_aligner(ty, nvb.SimpleGotohScheme(2, -1, -2, -1), nvb.TEXT_BLOCKING)
    ms = timed(lambda: nvb.batch_alignment_score(al, mp_, wt, L, N, None))
    print("n %6d %-12s %7.2f ms %6.0f GCUPS [%s]" % (n, name, ms, n * L * N / ms / 1e6, nvb.lib().nvbio_hip_last_kernel().decode()))
