"""Full-matrix scoring of LONG patterns (the one-job-per-wave kernel: 8 or 16 rows per lane): python tools/full_dp_long_probe.py [semi|global]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, nvbio_amd as nvb
from nvbio_amd import workloads as W
dev = "cuda"
TYPE = nvb.SEMI_GLOBAL if "semi" in sys.argv[1:] else nvb.GLOBAL if "global" in sys.argv[1:] else nvb.LOCAL
def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
g = torch.Generator(device=dev); g.manual_seed(8)
for nw, L, wl in ((200_000, 500, 1200), (100_000, 1000, 2400)):
    win = torch.randint(0, 4, (nw, wl), dtype=torch.uint8, generator=g, device=dev)
    off = torch.randint(0, wl - L, (nw,), generator=g, device=dev)
    mate = win.gather(1, off.unsqueeze(1) + torch.arange(L, device=dev).unsqueeze(0))
    mp_ = nvb.PackedStringSet(W._pack_chunked(mate.reshape(-1), 4, True), 4, True, torch.arange(nw, dtype=torch.int64, device=dev) * L, None, L)
    wt = nvb.PackedStringSet(W._pack_chunked(win.reshape(-1), 2, True), 2, True, torch.arange(nw, dtype=torch.int64, device=dev) * wl, None, wl)
    msc = torch.full((nw,), 100 if TYPE == nvb.LOCAL else -300, dtype=torch.int32, device=dev)
    for algo in (nvb.TEXT_BLOCKING, nvb.PATTERN_BLOCKING):
        al = nvb.make_gotoh_aligner(TYPE, nvb.SimpleGotohScheme(2, -1, -2, -1), algo)
        for ms_t, nm in ((None, "no min_score"), (msc, "min_score")):
            ms = timed(lambda: nvb.batch_alignment_score(al, mp_, wt, L, wl, ms_t))
            print("n %7d M %4d N %4d algo %d %-13s: %7.2f ms %6.0f GCUPS [%s]" % (nw, L, wl, algo, nm, ms, nw * L * wl / ms / 1e6, nvb.lib().nvbio_hip_last_kernel().decode()))
    del win, mate, mp_, wt
