"""Perf-cliff hunt: banded kernel in 32-bit arithmetic / ragged sets / 2-bit reads, full-matrix kernel across pattern lengths."""
import os, torch, numpy as np, nvbio_amd as nvb
from nvbio_amd import workloads as W
dev = "cuda"
def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
n = 4_000_000
al = nvb.make_gotoh_aligner(nvb.LOCAL, nvb.SimpleGotohScheme(2, -1, -2, -1))
for L, band in ((100, 15), (150, 31)):
    p, t = W.make_sw_batch(n, L, L + band, seed=3, device=dev)
    sc = torch.empty(n, dtype=torch.int32, device=dev); sk = torch.empty((n, 2), dtype=torch.int32, device=dev)
    b = nvb.BatchedBandedAlignmentScore(band)
    for force in ("0", "1"):
        nvb.set_test_switch("NVBIO_HIP_FORCE_32BIT", force)
        ms = timed(lambda: b.enact(al, p, t, sc, sk))
        print("banded L %3d band %2d force32=%s fixed : %6.2f ms %7.0f GCUPS [%s]" % (L, band, force, ms, n * L * band / ms / 1e6, nvb.lib().nvbio_hip_last_kernel().decode()))
    nvb.set_test_switch("NVBIO_HIP_FORCE_32BIT", "0")
    # ragged: the same strings with explicit length arrays (+ max_pattern_length hint, and without)
    pr = nvb.PackedStringSet(p.words, 4, True, p.begin, torch.full((n,), L, dtype=torch.int32, device=dev), 0)
    tr = nvb.PackedStringSet(t.words, 2, False, t.begin, torch.full((n,), L + band, dtype=torch.int32, device=dev), 0)
    for hint in (L, 0):
        ms = timed(lambda: b.enact(al, pr, tr, sc, sk, max_pattern_length=hint))
        print("banded L %3d band %2d ragged hint=%3d   : %6.2f ms %7.0f GCUPS" % (L, band, hint, ms, n * L * band / ms / 1e6))
    del p, t, pr, tr
# full matrix across pattern lengths
g = torch.Generator(device=dev); g.manual_seed(1)
ref_len, nr = 4096, 65536
ref = torch.randint(0, 4, (ref_len,), dtype=torch.uint8, generator=g, device=dev)
rt = nvb.PackedStringSet(W._pack_chunked(ref, 2, False), 2, False, torch.zeros(nr, dtype=torch.int64, device=dev), None, ref_len)
for M in (32, 64, 100, 128, 150, 192, 250, 256, 300, 512):
    st = torch.randint(0, ref_len - M, (nr,), generator=g, device=dev)
    reads = ref[st.unsqueeze(1) + torch.arange(M, device=dev).unsqueeze(0)]
    rp = nvb.PackedStringSet(W._pack_chunked(reads.reshape(-1), 4, True), 4, True, torch.arange(nr, dtype=torch.int64, device=dev) * M, None, M)
    for ty, nm in ((nvb.LOCAL, "LOCAL"), (nvb.SEMI_GLOBAL, "SEMI ")):
        a2 = nvb.make_gotoh_aligner(ty, nvb.SimpleGotohScheme(2, -1, -2, -1))
        ms = timed(lambda: nvb.batch_alignment_score(a2, rp, rt, M, ref_len))
        print("full M %3d N %d %s : %7.2f ms %7.0f GCUPS [%s]" % (M, ref_len, nm, ms, nr * M * ref_len / ms / 1e6, nvb.lib().nvbio_hip_last_kernel().decode()))
