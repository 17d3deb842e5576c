import torch, time, nvbio_amd as nvb
from nvbio_amd import workloads as W
n, M, band = 2_000_000, 100, 15
p, t = W.make_sw_batch(n, M, M + band, device="cuda", seed=5)
al = nvb.make_gotoh_aligner(nvb.LOCAL, nvb.SimpleGotohScheme(2, -1, -2, -1))
tb = nvb.BatchedBandedAlignmentTraceback(band)
dev = "cuda"
out = dict(score=torch.empty(n, dtype=torch.int32, device=dev), sink=torch.empty((n, 2), dtype=torch.int32, device=dev),
           source=torch.empty((n, 2), dtype=torch.int32, device=dev), cigar=torch.zeros((n, 32), dtype=torch.int16, device=dev),
           cigar_len=torch.empty(n, dtype=torch.int32, device=dev))
temp = torch.empty(tb.min_temp_storage(M, 0, n), dtype=torch.uint8, device=dev)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    tb.enact(al, p, t, out["score"], out["sink"], out["source"], out["cigar"], out["cigar_len"], temp=temp)
    torch.cuda.synchronize(); dt = time.time() - t0
    print("traceback %d x %d bp band %d: %.2f ms  %.1f M aln/s  temp %.2f GB" % (n, M, band, dt * 1e3, n / dt / 1e6, temp.numel() / 1e9))
