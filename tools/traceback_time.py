"""Banded traceback timing: band 15 LOCAL (simple scheme) and band 31 SEMI_GLOBAL in nvBowtie's quality-aware scheme (the drivers' case),
on ungapped-mostly reads (make_sw_batch: 4 % substitutions, 0.5 % indels)."""
import time

import torch

import nvbio_amd as nvb
from nvbio_amd import workloads as W

dev = "cuda"
for band, ty, scheme, n in ((15, nvb.LOCAL, nvb.SimpleGotohScheme(2, -1, -2, -1), 2_000_000), (31, nvb.SEMI_GLOBAL, nvb.SmithWatermanScoringScheme(), 2_000_000)):
    M = 100
    p, t = W.make_sw_batch(n, M, M + band, device=dev, seed=5)
    al = nvb.make_gotoh_aligner(ty, scheme)
    quals = torch.full((n * M + 8,), 30, dtype=torch.uint8, device=dev) if isinstance(scheme, nvb.SmithWatermanScoringScheme) else None
    tb = nvb.BatchedBandedAlignmentTraceback(band)
    out = dict(score=torch.empty(n, dtype=torch.int32, device=dev), sink=torch.empty((n, 2), dtype=torch.int32, device=dev),
               source=torch.empty((n, 2), dtype=torch.int32, device=dev), cigar=torch.zeros((n, 32), dtype=torch.int16, device=dev),
               cigar_len=torch.empty(n, dtype=torch.int32, device=dev))
    temp = torch.empty(tb.min_temp_storage(M, 0, n), dtype=torch.uint8, device=dev)
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.time()
        tb.enact(al, p, t, out["score"], out["sink"], out["source"], out["cigar"], out["cigar_len"], quals=quals, temp=temp)
        torch.cuda.synchronize(); dt = time.time() - t0
    cig = out["cigar"].to(torch.int32) & 0xFFFF
    gapped = ((((cig & 3) == 1) | ((cig & 3) == 2)) & (torch.arange(32, device=dev)[None, :] < out["cigar_len"][:, None])).any(1).float().mean().item()
    print("traceback %d x %d bp band %d type %d: %.2f ms  %.1f M aln/s  temp %.2f GB  gapped %.1f %%" % (n, M, band, ty, dt * 1e3, n / dt / 1e6, temp.numel() / 1e9, 100 * gapped))
