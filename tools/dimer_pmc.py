"""A short match-only run for PMC passes: 3 Gbp index, 20 M seeds, two launches per index flavour
(reference layout, then the two-symbol index).  The dispatches appear in that order in the counter CSV."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nvbio_amd as nvb
from nvbio_amd import workloads as W

ng = int(float(os.environ.get("GENOME", "3e9")))
ns = int(float(os.environ.get("SEEDS", "2e7")))
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(0x5EED0003)
text = torch.randint(0, 4, (ng,), dtype=torch.uint8, generator=g, device=dev)
fmi = W.build_fm_index(text)
fd = fmi.with_dimer()
seeds = W.make_seeds(text, ns, 22)
out = nvb.match(fmi, seeds)
flavours = (fmi, fmi, fd, fd)
if os.environ.get("TRIMER"):
    ft = fd.with_trimer()
    flavours = (fd, fd, ft, ft)
for f in flavours:
    nvb.match(f, seeds, out=out)
torch.cuda.synchronize()
rows = out[:, 0].contiguous()
for f in (fmi, fd):
    nvb.locate(f, rows)
torch.cuda.synchronize()
print("done")
if os.environ.get("MAP"):
    from nvbio_amd import pipeline as P
    nr = int(float(os.environ.get("READS", "4e6")))
    sym, pos, _ = P.make_reads(text, nr, 100, seed=0x5EED0004)
    reads, _ = P.pack_read_streams(sym)
    mp = nvb.MappingParams()
    for f in (fmi, fd, fd):
        nvb.map_exact(f, reads, mp, 100)
    torch.cuda.synchronize()
    print("map done")
