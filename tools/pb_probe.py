import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, nvbio_amd as nvb
from nvbio_amd import workloads as W
dev = "cuda"
TYPE = nvb.SEMI_GLOBAL if "semi" in sys.argv[1:] else nvb.LOCAL
g = torch.Generator(device=dev); g.manual_seed(8)
nw, wl = 1_000_000, 650
win = torch.randint(0, 4, (nw, wl), dtype=torch.uint8, generator=g, device=dev)
off = torch.randint(0, wl - 150, (nw,), generator=g, device=dev)
mate = win.gather(1, off.unsqueeze(1) + torch.arange(150, device=dev).unsqueeze(0))
mp_ = nvb.PackedStringSet(W._pack_chunked(mate.reshape(-1), 4, True), 4, True, torch.arange(nw, dtype=torch.int64, device=dev) * 150, None, 150)
wt = nvb.PackedStringSet(W._pack_chunked(win.reshape(-1), 2, True), 2, True, torch.arange(nw, dtype=torch.int64, device=dev) * wl, None, wl)
msc = torch.full((nw,), 100 if TYPE == nvb.LOCAL else -90, dtype=torch.int32, device=dev)
al = nvb.make_gotoh_aligner(TYPE, nvb.SimpleGotohScheme(2, -6, -8, -3), nvb.PATTERN_BLOCKING)
for _ in range(3):
    nvb.batch_alignment_score(al, mp_, wt, 150, wl, msc)
torch.cuda.synchronize()
