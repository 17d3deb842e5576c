#!/usr/bin/env python3
"""config 5's per-GPU-share leg alone (bench.config5_share_leg) on a smaller genome: python tools/share_probe.py [genome] [pairs_total] [pairs_per_batch]"""
import json
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nvbio_amd import workloads as W, aligner as AL, select as SEL   # noqa: E402
import bench                                                          # noqa: E402

ng = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
total = int(float(sys.argv[2])) if len(sys.argv) > 2 else 4_000_000
npairs = int(float(sys.argv[3])) if len(sys.argv) > 3 else 500_000
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(0x5EED0003)
text = torch.randint(0, 4, (ng,), dtype=torch.uint8, generator=g, device=dev)
fmi = W.build_fm_index(text)
genome_words = W._pack_chunked(text, 2, True)
pnames = SEL.pack_names(["p%d" % i for i in range(npairs)], dev)
prm5 = AL.Params(hits_stride=32, batch_size=npairs, local=True, seed_len=20, seed_freq=(2, 1.0, 0.75))
a = types.SimpleNamespace(share_pairs=total)
print(json.dumps(bench.config5_share_leg(a, dev, fmi, text, genome_words, ng, pnames, prm5, npairs)))
