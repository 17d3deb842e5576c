"""Probe: the banded traceback on the C++ suite's batch (tests/cxx/nvbio_hip_test.cpp:298-308: 32768 x 150 bp, 5 % substitutions, a deletion in every
third read, Ns, some texts shorter than their pattern), band 7 LOCAL (2,-1,-1,-1), repeated; every run must equal the first."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nvbio_amd as nvb
from oracle import pyoracle as O
dev = torch.device("cuda:0")
class LCG:
    def __init__(s, v): s.s = v
    def next(s): s.s = (s.s * 1664525 + 1013904223) & 0xFFFFFFFF; return s.s
    def sym(s): return s.next() >> 30
n, M = 32768, 150
N = M + 15
rnd = LCG(7)
pats, txts = [], []
for i in range(n):
    t = [rnd.sym() for _ in range(N)]
    p = t[7:7 + M]
    for j in range(M):
        if (rnd.next() >> 16) % 100 < 5: p[j] = rnd.sym()
    if i % 3 == 0: del p[40 + i % 50: 42 + i % 50]
    if i % 97 == 0: p[i % len(p)] = 4
    if i % 211 == 0: t = t[:len(p) - 3]
    pats.append(np.array(p, np.uint8)); txts.append(np.array(t, np.uint8))
hp, ht = O.StringSet.from_lists(pats, 4, True), O.StringSet.from_lists(txts, 2, True)
p = nvb.PackedStringSet.from_host(hp.words, 4, True, hp.begin, hp.length, device=dev)
t = nvb.PackedStringSet.from_host(ht.words, 2, True, ht.begin, ht.length, device=dev)
for band, ty, scheme in ((7, nvb.LOCAL, (2, -1, -1, -1)),):
    al = nvb.make_gotoh_aligner(ty, nvb.SimpleGotohScheme(*scheme))
    tb = nvb.BatchedBandedAlignmentTraceback(band)
    need = tb.min_temp_storage(150, 165, n)
    first = None
    for rep in range(40):
        temp = torch.randint(0, 256, (need,), dtype=torch.uint8, device=dev) if rep % 2 else torch.zeros(need, dtype=torch.uint8, device=dev)
        score = torch.empty(n, dtype=torch.int32, device=dev); sink = torch.empty((n, 2), dtype=torch.int32, device=dev); src = torch.empty((n, 2), dtype=torch.int32, device=dev)
        cig = torch.zeros((n, 48), dtype=torch.int16, device=dev); cl = torch.empty(n, dtype=torch.int32, device=dev)
        tb.enact(al, p, t, score, sink, src, cig, cl, max_pattern_length=150, max_text_length=165, temp=temp)
        torch.cuda.synchronize()
        lens = cl.cpu().numpy(); cg = cig.cpu().numpy()
        mask = np.arange(48)[None, :] < lens[:, None]
        cg = np.where(mask, cg, 0)
        cur = (score.cpu().numpy(), sink.cpu().numpy(), src.cpu().numpy(), lens, cg)
        if first is None: first = cur; continue
        bad = np.nonzero((first[4] != cur[4]).any(1) | (first[3] != cur[3]) | (first[0] != cur[0]))[0]
        if bad.size: print("rep", rep, "differing jobs", bad[:10], bad.size)
    print("done", band)
