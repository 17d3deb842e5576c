"""The paired-end driver on the extras leg's workload, with the sub-stages of its traceback stage.
usage: python tools/paired_probe.py [genome_symbols] [pairs]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nvbio_amd as nvb
from nvbio_amd import aligner as AL, pipeline as P, select as SEL, workloads as W


def main():
    ng = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
    n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 500_000
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev); g.manual_seed(0x5EED0007)
    text = torch.randint(0, 4, (ng,), dtype=torch.uint8, generator=g, device=dev)
    fmi = W.build_fm_index(text)
    rfmi = W.build_fm_index(torch.flip(text, dims=[0]))
    genome_words = W._pack_chunked(text, 2, True)
    s1, s2, _, _ = P.make_read_pairs(text, n, 150, seed=0x5EED0009)
    names = SEL.pack_names(["p%d" % i for i in range(n)], dev)
    for local in (True, False):
        prm = AL.Params(hits_stride=32, batch_size=n, **(dict(local=True, seed_len=20, seed_freq=(2, 1.0, 0.75)) if local else {}))
        for which in ("reference layout", "two-symbol index"):
            f, r = (fmi, rfmi) if which == "reference layout" else (fmi.with_dimer(), rfmi.with_dimer())
            run = lambda st=False: AL.best_approx_paired(f, r, s1, s2, genome_words, ng, prm, names=names, stage_times=st)
            run(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(); e1.record(); torch.cuda.synchronize()
            out = run(True)
            print("%s, %s: %.2f ms per %d pairs; stages %s" % ("local" if local else "end-to-end", which, e0.elapsed_time(e1), n,
                                                               {k: round(v, 2) for k, v in out["stats"]["ms"].items()}), flush=True)


if __name__ == "__main__":
    main()
