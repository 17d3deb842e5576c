#!/usr/bin/env python3
"""300 reads on which the unchanged nvBowtie and this repository's driver disagree at 3 Gbp (tools/nvbowtie_3gbp_probe_ids.txt, found by
tools/nvbowtie_nomulti_probe.py; the reads differ when aligned alone too), through both drivers under option settings that switch off one
stage's policy at a time -- to find the stage where the two part.  Works on the files a `tools/nvbowtie_3gbp.py --keep /tmp/w3g` run left.  GPU box only."""
import json
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
W = sys.argv[1] if len(sys.argv) > 1 else "/tmp/w3g"


def main():
    import align_fastq as AF
    from nvbio_amd import io as nio, aligner as A
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_nvBowtie")
    ids = [int(x) for x in open(os.path.join(ROOT, "tools", "nvbowtie_3gbp_probe_ids.txt")).read().split()]
    raw = np.fromfile(W + "/reads.fastq", dtype=np.uint8).reshape(-1, 215)
    raw[ids].tofile(W + "/sub.fastq")
    lut = np.full(256, 4, np.uint8)
    for c, v in zip(b"ACGT", range(4)):
        lut[c] = v
    sym = torch.from_numpy(lut[raw[ids][:, 11:111]]).cuda(); qual = torch.from_numpy(raw[ids][:, 114:214] - 33).cuda()
    data = nio.FMIndexDataDevice(W + "/genome", flags=nio.FORWARD | nio.SA, device="cuda")
    n_genome, g_words = nio.load_genome(W + "/genome")
    genome_words = torch.from_numpy(np.concatenate([g_words, np.zeros(8, np.uint32)]).view(np.int32)).cuda()
    ref = AF.Reference(W + "/genome", n_genome, "ref")
    names = ["r%08d" % i for i in ids]
    m = len(ids)
    index = torch.arange(0, (m + 1) * 100, 100, dtype=torch.int64, device="cuda")
    name_buf = np.frombuffer(("\0".join(names) + "\0").encode(), dtype=np.uint8)
    name_idx = np.arange(0, (m + 1) * 10, 10, dtype=np.uint32)
    cut = lambda l: [f.decode() for f in l.split(b"\t")[:9] + l.split(b"\t")[11:]]
    variants = [("no_multi_hits", ["--no-multi-hits", "1"], dict(no_multi_hits=True)),
                ("no_multi_hits+no_rand", ["--no-multi-hits", "1", "--no-rand"], dict(no_multi_hits=True, randomized=False)),
                ("no_multi_hits+no_reseed", ["--no-multi-hits", "1", "--max-reseed", "0"], dict(no_multi_hits=True, max_reseed=0)),
                ("no_multi_hits+exhaustive", ["--no-multi-hits", "1", "--max-effort", "2000", "--max-ext", "4000"], dict(no_multi_hits=True, max_effort=2000, max_ext=4000)),
                ("no_multi_hits+no_reseed+exhaustive", ["--no-multi-hits", "1", "--max-reseed", "0", "--max-effort", "2000", "--max-ext", "4000"],
                 dict(no_multi_hits=True, max_reseed=0, max_effort=2000, max_ext=4000)),
                ("no_multi_hits+seed_len_32", ["--no-multi-hits", "1", "-L", "32"], dict(no_multi_hits=True, seed_len=32)),
                ("no_multi_hits+max_hits_4", ["--no-multi-hits", "1", "--max-hits", "4"], dict(no_multi_hits=True, max_hits=4))]
    out = {}
    for tag, args, own in variants:
        try:
            r = subprocess.run([exe] + args + ["--file-ref", "-x", W + "/genome", "-U", W + "/sub.fastq", "-S", W + "/ref_sub.sam"], capture_output=True, text=True, timeout=120)
        except subprocess.TimeoutExpired:
            out[tag] = "nvBowtie hung"; continue
        if r.returncode != 0:
            out[tag] = "nvBowtie exit %d" % r.returncode; continue
        batch = A.ReadBatch.from_ragged(sym.reshape(-1), index, qual.reshape(-1))
        rr = A.best_approx(data.index(), data.rindex(), batch, genome_words, n_genome, A.Params(hits_stride=32, **own), names=names, cigar_stride=64, finish=True)
        torch.cuda.synchronize()
        AF.write_records_se_native(W + "/own_sub.sam", ref, (name_buf, name_idx), sym.reshape(-1).cpu().numpy(), index.cpu().numpy(), qual.reshape(-1).cpu().numpy(),
                                   rr["best"].cpu().numpy().view(np.uint64), rr["mapq"].cpu().numpy(), rr["cigar"].cpu().numpy().view(np.uint16), rr["cigar_len"].cpu().numpy(),
                                   rr["source"].cpu().numpy(), rr["mds"].cpu().numpy(), extra_flags=64)
        sa = [l for l in open(W + "/ref_sub.sam", "rb").read().split(b"\n") if l and not l.startswith(b"@")]
        sb = [l for l in open(W + "/own_sub.sam", "rb").read().split(b"\n") if l and not l.startswith(b"@")]
        bad = [k for k, (x, y) in enumerate(zip(sa, sb)) if x != y]
        worse_ref = sum(1 for k in bad if len(sa[k].split(b"\t")) > 12 and len(sb[k].split(b"\t")) > 12 and int(sa[k].split(b"\t")[12][5:]) < int(sb[k].split(b"\t")[12][5:]))
        worse_own = sum(1 for k in bad if len(sa[k].split(b"\t")) > 12 and len(sb[k].split(b"\t")) > 12 and int(sa[k].split(b"\t")[12][5:]) > int(sb[k].split(b"\t")[12][5:]))
        out[tag] = dict(differ=len(bad), of=m, nvbowtie_score_lower=worse_ref, own_score_lower=worse_own,
                        own_stats={k: v for k, v in rr["stats"].items() if k != "ms"},
                        examples=[dict(ref=cut(sa[k]), own=cut(sb[k])) for k in bad[:3]])
        print(tag, out[tag]["differ"], file=sys.stderr, flush=True)
    print(json.dumps(out, indent=1, default=str))


if __name__ == "__main__":
    main()
