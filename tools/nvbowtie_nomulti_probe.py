#!/usr/bin/env python3
"""own driver vs the unchanged nvBowtie WITHOUT multi-hit rounds (--no-multi-hits 1 / no_multi_hits=True) on the files a
`tools/nvbowtie_3gbp.py --keep /tmp/w3g` run left: with one hit per read and round there is no hits-per-read rule reasoning with the size of the
active queue, so a difference cannot cascade from one read to the others -- what is left are the reads that differ on their own.  GPU box only."""
import sys, os, json, subprocess, time
sys.path.insert(0, "tools")
import numpy as np, torch
import nvbowtie_3gbp as T
W = "/tmp/w3g"
exe = "oracle/_ref/ref_nvBowtie"
r = subprocess.run([exe, "--no-multi-hits", "1", "--file-ref", "-x", W + "/genome", "-U", W + "/reads.fastq", "-S", W + "/ref_nm.sam"], capture_output=True, text=True, timeout=120)
# the reads back from the FASTQ file (fixed-size records)
raw = np.fromfile(W + "/reads.fastq", dtype=np.uint8).reshape(-1, 215)
lut = np.full(256, 4, np.uint8)
for c, v in zip(b"ACGT", range(4)): lut[c] = v
sym = torch.from_numpy(lut[raw[:, 11:111]]).cuda(); qual = torch.from_numpy(raw[:, 114:214] - 33).cuda()
out = {}
T.own_driver(W + "/genome", sym, qual, W + "/own_nm.sam", torch.device("cuda:0"), 1 << 20, timings=out, overrides=dict(no_multi_hits=True))
n_a, n_b, same, diffs, cats = T.compare_sam(W + "/ref_nm.sam", W + "/own_nm.sam", show=12)
out.update(exit=r.returncode, records_ref=n_a, records_own=n_b, identical=same, categories=cats, first_differences=diffs)
print(json.dumps(out, indent=1, default=str))
# ---- the differing reads on their own: a FASTQ of (up to) 300 of them through both drivers again
la = [l for l in open(W + "/ref_nm.sam", "rb").read().split(b"\n") if l and not l.startswith(b"@")]
lb = [l for l in open(W + "/own_nm.sam", "rb").read().split(b"\n") if l and not l.startswith(b"@")]
bad = [i for i, (x, y) in enumerate(zip(la, lb)) if x != y]
sub = bad[:300]
from collections import Counter
print(json.dumps({"differing_reads": len(bad), "by_batch": dict(Counter(i >> 20 for i in bad))}))
if sub:
    raw[sub].tofile(W + "/sub.fastq")
    r2 = subprocess.run([exe, "--no-multi-hits", "1", "--file-ref", "-x", W + "/genome", "-U", W + "/sub.fastq", "-S", W + "/ref_sub.sam"], capture_output=True, text=True, timeout=120)
    import align_fastq as AF
    from nvbio_amd import io as nio, aligner as A
    idx = torch.tensor(sub, device="cuda")
    res = {}
    # own driver on the subset, names kept
    data = nio.FMIndexDataDevice(W + "/genome", flags=nio.FORWARD | nio.SA, device="cuda")
    n_genome, g_words = nio.load_genome(W + "/genome")
    genome_words = torch.from_numpy(np.concatenate([g_words, np.zeros(8, np.uint32)]).view(np.int32)).cuda()
    names = ["r%08d" % i for i in sub]
    m = len(sub)
    index = torch.arange(0, (m + 1) * 100, 100, dtype=torch.int64, device="cuda")
    batch = A.ReadBatch.from_ragged(sym[idx].reshape(-1), index, qual[idx].reshape(-1))
    rr = A.best_approx(data.index(), data.rindex(), batch, genome_words, n_genome, A.Params(hits_stride=32, no_multi_hits=True), names=names, cigar_stride=64, finish=True)
    torch.cuda.synchronize()
    ref = AF.Reference(W + "/genome", n_genome, "ref")
    name_buf = np.frombuffer(("\0".join(names) + "\0").encode(), dtype=np.uint8)
    name_idx = np.arange(0, (m + 1) * 10, 10, dtype=np.uint32)
    AF.write_records_se_native(W + "/own_sub.sam", ref, (name_buf, name_idx), sym[idx].reshape(-1).cpu().numpy(), index.cpu().numpy(), qual[idx].reshape(-1).cpu().numpy(),
                               rr["best"].cpu().numpy().view(np.uint64), rr["mapq"].cpu().numpy(), rr["cigar"].cpu().numpy().view(np.uint16), rr["cigar_len"].cpu().numpy(),
                               rr["source"].cpu().numpy(), rr["mds"].cpu().numpy(), extra_flags=64)
    sa = [l for l in open(W + "/ref_sub.sam", "rb").read().split(b"\n") if l and not l.startswith(b"@")]
    sb = [l for l in open(W + "/own_sub.sam", "rb").read().split(b"\n") if l and not l.startswith(b"@")]
    cut = lambda l: [f.decode() for f in l.split(b"\t")[:9] + l.split(b"\t")[11:]]
    still = [k for k, (x, y) in enumerate(zip(sa, sb)) if x != y]
    full_vs_sub_ref = sum(1 for k, i in enumerate(sub) if la[i] != sa[k])
    full_vs_sub_own = sum(1 for k, i in enumerate(sub) if lb[i] != sb[k])
    print(json.dumps({"subset": m, "still_differ_alone": len(still), "nvbowtie_changed_its_answer_alone": full_vs_sub_ref, "own_changed_its_answer_alone": full_vs_sub_own,
                      "examples": [dict(read=sub[k], ref_alone=cut(sa[k]), own_alone=cut(sb[k]), ref_in_batch=cut(la[sub[k]]), own_in_batch=cut(lb[sub[k]])) for k in (still[:6] or list(range(min(4, m))))],
                      "own_stats": {k: v for k, v in rr["stats"].items() if k != "ms"}}, indent=1))
    import shutil
    os.makedirs("gpurun_out/sub3g", exist_ok=True)
    for f in ("sub.fastq", "ref_sub.sam", "own_sub.sam"):
        shutil.copy(W + "/" + f, "gpurun_out/sub3g/" + f)
