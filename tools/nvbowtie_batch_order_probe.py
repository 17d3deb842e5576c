#!/usr/bin/env python3
"""Does a read's record depend on which batches the same nvBowtie Aligner processed BEFORE its batch?  The single-thread application is run over the
reads file of a `tools/nvbowtie_3gbp.py --keep DIR` run and over the same file with its batches in reverse order (and, as a control, twice over each):
every read sits in a batch of the same composition in both, only the history of the Aligner object differs.  In the two-thread mode
(`--device 0 --device 0`) which batches an Aligner sees is decided by timing, so any such dependence shows up there as run-to-run differences.
GPU box only."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REC = 215                      # bytes per FASTQ record written by nvbowtie_3gbp.write_fastq


def records(path):
    out = {}
    for l in open(path, "rb").read().split(b"\n"):
        if l and not l.startswith(b"@"):
            out[l.split(b"\t", 1)[0]] = l
    return out


def main():
    W = sys.argv[1]
    batch_k = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_nvBowtie")
    raw = open(os.path.join(W, "reads.fastq"), "rb").read()
    n = len(raw) // REC
    per = batch_k * 1024
    batches = [raw[b * per * REC: min(n, (b + 1) * per) * REC] for b in range((n + per - 1) // per)]
    full = [b for b in batches if len(b) == per * REC]                 # whole batches only: their composition survives the reordering
    files = {"forward": b"".join(full), "reverse": b"".join(reversed(full)), "rotated": b"".join(full[len(full) // 2:] + full[:len(full) // 2])}
    res = {}
    for tag, data in files.items():
        fq = os.path.join(W, "order_%s.fastq" % tag)
        open(fq, "wb").write(data)
        for rep in (0, 1):
            sam = os.path.join(W, "order_%s_%d.sam" % (tag, rep))
            r = subprocess.run([exe, "--batch-size", str(batch_k), "--file-ref", "-x", os.path.join(W, "genome"), "-U", fq, "-S", sam], capture_output=True, text=True, timeout=120)
            res[(tag, rep)] = records(sam) if r.returncode == 0 else None
    out = {"reads": len(full) * per, "batches": len(full), "batch_reads": per}
    base = res[("forward", 0)]
    for key, rec in res.items():
        if rec is None:
            out["%s_%d" % key] = "failed"; continue
        diff = [k for k in base if rec.get(k) != base[k]]
        per_batch = {}
        for k in diff:
            b = int(k[1:]) // per
            per_batch[b] = per_batch.get(b, 0) + 1
        out["%s_%d" % key] = {"records_that_differ_from_forward_0": len(diff), "per_original_batch": per_batch,
                              "examples": [[base[k].decode().split("\t")[:9] + base[k].decode().split("\t")[11:13], rec[k].decode().split("\t")[:9] + rec[k].decode().split("\t")[11:13]] for k in diff[:3]]}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
