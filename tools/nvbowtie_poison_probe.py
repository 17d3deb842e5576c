#!/usr/bin/env python3
"""Does the unchanged nvBowtie (or the drop-in layer under it) read device storage it never wrote?  The single-thread application is run over the files
of a `tools/nvbowtie_3gbp.py --keep DIR` run with every new device block filled with a different byte (NVBIO_HIP_POISON_ALLOC); records that change
with the byte come from such a read.  GPU box only."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def records(path):
    return [l for l in open(path, "rb").read().split(b"\n") if l and not l.startswith(b"@")]


def main():
    W = sys.argv[1]
    extra = sys.argv[2].split() if len(sys.argv) > 2 and sys.argv[2] else []
    cases = [dict(), dict(NVBIO_HIP_POISON_ALLOC="0"), dict(NVBIO_HIP_POISON_ALLOC="255"), dict(NVBIO_HIP_POISON_ALLOC="165"), dict(NVBIO_HIP_POISON_ALLOC="1")]
    for spec in sys.argv[3:]:                      # further cases: BYTE:MIN:MAX
        b, lo, hi = spec.split(":")
        cases.append(dict(NVBIO_HIP_POISON_ALLOC=b, NVBIO_HIP_POISON_MIN=lo, NVBIO_HIP_POISON_MAX=hi))
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_nvBowtie")
    base, out = None, {}
    for env in cases:
        tag = "poison_" + "_".join(env.get(k, "-") for k in ("NVBIO_HIP_POISON_ALLOC", "NVBIO_HIP_POISON_MIN", "NVBIO_HIP_POISON_MAX")) if env else "none"
        sam = os.path.join(W, tag + ".sam")
        try:
            r = subprocess.run([exe] + extra + ["--file-ref", "-x", os.path.join(W, "genome"), "-U", os.path.join(W, "reads.fastq"), "-S", sam], capture_output=True, text=True,
                               timeout=150, env=dict(os.environ, **env))
        except subprocess.TimeoutExpired:
            out[tag] = "HUNG"; continue
        if r.returncode != 0:
            out[tag] = dict(exit=r.returncode, tail=(r.stdout + r.stderr).replace("\r", "\n")[-600:]); continue
        rec = records(sam)
        os.remove(sam)
        if base is None:
            base = rec; out[tag] = dict(records=len(rec)); continue
        diff = [k for k in range(min(len(rec), len(base))) if rec[k] != base[k]]
        out[tag] = dict(records=len(rec), differ_from_unpoisoned=len(diff), first=diff[:6],
                        examples=[[base[k].decode().split("\t")[:9] + base[k].decode().split("\t")[11:14], rec[k].decode().split("\t")[:9] + rec[k].decode().split("\t")[11:14]] for k in diff[:2]])
        print(tag, json.dumps(out[tag])[:400], file=sys.stderr, flush=True)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
