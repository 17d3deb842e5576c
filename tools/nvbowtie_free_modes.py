#!/usr/bin/env python3
"""Which form of nvbio_hip_device_free keeps the unchanged nvBowtie's output what it is under the blocking form?  Runs oracle/_ref/ref_nvBowtie over the
files tools/nvbowtie_3gbp.py --keep left in a work directory, once per NVBIO_HIP_FREE_MODE, and counts the records that differ from mode 1's."""
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def body(path):
    return [l for l in open(path, "rb").read().split(b"\n") if l and not l.startswith(b"@")]


def main():
    work = sys.argv[1]
    modes = [int(m) for m in sys.argv[2:]] or [1, 0, 2, 3, 4]
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_nvBowtie")
    base = None
    out = {}
    for m in modes:
        sam = os.path.join(work, "mode%d.sam" % m)
        t0 = time.time()
        try:
            r = subprocess.run([exe, "--file-ref", "-x", os.path.join(work, "genome"), "-U", os.path.join(work, "reads.fastq"), "-S", sam], capture_output=True, text=True,
                               env=dict(os.environ, NVBIO_HIP_FREE_MODE=str(m)), timeout=120)
        except subprocess.TimeoutExpired:
            out[m] = dict(timeout=True); continue
        rec = dict(exit=r.returncode, wall_s=round(time.time() - t0, 2), stderr_tail=[l for l in r.stderr.replace("\r", "\n").splitlines() if "freed twice" in l][-3:])
        if r.returncode == 0:
            b = body(sam)
            rec["md5"] = hashlib.md5(b"\n".join(b)).hexdigest()
            if base is None:
                base = b
            else:
                diff = [k for k in range(min(len(b), len(base))) if b[k] != base[k]]
                rec["differ_from_first_mode"] = len(diff)
                per = {}
                for k in diff:
                    per[k >> 20] = per.get(k >> 20, 0) + 1
                rec["per_batch"] = per
                rec["first"] = diff[:8]
        os.remove(sam) if os.path.exists(sam) else None
        out[m] = rec
        print(m, json.dumps(rec), flush=True)
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
