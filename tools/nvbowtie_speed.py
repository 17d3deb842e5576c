#!/usr/bin/env python3
"""Throughput of the reference's own nvBowtie (oracle/_ref/ref_nvBowtie, built unchanged on the drop-in layer) on a simulated input, beside
this repository's C++-driver numbers in bench.py.  GPU box only.

    python tools/nvbowtie_speed.py [--genome 4000000] [--reads 1000000] [--paired]
"""
import argparse
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome", type=int, default=4_000_000)
    ap.add_argument("--reads", type=int, default=1_000_000)
    ap.add_argument("--extra", default="")
    ap.add_argument("--wrap", default="", help="a command prefix to run nvBowtie under, e.g. 'rocprofv3 --kernel-trace --stats -d /tmp/p -o n --'")
    args = ap.parse_args()
    from nvbio_amd import io as nio
    from oracle import pyoracle as O
    rng = np.random.default_rng(1)
    tmp = tempfile.mkdtemp(prefix="nvbspeed_")
    t0 = time.time()
    text = rng.integers(0, 4, args.genome, dtype=np.uint8)
    prefix = os.path.join(tmp, "genome")
    nio.save_fmindex(prefix, O.FMIndex(text)); nio.save_fmindex(prefix, O.FMIndex(text[::-1].copy()), reverse=True)
    nio.write_wpac(prefix + ".wpac", text.size, O.pack(text, 2, True)); nio.write_bns(prefix, ["chr1"], [args.genome])
    print("index built in %.1f s" % (time.time() - t0))
    L, n = 100, args.reads
    pos = rng.integers(0, args.genome - L, n)
    idx = pos[:, None] + np.arange(L)[None, :]
    reads = text[idx]
    mut = rng.random(reads.shape) < 0.02
    reads = np.where(mut, (reads + 1) & 3, reads).astype(np.uint8)
    rc = (np.arange(n) & 1).astype(bool)
    reads[rc] = (3 - reads[rc])[:, ::-1]
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    seq = lut[reads]
    fq = os.path.join(tmp, "reads.fastq")
    t0 = time.time()
    with open(fq, "wb") as f:
        qual = b"I" * L
        for i in range(n):
            f.write(b"@r%d\n" % i); f.write(seq[i].tobytes()); f.write(b"\n+\n"); f.write(qual); f.write(b"\n")
    print("fastq written in %.1f s" % (time.time() - t0))
    sam = os.path.join(tmp, "out.sam")
    cmd = args.wrap.split() + [os.path.join(ROOT, "oracle", "_ref", "ref_nvBowtie")] + args.extra.split() + ["--file-ref", "-x", prefix, "-U", fq, "-S", sam]
    t0 = time.time()
    r = subprocess.run(cmd, capture_output=True, text=True)
    dt = time.time() - t0
    log = (r.stdout + r.stderr).replace("\r", "\n")
    print("\n".join(l for l in log.splitlines() if l.startswith("stats") or "rror" in l or "xception" in l or "warn" in l))
    print(log[-1500:])
    print("nvBowtie wall time %.2f s for %d reads (%.2f M reads/s incl. file I/O and start-up), exit %d" % (dt, n, n / dt * 1e-6, r.returncode))
    return r.returncode


if __name__ == "__main__":
    sys.exit(main())
