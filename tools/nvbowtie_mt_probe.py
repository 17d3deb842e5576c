#!/usr/bin/env python3
"""Why does nvBowtie with two compute threads on one device (--device 0 --device 0) print other records than with one?  On the files a
`tools/nvbowtie_3gbp.py --keep DIR` run left: the second batch of reads (1024 K reads from read 1048576) is aligned ALONE by a fresh single-thread
run, and its records are compared with (a) the same reads' records in the full single-thread run (where the Aligner had processed a batch before),
(b) the two-thread run's.  Also runs the two-thread mode twice (is it at least repeatable?).  GPU box only."""
import json
import os
import subprocess
import sys
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def records(path):
    d = {}
    if path is None or not os.path.exists(path):
        return d
    for l in open(path, "rb").read().split(b"\n"):
        if l and not l.startswith(b"@"):
            d[l.split(b"\t", 1)[0]] = l
    return d


def main():
    W = sys.argv[1]
    extra_env = dict(kv.split("=") for kv in sys.argv[2:])
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_nvBowtie")
    rec = 215                                                          # bytes of one FASTQ record of tools/nvbowtie_3gbp.py
    b0, b1 = 1 << 20, 2 << 20
    with open(os.path.join(W, "reads.fastq"), "rb") as f:
        f.seek(b0 * rec); blob = f.read((b1 - b0) * rec)
    open(os.path.join(W, "batch1.fastq"), "wb").write(blob)
    out = {}

    def run(tag, args):
        try:
            r = subprocess.run([exe] + args + ["--file-ref", "-x", os.path.join(W, "genome"), "-S", os.path.join(W, tag + ".sam")], capture_output=True, text=True, timeout=90,
                               env=dict(os.environ, **extra_env))
            out[tag + "_exit"] = r.returncode
        except subprocess.TimeoutExpired as e:
            out[tag + "_exit"] = "HUNG (90 s)"
            txt = lambda x: x.decode(errors="replace") if isinstance(x, bytes) else (x or "")
            out[tag + "_log_tail"] = (txt(e.stdout) + txt(e.stderr)).replace("\r", "\n")[-1500:]
            return None
        return os.path.join(W, tag + ".sam")

    alone = records(run("b1_alone", ["-U", os.path.join(W, "batch1.fastq")]))
    full = records(os.path.join(W, "ref.sam"))
    mt1 = records(run("mt_a", ["--device", "0", "--device", "0", "-U", os.path.join(W, "reads.fastq")]))
    mt2 = records(run("mt_b", ["--device", "0", "--device", "0", "-U", os.path.join(W, "reads.fastq")]))
    print(json.dumps(out), file=sys.stderr, flush=True)
    own = records(os.path.join(W, "own.sam")) if os.path.exists(os.path.join(W, "own.sam")) else {}
    names = list(alone.keys())
    out["batch1_reads"] = len(names)
    out["alone_vs_full_single_thread"] = sum(1 for k in names if alone[k] != full.get(k))
    out["alone_vs_two_threads_a"] = sum(1 for k in names if alone[k] != mt1.get(k))
    out["full_vs_two_threads_a_in_batch1"] = sum(1 for k in names if full.get(k) != mt1.get(k))
    out["two_threads_a_vs_b_all_reads"] = sum(1 for k in mt1 if mt1[k] != mt2.get(k))
    if own:
        out["own_vs_alone_in_batch1"] = sum(1 for k in names if own.get(k) != alone[k].replace(b"\t64\t", b"\t64\t"))
        out["own_vs_full_in_batch1"] = sum(1 for k in names if own.get(k) != full.get(k))
    per_batch = Counter()
    for k in full:
        if full[k] != mt1.get(k):
            per_batch[int(k[1:]) >> 20] += 1
    out["full_vs_two_threads_a_by_batch"] = dict(per_batch)
    if own:
        pb = Counter()
        for k in full:
            if full[k] != own.get(k):
                pb[int(k[1:]) >> 20] += 1
        out["full_vs_own_by_batch"] = dict(pb)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
