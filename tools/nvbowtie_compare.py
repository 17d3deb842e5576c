#!/usr/bin/env python3
"""Run the reference's own nvBowtie (oracle/_ref/ref_nvBowtie: its application sources compiled unchanged on the drop-in layer,
tools/nvbowtie_tu_check.py --link) and this repository's from-scratch drivers (tools/align_fastq.py over the C-ABI) on the same simulated
input, and compare their SAM records field by field.  GPU box only.

    python tools/nvbowtie_compare.py [--mode se|local|all|paired] [--reads N] [--seed S] [--indels RATE]
"""
import argparse
import io as _io
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))


def write_reference(tmp, rng, n_genome, seqs, repeats=0.0):
    """repeats: that fraction of the genome is overwritten with copies of a few 1-2 kbp families, 1 % diverged from their source -- reads
    from there have several good placements (second-best scores, MAPQ below 42, randomized choice among equals)"""
    from nvbio_amd import io as nio
    from oracle import pyoracle as O
    text = rng.integers(0, 4, n_genome, dtype=np.uint8)
    if repeats > 0.0:
        families = [rng.integers(0, 4, int(rng.integers(1000, 2000)), dtype=np.uint8) for _ in range(6)]
        filled = 0
        while filled < repeats * n_genome:
            f = families[int(rng.integers(0, len(families)))]
            c = f.copy(); mut = rng.random(c.size) < 0.01; c[mut] = (c[mut] + 1) & 3
            at = int(rng.integers(0, n_genome - c.size))
            # (no copy across a sequence boundary: which hit nvBowtie's --all drops for a seed straddling two sequences depends on what its
            # radix sort left in a buffer it has already handed on, aligner_all.h:460-520 -- not something two implementations can agree on)
            bounds = np.cumsum([0] + [l for _, l in seqs])
            if any(at < b < at + c.size for b in bounds[1:-1]) and not os.environ.get("NVBOWTIE_COMPARE_STRADDLE"):
                continue                      # (NVBOWTIE_COMPARE_STRADDLE=1 lets copies cross: since nvbio_hip_sort_hits_pingpong the drivers replay that buffer)
            text[at:at + c.size] = c; filled += c.size
    prefix = os.path.join(tmp, "genome")
    nio.save_fmindex(prefix, O.FMIndex(text))
    nio.save_fmindex(prefix, O.FMIndex(text[::-1].copy()), reverse=True)
    nio.write_wpac(prefix + ".wpac", text.size, O.pack(text, 2, True))
    nio.write_bns(prefix, [n for n, _ in seqs], [l for _, l in seqs])
    return prefix, text


def mutate(rng, r, rate):
    mut = rng.random(r.size) < rate
    r = r.copy(); r[mut] = (r[mut] + 1) & 3
    return r


def write_fastq(path, reads, tag, qual="I", rng=None):
    """qual: one character for every base, or "random" (with rng): phred 2 .. 40 drawn per base"""
    with open(path, "w") as f:
        for i, r in enumerate(reads):
            q = qual * len(r) if qual != "random" else "".join(chr(33 + int(v)) for v in rng.integers(2, 41, len(r)))
            f.write("@%s%d\n%s\n+\n%s\n" % (tag, i, "".join("ACGTN"[c] for c in r), q))


def records(text):
    return [ln.rstrip("\n").split("\t") for ln in text.splitlines() if ln and not ln.startswith("@")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="se", choices=["se", "local", "all", "paired"])
    ap.add_argument("--reads", type=int, default=4000)
    ap.add_argument("--seed", type=int, default=5)
    ap.add_argument("--indels", type=float, default=0.2)
    ap.add_argument("--show", type=int, default=4)
    ap.add_argument("--len", type=int, default=100, help="read length (single-end modes)")
    ap.add_argument("--ns", type=float, default=0.0, help="fraction of read bases replaced by N (single-end modes)")
    ap.add_argument("--quals", default="I", help="one quality character for all bases, or 'random'")
    ap.add_argument("--repeats", type=float, default=0.0, help="fraction of the genome covered by diverged copies of a few repeat families")
    ap.add_argument("--mixed", action="store_true", help="paired mode: mates of different lengths (70 .. 130 bp each)")
    ap.add_argument("--extra", default="", help="further nvBowtie options, e.g. '-N 1 -L 18'")
    ap.add_argument("--own", default="", help="the same settings for this repository's driver: comma-separated Params fields, e.g. 'allow_sub=1,seed_len=18'")
    args = ap.parse_args()
    same, n_ref, n_own = compare(args)
    return 0 if same == max(n_ref, n_own) and n_ref else 1


def two_threads(args, repeats=2):
    """nvBowtie with two compute threads on the one GPU (`--device 0 --device 0`) against its own single-thread run on the same files, read by read
    -> (records of the single-thread run, [records of each two-thread run that equal them])"""
    cmd, _, _, sam = prepare(args)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        print((r.stdout + r.stderr)[-3000:]); return 0, []
    key = lambda a: (a[0], int(a[1]) & 0xC0)
    single = sorted(records(open(sam).read()), key=key)
    same = []
    for k in range(repeats):
        r = subprocess.run(cmd[:1] + ["--device", "0", "--device", "0"] + cmd[1:], capture_output=True, text=True, timeout=150)
        if r.returncode != 0:
            print((r.stdout + r.stderr)[-3000:]); same.append(-1); continue
        two = sorted(records(open(sam).read()), key=key)
        same.append(sum(1 for a, b in zip(single, two) if a == b) if len(two) == len(single) else -len(two))
        shown = 0
        for a, b in zip(single, two):
            if a != b and shown < args.show:
                shown += 1
                print("  run %d:\n    one thread : %s\n    two threads: %s" % (k, a[:9] + a[11:], b[:9] + b[11:]))
    return len(single), same


def prepare(args):
    """writes the reference files and the reads of a case -> (nvBowtie command, this repository's driver as a callable, its output buffer, nvBowtie's SAM path)"""
    import torch
    import align_fastq
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_nvBowtie")
    rng = np.random.default_rng(args.seed)
    tmp = tempfile.mkdtemp(prefix="nvbcmp_")
    overrides = {}
    for kv in filter(None, getattr(args, "own", "").split(",")):
        k, v = kv.split("=")
        overrides[k] = (v == "True") if v in ("True", "False") else (int(v) if v.lstrip("-").isdigit() else v)
    extra = getattr(args, "extra", "").split()
    n_genome, L, n = 200_000, (getattr(args, "len", 100) if args.mode != "paired" else 100), args.reads
    prefix, text = write_reference(tmp, rng, n_genome, [("chrA", 120_000), ("chrB", 80_000)], getattr(args, "repeats", 0.0))
    if args.mode == "all":                                   # some repeats, so that reads have several placements
        pass
    dev = torch.device("cuda:0")
    sam = os.path.join(tmp, "ref.sam")
    buf = _io.StringIO()
    if args.mode == "paired":
        frag = rng.integers(200, 400, n)
        pos = rng.integers(0, 120_000 - 420, n) + np.where(rng.random(n) < 0.4, 120_000, 0) * 0
        if getattr(args, "mixed", False):
            # mates of their own lengths (70 .. 130 bp each): mate 1 from the fragment's left end, mate 2 from its right end
            l1, l2 = rng.integers(70, 131, n), rng.integers(70, 131, n)
            m1 = [mutate(rng, text[p:p + a], 0.02) for p, a in zip(pos, l1)]
            m2 = [mutate(rng, (3 - text[p + f - b:p + f])[::-1], 0.02) for p, f, b in zip(pos, frag, l2)]
        else:
            m1 = [mutate(rng, text[p:p + L], 0.02) for p in pos]
            m2 = [mutate(rng, (3 - text[p + f - L:p + f])[::-1], 0.02) for p, f in zip(pos, frag)]
        f1, f2 = os.path.join(tmp, "m1.fastq"), os.path.join(tmp, "m2.fastq")
        q = getattr(args, "quals", "I")
        write_fastq(f1, m1, "pair", q, rng); write_fastq(f2, m2, "pair", q, rng)
        cmd = [exe] + extra + ["--file-ref", "-x", prefix, "-1", f1, "-2", f2, "-S", sam]
        own = lambda: align_fastq.main_paired(prefix, f1, f2, buf, device=dev, **overrides)
    else:
        pos = rng.integers(0, n_genome - L - 4, n)
        pos = np.where((pos < 120_000) & (pos + L + 4 > 120_000), pos - L - 4, pos)
        reads = []
        for i, p in enumerate(pos):
            r = text[p:p + L].copy()
            if rng.random() < args.indels:
                k, g = int(rng.integers(3 * L // 10, 7 * L // 10)), int(rng.integers(1, 3))
                r = np.concatenate([r[:k], rng.integers(0, 4, g).astype(np.uint8), r[k:]])[:L] if rng.random() < 0.5 else np.concatenate([text[p:p + k], text[p + k + g:p + L + g]])
            r = mutate(rng, r, 0.03)
            r = (3 - r)[::-1] if i % 2 else r
            if getattr(args, "ns", 0.0) > 0.0:
                r = np.where(rng.random(r.size) < args.ns, 4, r).astype(np.uint8)
            reads.append(r)
        fq = os.path.join(tmp, "reads.fastq")
        write_fastq(fq, reads, "read", getattr(args, "quals", "I"), rng)
        # (mode flags go first: nvBowtie reads argv[i + 1] after an option it does not know, nvBowtie.cpp:343)
        cmd = [exe] + (["--local"] if args.mode == "local" else ["--all"] if args.mode == "all" else []) + extra + ["--file-ref", "-x", prefix, "-U", fq, "-S", sam]
        if args.mode == "all":
            own = lambda: align_fastq.main_all(prefix, fq, buf, device=dev, **overrides)
        elif args.mode == "local":
            own = lambda: align_fastq.main(prefix, fq, buf, device=dev, local=True, **overrides)
        else:
            own = lambda: align_fastq.main(prefix, fq, buf, device=dev, **overrides)
    cmd += os.environ.get("NVBOWTIE_EXTRA_ARGS", "").split()
    return cmd, own, buf, sam


def compare(args):
    """-> (identical records, reference records, own records)"""
    cmd, own, buf, sam = prepare(args)
    import align_fastq
    extra = getattr(args, "extra", "").split()
    r = subprocess.run(cmd, capture_output=True, text=True)
    if os.environ.get("NVBOWTIE_EXTRA_ARGS"):          # (a debug build's device-side prints: everything that is not a log line)
        print("\n".join(ln for ln in (r.stdout + r.stderr).replace("\r", "\n").splitlines()
                        if ln.strip() and not ln.startswith(("stats", "verbose", "info", "visible", "debug", "warning", "[0]")))[-20000:])
    if r.returncode != 0:
        print((r.stdout + r.stderr)[-3000:]); return 0, 0, 0
    ref = records(open(sam).read())
    own()
    mine = records(buf.getvalue())
    if os.environ.get("OWN_STATS") and hasattr(align_fastq, "last_stats"):
        print("own driver stats:", align_fastq.last_stats)
    print("mode %s: reference %d records, own %d records" % (args.mode, len(ref), len(mine)))
    if args.mode == "all":
        key = lambda a: (a[0], a[2], int(a[3]), int(a[1]) & 16)
        ref.sort(key=key); mine.sort(key=key)
    if args.mode == "all":
        # (every placement of a read is a record: compare the two multisets, a lone extra record must not shift the rest)
        from collections import Counter
        ca, cb = Counter(tuple([a[0], str(int(a[1]) & ~64)] + list(a[2:])) for a in ref), Counter(tuple(b) for b in mine)      # (SamOutput sets READ_1 for single-end reads too)
        only_ref, only_own = list((ca - cb).elements()), list((cb - ca).elements())
        for rec in only_ref[:args.show]:
            print("  only the reference:", list(rec[:9]) + list(rec[11:]))
        for rec in only_own[:args.show]:
            print("  only this repository:", list(rec[:9]) + list(rec[11:]))
        same = sum((ca & cb).values())
        print("identical records: %d of %d" % (same, max(len(ref), len(mine))))
        return same, len(ref), len(mine)
    same, shown = 0, 0
    for a, b in zip(ref, mine):
        a = list(a)
        if args.mode != "paired":
            a[1] = str(int(a[1]) & ~64)
        if a == b:
            same += 1
        elif shown < args.show:
            shown += 1
            d = [i for i in range(min(len(a), len(b))) if a[i] != b[i]]
            print("  differ in fields", d, "\n    ref:", a[:9] + a[11:], "\n    own:", b[:9] + b[11:])
            if args.mode == "paired":
                k = int(a[0][4:])
                print("    simulated: mate 1 at %d, fragment %d" % (pos[k] + 1, frag[k]))
    print("identical records: %d of %d" % (same, max(len(ref), len(mine))))
    return same, len(ref), len(mine)


if __name__ == "__main__":
    sys.exit(main())
