#!/usr/bin/env python3
"""Find the reads on which oracle/_ref/ref_fmmap (examples/fmmap/fmmap.cu on the drop-in layer) and the CPU oracle's restatement of its pipeline
disagree, by bisection over FASTQ subsets (the program prints only an aligned percentage).  GPU box only; a debugging aid for
tests/test_ref_tests_gpu.py::test_reference_fmmap_runs_and_matches_the_oracle."""
import os
import re
import subprocess
import sys
import tempfile
import pathlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import test_ref_tests_gpu as T
    from oracle import pyoracle as O
    tmp = pathlib.Path(tempfile.mkdtemp(prefix="fmmapdbg_"))
    rng = np.random.default_rng(41)
    n_genome, n, L = 200_000, 4000, 100
    prefix, text = T._write_reference(tmp, rng, n_genome, [("chrA", 120_000), ("chrB", 80_000)])
    pos = rng.integers(300, n_genome - L - 300, n)
    reads = []
    for i, q in enumerate(pos):
        r = text[q:q + L].copy()
        if i % 3:
            m = rng.random(L) < 0.01; r[m] = (r[m] + 1) & 3
        if i % 7 == 0:
            r[int(rng.integers(0, L))] = 4
        reads.append((3 - r)[::-1] if i % 2 and r.max() < 4 else r)
    host = O.FMIndex(text)

    def oracle_best(ids):
        both = []
        for i in ids:
            r_ = reads[i]; both.append(r_); both.append(np.where(r_ < 4, 3 - r_, 4)[::-1].astype(np.uint8))
        seeds, owner, offs = [], [], []
        for sid, s in enumerate(both):
            for b in range(0, L - 22 + 1, 10):
                seeds.append(s[b:b + 22]); owner.append(sid); offs.append(b)
        ranges = host.match(O.StringSet.from_lists(seeds, 4, True))
        pats, txts, who, info = [], [], [], []
        for k, (lo, hi) in enumerate(ranges):
            if lo > hi:
                continue
            for tp in host.locate(np.arange(lo, hi + 1, dtype=np.uint32)):
                diag = int(tp) - offs[k]
                gb = diag - 15 if diag > 15 else 0
                ge = min(gb + L + 31, n_genome)
                pats.append(both[owner[k]]); txts.append(text[gb:ge]); who.append(owner[k] // 2); info.append((owner[k], offs[k], int(tp), gb, ge))
        best = np.full(len(ids), -32768, np.int64)
        if pats:
            score, _ = O.batch_banded_myers_score(31, O.SEMI_GLOBAL, 5, O.StringSet.from_lists(pats, 4, True), O.StringSet.from_lists(txts, 2, True), sink_bits=16)
            np.maximum.at(best, np.array(who), score)
        else:
            score = np.zeros(0)
        return best, info, score

    def program(ids):
        fq = str(tmp / "sub.fastq")
        with open(fq, "w") as f:
            for i in ids:
                f.write("@read%d\n%s\n+\n%s\n" % (i, "".join("ACGTN"[c] for c in reads[i]), "I" * L))
        r = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "ref_fmmap"), prefix, fq], capture_output=True, text=True, timeout=600)
        out = (r.stdout + r.stderr).replace("\r", "\n")
        m = re.findall(r"aligned\s+([0-9.]+) % reads", out)
        return int(round(float(m[-1]) * len(ids) / 100.0)), out

    ids = list(range(n))
    got, _ = program(ids)
    want = int((oracle_best(ids)[0] >= -20).sum())
    print("all reads: program %d aligned, oracle %d" % (got, want))
    while len(ids) > 1 and got != want:
        half = len(ids) // 2
        a = ids[:half]
        ga, _ = program(a)
        wa = int((oracle_best(a)[0] >= -20).sum())
        if ga != wa:
            ids, got, want = a, ga, wa
        else:
            ids = ids[half:]
            got, _ = program(ids)
            want = int((oracle_best(ids)[0] >= -20).sum())
        print("  %d reads: program %d, oracle %d" % (len(ids), got, want))
    if got != want:
        i = ids[0]
        best, info, score = oracle_best([i])
        print("read", i, "pos", int(pos[i]), "rc" if (i % 2 and reads[i].max() < 4) else "fw", "".join("ACGTN"[c] for c in reads[i]))
        print("oracle best", best, "hits (string, seed offset, text pos, window):", info, "scores", score)
        _, out = program([i])
        print(out[-1500:])


if __name__ == "__main__":
    main()
