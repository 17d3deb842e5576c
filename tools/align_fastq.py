#!/usr/bin/env python3
"""Real-data flow through the library: <prefix>.bwt/.sa (+ .wpac/.pac genome) and a FASTQ file of equal-length reads ->
SAM lines through nvBowtie's single-end best-mapping driver (nvbio_amd.aligner.best_approx = Aligner::best_approx: seeding
passes, randomized hit selection seeded by the read names, quality-aware extension, give-up counters, MAPQ, traceback).
Reads may differ in length.  A usage example, not nvBowtie's CLI: the mandatory SAM fields and the tags SamOutput writes (NM, AS, XM, XO, XG, MD from the finished
alignments; nvbio/io/output/output_sam.cpp:316-366), single reference sequence, no read groups.

    python tools/align_fastq.py <index prefix> <reads.fastq> [out.sam]"""
import sys

import numpy as np
import torch

import nvbio_amd as nvb
from nvbio_amd import io as nio, aligner as A


def cigar_string(words, length):
    """io::Cigar words are stored end-first; SAM wants them start-first"""
    ops = [(int(w) & 3, (int(w) & 0xFFFF) >> 2) for w in words[:length]][::-1]
    return "".join("%d%s" % (n, "MIDS"[t]) for t, n in ops) or "*"


def main(prefix, fastq, out=sys.stdout, device="cuda", ref_name="ref"):
    data = nio.FMIndexDataDevice(prefix, flags=nio.FORWARD | nio.SA, device=device)
    n_genome, g_words = nio.load_genome(prefix)
    genome_words = torch.from_numpy(np.concatenate([g_words, np.zeros(8, np.uint32)]).view(np.int32)).to(device)
    reads = nio.read_fastq(fastq)
    n = reads.size()
    if n == 0:
        raise SystemExit("align_fastq: no reads")
    index = np.asarray(reads.sequence_index, dtype=np.int64)
    batch = A.ReadBatch.from_ragged(torch.from_numpy(reads.symbols).to(device), torch.from_numpy(index).to(device), torch.from_numpy(reads.quals).to(device))
    params = A.Params(hits_stride=32)
    r = A.best_approx(data.index(), None, batch, genome_words, n_genome, params, names=list(reads.names), cigar_stride=64, finish=True)
    torch.cuda.synchronize()
    best = r["best"].cpu().numpy().view(np.uint64)            # finished: m_align = the traceback window's begin, m_ed, final score
    mapq, cig, clen = r["mapq"].cpu().numpy(), r["cigar"].cpu().numpy().view(np.uint16), r["cigar_len"].cpu().numpy()
    source, mds = r["source"].cpu().numpy(), r["mds"].cpu().numpy()
    out.write("@HD\tVN:1.0\tSO:unsorted\n@SQ\tSN:%s\tLN:%d\n@PG\tID:nvbio_amd\tPN:nvbio_amd\n" % (ref_name, n_genome))
    for i in range(n):
        w, pos = int(best[0, i] & 0xFFFFFFFF), int(best[0, i] >> 32)
        seq, qual = reads.symbols[index[i]:index[i + 1]], reads.quals[index[i]:index[i + 1]]
        if pos == 0xFFFFFFFF:
            out.write("%s\t4\t*\t0\t0\t*\t*\t0\t0\t%s\t%s\n" % (reads.names[i], "".join("ACGTN"[c] for c in seq), "".join(chr(int(q) + 33) for q in qual)))
            continue
        rc = (w >> 28) & 1
        score, ed = ((w >> 1) & 0x1FFFF) * (-1 if w & 1 else 1), (w >> 18) & 0x3FF
        s, q = (np.where(seq < 4, 3 - seq, 4)[::-1], qual[::-1]) if rc else (seq, qual)
        md, mm, gapo, gape = nio.sam_md_string(mds[i])
        out.write("%s\t%d\t%s\t%d\t%d\t%s\t*\t0\t0\t%s\t%s\tNM:i:%d\tAS:i:%d\tXM:i:%d\tXO:i:%d\tXG:i:%d\tMD:Z:%s\n" % (
            reads.names[i], 16 if rc else 0, ref_name, pos + int(source[i, 0]) + 1, int(mapq[i]), cigar_string(cig[i], int(clen[i])),
            "".join("ACGTN"[c] for c in s), "".join(chr(int(x) + 33) for x in q), ed, score, mm, gapo, gape, md or "*"))


if __name__ == "__main__":
    if len(sys.argv) < 3:
        raise SystemExit(__doc__)
    if len(sys.argv) > 3:
        with open(sys.argv[3], "w") as f:
            main(sys.argv[1], sys.argv[2], f)
    else:
        main(sys.argv[1], sys.argv[2])
