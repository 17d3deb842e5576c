#!/usr/bin/env python3
"""Real-data flow through the library: <prefix>.bwt/.sa (+ .wpac/.pac genome) and a FASTQ file of equal-length reads ->
SAM lines through nvBowtie's single-end best-mapping driver (nvbio_amd.aligner.best_approx = Aligner::best_approx: seeding
passes, randomized hit selection seeded by the read names, quality-aware extension, give-up counters, MAPQ, traceback).
Reads may differ in length.  A usage example, not nvBowtie's CLI: the mandatory SAM fields and the tags SamOutput writes (NM, AS, XM, XO, XG, MD from the finished
alignments; nvbio/io/output/output_sam.cpp:316-366); the reference's sequences come from <prefix>.ann / .amb when present (else one
sequence), no read groups.

    python tools/align_fastq.py <index prefix> <reads.fastq> [out.sam | out.bam]
    python tools/align_fastq.py --all <index prefix> <reads.fastq> [out.sam | out.bam]     (every alignment of every read, Aligner::all)
    python tools/align_fastq.py <index prefix> <mates1.fastq> <mates2.fastq> <out.sam | out.bam>"""
import sys

import numpy as np
import torch

import nvbio_amd as nvb
from nvbio_amd import io as nio, aligner as A

last_stats = None


def _rname_pos(ref, pos):
    """(RNAME, 1-based POS) of a genome coordinate"""
    name, p = ref.locate(pos)
    return name, p + 1


def cigar_ref_length(words, length):
    """reference symbols an alignment spans: its M and D operations"""
    return sum((int(w) & 0xFFFF) >> 2 for w in words[:length] if (int(w) & 3) in (0, 2))


def cigar_string(words, length):
    """io::Cigar words are stored end-first; SAM wants them start-first"""
    ops = [(int(w) & 3, (int(w) & 0xFFFF) >> 2) for w in words[:length]][::-1]
    return "".join("%d%s" % (n, "MIDS"[t]) for t, n in ops) or "*"


class Reference:
    """Sequence names and offsets of the reference: <prefix>.ann / .amb when they exist (BWA-style, io.read_bns), else one sequence `ref_name`."""

    def __init__(self, prefix, n_genome, ref_name):
        import os
        self.bns = nio.read_bns(prefix) if os.path.exists(prefix + ".ann") and os.path.exists(prefix + ".amb") else None
        self.names = self.bns.names if self.bns else [ref_name]
        self.index = self.bns.sequence_index() if self.bns else [0, n_genome]

    def header(self):
        return "@HD\tVN:1.0\tSO:unsorted\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % (nm, self.index[k + 1] - self.index[k]) for k, nm in enumerate(self.names)) + \
               "@PG\tID:nvbio_amd\tPN:nvbio_amd\n"

    def bridges(self, pos, ref_len):
        """an alignment that runs over the end of its reference sequence: SamOutput flags it UNMAPPED with mapping quality 0 and still
        prints it in full (output_sam.cpp:454-462)"""
        import bisect
        k = bisect.bisect_right(list(self.index), pos) - 1
        return pos + ref_len > int(self.index[k + 1])

    def locate(self, pos):
        """genome coordinate -> (RNAME, 0-based coordinate inside that sequence)"""
        if self.bns is None:
            return self.names[0], pos
        k, p = self.bns.locate(pos)
        return self.names[k], p


def main(prefix, fastq, out=sys.stdout, device="cuda", ref_name="ref", **param_overrides):
    # (the 1-mismatch mappers search the reverse index too, mapping_inl.h:128-220: -N 1 needs <prefix>.rbwt)
    data = nio.FMIndexDataDevice(prefix, flags=nio.FORWARD | nio.SA | (nio.REVERSE if param_overrides.get("allow_sub") else 0), device=device)
    n_genome, g_words = nio.load_genome(prefix)
    genome_words = torch.from_numpy(np.concatenate([g_words, np.zeros(8, np.uint32)]).view(np.int32)).to(device)
    reads = nio.read_fastq(fastq)
    n = reads.size()
    if n == 0:
        raise SystemExit("align_fastq: no reads")
    index = np.asarray(reads.sequence_index, dtype=np.int64)
    batch = A.ReadBatch.from_ragged(torch.from_numpy(reads.symbols).to(device), torch.from_numpy(index).to(device), torch.from_numpy(reads.quals).to(device))
    params = A.Params(hits_stride=32, **param_overrides)        # e.g. local=True = nvBowtie --local
    r = A.best_approx(data.index(), data.rindex(), batch, genome_words, n_genome, params, names=list(reads.names), cigar_stride=64, finish=True)
    torch.cuda.synchronize()
    global last_stats
    last_stats = dict((k, v) for k, v in r["stats"].items() if k != "ms")          # seeding passes, queue sizes, extension rounds
    best = r["best"].cpu().numpy().view(np.uint64)            # finished: m_align = the traceback window's begin, m_ed, final score
    mapq, cig, clen = r["mapq"].cpu().numpy(), r["cigar"].cpu().numpy().view(np.uint16), r["cigar_len"].cpu().numpy()
    source, mds = r["source"].cpu().numpy(), r["mds"].cpu().numpy()
    ref = Reference(prefix, n_genome, ref_name)
    out.write(ref.header())
    write_records_se(out, ref, reads.names, reads.symbols, index, reads.quals, best, mapq, cig, clen, source, mds)


def write_records_se(out, ref, names, symbols, index, quals, best, mapq, cig, clen, source, mds, extra_flags=0):
    """One SAM record per read from the arrays Aligner::best_approx (finish = True) returns (the Python spelling; write_records_se_native
    is the C++ host layer's, include/nvbio_hip/sam.h)"""
    for i in range(len(index) - 1):
        w, pos = int(best[0, i] & 0xFFFFFFFF), int(best[0, i] >> 32)
        seq, qual = symbols[index[i]:index[i + 1]], quals[index[i]:index[i + 1]]
        if pos == 0xFFFFFFFF:
            out.write("%s\t4\t*\t0\t0\t*\t*\t0\t0\t%s\t%s\n" % (names[i], "".join("ACGTN"[c] for c in seq), "".join(chr(int(q) + 33) for q in qual)))
            continue
        rc = (w >> 28) & 1
        score, ed = ((w >> 1) & 0x1FFFF) * (-1 if w & 1 else 1), (w >> 18) & 0x3FF
        s, q = (np.where(seq < 4, 3 - seq, 4)[::-1], qual[::-1]) if rc else (seq, qual)
        md, mm, gapo, gape = nio.sam_md_string(mds[i])
        over = ref.bridges(pos + int(source[i, 0]), cigar_ref_length(cig[i], int(clen[i])))
        out.write("%s\t%d\t%s\t%d\t%d\t%s\t*\t0\t0\t%s\t%s\tNM:i:%d\tAS:i:%d\tXM:i:%d\tXO:i:%d\tXG:i:%d\tMD:Z:%s\n" % (
            names[i], (16 if rc else 0) | (4 if over else 0) | extra_flags, *_rname_pos(ref, pos + int(source[i, 0])), 0 if over else int(mapq[i]), cigar_string(cig[i], int(clen[i])),
            "".join("ACGTN"[c] for c in s), "".join(chr(int(x) + 33) for x in q), ed, score, mm, gapo, gape, md or "*"))


def write_records_se_native(path, ref, names, symbols, index, quals, best, mapq, cig, clen, source, mds, extra_flags=0, append=False, header=True):
    """The same records through the C++ host layer's writer (include/nvbio_hip/sam.h: all host threads format, written in read order) --
    what a run of millions of reads uses.  `names`: a list of str, or (bytes buffer of 0-terminated names, uint32 offsets)."""
    import ctypes as C
    import os
    lib = C.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "cxx", "libaligner_shim.so"))
    if isinstance(names, tuple):
        name_buf, name_idx = names
    else:
        enc = [nm.encode() + b"\0" for nm in names]
        name_idx = np.zeros(len(enc) + 1, np.uint32); name_idx[1:] = np.cumsum([len(e) for e in enc])
        name_buf = np.frombuffer(b"".join(enc), dtype=np.uint8)
    arr = lambda a, t: np.ascontiguousarray(a, dtype=t)
    symbols, quals, index = arr(symbols, np.uint8), arr(quals, np.uint8), arr(index, np.uint64)
    best, mapq, cig, clen = arr(best[0] if np.ndim(best) == 2 else best, np.uint64), arr(mapq, np.uint8), arr(cig, np.uint16), arr(clen, np.uint32)
    source, mds, name_buf, name_idx = arr(source, np.uint32), arr(mds, np.uint8), arr(name_buf, np.uint8), arr(name_idx, np.uint32)
    seq_names = (C.c_char_p * len(ref.names))(*[nm.encode() for nm in ref.names])
    seq_index = arr(ref.index, np.uint64)
    pv = lambda a: a.ctypes.data_as(C.c_void_p)
    n = index.size - 1
    rc = lib.nvbio_write_sam_se(path.encode(), C.c_int(1 if append else 0), C.c_int(1 if header else 0), C.c_uint32(extra_flags), C.c_uint32(n), pv(name_buf), pv(name_idx),
                                pv(symbols), pv(index), pv(quals), pv(best), pv(mapq), pv(cig), C.c_uint32(cig.shape[1]), pv(clen), pv(source), pv(mds), C.c_uint32(mds.shape[1]),
                                C.c_uint32(len(ref.names)), seq_names, pv(seq_index))
    if rc != 0:
        raise IOError("nvbio_write_sam_se(%s) failed: %d" % (path, rc))


def main_all(prefix, fastq, out=sys.stdout, device="cuda", ref_name="ref", **param_overrides):
    """All-mapping flow (nvBowtie --all = Aligner::all): one SAM record per accepted alignment of every read (MAPQ 255, as the reference fills
    it, aligner_all.h:82), no record for reads without one.  Needs the reverse index too when one-mismatch seeds are asked for."""
    # (the 1-mismatch mappers search the reverse index too, mapping_inl.h:128-220: -N 1 needs <prefix>.rbwt)
    data = nio.FMIndexDataDevice(prefix, flags=nio.FORWARD | nio.SA | (nio.REVERSE if param_overrides.get("allow_sub") else 0), device=device)
    n_genome, g_words = nio.load_genome(prefix)
    genome_words = torch.from_numpy(np.concatenate([g_words, np.zeros(8, np.uint32)]).view(np.int32)).to(device)
    reads = nio.read_fastq(fastq)
    if reads.size() == 0:
        raise SystemExit("align_fastq: no reads")
    index = np.asarray(reads.sequence_index, dtype=np.int64)
    batch = A.ReadBatch.from_ragged(torch.from_numpy(reads.symbols).to(device), torch.from_numpy(index).to(device), torch.from_numpy(reads.quals).to(device))
    ref = Reference(prefix, n_genome, ref_name)
    r = A.all_mapping(data.index(), data.rindex(), batch, genome_words, n_genome, A.Params(hits_stride=32, **param_overrides), cigar_stride=64,
                      sequence_index=ref.index)                       # seeds straddling two sequences are dropped (mark_straddling)
    torch.cuda.synchronize()
    out.write(ref.header())
    m = int(r["read_id"].numel())
    if m == 0:
        return
    rid, aln = r["read_id"].cpu().numpy(), r["alignments"].cpu().numpy().view(np.uint64)
    cig, clen, source, mds = r["cigar"].cpu().numpy().view(np.uint16), r["cigar_len"].cpu().numpy(), r["source"].cpu().numpy(), r["mds"].cpu().numpy()
    for k in range(m):
        i = int(rid[k])
        w, pos = int(aln[k] & 0xFFFFFFFF), int(aln[k] >> 32)
        seq, qual = reads.symbols[index[i]:index[i + 1]], reads.quals[index[i]:index[i + 1]]
        rc = (w >> 28) & 1
        score, ed = ((w >> 1) & 0x1FFFF) * (-1 if w & 1 else 1), (w >> 18) & 0x3FF
        s, q = (np.where(seq < 4, 3 - seq, 4)[::-1], qual[::-1]) if rc else (seq, qual)
        md, mm, gapo, gape = nio.sam_md_string(mds[k])
        over = ref.bridges(pos + int(source[k, 0]), cigar_ref_length(cig[k], int(clen[k])))
        out.write("%s\t%d\t%s\t%d\t%d\t%s\t*\t0\t0\t%s\t%s\tNM:i:%d\tAS:i:%d\tXM:i:%d\tXO:i:%d\tXG:i:%d\tMD:Z:%s\n" % (
            reads.names[i], (16 if rc else 0) | (4 if over else 0), *_rname_pos(ref, pos + int(source[k, 0])), 0 if over else 255, cigar_string(cig[k], int(clen[k])),
            "".join("ACGTN"[c] for c in s), "".join(chr(int(x) + 33) for x in q), ed, score, mm, gapo, gape, md or "*"))


def main_paired(prefix, fastq1, fastq2, out=sys.stdout, device="cuda", ref_name="ref", **param_overrides):
    """Paired-end flow: two FASTQ files of equal-length mates -> SAM through nvbio_amd.aligner.best_approx_paired (Aligner::best_approx of
    aligner_best_approx_paired.h) with SamOutput's paired fields (output_sam.cpp:372-520): flags READ_1 / READ_2 by the alignment's mate, REVERSE,
    PAIRED, PROPER_PAIR when the mate's alignment is concordant, MATE_UNMAPPED, MATE_REVERSE; RNEXT '=', PNEXT, TLEN = span of the two alignments,
    negative for the rightmost one; unaligned reads carry the UNMAPPED flag alone, as the reference writes them."""
    # (the 1-mismatch mappers search the reverse index too, mapping_inl.h:128-220: -N 1 needs <prefix>.rbwt)
    data = nio.FMIndexDataDevice(prefix, flags=nio.FORWARD | nio.SA | (nio.REVERSE if param_overrides.get("allow_sub") else 0), device=device)
    n_genome, g_words = nio.load_genome(prefix)
    genome_words = torch.from_numpy(np.concatenate([g_words, np.zeros(8, np.uint32)]).view(np.int32)).to(device)
    r1, r2 = nio.read_fastq(fastq1), nio.read_fastq(fastq2)
    n = r1.size()
    lens = np.concatenate([np.diff(r1.sequence_index), np.diff(r2.sequence_index)])
    if n == 0 or r2.size() != n:
        raise SystemExit("align_fastq: the paired example needs two files with the same number of reads")
    params = A.Params(hits_stride=32, **param_overrides)
    idx = [np.asarray(r.sequence_index, dtype=np.int64) for r in (r1, r2)]
    if (lens != lens[0]).any():
        # mates of their own lengths: two ragged batches
        mates = [A.ReadBatch.from_ragged(torch.from_numpy(r.symbols).to(device), torch.from_numpy(ix).to(device), torch.from_numpy(r.quals).to(device)) for r, ix in ((r1, idx[0]), (r2, idx[1]))]
        r = A.best_approx_paired(data.index(), data.rindex(), mates[0], mates[1], genome_words, n_genome, params, names=list(r1.names), finish=True)
    else:
        L = int(lens[0])
        mats = [torch.from_numpy(r.symbols.reshape(n, L)).to(device) for r in (r1, r2)]
        quals = [torch.from_numpy(r.quals.reshape(n, L)).to(device) for r in (r1, r2)]
        r = A.best_approx_paired(data.index(), data.rindex(), mats[0], mats[1], genome_words, n_genome, params, names=list(r1.names), finish=True, quals1=quals[0], quals2=quals[1])
    torch.cuda.synchronize()
    slots = []
    for key_best, key_tb, key_mds, key_mapq in (("best", "tb1", "mds1", "mapq1"), ("best_o", "tb2", "mds2", "mapq2")):
        slots.append(dict(best=r[key_best].cpu().numpy().view(np.uint64)[0], cigar=r[key_tb]["cigar"].cpu().numpy().view(np.uint16),
                          clen=r[key_tb]["cigar_len"].cpu().numpy(), source=r[key_tb]["source"].cpu().numpy(), mds=r[key_mds].cpu().numpy(), mapq=r[key_mapq].cpu().numpy()))
    reads = (r1, r2)
    ref = Reference(prefix, n_genome, ref_name)
    out.write(ref.header())

    def fields(slot, i):
        w, pos = int(slot["best"][i] & 0xFFFFFFFF), int(slot["best"][i] >> 32)
        if pos == 0xFFFFFFFF:
            return None
        ops = [(int(c) & 3, int(c) >> 2) for c in slot["cigar"][i][:int(slot["clen"][i])]]
        ref_len = sum(l for t, l in ops if t in (0, 2))
        return dict(w=w, pos=pos + int(slot["source"][i, 0]), ref_len=ref_len, rc=(w >> 28) & 1, mate=(w >> 29) & 1,
                    concordant=bool((w >> 30) & 1) and not bool((w >> 31) & 1))

    for i in range(n):
        f = [fields(slots[0], i), fields(slots[1], i)]
        for k in (0, 1):
            a, m = f[k], f[1 - k]
            # SamOutput reads the strand and mate bits of a slot's io::Alignment word whether or not it holds an alignment
            # (output_sam.cpp:389-418, 438-452; get_anchor_mate / get_opposite_mate, output_priv.h): an unaligned read is printed on
            # the strand its word names, and its mate's MATE_REVERSE flag follows that bit too
            w_k, w_m = int(slots[k]["best"][i] & 0xFFFFFFFF), int(slots[1 - k]["best"][i] & 0xFFFFFFFF)
            mate = (w_k >> 29) & 1
            rd = reads[mate]
            seq, qual = rd.symbols[idx[mate][i]:idx[mate][i + 1]], rd.quals[idx[mate][i]:idx[mate][i + 1]]
            if a is None:
                s, q = (np.where(seq < 4, 3 - seq, 4)[::-1], qual[::-1]) if (w_k >> 28) & 1 else (seq, qual)
                out.write("%s\t4\t*\t0\t0\t*\t*\t0\t0\t%s\t%s\n" % (rd.names[i], "".join("ACGTN"[c] for c in s), "".join(chr(int(x) + 33) for x in q)))
                continue
            flags = (0x80 if a["mate"] else 0x40) | (0x10 if a["rc"] else 0) | 0x1
            if m is not None and m["concordant"]:
                flags |= 0x2
            if m is None:
                flags |= 0x8
            if (w_m >> 28) & 1:
                flags |= 0x20
            rname, lpos = ref.locate(a["pos"])
            over = ref.bridges(a["pos"], a["ref_len"])
            if over:
                flags |= 0x4
            if m is not None:
                m_rname, m_lpos = ref.locate(m["pos"])
                rnext = "=" if m_rname == rname else m_rname
                pnext = m_lpos + 1
                tlen = max(m["pos"] + m["ref_len"], a["pos"] + a["ref_len"]) - min(m["pos"], a["pos"])
                if m["pos"] < a["pos"]:
                    tlen = -tlen
                if rnext != "=":
                    tlen = 0
            else:
                rnext, pnext, tlen = "=", lpos + 1, 0
            w = a["w"]
            score, ed = ((w >> 1) & 0x1FFFF) * (-1 if w & 1 else 1), (w >> 18) & 0x3FF
            s, q = (np.where(seq < 4, 3 - seq, 4)[::-1], qual[::-1]) if a["rc"] else (seq, qual)
            md, mm, gapo, gape = nio.sam_md_string(slots[k]["mds"][i])
            out.write("%s\t%d\t%s\t%d\t%d\t%s\t%s\t%d\t%d\t%s\t%s\tNM:i:%d\tAS:i:%d\tXM:i:%d\tXO:i:%d\tXG:i:%d\tMD:Z:%s\n" % (
                rd.names[i], flags, rname, lpos + 1, 0 if over else int(slots[k]["mapq"][i]), cigar_string(slots[k]["cigar"][i], int(slots[k]["clen"][i])), rnext, pnext, tlen,
                "".join("ACGTN"[c] for c in s), "".join(chr(int(x) + 33) for x in q), ed, score, mm, gapo, gape, md or "*"))


if __name__ == "__main__":
    if len(sys.argv) < 3:
        raise SystemExit(__doc__)
    import io as _io
    all_mode = "--all" in sys.argv
    if all_mode:
        sys.argv.remove("--all")
    paired = len(sys.argv) > 4                  # <prefix> <mates 1> <mates 2> <out.sam|out.bam>
    if all_mode and paired:
        raise SystemExit("align_fastq: --all is single-end (as in nvBowtie)")
    if all_mode:
        main, main_best = main_all, main
    if len(sys.argv) > 3:
        path = sys.argv[4] if paired else sys.argv[3]
        buf = _io.StringIO()
        if paired:
            main_paired(sys.argv[1], sys.argv[2], sys.argv[3], buf)
        else:
            main(sys.argv[1], sys.argv[2], buf)
        if path.endswith(".bam"):
            nio.sam_to_bam(buf.getvalue(), path)           # the same records in BAM (nvbio/io/output/output_bam.cpp)
        else:
            with open(path, "w") as f:
                f.write(buf.getvalue())
    else:
        main(sys.argv[1], sys.argv[2])
