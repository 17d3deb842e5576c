#!/usr/bin/env python3
"""BASELINE config 4 through the reference's own application: the unchanged nvBowtie (oracle/_ref/ref_nvBowtie: its 29 translation units compiled
as they lie on the drop-in layer, linked with libnvbio_hip.so) and this repository's from-scratch driver (nvbio_amd.aligner.best_approx =
Aligner::best_approx) on the SAME 3 Gbp index files and the same FASTQ reads, every SAM record compared.  GPU box only.

  * a synthetic genome is drawn on the device (i.i.d. bases; with --repeats a fraction of it is overwritten by diverged copies of a few repeat
    families, the largest with millions of copies, so that seeds reach SA ranges of 2^20 rows and more: SeedHit::range_delta's 20 bits,
    nvBowtie/bowtie2/cuda/seed_hit.h), cut into 24 sequences;
  * forward and reverse FM-indices are built on the device (nvbio_amd.workloads.build_fm_index: prefix doubling) and written as
    <prefix>.bwt/.sa/.rbwt/.rsa/.wpac/.ann/.amb -- the files nvBWT leaves;
  * reads: 100 bp, 3 % substitutions, a fifth with a 1-2 bp indel, one in a hundred with an N, every other one reverse-complemented, per-base
    phred 2..40, written as FASTQ;
  * ref_nvBowtie --file-ref -x <prefix> -U reads.fastq -S ref.sam, then the own driver on the same arrays -> own.sam through the C++ host layer's
    SAM writer (include/nvbio_hip/sam.h); the records are compared byte for byte (SamOutput's READ_1 flag on single-end reads aside).

    python tools/nvbowtie_3gbp.py [--genome 3e9] [--reads 5000000] [--repeats 0.6] [--profile DIR] [--json OUT]
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))

# (length, share of the repeat budget, divergence of a copy from its family's consensus)
FAMILIES = [(200, 1.0 / 3.0, 0.01), (1500, 1.0 / 3.0, 0.01), (5000, 1.0 / 3.0, 0.015)]


def make_genome(n, repeats, seed, dev, families=None):
    """uint8 symbol tensor of length n.  Repeat copies sit on a grid of their family's length (no two copies of a family overlap), families are
    laid down one after the other: the result is a deterministic function of (n, repeats, seed)."""
    g = torch.Generator(device=dev); g.manual_seed(seed)
    text = torch.empty(n, dtype=torch.uint8, device=dev)
    for s in range(0, n, 1 << 28):
        e = min(n, s + (1 << 28))
        text[s:e] = torch.randint(0, 4, (e - s,), dtype=torch.uint8, generator=g, device=dev)
    placed = []
    for L, share, div in (families or FAMILIES):
        copies = int(repeats * share * n / L)
        if copies == 0:
            continue
        consensus = torch.randint(0, 4, (L,), dtype=torch.uint8, generator=g, device=dev)
        slots = torch.randperm(n // L, generator=g, device=dev)[:copies]
        for c0 in range(0, copies, 1 << 18):
            sl = slots[c0:c0 + (1 << 18)]
            idx = (sl * L).unsqueeze(1) + torch.arange(L, device=dev).unsqueeze(0)
            vals = consensus.unsqueeze(0).expand(sl.numel(), L)
            mut = torch.rand((sl.numel(), L), generator=g, device=dev) < div
            shift = torch.randint(1, 4, (sl.numel(), L), dtype=torch.uint8, generator=g, device=dev)
            text[idx.reshape(-1)] = torch.where(mut, (vals + shift) & 3, vals).reshape(-1)
        placed.append((L, copies))
    return text, placed


def save_index(prefix, fmi, reverse):
    """FMIndexDevice -> <prefix>.bwt/.sa (or .rbwt/.rsa), nvBWT's layout (nvbio_amd.io.write_bwt / write_sa)"""
    from nvbio_amd import io as nio
    n = fmi.length
    words = fmi.bwt_occ.view(-1, 8)[:, :4].contiguous().view(-1).cpu().numpy().view(np.uint32)
    cum = np.array(fmi.L2[1:5], dtype=np.uint32)
    nio.write_bwt(prefix + (".rbwt" if reverse else ".bwt"), fmi.primary, cum, words, n)
    nio.write_sa(prefix + (".rsa" if reverse else ".sa"), fmi.primary, cum, fmi.ssa.cpu().numpy().view(np.uint32), n, fmi.sa_int)


def make_reads(text, n_reads, L, seed, seq_bounds, dev):
    """-> (symbols uint8 [n, L] as they appear in the FASTQ file, phred uint8 [n, L]).  No read crosses a sequence boundary."""
    g = torch.Generator(device=dev); g.manual_seed(seed)
    n = text.numel()
    pos = torch.randint(0, n - L - 8, (n_reads,), generator=g, device=dev)
    b = torch.tensor(seq_bounds[1:-1], dtype=torch.int64, device=dev)
    if b.numel():
        k = torch.searchsorted(b, pos + L + 8, right=False)                 # boundaries below the window's end; the last of them may lie inside it
        kb = b[torch.clamp(k - 1, min=0)]
        cross = (k > 0) & (kb > pos)
        pos = torch.where(cross, kb - L - 8, pos)
    idx = pos.unsqueeze(1) + torch.arange(L + 4, device=dev).unsqueeze(0)
    win = text[idx]                                                           # L + 4 symbols: room for a deletion
    # a fifth of the reads carry a 1-2 bp indel at 30-70 % of their length
    r = torch.rand(n_reads, generator=g, device=dev)
    at = torch.randint(3 * L // 10, 7 * L // 10, (n_reads,), generator=g, device=dev)
    gl = torch.randint(1, 3, (n_reads,), generator=g, device=dev)
    col = torch.arange(L, device=dev).unsqueeze(0)
    dele, ins = (r < 0.1).unsqueeze(1), ((r >= 0.1) & (r < 0.2)).unsqueeze(1)
    src = torch.where(dele & (col >= at.unsqueeze(1)), col + gl.unsqueeze(1), col)                      # deletion: skip gl reference symbols
    src = torch.where(ins & (col >= (at + gl).unsqueeze(1)), col - gl.unsqueeze(1), src)                # insertion: gl new symbols, the rest shifted
    sym = torch.gather(win, 1, src)
    rnd = torch.randint(0, 4, (n_reads, L), dtype=torch.uint8, generator=g, device=dev)
    sym = torch.where(ins & (col >= at.unsqueeze(1)) & (col < (at + gl).unsqueeze(1)), rnd, sym)
    mut = torch.rand((n_reads, L), generator=g, device=dev) < 0.03
    sym = torch.where(mut, (sym + 1 + (rnd % 3)) & 3, sym)
    rc = (torch.arange(n_reads, device=dev) & 1).bool().unsqueeze(1)
    sym = torch.where(rc, (3 - sym).flip(1), sym)
    hasn = torch.rand(n_reads, generator=g, device=dev) < 0.01
    npos = torch.randint(0, L, (n_reads,), generator=g, device=dev)
    sym = torch.where(hasn.unsqueeze(1) & (col == npos.unsqueeze(1)), torch.full_like(sym, 4), sym)
    qual = torch.randint(2, 41, (n_reads, L), dtype=torch.uint8, generator=g, device=dev)
    return sym.contiguous(), qual.contiguous(), pos


def write_fastq(path, sym, qual, digits=8, tag="r"):
    """@r<8 digits>, sequence, +, phred+33: built as one byte matrix on the device"""
    n, L = sym.shape
    dev = sym.device
    rec = torch.empty((n, 2 + digits + 1 + L + 3 + L + 1), dtype=torch.uint8, device=dev)
    rec[:, 0] = ord("@"); rec[:, 1] = ord(tag)
    ids = torch.arange(n, device=dev)
    for d in range(digits):
        rec[:, 2 + d] = ((ids // (10 ** (digits - 1 - d))) % 10 + ord("0")).to(torch.uint8)
    o = 2 + digits
    rec[:, o] = ord("\n")
    lut = torch.tensor(list(b"ACGTN"), dtype=torch.uint8, device=dev)
    rec[:, o + 1:o + 1 + L] = lut[sym.long()]
    rec[:, o + 1 + L] = ord("\n"); rec[:, o + 2 + L] = ord("+"); rec[:, o + 3 + L] = ord("\n")
    rec[:, o + 4 + L:o + 4 + 2 * L] = qual + 33
    rec[:, o + 4 + 2 * L] = ord("\n")
    rec.cpu().numpy().tofile(path)


def own_driver(prefix, sym, qual, sam_path, dev, batch_reads, digits=8, timings=None, overrides=None):
    """this repository's driver over the same reads, batch by batch -> own.sam"""
    import align_fastq as AF
    from nvbio_amd import io as nio, aligner as A
    t0 = time.time()
    data = nio.FMIndexDataDevice(prefix, flags=nio.FORWARD | nio.SA, device=dev)
    n_genome, g_words = nio.load_genome(prefix)
    genome_words = torch.from_numpy(np.concatenate([g_words, np.zeros(8, np.uint32)]).view(np.int32)).to(dev)
    ref = AF.Reference(prefix, n_genome, "ref")
    torch.cuda.synchronize()
    t_load = time.time() - t0
    n, L = sym.shape
    params = A.Params(**(overrides or {}))
    t_align, t_write, stats = 0.0, 0.0, []
    for s in range(0, n, batch_reads):
        e = min(n, s + batch_reads)
        m = e - s
        names = ["r%0*d" % (digits, i) for i in range(s, e)]
        index = torch.arange(0, (m + 1) * L, L, dtype=torch.int64, device=dev)
        batch = A.ReadBatch.from_ragged(sym[s:e].reshape(-1), index, qual[s:e].reshape(-1))
        torch.cuda.synchronize(); t1 = time.time()
        r = A.best_approx(data.index(), data.rindex(), batch, genome_words, n_genome, params, names=names, cigar_stride=64, finish=True)
        torch.cuda.synchronize(); t_align += time.time() - t1
        stats.append(dict((k, v) for k, v in r["stats"].items() if k != "ms"))
        t1 = time.time()
        name_buf = np.frombuffer(("\0".join(names) + "\0").encode(), dtype=np.uint8)
        name_idx = np.arange(0, (m + 1) * (digits + 2), digits + 2, dtype=np.uint32)
        AF.write_records_se_native(sam_path, ref, (name_buf, name_idx), sym[s:e].reshape(-1).cpu().numpy(), index.cpu().numpy(), qual[s:e].reshape(-1).cpu().numpy(),
                                   r["best"].cpu().numpy().view(np.uint64), r["mapq"].cpu().numpy(), r["cigar"].cpu().numpy().view(np.uint16), r["cigar_len"].cpu().numpy(),
                                   r["source"].cpu().numpy(), r["mds"].cpu().numpy(), extra_flags=64, append=s > 0, header=s == 0)
        t_write += time.time() - t1
        del r, batch
    if timings is not None:
        timings.update(own_load_s=t_load, own_align_s=t_align, own_write_s=t_write, own_reads_per_s=n / t_align, own_batches=len(stats), own_stats_first_batch=stats[0])


def make_pairs(text, n_pairs, L, seed, seq_bounds, dev, frag=(250, 450)):
    """FR pairs with per-base qualities: mate 1 = the fragment's first L bases, mate 2 = the reverse complement of its last L; half of the fragments come
    from the reverse strand; 3 % substitutions, a tenth of the mates with a 1-2 bp indel, one mate in a hundred with an N.  No fragment crosses a sequence
    boundary.  -> (sym1, sym2, qual1, qual2) uint8 [n, L]"""
    g = torch.Generator(device=dev); g.manual_seed(seed)
    n = text.numel()
    flen = torch.randint(frag[0], frag[1] + 1, (n_pairs,), generator=g, device=dev)
    pos = torch.randint(0, n - frag[1] - 16, (n_pairs,), generator=g, device=dev)
    b = torch.tensor(seq_bounds[1:-1], dtype=torch.int64, device=dev)
    if b.numel():
        k = torch.searchsorted(b, pos + flen + 8, right=False)
        kb = b[torch.clamp(k - 1, min=0)]
        cross = (k > 0) & (kb > pos)
        pos = torch.where(cross, kb - flen - 8, pos)

    def mate(start, rc, salt):
        idx = start.unsqueeze(1) + torch.arange(L + 4, device=dev).unsqueeze(0)
        win = text[idx]
        r = torch.rand(n_pairs, generator=g, device=dev)
        at = torch.randint(3 * L // 10, 7 * L // 10, (n_pairs,), generator=g, device=dev)
        gl = torch.randint(1, 3, (n_pairs,), generator=g, device=dev)
        col = torch.arange(L, device=dev).unsqueeze(0)
        dele, ins = (r < 0.05).unsqueeze(1), ((r >= 0.05) & (r < 0.1)).unsqueeze(1)
        src = torch.where(dele & (col >= at.unsqueeze(1)), col + gl.unsqueeze(1), col)
        src = torch.where(ins & (col >= (at + gl).unsqueeze(1)), col - gl.unsqueeze(1), src)
        sym = torch.gather(win, 1, src)
        rnd = torch.randint(0, 4, (n_pairs, L), dtype=torch.uint8, generator=g, device=dev)
        sym = torch.where(ins & (col >= at.unsqueeze(1)) & (col < (at + gl).unsqueeze(1)), rnd, sym)
        mut = torch.rand((n_pairs, L), generator=g, device=dev) < 0.03
        sym = torch.where(mut, (sym + 1 + (rnd % 3)) & 3, sym)
        if rc:
            sym = (3 - sym).flip(1)
        hasn = torch.rand(n_pairs, generator=g, device=dev) < 0.01
        npos = torch.randint(0, L, (n_pairs,), generator=g, device=dev)
        sym = torch.where(hasn.unsqueeze(1) & (col == npos.unsqueeze(1)), torch.full_like(sym, 4), sym)
        q = torch.randint(2, 41, (n_pairs, L), dtype=torch.uint8, generator=g, device=dev)
        return sym.contiguous(), q.contiguous()
    m1, q1 = mate(pos, False, 0)
    m2, q2 = mate(pos + flen - L - 2, True, 1)              # (the window is L + 4 wide: the mate's last base sits near the fragment's end)
    swap = (torch.rand(n_pairs, generator=g, device=dev) < 0.5).unsqueeze(1)
    return (torch.where(swap, m2, m1).contiguous(), torch.where(swap, m1, m2).contiguous(), torch.where(swap, q2, q1).contiguous(), torch.where(swap, q1, q2).contiguous())


def own_driver_cxx_paired(prefix, s1, s2, q1, q2, sam_path, dev, batch_pairs, digits=8, timings=None, local=True):
    """the C++ paired-end driver (Aligner::best_approx over a PairedReadBatch) over the same pairs on the same index files -> own_pe.sam through
    the host layer's paired SAM writer (include/nvbio_hip/sam.h: write_sam_pe)"""
    import ctypes as C
    import align_fastq as AF
    import bench as B
    import nvbio_amd as nvb
    from nvbio_amd import io as nio, aligner as A, pipeline as P, select as SEL
    shim = C.CDLL(os.path.join(ROOT, "tests", "cxx", "libaligner_shim.so"))
    t0 = time.time()
    data = nio.FMIndexDataDevice(prefix, flags=nio.FORWARD | nio.SA, device=dev)
    n_genome, g_words = nio.load_genome(prefix)
    genome_words = torch.from_numpy(np.concatenate([g_words, np.zeros(8, np.uint32)]).view(np.int32)).to(dev)
    ref = AF.Reference(prefix, n_genome, "ref")
    torch.cuda.synchronize()
    t_load = time.time() - t0
    n, L = s1.shape
    prm = A.Params(batch_size=batch_pairs, local=local)
    scheme = nvb.SmithWatermanScoringScheme.local() if local else nvb.SmithWatermanScoringScheme()
    sp = B._shim_params(prm, scheme); sp.finish = 1

    class ShimPeParams(C.Structure):
        _fields_ = [("pe_policy", C.c_int32)] + [(k, C.c_uint32) for k in ("pe_overlap", "pe_unpaired", "pe_discordant", "min_frag_len", "max_frag_len")]
    pp = ShimPeParams(prm.pe_policy, int(prm.pe_overlap), int(prm.pe_unpaired), int(prm.pe_discordant), prm.min_frag_len, prm.max_frag_len)
    fs = data.index().struct()
    vp = lambda t: C.c_void_p(t.data_ptr())
    pair_ptrs = lambda ts: (C.c_void_p * 2)(*[t.data_ptr() for t in ts])
    pair_host = lambda arrs: (C.c_void_p * 2)(*[a.ctypes.data for a in arrs])
    u64x2 = lambda v: (C.c_uint64 * 2)(*v)
    t_align, t_write, first_stats = 0.0, 0.0, None
    seq_names = (C.c_char_p * len(ref.names))(*[nm.encode() for nm in ref.names])
    seq_index = np.ascontiguousarray(ref.index, dtype=np.uint64)
    for b0 in range(0, n, batch_pairs):
        e0 = min(n, b0 + batch_pairs); m = e0 - b0
        mates = [s1[b0:e0].contiguous(), s2[b0:e0].contiguous()]
        mq = [q1[b0:e0].contiguous(), q2[b0:e0].contiguous()]
        packed = [P.pack_read_streams(x) for x in mates]
        qs = [A._qual_stream(m, L, 30, x, dev) for x in mq]
        both = torch.cat([packed[0][1], packed[1][1]]); mate_offset = packed[0][1].numel() * 8
        both_q = torch.zeros(mate_offset + 2 * m * L + 8, dtype=torch.uint8, device=dev)
        both_q[:qs[0].numel()] = qs[0]; both_q[mate_offset:mate_offset + qs[1].numel()] = qs[1]
        names = ["p%0*d" % (digits, i) for i in range(b0, e0)]
        arena, nidx = SEL.pack_names(names, dev)
        out = dict(best=[np.zeros((2, m), np.uint64) for _ in range(2)], mapq=[np.zeros(m, np.uint8) for _ in range(2)], cigar=[np.zeros((m, 64), np.uint16) for _ in range(2)],
                   cigar_len=[np.zeros(m, np.uint32) for _ in range(2)], source=[np.zeros((m, 2), np.uint32) for _ in range(2)], sink=[np.zeros((m, 2), np.uint32) for _ in range(2)],
                   mds=[np.zeros((m, 256), np.uint8) for _ in range(2)], mds_len=[np.zeros(m, np.uint32) for _ in range(2)])
        stats = np.zeros(12, np.uint64)
        ms = (C.c_double * 1)()
        torch.cuda.synchronize(); t1 = time.time()
        rc = shim.nvbio_aligner_best_approx_paired_quals(
            C.byref(fs), None, C.c_uint32(m), C.c_uint32(L),
            pair_ptrs([packed[0][0].words, packed[1][0].words]), u64x2([packed[0][0].words.numel(), packed[1][0].words.numel()]), pair_ptrs([packed[0][0].begin, packed[1][0].begin]),
            pair_ptrs([packed[0][1], packed[1][1]]), u64x2([packed[0][1].numel(), packed[1][1].numel()]), pair_ptrs(qs), C.c_uint64(qs[0].numel()), vp(arena), vp(nidx),
            vp(both), C.c_uint64(both.numel()), C.c_uint64(mate_offset), vp(both_q), C.c_uint64(both_q.numel()),
            vp(genome_words), C.c_uint64(genome_words.numel()), C.c_uint32(n_genome), C.byref(sp), C.byref(pp),
            pair_host(out["best"]), pair_host(out["mapq"]), pair_host(out["cigar"]), pair_host(out["cigar_len"]), pair_host(out["source"]), pair_host(out["sink"]),
            pair_host(out["mds"]), pair_host(out["mds_len"]), stats.ctypes.data_as(C.c_void_p), ms)
        t_align += time.time() - t1
        if rc != 0:
            raise RuntimeError("nvbio_aligner_best_approx_paired_quals returned %d" % rc)
        if first_stats is None:
            first_stats = dict(anchor_extensions=int(stats[0]), rounds=int(stats[1]), seeding_passes=int(stats[2]))
        t1 = time.time()
        name_buf = np.frombuffer(("\0".join(names) + "\0").encode(), dtype=np.uint8)
        name_idx = np.arange(0, (m + 1) * (digits + 2), digits + 2, dtype=np.uint32)
        hs = [x.cpu().numpy() for x in mates]; hq = [x.cpu().numpy() for x in mq]
        best0 = [np.ascontiguousarray(x[0]) for x in out["best"]]
        pv = lambda a: a.ctypes.data_as(C.c_void_p)
        rc = shim.nvbio_write_sam_pe(sam_path.encode(), C.c_int(1 if b0 > 0 else 0), C.c_int(1 if b0 == 0 else 0), C.c_uint32(m), C.c_uint32(L), pv(name_buf), pv(name_idx),
                                     pair_host(hs), pair_host(hq), pair_host(best0), pair_host(out["mapq"]), pair_host(out["cigar"]), C.c_uint32(64), pair_host(out["cigar_len"]),
                                     pair_host(out["source"]), pair_host(out["mds"]), C.c_uint32(256), C.c_uint32(len(ref.names)), seq_names, pv(seq_index))
        if rc != 0:
            raise IOError("nvbio_write_sam_pe failed: %d" % rc)
        t_write += time.time() - t1
    if timings is not None:
        timings.update(cxx_load_s=t_load, cxx_align_s_incl_result_copies=t_align, cxx_write_s=t_write, cxx_pairs_per_s=n / t_align, cxx_index=data.description, cxx_stats_first_batch=first_stats)


def build_files(tmp, genome, repeats, seed, dev, families=None, out=None):
    """the synthetic genome, its forward / reverse index files and the BWA-style reference files under <tmp>/genome -> (prefix, text, sequence bounds)"""
    from nvbio_amd import workloads as W, io as nio
    from nvbio_amd.strings import pack_symbols
    out = out if out is not None else {}
    prefix = os.path.join(tmp, "genome")
    t0 = time.time()
    text, placed = make_genome(genome, repeats, seed, dev, families)
    out["repeat_families"] = [dict(length=L, copies=c) for L, c in placed]
    n_seq = 24
    lens = [genome // n_seq] * (n_seq - 1); lens.append(genome - sum(lens))
    bounds = [0] + [int(x) for x in np.cumsum(lens)]
    fmi = W.build_fm_index(text)
    save_index(prefix, fmi, False)
    del fmi
    rfmi = W.build_fm_index(text.flip(0).contiguous())
    save_index(prefix, rfmi, True)
    del rfmi
    gw = torch.cat([pack_symbols(text[s:s + (1 << 30)], 2, True, pad_words=0) for s in range(0, genome, 1 << 30)])
    nio.write_wpac(prefix + ".wpac", genome, gw.cpu().numpy().view(np.uint32)); del gw
    nio.write_bns(prefix, ["chr%d" % (k + 1) for k in range(n_seq)], lens)
    torch.cuda.synchronize(); out["files_s"] = time.time() - t0
    out["index_files_GB"] = sum(os.path.getsize(prefix + e) for e in (".bwt", ".sa", ".rbwt", ".rsa", ".wpac")) / 1e9
    return prefix, text, bounds


def run_paired(genome=3_000_000_000, pairs=1 << 20, repeats=0.6, seed=0x5EED0019, batch_pairs=1 << 20, read_len=150, workdir=None, keep=False, extra=("--local",)):
    """BASELINE config 5's shape at its index size: 2 x 150 bp FR pairs, --local (LOCAL Gotoh, band 31 from the default max_dist 15), on the 3 Gbp index
    files: the unchanged nvBowtie (-1 / -2) and the C++ paired-end driver on the same files and reads, every SAM record compared."""
    dev = torch.device("cuda:0")
    out = dict(genome=genome, pairs=pairs, repeats=repeats, read_len=read_len, mode=" ".join(extra))
    if workdir:
        os.makedirs(workdir, exist_ok=True)
    tmp = workdir or tempfile.mkdtemp(prefix="nvb3gpe_")
    try:
        prefix, text, bounds = build_files(tmp, genome, repeats, seed, dev, out=out)
        s1, s2, q1, q2 = make_pairs(text, pairs, read_len, seed + 1, bounds, dev)
        del text
        torch.cuda.empty_cache()
        f1, f2 = os.path.join(tmp, "m1.fastq"), os.path.join(tmp, "m2.fastq")
        write_fastq(f1, s1, q1, tag="p"); write_fastq(f2, s2, q2, tag="p")
        exe = os.path.join(ROOT, "oracle", "_ref", "ref_nvBowtie")
        ref_sam = os.path.join(tmp, "ref_pe.sam")
        cmd = [exe] + list(extra) + ["--file-ref", "-x", prefix, "-1", f1, "-2", f2, "-S", ref_sam]
        t0 = time.time()
        r = subprocess.run(cmd, capture_output=True, text=True)
        out["nvbowtie_wall_s"] = time.time() - t0
        log = (r.stdout + r.stderr).replace("\r", "\n")
        out["nvbowtie_exit"] = r.returncode
        out["nvbowtie_log_tail"] = [l for l in log.splitlines() if l.strip()][-40:]
        if r.returncode != 0:
            return out, log
        own_sam = os.path.join(tmp, "own_pe.sam")
        own_driver_cxx_paired(prefix, s1, s2, q1, q2, own_sam, dev, batch_pairs, timings=out, local="--local" in extra)
        n_ref, n_own, same, diffs, cats = compare_sam(ref_sam, own_sam)
        out.update(records_ref=n_ref, records_own=n_own, identical=same, difference_categories=cats, first_differences=diffs)
        return out, log
    finally:
        if not keep and workdir is None:
            import shutil
            shutil.rmtree(tmp, ignore_errors=True)


def own_driver_cxx(prefix, sym, qual, sam_path, dev, batch_reads, digits=8, timings=None, hbm_rich=True, overrides=None):
    """the C++ host driver (nvbio::bowtie2::cuda::Aligner::best_approx, include/nvbio_hip/aligner.h; entered through tests/cxx/aligner_shim.cpp)
    over the same reads on the same index files, batch by batch -> own_cxx.sam.  The index is what io::FMIndexDataDevice makes of the files on
    this device by default: the HBM-rich form (line-native records, 12-mer table, the densest suffix array that fits)."""
    import ctypes as C
    import align_fastq as AF
    import bench as B
    import nvbio_amd as nvb
    from nvbio_amd import io as nio, aligner as A, pipeline as P, select as SEL
    shim = C.CDLL(os.path.join(ROOT, "tests", "cxx", "libaligner_shim.so"))
    t0 = time.time()
    data = nio.FMIndexDataDevice(prefix, flags=nio.FORWARD | nio.SA, device=dev, hbm_rich=hbm_rich)
    n_genome, g_words = nio.load_genome(prefix)
    genome_words = torch.from_numpy(np.concatenate([g_words, np.zeros(8, np.uint32)]).view(np.int32)).to(dev)
    ref = AF.Reference(prefix, n_genome, "ref")
    torch.cuda.synchronize()
    t_load = time.time() - t0
    n, L = sym.shape
    prm = A.Params(batch_size=batch_reads, **(overrides or {}))
    scheme = nvb.SmithWatermanScoringScheme()
    sp = B._shim_params(prm, scheme); sp.finish = 1
    fs = data.index().struct()
    vp = lambda t: C.c_void_p(t.data_ptr())
    hp = lambda a: a.ctypes.data_as(C.c_void_p)
    t_align, t_write, first_stats = 0.0, 0.0, None
    for s0 in range(0, n, batch_reads):
        e0 = min(n, s0 + batch_reads); m = e0 - s0
        sb = sym[s0:e0].contiguous()
        rev, fwrc = P.pack_read_streams(sb)
        qs = A._qual_stream(m, L, 30, qual[s0:e0], dev)
        names = ["r%0*d" % (digits, i) for i in range(s0, e0)]
        arena, nidx = SEL.pack_names(names, dev)
        best = np.zeros((2, m), np.uint64); mapq = np.zeros(m, np.uint8); cigar = np.zeros((m, 64), np.uint16); cigar_len = np.zeros(m, np.uint32)
        source = np.zeros((m, 2), np.uint32); sink = np.zeros((m, 2), np.uint32); tb_score = np.zeros(m, np.int32); stats = np.zeros(12, np.uint64)
        mds = np.zeros((m, 256), np.uint8); mds_len = np.zeros(m, np.uint32)
        torch.cuda.synchronize(); t1 = time.time()
        rc = shim.nvbio_aligner_best_approx(C.byref(fs), None, C.c_uint32(m), C.c_uint32(L), vp(rev.words), C.c_uint64(rev.words.numel()), vp(rev.begin), vp(fwrc),
                                            C.c_uint64(fwrc.numel()), vp(qs), C.c_uint64(qs.numel()), vp(arena), vp(nidx), vp(genome_words), C.c_uint64(genome_words.numel()),
                                            C.c_uint32(n_genome), C.byref(sp), hp(best), hp(mapq), hp(cigar), hp(cigar_len), hp(source), hp(sink), hp(tb_score), hp(stats),
                                            hp(mds), hp(mds_len))
        t_align += time.time() - t1
        if rc != 0:
            raise RuntimeError("nvbio_aligner_best_approx returned %d" % rc)
        if first_stats is None:
            first_stats = dict(extensions=int(stats[0]), rounds=int(stats[1]), seeding_passes=int(stats[2]), queue=[int(x) for x in stats[4:4 + int(stats[3])]])
        t1 = time.time()
        name_buf = np.frombuffer(("\0".join(names) + "\0").encode(), dtype=np.uint8)
        name_idx = np.arange(0, (m + 1) * (digits + 2), digits + 2, dtype=np.uint32)
        index = np.arange(0, (m + 1) * L, L, dtype=np.int64)
        AF.write_records_se_native(sam_path, ref, (name_buf, name_idx), sb.reshape(-1).cpu().numpy(), index, qual[s0:e0].reshape(-1).cpu().numpy(),
                                   best, mapq, cigar, cigar_len, source, mds, extra_flags=64, append=s0 > 0, header=s0 == 0)
        t_write += time.time() - t1
    if timings is not None:
        timings.update(cxx_load_s=t_load, cxx_align_s_incl_result_copies=t_align, cxx_write_s=t_write, cxx_reads_per_s=n / t_align, cxx_index=data.description,
                       cxx_stats_first_batch=first_stats)


def compare_sam(ref_path, own_path, show=5):
    """-> (records of the reference, of the own driver, identical ones, examples, categories); byte comparison first, line by line only when that fails"""
    def body(path):
        raw = np.fromfile(path, dtype=np.uint8)
        at = 0
        while at < raw.size and raw[at] == ord("@"):                           # header lines
            nl = int(np.flatnonzero(raw[at:at + (1 << 20)] == 10)[0])
            at += nl + 1
        return raw[at:]
    a, b = body(ref_path), body(own_path)
    n_a, n_b = int((a == 10).sum()), int((b == 10).sum())
    if a.size == b.size and bool((a == b).all()):
        return n_a, n_b, n_a, [], {}
    la, lb = a.tobytes().split(b"\n"), b.tobytes().split(b"\n")
    same, diffs, cats = 0, [], {}
    for k, (x, y) in enumerate(zip(la, lb)):
        if x == y:
            same += 1 if x else 0
            continue
        fx, fy = x.split(b"\t"), y.split(b"\t")
        d = [i for i in range(min(len(fx), len(fy))) if fx[i] != fy[i]]
        ua, ub = (int(fx[1]) & 4) != 0, (int(fy[1]) & 4) != 0
        if ua != ub:
            cat = "aligned_vs_unaligned"
        elif len(fx) > 12 and len(fy) > 12 and fx[12] != fy[12]:
            cat = "different_score"                                            # AS:i differs
        elif fx[4] != fy[4]:
            cat = "same_score_different_mapq"
        elif fx[2] != fy[2] or fx[3] != fy[3]:
            cat = "same_score_and_mapq_other_placement"
        else:
            cat = "other_fields_%s" % ",".join(map(str, d))
        cats[cat] = cats.get(cat, 0) + 1
        cats["in_batch_%d" % (k >> 20)] = cats.get("in_batch_%d" % (k >> 20), 0) + 1
        if len(diffs) < show or (cats[cat] <= 3 and len(diffs) < 4 * show):
            diffs.append(dict(category=cat, fields=d, ref=[f.decode() for f in fx[:9] + fx[11:]], own=[f.decode() for f in fy[:9] + fy[11:]]))
    return n_a, n_b, same, diffs, cats


def multiset_difference(path_a, path_b, show=6):
    """records (header aside) of two SAM files as multisets -> (only in a, only in b, examples)"""
    from collections import Counter
    ca = Counter(l for l in open(path_a, "rb").read().split(b"\n") if l and not l.startswith(b"@"))
    cb = Counter(l for l in open(path_b, "rb").read().split(b"\n") if l and not l.startswith(b"@"))
    only_a, only_b = ca - cb, cb - ca
    ex = [[f.decode() for f in l.split(b"\t")[:9] + l.split(b"\t")[11:]] for l in sorted(only_a)[:show]]
    exb = [[f.decode() for f in l.split(b"\t")[:9] + l.split(b"\t")[11:]] for l in sorted(only_b)[:show]]
    return sum(only_a.values()), sum(only_b.values()), ex, exb


def run(genome=3_000_000_000, reads=5_000_000, repeats=0.6, seed=0x5EED0009, batch_reads=1 << 20, profile=None, workdir=None, keep=False, extra=(), threads_test=False, rerun=False, own_overrides=None, families=None):
    from nvbio_amd import workloads as W, io as nio
    dev = torch.device("cuda:0")
    out = dict(genome=genome, reads=reads, repeats=repeats, read_len=100)
    if workdir:
        os.makedirs(workdir, exist_ok=True)
    tmp = workdir or tempfile.mkdtemp(prefix="nvb3g_")
    try:
        return _run(tmp, genome, reads, repeats, seed, batch_reads, profile, extra, threads_test, rerun, own_overrides, families, out, dev)
    finally:
        if not keep and workdir is None:          # on every path: the directory holds 4 GB of index files
            import shutil
            shutil.rmtree(tmp, ignore_errors=True)


def _run(tmp, genome, reads, repeats, seed, batch_reads, profile, extra, threads_test, rerun, own_overrides, families, out, dev):
    from nvbio_amd import workloads as W, io as nio
    prefix = os.path.join(tmp, "genome")
    t0 = time.time()
    text, placed = make_genome(genome, repeats, seed, dev, families)
    out["repeat_families"] = [dict(length=L, copies=c) for L, c in placed]
    torch.cuda.synchronize(); out["genome_s"] = time.time() - t0
    n_seq = 24
    lens = [genome // n_seq] * (n_seq - 1); lens.append(genome - sum(lens))
    bounds = [0] + list(np.cumsum(lens))
    t0 = time.time()
    fmi = W.build_fm_index(text)
    torch.cuda.synchronize(); out["index_build_s"] = time.time() - t0
    t0 = time.time()
    save_index(prefix, fmi, False)
    del fmi
    rfmi = W.build_fm_index(text.flip(0).contiguous())
    torch.cuda.synchronize(); out["rindex_build_s"] = time.time() - t0
    save_index(prefix, rfmi, True)
    del rfmi
    from nvbio_amd.strings import pack_symbols
    gw = torch.cat([pack_symbols(text[s:s + (1 << 30)], 2, True, pad_words=0) for s in range(0, genome, 1 << 30)])
    nio.write_wpac(prefix + ".wpac", genome, gw.cpu().numpy().view(np.uint32)); del gw
    nio.write_bns(prefix, ["chr%d" % (k + 1) for k in range(n_seq)], lens)
    out["files_s"] = time.time() - t0
    sym, qual, pos = make_reads(text, reads, 100, seed + 1, bounds, dev)
    del text
    torch.cuda.empty_cache()
    fq = os.path.join(tmp, "reads.fastq")
    write_fastq(fq, sym, qual)
    out["index_files_GB"] = sum(os.path.getsize(prefix + e) for e in (".bwt", ".sa", ".rbwt", ".rsa", ".wpac")) / 1e9
    # ---- the reference's application
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_nvBowtie")
    ref_sam = os.path.join(tmp, "ref.sam")
    cmd = [exe] + list(extra) + ["--file-ref", "-x", prefix, "-U", fq, "-S", ref_sam]
    t0 = time.time()
    r = subprocess.run(cmd, capture_output=True, text=True)
    out["nvbowtie_wall_s"] = time.time() - t0
    log = (r.stdout + r.stderr).replace("\r", "\n")
    out["nvbowtie_exit"] = r.returncode
    out["nvbowtie_log_tail"] = [l for l in log.splitlines() if l.strip()][-60:]
    out["nvbowtie_reads_per_s_incl_io"] = reads / out["nvbowtie_wall_s"]
    if r.returncode != 0:
        return out, log
    # ---- this repository's driver
    own_sam = os.path.join(tmp, "own.sam")
    own_driver(prefix, sym, qual, own_sam, dev, batch_reads, timings=out, overrides=own_overrides)
    t0 = time.time()
    n_ref, n_own, same, diffs, cats = compare_sam(ref_sam, own_sam)
    out.update(records_ref=n_ref, records_own=n_own, identical=same, difference_categories=cats, first_differences=diffs, compare_s=time.time() - t0)
    # ---- ... and its C++ host driver on the HBM-rich index this device gets by default
    torch.cuda.empty_cache()
    cxx_sam = os.path.join(tmp, "own_cxx.sam")
    own_driver_cxx(prefix, sym, qual, cxx_sam, dev, batch_reads, timings=out, overrides=own_overrides)
    n_ref2, n_cxx, same2, diffs2, cats2 = compare_sam(ref_sam, cxx_sam)
    out.update(records_cxx=n_cxx, cxx_identical=same2, cxx_difference_categories=cats2, cxx_first_differences=diffs2)
    if rerun:
        # the reference's application against itself: a second run on the same files
        again = os.path.join(tmp, "ref2.sam")
        r3 = subprocess.run(cmd[:-1] + [again], capture_output=True, text=True)
        if r3.returncode == 0:
            oa, ob, ex, exb = multiset_difference(ref_sam, again)
            out["nvbowtie_rerun"] = dict(only_first=oa, only_second=ob, examples_first=ex, examples_second=exb)
    # aligned share and wide SA ranges seen, from the reference's SAM
    flags = []
    with open(ref_sam, "rb") as f:
        for ln in f:
            if not ln.startswith(b"@"):
                flags.append(int(ln.split(b"\t", 2)[1]))
                if len(flags) >= 200_000:
                    break
    out["aligned_share_first_200k"] = float(np.mean((np.array(flags) & 4) == 0)) if flags else None
    if threads_test:
        # nvBowtie's multi-device mode on one GPU: two compute threads, shared input thread, mutexed output (nvBowtie.cpp:809-864)
        mt_sam = os.path.join(tmp, "ref_mt.sam")
        t0 = time.time()
        r2 = subprocess.run([exe] + list(extra) + ["--device", "0", "--device", "0", "--file-ref", "-x", prefix, "-U", fq, "-S", mt_sam], capture_output=True, text=True)
        out["nvbowtie_two_threads_wall_s"] = time.time() - t0
        out["nvbowtie_two_threads_exit"] = r2.returncode
        if r2.returncode == 0:
            oa, ob, ex, exb = multiset_difference(ref_sam, mt_sam)
            out["two_threads_same_multiset"] = (oa == 0 and ob == 0)
            out["two_threads_difference"] = dict(only_single=oa, only_two_threads=ob, examples_single=ex, examples_two_threads=exb)
        else:
            out["two_threads_log_tail"] = (r2.stdout + r2.stderr).replace("\r", "\n")[-1500:]
    if profile:
        profile = os.path.abspath(profile)
        os.makedirs(profile, exist_ok=True)
        pr = subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", profile, "-o", "ref_nvbowtie_3gbp", "--"] + cmd[:-1] + [os.path.join(tmp, "prof.sam")],
                            capture_output=True, text=True, env=dict(os.environ, TMPDIR="/tmp"), cwd="/tmp")
        out["profile_exit"] = pr.returncode
    return out, log


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome", type=float, default=3e9)
    ap.add_argument("--reads", type=int, default=5_000_000)
    ap.add_argument("--repeats", type=float, default=0.6)
    ap.add_argument("--batch-reads", type=int, default=1 << 20, help="reads per batch of the own driver: nvBowtie's default batch (1024 K reads), because the hits-per-read rule of both drivers reasons with the batch")
    ap.add_argument("--profile", default=None, help="directory for a second nvBowtie run under rocprofv3 --kernel-trace --stats")
    ap.add_argument("--json", default=None)
    ap.add_argument("--log", default=None, help="where to keep nvBowtie's own log")
    ap.add_argument("--two-threads", action="store_true", help="also run nvBowtie with --device 0 --device 0 and compare the records as a multiset")
    ap.add_argument("--extra", default="")
    ap.add_argument("--families", default="", help="repeat families as length:share:divergence,... (shares of the --repeats budget; default 200:1/3:0.01,1500:1/3:0.01,5000:1/3:0.015)")
    ap.add_argument("--own", default="", help="the same settings for this repository's driver: comma-separated Params fields, e.g. 'no_multi_hits=True'")
    ap.add_argument("--rerun", action="store_true", help="run nvBowtie a second time and compare its two outputs")
    ap.add_argument("--keep", default=None, help="work in this directory and keep the files")
    ap.add_argument("--paired", type=int, default=0, help="run the paired-end comparison instead (config 5's shape: 2 x 150 bp, --local) with this many pairs")
    a = ap.parse_args()
    if a.paired:
        out, log = run_paired(int(a.genome), a.paired, a.repeats, workdir=a.keep)
        if a.log:
            open(a.log, "w").write(log)
        text = json.dumps(out, indent=1, default=str)
        if a.json:
            open(a.json, "w").write(text)
        print(text)
        return 0 if out.get("nvbowtie_exit") == 0 and out.get("identical") == out.get("records_ref") == out.get("records_own") == 2 * a.paired else 1
    overrides = {}
    for kv in filter(None, a.own.split(",")):
        k, v = kv.split("=")
        overrides[k] = (v == "True") if v in ("True", "False") else (int(v) if v.lstrip("-").isdigit() else v)
    out, log = run(int(a.genome), a.reads, a.repeats, batch_reads=a.batch_reads, profile=a.profile, extra=a.extra.split(), threads_test=a.two_threads, rerun=a.rerun,
                   workdir=a.keep, own_overrides=overrides,
                   families=[(int(f.split(":")[0]), float(f.split(":")[1]), float(f.split(":")[2])) for f in a.families.split(",")] if a.families else None)
    if a.keep:
        os.makedirs(a.keep, exist_ok=True)
    if a.log:
        open(a.log, "w").write(log)
    text = json.dumps(out, indent=1, default=str)
    if a.json:
        open(a.json, "w").write(text)
    print(text)
    ok = out.get("nvbowtie_exit") == 0 and out.get("identical") == out.get("records_ref") == out.get("records_own") == a.reads and out.get("cxx_identical") == a.reads
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
