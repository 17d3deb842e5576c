#!/usr/bin/env python3
"""This repository's C++ single-end driver (nvbio::bowtie2::cuda::Aligner::best_approx, include/nvbio_hip/aligner.h) on the repeat-rich
3 Gbp genome of tools/nvbowtie_3gbp.py -- the genome the record-for-record comparison with the unchanged nvBowtie runs on -- without
the files and without the reference's binary: where the time of a batch goes (stage clock), how many rounds / extensions / DP jobs it
takes, the rate with 1 / 2 / 4 batches in flight, and whether every batch's (best, mapq) equals the Python driver's (which
tests/test_ref_tests_gpu.py pins to the unchanged nvBowtie record for record).  GPU box only.

    python tools/own_driver_3gbp.py [--genome 3e9] [--reads 5000000] [--batch 1048576] [--repeats 0.6] [--workers 1,2,4] [--check] [--json OUT]
    rocprofv3 --kernel-trace --stats ... -- python tools/own_driver_3gbp.py --workers 1 --no-stage-clock
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))

STAGES = ("map", "select_init", "select", "locate", "score", "reduce", "mapq", "traceback", "finish")


def setup(genome, n_reads, repeats, seed, dev, index="line_native"):
    """-> (index, genome_words, sym [n, L], qual [n, L])"""
    import nvbowtie_3gbp as T
    from nvbio_amd import workloads as W
    text, placed = T.make_genome(genome, repeats, seed, dev)
    n_seq = 24
    lens = [genome // n_seq] * (n_seq - 1); lens.append(genome - sum(lens))
    bounds = [0]
    for l in lens:
        bounds.append(bounds[-1] + l)
    fmi = W.build_fm_index(text)
    genome_words = W._pack_chunked(text, 2, True)
    sym, qual, pos = T.make_reads(text, n_reads, 100, seed + 1, bounds, dev)
    del text
    torch.cuda.empty_cache()
    desc = {"line_native": False, "ktab_k": 0, "sa_int": fmi.sa_int, "policy": "lean"}
    if index == "line_native":
        fmi = fmi.with_dimer(); desc["line_native"] = True
    elif index == "default":
        fmi, desc = fmi.hbm_default()
    return fmi, genome_words, sym, qual, placed, desc


def shim_lib():
    return C.CDLL(os.path.join(ROOT, "tests", "cxx", "libaligner_shim.so"))


def pack_batches(sym, qual, batch, dev):
    """per batch: (reversed set, fw+rc words, quality stream); names are shared per batch index range"""
    from nvbio_amd import pipeline as P, aligner as A
    out = []
    n, L = sym.shape
    for s in range(0, n - batch + 1, batch):
        sb = sym[s:s + batch].contiguous()
        rev, fwrc = P.pack_read_streams(sb)
        out.append((rev, fwrc, A._qual_stream(batch, L, 30, qual[s:s + batch], dev), s))
    return out


def death_probe(fmi, genome_words, ng, sym, qual, prm, dev, rows=(20, 40, 60, 80)):
    """The Python driver on one batch; every DP job is also scored over its first K pattern rows (SEMI_GLOBAL: the best cell of row K - 1).
    In end-to-end mode no substitution score is positive, so a job whose row-K best is <= its min_score (the read's second-best score when
    the window was set up) ends <= min_score: the reduction would do the same with any such score (reduce_inl.h:111-135)."""
    from nvbio_amd import aligner as A, select as SEL
    from nvbio_amd.strings import PackedStringSet
    n, L = sym.shape
    tally = dict(jobs=0, final_le_min=0, final_le_min_mid_rounds=0, **{"dead_by_row_%d" % k: 0 for k in rows})
    per_round = []
    last = {}
    real_setup, real_score = SEL.score_best_setup, A.batch_banded_alignment_score

    def setup(*args, **kw):
        r = real_setup(*args, **kw)
        last["min_score"] = r[4]
        return r

    def score(band_len, aligner, patterns, texts, **kw):
        s, k = real_score(band_len, aligner, patterns, texts, **kw)
        ms = last["min_score"]
        nj = s.numel()
        row = dict(jobs=nj, final_le_min=int((s <= ms).sum().item()))
        for K in rows:
            pl = torch.full((nj,), K, dtype=torch.int32, device=dev)
            pk = PackedStringSet(patterns.words, 4, True, patterns.begin, pl, 0)
            sk, _ = real_score(band_len, aligner, pk, texts, max_pattern_length=K, quals=kw.get("quals"))
            row["dead_by_row_%d" % K] = int((sk <= ms).sum().item())
        for kk, v in row.items():
            tally[kk] += v
        per_round.append(row)
        return s, k

    SEL.score_best_setup, A.batch_banded_alignment_score = setup, score
    try:
        index = torch.arange(0, (n + 1) * L, L, dtype=torch.int64, device=dev)
        rb = A.ReadBatch.from_ragged(sym.reshape(-1), index, qual.reshape(-1))
        r = A.best_approx(fmi, None, rb, genome_words, ng, prm, names=["r%08d" % i for i in range(n)], cigar_stride=64, finish=False)
    finally:
        SEL.score_best_setup, A.batch_banded_alignment_score = real_setup, real_score
    tally.pop("final_le_min_mid_rounds")
    tally["rounds"] = len(per_round)
    tally["per_round_sample"] = per_round[:40:3] + per_round[40::10]
    return tally


def duplicate_probe(fmi, genome_words, ng, sym, qual, prm, dev):
    """The Python driver on one batch, recording every DP job's (pattern begin = read and strand, window begin): how many of a batch's DP
    jobs repeat a job the same read already ran (several seeds of a read point at one placement: same diagonal, same window)."""
    from nvbio_amd import aligner as A, select as SEL
    n, L = sym.shape
    keys, per_round = [], []
    memo = {K: torch.full((n, K), -1, dtype=torch.int64, device=dev) for K in (1, 2, 4, 8, 16)}
    memo_hits = {K: 0 for K in memo}
    real_setup = SEL.score_best_setup

    def setup(*args, **kw):
        r = real_setup(*args, **kw)
        pb, tb, tl = r[0], r[2], r[3]
        live = tl != 0
        k = (pb[live].to(torch.int64) << 33) | (tb[live].to(torch.int64) & 0x1FFFFFFFF)
        keys.append(k)
        per_round.append((int(k.numel()), int(torch.unique(k).numel())))
        # what a small per-read memo of scored windows would answer: K direct-mapped slots per read, looked up and refilled round by round
        rco = int(kw.get("rc_offset") or n * L)
        rd = (pb[live].to(torch.int64) % rco) // L
        for K in memo:
            slot = ((tb[live].to(torch.int64) * 0x9E3779B1) >> 12) % K
            cell = memo[K][rd, slot]
            memo_hits[K] += int((cell == k).sum().item())
            memo[K][rd, slot] = k
        return r

    SEL.score_best_setup = setup
    try:
        index = torch.arange(0, (n + 1) * L, L, dtype=torch.int64, device=dev)
        rb = A.ReadBatch.from_ragged(sym.reshape(-1), index, qual.reshape(-1))
        A.best_approx(fmi, None, rb, genome_words, ng, prm, names=["r%08d" % i for i in range(n)], cigar_stride=64, finish=False)
    finally:
        SEL.score_best_setup = real_setup
    allk = torch.cat(keys)
    return dict(jobs=int(allk.numel()), distinct=int(torch.unique(allk).numel()), distinct_within_rounds=sum(u for _, u in per_round), rounds=len(per_round), memo_hits_by_slots=memo_hits,
                per_round_sample=per_round[:12] + per_round[12::10])


def measure(genome=3_000_000_000, reads=5_000_000, batch=1 << 20, repeats=0.6, index="default", workers=(1, 2), check=False, stage_clock=True, reps=2,
            probe=False, verbose=False, lean_check=False):
    """-> dict.  check: batch 0's (best, mapq) against the Python driver's; lean_check: every batch's against the same C++ driver on the index in
    the reference's layout (bwt|occ records, SA every 16 rows)."""
    import bench as B
    from nvbio_amd import aligner as A, select as SEL
    import nvbio_amd as nvb
    dev = torch.device("cuda:0")
    t0 = time.time()
    fmi, genome_words, sym, qual, placed, index_desc = setup(int(genome), reads, repeats, 0x5EED0009, dev, "reference" if lean_check else index)
    lean = fmi
    if lean_check:
        index_desc = {"line_native": False, "ktab_k": 0, "sa_int": fmi.sa_int, "policy": "lean"}
        if index == "line_native":
            fmi = lean.with_dimer(); index_desc["line_native"] = True
        elif index == "default":
            fmi, index_desc = lean.hbm_default()
    torch.cuda.synchronize()
    out = dict(genome=int(genome), reads=reads, batch=batch, repeats=repeats, index=index, index_built=index_desc, setup_s=time.time() - t0,
               repeat_families=[dict(length=L, copies=c) for L, c in placed], driver="nvbio::bowtie2::cuda::Aligner::best_approx (C++, include/nvbio_hip/aligner.h)")
    n, L = sym.shape
    batches = pack_batches(sym, qual, batch, dev)
    nb = len(batches)
    prm = A.Params(batch_size=batch)
    scheme = nvb.SmithWatermanScoringScheme()
    sp = B._shim_params(prm, scheme)
    sp.finish = 0 if (check or lean_check) else 1        # finish_alignment rewrites best_data (window begin, final score): the comparisons are of the extension-stage words
    shim = shim_lib()
    fs = fmi.struct()
    vp = lambda t: C.c_void_p(t.data_ptr())
    ptrs = lambda ts: (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    ng = int(genome)
    # every batch uses its own names r%08d (the randomized selection is seeded by them)
    name_arenas = [SEL.pack_names(["r%08d" % i for i in range(b[3], b[3] + batch)], dev) for b in batches]

    # ---- one batch with the stage clock
    if stage_clock:
        rev, fwrc, qs, s0 = batches[0]
        best = torch.zeros((2, batch), dtype=torch.int64, device=dev); mapq = torch.zeros(batch, dtype=torch.uint8, device=dev)
        ms, stage, stats = (C.c_double * 1)(), (C.c_double * 9)(), (C.c_uint64 * 4)()
        torch.cuda.synchronize(); torch.cuda.empty_cache()
        rc = shim.nvbio_aligner_best_approx_timed(C.byref(fs), None, C.c_uint32(batch), C.c_uint32(L), vp(rev.words), C.c_uint64(rev.words.numel()), vp(rev.begin),
                                                  vp(fwrc), C.c_uint64(fwrc.numel()), vp(qs), C.c_uint64(qs.numel()), vp(name_arenas[0][0]), vp(name_arenas[0][1]),
                                                  vp(genome_words), C.c_uint64(genome_words.numel()), C.c_uint32(ng), C.byref(sp), C.c_uint32(reps), ms, stage, vp(best), vp(mapq), stats)
        assert rc == 0
        loc = (best[0] >> 32) & 0xFFFFFFFF
        out["one_batch"] = dict(ms_per_batch=ms[0], Mreads_per_s=batch / ms[0] / 1e3, extensions=int(stats[0]), rounds=int(stats[1]), seeding_passes=int(stats[2]),
                                dp_jobs=int(stats[3]), aligned=float((loc != 0xFFFFFFFF).float().mean().item()),
                                stage_ms_with_syncs={k: round(stage[i], 3) for i, k in enumerate(STAGES)}, stage_ms_sum=round(sum(stage), 3))
        if check:
            index_t = torch.arange(0, (batch + 1) * L, L, dtype=torch.int64, device=dev)
            rb = A.ReadBatch.from_ragged(sym[:batch].reshape(-1), index_t, qual[:batch].reshape(-1))
            t1 = time.time()
            r = A.best_approx(fmi, None, rb, genome_words, ng, prm, names=["r%08d" % i for i in range(batch)], cigar_stride=64, finish=False)
            torch.cuda.synchronize()
            out["python_driver"] = dict(s_per_batch=time.time() - t1, extensions=r["stats"]["extensions"], rounds=r["stats"]["rounds"],
                                        identical_best=bool(torch.equal(r["best"], best)), identical_mapq=bool(torch.equal(r["mapq"], mapq)))
            del r, rb
        if verbose:
            print(json.dumps(out["one_batch"]), flush=True)

    if os.environ.get("OWN_DRIVER_DUP_PROBE"):
        out["duplicate_probe"] = duplicate_probe(fmi, genome_words, ng, sym[:batch], qual[:batch], prm, dev)
        print(json.dumps(out["duplicate_probe"]), flush=True)
    if probe:
        out["death_probe"] = death_probe(fmi, genome_words, ng, sym[:batch], qual[:batch], prm, dev)
        if verbose:
            print(json.dumps(out["death_probe"]), flush=True)

    # ---- all batches, w in flight (every batch with its own qualities and names)
    def run_all(fstruct, w, r):
        best = [torch.zeros((2, batch), dtype=torch.int64, device=dev) for _ in range(nb)]
        mapq = [torch.zeros(batch, dtype=torch.uint8, device=dev) for _ in range(nb)]
        ms = (C.c_double * 1)()
        torch.cuda.synchronize(); torch.cuda.empty_cache()
        rc = shim.nvbio_aligner_best_approx_pipelined_names(
            C.byref(fstruct), None, C.c_uint32(batch), C.c_uint32(L), C.c_uint32(nb),
            ptrs([b[0].words for b in batches]), C.c_uint64(batches[0][0].words.numel()), ptrs([b[0].begin for b in batches]),
            ptrs([b[1] for b in batches]), C.c_uint64(batches[0][1].numel()),
            ptrs([b[2] for b in batches]), C.c_uint64(batches[0][2].numel()), ptrs([x[0] for x in name_arenas]), ptrs([x[1] for x in name_arenas]),
            vp(genome_words), C.c_uint64(genome_words.numel()), C.c_uint32(ng), C.byref(sp), C.c_uint32(w), C.c_uint32(r), ms, ptrs(best), ptrs(mapq))
        torch.cuda.synchronize()
        assert rc == 0
        return ms[0], best, mapq

    out["pipelined"] = {}
    first = None
    for w in workers:
        ms, best, mapq = run_all(fs, w, reps)
        loc = [(x[0] >> 32) & 0xFFFFFFFF for x in best]
        out["pipelined"][str(w)] = dict(host_threads=w, ms_total=ms, Mreads_per_s=batch * nb / ms / 1e3,
                                        aligned=sum(int((l != 0xFFFFFFFF).sum().item()) for l in loc) / (batch * nb))
        if first is None:
            first = (best, mapq)
        else:
            out["pipelined"][str(w)]["identical_to_serial"] = all(torch.equal(x, y) for x, y in zip(best, first[0])) and all(torch.equal(x, y) for x, y in zip(mapq, first[1]))
        if verbose:
            print(json.dumps({("workers_%d" % w): out["pipelined"][str(w)]}), flush=True)
    out["reads_aligned"] = batch * nb
    out["Mreads_per_s"] = max(v["Mreads_per_s"] for v in out["pipelined"].values()) if out["pipelined"] else None
    if lean_check and first is not None:
        ls = lean.struct()
        _, lb, lm = run_all(ls, 1, 1)
        out["identical_to_reference_layout"] = all(torch.equal(x, y) for x, y in zip(lb, first[0])) and all(torch.equal(x, y) for x, y in zip(lm, first[1]))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome", type=float, default=3e9)
    ap.add_argument("--reads", type=int, default=5_000_000)
    ap.add_argument("--batch", type=int, default=1 << 20, help="reads per batch = Aligner::BATCH_SIZE (nvBowtie's default: 1024 K)")
    ap.add_argument("--repeats", type=float, default=0.6)
    ap.add_argument("--index", default="default", choices=["reference", "line_native", "default"], help="default: FMIndexDevice.hbm_default(), what the loaders build on this device")
    ap.add_argument("--workers", default="1,2,4")
    ap.add_argument("--check", action="store_true", help="compare batch 0's (best, mapq) with the Python driver's")
    ap.add_argument("--lean-check", action="store_true", help="compare every batch with the same driver on the index in the reference's layout")
    ap.add_argument("--no-stage-clock", action="store_true")
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--json", default=None)
    ap.add_argument("--death-probe", action="store_true", help="Python driver on batch 0 with every DP job also scored over its first K rows: how early jobs fall to or below their min_score")
    a = ap.parse_args()
    out = measure(int(a.genome), a.reads, a.batch, a.repeats, a.index, [int(x) for x in a.workers.split(",") if x], a.check, not a.no_stage_clock, a.reps, a.death_probe,
                  verbose=True, lean_check=a.lean_check)
    text = json.dumps(out, indent=1)
    if a.json:
        os.makedirs(os.path.dirname(os.path.abspath(a.json)), exist_ok=True)
        open(a.json, "w").write(text)
    print(text)


if __name__ == "__main__":
    main()
