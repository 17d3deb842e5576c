"""Host restatement (numpy over oracle/) of nvBowtie's single-end best-mapping driver -- Aligner::best_approx /
best_approx_score (nvBowtie/bowtie2/cuda/aligner_best_approx.h:85-520, :522-840) -- written independently of
nvbio_amd/aligner.py.  Test infrastructure: the checker of the composed GPU pipeline."""
import numpy as np

from oracle import pyoracle as O

WORST_SCORE = -(1 << 16)


def band_length(max_dist):
    b = 4
    while b - 1 < 2 * max_dist + 1:
        b *= 2
    return b - 1


def pack_reads(sym):
    """(reversed reads as a 4-bit BE StringSet for the mapper; fw + rc pattern words)"""
    n, L = sym.shape
    begin = np.arange(n, dtype=np.uint64) * L
    lens = np.full(n, L, np.uint32)
    rev = O.StringSet(O.pack(sym[:, ::-1].reshape(-1), 4, True), 4, True, begin, lens)
    rc = np.where(sym > 3, sym, 3 - sym)[:, ::-1]
    ext = O.pack(np.concatenate([sym.reshape(-1), rc.reshape(-1)]), 4, True)
    return rev, ext


def qual_scheme(scheme):
    st = scheme.struct()
    return (st.match, st.pattern_gap_open, st.pattern_gap_ext, st.text_gap_open, st.text_gap_ext, 0), np.array([st.mismatch[q] for q in range(256)], np.int32)


def best_approx(host_fmi, host_rfmi, sym, genome_words, genome_len, params, scheme, names, aln_type, qual_value=30, traceback=True, cigar_stride=64):
    n, L = sym.shape
    band = band_length(params.max_dist)
    reads_rev, ext_words = pack_reads(sym)
    quals = np.full(2 * n * L + 8, qual_value, np.uint8)
    sch6, lut = qual_scheme(scheme)
    read_len = np.full(n, L, np.uint32)
    best = O.init_alignments(read_len, scheme.m_score_min)
    mp = params.mapping_params()
    sf = mp.seed_freq_table(L, "cpu").numpy().view(np.uint32)
    arena, idx = O.pack_names(names)
    stride = params.hits_stride or min(params.max_hits, 128)
    algorithm = 0 if not params.allow_sub else (2 if params.subseed_len == 0 else 1)
    queue = np.arange(n, dtype=np.uint32)
    stats = dict(extensions=0, rounds=0, seeding_passes=0, queue=[])
    for seeding_pass in range(params.max_reseed + 1):
        if queue.size == 0:
            break
        stats["queue"].append(int(queue.size)); stats["seeding_passes"] += 1
        pd = dict(seed_len=mp.seed_len, min_read_len=mp.min_read_len, max_hits=mp.max_hits, max_reseed=mp.max_reseed, retry=seeding_pass,
                  rep_seeds=mp.rep_seeds, fw=int(params.fw), rc=int(params.rc))
        hits, counts, reseed = O.map_seeds(algorithm, params.subseed_len, host_fmi, host_rfmi, reads_rev, pd, sf, stride, in_queue=queue)
        probs, trys, rseeds = O.select_init(hits, counts, arena if params.randomized else None, idx if params.randomized else None,
                                            params.max_effort_init, params.randomized, params.top_seed)
        active = queue | np.uint32((params.top_seed & 1) << 31)
        n_ext = 0
        while active.size and n_ext < params.max_ext:
            n_multi = 1
            if active.size <= params.batch_size // 2 and not params.no_multi_hits:
                n_multi = min(params.batch_size // active.size, min(4096, params.max_ext - n_ext))
            active, hit_begin, rid, loc, seed = O.select(params.randomized, n_multi, active, hits, counts, probs, rseeds, trys)
            if active.size == 0:
                break
            loc = O.locate_hits(host_fmi, host_rfmi, loc, seed)
            tb, tl, _ = O.score_best_setup(rid, loc, read_len, band, genome_len, best, WORST_SCORE)
            rc = (seed >> 13) & 1
            patterns = O.StringSet(ext_words, 4, True, rid.astype(np.uint64) * L + rc.astype(np.uint64) * (n * L), np.full(rid.size, L, np.uint32))
            texts = O.StringSet(genome_words, 2, True, tb, tl)
            score, _ = O.batch_banded_gotoh_score_qual(band, aln_type, sch6, lut, quals, patterns, texts)
            O.score_reduce_best_approx(best, active, hit_begin, score, loc, seed, read_len, WORST_SCORE, trys, counts, n_ext,
                                       params.min_ext, params.max_ext, params.max_effort)
            stats["extensions"] += int(loc.size); stats["rounds"] += 1
            n_ext += n_multi
        aligned = (best[0] >> np.uint64(32)) != np.uint64(0xFFFFFFFF)
        queue = queue[(reseed != 0) | ~aligned[queue]]
    out = dict(best=best, mapq=O.mapq(2, scheme.m_match, scheme.m_score_min, scheme.m_monotone, best, read_len), stats=stats)
    if traceback:
        align = (best[0] >> np.uint64(32)).astype(np.int64)
        ids = np.nonzero(align != 0xFFFFFFFF)[0]
        b_rc = ((best[0] >> np.uint64(28)) & np.uint64(1)).astype(np.int64)
        tbeg = np.maximum(align[ids] - band // 2, 0)
        tend = np.minimum(tbeg + L + band, genome_len)
        pat = O.StringSet(ext_words, 4, True, (ids * L + b_rc[ids] * (n * L)).astype(np.uint64), np.full(ids.size, L, np.uint32))
        txt = O.StringSet(genome_words, 2, True, tbeg.astype(np.uint64), (tend - tbeg).astype(np.uint32))
        r = O.batch_banded_gotoh_traceback(band, aln_type, sch6[:5], pat, txt, cigar_stride, mm_lut=lut, quals=quals)
        out.update(aligned_ids=ids, tb=r)
    return out
