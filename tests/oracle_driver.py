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
    """(reversed reads as a 4-bit BE StringSet for the mapper; fw + rc pattern words; offsets int64[n+1]).  `sym`: uint8 [n, L] matrix
    or a list of uint8 arrays (reads of different lengths)."""
    reads = [np.asarray(r, np.uint8) for r in sym]
    lens = np.array([r.size for r in reads], np.uint32)
    index = np.zeros(len(reads) + 1, np.int64); index[1:] = np.cumsum(lens)
    cat = lambda xs: np.concatenate(xs) if len(xs) else np.zeros(0, np.uint8)
    rev = O.StringSet(O.pack(cat([r[::-1] for r in reads]), 4, True), 4, True, index[:-1].astype(np.uint64), lens)
    ext = O.pack(np.concatenate([cat(reads), cat([np.where(r > 3, r, 3 - r)[::-1] for r in reads])]), 4, True)
    return rev, ext, index


def qual_scheme(scheme):
    st = scheme.struct()
    return (st.match, st.pattern_gap_open, st.pattern_gap_ext, st.text_gap_open, st.text_gap_ext, 0), np.array([st.mismatch[q] for q in range(256)], np.int32)


def best_approx(host_fmi, host_rfmi, sym, genome_words, genome_len, params, scheme, names, aln_type, qual_value=30, traceback=True, cigar_stride=64,
                read_quals=None, finish=False, mds_stride=256):
    band = band_length(params.max_dist)
    reads_rev, ext_words, index = pack_reads(sym)
    n, total = index.size - 1, int(index[-1])
    read_len = np.diff(index).astype(np.uint32)
    L = int(read_len.max())                                      # the longest read (sizes the seed-frequency table)
    quals = np.full(2 * total + 8, qual_value, np.uint8)
    if read_quals is not None:                                   # matrix or list of per-read arrays
        qs = [np.asarray(q, np.uint8) for q in read_quals]
        quals = np.concatenate(qs + [q[::-1] for q in qs] + [np.zeros(8, np.uint8)])
    sch6, lut = qual_scheme(scheme)
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
            patterns = O.StringSet(ext_words, 4, True, index[rid].astype(np.uint64) + rc.astype(np.uint64) * total, read_len[rid])
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
        tend = np.minimum(tbeg + read_len[ids].astype(np.int64) + band, genome_len)
        pat = O.StringSet(ext_words, 4, True, (index[ids] + b_rc[ids] * total).astype(np.uint64), read_len[ids])
        txt = O.StringSet(genome_words, 2, True, tbeg.astype(np.uint64), (tend - tbeg).astype(np.uint32))
        r = O.batch_banded_gotoh_traceback(band, aln_type, sch6[:5], pat, txt, cigar_stride, mm_lut=lut, quals=quals)
        out.update(aligned_ids=ids, tb=r)
        if finish:                                       # finish_alignment_best (traceback_inl.h:523-760)
            out["best_scored"] = best.copy()
            mds, mds_len = O.finish_alignment(np.ones(ids.size, np.uint8), pat, quals, txt, r["cigar"][: ids.size], r["cigar_len"], r["source"], scheme.m_match, lut, 1,
                                              best[0], idx=ids, mds_stride=mds_stride, gap_costs=(scheme.pattern_gap_open(), scheme.pattern_gap_extension(), scheme.text_gap_open(), scheme.text_gap_extension()))
            out["mds"] = np.zeros((n, mds_stride), np.uint8); out["mds"][ids] = mds[: ids.size]
            out["mds_len"] = np.zeros(n, np.uint32); out["mds_len"][ids] = mds_len
    return out


def best_approx_paired(host_fmi, host_rfmi, sym1, sym2, genome_words, genome_len, params, scheme, names, aln_type, qual_value=30, cigar_stride=64,
                       finish=False, mds_stride=256):
    """Aligner::best_approx for pairs (aligner_best_approx_paired.h:95-453, :455-700), numpy over the oracle."""
    n, L = sym1.shape
    band = band_length(params.max_dist)
    packed = [pack_reads(sym1)[:2], pack_reads(sym2)[:2]]
    quals = np.full(2 * n * L + 8, qual_value, np.uint8)
    sch6, lut = qual_scheme(scheme)
    read_len = np.full(n, L, np.uint32)
    best = O.init_alignments(read_len, scheme.m_score_min, 0)
    best_o = O.init_alignments(read_len, scheme.m_score_min, 1)
    mp = params.mapping_params()
    sf = mp.seed_freq_table(L, "cpu").numpy().view(np.uint32)
    arena, idx = O.pack_names(names)
    stride = params.hits_stride or min(params.max_hits, 128)
    algorithm = 0 if not params.allow_sub else (2 if params.subseed_len == 0 else 1)
    stats = dict(extensions=0, opposite_extensions=0, rounds=0, seeding_passes=0, queue=[])
    for anchor in (0, 1):
        queue = np.arange(n, dtype=np.uint32)
        fw_strand = params.pe_policy in (0, 1) if anchor == 0 else params.pe_policy in (0, 2)
        fw, rc = (params.fw, params.rc) if fw_strand else (params.rc, params.fw)
        reads_rev, a_words = packed[anchor]
        o_words = packed[1 - anchor][1]
        for seeding_pass in range(params.max_reseed + 1):
            if queue.size == 0:
                break
            stats["queue"].append(int(queue.size)); stats["seeding_passes"] += 1
            pd = dict(seed_len=mp.seed_len, min_read_len=mp.min_read_len, max_hits=mp.max_hits, max_reseed=mp.max_reseed, retry=seeding_pass,
                      rep_seeds=mp.rep_seeds, fw=int(fw), rc=int(rc))
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
                rcb = ((seed >> 13) & 1).astype(np.uint8)
                # anchor_score_best
                tb, tl, ms = O.anchor_score_setup(rid, loc, seed, read_len, read_len, band, genome_len, best, best_o, scheme.m_match, scheme.m_score_min, WORST_SCORE, anchor)
                pat = O.StringSet(a_words, 4, True, rid.astype(np.uint64) * L + rcb.astype(np.uint64) * (n * L), np.full(rid.size, L, np.uint32))
                raw, raw_sink = O.batch_banded_gotoh_score_qual(band, aln_type, sch6, lut, quals, pat, O.StringSet(genome_words, 2, True, tb, tl))
                hit_score, hit_sink = O.anchor_score_finish(raw, raw_sink, tb, ms, WORST_SCORE)
                # opposite_score_best over the hits whose anchor scored
                ow = O.opposite_windows(rid, rcb, loc, hit_score, read_len, read_len, best, best_o, scheme.m_match, scheme.m_score_min, scheme.text_gap_open(),
                                        scheme.text_gap_extension(), params.pe_policy, params.min_frag_len, params.max_frag_len, params.pe_overlap, WORST_SCORE,
                                        anchor, genome_len)
                valid = (ow["valid"] != 0) & (hit_score != WORST_SCORE)
                k = np.nonzero(valid)[0]
                o_score = np.full(rid.size, WORST_SCORE, np.int32); o_score2 = o_score.copy()
                o_loc = np.zeros(rid.size, np.uint32); o_sink = np.zeros(rid.size, np.uint32); o_sink2 = np.zeros(rid.size, np.uint32)
                if k.size:
                    ob, oe = ow["genome_begin"][k].astype(np.uint64), ow["genome_end"][k].astype(np.uint64)
                    o_pat = O.StringSet(o_words, 4, True, rid[k].astype(np.uint64) * L + ow["read_rc"][k].astype(np.uint64) * (n * L), np.full(k.size, L, np.uint32))
                    o_txt = O.StringSet(genome_words, 2, True, ob, (oe - ob).astype(np.uint32))
                    s_o, k_o, _ = O.batch_gotoh_score_qual(0, aln_type, sch6[:5], lut, quals, o_pat, o_txt, min_score=ow["min_score"][k])
                    o_score[k] = np.where(s_o >= ow["min_score"][k], s_o, WORST_SCORE)
                    sx = k_o.reshape(-1, 2)[:, 0]
                    o_loc[k] = ob.astype(np.uint32)
                    o_sink[k] = (ob + np.where(sx == 0xFFFFFFFF, 0, sx)).astype(np.uint32)
                    o_sink2[k] = ob.astype(np.uint32)
                O.score_reduce_paired_best_approx(best, best_o, active, hit_begin, loc, hit_sink, hit_score, seed, o_loc, o_sink, o_sink2, o_score, o_score2,
                                                  read_len, anchor, params.pe_policy, params.pe_unpaired, WORST_SCORE, trys, counts, n_ext,
                                                  params.min_ext, params.max_ext, params.max_effort)
                stats["extensions"] += int(loc.size); stats["opposite_extensions"] += int(k.size); stats["rounds"] += 1
                n_ext += n_multi
            queue = queue[reseed != 0]
    if params.pe_discordant:
        O.mark_discordant(best, best_o)
    mapq1 = O.mapq_paired(2, scheme.m_match, scheme.m_score_min, scheme.m_monotone, best, best_o, read_len, read_len)
    mapq2 = O.mapq_paired(2, scheme.m_match, scheme.m_score_min, scheme.m_monotone, best_o, best, read_len, read_len)
    out = dict(best=best, best_o=best_o, mapq1=mapq1, mapq2=mapq2, stats=stats)
    mate_words = (packed[0][1], packed[1][1])

    if finish:
        out["best_scored"], out["best_o_scored"] = best.copy(), best_o.copy()

    def trace(data, ids, full):
        res = dict(cigar=np.zeros((n, cigar_stride), np.uint16), cigar_len=np.zeros(n, np.uint32), source=np.full((n, 2), 0xFFFFFFFF, np.uint32),
                   sink=np.full((n, 2), 0xFFFFFFFF, np.uint32), score=np.full(n, WORST_SCORE, np.int32),
                   mds=np.zeros((n, mds_stride), np.uint8), mds_len=np.zeros(n, np.uint32))
        for i in ids:
            w = int(data[0][i])
            align, rcb, mate, g_len = w >> 32, (w >> 28) & 1, (w >> 29) & 1, (w >> 18) & 0x3FF
            pat = O.StringSet(mate_words[mate], 4, True, np.array([i * L + rcb * n * L], np.uint64), np.array([L], np.uint32))
            if full(i):
                tb0, tb1 = align, min(align + g_len, genome_len)
                r = O.batch_gotoh_traceback(aln_type, sch6[:5], pat, O.StringSet(genome_words, 2, True, np.array([tb0], np.uint64), np.array([tb1 - tb0], np.uint32)),
                                            cigar_stride, mm_lut=lut, quals=quals)
            else:
                tb0 = max(align - band // 2, 0); tb1 = min(tb0 + L + band, genome_len)
                r = O.batch_banded_gotoh_traceback(band, aln_type, sch6[:5], pat, O.StringSet(genome_words, 2, True, np.array([tb0], np.uint64), np.array([tb1 - tb0], np.uint32)),
                                                   cigar_stride, mm_lut=lut, quals=quals)
            for key in ("cigar", "cigar_len", "source", "sink", "score"):
                res[key][i] = r[key][0]
            if finish:                                   # finish_alignment_best / finish_opposite_alignment_best on this slot
                txt = O.StringSet(genome_words, 2, True, np.array([tb0], np.uint64), np.array([tb1 - tb0], np.uint32))
                m, ml = O.finish_alignment(np.ones(1, np.uint8), pat, quals, txt, r["cigar"][:1], r["cigar_len"], r["source"], scheme.m_match, lut, 1, data[0],
                                           idx=np.array([i], np.uint32), mds_stride=mds_stride, gap_costs=(scheme.pattern_gap_open(), scheme.pattern_gap_extension(), scheme.text_gap_open(), scheme.text_gap_extension()))
                res["mds"][i] = m[0]; res["mds_len"][i] = ml[0]
        return res

    aligned = lambda d: (d[0] >> np.uint64(32)) != np.uint64(0xFFFFFFFF)
    out["tb1"] = trace(best, np.nonzero(aligned(best))[0], lambda i: False)
    if finish:       # mate 2's MAPQ is evaluated after the anchor slots were finished (aligner_best_approx_paired.h:308-323)
        out["mapq2"] = O.mapq_paired(2, scheme.m_match, scheme.m_score_min, scheme.m_monotone, best_o, best, read_len, read_len)
    w_o = best_o[0]
    conc = (((w_o >> np.uint64(30)) & np.uint64(1)) != 0) & (((w_o >> np.uint64(31)) & np.uint64(1)) == 0)
    out["tb2"] = trace(best_o, np.nonzero(aligned(best_o))[0], lambda i: bool(conc[i]))
    return out


def all_mapping(host_fmi, host_rfmi, sym, genome_words, genome_len, params, scheme, aln_type, qual_value=30, cigar_stride=64, mds_stride=256,
                sequence_index=None, read_quals=None, straddle_index="sorted"):
    """Aligner::all / score_all (aligner_all.h:47-694), numpy over the oracle: one mapping pass with every seed, then all rows of all SA
    ranges in batches of params.batch_size hits -- hi-bits sort, locate, (read, strand, position) sort, de-duplication within the batch,
    straddling marks (with the reference's indexing), banded extension, acceptance at min_score(read_len), traceback, finish.  Returns
    the accepted alignments in batch order, sorted by (read, strand, position) inside each batch.
    straddle_index: which index mark_straddling reads.  The reference hands it `pipeline.idx_queue` (aligner_all.h:520), the pointer
    sort_hi_bits returned -- a half of the ping-pong index buffer that sort_64_bits has refilled and sorted through since.  "sorted": both
    sorts ended in the same half, the pointer sees the final (read, strand, position) index (what the radix sort of this image does up to
    ~10^5 hits per batch, i.e. at every size THIS numpy driver is run at; at a full batch of 2^20 hits the half holds the last pass but one,
    which only a replay of the sorts can know: nvbio_hip_sort_hits_pingpong, held against the unchanged nvBowtie in test_ref_tests_gpu.py); "locate": the
    hi-bits index survived (the reading of rounds 4-5)."""
    band = band_length(params.max_dist)
    reads_rev, ext_words, index = pack_reads(sym)
    n, total = index.size - 1, int(index[-1])
    read_len = np.diff(index).astype(np.uint32)
    L = int(read_len.max())
    quals = np.full(2 * total + 8, qual_value, np.uint8)
    if read_quals is not None:
        qs = [np.asarray(q, np.uint8) for q in read_quals]
        quals = np.concatenate(qs + [q[::-1] for q in qs] + [np.zeros(8, np.uint8)])
    sch6, lut = qual_scheme(scheme)
    mp = params.mapping_params()
    sf = mp.seed_freq_table(L, "cpu").numpy().view(np.uint32)
    stride = params.hits_stride or min(params.max_hits, 128)
    algorithm = 1 if params.allow_sub else 0          # map_exact or map_approx -- never case pruning (aligner_all.h:177-212)
    pd = dict(seed_len=mp.seed_len, min_read_len=mp.min_read_len, max_hits=mp.max_hits, max_reseed=mp.max_reseed, retry=0,
              rep_seeds=mp.rep_seeds, fw=int(params.fw), rc=int(params.rc))
    hits, counts, _ = O.map_seeds(algorithm, params.subseed_len, host_fmi, host_rfmi, reads_rev, pd, sf, stride)
    seq_index = np.asarray(sequence_index if sequence_index is not None else [0, genome_len], np.int64)
    min_score = np.array([scheme.min_score(int(l)) if l else 0 for l in range(L + 1)], np.int64)
    # every (read, range k, row) in order: the numbering select_all decodes
    h_read, h_sa, h_seed = [], [], []
    for r in range(n):
        for k in range(int(counts[r])):
            w = int(hits[r, k]); lo, hi = w & 0xFFFFFFFF, w >> 32
            delta, pos, rcb, idir = hi & 0xFFFFF, (hi >> 20) & 0x3FF, (hi >> 30) & 1, (hi >> 31) & 1
            h_read.append(np.full(delta, r, np.uint32)); h_sa.append((lo + np.arange(delta, dtype=np.int64)).astype(np.uint32))
            h_seed.append(np.full(delta, pos | (idir << 12) | (rcb << 13), np.uint32))
    cat = lambda xs, t: np.concatenate(xs) if xs else np.zeros(0, t)
    h_read, h_sa, h_seed = cat(h_read, np.uint32), cat(h_sa, np.uint32), cat(h_seed, np.uint32)
    n_hits = h_read.size
    out_aln, out_read, stats = [], [], dict(hits=int(n_hits), ranges=int(counts.sum()), unique=0)
    B = params.batch_size
    for off in range(0, n_hits, B):
        rid, sa, seed = h_read[off:off + B], h_sa[off:off + B], h_seed[off:off + B]
        cnt = rid.size
        idx_queue = np.argsort(sa >> 16, kind="stable")                               # sort_hi_bits (uint16 keys, stable radix sort)
        loc = O.locate_hits(host_fmi, host_rfmi, sa.copy(), seed)
        key = loc.astype(np.uint64) + (rid.astype(np.uint64) << np.uint64(33)) + (((seed >> 13) & 1).astype(np.uint64) << np.uint64(32))
        sidx = np.argsort(key, kind="stable")
        skey = key[sidx]
        flags = np.ones(cnt, bool); flags[1:] = skey[1:] != skey[:-1]
        g = loc[sidx if straddle_index == "sorted" else idx_queue].astype(np.int64)     # mark_straddling: hit idx[t], flag t
        s0 = np.searchsorted(seq_index, g, side="right") - 1
        s1 = np.searchsorted(seq_index, (g + params.seed_len) & 0xFFFFFFFF, side="right") - 1
        flags[s0 != s1] = False
        q = sidx[flags]
        stats["unique"] += int(q.size)
        if q.size == 0:
            continue
        pos = loc[q].astype(np.int64); rl = read_len[rid[q]].astype(np.int64); rcq = ((seed[q] >> 13) & 1).astype(np.int64)
        tbeg = np.where(pos > band // 2, pos - band // 2, 0)
        tend = np.minimum((tbeg + band + rl) & 0xFFFFFFFF, genome_len)
        tl = np.maximum(tend - tbeg, 0)
        pat = O.StringSet(ext_words, 4, True, (index[rid[q]] + rcq * total).astype(np.uint64), read_len[rid[q]])
        txt = O.StringSet(genome_words, 2, True, tbeg.astype(np.uint64), tl.astype(np.uint32))
        score, _ = O.batch_banded_gotoh_score_qual(band, aln_type, sch6, lut, quals, pat, txt)
        ok = score.astype(np.int64) >= min_score[rl]
        s = score[ok].astype(np.int64)
        w = (s < 0).astype(np.uint64) | ((np.abs(s).astype(np.uint64) & np.uint64(0x1FFFF)) << np.uint64(1)) | (rcq[ok].astype(np.uint64) << np.uint64(28))
        out_aln.append((pos[ok].astype(np.uint64) << np.uint64(32)) | w); out_read.append(rid[q][ok])
    aln, arid = cat(out_aln, np.uint64), cat(out_read, np.uint32)
    out = dict(alignments_scored=aln.copy(), read_id=arid, stats=stats)
    m = aln.size
    if m:
        pos = (aln >> np.uint64(32)).astype(np.int64); rcq = ((aln >> np.uint64(28)) & np.uint64(1)).astype(np.int64); rl = read_len[arid].astype(np.int64)
        tbeg = np.where(pos > band // 2, pos - band // 2, 0)
        tend = np.minimum((tbeg + band + rl) & 0xFFFFFFFF, genome_len)
        pat = O.StringSet(ext_words, 4, True, (index[arid] + rcq * total).astype(np.uint64), read_len[arid])
        txt = O.StringSet(genome_words, 2, True, tbeg.astype(np.uint64), np.maximum(tend - tbeg, 0).astype(np.uint32))
        r = O.batch_banded_gotoh_traceback(band, aln_type, sch6[:5], pat, txt, cigar_stride, mm_lut=lut, quals=quals)
        fin = aln.copy()
        mds, mds_len = O.finish_alignment(np.ones(m, np.uint8), pat, quals, txt, r["cigar"][:m], r["cigar_len"], r["source"], scheme.m_match, lut, 1, fin,
                                          mds_stride=mds_stride, gap_costs=(scheme.pattern_gap_open(), scheme.pattern_gap_extension(), scheme.text_gap_open(), scheme.text_gap_extension()))
        out.update(alignments=fin, tb=r, mds=mds[:m], mds_len=mds_len)
    return out
