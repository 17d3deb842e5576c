"""A minimal single-end seed-and-extend driver over the hot-path entry points: map_exact -> (take up
to `rows_per_hit` SA rows of every seed hit) -> locate -> banded Gotoh extension of the read against
the genome window around each located seed -> best score per read.

This is glue (torch tensor ops), not part of the hot path and not nvBowtie's selection / reduction
policy (out of scope); it exists to exercise and time the composed kernels on
BASELINE config 4's shape.  `backend` supplies the three hot-path calls so that the tests can run
the identical glue over a host checker."""
import torch

from .strings import PackedStringSet
from .workloads import _pack_chunked


class HipBackend:
    """The product path: every call goes to libnvbio_hip.so."""

    def __init__(self, fmi, genome, map_params, max_read_len):
        from . import mapping, fmindex, alignment
        self._m, self._f, self._a = mapping, fmindex, alignment
        self.fmi, self.genome, self.map_params, self.max_read_len = fmi, genome, map_params, max_read_len

    def map_exact(self, reads_rev, hits_stride):
        return self._m.map_exact(self.fmi, reads_rev, self.map_params, self.max_read_len, hits_stride=hits_stride)[:2]

    def locate(self, rows):
        return self._f.locate(self.fmi, rows)

    def score(self, band, aligner, patterns, texts):
        return self._a.batch_banded_alignment_score(band, aligner, patterns, texts)

    # the stages after scoring (align_single_end)
    def init_best(self, n, scheme, read_len):
        from . import reduce
        return reduce.BestAlignments(n, scheme, fixed_read_len=read_len, device=self.fmi.bwt_occ.device).data

    def reduce(self, best, hit_begin, score, loc, rc, read_len):
        from . import reduce
        b = reduce.BestAlignments.__new__(reduce.BestAlignments)
        b.n, b.stride, b.data = best.shape[1], best.shape[1], best
        reduce.score_reduce(b, hit_begin, score, loc, rc, fixed_read_len=read_len)
        return best

    def mapq(self, best, scheme, read_len, version=2):
        from . import reduce
        b = reduce.BestAlignments.__new__(reduce.BestAlignments)
        b.n, b.stride, b.data = best.shape[1], best.shape[1], best
        return reduce.mapq(b, scheme, fixed_read_len=read_len, version=version)

    def traceback(self, band, aligner, patterns, texts, cigar_stride):
        return self._a.batch_banded_alignment_traceback(band, aligner, patterns, texts, cigar_stride=cigar_stride)

    # the paired-end stages (align_paired_end)
    def _best(self, data):
        from . import reduce
        b = reduce.BestAlignments.__new__(reduce.BestAlignments)
        b.n, b.stride, b.data = data.shape[1], data.shape[1], data
        return b

    def init_best_mate(self, n, scheme, read_len, mate):
        from . import reduce
        return reduce.BestAlignments(n, scheme, fixed_read_len=read_len, device=self.fmi.bwt_occ.device, mate=mate).data

    def score_qual(self, band, aligner, patterns, texts, quals):
        return self._a.batch_banded_alignment_score(band, aligner, patterns, texts, quals=quals)

    def opposite_windows(self, read_id, rc, loc, score, best, best_o, scheme, anchor, genome_len, read_len, **kw):
        from . import reduce
        return reduce.opposite_mate_windows(read_id, rc, loc, score, self._best(best), self._best(best_o), scheme, anchor, genome_len,
                                            a_fixed_len=read_len, o_fixed_len=read_len, **kw)

    def full_score_qual(self, aligner, patterns, texts, max_m, max_n, min_score, quals):
        return self._a.batch_alignment_score(aligner, patterns, texts, max_m, max_n, min_score, quals=quals)

    def reduce_paired(self, best, best_o, hit_begin, loc, sink, score, rc, o_loc, o_sink, o_sink2, o_score, o_score2, anchor, read_len, **kw):
        from . import reduce
        reduce.score_reduce_paired(self._best(best), self._best(best_o), hit_begin, loc, sink, score, rc, o_loc, o_sink, o_sink2, o_score, o_score2,
                                   anchor, fixed_read_len=read_len, **kw)

    def mapq_paired(self, best, best_o, scheme, read_len):
        from . import reduce
        return reduce.mapq_paired(self._best(best), self._best(best_o), scheme, fixed_read_len=read_len, o_fixed_read_len=read_len)


def make_reads(text, n, read_len=100, seed=0x5EED0004, sub_rate=0.04):
    """Single-end reads sampled from the genome, half of them from the reverse strand, with
    substitutions.  Returns (fw symbols [n,L] uint8, true position int64[n], is_rc bool[n])."""
    dev = text.device
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    pos = torch.randint(0, text.numel() - read_len, (n,), generator=g, device=dev)
    sym = text[pos.unsqueeze(1) + torch.arange(read_len, device=dev).unsqueeze(0)]
    sub = torch.rand((n, read_len), generator=g, device=dev) < sub_rate
    delta = torch.randint(1, 4, (n, read_len), dtype=torch.uint8, generator=g, device=dev)
    sym = torch.where(sub, (sym + delta) & 3, sym)
    is_rc = torch.rand(n, generator=g, device=dev) < 0.5
    rc_sym = (3 - sym).flip(1)
    sym = torch.where(is_rc.unsqueeze(1), rc_sym, sym)
    return sym, pos, is_rc


def pack_read_streams(sym):
    """The three orientations the stages need, all 4-bit big-endian (nvBowtie's read format):
    reversed (what the mapping stage scans, io::REVERSE), forward and reverse-complement (patterns of
    the extension stage for STANDARD / COMPLEMENT hits)."""
    n, L = sym.shape
    dev = sym.device
    idx = torch.arange(n, dtype=torch.int64, device=dev) * L
    rev = PackedStringSet(_pack_chunked(sym.flip(1).reshape(-1), 4, True), 4, True, idx, None, L)
    both = torch.cat([sym.reshape(-1), torch.where(sym > 3, sym, 3 - sym).flip(1).reshape(-1)])       # complement_functor<4>: N stays N
    ext_words = _pack_chunked(both, 4, True)
    return rev, ext_words


def _first_occurrences(jr, jrc, loc):
    """Indices (ascending) of the first job of every distinct (read, strand, read start): several seeds of one read usually locate
    to the same placement, and extending it once is enough (the reference gets the same effect from extending one hit per round and
    skipping recorded locations, score_best / reduce_inl.h:111-114)."""
    key = (jr << 33) | (jrc << 32) | loc
    uniq, inv = torch.unique(key, return_inverse=True)
    first = torch.full((uniq.numel(),), key.numel(), dtype=torch.int64, device=key.device)
    first = first.scatter_reduce(0, inv, torch.arange(key.numel(), device=key.device), reduce="amin", include_self=True)
    return torch.sort(first).values


def seed_and_extend(backend, sym, genome_words, genome_len, band=15, rows_per_hit=2, hits_stride=16, aligner=None, packed=None):
    """Returns (best_score int32[n], best_pos int64[n] (window begin of the best job, -1 if none),
    n_jobs).  `packed` = pack_read_streams(sym) when the caller keeps the packed reads resident."""
    from .alignment import make_gotoh_aligner, SimpleGotohScheme, SEMI_GLOBAL
    n, L = sym.shape
    dev = sym.device
    reads_rev, ext_words = packed if packed is not None else pack_read_streams(sym)
    hits, counts = backend.map_exact(reads_rev, hits_stride)
    # flatten the hits, expand each into up to rows_per_hit SA rows
    k = torch.arange(hits.shape[1], device=dev).unsqueeze(0)
    valid = k < (counts.to(torch.int64) & 0xFFFFFFFF).unsqueeze(1)
    read_id = torch.arange(n, device=dev).unsqueeze(1).expand_as(hits)[valid]
    w = hits[valid]
    lo, hi = w & 0xFFFFFFFF, (w >> 32) & 0xFFFFFFFF
    delta, pir, rc = hi & 0xFFFFF, (hi >> 20) & 0x3FF, (hi >> 30) & 1
    take = torch.clamp(delta, max=rows_per_hit)
    rep = torch.repeat_interleave(torch.arange(w.numel(), device=dev), take)
    first = torch.cumsum(take, 0) - take
    row = lo[rep] + (torch.arange(rep.numel(), device=dev) - first[rep])
    gpos = backend.locate(row.to(torch.int32)).to(torch.int64) & 0xFFFFFFFF
    # window of the read around the located seed (the arithmetic of score_best_inl.h:95-126)
    jr, jrc = read_id[rep], rc[rep]
    sel = _first_occurrences(jr, jrc, torch.clamp(gpos - pir[rep], min=0))
    jr, jrc, gpos, rep = jr[sel], jrc[sel], gpos[sel], rep[sel]
    wbeg = torch.clamp(gpos - pir[rep] - band // 2, min=0)
    wend = torch.clamp(wbeg + L + band, max=genome_len)
    patterns = PackedStringSet(ext_words, 4, True, (jr * L + jrc * (n * L)).contiguous(), None, L)
    texts = PackedStringSet(genome_words, 2, True, wbeg.contiguous(), (wend - wbeg).to(torch.int32).contiguous(), 0)
    if aligner is None:
        aligner = make_gotoh_aligner(SEMI_GLOBAL, SimpleGotohScheme(0, -6, -8, -3))
    score, _ = backend.score(band, aligner, patterns, texts)
    # best job per read; ties broken toward the smallest window begin so that the result is order-free
    key = (score.to(torch.int64) + (1 << 31)) * (1 << 32) + ((1 << 32) - 1 - wbeg)
    best = torch.full((n,), -1, dtype=torch.int64, device=dev)
    best = best.scatter_reduce(0, jr, key, reduce="amax", include_self=True)
    has = best >= 0
    best_score = torch.where(has, (best >> 32) - (1 << 31), torch.full_like(best, -(1 << 30))).to(torch.int32)
    best_pos = torch.where(has, (1 << 32) - 1 - (best & 0xFFFFFFFF), torch.full_like(best, -1))
    return best_score, best_pos, int(rep.numel())


def align_single_end(backend, sym, genome_words, genome_len, band=15, rows_per_hit=2, hits_stride=16, aligner=None,
                     mapq_scheme=None, packed=None, cigar_stride=32):
    """seed -> locate -> extend -> score_reduce (best / second best, nvBowtie's rule) -> MAPQ (BowtieMapq2) ->
    banded traceback of the best alignment: the single-end stages of nvBowtie's best-approx driver
    (aligner_best_approx.h:522-840) minus its hit-selection heuristics -- every located row is extended, in
    (read, sorted hit, row) order.  Returns dict(best int64[2,n] io::Alignment words, mapq uint8[n],
    cigar int16[n,stride], cigar_len int32[n], source int32[n,2], sink int32[n,2], n_jobs)."""
    from .alignment import make_gotoh_aligner, SimpleGotohScheme, SmithWatermanScoringScheme, SEMI_GLOBAL
    n, L = sym.shape
    dev = sym.device
    reads_rev, ext_words = packed if packed is not None else pack_read_streams(sym)
    if aligner is None:
        aligner = make_gotoh_aligner(SEMI_GLOBAL, SimpleGotohScheme(0, -6, -8, -3))
    if mapq_scheme is None:
        mapq_scheme = SmithWatermanScoringScheme()              # end-to-end: perfect score 0, min score -0.6 - 0.6 L
    hits, counts = backend.map_exact(reads_rev, hits_stride)
    k = torch.arange(hits.shape[1], device=dev).unsqueeze(0)
    valid = k < (counts.to(torch.int64) & 0xFFFFFFFF).unsqueeze(1)
    # a deterministic extension order: each read's hits sorted by their words (the reference's order is run dependent)
    hits = torch.where(valid, hits, torch.full_like(hits, (1 << 63) - 1))
    hits, _ = torch.sort(hits, dim=1)
    read_id = torch.arange(n, device=dev).unsqueeze(1).expand_as(hits)[valid.sum(1, keepdim=True) > k]
    w = hits[valid.sum(1, keepdim=True) > k]
    lo, hi = w & 0xFFFFFFFF, (w >> 32) & 0xFFFFFFFF
    delta, pir, rc = hi & 0xFFFFF, (hi >> 20) & 0x3FF, (hi >> 30) & 1
    take = torch.clamp(delta, max=rows_per_hit)
    rep = torch.repeat_interleave(torch.arange(w.numel(), device=dev), take)
    first = torch.cumsum(take, 0) - take
    row = lo[rep] + (torch.arange(rep.numel(), device=dev) - first[rep])
    gpos = backend.locate(row.to(torch.int32)).to(torch.int64) & 0xFFFFFFFF
    jr, jrc = read_id[rep], rc[rep]
    sel = _first_occurrences(jr, jrc, torch.clamp(gpos - pir[rep], min=0))
    jr, jrc, gpos, rep = jr[sel], jrc[sel], gpos[sel], rep[sel]
    wbeg = torch.clamp(gpos - pir[rep] - band // 2, min=0)
    wend = torch.clamp(wbeg + L + band, max=genome_len)
    pat_begin = (jr * L + jrc * (n * L)).contiguous()
    patterns = PackedStringSet(ext_words, 4, True, pat_begin, None, L)
    texts = PackedStringSet(genome_words, 2, True, wbeg.contiguous(), (wend - wbeg).to(torch.int32).contiguous(), 0)
    score, _ = backend.score(band, aligner, patterns, texts)
    # reduce: jobs are grouped by read (jr is non-decreasing); results below the threshold never beat the initial worst score
    per_read = torch.bincount(jr, minlength=n)
    hit_begin = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    hit_begin[1:] = torch.cumsum(per_read, 0)
    best = backend.init_best(n, mapq_scheme, L)
    best = backend.reduce(best, hit_begin, score.contiguous(), wbeg.to(torch.int32).contiguous(), jrc.to(torch.uint8).contiguous(), L)
    mapq = backend.mapq(best, mapq_scheme, L)
    # traceback of the best alignment of every aligned read (traceback_inl.h: window = the scored window)
    b_align = (best[0] >> 32) & 0xFFFFFFFF
    aligned = b_align != 0xFFFFFFFF
    b_rc = (best[0] >> 28) & 1
    ids = torch.nonzero(aligned).squeeze(1)
    tb_begin = b_align[ids]
    tb_end = torch.clamp(tb_begin + L + band, max=genome_len)
    tb_pat = PackedStringSet(ext_words, 4, True, (ids * L + b_rc[ids] * (n * L)).contiguous(), None, L)
    tb_txt = PackedStringSet(genome_words, 2, True, tb_begin.contiguous(), (tb_end - tb_begin).to(torch.int32).contiguous(), 0)
    tb = backend.traceback(band, aligner, tb_pat, tb_txt, cigar_stride)
    cigar = torch.zeros((n, cigar_stride), dtype=torch.int16, device=dev)
    cigar_len = torch.zeros(n, dtype=torch.int32, device=dev)
    source = torch.full((n, 2), -1, dtype=torch.int32, device=dev)
    sink = torch.full((n, 2), -1, dtype=torch.int32, device=dev)
    cigar[ids] = tb["cigar"][: ids.numel()]; cigar_len[ids] = tb["cigar_len"]; source[ids] = tb["source"]; sink[ids] = tb["sink"]
    return dict(best=best, mapq=mapq, cigar=cigar, cigar_len=cigar_len, source=source, sink=sink, tb_score=tb["score"], aligned_ids=ids,
                n_jobs=int(rep.numel()))


def make_read_pairs(text, n, read_len=150, seed=0x5EED0005, sub_rate=0.03, frag=(250, 450)):
    """FR pairs: mate 1 = the fragment's first read_len bases (forward strand), mate 2 = the reverse complement of its last
    read_len bases; half of the fragments come from the reverse strand.  -> (sym1, sym2 [n,L] uint8, frag_pos int64[n], frag_len)"""
    dev = text.device
    g = torch.Generator(device=dev); g.manual_seed(seed)
    flen = torch.randint(frag[0], frag[1] + 1, (n,), generator=g, device=dev)
    pos = torch.randint(0, text.numel() - frag[1] - 1, (n,), generator=g, device=dev)
    ar = torch.arange(read_len, device=dev).unsqueeze(0)
    left = text[pos.unsqueeze(1) + ar]
    right = text[(pos + flen - read_len).unsqueeze(1) + ar]
    def mutate(x):
        sub = torch.rand(x.shape, generator=g, device=dev) < sub_rate
        return torch.where(sub, (x + torch.randint(1, 4, x.shape, dtype=torch.uint8, generator=g, device=dev)) & 3, x)
    m1, m2 = mutate(left), mutate((3 - right).flip(1))
    swap = torch.rand(n, generator=g, device=dev) < 0.5          # fragment from the reverse strand: the mates trade places
    sym1 = torch.where(swap.unsqueeze(1), m2, m1)
    sym2 = torch.where(swap.unsqueeze(1), m1, m2)
    return sym1, sym2, pos, flen


def align_paired_end(backend, sym1, sym2, genome_words, genome_len, scheme=None, band=31, rows_per_hit=2, hits_stride=16,
                     min_frag_len=0, max_frag_len=500, qual_value=30):
    """BASELINE config 5's stage sequence for FR pairs, per anchor mate (aligner_best_approx_paired.h): seed + locate the anchor,
    banded LOCAL extension in nvBowtie's quality-aware scheme, opposite-mate windows and thresholds, full-matrix scoring of the
    opposite mate, paired reduction; then paired MAPQ.  Hit selection is not the reference's (every located row is extended, in
    (read, sorted hit, row) order).  Returns dict(best, best_o int64[2,n] io::Alignment words, mapq uint8[n], n_jobs, n_opposite)."""
    from .alignment import make_gotoh_aligner, SmithWatermanScoringScheme, LOCAL, PATTERN_BLOCKING
    n, L = sym1.shape
    dev = sym1.device
    scheme = scheme or SmithWatermanScoringScheme.local()
    banded_aligner = make_gotoh_aligner(LOCAL, scheme)
    full_aligner = make_gotoh_aligner(LOCAL, scheme, PATTERN_BLOCKING)
    worst = -(1 << 16)                                            # scheme_type::worst_score (scoring.h:226-227)
    score_limit = worst
    mates = (sym1, sym2)
    packed = [pack_read_streams(m) for m in mates]                # (reversed reads for mapping, fw + rc extension words) per mate
    quals = torch.full((2 * n * L + 8,), qual_value, dtype=torch.uint8, device=dev)
    best = backend.init_best_mate(n, scheme, L, 0)
    best_o = backend.init_best_mate(n, scheme, L, 1)
    n_jobs = n_opp = 0
    for anchor in (0, 1):
        reads_rev, ext_words = packed[anchor]
        o_ext_words = packed[1 - anchor][1]
        hits, counts = backend.map_exact(reads_rev, hits_stride)
        k = torch.arange(hits.shape[1], device=dev).unsqueeze(0)
        valid = k < (counts.to(torch.int64) & 0xFFFFFFFF).unsqueeze(1)
        hits, _ = torch.sort(torch.where(valid, hits, torch.full_like(hits, (1 << 63) - 1)), dim=1)
        keep = valid.sum(1, keepdim=True) > k
        read_id = torch.arange(n, device=dev).unsqueeze(1).expand_as(hits)[keep]
        w = hits[keep]
        lo, hi = w & 0xFFFFFFFF, (w >> 32) & 0xFFFFFFFF
        delta, pir, rc = hi & 0xFFFFF, (hi >> 20) & 0x3FF, (hi >> 30) & 1
        take = torch.clamp(delta, max=rows_per_hit)
        rep = torch.repeat_interleave(torch.arange(w.numel(), device=dev), take)
        first = torch.cumsum(take, 0) - take
        row = lo[rep] + (torch.arange(rep.numel(), device=dev) - first[rep])
        gpos = backend.locate(row.to(torch.int32)).to(torch.int64) & 0xFFFFFFFF
        jr, jrc = read_id[rep], rc[rep]
        sel = _first_occurrences(jr, jrc, torch.clamp(gpos - pir[rep], min=0))
        jr, jrc, gpos, rep = jr[sel], jrc[sel], gpos[sel], rep[sel]
        loc = torch.clamp(gpos - pir[rep], min=0)                 # hit.loc: where the read starts (locate_inl.h:142)
        wbeg = torch.clamp(loc - band // 2, min=0)                # score_best_inl.h:112-116
        wend = torch.clamp(wbeg + L + band, max=genome_len)
        patterns = PackedStringSet(ext_words, 4, True, (jr * L + jrc * (n * L)).contiguous(), None, L)
        texts = PackedStringSet(genome_words, 2, True, wbeg.contiguous(), (wend - wbeg).to(torch.int32).contiguous(), 0)
        a_score, a_sink = backend.score_qual(band, banded_aligner, patterns, texts, quals)
        a_score = torch.clamp(a_score, min=worst)                 # hit.score = max(sink.score, worst_score) (:139)
        a_sinkx = (a_sink.view(-1, 2)[:, 0].to(torch.int64) & 0xFFFFFFFF)
        hit_sink = torch.where(a_sinkx == 0xFFFFFFFF, wbeg, wbeg + a_sinkx)
        i32 = lambda t: t.to(torch.int32).contiguous()
        ow = backend.opposite_windows(i32(jr), jrc.to(torch.uint8).contiguous(), i32(loc), i32(a_score), best, best_o, scheme, anchor, genome_len, L,
                                      min_frag_len=min_frag_len, max_frag_len=max_frag_len, score_limit=score_limit)
        ok = ow["valid"] != 0
        idx = torch.nonzero(ok).squeeze(1)
        ob = ow["genome_begin"].to(torch.int64)[idx] & 0xFFFFFFFF
        oe = ow["genome_end"].to(torch.int64)[idx] & 0xFFFFFFFF
        orc = ow["read_rc"].to(torch.int64)[idx]
        o_pat = PackedStringSet(o_ext_words, 4, True, (jr[idx] * L + orc * (n * L)).contiguous(), None, L)
        o_txt = PackedStringSet(genome_words, 2, True, ob.contiguous(), (oe - ob).to(torch.int32).contiguous(), 0)
        ms = ow["min_score"][idx].contiguous()
        s_o, k_o, _ = backend.full_score_qual(full_aligner, o_pat, o_txt, L, int(max_frag_len) + L, ms, quals)
        # hit.opposite_* (score_opposite_inl.h:224-235)
        o_score = torch.full((rep.numel(),), worst, dtype=torch.int32, device=dev)
        o_score[idx] = torch.where(s_o >= ms, s_o, torch.full_like(s_o, worst))
        o_loc = torch.zeros(rep.numel(), dtype=torch.int64, device=dev); o_loc[idx] = ob
        kx = k_o.view(-1, 2)[:, 0].to(torch.int64) & 0xFFFFFFFF
        o_sink = o_loc.clone(); o_sink[idx] = ob + torch.where(kx == 0xFFFFFFFF, torch.zeros_like(kx), kx)
        o_score2 = torch.full_like(o_score, worst)
        per_read = torch.bincount(jr, minlength=n)
        hit_begin = torch.zeros(n + 1, dtype=torch.int64, device=dev); hit_begin[1:] = torch.cumsum(per_read, 0)
        backend.reduce_paired(best, best_o, hit_begin, i32(loc), i32(hit_sink), i32(a_score), jrc.to(torch.uint8).contiguous(),
                              i32(o_loc), i32(o_sink), i32(o_loc), o_score, o_score2, anchor, L, score_limit=score_limit)
        n_jobs += int(rep.numel()); n_opp += int(idx.numel())
    mapq = backend.mapq_paired(best, best_o, scheme, L)
    return dict(best=best, best_o=best_o, mapq=mapq, n_jobs=n_jobs, n_opposite=n_opp)
