"""nvBowtie's single-end best-mapping driver over the C-ABI stages: Aligner::best_approx and
Aligner::best_approx_score (nvBowtie/bowtie2/cuda/aligner_best_approx.h:85-520, :522-840).

Per seeding pass: map the queued reads' seeds into their hit deques, then run extension rounds -- select the next SA
row(s) of every active read (deterministic or randomized, one or several per round), locate them, score the read
against the genome window around each, fold the scores into the best / second-best alignments while the
give-up counters decide which reads stay active -- and queue for re-seeding the reads that are still unaligned or
whose seeds were too repetitive.  Then MAPQ and the banded traceback of every best alignment.

Only queue bookkeeping lives here (compaction of the re-seed queue, the choice of hits-per-read per round): what
the reference's driver does on the host between kernel launches.  Every stage is a libnvbio_hip.so call; nothing
falls back to the CPU."""
import torch

from . import mapping, reduce, select as sel
from .alignment import (make_gotoh_aligner, make_edit_distance_aligner, SmithWatermanScoringScheme, SEMI_GLOBAL, LOCAL, PATTERN_BLOCKING, batch_banded_alignment_score,
                        batch_banded_alignment_traceback, batch_alignment_score, batch_alignment_traceback)
from .strings import PackedStringSet

WORST_SCORE = -(1 << 16)          # SmithWatermanScoringScheme::worst_score (scoring.h:226-227)


class _Stage:
    """Accumulates a stage's device time into stats["ms"][name] when stats["ms"] exists (as the reference's Stats does with
    its device timers); otherwise free."""

    def __init__(self, stats, name):
        self.ms, self.name = stats.get("ms"), name

    def __enter__(self):
        if self.ms is not None:
            self.e0, self.e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.e0.record()

    def __exit__(self, *exc):
        if self.ms is not None:
            self.e1.record(); self.e1.synchronize()
            self.ms[self.name] = self.ms.get(self.name, 0.0) + self.e0.elapsed_time(self.e1)


class Params:
    """The fields of nvBowtie's Params this driver reads, with its defaults (params.cpp:116-197; end-to-end)."""

    def __init__(self, **kw):
        self.max_hits, self.max_dist = 100, 15
        self.max_effort_init, self.max_effort, self.min_ext, self.max_ext = 15, 15, 30, 400
        self.max_reseed, self.rep_seeds, self.allow_sub, self.subseed_len = 2, 300, 0, 0
        self.randomized, self.top_seed, self.no_multi_hits = True, 0, False
        self.seed_len, self.seed_freq, self.min_read_len = 22, (mapping.SQRT_FUNC, 1.0, 1.15), 12
        self.local = False
        self.scoring_mode = "sw"                       # --scoring sw|ed (params.cpp:117): "ed" extends, reduces and traces hits with the edit-distance aligner
        self.fw, self.rc = True, True
        # paired-end (params.cpp:165-172; io::PE_POLICY_FR)
        self.pe_policy, self.pe_overlap, self.pe_unpaired, self.pe_discordant, self.min_frag_len, self.max_frag_len = 1, True, True, True, 0, 500
        self.batch_size = 1 << 20                      # Aligner::BATCH_SIZE
        self.hits_stride = None                        # arena slots per read (default: resolved_hits_stride)
        for k, v in kw.items():
            if not hasattr(self, k):
                raise TypeError("unknown parameter %s" % k)
            setattr(self, k, v)
        if self.local:                                 # --local moves the seeding defaults (params.cpp:156-160): 20-bp seeds every 1 + 0.75 sqrt(L)
            if "seed_len" not in kw:
                self.seed_len = 20
            if "seed_freq" not in kw:
                self.seed_freq = (mapping.SQRT_FUNC, 1.0, 0.75)
        self.max_effort_init = max(self.max_effort_init, self.max_effort)      # params.cpp:197-198
        self.max_ext = max(self.max_ext, self.max_effort)

    def resolved_hits_stride(self, max_read_len):
        """Slots of a read's hit deque when none are named (Params::resolved_hits_stride, include/nvbio_hip/aligner.h): the reference's capacity
        min(max_hits, 128), or -- exact seeding, where a read yields at most one range per seed and strand -- the 16 or 32 slots that hold them all"""
        if self.hits_stride:
            return self.hits_stride
        cap = min(self.max_hits, 128)
        if self.allow_sub:
            return cap
        most = 0
        for L in range(max(self.min_read_len, 1), max_read_len + 1):
            f = mapping.simple_func(*self.seed_freq, L)
            if f > 0:
                most = max(most, 2 * ((L - min(self.seed_len, L)) // f + 1))
        rows = 16 if most <= 16 else 32 if most <= 32 else most
        return min(cap, rows)

    def mapping_params(self):
        return mapping.MappingParams(self.seed_len, self.seed_freq, self.min_read_len, self.max_hits, self.max_reseed, self.rep_seeds)


def band_length(max_dist):
    """Aligner::band_length (aligner.h:165-174): the smallest 2^k - 1 >= 2 * max_dist + 1"""
    b = 4
    while b - 1 < max_dist * 2 + 1:
        b *= 2
    return b - 1


def hits_per_read(n_active, n_ext, params):
    """The choice at aligner_best_approx.h:627-650."""
    if n_active <= params.batch_size // 2 and not params.no_multi_hits:
        return min(params.batch_size // n_active, min(4096, params.max_ext - n_ext))
    return 1


TRACE = None        # {"read": id in the batch, "events": []} to record one read's picks


def best_approx_score(fmi, rfmi, state, seed_queue, best, batch, genome_words, genome_len, aligner, params, band_len, stats, best_sink=None):
    """Aligner::best_approx_score: the extension rounds of one seeding pass (`state` = select_init's output)."""
    active = seed_queue.to(torch.int32)                                   # pack_read(params.top_seed), defs.h:185-205
    if params.top_seed & 1:
        active = active | torch.tensor(-(1 << 31), dtype=torch.int32, device=active.device)
    n_ext = 0
    while active.numel() and n_ext < params.max_ext:
        n_multi = hits_per_read(active.numel(), n_ext, params)
        with _Stage(stats, "select"):
            active, hit_begin, rid, loc, seed = sel.select(state, active, n_multi)
        if active.numel() == 0:
            break
        if loc.numel() == 0:
            continue
        traced = None
        if TRACE is not None:                   # debugging aid: the SA rows picked for one read, round by round
            traced = (rid == TRACE["read"]).nonzero().flatten()
            rows = loc[traced].cpu().tolist()
        with _Stage(stats, "locate"):
            sel.locate_hits(fmi, rfmi, loc, seed)
        with _Stage(stats, "score"):
            # hits at a placement the read already recorded keep the recorded score; only the others become DP jobs (compacted)
            pb, pl, tb, tl, _, score, job_hit = sel.score_best_setup(rid, loc, seed, best, band_len, genome_len, WORST_SCORE,
                                                                     fixed_read_len=batch.fixed_len, read_begin=batch.read_begin,
                                                                     read_len=batch.read_len, rc_offset=batch.rc_offset, compact=True)
            hit_sink = torch.empty((loc.numel(), 2), dtype=torch.int32, device=loc.device) if best_sink is not None else None
            if job_hit.numel():
                patterns = PackedStringSet(batch.fw_rc_words, 4, True, pb, pl, batch.fixed_len)
                texts = PackedStringSet(genome_words, 2, True, tb, tl, 0)
                job_score, job_sink = batch_banded_alignment_score(band_len, aligner, patterns, texts, max_pattern_length=batch.max_len, quals=batch.quals)
                sel.scatter_scores(job_hit, job_score, score)
                if hit_sink is not None:
                    sel.scatter_sinks(job_hit, job_sink, hit_sink)
            stats["dp_jobs"] = stats.get("dp_jobs", 0) + int(job_hit.numel())
        with _Stage(stats, "reduce"):
            sel.score_reduce_best_approx(best, state, active, hit_begin, score, loc, seed, WORST_SCORE, n_ext, params.min_ext, params.max_ext,
                                         params.max_effort, fixed_read_len=batch.fixed_len, read_len=batch.read_len, hit_sink=hit_sink, best_sink=best_sink)
        if traced is not None and traced.numel():
            TRACE["events"].append(dict(n_ext=n_ext, sa_rows=[r & 0xFFFFFFFF for r in rows], seeds=[x & 0xFFFFFFFF for x in seed[traced].cpu().tolist()],
                                        positions=[x & 0xFFFFFFFF for x in loc[traced].cpu().tolist()], scores=score[traced].cpu().tolist()))
        stats["extensions"] += int(loc.numel()); stats["rounds"] += 1
        n_ext += n_multi


def _qual_stream(n, L, qual_value, quals, dev):
    """One quality byte per pattern symbol, laid out like the fw + rc pattern words (the rc copy's qualities reversed)."""
    if quals is None:
        return torch.full((2 * n * L + 8,), qual_value, dtype=torch.uint8, device=dev)
    q = quals.to(dev).to(torch.uint8).reshape(n, L)
    return torch.cat([q.reshape(-1), q.flip(1).reshape(-1), torch.zeros(8, dtype=torch.uint8, device=dev)])


class ReadBatch:
    """A batch of reads on the device in the layouts the stages read (io::SequenceDataDevice's role): stored reversed (io::REVERSE: what
    the mappers scan), forward copies followed rc_offset symbols later by the reverse complements (extension / traceback patterns), one
    quality byte per pattern symbol.  Equal-length batches keep fixed_len (no per-read arrays); ragged ones carry begin / length."""

    def __init__(self, n, max_len, fixed_len, read_begin, read_len, reversed_set, fw_rc_words, rc_offset, quals):
        self.n, self.max_len, self.fixed_len, self.read_begin, self.read_len = n, max_len, fixed_len, read_begin, read_len
        self.reversed, self.fw_rc_words, self.rc_offset, self.quals = reversed_set, fw_rc_words, rc_offset, quals

    @staticmethod
    def from_matrix(sym, qual_value=30, quals=None, packed=None):
        from .pipeline import pack_read_streams
        n, L = sym.shape
        reads_rev, fw_rc = packed if packed is not None else pack_read_streams(sym)
        return ReadBatch(n, L, L, None, None, reads_rev, fw_rc, n * L, _qual_stream(n, L, qual_value, quals, sym.device))

    @staticmethod
    def from_ragged(symbols, index, quals=None, qual_value=30):
        """symbols: uint8 [total] (0..4), index: int64 [n+1] offsets, quals: uint8 [total] or None"""
        from .workloads import _pack_chunked
        dev = symbols.device
        index = index.to(torch.int64)
        n, total = index.numel() - 1, int(index[-1])
        length = (index[1:] - index[:-1])
        pos = torch.arange(total, device=dev)
        r = torch.searchsorted(index, pos, right=True) - 1
        mirror = index[r + 1] - 1 - (pos - index[r])                                  # the same read's symbol counted from its end
        rev = symbols[mirror]
        rc = torch.where(rev > 3, rev, 3 - rev)                                       # complement_functor<4>: N stays N
        begin = index[:-1].contiguous()
        len32 = length.to(torch.int32).contiguous()
        reversed_set = PackedStringSet(_pack_chunked(rev, 4, True), 4, True, begin, len32, 0)
        fw_rc = _pack_chunked(torch.cat([symbols, rc]), 4, True)
        q = torch.full((total,), qual_value, dtype=torch.uint8, device=dev) if quals is None else quals.to(dev).to(torch.uint8)
        qs = torch.cat([q, q[mirror], torch.zeros(8, dtype=torch.uint8, device=dev)])
        return ReadBatch(n, int(length.max()) if n else 0, 0, begin, len32, reversed_set, fw_rc, total, qs)


def best_approx(fmi, rfmi, sym, genome_words, genome_len, params=None, scheme=None, names=None, qual_value=30, traceback=True,
                cigar_stride=None, stage_times=False, packed=None, quals=None, finish=False, mds_stride=256):
    """Aligner::best_approx for a batch of reads: `sym` is a uint8 [n, L] matrix of equal-length reads (symbols 0..4) or a ReadBatch
    (ReadBatch.from_ragged for reads of different lengths).  `names`: list of read names (they seed the randomized selection).
    Returns dict(best int64[2,n] io::Alignment words, mapq uint8[n], and with traceback: cigar int16[n,stride], cigar_len, source,
    sink (-1 for unaligned reads), stats)."""
    params = params or Params()
    batch = sym if isinstance(sym, ReadBatch) else ReadBatch.from_matrix(sym, qual_value, quals, packed)
    n, L = batch.n, batch.max_len                                                    # L: the longest read
    dev = batch.fw_rc_words.device
    scheme = scheme or (SmithWatermanScoringScheme.local() if params.local else SmithWatermanScoringScheme())
    # edit-distance mode (compute_thread.cu:296, scoring.h:133-200): hits are extended and reduced against the edit-distance costs and
    # score-min = -max_dist; MAPQ and finish_alignment's final scores still read the Smith-Waterman scheme (aligner_best_approx.h:294-296, :358)
    ed_mode = params.scoring_mode == "ed"
    final_scheme, scheme = scheme, (SmithWatermanScoringScheme.edit_distance(params.max_dist) if ed_mode else scheme)
    aligner = make_gotoh_aligner(LOCAL if params.local else SEMI_GLOBAL, scheme)
    band_len = band_length(params.max_dist)
    reads_rev, reads_fw_rc, quals = batch.reversed, batch.fw_rc_words, batch.quals
    if not params.randomized:
        name_arena = None
    elif isinstance(names, tuple):                       # already packed: (uint8 arena, int32 index[n+1]) on the device
        name_arena = names
    else:
        name_arena = sel.pack_names(names if names is not None else ["%d" % i for i in range(n)], dev)
    mp = params.mapping_params()
    best = reduce.BestAlignments(n, scheme, read_len=batch.read_len, fixed_read_len=batch.fixed_len, max_read_len=L, device=dev)     # init_alignments
    seed_queue = torch.arange(n, dtype=torch.int32, device=dev)
    hits_stride = params.resolved_hits_stride(L)
    stats = dict(extensions=0, rounds=0, seeding_passes=0, queue=[])
    if stage_times:
        stats["ms"] = {}
    # the DP sink of every read's best alignment, kept by the reduction: the traceback re-scores the very job the extension scored
    # (same window, pattern, scheme), so it can start from that score and sink instead
    best_sink = torch.full((n, 2), -1, dtype=torch.int32, device=dev) if traceback else None
    for seeding_pass in range(params.max_reseed + 1):
        if seed_queue.numel() == 0:
            break
        stats["queue"].append(int(seed_queue.numel())); stats["seeding_passes"] += 1
        with _Stage(stats, "map"):
            hits, counts, reseed = mapping.map_seeds(fmi, rfmi, reads_rev, mp, L, allow_sub=params.allow_sub, subseed_len=params.subseed_len,
                                                     retry=seeding_pass, fw=params.fw, rc=params.rc, in_queue=seed_queue, hits_stride=hits_stride)
        with _Stage(stats, "select_init"):
            state = sel.SelectState(hits, counts, name_arena, params.max_effort_init, params.randomized, params.top_seed)
        best_approx_score(fmi, rfmi, state, seed_queue, best, batch, genome_words, genome_len, aligner, params, band_len, stats, best_sink)
        sel.mark_unaligned(seed_queue, best, reseed)                               # aligner_init.cu:421-444
        seed_queue = sel.copy_flagged(seed_queue, reseed)                          # aligner_best_approx.h:273-283
    with _Stage(stats, "mapq"):
        mapq = reduce.mapq(best, final_scheme, read_len=batch.read_len, fixed_read_len=batch.fixed_len, max_read_len=L)
    out = dict(best=best.data, mapq=mapq, dp_jobs=stats.pop("dp_jobs", 0), stats=stats)    # (dp_jobs: the extensions that needed a DP)
    if traceback:
        # banded_traceback_best (traceback_inl.h:104-136) over every read; unaligned reads get an empty window, fail at once and
        # come back with no CIGAR and source = sink = (-1, -1)
        with _Stage(stats, "traceback"):
            if batch.read_len is None:
                valid, pb, tbeg, tlen = sel.traceback_best_setup(best.data, n, band_len, genome_len, L, batch.rc_offset)
                plen = None
            else:
                valid, pb, tbeg, tlen, plen = sel.traceback_best_setup(best.data, n, band_len, genome_len, 0, batch.rc_offset, read_begin=batch.read_begin,
                                                                      read_len=batch.read_len)
            pat, txt = PackedStringSet(reads_fw_rc, 4, True, pb, plen, batch.fixed_len), PackedStringSet(genome_words, 2, True, tbeg, tlen, 0)
            if ed_mode:       # the edit-distance aligner's own walk (sw_banded_inl.h:405-470): among equal moves it does not choose as the Gotoh walk does
                tb = batch_banded_alignment_traceback(band_len, make_edit_distance_aligner(LOCAL if params.local else SEMI_GLOBAL), pat, txt, max_pattern_length=L,
                                                      max_text_length=L + band_len, cigar_stride=cigar_stride)
            else:
                tb = batch_banded_alignment_traceback(band_len, aligner, pat, txt, max_pattern_length=L, quals=quals, cigar_stride=cigar_stride,
                                                      known=sel.traceback_best_known(best.data, best_sink, n))
        out.update(cigar=tb["cigar"], cigar_len=tb["cigar_len"], source=tb["source"], sink=tb["sink"], tb_score=tb["score"],
                   aligned_ids=torch.nonzero(best.is_aligned(0)).squeeze(1))
        if finish:
            # finish_alignment_best (traceback_inl.h:523-760): MD strings, edit distances, final scores; out["best"] becomes what the
            # reference hands to its output stage (m_align = window begin), the extension-stage words stay in out["best_scored"]
            out["best_scored"] = best.data.clone()
            with _Stage(stats, "finish"):
                out["mds"], out["mds_len"] = sel.finish_alignment(valid, pat, quals, txt, tb["cigar"], tb["cigar_len"], tb["source"], final_scheme, best.data,
                                                                  mds_stride=mds_stride)
    return out


# ------------------------------------------------------------------------------------------------------------------
# paired-end: Aligner::best_approx / best_approx_score of aligner_best_approx_paired.h (:95-453, :455-700)
# ------------------------------------------------------------------------------------------------------------------
def best_approx_score_paired(fmi, rfmi, state, seed_queue, anchor, best, best_o, a_words, o_words, n_reads, read_len, genome_words, genome_len,
                             scheme, banded_aligner, full_aligner, quals, table, params, band_len, stats, memo, o_quals=None, a_batch=None, o_batch=None):
    """The extension rounds of one seeding pass of one anchor mate (`quals`: the anchor mate's quality stream, `o_quals`: the opposite mate's): select, locate, anchor_score_best, opposite_score_best over the
    hits whose anchor scored, score_reduce_paired with the give-up counters."""
    L = read_len
    # mates of their own lengths (a_batch / o_batch: the anchor and opposite mates' ReadBatch): per-read begins and lengths; else every read is L long
    ragged = a_batch is not None and (a_batch.read_len is not None or o_batch.read_len is not None)
    a_fix, o_fix = (a_batch.fixed_len, o_batch.fixed_len) if ragged else (L, L)
    a_rc, o_rc = (a_batch.rc_offset, o_batch.rc_offset) if ragged else (n_reads * L, n_reads * L)
    a_begin, a_len, o_begin, o_len = (a_batch.read_begin, a_batch.read_len, o_batch.read_begin, o_batch.read_len) if ragged else (None, None, None, None)
    active = seed_queue.to(torch.int32)
    if params.top_seed & 1:
        active = active | torch.tensor(-(1 << 31), dtype=torch.int32, device=active.device)
    n_ext = 0
    n_valid_dev = torch.zeros((), dtype=torch.int64, device=active.device)
    while active.numel() and n_ext < params.max_ext:
        n_multi = hits_per_read(active.numel(), n_ext, params)
        with _Stage(stats, "select"):
            active, hit_begin, rid, loc, seed = sel.select(state, active, n_multi)
        if active.numel() == 0:
            break
        if loc.numel() == 0:
            continue
        traced = None
        if TRACE is not None:                   # debugging aid: the SA rows picked for one read, round by round
            traced = (rid == TRACE["read"]).nonzero().flatten()
            rows = loc[traced].cpu().tolist()
        with _Stage(stats, "locate"):
            sel.locate_hits(fmi, rfmi, loc, seed)
        with _Stage(stats, "anchor_score"):
            r = sel.anchor_score_setup(rid, loc, seed, best, best_o, scheme, anchor, band_len, genome_len, WORST_SCORE, a_fix, o_fix, a_rc, table,
                                       a_read_begin=a_begin, a_read_len=a_len, o_read_len=o_len)
            pb, tb, tl, ms = r[:4]
            pl = r[4] if len(r) > 4 else None
            raw, raw_sink = batch_banded_alignment_score(band_len, banded_aligner, PackedStringSet(a_words, 4, True, pb, pl, 0 if pl is not None else L),
                                                         PackedStringSet(genome_words, 2, True, tb, tl, 0), max_pattern_length=L, quals=quals)
            hit_score, hit_sink = sel.anchor_score_finish(raw, raw_sink, tb, ms, WORST_SCORE)
        with _Stage(stats, "opposite_score"):
            ow = sel.opposite_score_setup(rid, seed, loc, hit_score, WORST_SCORE, best, best_o, scheme, anchor, genome_len, a_fix, o_fix, params.pe_policy,
                                          params.min_frag_len, params.max_frag_len, params.pe_overlap, WORST_SCORE, table, a_read_len=a_len, o_read_len=o_len)
            # jobs whose (window, threshold, strand) equal the pair's last scored job are answered from the memo: the reference
            # re-runs them and absorbs the identical result
            o_out = sel.opposite_outputs(int(loc.numel()), WORST_SCORE, loc.device)
            sel.opposite_memo_lookup(rid, ow, anchor, memo, WORST_SCORE, o_out)
            idx = torch.nonzero(ow["valid"] == 1).squeeze(1)               # the jobs that are actually scored
            n_valid_dev = n_valid_dev + torch.count_nonzero(ow["valid"])   # what the reference scores (stats; read once, after the rounds)
            stats["opposite_dp_jobs"] = stats.get("opposite_dp_jobs", 0) + int(idx.numel())
            if idx.numel():
                ob = ow["genome_begin"].to(torch.int64)[idx] & 0xFFFFFFFF
                oe = ow["genome_end"].to(torch.int64)[idx] & 0xFFFFFFFF
                r_o = rid.to(torch.int64)[idx] & 0xFFFFFFFF
                o_first = o_begin[r_o] if o_begin is not None else r_o * o_fix
                o_plen = o_len[r_o].contiguous() if o_len is not None else None
                o_pat = PackedStringSet(o_words, 4, True, (o_first + ow["read_rc"].to(torch.int64)[idx] * o_rc).contiguous(), o_plen, 0 if o_plen is not None else L)
                o_txt = PackedStringSet(genome_words, 2, True, ob.contiguous(), (oe - ob).to(torch.int32).contiguous(), 0)
                o_ms = ow["min_score"][idx].contiguous()
                max_n = int(params.max_frag_len) + L
                s_o, k_o, _ = batch_alignment_score(full_aligner, o_pat, o_txt, L, max_n, o_ms, quals=quals if o_quals is None else o_quals)
            else:
                s_o = torch.empty(0, dtype=torch.int32, device=loc.device); k_o = torch.empty((0, 2), dtype=torch.int32, device=loc.device)
            o_score, o_score2, o_loc, o_sink, o_sink2 = sel.opposite_score_finish(idx.to(torch.int32), s_o, k_o, ow["min_score"], ow["genome_begin"],
                                                                                   WORST_SCORE, int(loc.numel()), out=o_out)
            sel.opposite_memo_update(active, hit_begin, ow, o_out, anchor, memo)
        with _Stage(stats, "reduce"):
            sel.score_reduce_paired_best_approx(best, best_o, state, active, hit_begin, loc, hit_sink, hit_score, seed, o_loc, o_sink, o_sink2, o_score, o_score2,
                                                anchor, params.pe_policy, params.pe_unpaired, WORST_SCORE, n_ext, params.min_ext, params.max_ext,
                                                params.max_effort, a_fix, read_len=a_len)
        if traced is not None and traced.numel():
            TRACE["events"].append(dict(n_ext=n_ext, sa_rows=[r & 0xFFFFFFFF for r in rows], seeds=[x & 0xFFFFFFFF for x in seed[traced].cpu().tolist()],
                                        positions=[x & 0xFFFFFFFF for x in loc[traced].cpu().tolist()], scores=score[traced].cpu().tolist()))
        stats["extensions"] += int(loc.numel()); stats["rounds"] += 1
        n_ext += n_multi
    stats["opposite_extensions"] += int(n_valid_dev)


def best_approx_paired(fmi, rfmi, sym1, sym2, genome_words, genome_len, params=None, scheme=None, names=None, qual_value=30, traceback=True,
                       cigar_stride=64, stage_times=False, finish=False, mds_stride=256, quals1=None, quals2=None):
    """Aligner::best_approx for read pairs (equal-length mates sym1 / sym2, uint8 [n, L]; quals1 / quals2: their phred qualities, uint8
    [n, L], or None for `qual_value` throughout).  Returns dict(best, best_o int64[2,n]
    io::Alignment words of the anchor / opposite slots, mapq1, mapq2 uint8[n], and with traceback per slot set ("1" = best_data,
    "2" = best_data_o): cigar, cigar_len, source, sink; stats)."""
    from .pipeline import pack_read_streams
    params = params or Params()
    # mates of their own lengths: two ReadBatch objects (ReadBatch.from_ragged) instead of two [n, L] matrices -- the reference's paired driver
    # takes the mates as they come (aligner_best_approx_paired.h); L is then the longest read of either mate
    mates = [sym1, sym2] if isinstance(sym1, ReadBatch) else None
    if mates is not None:
        assert isinstance(sym2, ReadBatch) and sym2.n == sym1.n
        n, L = sym1.n, max(sym1.max_len, sym2.max_len)
        dev = sym1.fw_rc_words.device
    else:
        n, L = sym1.shape
        assert sym2.shape == sym1.shape
        dev = sym1.device
    scheme = scheme or (SmithWatermanScoringScheme.local() if params.local else SmithWatermanScoringScheme())
    aln_type = LOCAL if params.local else SEMI_GLOBAL
    # edit-distance mode: hits are extended against the edit-distance costs and score-min = -max_dist and traced with the edit-distance
    # aligner's own walks; MAPQ and finish_alignment read the Smith-Waterman scheme (see best_approx)
    ed_mode = params.scoring_mode == "ed"
    final_scheme, scheme = scheme, (SmithWatermanScoringScheme.edit_distance(params.max_dist) if ed_mode else scheme)
    banded_aligner = make_gotoh_aligner(aln_type, scheme)
    full_aligner = make_gotoh_aligner(aln_type, scheme, PATTERN_BLOCKING)            # nvBowtie's aligners carry the default tag
    tb_banded = make_edit_distance_aligner(aln_type) if ed_mode else banded_aligner
    tb_full = make_edit_distance_aligner(aln_type, PATTERN_BLOCKING) if ed_mode else full_aligner
    band_len = band_length(params.max_dist)
    if mates is not None:
        packed = [(m.reversed, m.fw_rc_words) for m in mates]
        mate_quals = [m.quals for m in mates]
    else:
        packed = [pack_read_streams(sym1), pack_read_streams(sym2)]                       # per mate: (reversed reads, fw + rc words)
        mate_quals = [_qual_stream(n, L, qual_value, quals1, dev), _qual_stream(n, L, qual_value, quals2, dev)]   # laid out like each mate's fw + rc words
    mlen = (lambda m: dict(read_len=mates[m].read_len, fixed_read_len=mates[m].fixed_len)) if mates is not None else (lambda m: dict(fixed_read_len=L))
    if not params.randomized:
        name_arena = None
    elif isinstance(names, tuple):
        name_arena = names
    else:
        name_arena = sel.pack_names(names if names is not None else ["%d" % i for i in range(n)], dev)
    mp = params.mapping_params()
    table = sel._min_score_table(scheme, L, dev)
    best = reduce.BestAlignments(n, scheme, max_read_len=L, device=dev, mate=0, **mlen(0))
    best_o = reduce.BestAlignments(n, scheme, max_read_len=L, device=dev, mate=1, **mlen(1))
    hits_stride = params.resolved_hits_stride(L)
    stats = dict(extensions=0, opposite_extensions=0, rounds=0, seeding_passes=0, queue=[])
    if stage_times:
        stats["ms"] = {}
    memo = sel.opposite_memo(n, dev)
    for anchor in (0, 1):
        seed_queue = torch.arange(n, dtype=torch.int32, device=dev)
        fw_strand = params.pe_policy in (0, 1) if anchor == 0 else params.pe_policy in (0, 2)      # :168-176
        fw, rc = (params.fw, params.rc) if fw_strand else (params.rc, params.fw)
        reads_rev, a_words = packed[anchor]
        o_words = packed[1 - anchor][1]
        for seeding_pass in range(params.max_reseed + 1):
            if seed_queue.numel() == 0:
                break
            stats["queue"].append(int(seed_queue.numel())); stats["seeding_passes"] += 1
            with _Stage(stats, "map"):
                hits, counts, reseed = mapping.map_seeds(fmi, rfmi, reads_rev, mp, L, allow_sub=params.allow_sub, subseed_len=params.subseed_len,
                                                         retry=seeding_pass, fw=fw, rc=rc, in_queue=seed_queue, hits_stride=hits_stride)
            with _Stage(stats, "select_init"):
                state = sel.SelectState(hits, counts, name_arena, params.max_effort_init, params.randomized, params.top_seed)
            best_approx_score_paired(fmi, rfmi, state, seed_queue, anchor, best, best_o, a_words, o_words, n, L, genome_words, genome_len, scheme,
                                     banded_aligner, full_aligner, mate_quals[anchor], table, params, band_len, stats, memo, o_quals=mate_quals[1 - anchor],
                                     a_batch=mates[anchor] if mates is not None else None, o_batch=mates[1 - anchor] if mates is not None else None)
            seed_queue = seed_queue[reseed != 0]                                      # copy_flagged (no mark_unaligned in the paired driver)
    if params.pe_discordant:
        sel.mark_discordant(best, best_o)
    with _Stage(stats, "mapq"):
        ml = lambda a, o: (dict(read_len=mates[a].read_len, o_read_len=mates[o].read_len, fixed_read_len=mates[a].fixed_len, o_fixed_read_len=mates[o].fixed_len, max_read_len=L)
                           if mates is not None else dict(fixed_read_len=L, o_fixed_read_len=L))
        mapq1 = reduce.mapq_paired(best, best_o, final_scheme, **ml(0, 1))      # MapqFunctorPE(mate 0)
        mapq2 = reduce.mapq_paired(best_o, best, final_scheme, **ml(1, 0))      # MapqFunctorPE(mate 1)
    out = dict(best=best.data, best_o=best_o.data, mapq1=mapq1, mapq2=mapq2, opposite_dp_jobs=stats.pop("opposite_dp_jobs", 0), stats=stats)
    if traceback:
        # both mates' fw + rc patterns in one stream: a traceback picks its read by the alignment's mate bit (traceback_inl.h:117-120)
        mate_words = torch.cat([packed[0][1], packed[1][1]])
        mate_offset = int(packed[0][1].numel()) * 8
        q_syms = [2 * int(m.rc_offset) for m in mates] if mates is not None else [2 * n * L, 2 * n * L]          # fw + rc quality bytes per mate
        tq = torch.zeros(mate_offset + q_syms[1] + 8, dtype=torch.uint8, device=dev)
        tq[: q_syms[0]] = mate_quals[0][: q_syms[0]]; tq[mate_offset: mate_offset + q_syms[1]] = mate_quals[1][: q_syms[1]]
        sets = lambda pb, tbeg, tlen, plen=None: (PackedStringSet(mate_words, 4, True, pb, plen, 0 if plen is not None else L), PackedStringSet(genome_words, 2, True, tbeg, tlen, 0))

        def tb_setup(slots, want, idx=None):
            """-> (valid, pattern set, text set)"""
            if mates is not None:
                v, pb, tbeg, tlen, plen = sel.traceback_best_setup_mates(slots, n, band_len, genome_len, mates, mate_offset, want=want, idx=idx)
                return (v,) + sets(pb, tbeg, tlen, plen)
            v, pb, tbeg, tlen = sel.traceback_best_setup(slots, n, band_len, genome_len, L, n * L, mate_offset, want=want, idx=idx)
            return (v,) + sets(pb, tbeg, tlen)
        with _Stage(stats, "traceback"):
            # banded_traceback_best over the anchor slots (every aligned entry)
            with _Stage(stats, "traceback.anchor"):
                v1, pat1, txt1 = tb_setup(best.data, 0)
                tb1 = batch_banded_alignment_traceback(band_len, tb_banded, pat1, txt1, max_pattern_length=L, quals=tq, cigar_stride=cigar_stride)
            # the opposite slots: opposite_traceback_best (full matrix over [alignment, alignment + sink)) for the concordant ones,
            # banded_traceback_best for the other aligned ones
            w_o = best_o.data[0]
            concordant = (((w_o >> 30) & 1) != 0) & (((w_o >> 31) & 1) == 0) & best_o.is_aligned(0)
            ids_c = torch.nonzero(concordant).squeeze(1).to(torch.int32)
            with _Stage(stats, "traceback.opposite_banded"):
                vu, pat_u, txt_u = tb_setup(best_o.data, 2)
                tb_u = batch_banded_alignment_traceback(band_len, tb_banded, pat_u, txt_u, max_pattern_length=L, quals=tq, cigar_stride=cigar_stride)
            if ids_c.numel():
                with _Stage(stats, "traceback.opposite_full"):
                    vc, pat_c, txt_c = tb_setup(best_o.data, 1, ids_c)
                    # these windows end at the sink of the opposite-mate scoring pass, whose score the slot holds: the traceback drops
                    # the rows of the window no alignment with that score can reach
                    known = sel.traceback_best_known(best_o.data, None, n, idx=ids_c)[0]
                    tb_c = batch_alignment_traceback(tb_full, pat_c, txt_c, L, 1024, cigar_stride=cigar_stride, quals=tq, known_score=None if ed_mode else known)
        if finish:
            out["best_scored"], out["best_o_scored"] = best.data.clone(), best_o.data.clone()
            with _Stage(stats, "finish"):
                out["mds1"], out["mds1_len"] = sel.finish_alignment(v1, pat1, tq, txt1, tb1["cigar"], tb1["cigar_len"], tb1["source"], final_scheme, best.data, mds_stride=mds_stride)
                # the reference evaluates mate 2's MAPQ functor here, after the anchor slots were finished and before the opposite ones
                # are (aligner_best_approx_paired.h:308-323)
                out["mapq2"] = reduce.mapq_paired(best_o, best, final_scheme, **ml(1, 0))
                mds2, mds2_len = sel.finish_alignment(vu, pat_u, tq, txt_u, tb_u["cigar"], tb_u["cigar_len"], tb_u["source"], final_scheme, best_o.data, mds_stride=mds_stride)
                if ids_c.numel():
                    mc, mc_len = sel.finish_alignment(vc, pat_c, tq, txt_c, tb_c["cigar"], tb_c["cigar_len"], tb_c["source"], final_scheme, best_o.data, idx=ids_c,
                                                      mds_stride=mds_stride)
                    k = ids_c.to(torch.int64)
                    mds2[k] = mc[: k.numel()]; mds2_len[k] = mc_len
                out["mds2"], out["mds2_len"] = mds2, mds2_len
        if ids_c.numel():                                   # merge the two kinds of opposite-slot tracebacks
            k = ids_c.to(torch.int64)
            for key in ("cigar", "cigar_len", "source", "sink", "score"):
                tb_u[key][k] = tb_c[key][: k.numel()]
        out["tb1"], out["tb2"] = tb1, tb_u
    return out


# ------------------------------------------------------------------------------------------------------------------
# all-mapping: Aligner::all / score_all of aligner_all.h (:47-227, :264-694)
# ------------------------------------------------------------------------------------------------------------------
def all_mapping(fmi, rfmi, sym, genome_words, genome_len, params=None, scheme=None, qual_value=30, quals=None, packed=None, sequence_index=None,
                cigar_stride=None, mds_stride=256, traceback=True, stage_times=False):
    """nvBowtie's all-mapping mode: one mapping pass with every seed of every read, then every row of every SA range is located,
    de-duplicated (within a batch of params.batch_size hits, as in the reference), extended, and reported when its score reaches
    scheme.min_score(read_len).  The reference streams the accepted alignments through a ring buffer in atomic order and sorts each
    traceback batch by read; here they come back in batch order, sorted by (read, strand, position) inside a batch.
    Returns dict(read_id int32[m], alignments int64[m] (finished io::Alignment words: window begin, edit distance, final score),
    alignments_scored (as accepted: read start, extension score), cigar, cigar_len, source, sink, mds, mds_len, stats)."""
    params = params or Params()
    batch = sym if isinstance(sym, ReadBatch) else ReadBatch.from_matrix(sym, qual_value, quals, packed)
    n, L = batch.n, batch.max_len
    dev = batch.fw_rc_words.device
    scheme = scheme or (SmithWatermanScoringScheme.local() if params.local else SmithWatermanScoringScheme())
    ed_mode = params.scoring_mode == "ed"                  # all_ed (compute_thread.cu:265-278): see best_approx
    final_scheme, scheme = scheme, (SmithWatermanScoringScheme.edit_distance(params.max_dist) if ed_mode else scheme)
    aligner = make_gotoh_aligner(LOCAL if params.local else SEMI_GLOBAL, scheme)
    band_len = band_length(params.max_dist)
    mp = params.mapping_params()
    hits_stride = params.resolved_hits_stride(L)
    stats = dict(hits=0, ranges=0, unique=0)
    if stage_times:
        stats["ms"] = {}
    # map_kernel (mapping_inl.h:598-687) searches seeds 0 .. max_seeds-1 of the first seeding pattern, max_seeds = max over the read
    # lengths of uint32(len / seed_freq(len)) (aligner_all.h:93-95); map_seeds(retry 0) searches every seed that fits, which is the
    # same set whenever the cap does not bind -- checked here, loudly, instead of silently mapping more seeds than the reference
    lens = [L] if batch.read_len is None else sorted(set(int(x) for x in torch.unique(batch.read_len).tolist()))
    freq = lambda l: max(mapping.simple_func(*mp.seed_freq, l), 0)
    max_seeds = max([l // freq(l) for l in range(max(lens[0], 1), lens[-1] + 1) if freq(l) > 0] or [0])
    for l in lens:
        if l >= mp.min_read_len and freq(l) > 0 and (l - min(mp.seed_len, l)) // freq(l) + 1 > max_seeds:
            raise ValueError("all_mapping: the reference's seed cap (%d) would drop seeds of %d-bp reads; this seeding interval is not supported" % (max_seeds, l))
    with _Stage(stats, "map"):
        hits, counts, _ = mapping.map_seeds(fmi, rfmi, batch.reversed, mp, L, allow_sub=params.allow_sub, subseed_len=params.subseed_len, retry=0,
                                            fw=params.fw, rc=params.rc, in_queue=None, hits_stride=hits_stride,
                                            algorithm=mapping.APPROX_MAPPING if params.allow_sub else mapping.EXACT_MAPPING)
    empty = dict(read_id=torch.zeros(0, dtype=torch.int32, device=dev), alignments=torch.zeros(0, dtype=torch.int64, device=dev),
                 alignments_scored=torch.zeros(0, dtype=torch.int64, device=dev), stats=stats)
    # scans (thrust::inclusive_scan in the reference) and the range sizes
    count_scan = sel.inclusive_scan(counts)
    n_ranges = int(count_scan[-1]) if n else 0
    if n_ranges == 0:
        return empty
    range_scan = sel.inclusive_scan(sel.gather_ranges(hits, counts, count_scan, n_ranges))
    n_hits = int(range_scan[-1])
    stats["hits"], stats["ranges"] = n_hits, n_ranges
    seq_index = torch.tensor(sequence_index if sequence_index is not None else [0, genome_len], dtype=torch.int64, device=dev).to(torch.int32)
    table = sel._min_score_table(scheme, L, dev)
    out_aln, out_read, out_known = [], [], []
    B = params.batch_size
    for off in range(0, n_hits, B):
        cnt = min(n_hits - off, B)
        with _Stage(stats, "select"):
            loc, seed, rid = sel.select_all(off, cnt, hits, count_scan, range_scan)
        with _Stage(stats, "locate"):
            sel.locate_hits(fmi, rfmi, loc, seed)
        with _Stage(stats, "sort"):
            # SortingKeys order + first-of-run flags; mark_straddling reads the reference's stale `pipeline.idx_queue` (aligner_all.h:520): the half of
            # the ping-pong index buffer sort_hi_bits ended in, as sort_64_bits left it -- replayed by sort_hits_pingpong
            sidx, flags, stale = sel.sort_hits_pingpong(rid, loc, seed)
            sel.mark_straddling(stale, seq_index, loc, params.seed_len, flags)
            q = sel.copy_flagged(sidx, flags)
        stats["unique"] += int(q.numel())
        if q.numel() == 0:
            continue
        with _Stage(stats, "score"):
            pb, pl, tb, tl = sel.score_all_setup(q, rid, loc, seed, band_len, genome_len, fixed_read_len=batch.fixed_len, read_begin=batch.read_begin,
                                                 read_len=batch.read_len, rc_offset=batch.rc_offset)
            score, sink = batch_banded_alignment_score(band_len, aligner, PackedStringSet(batch.fw_rc_words, 4, True, pb, pl, batch.fixed_len),
                                                       PackedStringSet(genome_words, 2, True, tb, tl, 0), max_pattern_length=L, quals=batch.quals)
            fl, aln, arid = sel.score_all_output(q, rid, loc, seed, score, table, fixed_read_len=batch.fixed_len, read_len=batch.read_len)
            keep = fl.bool()
            out_aln.append(aln[keep]); out_read.append(arid[keep])
            out_known.append((score[keep], sink[keep]))              # the traceback re-scores these very jobs: it starts from their sinks instead
    if not out_aln:
        return empty
    aln, arid = torch.cat(out_aln), torch.cat(out_read)
    k_score, k_sink = torch.cat([k[0] for k in out_known]), torch.cat([k[1] for k in out_known])
    out = dict(read_id=arid, alignments_scored=aln.clone(), alignments=aln, stats=stats)
    m = aln.numel()
    if traceback and m:
        cig, cl, src, snk, mds, ml = [], [], [], [], [], []
        for off in range(0, m, B):                     # banded_traceback_all + finish_alignment_all, a batch at a time
            a, r = aln[off:off + B], arid[off:off + B].contiguous()
            with _Stage(stats, "traceback"):
                pb, pl, tb, tl = sel.traceback_all_setup(a, r, band_len, genome_len, fixed_read_len=batch.fixed_len, read_begin=batch.read_begin,
                                                         read_len=batch.read_len, rc_offset=batch.rc_offset)
                pat, txt = PackedStringSet(batch.fw_rc_words, 4, True, pb, pl, batch.fixed_len), PackedStringSet(genome_words, 2, True, tb, tl, 0)
                if ed_mode:
                    t = batch_banded_alignment_traceback(band_len, make_edit_distance_aligner(LOCAL if params.local else SEMI_GLOBAL), pat, txt, max_pattern_length=L,
                                                         max_text_length=L + band_len, cigar_stride=cigar_stride)
                else:
                    t = batch_banded_alignment_traceback(band_len, aligner, pat, txt, max_pattern_length=L, quals=batch.quals, cigar_stride=cigar_stride,
                                                         known=(k_score[off:off + B].contiguous(), k_sink[off:off + B].contiguous()))
            with _Stage(stats, "finish"):
                valid = torch.ones(a.numel(), dtype=torch.uint8, device=dev)
                md, mdl = sel.finish_alignment(valid, pat, batch.quals, txt, t["cigar"], t["cigar_len"], t["source"], final_scheme, a, mds_stride=mds_stride)
            cig.append(t["cigar"][: a.numel()]); cl.append(t["cigar_len"][: a.numel()]); src.append(t["source"][: a.numel()]); snk.append(t["sink"][: a.numel()])
            mds.append(md[: a.numel()]); ml.append(mdl)
        out.update(cigar=torch.cat(cig), cigar_len=torch.cat(cl), source=torch.cat(src), sink=torch.cat(snk), mds=torch.cat(mds), mds_len=torch.cat(ml))
    return out
