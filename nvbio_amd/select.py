"""nvBowtie's hit-selection stage and the per-round stages of its best-approx extension loop, over the C-ABI
(nvBowtie/bowtie2/cuda/select.h, select_inl.h, locate_inl.h, score_best_inl.h, reduce.h).

The hit arena returned by `mapping.map_exact` / `map_seeds` is the reference's SeedHitDequeArray content (a read's
hits in the array order of its priority deque); `SelectState` adds what `select_init` sets up next to it."""
import ctypes as C

import torch

from ._lib import lib, check, current_stream_ptr


def _vp(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def sum_tree_node_count(size):
    return int(lib().nvbio_hip_sum_tree_node_count(int(size)))


def pack_names(names, device):
    """NUL-terminated names -> (uint8 arena, int32 index[n+1]) as SequenceData keeps them (name_stream / name_index)."""
    import numpy as np
    blob = ("\0".join(names) + "\0").encode() if len(names) else b""
    idx = np.zeros(len(names) + 1, dtype=np.int64)
    if len(names):
        idx[1:] = np.cumsum(np.fromiter((len(nm.encode()) if not nm.isascii() else len(nm) for nm in names), dtype=np.int64, count=len(names)) + 1)
    return (torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(device) if blob else torch.zeros(0, dtype=torch.uint8, device=device),
            torch.from_numpy(idx).to(torch.int32).to(device))


class SelectState:
    """trys / rseeds / probability trees of every read of a batch (BestApproxScoringPipelineState's trys, rseeds
    and the SeedHitDequeArray's m_probs)."""

    def __init__(self, hits, counts, names=None, max_effort_init=15, randomized=True, top_seed=0, rseeds=None):
        n, stride = hits.shape
        dev = hits.device
        self.hits, self.counts = hits, counts
        self.randomized = bool(randomized)
        # a read's row holds the LEAVES of its probability tree (the sums are rebuilt on chip); rows wider than 32 hit slots also hold the sums
        self.probs_stride = ((stride + 3) & ~3) if stride <= 32 else sum_tree_node_count(stride)
        self.probs = torch.zeros((n, self.probs_stride), dtype=torch.float32, device=dev) if randomized else None
        self.trys = torch.zeros(n, dtype=torch.int32, device=dev)
        self.rseeds = (rseeds.clone() if rseeds is not None else torch.zeros(n, dtype=torch.int32, device=dev))
        arena, idx = names if names is not None else (None, None)
        check(lib().nvbio_hip_select_init(n, _vp(arena), _vp(idx), _vp(hits), stride, _vp(counts), _vp(self.probs), self.probs_stride,
                                          _vp(self.trys), _vp(self.rseeds), int(max_effort_init), int(self.randomized), int(top_seed),
                                          current_stream_ptr()), "nvbio_hip_select_init")


def select(state, active_in, n_multi=1):
    """One selection round.  active_in: int32 packed_read words.  Returns (active_out, hit_begin int64[n_out+1],
    hit_read_id, hit_loc (SA rows), hit_seed (packed_seed words)), trimmed to their sizes (one host sync, as in the
    reference, which reads the queue sizes back at this point: aligner_best_approx.h:678-692)."""
    n = active_in.numel()
    dev = active_in.device
    active_out = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    hit_begin = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    cap = max(n * n_multi, 1)
    rid = torch.empty(cap, dtype=torch.int32, device=dev)
    loc = torch.empty(cap, dtype=torch.int32, device=dev)
    seed = torch.empty(cap, dtype=torch.int32, device=dev)
    sizes = torch.zeros(2, dtype=torch.int32, device=dev)
    tb = int(lib().nvbio_hip_select_temp_bytes(n, n_multi))
    temp = torch.empty(tb, dtype=torch.uint8, device=dev)
    check(lib().nvbio_hip_select(int(state.randomized), int(n_multi), _vp(active_in), n, _vp(state.hits), state.hits.shape[1], _vp(state.counts),
                                 _vp(state.probs), state.probs_stride, _vp(state.rseeds), _vp(state.trys),
                                 _vp(active_out), _vp(hit_begin), _vp(rid), _vp(loc), _vp(seed), _vp(sizes), _vp(temp), tb,
                                 current_stream_ptr()), "nvbio_hip_select")
    n_out, n_hits = (int(v) for v in sizes.tolist())
    return active_out[:n_out], hit_begin[:n_out + 1], rid[:n_hits], loc[:n_hits], seed[:n_hits]


def locate_hits(fmi, rfmi, hit_loc, hit_seed):
    """In place: SA rows -> read-start genome coordinates (locate_kernel)."""
    s = fmi.struct()
    r = rfmi.struct() if rfmi is not None else None
    check(lib().nvbio_hip_locate_hits(C.byref(s), C.byref(r) if r is not None else None, hit_loc.numel(), _vp(hit_loc), _vp(hit_seed),
                                      current_stream_ptr()), "nvbio_hip_locate_hits")
    return hit_loc


def score_best_setup(hit_read_id, hit_loc, hit_seed, best, band_len, genome_len, score_limit, fixed_read_len=0, read_begin=None,
                     read_len=None, rc_offset=0, known=False, compact=False):
    """Per hit: (pattern_begin int64, pattern_len int32 | None, text_begin int64, text_len int32, min_score int32[, known_score int32 when
    `known`: the recorded score of hits at an already recorded placement, whose windows come back empty]).
    compact: only the hits that still need a DP become jobs; returns (pb, pl, tb, tl, ms, known_score, job_hit), the job arrays cut to
    the job count (one device->host read)."""
    known = known or compact
    n = hit_read_id.numel()
    dev = hit_read_id.device
    pb = torch.empty(n, dtype=torch.int64, device=dev)
    pl = torch.empty(n, dtype=torch.int32, device=dev) if read_len is not None else None
    tb = torch.empty(n, dtype=torch.int64, device=dev)
    tl = torch.empty(n, dtype=torch.int32, device=dev)
    ms = torch.empty(n, dtype=torch.int32, device=dev)
    ks = torch.empty(n, dtype=torch.int32, device=dev) if known else None
    jc = torch.empty(1, dtype=torch.int32, device=dev) if compact else None
    jh = torch.empty(n, dtype=torch.int32, device=dev) if compact else None
    data = best.data if hasattr(best, "data") else best
    check(lib().nvbio_hip_score_best_setup(n, _vp(hit_read_id), _vp(hit_loc), _vp(hit_seed), _vp(read_begin), _vp(read_len), int(fixed_read_len),
                                           int(rc_offset), int(band_len), int(genome_len), _vp(data), data.shape[1], int(score_limit),
                                           _vp(pb), _vp(pl), _vp(tb), _vp(tl), _vp(ms), _vp(ks), _vp(jc), _vp(jh), current_stream_ptr()),
          "nvbio_hip_score_best_setup")
    if compact:
        nj = int(jc.item())
        return pb[:nj], (pl[:nj] if pl is not None else None), tb[:nj], tl[:nj], ms[:nj], ks, jh[:nj]
    return (pb, pl, tb, tl, ms, ks) if known else (pb, pl, tb, tl, ms)


def scatter_sinks(job_hit, sinks, hit_sink):
    """hit_sink[job_hit[j]] = sinks[j] (uint2 rows)."""
    check(lib().nvbio_hip_scatter_rows(job_hit.numel(), _vp(job_hit), _vp(sinks), _vp(hit_sink), 8, current_stream_ptr()), "nvbio_hip_scatter_rows")
    return hit_sink


def scatter_scores(job_hit, scores, known_score):
    """known_score[job_hit[j]] = scores[j]: the DP scores of compacted jobs back at their hits."""
    check(lib().nvbio_hip_scatter_rows(job_hit.numel(), _vp(job_hit), _vp(scores), _vp(known_score), 4, current_stream_ptr()), "nvbio_hip_scatter_rows")
    return known_score


def score_reduce_best_approx(best, state, active, hit_begin, hit_score, hit_loc, hit_seed, worst_score, n_ext, min_ext, max_ext, max_effort,
                             fixed_read_len=0, read_len=None, known_score=None, hit_sink=None, best_sink=None):
    data = best.data if hasattr(best, "data") else best
    check(lib().nvbio_hip_score_reduce_best_approx(active.numel(), _vp(active), _vp(hit_begin), _vp(hit_score), _vp(hit_loc), _vp(hit_seed),
                                                   _vp(read_len), int(fixed_read_len), _vp(data), data.shape[1], int(worst_score),
                                                   _vp(state.trys), _vp(state.counts), int(n_ext), int(min_ext), int(max_ext), int(max_effort), _vp(known_score),
                                                   _vp(hit_sink), _vp(best_sink), current_stream_ptr()), "nvbio_hip_score_reduce_best_approx")
    return best


# ---- the paired-end loop's stages (aligner_best_approx_paired.h:455-700) ------------------------------------------------
def _min_score_table(scheme, max_len, dev):
    return torch.tensor([scheme.min_score(L) if L > 0 else 0 for L in range(max_len + 1)], dtype=torch.int32, device=dev)


def anchor_score_setup(hit_read_id, hit_loc, hit_seed, best, best_o, scheme, anchor, band_len, genome_len, score_limit, fixed_read_len,
                       o_fixed_read_len, rc_offset, table=None, a_read_begin=None, a_read_len=None, o_read_len=None):
    """BestAnchorScoreStream::init_context -> (pattern_begin, text_begin, text_len, min_score); with a_read_len (mates of their own lengths:
    a_read_begin / a_read_len of the anchor mate, o_read_len of the opposite one) also pattern_len, as a fifth value."""
    n = hit_read_id.numel()
    dev = hit_read_id.device
    table = table if table is not None else _min_score_table(scheme, max(fixed_read_len, o_fixed_read_len), dev)
    pb = torch.empty(n, dtype=torch.int64, device=dev); tb = torch.empty(n, dtype=torch.int64, device=dev)
    tl = torch.empty(n, dtype=torch.int32, device=dev); ms = torch.empty(n, dtype=torch.int32, device=dev)
    pl = torch.empty(n, dtype=torch.int32, device=dev) if a_read_len is not None else None
    check(lib().nvbio_hip_anchor_score_setup(n, _vp(hit_read_id), _vp(hit_loc), _vp(hit_seed), _vp(a_read_begin), _vp(a_read_len), _vp(o_read_len),
                                             int(fixed_read_len), int(o_fixed_read_len),
                                             int(rc_offset), int(band_len), int(genome_len), _vp(best.data), _vp(best_o.data), best.stride,
                                             int(scheme.m_match), _vp(table), int(score_limit), int(anchor), _vp(pb), _vp(pl), _vp(tb), _vp(tl), _vp(ms),
                                             current_stream_ptr()), "nvbio_hip_anchor_score_setup")
    table.record_stream(torch.cuda.current_stream())
    return (pb, tb, tl, ms) if pl is None else (pb, tb, tl, ms, pl)


def anchor_score_finish(raw_score, raw_sink, text_begin, min_score, worst_score):
    n = raw_score.numel()
    hs = torch.empty(n, dtype=torch.int32, device=raw_score.device); hk = torch.empty(n, dtype=torch.int32, device=raw_score.device)
    check(lib().nvbio_hip_anchor_score_finish(n, _vp(raw_score), _vp(raw_sink), _vp(text_begin), _vp(min_score), int(worst_score), _vp(hs), _vp(hk),
                                              current_stream_ptr()), "nvbio_hip_anchor_score_finish")
    return hs, hk


def opposite_score_setup(hit_read_id, hit_seed, hit_loc, hit_score, worst_score, best, best_o, scheme, anchor, genome_len, fixed_read_len, o_fixed_read_len,
                         pe_policy, min_frag_len, max_frag_len, pe_overlap, score_limit, table=None, a_read_len=None, o_read_len=None):
    """BestOppositeScoreStream::init_context for the hits of the opposite queue -> dict(valid, min_score, read_rc, genome_begin, genome_end)."""
    from ._lib import PeParamsStruct
    n = hit_read_id.numel()
    dev = hit_read_id.device
    table = table if table is not None else _min_score_table(scheme, max(fixed_read_len, o_fixed_read_len), dev)
    out = dict(valid=torch.empty(n, dtype=torch.uint8, device=dev), min_score=torch.empty(n, dtype=torch.int32, device=dev),
               read_rc=torch.empty(n, dtype=torch.uint8, device=dev), genome_begin=torch.empty(n, dtype=torch.int32, device=dev),
               genome_end=torch.empty(n, dtype=torch.int32, device=dev))
    pp = PeParamsStruct(int(pe_policy), int(min_frag_len), int(max_frag_len), int(bool(pe_overlap)), int(score_limit), int(anchor), int(genome_len))
    check(lib().nvbio_hip_opposite_score_setup(n, _vp(hit_read_id), _vp(hit_seed), _vp(hit_loc), _vp(hit_score), int(worst_score), _vp(a_read_len), _vp(o_read_len),
                                               int(fixed_read_len), int(o_fixed_read_len), _vp(best.data), _vp(best_o.data), best.stride,
                                               int(scheme.m_match), _vp(table), int(scheme.text_gap_open()), int(scheme.text_gap_extension()), C.byref(pp),
                                               _vp(out["valid"]), _vp(out["min_score"]), _vp(out["read_rc"]), _vp(out["genome_begin"]), _vp(out["genome_end"]),
                                               None, 0, None, None, None, current_stream_ptr()), "nvbio_hip_opposite_score_setup")
    table.record_stream(torch.cuda.current_stream())
    return out


def opposite_outputs(n_hits, worst_score, dev):
    """hit.opposite_* of a round as the driver initialises them (aligner_best_approx_paired.h:641-645): worst_score, zeros."""
    o_score = torch.full((n_hits,), worst_score, dtype=torch.int32, device=dev)
    o_score2 = torch.full((n_hits,), worst_score, dtype=torch.int32, device=dev)
    o_loc = torch.zeros(n_hits, dtype=torch.int32, device=dev); o_sink = torch.zeros(n_hits, dtype=torch.int32, device=dev); o_sink2 = torch.zeros(n_hits, dtype=torch.int32, device=dev)
    return o_score, o_score2, o_loc, o_sink, o_sink2


def opposite_memo(n_reads, dev):
    """The opposite-mate memo of a run: 6 words per pair, empty."""
    return torch.zeros((n_reads, 6), dtype=torch.int32, device=dev)


def opposite_memo_lookup(hit_read_id, ow, anchor, memo, worst_score, outputs):
    """Hits whose opposite-mate job equals their pair's memo entry: outputs filled in, ow["valid"] set to 2 (no DP needed)."""
    o_score, o_score2, o_loc, o_sink, o_sink2 = outputs
    check(lib().nvbio_hip_opposite_memo_lookup(hit_read_id.numel(), _vp(hit_read_id), _vp(ow["valid"]), _vp(ow["read_rc"]), _vp(ow["genome_begin"]), _vp(ow["genome_end"]),
                                               _vp(ow["min_score"]), int(anchor), _vp(memo), int(worst_score), _vp(o_score), _vp(o_score2), _vp(o_loc), _vp(o_sink),
                                               _vp(o_sink2), None, current_stream_ptr()), "nvbio_hip_opposite_memo_lookup")


def opposite_memo_update(active, hit_begin, ow, outputs, anchor, memo):
    """Per active pair: its last scored hit of the round becomes the memo entry."""
    check(lib().nvbio_hip_opposite_memo_update(active.numel(), _vp(active), _vp(hit_begin), _vp(ow["valid"]), _vp(ow["read_rc"]), _vp(ow["genome_begin"]),
                                               _vp(ow["genome_end"]), _vp(ow["min_score"]), _vp(outputs[0]), _vp(outputs[3]), int(anchor), _vp(memo),
                                               current_stream_ptr()), "nvbio_hip_opposite_memo_update")


def opposite_score_finish(valid_idx, raw_score, raw_sink, min_score, genome_begin, worst_score, n_hits, out=None):
    """hit.opposite_* of a round: worst_score everywhere, BestOppositeScoreStream::output for the scored hits."""
    dev = min_score.device
    o_score, o_score2, o_loc, o_sink, o_sink2 = out if out is not None else opposite_outputs(n_hits, worst_score, dev)
    check(lib().nvbio_hip_opposite_score_finish(valid_idx.numel(), _vp(valid_idx), None, _vp(raw_score), _vp(raw_sink), _vp(min_score), _vp(genome_begin), int(worst_score),
                                                _vp(o_score), _vp(o_score2), _vp(o_loc), _vp(o_sink), _vp(o_sink2), current_stream_ptr()), "nvbio_hip_opposite_score_finish")
    return o_score, o_score2, o_loc, o_sink, o_sink2


def score_reduce_paired_best_approx(best, best_o, state, active, hit_begin, hit_loc, hit_sink, hit_score, hit_seed, o_loc, o_sink, o_sink2, o_score, o_score2,
                                    anchor, pe_policy, pe_unpaired, score_limit, n_ext, min_ext, max_ext, max_effort, fixed_read_len, read_len=None):
    check(lib().nvbio_hip_score_reduce_paired_best_approx(active.numel(), _vp(active), _vp(hit_begin), _vp(hit_loc), _vp(hit_sink), _vp(hit_score), _vp(hit_seed),
                                                          _vp(o_loc), _vp(o_sink), _vp(o_sink2), _vp(o_score), _vp(o_score2), _vp(read_len), int(fixed_read_len),
                                                          int(anchor), int(pe_policy), int(bool(pe_unpaired)), int(score_limit), _vp(best.data), _vp(best_o.data),
                                                          best.stride, _vp(state.trys), _vp(state.counts), int(n_ext), int(min_ext), int(max_ext), int(max_effort),
                                                          current_stream_ptr()), "nvbio_hip_score_reduce_paired_best_approx")


def mark_discordant(best, best_o):
    check(lib().nvbio_hip_mark_discordant(best.n, _vp(best.data), _vp(best_o.data), best.stride, current_stream_ptr()), "nvbio_hip_mark_discordant")


# ---- driver utilities (what the reference's host code does with thrust between the stages) ------------------------------
def mark_unaligned(active, best, reseed):
    check(lib().nvbio_hip_mark_unaligned(active.numel(), _vp(active), _vp(best.data), _vp(reseed), current_stream_ptr()), "nvbio_hip_mark_unaligned")


def copy_flagged(values, flags):
    """nvbio::copy_flagged: the values whose flag is set, in order (one host sync for the count, as in the reference)."""
    n = values.numel()
    dev = values.device
    out = torch.empty(max(n, 1), dtype=values.dtype, device=dev)
    count = torch.zeros(1, dtype=torch.int32, device=dev)
    tb = int(lib().nvbio_hip_copy_flagged_temp_bytes(n))
    temp = torch.empty(tb, dtype=torch.uint8, device=dev)
    check(lib().nvbio_hip_copy_flagged(n, _vp(values), _vp(flags), _vp(out), _vp(count), _vp(temp), tb, current_stream_ptr()), "nvbio_hip_copy_flagged")
    return out[: int(count.item())]


def traceback_best_setup_mates(best_data, n, band_len, genome_len, mates, mate_offset, want=0, idx=None):
    """traceback_best_setup for pairs whose mates have their own lengths: mates = two ReadBatch-like objects (read_begin / read_len or fixed_len,
    rc_offset); -> (valid, pattern_begin, text_begin, text_len, pattern_len)"""
    import ctypes as C
    dev = best_data.device
    m = idx.numel() if idx is not None else n
    valid = torch.empty(m, dtype=torch.uint8, device=dev)
    pb = torch.empty(m, dtype=torch.int64, device=dev); tb = torch.empty(m, dtype=torch.int64, device=dev)
    tl = torch.empty(m, dtype=torch.int32, device=dev); pl = torch.empty(m, dtype=torch.int32, device=dev)
    ptr = lambda t: C.c_void_p(t.data_ptr() if t is not None else None)
    rb = (C.c_void_p * 2)(ptr(mates[0].read_begin), ptr(mates[1].read_begin))
    rl = (C.c_void_p * 2)(ptr(mates[0].read_len), ptr(mates[1].read_len))
    fl = (C.c_uint32 * 2)(int(mates[0].fixed_len), int(mates[1].fixed_len))
    ro = (C.c_uint64 * 2)(int(mates[0].rc_offset), int(mates[1].rc_offset))
    check(lib().nvbio_hip_traceback_best_setup_mates(m, _vp(idx), _vp(best_data), int(band_len), int(genome_len), rb, rl, fl, ro, C.c_uint64(int(mate_offset)), int(want),
                                                     _vp(valid), _vp(pb), _vp(pl), _vp(tb), _vp(tl), current_stream_ptr()), "nvbio_hip_traceback_best_setup_mates")
    return valid, pb, tb, tl, pl


def traceback_best_setup(best_data, n, band_len, genome_len, fixed_read_len, rc_offset, mate_offset=0, want=0, idx=None, read_begin=None, read_len=None):
    """BestTracebackStream::init_context over best_data[0][idx] -> (valid uint8, pattern_begin, text_begin, text_len[, pattern_len when ragged])."""
    dev = best_data.device
    m = idx.numel() if idx is not None else n
    valid = torch.empty(m, dtype=torch.uint8, device=dev)
    pb = torch.empty(m, dtype=torch.int64, device=dev); tb = torch.empty(m, dtype=torch.int64, device=dev); tl = torch.empty(m, dtype=torch.int32, device=dev)
    pl = torch.empty(m, dtype=torch.int32, device=dev) if read_len is not None else None
    check(lib().nvbio_hip_traceback_best_setup(m, _vp(idx), _vp(best_data), int(band_len), int(genome_len), _vp(read_begin), _vp(read_len), int(fixed_read_len),
                                               int(rc_offset), int(mate_offset), int(want), _vp(valid), _vp(pb), _vp(pl), _vp(tb), _vp(tl), current_stream_ptr()),
          "nvbio_hip_traceback_best_setup")
    return (valid, pb, tb, tl) if read_len is None else (valid, pb, tb, tl, pl)


def finish_alignment(valid, patterns, quals, texts, cigar, cigar_len, source, scheme, best_data, idx=None, mds_stride=256):
    """finish_alignment_kernel: MD strings (nvbio byte code) + the alignments of best_data[0] rewritten (window begin, edit distance, final score).
    Returns (mds uint8[n, stride], mds_len int32[n])."""
    import numpy as np
    n = len(patterns)
    dev = cigar.device
    mds = torch.zeros((max(n, 1), mds_stride), dtype=torch.uint8, device=dev)
    mds_len = torch.zeros(max(n, 1), dtype=torch.int32, device=dev)
    st = scheme.struct()
    lut = (C.c_int32 * 256)(*[st.mismatch[q] for q in range(256)])
    gaps = (C.c_int32 * 4)(st.pattern_gap_open, st.pattern_gap_ext, st.text_gap_open, st.text_gap_ext)
    ps, ts = patterns.struct(), texts.struct()
    check(lib().nvbio_hip_finish_alignment(n, _vp(valid), C.byref(ps), _vp(quals), quals.numel() if quals is not None else 0, C.byref(ts), _vp(cigar), cigar.shape[1],
                                           _vp(cigar_len), _vp(source), int(scheme.m_match), lut, int(getattr(scheme, "m_n_penalty", 1)), gaps, _vp(idx), _vp(best_data),
                                           _vp(mds), mds_stride, _vp(mds_len), current_stream_ptr()), "nvbio_hip_finish_alignment")
    return mds, mds_len[:n]


# ------------------------------------------------------------------------------------------------------------------
# all-mapping mode (aligner_all.h): device stages
# ------------------------------------------------------------------------------------------------------------------
def gather_ranges(hits, counts, count_scan, n_ranges):
    """gather_ranges (mapping.cu:39-67): the size of every SA range, read by read in deque array order -> int64[n_ranges]."""
    out = torch.empty(n_ranges, dtype=torch.int64, device=hits.device)
    check(lib().nvbio_hip_gather_ranges(n_ranges, counts.numel(), _vp(hits), hits.shape[1], _vp(count_scan), _vp(out), current_stream_ptr()),
          "nvbio_hip_gather_ranges")
    return out


def select_all(begin, count, hits, count_scan, range_scan):
    """select_all (select.cu:175-219): hits begin .. begin + count of the global numbering -> (SA row, packed seed, read id), int32[count] each."""
    dev = hits.device
    loc, seed, rid = (torch.empty(count, dtype=torch.int32, device=dev) for _ in range(3))
    check(lib().nvbio_hip_select_all(int(begin), int(count), count_scan.numel(), range_scan.numel(), _vp(hits), hits.shape[1], _vp(count_scan), _vp(range_scan),
                                     _vp(loc), _vp(seed), _vp(rid), current_stream_ptr()), "nvbio_hip_select_all")
    return loc, seed, rid


def mark_straddling(idx_queue, sequence_index, hit_loc, seed_len, flags):
    """mark_straddling (locate_inl.h:213-244) into flags (uint8, in place)."""
    check(lib().nvbio_hip_mark_straddling(idx_queue.numel(), _vp(idx_queue), sequence_index.numel() - 1, _vp(sequence_index), _vp(hit_loc), int(seed_len), _vp(flags),
                                          current_stream_ptr()), "nvbio_hip_mark_straddling")
    return flags


def _window_outputs(m, dev, ragged):
    pb = torch.empty(m, dtype=torch.int64, device=dev); tb = torch.empty(m, dtype=torch.int64, device=dev); tl = torch.empty(m, dtype=torch.int32, device=dev)
    return pb, (torch.empty(m, dtype=torch.int32, device=dev) if ragged else None), tb, tl


def score_all_setup(idx, hit_read_id, hit_loc, hit_seed, band_len, genome_len, fixed_read_len=0, read_begin=None, read_len=None, rc_offset=0):
    """AllScoreStream::init_context over the hits idx[i] -> (pattern_begin, pattern_len | None, text_begin, text_len)."""
    m = idx.numel() if idx is not None else hit_loc.numel()
    pb, pl, tb, tl = _window_outputs(m, hit_loc.device, read_len is not None)
    check(lib().nvbio_hip_score_all_setup(m, _vp(idx), _vp(hit_read_id), _vp(hit_loc), _vp(hit_seed), _vp(read_begin), _vp(read_len), int(fixed_read_len), int(rc_offset),
                                          int(band_len), int(genome_len), _vp(pb), _vp(pl), _vp(tb), _vp(tl), current_stream_ptr()), "nvbio_hip_score_all_setup")
    return pb, pl, tb, tl


def score_all_output(idx, hit_read_id, hit_loc, hit_seed, score, min_score_by_len, fixed_read_len=0, read_len=None):
    """AllScoreStream::output: (flags uint8, io::Alignment words int64, read ids int32) per job."""
    m = score.numel()
    dev = score.device
    flags = torch.empty(m, dtype=torch.uint8, device=dev); aln = torch.empty(m, dtype=torch.int64, device=dev); rid = torch.empty(m, dtype=torch.int32, device=dev)
    check(lib().nvbio_hip_score_all_output(m, _vp(idx), _vp(hit_read_id), _vp(hit_loc), _vp(hit_seed), _vp(score), _vp(min_score_by_len), _vp(read_len),
                                           int(fixed_read_len), _vp(flags), _vp(aln), _vp(rid), current_stream_ptr()), "nvbio_hip_score_all_output")
    return flags, aln, rid


def traceback_all_setup(alignments, read_id, band_len, genome_len, fixed_read_len=0, read_begin=None, read_len=None, rc_offset=0):
    """AllTracebackStream::init_context -> (pattern_begin, pattern_len | None, text_begin, text_len)."""
    m = alignments.numel()
    pb, pl, tb, tl = _window_outputs(m, alignments.device, read_len is not None)
    check(lib().nvbio_hip_traceback_all_setup(m, _vp(alignments), _vp(read_id), _vp(read_begin), _vp(read_len), int(fixed_read_len), int(rc_offset), int(band_len),
                                              int(genome_len), _vp(pb), _vp(pl), _vp(tb), _vp(tl), current_stream_ptr()), "nvbio_hip_traceback_all_setup")
    return pb, pl, tb, tl


def _all_temp(n, dev):
    return torch.empty(int(lib().nvbio_hip_all_mapping_temp_bytes(int(n))), dtype=torch.uint8, device=dev)


def inclusive_scan(x):
    """thrust::inclusive_scan over int32 (as uint32) or int64 (as uint64) values."""
    out = torch.empty_like(x)
    t = _all_temp(x.numel(), x.device)
    fn = lib().nvbio_hip_inclusive_scan_u32 if x.dtype == torch.int32 else lib().nvbio_hip_inclusive_scan_u64
    check(fn(x.numel(), _vp(x), _vp(out), _vp(t), t.numel(), current_stream_ptr()), "nvbio_hip_inclusive_scan")
    return out


def sort_hi_bits(keys):
    """Aligner::sort_hi_bits (aligner_sort.cu:39-64): the permutation that stably sorts keys >> 16 -> int32[n]."""
    idx = torch.empty(keys.numel(), dtype=torch.int32, device=keys.device)
    t = _all_temp(keys.numel(), keys.device)
    check(lib().nvbio_hip_sort_hi_bits(keys.numel(), _vp(keys), _vp(idx), _vp(t), t.numel(), current_stream_ptr()), "nvbio_hip_sort_hi_bits")
    return idx


def sort_hits(hit_read_id, hit_loc, hit_seed):
    """Aligner::sort_64_bits over SortingKeys + the dedup flags (aligner_all.h:229-247, :492-509) -> (idx int32[n], first-of-run flags uint8[n])."""
    n = hit_loc.numel()
    idx = torch.empty(n, dtype=torch.int32, device=hit_loc.device); first = torch.empty(n, dtype=torch.uint8, device=hit_loc.device)
    t = _all_temp(n, hit_loc.device)
    check(lib().nvbio_hip_sort_hits(n, _vp(hit_read_id), _vp(hit_loc), _vp(hit_seed), _vp(idx), _vp(first), _vp(t), t.numel(), current_stream_ptr()), "nvbio_hip_sort_hits")
    return idx, first


def sort_hits_pingpong(hit_read_id, hit_loc, hit_seed):
    """sort_hits + the index the reference's mark_straddling reads (aligner_all.h:520: the result pointer of sort_hi_bits, a half of the ping-pong
    index buffer that sort_64_bits has sorted through since) -> (idx int32[n], first-of-run flags uint8[n], stale idx int32[n])."""
    n = hit_loc.numel()
    idx = torch.empty(n, dtype=torch.int32, device=hit_loc.device); first = torch.empty(n, dtype=torch.uint8, device=hit_loc.device)
    stale = torch.empty(n, dtype=torch.int32, device=hit_loc.device)
    t = _all_temp(n, hit_loc.device)
    check(lib().nvbio_hip_sort_hits_pingpong(n, _vp(hit_read_id), _vp(hit_loc), _vp(hit_seed), _vp(idx), _vp(first), _vp(stale), _vp(t), t.numel(), current_stream_ptr()),
          "nvbio_hip_sort_hits_pingpong")
    return idx, first, stale


def traceback_best_known(best_data, best_sink, n, idx=None):
    """Score and sink of every best alignment as the banded scorer reports them over the traceback's window (kept by the reduction):
    (score int32[m], sink int32[m, 2]) for batch_banded_alignment_traceback(known=...).  best_sink None: the scores alone (sink None)."""
    m = idx.numel() if idx is not None else n
    dev = best_data.device
    score = torch.empty(m, dtype=torch.int32, device=dev); sink = torch.empty((m, 2), dtype=torch.int32, device=dev) if best_sink is not None else None
    check(lib().nvbio_hip_traceback_best_known(m, _vp(idx), _vp(best_data), _vp(best_sink), _vp(score), _vp(sink), current_stream_ptr()), "nvbio_hip_traceback_best_known")
    return score, sink
