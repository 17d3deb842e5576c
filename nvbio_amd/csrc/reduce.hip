// reduce.hip -- nvBowtie's score reduction and mapping-quality stages on gfx950.
//   score_reduce_kernel   nvBowtie/bowtie2/cuda/reduce_inl.h:71-160   (best / second-best per read)
//   BowtieMapq2 / Mapq3   nvBowtie/bowtie2/cuda/mapq.h:42-330
//   io::Alignment         nvbio/io/alignments.h:80-131 ; distinct_alignments alignments_inl.h:35-47
// Both are thin, HBM-streaming stages: one lane per read, a sequential walk over the read's extension
// results (the update rule is order dependent), 16 B of state per read.
#include "common.h"
#include <limits.h>

namespace nvb {

struct IoAln { uint32_t w, align; };      // {score_sgn:1, score:17, ed:10, rc:1, mate:1, paired:1, discordant:1}, m_align

__device__ __forceinline__ IoAln io_aln_make(uint32_t pos, uint32_t ed, int32_t score, uint32_t rc)
{
    const uint32_t mag = score < 0 ? uint32_t(-score) : uint32_t(score);
    IoAln a;
    a.w = (score < 0 ? 1u : 0u) | ((mag & 0x1FFFFu) << 1) | ((ed & 0x3FFu) << 18) | ((rc & 1u) << 28);
    a.align = pos;
    return a;
}
__device__ __forceinline__ int32_t  io_aln_score(const IoAln a) { const int32_t m = int32_t((a.w >> 1) & 0x1FFFFu); return (a.w & 1u) ? -m : m; }
__device__ __forceinline__ uint32_t io_aln_rc(const IoAln a) { return (a.w >> 28) & 1u; }
__device__ __forceinline__ bool     distinct_alignments(uint32_t pos1, uint32_t rc1, uint32_t pos2, uint32_t rc2, uint32_t dist)
{
    if (rc1 != rc2) return true;
    return !(pos1 >= pos2 - min(pos2, dist) && pos1 <= pos2 + dist);
}

__global__ void __launch_bounds__(256)
score_reduce_kernel(uint32_t n_active, const uint32_t* __restrict__ read_ids, const uint64_t* __restrict__ hit_begin,
                    const int32_t* __restrict__ hit_score, const uint32_t* __restrict__ hit_loc, const uint8_t* __restrict__ hit_rc,
                    const uint32_t* __restrict__ read_len, uint32_t fixed_len, uint2* __restrict__ best, uint32_t best_stride)
{
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= n_active) return;
    const uint32_t read_id = read_ids ? read_ids[t] : t;
    const uint2 b1 = best[read_id], b2 = best[read_id + best_stride];
    IoAln a1 = { b1.x, b1.y }, a2 = { b2.x, b2.y };
    const uint32_t len = read_len ? read_len[read_id] : fixed_len;
    const uint64_t hb = hit_begin[t], he = hit_begin[t + 1];
    for (uint64_t i = hb; i < he; ++i)
    {
        const int32_t score = hit_score[i]; const uint32_t g_pos = hit_loc[i], rc = hit_rc[i];
        // locations already visited are skipped (reduce_inl.h:111-114)
        if ((rc == io_aln_rc(a1) && g_pos == a1.align) || (rc == io_aln_rc(a2) && g_pos == a2.align)) continue;
        if (score > io_aln_score(a1)) { a2 = a1; a1 = io_aln_make(g_pos, 0u, score, rc); }
        else if (score > io_aln_score(a2) && distinct_alignments(a1.align, io_aln_rc(a1), g_pos, rc, len / 2u)) a2 = io_aln_make(g_pos, 0u, score, rc);
    }
    best[read_id] = make_uint2(a1.w, a1.align);
    best[read_id + best_stride] = make_uint2(a2.w, a2.align);
}

// score_reduce_kernel with ReduceBestApproxContext (reduce.h:63-105): the same walk, plus the extension give-up
// counters of the best-approx loop.  An update of the best or second best refills the read's try counter with
// max_effort; any other result, once n_ext + idx reaches min_ext and the hit did not come from the top seed range,
// burns one try, and the read's hit deque is erased (so the next selection round drops the read) when the tries
// run out or n_ext + idx reaches max_ext.  hit_score is the raw DP result, clamped to worst_score first as
// BestScoreStream::output does (score_best_inl.h:139); `active` holds packed_read words.
__global__ void __launch_bounds__(256)
score_reduce_best_approx_kernel(uint32_t n_active, const uint32_t* __restrict__ active, const uint64_t* __restrict__ hit_begin,
                                const int32_t* __restrict__ hit_score, const uint32_t* __restrict__ hit_loc, const uint32_t* __restrict__ hit_seed,
                                const uint32_t* __restrict__ read_len, uint32_t fixed_len, uint2* __restrict__ best, uint32_t best_stride,
                                int32_t worst_score, uint32_t* __restrict__ trys, uint32_t* __restrict__ hit_counts,
                                uint32_t n_ext, uint32_t min_ext, uint32_t max_ext, uint32_t max_effort, const int32_t* __restrict__ known_score,
                                const uint2* __restrict__ hit_sink, uint2* __restrict__ best_sink)
{
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= n_active) return;
    const uint32_t read_id = active[t] & 0x7FFFFFFFu;
    uint2 s1 = make_uint2(0u, 0u); bool s1_new = false;      // the DP sink of a new best alignment (kept for the traceback, which can then skip its score pass)
    const uint2 b1 = best[read_id], b2 = best[read_id + best_stride];
    IoAln a1 = { b1.x, b1.y }, a2 = { b2.x, b2.y };
    const uint32_t len = read_len ? read_len[read_id] : fixed_len;
    const uint64_t hb = hit_begin[t], he = hit_begin[t + 1];
    uint32_t tr = trys[read_id];
    bool erase = false;
    for (uint64_t i = hb; i < he; ++i)
    {
        // (known_score: the score of a hit whose placement was already recorded, set by score_best_setup instead of re-running its DP)
        const int32_t known = known_score ? known_score[i] : INT32_MIN;
        const int32_t score = known != INT32_MIN ? known : max(hit_score[i], worst_score);
        const uint32_t g_pos = hit_loc[i], seed = hit_seed[i], rc = (seed >> 13) & 1u, top_flag = (seed >> 14) & 1u;
        if ((rc == io_aln_rc(a1) && g_pos == a1.align) || (rc == io_aln_rc(a2) && g_pos == a2.align)) continue;
        if (score > io_aln_score(a1)) { tr = max_effort; a2 = a1; a1 = io_aln_make(g_pos, 0u, score, rc); if (best_sink) { s1 = hit_sink[i]; s1_new = true; } }
        else if (score > io_aln_score(a2) && distinct_alignments(a1.align, io_aln_rc(a1), g_pos, rc, len / 2u)) { tr = max_effort; a2 = io_aln_make(g_pos, 0u, score, rc); }
        else if (tr > 0u) {
            const uint32_t idx = uint32_t(i - hb);
            if ((n_ext + idx >= min_ext && top_flag == 0u && --tr == 0u) || (n_ext + idx >= max_ext)) erase = true;
        }
    }
    trys[read_id] = tr;
    if (erase) hit_counts[read_id] = 0u;
    best[read_id] = make_uint2(a1.w, a1.align);
    best[read_id + best_stride] = make_uint2(a2.w, a2.align);
    if (s1_new) best_sink[read_id] = s1;
}

// score and sink of every best alignment as the banded scorer reports them over the traceback's window (the same window and pattern
// as the extension that found it): what nvbio_hip_banded_gotoh_traceback_qual_known takes.  Unaligned entries: a failed alignment.
__global__ void __launch_bounds__(256)
traceback_best_known_kernel(uint32_t n, const uint32_t* __restrict__ idx, const uint2* __restrict__ best, const uint2* __restrict__ best_sink,
                            int32_t* __restrict__ out_score, uint2* __restrict__ out_sink)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t r = idx ? idx[i] : i;
    const uint2 a = best[r];
    const bool aligned = a.y != 0xFFFFFFFFu;
    const int32_t m = int32_t((a.x >> 1) & 0x1FFFFu);
    out_score[i] = aligned ? ((a.x & 1u) ? -m : m) : -(1 << 30);
    if (out_sink) out_sink[i] = aligned ? best_sink[r] : make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
}

// init_alignments_kernel (nvBowtie/bowtie2/cuda/aligner.h:323-346): both slots unaligned (pos -1, ed max) with the
// read's worst acceptable score, so that only extensions above the threshold are ever recorded.  The reference
// passes `mate` in the constructor's rc position (aligner.h:342-343); kept.
__global__ void __launch_bounds__(256)
init_alignments_kernel(uint32_t n_reads, const uint32_t* __restrict__ read_len, uint32_t fixed_len,
                       const int32_t* __restrict__ worst_score_by_len, uint32_t mate, uint2* __restrict__ best, uint32_t best_stride)
{
    const uint32_t r = blockIdx.x * 256u + threadIdx.x;
    if (r >= n_reads) return;
    const uint32_t len = read_len ? read_len[r] : fixed_len;
    const IoAln a = io_aln_make(0xFFFFFFFFu, 255u, worst_score_by_len[len], mate);
    best[r] = make_uint2(a.w, a.align);
    best[r + best_stride] = make_uint2(a.w, a.align);
}

// ------------------------------------------------------------------ paired-end reduction
// score_reduce_paired_kernel (reduce_inl.h:355-500) with try_update / update_* / replace_best (:160-350),
// frame_opposite_mate (alignment_utils.h:61-98), the paired distinct_alignments (alignments_inl.h:66-80,121-136)
__device__ __forceinline__ IoAln io_aln_make_full(uint32_t pos, uint32_t ed, int32_t score, uint32_t rc, uint32_t mate, bool paired)
{
    IoAln a = io_aln_make(pos, ed, score, rc);
    a.w |= ((mate & 1u) << 29) | ((paired ? 1u : 0u) << 30);
    return a;
}
__device__ __forceinline__ uint32_t io_aln_sink(const IoAln a)   { return (a.w >> 18) & 0x3FFu; }
__device__ __forceinline__ uint32_t io_aln_mate(const IoAln a)   { return (a.w >> 29) & 1u; }
__device__ __forceinline__ bool     io_aln_paired(const IoAln a) { return ((a.w >> 30) & 1u) != 0u && a.align != 0xFFFFFFFFu; }
struct IoPair { IoAln a, o; };
struct IoBestPairs { IoAln a1, a2, o1, o2; };
__device__ __forceinline__ IoAln   pair_mate(const IoPair& p, uint32_t m) { return m == io_aln_mate(p.a) ? p.a : p.o; }
__device__ __forceinline__ int32_t pair_score(const IoPair& p) { return io_aln_score(p.a) + io_aln_score(p.o); }
__device__ __forceinline__ int32_t bp_best_score(const IoBestPairs& b)   { return io_aln_score(b.a1) + (io_aln_paired(b.a1) ? io_aln_score(b.o1) : 0); }
__device__ __forceinline__ int32_t bp_second_score(const IoBestPairs& b) { return io_aln_score(b.a2) + (io_aln_paired(b.a2) ? io_aln_score(b.o2) : 0); }
__device__ __forceinline__ bool distinct_pairs(const IoPair& p1, const IoPair& p2, uint32_t dist)
{
    const IoAln a1 = pair_mate(p1, 0), o1 = pair_mate(p1, 1), a2 = pair_mate(p2, 0), o2 = pair_mate(p2, 1);
    const uint32_t apos1 = a1.align + io_aln_sink(a1), opos1 = o1.align + io_aln_sink(o1);
    const uint32_t apos2 = a2.align + io_aln_sink(a2), opos2 = o2.align + io_aln_sink(o2);
    if (io_aln_rc(a1) != io_aln_rc(a2) || io_aln_rc(o1) != io_aln_rc(o2)) return true;
    return !((apos1 >= apos2 - min(apos2, dist) && apos1 <= apos2 + dist) && (opos1 >= opos2 - min(opos2, dist) && opos1 <= opos2 + dist));
}
__device__ __forceinline__ bool distinct_alns(const IoAln p1, const IoAln p2, uint32_t dist)
{
    return distinct_alignments(p1.align + io_aln_sink(p1), io_aln_rc(p1), p2.align + io_aln_sink(p2), io_aln_rc(p2), dist);
}
// try_update (reduce_inl.h:204-340): true when the result was absorbed (an update, or a location already recorded); every
// actual update refills the read's try counter (ReduceBestApproxContext::best_score / second_score) when CTX
template <bool CTX>
__device__ __forceinline__ bool try_update_pair(IoBestPairs& b, const IoPair& pair, uint32_t min_distance, uint32_t& tr, uint32_t max_effort)
{
    const int32_t score = pair_score(pair);
    const IoPair p0 = { b.a1, b.o1 }, p1 = { b.a2, b.o2 };
    if (!distinct_pairs(p0, pair, min_distance)) { if (score > bp_best_score(b)) { if (CTX) tr = max_effort; b.a1 = pair.a; b.o1 = pair.o; } return true; }
    else if (!distinct_pairs(p1, pair, min_distance)) {
        if (score > bp_best_score(b)) { if (CTX) tr = max_effort; b.a2 = b.a1; b.o2 = b.o1; b.a1 = pair.a; b.o1 = pair.o; }
        else if (score > bp_second_score(b)) { if (CTX) tr = max_effort; b.a2 = pair.a; b.o2 = pair.o; }
        return true;
    }
    else if (!io_aln_paired(b.a1) || score > bp_best_score(b)) { if (CTX) tr = max_effort; b.a2 = b.a1; b.o2 = b.o1; b.a1 = pair.a; b.o1 = pair.o; return true; }
    else if (!io_aln_paired(b.a2) || score > bp_second_score(b)) { if (CTX) tr = max_effort; b.a2 = pair.a; b.o2 = pair.o; return true; }
    return false;
}
template <bool CTX>
__device__ __forceinline__ bool try_update_single(IoAln& a1, IoAln& a2, const IoAln a, uint32_t min_distance, uint32_t& tr, uint32_t max_effort)
{
    if (!distinct_alns(a1, a, min_distance)) { if (io_aln_score(a) > io_aln_score(a1)) { if (CTX) tr = max_effort; a1 = a; } return true; }
    else if (!distinct_alns(a2, a, min_distance)) {
        if (io_aln_score(a) > io_aln_score(a1)) { if (CTX) tr = max_effort; a2 = a1; a1 = a; }
        else if (io_aln_score(a) > io_aln_score(a2)) { if (CTX) tr = max_effort; a2 = a; }
        return true;
    }
    else if (io_aln_score(a) > io_aln_score(a1)) { if (CTX) tr = max_effort; a2 = a1; a1 = a; return true; }
    else if (io_aln_score(a) > io_aln_score(a2)) { if (CTX) tr = max_effort; a2 = a; return true; }
    return false;
}

struct PairedReduceParams {
    uint32_t n_active; const uint32_t* read_ids; const uint64_t* hit_begin;
    const uint32_t* hit_loc; const uint32_t* hit_sink; const int32_t* hit_score; const uint8_t* hit_rc;
    const uint32_t* o_loc; const uint32_t* o_sink; const uint32_t* o_sink2; const int32_t* o_score; const int32_t* o_score2;
    const uint32_t* read_len; uint32_t fixed_len;
    uint32_t anchor; int32_t pe_policy, pe_unpaired, score_limit;
    uint2* best; uint2* best_o; uint32_t best_stride;
    // CTX (the best-approx loop): read_ids are packed_read words, strands / top flags come from packed seeds, and the
    // give-up counters of ReduceBestApproxContext (reduce.h:63-105) are maintained
    const uint32_t* hit_seed; uint32_t* trys; uint32_t* hit_counts; uint32_t n_ext, min_ext, max_ext, max_effort;
};

template <bool CTX>
__global__ void __launch_bounds__(256) score_reduce_paired_kernel(const PairedReduceParams p)
{
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= p.n_active) return;
    const uint32_t read_id = CTX ? (p.read_ids[t] & 0x7FFFFFFFu) : (p.read_ids ? p.read_ids[t] : t);
    auto ld = [](const uint2* q, uint32_t i) { const uint2 v = q[i]; IoAln a = { v.x, v.y }; return a; };
    IoBestPairs b = { ld(p.best, read_id), ld(p.best, read_id + p.best_stride), ld(p.best_o, read_id), ld(p.best_o, read_id + p.best_stride) };
    const uint32_t min_distance = (p.read_len ? p.read_len[read_id] : p.fixed_len) / 4u;
    uint32_t tr = CTX ? p.trys[read_id] : 0u;
    bool erase = false;
    const uint64_t hb = p.hit_begin[t];
    for (uint64_t i = hb; i < p.hit_begin[t + 1]; ++i)
    {
        const uint32_t seed = CTX ? p.hit_seed[i] : 0u;
        const uint32_t rc = CTX ? ((seed >> 13) & 1u) : p.hit_rc[i];
        const bool anchor_fw = !rc;
        bool o_fw;      // frame_opposite_mate: only the orientation is needed here
        switch (p.pe_policy) {
        case 0: case 3: o_fw = anchor_fw; break;         // FF, RR
        default:        o_fw = !anchor_fw; break;        // FR, RF
        }
        const uint32_t o_rc = !o_fw;
        const IoPair pair  = { io_aln_make_full(p.hit_loc[i], p.hit_sink[i] - p.hit_loc[i], p.hit_score[i], rc, p.anchor, p.o_score[i] > p.score_limit),
                               io_aln_make_full(p.o_loc[i], p.o_sink[i] - p.o_loc[i], p.o_score[i], o_rc, p.anchor ^ 1u, p.o_score[i] > p.score_limit) };
        const IoPair pair2 = { io_aln_make_full(p.hit_loc[i], p.hit_sink[i] - p.hit_loc[i], p.hit_score[i], rc, p.anchor, p.o_score2[i] > p.score_limit),
                               io_aln_make_full(p.o_loc[i], p.o_sink2[i] - p.o_loc[i], p.o_score2[i], o_rc, p.anchor ^ 1u, p.o_score2[i] > p.score_limit) };
        bool updated = false;
        if (io_aln_paired(pair.a)) {
            updated |= try_update_pair<CTX>(b, pair, min_distance, tr, p.max_effort);
            if (io_aln_paired(pair2.a)) updated |= try_update_pair<CTX>(b, pair2, min_distance, tr, p.max_effort);
        } else if (p.pe_unpaired && !io_aln_paired(b.a1)) {
            // no paired alignment yet: best two of mate 1 live in m_a*, of mate 2 in m_o*
            if (p.anchor) updated |= try_update_single<CTX>(b.o1, b.o2, pair.a, min_distance, tr, p.max_effort);
            else          updated |= try_update_single<CTX>(b.a1, b.a2, pair.a, min_distance, tr, p.max_effort);
        }
        if (CTX && !updated && tr > 0u) {                  // ReduceBestApproxContext::failure
            const uint32_t idx = uint32_t(i - hb), top_flag = (seed >> 14) & 1u;
            if ((p.n_ext + idx >= p.min_ext && top_flag == 0u && --tr == 0u) || (p.n_ext + idx >= p.max_ext)) erase = true;
        }
    }
    if (CTX) { p.trys[read_id] = tr; if (erase) p.hit_counts[read_id] = 0u; }
    p.best[read_id] = make_uint2(b.a1.w, b.a1.align); p.best[read_id + p.best_stride] = make_uint2(b.a2.w, b.a2.align);
    p.best_o[read_id] = make_uint2(b.o1.w, b.o1.align); p.best_o[read_id + p.best_stride] = make_uint2(b.o2.w, b.o2.align);
}

// mark_discordant_kernel (aligner_init.cu:457-480): a pair whose mates are both uniquely aligned but not concordant
__global__ void __launch_bounds__(256) mark_discordant_kernel(uint32_t n_reads, uint2* __restrict__ best, uint2* __restrict__ best_o, uint32_t stride)
{
    const uint32_t r = blockIdx.x * 256u + threadIdx.x;
    if (r >= n_reads) return;
    const uint2 a1 = best[r], o1 = best_o[r];
    const bool concordant = ((a1.x >> 30) & 1u) && !((a1.x >> 31) & 1u);
    if (!concordant && a1.y != 0xFFFFFFFFu && best[r + stride].y == 0xFFFFFFFFu && o1.y != 0xFFFFFFFFu && best_o[r + stride].y == 0xFFFFFFFFu) {
        best[r] = make_uint2(a1.x | 0xC0000000u, a1.y);
        best_o[r] = make_uint2(o1.x | 0xC0000000u, o1.y);
    }
}

// compute_target_score (alignment_utils.h:100-111) bounded by the pair's perfect score
__device__ __forceinline__ int32_t target_pair_score(const IoBestPairs& b, int32_t a_worst, int32_t o_worst, int32_t a_optimal, int32_t o_optimal)
{
    int32_t target;
    if (!io_aln_paired(b.a2)) target = a_worst + o_worst;
    else { const int32_t delta = bp_best_score(b) - bp_second_score(b); target = bp_second_score(b) + (delta * 3) / 4; }      // bowtie2's 'tighten = 3'
    return min(target + 1, a_optimal + o_optimal);
}

// BestAnchorScoreStream::init_context (score_paired_inl.h:54-135): window, pattern and threshold of every anchor hit.  A hit at a
// location already recorded in the read's best pairs is skipped: threshold INT32_MAX and an empty window (the DP then fails and
// the hit scores worst_score, as when the reference does not run it).  The reference's skip test also reads context->min_score
// before setting it (:128, an uninitialised read); that term is taken as false.
struct AnchorSetupParams {
    uint32_t n; const uint32_t* hit_read_id; const uint32_t* hit_loc; const uint32_t* hit_seed;
    const uint64_t* a_read_begin; const uint32_t* a_read_len; const uint32_t* o_read_len; uint32_t a_fixed_len, o_fixed_len; uint64_t rc_offset;
    uint32_t band_len, genome_len; const uint2* best; const uint2* best_o; uint32_t best_stride;
    int32_t match; const int32_t* min_score_by_len; int32_t score_limit; uint32_t anchor;
    uint64_t* pat_begin; uint32_t* pat_len; uint64_t* text_begin; uint32_t* text_len; int32_t* min_score;
};
__global__ void __launch_bounds__(256) anchor_score_setup_kernel(const AnchorSetupParams p)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= p.n) return;
    const uint32_t read_id = p.hit_read_id[i], g_pos = p.hit_loc[i], read_rc = (p.hit_seed[i] >> 13) & 1u;
    const uint32_t a_len = p.a_read_len ? p.a_read_len[read_id] : p.a_fixed_len, o_len = p.o_read_len ? p.o_read_len[read_id] : p.o_fixed_len;
    const int32_t a_optimal = int32_t(a_len) * p.match, a_worst = p.min_score_by_len[a_len];
    const int32_t o_optimal = int32_t(o_len) * p.match, o_worst = p.min_score_by_len[o_len];
    auto ld = [](const uint2* q, uint32_t k) { const uint2 v = q[k]; IoAln a = { v.x, v.y }; return a; };
    const IoBestPairs b = { ld(p.best, read_id), ld(p.best, read_id + p.best_stride), ld(p.best_o, read_id), ld(p.best_o, read_id + p.best_stride) };
    const int32_t target_mate = max(target_pair_score(b, a_worst, o_worst, a_optimal, o_optimal) - o_optimal, a_worst);
    const uint32_t gb = g_pos > p.band_len / 2u ? g_pos - p.band_len / 2u : 0u;
    const uint32_t sum = gb + p.band_len + a_len;
    const uint32_t ge = sum < p.genome_len ? sum : p.genome_len;
    const uint32_t mate = p.anchor;
    const bool skip = (mate == io_aln_mate(b.a1) && read_rc == io_aln_rc(b.a1) && g_pos == b.a1.align) ||
                      (mate == io_aln_mate(b.o1) && read_rc == io_aln_rc(b.o1) && g_pos == b.o1.align) ||
                      (mate == io_aln_mate(b.a2) && read_rc == io_aln_rc(b.a2) && g_pos == b.a2.align) ||
                      (mate == io_aln_mate(b.o2) && read_rc == io_aln_rc(b.o2) && g_pos == b.o2.align);
    p.text_begin[i] = gb;
    p.text_len[i] = (skip || ge <= gb) ? 0u : ge - gb;
    p.min_score[i] = skip ? 0x7FFFFFFF : max(target_mate, p.score_limit);
    p.pat_begin[i] = (p.a_read_begin ? p.a_read_begin[read_id] : uint64_t(read_id) * p.a_fixed_len) + (read_rc ? p.rc_offset : 0ull);
    if (p.pat_len) p.pat_len[i] = a_len;
}
// BestAnchorScoreStream::output (:137-150)
__global__ void __launch_bounds__(256)
anchor_score_finish_kernel(uint32_t n, const int32_t* __restrict__ raw_score, const uint2* __restrict__ raw_sink, const uint64_t* __restrict__ text_begin,
                           const int32_t* __restrict__ min_score, int32_t worst_score, int32_t* __restrict__ hit_score, uint32_t* __restrict__ hit_sink)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const int32_t s = raw_score[i];
    hit_score[i] = s >= min_score[i] ? s : worst_score;
    hit_sink[i] = uint32_t(text_begin[i]) + raw_sink[i].x;
}
// BestOppositeScoreStream::output (score_opposite_inl.h:203-235) for the scored (valid) hits idx[k]; all other hits keep the
// worst score the driver filled in (aligner_best_approx_paired.h:641-645)
__global__ void __launch_bounds__(256)
opposite_score_finish_kernel(uint32_t n_valid, const uint32_t* __restrict__ idx, const int32_t* __restrict__ raw_score, const uint2* __restrict__ raw_sink,
                             const int32_t* __restrict__ min_score, const uint32_t* __restrict__ genome_begin, int32_t worst_score, const uint8_t* __restrict__ valid_flags,
                             int32_t* __restrict__ o_score, int32_t* __restrict__ o_score2, uint32_t* __restrict__ o_loc, uint32_t* __restrict__ o_sink, uint32_t* __restrict__ o_sink2)
{
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (k >= n_valid) return;
    const uint32_t h = idx ? idx[k] : k;          // idx == NULL: one raw result per hit, `valid_flags` says which were scored
    if (!idx && valid_flags && valid_flags[k] == 2u) return;          // answered from the opposite-mate memo (opposite_memo_lookup)
    if (!idx && valid_flags && !valid_flags[k]) { o_score[h] = worst_score; o_score2[h] = worst_score; o_loc[h] = 0u; o_sink[h] = 0u; o_sink2[h] = 0u; return; }
    const int32_t s = raw_score[k];
    const uint32_t gb = genome_begin[h], sx = raw_sink[k].x;
    o_score[h] = s >= min_score[h] ? s : worst_score;
    o_score2[h] = worst_score;
    o_loc[h] = gb;
    o_sink[h] = gb + (sx != 0xFFFFFFFFu ? sx : 0u);
    o_sink2[h] = gb;
}

// ------------------------------------------------------------------ anchor memo
// The anchor's banded DP is a pure function of (pair, which mate, strand, window).  The reference's skip test (anchor_score_setup above)
// compares a hit's position with the recorded alignments' *window* begins and so almost never fires: every seed of a read that points at
// the placement the read already tried -- a dozen per read on a unique genome -- pays the DP again and gets the identical score (measured:
// 7.7 M anchor DPs per 500 k pairs where ~1.5 M are distinct, bench config 5).  One entry per pair remembers the last scored job
// {window begin, strand, mate} and its raw outputs; a later hit with the same job is answered from it.
//   mark    (after anchor_score_setup, before the DP): a hit whose job is the pair's entry, or the job of the hit just before it in the same
//           round, gets an empty text (its lane of the scorer returns at once) and from_memo[i] = 1 (entry) / 2 (same as hit i - 1)
//   finish  the reference's output step, taking a marked hit's raw score and sink from where they are
//   update  per active read, the last hit of the round becomes the pair's entry
// memo: 6 words per pair {window begin, 1 | strand << 1 | mate << 2, raw score, sink.x, sink.y, -}, zero-initialised by the caller.
__global__ void __launch_bounds__(256)
anchor_memo_mark_kernel(uint32_t n_hits, const uint32_t* __restrict__ hit_read_id, const uint32_t* __restrict__ hit_seed, const uint64_t* __restrict__ text_begin,
                        const uint32_t* __restrict__ text_len_in, uint32_t anchor, const uint32_t* __restrict__ memo, uint8_t* __restrict__ from_memo,
                        uint32_t* __restrict__ text_len_out, uint32_t* __restrict__ live_count, uint32_t* __restrict__ live_idx)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    __shared__ uint32_t wave_live[4], block_base;
    uint8_t f = 0u;
    const uint32_t tl = i < n_hits ? text_len_in[i] : 0u;
    if (i < n_hits && tl != 0u)
    {
        const uint32_t r = hit_read_id[i], rc = (hit_seed[i] >> 13) & 1u, gb = uint32_t(text_begin[i]);
        // the hit before this one in the round: same pair, same strand, same window, and scored (not skipped)
        if (i > 0u && hit_read_id[i - 1u] == r && ((hit_seed[i - 1u] >> 13) & 1u) == rc && uint32_t(text_begin[i - 1u]) == gb && text_len_in[i - 1u] != 0u) f = 2u;
        else
        {
            const uint32_t* m = memo + uint64_t(r) * 6u;
            if (m[1] == (1u | (rc << 1) | (anchor << 2)) && m[0] == gb) f = 1u;
        }
    }
    if (i < n_hits) { from_memo[i] = f; text_len_out[i] = f ? 0u : tl; }
    // the hits that still need their DP, as a list (one atomic per block): what the wave-per-job scorer runs when there are few of them
    if (live_count)
    {
        const bool live = i < n_hits && tl != 0u && f == 0u;
        const uint64_t m = __ballot(live);
        const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
        if (lane == 0u) wave_live[wv] = uint32_t(__popcll(m));
        __syncthreads();
        if (threadIdx.x == 0u) { const uint32_t tot = wave_live[0] + wave_live[1] + wave_live[2] + wave_live[3]; block_base = tot ? atomicAdd(live_count, tot) : 0u; }
        __syncthreads();
        uint32_t base = block_base;
        for (uint32_t k = 0; k < wv; ++k) base += wave_live[k];
        if (live) live_idx[base + uint32_t(__popcll(m & ((1ull << lane) - 1ull)))] = i;
    }
}
// the raw (score, sink) of hit i: from the scorer, the memo, or the hit it repeats
__device__ __forceinline__ void anchor_raw(uint32_t i, const uint8_t* __restrict__ from_memo, const uint32_t* __restrict__ hit_read_id, const uint32_t* __restrict__ memo,
                                           const int32_t* __restrict__ raw_score, const uint2* __restrict__ raw_sink, int32_t& s, uint2& k)
{
    while (from_memo[i] == 2u) --i;                    // (a run of repeats ends at the hit that was scored, or answered from the entry)
    if (from_memo[i] == 1u) { const uint32_t* m = memo + uint64_t(hit_read_id[i]) * 6u; s = int32_t(m[2]); k = make_uint2(m[3], m[4]); }
    else { s = raw_score[i]; k = raw_sink[i]; }
}
__global__ void __launch_bounds__(256)
anchor_score_finish_memo_kernel(uint32_t n, const int32_t* __restrict__ raw_score, const uint2* __restrict__ raw_sink, const uint64_t* __restrict__ text_begin,
                                const int32_t* __restrict__ min_score, int32_t worst_score, const uint8_t* __restrict__ from_memo,
                                const uint32_t* __restrict__ hit_read_id, const uint32_t* __restrict__ memo, int32_t* __restrict__ hit_score, uint32_t* __restrict__ hit_sink)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    int32_t s; uint2 k;
    anchor_raw(i, from_memo, hit_read_id, memo, raw_score, raw_sink, s, k);
    hit_score[i] = s >= min_score[i] ? s : worst_score;
    hit_sink[i] = uint32_t(text_begin[i]) + k.x;
}
__global__ void __launch_bounds__(256)
anchor_memo_update_kernel(uint32_t n_active, const uint32_t* __restrict__ active, const uint64_t* __restrict__ hit_begin, const uint32_t* __restrict__ hit_read_id,
                          const uint32_t* __restrict__ hit_seed, const uint64_t* __restrict__ text_begin, const uint32_t* __restrict__ text_len_setup,
                          const uint8_t* __restrict__ from_memo, const int32_t* __restrict__ raw_score, const uint2* __restrict__ raw_sink, uint32_t anchor,
                          uint32_t* __restrict__ memo)
{
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= n_active) return;
    uint32_t* m = memo + uint64_t(active[t] & 0x7FFFFFFFu) * 6u;
    for (uint64_t i = hit_begin[t + 1]; i > hit_begin[t]; --i)            // the last hit of the round that had a window
    {
        const uint32_t h = uint32_t(i - 1u);
        if (text_len_setup[h] == 0u) continue;
        int32_t s; uint2 k;
        anchor_raw(h, from_memo, hit_read_id, memo, raw_score, raw_sink, s, k);      // (reads the entry before it is overwritten below: one lane per pair)
        m[0] = uint32_t(text_begin[h]); m[1] = 1u | (((hit_seed[h] >> 13) & 1u) << 1) | (anchor << 2); m[2] = uint32_t(s); m[3] = k.x; m[4] = k.y;
        break;
    }
}

// the indices i < n with flags[i] == value, listed in idx (block by block: the order inside the list is not the index order), counted in *count
__global__ void __launch_bounds__(256)
list_flagged_kernel(uint32_t n, const uint8_t* __restrict__ flags, uint32_t value, uint32_t* __restrict__ count, uint32_t* __restrict__ idx)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    __shared__ uint32_t wave_n[4], block_base;
    const bool hit = i < n && uint32_t(flags[i]) == value;
    const uint64_t m = __ballot(hit);
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    if (lane == 0u) wave_n[wv] = uint32_t(__popcll(m));
    __syncthreads();
    if (threadIdx.x == 0u) { const uint32_t tot = wave_n[0] + wave_n[1] + wave_n[2] + wave_n[3]; block_base = tot ? atomicAdd(count, tot) : 0u; }
    __syncthreads();
    uint32_t base = block_base;
    for (uint32_t k = 0; k < wv; ++k) base += wave_n[k];
    if (hit) idx[base + uint32_t(__popcll(m & ((1ull << lane) - 1ull)))] = i;
}

// ------------------------------------------------------------------ opposite-mate memo
// The opposite mate's DP is a pure function of (pair, opposite strand, window, threshold).  The reference re-runs it for every
// anchor hit that lands on a placement it has already tried -- in the second anchor pass that is every seed of every read, because
// its skip test compares the hit with the recorded *window* begin -- and absorbs the identical result.  One entry
// per pair remembers the last scored job {window begin, end, threshold, strand, which mate} and its outputs; a later hit with the
// same job is answered from it.  Written by one lane per pair (opposite_memo_update), read in the next round (opposite_memo_lookup).
__global__ void __launch_bounds__(256)
opposite_memo_lookup_kernel(uint32_t n_hits, const uint32_t* __restrict__ hit_read_id, uint8_t* __restrict__ valid, const uint8_t* __restrict__ read_rc,
                            const uint32_t* __restrict__ gb, const uint32_t* __restrict__ ge, const int32_t* __restrict__ min_score, uint32_t anchor,
                            const uint32_t* __restrict__ memo, int32_t worst_score,
                            int32_t* __restrict__ o_score, int32_t* __restrict__ o_score2, uint32_t* __restrict__ o_loc, uint32_t* __restrict__ o_sink, uint32_t* __restrict__ o_sink2,
                            uint32_t* __restrict__ text_len)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n_hits || valid[i] != 1u) return;
    const uint32_t* m = memo + uint64_t(hit_read_id[i]) * 6u;
    if (m[3] != (1u | (uint32_t(read_rc[i]) << 1) | (anchor << 2)) || m[0] != gb[i] || m[1] != ge[i] || int32_t(m[2]) != min_score[i]) return;
    valid[i] = 2u;
    o_score[i] = int32_t(m[4]); o_score2[i] = worst_score; o_loc[i] = gb[i]; o_sink[i] = m[5]; o_sink2[i] = gb[i];
    if (text_len) text_len[i] = 0u;
}

__global__ void __launch_bounds__(256)
opposite_memo_update_kernel(uint32_t n_active, const uint32_t* __restrict__ active, const uint64_t* __restrict__ hit_begin, const uint8_t* __restrict__ valid,
                            const uint8_t* __restrict__ read_rc, const uint32_t* __restrict__ gb, const uint32_t* __restrict__ ge, const int32_t* __restrict__ min_score,
                            const int32_t* __restrict__ o_score, const uint32_t* __restrict__ o_sink, uint32_t anchor, uint32_t* __restrict__ memo)
{
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= n_active) return;
    uint32_t* m = memo + uint64_t(active[t] & 0x7FFFFFFFu) * 6u;
    for (uint64_t i = hit_begin[t + 1]; i > hit_begin[t]; --i)            // the last scored hit of the round
    {
        const uint64_t h = i - 1u;
        if (valid[h] != 1u) continue;
        m[0] = gb[h]; m[1] = ge[h]; m[2] = uint32_t(min_score[h]); m[3] = 1u | (uint32_t(read_rc[h]) << 1) | (anchor << 2); m[4] = uint32_t(o_score[h]); m[5] = o_sink[h];
        break;
    }
}

// ------------------------------------------------------------------ opposite-mate windows
// BestOppositeScoreStream::init_context (score_opposite_inl.h:92-200), compute_target_score (alignment_utils.h:100-111),
// frame_opposite_mate (:61-98), max_text_gaps for the Gotoh aligner (nvbio/alignment/utils_inl.h:181-204)
struct OppositeParams {
    uint32_t n_hits; const uint32_t* hit_read_id; const uint8_t* hit_rc; const uint32_t* hit_loc; const int32_t* hit_score;
    const uint32_t* a_read_len; const uint32_t* o_read_len; uint32_t a_fixed_len, o_fixed_len;
    const uint2* best; const uint2* best_o; uint32_t best_stride;
    int32_t match; const int32_t* min_score_by_len; int32_t text_gap_open, text_gap_ext;
    int32_t pe_policy, min_frag_len, max_frag_len, pe_overlap, score_limit; uint32_t anchor, genome_length;
    uint8_t* out_valid; int32_t* out_min_score; uint8_t* out_read_rc; uint32_t* out_genome_begin; uint32_t* out_genome_end;
    // the best-approx loop's form: strands from packed seeds, and only hits whose anchor score is not gate_worst are in the
    // opposite queue (aligner_best_approx_paired.h:632-640)
    const uint32_t* hit_seed; int32_t use_gate, gate_worst;
    // optional job description of every hit for the full-matrix scorer (an invalid hit gets an empty text: nothing to score):
    // the opposite mate's pattern (forward copy at o_read_begin[r] | r * o_fixed_len, reverse complement o_rc_offset further)
    const uint64_t* o_read_begin; uint64_t o_rc_offset; uint64_t* out_pat_begin; uint64_t* out_text_begin; uint32_t* out_text_len;
};

__global__ void __launch_bounds__(256) opposite_windows_kernel(const OppositeParams p)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= p.n_hits) return;
    const uint32_t read_rc = p.hit_seed ? ((p.hit_seed[i] >> 13) & 1u) : p.hit_rc[i], read_id = p.hit_read_id[i], g_pos = p.hit_loc[i];
    const uint32_t a_len = p.a_read_len ? p.a_read_len[read_id] : p.a_fixed_len, o_len = p.o_read_len ? p.o_read_len[read_id] : p.o_fixed_len;
    const int32_t a_optimal = int32_t(a_len) * p.match, a_worst = p.min_score_by_len[a_len];
    const int32_t o_optimal = int32_t(o_len) * p.match, o_worst = p.min_score_by_len[o_len];
    auto ld = [](const uint2* q, uint32_t k) { const uint2 v = q[k]; IoAln a = { v.x, v.y }; return a; };
    const IoBestPairs b = { ld(p.best, read_id), ld(p.best, read_id + p.best_stride), ld(p.best_o, read_id), ld(p.best_o, read_id + p.best_stride) };
    int32_t target;
    if (!io_aln_paired(b.a2)) target = a_worst + o_worst;
    else { const int32_t delta = bp_best_score(b) - bp_second_score(b); target = bp_second_score(b) + (delta * 3) / 4; }      // bowtie2's 'tighten = 3'
    const int32_t target_pair = min(target + 1, a_optimal + o_optimal);
    const int32_t target_mate = max(target_pair - p.hit_score[i], o_worst);
    const int32_t min_score = max(target_mate, p.score_limit);
    p.out_min_score[i] = min_score;
    uint32_t valid = 0, o_rc = 0, gb = 0, ge = 0;
    if (min_score <= o_optimal)
    {
        const bool anchor_fw = !read_rc, anchor_1 = (p.anchor == 0u);
        bool o_left, o_fw;
        switch (p.pe_policy) {
        case 0:  o_left = (anchor_1 != anchor_fw); o_fw = anchor_fw;  break;    // FF
        case 3:  o_left = (anchor_1 == anchor_fw); o_fw = anchor_fw;  break;    // RR
        case 1:  o_left = !anchor_fw;              o_fw = !anchor_fw; break;    // FR
        default: o_left = anchor_fw;               o_fw = !anchor_fw; break;    // RF
        }
        o_rc = o_fw ? 0u : 1u;
        int32_t max_ref_gaps;
        {
            int32_t score = int32_t(o_len) * p.match;
            if (score < min_score) max_ref_gaps = 0;
            else {
                score += p.text_gap_open;
                uint32_t n = 0;
                while (score >= min_score && n < o_len) { score += p.text_gap_ext; ++n; }
                max_ref_gaps = int32_t(n - 1u);             // (n == 0 wraps, as in the reference)
            }
        }
        const uint32_t o_gapped_len = o_len + uint32_t(max_ref_gaps);
        const uint32_t min_frag = uint32_t(p.min_frag_len), max_frag = uint32_t(p.max_frag_len);
        if (o_left) {
            const uint32_t max_end = g_pos + a_len + o_gapped_len > min_frag ? g_pos + a_len + o_gapped_len - min_frag : 0u;
            gb = g_pos + a_len > max_frag ? (g_pos + a_len) - max_frag : 0u;
            ge = p.pe_overlap ? g_pos + a_len : g_pos;
            ge = min(ge, max_end);
        } else {
            const uint32_t min_begin = g_pos + min_frag > o_gapped_len ? g_pos + min_frag - o_gapped_len : 0u;
            ge = g_pos + max_frag;
            gb = p.pe_overlap ? g_pos : g_pos + a_len;
            gb = max(gb, min_begin);
        }
        ge = min(ge, p.genome_length);
        if (gb < p.genome_length)
        {
            const uint32_t mate = p.anchor ? 0u : 1u;
            const bool skip = (mate == io_aln_mate(b.a1) && o_rc == io_aln_rc(b.a1) && g_pos == b.a1.align) ||
                              (mate == io_aln_mate(b.o1) && o_rc == io_aln_rc(b.o1) && g_pos == b.o1.align) ||
                              (mate == io_aln_mate(b.a2) && o_rc == io_aln_rc(b.a2) && g_pos == b.a2.align) ||
                              (mate == io_aln_mate(b.o2) && o_rc == io_aln_rc(b.o2) && g_pos == b.o2.align) || (gb == ge);
            valid = skip ? 0u : 1u;
        }
    }
    if (p.use_gate && p.hit_score[i] == p.gate_worst) valid = 0u;
    p.out_valid[i] = uint8_t(valid); p.out_read_rc[i] = uint8_t(o_rc); p.out_genome_begin[i] = gb; p.out_genome_end[i] = ge;
    if (p.out_pat_begin) {
        p.out_pat_begin[i]  = (p.o_read_begin ? p.o_read_begin[read_id] : uint64_t(read_id) * p.o_fixed_len) + (o_rc ? p.o_rc_offset : 0ull);
        p.out_text_begin[i] = gb;
        p.out_text_len[i]   = (valid && ge > gb) ? ge - gb : 0u;
    }
}

// single-precision arithmetic without contraction, so the thresholds fall where the host code puts them
__device__ __forceinline__ float fmul(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float fadd(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ int clamp10(int v) { return v < 0 ? 0 : v > 10 ? 10 : v; }

__constant__ int8_t c_unpaired_one[11] = { 43, 42, 41, 36, 32, 27, 20, 11, 4, 1, 0 };
__constant__ int8_t c_unpaired_two_perfect[11] = { 2, 16, 23, 30, 31, 32, 34, 36, 38, 40, 42 };
__constant__ int8_t c_unpaired_two[11][11] = {
    {  2,  2,  2,  1,  1, 0, 0, 0, 0, 0, 0 }, { 20, 14,  7,  3,  2, 1, 0, 0, 0, 0, 0 }, { 20, 16, 10,  6,  3, 1, 0, 0, 0, 0, 0 },
    { 20, 17, 13,  9,  3, 1, 1, 0, 0, 0, 0 }, { 21, 19, 15,  9,  5, 2, 2, 0, 0, 0, 0 }, { 22, 21, 16, 11, 10, 5, 0, 0, 0, 0, 0 },
    { 23, 22, 19, 16, 11, 0, 0, 0, 0, 0, 0 }, { 24, 25, 21, 30,  0, 0, 0, 0, 0, 0, 0 }, { 30, 26, 29,  0,  0, 0, 0, 0, 0, 0, 0 },
    { 30, 27,  0,  0,  0, 0, 0, 0, 0, 0, 0 }, { 30,  0,  0,  0,  0, 0, 0, 0, 0, 0, 0 } };

__device__ uint32_t mapq_v3(int32_t best_score, bool has_second, int32_t second_score, float max_score, float min_score, bool is_paired = false)
{
    if (is_paired) return 44u;          // paired_one_perfect (mapq.h:96-102)
    const float norm_factor = __fdiv_rn(10.0f, fadd(max_score, -min_score));
    if (float(best_score) < min_score) return 0u;
    const int best = max(int(max_score) - best_score, 0);
    const int best_bin = clamp10(int(fadd(fmul(float(best), norm_factor), 0.5f)));
    if (has_second) {
        const int diff = best_score - second_score;
        const int diff_bin = clamp10(int(fadd(fmul(float(diff), norm_factor), 0.5f)));
        return (float(best) == max_score) ? uint32_t(c_unpaired_two_perfect[best_bin]) : uint32_t(c_unpaired_two[diff_bin][best_bin]);
    }
    return (float(best) == max_score) ? 44u : uint32_t(c_unpaired_one[best_bin]);
}
__device__ uint32_t mapq_v2(int32_t best_score, bool has_second, int32_t second_score, float max_score, float min_score, bool monotone)
{
    const float diff = fadd(max_score, -min_score), best = float(best_score);
    if (best < min_score) return 0u;
    const float best_over = fadd(best, -min_score);
    #define GE(x, f) ((x) >= fmul(diff, f))
    if (monotone) {
        if (!has_second) {
            if (GE(best_over, 0.8f)) return 42; if (GE(best_over, 0.7f)) return 40; if (GE(best_over, 0.6f)) return 24;
            if (GE(best_over, 0.5f)) return 23; if (GE(best_over, 0.4f)) return 8;  if (GE(best_over, 0.3f)) return 3;
            return 0;
        }
        const float best_diff = fabsf(fadd(fabsf(best), -fabsf(float(second_score))));
        if (GE(best_diff, 0.9f)) return (best_over == diff) ? 39 : 33;
        if (GE(best_diff, 0.8f)) return (best_over == diff) ? 38 : 27;
        if (GE(best_diff, 0.7f)) return (best_over == diff) ? 37 : 26;
        if (GE(best_diff, 0.6f)) return (best_over == diff) ? 36 : 22;
        if (GE(best_diff, 0.5f)) { if (best_over == diff) return 35; if (GE(best_over, 0.84f)) return 25; if (GE(best_over, 0.68f)) return 16; return 5; }
        if (GE(best_diff, 0.4f)) { if (best_over == diff) return 34; if (GE(best_over, 0.84f)) return 21; if (GE(best_over, 0.68f)) return 14; return 4; }
        if (GE(best_diff, 0.3f)) { if (best_over == diff) return 32; if (GE(best_over, 0.88f)) return 18; if (GE(best_over, 0.67f)) return 15; return 3; }
        if (GE(best_diff, 0.2f)) { if (best_over == diff) return 31; if (GE(best_over, 0.88f)) return 17; if (GE(best_over, 0.67f)) return 11; return 0; }
        if (GE(best_diff, 0.1f)) { if (best_over == diff) return 30; if (GE(best_over, 0.88f)) return 12; if (GE(best_over, 0.67f)) return 7;  return 0; }
        if (best_diff > 0) return GE(best_over, 0.67f) ? 6 : 2;
        return GE(best_over, 0.67f) ? 1 : 0;
    }
    if (!has_second) {
        if (GE(best_over, 0.8f)) return 44; if (GE(best_over, 0.7f)) return 42; if (GE(best_over, 0.6f)) return 41;
        if (GE(best_over, 0.5f)) return 36; if (GE(best_over, 0.4f)) return 28; if (GE(best_over, 0.3f)) return 24;
        return 22;
    }
    const float best_diff = fabsf(fadd(fabsf(best), -fabsf(float(second_score))));
    if (GE(best_diff, 0.9f)) return 40; if (GE(best_diff, 0.8f)) return 39; if (GE(best_diff, 0.7f)) return 38; if (GE(best_diff, 0.6f)) return 37;
    if (GE(best_diff, 0.5f)) { if (best_over == diff) return 35; return GE(best_over, 0.50f) ? 25 : 20; }
    if (GE(best_diff, 0.4f)) { if (best_over == diff) return 34; return GE(best_over, 0.50f) ? 21 : 19; }
    if (GE(best_diff, 0.3f)) { if (best_over == diff) return 33; return GE(best_over, 0.5f) ? 18 : 16; }
    if (GE(best_diff, 0.2f)) { if (best_over == diff) return 32; return GE(best_over, 0.5f) ? 17 : 12; }
    if (GE(best_diff, 0.1f)) { if (best_over == diff) return 31; return GE(best_over, 0.5f) ? 14 : 9; }
    if (best_diff > 0) return GE(best_over, 0.5f) ? 11 : 2;
    return GE(best_over, 0.5f) ? 1 : 0;
    #undef GE
}

__global__ void __launch_bounds__(256)
mapq_kernel(int32_t version, int32_t match, int32_t monotone, uint32_t n_reads, const uint2* __restrict__ best, uint32_t best_stride,
            const uint32_t* __restrict__ read_len, uint32_t fixed_len, const int32_t* __restrict__ min_score_by_len, uint8_t* __restrict__ out)
{
    const uint32_t r = blockIdx.x * 256u + threadIdx.x;
    if (r >= n_reads) return;
    const uint2 b1 = best[r], b2 = best[r + best_stride];
    const IoAln a1 = { b1.x, b1.y }, a2 = { b2.x, b2.y };
    const uint32_t len = read_len ? read_len[r] : fixed_len;
    const float max_score = float(int32_t(len) * match), min_score = float(min_score_by_len[len]);
    const bool has_second = a2.align != 0xFFFFFFFFu;
    out[r] = uint8_t(version == 3 ? mapq_v3(io_aln_score(a1), has_second, io_aln_score(a2), max_score, min_score)
                                  : mapq_v2(io_aln_score(a1), has_second, io_aln_score(a2), max_score, min_score, monotone != 0));
}

__global__ void __launch_bounds__(256)
mapq_paired_kernel(int32_t version, int32_t match, int32_t monotone, uint32_t n_reads, const uint2* __restrict__ best, const uint2* __restrict__ best_o,
                   uint32_t best_stride, const uint32_t* __restrict__ read_len, const uint32_t* __restrict__ o_read_len, uint32_t fixed_len, uint32_t o_fixed_len,
                   const int32_t* __restrict__ min_score_by_len, uint8_t* __restrict__ out)
{
    const uint32_t r = blockIdx.x * 256u + threadIdx.x;
    if (r >= n_reads) return;
    auto ld = [](const uint2* q, uint32_t i) { const uint2 v = q[i]; IoAln a = { v.x, v.y }; return a; };
    const IoBestPairs b = { ld(best, r), ld(best, r + best_stride), ld(best_o, r), ld(best_o, r + best_stride) };
    const bool paired = io_aln_paired(b.a1);
    const bool has_second = paired ? io_aln_paired(b.a2) : (b.a2.align != 0xFFFFFFFFu);
    const uint32_t len = read_len ? read_len[r] : fixed_len, olen = o_read_len ? o_read_len[r] : o_fixed_len;
    if (version == 3) {
        out[r] = uint8_t(mapq_v3(bp_best_score(b), has_second, bp_second_score(b), float(int32_t(len) * match), float(min_score_by_len[len]), paired));
    } else {
        const float max_score = fadd(float(int32_t(len) * match), paired ? float(int32_t(olen) * match) : 0.0f);
        const float min_score = fadd(float(min_score_by_len[len]), paired ? float(min_score_by_len[olen]) : 0.0f);
        out[r] = uint8_t(mapq_v2(bp_best_score(b), has_second, bp_second_score(b), max_score, min_score, monotone != 0));
    }
}

} // namespace nvb

using namespace nvb;

NVB_API uint64_t nvbio_hip_alignment_invalid(void)
{   // io::Alignment::invalid(): pos -1, ed 255, score 2^17-1 (alignments.h:127-128)
    return (uint64_t(0xFFFFFFFFu) << 32) | (uint64_t(0x1FFFFu) << 1) | (uint64_t(255u) << 18);
}

NVB_API int nvbio_hip_init_alignments(uint32_t n_reads, const uint32_t* read_len, uint32_t fixed_read_len,
                                      const int32_t* worst_score_by_len, uint32_t mate,
                                      uint64_t* best_alignments, uint32_t best_stride, void* stream)
{
    if (n_reads == 0) return hipSuccess;
    if (!worst_score_by_len || !best_alignments || best_stride < n_reads) return hipErrorInvalidValue;
    if (!read_len && fixed_read_len == 0) return hipErrorInvalidValue;
    g_last_kernel = "init_alignments_kernel";
    hipLaunchKernelGGL(init_alignments_kernel, dim3((n_reads + 255u) / 256u), dim3(256), 0, to_stream(stream), n_reads, read_len, fixed_read_len,
                       worst_score_by_len, mate, reinterpret_cast<uint2*>(best_alignments), best_stride);
    return hipGetLastError();
}

NVB_API int nvbio_hip_score_reduce(uint32_t n_active, const uint32_t* read_ids, const uint64_t* hit_begin,
                                   const int32_t* hit_score, const uint32_t* hit_loc, const uint8_t* hit_rc,
                                   const uint32_t* read_len, uint32_t fixed_read_len,
                                   uint64_t* best_alignments, uint32_t best_stride, void* stream)
{
    if (n_active == 0) return hipSuccess;
    if (!hit_begin || !hit_score || !hit_loc || !hit_rc || !best_alignments || best_stride == 0) return hipErrorInvalidValue;
    if (!read_len && fixed_read_len == 0) return hipErrorInvalidValue;
    g_last_kernel = "score_reduce_kernel";
    hipLaunchKernelGGL(score_reduce_kernel, dim3((n_active + 255u) / 256u), dim3(256), 0, to_stream(stream), n_active, read_ids, hit_begin,
                       hit_score, hit_loc, hit_rc, read_len, fixed_read_len, reinterpret_cast<uint2*>(best_alignments), best_stride);
    return hipGetLastError();
}

NVB_API int nvbio_hip_score_reduce_best_approx(uint32_t n_active, const uint32_t* active_reads, const uint64_t* hit_begin,
                                               const int32_t* hit_score, const uint32_t* hit_loc, const uint32_t* hit_seed,
                                               const uint32_t* read_len, uint32_t fixed_read_len,
                                               uint64_t* best_alignments, uint32_t best_stride, int32_t worst_score,
                                               uint32_t* trys, uint32_t* hit_counts,
                                               uint32_t n_ext, uint32_t min_ext, uint32_t max_ext, uint32_t max_effort, const int32_t* known_score,
                                               const uint32_t* hit_sink, uint32_t* best_sink, void* stream)
{
    if ((hit_sink != nullptr) != (best_sink != nullptr)) return hipErrorInvalidValue;
    if (n_active == 0) return hipSuccess;
    if (!active_reads || !hit_begin || !hit_score || !hit_loc || !hit_seed || !best_alignments || best_stride == 0 || !trys || !hit_counts)
        return hipErrorInvalidValue;
    if (!read_len && fixed_read_len == 0) return hipErrorInvalidValue;
    g_last_kernel = "score_reduce_best_approx_kernel";
    hipLaunchKernelGGL(score_reduce_best_approx_kernel, dim3((n_active + 255u) / 256u), dim3(256), 0, to_stream(stream), n_active, active_reads,
                       hit_begin, hit_score, hit_loc, hit_seed, read_len, fixed_read_len, reinterpret_cast<uint2*>(best_alignments), best_stride,
                       worst_score, trys, hit_counts, n_ext, min_ext, max_ext, max_effort, known_score, reinterpret_cast<const uint2*>(hit_sink),
                       reinterpret_cast<uint2*>(best_sink));
    return hipGetLastError();
}

NVB_API int nvbio_hip_traceback_best_known(uint32_t n, const uint32_t* idx, const uint64_t* best_alignments, const uint32_t* best_sink,
                                           int32_t* out_score, uint32_t* out_sink, void* stream)
{
    if (n == 0) return hipSuccess;
    if (!best_alignments || !out_score || (best_sink == nullptr) != (out_sink == nullptr)) return hipErrorInvalidValue;      // sinks: both or neither
    g_last_kernel = "traceback_best_known_kernel";
    hipLaunchKernelGGL(traceback_best_known_kernel, dim3((n + 255u) / 256u), dim3(256), 0, to_stream(stream), n, idx, reinterpret_cast<const uint2*>(best_alignments),
                       reinterpret_cast<const uint2*>(best_sink), out_score, reinterpret_cast<uint2*>(out_sink));
    return hipGetLastError();
}

NVB_API int nvbio_hip_mapq(int32_t version, int32_t match, int32_t monotone, const int32_t* min_score_by_len,
                           uint32_t n_reads, const uint64_t* best_alignments, uint32_t best_stride,
                           const uint32_t* read_len, uint32_t fixed_read_len, uint8_t* out_mapq, void* stream)
{
    if (version != 2 && version != 3) return hipErrorInvalidValue;
    if (n_reads == 0) return hipSuccess;
    if (!min_score_by_len || !best_alignments || !out_mapq || best_stride == 0) return hipErrorInvalidValue;
    if (!read_len && fixed_read_len == 0) return hipErrorInvalidValue;
    g_last_kernel = "mapq_kernel";
    hipLaunchKernelGGL(mapq_kernel, dim3((n_reads + 255u) / 256u), dim3(256), 0, to_stream(stream), version, match, monotone, n_reads,
                       reinterpret_cast<const uint2*>(best_alignments), best_stride, read_len, fixed_read_len, min_score_by_len, out_mapq);
    return hipGetLastError();
}

NVB_API int nvbio_hip_score_reduce_paired(uint32_t n_active, const uint32_t* read_ids, const uint64_t* hit_begin,
    const uint32_t* hit_loc, const uint32_t* hit_sink, const int32_t* hit_score, const uint8_t* hit_rc,
    const uint32_t* opposite_loc, const uint32_t* opposite_sink, const uint32_t* opposite_sink2,
    const int32_t* opposite_score, const int32_t* opposite_score2,
    const uint32_t* read_len, uint32_t fixed_read_len, uint32_t anchor, int32_t pe_policy, int32_t pe_unpaired, int32_t score_limit,
    uint64_t* best_alignments, uint64_t* best_alignments_o, uint32_t best_stride, void* stream)
{
    if (n_active == 0) return hipSuccess;
    if (!hit_begin || !hit_loc || !hit_sink || !hit_score || !hit_rc || !opposite_loc || !opposite_sink || !opposite_sink2 || !opposite_score || !opposite_score2 ||
        !best_alignments || !best_alignments_o || best_stride == 0 || anchor > 1u || pe_policy < 0 || pe_policy > 3) return hipErrorInvalidValue;
    if (!read_len && fixed_read_len == 0) return hipErrorInvalidValue;
    PairedReduceParams p = { n_active, read_ids, hit_begin, hit_loc, hit_sink, hit_score, hit_rc, opposite_loc, opposite_sink, opposite_sink2, opposite_score, opposite_score2,
                             read_len, fixed_read_len, anchor, pe_policy, pe_unpaired, score_limit,
                             reinterpret_cast<uint2*>(best_alignments), reinterpret_cast<uint2*>(best_alignments_o), best_stride };
    g_last_kernel = "score_reduce_paired_kernel";
    hipLaunchKernelGGL(score_reduce_paired_kernel<false>, dim3((n_active + 255u) / 256u), dim3(256), 0, to_stream(stream), p);
    return hipGetLastError();
}

NVB_API int nvbio_hip_mapq_paired(int32_t version, int32_t match, int32_t monotone, const int32_t* min_score_by_len,
                                  uint32_t n_reads, const uint64_t* best_alignments, const uint64_t* best_alignments_o, uint32_t best_stride,
                                  const uint32_t* read_len, const uint32_t* o_read_len, uint32_t fixed_read_len, uint32_t o_fixed_read_len,
                                  uint8_t* out_mapq, void* stream)
{
    if (version != 2 && version != 3) return hipErrorInvalidValue;
    if (n_reads == 0) return hipSuccess;
    if (!min_score_by_len || !best_alignments || !best_alignments_o || !out_mapq || best_stride == 0) return hipErrorInvalidValue;
    if ((!read_len && fixed_read_len == 0) || (!o_read_len && o_fixed_read_len == 0)) return hipErrorInvalidValue;
    g_last_kernel = "mapq_paired_kernel";
    hipLaunchKernelGGL(mapq_paired_kernel, dim3((n_reads + 255u) / 256u), dim3(256), 0, to_stream(stream), version, match, monotone, n_reads,
                       reinterpret_cast<const uint2*>(best_alignments), reinterpret_cast<const uint2*>(best_alignments_o), best_stride,
                       read_len, o_read_len, fixed_read_len, o_fixed_read_len, min_score_by_len, out_mapq);
    return hipGetLastError();
}

NVB_API int nvbio_hip_opposite_mate_windows(uint32_t n_hits, const uint32_t* hit_read_id, const uint8_t* hit_rc, const uint32_t* hit_loc, const int32_t* hit_score,
    const uint32_t* a_read_len, const uint32_t* o_read_len, uint32_t a_fixed_len, uint32_t o_fixed_len,
    const uint64_t* best_alignments, const uint64_t* best_alignments_o, uint32_t best_stride,
    int32_t match, const int32_t* min_score_by_len, int32_t text_gap_open, int32_t text_gap_ext, const nvbio_hip_pe_params* params,
    uint8_t* out_valid, int32_t* out_min_score, uint8_t* out_read_rc, uint32_t* out_genome_begin, uint32_t* out_genome_end, void* stream)
{
    if (n_hits == 0) return hipSuccess;
    if (!hit_read_id || !hit_rc || !hit_loc || !hit_score || !best_alignments || !best_alignments_o || best_stride == 0 || !min_score_by_len || !params ||
        !out_valid || !out_min_score || !out_read_rc || !out_genome_begin || !out_genome_end) return hipErrorInvalidValue;
    if ((!a_read_len && a_fixed_len == 0) || (!o_read_len && o_fixed_len == 0) || params->anchor > 1u || params->pe_policy < 0 || params->pe_policy > 3) return hipErrorInvalidValue;
    OppositeParams p = { n_hits, hit_read_id, hit_rc, hit_loc, hit_score, a_read_len, o_read_len, a_fixed_len, o_fixed_len,
                         reinterpret_cast<const uint2*>(best_alignments), reinterpret_cast<const uint2*>(best_alignments_o), best_stride,
                         match, min_score_by_len, text_gap_open, text_gap_ext,
                         params->pe_policy, params->min_frag_len, params->max_frag_len, params->pe_overlap, params->score_limit, params->anchor, params->genome_length,
                         out_valid, out_min_score, out_read_rc, out_genome_begin, out_genome_end };
    g_last_kernel = "opposite_windows_kernel";
    hipLaunchKernelGGL(opposite_windows_kernel, dim3((n_hits + 255u) / 256u), dim3(256), 0, to_stream(stream), p);
    return hipGetLastError();
}

NVB_API int nvbio_hip_anchor_score_setup(uint32_t n_hits, const uint32_t* hit_read_id, const uint32_t* hit_loc, const uint32_t* hit_seed,
    const uint64_t* a_read_begin, const uint32_t* a_read_len, const uint32_t* o_read_len, uint32_t a_fixed_len, uint32_t o_fixed_len, uint64_t rc_offset,
    uint32_t band_len, uint32_t genome_length, const uint64_t* best_alignments, const uint64_t* best_alignments_o, uint32_t best_stride,
    int32_t match, const int32_t* min_score_by_len, int32_t score_limit, uint32_t anchor,
    uint64_t* pattern_begin, uint32_t* pattern_len, uint64_t* text_begin, uint32_t* text_len, int32_t* min_score, void* stream)
{
    if (n_hits == 0) return hipSuccess;
    if (!hit_read_id || !hit_loc || !hit_seed || !best_alignments || !best_alignments_o || best_stride == 0 || !min_score_by_len || anchor > 1u ||
        !pattern_begin || !text_begin || !text_len || !min_score) return hipErrorInvalidValue;
    if ((!a_read_len && a_fixed_len == 0) || (!o_read_len && o_fixed_len == 0) || (a_read_len && !pattern_len)) return hipErrorInvalidValue;
    AnchorSetupParams p = { n_hits, hit_read_id, hit_loc, hit_seed, a_read_begin, a_read_len, o_read_len, a_fixed_len, o_fixed_len, rc_offset,
                            band_len, genome_length, reinterpret_cast<const uint2*>(best_alignments), reinterpret_cast<const uint2*>(best_alignments_o), best_stride,
                            match, min_score_by_len, score_limit, anchor, pattern_begin, pattern_len, text_begin, text_len, min_score };
    g_last_kernel = "anchor_score_setup_kernel";
    hipLaunchKernelGGL(anchor_score_setup_kernel, dim3((n_hits + 255u) / 256u), dim3(256), 0, to_stream(stream), p);
    return hipGetLastError();
}

NVB_API int nvbio_hip_anchor_score_finish(uint32_t n_hits, const int32_t* raw_score, const uint32_t* raw_sink, const uint64_t* text_begin,
    const int32_t* min_score, int32_t worst_score, int32_t* hit_score, uint32_t* hit_sink, void* stream)
{
    if (n_hits == 0) return hipSuccess;
    if (!raw_score || !raw_sink || !text_begin || !min_score || !hit_score || !hit_sink) return hipErrorInvalidValue;
    g_last_kernel = "anchor_score_finish_kernel";
    hipLaunchKernelGGL(anchor_score_finish_kernel, dim3((n_hits + 255u) / 256u), dim3(256), 0, to_stream(stream), n_hits, raw_score,
                       reinterpret_cast<const uint2*>(raw_sink), text_begin, min_score, worst_score, hit_score, hit_sink);
    return hipGetLastError();
}

NVB_API int nvbio_hip_anchor_memo_mark(uint32_t n_hits, const uint32_t* hit_read_id, const uint32_t* hit_seed, const uint64_t* text_begin, const uint32_t* text_len,
    uint32_t anchor, const uint32_t* memo, uint8_t* from_memo, uint32_t* text_len_out, uint32_t* live_count, uint32_t* live_idx, void* stream)
{
    if ((live_count != nullptr) != (live_idx != nullptr)) return hipErrorInvalidValue;
    if (live_count) { const hipError_t e = hipMemsetAsync(live_count, 0, sizeof(uint32_t), to_stream(stream)); if (e != hipSuccess) return e; }
    if (n_hits == 0) return hipSuccess;
    if (!hit_read_id || !hit_seed || !text_begin || !text_len || anchor > 1u || !memo || !from_memo || !text_len_out || text_len_out == text_len) return hipErrorInvalidValue;
    g_last_kernel = "anchor_memo_mark_kernel";
    hipLaunchKernelGGL(anchor_memo_mark_kernel, dim3((n_hits + 255u) / 256u), dim3(256), 0, to_stream(stream), n_hits, hit_read_id, hit_seed, text_begin, text_len, anchor, memo,
                       from_memo, text_len_out, live_count, live_idx);
    return hipGetLastError();
}
NVB_API int nvbio_hip_anchor_score_finish_memo(uint32_t n_hits, const int32_t* raw_score, const uint32_t* raw_sink, const uint64_t* text_begin, const int32_t* min_score,
    int32_t worst_score, const uint8_t* from_memo, const uint32_t* hit_read_id, const uint32_t* memo, int32_t* hit_score, uint32_t* hit_sink, void* stream)
{
    if (n_hits == 0) return hipSuccess;
    if (!raw_score || !raw_sink || !text_begin || !min_score || !from_memo || !hit_read_id || !memo || !hit_score || !hit_sink) return hipErrorInvalidValue;
    g_last_kernel = "anchor_score_finish_memo_kernel";
    hipLaunchKernelGGL(anchor_score_finish_memo_kernel, dim3((n_hits + 255u) / 256u), dim3(256), 0, to_stream(stream), n_hits, raw_score, reinterpret_cast<const uint2*>(raw_sink),
                       text_begin, min_score, worst_score, from_memo, hit_read_id, memo, hit_score, hit_sink);
    return hipGetLastError();
}
NVB_API int nvbio_hip_anchor_memo_update(uint32_t n_active, const uint32_t* active_reads, const uint64_t* hit_begin, const uint32_t* hit_read_id, const uint32_t* hit_seed,
    const uint64_t* text_begin, const uint32_t* text_len_setup, const uint8_t* from_memo, const int32_t* raw_score, const uint32_t* raw_sink, uint32_t anchor, uint32_t* memo, void* stream)
{
    if (n_active == 0) return hipSuccess;
    if (!active_reads || !hit_begin || !hit_read_id || !hit_seed || !text_begin || !text_len_setup || !from_memo || !raw_score || !raw_sink || anchor > 1u || !memo) return hipErrorInvalidValue;
    g_last_kernel = "anchor_memo_update_kernel";
    hipLaunchKernelGGL(anchor_memo_update_kernel, dim3((n_active + 255u) / 256u), dim3(256), 0, to_stream(stream), n_active, active_reads, hit_begin, hit_read_id, hit_seed, text_begin,
                       text_len_setup, from_memo, raw_score, reinterpret_cast<const uint2*>(raw_sink), anchor, memo);
    return hipGetLastError();
}

NVB_API int nvbio_hip_opposite_score_setup(uint32_t n_hits, const uint32_t* hit_read_id, const uint32_t* hit_seed, const uint32_t* hit_loc, const int32_t* hit_score,
    int32_t worst_score, const uint32_t* a_read_len, const uint32_t* o_read_len, uint32_t a_fixed_len, uint32_t o_fixed_len,
    const uint64_t* best_alignments, const uint64_t* best_alignments_o, uint32_t best_stride,
    int32_t match, const int32_t* min_score_by_len, int32_t text_gap_open, int32_t text_gap_ext, const nvbio_hip_pe_params* params,
    uint8_t* out_valid, int32_t* out_min_score, uint8_t* out_read_rc, uint32_t* out_genome_begin, uint32_t* out_genome_end,
    const uint64_t* o_read_begin, uint64_t o_rc_offset, uint64_t* out_pattern_begin, uint64_t* out_text_begin, uint32_t* out_text_len, void* stream)
{
    if (n_hits == 0) return hipSuccess;
    if (out_pattern_begin && (!out_text_begin || !out_text_len)) return hipErrorInvalidValue;
    if (!hit_read_id || !hit_seed || !hit_loc || !hit_score || !best_alignments || !best_alignments_o || best_stride == 0 || !min_score_by_len || !params ||
        !out_valid || !out_min_score || !out_read_rc || !out_genome_begin || !out_genome_end) return hipErrorInvalidValue;
    if ((!a_read_len && a_fixed_len == 0) || (!o_read_len && o_fixed_len == 0) || params->anchor > 1u || params->pe_policy < 0 || params->pe_policy > 3) return hipErrorInvalidValue;
    OppositeParams p = { n_hits, hit_read_id, nullptr, hit_loc, hit_score, a_read_len, o_read_len, a_fixed_len, o_fixed_len,
                         reinterpret_cast<const uint2*>(best_alignments), reinterpret_cast<const uint2*>(best_alignments_o), best_stride,
                         match, min_score_by_len, text_gap_open, text_gap_ext,
                         params->pe_policy, params->min_frag_len, params->max_frag_len, params->pe_overlap, params->score_limit, params->anchor, params->genome_length,
                         out_valid, out_min_score, out_read_rc, out_genome_begin, out_genome_end, hit_seed, 1, worst_score,
                         o_read_begin, o_rc_offset, out_pattern_begin, out_text_begin, out_text_len };
    g_last_kernel = "opposite_windows_kernel";
    hipLaunchKernelGGL(opposite_windows_kernel, dim3((n_hits + 255u) / 256u), dim3(256), 0, to_stream(stream), p);
    return hipGetLastError();
}

NVB_API int nvbio_hip_opposite_score_finish(uint32_t n_valid, const uint32_t* valid_idx, const uint8_t* valid_flags, const int32_t* raw_score, const uint32_t* raw_sink,
    const int32_t* min_score, const uint32_t* genome_begin, int32_t worst_score,
    int32_t* opposite_score, int32_t* opposite_score2, uint32_t* opposite_loc, uint32_t* opposite_sink, uint32_t* opposite_sink2, void* stream)
{
    if (n_valid == 0) return hipSuccess;
    if ((!valid_idx && !valid_flags) || !raw_score || !raw_sink || !min_score || !genome_begin || !opposite_score || !opposite_score2 || !opposite_loc || !opposite_sink || !opposite_sink2)
        return hipErrorInvalidValue;
    g_last_kernel = "opposite_score_finish_kernel";
    hipLaunchKernelGGL(opposite_score_finish_kernel, dim3((n_valid + 255u) / 256u), dim3(256), 0, to_stream(stream), n_valid, valid_idx, raw_score,
                       reinterpret_cast<const uint2*>(raw_sink), min_score, genome_begin, worst_score, valid_idx ? nullptr : valid_flags, opposite_score, opposite_score2, opposite_loc, opposite_sink, opposite_sink2);
    return hipGetLastError();
}

NVB_API int nvbio_hip_list_flagged(uint32_t n, const uint8_t* flags, uint32_t value, uint32_t* count, uint32_t* idx, void* stream)
{
    if (!count || !idx) return hipErrorInvalidValue;
    if (hipError_t e = hipMemsetAsync(count, 0, sizeof(uint32_t), to_stream(stream))) return e;
    if (n == 0) return hipSuccess;
    if (!flags) return hipErrorInvalidValue;
    g_last_kernel = "list_flagged_kernel";
    hipLaunchKernelGGL(list_flagged_kernel, dim3((n + 255u) / 256u), dim3(256), 0, to_stream(stream), n, flags, value, count, idx);
    return hipGetLastError();
}

NVB_API int nvbio_hip_opposite_memo_lookup(uint32_t n_hits, const uint32_t* hit_read_id, uint8_t* valid, const uint8_t* read_rc, const uint32_t* genome_begin,
    const uint32_t* genome_end, const int32_t* min_score, uint32_t anchor, const uint32_t* memo, int32_t worst_score,
    int32_t* opposite_score, int32_t* opposite_score2, uint32_t* opposite_loc, uint32_t* opposite_sink, uint32_t* opposite_sink2, uint32_t* text_len, void* stream)
{
    if (n_hits == 0) return hipSuccess;
    if (!hit_read_id || !valid || !read_rc || !genome_begin || !genome_end || !min_score || anchor > 1u || !memo || !opposite_score || !opposite_score2 ||
        !opposite_loc || !opposite_sink || !opposite_sink2) return hipErrorInvalidValue;
    g_last_kernel = "opposite_memo_lookup_kernel";
    hipLaunchKernelGGL(opposite_memo_lookup_kernel, dim3((n_hits + 255u) / 256u), dim3(256), 0, to_stream(stream), n_hits, hit_read_id, valid, read_rc, genome_begin,
                       genome_end, min_score, anchor, memo, worst_score, opposite_score, opposite_score2, opposite_loc, opposite_sink, opposite_sink2, text_len);
    return hipGetLastError();
}

NVB_API int nvbio_hip_opposite_memo_update(uint32_t n_active, const uint32_t* active_reads, const uint64_t* hit_begin, const uint8_t* valid, const uint8_t* read_rc,
    const uint32_t* genome_begin, const uint32_t* genome_end, const int32_t* min_score, const int32_t* opposite_score, const uint32_t* opposite_sink,
    uint32_t anchor, uint32_t* memo, void* stream)
{
    if (n_active == 0) return hipSuccess;
    if (!active_reads || !hit_begin || !valid || !read_rc || !genome_begin || !genome_end || !min_score || !opposite_score || !opposite_sink || anchor > 1u || !memo)
        return hipErrorInvalidValue;
    g_last_kernel = "opposite_memo_update_kernel";
    hipLaunchKernelGGL(opposite_memo_update_kernel, dim3((n_active + 255u) / 256u), dim3(256), 0, to_stream(stream), n_active, active_reads, hit_begin, valid, read_rc,
                       genome_begin, genome_end, min_score, opposite_score, opposite_sink, anchor, memo);
    return hipGetLastError();
}

NVB_API int nvbio_hip_score_reduce_paired_best_approx(uint32_t n_active, const uint32_t* active_reads, const uint64_t* hit_begin,
    const uint32_t* hit_loc, const uint32_t* hit_sink, const int32_t* hit_score, const uint32_t* hit_seed,
    const uint32_t* opposite_loc, const uint32_t* opposite_sink, const uint32_t* opposite_sink2, const int32_t* opposite_score, const int32_t* opposite_score2,
    const uint32_t* read_len, uint32_t fixed_read_len, uint32_t anchor, int32_t pe_policy, int32_t pe_unpaired, int32_t score_limit,
    uint64_t* best_alignments, uint64_t* best_alignments_o, uint32_t best_stride,
    uint32_t* trys, uint32_t* hit_counts, uint32_t n_ext, uint32_t min_ext, uint32_t max_ext, uint32_t max_effort, void* stream)
{
    if (n_active == 0) return hipSuccess;
    if (!active_reads || !hit_begin || !hit_loc || !hit_sink || !hit_score || !hit_seed || !opposite_loc || !opposite_sink || !opposite_sink2 || !opposite_score ||
        !opposite_score2 || !best_alignments || !best_alignments_o || best_stride == 0 || anchor > 1u || pe_policy < 0 || pe_policy > 3 || !trys || !hit_counts)
        return hipErrorInvalidValue;
    if (!read_len && fixed_read_len == 0) return hipErrorInvalidValue;
    PairedReduceParams p = { n_active, active_reads, hit_begin, hit_loc, hit_sink, hit_score, nullptr, opposite_loc, opposite_sink, opposite_sink2, opposite_score, opposite_score2,
                             read_len, fixed_read_len, anchor, pe_policy, pe_unpaired, score_limit,
                             reinterpret_cast<uint2*>(best_alignments), reinterpret_cast<uint2*>(best_alignments_o), best_stride,
                             hit_seed, trys, hit_counts, n_ext, min_ext, max_ext, max_effort };
    g_last_kernel = "score_reduce_paired_kernel<ctx>";
    hipLaunchKernelGGL(score_reduce_paired_kernel<true>, dim3((n_active + 255u) / 256u), dim3(256), 0, to_stream(stream), p);
    return hipGetLastError();
}

NVB_API int nvbio_hip_mark_discordant(uint32_t n_reads, uint64_t* best_alignments, uint64_t* best_alignments_o, uint32_t best_stride, void* stream)
{
    if (n_reads == 0) return hipSuccess;
    if (!best_alignments || !best_alignments_o || best_stride == 0) return hipErrorInvalidValue;
    g_last_kernel = "mark_discordant_kernel";
    hipLaunchKernelGGL(mark_discordant_kernel, dim3((n_reads + 255u) / 256u), dim3(256), 0, to_stream(stream), n_reads,
                       reinterpret_cast<uint2*>(best_alignments), reinterpret_cast<uint2*>(best_alignments_o), best_stride);
    return hipGetLastError();
}
