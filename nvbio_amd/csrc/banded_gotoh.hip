// banded_gotoh.hip -- batched banded Gotoh / Smith-Waterman score for gfx950.
//
// Computes, per job, exactly what the reference's
//   priv::banded::gotoh_alignment_score_dispatch<BAND_LEN,TYPE>::run
//   (nvbio/alignment/gotoh/gotoh_banded_inl.h:415-658)
// reports into a fresh BestSink<int32> (nvbio/alignment/sink_inl.h:38-68), for
// GotohAligner<TYPE,SimpleGotohScheme> with trivial qualities, scheduled as
//   BatchedBandedAlignmentScore<BAND_LEN,stream,DeviceThreadBlockScheduler<128,1>>
//   (nvbio/alignment/batched_banded_inl.h:135-162).
//
// This is NOT a translation of that code.  The arithmetic is re-derived for CDNA4:
//
//  * one lane = one alignment (integer DP with a serial E chain per row: there
//    is no contraction for MFMA to do); H+G_o ("HG") is the only H form kept in
//    registers: it feeds E of the same row, F of the next row, and the diagonal
//    (with the substitution scores pre-biased by -G_o), which removes one add
//    per cell relative to the textbook recurrence;
//  * rows are unrolled in blocks of 16 so that (a) the text window lives in a
//    statically indexed 16-slot register ring (slot = text index mod 16) -- no
//    per-cell register shifting as in the reference's text_cache -- and (b) the
//    16 pattern symbols / 16 new text symbols of a block are pulled with 2-3
//    dword loads + one v_alignbit each and normalised in-register, instead of a
//    PackedStream access per symbol;  the next block's words are prefetched
//    while the current block computes;
//  * LOCAL: all scores are carried scaled by 32, so a cell's sink key
//    (score*32 + j) is ONE add; a row keeps the max key and folds it into the
//    running best once per row with a >= test, which reproduces BestSink's
//    "last maximum wins" tie-break (i-major, j-minor) without its compare + two
//    selects per cell;
//  * 16-bit arithmetic where it is provably exact: on gfx950 v_add_u16 /
//    v_max_i16 issue in 2 cycles per wave64 while v_max_i32 / v_max3_i32 /
//    v_lshl_or_b32 take 4 (tools/valu_probe.hip, profiles/r01/valu_probe.txt).
//    The host computes, from the scheme, band and type alone, the longest
//    pattern for which no DP value can leave the int16 range (and no value can
//    come near the sentinel that stands in for the reference's infimum); jobs up
//    to that length run the 16-bit kernel, longer ones the 32-bit kernel.
//    Both produce the reference's int32 results bit for bit.
//
#include "banded_gotoh_bounded.h"
#include <mutex>
#include <unordered_map>
#include <atomic>
#include <vector>
#include <algorithm>
#include <cstring>
#include <cstdlib>

namespace nvb {

thread_local const char* g_last_kernel = "";

#define NVB_DECL(B) \
    extern template hipError_t launch_band_width<B, NoQual>(const GotohParams&, const NoQual&, int, bool, hipStream_t); \
    extern template hipError_t launch_band_width<B, QualArgs>(const GotohParams&, const QualArgs&, int, bool, hipStream_t);
NVB_DECL(3) NVB_DECL(5) NVB_DECL(7) NVB_DECL(15) NVB_DECL(31)
#undef NVB_DECL
#define NVB_DECL(B) extern template hipError_t launch_band_width_asym<B>(const GotohParams&, int, bool, hipStream_t);
NVB_DECL(3) NVB_DECL(5) NVB_DECL(7) NVB_DECL(15) NVB_DECL(31)
#undef NVB_DECL
#define NVB_DECL(B) \
    extern template hipError_t launch_band_width_bounded<B, NoQual>(const GotohParams&, const NoQual&, const BoundArgs&, int, bool, hipStream_t); \
    extern template hipError_t launch_band_width_bounded<B, QualArgs>(const GotohParams&, const QualArgs&, const BoundArgs&, int, bool, hipStream_t);
NVB_DECL(3) NVB_DECL(5) NVB_DECL(7) NVB_DECL(15) NVB_DECL(31)
#undef NVB_DECL

template <typename QA>
static hipError_t launch_bounded(const GotohParams& p, const QA& qa, const BoundArgs& ba, int type, uint32_t band, bool width16, hipStream_t s)
{
    switch (band) {
    case 3:  return launch_band_width_bounded<3, QA>(p, qa, ba, type, width16, s);
    case 5:  return launch_band_width_bounded<5, QA>(p, qa, ba, type, width16, s);
    case 7:  return launch_band_width_bounded<7, QA>(p, qa, ba, type, width16, s);
    case 15: return launch_band_width_bounded<15, QA>(p, qa, ba, type, width16, s);
    case 31: return launch_band_width_bounded<31, QA>(p, qa, ba, type, width16, s);
    default: return hipErrorNotSupported;
    }
}

static hipError_t launch_asym(const GotohParams& p, int type, uint32_t band, bool width16, hipStream_t s)
{
    switch (band) {
    case 3:  return launch_band_width_asym<3>(p, type, width16, s);
    case 5:  return launch_band_width_asym<5>(p, type, width16, s);
    case 7:  return launch_band_width_asym<7>(p, type, width16, s);
    case 15: return launch_band_width_asym<15>(p, type, width16, s);
    case 31: return launch_band_width_asym<31>(p, type, width16, s);
    default: return hipErrorNotSupported;
    }
}

template <typename QA>
static hipError_t launch(const GotohParams& p, const QA& qa, int type, uint32_t band, bool width16, hipStream_t s)
{
    switch (band) {
    case 3:  return launch_band_width<3, QA>(p, qa, type, width16, s);
    case 5:  return launch_band_width<5, QA>(p, qa, type, width16, s);
    case 7:  return launch_band_width<7, QA>(p, qa, type, width16, s);
    case 15: return launch_band_width<15, QA>(p, qa, type, width16, s);
    case 31: return launch_band_width<31, QA>(p, qa, type, width16, s);
    default: return hipErrorNotSupported;
    }
}

// Longest pattern for which the 16-bit kernel is exact, from the scheme, band and type alone
// (0 = never).  Reachable DP values:
//   LOCAL (scores x32):  0 <= H <= M*S, S = the largest substitution score (match -- or a mismatch / quality-LUT
//                        entry above it: the scheme is arbitrary ints);  E, F, diagonal >= -(|G_o| + |G_e| + |mismatch|);
//                        keys add j < 32.  Need 32*(M*S) + 31 <= 32767 and the negative side far above the sentinel.
//   GLOBAL/SEMI_GLOBAL:  |value| <= (M + BAND + 2) * max|cost|; need that below 15000 so that
//                        no real value comes within a band's worth of gap steps of the sentinel.
// The kernels hold row i's values plus (i + 1) |G_e| (the row frame, banded_gotoh_impl.h: A32::cell_rt): a LOCAL value reaches
// 32 * M * (S + |G_e|) + 31, a GLOBAL / SEMI_GLOBAL one gains at most M * max|cost|.
static uint32_t max_len_16bit(int32_t match, int32_t best_pair /* max substitution score */, int64_t A /* max |cost| */, int32_t gap_open, int32_t gap_ext, int type, uint32_t band,
                              int32_t row_step /* |G_e| of the pattern's gaps */)
{
    if (gap_open > 0 || gap_ext > 0) return 0;
    if (A == 0) return 0xFFFFFFFFu;
    if (type == NVBIO_HIP_LOCAL) {
        if (match < 0 || A > 100) return 0;                          // 32*4*A stays far above -32768
        const int64_t per_row = int64_t(std::max(best_pair, 0)) + row_step;
        if (per_row <= 0) return 0xFFFFFFFFu;
        return uint32_t(1022 / per_row);
    }
    const int64_t lim = (15000 / A - int64_t(band) - 2) / (row_step ? 2 : 1);
    return lim <= 0 ? 0u : uint32_t(lim);
}

} // namespace nvb

template <typename QA>
static int banded_gotoh_dispatch(nvb::GotohParams& p, const QA& qa, int64_t max_abs_cost, int32_t best_pair, int32_t type, uint32_t band_len,
                                 const nvbio_hip_string_set* patterns, hipStream_t s, const char* tag16, const char* tag32, const bool views = false,
                                 const nvb::BoundArgs* bound = nullptr, const bool asym = false)
{
    using namespace nvb;
    // NVBIO_HIP_FORCE_32BIT=1 disables the 16-bit kernels (used by the tests to cover both widths)
    // the row-frame kernels' limit (the plain launch); the limit of the recurrence as written (asymmetric costs, the bounded form -- and LOCAL jobs
    // between the two limits, which the plain launch hands to the A16P instance)
    const bool rt = !asym && !bound;
    const uint32_t lim_plain = test_switch(SW_FORCE_32BIT) == 1 ? 0u
                             : max_len_16bit(p.match, best_pair, max_abs_cost, std::max(p.gap_open, p.txt_gap_open), std::max(p.gap_ext, p.txt_gap_ext), type, band_len, 0);
    const uint32_t lim16 = !rt ? lim_plain : test_switch(SW_FORCE_32BIT) == 1 ? 0u
                         : max_len_16bit(p.match, best_pair, max_abs_cost, std::max(p.gap_open, p.txt_gap_open), std::max(p.gap_ext, p.txt_gap_ext), type, band_len,
                                         p.gap_ext < 0 ? -p.gap_ext : 0);
    const uint32_t lim16p = (rt && type == NVBIO_HIP_LOCAL && lim_plain > lim16) ? lim_plain : lim16;
    const bool fixed = (patterns->length == nullptr);
    hipError_t e = hipSuccess;
    // LDS staging of each lane's words, sized from the longest pattern the caller announces
    // (max_pattern_length, or the fixed length); unknown or too long for 32 KiB per block -> off
    {
        const uint32_t maxM = fixed ? patterns->fixed_length : p.stage_pw /* carries the hint */;
        p.stage_pw = p.stage_tw = 0;
        if (maxM != 0 && test_switch(SW_NO_STAGING) != 1) {
            uint32_t pw = stage_words_pattern(32u / patterns->bits - 1u, maxM, patterns->bits);
            if (views) {     // a reversed view spans the groups [last - ceil16(M) - 15, last] plus one group fetch from the word of last - 15
                const uint32_t per = 32u / patterns->bits, c16 = (maxM + 15u) & ~15u;
                pw = std::max(pw, (per - 1u + c16 + 16u + per - 1u) / per + (patterns->bits == 4 ? 3u : 2u));
            }
            pw = (pw + 3u) & ~3u;
            const uint32_t tw = (stage_words_text(15u, maxM, band_len) + 3u) & ~3u;
            if ((pw + tw) * 1024u <= 32768u) { p.stage_pw = pw; p.stage_tw = tw; }
        }
    }
    // jobs with pattern_len <= lim16 : 16-bit arithmetic;  longer ones : 32-bit arithmetic
    // the bounded form (banded_gotoh_bounded.h): persistent waves over a work counter, zeroed before each of the (at most two) launches
    auto go = [&](const bool width16) -> hipError_t {
        if (asym)   return launch_asym(p, type, band_len, width16, s);
        if (!bound) return launch<QA>(p, qa, type, band_len, width16, s);
        if (hipError_t z = hipMemsetAsync(bound->counter, 0, 4u, s)) return z;
        return launch_bounded<QA>(p, qa, *bound, type, band_len, width16, s);
    };
    p.plain16 = 0u;
    if (lim16 > 0 && (!fixed || patterns->fixed_length <= lim16)) {
        p.len_lo = 0; p.len_hi = lim16;
        g_last_kernel = tag16;
        e = go(true);
        if (e != hipSuccess) return e;
    }
    if (lim16p > lim16 && (!fixed || (patterns->fixed_length > lim16 && patterns->fixed_length <= lim16p))) {
        p.len_lo = lim16 + 1u; p.len_hi = lim16p; p.plain16 = 1u;
        if (fixed) g_last_kernel = tag16;
        e = go(true);
        p.plain16 = 0u;
        if (e != hipSuccess) return e;
    }
    if (lim16p != 0xFFFFFFFFu && (!fixed || patterns->fixed_length > lim16p)) {
        p.len_lo = lim16p > 0 ? lim16p + 1 : 0; p.len_hi = 0xFFFFFFFFu;
        if (fixed || lim16p == 0) g_last_kernel = tag32;
        e = go(false);
    }
    return e;
}

static int check_banded_args(int32_t type, uint32_t band_len, const nvbio_hip_string_set* patterns, const nvbio_hip_string_set* texts)
{
    if (!patterns || !texts) return hipErrorInvalidValue;
    if (type < 0 || type > 2) return hipErrorInvalidValue;
    if (!(patterns->bits == 2 || patterns->bits == 4 || patterns->bits == 8) || texts->bits != 2) return hipErrorNotSupported;
    if (!(band_len == 3 || band_len == 5 || band_len == 7 || band_len == 15 || band_len == 31)) return hipErrorNotSupported;
    return hipSuccess;
}
static int check_banded_ptrs(const nvbio_hip_string_set* patterns, const nvbio_hip_string_set* texts, const int32_t* out_score, const uint32_t* out_sink)
{
    if (!out_score || !out_sink) return hipErrorInvalidValue;
    if (!patterns->words || !texts->words || !patterns->begin || !texts->begin ||
        patterns->n_words == 0 || texts->n_words == 0) return hipErrorInvalidValue;
    return hipSuccess;
}
static inline int64_t iabs64(int32_t v) { return v < 0 ? -int64_t(v) : int64_t(v); }

NVB_API int nvbio_hip_banded_gotoh_score(
    const nvbio_hip_gotoh_scheme* scheme, int32_t type, uint32_t band_len,
    const nvbio_hip_string_set* patterns, const nvbio_hip_string_set* texts,
    uint32_t max_pattern_len, uint32_t max_text_len,
    uint32_t n, int32_t* out_score, uint32_t* out_sink, void* stream)
{
    (void)max_text_len;     // arithmetic width is chosen per job from its own length; max_pattern_len only sizes the LDS staging
    using namespace nvb;
    if (!scheme) return hipErrorInvalidValue;
    if (int e = check_banded_args(type, band_len, patterns, texts)) return e;
    if (n == 0) return hipSuccess;                 // an empty batch is legal and touches nothing
    if (int e = check_banded_ptrs(patterns, texts, out_score, out_sink)) return e;

    GotohParams p;
    p.pat = make_string_set(patterns);
    p.txt = make_string_set(texts);
    p.match = scheme->match; p.mismatch = scheme->mismatch;
    p.gap_open = scheme->gap_open; p.gap_ext = scheme->gap_ext;
    p.txt_gap_open = scheme->gap_open; p.txt_gap_ext = scheme->gap_ext;      // SimpleGotohScheme: utils.h:128-131
    p.n = n; p.out_score = out_score; p.out_sink = out_sink; p.n_dev = nullptr; p.out_index = nullptr; p.gate = nullptr; p.gate_limit = 0;
    p.stage_pw = max_pattern_len; p.stage_tw = 0;       // the hint, consumed by banded_gotoh_dispatch
    const int64_t A = std::max(std::max(iabs64(scheme->match), iabs64(scheme->mismatch)), std::max(iabs64(scheme->gap_open), iabs64(scheme->gap_ext)));
    return banded_gotoh_dispatch(p, NoQual(), A, std::max(scheme->match, scheme->mismatch), type, band_len, patterns, to_stream(stream),
                                 "banded_gotoh_score_kernel<A16>", "banded_gotoh_score_kernel<A32>");
}

NVB_API int nvbio_hip_banded_gotoh_score_qual(
    const nvbio_hip_gotoh_qual_scheme* scheme, int32_t type, uint32_t band_len,
    const nvbio_hip_string_set* patterns, const uint8_t* quals, uint64_t n_quals,
    const nvbio_hip_string_set* texts,
    uint32_t max_pattern_len, uint32_t max_text_len,
    uint32_t n, int32_t* out_score, uint32_t* out_sink, void* stream)
{
    return nvbio_hip_banded_gotoh_score_qual_views(scheme, type, band_len, patterns, quals, n_quals, nullptr, texts, max_pattern_len, max_text_len, n, out_score, out_sink, stream);
}

NVB_API int nvbio_hip_banded_gotoh_score_qual_views(
    const nvbio_hip_gotoh_qual_scheme* scheme, int32_t type, uint32_t band_len,
    const nvbio_hip_string_set* patterns, const uint8_t* quals, uint64_t n_quals, const uint8_t* pattern_flags,
    const nvbio_hip_string_set* texts,
    uint32_t max_pattern_len, uint32_t max_text_len,
    uint32_t n, int32_t* out_score, uint32_t* out_sink, void* stream)
{
    (void)max_text_len;
    using namespace nvb;
    if (!scheme) return hipErrorInvalidValue;
    if (int e = check_banded_args(type, band_len, patterns, texts)) return e;
    if (n == 0) return hipSuccess;
    if (int e = check_banded_ptrs(patterns, texts, out_score, out_sink)) return e;
    if (!quals || n_quals < 4) return hipErrorInvalidValue;
    if (pattern_flags && patterns->bits == 8u) return hipErrorNotSupported;      // reversed / complemented views are a feature of the packed read streams

    GotohParams p;
    p.pat = make_string_set(patterns);
    p.txt = make_string_set(texts);
    p.match = scheme->match; p.mismatch = 0;
    p.gap_open = scheme->pattern_gap_open; p.gap_ext = scheme->pattern_gap_ext;
    p.txt_gap_open = scheme->text_gap_open; p.txt_gap_ext = scheme->text_gap_ext;
    p.n = n; p.out_score = out_score; p.out_sink = out_sink; p.n_dev = nullptr; p.out_index = nullptr; p.gate = nullptr; p.gate_limit = 0;
    p.stage_pw = max_pattern_len; p.stage_tw = 0;       // the hint, consumed by banded_gotoh_dispatch
    QualArgs qa;
    qa.quals = quals; qa.n_quals = n_quals; qa.flags = pattern_flags;
    int64_t A = std::max(std::max(iabs64(scheme->match), iabs64(scheme->pattern_gap_open)), std::max(iabs64(scheme->pattern_gap_ext),
                std::max(iabs64(scheme->text_gap_open), iabs64(scheme->text_gap_ext))));
    int32_t best_pair = scheme->match;
    for (int i = 0; i < 256; ++i) { qa.lut[i] = scheme->mismatch[i]; A = std::max(A, iabs64(scheme->mismatch[i])); best_pair = std::max(best_pair, scheme->mismatch[i]); }
    return banded_gotoh_dispatch(p, qa, A, best_pair, type, band_len, patterns, to_stream(stream),
                                 pattern_flags ? "banded_gotoh_score_kernel<A16,qual,views>" : "banded_gotoh_score_kernel<A16,qual>",
                                 pattern_flags ? "banded_gotoh_score_kernel<A32,qual,views>" : "banded_gotoh_score_kernel<A32,qual>", pattern_flags != nullptr);
}

// The same scorer with a threshold per job (banded_gotoh_bounded.h): a job whose score cannot exceed min_score[i] is given up; its lane takes the next one.
NVB_API int nvbio_hip_banded_gotoh_score_qual_bounded(
    const nvbio_hip_gotoh_qual_scheme* scheme, int32_t type, uint32_t band_len,
    const nvbio_hip_string_set* patterns, const uint8_t* quals, uint64_t n_quals, const uint8_t* pattern_flags,
    const nvbio_hip_string_set* texts,
    uint32_t max_pattern_len, uint32_t max_text_len,
    uint32_t n, const uint32_t* n_on_device, const int32_t* min_score, uint32_t* work_counter, const uint32_t* out_index,
    const uint32_t* gate, uint32_t gate_limit,
    int32_t* out_score, uint32_t* out_sink, void* stream)
{
    (void)max_text_len;
    using namespace nvb;
    if (!scheme || (min_score && !work_counter) || (min_score && gate)) return hipErrorInvalidValue;
    if (int e = check_banded_args(type, band_len, patterns, texts)) return e;
    if (n == 0) return hipSuccess;
    if (int e = check_banded_ptrs(patterns, texts, out_score, out_sink)) return e;
    if (!quals || n_quals < 4) return hipErrorInvalidValue;
    if (pattern_flags && patterns->bits == 8u) return hipErrorNotSupported;      // reversed / complemented views are a feature of the packed read streams

    GotohParams p;
    p.pat = make_string_set(patterns);
    p.txt = make_string_set(texts);
    p.match = scheme->match; p.mismatch = 0;
    p.gap_open = scheme->pattern_gap_open; p.gap_ext = scheme->pattern_gap_ext;
    p.txt_gap_open = scheme->text_gap_open; p.txt_gap_ext = scheme->text_gap_ext;
    p.n = n; p.out_score = out_score; p.out_sink = out_sink; p.n_dev = nullptr; p.out_index = nullptr; p.gate = nullptr; p.gate_limit = 0;
    p.stage_pw = max_pattern_len; p.stage_tw = 0;
    QualArgs qa;
    qa.quals = quals; qa.n_quals = n_quals; qa.flags = pattern_flags;
    int64_t A = std::max(std::max(iabs64(scheme->match), iabs64(scheme->pattern_gap_open)), std::max(iabs64(scheme->pattern_gap_ext),
                std::max(iabs64(scheme->text_gap_open), iabs64(scheme->text_gap_ext))));
    int32_t best_pair = scheme->match;
    for (int i = 0; i < 256; ++i) { qa.lut[i] = scheme->mismatch[i]; A = std::max(A, iabs64(scheme->mismatch[i])); best_pair = std::max(best_pair, scheme->mismatch[i]); }
    // without thresholds this is the plain kernel over a device-side count, writing through the index
    if (!min_score)
    {
        p.n_dev = n_on_device; p.out_index = out_index; p.gate = gate; p.gate_limit = gate_limit;
        return banded_gotoh_dispatch(p, qa, A, best_pair, type, band_len, patterns, to_stream(stream),
                                     pattern_flags ? "banded_gotoh_score_kernel<A16,qual,views>" : "banded_gotoh_score_kernel<A16,qual>",
                                     pattern_flags ? "banded_gotoh_score_kernel<A32,qual,views>" : "banded_gotoh_score_kernel<A32,qual>", pattern_flags != nullptr);
    }
    BoundArgs ba;
    // a gap that scores above zero could lift a path after the bound was taken: no thresholds then (every job exact)
    const bool gaps_ok = scheme->pattern_gap_open <= 0 && scheme->pattern_gap_ext <= 0;
    ba.min_score = gaps_ok ? min_score : nullptr;
    ba.n_dev = n_on_device; ba.counter = work_counter; ba.out_index = out_index;
    ba.cap = std::max(best_pair, 0);
    static const int refill = [] { const char* e = getenv("NVBIO_HIP_BOUNDED_REFILL"); return std::min(64, std::max(1, e ? atoi(e) : 16)); }();
    ba.refill = uint32_t(refill);
    return banded_gotoh_dispatch(p, qa, A, best_pair, type, band_len, patterns, to_stream(stream),
                                 "banded_gotoh_score_bounded_kernel<A16,qual>", "banded_gotoh_score_bounded_kernel<A32,qual>", pattern_flags != nullptr, &ba);
}

// SmithWatermanAligner / EditDistanceAligner in the band (sw_banded_inl.h:340-520): with deletion == insertion the
// recurrence is the Gotoh one with gap_open == gap_ext cell for cell (H >= E, F), same band geometry, same reports.
NVB_API int nvbio_hip_banded_sw_score(
    const nvbio_hip_sw_scheme* scheme, int32_t type, uint32_t band_len,
    const nvbio_hip_string_set* patterns, const nvbio_hip_string_set* texts,
    uint32_t max_pattern_len, uint32_t max_text_len,
    uint32_t n, int32_t* out_score, uint32_t* out_sink, void* stream)
{
    if (!scheme) return hipErrorInvalidValue;
    if (scheme->deletion == scheme->insertion)
    {
        const nvbio_hip_gotoh_scheme g = { scheme->match, scheme->mismatch, scheme->deletion, scheme->deletion };
        return nvbio_hip_banded_gotoh_score(&g, type, band_len, patterns, texts, max_pattern_len, max_text_len, n, out_score, out_sink, stream);
    }
    // deletion != insertion: the move from the previous row (band[j+1] + G, :413 / :432) costs `deletion`, the move along the row
    // (band[j-1] + I, :433 / :459) `insertion`, row zero of a GLOBAL alignment j * deletion (:53) -- the kernels' asymmetric instances
    (void)max_text_len;
    using namespace nvb;
    if (int e = check_banded_args(type, band_len, patterns, texts)) return e;
    if (n == 0) return hipSuccess;
    if (int e = check_banded_ptrs(patterns, texts, out_score, out_sink)) return e;
    GotohParams p;
    p.pat = make_string_set(patterns);
    p.txt = make_string_set(texts);
    p.match = scheme->match; p.mismatch = scheme->mismatch;
    p.gap_open = p.gap_ext = scheme->insertion;
    p.f_gap_open = p.f_gap_ext = scheme->deletion;
    p.txt_gap_open = p.txt_gap_ext = scheme->deletion;
    p.n = n; p.out_score = out_score; p.out_sink = out_sink; p.n_dev = nullptr; p.out_index = nullptr; p.gate = nullptr; p.gate_limit = 0;
    p.stage_pw = max_pattern_len; p.stage_tw = 0;
    const int64_t A = std::max(std::max(iabs64(scheme->match), iabs64(scheme->mismatch)), std::max(iabs64(scheme->deletion), iabs64(scheme->insertion)));
    return banded_gotoh_dispatch(p, NoQual(), A, std::max(scheme->match, scheme->mismatch), type, band_len, patterns, to_stream(stream),
                                 "banded_gotoh_score_kernel<A16X>", "banded_gotoh_score_kernel<A32X>", false, nullptr, true);
}

// Device memory for the C++ host layer's containers and the drop-in layer's vectors: hipMalloc'ed blocks kept in a per-device cache.
//
// Rounds 1-4 took them from a private hipMemPool (hipMallocFromPoolAsync / hipFreeAsync on the legacy default stream).  Under the reference's
// own multi-threaded mode -- nvBowtie --device 0 --device 0: two compute threads, each with its own Aligner, both on the default stream of one
// device -- that pool loses the contents of live blocks: a ~2 MB stretch of one thread's traceback scratch reads back as zeros while the other
// thread allocates and frees (profiles/r05/two_threads_pool.txt: 0 of 9 runs differ from the single-thread run with plain hipMalloc / hipFree,
// 2 of 3 with the pool; no two live blocks ever overlapped, a mutex around every pool call, kernel / copy serialisation and a blocking free all
// left it in place).  So the pool is gone: a freed block goes on a free list, a request takes the smallest listed block of at least its size
// (and at most twice it) or calls hipMalloc.
//   free    nvbio_hip_device_free keeps hipFree's contract: the device is idle before the block goes on the list (a drop-in caller may have
//           used the block on a stream this library never saw -- torch's pool streams, its own hipStreamNonBlocking streams).
//           nvbio_hip_device_free_ordered is the opt-in form that does not stop the host, for callers that vouch for their streams (this
//           repository's C++ host layer, include/nvbio_hip/types.h): work that used the block was queued before the call -- on the default
//           stream, a blocking stream, or a stream made by nvbio_hip_stream_create; for each of the latter the default stream is made to
//           wait (hipStreamWaitEvent) for what that stream holds now.
//   malloc  returns after hipStreamSynchronize(default stream), as it always did (hipMalloc's contract: usable from every stream): everything
//           queued before the block was freed -- the waits included -- has finished by then, whichever thread freed it.
// Blocks stay cached up to NVBIO_HIP_POOL_KEEP_MB (default 2048) per device (nvbio_hip_device_trim hands them back), least recently freed first out (hipFree).  A driver's per-batch
// working set lives in a hip::device_arena (include/nvbio_hip/types.h) and comes through here once.
namespace nvb {
static std::mutex g_streams_mtx;
static std::vector<hipStream_t> g_streams[64];
static void register_stream(int dev, hipStream_t s) { if (dev >= 0 && dev < 64) { std::lock_guard<std::mutex> lock(g_streams_mtx); g_streams[dev].push_back(s); } }
static void forget_stream(hipStream_t s)
{
    std::lock_guard<std::mutex> lock(g_streams_mtx);
    for (auto& v : g_streams) v.erase(std::remove(v.begin(), v.end(), s), v.end());
}

struct CachedBlock { void* ptr; uint64_t bytes; uint64_t stamp; };
struct BlockCache
{
    std::mutex mtx;
    std::vector<CachedBlock> idle;                       // freed, ready to be handed out again
    std::unordered_map<void*, uint64_t> live;            // handed out: size
    uint64_t idle_bytes = 0, clock = 0;
};
static BlockCache g_cache[64];
static int env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
static uint64_t cache_keep_bytes() { static const uint64_t keep = uint64_t(std::max(0, env_int("NVBIO_HIP_POOL_KEEP_MB", 2048))) << 20; return keep; }
// the rounds 1-4 allocator, kept for reproducing what it does under two host threads: NVBIO_HIP_ROCM_POOL=1
static hipMemPool_t rocm_pool(int dev)
{
    static std::mutex mtx;
    static hipMemPool_t pools[64] = {};
    static bool tried[64] = {};
    static const bool use = env_int("NVBIO_HIP_ROCM_POOL", 0) == 1;
    if (!use || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(mtx);
    if (!tried[dev]) {
        tried[dev] = true;
        hipMemPoolProps props = {};
        props.allocType = hipMemAllocationTypePinned;
        props.handleTypes = hipMemHandleTypeNone;
        props.location.type = hipMemLocationTypeDevice;
        props.location.id = dev;
        hipMemPool_t pool = nullptr;
        if (hipMemPoolCreate(&pool, &props) == hipSuccess) {
            uint64_t keep = uint64_t(256) << 20;
            (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
            pools[dev] = pool;
        } else (void)hipGetLastError();
    }
    return pools[dev];
}
static void drop_idle(BlockCache& c, std::vector<void*>& victims, uint64_t keep)          // (c.mtx held) least recently freed first
{
    while (c.idle_bytes > keep && !c.idle.empty())
    {
        size_t k = 0;
        for (size_t j = 1; j < c.idle.size(); ++j) if (c.idle[j].stamp < c.idle[k].stamp) k = j;
        victims.push_back(c.idle[k].ptr);
        c.idle_bytes -= c.idle[k].bytes;
        c.idle[k] = c.idle.back(); c.idle.pop_back();
    }
}
} // namespace nvb
NVB_API int nvbio_hip_device_malloc(void** ptr, uint64_t bytes)
{
    if (!ptr) return hipErrorInvalidValue;
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev)) return e;
    if (dev < 0 || dev >= 64) return hipMalloc(ptr, bytes ? bytes : 1);
    if (hipMemPool_t pool = nvb::rocm_pool(dev))
    {
        if (hipError_t e = hipMallocFromPoolAsync(ptr, bytes ? bytes : 1, pool, nullptr)) return e;
        return hipStreamSynchronize(nullptr);
    }
    const uint64_t n = ((bytes ? bytes : 1) + 511ull) & ~511ull;
    nvb::BlockCache& c = nvb::g_cache[dev];
    void* p = nullptr;
    {
        std::lock_guard<std::mutex> lock(c.mtx);
        size_t best = c.idle.size();
        for (size_t k = 0; k < c.idle.size(); ++k)
            if (c.idle[k].bytes >= n && c.idle[k].bytes <= 2u * n && (best == c.idle.size() || c.idle[k].bytes < c.idle[best].bytes)) best = k;
        if (best != c.idle.size())
        {
            p = c.idle[best].ptr;
            c.live[p] = c.idle[best].bytes;
            c.idle_bytes -= c.idle[best].bytes;
            c.idle[best] = c.idle.back(); c.idle.pop_back();
        }
    }
    if (!p)
    {
        hipError_t e = hipMalloc(&p, n);
        if (e != hipSuccess)
        {   // out of memory with blocks sitting idle: give them back and ask again
            (void)hipGetLastError();
            std::vector<void*> victims;
            { std::lock_guard<std::mutex> lock(c.mtx); nvb::drop_idle(c, victims, 0); }
            for (void* v : victims) (void)hipFree(v);
            e = hipMalloc(&p, n);
            if (e != hipSuccess) return e;
        }
        std::lock_guard<std::mutex> lock(c.mtx);
        c.live[p] = n;
    }
    *ptr = p;
    // NVBIO_HIP_POISON_ALLOC=<byte> (debugging aid, read once): every block handed out is filled with that byte, so a caller that reads storage
    // it never wrote gives results that change with the byte
    static const int poison = nvb::env_int("NVBIO_HIP_POISON_ALLOC", -1);
    if (poison >= 0) (void)hipMemsetAsync(p, poison & 255, bytes ? bytes : 1, nullptr);
    return hipStreamSynchronize(nullptr);          // like hipMalloc: the block is usable from every stream on return
}
// free_impl: `ordered` = the non-blocking form (the caller vouches that every stream that touched the block is the default stream, a blocking
// stream or one made by nvbio_hip_stream_create); otherwise hipFree's contract -- the device is idle before the block can change hands.
static int free_impl(void* ptr, bool ordered, hipStream_t extra = nullptr)
{
    if (!ptr) return hipSuccess;
    int dev = 0;
    static const bool sync_free = nvb::env_int("NVBIO_HIP_SYNC_FREE", 0) == 1;     // debugging aid: the blocking form everywhere
    const bool have_dev = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64;
    if (!have_dev) { if (hipError_t e = hipDeviceSynchronize()) return e; return hipFree(ptr); }
    if (sync_free || !ordered) { if (hipError_t e = hipDeviceSynchronize()) return e; }
    else
    {
        // the library's own non-blocking streams are not ordered with the default stream: put it behind what each of them holds now
        std::lock_guard<std::mutex> lock(nvb::g_streams_mtx);
        std::vector<hipStream_t> behind(nvb::g_streams[dev]);
        if (extra && std::find(behind.begin(), behind.end(), extra) == behind.end()) behind.push_back(extra);
        for (hipStream_t s : behind)
        {
            hipEvent_t ev = nullptr;
            hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventRecord(ev, s);
            if (e == hipSuccess) e = hipStreamWaitEvent(nullptr, ev, 0);
            if (ev) (void)hipEventDestroy(ev);               // (released by the runtime once the wait has been through)
            if (e != hipSuccess)
            {   // could not order behind this stream: the blocking form for this block
                (void)hipGetLastError();
                if (hipError_t es = hipDeviceSynchronize()) return es;
                break;
            }
        }
    }
    if (nvb::rocm_pool(dev)) return hipFreeAsync(ptr, nullptr);
    nvb::BlockCache& c = nvb::g_cache[dev];
    std::vector<void*> victims;
    {
        std::lock_guard<std::mutex> lock(c.mtx);
        auto it = c.live.find(ptr);
        if (it == c.live.end()) victims.push_back(ptr);      // not from here (or freed twice): hipFree says which
        else
        {
            c.idle.push_back(nvb::CachedBlock{ ptr, it->second, ++c.clock });
            c.idle_bytes += it->second;
            c.live.erase(it);
            nvb::drop_idle(c, victims, nvb::cache_keep_bytes());
        }
    }
    hipError_t r = hipSuccess;
    for (void* v : victims) { const hipError_t e = hipFree(v); if (e != hipSuccess) r = e; }      // (hipFree waits for the device itself)
    return r;
}
NVB_API int nvbio_hip_device_free(void* ptr)         { return free_impl(ptr, false); }
NVB_API int nvbio_hip_device_free_ordered(void* ptr) { return free_impl(ptr, true); }
NVB_API int nvbio_hip_device_free_after(void* ptr, void* stream) { return free_impl(ptr, true, nvb::to_stream(stream)); }
// hand every idle block of the calling thread's device back to the runtime (hipFree): for a process that shares the device with another
// allocator (torch's) and is about to let that one grow
NVB_API int nvbio_hip_device_trim(void)
{
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev)) return e;
    if (dev < 0 || dev >= 64) return hipSuccess;
    std::vector<void*> victims;
    { std::lock_guard<std::mutex> lock(nvb::g_cache[dev].mtx); nvb::drop_idle(nvb::g_cache[dev], victims, 0); }
    hipError_t r = hipSuccess;
    for (void* v : victims) { const hipError_t e = hipFree(v); if (e != hipSuccess) r = e; }
    return r;
}
// free / total bytes of the calling thread's device (hipMemGetInfo) plus what the library's block cache holds idle
NVB_API int nvbio_hip_device_mem_info(uint64_t* free_bytes, uint64_t* total_bytes, uint64_t* idle_cached_bytes)
{
    size_t f = 0, t = 0;
    if (hipError_t e = hipMemGetInfo(&f, &t)) return e;
    int dev = 0;
    uint64_t idle = 0;
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) { std::lock_guard<std::mutex> lock(nvb::g_cache[dev].mtx); idle = nvb::g_cache[dev].idle_bytes; }
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    if (idle_cached_bytes) *idle_cached_bytes = idle;
    return hipSuccess;
}
// ---- streams for the C++ host layer (which has no HIP headers).  The reference's drivers run on the default stream of one host thread
// per device (nvBowtie.cpp:809-864); here one device serves several batches at once: a driver object per host thread, each on its own
// non-blocking stream, so that one batch's fabric-bound seeding overlaps another's VALU-bound extension.
namespace nvb {
uint32_t seeding_grid(uint64_t n_items) { return uint32_t((n_items + 255u) / 256u); }
} // namespace nvb
NVB_API int nvbio_hip_stream_create(void** stream, uint32_t non_blocking)
{
    if (!stream) return hipErrorInvalidValue;
    hipStream_t s = nullptr;
    const hipError_t e = hipStreamCreateWithFlags(&s, non_blocking ? hipStreamNonBlocking : hipStreamDefault);
    *stream = s;
    int dev = 0;
    if (e == hipSuccess && hipGetDevice(&dev) == hipSuccess) nvb::register_stream(dev, s);
    return e;
}
NVB_API int nvbio_hip_stream_destroy(void* stream)
{
    if (!stream) return hipSuccess;
    nvb::forget_stream(nvb::to_stream(stream));
    return hipStreamDestroy(nvb::to_stream(stream));
}

// ---- test switches
namespace nvb {
static std::atomic<int> g_switch[SW_COUNT];
static const char* const g_switch_name[SW_COUNT] = { "NVBIO_HIP_FORCE_32BIT", "NVBIO_HIP_NO_STAGING", "NVBIO_HIP_FULL_GENERIC", "NVBIO_HIP_ED_SWEEP",
                                                     "NVBIO_HIP_FULL_SINGLE_JOB", "NVBIO_HIP_FULL_ROWS", "NVBIO_HIP_TRACEBACK_LANES", "NVBIO_HIP_SELECT_LANES" };
static void seed_switches()
{
    static std::once_flag once;
    std::call_once(once, [] { for (int k = 0; k < SW_COUNT; ++k) { const char* e = getenv(g_switch_name[k]); g_switch[k].store(e ? atoi(e) : 0, std::memory_order_relaxed); } });
}
int test_switch(TestSwitch which) { seed_switches(); return g_switch[which].load(std::memory_order_relaxed); }
} // namespace nvb
NVB_API int nvbio_hip_set_test_switch(const char* name, int value)
{
    if (!name) return hipErrorInvalidValue;
    nvb::seed_switches();
    for (int k = 0; k < nvb::SW_COUNT; ++k)
        if (strcmp(name, nvb::g_switch_name[k]) == 0) { nvb::g_switch[k].store(value, std::memory_order_relaxed); return hipSuccess; }
    return hipErrorInvalidValue;
}
NVB_API const char* nvbio_hip_test_switch_name(int index) { return (index >= 0 && index < nvb::SW_COUNT) ? nvb::g_switch_name[index] : nullptr; }
NVB_API int nvbio_hip_get_test_switch(const char* name)
{
    if (!name) return -1;
    for (int k = 0; k < nvb::SW_COUNT; ++k) if (strcmp(name, nvb::g_switch_name[k]) == 0) return nvb::test_switch(nvb::TestSwitch(k));
    return -1;
}

NVB_API int nvbio_hip_memcpy(void* dst, const void* src, uint64_t bytes, int kind, void* stream)
{
    const hipMemcpyKind k = kind == 1 ? hipMemcpyHostToDevice : kind == 2 ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
    if (bytes == 0) return hipSuccess;
    hipError_t e = hipMemcpyAsync(dst, src, bytes, k, nvb::to_stream(stream));
    if (e != hipSuccess) return e;
    return kind == 2 ? hipStreamSynchronize(nvb::to_stream(stream)) : hipSuccess;
}
NVB_API int nvbio_hip_memset(void* dst, int value, uint64_t bytes, void* stream) { return bytes ? hipMemsetAsync(dst, value, bytes, nvb::to_stream(stream)) : hipSuccess; }
NVB_API int nvbio_hip_stream_synchronize(void* stream) { return hipStreamSynchronize(nvb::to_stream(stream)); }
NVB_API int nvbio_hip_stream_query(void* stream)
{
    const hipError_t e = hipStreamQuery(nvb::to_stream(stream));
    if (e == hipErrorNotReady) (void)hipGetLastError();          // "not yet" is an answer, not a sticky error
    return e;
}
NVB_API int nvbio_hip_host_malloc(void** ptr, uint64_t bytes)
{
    if (!ptr) return hipErrorInvalidValue;
    *ptr = nullptr;
    return hipHostMalloc(ptr, bytes ? bytes : 1u, hipHostMallocDefault);
}
NVB_API int nvbio_hip_host_free(void* ptr) { return ptr ? hipHostFree(ptr) : hipSuccess; }

NVB_API int         nvbio_hip_abi_version(void) { return NVBIO_HIP_ABI_VERSION; }
NVB_API const char* nvbio_hip_arch(void)        { return "gfx950"; }
NVB_API const char* nvbio_hip_last_kernel(void) { return nvb::g_last_kernel; }
