// nvbio_hip/alignment.h -- nvbio::aln batch interface of the banded Gotoh score on MI355X.
//
// Mirrors, name for name:
//   AlignmentType, GotohAligner, make_gotoh_aligner        nvbio/alignment/alignment_base.h:54,255-298
//   SimpleGotohScheme                                      nvbio/alignment/utils.h:114-134
//   BestSink<int32>                                        nvbio/alignment/sink.h:68-89
//   DeviceThreadScheduler / DeviceThreadBlockScheduler     nvbio/alignment/batched.h:51-76
//   batch_banded_alignment_score<BAND_LEN>(...)            nvbio/alignment/batched.h:217-231
//   BatchedBandedAlignmentScore<BAND_LEN,stream,scheduler> nvbio/alignment/batched.h:333-353
// Work is done by nvbio_hip_banded_gotoh_score (include/nvbio_hip.h); like the reference's
// enact(), calls are asynchronous on the current (null) stream and return nothing.
#pragma once
#include <algorithm>
#include "strings.h"

namespace nvbio {
namespace aln {

enum AlignmentType { GLOBAL = 0, LOCAL = 1, SEMI_GLOBAL = 2 };

/// algorithm tags (alignment_base.h:72-79).  The banded aligners ignore them; the full-matrix batch
/// implements the text-blocking form (its visiting order decides LOCAL ties).
struct PatternBlockingTag {};
struct TextBlockingTag {};

/// SimpleSmithWatermanScheme (utils.h:92-109): linear gap costs
struct SimpleSmithWatermanScheme
{
    SimpleSmithWatermanScheme() : m_match(0), m_mismatch(0), m_deletion(0), m_insertion(0) {}
    SimpleSmithWatermanScheme(const int32 match, const int32 mm, const int32 del, const int32 ins)
        : m_match(match), m_mismatch(mm), m_deletion(del), m_insertion(ins) {}
    int32 match(const uint8 = 0) const { return m_match; }
    int32 mismatch(const uint8 = 0) const { return m_mismatch; }
    int32 mismatch(const uint8, const uint8, const uint8 = 0) const { return m_mismatch; }
    int32 deletion() const { return m_deletion; }
    int32 insertion() const { return m_insertion; }
    int32 m_match, m_mismatch, m_deletion, m_insertion;
};

struct SimpleGotohScheme
{
    SimpleGotohScheme() : m_match(0), m_mismatch(0), m_gap_open(0), m_gap_ext(0) {}
    SimpleGotohScheme(const int32 match, const int32 mm, const int32 gap_open, const int32 gap_ext)
        : m_match(match), m_mismatch(mm), m_gap_open(gap_open), m_gap_ext(gap_ext) {}
    int32 match(const uint8 = 0) const { return m_match; }
    int32 mismatch(const uint8 = 0) const { return m_mismatch; }
    int32 pattern_gap_open() const { return m_gap_open; }
    int32 pattern_gap_extension() const { return m_gap_ext; }
    int32 text_gap_open() const { return m_gap_open; }
    int32 text_gap_extension() const { return m_gap_ext; }
    int32 m_match, m_mismatch, m_gap_open, m_gap_ext;
};

/// nvBowtie's SmithWatermanScoringScheme<QualCost<int>,ConstantCost<int>> restricted to what the
/// Gotoh aligner reads (nvBowtie/bowtie2/cuda/scoring.h:206-356): same member names.
struct SmithWatermanScoringScheme
{
    SmithWatermanScoringScheme() : m_read_gap_const(5), m_read_gap_coeff(3), m_ref_gap_const(5), m_ref_gap_coeff(3),
                                   m_match(0), m_mmp_min(2), m_mmp_max(6), m_mmp_constant(false) {}
    static SmithWatermanScoringScheme local() { SmithWatermanScoringScheme s; s.m_match = 2; return s; }
    /// the costs of EditDistanceAligner (ed_utils.h:44-51: match 0, everything else -1) in this scheme's terms: with them the Gotoh
    /// recurrences are the linear-gap ones cell for cell (gap open == gap extension, so H >= E, F), which is how the drivers' scoring
    /// stages run nvBowtie's edit-distance mode (EditDistanceScoringScheme, scoring.h:133-200) on the quality-scheme kernels
    static SmithWatermanScoringScheme edit_distance()
    { SmithWatermanScoringScheme s; s.m_read_gap_const = s.m_ref_gap_const = 0; s.m_read_gap_coeff = s.m_ref_gap_coeff = 1; s.m_match = 0; s.m_mmp_min = s.m_mmp_max = 1; s.m_mmp_constant = true; return s; }

    /// QualCost<int>::operator() (scoring.h:96-100), single precision with truncation as written there
    int32 mmp(const int q) const {
        if (m_mmp_constant) return m_mmp_max;
        const float frac = (float)((q < 40 ? q : 40) / 40.0f);
        return m_mmp_min + int32(frac * (m_mmp_max - m_mmp_min));
    }
    int32 match(const uint8 = 0) const { return m_match; }
    int32 mismatch(const uint8 q = 0) const { return -mmp(q); }
    int32 pattern_gap_open() const { return -m_read_gap_const - m_read_gap_coeff; }
    int32 pattern_gap_extension() const { return -m_read_gap_coeff; }
    int32 text_gap_open() const { return -m_ref_gap_const - m_ref_gap_coeff; }
    int32 text_gap_extension() const { return -m_ref_gap_coeff; }

    nvbio_hip_gotoh_qual_scheme abi() const {
        nvbio_hip_gotoh_qual_scheme s;
        s.match = match(); s.pattern_gap_open = pattern_gap_open(); s.pattern_gap_ext = pattern_gap_extension();
        s.text_gap_open = text_gap_open(); s.text_gap_ext = text_gap_extension();
        for (int q = 0; q < 256; ++q) s.mismatch[q] = mismatch(uint8(q));
        return s;
    }
    int m_read_gap_const, m_read_gap_coeff, m_ref_gap_const, m_ref_gap_coeff;
    int m_match, m_mmp_min, m_mmp_max; bool m_mmp_constant;
};

struct GotohTag {};
struct SmithWatermanTag {};
struct EditDistanceTag {};

template <AlignmentType T, typename scoring_scheme_type, typename algorithm_tag = PatternBlockingTag>
struct GotohAligner
{
    static const AlignmentType TYPE = T;
    typedef GotohTag            aligner_tag;
    typedef algorithm_tag       algorithm_type;
    typedef scoring_scheme_type scoring_scheme;
    GotohAligner(const scoring_scheme_type _scheme) : scheme(_scheme) {}
    scoring_scheme_type scheme;
};
template <AlignmentType TYPE, typename scoring_scheme_type>
GotohAligner<TYPE, scoring_scheme_type> make_gotoh_aligner(const scoring_scheme_type& scheme) { return GotohAligner<TYPE, scoring_scheme_type>(scheme); }
template <AlignmentType TYPE, typename algorithm_tag, typename scoring_scheme_type>
GotohAligner<TYPE, scoring_scheme_type, algorithm_tag> make_gotoh_aligner(const scoring_scheme_type& scheme) { return GotohAligner<TYPE, scoring_scheme_type, algorithm_tag>(scheme); }

/// SmithWatermanAligner<TYPE, scheme, algorithm_tag> (alignment_base.h): linear-gap DP
template <AlignmentType T, typename scoring_scheme_type, typename algorithm_tag = PatternBlockingTag>
struct SmithWatermanAligner
{
    static const AlignmentType TYPE = T;
    typedef SmithWatermanTag    aligner_tag;
    typedef algorithm_tag       algorithm_type;
    typedef scoring_scheme_type scoring_scheme;
    SmithWatermanAligner(const scoring_scheme_type _scheme) : scheme(_scheme) {}
    scoring_scheme_type scheme;
};
template <AlignmentType TYPE, typename scoring_scheme_type>
SmithWatermanAligner<TYPE, scoring_scheme_type> make_smith_waterman_aligner(const scoring_scheme_type& scheme) { return SmithWatermanAligner<TYPE, scoring_scheme_type>(scheme); }
template <AlignmentType TYPE, typename algorithm_tag, typename scoring_scheme_type>
SmithWatermanAligner<TYPE, scoring_scheme_type, algorithm_tag> make_smith_waterman_aligner(const scoring_scheme_type& scheme) { return SmithWatermanAligner<TYPE, scoring_scheme_type, algorithm_tag>(scheme); }

/// EditDistanceAligner<TYPE, algorithm_tag>: the SW code with EditDistanceSWScheme (ed_utils.h:44-51)
template <AlignmentType T, typename algorithm_tag = PatternBlockingTag>
struct EditDistanceAligner
{
    static const AlignmentType TYPE = T;
    typedef EditDistanceTag           aligner_tag;
    typedef algorithm_tag             algorithm_type;
    typedef SimpleSmithWatermanScheme scoring_scheme;
    EditDistanceAligner() : scheme(0, -1, -1, -1) {}
    SimpleSmithWatermanScheme scheme;
};
template <AlignmentType TYPE> EditDistanceAligner<TYPE> make_edit_distance_aligner() { return EditDistanceAligner<TYPE>(); }
template <AlignmentType TYPE, typename algorithm_tag> EditDistanceAligner<TYPE, algorithm_tag> make_edit_distance_aligner() { return EditDistanceAligner<TYPE, algorithm_tag>(); }

namespace priv {
inline nvbio_hip_gotoh_scheme abi_scheme(const SimpleGotohScheme& s) { const nvbio_hip_gotoh_scheme r = { s.m_match, s.m_mismatch, s.m_gap_open, s.m_gap_ext }; return r; }
inline nvbio_hip_sw_scheme    abi_scheme(const SimpleSmithWatermanScheme& s) { const nvbio_hip_sw_scheme r = { s.m_match, s.m_mismatch, s.m_deletion, s.m_insertion }; return r; }
/// the C-ABI entry for each scheme kind: Gotoh / linear-gap (SW, ED)
inline int banded_score(const nvbio_hip_gotoh_scheme& sc, int32 type, uint32 band, const nvbio_hip_string_set& p, const nvbio_hip_string_set& t,
                        uint32 maxP, uint32 maxT, uint32 n, int32* score, uint32* sink, void* stream)
{ return nvbio_hip_banded_gotoh_score(&sc, type, band, &p, &t, maxP, maxT, n, score, sink, stream); }
inline int banded_score(const nvbio_hip_sw_scheme& sc, int32 type, uint32 band, const nvbio_hip_string_set& p, const nvbio_hip_string_set& t,
                        uint32 maxP, uint32 maxT, uint32 n, int32* score, uint32* sink, void* stream)
{ return nvbio_hip_banded_sw_score(&sc, type, band, &p, &t, maxP, maxT, n, score, sink, stream); }
inline int full_score(const nvbio_hip_gotoh_scheme& sc, int32 type, const nvbio_hip_string_set& p, const nvbio_hip_string_set& t,
                      uint32 maxP, uint32 maxT, const int32* min_score, uint32 n, int32* score, uint32* sink, uint8* ok, void* stream)
{ return nvbio_hip_gotoh_score(&sc, type, &p, &t, maxP, maxT, min_score, n, score, sink, ok, stream); }
inline int full_score(const nvbio_hip_sw_scheme& sc, int32 type, const nvbio_hip_string_set& p, const nvbio_hip_string_set& t,
                      uint32 maxP, uint32 maxT, const int32*, uint32 n, int32* score, uint32* sink, uint8* ok, void* stream)
{   // the text-blocking SW / ED form never exits early (sw_inl.h:1075-1222): every job returns true
    const int e = nvbio_hip_sw_score(&sc, type, &p, &t, maxP, maxT, n, score, sink, stream);
    if (e == 0 && ok && n) return nvbio_hip_memset(ok, 1, n, stream);
    return e;
}
struct int4_scheme { int32 v[4]; };
inline int4_scheme scheme4(const SimpleGotohScheme& s) { const int4_scheme r = { { s.m_match, s.m_mismatch, s.m_gap_open, s.m_gap_ext } }; return r; }
inline int4_scheme scheme4(const SimpleSmithWatermanScheme& s) { const int4_scheme r = { { s.m_match, s.m_mismatch, s.m_deletion, s.m_insertion } }; return r; }
inline int32 aligner_kind(const SimpleGotohScheme&) { return NVBIO_HIP_GOTOH_ALIGNER; }
inline int32 aligner_kind(const SimpleSmithWatermanScheme&) { return NVBIO_HIP_SW_ALIGNER; }
template <typename A, typename B> struct same_type { static const bool value = false; };
template <typename A> struct same_type<A, A> { static const bool value = true; };
} // namespace priv

/// BestSink<int32> in device memory is {int32 score; uint2 sink}
template <typename ScoreType> struct BestSink { ScoreType score; uint2 sink; };

/// structure-of-arrays iterator over device BestSink storage: what the C-ABI writes
struct BestSinkArrays { int32* score; uint32* sink; };

struct HostThreadScheduler {};
template <uint32 BLOCKDIM, uint32 MINBLOCKS> struct DeviceThreadBlockScheduler {};
typedef DeviceThreadBlockScheduler<128, 1> DeviceThreadScheduler;

/// The stream of alignment jobs handed to BatchedBandedAlignmentScore::enact.  The reference's
/// stream concept (batched.h:239-296) is a user functor bundle evaluated per thread; across a
/// C-ABI the kernels need its data instead, so this stream exposes the same accessors
/// (aligner(), size(), max_pattern_length(), max_text_length()) plus the string sets and sinks.
template <typename t_aligner_type, typename pattern_set_type, typename text_set_type>
struct PackedAlignmentStream
{
    typedef t_aligner_type aligner_type;
    PackedAlignmentStream(aligner_type _aligner, pattern_set_type _patterns, text_set_type _texts, BestSinkArrays _sinks,
                          uint32 _max_pattern_len = 0, uint32 _max_text_len = 0)
        : m_aligner(_aligner), m_patterns(_patterns), m_texts(_texts), m_sinks(_sinks),
          m_max_pattern_len(_max_pattern_len), m_max_text_len(_max_text_len) {}
    const aligner_type& aligner() const { return m_aligner; }
    uint32 size() const { return m_patterns.size(); }
    uint32 max_pattern_length() const { return m_max_pattern_len; }
    uint32 max_text_length() const { return m_max_text_len; }
    aligner_type m_aligner; pattern_set_type m_patterns; text_set_type m_texts; BestSinkArrays m_sinks;
    uint32 m_max_pattern_len, m_max_text_len;
};

template <uint32 BAND_LEN, typename stream_type, typename algorithm_type = DeviceThreadScheduler>
struct BatchedBandedAlignmentScore
{
    typedef typename stream_type::aligner_type aligner_type;
    static uint64 min_temp_storage(const uint32, const uint32, const uint32) { return 0u; }   // batched_banded_inl.h:143-147
    static uint64 max_temp_storage(const uint32, const uint32, const uint32) { return 0u; }

    void enact(stream_type stream, uint64 temp_size = 0u, uint8* temp = nullptr, void* hip_stream = nullptr)
    {
        (void)temp_size; (void)temp;
        static_assert(BAND_LEN == 3 || BAND_LEN == 5 || BAND_LEN == 7 || BAND_LEN == 15 || BAND_LEN == 31, "unsupported BAND_LEN");
        const nvbio_hip_string_set p = stream.m_patterns.abi(), t = stream.m_texts.abi();
        hip_check(priv::banded_score(priv::abi_scheme(stream.aligner().scheme), int32(aligner_type::TYPE), BAND_LEN, p, t,
                                     stream.max_pattern_length(), stream.max_text_length(),
                                     stream.size(), stream.m_sinks.score, stream.m_sinks.sink, hip_stream),
                  "nvbio_hip_banded_{gotoh,sw}_score");
    }
};

/// The quality-aware form used by nvBowtie's extension stage (score_best_inl.h:153-201): aligner =
/// GotohAligner<TYPE, SmithWatermanScoringScheme>, one quality byte per read symbol at the reads' offsets.
template <uint32 BAND_LEN, AlignmentType TYPE, typename pattern_set_type, typename text_set_type>
void batch_banded_alignment_score(
    const GotohAligner<TYPE, SmithWatermanScoringScheme> aligner,
    const pattern_set_type  patterns,
    const uint8*            quals,
    const uint64            n_quals,
    const text_set_type     texts,
          BestSinkArrays    sinks,
    const uint32            max_pattern_length,
    const uint32            max_text_length,
    void*                   hip_stream = nullptr)
{
    const nvbio_hip_gotoh_qual_scheme sc = aligner.scheme.abi();
    const nvbio_hip_string_set p = patterns.abi(), t = texts.abi();
    hip_check(nvbio_hip_banded_gotoh_score_qual(&sc, int32(TYPE), BAND_LEN, &p, quals, n_quals, &t, max_pattern_length, max_text_length,
                                                patterns.size(), sinks.score, sinks.sink, hip_stream), "nvbio_hip_banded_gotoh_score_qual");
}

/// ... with the stream's per-job min_score (BestScoreStream::init_context sets it to the read's second-best score, score_best_inl.h:113-116;
/// banded_alignment_score takes it for its early-out, gotoh_banded_inl.h:622-634): a job that cannot end above min_score[i] is given up and
/// reports some score <= min_score[i] -- what nvBowtie's reduction does with it is the same (reduce_inl.h:111-135); jobs ending above it are
/// exact.  `n_jobs_on_device` (optional): the job count lives on the device and patterns.size() is only the arrays' capacity -- no host
/// round trip between the kernel that counts the jobs and this one.  `work_counter`: 4 bytes of device scratch owned by the caller.
/// `out_index` (optional): job i's score and sink are written at sinks.score[out_index[i]] / sinks.sink[out_index[i]].
template <uint32 BAND_LEN, AlignmentType TYPE, typename pattern_set_type, typename text_set_type>
void batch_banded_alignment_score(
    const GotohAligner<TYPE, SmithWatermanScoringScheme> aligner,
    const pattern_set_type  patterns,
    const uint8*            quals,
    const uint64            n_quals,
    const text_set_type     texts,
    const int32*            min_score,
    const uint32*           n_jobs_on_device,
          uint32*           work_counter,
    const uint32*           out_index,
          BestSinkArrays    sinks,
    const uint32            max_pattern_length,
    const uint32            max_text_length,
    void*                   hip_stream = nullptr,
    const uint32            wave_form_up_to = 0u)
{
    const nvbio_hip_gotoh_qual_scheme sc = aligner.scheme.abi();
    const nvbio_hip_string_set p = patterns.abi(), t = texts.abi();
    // wave_form_up_to (needs the count on the device, no thresholds, patterns the wave kernel takes): both forms are queued, the device runs the
    // wave-per-job sweep when the batch holds at most that many jobs and the lane-per-job kernel otherwise
    const bool both = wave_form_up_to != 0u && n_jobs_on_device != nullptr && min_score == nullptr && max_pattern_length != 0u && max_pattern_length <= 512u;
    hip_check(nvbio_hip_banded_gotoh_score_qual_bounded(&sc, int32(TYPE), BAND_LEN, &p, quals, n_quals, nullptr, &t, max_pattern_length, max_text_length,
                                                        patterns.size(), n_jobs_on_device, min_score, work_counter, out_index, both ? n_jobs_on_device : nullptr, wave_form_up_to,
                                                        sinks.score, sinks.sink, hip_stream),
              "nvbio_hip_banded_gotoh_score_qual_bounded");
    if (both)
        hip_check(nvbio_hip_banded_gotoh_score_qual_wave(&sc, int32(TYPE), BAND_LEN, &p, quals, n_quals, &t, max_pattern_length, std::min<uint32>(patterns.size(), wave_form_up_to),
                                                         n_jobs_on_device, nullptr, out_index, n_jobs_on_device, wave_form_up_to, sinks.score, sinks.sink, hip_stream),
                  "nvbio_hip_banded_gotoh_score_qual_wave");
}

} // namespace aln

/// io::Cigar (nvbio/io/alignments.h:57-75): what nvBowtie's Backtracker writes
namespace io {
struct Cigar
{
    enum Operation { SUBSTITUTION = 0, INSERTION = 1, DELETION = 2, SOFT_CLIPPING = 3 };
    Cigar() {}
    Cigar(const uint8 type, const uint16 len) : m_type(type), m_len(len) {}
    uint16 m_type:2, m_len:14;
};
} // namespace io

namespace aln {

/// Alignment<int32> (alignment.h: score, source, sink) as device structure-of-arrays + the CIGARs the
/// backtracer formed: job i's entries are cigar[i*cigar_stride .. + cigar_len[i]), end of the alignment
/// first (the Backtracker stores them backwards, alignment_utils.h:121-123)
struct AlignmentArrays { int32* score; uint32* source; uint32* sink; };
struct CigarArrays     { io::Cigar* cigar; uint32 cigar_stride; uint32* cigar_len; };

/// The stream handed to BatchedBandedAlignmentTraceback::enact: the score stream's data plus the
/// traceback outputs (and, for nvBowtie's quality-aware scheme, the read qualities).
template <typename t_aligner_type, typename pattern_set_type, typename text_set_type>
struct PackedTracebackStream
{
    typedef t_aligner_type aligner_type;
    PackedTracebackStream(aligner_type _aligner, pattern_set_type _patterns, text_set_type _texts,
                          AlignmentArrays _alignments, CigarArrays _cigars, uint32 _max_pattern_len, uint32 _max_text_len = 0,
                          const uint8* _quals = nullptr, uint64 _n_quals = 0)
        : m_aligner(_aligner), m_patterns(_patterns), m_texts(_texts), m_alignments(_alignments), m_cigars(_cigars),
          m_max_pattern_len(_max_pattern_len), m_max_text_len(_max_text_len), m_quals(_quals), m_n_quals(_n_quals) {}
    const aligner_type& aligner() const { return m_aligner; }
    uint32 size() const { return m_patterns.size(); }
    uint32 max_pattern_length() const { return m_max_pattern_len; }
    uint32 max_text_length() const { return m_max_text_len; }
    aligner_type m_aligner; pattern_set_type m_patterns; text_set_type m_texts;
    AlignmentArrays m_alignments; CigarArrays m_cigars;
    uint32 m_max_pattern_len, m_max_text_len; const uint8* m_quals; uint64 m_n_quals;
};

/// BatchedBandedAlignmentTraceback<BAND_LEN, CHECKPOINTS, stream, scheduler> (batched.h:460-476).
/// CHECKPOINTS is kept for signature parity; the temp storage holds the flow flags of the whole band
/// (min_temp_storage == max_temp_storage, and enact requires it).
template <uint32 BAND_LEN, uint32 CHECKPOINTS, typename stream_type, typename algorithm_type = DeviceThreadScheduler>
struct BatchedBandedAlignmentTraceback
{
    typedef typename stream_type::aligner_type aligner_type;
    static uint64 min_temp_storage(const uint32 max_pattern_len, const uint32, const uint32 stream_size)
    { return nvbio_hip_banded_gotoh_traceback_temp_bytes(BAND_LEN, max_pattern_len, stream_size); }
    static uint64 max_temp_storage(const uint32 max_pattern_len, const uint32 max_text_len, const uint32 stream_size)
    { return min_temp_storage(max_pattern_len, max_text_len, stream_size); }

    /// known_sinks (not in the reference; nvBowtie's scheme only): the stream's score / sink arrays already hold what the banded scorer
    /// reports for these jobs (the extension stage ran that DP), so the score pass is skipped
    bool known_sinks;
    BatchedBandedAlignmentTraceback() : known_sinks(false) {}

    void enact(stream_type stream, uint64 temp_size, uint8* temp, void* hip_stream = nullptr)
    {
        static_assert(BAND_LEN == 3 || BAND_LEN == 5 || BAND_LEN == 7 || BAND_LEN == 15 || BAND_LEN == 31, "unsupported BAND_LEN");
        static_assert(sizeof(io::Cigar) == 2, "io::Cigar must be a uint16 bit-field");
        call(stream.aligner().scheme, stream, temp_size, temp, hip_stream, known_sinks);
    }
private:
    static void call(const SimpleGotohScheme& scheme, stream_type& stream, uint64 temp_size, uint8* temp, void* hip_stream, bool)
    {
        const nvbio_hip_gotoh_scheme sc = { scheme.m_match, scheme.m_mismatch, scheme.m_gap_open, scheme.m_gap_ext };
        const nvbio_hip_string_set p = stream.m_patterns.abi(), t = stream.m_texts.abi();
        hip_check(nvbio_hip_banded_gotoh_traceback(&sc, int32(aligner_type::TYPE), BAND_LEN, &p, &t,
                      stream.max_pattern_length(), stream.max_text_length(), stream.size(),
                      stream.m_alignments.score, stream.m_alignments.sink, stream.m_alignments.source,
                      reinterpret_cast<uint16*>(stream.m_cigars.cigar), stream.m_cigars.cigar_stride, stream.m_cigars.cigar_len,
                      temp, temp_size, hip_stream), "nvbio_hip_banded_gotoh_traceback");
    }
    static void call(const SmithWatermanScoringScheme& scheme, stream_type& stream, uint64 temp_size, uint8* temp, void* hip_stream, bool known)
    {
        const nvbio_hip_gotoh_qual_scheme sc = scheme.abi();
        const nvbio_hip_string_set p = stream.m_patterns.abi(), t = stream.m_texts.abi();
        hip_check((known ? nvbio_hip_banded_gotoh_traceback_qual_known : nvbio_hip_banded_gotoh_traceback_qual)(&sc, int32(aligner_type::TYPE), BAND_LEN, &p, stream.m_quals, stream.m_n_quals, &t,
                      stream.max_pattern_length(), stream.max_text_length(), stream.size(),
                      stream.m_alignments.score, stream.m_alignments.sink, stream.m_alignments.source,
                      reinterpret_cast<uint16*>(stream.m_cigars.cigar), stream.m_cigars.cigar_stride, stream.m_cigars.cigar_len,
                      temp, temp_size, hip_stream), "nvbio_hip_banded_gotoh_traceback_qual");
    }
};

/// BatchedAlignmentScore<stream, scheduler> (batched.h:310-329) for the full-matrix Gotoh score with
/// the text-blocking aligner, as sw-benchmark instantiates it (sw-benchmark.cu:604-631)
template <typename stream_type, typename algorithm_type = DeviceThreadScheduler>
struct BatchedAlignmentScore
{
    typedef typename stream_type::aligner_type aligner_type;
    static uint64 min_temp_storage(const uint32, const uint32, const uint32) { return 0u; }   // no column storage: the
    static uint64 max_temp_storage(const uint32, const uint32, const uint32) { return 0u; }   // pattern lives in a wave's registers

    /// min_score / ok: optional device arrays (the reference's context.min_score and per-job bool)
    void enact(stream_type stream, uint64 temp_size = 0u, uint8* temp = nullptr,
               const int32* min_score = nullptr, uint8* ok = nullptr, void* hip_stream = nullptr)
    {
        (void)temp_size; (void)temp;
        const nvbio_hip_string_set p = stream.m_patterns.abi(), t = stream.m_texts.abi();
        const bool text_blocking = priv::same_type<typename aligner_type::algorithm_type, TextBlockingTag>::value;
        hip_check(nvbio_hip_alignment_score(priv::aligner_kind(stream.aligner().scheme), text_blocking ? NVBIO_HIP_TEXT_BLOCKING : NVBIO_HIP_PATTERN_BLOCKING,
                                            priv::scheme4(stream.aligner().scheme).v, int32(aligner_type::TYPE), &p, &t,
                                            stream.max_pattern_length(), stream.max_text_length(),
                                            min_score, stream.size(), stream.m_sinks.score, stream.m_sinks.sink, ok, hip_stream),
                  "nvbio_hip_alignment_score");
    }
};

/// BatchedAlignmentTraceback<CHECKPOINTS, stream, scheduler> (batched.h:432-452): full-matrix Gotoh traceback with
/// nvBowtie's CIGAR-forming backtracer; the stream is the PackedTracebackStream of the banded form.
template <uint32 CHECKPOINTS, typename stream_type, typename algorithm_type = DeviceThreadScheduler>
struct BatchedAlignmentTraceback
{
    typedef typename stream_type::aligner_type aligner_type;
    static uint64 min_temp_storage(const uint32 max_pattern_len, const uint32 max_text_len, const uint32 stream_size)
    { return nvbio_hip_gotoh_traceback_temp_bytes(max_pattern_len, max_text_len, stream_size); }
    static uint64 max_temp_storage(const uint32 max_pattern_len, const uint32 max_text_len, const uint32 stream_size)
    { return min_temp_storage(max_pattern_len, max_text_len, stream_size); }

    void enact(stream_type stream, uint64 temp_size, uint8* temp, void* hip_stream = nullptr)
    {
        static_assert(sizeof(io::Cigar) == 2, "io::Cigar must be a uint16 bit-field");
        const nvbio_hip_gotoh_scheme sc = priv::abi_scheme(stream.aligner().scheme);
        const nvbio_hip_string_set p = stream.m_patterns.abi(), t = stream.m_texts.abi();
        hip_check(nvbio_hip_gotoh_traceback(&sc, int32(aligner_type::TYPE), &p, &t, stream.max_pattern_length(), stream.max_text_length(), stream.size(),
                      stream.m_alignments.score, stream.m_alignments.sink, stream.m_alignments.source,
                      reinterpret_cast<uint16*>(stream.m_cigars.cigar), stream.m_cigars.cigar_stride, stream.m_cigars.cigar_len,
                      temp, temp_size, hip_stream), "nvbio_hip_gotoh_traceback");
    }
};

/// batch_alignment_score(aligner, patterns, texts, sinks, scheduler, maxP, maxT)   (batched.h:160-190)
template <typename aligner_type, typename pattern_set_type, typename text_set_type, typename scheduler_type>
void batch_alignment_score(
    const aligner_type      aligner,
    const pattern_set_type  patterns,
    const text_set_type     texts,
          BestSinkArrays    sinks,
    const scheduler_type    scheduler,
    const uint32            max_pattern_length,
    const uint32            max_text_length)
{
    (void)scheduler;
    typedef PackedAlignmentStream<aligner_type, pattern_set_type, text_set_type> stream_type;
    stream_type stream(aligner, patterns, texts, sinks, max_pattern_length, max_text_length);
    BatchedAlignmentScore<stream_type, scheduler_type> batch;
    batch.enact(stream);
}

/// batch_banded_alignment_score<BAND_LEN>(aligner, patterns, texts, sinks, scheduler, maxP, maxT)
template <uint32 BAND_LEN, typename aligner_type, typename pattern_set_type, typename text_set_type, typename scheduler_type>
void batch_banded_alignment_score(
    const aligner_type      aligner,
    const pattern_set_type  patterns,
    const text_set_type     texts,
          BestSinkArrays    sinks,
    const scheduler_type    scheduler,
    const uint32            max_pattern_length,
    const uint32            max_text_length)
{
    (void)scheduler;
    typedef PackedAlignmentStream<aligner_type, pattern_set_type, text_set_type> stream_type;
    stream_type stream(aligner, patterns, texts, sinks, max_pattern_length, max_text_length);
    BatchedBandedAlignmentScore<BAND_LEN, stream_type, scheduler_type> batch;
    batch.enact(stream);
}

} // namespace aln
} // namespace nvbio
