// compat/nvbio/alignment/alignment_base.h -- aligner / scoring-scheme / sink value types of nvbio::aln
// (nvbio/alignment/alignment_base.h:54-410, utils.h:88-260, sink.h:44-180), with the reference's names,
// members and host-device qualifiers, for callers compiled against the drop-in template layer.
#pragma once
#include "../basic/types.h"
#include "../basic/vector_view.h"

namespace nvbio {
namespace aln {

enum AlignmentType { GLOBAL = 0, LOCAL = 1, SEMI_GLOBAL = 2 };

/// DP flow directions (alignment_base.h:56-70)
enum DirectionVector { SUBSTITUTION = 0u, INSERTION = 1u, DELETION = 2u, SINK = 3u };

/// an alignment result: score, start cell and terminal cell, x = text, y = pattern (alignment_base.h:122-135)
template <typename ScoreType>
struct Alignment
{
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE Alignment() {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE Alignment(const ScoreType _score, const uint2 _source, const uint2 _sink) : score(_score), source(_source), sink(_sink) {}
    ScoreType score; uint2 source; uint2 sink;
};

struct PatternBlockingTag {};
struct TextBlockingTag {};
/// the Myers bit-vector algorithm for edit distance (alignment_base.h:86-87): an algorithm choice, the scores are the edit-distance aligner's
template <uint32 ALPHABET_SIZE_T> struct MyersTag { static const uint32 ALPHABET_SIZE = ALPHABET_SIZE_T; };
template <typename T> struct transpose_tag {};
template <uint32 N> struct transpose_tag< MyersTag<N> > { typedef MyersTag<N> type; };
template <> struct transpose_tag<PatternBlockingTag> { typedef TextBlockingTag type; };
template <> struct transpose_tag<TextBlockingTag>    { typedef PatternBlockingTag type; };

struct SmithWatermanTag {};
struct GotohTag {};
struct EditDistanceTag {};
struct HammingDistanceTag {};
template <typename aligner_type> struct aligner_tag { typedef typename aligner_type::aligner_tag type; };

// ---------------------------------------------------------------------------------------- scoring schemes
struct SimpleSmithWatermanScheme
{
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE SimpleSmithWatermanScheme() : m_match(0), m_mismatch(0), m_deletion(0), m_insertion(0) {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE SimpleSmithWatermanScheme(const int32 match, const int32 mm, const int32 del, const int32 ins)
        : m_match(match), m_mismatch(mm), m_deletion(del), m_insertion(ins) {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE int32 match(const uint8 = 0) const { return m_match; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE int32 mismatch(const uint8 = 0) const { return m_mismatch; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE int32 mismatch(const uint8, const uint8, const uint8 = 0) const { return m_mismatch; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE int32 deletion() const { return m_deletion; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE int32 insertion() const { return m_insertion; }
    int32 m_match, m_mismatch, m_deletion, m_insertion;
};

struct SimpleGotohScheme
{
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE SimpleGotohScheme() : m_match(0), m_mismatch(0), m_gap_open(0), m_gap_ext(0) {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE SimpleGotohScheme(const int32 match, const int32 mm, const int32 gap_open, const int32 gap_ext)
        : m_match(match), m_mismatch(mm), m_gap_open(gap_open), m_gap_ext(gap_ext) {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE int32 match(const uint8 = 0) const { return m_match; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE int32 mismatch(const uint8 = 0) const { return m_mismatch; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE int32 mismatch(const uint8, const uint8, const uint8 = 0) const { return m_mismatch; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE int32 substitution(const uint32, const uint32, const uint8 r, const uint8 q, const uint8 = 0) const { return q == r ? m_match : m_mismatch; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE int32 pattern_gap_open() const { return m_gap_open; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE int32 pattern_gap_extension() const { return m_gap_ext; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE int32 text_gap_open() const { return m_gap_open; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE int32 text_gap_extension() const { return m_gap_ext; }
    int32 m_match, m_mismatch, m_gap_open, m_gap_ext;
};

/// the scheme the edit-distance aligner runs the Smith-Waterman code with (ed/ed_utils.h:44-51)
struct EditDistanceSWScheme
{
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE int32 match(const uint8 = 0) const { return 0; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE int32 mismatch(const uint8 = 0) const { return -1; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE int32 mismatch(const uint8, const uint8, const uint8 = 0) const { return -1; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE int32 deletion() const { return -1; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE int32 insertion() const { return -1; }
};

// ---------------------------------------------------------------------------------------- aligners
template <AlignmentType T_TYPE, typename AlgorithmType = PatternBlockingTag>
struct EditDistanceAligner
{
    static const AlignmentType TYPE = T_TYPE;
    typedef EditDistanceTag aligner_tag;
    typedef AlgorithmType   algorithm_tag;
};
template <AlignmentType TYPE> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE
EditDistanceAligner<TYPE> make_edit_distance_aligner() { return EditDistanceAligner<TYPE>(); }
template <AlignmentType TYPE, typename algorithm_tag> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE
EditDistanceAligner<TYPE, algorithm_tag> make_edit_distance_aligner() { return EditDistanceAligner<TYPE, algorithm_tag>(); }

template <AlignmentType T_TYPE, typename scoring_scheme_type, typename AlgorithmType = PatternBlockingTag>
struct GotohAligner
{
    static const AlignmentType TYPE = T_TYPE;
    typedef GotohTag            aligner_tag;
    typedef AlgorithmType       algorithm_tag;
    typedef scoring_scheme_type scheme_type;
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE GotohAligner(const scoring_scheme_type _scheme) : scheme(_scheme) {}
    scoring_scheme_type scheme;
};
template <AlignmentType TYPE, typename scoring_scheme_type> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE
GotohAligner<TYPE, scoring_scheme_type> make_gotoh_aligner(const scoring_scheme_type& scheme) { return GotohAligner<TYPE, scoring_scheme_type>(scheme); }
template <AlignmentType TYPE, typename algorithm_tag, typename scoring_scheme_type> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE
GotohAligner<TYPE, scoring_scheme_type, algorithm_tag> make_gotoh_aligner(const scoring_scheme_type& scheme) { return GotohAligner<TYPE, scoring_scheme_type, algorithm_tag>(scheme); }

template <AlignmentType T_TYPE, typename scoring_scheme_type, typename AlgorithmType = PatternBlockingTag>
struct SmithWatermanAligner
{
    static const AlignmentType TYPE = T_TYPE;
    typedef SmithWatermanTag    aligner_tag;
    typedef AlgorithmType       algorithm_tag;
    typedef scoring_scheme_type scheme_type;
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE SmithWatermanAligner(const scoring_scheme_type _scheme) : scheme(_scheme) {}
    scoring_scheme_type scheme;
};
template <AlignmentType TYPE, typename scoring_scheme_type> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE
SmithWatermanAligner<TYPE, scoring_scheme_type> make_smith_waterman_aligner(const scoring_scheme_type& scheme) { return SmithWatermanAligner<TYPE, scoring_scheme_type>(scheme); }
template <AlignmentType TYPE, typename algorithm_tag, typename scoring_scheme_type> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE
SmithWatermanAligner<TYPE, scoring_scheme_type, algorithm_tag> make_smith_waterman_aligner(const scoring_scheme_type& scheme) { return SmithWatermanAligner<TYPE, scoring_scheme_type, algorithm_tag>(scheme); }

/// ungapped alignment (alignment_base.h:365-393): a value type here -- nvBowtie's scheme names it, no batch function of the hot path takes it
template <AlignmentType T_TYPE, typename scoring_scheme_type, typename AlgorithmType = PatternBlockingTag>
struct HammingDistanceAligner
{
    static const AlignmentType TYPE = T_TYPE;
    typedef HammingDistanceTag aligner_tag;
    typedef AlgorithmType      algorithm_tag;
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE HammingDistanceAligner(const scoring_scheme_type _scheme) : scheme(_scheme) {}
    scoring_scheme_type scheme;
};
template <AlignmentType TYPE, typename scoring_scheme_type> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE
HammingDistanceAligner<TYPE, scoring_scheme_type> make_hamming_distance_aligner(const scoring_scheme_type& scheme) { return HammingDistanceAligner<TYPE, scoring_scheme_type>(scheme); }
template <AlignmentType TYPE, typename scoring_scheme_type, typename algorithm_tag> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE
HammingDistanceAligner<TYPE, scoring_scheme_type, algorithm_tag> make_hamming_distance_aligner(const scoring_scheme_type& scheme) { return HammingDistanceAligner<TYPE, scoring_scheme_type, algorithm_tag>(scheme); }

namespace priv { typedef nvbio::aln::EditDistanceSWScheme EditDistanceSWScheme; }       // the reference keeps it in priv (ed/ed_utils.h)

/// the cell type of the boundary column a full-matrix score pass is handed (alignment/utils.h:58-64)
template <typename aligner_type> struct column_storage_type { typedef null_type type; };
template <AlignmentType T, typename A>             struct column_storage_type< EditDistanceAligner<T, A> >       { typedef int16 type; };
template <AlignmentType T, typename S, typename A> struct column_storage_type< HammingDistanceAligner<T, S, A> > { typedef int16 type; };
template <AlignmentType T, typename S, typename A> struct column_storage_type< SmithWatermanAligner<T, S, A> >   { typedef int16 type; };
#if defined(__HIPCC__)
template <AlignmentType T, typename S, typename A> struct column_storage_type< GotohAligner<T, S, A> >           { typedef short2 type; };
#else
template <AlignmentType T, typename S, typename A> struct column_storage_type< GotohAligner<T, S, A> >           { struct type { int16 x, y; }; };
#endif

/// max_pattern_gaps / max_text_gaps(aligner, min_score, pattern_len) (alignment/utils.h:136-222, utils_inl.h:36-200): how many gap
/// symbols an alignment of a pattern_len pattern can hold and still reach min_score -- nvBowtie sizes the opposite-mate window
/// with it (score_opposite_inl.h:139).  Start from the all-match score (match(30) per symbol), pay the opening once (affine
/// gaps), then count extensions while the score holds; the count is capped at pattern_len and the result is "steps - 1", which
/// wraps to 0xFFFFFFFF when the opening alone already sinks the score (kept: callers add it to a length as a signed value).
namespace priv {
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 affordable_gaps(int32 score, const int32 min_score, const int32 open, const int32 step, const int32 pattern_len)
{
    if (score < min_score) return 0u;
    score += open;
    uint32 n = 0;
    for (; score >= min_score && n < uint32(pattern_len); ++n) score += step;
    return n - 1u;
}
} // namespace priv
template <AlignmentType T, typename A> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE
uint32 max_pattern_gaps(const EditDistanceAligner<T, A>&, const int32 min_score, const int32) { return uint32(-min_score); }
template <AlignmentType T, typename A> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE
uint32 max_text_gaps(const EditDistanceAligner<T, A>&, const int32 min_score, const int32) { return uint32(-min_score); }
template <AlignmentType T, typename S, typename A> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE
uint32 max_pattern_gaps(const HammingDistanceAligner<T, S, A>&, const int32, const int32) { return 0u; }
template <AlignmentType T, typename S, typename A> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE
uint32 max_text_gaps(const HammingDistanceAligner<T, S, A>&, const int32, const int32) { return 0u; }
template <AlignmentType T, typename S, typename A> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE
uint32 max_pattern_gaps(const SmithWatermanAligner<T, S, A>& al, const int32 min_score, const int32 pattern_len)
{ return priv::affordable_gaps(pattern_len * al.scheme.match(30), min_score, 0, al.scheme.deletion(), pattern_len); }
template <AlignmentType T, typename S, typename A> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE
uint32 max_text_gaps(const SmithWatermanAligner<T, S, A>& al, const int32 min_score, const int32 pattern_len)
{ return priv::affordable_gaps(pattern_len * al.scheme.match(30), min_score, 0, al.scheme.insertion(), pattern_len); }
template <AlignmentType T, typename S, typename A> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE
uint32 max_pattern_gaps(const GotohAligner<T, S, A>& al, const int32 min_score, const int32 pattern_len)
{ return priv::affordable_gaps(pattern_len * al.scheme.match(30), min_score, al.scheme.pattern_gap_open(), al.scheme.pattern_gap_extension(), pattern_len); }
template <AlignmentType T, typename S, typename A> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE
uint32 max_text_gaps(const GotohAligner<T, S, A>& al, const int32 min_score, const int32 pattern_len)
{ return priv::affordable_gaps(pattern_len * al.scheme.match(30), min_score, al.scheme.text_gap_open(), al.scheme.text_gap_extension(), pattern_len); }

template <typename T> struct transpose_aligner {};
template <AlignmentType T, typename S, typename A> struct transpose_aligner< HammingDistanceAligner<T, S, A> > { typedef HammingDistanceAligner<T, S, typename transpose_tag<A>::type> type; };
template <AlignmentType T, typename A> struct transpose_aligner< EditDistanceAligner<T, A> > { typedef EditDistanceAligner<T, typename transpose_tag<A>::type> type; };
template <AlignmentType T, typename S, typename A> struct transpose_aligner< GotohAligner<T, S, A> > { typedef GotohAligner<T, S, typename transpose_tag<A>::type> type; };
template <AlignmentType T, typename S, typename A> struct transpose_aligner< SmithWatermanAligner<T, S, A> > { typedef SmithWatermanAligner<T, S, typename transpose_tag<A>::type> type; };

// ---------------------------------------------------------------------------------------- sinks
/// a valid alignment: score + end cell (text position, pattern position), both 1-based ends of the path
template <typename ScoreType>
struct BestSink
{
    typedef ScoreType score_type;
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE BestSink() : score(Field_traits<ScoreType>::min()), sink(make_uint2(uint32(-1), uint32(-1))) {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE void invalidate() { score = Field_traits<ScoreType>::min(); sink = make_uint2(uint32(-1), uint32(-1)); }
    /// `<=`: of several cells with the best score the one reported last wins (sink_inl.h:57-68)
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE void report(const ScoreType _score, const uint2 _sink) { if (score <= _score) { score = _score; sink = _sink; } }
    ScoreType score;
    uint2     sink;
};

/// best two alignments at least `distinct_dist` text positions apart (sink_inl.h:70-130)
template <typename ScoreType>
struct Best2Sink
{
    typedef ScoreType score_type;
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE Best2Sink(const uint32 distinct_dist = 0)
        : score1(Field_traits<ScoreType>::min()), score2(Field_traits<ScoreType>::min()),
          sink1(make_uint2(uint32(-1), uint32(-1))), sink2(make_uint2(uint32(-1), uint32(-1))), m_distinct_dist(distinct_dist) {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE void invalidate()
    { score1 = score2 = Field_traits<ScoreType>::min(); sink1 = sink2 = make_uint2(uint32(-1), uint32(-1)); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE void report(const ScoreType score, const uint2 sink)
    {
        if (score1 <= score) { score1 = score; sink1 = sink; }
        else if (score2 <= score && (sink.x + m_distinct_dist < sink1.x || sink.x > sink1.x + m_distinct_dist)) { score2 = score; sink2 = sink; }
    }
    ScoreType score1, score2;
    uint2     sink1, sink2;
    uint32    m_distinct_dist;
};

struct NullSink { NVBIO_FORCEINLINE NVBIO_HOST_DEVICE void invalidate() {} template <typename S> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE void report(const S, const uint2) {} };

/// an all-zero quality string that is its own iterator (utils.h:228-260)
struct trivial_quality_string
{
    static const uint32 SYMBOL_SIZE = 8u;
    typedef std::random_access_iterator_tag  iterator_category;
    typedef uint8                            value_type;
    typedef uint8                            reference;
    typedef const uint8*                     pointer;
    typedef int32                            difference_type;
    typedef uint32                           index_type;
    typedef trivial_quality_string           iterator;
    typedef trivial_quality_string           const_iterator;
    typedef trivial_quality_string           forward_iterator;
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint8 operator[](const index_type) const { return 0u; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint8 operator*() const { return 0u; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE trivial_quality_string& operator++() { return *this; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE trivial_quality_string  operator++(int) { return *this; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE trivial_quality_string& operator--() { return *this; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE trivial_quality_string  operator--(int) { return *this; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE trivial_quality_string  operator+(const difference_type) const { return *this; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE trivial_quality_string  operator-(const difference_type) const { return *this; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE iterator begin() const { return *this; }
};
/// a set of them (utils.h:262-280)
struct trivial_quality_string_set
{
    typedef trivial_quality_string string_type;
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE trivial_quality_string operator[](const uint32) const { return trivial_quality_string(); }
};

} // namespace aln
} // namespace nvbio
