// tests/compat/nvbowtie_streams.hip -- a caller written to the shape of nvBowtie's own alignment streams:
//   * the strings container of nvBowtie/bowtie2/cuda/alignment_utils.h:170-218 (AlignmentStrings: a ReadLoader pattern viewed
//     REVERSE / STANDARD or FORWARD / COMPLEMENT by the hit's strand, pattern.qualities(), a PackedStringLoader genome window,
//     all with the lmem cache tag),
//   * the stream base of :257-340 (context {idx, mate, read_range, read_id, read_rc, genome_begin, genome_end, min_score} + sink or
//     backtracer; load_strings() = strings->load(pipeline, context)),
//   * a score stream in the shape of score_best_inl.h:54-148 (window = [loc - band/2, + band + read_len) clamped to the genome,
//     min_score = max(second best, score_limit), output = hit.score / hit.sink),
//   * an opposite-mate score stream over the full matrix (score_opposite_inl.h:54-260) and a traceback stream with nvBowtie's
//     CIGAR-forming Backtracker (alignment_utils.h:125-168, traceback_inl.h:53-189),
//   * a scoring scheme in the shape of scoring.h:206-356 (QualCost mismatch penalties, constant match bonus, separate read /
//     reference gap costs).
// It includes the reference's header names and is compiled with `hipcc -I include/nvbio_hip/compat`; nothing in these classes
// knows about this build.  The extern "C" entry points at the bottom exist so that the Python tests can drive it; they report
// which execution the batch objects chose (nvBowtie's streams must run "tuned").
#include <nvbio/basic/types.h>
#include <nvbio/basic/packedstream.h>
#include <nvbio/basic/packedstream_loader.h>
#include <nvbio/basic/vector_view.h>
#include <nvbio/basic/cuda/ldg.h>
#include <nvbio/io/utils.h>
#include <nvbio/io/sequence/sequence.h>
#include <nvbio/alignment/alignment.h>
#include <nvbio/alignment/batched.h>
#include <string.h>

using namespace nvbio;

namespace bt2 {

using namespace nvbio::io;

// ---------------------------------------------------------------------------------------------------------------------
// the scheme (shape of scoring.h:86-125, 206-356)
// ---------------------------------------------------------------------------------------------------------------------
template <typename T> struct QualCost
{
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE QualCost() : m_min_val(0), m_max_val(0) {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE QualCost(const T min_val, const T max_val) : m_min_val(min_val), m_max_val(max_val) {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE T operator()(const int i) const
    {
        const float frac = (float)(nvbio::min(i, 40) / 40.0f);
        return m_min_val + T(frac * (m_max_val - m_min_val));
    }
    T m_min_val, m_max_val;
};
template <typename T> struct ConstantCost
{
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE ConstantCost() : m_val(0) {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE ConstantCost(const T, const T max_val) : m_val(max_val) {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE T operator()(const int) const { return T(m_val); }
    T m_val;
};

template <typename MMCost = QualCost<int>, typename NCost = ConstantCost<int> >
struct SmithWatermanScoringScheme
{
    typedef SmithWatermanScoringScheme<MMCost, NCost>        scheme_type;
    typedef aln::GotohAligner<aln::LOCAL, scheme_type>       local_aligner_type;
    typedef aln::GotohAligner<aln::SEMI_GLOBAL, scheme_type> end_to_end_aligner_type;
    typedef ConstantCost<int> MatchCost;
    typedef MMCost            MismatchCost;
    typedef MatchCost         match_cost_function;
    typedef MismatchCost      mismatch_cost_function;
    typedef NCost             N_cost_function;
    static const int32 inf_score   = -(1 << 16);
    static const int32 worst_score = inf_score;

    local_aligner_type      local_aligner()      const { return local_aligner_type(*this); }
    end_to_end_aligner_type end_to_end_aligner() const { return end_to_end_aligner_type(*this); }

    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE int32 match(const uint8 q = 0)    const { return  m_match(q); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE int32 mismatch(const uint8 q = 0) const { return -m_mmp(q); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE int32 mismatch(const uint8, const uint8, const uint8 qq = 0) const { return -m_mmp(qq); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE int32 substitution(const uint32, const uint32, const uint8 r, const uint8 q, const uint8 qq = 0) const { return r == q ? m_match(qq) : -m_mmp(qq); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE int32 pattern_gap_open()      const { return -m_read_gap_const - m_read_gap_coeff; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE int32 pattern_gap_extension() const { return -m_read_gap_coeff; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE int32 text_gap_open()         const { return -m_ref_gap_const - m_ref_gap_coeff; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE int32 text_gap_extension()    const { return -m_ref_gap_coeff; }

    int       m_read_gap_const, m_read_gap_coeff, m_ref_gap_const, m_ref_gap_coeff;
    MatchCost m_match;
    MMCost    m_mmp;
    NCost     m_np;
};

// ---------------------------------------------------------------------------------------------------------------------
// the batch of reads (the accessors io::SequenceDataAccess offers) and the pipeline state
// ---------------------------------------------------------------------------------------------------------------------
struct ReadBatch
{
    static const uint32 SEQUENCE_BITS       = io::SequenceDataTraits<DNA_N>::SEQUENCE_BITS;
    static const bool   SEQUENCE_BIG_ENDIAN = io::SequenceDataTraits<DNA_N>::SEQUENCE_BIG_ENDIAN;
    typedef cuda::ldg_pointer<uint32>                                                       sequence_storage_iterator;
    typedef cuda::ldg_pointer<uint8>                                                        qual_storage_iterator;
    typedef PackedStream<sequence_storage_iterator, uint8, SEQUENCE_BITS, SEQUENCE_BIG_ENDIAN> sequence_stream_type;

    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE sequence_stream_type  sequence_stream() const { return sequence_stream_type(sequence_storage_iterator(words)); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE qual_storage_iterator qual_stream()     const { return qual_storage_iterator(quals); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint2  get_range(const uint32 i) const { return make_uint2(index[i], index[i + 1]); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 max_read_len() const { return longest; }
    const uint32* words; const uint8* quals; const uint32* index; uint32 longest;
};

struct Hit { uint32 read_id, loc, rc; };

template <typename scheme_t>
struct Pipeline
{
    typedef scheme_t                                                   scheme_type;
    typedef ReadBatch                                                  read_batch_type;
    typedef PackedStream<cuda::ldg_pointer<uint32>, uint8, 2, true>    genome_iterator;
    read_batch_type  reads, reads_o;
    genome_iterator  genome;
    uint32           genome_length;
    const uint32*    idx_queue;
    const Hit*       hits;
    uint32           hits_queue_size;
    const int32*     second_best;         // per read
    int32            score_limit;
    int32*           hit_score; uint32* hit_sink; int32* raw_score; uint2* raw_sink;      // outputs
    // opposite-mate windows
    const uint2*     o_windows;
    // tracebacks
    uint16*          cigar; uint32 cigar_stride; uint32* cigar_len; int32* aln_score; uint2* aln_source; uint2* aln_sink;
};

enum AlignmentStreamType { SCORE_STREAM = 0, OPPOSITE_SCORE_STREAM = 1, TRACEBACK_STREAM = 2 };

/// the CIGAR op as nvBowtie stores it (nvbio/io/alignments.h:57-75)
struct Cigar
{
    enum Type { SUBSTITUTION = 0, INSERTION = 1, DELETION = 2, SOFT_CLIPPING = 3 };
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE Cigar() {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE Cigar(const uint8 type, const uint16 len) : m_type(type), m_len(len) {}
    uint16 m_type:2, m_len:14;
};

/// forms a CIGAR on the go, stored backwards (alignment_utils.h:125-168)
template <typename vector>
struct Backtracker
{
    NVBIO_FORCEINLINE NVBIO_DEVICE Backtracker(vector vec, const uint32 _capacity) : out(vec), size(0), prev(255), capacity(_capacity) {}
    NVBIO_FORCEINLINE NVBIO_DEVICE void clip(const uint32 l) { if (l) out[size++] = Cigar(Cigar::SOFT_CLIPPING, l); }
    NVBIO_FORCEINLINE NVBIO_DEVICE void push(uint8 type)
    {
        if (prev == type) out[size - 1u].m_len++;
        else { out[size++] = Cigar(type, 1u); prev = type; }
    }
    vector out; uint32 size; uint8 prev; uint32 capacity;
};

// ---------------------------------------------------------------------------------------------------------------------
// strings + stream base (shape of alignment_utils.h:170-340)
// ---------------------------------------------------------------------------------------------------------------------
template <typename AlignerType, typename PipelineType>
struct AlignmentStrings
{
    typedef typename PipelineType::genome_iterator   genome_iterator;
    typedef typename PipelineType::read_batch_type   read_batch_type;
    typedef typename PipelineType::scheme_type       scheme_type;
    typedef AlignerType                              aligner_type;
    static const uint32 CACHE_SIZE = 64;
    typedef nvbio::lmem_cache_tag<CACHE_SIZE>        lmem_cache_type;

    typedef ReadLoader<read_batch_type, lmem_cache_type>        pattern_loader_type;
    typedef typename pattern_loader_type::string_type           pattern_string;
    typedef typename pattern_string::qual_string_type           qual_string;
    typedef PackedStringLoader<typename genome_iterator::storage_iterator, genome_iterator::SYMBOL_SIZE, genome_iterator::BIG_ENDIAN, lmem_cache_type> text_loader_type;
    typedef typename text_loader_type::iterator                 text_iterator;
    typedef vector_view<text_iterator>                          text_string;

    template <typename context_type>
    NVBIO_HOST_DEVICE void load(const PipelineType& pipeline, const context_type* context)
    {
        read_batch_type reads = context->mate ? pipeline.reads_o : pipeline.reads;
        const DirType  read_dir  = context->read_rc ? FORWARD    : REVERSE;       // the reads are stored reversed
        const ReadType read_type = context->read_rc ? COMPLEMENT : STANDARD;
        pattern = pattern_loader.load(reads, context->read_range, read_dir, read_type);
        quals   = pattern.qualities();
        text    = text_string(context->genome_end - context->genome_begin,
                              text_loader.load(pipeline.genome + context->genome_begin, context->genome_end - context->genome_begin));
    }
    pattern_loader_type pattern_loader;
    text_loader_type    text_loader;
    pattern_string      pattern;
    qual_string         quals;
    text_string         text;
};

template <AlignmentStreamType TYPE> struct AlignmentStreamContext {};
template <> struct AlignmentStreamContext<SCORE_STREAM>          { aln::BestSink<int32> sink; };
template <> struct AlignmentStreamContext<OPPOSITE_SCORE_STREAM> { aln::BestSink<int32> sink; };
template <> struct AlignmentStreamContext<TRACEBACK_STREAM>
{
    Cigar                   cigar[1024];
    Backtracker<Cigar*>     backtracer;
    aln::Alignment<int32>   alignment;
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE AlignmentStreamContext() : backtracer(cigar, 1024u) {}
};

template <AlignmentStreamType TYPE, typename AlignerType, typename PipelineType>
struct AlignmentStreamBase
{
    typedef AlignmentStrings<AlignerType, PipelineType> strings_type;
    typedef typename PipelineType::scheme_type          scheme_type;
    typedef AlignerType                                 aligner_type;
    struct context_type : AlignmentStreamContext<TYPE>
    {
        uint32 idx, mate; uint2 read_range; uint32 read_id, read_rc, genome_begin, genome_end; int32 min_score;
    };
    AlignmentStreamBase(const PipelineType _pipeline, const aligner_type _aligner) : m_pipeline(_pipeline), m_aligner(_aligner) {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE const aligner_type& aligner() const { return m_aligner; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 pattern_length(const uint32, const context_type* context) const { return context->read_range.y - context->read_range.x; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 text_length(const uint32, const context_type* context) const { return context->genome_end - context->genome_begin; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE void load_strings(const uint32, const uint32, const uint32, const context_type* context, strings_type* strings) const
    { strings->load(m_pipeline, context); }
    PipelineType m_pipeline;
    aligner_type m_aligner;
};

// ---------------------------------------------------------------------------------------------------------------------
// the three streams
// ---------------------------------------------------------------------------------------------------------------------
template <typename AlignerType, typename PipelineType>
struct BestScoreStream : public AlignmentStreamBase<SCORE_STREAM, AlignerType, PipelineType>
{
    typedef AlignmentStreamBase<SCORE_STREAM, AlignerType, PipelineType> base_type;
    typedef typename base_type::context_type context_type;
    typedef typename base_type::scheme_type  scheme_type;
    BestScoreStream(const uint32 _band_len, const PipelineType _pipeline, const AlignerType _aligner) : base_type(_pipeline, _aligner), m_band_len(_band_len) {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 max_pattern_length() const { return base_type::m_pipeline.reads.max_read_len(); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 max_text_length() const { return base_type::m_pipeline.reads.max_read_len() + m_band_len; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 size() const { return base_type::m_pipeline.hits_queue_size; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool init_context(const uint32 i, context_type* context) const
    {
        context->idx = base_type::m_pipeline.idx_queue[i];
        const Hit hit = base_type::m_pipeline.hits[context->idx];
        context->mate       = 0u;
        context->read_rc    = hit.rc;
        context->read_id    = hit.read_id;
        context->read_range = base_type::m_pipeline.reads.get_range(context->read_id);
        const uint32 g_pos = hit.loc;
        const uint32 read_len = context->read_range.y - context->read_range.x;
        context->genome_begin = g_pos > m_band_len / 2 ? g_pos - m_band_len / 2 : 0u;
        context->genome_end   = nvbio::min(context->genome_begin + m_band_len + read_len, base_type::m_pipeline.genome_length);
        context->sink = aln::BestSink<int32>();
        context->min_score = nvbio::max(base_type::m_pipeline.second_best[context->read_id], base_type::m_pipeline.score_limit);
        return true;
    }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE void output(const uint32, const context_type* context) const
    {
        const aln::BestSink<int32> sink = context->sink;
        base_type::m_pipeline.hit_score[context->idx] = nvbio::max(sink.score, scheme_type::worst_score);
        base_type::m_pipeline.hit_sink[context->idx]  = context->genome_begin + sink.sink.x;
        base_type::m_pipeline.raw_score[context->idx] = sink.score;
        base_type::m_pipeline.raw_sink[context->idx]  = sink.sink;
    }
    const uint32 m_band_len;
};

/// the opposite mate scored over a whole window with the full DP (the mate batch, its strand and window given per hit)
template <typename AlignerType, typename PipelineType>
struct OppositeScoreStream : public AlignmentStreamBase<OPPOSITE_SCORE_STREAM, AlignerType, PipelineType>
{
    typedef AlignmentStreamBase<OPPOSITE_SCORE_STREAM, AlignerType, PipelineType> base_type;
    typedef typename base_type::context_type context_type;
    OppositeScoreStream(const uint32 _max_window, const PipelineType _pipeline, const AlignerType _aligner) : base_type(_pipeline, _aligner), m_max_window(_max_window) {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 max_pattern_length() const { return base_type::m_pipeline.reads_o.max_read_len(); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 max_text_length() const { return m_max_window; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 size() const { return base_type::m_pipeline.hits_queue_size; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool init_context(const uint32 i, context_type* context) const
    {
        context->idx = base_type::m_pipeline.idx_queue[i];
        const Hit hit = base_type::m_pipeline.hits[context->idx];
        context->mate       = 1u;
        context->read_rc    = !hit.rc;
        context->read_id    = hit.read_id;
        context->read_range = base_type::m_pipeline.reads_o.get_range(context->read_id);
        context->genome_begin = base_type::m_pipeline.o_windows[context->idx].x;
        context->genome_end   = base_type::m_pipeline.o_windows[context->idx].y;
        context->sink = aln::BestSink<int32>();
        context->min_score = base_type::m_pipeline.score_limit;
        return context->genome_end > context->genome_begin;          // hits without a window are skipped, as score_opposite_inl.h:121-160 does
    }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE void output(const uint32, const context_type* context) const
    {
        base_type::m_pipeline.raw_score[context->idx] = context->sink.score;
        base_type::m_pipeline.raw_sink[context->idx]  = context->sink.sink;
    }
    const uint32 m_max_window;
};

/// tracebacks of the hits' alignments with the CIGAR-forming backtracer (traceback_inl.h:53-189); FULL: opposite-mate windows
template <bool FULL, typename AlignerType, typename PipelineType>
struct TracebackStream : public AlignmentStreamBase<TRACEBACK_STREAM, AlignerType, PipelineType>
{
    typedef AlignmentStreamBase<TRACEBACK_STREAM, AlignerType, PipelineType> base_type;
    typedef typename base_type::context_type context_type;
    TracebackStream(const uint32 _band_len, const uint32 _max_window, const PipelineType _pipeline, const AlignerType _aligner)
        : base_type(_pipeline, _aligner), m_band_len(_band_len), m_max_window(_max_window) {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 max_pattern_length() const { return FULL ? base_type::m_pipeline.reads_o.max_read_len() : base_type::m_pipeline.reads.max_read_len(); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 max_text_length() const { return FULL ? m_max_window : base_type::m_pipeline.reads.max_read_len() + m_band_len; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 size() const { return base_type::m_pipeline.hits_queue_size; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool init_context(const uint32 i, context_type* context) const
    {
        context->idx = base_type::m_pipeline.idx_queue[i];
        const Hit hit = base_type::m_pipeline.hits[context->idx];
        context->mate       = FULL ? 1u : 0u;
        context->read_rc    = FULL ? !hit.rc : hit.rc;
        context->read_id    = hit.read_id;
        context->read_range = FULL ? base_type::m_pipeline.reads_o.get_range(context->read_id) : base_type::m_pipeline.reads.get_range(context->read_id);
        if (FULL)
        {
            context->genome_begin = base_type::m_pipeline.o_windows[context->idx].x;
            context->genome_end   = base_type::m_pipeline.o_windows[context->idx].y;
        }
        else
        {
            const uint32 read_len = context->read_range.y - context->read_range.x;
            context->genome_begin = hit.loc > m_band_len / 2 ? hit.loc - m_band_len / 2 : 0u;
            context->genome_end   = nvbio::min(context->genome_begin + m_band_len + read_len, base_type::m_pipeline.genome_length);
        }
        context->min_score = base_type::m_pipeline.score_limit;
        return true;
    }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE void output(const uint32, const context_type* context) const
    {
        const PipelineType& p = base_type::m_pipeline;
        p.aln_score[context->idx]  = context->alignment.score;
        p.aln_source[context->idx] = context->alignment.source;
        p.aln_sink[context->idx]   = context->alignment.sink;
        const uint32 size = context->backtracer.size;
        p.cigar_len[context->idx] = size;
        for (uint32 k = 0; k < size && k < p.cigar_stride; ++k)
            p.cigar[uint64(context->idx) * p.cigar_stride + k] = uint16(context->cigar[k].m_type | (context->cigar[k].m_len << 2));
    }
    const uint32 m_band_len, m_max_window;
};

typedef SmithWatermanScoringScheme<>   scheme_type;
typedef Pipeline<scheme_type>          pipeline_type;

} // namespace bt2

// ---------------------------------------------------------------------------------------------------------------------
// C entry points for the tests
// ---------------------------------------------------------------------------------------------------------------------
#define API extern "C" __attribute__((visibility("default")))

struct bt2_args
{
    // scheme: read gap const / coeff, ref gap const / coeff, match bonus, mismatch penalty min / max
    int32_t rdg_c, rdg_k, rfg_c, rfg_k, match, mmp_min, mmp_max;
    int32_t local;                     // 1 = LOCAL, 0 = SEMI_GLOBAL (end-to-end)
    uint32_t band_len;                 // 15 or 31
    const uint32_t* read_words; const uint8_t* read_quals; const uint32_t* read_index; uint32_t longest;
    const uint32_t* mate_words; const uint8_t* mate_quals; const uint32_t* mate_index; uint32_t mate_longest;
    const uint32_t* genome_words; uint32_t genome_length;
    const uint32_t* idx_queue; const void* hits; uint32_t n_hits;
    const int32_t* second_best; int32_t score_limit;
    int32_t* hit_score; uint32_t* hit_sink; int32_t* raw_score; uint32_t* raw_sink;
    const uint32_t* o_windows; uint32_t max_window;
    uint16_t* cigar; uint32_t cigar_stride; uint32_t* cigar_len; int32_t* aln_score; uint32_t* aln_source; uint32_t* aln_sink;
};

static bt2::pipeline_type make_pipeline(const bt2_args* a)
{
    bt2::pipeline_type p;
    p.reads.words = a->read_words; p.reads.quals = a->read_quals; p.reads.index = a->read_index; p.reads.longest = a->longest;
    p.reads_o.words = a->mate_words; p.reads_o.quals = a->mate_quals; p.reads_o.index = a->mate_index; p.reads_o.longest = a->mate_longest;
    p.genome = bt2::pipeline_type::genome_iterator(cuda::ldg_pointer<uint32>(a->genome_words));
    p.genome_length = a->genome_length;
    p.idx_queue = a->idx_queue; p.hits = (const bt2::Hit*)a->hits; p.hits_queue_size = a->n_hits;
    p.second_best = a->second_best; p.score_limit = a->score_limit;
    p.hit_score = a->hit_score; p.hit_sink = a->hit_sink; p.raw_score = a->raw_score; p.raw_sink = (uint2*)a->raw_sink;
    p.o_windows = (const uint2*)a->o_windows;
    p.cigar = a->cigar; p.cigar_stride = a->cigar_stride; p.cigar_len = a->cigar_len; p.aln_score = a->aln_score;
    p.aln_source = (uint2*)a->aln_source; p.aln_sink = (uint2*)a->aln_sink;
    return p;
}
static bt2::scheme_type make_scheme(const bt2_args* a)
{
    bt2::scheme_type s;
    s.m_read_gap_const = a->rdg_c; s.m_read_gap_coeff = a->rdg_k; s.m_ref_gap_const = a->rfg_c; s.m_ref_gap_coeff = a->rfg_k;
    s.m_match = bt2::ConstantCost<int>(0, a->match); s.m_mmp = bt2::QualCost<int>(a->mmp_min, a->mmp_max); s.m_np = bt2::ConstantCost<int>(0, 1);
    return s;
}
static void copy_path(char* out, const char* path) { strncpy(out, path, 15); out[15] = 0; }

template <typename aligner_type>
static int run_banded_score(const bt2_args* a, const aligner_type aligner, char* path)
{
    typedef bt2::BestScoreStream<aligner_type, bt2::pipeline_type> stream_type;
    static_assert(aln::priv::recognised<stream_type>::value && aln::priv::recognised<stream_type>::staged && aln::priv::recognised<stream_type>::stage_quals,
                  "nvBowtie's score stream must be recognised (staged patterns and qualities)");
    const bt2::pipeline_type p = make_pipeline(a);
    if (a->band_len < 16) { aln::BatchedBandedAlignmentScore<15u, stream_type, aln::DeviceThreadScheduler> batch; batch.enact(stream_type(15u, p, aligner), 0, NULL); copy_path(path, batch.last_path()); }
    else                  { aln::BatchedBandedAlignmentScore<31u, stream_type, aln::DeviceThreadScheduler> batch; batch.enact(stream_type(31u, p, aligner), 0, NULL); copy_path(path, batch.last_path()); }
    return int(hipDeviceSynchronize());
}
API int bt2_banded_score(const bt2_args* a, char* path)
{
    try {
        const bt2::scheme_type s = make_scheme(a);
        return a->local ? run_banded_score(a, s.local_aligner(), path) : run_banded_score(a, s.end_to_end_aligner(), path);
    } catch (const std::exception& e) { fprintf(stderr, "bt2_banded_score: %s\n", e.what()); return -1; }
}
/// the same stream forced down the generic lane (what a build without the tuned recognition would run): for timing it beside the tuned one
API int bt2_banded_score_generic(const bt2_args* a)
{
    typedef bt2::scheme_type::local_aligner_type aligner_type;
    typedef bt2::BestScoreStream<aligner_type, bt2::pipeline_type> stream_type;
    const bt2::pipeline_type p = make_pipeline(a);
    const stream_type stream(15u, p, make_scheme(a).local_aligner());
    hipLaunchKernelGGL((aln::priv::batched_banded_score_kernel<15u, stream_type>), dim3((a->n_hits + 127u) / 128u), dim3(128), 0, 0, stream);
    return int(hipDeviceSynchronize());
}
template <typename aligner_type>
static int run_full_score(const bt2_args* a, const aligner_type aligner, char* path)
{
    typedef bt2::OppositeScoreStream<aligner_type, bt2::pipeline_type> stream_type;
    static_assert(aln::priv::recognised<stream_type>::staged, "nvBowtie's opposite-mate stream must be recognised");
    aln::BatchedAlignmentScore<stream_type, aln::DeviceThreadScheduler> batch;
    batch.enact(stream_type(a->max_window, make_pipeline(a), aligner), 0, NULL);
    copy_path(path, batch.last_path());
    return int(hipDeviceSynchronize());
}
API int bt2_opposite_score(const bt2_args* a, char* path)
{
    try {
        const bt2::scheme_type s = make_scheme(a);
        return a->local ? run_full_score(a, s.local_aligner(), path) : run_full_score(a, s.end_to_end_aligner(), path);
    } catch (const std::exception& e) { fprintf(stderr, "bt2_opposite_score: %s\n", e.what()); return -1; }
}
template <typename aligner_type>
static int run_traceback(const bt2_args* a, const aligner_type aligner, const bool full, char* path)
{
    const bt2::pipeline_type p = make_pipeline(a);
    if (full)
    {
        typedef bt2::TracebackStream<true, aligner_type, bt2::pipeline_type> stream_type;
        static_assert(aln::priv::recognised_tb<stream_type>::staged, "nvBowtie's traceback stream must be recognised");
        aln::BatchedAlignmentTraceback<1024u, stream_type> batch;
        batch.enact(stream_type(0u, a->max_window, p, aligner), 0, NULL);
        copy_path(path, batch.last_path());
    }
    else
    {
        typedef bt2::TracebackStream<false, aligner_type, bt2::pipeline_type> stream_type;
        static_assert(aln::priv::recognised_tb<stream_type>::staged, "nvBowtie's traceback stream must be recognised");
        if (a->band_len < 16) { aln::BatchedBandedAlignmentTraceback<15u, 1024u, stream_type> batch; batch.enact(stream_type(15u, 0u, p, aligner), 0, NULL); copy_path(path, batch.last_path()); }
        else                  { aln::BatchedBandedAlignmentTraceback<31u, 1024u, stream_type> batch; batch.enact(stream_type(31u, 0u, p, aligner), 0, NULL); copy_path(path, batch.last_path()); }
    }
    return int(hipDeviceSynchronize());
}
API int bt2_traceback(const bt2_args* a, int full, char* path)
{
    try {
        const bt2::scheme_type s = make_scheme(a);
        return a->local ? run_traceback(a, s.local_aligner(), full != 0, path) : run_traceback(a, s.end_to_end_aligner(), full != 0, path);
    } catch (const std::exception& e) { fprintf(stderr, "bt2_traceback: %s\n", e.what()); return -1; }
}
