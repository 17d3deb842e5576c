// tests/compat/nvbowtie_streams.hip -- a GPU-side client of the drop-in template layer that scores and traces back VIEWS of stored
// reads, the concept nvBowtie's extension streams are built on.  (That the reference's own stream classes -- verbatim -- bind and are
// routed to the tuned kernels is proved at compile time by tools/ref_bind_check.py in the build container; this file exists so that
// the same ROUTES run on the GPU box, against the oracle, from a caller that owns nothing but the library's documented concepts.)
//
// The concepts it relies on, all from the layer's public headers:
//   * a read batch type with the members io::ReadLoader asks for (sequence_stream(), qual_stream(), the two storage iterator
//     typedefs, SEQUENCE_BITS / SEQUENCE_BIG_ENDIAN);  ReadLoader::load(batch, range, direction, complement) -> io::ReadStream;
//     ReadStream::qualities() -> the matching quality string;
//   * a text that is a vector_view over a PackedStringLoader window of a 2-bit genome;
//   * a scoring scheme with match / mismatch(q) / substitution / four gap accessors and the two cost-function typedefs;
//   * the batch stream concept (batched.h:239-296): aligner(), size(), max_*_length(), init_context, pattern / text length,
//     load_strings, output; context {min_score, sink} for scores, {min_score, backtracer, alignment} for tracebacks.
// One stream template covers the four uses (banded score, whole-window score, banded traceback, whole-window traceback); a job is a
// candidate placement {read, position, strand} of a read (or of its mate) on the genome.
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

namespace client {

// ------------------------------------------------------------------------------------------------------------------ scoring
/// mismatch penalty growing linearly with the phred quality up to 40, truncated the way float -> int conversion truncates
struct PhredPenalty
{
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE PhredPenalty(const int at0 = 0, const int at40 = 0) : lo(at0), hi(at40) {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE int operator()(const int phred) const { return lo + int((float)(phred < 40 ? phred : 40) / 40.0f * (hi - lo)); }
    int lo, hi;
};
struct FixedBonus
{
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE FixedBonus(const int v = 0) : value(v) {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE int operator()(const int) const { return value; }
    int value;
};
/// affine gaps with separate costs on the read and on the reference side: a gap of length L costs base + L * step
struct ViewScheme
{
    typedef FixedBonus   match_cost_function;
    typedef PhredPenalty mismatch_cost_function;
    static const int32 worst_score = -(1 << 16);

    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE int32 match(const uint8 q = 0) const { return bonus(q); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE int32 mismatch(const uint8 q = 0) const { return -penalty(q); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE int32 mismatch(const uint8, const uint8, const uint8 q = 0) const { return -penalty(q); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE int32 substitution(const uint32, const uint32, const uint8 ref, const uint8 sym, const uint8 q = 0) const
    { return ref == sym ? bonus(q) : -penalty(q); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE int32 pattern_gap_open()      const { return -(read_gap_base + read_gap_step); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE int32 pattern_gap_extension() const { return -read_gap_step; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE int32 text_gap_open()         const { return -(ref_gap_base + ref_gap_step); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE int32 text_gap_extension()    const { return -ref_gap_step; }

    FixedBonus   bonus;
    PhredPenalty penalty;
    int          read_gap_base, read_gap_step, ref_gap_base, ref_gap_step;
};

// ------------------------------------------------------------------------------------------------------------------ data
/// 4-bit big-endian reads, stored back to front, with a quality byte per stored symbol -- what io::ReadLoader loads from
struct StoredReads
{
    static const uint32 SEQUENCE_BITS       = io::SequenceDataTraits<DNA_N>::SEQUENCE_BITS;
    static const bool   SEQUENCE_BIG_ENDIAN = io::SequenceDataTraits<DNA_N>::SEQUENCE_BIG_ENDIAN;
    typedef cuda::ldg_pointer<uint32>  sequence_storage_iterator;
    typedef cuda::ldg_pointer<uint8>   qual_storage_iterator;
    typedef PackedStream<sequence_storage_iterator, uint8, SEQUENCE_BITS, SEQUENCE_BIG_ENDIAN> sequence_stream_type;
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE sequence_stream_type  sequence_stream() const { return sequence_stream_type(sequence_storage_iterator(words)); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE qual_storage_iterator qual_stream()     const { return qual_storage_iterator(quals); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint2 extent(const uint32 r) const { return make_uint2(offsets[r], offsets[r + 1]); }
    const uint32* words; const uint8* quals; const uint32* offsets; uint32 longest;
};
struct Candidate { uint32 read, position, reverse_strand; };

/// everything a batch reads and writes
struct Workspace
{
    typedef PackedStream<cuda::ldg_pointer<uint32>, uint8, 2, true> genome_type;
    StoredReads       reads, mates;
    genome_type       genome;
    uint32            genome_length;
    const uint32*     order;            // job i works on candidate order[i]
    const Candidate*  candidates;
    uint32            n_jobs;
    const int32*      runner_up;        // per read: the score to beat
    int32             floor;
    const uint2*      windows;          // whole-window modes: the genome window of each candidate's mate
    uint32            widest_window;
    int32*  placed_score; uint32* placed_end; int32* raw_score; uint2* raw_sink;
    uint16* cigar; uint32 cigar_stride; uint32* cigar_len; int32* aln_score; uint2* aln_source; uint2* aln_sink;
};

/// one run-length CIGAR element in 16 bits: operation in the low two bits, length above (soft clip = 3)
struct Run
{
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE Run() {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE Run(const uint32 op, const uint32 len) : packed(uint16(op | (len << 2))) {}
    uint16 packed;
};
/// the backtracer: receives the alignment from its end backwards and run-length encodes it as it comes
struct RunRecorder
{
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE RunRecorder() : n(0), last(0xFFu) {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE void clip(const uint32 len) { if (len) { runs[n++] = Run(3u, len); last = 0xFFu; } }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE void push(const uint8 op)
    {
        if (op == last) runs[n - 1u].packed += 4u;
        else { runs[n++] = Run(op, 1u); last = op; }
    }
    Run runs[1024]; uint32 n; uint8 last;
};

enum Mode { BAND_SCORE = 0, WINDOW_SCORE = 1, BAND_TRACE = 2, WINDOW_TRACE = 3 };
template <Mode M> struct mode_traits { static const bool whole_window = (M == WINDOW_SCORE || M == WINDOW_TRACE); static const bool trace = (M == BAND_TRACE || M == WINDOW_TRACE); };

template <bool TRACE> struct Result { aln::BestSink<int32> sink; };
template <> struct Result<true> { RunRecorder backtracer; aln::Alignment<int32> alignment; };

// ------------------------------------------------------------------------------------------------------------------ the stream
template <Mode MODE, typename Aligner>
struct PlacementStream
{
    typedef Aligner aligner_type;
    typedef mode_traits<MODE> traits;
    typedef lmem_cache_tag<64> cache_tag;
    typedef ReadLoader<StoredReads, cache_tag>                          read_loader;
    typedef PackedStringLoader<cuda::ldg_pointer<uint32>, 2, true, cache_tag> genome_loader;

    /// the job: which candidate, which read range, which strand, which genome window, the score to beat -- and where the result goes
    struct context_type : Result<traits::trace>
    {
        uint32 candidate; uint2 stored; bool reverse_strand; uint32 window_lo, window_hi; int32 min_score;
    };
    /// the strings of a job: a view of the stored read, its qualities in the same order, the genome window
    struct strings_type
    {
        typename read_loader::string_type                        pattern;
        typename read_loader::string_type::qual_string_type      quals;
        vector_view<typename genome_loader::iterator>            text;
        read_loader   loads_read;
        genome_loader loads_genome;
    };

    PlacementStream(const Workspace& workspace, const Aligner aligner, const uint32 band) : ws(workspace), al(aligner), band_len(band) {}

    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE const Aligner& aligner() const { return al; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 size() const { return ws.n_jobs; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 max_pattern_length() const { return traits::whole_window ? ws.mates.longest : ws.reads.longest; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 max_text_length() const { return traits::whole_window ? ws.widest_window : ws.reads.longest + band_len; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 pattern_length(const uint32, const context_type* job) const { return job->stored.y - job->stored.x; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 text_length(const uint32, const context_type* job) const { return job->window_hi - job->window_lo; }

    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool init_context(const uint32 i, context_type* job) const
    {
        job->candidate = ws.order[i];
        const Candidate c = ws.candidates[job->candidate];
        if (traits::whole_window)
        {
            // the mate lies on the other strand, somewhere in a window computed elsewhere; an empty window means "do not look"
            job->stored = ws.mates.extent(c.read);
            job->reverse_strand = !c.reverse_strand;
            job->window_lo = ws.windows[job->candidate].x; job->window_hi = ws.windows[job->candidate].y;
            job->min_score = ws.floor;
            if constexpr (!traits::trace) job->sink = aln::BestSink<int32>();
            return traits::trace || job->window_hi > job->window_lo;
        }
        // the read is tried around its candidate position: half a band before it, a band plus its length wide, cut at the genome's end
        job->stored = ws.reads.extent(c.read);
        job->reverse_strand = c.reverse_strand != 0u;
        const uint32 len = job->stored.y - job->stored.x, half = band_len / 2u;
        job->window_lo = c.position > half ? c.position - half : 0u;
        job->window_hi = nvbio::min(job->window_lo + band_len + len, ws.genome_length);
        job->min_score = traits::trace ? ws.floor : nvbio::max(ws.runner_up[c.read], ws.floor);
        if constexpr (!traits::trace) job->sink = aln::BestSink<int32>();
        return true;
    }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE void load_strings(const uint32, const uint32, const uint32, const context_type* job, strings_type* s) const
    {
        // reads are stored back to front: the forward strand is the stored read walked backwards, the reverse complement is the stored
        // read walked forwards with every base complemented
        const StoredReads& from = traits::whole_window ? ws.mates : ws.reads;
        s->pattern = s->loads_read.load(from, job->stored, job->reverse_strand ? FORWARD : REVERSE, job->reverse_strand ? COMPLEMENT : STANDARD);
        s->quals   = s->pattern.qualities();
        const uint32 span = job->window_hi - job->window_lo;
        s->text    = vector_view<typename genome_loader::iterator>(span, s->loads_genome.load(ws.genome + job->window_lo, span));
    }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE void output(const uint32, const context_type* job) const
    {
        const uint32 k = job->candidate;
        if constexpr (traits::trace)
        {
            ws.aln_score[k] = job->alignment.score; ws.aln_source[k] = job->alignment.source; ws.aln_sink[k] = job->alignment.sink;
            const uint32 n = job->backtracer.n;
            ws.cigar_len[k] = n;
            for (uint32 r = 0; r < n && r < ws.cigar_stride; ++r) ws.cigar[uint64(k) * ws.cigar_stride + r] = job->backtracer.runs[r].packed;
        }
        else
        {
            ws.raw_score[k] = job->sink.score; ws.raw_sink[k] = job->sink.sink;
            if (!traits::whole_window)
            {
                ws.placed_score[k] = nvbio::max(job->sink.score, int32(ViewScheme::worst_score));
                ws.placed_end[k]   = job->window_lo + job->sink.sink.x;
            }
        }
    }
    Workspace ws; Aligner al; uint32 band_len;
};

typedef aln::GotohAligner<aln::LOCAL, ViewScheme>       local_aligner;
typedef aln::GotohAligner<aln::SEMI_GLOBAL, ViewScheme> end_to_end_aligner;

} // namespace client

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

static client::Workspace make_workspace(const bt2_args* a)
{
    client::Workspace w;
    w.reads.words = a->read_words; w.reads.quals = a->read_quals; w.reads.offsets = a->read_index; w.reads.longest = a->longest;
    w.mates.words = a->mate_words; w.mates.quals = a->mate_quals; w.mates.offsets = a->mate_index; w.mates.longest = a->mate_longest;
    w.genome = client::Workspace::genome_type(cuda::ldg_pointer<uint32>(a->genome_words));
    w.genome_length = a->genome_length;
    w.order = a->idx_queue; w.candidates = (const client::Candidate*)a->hits; w.n_jobs = a->n_hits;
    w.runner_up = a->second_best; w.floor = a->score_limit;
    w.windows = (const uint2*)a->o_windows; w.widest_window = a->max_window;
    w.placed_score = a->hit_score; w.placed_end = a->hit_sink; w.raw_score = a->raw_score; w.raw_sink = (uint2*)a->raw_sink;
    w.cigar = a->cigar; w.cigar_stride = a->cigar_stride; w.cigar_len = a->cigar_len; w.aln_score = a->aln_score;
    w.aln_source = (uint2*)a->aln_source; w.aln_sink = (uint2*)a->aln_sink;
    return w;
}
static client::ViewScheme make_scheme(const bt2_args* a)
{
    client::ViewScheme s;
    s.read_gap_base = a->rdg_c; s.read_gap_step = a->rdg_k; s.ref_gap_base = a->rfg_c; s.ref_gap_step = a->rfg_k;
    s.bonus = client::FixedBonus(a->match); s.penalty = client::PhredPenalty(a->mmp_min, a->mmp_max);
    return s;
}
static void copy_path(char* out, const char* path) { strncpy(out, path, 15); out[15] = 0; }

template <typename aligner_type>
static int run_banded_score(const bt2_args* a, const aligner_type aligner, char* path)
{
    typedef client::PlacementStream<client::BAND_SCORE, aligner_type> stream_type;
    typedef aln::priv::recognised<stream_type> R;
    static_assert(R::value && R::staged && R::stage_quals, "a stream of read views under a quality scheme is recognised");
    static_assert(R::view, "... and its banded score batches run on the views in place");
    const client::Workspace w = make_workspace(a);
    if (a->band_len < 16) { aln::BatchedBandedAlignmentScore<15u, stream_type, aln::DeviceThreadScheduler> batch; batch.enact(stream_type(w, aligner, 15u), 0, NULL); copy_path(path, batch.last_path()); }
    else                  { aln::BatchedBandedAlignmentScore<31u, stream_type, aln::DeviceThreadScheduler> batch; batch.enact(stream_type(w, aligner, 31u), 0, NULL); copy_path(path, batch.last_path()); }
    return int(hipDeviceSynchronize());
}
API int bt2_banded_score(const bt2_args* a, char* path)
{
    try {
        const client::ViewScheme s = make_scheme(a);
        return a->local ? run_banded_score(a, client::local_aligner(s), path) : run_banded_score(a, client::end_to_end_aligner(s), path);
    } catch (const std::exception& e) { fprintf(stderr, "bt2_banded_score: %s\n", e.what()); return -1; }
}
/// the same stream through the STAGED tuned route (patterns and qualities copied to a scratch first -- what the view route replaces;
/// NVBIO_HIP_COMPAT_NO_VIEWS selects it) and down the generic lane (what a build without the tuned recognition runs): for timing
/// them beside the in-place route, and for checking that all three agree
API int bt2_banded_score_staged(const bt2_args* a, char* path)
{
    setenv("NVBIO_HIP_COMPAT_NO_VIEWS", "1", 1);
    const int r = bt2_banded_score(a, path);
    unsetenv("NVBIO_HIP_COMPAT_NO_VIEWS");
    return r;
}
API int bt2_banded_score_generic(const bt2_args* a)
{
    typedef client::PlacementStream<client::BAND_SCORE, client::local_aligner> stream_type;
    const stream_type stream(make_workspace(a), client::local_aligner(make_scheme(a)), 15u);
    hipLaunchKernelGGL((aln::priv::batched_banded_score_kernel<15u, stream_type>), dim3((a->n_hits + 127u) / 128u), dim3(128), 0, 0, stream);
    return int(hipDeviceSynchronize());
}
template <typename aligner_type>
static int run_full_score(const bt2_args* a, const aligner_type aligner, char* path)
{
    typedef client::PlacementStream<client::WINDOW_SCORE, aligner_type> stream_type;
    static_assert(aln::priv::recognised<stream_type>::staged, "the whole-window score stream is recognised");
    aln::BatchedAlignmentScore<stream_type, aln::DeviceThreadScheduler> batch;
    batch.enact(stream_type(make_workspace(a), aligner, 0u), 0, NULL);
    copy_path(path, batch.last_path());
    return int(hipDeviceSynchronize());
}
API int bt2_opposite_score(const bt2_args* a, char* path)
{
    try {
        const client::ViewScheme s = make_scheme(a);
        return a->local ? run_full_score(a, client::local_aligner(s), path) : run_full_score(a, client::end_to_end_aligner(s), path);
    } catch (const std::exception& e) { fprintf(stderr, "bt2_opposite_score: %s\n", e.what()); return -1; }
}
template <typename aligner_type>
static int run_traceback(const bt2_args* a, const aligner_type aligner, const bool full, char* path)
{
    const client::Workspace w = make_workspace(a);
    if (full)
    {
        typedef client::PlacementStream<client::WINDOW_TRACE, aligner_type> stream_type;
        static_assert(aln::priv::recognised_tb<stream_type>::staged, "the whole-window traceback stream is recognised");
        aln::BatchedAlignmentTraceback<1024u, stream_type> batch;
        batch.enact(stream_type(w, aligner, 0u), 0, NULL);
        copy_path(path, batch.last_path());
    }
    else
    {
        typedef client::PlacementStream<client::BAND_TRACE, aligner_type> stream_type;
        static_assert(aln::priv::recognised_tb<stream_type>::staged, "the banded traceback stream is recognised");
        if (a->band_len < 16) { aln::BatchedBandedAlignmentTraceback<15u, 1024u, stream_type> batch; batch.enact(stream_type(w, aligner, 15u), 0, NULL); copy_path(path, batch.last_path()); }
        else                  { aln::BatchedBandedAlignmentTraceback<31u, 1024u, stream_type> batch; batch.enact(stream_type(w, aligner, 31u), 0, NULL); copy_path(path, batch.last_path()); }
    }
    return int(hipDeviceSynchronize());
}
API int bt2_traceback(const bt2_args* a, int full, char* path)
{
    try {
        const client::ViewScheme s = make_scheme(a);
        return a->local ? run_traceback(a, client::local_aligner(s), full != 0, path) : run_traceback(a, client::end_to_end_aligner(s), full != 0, path);
    } catch (const std::exception& e) { fprintf(stderr, "bt2_traceback: %s\n", e.what()); return -1; }
}
