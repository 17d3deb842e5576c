// tests/compat/aln_callers.hip -- a caller of nvbio::aln written against the reference's documented interface
// (nvbio/alignment/batched.h:239-296, the way sw-benchmark/sw-benchmark.cu:79-218 and nvBowtie's score streams use it):
// user-side stream classes whose functors are evaluated per job, handed to BatchedBandedAlignmentScore /
// BatchedAlignmentScore.  It includes the reference's header names and is compiled with
// `hipcc -I include/nvbio_hip/compat`; nothing in the stream classes knows about this build.
// The extern "C" entry points at the bottom exist only so that the Python tests can drive it.
#include <nvbio/basic/types.h>
#include <nvbio/basic/packedstream.h>
#include <nvbio/basic/packedstream_loader.h>
#include <nvbio/basic/vector_view.h>
#include <nvbio/basic/cuda/ldg.h>
#include <nvbio/io/sequence/sequence.h>
#include <nvbio/alignment/alignment.h>
#include <nvbio/alignment/batched.h>
#include <string.h>

using namespace nvbio;

// ---------------------------------------------------------------------------------------------------------------------
// stream 1: reads packed 4 bits per base (big-endian, io::SequenceDataTraits<DNA_N>), reference windows packed 2 bits
// per base (little-endian), strings fetched through PackedStringLoader over read-only word pointers
// ---------------------------------------------------------------------------------------------------------------------
template <typename t_aligner_type, typename cache_tag>
struct PackedReadStream
{
    typedef t_aligner_type aligner_type;
    typedef cuda::ldg_pointer<uint32> word_iterator;
    typedef PackedStringLoader<word_iterator, io::SequenceDataTraits<DNA_N>::SEQUENCE_BITS, io::SequenceDataTraits<DNA_N>::SEQUENCE_BIG_ENDIAN, cache_tag> read_loader_type;
    typedef PackedStringLoader<word_iterator, 2, false, uncached_tag>                                                                                         window_loader_type;
    typedef vector_view<typename read_loader_type::iterator>   read_string;
    typedef vector_view<typename window_loader_type::iterator> window_string;

    struct context_type { int32 min_score; aln::BestSink<int32> sink; };
    struct strings_type
    {
        read_loader_type            read_loader;
        window_loader_type          window_loader;
        read_string                 pattern;
        aln::trivial_quality_string quals;
        window_string               text;
    };

    PackedReadStream(aligner_type aligner, uint32 count, const uint32* read_offsets, const uint32* read_words, uint32 longest_read,
                     const uint32* window_offsets, const uint32* window_words, uint32 longest_window,
                     const int32* thresholds, int32* scores, uint2* sinks)
        : m_aligner(aligner), m_count(count), m_read_offsets(read_offsets), m_reads(word_iterator(read_words)), m_longest_read(longest_read),
          m_window_offsets(window_offsets), m_windows(word_iterator(window_words)), m_longest_window(longest_window),
          m_thresholds(thresholds), m_scores(scores), m_sinks(sinks) {}

    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE const aligner_type& aligner() const { return m_aligner; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 max_pattern_length() const { return m_longest_read; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 max_text_length() const { return m_longest_window; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 size() const { return m_count; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 pattern_length(const uint32 i, context_type*) const { return m_read_offsets[i + 1] - m_read_offsets[i]; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 text_length(const uint32 i, context_type*) const { return m_window_offsets[i + 1] - m_window_offsets[i]; }

    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool init_context(const uint32 i, context_type* context) const
    {
        context->min_score = m_thresholds ? m_thresholds[i] : Field_traits<int32>::min();
        return true;
    }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE void load_strings(const uint32 i, const uint32 window_begin, const uint32 window_end,
                                                          const context_type*, strings_type* strings) const
    {
        const uint32 r0 = m_read_offsets[i],   rn = m_read_offsets[i + 1] - r0;
        const uint32 w0 = m_window_offsets[i], wn = m_window_offsets[i + 1] - w0;
        strings->text    = window_string(wn, strings->window_loader.load(m_windows + w0, wn, make_uint2(window_begin, window_end), false));
        strings->pattern = read_string(rn, strings->read_loader.load(m_reads + r0, rn));
    }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE void output(const uint32 i, const context_type* context) const
    {
        m_scores[i] = context->sink.score;
        m_sinks[i]  = context->sink.sink;
    }

    aligner_type m_aligner;
    uint32       m_count;
    const uint32* m_read_offsets;
    typename read_loader_type::input_iterator m_reads;
    uint32       m_longest_read;
    const uint32* m_window_offsets;
    typename window_loader_type::input_iterator m_windows;
    uint32       m_longest_window;
    const int32* m_thresholds;
    int32*       m_scores;
    uint2*       m_sinks;
};

// ---------------------------------------------------------------------------------------------------------------------
// stream 2: plain byte strings (the reference's own tests align uint8 / char arrays, nvbio-test/alignment_test.cu:677-795),
// per-base qualities, and a Best2Sink -- none of which the tuned kernels know
// ---------------------------------------------------------------------------------------------------------------------
template <typename t_aligner_type, typename sink_t>
struct ByteStringStream
{
    typedef t_aligner_type aligner_type;
    typedef vector_view<const uint8*> byte_string;
    struct context_type { int32 min_score; sink_t sink; };
    struct strings_type { byte_string pattern; byte_string quals; byte_string text; };

    ByteStringStream(aligner_type aligner, uint32 count, const uint32* read_offsets, const uint8* reads, const uint8* quals, uint32 longest_read,
                     const uint32* window_offsets, const uint8* windows, uint32 longest_window, const int32* thresholds, int32* scores, uint2* sinks)
        : m_aligner(aligner), m_count(count), m_read_offsets(read_offsets), m_reads(reads), m_quals(quals), m_longest_read(longest_read),
          m_window_offsets(window_offsets), m_windows(windows), m_longest_window(longest_window), m_thresholds(thresholds), m_scores(scores), m_sinks(sinks) {}

    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE const aligner_type& aligner() const { return m_aligner; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 max_pattern_length() const { return m_longest_read; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 max_text_length() const { return m_longest_window; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 size() const { return m_count; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 pattern_length(const uint32 i, context_type*) const { return m_read_offsets[i + 1] - m_read_offsets[i]; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 text_length(const uint32 i, context_type*) const { return m_window_offsets[i + 1] - m_window_offsets[i]; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool init_context(const uint32 i, context_type* context) const
    {
        context->min_score = m_thresholds ? m_thresholds[i] : Field_traits<int32>::min();
        return (i % 97u) != 96u;               // some jobs are declined: they are still output, with the sink as the context was built (batched_banded_inl.h:53-75)
    }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE void load_strings(const uint32 i, const uint32, const uint32, const context_type*, strings_type* strings) const
    {
        const uint32 r0 = m_read_offsets[i],   rn = m_read_offsets[i + 1] - r0;
        const uint32 w0 = m_window_offsets[i], wn = m_window_offsets[i + 1] - w0;
        strings->pattern = byte_string(rn, m_reads + r0);
        strings->quals   = byte_string(rn, m_quals + r0);
        strings->text    = byte_string(wn, m_windows + w0);
    }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE void output(const uint32 i, const context_type* context) const { write(i, context->sink); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE void write(const uint32 i, const aln::BestSink<int32>& s) const { m_scores[i] = s.score; m_sinks[i] = s.sink; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE void write(const uint32 i, const aln::Best2Sink<int32>& s) const { m_scores[i] = s.score1; m_sinks[i] = s.sink1; }

    aligner_type m_aligner; uint32 m_count;
    const uint32* m_read_offsets; const uint8* m_reads; const uint8* m_quals; uint32 m_longest_read;
    const uint32* m_window_offsets; const uint8* m_windows; uint32 m_longest_window;
    const int32* m_thresholds; int32* m_scores; uint2* m_sinks;
};

// ---------------------------------------------------------------------------------------------------------------------
// stream 3: patterns as plain bytes next to a packed 2-bit reference -- an 8-bit pattern string in place (nvbio_hip.h: bits == 8)
// ---------------------------------------------------------------------------------------------------------------------
template <typename t_aligner_type>
struct BytePatternStream
{
    typedef t_aligner_type aligner_type;
    typedef cuda::ldg_pointer<uint32> word_iterator;
    typedef PackedStringLoader<word_iterator, 2, false, uncached_tag> window_loader_type;
    typedef vector_view<const uint8*>                           read_string;
    typedef vector_view<typename window_loader_type::iterator>  window_string;
    struct context_type { int32 min_score; aln::BestSink<int32> sink; };
    struct strings_type { window_loader_type window_loader; read_string pattern; aln::trivial_quality_string quals; window_string text; };

    BytePatternStream(aligner_type aligner, uint32 count, const uint32* read_offsets, const uint8* reads, uint32 longest_read,
                      const uint32* window_offsets, const uint32* window_words, uint32 longest_window, const int32* thresholds, int32* scores, uint2* sinks)
        : m_aligner(aligner), m_count(count), m_read_offsets(read_offsets), m_reads(reads), m_longest_read(longest_read),
          m_window_offsets(window_offsets), m_windows(word_iterator(window_words)), m_longest_window(longest_window), m_thresholds(thresholds), m_scores(scores), m_sinks(sinks) {}

    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE const aligner_type& aligner() const { return m_aligner; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 max_pattern_length() const { return m_longest_read; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 max_text_length() const { return m_longest_window; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 size() const { return m_count; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 pattern_length(const uint32 i, context_type*) const { return m_read_offsets[i + 1] - m_read_offsets[i]; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 text_length(const uint32 i, context_type*) const { return m_window_offsets[i + 1] - m_window_offsets[i]; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool init_context(const uint32 i, context_type* context) const
    { context->min_score = m_thresholds ? m_thresholds[i] : Field_traits<int32>::min(); return true; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE void load_strings(const uint32 i, const uint32 window_begin, const uint32 window_end, const context_type*, strings_type* strings) const
    {
        const uint32 r0 = m_read_offsets[i],   rn = m_read_offsets[i + 1] - r0;
        const uint32 w0 = m_window_offsets[i], wn = m_window_offsets[i + 1] - w0;
        strings->text    = window_string(wn, strings->window_loader.load(m_windows + w0, wn, make_uint2(window_begin, window_end), false));
        strings->pattern = read_string(rn, m_reads + r0);
    }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE void output(const uint32 i, const context_type* context) const { m_scores[i] = context->sink.score; m_sinks[i] = context->sink.sink; }

    aligner_type m_aligner; uint32 m_count;
    const uint32* m_read_offsets; const uint8* m_reads; uint32 m_longest_read;
    const uint32* m_window_offsets; typename window_loader_type::input_iterator m_windows; uint32 m_longest_window;
    const int32* m_thresholds; int32* m_scores; uint2* m_sinks;
};

// a user-defined Gotoh scoring scheme: quality-dependent mismatches, different gap costs on the two strings
struct PhredGotohScheme
{
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE int32 match(const uint8 = 0) const { return 2; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE int32 substitution(const uint32, const uint32, const uint8 r, const uint8 q, const uint8 qq = 0) const
    { return r == q ? 2 : -(2 + int32(qq < 40 ? qq : 40) / 10); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE int32 pattern_gap_open() const { return -8; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE int32 pattern_gap_extension() const { return -3; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE int32 text_gap_open() const { return -6; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE int32 text_gap_extension() const { return -2; }
};

// ---------------------------------------------------------------------------------------------------------------------
// drivers
// ---------------------------------------------------------------------------------------------------------------------
struct Args
{
    uint32 n; const uint32* read_offsets; const void* reads; const uint8* quals; uint32 longest_read;
    const uint32* window_offsets; const void* windows; uint32 longest_window; const int32* thresholds; int32* scores; uint2* sinks;
};

template <typename batch_type, typename stream_type>
static const char* run(const stream_type& stream, const bool device)
{
    batch_type batch;
    const uint64 temp = batch_type::max_temp_storage(stream.max_pattern_length(), stream.max_text_length(), stream.size());
    batch.enact(stream, temp, NULL);
    if (device && hipDeviceSynchronize() != hipSuccess) return "error";
    return batch.last_path();
}

template <uint32 BAND, typename scheduler, typename aligner_type>
static const char* banded_packed(const aligner_type aligner, const Args& a, const bool device)
{
    typedef PackedReadStream<aligner_type, lmem_cache_tag<64> > stream_type;
    const stream_type stream(aligner, a.n, a.read_offsets, (const uint32*)a.reads, a.longest_read, a.window_offsets, (const uint32*)a.windows, a.longest_window,
                             a.thresholds, a.scores, a.sinks);
    return run< aln::BatchedBandedAlignmentScore<BAND, stream_type, scheduler> >(stream, device);
}
template <typename scheduler, typename aligner_type>
static const char* full_packed(const aligner_type aligner, const Args& a, const bool device)
{
    typedef PackedReadStream<aligner_type, uncached_tag> stream_type;
    const stream_type stream(aligner, a.n, a.read_offsets, (const uint32*)a.reads, a.longest_read, a.window_offsets, (const uint32*)a.windows, a.longest_window,
                             a.thresholds, a.scores, a.sinks);
    return run< aln::BatchedAlignmentScore<stream_type, scheduler> >(stream, device);
}
template <uint32 BAND, typename scheduler, typename sink_t, typename aligner_type>
static const char* banded_bytes(const aligner_type aligner, const Args& a, const bool device)
{
    typedef ByteStringStream<aligner_type, sink_t> stream_type;
    const stream_type stream(aligner, a.n, a.read_offsets, (const uint8*)a.reads, a.quals, a.longest_read, a.window_offsets, (const uint8*)a.windows, a.longest_window,
                             a.thresholds, a.scores, a.sinks);
    return run< aln::BatchedBandedAlignmentScore<BAND, stream_type, scheduler> >(stream, device);
}
template <typename scheduler, typename aligner_type>
static const char* full_bytes(const aligner_type aligner, const Args& a, const bool device)
{
    typedef ByteStringStream<aligner_type, aln::BestSink<int32> > stream_type;
    const stream_type stream(aligner, a.n, a.read_offsets, (const uint8*)a.reads, a.quals, a.longest_read, a.window_offsets, (const uint8*)a.windows, a.longest_window,
                             a.thresholds, a.scores, a.sinks);
    return run< aln::BatchedAlignmentScore<stream_type, scheduler> >(stream, device);
}

template <uint32 BAND, typename aligner_type>
static const char* banded_byte_patterns(const aligner_type aligner, const Args& a)
{
    typedef BytePatternStream<aligner_type> stream_type;
    const stream_type stream(aligner, a.n, a.read_offsets, (const uint8*)a.reads, a.longest_read, a.window_offsets, (const uint32*)a.windows, a.longest_window, a.thresholds, a.scores, a.sinks);
    return run< aln::BatchedBandedAlignmentScore<BAND, stream_type, aln::DeviceThreadScheduler> >(stream, true);
}
template <typename aligner_type>
static const char* full_byte_patterns(const aligner_type aligner, const Args& a)
{
    typedef BytePatternStream<aligner_type> stream_type;
    const stream_type stream(aligner, a.n, a.read_offsets, (const uint8*)a.reads, a.longest_read, a.window_offsets, (const uint32*)a.windows, a.longest_window, a.thresholds, a.scores, a.sinks);
    return run< aln::BatchedAlignmentScore<stream_type, aln::DeviceThreadScheduler> >(stream, true);
}

// kind: 0 Gotoh, 1 Smith-Waterman, 2 edit distance, 3 Gotoh with the user-defined PhredGotohScheme, 4 the bit-vector banded edit distance (MyersTag<5>)
template <typename F> static const char* with_type(const int type, F f)
{
    if (type == 0) return f(std::integral_constant<aln::AlignmentType, aln::GLOBAL>());
    if (type == 1) return f(std::integral_constant<aln::AlignmentType, aln::LOCAL>());
    return f(std::integral_constant<aln::AlignmentType, aln::SEMI_GLOBAL>());
}

#define API extern "C" __attribute__((visibility("default")))

// strings: 0 = packed words (reads 4-bit BE, windows 2-bit LE), 1 = bytes, 2 = byte patterns next to packed 2-bit windows (device only);  where: 0 = device pointers + DeviceThreadScheduler, 1 = host pointers + HostThreadScheduler
API int compat_banded_score(int strings, int where, int kind, int type, int band, const int* sc, unsigned n,
                            const unsigned* read_offsets, const void* reads, const unsigned char* quals, unsigned longest_read,
                            const unsigned* window_offsets, const void* windows, unsigned longest_window, int* scores, unsigned* sinks, char* path /* 16 bytes */)
{
    const Args a = { n, read_offsets, reads, quals, longest_read, window_offsets, windows, longest_window, NULL, scores, reinterpret_cast<uint2*>(sinks) };
    const char* p = NULL;
    try {
        p = with_type(type, [&](auto T) -> const char* {
            const aln::AlignmentType TYPE = decltype(T)::value;
            const aln::SimpleGotohScheme g(sc[0], sc[1], sc[2], sc[3]);
            const aln::SimpleSmithWatermanScheme w(sc[0], sc[1], sc[2], sc[3]);
            #define CASE(B) \
                if (band == B) { \
                    if (strings == 0 && where == 0) { \
                        if (kind == 0) return banded_packed<B, aln::DeviceThreadScheduler>(aln::make_gotoh_aligner<TYPE>(g), a, true); \
                        if (kind == 1) return banded_packed<B, aln::DeviceThreadScheduler>(aln::make_smith_waterman_aligner<TYPE>(w), a, true); \
                        if (kind == 2) return banded_packed<B, aln::DeviceThreadScheduler>(aln::make_edit_distance_aligner<TYPE>(), a, true); \
                        if (kind == 4) return banded_packed<B, aln::DeviceThreadScheduler>(aln::make_edit_distance_aligner<TYPE, aln::MyersTag<5> >(), a, true); \
                    } \
                    if (strings == 0 && where == 1 && kind == 0) return banded_packed<B, aln::HostThreadScheduler>(aln::make_gotoh_aligner<TYPE>(g), a, false); \
                    if (strings == 1 && where == 0) { \
                        if (kind == 0) return banded_bytes<B, aln::DeviceThreadScheduler, aln::BestSink<int32> >(aln::make_gotoh_aligner<TYPE>(g), a, true); \
                        if (kind == 1) return banded_bytes<B, aln::DeviceThreadScheduler, aln::Best2Sink<int32> >(aln::make_smith_waterman_aligner<TYPE>(w), a, true); \
                        if (kind == 2) return banded_bytes<B, aln::DeviceThreadScheduler, aln::BestSink<int32> >(aln::make_edit_distance_aligner<TYPE>(), a, true); \
                        if (kind == 3) return banded_bytes<B, aln::DeviceThreadScheduler, aln::BestSink<int32> >(aln::make_gotoh_aligner<TYPE>(PhredGotohScheme()), a, true); \
                        if (kind == 4) return banded_bytes<B, aln::DeviceThreadScheduler, aln::BestSink<int32> >(aln::make_edit_distance_aligner<TYPE, aln::MyersTag<5> >(), a, true); \
                    } \
                    if (strings == 1 && where == 1 && kind == 1) return banded_bytes<B, aln::HostThreadScheduler, aln::BestSink<int32> >(aln::make_smith_waterman_aligner<TYPE>(w), a, false); \
                    if (strings == 2 && where == 0) { \
                        if (kind == 0) return banded_byte_patterns<B>(aln::make_gotoh_aligner<TYPE>(g), a); \
                        if (kind == 1) return banded_byte_patterns<B>(aln::make_smith_waterman_aligner<TYPE>(w), a); \
                    } \
                }
            CASE(15) CASE(31) CASE(9)
            #undef CASE
            return (const char*)NULL;
        });
    } catch (const std::exception& e) { strncpy(path, e.what(), 15); path[15] = 0; return 2; }
    if (!p) return 1;
    strncpy(path, p, 15); path[15] = 0;
    return 0;
}

// tag: 0 pattern blocking, 1 text blocking
API int compat_full_score(int strings, int where, int kind, int type, int tag, const int* sc, unsigned n,
                          const unsigned* read_offsets, const void* reads, const unsigned char* quals, unsigned longest_read,
                          const unsigned* window_offsets, const void* windows, unsigned longest_window, const int* thresholds,
                          int* scores, unsigned* sinks, char* path)
{
    const Args a = { n, read_offsets, reads, quals, longest_read, window_offsets, windows, longest_window, thresholds, scores, reinterpret_cast<uint2*>(sinks) };
    const char* p = NULL;
    try {
        p = with_type(type, [&](auto T) -> const char* {
            const aln::AlignmentType TYPE = decltype(T)::value;
            const aln::SimpleGotohScheme g(sc[0], sc[1], sc[2], sc[3]);
            const aln::SimpleSmithWatermanScheme w(sc[0], sc[1], sc[2], sc[3]);
            #define TAGGED(TAG) \
                if (strings == 0 && where == 0) { \
                    if (kind == 0) return full_packed<aln::DeviceThreadScheduler>(aln::make_gotoh_aligner<TYPE, TAG>(g), a, true); \
                    if (kind == 1) return full_packed<aln::DeviceThreadScheduler>(aln::make_smith_waterman_aligner<TYPE, TAG>(w), a, true); \
                    if (kind == 2) return full_packed<aln::DeviceThreadScheduler>(aln::make_edit_distance_aligner<TYPE, TAG>(), a, true); \
                } \
                if (strings == 0 && where == 1 && kind == 0) return full_packed<aln::HostThreadScheduler>(aln::make_gotoh_aligner<TYPE, TAG>(g), a, false); \
                if (strings == 1 && where == 0) { \
                    if (kind == 0) return full_bytes<aln::DeviceThreadScheduler>(aln::make_gotoh_aligner<TYPE, TAG>(g), a, true); \
                    if (kind == 1) return full_bytes<aln::DeviceThreadScheduler>(aln::make_smith_waterman_aligner<TYPE, TAG>(w), a, true); \
                    if (kind == 3) return full_bytes<aln::DeviceThreadScheduler>(aln::make_gotoh_aligner<TYPE, TAG>(PhredGotohScheme()), a, true); \
                } \
                if (strings == 1 && where == 1 && kind == 1) return full_bytes<aln::HostThreadScheduler>(aln::make_smith_waterman_aligner<TYPE, TAG>(w), a, false); \
                if (strings == 2 && where == 0) { \
                    if (kind == 0) return full_byte_patterns(aln::make_gotoh_aligner<TYPE, TAG>(g), a); \
                    if (kind == 1) return full_byte_patterns(aln::make_smith_waterman_aligner<TYPE, TAG>(w), a); \
                }
            if (tag == 0) { TAGGED(aln::PatternBlockingTag) } else { TAGGED(aln::TextBlockingTag) }
            #undef TAGGED
            return (const char*)NULL;
        });
    } catch (const std::exception& e) { strncpy(path, e.what(), 15); path[15] = 0; return 2; }
    if (!p) return 1;
    strncpy(path, p, 15); path[15] = 0;
    return 0;
}

// the per-thread functions straight from a user kernel (alignment.h:257-300): one alignment per thread, byte strings
template <uint32 BAND>
__global__ void per_thread_kernel(const uint32 n, const uint32* ro, const uint8* reads, const uint32* wo, const uint8* windows, const aln::SimpleGotohScheme scheme, int32* scores)
{
    const uint32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    typedef vector_view<const uint8*> str;
    scores[i] = aln::banded_alignment_score<BAND>(aln::make_gotoh_aligner<aln::SEMI_GLOBAL>(scheme),
                                                  str(ro[i + 1] - ro[i], reads + ro[i]), str(wo[i + 1] - wo[i], windows + wo[i]), Field_traits<int32>::min());
}
API int compat_per_thread_score(const int* sc, unsigned n, const unsigned* ro, const unsigned char* reads, const unsigned* wo, const unsigned char* windows, int* scores)
{
    hipLaunchKernelGGL(per_thread_kernel<7>, dim3((n + 127u) / 128u), dim3(128), 0, 0, n, ro, reads, wo, windows, aln::SimpleGotohScheme(sc[0], sc[1], sc[2], sc[3]), scores);
    return int(hipDeviceSynchronize());
}

// ---------------------------------------------------------------------------------------------------------------------
// stream 3: a traceback stream (batched.h:359-420): the context carries the caller's backtracer and receives the Alignment.
// The backtracer below forms a run-length CIGAR the way an aligner's own would (clip -> 'S', push -> M / I / D); output()
// stores it with the alignment's score, source and sink.
// ---------------------------------------------------------------------------------------------------------------------
struct RunLengthBacktracer
{
    uint16* words; uint32 capacity, size; uint32 run_op, run_len;
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE void clear() { size = 0; run_op = 255u; run_len = 0; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE void flush() { if (run_len) { if (size < capacity) words[size] = uint16(run_op | (run_len << 2)); ++size; run_len = 0; } }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE void clip(const uint32 len) { flush(); run_op = 255u; if (len) { if (size < capacity) words[size] = uint16(3u | (len << 2)); ++size; } }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE void push(const uint8 op) { if (op != run_op) { flush(); run_op = op; } ++run_len; }
};
template <typename t_aligner_type>
struct PackedTracebackStream
{
    typedef t_aligner_type aligner_type;
    typedef cuda::ldg_pointer<uint32> word_iterator;
    typedef PackedStringLoader<word_iterator, 4, true, uncached_tag>  read_loader_type;
    typedef PackedStringLoader<word_iterator, 2, false, uncached_tag> window_loader_type;
    typedef vector_view<typename read_loader_type::iterator>   read_string;
    typedef vector_view<typename window_loader_type::iterator> window_string;
    struct context_type { int32 min_score; RunLengthBacktracer backtracer; aln::Alignment<int32> alignment; };
    struct strings_type { read_loader_type read_loader; window_loader_type window_loader; read_string pattern; aln::trivial_quality_string quals; window_string text; };

    PackedTracebackStream(aligner_type aligner, uint32 count, const uint32* read_offsets, const uint32* read_words, uint32 longest_read,
                          const uint32* window_offsets, const uint32* window_words, uint32 longest_window,
                          int32* scores, uint2* sinks, uint2* sources, uint16* cigars, uint32 cigar_stride, uint32* cigar_lens)
        : m_aligner(aligner), m_count(count), m_read_offsets(read_offsets), m_reads(word_iterator(read_words)), m_longest_read(longest_read),
          m_window_offsets(window_offsets), m_windows(word_iterator(window_words)), m_longest_window(longest_window),
          m_scores(scores), m_sinks(sinks), m_sources(sources), m_cigars(cigars), m_cigar_stride(cigar_stride), m_cigar_lens(cigar_lens) {}

    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE const aligner_type& aligner() const { return m_aligner; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 max_pattern_length() const { return m_longest_read; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 max_text_length() const { return m_longest_window; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 size() const { return m_count; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 pattern_length(const uint32 i, context_type*) const { return m_read_offsets[i + 1] - m_read_offsets[i]; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 text_length(const uint32 i, context_type*) const { return m_window_offsets[i + 1] - m_window_offsets[i]; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool init_context(const uint32 i, context_type* context) const
    {
        context->min_score = Field_traits<int32>::min();
        context->backtracer.words = m_cigars + uint64(i) * m_cigar_stride; context->backtracer.capacity = m_cigar_stride; context->backtracer.clear();
        context->alignment = aln::Alignment<int32>(-77, make_uint2(7u, 7u), make_uint2(7u, 7u));         // what a declined job leaves behind
        return (i % 53u) != 52u;               // some jobs are declined: they are output as they are
    }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE void load_strings(const uint32 i, const uint32 window_begin, const uint32 window_end, const context_type*, strings_type* strings) const
    {
        const uint32 r0 = m_read_offsets[i],   rn = m_read_offsets[i + 1] - r0;
        const uint32 w0 = m_window_offsets[i], wn = m_window_offsets[i + 1] - w0;
        strings->text    = window_string(wn, strings->window_loader.load(m_windows + w0, wn, make_uint2(window_begin, window_end), false));
        strings->pattern = read_string(rn, strings->read_loader.load(m_reads + r0, rn));
    }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE void output(const uint32 i, context_type* context) const
    {
        context->backtracer.flush();
        m_scores[i] = context->alignment.score;
        m_sinks[i]  = context->alignment.sink;
        m_sources[i] = context->alignment.source;
        m_cigar_lens[i] = context->backtracer.size;
    }
    aligner_type m_aligner; uint32 m_count;
    const uint32* m_read_offsets; typename read_loader_type::input_iterator m_reads; uint32 m_longest_read;
    const uint32* m_window_offsets; typename window_loader_type::input_iterator m_windows; uint32 m_longest_window;
    int32* m_scores; uint2* m_sinks; uint2* m_sources; uint16* m_cigars; uint32 m_cigar_stride; uint32* m_cigar_lens;
};

// band: 0 = full matrix; kind as above (0 Gotoh, 1 Smith-Waterman, 2 edit distance)
extern "C" __attribute__((visibility("default")))
int compat_traceback(int kind, int type, int band, const int* sc, unsigned n, const unsigned* read_offsets, const unsigned* reads, unsigned longest_read,
                     const unsigned* window_offsets, const unsigned* windows, unsigned longest_window,
                     int* scores, unsigned* sinks, unsigned* sources, unsigned short* cigars, unsigned cigar_stride, unsigned* cigar_lens)
{
    try {
        const int rc = [&]() -> int {
            #define RUN(ALIGNER) { \
                auto al = ALIGNER; typedef PackedTracebackStream<decltype(al)> stream_type; \
                const stream_type st(al, n, read_offsets, reads, longest_read, window_offsets, windows, longest_window, scores, \
                                     reinterpret_cast<uint2*>(sinks), reinterpret_cast<uint2*>(sources), cigars, cigar_stride, cigar_lens); \
                if (band == 0)       { aln::BatchedAlignmentTraceback<64, stream_type> b; b.enact(st); } \
                else if (band == 15) { aln::BatchedBandedAlignmentTraceback<15, 64, stream_type> b; b.enact(st); } \
                else if (band == 31) { aln::BatchedBandedAlignmentTraceback<31, 64, stream_type> b; b.enact(st); } \
                else return 1; \
                return hipDeviceSynchronize() == hipSuccess ? 0 : 3; }
            #define TYPED(TYPE) \
                if (kind == 0) RUN(aln::make_gotoh_aligner<TYPE>(aln::SimpleGotohScheme(sc[0], sc[1], sc[2], sc[3]))) \
                if (kind == 1) RUN(aln::make_smith_waterman_aligner<TYPE>(aln::SimpleSmithWatermanScheme(sc[0], sc[1], sc[2], sc[3]))) \
                if (kind == 2) RUN(aln::make_edit_distance_aligner<TYPE>())
            if (type == 0) { TYPED(aln::GLOBAL) } else if (type == 1) { TYPED(aln::LOCAL) } else { TYPED(aln::SEMI_GLOBAL) }
            #undef TYPED
            #undef RUN
            return 1;
        }();
        return rc;
    } catch (const std::exception&) { return 2; }
}

// ---------------------------------------------------------------------------------------------------------------------
// stream 4: a traceback stream over plain byte strings with per-base qualities (the reference's own tests trace uint8 strings
// with user schemes, nvbio-test/alignment_test.cu:417-426; batched_banded_inl.h:299-326 takes any stream): nothing the tuned
// kernels know, so the batch classes run the per-lane templates -- on the device, and under HostThreadScheduler on the host.
// ---------------------------------------------------------------------------------------------------------------------
template <typename t_aligner_type>
struct ByteTracebackStream
{
    typedef t_aligner_type aligner_type;
    typedef vector_view<const uint8*> byte_string;
    struct context_type { int32 min_score; RunLengthBacktracer backtracer; aln::Alignment<int32> alignment; };
    struct strings_type { byte_string pattern; byte_string quals; byte_string text; };

    ByteTracebackStream(aligner_type aligner, uint32 count, const uint32* read_offsets, const uint8* reads, const uint8* quals, uint32 longest_read,
                        const uint32* window_offsets, const uint8* windows, uint32 longest_window,
                        int32* scores, uint2* sinks, uint2* sources, uint16* cigars, uint32 cigar_stride, uint32* cigar_lens)
        : m_aligner(aligner), m_count(count), m_read_offsets(read_offsets), m_reads(reads), m_quals(quals), m_longest_read(longest_read),
          m_window_offsets(window_offsets), m_windows(windows), m_longest_window(longest_window),
          m_scores(scores), m_sinks(sinks), m_sources(sources), m_cigars(cigars), m_cigar_stride(cigar_stride), m_cigar_lens(cigar_lens) {}

    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE const aligner_type& aligner() const { return m_aligner; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 max_pattern_length() const { return m_longest_read; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 max_text_length() const { return m_longest_window; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 size() const { return m_count; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 pattern_length(const uint32 i, context_type*) const { return m_read_offsets[i + 1] - m_read_offsets[i]; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 text_length(const uint32 i, context_type*) const { return m_window_offsets[i + 1] - m_window_offsets[i]; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool init_context(const uint32 i, context_type* context) const
    {
        context->min_score = Field_traits<int32>::min();
        context->backtracer.words = m_cigars + uint64(i) * m_cigar_stride; context->backtracer.capacity = m_cigar_stride; context->backtracer.clear();
        context->alignment = aln::Alignment<int32>(-77, make_uint2(7u, 7u), make_uint2(7u, 7u));
        return (i % 53u) != 52u;
    }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE void load_strings(const uint32 i, const uint32, const uint32, const context_type*, strings_type* strings) const
    {
        const uint32 r0 = m_read_offsets[i],   rn = m_read_offsets[i + 1] - r0;
        const uint32 w0 = m_window_offsets[i], wn = m_window_offsets[i + 1] - w0;
        strings->pattern = byte_string(rn, m_reads + r0);
        strings->quals   = byte_string(rn, m_quals + r0);
        strings->text    = byte_string(wn, m_windows + w0);
    }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE void output(const uint32 i, context_type* context) const
    {
        context->backtracer.flush();
        m_scores[i] = context->alignment.score; m_sinks[i] = context->alignment.sink; m_sources[i] = context->alignment.source;
        m_cigar_lens[i] = context->backtracer.size;
    }
    aligner_type m_aligner; uint32 m_count;
    const uint32* m_read_offsets; const uint8* m_reads; const uint8* m_quals; uint32 m_longest_read;
    const uint32* m_window_offsets; const uint8* m_windows; uint32 m_longest_window;
    int32* m_scores; uint2* m_sinks; uint2* m_sources; uint16* m_cigars; uint32 m_cigar_stride; uint32* m_cigar_lens;
};

// where: 0 = device pointers + DeviceThreadScheduler, 1 = host pointers + HostThreadScheduler; kind: 0 Gotoh, 1 Smith-Waterman (any
// deletion / insertion costs), 2 edit distance, 3 Gotoh with the user-defined PhredGotohScheme; band: 0 = full matrix, 15, 31
extern "C" __attribute__((visibility("default")))
int compat_traceback_bytes(int where, int kind, int type, int band, const int* sc, unsigned n, const unsigned* read_offsets, const unsigned char* reads, const unsigned char* quals,
                           unsigned longest_read, const unsigned* window_offsets, const unsigned char* windows, unsigned longest_window,
                           int* scores, unsigned* sinks, unsigned* sources, unsigned short* cigars, unsigned cigar_stride, unsigned* cigar_lens, char* path)
{
    try {
        const char* p = NULL;
        const int rc = [&]() -> int {
            #define RUN_S(ALIGNER, SCHED) { \
                auto al = ALIGNER; typedef ByteTracebackStream<decltype(al)> stream_type; \
                const stream_type st(al, n, read_offsets, reads, quals, longest_read, window_offsets, windows, longest_window, scores, \
                                     reinterpret_cast<uint2*>(sinks), reinterpret_cast<uint2*>(sources), cigars, cigar_stride, cigar_lens); \
                if (band == 0)       { aln::BatchedAlignmentTraceback<64, stream_type, SCHED> b; b.enact(st); p = b.last_path(); } \
                else if (band == 15) { aln::BatchedBandedAlignmentTraceback<15, 64, stream_type, SCHED> b; b.enact(st); p = b.last_path(); } \
                else if (band == 31) { aln::BatchedBandedAlignmentTraceback<31, 64, stream_type, SCHED> b; b.enact(st); p = b.last_path(); } \
                else return 1; \
                return (where != 0 || hipDeviceSynchronize() == hipSuccess) ? 0 : 3; }
            #define RUN(ALIGNER) { if (where == 0) RUN_S(ALIGNER, aln::DeviceThreadScheduler) else RUN_S(ALIGNER, aln::HostThreadScheduler) }
            #define TYPED(TYPE) \
                if (kind == 0) RUN(aln::make_gotoh_aligner<TYPE>(aln::SimpleGotohScheme(sc[0], sc[1], sc[2], sc[3]))) \
                if (kind == 1) RUN(aln::make_smith_waterman_aligner<TYPE>(aln::SimpleSmithWatermanScheme(sc[0], sc[1], sc[2], sc[3]))) \
                if (kind == 2) RUN(aln::make_edit_distance_aligner<TYPE>()) \
                if (kind == 3) RUN(aln::make_gotoh_aligner<TYPE>(PhredGotohScheme()))
            if (type == 0) { TYPED(aln::GLOBAL) } else if (type == 1) { TYPED(aln::LOCAL) } else { TYPED(aln::SEMI_GLOBAL) }
            #undef TYPED
            #undef RUN
            #undef RUN_S
            return 1;
        }();
        if (p) { strncpy(path, p, 15); path[15] = 0; }
        return rc;
    } catch (const std::exception&) { return 2; }
}
