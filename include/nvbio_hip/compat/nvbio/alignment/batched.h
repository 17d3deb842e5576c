// compat/nvbio/alignment/batched.h -- BatchedBandedAlignmentScore / BatchedAlignmentScore and the batch_* convenience
// functions of nvbio::aln (nvbio/alignment/batched.h:160-353), taking the reference's own `stream_type` concept
// (batched.h:239-296: aligner(), size(), max_pattern_length(), max_text_length(), pattern_length(i,ctx),
// text_length(i,ctx), init_context(i,ctx), load_strings(i,begin,end,ctx,strings), output(i,ctx), nested context_type
// {min_score, sink} and strings_type {pattern, quals, text}) -- so a caller's stream class binds unchanged.
//
// enact() picks one of three executions, all producing the reference's results:
//   * DeviceThreadScheduler, recognised stream: when strings.pattern / strings.text are vector_views over PackedStreams of
//     32-bit words in device memory (2- or 4-bit patterns, 2-bit texts), the qualities are trivial, the aligner is a
//     Gotoh / Smith-Waterman / edit-distance aligner over the library's Simple*Scheme and the sink a BestSink<int32>, a
//     small kernel evaluates the stream's functors per job into a job table (string offsets, lengths, min_score), the
//     tuned gfx950 kernels behind the C-ABI (include/nvbio_hip.h) score the table, and a second small kernel hands each
//     result to stream.output().  init_context() runs in both small kernels; streams are pure functors of i.
//   * DeviceThreadScheduler, staged stream: the same tuned kernels for streams whose patterns are not plain windows of packed
//     words or whose scheme reads qualities -- nvBowtie's own streams (nvBowtie/bowtie2/cuda/alignment_utils.h:170-218,
//     score_best_inl.h:54-148): io::ReadStream patterns (reads stored reversed, viewed forward or reverse-complemented), the
//     reads' quality strings, SmithWatermanScoringScheme (scoring.h:206-356).  The job-table kernel then also writes each job's
//     pattern (4 bits per symbol) and qualities (a byte per symbol), read through the stream's OWN iterators, into the batch
//     object's scratch; the scheme's mismatch(q) is tabulated on the host.  last_path() still says "tuned".
//   * DeviceThreadScheduler, any other stream: one lane per job runs the generic templates of alignment.h on whatever
//     iterators the stream hands out (8-bit strings, user schemes that are not a function of (equal?, quality), other sinks ...).
//   * HostThreadScheduler: the same templates over host pointers, OpenMP over jobs (batched_banded_inl.h:97-128).
// DeviceStagedThreadScheduler / DeviceWarpScheduler are accepted as aliases of the device path (they are alternative
// schedules of the same computation in the reference, batched.h:51-76).
#pragma once
#include "alignment.h"
#include "traceback.h"
#include "../io/utils.h"
#include "../basic/packedstream.h"
#include "../basic/packedstream_loader.h"
#include "../basic/packed_view.h"
#include "../basic/cuda/ldg.h"
#include "../strings/string_set.h"
#include <stdexcept>
#include <algorithm>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include <type_traits>
#include "../basic/omp.h"
#if defined(__HIPCC__) && !defined(NVBIO_HIP_COMPAT_NO_TUNED)
#include "../../../../nvbio_hip.h"
#define NVBIO_HIP_COMPAT_TUNED 1
#endif

namespace nvbio {
namespace aln {

struct HostThreadScheduler {};
template <uint32 BLOCKDIM, uint32 MINBLOCKS> struct DeviceThreadBlockScheduler {};
typedef DeviceThreadBlockScheduler<128, 1> DeviceThreadScheduler;
struct DeviceStagedThreadScheduler {};
struct DeviceWarpScheduler {};

/// which schedulers a batch of a given aligner runs under (batched.h:78-100): all four for the gapped aligners, all but the
/// warp scheduler for the ungapped one
template <typename aligner_type, typename scheduler_type> struct supports_scheduler { static const bool pred = true; };
template <AlignmentType TYPE, typename S, typename A> struct supports_scheduler< HammingDistanceAligner<TYPE, S, A>, DeviceWarpScheduler > { static const bool pred = false; };

struct hip_error : public std::runtime_error {
    int code;
    hip_error(const char* what, int c) : std::runtime_error(std::string(what) + " failed, hipError " + std::to_string(c)), code(c) {}
};

namespace priv {
/// A job's context, starting from all-zero bytes.  The reference declares its contexts uninitialised, and nvBowtie's BestAnchorScoreStream
/// reads context->min_score before writing it (score_paired_inl.h:150: `skip = ... || (context->min_score > a_optimal_score)`): whether a
/// hit is skipped then depends on what the register held.  This layer evaluates init_context in up to three kernels per batch (lengths,
/// job description, output), so the three must agree: every context is built over zeroed storage, which makes that test deterministic
/// (false, as it is for the zero a fresh CUDA local usually holds).
template <typename C>
struct fresh_context
{
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE fresh_context() { for (unsigned i = 0; i < sizeof(C); ++i) raw[i] = 0; new (raw) C; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE ~fresh_context() { get().~C(); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE C& get() { return *reinterpret_cast<C*>(raw); }
    alignas(C) unsigned char raw[sizeof(C)];
};
} // namespace priv

namespace priv {

inline void check(const int err, const char* what) { if (err != 0) throw hip_error(what, err); }

// ---------------------------------------------------------------------------------------- the per-job bodies
template <uint32 BAND_LEN, typename stream_type>
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE void banded_job(const stream_type& stream, const uint32 i)
{
    priv::fresh_context<typename stream_type::context_type> ctx_storage; typename stream_type::context_type& ctx = ctx_storage.get();
    typename stream_type::strings_type strings;
    if (!stream.init_context(i, &ctx)) { stream.output(i, &ctx); return; }      // a declined job is still output, as init_context left it (batched_banded_inl.h:53-75, batched_inl.h:58-63)
    const uint32 len = stream.pattern_length(i, &ctx);
    stream.load_strings(i, 0u, len, &ctx, &strings);
    banded_alignment_score<BAND_LEN>(stream.aligner(), strings.pattern, strings.quals, strings.text, ctx.min_score, ctx.sink);
    stream.output(i, &ctx);
}
template <typename stream_type>
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE void full_job(const stream_type& stream, const uint32 i, int16* column)
{
    priv::fresh_context<typename stream_type::context_type> ctx_storage; typename stream_type::context_type& ctx = ctx_storage.get();
    typename stream_type::strings_type strings;
    if (!stream.init_context(i, &ctx)) { stream.output(i, &ctx); return; }      // a declined job is still output, as init_context left it (batched_banded_inl.h:53-75, batched_inl.h:58-63)
    const uint32 len = stream.pattern_length(i, &ctx);
    stream.load_strings(i, 0u, len, &ctx, &strings);
    alignment_score(stream.aligner(), strings.pattern, strings.quals, strings.text, ctx.min_score, ctx.sink, column);
    stream.output(i, &ctx);
}

#if defined(__HIPCC__)
template <uint32 BAND_LEN, typename stream_type>
__global__ void __launch_bounds__(128) batched_banded_score_kernel(const stream_type stream)
{
    const uint32 i = blockIdx.x * 128u + threadIdx.x;
    if (i < stream.size()) banded_job<BAND_LEN>(stream, i);
}
template <typename stream_type>
__global__ void __launch_bounds__(128) batched_full_score_kernel(const stream_type stream, int16* columns, const uint64 column_stride)
{
    const uint32 i = blockIdx.x * 128u + threadIdx.x;
    if (i < stream.size()) full_job(stream, i, columns + uint64(i) * column_stride);
}

/// a growable device buffer owned by a batch object (the reference keeps a thrust::device_vector<uint8> there)
/// a batch object's scratch (job tables, staged strings, boundary columns).  Callers make a fresh batch object per call (sw-benchmark.cu:373-378,
/// nvBowtie's score / traceback functions), and a plain hipMalloc costs ~4 ms here -- a third of sw-benchmark's timed enact() -- so the
/// blocks come from the library's block cache (nvbio_hip_device_malloc; freed blocks kept).  The batch object dies -- and frees its scratch -- right
/// after enact() returns, with its kernels still queued on enact's stream: the buffer remembers that stream and frees with
/// nvbio_hip_device_free_after, which orders the next owner behind it without stopping the host
struct device_buffer
{
    device_buffer() : ptr(nullptr), bytes(0), last(nullptr) {}
    ~device_buffer() { release(); }
    device_buffer(const device_buffer&) = delete;
    device_buffer& operator=(const device_buffer&) = delete;
    /// `hs`: the stream the caller is about to use the buffer on
    uint8* reserve(const uint64 n, void* hs = nullptr)
    {
        if (hs != last && ptr && n > bytes) release();     // (a buffer that moves to another stream AND grows: free it behind the old one)
        last = hs;
        if (n > bytes)
        {
            release();
#if defined(NVBIO_HIP_COMPAT_TUNED)
            void* p = nullptr; check(nvbio_hip_device_malloc(&p, n), "nvbio_hip_device_malloc"); ptr = static_cast<uint8*>(p);
#else
            check(hipMalloc(reinterpret_cast<void**>(&ptr), n), "hipMalloc");
#endif
            bytes = n;
        }
        return ptr;
    }
    void release()
    {
        if (!ptr) return;
#if defined(NVBIO_HIP_COMPAT_TUNED)
        (void)nvbio_hip_device_free_after(ptr, last);
#else
        (void)hipFree(ptr);
#endif
        ptr = nullptr; bytes = 0;
    }
    uint8* ptr; uint64 bytes; void* last;
};
/// a few words of PINNED host memory per host thread, for the small read-backs between a batch's kernels (bounds, limits): hipMemcpyAsync
/// into pageable memory took 4.5 ms per call on the GPU box (a third of sw-benchmark's timed enact()); into pinned memory it is a DMA
inline unsigned long long* pinned_words()
{
    // (never freed: a thread's 128 bytes, against a hipHostFree that could run after the runtime has shut down)
    struct holder { unsigned long long* p; holder() : p(nullptr) { if (hipHostMalloc(reinterpret_cast<void**>(&p), 128u, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); p = nullptr; } } };
    static thread_local holder h;
    return h.p;
}
#endif

// ---------------------------------------------------------------------------------------- stream recognition
using nvbio::priv::word_pointer;         // basic/packed_view.h: "this iterator is a pointer to 32-bit words in memory"
using nvbio::priv::packed_view;          //                      "this string is a window of a 2/4-bit PackedStream over such words"

/// sinks the tuned route can fill: BestSink<int32>, and BestSink<int16> (examples/fmmap/fmmap.cu:306) while scores fit it
template <typename S> struct best_sink { static const bool ok = false; static const bool narrow = false; };
template <> struct best_sink< BestSink<int32> > { static const bool ok = true; static const bool narrow = false; };
template <> struct best_sink< BestSink<int16> > { static const bool ok = true; static const bool narrow = true; };

/// nvBowtie's scheme concept (nvBowtie/bowtie2/cuda/scoring.h:206-356): cost functors of the quality byte behind match(q) /
/// mismatch(q), substitution(.,.,r,q,qq) = r == q ? match(qq) : mismatch(qq), four gap accessors.  Detected by the two nested
/// cost-function typedefs; the host then CHECKS the concept on the values (quality_scheme_table) before any kernel relies on it.
template <typename S, typename = void> struct quality_scheme : std::false_type {};
template <typename S> struct quality_scheme<S, std::void_t<typename S::match_cost_function, typename S::mismatch_cost_function> > : std::true_type {};

template <typename A> struct tuned_aligner { static const bool ok = false; static const bool QUAL = false; };
#if defined(NVBIO_HIP_COMPAT_TUNED)
template <AlignmentType T, typename G> struct tuned_aligner< GotohAligner<T, SimpleGotohScheme, G> > {
    static const bool ok = true; static const bool QUAL = false; static const int32 KIND = NVBIO_HIP_GOTOH_ALIGNER; static const bool TEXT_BLOCKING = same_type<G, TextBlockingTag>::pred;
    static void scheme4(const GotohAligner<T, SimpleGotohScheme, G>& a, int32* v) { v[0] = a.scheme.m_match; v[1] = a.scheme.m_mismatch; v[2] = a.scheme.m_gap_open; v[3] = a.scheme.m_gap_ext; } };
template <AlignmentType T, typename G> struct tuned_aligner< SmithWatermanAligner<T, SimpleSmithWatermanScheme, G> > {
    static const bool ok = true; static const bool QUAL = false; static const int32 KIND = NVBIO_HIP_SW_ALIGNER; static const bool TEXT_BLOCKING = same_type<G, TextBlockingTag>::pred;
    static void scheme4(const SmithWatermanAligner<T, SimpleSmithWatermanScheme, G>& a, int32* v) { v[0] = a.scheme.m_match; v[1] = a.scheme.m_mismatch; v[2] = a.scheme.m_deletion; v[3] = a.scheme.m_insertion; } };
template <AlignmentType T, typename G> struct tuned_aligner< EditDistanceAligner<T, G> > {
    static const bool ok = true; static const bool QUAL = false; static const int32 KIND = NVBIO_HIP_SW_ALIGNER; static const bool TEXT_BLOCKING = same_type<G, TextBlockingTag>::pred;
    static void scheme4(const EditDistanceAligner<T, G>&, int32* v) { v[0] = 0; v[1] = -1; v[2] = -1; v[3] = -1; } };
/// the bit-vector edit distance is its own algorithm with its own reporting rules (alignment.h: banded_bitvector_score), not the DP the
/// tuned kernels run: such streams take the generic lane
template <AlignmentType T, uint32 N> struct tuned_aligner< EditDistanceAligner<T, MyersTag<N> > > { static const bool ok = false; static const bool QUAL = false; };
/// a Gotoh aligner over a scheme of nvBowtie's concept: its 256 mismatch penalties are tabulated on the host
template <AlignmentType T, typename S, typename G> struct tuned_aligner< GotohAligner<T, S, G> > {
    static const bool ok = quality_scheme<S>::value; static const bool QUAL = true; static const int32 KIND = NVBIO_HIP_GOTOH_ALIGNER; static const bool TEXT_BLOCKING = same_type<G, TextBlockingTag>::pred;
    /// false when the scheme's values do not follow the concept (a match bonus that depends on the quality, a substitution score
    /// that depends on more than (equal?, quality)): such a stream runs on the generic lane
    static bool table(const GotohAligner<T, S, G>& a, nvbio_hip_gotoh_qual_scheme& q)
    {
        const S& s = a.scheme;
        q.match = s.match(uint8(0));
        for (uint32 qq = 0; qq < 256u; ++qq)
        {
            q.mismatch[qq] = s.mismatch(uint8(qq));
            if (s.match(uint8(qq)) != q.match) return false;
            for (uint32 r = 0; r < 5u; ++r)
                for (uint32 c = 0; c < 5u; ++c)
                    if (s.substitution(7u * r, 3u * c, uint8(r), uint8(c), uint8(qq)) != (r == c ? q.match : q.mismatch[qq])) return false;
        }
        q.pattern_gap_open = s.pattern_gap_open(); q.pattern_gap_ext = s.pattern_gap_extension();
        q.text_gap_open = s.text_gap_open();       q.text_gap_ext = s.text_gap_extension();
        return true;
    }
};
#endif

/// how a pattern reaches the tuned kernels: in place (a window of packed words), staged (read through its own operator[] into the
/// batch's scratch: io::ReadStream views of <= 4-bit symbols), or not at all
/// "this iterator is a pointer to bytes in memory" (a read batch's quality stream)
template <typename It> struct byte_pointer { static const bool ok = false; };
template <> struct byte_pointer<const uint8*> { static const bool ok = true; NVBIO_FORCEINLINE NVBIO_HOST_DEVICE static const uint8* get(const uint8* p) { return p; } };
template <> struct byte_pointer<uint8*>       { static const bool ok = true; NVBIO_FORCEINLINE NVBIO_HOST_DEVICE static const uint8* get(uint8* p) { return p; } };
/// (the io layer keeps qualities as `char`: SequenceDataViewCore<..., const char*, ...>, io/sequence/sequence.h)
template <> struct byte_pointer<const char*>  { static const bool ok = true; NVBIO_FORCEINLINE NVBIO_HOST_DEVICE static const uint8* get(const char* p) { return reinterpret_cast<const uint8*>(p); } };
template <> struct byte_pointer<char*>        { static const bool ok = true; NVBIO_FORCEINLINE NVBIO_HOST_DEVICE static const uint8* get(char* p) { return reinterpret_cast<const uint8*>(p); } };
template <> struct byte_pointer< cuda::ldg_pointer<char> >  { static const bool ok = true; NVBIO_FORCEINLINE NVBIO_HOST_DEVICE static const uint8* get(cuda::ldg_pointer<char> p) { return reinterpret_cast<const uint8*>(p.base); } };
template <> struct byte_pointer< cuda::ldg_pointer<uint8> > { static const bool ok = true; NVBIO_FORCEINLINE NVBIO_HOST_DEVICE static const uint8* get(cuda::ldg_pointer<uint8> p) { return p.base; } };

/// ... or a VIEW run in place: an io::ReadStream over packed words with a byte-pointer quality stream -- the only pattern nvBowtie's
/// streams hand out (alignment_utils.h:170-218).  The banded quality-scheme kernels take (stored range, reverse / complement flags)
/// per job and turn each 16-symbol group round as they fetch it (nvbio_hip_banded_gotoh_score_qual_views): nothing is staged.
template <typename P> struct pattern_source { static const bool direct = packed_view<P>::ok; static const bool stageable = packed_view<P>::ok; static const bool viewable = false; };
template <typename St, typename Q> struct pattern_source< ReadStream<St, Q> >
{
    static const bool direct = false; static const bool stageable = (ReadStream<St, Q>::SYMBOL_SIZE <= 4u);
    static const bool viewable = nvbio::priv::packed_stream_where<St>::ok && byte_pointer<Q>::ok;
    typedef nvbio::priv::packed_stream_where<St> where_type; typedef byte_pointer<Q> qual_pointer;
};

template <typename stream_type, typename sink_like>
struct recognition
{
    typedef typename stream_type::strings_type strings_type;
    typedef decltype(strings_type().pattern) pattern_type;
    typedef decltype(strings_type().text)    text_type;
    typedef decltype(strings_type().quals)   quals_type;
    typedef tuned_aligner<typename stream_type::aligner_type> aligner;
    static const bool text_ok = packed_view<text_type>::ok && packed_view<text_type>::BITS == 2u;
    /// patterns in place: the strings are windows of packed words and the scheme ignores qualities
    static const bool zero_copy = aligner::ok && !aligner::QUAL && text_ok && pattern_source<pattern_type>::direct && sink_like::value;
    /// patterns (and qualities, for a quality scheme) staged through the stream's own iterators
    static const bool staged = aligner::ok && text_ok && pattern_source<pattern_type>::stageable && sink_like::value && !zero_copy;
    static const bool value = zero_copy || staged;
    static const bool stage_quals = staged && aligner::QUAL;
    /// a staged stream whose banded SCORE batches run on views in place instead (full-matrix and traceback batches still stage)
    static const bool view = staged && aligner::QUAL && pattern_source<pattern_type>::viewable;
};
template <typename stream_type> struct score_sink_ok { static const bool value = best_sink< decltype(typename stream_type::context_type().sink) >::ok; };
struct any_sink_ok { static const bool value = true; };
/// score streams (context holds a sink) and traceback streams (batched.h:359-420: a backtracer and an Alignment<int32> instead)
template <typename stream_type> struct recognised    : recognition<stream_type, score_sink_ok<stream_type> > {};
template <typename stream_type> struct recognised_tb : recognition<stream_type, any_sink_ok> {};

/// The longest pattern and text of a batch.  The stream concept offers max_pattern_length() / max_text_length() (batched.h:252-256), and a
/// stream of plain packed strings is taken at its word.  nvBowtie's own streams are not: their max_pattern_length() names a member its
/// read batch does not have (score_best_inl.h:80) and only compiles in the reference because DeviceThreadScheduler never instantiates it.
/// For every stream that is not `zero_copy` the two lengths are therefore MEASURED -- one light pass over init_context / pattern_length /
/// text_length -- and published for the duration of the enact() call through limits_scope; code below reads them with maxP_of / maxT_of.
/// NVBIO_HIP_COMPAT_GENERIC=<names> (test switch): run the named batch classes -- "banded", "full", "traceback" -- on the generic lane code
inline bool forced_generic(const char* name) { const char* e = getenv("NVBIO_HIP_COMPAT_GENERIC"); return e && strstr(e, name) != NULL; }
struct stream_limits { const void* stream; uint32 maxP, maxT; };
inline stream_limits& current_limits() { static thread_local stream_limits l = { NULL, 0u, 0u }; return l; }
template <typename stream_type> inline uint32 maxP_of(const stream_type& s) { const stream_limits& l = current_limits(); if (l.stream != &s) throw std::logic_error("batched: stream limits read outside an enact() scope"); return l.maxP; }
template <typename stream_type> inline uint32 maxT_of(const stream_type& s) { const stream_limits& l = current_limits(); if (l.stream != &s) throw std::logic_error("batched: stream limits read outside an enact() scope"); return l.maxT; }

template <typename stream_type>
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE void job_lengths(const stream_type& stream, const uint32 i, uint32& pl, uint32& tl)
{
    priv::fresh_context<typename stream_type::context_type> ctx_storage; typename stream_type::context_type& ctx = ctx_storage.get();
    pl = tl = 0;
    if (stream.init_context(i, &ctx)) { pl = stream.pattern_length(i, &ctx); tl = stream.text_length(i, &ctx); }
}
#if defined(__HIPCC__)
template <typename stream_type>
__global__ void __launch_bounds__(256) measure_limits_kernel(const stream_type stream, uint32* out)
{
    __shared__ uint32 s_p, s_t;
    if (threadIdx.x == 0) { s_p = 0; s_t = 0; }
    __syncthreads();
    uint32 mp = 0, mt = 0;
    const uint32 n = stream.size();
    for (uint64 i = uint64(blockIdx.x) * 256u + threadIdx.x; i < n; i += uint64(gridDim.x) * 256u)
    {
        uint32 pl, tl; job_lengths(stream, uint32(i), pl, tl);
        mp = pl > mp ? pl : mp; mt = tl > mt ? tl : mt;
    }
    atomicMax(&s_p, mp); atomicMax(&s_t, mt);
    __syncthreads();
    if (threadIdx.x == 0) { atomicMax(out, s_p); atomicMax(out + 1, s_t); }
}
#endif
template <typename stream_type, bool TRUSTED> struct limits_reader
{ static void read(const stream_type& s, uint32& p, uint32& t) { p = s.max_pattern_length(); t = s.max_text_length(); } };
template <typename stream_type> struct limits_reader<stream_type, false> { static void read(const stream_type&, uint32& p, uint32& t) { p = t = 0; } };

template <typename stream_type>
struct limits_scope
{
    static const bool trusted = recognised_tb<stream_type>::zero_copy;
    /// host streams: walk the jobs
    limits_scope(const stream_type& s) : m_saved(current_limits())
    {
        uint32 p = 0, t = 0;
        if (trusted) limits_reader<stream_type, trusted>::read(s, p, t);
        else
        {
            const int64 n = int64(s.size());
            #pragma omp parallel for reduction(max : p, t) schedule(static)
            for (int64 i = 0; i < n; ++i) { uint32 pl, tl; job_lengths(s, uint32(i), pl, tl); p = pl > p ? pl : p; t = tl > t ? tl : t; }
        }
        const stream_limits l = { &s, p, t }; current_limits() = l;
    }
#if defined(__HIPCC__)
    /// device streams: one reduction kernel (and a synchronization: the lengths size the launches that follow)
    limits_scope(const stream_type& s, hipStream_t hs) : m_saved(current_limits())
    {
        static thread_local device_buffer scratch;
        uint32 v[2] = { 0u, 0u };
        if (trusted) limits_reader<stream_type, trusted>::read(s, v[0], v[1]);
        else if (s.size())
        {
            uint32* d = reinterpret_cast<uint32*>(scratch.reserve(64u, hs));
            check(hipMemsetAsync(d, 0, 8u, hs), "hipMemsetAsync");
            const uint32 blocks = uint32(std::min<uint64>((uint64(s.size()) + 255u) / 256u, 4096u));
            hipLaunchKernelGGL((measure_limits_kernel<stream_type>), dim3(blocks), dim3(256), 0, hs, s, d);
            unsigned long long* pin = pinned_words();
            void* dst = pin ? static_cast<void*>(pin) : static_cast<void*>(v);
            check(hipMemcpyAsync(dst, d, 8u, hipMemcpyDeviceToHost, hs), "hipMemcpyAsync");
            check(hipStreamSynchronize(hs), "hipStreamSynchronize");
            if (pin) memcpy(v, pin, 8u);
        }
        const stream_limits l = { &s, v[0], v[1] }; current_limits() = l;
    }
#endif
    ~limits_scope() { current_limits() = m_saved; }
    stream_limits m_saved;
};

/// the same publication for limits that were measured on the way (the view route's description pass counts them: no pass of their own)
struct known_limits_scope
{
    template <typename stream_type>
    known_limits_scope(const stream_type& s, const uint32 p, const uint32 t) : m_saved(current_limits()) { const stream_limits l = { &s, p, t }; current_limits() = l; }
    ~known_limits_scope() { current_limits() = m_saved; }
    stream_limits m_saved;
};

#if defined(NVBIO_HIP_COMPAT_TUNED)
/// the job table the tuned kernels consume (structure of arrays in one device buffer)
struct job_table
{
    uint64* pat_begin; uint32* pat_len; uint64* txt_begin; uint32* txt_len; int32* min_score; int32* score; uint32* sink; uint8* ok;
    unsigned long long* bounds;        // [0] lowest pattern word, [1] end pattern word, [2] lowest text word, [3] end text word
                                       // views: [4] / [5] min / max of (quality address - pattern symbol address), [7] end pattern symbol,
                                       //        [9] / [11] the longest pattern / text (what limits_scope would have measured in a pass of its own)
    uint32* stage_words; uint8* stage_quals; uint32 stage_stride;         // staged patterns: job i at symbol i * stage_stride (4-bit, little-endian)
    // set by the host after the description pass: no job's min_score can ever bind (all <= -2^29; bounds[5] holds their maximum, biased by 2^31).
    // A stream that does not use thresholds says so with Field_traits<int32>::min() in every context (sw-benchmark.cu:204-215, batched_inl.h:944);
    // the full-matrix kernels then run their plain sweep instead of the one that watches the column maxima (12.2 instead of 16.2 ms for
    // sw-benchmark's LOCAL batch)
    bool    no_thresholds;
    static uint64 bytes(const uint32 n) { return 128u + uint64(n) * (8u + 4u + 8u + 4u + 4u + 4u + 8u + 1u) + 8u * 16u; }
    static uint32 stride_for(const uint32 maxP) { const uint32 s = (maxP + 7u) & ~7u; return s ? s : 8u; }
    static uint64 stage_bytes(const uint32 n, const uint32 maxP, const bool quals)
    { const uint64 sym = uint64(n) * stride_for(maxP) + 64u; return sym / 2u + (quals ? sym : 0u) + 32u; }
    void carve(uint8* p, const uint32 n)
    {
        bounds = reinterpret_cast<unsigned long long*>(p); p += 128u;
        pat_begin = reinterpret_cast<uint64*>(p); p += uint64(n) * 8u;
        txt_begin = reinterpret_cast<uint64*>(p); p += uint64(n) * 8u;
        sink      = reinterpret_cast<uint32*>(p); p += uint64(n) * 8u;
        pat_len   = reinterpret_cast<uint32*>(p); p += uint64(n) * 4u;
        txt_len   = reinterpret_cast<uint32*>(p); p += uint64(n) * 4u;
        min_score = reinterpret_cast<int32*>(p);  p += uint64(n) * 4u;
        score     = reinterpret_cast<int32*>(p);  p += uint64(n) * 4u;
        ok        = p;
        stage_words = nullptr; stage_quals = nullptr; stage_stride = 0; no_thresholds = false;
    }
};

__device__ __forceinline__ unsigned long long wave_min(unsigned long long v)
{ for (int o = 32; o > 0; o >>= 1) { const unsigned long long w = __shfl_xor(v, o, 64); v = w < v ? w : v; } return v; }
__device__ __forceinline__ unsigned long long wave_max(unsigned long long v)
{ for (int o = 32; o > 0; o >>= 1) { const unsigned long long w = __shfl_xor(v, o, 64); v = w > v ? w : v; } return v; }

static __global__ void init_bounds_kernel(unsigned long long* b) { if (threadIdx.x < 16u) b[threadIdx.x] = (threadIdx.x & 1u) ? 0ull : ~0ull; }

/// where a pattern lives (in place), or nothing (staged)
template <typename P, bool DIRECT> struct pattern_where {
    static const uint32 BITS = packed_view<P>::BITS; static const bool BE = packed_view<P>::BE;
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE static void get(const P& p, uint64& w, uint32& f) { packed_view<P>::where(p, w, f); } };
template <typename P> struct pattern_where<P, false> {
    static const uint32 BITS = 4u; static const bool BE = false;
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE static void get(const P&, uint64& w, uint32& f) { w = 0; f = 0; } };

/// evaluate the stream's functors per job: where its strings live, how long they are, its min_score.  A staged stream also has its
/// pattern (and qualities) read through its own iterators into the scratch, as 4-bit little-endian words and a byte per quality: then
/// SIXTEEN lanes work on a job, eight symbols = one word per lane and turn, so that a job's row is written in contiguous pieces (one lane
/// per job writes 38 scattered words per 100-bp read and is bound by exactly that: 12 of 14.7 ms per 10 M jobs).  Every one of the sixteen
/// evaluates the stream's context -- the same addresses in all of them, so no more lines move; the views a stream hands out may point into
/// the lane's own context (nvBowtie's loaders cache words there), so they cannot be passed between lanes.
template <typename stream_type, typename R> struct describe_lanes
{ static const uint32 value = R::staged ? (sizeof(typename stream_type::context_type) > 1024u ? 2u : 16u) : 1u; };
template <typename stream_type, typename R>
__global__ void __launch_bounds__(128) describe_jobs_kernel(const stream_type stream, const job_table t)
{
    typedef pattern_where<typename R::pattern_type, !R::staged> pwhere;
    // lanes per job.  Every lane of a job evaluates its own zeroed context: with a traceback stream's 4 KB context (nvBowtie keeps a
    // cigar[1024] in it, alignment_utils.h:252-258) sixteen lanes zeroed 64 KB of scratch per job -- 7.3 ms per 1 M jobs, more than
    // the traceback itself -- so large contexts are shared by two lanes only
    constexpr uint32 G = priv::describe_lanes<stream_type, R>::value;
    // a fixed grid strides over the jobs, every lane keeps its own bounds, and the four global bounds cost four atomics per BLOCK
    // (one per wave and bound made waves that see jobs in storage order queue up on the same addresses: see describe_views_kernel)
    unsigned long long plo = ~0ull, phi = 0ull, tlo = ~0ull, thi = 0ull, mshi = 0ull;
    const uint64 total = uint64(stream.size()) * G;
    for (uint64 gt = uint64(blockIdx.x) * 128u + threadIdx.x; gt < total; gt += uint64(gridDim.x) * 128u)
    {
        const uint32 i = uint32(gt / G), j = uint32(gt % G);
        priv::fresh_context<typename stream_type::context_type> ctx_storage; typename stream_type::context_type& ctx = ctx_storage.get();
        typename stream_type::strings_type strings;
        uint32 pl = 0, tl = 0; uint64 pw = 0, tw = 0; uint32 pf = 0, tf = 0; int32 ms = 0;
        if (stream.init_context(i, &ctx))
        {
            const uint32 len = stream.pattern_length(i, &ctx);
            stream.load_strings(i, 0u, len, &ctx, &strings);
            pwhere::get(strings.pattern, pw, pf);
            packed_view<typename R::text_type>::where(strings.text, tw, tf);
            pl = strings.pattern.length(); tl = strings.text.length(); ms = ctx.min_score;
            { const unsigned long long mb = static_cast<unsigned long long>(static_cast<long long>(ms) + (1ll << 31)); mshi = mb > mshi ? mb : mshi; }
            { const unsigned long long te = tw + (tf + tl + 15u) / 16u; tlo = tw < tlo ? tw : tlo; thi = te > thi ? te : thi; }
            if (R::staged)
            {
                // a job longer than the announced maximum cannot be staged: it is given an empty text, which the kernels refuse like
                // any text shorter than its pattern (the stream's max_pattern_length() is a contract, batched.h:252-256)
                if (pl > t.stage_stride) { tl = 0; }
                else
                {
                    uint32* w = t.stage_words + uint64(i) * (t.stage_stride / 8u);
                    for (uint32 k0 = 8u * j; k0 < pl; k0 += 8u * G)
                    {
                        uint32 word = 0;
                        for (uint32 k = 0; k < 8u; ++k) if (k0 + k < pl) word |= (uint32(strings.pattern[k0 + k]) & 15u) << (4u * k);
                        w[k0 / 8u] = word;
                    }
                    if (R::stage_quals)
                    {
                        uint32* qw = reinterpret_cast<uint32*>(t.stage_quals + uint64(i) * t.stage_stride);
                        for (uint32 k0 = 4u * j; k0 < pl; k0 += 4u * G)
                        {
                            uint32 word = 0;
                            for (uint32 k = 0; k < 4u; ++k) if (k0 + k < pl) word |= uint32(uint8(strings.quals[k0 + k])) << (8u * k);
                            qw[k0 / 4u] = word;
                        }
                    }
                }
            }
            else { const unsigned long long pe = pw + (pf + pl + 32u / pwhere::BITS - 1u) / (32u / pwhere::BITS); plo = pw < plo ? pw : plo; phi = pe > phi ? pe : phi; }
        }
        // offsets are kept absolute (symbols from address 0) until the host knows the lowest word
        if (j == 0u)
        {
            t.pat_begin[i] = R::staged ? uint64(i) * t.stage_stride : pw * (32u / pwhere::BITS) + pf; t.pat_len[i] = pl;
            t.txt_begin[i] = tw * 16u + tf; t.txt_len[i] = tl; t.min_score[i] = ms;
        }
    }
    __shared__ unsigned long long s_b[2][5];
    plo = wave_min(plo); phi = wave_max(phi); tlo = wave_min(tlo); thi = wave_max(thi); mshi = wave_max(mshi);
    if ((threadIdx.x & 63u) == 0u) { unsigned long long* o = s_b[threadIdx.x >> 6]; o[0] = plo; o[1] = phi; o[2] = tlo; o[3] = thi; o[4] = mshi; }
    __syncthreads();
    if (threadIdx.x < 4u)
    {
        const unsigned long long a = s_b[0][threadIdx.x], b = s_b[1][threadIdx.x];
        if ((threadIdx.x & 1u) == 0u) { const unsigned long long v = a < b ? a : b; if (v != ~0ull) atomicMin(&t.bounds[threadIdx.x], v); }
        else                          { const unsigned long long v = a > b ? a : b; if (v != 0ull) atomicMax(&t.bounds[threadIdx.x], v); }
    }
    else if (threadIdx.x == 4u)
    { const unsigned long long a = s_b[0][4], b = s_b[1][4], v = a > b ? a : b; if (v != 0ull) atomicMax(&t.bounds[5], v); }      // [5]: the largest min_score, biased by 2^31
}
static __global__ void __launch_bounds__(256) rebase_jobs_kernel(const uint32 n, uint64* pat_begin, const uint64 pat_delta, uint64* txt_begin, const uint64 txt_delta)
{
    const uint32 i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) { pat_begin[i] -= pat_delta; txt_begin[i] -= txt_delta; }
}
/// the view route's description of a job: where the STORED read lies (absolute symbol address), how the stream looks at it (t.ok[i]:
/// bit 0 reversed, bit 1 complemented), and where its qualities lie relative to its symbols -- one lane per job, nothing copied
template <typename stream_type, typename R>
__global__ void __launch_bounds__(256) describe_views_kernel(const stream_type stream, const job_table t)
{
    typedef pattern_source<typename R::pattern_type> source;
    typedef typename source::where_type              where_type;
    const uint32 per = 32u / where_type::BITS;
    // a fixed grid strides over the jobs and every lane keeps its own bounds: the seven global bounds then cost seven atomics per
    // BLOCK.  (One atomic per wave and bound is what the first version did; jobs usually come in storage order, so every wave moved
    // the upper bounds, and 156 k waves x 3 atomics on three addresses were 3 of its 4.1 ms per 10 M jobs.)
    unsigned long long plo = ~0ull, phi = 0ull, tlo = ~0ull, thi = 0ull, dlo = ~0ull, dhi = 0ull, send = 0ull, pmax = 0ull, tmax = 0ull;
    const uint32 n = stream.size();
    for (uint32 i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u)
    {
        priv::fresh_context<typename stream_type::context_type> ctx_storage; typename stream_type::context_type& ctx = ctx_storage.get();
        typename stream_type::strings_type strings;
        uint32 pl = 0, tl = 0, tf = 0, fl = 0; uint64 ps = 0, tw = 0; int32 ms = 0;
        if (stream.init_context(i, &ctx))
        {
            const uint32 len = stream.pattern_length(i, &ctx);
            stream.load_strings(i, 0u, len, &ctx, &strings);
            uint64 w; uint32 f;
            where_type::where(strings.pattern.stream, w, f);
            ps = w * per + f + strings.pattern.first;
            pl = strings.pattern.length();
            fl = (strings.pattern.rev ? 1u : 0u) | (strings.pattern.comp ? 2u : 0u);
            packed_view<typename R::text_type>::where(strings.text, tw, tf);
            tl = strings.text.length(); ms = ctx.min_score;
            pmax = pl > pmax ? pl : pmax; tmax = tl > tmax ? tl : tmax;
            const unsigned long long te = tw + (tf + tl + 15u) / 16u, pe = (ps + pl + per - 1u) / per;
            tlo = tw < tlo ? tw : tlo; thi = te > thi ? te : thi;
            plo = ps / per < plo ? ps / per : plo; phi = pe > phi ? pe : phi; send = ps + pl > send ? ps + pl : send;
            const unsigned long long qa = (unsigned long long)(reinterpret_cast<uintptr_t>(source::qual_pointer::get(strings.pattern.qual))) + strings.pattern.first;
            const unsigned long long d = qa - ps;       // modular: equal for all jobs of one read batch
            dlo = d < dlo ? d : dlo; dhi = d > dhi ? d : dhi;
        }
        t.pat_begin[i] = ps; t.pat_len[i] = pl; t.ok[i] = uint8(fl);
        t.txt_begin[i] = tw * 16u + tf; t.txt_len[i] = tl; t.min_score[i] = ms;
    }
    __shared__ unsigned long long s_b[4][9];
    plo = wave_min(plo); phi = wave_max(phi); tlo = wave_min(tlo); thi = wave_max(thi); send = wave_max(send); dlo = wave_min(dlo); dhi = wave_max(dhi);
    pmax = wave_max(pmax); tmax = wave_max(tmax);
    if ((threadIdx.x & 63u) == 0u)
    { unsigned long long* o = s_b[threadIdx.x >> 6]; o[0] = plo; o[1] = phi; o[2] = tlo; o[3] = thi; o[4] = dlo; o[5] = dhi; o[6] = send; o[7] = pmax; o[8] = tmax; }
    __syncthreads();
    if (threadIdx.x < 9u)
    {
        const bool is_min = (threadIdx.x == 0u || threadIdx.x == 2u || threadIdx.x == 4u);
        unsigned long long v = s_b[0][threadIdx.x];
        for (uint32 w = 1; w < 4u; ++w) { const unsigned long long x = s_b[w][threadIdx.x]; v = is_min ? (x < v ? x : v) : (x > v ? x : v); }
        const uint32 slot = threadIdx.x == 6u ? 7u : threadIdx.x == 7u ? 9u : threadIdx.x == 8u ? 11u : threadIdx.x;
        if (is_min) { if (v != ~0ull) atomicMin(&t.bounds[slot], v); } else if (v != 0ull) atomicMax(&t.bounds[slot], v);
    }
}
/// hand each result to the stream: a job the scorer refused (text shorter than pattern) leaves the fresh sink untouched
template <typename stream_type>
__global__ void __launch_bounds__(128) output_jobs_kernel(const stream_type stream, const job_table t)
{
    const uint32 i = blockIdx.x * 128u + threadIdx.x;
    if (i >= stream.size()) return;
    priv::fresh_context<typename stream_type::context_type> ctx_storage; typename stream_type::context_type& ctx = ctx_storage.get();
    if (!stream.init_context(i, &ctx)) { stream.output(i, &ctx); return; }      // a declined job is still output, as init_context left it (batched_banded_inl.h:53-75, batched_inl.h:58-63)
    const uint2 k = make_uint2(t.sink[2u * i], t.sink[2u * i + 1u]);
    if (!(k.x == 0xFFFFFFFFu && k.y == 0xFFFFFFFFu)) { ctx.sink.score = t.score[i]; ctx.sink.sink = k; }
    stream.output(i, &ctx);
}

/// hand each traceback to the stream the way the reference's per-job body does (batched_banded_inl.h:248-296, banded_inl.h:383-426):
/// a declined job is output untouched; a job without a sink gets Alignment(score, (-1,-1), (-1,-1)) and no backtracer call; otherwise
/// clip(end of the pattern past the sink), the walk's operations from the sink backwards, clip(start of the pattern), the Alignment.
template <typename stream_type>
__global__ void __launch_bounds__(128) replay_tracebacks_kernel(stream_type stream, const job_table t, const uint32* source, const uint16* cigar,
                                                                const uint32 cigar_stride, const uint32* cigar_len)
{
    const uint32 i = blockIdx.x * 128u + threadIdx.x;
    if (i >= stream.size()) return;
    priv::fresh_context<typename stream_type::context_type> ctx_storage; typename stream_type::context_type& ctx = ctx_storage.get();
    if (!stream.init_context(i, &ctx)) { stream.output(i, &ctx); return; }
    const uint2 snk = make_uint2(t.sink[2u * i], t.sink[2u * i + 1u]), src = make_uint2(source[2u * i], source[2u * i + 1u]);
    if (snk.x == 0xFFFFFFFFu || snk.y == 0xFFFFFFFFu)
        ctx.alignment = Alignment<int32>(t.score[i], make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu), make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu));
    else
    {
        ctx.backtracer.clip(t.pat_len[i] - snk.y);
        const uint16* c = cigar + uint64(i) * cigar_stride;
        const uint32 words = cigar_len[i] < cigar_stride ? cigar_len[i] : cigar_stride;
        for (uint32 k = 0; k < words; ++k)
        {
            const uint32 op = c[k] & 3u, len = uint32(c[k]) >> 2;
            if (op == 3u) continue;                                   // the two clips are made from sink / source
            for (uint32 l = 0; l < len; ++l) ctx.backtracer.push(DirectionVector(op));
        }
        ctx.backtracer.clip(src.y);
        ctx.alignment = Alignment<int32>(t.score[i], src, snk);
    }
    stream.output(i, &ctx);
}

/// describe (+ stage) -> (sync: the four bounds) -> rebase; fills the C-ABI string sets.  quals / n_quals: the staged qualities.
template <typename stream_type, typename R = recognised<stream_type> >
inline void build_job_table(const stream_type& stream, device_buffer& buf, job_table& t, nvbio_hip_string_set& ps, nvbio_hip_string_set& ts, hipStream_t hs,
                            const uint64 extra_bytes = 0, uint8** extra = NULL, const uint8** quals = NULL, uint64* n_quals = NULL)
{
    const uint32 n = stream.size();
    const uint32 maxP = maxP_of(stream);
    const uint64 table = (job_table::bytes(n) + 15u) & ~uint64(15);
    const uint64 stage = R::staged ? ((job_table::stage_bytes(n, maxP, R::stage_quals) + 15u) & ~uint64(15)) : 0u;
    uint8* base = buf.reserve(table + stage + extra_bytes + 16u, hs);
    t.carve(base, n);
    if (R::staged)
    {
        const uint64 sym = uint64(n) * job_table::stride_for(maxP) + 64u;
        t.stage_stride = job_table::stride_for(maxP);
        t.stage_words  = reinterpret_cast<uint32*>(base + table);
        t.stage_quals  = R::stage_quals ? base + table + ((sym / 2u + 15u) & ~uint64(15)) : nullptr;
        if (quals)   *quals   = t.stage_quals;
        if (n_quals) *n_quals = R::stage_quals ? sym : 0u;
    }
    if (extra) *extra = base + table + stage;
    hipLaunchKernelGGL(init_bounds_kernel, dim3(1), dim3(64), 0, hs, t.bounds);
    hipLaunchKernelGGL((describe_jobs_kernel<stream_type, R>), dim3(uint32(std::min<uint64>((uint64(n) * describe_lanes<stream_type, R>::value + 127u) / 128u, 65536u))), dim3(128), 0, hs, stream, t);
    unsigned long long b[6];
    { unsigned long long* pin = pinned_words();
      check(hipMemcpyAsync(pin ? pin : b, t.bounds, sizeof(b), hipMemcpyDeviceToHost, hs), "hipMemcpyAsync");
      check(hipStreamSynchronize(hs), "hipStreamSynchronize");
      if (pin) memcpy(b, pin, sizeof(b)); }
    t.no_thresholds = (b[5] != 0ull) && static_cast<long long>(b[5]) - (1ll << 31) <= -(1ll << 29);
    if (b[1] == 0ull) { b[0] = 0ull; b[1] = 1ull; }          // no job has a pattern / text: any valid range will do
    if (b[3] == 0ull) { b[2] = 0ull; b[3] = 1ull; }
    typedef pattern_where<typename R::pattern_type, !R::staged> pwhere;
    const uint32 pper = 32u / pwhere::BITS;
    hipLaunchKernelGGL(rebase_jobs_kernel, dim3((n + 255u) / 256u), dim3(256), 0, hs, n, t.pat_begin, R::staged ? uint64(0) : uint64(b[0]) * pper, t.txt_begin, uint64(b[2]) * 16u);
    if (R::staged) { ps.words = t.stage_words; ps.n_words = (uint64(n) * t.stage_stride + 64u) / 8u; ps.bits = 4u; ps.big_endian = 0u; }
    else           { ps.words = reinterpret_cast<const uint32*>(uintptr_t(b[0]) * 4u); ps.n_words = b[1] - b[0]; ps.bits = pwhere::BITS; ps.big_endian = pwhere::BE ? 1u : 0u; }
    ps.begin = t.pat_begin; ps.length = t.pat_len; ps.fixed_length = 0; ps._pad = 0;
    ts.words = reinterpret_cast<const uint32*>(uintptr_t(b[2]) * 4u); ts.n_words = b[3] - b[2];
    ts.bits = 2u; ts.big_endian = packed_view<typename R::text_type>::BE ? 1u : 0u;
    ts.begin = t.txt_begin; ts.length = t.txt_len; ts.fixed_length = 0; ts._pad = 0;
}

/// the view route: describe -> (sync: the bounds) -> rebase.  false when the jobs' qualities do not sit at one common distance from their
/// symbols (several read batches behind one stream): the caller then stages, as before.
template <typename stream_type, typename R = recognised<stream_type> >
inline bool build_view_table(const stream_type& stream, device_buffer& buf, job_table& t, nvbio_hip_string_set& ps, nvbio_hip_string_set& ts,
                             const uint8** quals, uint64* n_quals, const uint8** flags, hipStream_t hs, uint32* maxP = NULL, uint32* maxT = NULL)
{
    typedef typename pattern_source<typename R::pattern_type>::where_type where_type;
    const uint32 n = stream.size();
    const uint64 table = (job_table::bytes(n) + 15u) & ~uint64(15);
    t.carve(buf.reserve(table + 16u, hs), n);
    hipLaunchKernelGGL(init_bounds_kernel, dim3(1), dim3(64), 0, hs, t.bounds);
    hipLaunchKernelGGL((describe_views_kernel<stream_type, R>), dim3(std::min<uint32>((n + 255u) / 256u, 8192u)), dim3(256), 0, hs, stream, t);
    unsigned long long b[12];
    { unsigned long long* pin = pinned_words();
      check(hipMemcpyAsync(pin ? pin : b, t.bounds, sizeof(b), hipMemcpyDeviceToHost, hs), "hipMemcpyAsync");
      check(hipStreamSynchronize(hs), "hipStreamSynchronize");
      if (pin) memcpy(b, pin, sizeof(b)); }
    if (maxP) *maxP = uint32(b[9]);
    if (maxT) *maxT = uint32(b[11]);
    if (b[1] == 0ull || b[4] != b[5]) return false;
    if (b[3] == 0ull) { b[2] = 0ull; b[3] = 1ull; }
    const uint32 per = 32u / where_type::BITS;
    hipLaunchKernelGGL(rebase_jobs_kernel, dim3((n + 255u) / 256u), dim3(256), 0, hs, n, t.pat_begin, uint64(b[0]) * per, t.txt_begin, uint64(b[2]) * 16u);
    ps.words = reinterpret_cast<const uint32*>(uintptr_t(b[0]) * 4u); ps.n_words = b[1] - b[0]; ps.bits = where_type::BITS; ps.big_endian = where_type::BE ? 1u : 0u;
    ps.begin = t.pat_begin; ps.length = t.pat_len; ps.fixed_length = 0; ps._pad = 0;
    ts.words = reinterpret_cast<const uint32*>(uintptr_t(b[2]) * 4u); ts.n_words = b[3] - b[2];
    ts.bits = 2u; ts.big_endian = packed_view<typename R::text_type>::BE ? 1u : 0u;
    ts.begin = t.txt_begin; ts.length = t.txt_len; ts.fixed_length = 0; ts._pad = 0;
    // quals[begin + k] must be the quality of stored symbol begin + k: the array starts where the lowest pattern word's first symbol would.
    // That address is never below the caller's quality storage: a quality string runs parallel to its symbol stream from symbol 0 (b[4] ==
    // b[5]: one offset for every job), symbol 0 opens a word, so the first symbol of ANY word of the stream has a quality byte -- when the
    // lowest read starts mid-word the bytes in front of it are an earlier read's.  Positions below the array (a reversed read at symbol 0)
    // are never fetched: fetch_quals16_signed supplies zeros for them (csrc/banded_gotoh_impl.h).
    *quals = reinterpret_cast<const uint8*>(uintptr_t(b[4] + uint64(b[0]) * per));
    *n_quals = b[7] - uint64(b[0]) * per;
    *flags = t.ok;
    return true;
}

/// the largest |cost| of an aligner's scheme (to bound what an int16 sink can hold)
inline int64 abs64(const int32 v) { return v < 0 ? -int64(v) : int64(v); }
#endif // NVBIO_HIP_COMPAT_TUNED

template <uint32 BAND_LEN> struct tuned_band { static const bool value = (BAND_LEN == 3u || BAND_LEN == 5u || BAND_LEN == 7u || BAND_LEN == 15u || BAND_LEN == 31u); };

} // namespace priv

#if defined(NVBIO_HIP_COMPAT_TUNED)
namespace priv {
/// the scheme of a recognised stream as the C-ABI takes it, and the call of the matching entry point
template <typename stream_type>
struct tuned_scheme
{
    typedef typename stream_type::aligner_type aligner_type;
    typedef tuned_aligner<aligner_type>        TA;
    int32 sc[4]; nvbio_hip_gotoh_qual_scheme q; int64 A;

    /// false: the scheme's values are outside the tuned kernels' contract -> generic lane
    bool init(const stream_type& stream)
    {
        A = 0;
        if constexpr (TA::QUAL)
        {
            if (!TA::table(stream.aligner(), q)) return false;
            A = std::max(std::max(abs64(q.match), abs64(q.pattern_gap_open)), std::max(abs64(q.pattern_gap_ext), std::max(abs64(q.text_gap_open), abs64(q.text_gap_ext))));
            for (int i = 0; i < 256; ++i) A = std::max(A, abs64(q.mismatch[i]));
        }
        else
        {
            TA::scheme4(stream.aligner(), sc);
            for (int i = 0; i < 4; ++i) A = std::max(A, abs64(sc[i]));
        }
        return true;
    }
    /// an int16 sink holds every score of the batch (the reference narrows each reported score to the sink's type)
    template <typename sink_type>
    bool sink_fits(const stream_type& stream) const
    { return !best_sink<sink_type>::narrow || (int64(maxP_of(stream)) + maxT_of(stream) + 2) * A < 32000; }

    int banded_score(const uint32 band, const stream_type& stream, const job_table& t, const nvbio_hip_string_set& ps, const nvbio_hip_string_set& ts,
                     const uint8* quals, const uint64 n_quals, hipStream_t hs) const
    {
        const uint32 n = stream.size(), maxP = maxP_of(stream), maxT = maxT_of(stream);
        if constexpr (TA::QUAL) return nvbio_hip_banded_gotoh_score_qual(&q, int32(aligner_type::TYPE), band, &ps, quals, n_quals, &ts, maxP, maxT, n, t.score, t.sink, hs);
        else if (TA::KIND == NVBIO_HIP_GOTOH_ALIGNER) { const nvbio_hip_gotoh_scheme g = { sc[0], sc[1], sc[2], sc[3] };
            return nvbio_hip_banded_gotoh_score(&g, int32(aligner_type::TYPE), band, &ps, &ts, maxP, maxT, n, t.score, t.sink, hs); }
        else { const nvbio_hip_sw_scheme w = { sc[0], sc[1], sc[2], sc[3] };
            return nvbio_hip_banded_sw_score(&w, int32(aligner_type::TYPE), band, &ps, &ts, maxP, maxT, n, t.score, t.sink, hs); }
    }
    int full_score(const stream_type& stream, const job_table& t, const nvbio_hip_string_set& ps, const nvbio_hip_string_set& ts,
                   const uint8* quals, const uint64 n_quals, hipStream_t hs) const
    {
        const uint32 n = stream.size(), maxP = maxP_of(stream), maxT = maxT_of(stream);
        const int32 algo = TA::TEXT_BLOCKING ? NVBIO_HIP_TEXT_BLOCKING : NVBIO_HIP_PATTERN_BLOCKING;
        // thresholds that can never bind are not passed on (a non-negative match score: with a negative one even -2^30 takes part in the
        // reference's exit test for empty patterns, gotoh_inl.h:1203-1207)
        if constexpr (TA::QUAL) return nvbio_hip_alignment_score_qual(&q, algo, int32(aligner_type::TYPE), &ps, quals, n_quals, &ts, maxP, maxT, (t.no_thresholds && q.match >= 0) ? nullptr : t.min_score, n, t.score, t.sink, t.ok, hs);
        else return nvbio_hip_alignment_score(TA::KIND, algo, sc, int32(aligner_type::TYPE), &ps, &ts, maxP, maxT, (t.no_thresholds && sc[0] >= 0) ? nullptr : t.min_score, n, t.score, t.sink, t.ok, hs);
    }
    /// band == 0: full matrix
    int traceback(const uint32 band, const stream_type& stream, const job_table& t, const nvbio_hip_string_set& ps, const nvbio_hip_string_set& ts,
                  const uint8* quals, const uint64 n_quals, uint32* source, uint16* cigar, const uint32 stride, uint32* cigar_len, uint8* temp, const uint64 tb, hipStream_t hs) const
    {
        const uint32 n = stream.size(), maxP = maxP_of(stream), maxT = maxT_of(stream);
        const int32 ty = int32(aligner_type::TYPE);
        if constexpr (TA::QUAL)
            return band ? nvbio_hip_banded_gotoh_traceback_qual(&q, ty, band, &ps, quals, n_quals, &ts, maxP, maxT, n, t.score, t.sink, source, cigar, stride, cigar_len, temp, tb, hs)
                        : nvbio_hip_gotoh_traceback_qual(&q, ty, &ps, quals, n_quals, &ts, maxP, maxT, n, t.score, t.sink, source, cigar, stride, cigar_len, temp, tb, hs);
        else if (TA::KIND == NVBIO_HIP_GOTOH_ALIGNER) { const nvbio_hip_gotoh_scheme g = { sc[0], sc[1], sc[2], sc[3] };
            return band ? nvbio_hip_banded_gotoh_traceback(&g, ty, band, &ps, &ts, maxP, maxT, n, t.score, t.sink, source, cigar, stride, cigar_len, temp, tb, hs)
                        : nvbio_hip_gotoh_traceback(&g, ty, &ps, &ts, maxP, maxT, n, t.score, t.sink, source, cigar, stride, cigar_len, temp, tb, hs); }
        else { const nvbio_hip_sw_scheme w = { sc[0], sc[1], sc[2], sc[3] };
            return band ? nvbio_hip_banded_sw_traceback(&w, ty, band, &ps, &ts, maxP, maxT, n, t.score, t.sink, source, cigar, stride, cigar_len, temp, tb, hs)
                        : nvbio_hip_sw_traceback(&w, ty, &ps, &ts, maxP, maxT, n, t.score, t.sink, source, cigar, stride, cigar_len, temp, tb, hs); }
    }
};
} // namespace priv
#endif

// ---------------------------------------------------------------------------------------- BatchedBandedAlignmentScore
template <uint32 BAND_LEN, typename stream_type, typename algorithm_type = DeviceThreadScheduler>
struct BatchedBandedAlignmentScore
{
    typedef typename stream_type::aligner_type aligner_type;
    static uint64 min_temp_storage(const uint32, const uint32, const uint32) { return 0u; }       // batched_banded_inl.h:107-111,143-147
    static uint64 max_temp_storage(const uint32, const uint32, const uint32) { return 0u; }

    /// enact the batch (batched.h:349-352).  hip_stream: the HIP stream the work is queued on (the reference uses the
    /// default stream of the current device).  last_path() tells which execution ran.
    void enact(stream_type stream, uint64 temp_size = 0u, uint8* temp = NULL
#if defined(__HIPCC__)
               , hipStream_t hip_stream = 0
#endif
               )
    {
        (void)temp_size; (void)temp;
#if defined(__HIPCC__)
        run(stream, algorithm_type(), hip_stream);
#else
        run(stream, algorithm_type());
#endif
    }
    const char* last_path() const { return m_path; }
    BatchedBandedAlignmentScore() : m_path("none") {}

private:
    template <typename any_device_scheduler>
    void run(const stream_type& stream, const any_device_scheduler
#if defined(__HIPCC__)
             , hipStream_t hs
#endif
             )
    {
#if defined(__HIPCC__)
        const uint32 n = stream.size();
        if (n == 0) return;
#if defined(NVBIO_HIP_COMPAT_TUNED)
        // nvBowtie's streams: the description pass of the view route measures the batch's lengths on its way, so the pass (and the host
        // round trip) limits_scope would spend on them is not run; anything the view route declines goes the usual way below
        if constexpr (priv::recognised<stream_type>::value && priv::recognised<stream_type>::view && priv::tuned_band<BAND_LEN>::value)
            if (!priv::forced_generic("banded") && run_views(stream, hs)) return;
#endif
        const priv::limits_scope<stream_type> limits(stream, hs);
        if (!priv::limits_scope<stream_type>::trusted && priv::maxP_of(stream) == 0u) { m_path = "empty"; return; }
        if (priv::forced_generic("banded")) { run_device(stream, hs, std::false_type()); return; }
        run_device(stream, hs, std::integral_constant<bool, priv::recognised<stream_type>::value && priv::tuned_band<BAND_LEN>::value>());
#else
        static_assert(sizeof(any_device_scheduler) == 0, "the device schedulers need a translation unit compiled by hipcc");
#endif
    }
    void run(const stream_type& stream, const HostThreadScheduler
#if defined(__HIPCC__)
             , hipStream_t
#endif
             )
    {
        // the stream's pointers are host pointers here, exactly as with the reference's HostThreadScheduler
        const int64 n = int64(stream.size());
        #pragma omp parallel for schedule(dynamic, 256)
        for (int64 i = 0; i < n; ++i) priv::banded_job<BAND_LEN>(stream, uint32(i));
        m_path = "host";
    }
#if defined(__HIPCC__)
    void run_device(const stream_type& stream, hipStream_t hs, std::false_type)
    {
        const uint32 n = stream.size();
        hipLaunchKernelGGL((priv::batched_banded_score_kernel<BAND_LEN, stream_type>), dim3((n + 127u) / 128u), dim3(128), 0, hs, stream);
        priv::check(hipGetLastError(), "batched_banded_score_kernel");
        m_path = "generic";
    }
#if defined(NVBIO_HIP_COMPAT_TUNED)
    /// the view route on its own: the stored reads are scored through their views, in place, and the batch's lengths come out of the
    /// description pass.  false: not taken (the scheme, the sink or the jobs' quality layout are outside it) -- nothing was output.
    /// (NVBIO_HIP_COMPAT_NO_VIEWS=1 keeps the staged route, for timing the two side by side)
    bool run_views(const stream_type& stream, hipStream_t hs)
    {
        if constexpr (priv::recognised<stream_type>::view)
        {
            const char* no_views = getenv("NVBIO_HIP_COMPAT_NO_VIEWS");
            if (no_views && no_views[0] == '1') return false;
            const uint32 n = stream.size();
            priv::tuned_scheme<stream_type> scheme;
            typedef decltype(typename stream_type::context_type().sink) sink_type;
            if (!scheme.init(stream)) return false;
            priv::job_table t; nvbio_hip_string_set ps, ts; const uint8* quals = NULL; uint64 n_quals = 0; const uint8* flags = NULL;
            uint32 maxP = 0, maxT = 0;
            const bool described = priv::build_view_table(stream, m_jobs, t, ps, ts, &quals, &n_quals, &flags, hs, &maxP, &maxT);
            if (maxP == 0u) { m_path = "empty"; return true; }           // measured: no job has a valid context, nothing to score or output
            if (!described || n_quals < 4u) return false;
            const priv::known_limits_scope limits(stream, maxP, maxT);
            if (!scheme.template sink_fits<sink_type>(stream)) return false;
            const int err = nvbio_hip_banded_gotoh_score_qual_views(&scheme.q, int32(stream_type::aligner_type::TYPE), BAND_LEN, &ps, quals, n_quals, flags, &ts,
                                                                    maxP, maxT, n, t.score, t.sink, hs);
            priv::check(err, "nvbio_hip_banded_gotoh_score_qual_views");
            hipLaunchKernelGGL((priv::output_jobs_kernel<stream_type>), dim3((n + 127u) / 128u), dim3(128), 0, hs, stream, t);
            priv::check(hipGetLastError(), "output_jobs_kernel");
            m_path = "tuned-views";
            return true;
        }
        else return false;
    }
#endif
    void run_device(const stream_type& stream, hipStream_t hs, std::true_type)
    {
#if defined(NVBIO_HIP_COMPAT_TUNED)
        const uint32 n = stream.size();
        priv::tuned_scheme<stream_type> scheme;
        typedef decltype(typename stream_type::context_type().sink) sink_type;
        if (!scheme.init(stream) || !scheme.template sink_fits<sink_type>(stream)) { run_device(stream, hs, std::false_type()); return; }
        priv::job_table t; nvbio_hip_string_set ps, ts; const uint8* quals = NULL; uint64 n_quals = 0;
        priv::build_job_table(stream, m_jobs, t, ps, ts, hs, 0u, NULL, &quals, &n_quals);
        if (ts.words == NULL || ps.words == NULL) { run_device(stream, hs, std::false_type()); return; }      // no job with a text: nothing for the tuned kernels to read
        const int err = scheme.banded_score(BAND_LEN, stream, t, ps, ts, quals, n_quals, hs);
        if (err == 801) { run_device(stream, hs, std::false_type()); return; }       // outside the tuned kernels' contract (a band they are not instantiated for is caught above; this is the belt to that)
        if (err != 0) fprintf(stderr, "compat banded score: err %d band %u n %u maxP %u maxT %u quals %p n_quals %llu ps{words %p n %llu bits %u begin %p len %p} ts{words %p n %llu begin %p len %p}\n", err, BAND_LEN, n,
                              priv::maxP_of(stream), priv::maxT_of(stream), (const void*)quals, (unsigned long long)n_quals, (const void*)ps.words, (unsigned long long)ps.n_words, ps.bits, (const void*)ps.begin, (const void*)ps.length,
                              (const void*)ts.words, (unsigned long long)ts.n_words, (const void*)ts.begin, (const void*)ts.length);
        priv::check(err, "nvbio_hip_banded_score");
        hipLaunchKernelGGL((priv::output_jobs_kernel<stream_type>), dim3((n + 127u) / 128u), dim3(128), 0, hs, stream, t);
        priv::check(hipGetLastError(), "output_jobs_kernel");
        m_path = "tuned";
#else
        run_device(stream, hs, std::false_type());
#endif
    }
    priv::device_buffer m_jobs;
#endif
    const char* m_path;
};

// ---------------------------------------------------------------------------------------- BatchedAlignmentScore
template <typename stream_type, typename algorithm_type = DeviceThreadScheduler>
struct BatchedAlignmentScore
{
    typedef typename stream_type::aligner_type aligner_type;
    /// bytes of boundary columns for a batch: one column of int16 per job (batched_inl.h:236-300 sizes one per thread)
    static uint64 min_temp_storage(const uint32 max_pattern_len, const uint32 max_text_len, const uint32 stream_size)
    { return uint64(stream_size) * column_stride(max_pattern_len, max_text_len) * sizeof(int16); }
    static uint64 max_temp_storage(const uint32 max_pattern_len, const uint32 max_text_len, const uint32 stream_size)
    { return min_temp_storage(max_pattern_len, max_text_len, stream_size); }

    void enact(stream_type stream, uint64 temp_size = 0u, uint8* temp = NULL
#if defined(__HIPCC__)
               , hipStream_t hip_stream = 0
#endif
               )
    {
#if defined(__HIPCC__)
        run(stream, temp_size, temp, algorithm_type(), hip_stream);
#else
        run(stream, temp_size, temp, algorithm_type());
#endif
    }
    const char* last_path() const { return m_path; }
    BatchedAlignmentScore() : m_path("none") {}

private:
    static uint64 column_stride(const uint32 maxP, const uint32 maxT) { return (uint64(priv::column_entries<aligner_type>::get(maxP, maxT)) + 7u) & ~7ull; }

    void run(const stream_type& stream, uint64, uint8*, const HostThreadScheduler
#if defined(__HIPCC__)
             , hipStream_t
#endif
             )
    {
        const int64 n = int64(stream.size());
        const priv::limits_scope<stream_type> limits(stream);
        const uint64 stride = column_stride(priv::maxP_of(stream), priv::maxT_of(stream));
        #pragma omp parallel
        {
            std::vector<int16> column(stride + 8u);
            #pragma omp for schedule(dynamic, 64)
            for (int64 i = 0; i < n; ++i) priv::full_job(stream, uint32(i), column.data());
        }
        m_path = "host";
    }
    template <typename any_device_scheduler>
    void run(const stream_type& stream, uint64 temp_size, uint8* temp, const any_device_scheduler
#if defined(__HIPCC__)
             , hipStream_t hs
#endif
             )
    {
#if defined(__HIPCC__)
        if (stream.size() == 0) return;
        const priv::limits_scope<stream_type> limits(stream, hs);
        if (!priv::limits_scope<stream_type>::trusted && priv::maxP_of(stream) == 0u) { m_path = "empty"; return; }
        if (priv::forced_generic("full")) { run_device(stream, temp_size, temp, hs, std::false_type()); return; }
        run_device(stream, temp_size, temp, hs, std::integral_constant<bool, priv::recognised<stream_type>::value>());
#else
        static_assert(sizeof(any_device_scheduler) == 0, "the device schedulers need a translation unit compiled by hipcc");
#endif
    }
#if defined(__HIPCC__)
    void run_device(const stream_type& stream, uint64 temp_size, uint8* temp, hipStream_t hs, std::false_type)
    {
        const uint32 n = stream.size();
        const uint64 stride = column_stride(priv::maxP_of(stream), priv::maxT_of(stream));
        const uint64 need = uint64(n) * stride * sizeof(int16);
        if (temp == NULL || temp_size < need) temp = m_columns.reserve(need, hs);          // batched_inl.h:402-408: allocate when the caller gave none
        hipLaunchKernelGGL((priv::batched_full_score_kernel<stream_type>), dim3((n + 127u) / 128u), dim3(128), 0, hs, stream, reinterpret_cast<int16*>(temp), stride);
        priv::check(hipGetLastError(), "batched_full_score_kernel");
        m_path = "generic";
    }
    void run_device(const stream_type& stream, uint64 temp_size, uint8* temp, hipStream_t hs, std::true_type)
    {
#if defined(NVBIO_HIP_COMPAT_TUNED)
        const uint32 n = stream.size();
        priv::tuned_scheme<stream_type> scheme;
        typedef decltype(typename stream_type::context_type().sink) sink_type;
        // (patterns beyond the 1 024 rows a wave holds run in stripes: csrc/full_gotoh_striped.hip)
        if (!scheme.init(stream) || !scheme.template sink_fits<sink_type>(stream)) { run_device(stream, temp_size, temp, hs, std::false_type()); return; }
        priv::job_table t; nvbio_hip_string_set ps, ts; const uint8* quals = NULL; uint64 n_quals = 0;
        priv::build_job_table(stream, m_jobs, t, ps, ts, hs, 0u, NULL, &quals, &n_quals);
        if (ts.words == NULL || ps.words == NULL) { run_device(stream, temp_size, temp, hs, std::false_type()); return; }
        const int err = scheme.full_score(stream, t, ps, ts, quals, n_quals, hs);
        if (err == 801) { run_device(stream, temp_size, temp, hs, std::false_type()); return; }
        priv::check(err, "nvbio_hip_alignment_score");
        hipLaunchKernelGGL((priv::output_jobs_kernel<stream_type>), dim3((n + 127u) / 128u), dim3(128), 0, hs, stream, t);
        priv::check(hipGetLastError(), "output_jobs_kernel");
        m_path = "tuned";
#else
        run_device(stream, temp_size, temp, hs, std::false_type());
#endif
    }
    priv::device_buffer m_jobs, m_columns;
#endif
    const char* m_path;
};

// ---------------------------------------------------------------------------------------- tracebacks
// BatchedBandedAlignmentTraceback / BatchedAlignmentTraceback (batched.h:432-476) over the reference's traceback stream concept
// (context_type {min_score, backtracer, alignment}; batched_banded_inl.h:248-451, batched_inl.h:610-850), with the same three
// executions as the scoring classes:
//   * tuned   -- recognised streams (packed strings in place, or staged through the stream's own iterators as for scoring; the
//                library's Simple*Scheme / edit-distance aligners or a scheme of nvBowtie's concept; any backtracer with clip(n) /
//                push(op)): the C-ABI traceback produces the run-length CIGAR, and a second kernel replays it into the stream's own
//                backtracer and output();
//   * generic -- any other stream (8-bit strings, user schemes, asymmetric linear gaps): one lane per job runs the templates of
//                traceback.h on the stream's iterators, flow flags in the batch object's scratch;
//   * host    -- HostThreadScheduler: the same templates under OpenMP over host pointers.
// CHECKPOINTS is accepted and ignored (every execution keeps the whole flow matrix of a job).
namespace priv {

template <uint32 BAND_LEN, typename stream_type>
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE void banded_traceback_job(stream_type& stream, const uint32 i, uint8* flags)
{
    priv::fresh_context<typename stream_type::context_type> ctx_storage; typename stream_type::context_type& ctx = ctx_storage.get();
    typename stream_type::strings_type strings;
    if (!stream.init_context(i, &ctx)) { stream.output(i, &ctx); return; }          // batched_banded_inl.h:262-268: declined jobs are output as they are
    const uint32 len = stream.pattern_length(i, &ctx);
    stream.load_strings(i, 0u, len, &ctx, &strings);
    ctx.alignment = banded_traceback<BAND_LEN>(stream.aligner(), strings.pattern, strings.quals, strings.text, ctx.backtracer, flags);
    stream.output(i, &ctx);
}
template <typename stream_type>
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE void full_traceback_job(stream_type& stream, const uint32 i, uint8* scratch, const uint32 maxP, const uint32 maxT)
{
    priv::fresh_context<typename stream_type::context_type> ctx_storage; typename stream_type::context_type& ctx = ctx_storage.get();
    typename stream_type::strings_type strings;
    if (!stream.init_context(i, &ctx)) { stream.output(i, &ctx); return; }
    const uint32 len = stream.pattern_length(i, &ctx);
    stream.load_strings(i, 0u, len, &ctx, &strings);
    uint8* flags = scratch;
    int32* rows  = reinterpret_cast<int32*>(scratch + full_traceback_flag_bytes(maxP, maxT));
    int16* col   = reinterpret_cast<int16*>(scratch + full_traceback_flag_bytes(maxP, maxT) + full_traceback_row_bytes(maxP));
    ctx.alignment = matrix_traceback(stream.aligner(), strings.pattern, strings.quals, strings.text, ctx.backtracer, flags, rows, col);
    stream.output(i, &ctx);
}

#if defined(__HIPCC__)
template <uint32 BAND_LEN, typename stream_type>
__global__ void __launch_bounds__(128) batched_banded_traceback_kernel(stream_type stream, uint8* scratch, const uint64 stride)     // by value, not const: a traceback stream's output() may be non-const (traceback_inl.h:140)
{
    const uint32 i = blockIdx.x * 128u + threadIdx.x;
    if (i < stream.size()) banded_traceback_job<BAND_LEN>(stream, i, scratch + uint64(i) * stride);
}
template <typename stream_type>
__global__ void __launch_bounds__(128) batched_full_traceback_kernel(stream_type stream, uint8* scratch, const uint64 stride, const uint32 maxP, const uint32 maxT)
{
    const uint32 i = blockIdx.x * 128u + threadIdx.x;
    if (i < stream.size()) full_traceback_job(stream, i, scratch + uint64(i) * stride, maxP, maxT);
}
#endif

/// BAND_LEN == 0: full matrix
template <uint32 BAND_LEN, typename stream_type>
struct traceback_runner
{
    typedef typename stream_type::aligner_type aligner_type;
    traceback_runner() : m_path("none") {}

    void run_host(const stream_type& in_stream)
    {
        stream_type stream(in_stream);
        const int64 n = int64(stream.size());
        const limits_scope<stream_type> limits(stream);
        const uint32 maxP = maxP_of(stream), maxT = maxT_of(stream);
        const uint64 bytes = BAND_LEN ? banded_traceback_scratch(BAND_LEN ? BAND_LEN : 1u, maxP) : full_traceback_scratch(maxP, maxT);
        #pragma omp parallel
        {
            std::vector<uint64> scratch(bytes / 8u + 2u);
            #pragma omp for schedule(dynamic, 64)
            for (int64 i = 0; i < n; ++i) job(stream, uint32(i), reinterpret_cast<uint8*>(scratch.data()), maxP, maxT);
        }
        m_path = "host";
    }
#if defined(__HIPCC__)
    void run(const stream_type& stream, hipStream_t hs)
    {
        if (stream.size() == 0) return;
        const limits_scope<stream_type> limits(stream, hs);
        if (!limits_scope<stream_type>::trusted && maxP_of(stream) == 0u) { m_path = "empty"; return; }       // measured: no job has a valid context, nothing to trace or output
        if (forced_generic("traceback")) { run_device(stream, hs, std::false_type()); return; }
        run_device(stream, hs, std::integral_constant<bool, recognised_tb<stream_type>::value && (BAND_LEN == 0u || tuned_band<BAND_LEN ? BAND_LEN : 3u>::value)>());
    }
    void run_device(const stream_type& stream, hipStream_t hs, std::false_type)
    {
        const uint32 n = stream.size(), maxP = maxP_of(stream), maxT = maxT_of(stream);
        const uint64 stride = BAND_LEN ? banded_traceback_scratch(BAND_LEN ? BAND_LEN : 1u, maxP) : full_traceback_scratch(maxP, maxT);
        uint8* scratch = m_temp.reserve(uint64(n) * stride + 16u, hs);
        launch_generic(stream, scratch, stride, maxP, maxT, hs, std::integral_constant<bool, BAND_LEN != 0u>());
        check(hipGetLastError(), "batched_traceback_kernel");
        m_path = "generic";
    }
    void launch_generic(const stream_type& stream, uint8* scratch, const uint64 stride, uint32, uint32, hipStream_t hs, std::true_type)
    { hipLaunchKernelGGL((batched_banded_traceback_kernel<(BAND_LEN ? BAND_LEN : 3u), stream_type>), dim3((stream.size() + 127u) / 128u), dim3(128), 0, hs, stream, scratch, stride); }
    void launch_generic(const stream_type& stream, uint8* scratch, const uint64 stride, uint32 maxP, uint32 maxT, hipStream_t hs, std::false_type)
    { hipLaunchKernelGGL((batched_full_traceback_kernel<stream_type>), dim3((stream.size() + 127u) / 128u), dim3(128), 0, hs, stream, scratch, stride, maxP, maxT); }

    void run_device(const stream_type& stream, hipStream_t hs, std::true_type)
    {
#if defined(NVBIO_HIP_COMPAT_TUNED)
        const uint32 n = stream.size();
        const uint32 band = BAND_LEN;
        const uint32 maxP = maxP_of(stream), maxT = maxT_of(stream);
        tuned_scheme<stream_type> scheme;
        if (!scheme.init(stream)) { run_device(stream, hs, std::false_type()); return; }
        const uint32 stride = band ? maxP + band + 4u : maxP + maxT + 4u;          // run-length words never exceed the walk's length
        const uint64 extra = uint64(n) * (8u + 4u + uint64(stride) * 2u) + 64u;
        job_table t; nvbio_hip_string_set ps, ts; uint8* x = NULL; const uint8* quals = NULL; uint64 n_quals = 0;
        build_job_table<stream_type, recognised_tb<stream_type> >(stream, m_jobs, t, ps, ts, hs, extra, &x, &quals, &n_quals);
        if (ts.words == NULL || ps.words == NULL) { run_device(stream, hs, std::false_type()); return; }
        uint32* source = reinterpret_cast<uint32*>(x);
        uint32* cigar_len = source + 2u * uint64(n);
        uint16* cigar = reinterpret_cast<uint16*>(cigar_len + n);
        const uint64 tb = band ? nvbio_hip_banded_gotoh_traceback_temp_bytes(band, maxP, n) : nvbio_hip_gotoh_traceback_temp_bytes(maxP, maxT, n);
        uint8* temp = m_temp.reserve(tb + 16u, hs);
        const int err = scheme.traceback(band, stream, t, ps, ts, quals, n_quals, source, cigar, stride, cigar_len, temp, tb, hs);
        if (err == 801) { run_device(stream, hs, std::false_type()); return; }     // e.g. asymmetric linear gaps (scores run tuned, tracebacks do not), 8-bit patterns, values beyond int16
        if (err != 0) fprintf(stderr, "compat traceback: err %d band %u n %u maxP %u maxT %u stride %u tb %llu quals %p n_quals %llu ps{words %p n %llu bits %u} ts{words %p n %llu}\n", err, band, n, maxP, maxT, stride,
                              (unsigned long long)tb, (const void*)quals, (unsigned long long)n_quals, (const void*)ps.words, (unsigned long long)ps.n_words, ps.bits, (const void*)ts.words, (unsigned long long)ts.n_words);
        check(err, "nvbio_hip_*_traceback");
        hipLaunchKernelGGL((replay_tracebacks_kernel<stream_type>), dim3((n + 127u) / 128u), dim3(128), 0, hs, stream, t, source, cigar, stride, cigar_len);
        check(hipGetLastError(), "replay_tracebacks_kernel");
        m_path = "tuned";
#else
        run_device(stream, hs, std::false_type());
#endif
    }
    device_buffer m_jobs, m_temp;
#endif
    const char* m_path;

private:
    static void job(stream_type& stream, const uint32 i, uint8* scratch, const uint32 maxP, const uint32 maxT)
    { job(stream, i, scratch, maxP, maxT, std::integral_constant<bool, BAND_LEN != 0u>()); }
    static void job(stream_type& stream, const uint32 i, uint8* scratch, uint32, uint32, std::true_type) { banded_traceback_job<(BAND_LEN ? BAND_LEN : 3u)>(stream, i, scratch); }
    static void job(stream_type& stream, const uint32 i, uint8* scratch, uint32 maxP, uint32 maxT, std::false_type) { full_traceback_job(stream, i, scratch, maxP, maxT); }
};
} // namespace priv

template <uint32 BAND_LEN, uint32 CHECKPOINTS, typename stream_type, typename algorithm_type = DeviceThreadScheduler>
struct BatchedBandedAlignmentTraceback
{
    static_assert(BAND_LEN >= 2u, "a band of at least two cells");
    static uint64 min_temp_storage(const uint32, const uint32, const uint32) { return 0u; }        // the batch object owns its storage
    static uint64 max_temp_storage(const uint32, const uint32, const uint32) { return 0u; }
    void enact(stream_type stream, uint64 = 0u, uint8* = NULL
#if defined(__HIPCC__)
               , hipStream_t hip_stream = 0
#endif
               )
    {
#if defined(__HIPCC__)
        dispatch(stream, hip_stream, std::integral_constant<bool, same_type<algorithm_type, HostThreadScheduler>::pred>());
#else
        static_assert(same_type<algorithm_type, HostThreadScheduler>::pred, "the device schedulers need a translation unit compiled by hipcc");
        m_run.run_host(stream);
#endif
    }
    const char* last_path() const { return m_run.m_path; }
private:
#if defined(__HIPCC__)
    void dispatch(const stream_type& stream, hipStream_t, std::true_type)     { m_run.run_host(stream); }       // only instantiated for host streams
    void dispatch(const stream_type& stream, hipStream_t hs, std::false_type) { m_run.run(stream, hs); }
#endif
    priv::traceback_runner<BAND_LEN, stream_type> m_run;
};
template <uint32 CHECKPOINTS, typename stream_type, typename algorithm_type = DeviceThreadScheduler>
struct BatchedAlignmentTraceback
{
    static uint64 min_temp_storage(const uint32, const uint32, const uint32) { return 0u; }
    static uint64 max_temp_storage(const uint32, const uint32, const uint32) { return 0u; }
    void enact(stream_type stream, uint64 = 0u, uint8* = NULL
#if defined(__HIPCC__)
               , hipStream_t hip_stream = 0
#endif
               )
    {
#if defined(__HIPCC__)
        dispatch(stream, hip_stream, std::integral_constant<bool, same_type<algorithm_type, HostThreadScheduler>::pred>());
#else
        static_assert(same_type<algorithm_type, HostThreadScheduler>::pred, "the device schedulers need a translation unit compiled by hipcc");
        m_run.run_host(stream);
#endif
    }
    const char* last_path() const { return m_run.m_path; }
private:
#if defined(__HIPCC__)
    void dispatch(const stream_type& stream, hipStream_t, std::true_type)     { m_run.run_host(stream); }       // only instantiated for host streams
    void dispatch(const stream_type& stream, hipStream_t hs, std::false_type) { m_run.run(stream, hs); }
#endif
    priv::traceback_runner<0u, stream_type> m_run;
};

// ---------------------------------------------------------------------------------------- convenience functions
namespace priv {

/// the stream batch_*_alignment_score builds around two string sets and a sink iterator (batched_inl.h:857-982)
template <typename t_aligner_type, typename pattern_set_type, typename text_set_type, typename sink_iterator>
struct StringSetAlignmentStream
{
    typedef t_aligner_type                          aligner_type;
    typedef typename pattern_set_type::string_type  pattern_string;
    typedef typename text_set_type::string_type     text_string;
    typedef typename std::iterator_traits<sink_iterator>::value_type sink_type;
    struct context_type { int32 min_score; sink_type sink; };
    struct strings_type { pattern_string pattern; trivial_quality_string quals; text_string text; };

    StringSetAlignmentStream(aligner_type aligner, uint32 count, pattern_set_type patterns, text_set_type texts, sink_iterator sinks, uint32 maxP, uint32 maxT)
        : m_aligner(aligner), m_count(count), m_patterns(patterns), m_texts(texts), m_sinks(sinks), m_max_pattern_len(maxP), m_max_text_len(maxT) {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE const aligner_type& aligner() const { return m_aligner; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 max_pattern_length() const { return m_max_pattern_len; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 max_text_length() const { return m_max_text_len; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 size() const { return m_count; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 pattern_length(const uint32 i, context_type*) const { return m_patterns[i].length(); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 text_length(const uint32 i, context_type*) const { return m_texts[i].length(); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool init_context(const uint32, context_type* ctx) const { ctx->min_score = Field_traits<int32>::min(); ctx->sink = sink_type(); return true; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE void load_strings(const uint32 i, const uint32, const uint32, const context_type*, strings_type* s) const
    { s->pattern = m_patterns[i]; s->text = m_texts[i]; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE void output(const uint32 i, const context_type* ctx) const { m_sinks[i] = ctx->sink; }

    aligner_type m_aligner; uint32 m_count; pattern_set_type m_patterns; text_set_type m_texts; sink_iterator m_sinks; uint32 m_max_pattern_len, m_max_text_len;
};

} // namespace priv

/// batch_banded_alignment_score<BAND_LEN>(aligner, patterns, texts, sinks, scheduler, maxP, maxT)   (batched.h:217-231)
template <uint32 BAND_LEN, typename aligner_type, typename pattern_set_type, typename text_set_type, typename sink_iterator, typename scheduler_type>
void batch_banded_alignment_score(const aligner_type aligner, const pattern_set_type patterns, const text_set_type texts, sink_iterator sinks,
                                  const scheduler_type, const uint32 max_pattern_length, const uint32 max_text_length)
{
    typedef priv::StringSetAlignmentStream<aligner_type, pattern_set_type, text_set_type, sink_iterator> stream_type;
    BatchedBandedAlignmentScore<BAND_LEN, stream_type, scheduler_type> batch;
    batch.enact(stream_type(aligner, patterns.size(), patterns, texts, sinks, max_pattern_length, max_text_length));
#if defined(__HIPCC__)
    if (!same_type<scheduler_type, HostThreadScheduler>::pred) priv::check(hipStreamSynchronize(0), "hipStreamSynchronize");    // the batch object's buffers die here
#endif
}
/// batch_alignment_score(aligner, patterns, texts, sinks, scheduler, maxP, maxT)   (batched.h:160-190)
template <typename aligner_type, typename pattern_set_type, typename text_set_type, typename sink_iterator, typename scheduler_type>
void batch_alignment_score(const aligner_type aligner, const pattern_set_type patterns, const text_set_type texts, sink_iterator sinks,
                           const scheduler_type, const uint32 max_pattern_length, const uint32 max_text_length)
{
    typedef priv::StringSetAlignmentStream<aligner_type, pattern_set_type, text_set_type, sink_iterator> stream_type;
    BatchedAlignmentScore<stream_type, scheduler_type> batch;
    batch.enact(stream_type(aligner, patterns.size(), patterns, texts, sinks, max_pattern_length, max_text_length));
#if defined(__HIPCC__)
    if (!same_type<scheduler_type, HostThreadScheduler>::pred) priv::check(hipStreamSynchronize(0), "hipStreamSynchronize");
#endif
}

} // namespace aln
} // namespace nvbio
