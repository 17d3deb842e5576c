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
//   * DeviceThreadScheduler, any other stream: one lane per job runs the generic templates of alignment.h on whatever
//     iterators the stream hands out (8-bit strings, user schemes, quality strings, other sinks ...).
//   * HostThreadScheduler: the same templates over host pointers, OpenMP over jobs (batched_banded_inl.h:97-128).
// DeviceStagedThreadScheduler / DeviceWarpScheduler are accepted as aliases of the device path (they are alternative
// schedules of the same computation in the reference, batched.h:51-76).
#pragma once
#include "alignment.h"
#include "../basic/packedstream.h"
#include "../basic/packedstream_loader.h"
#include "../basic/cuda/ldg.h"
#include "../strings/string_set.h"
#include <stdexcept>
#include <string>
#include <vector>
#include <type_traits>
#if defined(_OPENMP)
#include <omp.h>
#endif
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

struct hip_error : public std::runtime_error {
    int code;
    hip_error(const char* what, int c) : std::runtime_error(std::string(what) + " failed, hipError " + std::to_string(c)), code(c) {}
};

namespace priv {

inline void check(const int err, const char* what) { if (err != 0) throw hip_error(what, err); }

// ---------------------------------------------------------------------------------------- the per-job bodies
template <uint32 BAND_LEN, typename stream_type>
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE void banded_job(const stream_type& stream, const uint32 i)
{
    typename stream_type::context_type ctx;
    typename stream_type::strings_type strings;
    if (!stream.init_context(i, &ctx)) return;
    const uint32 len = stream.pattern_length(i, &ctx);
    stream.load_strings(i, 0u, len, &ctx, &strings);
    banded_alignment_score<BAND_LEN>(stream.aligner(), strings.pattern, strings.quals, strings.text, ctx.min_score, ctx.sink);
    stream.output(i, &ctx);
}
template <typename stream_type>
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE void full_job(const stream_type& stream, const uint32 i, int16* column)
{
    typename stream_type::context_type ctx;
    typename stream_type::strings_type strings;
    if (!stream.init_context(i, &ctx)) return;
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
struct device_buffer
{
    device_buffer() : ptr(nullptr), bytes(0) {}
    ~device_buffer() { if (ptr) (void)hipFree(ptr); }
    device_buffer(const device_buffer&) = delete;
    device_buffer& operator=(const device_buffer&) = delete;
    uint8* reserve(const uint64 n)
    {
        if (n > bytes) { if (ptr) (void)hipFree(ptr); ptr = nullptr; bytes = 0; check(hipMalloc(reinterpret_cast<void**>(&ptr), n), "hipMalloc"); bytes = n; }
        return ptr;
    }
    uint8* ptr; uint64 bytes;
};
#endif

// ---------------------------------------------------------------------------------------- stream recognition
template <typename It> struct word_pointer { static const bool ok = false; };
template <> struct word_pointer<const uint32*> { static const bool ok = true; NVBIO_FORCEINLINE NVBIO_HOST_DEVICE static const uint32* get(const uint32* p) { return p; } };
template <> struct word_pointer<uint32*>       { static const bool ok = true; NVBIO_FORCEINLINE NVBIO_HOST_DEVICE static const uint32* get(uint32* p) { return p; } };
template <> struct word_pointer< cuda::ldg_pointer<uint32> > { static const bool ok = true; NVBIO_FORCEINLINE NVBIO_HOST_DEVICE static const uint32* get(cuda::ldg_pointer<uint32> p) { return p.base; } };
template <typename It> struct word_pointer< const_cached_iterator<It> > { static const bool ok = word_pointer<It>::ok;
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE static const uint32* get(const_cached_iterator<It> p) { return word_pointer<It>::get(p.base()); } };

template <typename S> struct packed_view { static const bool ok = false; static const uint32 BITS = 0; static const bool BE = false; };
template <typename I, uint32 B, bool E, typename VI>
struct packed_view< vector_view< PackedStream<I, uint8, B, E, uint32>, VI > >
{
    static const bool   ok   = word_pointer<I>::ok && (B == 2u || B == 4u);
    static const uint32 BITS = B;
    static const bool   BE   = E;
    typedef vector_view< PackedStream<I, uint8, B, E, uint32>, VI > view_type;
    /// first word address (in words from address 0) and symbol offset of the view from it
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE static void where(const view_type& v, uint64& word0, uint32& first)
    {
        const PackedStream<I, uint8, B, E, uint32> ps = v.begin();
        const uint32 per = 32u / B;
        word0 = uint64(reinterpret_cast<uintptr_t>(word_pointer<I>::get(ps.stream()))) / 4u + ps.index() / per;
        first = ps.index() % per;
    }
};

template <typename A> struct simple_aligner { static const bool ok = false; };
#if defined(NVBIO_HIP_COMPAT_TUNED)
template <AlignmentType T, typename G> struct simple_aligner< GotohAligner<T, SimpleGotohScheme, G> > {
    static const bool ok = true; static const int32 KIND = NVBIO_HIP_GOTOH_ALIGNER; static const bool TEXT_BLOCKING = equal<G, TextBlockingTag>::pred;
    static void scheme4(const GotohAligner<T, SimpleGotohScheme, G>& a, int32* v) { v[0] = a.scheme.m_match; v[1] = a.scheme.m_mismatch; v[2] = a.scheme.m_gap_open; v[3] = a.scheme.m_gap_ext; } };
template <AlignmentType T, typename G> struct simple_aligner< SmithWatermanAligner<T, SimpleSmithWatermanScheme, G> > {
    static const bool ok = true; static const int32 KIND = NVBIO_HIP_SW_ALIGNER; static const bool TEXT_BLOCKING = equal<G, TextBlockingTag>::pred;
    static void scheme4(const SmithWatermanAligner<T, SimpleSmithWatermanScheme, G>& a, int32* v) { v[0] = a.scheme.m_match; v[1] = a.scheme.m_mismatch; v[2] = a.scheme.m_deletion; v[3] = a.scheme.m_insertion; } };
template <AlignmentType T, typename G> struct simple_aligner< EditDistanceAligner<T, G> > {
    static const bool ok = true; static const int32 KIND = NVBIO_HIP_SW_ALIGNER; static const bool TEXT_BLOCKING = equal<G, TextBlockingTag>::pred;
    static void scheme4(const EditDistanceAligner<T, G>&, int32* v) { v[0] = 0; v[1] = -1; v[2] = -1; v[3] = -1; } };
#endif

template <typename stream_type>
struct recognised
{
    typedef typename stream_type::strings_type strings_type;
    typedef typename stream_type::context_type context_type;
    typedef decltype(strings_type().pattern) pattern_type;
    typedef decltype(strings_type().text)    text_type;
    typedef decltype(strings_type().quals)   quals_type;
    typedef decltype(context_type().sink)    sink_type;
    static const bool value = packed_view<pattern_type>::ok && packed_view<text_type>::ok && packed_view<text_type>::BITS == 2u &&
                              equal<quals_type, trivial_quality_string>::pred && equal<sink_type, BestSink<int32> >::pred &&
                              simple_aligner<typename stream_type::aligner_type>::ok;
};

/// the same test for a traceback stream (batched.h:359-420: its context holds a backtracer and an Alignment<int32> instead of a sink)
template <typename stream_type>
struct recognised_tb
{
    typedef typename stream_type::strings_type strings_type;
    typedef decltype(strings_type().pattern) pattern_type;
    typedef decltype(strings_type().text)    text_type;
    typedef decltype(strings_type().quals)   quals_type;
    static const bool value = packed_view<pattern_type>::ok && packed_view<text_type>::ok && packed_view<text_type>::BITS == 2u &&
                              equal<quals_type, trivial_quality_string>::pred && simple_aligner<typename stream_type::aligner_type>::ok;
};

#if defined(NVBIO_HIP_COMPAT_TUNED)
/// the job table the tuned kernels consume (structure of arrays in one device buffer)
struct job_table
{
    uint64* pat_begin; uint32* pat_len; uint64* txt_begin; uint32* txt_len; int32* min_score; int32* score; uint32* sink; uint8* ok;
    unsigned long long* bounds;        // [0] lowest pattern word, [1] end pattern word, [2] lowest text word, [3] end text word
    static uint64 bytes(const uint32 n) { return 64u + uint64(n) * (8u + 4u + 8u + 4u + 4u + 4u + 8u + 1u) + 8u * 16u; }
    void carve(uint8* p, const uint32 n)
    {
        bounds = reinterpret_cast<unsigned long long*>(p); p += 64u;
        pat_begin = reinterpret_cast<uint64*>(p); p += uint64(n) * 8u;
        txt_begin = reinterpret_cast<uint64*>(p); p += uint64(n) * 8u;
        sink      = reinterpret_cast<uint32*>(p); p += uint64(n) * 8u;
        pat_len   = reinterpret_cast<uint32*>(p); p += uint64(n) * 4u;
        txt_len   = reinterpret_cast<uint32*>(p); p += uint64(n) * 4u;
        min_score = reinterpret_cast<int32*>(p);  p += uint64(n) * 4u;
        score     = reinterpret_cast<int32*>(p);  p += uint64(n) * 4u;
        ok        = p;
    }
};

__device__ __forceinline__ unsigned long long wave_min(unsigned long long v)
{ for (int o = 32; o > 0; o >>= 1) { const unsigned long long w = __shfl_xor(v, o, 64); v = w < v ? w : v; } return v; }
__device__ __forceinline__ unsigned long long wave_max(unsigned long long v)
{ for (int o = 32; o > 0; o >>= 1) { const unsigned long long w = __shfl_xor(v, o, 64); v = w > v ? w : v; } return v; }

__global__ void init_bounds_kernel(unsigned long long* b) { if (threadIdx.x < 4u) b[threadIdx.x] = (threadIdx.x & 1u) ? 0ull : ~0ull; }

/// evaluate the stream's functors per job: where its strings live, how long they are, its min_score
template <typename stream_type, typename R>
__global__ void __launch_bounds__(128) describe_jobs_kernel(const stream_type stream, const job_table t)
{
    const uint32 i = blockIdx.x * 128u + threadIdx.x;
    unsigned long long plo = ~0ull, phi = 0ull, tlo = ~0ull, thi = 0ull;
    if (i < stream.size())
    {
        typename stream_type::context_type ctx;
        typename stream_type::strings_type strings;
        uint32 pl = 0, tl = 0; uint64 pw = 0, tw = 0; uint32 pf = 0, tf = 0; int32 ms = 0;
        if (stream.init_context(i, &ctx))
        {
            const uint32 len = stream.pattern_length(i, &ctx);
            stream.load_strings(i, 0u, len, &ctx, &strings);
            packed_view<typename R::pattern_type>::where(strings.pattern, pw, pf);
            packed_view<typename R::text_type>::where(strings.text, tw, tf);
            pl = strings.pattern.length(); tl = strings.text.length(); ms = ctx.min_score;
            plo = pw; phi = pw + (pf + pl + 32u / packed_view<typename R::pattern_type>::BITS - 1u) / (32u / packed_view<typename R::pattern_type>::BITS);
            tlo = tw; thi = tw + (tf + tl + 15u) / 16u;
        }
        // offsets are kept absolute (symbols from address 0) until the host knows the lowest word
        t.pat_begin[i] = pw * (32u / packed_view<typename R::pattern_type>::BITS) + pf; t.pat_len[i] = pl;
        t.txt_begin[i] = tw * 16u + tf; t.txt_len[i] = tl; t.min_score[i] = ms;
    }
    plo = wave_min(plo); phi = wave_max(phi); tlo = wave_min(tlo); thi = wave_max(thi);
    if ((threadIdx.x & 63u) == 0u)
    {
        if (phi) { atomicMin(&t.bounds[0], plo); atomicMax(&t.bounds[1], phi); }
        if (thi) { atomicMin(&t.bounds[2], tlo); atomicMax(&t.bounds[3], thi); }
    }
}
__global__ void __launch_bounds__(256) rebase_jobs_kernel(const uint32 n, uint64* pat_begin, const uint64 pat_delta, uint64* txt_begin, const uint64 txt_delta)
{
    const uint32 i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) { pat_begin[i] -= pat_delta; txt_begin[i] -= txt_delta; }
}
/// hand each result to the stream: a job the scorer refused (text shorter than pattern) leaves the fresh sink untouched
template <typename stream_type>
__global__ void __launch_bounds__(128) output_jobs_kernel(const stream_type stream, const job_table t)
{
    const uint32 i = blockIdx.x * 128u + threadIdx.x;
    if (i >= stream.size()) return;
    typename stream_type::context_type ctx;
    if (!stream.init_context(i, &ctx)) return;
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
    typename stream_type::context_type ctx;
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

/// describe -> (sync: two pointers) -> rebase; fills the C-ABI string sets
template <typename stream_type, typename R = recognised<stream_type> >
inline void build_job_table(const stream_type& stream, device_buffer& buf, job_table& t, nvbio_hip_string_set& ps, nvbio_hip_string_set& ts, hipStream_t hs,
                            const uint64 extra_bytes = 0, uint8** extra = NULL)
{
    const uint32 n = stream.size();
    uint8* base = buf.reserve(job_table::bytes(n) + extra_bytes + 16u);
    t.carve(base, n);
    if (extra) *extra = base + ((job_table::bytes(n) + 15u) & ~uint64(15));
    hipLaunchKernelGGL(init_bounds_kernel, dim3(1), dim3(64), 0, hs, t.bounds);
    hipLaunchKernelGGL((describe_jobs_kernel<stream_type, R>), dim3((n + 127u) / 128u), dim3(128), 0, hs, stream, t);
    unsigned long long b[4];
    check(hipMemcpyAsync(b, t.bounds, sizeof(b), hipMemcpyDeviceToHost, hs), "hipMemcpyAsync");
    check(hipStreamSynchronize(hs), "hipStreamSynchronize");
    if (b[1] == 0ull) { b[0] = 0ull; b[1] = 1ull; }          // no job has a pattern / text: any valid range will do
    if (b[3] == 0ull) { b[2] = 0ull; b[3] = 1ull; }
    const uint32 pper = 32u / packed_view<typename R::pattern_type>::BITS;
    hipLaunchKernelGGL(rebase_jobs_kernel, dim3((n + 255u) / 256u), dim3(256), 0, hs, n, t.pat_begin, uint64(b[0]) * pper, t.txt_begin, uint64(b[2]) * 16u);
    ps.words = reinterpret_cast<const uint32*>(uintptr_t(b[0]) * 4u); ps.n_words = b[1] - b[0];
    ps.bits = packed_view<typename R::pattern_type>::BITS; ps.big_endian = packed_view<typename R::pattern_type>::BE ? 1u : 0u;
    ps.begin = t.pat_begin; ps.length = t.pat_len; ps.fixed_length = 0; ps._pad = 0;
    ts.words = reinterpret_cast<const uint32*>(uintptr_t(b[2]) * 4u); ts.n_words = b[3] - b[2];
    ts.bits = 2u; ts.big_endian = packed_view<typename R::text_type>::BE ? 1u : 0u;
    ts.begin = t.txt_begin; ts.length = t.txt_len; ts.fixed_length = 0; ts._pad = 0;
}
#endif // NVBIO_HIP_COMPAT_TUNED

template <uint32 BAND_LEN> struct tuned_band { static const bool value = (BAND_LEN == 3u || BAND_LEN == 5u || BAND_LEN == 7u || BAND_LEN == 15u || BAND_LEN == 31u); };

} // namespace priv

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
    void run_device(const stream_type& stream, hipStream_t hs, std::true_type)
    {
#if defined(NVBIO_HIP_COMPAT_TUNED)
        const uint32 n = stream.size();
        priv::job_table t; nvbio_hip_string_set ps, ts;
        priv::build_job_table(stream, m_jobs, t, ps, ts, hs);
        int32 sc[4];
        priv::simple_aligner<aligner_type>::scheme4(stream.aligner(), sc);
        int err;
        if (priv::simple_aligner<aligner_type>::KIND == NVBIO_HIP_GOTOH_ALIGNER) {
            const nvbio_hip_gotoh_scheme g = { sc[0], sc[1], sc[2], sc[3] };
            err = nvbio_hip_banded_gotoh_score(&g, int32(aligner_type::TYPE), BAND_LEN, &ps, &ts, stream.max_pattern_length(), stream.max_text_length(), n, t.score, t.sink, hs);
        } else {
            const nvbio_hip_sw_scheme w = { sc[0], sc[1], sc[2], sc[3] };
            err = nvbio_hip_banded_sw_score(&w, int32(aligner_type::TYPE), BAND_LEN, &ps, &ts, stream.max_pattern_length(), stream.max_text_length(), n, t.score, t.sink, hs);
        }
        if (err == 801) { run_device(stream, hs, std::false_type()); return; }       // outside the tuned kernels' contract (e.g. asymmetric linear gaps)
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
        const uint64 stride = column_stride(stream.max_pattern_length(), stream.max_text_length());
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
        run_device(stream, temp_size, temp, hs, std::integral_constant<bool, priv::recognised<stream_type>::value>());
#else
        static_assert(sizeof(any_device_scheduler) == 0, "the device schedulers need a translation unit compiled by hipcc");
#endif
    }
#if defined(__HIPCC__)
    void run_device(const stream_type& stream, uint64 temp_size, uint8* temp, hipStream_t hs, std::false_type)
    {
        const uint32 n = stream.size();
        const uint64 stride = column_stride(stream.max_pattern_length(), stream.max_text_length());
        const uint64 need = uint64(n) * stride * sizeof(int16);
        if (temp == NULL || temp_size < need) temp = m_columns.reserve(need);          // batched_inl.h:402-408: allocate when the caller gave none
        hipLaunchKernelGGL((priv::batched_full_score_kernel<stream_type>), dim3((n + 127u) / 128u), dim3(128), 0, hs, stream, reinterpret_cast<int16*>(temp), stride);
        priv::check(hipGetLastError(), "batched_full_score_kernel");
        m_path = "generic";
    }
    void run_device(const stream_type& stream, uint64 temp_size, uint8* temp, hipStream_t hs, std::true_type)
    {
#if defined(NVBIO_HIP_COMPAT_TUNED)
        const uint32 n = stream.size();
        if (stream.max_pattern_length() > 1024u) { run_device(stream, temp_size, temp, hs, std::false_type()); return; }    // the tuned sweep keeps <= 1024 rows in a wave
        priv::job_table t; nvbio_hip_string_set ps, ts;
        priv::build_job_table(stream, m_jobs, t, ps, ts, hs);
        int32 sc[4];
        priv::simple_aligner<aligner_type>::scheme4(stream.aligner(), sc);
        const int err = nvbio_hip_alignment_score(priv::simple_aligner<aligner_type>::KIND,
                            priv::simple_aligner<aligner_type>::TEXT_BLOCKING ? NVBIO_HIP_TEXT_BLOCKING : NVBIO_HIP_PATTERN_BLOCKING,
                            sc, int32(aligner_type::TYPE), &ps, &ts, stream.max_pattern_length(), stream.max_text_length(),
                            t.min_score, n, t.score, t.sink, t.ok, hs);
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
// (context_type {min_score, backtracer, alignment}).  Offered for the streams the tuned kernels recognise -- packed strings in
// device memory, trivial qualities, the library's Simple*Scheme / edit-distance aligners, any user backtracer with clip(n) /
// push(op): the C-ABI traceback produces the run-length CIGAR, and a second kernel replays it into the stream's own backtracer
// and output().  CHECKPOINTS is accepted and ignored (the kernels keep the whole flow matrix).  Other streams and the host
// scheduler are not offered here: there is no generic per-lane traceback template in this layer.
#if defined(NVBIO_HIP_COMPAT_TUNED)
namespace priv {
template <typename stream_type>
struct traceback_runner
{
    typedef typename stream_type::aligner_type aligner_type;
    static_assert(recognised_tb<stream_type>::value, "compat tracebacks take packed-string streams with trivial qualities and Simple*Scheme / edit-distance aligners");

    /// band == 0: full matrix
    void run(const stream_type& stream, const uint32 band, hipStream_t hs)
    {
        const uint32 n = stream.size();
        if (n == 0) return;
        const uint32 maxP = stream.max_pattern_length(), maxT = stream.max_text_length();
        const uint32 stride = band ? maxP + band + 4u : maxP + maxT + 4u;          // run-length words never exceed the walk's length
        const uint64 extra = uint64(n) * (8u + 4u + uint64(stride) * 2u) + 64u;
        job_table t; nvbio_hip_string_set ps, ts; uint8* x = NULL;
        build_job_table<stream_type, recognised_tb<stream_type> >(stream, m_jobs, t, ps, ts, hs, extra, &x);
        uint32* source = reinterpret_cast<uint32*>(x);
        uint32* cigar_len = source + 2u * uint64(n);
        uint16* cigar = reinterpret_cast<uint16*>(cigar_len + n);
        const uint64 tb = band ? nvbio_hip_banded_gotoh_traceback_temp_bytes(band, maxP, n) : nvbio_hip_gotoh_traceback_temp_bytes(maxP, maxT, n);
        uint8* temp = m_temp.reserve(tb + 16u);
        int32 sc[4];
        simple_aligner<aligner_type>::scheme4(stream.aligner(), sc);
        int err;
        if (simple_aligner<aligner_type>::KIND == NVBIO_HIP_GOTOH_ALIGNER) {
            const nvbio_hip_gotoh_scheme g = { sc[0], sc[1], sc[2], sc[3] };
            err = band ? nvbio_hip_banded_gotoh_traceback(&g, int32(aligner_type::TYPE), band, &ps, &ts, maxP, maxT, n, t.score, t.sink, source, cigar, stride, cigar_len, temp, tb, hs)
                       : nvbio_hip_gotoh_traceback(&g, int32(aligner_type::TYPE), &ps, &ts, maxP, maxT, n, t.score, t.sink, source, cigar, stride, cigar_len, temp, tb, hs);
        } else {
            const nvbio_hip_sw_scheme w = { sc[0], sc[1], sc[2], sc[3] };
            err = band ? nvbio_hip_banded_sw_traceback(&w, int32(aligner_type::TYPE), band, &ps, &ts, maxP, maxT, n, t.score, t.sink, source, cigar, stride, cigar_len, temp, tb, hs)
                       : nvbio_hip_sw_traceback(&w, int32(aligner_type::TYPE), &ps, &ts, maxP, maxT, n, t.score, t.sink, source, cigar, stride, cigar_len, temp, tb, hs);
        }
        check(err, "nvbio_hip_*_traceback");
        hipLaunchKernelGGL((replay_tracebacks_kernel<stream_type>), dim3((n + 127u) / 128u), dim3(128), 0, hs, stream, t, source, cigar, stride, cigar_len);
        check(hipGetLastError(), "replay_tracebacks_kernel");
    }
    device_buffer m_jobs, m_temp;
};
} // namespace priv

template <uint32 BAND_LEN, uint32 CHECKPOINTS, typename stream_type, typename algorithm_type = DeviceThreadScheduler>
struct BatchedBandedAlignmentTraceback
{
    static_assert(!equal<algorithm_type, HostThreadScheduler>::pred, "compat tracebacks run on the device schedulers");
    static_assert(priv::tuned_band<BAND_LEN>::value, "bands 3, 5, 7, 15 and 31");
    static uint64 min_temp_storage(const uint32, const uint32, const uint32) { return 0u; }        // the batch object owns its storage
    static uint64 max_temp_storage(const uint32, const uint32, const uint32) { return 0u; }
    void enact(stream_type stream, uint64 = 0u, uint8* = NULL, hipStream_t hip_stream = 0) { m_run.run(stream, BAND_LEN, hip_stream); }
private:
    priv::traceback_runner<stream_type> m_run;
};
template <uint32 CHECKPOINTS, typename stream_type, typename algorithm_type = DeviceThreadScheduler>
struct BatchedAlignmentTraceback
{
    static_assert(!equal<algorithm_type, HostThreadScheduler>::pred, "compat tracebacks run on the device schedulers");
    static uint64 min_temp_storage(const uint32, const uint32, const uint32) { return 0u; }
    static uint64 max_temp_storage(const uint32, const uint32, const uint32) { return 0u; }
    void enact(stream_type stream, uint64 = 0u, uint8* = NULL, hipStream_t hip_stream = 0) { m_run.run(stream, 0u, hip_stream); }
private:
    priv::traceback_runner<stream_type> m_run;
};
#endif // NVBIO_HIP_COMPAT_TUNED

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
    if (!equal<scheduler_type, HostThreadScheduler>::pred) priv::check(hipStreamSynchronize(0), "hipStreamSynchronize");    // the batch object's buffers die here
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
    if (!equal<scheduler_type, HostThreadScheduler>::pred) priv::check(hipStreamSynchronize(0), "hipStreamSynchronize");
#endif
}

} // namespace aln
} // namespace nvbio
