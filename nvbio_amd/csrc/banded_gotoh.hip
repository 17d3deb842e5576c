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
#include "banded_gotoh_impl.h"
#include <mutex>
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
static uint32_t max_len_16bit(int32_t match, int32_t best_pair /* max substitution score */, int64_t A /* max |cost| */, int32_t gap_open, int32_t gap_ext, int type, uint32_t band)
{
    if (gap_open > 0 || gap_ext > 0) return 0;
    if (A == 0) return 0xFFFFFFFFu;
    if (type == NVBIO_HIP_LOCAL) {
        if (match < 0 || A > 100) return 0;                          // 32*4*A stays far above -32768
        if (best_pair <= 0) return 0xFFFFFFFFu;
        return uint32_t(1022 / best_pair);
    }
    const int64_t lim = 15000 / A - int64_t(band) - 2;
    return lim <= 0 ? 0u : uint32_t(lim);
}

} // namespace nvb

template <typename QA>
static int banded_gotoh_dispatch(nvb::GotohParams& p, const QA& qa, int64_t max_abs_cost, int32_t best_pair, int32_t type, uint32_t band_len,
                                 const nvbio_hip_string_set* patterns, hipStream_t s, const char* tag16, const char* tag32, const bool views = false)
{
    using namespace nvb;
    // NVBIO_HIP_FORCE_32BIT=1 disables the 16-bit kernels (used by the tests to cover both widths)
    const uint32_t lim16 = test_switch(SW_FORCE_32BIT) == 1 ? 0u
                         : max_len_16bit(p.match, best_pair, max_abs_cost, std::max(p.gap_open, p.txt_gap_open), std::max(p.gap_ext, p.txt_gap_ext), type, band_len);
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
    if (lim16 > 0 && (!fixed || patterns->fixed_length <= lim16)) {
        p.len_lo = 0; p.len_hi = lim16;
        g_last_kernel = tag16;
        e = launch<QA>(p, qa, type, band_len, true, s);
        if (e != hipSuccess) return e;
    }
    if (lim16 != 0xFFFFFFFFu && (!fixed || patterns->fixed_length > lim16)) {
        p.len_lo = lim16 > 0 ? lim16 + 1 : 0; p.len_hi = 0xFFFFFFFFu;
        if (fixed || lim16 == 0) g_last_kernel = tag32;
        e = launch<QA>(p, qa, type, band_len, false, s);
    }
    return e;
}

static int check_banded_args(int32_t type, uint32_t band_len, const nvbio_hip_string_set* patterns, const nvbio_hip_string_set* texts)
{
    if (!patterns || !texts) return hipErrorInvalidValue;
    if (type < 0 || type > 2) return hipErrorInvalidValue;
    if (!(patterns->bits == 2 || patterns->bits == 4) || texts->bits != 2) return hipErrorNotSupported;
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
    p.n = n; p.out_score = out_score; p.out_sink = out_sink;
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

    GotohParams p;
    p.pat = make_string_set(patterns);
    p.txt = make_string_set(texts);
    p.match = scheme->match; p.mismatch = 0;
    p.gap_open = scheme->pattern_gap_open; p.gap_ext = scheme->pattern_gap_ext;
    p.txt_gap_open = scheme->text_gap_open; p.txt_gap_ext = scheme->text_gap_ext;
    p.n = n; p.out_score = out_score; p.out_sink = out_sink;
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

// SmithWatermanAligner / EditDistanceAligner in the band (sw_banded_inl.h:340-520): with deletion == insertion the
// recurrence is the Gotoh one with gap_open == gap_ext cell for cell (H >= E, F), same band geometry, same reports.
NVB_API int nvbio_hip_banded_sw_score(
    const nvbio_hip_sw_scheme* scheme, int32_t type, uint32_t band_len,
    const nvbio_hip_string_set* patterns, const nvbio_hip_string_set* texts,
    uint32_t max_pattern_len, uint32_t max_text_len,
    uint32_t n, int32_t* out_score, uint32_t* out_sink, void* stream)
{
    if (!scheme) return hipErrorInvalidValue;
    if (scheme->deletion != scheme->insertion) return hipErrorNotSupported;
    const nvbio_hip_gotoh_scheme g = { scheme->match, scheme->mismatch, scheme->deletion, scheme->deletion };
    return nvbio_hip_banded_gotoh_score(&g, type, band_len, patterns, texts, max_pattern_len, max_text_len, n, out_score, out_sink, stream);
}

// Device memory for the C++ host layer's containers.  A private, per-device stream-ordered pool (the device's default pool and
// its attributes are left alone: this library shares its process with torch's allocator in bench.py) with a bounded release
// threshold, so freed blocks above 256 MiB go back to the driver.  nvbio_hip_device_free keeps hipFree's contract -- the block
// may be in use by ANY stream of the device until the call returns, so the device is synchronised before the block re-enters the
// pool; a driver's per-batch working set lives in a hip::device_arena (include/nvbio_hip/types.h) and never comes through here.
namespace nvb {
static hipMemPool_t private_pool(int dev)
{
    static std::mutex mtx;
    static hipMemPool_t pools[64] = {};
    static bool tried[64] = {};
    if (dev < 0 || dev >= 64) return nullptr;
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
} // namespace nvb
// nvbio_hip_device_free keeps hipFree's contract for the streams this library knows -- the legacy default stream (and through it every blocking
// stream) and the streams made by nvbio_hip_stream_create -- WITHOUT stopping the host: the block is parked with one event recorded in each of
// those streams and goes back to the pool (on the default stream, by which time nothing uses it) once every event has completed; the parked
// blocks are polled at the next malloc / free.  The calling thread returns at once; another driver thread's batch in flight is not waited for.
// Nothing is ever freed on, or waited for by, a stream other than the default one: the pool sees plain same-stream alloc / free traffic.
// (Work queued on a non-blocking stream created elsewhere is not covered: synchronise such a stream before freeing.  More than 512 parked
// blocks: the oldest is waited for.)
namespace nvb {
static std::mutex g_streams_mtx;
static std::vector<hipStream_t> g_streams[64];
static void register_stream(int dev, hipStream_t s) { if (dev >= 0 && dev < 64) { std::lock_guard<std::mutex> lock(g_streams_mtx); g_streams[dev].push_back(s); } }
static void forget_stream(hipStream_t s)
{
    std::lock_guard<std::mutex> lock(g_streams_mtx);
    for (auto& v : g_streams) v.erase(std::remove(v.begin(), v.end(), s), v.end());
}
} // namespace nvb
NVB_API int nvbio_hip_device_malloc(void** ptr, uint64_t bytes)
{
    if (!ptr) return hipErrorInvalidValue;
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev)) return e;
    hipMemPool_t pool = nvb::private_pool(dev);
    if (!pool) return hipMalloc(ptr, bytes ? bytes : 1);
    if (hipError_t e = hipMallocFromPoolAsync(ptr, bytes ? bytes : 1, pool, nullptr)) return e;
    return hipStreamSynchronize(nullptr);          // like hipMalloc: the block is usable from every stream on return
}
NVB_API int nvbio_hip_device_free(void* ptr)
{
    if (!ptr) return hipSuccess;
    int dev = 0;
    static const bool sync_free = [] { const char* e = getenv("NVBIO_HIP_SYNC_FREE"); return e && e[0] == '1'; }();     // debugging aid: hipFree's blocking form
    const bool have_dev = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64;
    if (sync_free || !have_dev || !nvb::private_pool(dev)) {
        if (hipError_t e = hipDeviceSynchronize()) return e;
        return (have_dev && nvb::private_pool(dev)) ? hipFreeAsync(ptr, nullptr) : hipFree(ptr);
    }
    // Stream-ordered: the block goes back to the pool in the default stream's order, and the pool only ever serves default-stream requests
    // (nvbio_hip_device_malloc), so whoever gets it next is ordered behind everything the default stream held at this point -- the host does
    // not wait.  Work on the library's own non-blocking streams is not ordered with the default stream: the free is put behind one event per
    // such stream (a wait executed by the device).  A program that only uses the default stream -- the reference's applications -- pays one
    // hipFreeAsync.  (A form that parked blocks on the host and polled events did not survive the unchanged nvBowtie at 3 Gbp:
    // profiles/r05/device_free_forms.txt.)
    std::lock_guard<std::mutex> lock(nvb::g_streams_mtx);
    for (hipStream_t s : nvb::g_streams[dev])
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
    return hipFreeAsync(ptr, nullptr);
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

NVB_API int         nvbio_hip_abi_version(void) { return NVBIO_HIP_ABI_VERSION; }
NVB_API const char* nvbio_hip_arch(void)        { return "gfx950"; }
NVB_API const char* nvbio_hip_last_kernel(void) { return nvb::g_last_kernel; }
