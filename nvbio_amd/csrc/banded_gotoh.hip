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
//  * LOCAL: instead of BestSink's compare + two selects per cell, each cell
//    packs (score << 5 | j) and a row keeps the max key (v_max3); the row's
//    best is folded into the running best once per row with a >= test, which
//    reproduces BestSink's "last maximum wins" tie-break (i-major, j-minor).
//
#include "common.h"
#include <limits.h>

namespace nvb {

thread_local const char* g_last_kernel = "";

struct GotohParams {
    StringSet pat, txt;
    int32_t   match, mismatch, gap_open, gap_ext;
    uint32_t  n;
    int32_t*  out_score;
    uint32_t* out_sink;
};

template <int BAND> struct BandTraits {
    // bands 3,5,7,15: the reference's text cache is a plain uint32 array; any other band uses a
    // 2-bit PackedStream cache which truncates what it stores to 2 bits
    // (nvbio/alignment/alignment_base_inl.h:75-98, packedstream_inl.h:352-369)
    static constexpr bool QUIRK = !(BAND == 3 || BAND == 5 || BAND == 7 || BAND == 15);
    static constexpr bool RING  = (BAND <= 16);
    static constexpr int  ROWS  = RING ? 16 : 8;
    static constexpr int  NTC   = RING ? 16 : BAND - 1;
};

template <int BAND, int TYPE>
struct DPState {
    int32_t  HG[BAND];                       // H + G_o of the previous row
    int32_t  F[BAND - 1];                    // F[BAND-1] is always `infimum`
    uint32_t tc[BandTraits<BAND>::NTC];      // text symbols of the band
    int32_t  bestkey;                        // LOCAL: (score << 5 | j) of the best cell so far
    uint32_t besti;                          //        and its row
};

struct DPConsts {
    int32_t Go, Ge, sM, sX, inf_ge;          // sM/sX = match/mismatch - G_o ; inf_ge = infimum + G_e
};

template <int BAND, int TYPE, int R>
__device__ __forceinline__ void dp_row(DPState<BAND, TYPE>& st, const DPConsts& k,
                                       const uint32_t i, const uint32_t q, const uint32_t g_new)
{
    typedef BandTraits<BAND> BT;
    int32_t rowkey = INT_MIN;
    int32_t E;

    // j == 0  (gotoh_banded_inl.h:483-517)
    {
        const int32_t fnext = (1 == BAND - 1) ? k.inf_ge : st.F[1 < BAND - 1 ? 1 : 0] + k.Ge;
        st.F[0] = max(fnext, st.HG[1]);
        const uint32_t g = st.tc[BT::RING ? (R & 15) : 0];
        const int32_t diag = st.HG[0] + (g == q ? k.sM : k.sX);
        int32_t hi = max(st.F[0], diag);
        if (TYPE == NVBIO_HIP_LOCAL) { hi = max(hi, 0); rowkey = (hi << 5); }
        st.HG[0] = hi + k.Go;
        E = st.HG[0];
    }
    // 1 <= j <= BAND-2  (:520-577)
    #pragma unroll
    for (int j = 1; j < BAND - 1; ++j)
    {
        const int32_t fnext = (j + 1 == BAND - 1) ? k.inf_ge : st.F[j + 1 < BAND - 1 ? j + 1 : 0] + k.Ge;
        st.F[j] = max(fnext, st.HG[j + 1]);
        const uint32_t g = st.tc[BT::RING ? ((R + j) & 15) : j];
        if (!BT::RING) st.tc[j - 1] = g;                                   // :542
        const int32_t diag = st.HG[j] + (g == q ? k.sM : k.sX);
        int32_t hi = max(max(st.F[j], E), diag);
        if (TYPE == NVBIO_HIP_LOCAL) { hi = max(hi, 0); rowkey = max(rowkey, (hi << 5) | j); }
        st.HG[j] = hi + k.Go;
        E = max(E + k.Ge, st.HG[j]);
    }
    // the new text symbol enters the band (:580-581); the cached copy is what later rows see
    {
        const uint32_t stored = BT::QUIRK ? (g_new & 3u) : g_new;
        if (BT::RING) st.tc[(R + BAND - 1) & 15] = stored;
        else          st.tc[BAND - 2] = stored;
    }
    // j == BAND-1  (:584-614) -- compares against the raw symbol
    {
        const int32_t diag = st.HG[BAND - 1] + (g_new == q ? k.sM : k.sX);
        int32_t hi = max(E, diag);
        if (TYPE == NVBIO_HIP_LOCAL) { hi = max(hi, 0); rowkey = max(rowkey, (hi << 5) | (BAND - 1)); }
        st.HG[BAND - 1] = hi + k.Go;
    }
    if (TYPE == NVBIO_HIP_LOCAL)
    {
        // BestSink::report uses '<=' (sink_inl.h:57-68): a later cell with an equal score wins
        const bool upd = (rowkey | 31) >= st.bestkey;
        st.bestkey = upd ? rowkey : st.bestkey;
        st.besti   = upd ? i : st.besti;
    }
}

template <int BAND, int TYPE, int R, int END>
struct RowUnrollN {
    __device__ __forceinline__ static void run(DPState<BAND, TYPE>& st, const DPConsts& k,
        const uint32_t i0, const uint32_t M, const uint32_t N, const uint64_t P, const uint32_t T)
    {
        const uint32_t i = i0 + R;
        if (i < M)
        {
            const uint32_t q = uint32_t(P >> (4 * R)) & 15u;
            uint32_t g = (T >> (2 * R)) & 3u;
            if (i + BAND - 1 >= N) g = 255u;
            dp_row<BAND, TYPE, R>(st, k, i, q, g);
        }
        RowUnrollN<BAND, TYPE, R + 1, END>::run(st, k, i0, M, N, P, T);
    }
};
template <int BAND, int TYPE, int END> struct RowUnrollN<BAND, TYPE, END, END> {
    __device__ __forceinline__ static void run(DPState<BAND, TYPE>&, const DPConsts&, uint32_t, uint32_t, uint32_t, uint64_t, uint32_t) {}
};

__device__ __forceinline__ uint64_t fetch_pattern16(const Stream& s, uint64_t sym)
{
    return (s.bits == 4) ? fetch16_4bit(s, sym) : expand_2to4(fetch16_2bit(s, sym));
}

template <int BAND, int TYPE>
__global__ void __launch_bounds__(256)
banded_gotoh_score_kernel(const GotohParams p)
{
    typedef BandTraits<BAND> BT;
    const uint32_t id = blockIdx.x * 256u + threadIdx.x;
    if (id >= p.n) return;

    const uint64_t pb = p.pat.begin[id];
    const uint64_t tb = p.txt.begin[id];
    const uint32_t M  = p.pat.length ? p.pat.length[id] : p.pat.fixed_length;
    const uint32_t N  = p.txt.length ? p.txt.length[id] : p.txt.fixed_length;

    int32_t  score = -(1 << 30);                 // BestSink<int32>() : numbers.h:832-835
    uint32_t sx = 0xFFFFFFFFu, sy = 0xFFFFFFFFu;

    if (N >= M)                                  // gotoh_banded_inl.h:431-432
    {
        DPConsts k;
        k.Go = p.gap_open; k.Ge = p.gap_ext;
        k.sM = p.match - p.gap_open; k.sX = p.mismatch - p.gap_open;
        const int32_t infimum = -32768 - max(p.gap_open, p.gap_ext);       // :446-448
        k.inf_ge = infimum + p.gap_ext;

        DPState<BAND, TYPE> st;
        // init_row_zero (:46-77), stored as H + G_o
        st.HG[0] = k.Go;
        #pragma unroll
        for (int j = 1; j < BAND; ++j)
            st.HG[j] = (TYPE == NVBIO_HIP_GLOBAL ? p.gap_open + (j - 1) * p.gap_ext : 0) + k.Go;
        #pragma unroll
        for (int j = 0; j < BAND - 1; ++j) st.F[j] = infimum;
        st.bestkey = INT_MIN; st.besti = 0;

        // first band of text (:441-442): symbols 0..BAND-2, no bounds check in the reference either
        {
            #pragma unroll
            for (int b = 0; b < BAND - 1; b += 16)
            {
                const uint32_t T0 = fetch16_2bit(p.txt.s, tb + b);
                #pragma unroll
                for (int j = b; j < BAND - 1 && j < b + 16; ++j)
                    st.tc[BT::RING ? (j & 15) : j] = (T0 >> (2 * (j - b))) & 3u;
            }
        }

        uint64_t P = fetch_pattern16(p.pat.s, pb);
        uint32_t T = fetch16_2bit(p.txt.s, tb + BAND - 1);
        for (uint32_t i0 = 0; i0 < M; i0 += BT::ROWS)
        {
            // prefetch the next block's symbols while this one computes
            const uint64_t Pn = fetch_pattern16(p.pat.s, pb + i0 + BT::ROWS);
            const uint32_t Tn = fetch16_2bit(p.txt.s, tb + i0 + BT::ROWS + BAND - 1);
            RowUnrollN<BAND, TYPE, 0, BT::ROWS>::run(st, k, i0, M, N, P, T);
            P = Pn; T = Tn;
        }

        if (TYPE == NVBIO_HIP_LOCAL)
        {
            if (M > 0) {
                const uint32_t j = uint32_t(st.bestkey) & 31u;
                score = st.bestkey >> 5;
                sx = st.besti + j + 1; sy = st.besti + 1;
            }
        }
        else if (TYPE == NVBIO_HIP_GLOBAL)
        {
            score = st.HG[BAND - 1] - k.Go;      // :641-642  (-(1<<30) <= any reachable score)
            sx = M + BAND - 1; sy = M;
        }
        else
        {
            // :643-655
            const uint32_t a = M + BAND - 1u;
            const uint32_t m = (a < N ? a : N) - (M - 1u);
            #pragma unroll
            for (int j = 0; j < BAND; ++j)
            {
                const int32_t h = st.HG[j] - k.Go;
                if ((j == 0 || uint32_t(j) < m) && score <= h) { score = h; sx = M + j; sy = M; }
            }
        }
    }
    p.out_score[id] = score;
    reinterpret_cast<uint2*>(p.out_sink)[id] = make_uint2(sx, sy);
}

template <int BAND>
static hipError_t launch_band(const GotohParams& p, int type, hipStream_t stream)
{
    const dim3 grid((p.n + 255u) / 256u), block(256);
    switch (type) {
    case NVBIO_HIP_GLOBAL:      hipLaunchKernelGGL((banded_gotoh_score_kernel<BAND, NVBIO_HIP_GLOBAL>),      grid, block, 0, stream, p); break;
    case NVBIO_HIP_LOCAL:       hipLaunchKernelGGL((banded_gotoh_score_kernel<BAND, NVBIO_HIP_LOCAL>),       grid, block, 0, stream, p); break;
    case NVBIO_HIP_SEMI_GLOBAL: hipLaunchKernelGGL((banded_gotoh_score_kernel<BAND, NVBIO_HIP_SEMI_GLOBAL>), grid, block, 0, stream, p); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

} // namespace nvb

NVB_API int nvbio_hip_banded_gotoh_score(
    const nvbio_hip_gotoh_scheme* scheme, int32_t type, uint32_t band_len,
    const nvbio_hip_string_set* patterns, const nvbio_hip_string_set* texts,
    uint32_t max_pattern_len, uint32_t max_text_len,
    uint32_t n, int32_t* out_score, uint32_t* out_sink, void* stream)
{
    (void)max_pattern_len; (void)max_text_len;
    using namespace nvb;
    if (!scheme || !patterns || !texts) return hipErrorInvalidValue;
    if (type < 0 || type > 2) return hipErrorInvalidValue;
    if (!(patterns->bits == 2 || patterns->bits == 4) || texts->bits != 2) return hipErrorNotSupported;
    if (n == 0) return hipSuccess;                 // an empty batch is legal and touches nothing
    if (!out_score || !out_sink) return hipErrorInvalidValue;
    if (!patterns->words || !texts->words || !patterns->begin || !texts->begin ||
        patterns->n_words == 0 || texts->n_words == 0) return hipErrorInvalidValue;

    GotohParams p;
    p.pat = make_string_set(patterns);
    p.txt = make_string_set(texts);
    p.match = scheme->match; p.mismatch = scheme->mismatch;
    p.gap_open = scheme->gap_open; p.gap_ext = scheme->gap_ext;
    p.n = n; p.out_score = out_score; p.out_sink = out_sink;

    hipStream_t s = to_stream(stream);
    switch (band_len) {
    case 3:  g_last_kernel = "banded_gotoh_score_kernel<3>";  return launch_band<3>(p, type, s);
    case 5:  g_last_kernel = "banded_gotoh_score_kernel<5>";  return launch_band<5>(p, type, s);
    case 7:  g_last_kernel = "banded_gotoh_score_kernel<7>";  return launch_band<7>(p, type, s);
    case 15: g_last_kernel = "banded_gotoh_score_kernel<15>"; return launch_band<15>(p, type, s);
    case 31: g_last_kernel = "banded_gotoh_score_kernel<31>"; return launch_band<31>(p, type, s);
    default: return hipErrorNotSupported;
    }
}

NVB_API int nvbio_hip_device_malloc(void** ptr, uint64_t bytes) { return ptr ? hipMalloc(ptr, bytes ? bytes : 1) : hipErrorInvalidValue; }
NVB_API int nvbio_hip_device_free(void* ptr) { return hipFree(ptr); }
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
