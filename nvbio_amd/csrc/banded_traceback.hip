// banded_traceback.hip -- batched banded Gotoh traceback -> CIGAR, gfx950.
//
// Replaces, for GotohAligner<TYPE, scheme>:
//   aln::BatchedBandedAlignmentTraceback<BAND_LEN, CHECKPOINTS, stream>::enact  (nvbio/alignment/batched.h:460-476,
//   batched_banded_inl.h:250-420) = per job banded_alignment_traceback (banded_inl.h:352-489) with
//   priv::banded_alignment_traceback (gotoh/gotoh_banded_inl.h:878-960) and nvBowtie's Backtracker
//   (nvBowtie/bowtie2/cuda/alignment_utils.h:125-168) as the backtracer.
//
// The reference keeps one short2 checkpoint row every CHECKPOINTS rows and recomputes a window of
// 4-bit flow flags per checkpoint while walking back, because a Kepler thread had no memory for the
// whole band.  Here every lane streams the flags of its whole band to HBM once (one uint64 per row
// for BAND <= 16, lane-interleaved so a wavefront writes 512 contiguous bytes per row) and walks them
// back: 100 bp x band 15 is 800 B per alignment, i.e. 1 M alignments need 0.8 GB of the 288 GB.  The
// flags are the same ones the reference's submatrix pass regenerates whenever the DP values fit the
// int16 checkpoints; the host refuses (NotSupported) schemes/lengths that could leave that range.
#include "common.h"
#include <algorithm>

namespace nvb {

enum : uint32_t { SUBSTITUTION = 0, INSERTION = 1, DELETION = 2, SINK = 3, INSERTION_EXT = 4, DELETION_EXT = 8 };   // alignment_base.h:139-150

struct TracebackParams {
    StringSet       pat, txt;
    const uint8_t*  quals;          // nullable: quality byte of pattern symbol at stream index b + i
    uint64_t        n_quals;
    int32_t         match, gap_open, gap_ext, txt_gap_open, txt_gap_ext;
    int32_t         mismatch[256];  // by quality byte (constant for SimpleGotohScheme)
    uint32_t        n;
    int32_t*        out_score;
    uint2*          out_sink;
    uint2*          out_source;
    uint16_t*       out_cigar;
    uint32_t        cigar_stride;
    uint32_t*       out_cigar_len;
    uint64_t*       flags;          // [row][word][job]
    uint32_t        no_sink;        // SW / ED aligners: the banded submatrix context stores the direction only (sw_banded_inl.h:269-279)
};

template <uint32_t BAND>
struct FlagRow {
    static constexpr uint32_t W = (BAND + 15u) / 16u;
    uint64_t w[W];
    __device__ __forceinline__ void clear() { for (uint32_t k = 0; k < W; ++k) w[k] = 0; }
    __device__ __forceinline__ void set(const uint32_t j, const uint32_t f) { w[j >> 4] |= uint64_t(f) << ((j & 15u) * 4u); }
};

template <uint32_t BAND, int TYPE>
__global__ __launch_bounds__(256) void banded_gotoh_traceback_kernel(const TracebackParams p)
{
    constexpr bool     QUIRK = !(BAND == 3 || BAND == 5 || BAND == 7 || BAND == 15);   // Reference_cache<BAND>: 2-bit storage
    constexpr uint32_t W     = FlagRow<BAND>::W;

    __shared__ int32_t mm[256];
    mm[threadIdx.x] = p.mismatch[threadIdx.x];
    __syncthreads();

    const uint32_t tid = blockIdx.x * 256u + threadIdx.x;
    if (tid >= p.n) return;

    const uint64_t pb = p.pat.begin[tid], tb = p.txt.begin[tid];
    const uint32_t M  = p.pat.length ? p.pat.length[tid] : p.pat.fixed_length;
    const uint32_t N  = p.txt.length ? p.txt.length[tid] : p.txt.fixed_length;

    int32_t  best = -(1 << 30);
    uint32_t bx = 0xFFFFFFFFu, by = 0xFFFFFFFFu;
    auto report = [&](const int32_t s, const uint32_t x, const uint32_t y) { if (best <= s) { best = s; bx = x; by = y; } };

    if (N >= M)
    {
        // ---- forward pass: gotoh_banded_inl.h:434-640 with new_cell's flags kept ----
        const int32_t G_o = p.gap_open, G_e = p.gap_ext;
        const int32_t infimum = -32768 - max(max(G_o, G_e), max(p.txt_gap_open, p.txt_gap_ext));
        uint32_t tc[BAND - 1];
        int32_t  H[BAND], F[BAND];
#pragma unroll
        for (uint32_t j = 0; j < BAND - 1; ++j) {
            const uint32_t g = get_symbol(p.txt.s, tb + j);
            tc[j] = QUIRK ? (g & 3u) : g;
        }
        H[0] = 0;
#pragma unroll
        for (uint32_t j = 1; j < BAND; ++j) H[j] = (TYPE == NVBIO_HIP_GLOBAL) ? p.txt_gap_open + int32_t(j - 1) * p.txt_gap_ext : 0;
#pragma unroll
        for (uint32_t j = 0; j < BAND; ++j) F[j] = infimum;

        for (uint32_t i = 0; i < M; ++i)
        {
            const uint32_t q  = get_symbol(p.pat.s, pb + i);
            const uint32_t qq = p.quals ? p.quals[min(pb + i, p.n_quals - 1)] : 0u;
            const int32_t  S  = p.match, X = mm[qq];
            FlagRow<BAND> row; row.clear();
            uint32_t edir = SUBSTITUTION;
            {
                const int32_t ftop = F[1] + G_e, htop = H[1] + G_o;
                F[0] = max(ftop, htop);
                const uint32_t fdir = ftop > htop ? DELETION_EXT : SUBSTITUTION;
                const int32_t diagonal = H[0] + (tc[0] == q ? S : X);
                const int32_t top = F[0];
                int32_t  hi   = max(top, diagonal);
                uint32_t hdir = top > diagonal ? INSERTION : SUBSTITUTION;
                if (TYPE == NVBIO_HIP_LOCAL) { hi = max(hi, 0); if (hi == 0 && !p.no_sink) hdir = SINK; report(hi, i + 1, i + 1); }
                H[0] = hi;
                row.set(0, hdir | fdir);
            }
            int32_t E = H[0] + G_o;
#pragma unroll
            for (uint32_t j = 1; j < BAND - 1; ++j)
            {
                const int32_t ftop = F[j + 1] + G_e, htop = H[j + 1] + G_o;
                F[j] = max(ftop, htop);
                const uint32_t fdir = ftop > htop ? DELETION_EXT : SUBSTITUTION;
                const uint32_t g = tc[j]; tc[j - 1] = g;
                const int32_t diagonal = H[j] + (g == q ? S : X);
                const int32_t top = F[j], left = E;
                int32_t  hi   = max(max(top, left), diagonal);
                uint32_t hdir = top > left ? (top > diagonal ? INSERTION : SUBSTITUTION) : (left > diagonal ? DELETION : SUBSTITUTION);
                if (TYPE == NVBIO_HIP_LOCAL) { hi = max(hi, 0); if (hi == 0 && !p.no_sink) hdir = SINK; report(hi, i + j + 1, i + 1); }
                H[j] = hi;
                row.set(j, hdir | edir | fdir);
                const int32_t eleft = E + G_e, ediagonal = hi + G_o;
                edir = eleft > ediagonal ? INSERTION_EXT : SUBSTITUTION;
                E = max(ediagonal, eleft);
            }
            const uint32_t g = (i + BAND - 1 < N) ? get_symbol(p.txt.s, tb + i + BAND - 1) : 255u;
            tc[BAND - 2] = QUIRK ? (g & 3u) : g;
            {
                F[BAND - 1] = infimum;
                const int32_t diagonal = H[BAND - 1] + (g == q ? S : X);
                const int32_t left = E;
                int32_t  hi   = max(left, diagonal);
                uint32_t hdir = left > diagonal ? DELETION : SUBSTITUTION;
                if (TYPE == NVBIO_HIP_LOCAL) { hi = max(hi, 0); if (hi == 0 && !p.no_sink) hdir = SINK; report(hi, i + BAND, i + 1); }
                H[BAND - 1] = hi;
                row.set(BAND - 1, hdir | edir);
            }
#pragma unroll
            for (uint32_t k = 0; k < W; ++k)
                __builtin_nontemporal_store(row.w[k], p.flags + (uint64_t(i) * W + k) * p.n + tid);
        }
        if (TYPE == NVBIO_HIP_GLOBAL)
            report(H[BAND - 1], M + BAND - 1, M);
        else if (TYPE == NVBIO_HIP_SEMI_GLOBAL) {
            const uint32_t m = min(M + BAND - 1u, N) - (M - 1u);
            report(H[0], M, M);
#pragma unroll
            for (uint32_t j = 1; j < BAND; ++j) if (j < m) report(H[j], M + j, M);
        }
    }

    p.out_score[tid] = best;
    p.out_sink[tid]  = make_uint2(bx, by);

    // ---- walk back: banded_inl.h:383-423, gotoh_banded_inl.h:898-960, Backtracker::clip/push ----
    uint16_t* cigar = p.out_cigar + uint64_t(tid) * p.cigar_stride;
    uint32_t  size = 0, run_type = 255u, run_len = 0;
    auto flush = [&]() { if (run_len) { if (size < p.cigar_stride) cigar[size] = uint16_t(run_type | (run_len << 2)); ++size; run_len = 0; } };
    auto clip  = [&](const uint32_t l) { if (l) { if (size < p.cigar_stride) cigar[size] = uint16_t(3u | (l << 2)); ++size; } };
    auto push  = [&](const uint32_t t) { if (t != run_type) { flush(); run_type = t; } ++run_len; };

    if (bx == 0xFFFFFFFFu || by == 0xFFFFFFFFu) {
        p.out_source[tid]    = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
        p.out_cigar_len[tid] = 0;
        return;
    }
    clip(M - by);
    int32_t  entry = int32_t(bx - by), row = int32_t(by) - 1;
    uint32_t state = 0;     // 0 = H, 1 = E, 2 = F
    uint2    source = make_uint2(0, 0);
    bool     stopped = false;
    while (row >= 0)
    {
        const uint64_t w  = p.flags[(uint64_t(row) * W + (uint32_t(entry) >> 4)) * p.n + tid];
        const uint32_t op = uint32_t(w >> ((uint32_t(entry) & 15u) * 4u)) & 15u, h_op = op & 3u;
        if (TYPE == NVBIO_HIP_LOCAL && state == 0 && h_op == SINK) { source.y = uint32_t(row) + 1u; source.x = uint32_t(entry) + source.y; stopped = true; break; }
        if (state == 1)      { if ((op & INSERTION_EXT) == 0u) state = 0; --entry; push(DELETION); }
        else if (state == 2) { if ((op & DELETION_EXT)  == 0u) state = 0; ++entry; --row; push(INSERTION); }
        else if (h_op == DELETION)  state = 1;
        else if (h_op == INSERTION) state = 2;
        else { --row; push(SUBSTITUTION); }
    }
    if (!stopped) { source.y = 0; source.x = uint32_t(entry); }
    flush();
    clip(source.y);
    p.out_source[tid]    = source;
    p.out_cigar_len[tid] = size;
}

template <uint32_t BAND>
static hipError_t launch_tb(const TracebackParams& p, const int32_t type, hipStream_t s)
{
    const dim3 grid((p.n + 255u) / 256u), block(256);
    switch (type) {
    case NVBIO_HIP_LOCAL:       hipLaunchKernelGGL((banded_gotoh_traceback_kernel<BAND, NVBIO_HIP_LOCAL>),       grid, block, 0, s, p); break;
    case NVBIO_HIP_SEMI_GLOBAL: hipLaunchKernelGGL((banded_gotoh_traceback_kernel<BAND, NVBIO_HIP_SEMI_GLOBAL>), grid, block, 0, s, p); break;
    default:                    hipLaunchKernelGGL((banded_gotoh_traceback_kernel<BAND, NVBIO_HIP_GLOBAL>),      grid, block, 0, s, p); break;
    }
    return hipGetLastError();
}

static inline int64_t tb_abs(int32_t v) { return v < 0 ? -int64_t(v) : int64_t(v); }

static int traceback_common(TracebackParams& p, int64_t A, int32_t type, uint32_t band_len,
                            const nvbio_hip_string_set* patterns, const nvbio_hip_string_set* texts,
                            uint32_t max_pattern_len, uint32_t n,
                            int32_t* out_score, uint32_t* out_sink, uint32_t* out_source,
                            uint16_t* out_cigar, uint32_t cigar_stride, uint32_t* out_cigar_len,
                            void* temp, uint64_t temp_bytes, hipStream_t s)
{
    if (!patterns || !texts) return hipErrorInvalidValue;
    if (type < 0 || type > 2) return hipErrorInvalidValue;
    if (!(patterns->bits == 2 || patterns->bits == 4) || texts->bits != 2) return hipErrorNotSupported;
    if (!(band_len == 3 || band_len == 5 || band_len == 7 || band_len == 15 || band_len == 31)) return hipErrorNotSupported;
    if (n == 0) return hipSuccess;
    if (!out_score || !out_sink || !out_source || !out_cigar || !out_cigar_len || cigar_stride == 0) return hipErrorInvalidValue;
    if (!patterns->words || !texts->words || !patterns->begin || !texts->begin || patterns->n_words == 0 || texts->n_words == 0) return hipErrorInvalidValue;
    const uint32_t maxM = patterns->length ? max_pattern_len : patterns->fixed_length;
    if (maxM == 0 && patterns->length) return hipErrorInvalidValue;       // ragged sets must announce their longest pattern
    // the reference's int16 checkpoints (gotoh_banded_inl.h:205-262) are lossless only inside this range
    if ((int64_t(maxM) + band_len + 2) * A >= 32000) return hipErrorNotSupported;
    if (maxM >= (1u << 14)) return hipErrorNotSupported;                  // io::Cigar::m_len is 14 bits
    const uint64_t need = nvbio_hip_banded_gotoh_traceback_temp_bytes(band_len, maxM, n);
    if (need && (!temp || temp_bytes < need)) return hipErrorInvalidValue;

    p.pat = make_string_set(patterns);
    p.txt = make_string_set(texts);
    p.n = n; p.out_score = out_score; p.out_sink = reinterpret_cast<uint2*>(out_sink); p.out_source = reinterpret_cast<uint2*>(out_source);
    p.out_cigar = out_cigar; p.cigar_stride = cigar_stride; p.out_cigar_len = out_cigar_len;
    p.flags = static_cast<uint64_t*>(temp);
    g_last_kernel = "banded_gotoh_traceback_kernel";
    switch (band_len) {
    case 3:  return launch_tb<3>(p, type, s);
    case 5:  return launch_tb<5>(p, type, s);
    case 7:  return launch_tb<7>(p, type, s);
    case 15: return launch_tb<15>(p, type, s);
    default: return launch_tb<31>(p, type, s);
    }
}

} // namespace nvb

NVB_API uint64_t nvbio_hip_banded_gotoh_traceback_temp_bytes(uint32_t band_len, uint32_t max_pattern_len, uint32_t n)
{
    return uint64_t(max_pattern_len) * ((band_len + 15u) / 16u) * uint64_t(n) * 8u;
}

NVB_API int nvbio_hip_banded_gotoh_traceback(
    const nvbio_hip_gotoh_scheme* scheme, int32_t type, uint32_t band_len,
    const nvbio_hip_string_set* patterns, const nvbio_hip_string_set* texts,
    uint32_t max_pattern_len, uint32_t max_text_len, uint32_t n,
    int32_t* out_score, uint32_t* out_sink, uint32_t* out_source,
    uint16_t* out_cigar, uint32_t cigar_stride, uint32_t* out_cigar_len,
    void* temp, uint64_t temp_bytes, void* stream)
{
    (void)max_text_len;
    using namespace nvb;
    if (!scheme) return hipErrorInvalidValue;
    TracebackParams p;
    p.quals = nullptr; p.n_quals = 0; p.no_sink = 0;
    p.match = scheme->match;
    p.gap_open = scheme->gap_open; p.gap_ext = scheme->gap_ext;
    p.txt_gap_open = scheme->gap_open; p.txt_gap_ext = scheme->gap_ext;
    for (int i = 0; i < 256; ++i) p.mismatch[i] = scheme->mismatch;
    const int64_t A = std::max(std::max(tb_abs(scheme->match), tb_abs(scheme->mismatch)), std::max(tb_abs(scheme->gap_open), tb_abs(scheme->gap_ext)));
    return traceback_common(p, A, type, band_len, patterns, texts, max_pattern_len, n, out_score, out_sink, out_source,
                            out_cigar, cigar_stride, out_cigar_len, temp, temp_bytes, to_stream(stream));
}

NVB_API int nvbio_hip_banded_gotoh_traceback_qual(
    const nvbio_hip_gotoh_qual_scheme* scheme, int32_t type, uint32_t band_len,
    const nvbio_hip_string_set* patterns, const uint8_t* quals, uint64_t n_quals,
    const nvbio_hip_string_set* texts,
    uint32_t max_pattern_len, uint32_t max_text_len, uint32_t n,
    int32_t* out_score, uint32_t* out_sink, uint32_t* out_source,
    uint16_t* out_cigar, uint32_t cigar_stride, uint32_t* out_cigar_len,
    void* temp, uint64_t temp_bytes, void* stream)
{
    (void)max_text_len;
    using namespace nvb;
    if (!scheme) return hipErrorInvalidValue;
    if (n != 0 && (!quals || n_quals == 0)) return hipErrorInvalidValue;
    TracebackParams p;
    p.quals = quals; p.n_quals = n_quals; p.no_sink = 0;
    p.match = scheme->match;
    p.gap_open = scheme->pattern_gap_open; p.gap_ext = scheme->pattern_gap_ext;
    p.txt_gap_open = scheme->text_gap_open; p.txt_gap_ext = scheme->text_gap_ext;
    int64_t A = std::max(std::max(tb_abs(scheme->match), tb_abs(scheme->pattern_gap_open)), std::max(tb_abs(scheme->pattern_gap_ext),
                std::max(tb_abs(scheme->text_gap_open), tb_abs(scheme->text_gap_ext))));
    for (int i = 0; i < 256; ++i) { p.mismatch[i] = scheme->mismatch[i]; A = std::max(A, tb_abs(scheme->mismatch[i])); }
    return traceback_common(p, A, type, band_len, patterns, texts, max_pattern_len, n, out_score, out_sink, out_source,
                            out_cigar, cigar_stride, out_cigar_len, temp, temp_bytes, to_stream(stream));
}

// SmithWatermanAligner / EditDistanceAligner in the band (sw_banded_inl.h:405-470, 748-800): with deletion == insertion the
// directions are those of the Gotoh recurrence with gap_open == gap_ext; the reference's banded SW context does not mark
// zero cells as SINK, so a LOCAL walk runs to the first pattern row -- kept.
NVB_API int nvbio_hip_banded_sw_traceback(
    const nvbio_hip_sw_scheme* scheme, int32_t type, uint32_t band_len,
    const nvbio_hip_string_set* patterns, const nvbio_hip_string_set* texts,
    uint32_t max_pattern_len, uint32_t max_text_len, uint32_t n,
    int32_t* out_score, uint32_t* out_sink, uint32_t* out_source,
    uint16_t* out_cigar, uint32_t cigar_stride, uint32_t* out_cigar_len,
    void* temp, uint64_t temp_bytes, void* stream)
{
    (void)max_text_len;
    using namespace nvb;
    if (!scheme) return hipErrorInvalidValue;
    if (scheme->deletion != scheme->insertion) return hipErrorNotSupported;
    TracebackParams p;
    p.quals = nullptr; p.n_quals = 0; p.no_sink = 1;
    p.match = scheme->match;
    p.gap_open = p.gap_ext = p.txt_gap_open = p.txt_gap_ext = scheme->deletion;
    for (int i = 0; i < 256; ++i) p.mismatch[i] = scheme->mismatch;
    const int64_t A = std::max(std::max(tb_abs(scheme->match), tb_abs(scheme->mismatch)), tb_abs(scheme->deletion));
    return traceback_common(p, A, type, band_len, patterns, texts, max_pattern_len, n, out_score, out_sink, out_source,
                            out_cigar, cigar_stride, out_cigar_len, temp, temp_bytes, to_stream(stream));
}
