// banded_gotoh_wave.hip -- the banded Gotoh score with ONE WAVE PER JOB: the anti-diagonal sweep, for batches too small to fill the chip.
//
// The lane-per-job kernel (banded_gotoh_impl.h) is the throughput form: at 3+ waves per SIMD it runs at the VALU's issue rate.  A batch of a
// few thousand jobs cannot give it that: it occupies a handful of waves and takes as long as ONE job's 4 650 sequential cells -- 150-600 us
// for a 150 x 31 job, whatever the batch size.  nvBowtie's extension rounds after the first are such batches (a read's later seeds land on a
// placement already scored and are answered from the memos; what is left per round is a few thousand new placements), 35 rounds per batch of
// pairs: the driver was spending more time waiting for single jobs than computing (profiles/r06/README.md, "paired tail").
//
// Here a wave owns a job and lane j owns band cell j.  Cell (i, j) needs H and F of (i-1, j+1), H of (i-1, j) and the E chain of (i, j-1)
// (gotoh_banded_inl.h:483-614), so all cells with 2 i + j = w are independent: the wave steps w = 0 .. 2 (M - 1) + BAND - 1, lanes whose
// (w - j) is even and in range compute, the hand-offs are three DPP moves (lane j+1's H and F, lane j-1's E).  2 M + BAND steps instead of
// M * BAND cells in sequence: ~14x shorter at 150 x 31.  int32 arithmetic (the reference's), the reference's infimum, its text-cache quirk
// for bands other than 3 / 5 / 7 / 15 (a symbol past the text's end is 255 where it enters the band and 255 & 3 once cached), its sink order
// (the last maximal cell in row-major order; SEMI_GLOBAL: the last of the final row's maxima).  Results: bit for bit the lane kernel's
// (tests/test_banded_gpu.py::test_wave_kernel_equals_lane_kernel).
#include "banded_gotoh_impl.h"

namespace nvb {

struct WaveParams {
    StringSet      pat, txt;
    const uint8_t* quals; uint64_t n_quals;      // nullptr: constant mismatch (lut[0])
    int32_t        match, gap_open, gap_ext, txt_gap_open, txt_gap_ext;
    uint32_t       n; const uint32_t* n_dev; const uint32_t* job_index; const uint32_t* out_index; const uint32_t* gate; uint32_t gate_limit;
    int32_t*       out_score; uint2* out_sink;
    int32_t        lut[256];
};

constexpr uint32_t WAVE_MAX_M = 512u;            // rows a job may have (LDS: 4 bytes per row + 1 per text symbol, four jobs per block)

__device__ __forceinline__ int32_t lane_from_above(int32_t x) { return __builtin_amdgcn_update_dpp(0, x, 0x130, 0xf, 0xf, false); }   // wave_shl:1 -- lane j gets lane j+1's
__device__ __forceinline__ int32_t lane_from_below(int32_t x) { return __builtin_amdgcn_update_dpp(0, x, 0x138, 0xf, 0xf, false); }   // wave_shr:1 -- lane j gets lane j-1's

template <int BAND, int TYPE>
__global__ void __launch_bounds__(256)
banded_gotoh_wave_kernel(const WaveParams p)
{
    constexpr bool QUIRK = BandTraits<BAND>::QUIRK;
    __shared__ uint32_t s_row[4][WAVE_MAX_M];                    // per pattern row: symbol | (mismatch score + 32768) << 8
    __shared__ uint8_t  s_txt[4][WAVE_MAX_M + 64];               // text symbol at band-relative index t, as the reference's cache holds it
    __shared__ int32_t  s_best[4][64][2];
    const uint32_t wv = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t k = blockIdx.x * 4u + wv;
    const uint32_t n = p.n_dev ? *p.n_dev : p.n;
    if (k >= n) return;
    if (p.gate && *p.gate > p.gate_limit) return;                // (the lane-per-job kernel takes this batch)
    const uint32_t id = p.job_index ? p.job_index[k] : k;
    const uint32_t M = p.pat.length ? p.pat.length[id] : p.pat.fixed_length;
    const uint32_t N = p.txt.length ? p.txt.length[id] : p.txt.fixed_length;
    const uint64_t pb = p.pat.begin[id], tb = p.txt.begin[id];

    int32_t  score = -(1 << 30);
    uint32_t sx = 0xFFFFFFFFu, sy = 0xFFFFFFFFu;
    if (N >= M && M > 0u && M <= WAVE_MAX_M)
    {
        // ---- the job's rows and text into LDS
        for (uint32_t i = lane; i < M; i += 64u)
        {
            const uint32_t q = get_symbol(p.pat.s, pb + i);
            uint32_t qq = 0u;
            if (p.quals) { const uint64_t o = pb + i; qq = p.quals[o < p.n_quals ? o : p.n_quals - 1u]; }
            s_row[wv][i] = q | (uint32_t(p.lut[qq] + 32768) << 8);
        }
        const uint32_t n_txt = M + uint32_t(BAND) - 1u;          // text indices 0 .. M + BAND - 2 are ever looked at
        for (uint32_t t = lane; t < n_txt; t += 64u)
        {
            // the first BAND - 1 symbols are loaded without a bounds check (gotoh_banded_inl.h:441-442); later ones enter checked and are cached
            // (:580-581, :542): past the end the cache holds 255, or 255 & 3 where it is a 2-bit packed cache
            const bool real = t < uint32_t(BAND) - 1u || t < N;
            s_txt[wv][t] = real ? uint8_t(get_symbol(p.txt.s, tb + t)) : uint8_t(QUIRK ? 3u : 255u);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();

        const int32_t Go = p.gap_open, Ge = p.gap_ext;
        const int32_t inf = -32768 - max(max(Go, Ge), max(p.txt_gap_open, p.txt_gap_ext));
        const uint32_t j = lane;
        int32_t H = (j == 0u) ? 0 : (TYPE == NVBIO_HIP_GLOBAL ? p.txt_gap_open + (int32_t(j) - 1) * p.txt_gap_ext : 0);
        int32_t F = inf, E = inf;
        int32_t best = -1, best_i = 0;                                    // LOCAL: this column's last maximum
        const uint32_t steps = 2u * (M - 1u) + uint32_t(BAND);           // w = 0 .. 2 (M - 1) + BAND - 1
        for (uint32_t w = 0; w < steps; ++w)
        {
            const int32_t Hn = lane_from_above(H), Fn = lane_from_above(F), Ein = lane_from_below(E);
            const uint32_t d = w - j;                                     // = 2 i when this lane is on the wavefront
            const bool on = j < uint32_t(BAND) && w >= j && (d & 1u) == 0u && (d >> 1) < M;
            if (on)
            {
                const uint32_t i = d >> 1, t = i + j;
                const uint32_t ri = s_row[wv][i];
                const uint32_t q = ri & 255u;
                const int32_t sX = int32_t(ri >> 8) - 32768;
                uint32_t g = s_txt[wv][t];
                if (j == uint32_t(BAND) - 1u && t >= N) g = 255u;          // the entering symbol is compared raw
                const int32_t S = (g == q) ? p.match : sX;
                int32_t h;
                if (j == uint32_t(BAND) - 1u) { F = inf; h = max(Ein, H + S); }
                else
                {
                    F = max(Fn + Ge, Hn + Go);
                    h = (j == 0u) ? max(F, H + S) : max(max(F, Ein), H + S);
                }
                if (TYPE == NVBIO_HIP_LOCAL) { h = max(h, 0); if (h >= best) { best = h; best_i = int32_t(i); } }
                E = (j == 0u) ? h + Go : max(h + Go, Ein + Ge);
                H = h;
            }
        }
        // ---- the sink
        s_best[wv][lane][0] = (TYPE == NVBIO_HIP_LOCAL) ? best : H;
        s_best[wv][lane][1] = best_i;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (lane == 0u)
        {
            if (TYPE == NVBIO_HIP_LOCAL)
            {
                // BestSink keeps the last maximal report, reports come row by row, cell by cell (sink_inl.h:57-68)
                int32_t bs = -1, bi = 0; uint32_t bj = 0;
                for (uint32_t c = 0; c < uint32_t(BAND); ++c)
                {
                    const int32_t s = s_best[wv][c][0], ii = s_best[wv][c][1];
                    if (s > bs || (s == bs && (ii > bi || (ii == bi && c >= bj)))) { bs = s; bi = ii; bj = c; }
                }
                score = bs; sx = uint32_t(bi) + bj + 1u; sy = uint32_t(bi) + 1u;
            }
            else if (TYPE == NVBIO_HIP_GLOBAL) { score = s_best[wv][BAND - 1][0]; sx = M + uint32_t(BAND) - 1u; sy = M; }
            else
            {
                const uint32_t a = M + uint32_t(BAND) - 1u;
                const uint32_t m = (a < N ? a : N) - (M - 1u);
                for (uint32_t c = 0; c < uint32_t(BAND); ++c)
                {
                    const int32_t h = s_best[wv][c][0];
                    if ((c == 0u || c < m) && score <= h) { score = h; sx = M + c; sy = M; }
                }
            }
        }
    }
    else if (N >= M && M == 0u && TYPE != NVBIO_HIP_LOCAL)
    {
        if (TYPE == NVBIO_HIP_GLOBAL) { score = p.txt_gap_open + (BAND - 2) * p.txt_gap_ext; sx = BAND - 1; sy = 0; }
        else { score = 0; sx = N < uint32_t(BAND - 1) ? N : uint32_t(BAND - 1); sy = 0; }
    }
    if (lane == 0u) { const uint32_t o = p.out_index ? p.out_index[id] : id; p.out_score[o] = score; p.out_sink[o] = make_uint2(sx, sy); }
}

template <int BAND>
static hipError_t launch_wave(const WaveParams& p, int type, hipStream_t s)
{
    const dim3 grid((p.n + 3u) / 4u), block(256);
    switch (type) {
    case NVBIO_HIP_GLOBAL:      hipLaunchKernelGGL((banded_gotoh_wave_kernel<BAND, NVBIO_HIP_GLOBAL>),      grid, block, 0, s, p); break;
    case NVBIO_HIP_LOCAL:       hipLaunchKernelGGL((banded_gotoh_wave_kernel<BAND, NVBIO_HIP_LOCAL>),       grid, block, 0, s, p); break;
    case NVBIO_HIP_SEMI_GLOBAL: hipLaunchKernelGGL((banded_gotoh_wave_kernel<BAND, NVBIO_HIP_SEMI_GLOBAL>), grid, block, 0, s, p); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

} // namespace nvb

using namespace nvb;

NVB_API int nvbio_hip_banded_gotoh_score_qual_wave(
    const nvbio_hip_gotoh_qual_scheme* scheme, int32_t type, uint32_t band_len,
    const nvbio_hip_string_set* patterns, const uint8_t* quals, uint64_t n_quals,
    const nvbio_hip_string_set* texts, uint32_t max_pattern_len,
    uint32_t n, const uint32_t* n_on_device, const uint32_t* job_index, const uint32_t* out_index,
    const uint32_t* gate, uint32_t gate_limit,
    int32_t* out_score, uint32_t* out_sink, void* stream)
{
    if (!scheme || !patterns || !texts || type < 0 || type > 2) return hipErrorInvalidValue;
    if (!(patterns->bits == 2 || patterns->bits == 4) || texts->bits != 2) return hipErrorNotSupported;
    if (!(band_len == 3 || band_len == 5 || band_len == 7 || band_len == 15 || band_len == 31)) return hipErrorNotSupported;
    if (n == 0) return hipSuccess;
    if (!out_score || !out_sink || !patterns->words || !texts->words || !patterns->begin || !texts->begin || patterns->n_words == 0 || texts->n_words == 0) return hipErrorInvalidValue;
    if (quals && n_quals == 0) return hipErrorInvalidValue;
    const uint32_t maxM = patterns->length ? max_pattern_len : patterns->fixed_length;
    if (maxM == 0 || maxM > WAVE_MAX_M) return hipErrorNotSupported;         // (ragged sets must announce their longest pattern)
    WaveParams p;
    p.pat = make_string_set(patterns); p.txt = make_string_set(texts);
    p.quals = quals; p.n_quals = n_quals;
    p.match = scheme->match; p.gap_open = scheme->pattern_gap_open; p.gap_ext = scheme->pattern_gap_ext;
    p.txt_gap_open = scheme->text_gap_open; p.txt_gap_ext = scheme->text_gap_ext;
    p.n = n; p.n_dev = n_on_device; p.job_index = job_index; p.out_index = out_index; p.gate = gate; p.gate_limit = gate_limit;
    p.out_score = out_score; p.out_sink = reinterpret_cast<uint2*>(out_sink);
    for (int i = 0; i < 256; ++i) { p.lut[i] = scheme->mismatch[i]; if (p.lut[i] < -32768 || p.lut[i] > 32767) return hipErrorNotSupported; }
    g_last_kernel = "banded_gotoh_wave_kernel";
    hipStream_t s = to_stream(stream);
    switch (band_len) {
    case 3:  return launch_wave<3>(p, type, s);
    case 5:  return launch_wave<5>(p, type, s);
    case 7:  return launch_wave<7>(p, type, s);
    case 15: return launch_wave<15>(p, type, s);
    default: return launch_wave<31>(p, type, s);
    }
}
