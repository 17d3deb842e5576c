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
    const uint32_t* pending;        // nullable: the jobs this launch works on (slot -> job), *pending_count of them
    const uint32_t* pending_count;
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

    const uint32_t slot = blockIdx.x * 256u + threadIdx.x;       // where this lane's flags live
    if (slot >= (p.pending ? *p.pending_count : p.n)) return;
    const uint32_t tid = p.pending ? p.pending[slot] : slot;      // the job

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

        uint64_t pq = 0, tg = 0;          // read / new-text symbols of rows [i & ~15, +16), one nibble per row
        for (uint32_t i = 0; i < M; ++i)
        {
            if ((i & 15u) == 0u) {
                pq = (p.pat.s.bits == 2) ? expand_2to4(fetch16_2bit(p.pat.s, pb + i)) : fetch16_4bit(p.pat.s, pb + i);
                tg = expand_2to4(fetch16_2bit(p.txt.s, tb + i + BAND - 1));
            }
            const uint32_t q  = uint32_t(pq >> ((i & 15u) * 4u)) & 15u;
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
            const uint32_t g = (i + BAND - 1 < N) ? (uint32_t(tg >> ((i & 15u) * 4u)) & 15u) : 255u;
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
                // A PLAIN store.  This was a nontemporal store (the flags are written once and read once, much later): with it the C++ suite's
                // band-7 LOCAL batch came back with a different -- equally scored -- traceback for one job in about one run of six (round 4: 5 of
                // 28 runs of tests/cxx/nvbio_hip_test -aln; 0 of 24 with this store; the Python suite on torch-allocated scratch never showed it).
                // The walk below re-reads these words from the same lane; an `nt` store is not ordered against that later load the way a normal
                // store is on this path.  Bit-exactness is the contract, the stream hint was worth < 2 % of this kernel.
                p.flags[(uint64_t(i) * W + k) * p.n + slot] = row.w[k];
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
        const uint64_t w  = p.flags[(uint64_t(row) * W + (uint32_t(entry) >> 4)) * p.n + slot];
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

// ---- the ungapped fast path ------------------------------------------------------------------------------
// Most reads align without a gap.  Given the score and sink of a job (from the score kernel, the same DP), walk up the band
// column of the sink adding the substitution scores exactly as the DP sees them (qualities, N, and the reference's text cache:
// symbols beyond the text read 255 when first loaded and 3 when re-read from the 2-bit cache of bands other than 3,5,7,15).
// If the sum from some row t to the sink row equals the score, then every H on that diagonal segment equals its partial sum
// (H >= partial sum through the diagonal candidate, and H(sink) >= H(k) + rest forces <=), so at each of those cells the
// diagonal candidate equals H and the reference's direction is SUBSTITUTION (it wins every tie), i.e. the traceback IS that
// diagonal: GLOBAL / SEMI_GLOBAL need t = 0 (plus the row-zero value of the column), LOCAL takes the largest such t, where
// H(t-1) = 0 makes the walk stop (SINK) or t = 0.  Such jobs get their CIGAR here; the others are queued for the full kernel.
template <int TYPE>
__global__ __launch_bounds__(256) void banded_traceback_diagonal_kernel(const TracebackParams p, const uint32_t band, const uint32_t quirk,
                                                                        uint32_t* __restrict__ pending, uint32_t* __restrict__ pending_count)
{
    __shared__ int32_t mm[256];
    mm[threadIdx.x] = p.mismatch[threadIdx.x];
    __syncthreads();
    const uint32_t tid = blockIdx.x * 256u + threadIdx.x;
    if (tid >= p.n) return;
    const uint2 sink = p.out_sink[tid];
    if (sink.x == 0xFFFFFFFFu || sink.y == 0xFFFFFFFFu) {          // no alignment: what the full kernel reports
        p.out_source[tid] = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
        p.out_cigar_len[tid] = 0;
        return;
    }
    const int32_t  best = p.out_score[tid];
    const uint64_t pb = p.pat.begin[tid], tb = p.txt.begin[tid];
    const uint32_t M  = p.pat.length ? p.pat.length[tid] : p.pat.fixed_length;
    const uint32_t N  = p.txt.length ? p.txt.length[tid] : p.txt.fixed_length;
    const uint32_t by = sink.y, j = sink.x - sink.y;
    bool     found = false;
    uint32_t t = 0;
    if (!(TYPE == NVBIO_HIP_LOCAL && best <= 0))
    {
        // 16 rows per fetch (one 64-bit group of read symbols, one of text symbols); quality bytes only where symbols differ
        int32_t c = 0;
        // qv = the group's 16 quality bytes when they were fetched ahead (have_q), else they are read where symbols differ
        auto walk_group = [&](const int32_t c0, const uint64_t pq, const uint64_t tg, const uint4 qv, const bool have_q)
        {
            for (int32_t k = min(15, int32_t(by) - 1 - c0); k >= 0; --k)
            {
                const uint32_t i = uint32_t(c0 + k), idx = i + j;
                const uint32_t q = uint32_t(pq >> (4 * k)) & 15u;
                uint32_t g = uint32_t(tg >> (4 * k)) & 15u;
                if (!(idx + 2u <= band || idx < N)) g = (j == band - 1u) ? 255u : (quirk ? 3u : 255u);     // (the initial cache load is not range checked)
                int32_t sc = p.match;
                if (g != q)
                {
                    uint32_t qual = 0u;
                    if (have_q) { const uint32_t w = k < 8 ? (k < 4 ? qv.x : qv.y) : (k < 12 ? qv.z : qv.w); qual = (w >> (8u * (uint32_t(k) & 3u))) & 255u; }
                    else if (p.quals) qual = p.quals[min(pb + i, p.n_quals - 1)];
                    sc = mm[qual];
                }
                c += sc;
                if (TYPE == NVBIO_HIP_LOCAL && c == best) { found = true; t = i; break; }
            }
        };
        // The same walk by RUNS of matching rows: a group's differing rows are the set nibbles of pq ^ tg, a run of n matching rows adds
        // n * match, and the row at which a run reaches `best` (LOCAL) follows from one division -- ~10 instructions per differing row
        // instead of ~30 per row.  Only for a group whose rows all lie inside the text (the last window of a genome takes walk_group).
        auto walk_runs = [&](const int32_t c0, const uint64_t pq, const uint64_t tg, const uint4 qv, const bool have_q)
        {
            const int32_t kmax = min(15, int32_t(by) - 1 - c0);
            uint64_t x = pq ^ tg;
            x |= x >> 1; x |= x >> 2; x &= 0x1111111111111111ull;                       // row k differs -> bit 4k
            if (kmax < 15) x &= (1ull << (4 * (kmax + 1))) - 1ull;
            int32_t k = kmax;
            while (k >= 0)
            {
                const uint64_t below = (k == 15) ? x : (x & ((1ull << (4 * (k + 1))) - 1ull));
                const int32_t pos = below ? int32_t((63 - __clzll((long long)below)) >> 2) : -1;      // the next differing row at or below k
                const int32_t run = k - pos;                                            // rows k .. pos + 1 match
                if (run > 0)
                {
                    if (TYPE == NVBIO_HIP_LOCAL)
                    {
                        // the first row, going down, after which c == best: c + n * match == best with 1 <= n <= run
                        const int32_t need = best - c;
                        int32_t n = 0;
                        if (p.match == 0) n = (need == 0) ? 1 : 0;
                        else if (need % p.match == 0) { const int32_t qn = need / p.match; if (qn >= 1 && qn <= run) n = qn; }
                        if (n) { found = true; t = uint32_t(c0 + k - n + 1); return; }
                    }
                    c += run * p.match;
                }
                if (pos < 0) return;
                uint32_t qual = 0u;
                if (have_q) { const uint32_t w = pos < 8 ? (pos < 4 ? qv.x : qv.y) : (pos < 12 ? qv.z : qv.w); qual = (w >> (8u * (uint32_t(pos) & 3u))) & 255u; }
                else if (p.quals) qual = p.quals[min(pb + uint32_t(c0 + pos), p.n_quals - 1)];
                c += mm[qual];
                if (TYPE == NVBIO_HIP_LOCAL && c == best) { found = true; t = uint32_t(c0 + pos); return; }
                k = pos - 1;
            }
        };
        // a group's rows c0 .. c0 + 15 (those below `by`) all compare against real text symbols
        auto inside = [&](const int32_t c0) { return uint32_t(min(c0 + 15, int32_t(by) - 1)) + j < N; };
        auto load_p = [&](const int32_t c0) { return (p.pat.s.bits == 2) ? expand_2to4(fetch16_2bit(p.pat.s, pb + uint32_t(c0))) : fetch16_4bit(p.pat.s, pb + uint32_t(c0)); };
        auto load_t = [&](const int32_t c0) { return expand_2to4(fetch16_2bit(p.txt.s, tb + uint32_t(c0) + j)); };
        const uint4 no_q = make_uint4(0u, 0u, 0u, 0u);
        if (by <= 128u && p.pat.s.bits == 4u && !p.pat.s.lds && !p.txt.s.lds)
        {
            // Reads up to 128 rows (4-bit patterns in HBM): every word the walk will touch -- read symbols, window symbols, quality bytes -- is
            // requested BEFORE the walk starts, as independent loads with nothing but address arithmetic between them.  Fetching a group per
            // turn of the walk re-requests the read's and the window's lines once per group (between two turns of a lane the other 500 k
            // resident lanes have pushed them out of L2: 15 fabric requests per read, 55 G/s), and a quality byte fetched where symbols differ
            // puts a memory round trip into nearly every row of a wave (some lane of 64 differs in 93 % of the rows at 4 % error).
            typedef uint4 __attribute__((aligned(1))) uint4_u;
            const int32_t g_top = int32_t((by - 1u) >> 4);
            const bool have_q = p.quals != nullptr && pb + 16u * uint64_t(g_top) + 16u <= p.n_quals;
            const uint64_t pk = pb >> 3, tk = (tb + j) >> 4;
            const uint32_t psh = (uint32_t(pb) & 7u) << 2, tsh = (uint32_t(tb + j) & 15u) << 1;
            const uint64_t pn = p.pat.s.n_words - 1u, tn = p.txt.s.n_words - 1u;
            uint32_t pw[17], tw[9]; uint4 qv[8];
            #pragma unroll
            for (int k = 0; k < 17; ++k) { pw[k] = 0u; if (k <= 2 * g_top + 2) pw[k] = p.pat.s.words[min(pk + uint32_t(k), pn)]; }
            #pragma unroll
            for (int k = 0; k < 9; ++k)  { tw[k] = 0u; if (k <= g_top + 1) tw[k] = p.txt.s.words[min(tk + uint32_t(k), tn)]; }
            #pragma unroll
            for (int gi = 0; gi < 8; ++gi) { qv[gi] = no_q; if (have_q && gi <= g_top) qv[gi] = *reinterpret_cast<const uint4_u*>(p.quals + pb + 16u * uint32_t(gi)); }
            if (p.pat.s.big_endian) {
                #pragma unroll
                for (int k = 0; k < 17; ++k) pw[k] = rev4(pw[k]);
            }
            if (p.txt.s.big_endian) {
                #pragma unroll
                for (int k = 0; k < 9; ++k) tw[k] = rev2(tw[k]);
            }
            #pragma unroll
            for (int gi = 7; gi >= 0; --gi)
                if (gi <= g_top && !found)
                {
                    const uint64_t pq = (uint64_t(funnel(pw[2 * gi + 1], pw[2 * gi + 2], psh)) << 32) | funnel(pw[2 * gi], pw[2 * gi + 1], psh);
                    const uint64_t tgi = expand_2to4(funnel(tw[gi], tw[gi + 1], tsh));
                    if (inside(gi * 16)) walk_runs(gi * 16, pq, tgi, qv[gi], have_q);
                    else                 walk_group(gi * 16, pq, tgi, qv[gi], have_q);
                }
        }
        else
            for (int32_t c0 = int32_t((by - 1u) & ~15u); c0 >= 0 && !found; c0 -= 16)
            {
                if (inside(c0)) walk_runs(c0, load_p(c0), load_t(c0), no_q, false);
                else            walk_group(c0, load_p(c0), load_t(c0), no_q, false);
            }
        if (TYPE != NVBIO_HIP_LOCAL) {
            const int32_t init = (TYPE == NVBIO_HIP_GLOBAL && j != 0u) ? p.txt_gap_open + int32_t(j - 1u) * p.txt_gap_ext : 0;
            found = (c + init == best);
        }
    }
    {
        // queue the jobs that need the full kernel: one atomic per wavefront, slots in lane order
        const uint64_t need = __ballot(!found);
        if (!found) {
            const uint32_t lane = threadIdx.x & 63u, leader = uint32_t(__ffsll((long long)need)) - 1u;
            uint32_t base = 0;
            if (lane == leader) base = atomicAdd(pending_count, uint32_t(__popcll(need)));
            base = __shfl(base, int(leader));
            pending[base + uint32_t(__popcll(need & ((1ull << lane) - 1ull)))] = tid;
            return;
        }
    }

    uint16_t* cigar = p.out_cigar + uint64_t(tid) * p.cigar_stride;
    uint32_t  size = 0;
    auto emit = [&](const uint32_t type, const uint32_t len) { if (len) { if (size < p.cigar_stride) cigar[size] = uint16_t(type | (len << 2)); ++size; } };
    emit(3u, M - by);                      // soft clip at the read's end
    emit(SUBSTITUTION, by - t);
    emit(3u, t);                           // ... and at its start
    p.out_source[tid]    = make_uint2(j + t, t);
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

static inline uint64_t tb_flag_bytes(uint32_t band_len, uint32_t max_pattern_len, uint32_t n)
{
    const uint64_t b = uint64_t(max_pattern_len) * ((band_len + 15u) / 16u) * uint64_t(n) * 8u;
    return (b + 255ull) & ~255ull;
}
static inline int64_t tb_abs(int32_t v) { return v < 0 ? -int64_t(v) : int64_t(v); }

typedef int (*score_launch_fn)(const void* scheme, int32_t type, uint32_t band_len, const nvbio_hip_string_set* patterns, const uint8_t* quals, uint64_t n_quals,
                               const nvbio_hip_string_set* texts, uint32_t max_pattern_len, uint32_t n, int32_t* out_score, uint32_t* out_sink, void* stream);

static int traceback_common(TracebackParams& p, int64_t A, int32_t type, uint32_t band_len,
                            const nvbio_hip_string_set* patterns, const nvbio_hip_string_set* texts,
                            uint32_t max_pattern_len, uint32_t n,
                            int32_t* out_score, uint32_t* out_sink, uint32_t* out_source,
                            uint16_t* out_cigar, uint32_t cigar_stride, uint32_t* out_cigar_len,
                            void* temp, uint64_t temp_bytes, hipStream_t s, score_launch_fn score_launch = nullptr, const void* score_scheme = nullptr)
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
    p.pending = nullptr; p.pending_count = nullptr;
    if (score_launch)
    {
        // score + sink of every job from the score kernel, CIGARs of the ungapped ones from the diagonal check, the rest queued
        uint32_t* pending = reinterpret_cast<uint32_t*>(static_cast<uint8_t*>(temp) + tb_flag_bytes(band_len, maxM, n));
        uint32_t* pending_count = pending + n;
        if (hipError_t e = hipMemsetAsync(pending_count, 0, 4, s)) return e;
        if (int e = score_launch(score_scheme, type, band_len, patterns, p.quals, p.n_quals, texts, max_pattern_len, n, out_score, out_sink, s)) return e;
        const uint32_t quirk = !(band_len == 3 || band_len == 5 || band_len == 7 || band_len == 15) ? 1u : 0u;
        const dim3 grid((n + 255u) / 256u), block(256);
        switch (type) {
        case NVBIO_HIP_LOCAL:       hipLaunchKernelGGL(banded_traceback_diagonal_kernel<NVBIO_HIP_LOCAL>,       grid, block, 0, s, p, band_len, quirk, pending, pending_count); break;
        case NVBIO_HIP_SEMI_GLOBAL: hipLaunchKernelGGL(banded_traceback_diagonal_kernel<NVBIO_HIP_SEMI_GLOBAL>, grid, block, 0, s, p, band_len, quirk, pending, pending_count); break;
        default:                    hipLaunchKernelGGL(banded_traceback_diagonal_kernel<NVBIO_HIP_GLOBAL>,      grid, block, 0, s, p, band_len, quirk, pending, pending_count); break;
        }
        if (hipError_t e = hipGetLastError()) return e;
        p.pending = pending; p.pending_count = pending_count;
    }
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
    // the flow flags of every job's band + the queue of jobs left to the full kernel and its counter
    return nvb::tb_flag_bytes(band_len, max_pattern_len, n) + uint64_t(n) * 4u + 256u;
}

namespace {
int score_noqual(const void* scheme, int32_t type, uint32_t band_len, const nvbio_hip_string_set* patterns, const uint8_t*, uint64_t,
                 const nvbio_hip_string_set* texts, uint32_t max_pattern_len, uint32_t n, int32_t* out_score, uint32_t* out_sink, void* stream)
{ return nvbio_hip_banded_gotoh_score(static_cast<const nvbio_hip_gotoh_scheme*>(scheme), type, band_len, patterns, texts, max_pattern_len, 0, n, out_score, out_sink, stream); }
// the caller filled out_score / out_sink with what the scorer reports for every job (nvbio_hip_banded_gotoh_traceback_qual_known)
int score_prefilled(const void*, int32_t, uint32_t, const nvbio_hip_string_set*, const uint8_t*, uint64_t, const nvbio_hip_string_set*, uint32_t, uint32_t, int32_t*, uint32_t*, void*)
{ return hipSuccess; }
int score_qual(const void* scheme, int32_t type, uint32_t band_len, const nvbio_hip_string_set* patterns, const uint8_t* quals, uint64_t n_quals,
               const nvbio_hip_string_set* texts, uint32_t max_pattern_len, uint32_t n, int32_t* out_score, uint32_t* out_sink, void* stream)
{ return nvbio_hip_banded_gotoh_score_qual(static_cast<const nvbio_hip_gotoh_qual_scheme*>(scheme), type, band_len, patterns, quals, n_quals, texts, max_pattern_len, 0, n, out_score, out_sink, stream); }
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
                            out_cigar, cigar_stride, out_cigar_len, temp, temp_bytes, to_stream(stream), score_noqual, scheme);
}

static int banded_traceback_qual(
    const nvbio_hip_gotoh_qual_scheme* scheme, int32_t type, uint32_t band_len,
    const nvbio_hip_string_set* patterns, const uint8_t* quals, uint64_t n_quals,
    const nvbio_hip_string_set* texts,
    uint32_t max_pattern_len, uint32_t max_text_len, uint32_t n,
    int32_t* out_score, uint32_t* out_sink, uint32_t* out_source,
    uint16_t* out_cigar, uint32_t cigar_stride, uint32_t* out_cigar_len,
    void* temp, uint64_t temp_bytes, void* stream, bool known)
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
                            out_cigar, cigar_stride, out_cigar_len, temp, temp_bytes, to_stream(stream), known ? score_prefilled : score_qual, scheme);
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

NVB_API int nvbio_hip_banded_gotoh_traceback_qual(
    const nvbio_hip_gotoh_qual_scheme* scheme, int32_t type, uint32_t band_len,
    const nvbio_hip_string_set* patterns, const uint8_t* quals, uint64_t n_quals, const nvbio_hip_string_set* texts,
    uint32_t max_pattern_len, uint32_t max_text_len, uint32_t n, int32_t* out_score, uint32_t* out_sink, uint32_t* out_source,
    uint16_t* out_cigar, uint32_t cigar_stride, uint32_t* out_cigar_len, void* temp, uint64_t temp_bytes, void* stream)
{
    return banded_traceback_qual(scheme, type, band_len, patterns, quals, n_quals, texts, max_pattern_len, max_text_len, n, out_score, out_sink, out_source,
                                 out_cigar, cigar_stride, out_cigar_len, temp, temp_bytes, stream, false);
}

// The same with score and sink of every job known beforehand (out_score / out_sink filled by the caller with what
// nvbio_hip_banded_gotoh_score_qual reports for these very jobs: the extension stage already ran that DP): skips the score pass.
NVB_API int nvbio_hip_banded_gotoh_traceback_qual_known(
    const nvbio_hip_gotoh_qual_scheme* scheme, int32_t type, uint32_t band_len,
    const nvbio_hip_string_set* patterns, const uint8_t* quals, uint64_t n_quals, const nvbio_hip_string_set* texts,
    uint32_t max_pattern_len, uint32_t max_text_len, uint32_t n, int32_t* score, uint32_t* sink, uint32_t* out_source,
    uint16_t* out_cigar, uint32_t cigar_stride, uint32_t* out_cigar_len, void* temp, uint64_t temp_bytes, void* stream)
{
    return banded_traceback_qual(scheme, type, band_len, patterns, quals, n_quals, texts, max_pattern_len, max_text_len, n, score, sink, out_source,
                                 out_cigar, cigar_stride, out_cigar_len, temp, temp_bytes, stream, true);
}
