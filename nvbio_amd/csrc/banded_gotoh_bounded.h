// banded_gotoh_bounded.h -- the banded Gotoh score with a per-job score bound: a job whose score cannot exceed its bound stops early and
// its lane takes the next job.
//
// What the reference has: banded_alignment_score(aligner, pattern, quals, text, min_score, sink) takes a `min_score` and, in its windowed
// form, gives up on a job once max(H) < min_score + remaining_rows * match (gotoh_banded_inl.h:622-634).  nvBowtie's scoring stage passes
// the read's second-best score as that threshold (BestScoreStream::init_context, score_best_inl.h:113-116) and its reduction
// (reduce_inl.h:111-135) only ever asks of a hit's score "is it above the best / the second best", both of which are >= the threshold: a
// score at or below it leaves the read's record exactly as any other such score would.
//
// What this kernel does with that: it is the lane-per-job kernel of banded_gotoh_impl.h (same cells, same 16-bit / 32-bit arithmetic, same
// tables and staging), run by PERSISTENT waves over a work counter.  After every block of rows a lane forms the largest score its job can
// still reach -- the row's best H plus `cap` per remaining row (cap = the largest substitution score, or 0; LOCAL: or the best cell so far)
// -- and a lane whose job cannot exceed its threshold retires it (the reported score is that upper bound, <= the threshold; no sink).  Retired and
// finished lanes idle until `refill` of the wave's 64 are free, then take new jobs together (one atomic per wave), so that the divergent set-up
// code is paid once per group.  On nvBowtie's repeat-rich extension rounds 95 % of the jobs end at or below their threshold, at rows spread
// evenly over the read: lane-rows drop to ~0.57 of the plain kernel's (profiles/r06/own_driver_3gbp_death.json).
//
// MEASURED (round 6, profiles/r06/bounded_probe.txt): correct, and NOT faster.  The band's ~90 state registers now live across the refill code; at
// band 31 the kernel needs > 256 VGPRs where the plain one fits 165 (occupancy 3 -> 2, 28 bytes of scratch), a block of rows costs 1.4x the plain
// kernel's, and with 92 % of the jobs given up at evenly spread rows the launch still takes 1.1x the plain kernel's time (at occupancy 3 the
// spills make it 3.7x).  The driver therefore keeps the plain kernel (NVBIO_HIP_BOUNDED_DP=1 selects this one); the entry point stays because
// its contract -- thresholds, a device-side job count, results written through an index -- is the reference's min_score interface.
//
// Results: jobs ending above their threshold -- score and sink bit for bit what the plain kernel reports; the others -- some score <= threshold,
// sink (0xFFFFFFFF, 0xFFFFFFFF).  min_score[i] = INT32_MIN makes job i exact.
#pragma once
#include "banded_gotoh_impl.h"

namespace nvb {

struct BoundArgs {
    const int32_t*  min_score;     // per job: scores at or below it need not be exact
    const uint32_t* n_dev;         // the number of jobs, on the device (NULL: GotohParams::n) -- spares the host a round trip
    uint32_t*       counter;       // work counter, zero when the kernel starts
    int32_t         cap;           // max(largest substitution score, 0): the most one more row can add to a path
    uint32_t        refill;        // idle lanes of a wave that trigger a refill (1..64)
    const uint32_t* out_index;     // optional: job i's results go to out_score[out_index[i]] / out_sink[out_index[i]] (a compacted batch writing back at its hits)
};

template <int BAND, int TYPE, typename A, typename QA>
#ifndef NVB_EXP_OCC
#define NVB_EXP_OCC 2
#endif
__global__ void __launch_bounds__(256, (BAND == 31 ? NVB_EXP_OCC : 1))
banded_gotoh_score_bounded_kernel(const GotohParams p, const QA qa, const BoundArgs ba)
{
    typedef BandTraits<BAND> BT;
    typedef typename A::T T;
    constexpr bool QUAL = IsQual<QA>::value;
    constexpr int SH = (TYPE == NVBIO_HIP_LOCAL) ? 5 : 0;
    constexpr uint32_t NONE = 0xFFFFFFFFu;
    __shared__ T s_lut[QUAL ? 256 : 1];
    __shared__ uint2 s_masks[16];
    if (A::TABLE && threadIdx.x < 16u)
        s_masks[threadIdx.x] = make_uint2(threadIdx.x == 0u ? 0x0000FFFFu : threadIdx.x == 1u ? 0xFFFF0000u : 0u,
                                          threadIdx.x == 2u ? 0x0000FFFFu : threadIdx.x == 3u ? 0xFFFF0000u : 0u);
    if (A::TABLE && !QUAL) __syncthreads();
    extern __shared__ __attribute__((aligned(16))) uint32_t s_stage[];
    fill_lut<A>(s_lut, qa, p.gap_open, SH);

    const uint32_t n = ba.n_dev ? *ba.n_dev : p.n;
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t below = (1ull << lane) - 1ull;

    DPConsts<A> k;
    k.Go = A::cnst(p.gap_open * (1 << SH)); k.Ge = A::cnst(p.gap_ext * (1 << SH)); k.GeF = k.dF = A::cnst(0); k.bytes = false;
    k.sM = A::cnst((p.match - p.gap_open) * (1 << SH)); k.sX = A::cnst((p.mismatch - p.gap_open) * (1 << SH));
    k.inf = Sentinel<A>::get(p.gap_open, p.gap_ext, p.txt_gap_open, p.txt_gap_ext, SH);
    k.sMM = (uint32_t(k.sM) & 0xFFFFu) * 0x10001u; k.sXX = (uint32_t(k.sX) & 0xFFFFu) * 0x10001u;
    const T infimum = k.inf;

    // the lane's job
    uint32_t id = NONE, M = 0, N = 0, fl = 0, i0 = 0;
    int32_t  thr = INT32_MIN;
    uint64_t pb = 0, tb = 0;
    Stream   ps = p.pat.s, ts = p.txt.s;
    DPState<BAND, A> st;
    uint64_t P = 0; uint4 Q = make_uint4(0, 0, 0, 0); uint32_t Tx = 0;
    bool drained = false;

    for (;;)
    {
        // ---- refill: the idle lanes of the wave take the next jobs together
        const uint64_t want = __ballot(id == NONE && !drained);
        const uint64_t busy = __ballot(id != NONE);
        if (want != 0ull && (busy == 0ull || uint32_t(__popcll(want)) >= ba.refill))
        {
            const int first = __ffsll((long long)want) - 1;
            uint32_t base = 0;
            if (int(lane) == first) base = atomicAdd(ba.counter, uint32_t(__popcll(want)));
            base = __shfl(base, first, 64);
            bool fresh = false;                       // this lane starts a job now
            if (id == NONE && !drained)
            {
                const uint32_t nid = base + uint32_t(__popcll(want & below));
                if (nid >= n) drained = true;
                else
                {
                    M  = p.pat.length ? p.pat.length[nid] : p.pat.fixed_length;
                    N  = p.txt.length ? p.txt.length[nid] : p.txt.fixed_length;
                    pb = p.pat.begin[nid];
                    tb = p.txt.begin[nid];
                    fl = view_flags(qa, nid);
                    thr = ba.min_score ? ba.min_score[nid] : INT32_MIN;
                    if (M < p.len_lo || M > p.len_hi) { /* the other arithmetic width owns this job */ }
                    else if (N < M || M == 0u)
                    {
                        // no alignment (gotoh_banded_inl.h:431-432) / an empty pattern: what the plain kernel reports
                        int32_t score = -(1 << 30); uint32_t sx = 0xFFFFFFFFu, sy = 0xFFFFFFFFu;
                        if (N >= M && TYPE != NVBIO_HIP_LOCAL)
                        {
                            if (TYPE == NVBIO_HIP_GLOBAL) { score = p.txt_gap_open + (BAND - 2) * p.txt_gap_ext; sx = BAND - 1; sy = 0; }
                            else { score = 0; sx = N < uint32_t(BAND - 1) ? N : uint32_t(BAND - 1); sy = 0; }      // SEMI_GLOBAL, M == 0: row zero's cells 0 .. min(BAND-1, N), all 0, the last one wins
                        }
                        const uint32_t o = ba.out_index ? ba.out_index[nid] : nid;
                        p.out_score[o] = score;
                        reinterpret_cast<uint2*>(p.out_sink)[o] = make_uint2(sx, sy);
                    }
                    else { id = nid; i0 = 0; fresh = true; }
                }
            }
            // The fresh lanes' state.  Everything below is either under a small divergent `if` that only moves words (staging) or a SELECT
            // between the value a fresh lane starts from and the value a running lane holds: the band's ~90 registers are never assigned in
            // divergent control flow, where the compiler would keep a second copy of them alive across the join.
            if (__ballot(fresh) != 0ull)
            {
                uint32_t lds_p = 0u, lds_t = 0u;      // 1: this fresh lane's words are staged
                uint64_t kbp = 0, kbt = 0;
                if (fresh && p.stage_pw != 0u)
                {
                    uint32_t pwords;
                    pattern_words_span(pb, M, p.pat.s.bits, fl, kbp, pwords);
                    kbt = tb >> 4;
                    if (pwords <= p.stage_pw && stage_words_text(uint32_t(tb & 15u), M, BAND) <= p.stage_tw)
                    {
                        uint32_t* lp = s_stage + threadIdx.x;
                        uint32_t* lt = s_stage + p.stage_pw * 256u + threadIdx.x;
                        for (uint32_t w = 0; w < p.stage_pw; w += 4u) {
                            const uint32_t a = ld_word_global(p.pat.s, kbp + w), b = ld_word_global(p.pat.s, kbp + w + 1u),
                                           c = ld_word_global(p.pat.s, kbp + w + 2u), d = ld_word_global(p.pat.s, kbp + w + 3u);
                            lp[w * 256u] = a; lp[(w + 1u) * 256u] = b; lp[(w + 2u) * 256u] = c; lp[(w + 3u) * 256u] = d;
                        }
                        for (uint32_t w = 0; w < p.stage_tw; w += 4u) {
                            const uint32_t a = ld_word_global(p.txt.s, kbt + w), b = ld_word_global(p.txt.s, kbt + w + 1u),
                                           c = ld_word_global(p.txt.s, kbt + w + 2u), d = ld_word_global(p.txt.s, kbt + w + 3u);
                            lt[w * 256u] = a; lt[(w + 1u) * 256u] = b; lt[(w + 2u) * 256u] = c; lt[(w + 3u) * 256u] = d;
                        }
                        lds_p = lds_t = 1u;
                    }
                }
                if (fresh)
                {
                    ps.lds = lds_p ? (lds_words_t)(s_stage + threadIdx.x) : (lds_words_t)nullptr;                         ps.kb = lds_p ? kbp : 0ull;
                    ts.lds = lds_t ? (lds_words_t)(s_stage + p.stage_pw * 256u + threadIdx.x) : (lds_words_t)nullptr;    ts.kb = lds_t ? kbt : 0ull;
                }
                st.HG[0] = fresh ? k.Go : st.HG[0];
                #pragma unroll
                for (int j = 1; j < BAND; ++j)
                {
                    const T h0 = A::cnst(((TYPE == NVBIO_HIP_GLOBAL ? p.txt_gap_open + (j - 1) * p.txt_gap_ext : 0) + p.gap_open) * (1 << SH));
                    st.HG[j] = fresh ? h0 : st.HG[j];
                }
                #pragma unroll
                for (int j = 0; j < BAND - 1; ++j) st.F[j] = fresh ? infimum : st.F[j];
                st.bestkey = fresh ? A::cnst(0) : st.bestkey; st.besti = fresh ? 0u : st.besti;
                // first band of text and the first groups: fetched by every lane from its own strings (a running lane's are discarded)
                #pragma unroll
                for (int b = 0; b < BAND - 1; b += 16)
                {
                    const uint32_t T0 = fetch16_2bit(ts, tb + b);
                    #pragma unroll
                    for (int j = b; j < BAND - 1 && j < b + 16; ++j)
                    {
                        const uint32_t e = A::enc((T0 >> (2 * (j - b))) & 3u);
                        st.tc[j & BT::MASK] = fresh ? e : st.tc[j & BT::MASK];
                    }
                }
                uint64_t P0; uint4 Q0;
                fetch_group(ps, qa, pb, M, fl, 0u, P0, Q0);
                const uint32_t T1 = fetch16_2bit(ts, tb + BAND - 1);
                P = fresh ? P0 : P; Tx = fresh ? T1 : Tx;
                Q.x = fresh ? Q0.x : Q.x; Q.y = fresh ? Q0.y : Q.y; Q.z = fresh ? Q0.z : Q.z; Q.w = fresh ? Q0.w : Q.w;
            }
        }
        if (__ballot(id != NONE) == 0ull) { if (__ballot(!drained) == 0ull) break; else continue; }

        // ---- one block of rows for the lanes that hold a job
        if (id != NONE)
        {
            uint64_t Pn; uint4 Qn;
            fetch_group(ps, qa, pb, M, fl, i0 + BT::ROWS, Pn, Qn);
            const uint32_t Tn = fetch16_2bit(ts, tb + i0 + BT::ROWS + BAND - 1);
            const uint32_t last_row = (i0 + BT::ROWS < M ? i0 + BT::ROWS : M) - 1u;
#ifdef NVB_EXP_NOSLOW
                RowUnrollN<BAND, TYPE, A, QUAL, true, 0, BT::ROWS>::run(st, k, i0, M, N, P, Tx, Q, s_lut, s_masks);
#else
            if (A::TABLE && last_row + BAND - 1u < N)
                RowUnrollN<BAND, TYPE, A, QUAL, true, 0, BT::ROWS>::run(st, k, i0, M, N, P, Tx, Q, s_lut, s_masks);
            else
                RowUnrollN<BAND, TYPE, A, QUAL, false, 0, BT::ROWS>::run(st, k, i0, M, N, P, Tx, Q, s_lut, s_masks);
#endif
            ring_advance<BAND, A>(st);
            P = Pn; Tx = Tn; Q = Qn;
            i0 += BT::ROWS;

            if (i0 >= M)
            {
                int32_t score = -(1 << 30); uint32_t sx = 0xFFFFFFFFu, sy = 0xFFFFFFFFu;
                if (TYPE == NVBIO_HIP_LOCAL)
                {
                    const uint32_t key = uint32_t(A::to_int(st.bestkey));
                    const uint32_t j = key & 31u;
                    score = int32_t(key >> 5);
                    sx = st.besti + j + 1; sy = st.besti + 1;
                }
                else if (TYPE == NVBIO_HIP_GLOBAL)
                {
                    score = A::to_int(st.HG[BAND - 1]) - p.gap_open;
                    sx = M + BAND - 1; sy = M;
                }
                else
                {
                    const uint32_t a = M + BAND - 1u;
                    const uint32_t m = (a < N ? a : N) - (M - 1u);
                    #pragma unroll
                    for (int j = 0; j < BAND; ++j)
                    {
                        const int32_t h = A::to_int(st.HG[j]) - p.gap_open;
                        if ((j == 0 || uint32_t(j) < m) && score <= h) { score = h; sx = M + j; sy = M; }
                    }
                }
                const uint32_t o = ba.out_index ? ba.out_index[id] : id;
                p.out_score[o] = score;
                reinterpret_cast<uint2*>(p.out_sink)[o] = make_uint2(sx, sy);
                id = NONE;
            }
            else if (thr != INT32_MIN)
            {
                // the most this job can still score: the best H of the last row done, plus `cap` for every row to come
                T rm = st.HG[0];
                #pragma unroll
                for (int j = 1; j < BAND; ++j) rm = A::mx(rm, st.HG[j]);
                int32_t bound = (A::to_int(rm) >> SH) - p.gap_open + ba.cap * int32_t(M - i0);
                if (TYPE == NVBIO_HIP_LOCAL) { const int32_t bs = int32_t(uint32_t(A::to_int(st.bestkey)) >> 5); bound = bound > bs ? bound : bs; }
                if (bound <= thr)
                {
                    const uint32_t o = ba.out_index ? ba.out_index[id] : id;
                    p.out_score[o] = bound;
                    reinterpret_cast<uint2*>(p.out_sink)[o] = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
                    id = NONE;
                }
            }
        }
    }
}

// persistent grid: as many blocks as the device holds at once (occupancy x compute units), or fewer when there are fewer jobs
template <typename K>
inline uint32_t persistent_blocks(K kernel, unsigned lds_bytes, uint32_t n_jobs)
{
    int dev = 0, cus = 256, per_cu = 2;
    (void)hipGetDevice(&dev);
    static int cu_count[64] = {};
    if (dev >= 0 && dev < 64) {
        if (cu_count[dev] == 0) { int v = 0; if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cu_count[dev] = v; else cu_count[dev] = 256; }
        cus = cu_count[dev];
    }
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, lds_bytes) != hipSuccess || per_cu < 1) { (void)hipGetLastError(); per_cu = 2; }
    const uint64_t want = (uint64_t(n_jobs) + 255u) / 256u, cap = uint64_t(cus) * uint64_t(per_cu);
    return uint32_t(want < cap ? (want ? want : 1u) : cap);
}

template <int BAND, typename A, typename QA>
hipError_t launch_band_bounded(const GotohParams& p, const QA& qa, const BoundArgs& ba, int type, hipStream_t stream)
{
    const dim3 block(256);
    const unsigned lds_pad = (p.stage_pw + p.stage_tw) * 256u * 4u;
    #define NVB_BOUNDED(TYPE) { auto kern = banded_gotoh_score_bounded_kernel<BAND, TYPE, A, QA>; \
                                hipLaunchKernelGGL(kern, dim3(persistent_blocks(kern, lds_pad, p.n)), block, lds_pad, stream, p, qa, ba); }
    switch (type) {
    case NVBIO_HIP_GLOBAL:      NVB_BOUNDED(NVBIO_HIP_GLOBAL) break;
    case NVBIO_HIP_LOCAL:       NVB_BOUNDED(NVBIO_HIP_LOCAL) break;
    case NVBIO_HIP_SEMI_GLOBAL: NVB_BOUNDED(NVBIO_HIP_SEMI_GLOBAL) break;
    default: return hipErrorInvalidValue;
    }
    #undef NVB_BOUNDED
    return hipGetLastError();
}
template <int BAND, typename QA>
hipError_t launch_band_width_bounded(const GotohParams& p, const QA& qa, const BoundArgs& ba, int type, bool width16, hipStream_t s)
{
    return width16 ? launch_band_bounded<BAND, A16P, QA>(p, qa, ba, type, s) : launch_band_bounded<BAND, A32P, QA>(p, qa, ba, type, s);
}

} // namespace nvb
