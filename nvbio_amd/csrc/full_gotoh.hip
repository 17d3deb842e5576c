// full_gotoh.hip -- batched full-matrix Gotoh / Smith-Waterman score for gfx950.
//
// Computes, per job, what the reference's text-blocking full DP
//   priv::gotoh_alignment_score_dispatch<8,TYPE,TextBlockingTag,symbol>::run
//   (nvbio/alignment/gotoh/gotoh_inl.h:969-1489; BAND_LEN = 8 text columns per block, :1491-1495)
// reports into a fresh BestSink<int32>, as instantiated by sw-benchmark
// (make_gotoh_aligner<TYPE,TextBlockingTag>, sw-benchmark/sw-benchmark.cu:604-631) through
//   BatchedAlignmentScore<stream,DeviceThreadScheduler>::enact (nvbio/alignment/batched_inl.h:383-460).
//
// The reference runs one thread per alignment and keeps a boundary column of short2 {H,E} per
// pattern row in local / global memory, sweeping the text in blocks of 8 columns.  That mapping
// needs O(pattern) private memory per lane.  Here ONE WAVE owns one alignment and sweeps the text
// as a systolic array: lane l holds R consecutive pattern rows in registers (H and E of the
// previous column), processes text column c = step - l, and hands its bottom row's H and F, the
// column's text symbol and the running column maximum to lane l+1 with four wave_shr:1 DPP moves
// per step -- no column storage at all, text symbols are read once per wave.  (This is the
// "wavefront / shuffle sweep" shape; the banded kernel does not use it because a 15-wide band
// leaves 3/4 of a wave idle.)
//
// What the reference's blocking makes observable is reproduced exactly:
//  * LOCAL ties: BestSink keeps the LAST maximal cell in the reference's visiting order
//    (block of 8 columns, then row, then column) -- each lane tracks (score, order key) pairs and
//    the wave reduces them;
//  * the boundary column is stored as int16: H and E crossing a block boundary are truncated
//    (gotoh_inl.h:1065, 1476-1477) -- the TRUNC variants do the same, and are selected whenever
//    the host cannot prove the values stay inside int16;
//  * the early exit after each full block, max_i H(i, block end) + missing_cols * match < min_score
//    (:1212-1214): the column maximum flows down the lanes with the data; when the test fires the
//    wave recomputes the prefix it covers, so the sink holds exactly what the reference's holds.
#include "common.h"
#include "full_gotoh_striped.h"
#include <algorithm>
#include <stdlib.h>
#include <string.h>

namespace nvb {

struct FullParams {
    StringSet      pat, txt;
    int32_t        match, mismatch, gap_open, gap_ext;
    const int32_t* min_score;      // nullable: no early exit
    uint32_t       n;
    int32_t*       out_score;
    uint32_t*      out_sink;
    uint8_t*       out_ok;         // nullable: the reference's per-job bool
    uint32_t       max_m, max_n;   // the bounds the register layout and the key packing were chosen for
    uint32_t       blk_log2;       // log2 of the reference's block width: fixes the LOCAL tie order (3 = Gotoh, 4 = SW / ED)
    uint32_t       pattern_blocking;   // 0: blocks of text columns (TextBlockingTag); 1: blocks of pattern rows (PatternBlockingTag)
    // quality-aware scheme (nvBowtie): mismatch by the quality byte of the pattern symbol; nullptr = constant `mismatch`
    const uint8_t* quals; uint64_t n_quals;
    int32_t        mm_lut[256];
    // gap costs of the two boundary lines (they are the text / pattern gap costs, which line gets which depends on the tag)
    int32_t        col_go, col_ge;     // column before the text:  H(r, -1) = col_go + col_ge * r      (non-LOCAL)
    int32_t        row_go, row_ge;     // row above the pattern:   H(-1, c) = row_go + row_ge * c      (GLOBAL)
    // job control (nvbio_hip_alignment_score_qual_jobs): the count on the device, a list of jobs run in place, a device-side gate
    const uint32_t* n_dev;             // nullable: the number of jobs (of list entries, with job_index)
    const uint32_t* job_index;         // nullable: entry k is job job_index[k] of the arrays (one-job-per-wave kernel only)
    const uint32_t* gate;              // nullable: the launch runs iff (*gate > gate_limit) == gate_above
    uint32_t        gate_limit, gate_above;
};
__device__ __forceinline__ bool gate_closed(const FullParams& p) { return p.gate != nullptr && ((*p.gate > p.gate_limit) != (p.gate_above != 0u)); }

__device__ __forceinline__ int32_t dpp_shr1(int32_t first_lane_value, int32_t x)
{
    // lane l receives x of lane l-1; lane 0 keeps first_lane_value   (wave_shr:1 = 0x138 on gfx9)
    return __builtin_amdgcn_update_dpp(first_lane_value, x, 0x138, 0xf, 0xf, false);
}
__device__ __forceinline__ int32_t sext16(int32_t v) { return int32_t(int16_t(v)); }

struct SweepResult { int32_t score; uint32_t sx, sy; uint32_t exit_col; uint32_t pb_exit_row; };   // exit_col / pb_exit_row = 0xFFFFFFFF: ran to the end

// One sweep of the wave over text columns [0, Ncols).  CHECK: evaluate the early-exit test.
template <int TYPE, int R, bool TRUNC>
__device__ __forceinline__ SweepResult sweep(const FullParams& p, const uint64_t pb, const uint64_t tb,
                                             const uint32_t M, const uint32_t Ncols, const uint32_t Nfull,
                                             const bool check, const int32_t min_score)
{
    const uint32_t lane = threadIdx.x & 63u;
    const int32_t  Go = p.gap_open, Ge = p.gap_ext;
    const int32_t  infimum = -32768 - min(Go, Ge);                               // :1153
    const uint32_t lane_last = (M - 1u) / uint32_t(R);
    const uint32_t BS = p.blk_log2, BLK = 1u << BS;                              // columns per reference block (8 Gotoh, 16 SW/ED)
    const uint32_t KM = BLK * 64u * uint32_t(R);                                 // order-key stride per block

    // this lane's rows
    uint32_t q[R]; int32_t Hleft[R], E[R];
    #pragma unroll
    for (int k = 0; k < R; ++k)
    {
        const uint32_t r = lane * R + k;
        q[k] = r < M ? get_symbol(p.pat.s, pb + r) : 255u;
        // context.init (:69-93): the column left of the matrix, kept as short2
        int32_t h0 = (TYPE != NVBIO_HIP_LOCAL) ? p.gap_open + p.gap_ext * int32_t(r) : 0;
        int32_t e0 = (TYPE == NVBIO_HIP_LOCAL) ? 0 : infimum;
        Hleft[k] = TRUNC ? sext16(h0) : h0;
        E[k]     = TRUNC ? sext16(e0) : e0;
    }

    // values handed down the lanes (produced in the previous step)
    int32_t out_h = 0, out_f = 0, out_ch = 0, out_cm = 0;
    int32_t prev_in_h = 0;          // H(row above this lane, c-1): the first row's diagonal
    uint64_t best64 = 0;            // LOCAL: (score << 32 | order key) of the best cell of this lane
    bool     have = false;
    int32_t  sg_score = -(1 << 30); uint32_t sg_col = 0;     // SEMI_GLOBAL / GLOBAL, last-row lane
    uint32_t exit_col = 0xFFFFFFFFu;
    uint32_t grp = 0;

    const uint32_t n_steps = Ncols + lane_last;
    for (uint32_t s = 0; s < n_steps; ++s)
    {
        if ((s & 15u) == 0u && s < Ncols) grp = fetch16_2bit(p.txt.s, tb + s);   // wave-uniform
        const int32_t  c_signed = int32_t(s) - int32_t(lane);
        const uint32_t c = uint32_t(c_signed);
        const bool active = c_signed >= 0 && c < Ncols && lane <= lane_last;

        // inputs from the lane above (its outputs of the previous step = same column c)
        // row -1 for lane 0: H(-1,c) = GLOBAL ? G_o + G_e*c : 0 ; F = infimum ; (:1172-1176)
        const int32_t top_h  = (TYPE == NVBIO_HIP_GLOBAL) ? Go + Ge * int32_t(s) : 0;
        const int32_t ch0    = int32_t((grp >> (2u * (s & 15u))) & 3u);
        int32_t in_h  = dpp_shr1(top_h, out_h);
        int32_t in_f  = dpp_shr1(infimum, out_f);
        int32_t in_ch = dpp_shr1(ch0, out_ch);
        int32_t in_cm = dpp_shr1(-(1 << 30), out_cm);
        // diagonal of the first row: H(row above, c-1); for lane 0: H(-1,-1) = 0, else G_o + G_e*(c-1)
        int32_t diag = prev_in_h;
        if (lane == 0u) diag = (TYPE == NVBIO_HIP_GLOBAL && s > 0u) ? Go + Ge * int32_t(s - 1u) : 0;
        else if (c_signed == 0)
        {
            // first column: the diagonal is the init column's entry of the row above (context.init)
            const int32_t h0 = (TYPE != NVBIO_HIP_LOCAL) ? p.gap_open + p.gap_ext * int32_t(lane * R - 1u) : 0;
            diag = TRUNC ? sext16(h0) : h0;
        }
        if (TRUNC && lane != 0u && (c & 7u) == 0u && c_signed > 0) diag = sext16(diag);   // crossed a block boundary via temp[]
        prev_in_h = in_h;

        int32_t habove = in_h, fabove = in_f, cm = in_cm;
        const bool crossing = TRUNC && (c & 7u) == 0u && c_signed > 0;      // column c starts a block: left values came through temp[]
        #pragma unroll
        for (int k = 0; k < R; ++k)
        {
            const uint32_t r = lane * R + k;
            int32_t hl = Hleft[k], e = E[k];
            if (crossing) { hl = sext16(hl); e = sext16(e); }
            const int32_t f  = max(fabove + Ge, habove + Go);
            e = max(e + Ge, hl + Go);
            const int32_t d  = diag + ((uint32_t(in_ch) == q[k]) ? p.match : p.mismatch);
            int32_t h = max(max(e, f), d);
            if (TYPE == NVBIO_HIP_LOCAL) h = max(h, 0);
            diag = hl;                  // H(r, c-1) is the next row's diagonal (already truncated if crossing)
            if (active)
            {
                Hleft[k] = h; E[k] = e;
                if (r < M)
                {
                    cm = max(cm, h);
                    if (TYPE == NVBIO_HIP_LOCAL)
                    {
                        // order key: block-major, then row, then column within the block
                        const uint32_t key = (c >> BS) * KM + r * BLK + (c & (BLK - 1u));
                        const uint64_t cand = (uint64_t(uint32_t(h)) << 32) | key;
                        if (!have || cand >= best64) { best64 = cand; have = true; }
                    }
                }
            }
            habove = h; fabove = f;
        }
        if (active)
        {
            out_h = habove; out_f = fabove; out_ch = in_ch; out_cm = cm;
        }
        // the lane holding the last pattern row: semi-global / global reports and the early-exit test
        if (active && lane == lane_last)
        {
            const uint32_t klast = (M - 1u) - lane_last * uint32_t(R);
            int32_t hlast = Hleft[0];
            #pragma unroll
            for (int k = 1; k < R; ++k) if (uint32_t(k) == klast) hlast = Hleft[k];
            if (TYPE == NVBIO_HIP_SEMI_GLOBAL) { if (sg_score <= hlast) { sg_score = hlast; sg_col = c; } }
            if (TYPE == NVBIO_HIP_GLOBAL && c + 1u == Nfull) { sg_score = hlast; sg_col = c; }
            // early exit (:1212-1214): only after blocks that are not the last one
            if (check && (c & 7u) == 7u && exit_col == 0xFFFFFFFFu)
            {
                const uint32_t nb = 8u * ((Nfull + 7u) / 8u);
                const uint32_t end_block = nb > 8u ? nb : 8u;
                const uint32_t block = c - 7u;
                if (block + 8u < end_block)
                {
                    const int32_t missing = int32_t(Nfull - block - 8u);
                    if (cm + missing * p.match < min_score) exit_col = c;
                }
            }
        }
    }

    // gather the result in every lane
    SweepResult res;
    res.exit_col = uint32_t(__shfl(int32_t(exit_col), int32_t(lane_last)));
    res.score = -(1 << 30); res.sx = res.sy = 0xFFFFFFFFu; res.pb_exit_row = 0xFFFFFFFFu;
    if (TYPE == NVBIO_HIP_LOCAL)
    {
        uint64_t b = have ? best64 : 0ull; uint32_t hv = have ? 1u : 0u;
        #pragma unroll
        for (int off = 32; off >= 1; off >>= 1)
        {
            const uint32_t olo = uint32_t(__shfl_xor(int32_t(uint32_t(b)), off));
            const uint32_t ohi = uint32_t(__shfl_xor(int32_t(uint32_t(b >> 32)), off));
            const uint32_t ohv = uint32_t(__shfl_xor(int32_t(hv), off));
            const uint64_t o = (uint64_t(ohi) << 32) | olo;
            if (ohv && (!hv || o > b)) { b = o; hv = 1u; }
        }
        if (hv) {
            const uint32_t key = uint32_t(b);
            const uint32_t col = (key / KM) * BLK + (key & (BLK - 1u)), row = (key % KM) >> BS;
            res.score = int32_t(uint32_t(b >> 32)); res.sx = col + 1u; res.sy = row + 1u;
        }
    }
    else
    {
        const int32_t  sc  = __shfl(sg_score, int32_t(lane_last));
        const uint32_t col = uint32_t(__shfl(int32_t(sg_col), int32_t(lane_last)));
        const bool reported = (TYPE == NVBIO_HIP_SEMI_GLOBAL) ? (Ncols > 0u) : (Ncols == Nfull && Nfull > 0u);
        if (reported) { res.score = sc; res.sx = col + 1u; res.sy = M; }
    }
    return res;
}

// ---------------------------------------------------------------------------------------------
// Fast sweep: used when the host has proved that no DP value can leave int16 (so the boundary
// column's truncation is the identity), scores stay below 2048 and texts below 2^20 symbols.
// Everything a cell needs is a 2-cycle 16-bit VOP2 op (profiles/r01/valu_probe.txt) plus the
// compare/select of the substitution score; H is carried as HG = H + G_o exactly as in the banded
// kernel, which serves the E of the next column, the F of the next row and, with the substitution
// scores pre-biased by -G_o, the diagonal.  LOCAL: each ROW keeps max(score << 20 | column) -- for
// one row the reference's visiting order is the column order, so a per-row maximum with the column
// in the low bits IS "last maximal cell of the row"; rows and lanes are merged once at the end with
// the full (block, row, column) order key.
// ---------------------------------------------------------------------------------------------
template <int TYPE>
__device__ __forceinline__ void cell16(uint32_t& e, uint32_t& hlg, uint32_t& hab_g, uint32_t& fab, uint32_t& diag_g,
                                       const uint32_t ch, const uint32_t tlo, const uint32_t thi, const uint32_t go, const uint32_t ge,
                                       uint32_t& h_out, uint32_t& bk, const uint32_t s15)
{
    // in : e = E(r,c-1), hlg = HG(r,c-1), hab_g = HG(r-1,c), fab = F(r-1,c), diag_g = HG(r-1,c-1)
    // out: e = E(r,c),   hlg = HG(r,c),   hab_g = HG(r,c),   fab = F(r,c),   diag_g = HG(r,c-1), h_out = H(r,c)
    // The substitution score is a table lookup, not a compare + select (6.6 issue cycles -> 4.3, no VCC; profiles/r02/valu_probe.txt):
    // ch is the text symbol as the byte selector of its 16-bit entry (0x0C0C0100 + 0x0202 * g), {tlo, thi} the row's four entries
    // (match score - G_o where the entry's symbol is the row's, its mismatch score - G_o elsewhere; a row past the pattern: all mismatch).
    uint32_t f, d, h, t;
    if (TYPE == NVBIO_HIP_LOCAL)
        asm("v_perm_b32 %[d], %[thi], %[tlo], %[ch]\n\t"
            "v_add_u16 %[f], %[fab], %[ge]\n\t"
            "v_add_u16 %[e], %[e], %[ge]\n\t"
            "v_max_i16 %[f], %[f], %[hab]\n\t"
            "v_add_u16 %[d], %[dg], %[d]\n\t"
            "v_max_i16 %[e], %[e], %[hl]\n\t"
            "v_max_i16 %[h], %[f], %[d]\n\t"
            "v_max_i16 %[h], %[h], %[e]\n\t"
            "v_max_i16 %[h], 0, %[h]\n\t"
            "v_add_u16 %[d], %[s15], %[h]\n\t"
            "v_add_u16 %[t], %[h], %[go]\n\t"
            "v_max_i16 %[bk], %[bk], %[d]"
            : [f] "=&v"(f), [d] "=&v"(d), [h] "=&v"(h), [t] "=&v"(t), [e] "+v"(e), [bk] "+v"(bk)
            : [ch] "v"(ch), [tlo] "v"(tlo), [thi] "v"(thi), [fab] "v"(fab), [ge] "v"(ge), [hab] "v"(hab_g), [dg] "v"(diag_g), [go] "v"(go), [hl] "v"(hlg),
              [s15] "v"(s15));       // (a VGPR: VOP2 with an SGPR source issues at half rate, 4.2 against 2.3 cycles)
    else
        asm("v_perm_b32 %[d], %[thi], %[tlo], %[ch]\n\t"
            "v_add_u16 %[f], %[fab], %[ge]\n\t"
            "v_add_u16 %[e], %[e], %[ge]\n\t"
            "v_max_i16 %[f], %[f], %[hab]\n\t"
            "v_add_u16 %[d], %[dg], %[d]\n\t"
            "v_max_i16 %[e], %[e], %[hl]\n\t"
            "v_max_i16 %[h], %[f], %[d]\n\t"
            "v_max_i16 %[h], %[h], %[e]\n\t"
            "v_add_u16 %[t], %[h], %[go]"
            : [f] "=&v"(f), [d] "=&v"(d), [h] "=&v"(h), [t] "=&v"(t), [e] "+v"(e)
            : [ch] "v"(ch), [tlo] "v"(tlo), [thi] "v"(thi), [fab] "v"(fab), [ge] "v"(ge), [hab] "v"(hab_g), [dg] "v"(diag_g), [go] "v"(go), [hl] "v"(hlg));
    diag_g = hlg; hlg = t; hab_g = t; fab = f; h_out = h;      // old HG(r,c-1) is the next row's diagonal: a renaming
}

__device__ __forceinline__ uint32_t c16(int32_t v) { return uint32_t(v) & 0xFFFFu; }
__device__ __forceinline__ uint32_t max16u(uint32_t a, uint32_t b) { uint32_t r; asm("v_max_i16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ uint32_t min16u(uint32_t a, uint32_t b) { uint32_t r; asm("v_min_i16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

// MULTI: several jobs share the wave (full_gotoh_score_multi_kernel): the wave is cut in segments of `W` lanes, segment j sweeps job j,
// and `lane` is the lane's index INSIDE its segment.  The hand-off between lanes is the same two instructions per value for every shape:
// a wave_shr:1 DPP move and a select on the mask of the segments' first lanes -- a first lane takes the row above its
// matrix (H(-1, c) + G_o, F = infimum, the text symbol of column c of ITS job, an empty column maximum), every other lane what the lane
// above computed one step ago.  Everything a job owns (M, Ncols, lane_last, klast ...) is a per-lane value, and the loop bounds are the
// wave's maxima / minima.
template <int TYPE, int R, bool CHECK, bool PBX = false, bool MULTI = false, bool BKL = false>      // PBX: keep every row's maximum over the text (pattern-blocking early exit, non-LOCAL types)
struct Sweep16                                                                                     // BKL: LOCAL's per-row records live in LDS (bkl), not in registers
{
    const FullParams& p;
    uint32_t lane, lane_last, klast, M, Ncols, Nfull;
    uint32_t wlane, seg_base, seg_w; bool seg_valid;              // MULTI: lane in the wave, first lane / width of the segment, segment holds a job
    uint64_t head_mask;                                           // the first lane of every segment (lane 0 of a wave that sweeps one job)
    int32_t  Go, Ge, min_score;
    uint32_t go, ge, rge, inf16, init_above_g;
    uint32_t tlo[R], thi[R];              // per-row substitution table: four 16-bit entries, scores pre-biased by -G_o (cell16)
    uint64_t tb;
    uint32_t q[R], HLG[R], E[R], bestk[BKL ? 1 : R], rmax[PBX ? R : 1];
    // BKL: bestk[k] of this lane at bkl[k * 64 + wlane].  The records are touched once per 16 steps (fold) and at the end, so LDS costs
    // nothing measurable -- and ten rows per lane of LOCAL then fit 168 VGPRs, i.e. three waves per SIMD instead of two
    uint32_t* bkl;
    __device__ __forceinline__ uint32_t get_bk(const int k) const { return BKL ? bkl[k * 64 + int(wlane)] : bestk[BKL ? 0 : k]; }
    __device__ __forceinline__ void     set_bk(const int k, const uint32_t v) { if (BKL) bkl[k * 64 + int(wlane)] = v; else bestk[BKL ? 0 : k] = v; }
    // LOCAL carries every value x16 (the host admits LOCAL only below 2048) so that "score, then later column" is one 16-bit maximum:
    // bk16 = max over the current group of 16 steps of H*16 + (step & 15); the groups are folded into bestk = score << 20 | column
    // at wave-uniform times.  (The 32-bit form, (h << 20 | c) and v_max_u32 per cell, cost 8.6 issue cycles of a cell's 36; this costs 4.6.)
    static constexpr int SC = (TYPE == NVBIO_HIP_LOCAL) ? 16 : 1;
    uint32_t bk16[TYPE == NVBIO_HIP_LOCAL ? R : 1];
    uint32_t lim[CHECK ? R : 1];          // CHECK: 0x7FFF for this lane's valid rows, 0x8000 for rows past the pattern (they drop out of the column maximum)
    uint32_t out_hg, out_f, out_ch, out_cm, prev_in_hg;
    int32_t  sg_score; uint32_t sg_col, exit_col, grp;
    uint32_t sg_hg16;                     // SEMI_GLOBAL: best HG of this lane's last row so far (16-bit), its column in sg_col
    uint32_t top_hg, top_prev_hg;         // GLOBAL: HG(-1,s), HG(-1,s-1)
    uint32_t kl;                          // this lane's last valid row (for the last-row reports)
    bool     pb_check;                    // pattern blocking with a min_score: evaluate its early exit after the sweep

    __device__ __forceinline__ Sweep16(const FullParams& _p) : p(_p) {}

    /// MULTI: _seg_w = lanes per segment; _valid = this lane's segment holds a job
    __device__ __forceinline__ void init(const uint64_t pb, const uint64_t _tb, uint32_t _M, uint32_t _Ncols, uint32_t _Nfull, int32_t _min_score,
                                         const uint32_t _seg_w = 64u, const bool _valid = true)
    {
        wlane = threadIdx.x & 63u; seg_w = _seg_w; seg_valid = _valid;
        lane = MULTI ? wlane % seg_w : wlane; seg_base = wlane - lane;
        head_mask = __ballot(lane == 0u);
        M = _M; Ncols = _Ncols; Nfull = _Nfull; min_score = _min_score; tb = _tb; pb_check = false;
        Go = p.gap_open; Ge = p.gap_ext;
        const int32_t infimum = -32768 - min(Go, Ge) * SC;
        lane_last = (M - 1u) / uint32_t(R);
        klast = (M - 1u) - lane_last * uint32_t(R);
        kl = lane < lane_last ? uint32_t(R - 1) : (lane == lane_last ? klast : 0u);
        go = c16(Go * SC); ge = c16(Ge * SC); rge = c16(p.row_ge); inf16 = c16(infimum);
        const uint32_t sM = c16((p.match - Go) * SC), sX = c16((p.mismatch - Go) * SC);
        #pragma unroll
        for (int k = 0; k < R; ++k)
        {
            const uint32_t r = lane * R + k;
            q[k] = r < M ? get_symbol(p.pat.s, pb + r) : 255u;
            HLG[k] = c16(((TYPE != NVBIO_HIP_LOCAL) ? p.col_go + p.col_ge * int32_t(r) : 0) + Go * SC);
            const uint32_t sXr = (p.quals && r < M) ? c16((p.mm_lut[p.quals[min(pb + r, p.n_quals - 1u)]] - Go) * SC) : sX;
            tlo[k] = (q[k] == 0u ? sM : sXr) | ((q[k] == 1u ? sM : sXr) << 16);
            thi[k] = (q[k] == 2u ? sM : sXr) | ((q[k] == 3u ? sM : sXr) << 16);
            E[k]   = c16((TYPE == NVBIO_HIP_LOCAL) ? 0 : infimum);
            set_bk(k, 0u);
            if (TYPE == NVBIO_HIP_LOCAL) bk16[k] = 0x8000u;
            if (PBX) rmax[k] = 0x8000u;
            if (CHECK) lim[k] = (uint32_t(k) <= kl) ? 0x7FFFu : 0x8000u;
        }
        out_hg = out_f = out_ch = out_cm = 0;
        if (MULTI && !seg_valid) { Ncols = 0u; }       // lanes of a segment without a job never enter the matrix
        prev_in_hg = (lane == 0u) ? go : 0u;          // lane 0's first diagonal: the corner above the matrix, H(-1,-1) = 0
        sg_score = -(1 << 30); sg_col = 0; exit_col = 0xFFFFFFFFu; grp = 0; sg_hg16 = 0x8000u;
        top_hg = c16(p.row_go + Go); top_prev_hg = go;
        init_above_g = c16(((TYPE != NVBIO_HIP_LOCAL) ? p.col_go + p.col_ge * int32_t(lane * R - 1u) : 0) + Go * SC);
    }

    // the last-row lane's reports and the early-exit test for column c (cm = full column maximum)
    __device__ __forceinline__ void last_row(const uint32_t c, const uint32_t cm, const uint32_t hg_last)
    {
        const int32_t hlast = int32_t(int16_t(hg_last)) - Go;
        if (TYPE == NVBIO_HIP_SEMI_GLOBAL) { if (int16_t(hg_last) >= int16_t(sg_hg16) && (!CHECK || exit_col == 0xFFFFFFFFu)) { sg_hg16 = hg_last; sg_col = c; } }
        if (TYPE == NVBIO_HIP_GLOBAL && c + 1u == Nfull) { sg_score = hlast; sg_col = c; }
        early_exit_test(c, cm);
    }
    __device__ __forceinline__ void early_exit_test(const uint32_t c, const uint32_t cm)
    {
        if (CHECK && (c & 7u) == 7u && exit_col == 0xFFFFFFFFu)
        {
            const uint32_t nb = 8u * ((Nfull + 7u) / 8u);
            const uint32_t end_block = nb > 8u ? nb : 8u;
            const uint32_t block = c - 7u;
            if (block + 8u < end_block && int32_t(int16_t(cm)) / SC + int32_t(Nfull - block - 8u) * p.match < min_score) exit_col = c;
        }
    }

    // one step; PRED = lanes may be outside the matrix (ramp-up / ramp-down)
    template <bool PRED>
    __device__ __forceinline__ void step(const uint32_t s, const uint32_t ch0, const uint32_t s15)     // s15 = s & 15 (a constant in the unrolled steady state)
    {
        const uint32_t c = s - lane;
        const uint32_t th = (TYPE == NVBIO_HIP_GLOBAL) ? top_hg : go;          // HG(-1,c) for a first lane
        // The hand-off: in = first lane ? (row above the matrix) : (the lane above's out).  One VOP2 each -- the DPP source shifts, the
        // select takes the first lanes' mask from VCC -- where a DPP move into a register preset with the first lane's value costs two and
        // knows lane 0 only.  (In isolation v_cndmask_b32_dpp on VCC measures 22.7 issue cycles against 6.9 for a DPP move + an e64 select
        // on an SGPR mask, profiles/r03/valu_probe.txt; inside this sweep the order is the other way round, 3.73 against 3.54 TCUPS.)
        // gfx9 wants two wait states between a VALU write of a VGPR and a DPP read of it, and the compiler's hazard recognizer does not
        // look inside an asm block: whatever the scheduler placed before this block, `s_nop 1` after the s_mov guarantees them (the registers
        // the step before wrote last -- out_hg, and out_cm in the CHECK form -- are also read last).
        uint32_t in_hg, in_f, in_ch, in_cm = 0u;
        if (CHECK)
            asm("s_mov_b64 vcc, %[m]\n\t"
                "s_nop 1\n\t"
                "v_cndmask_b32_dpp %[ch], %[och], %[hch], vcc wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                "v_cndmask_b32_dpp %[f], %[of], %[hf], vcc wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                "v_cndmask_b32_dpp %[hg], %[ohg], %[hh], vcc wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                "v_cndmask_b32_dpp %[cm], %[ocm], %[hcm], vcc wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0"
                : [cm] "=&v"(in_cm), [ch] "=&v"(in_ch), [f] "=&v"(in_f), [hg] "=&v"(in_hg)
                : [ocm] "v"(out_cm), [och] "v"(out_ch), [of] "v"(out_f), [ohg] "v"(out_hg),
                  [hcm] "v"(0x8000u), [hch] "v"(ch0), [hf] "v"(inf16), [hh] "v"(th), [m] "s"(head_mask) : "vcc");
        else
            asm("s_mov_b64 vcc, %[m]\n\t"
                "s_nop 1\n\t"
                "v_cndmask_b32_dpp %[ch], %[och], %[hch], vcc wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                "v_cndmask_b32_dpp %[f], %[of], %[hf], vcc wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                "v_cndmask_b32_dpp %[hg], %[ohg], %[hh], vcc wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0"
                : [ch] "=&v"(in_ch), [f] "=&v"(in_f), [hg] "=&v"(in_hg)
                : [och] "v"(out_ch), [of] "v"(out_f), [ohg] "v"(out_hg),
                  [hch] "v"(ch0), [hf] "v"(inf16), [hh] "v"(th), [m] "s"(head_mask) : "vcc");
        // HG(r-1, c-1): what came down the lanes one step ago.  Lane 0 needs no case of its own -- its in_hg is the row above the matrix,
        // th, so one step ago it was that row's previous column (prev_in_hg starts as the corner) -- and a lane's first column (the
        // boundary column's value instead) only occurs while lanes are still entering the matrix.
        uint32_t diag_g = prev_in_hg;
        if (PRED && lane != 0u && c == 0u) diag_g = init_above_g;
        prev_in_hg = in_hg;
        if (TYPE == NVBIO_HIP_GLOBAL) { top_prev_hg = top_hg; uint32_t t; asm("v_add_u16 %0, %1, %2" : "=v"(t) : "v"(top_hg), "v"(rge)); top_hg = t; }

        const bool active = !PRED || (int32_t(c) >= 0 && c < Ncols && lane <= lane_last);
        if (active)
        {
            uint32_t hab_g = in_hg, fab = in_f, cm = in_cm, h = 0, hg_last = 0;
            #pragma unroll
            for (int k = 0; k < R; ++k)
            {
                // LOCAL: bk16 = max(bk16, h + s15) inside the cell (h < 2^15 and a multiple of 16: no carry out of the low half)
                cell16<TYPE>(E[k], HLG[k], hab_g, fab, diag_g, in_ch, tlo[k], thi[k], go, ge, h, bk16[TYPE == NVBIO_HIP_LOCAL ? k : 0], s15);
                if (PBX) rmax[k] = max16u(rmax[k], h);
                if (CHECK) cm = max16u(cm, min16u(h, lim[k]));
            }
            if (TYPE != NVBIO_HIP_LOCAL)
            {
                // HG of the pattern's last row, for the lane that holds it: row klast of that lane.  klast is the same number in every
                // lane, and only the last-row lane's record is ever read, so every lane follows its own row klast: selects on a
                // wave-uniform condition instead of a compare + select per cell.
                hg_last = HLG[0];
                #pragma unroll
                for (int k = 1; k < R; ++k) hg_last = (klast == uint32_t(k)) ? HLG[k] : hg_last;
            }
            out_hg = hab_g; out_f = fab; out_ch = in_ch; out_cm = cm;
            if (PRED)
            {
                if (TYPE != NVBIO_HIP_LOCAL || CHECK) { if (lane == lane_last) last_row(c, cm, hg_last); }
            }
            else
            {
                // steady state: no branch.  Every lane follows its own last row (only the last-row
                // lane's record is read at the end); the final column never falls in this phase.
                if (TYPE == NVBIO_HIP_SEMI_GLOBAL)
                {
                    // (after the reference's early exit the last row's record is frozen: its sink saw columns <= exit only)
                    const bool upd = int16_t(hg_last) >= int16_t(sg_hg16) && (!CHECK || exit_col == 0xFFFFFFFFu);
                    sg_hg16 = upd ? hg_last : sg_hg16;
                    sg_col  = upd ? c : sg_col;
                }
                if (CHECK) { if (((s - lane_last) & 7u) == 7u) { if (lane == lane_last) early_exit_test(c, cm); } }   // wave-uniform outer test
            }
        }
    }

    // LOCAL: the finished group of 16 steps [base, base + 16) into the rows' all-time records (every lane at once)
    __device__ __forceinline__ void fold(const uint32_t base)
    {
        if (TYPE == NVBIO_HIP_LOCAL)
        {
            #pragma unroll
            for (int k = 0; k < R; ++k)
            {
                const uint32_t t = bk16[k];
                const uint32_t cand = ((t >> 4) << 20) | ((base + (t & 15u) - lane) & 0xFFFFFu);
                if (BKL) { if (t != 0x8000u) { const uint32_t b0 = get_bk(k); if (cand > b0) set_bk(k, cand); } }
                else bestk[BKL ? 0 : k] = (t != 0x8000u) ? max(bestk[BKL ? 0 : k], cand) : bestk[BKL ? 0 : k];      // 0x8000: the lane sat outside the matrix for the whole group
                bk16[k] = 0x8000u;
            }
        }
    }

    __device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) const
    { for (int o = 32; o >= 1; o >>= 1) v = max(v, uint32_t(__shfl_xor(int32_t(v), o))); return uint32_t(__builtin_amdgcn_readfirstlane(int(v))); }
    __device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) const
    { for (int o = 32; o >= 1; o >>= 1) v = min(v, uint32_t(__shfl_xor(int32_t(v), o))); return uint32_t(__builtin_amdgcn_readfirstlane(int(v))); }
    /// the text group of 16 symbols starting at column s: the job's own (wave-uniform -> scalar) or, MULTI, per lane of the job the lane feeds
    __device__ __forceinline__ uint32_t text_group(const uint32_t s) const
    {
        if (MULTI) return fetch16_2bit(p.txt.s, tb + s);
        return uint32_t(__builtin_amdgcn_readfirstlane(int(fetch16_2bit(p.txt.s, tb + s))));
    }

    __device__ __forceinline__ SweepResult run()
    {
        // MULTI: the wave runs to the longest job's last step; the unpredicated steady state covers the steps at which every job is inside it
        const bool rows = !MULTI || (seg_valid && Ncols > 0u);
        const uint32_t n_steps = MULTI ? wave_max_u32(rows ? Ncols + lane_last : 0u) : Ncols + lane_last;
        const uint32_t fetch_cols = MULTI ? wave_max_u32(Ncols) : Ncols;
        uint32_t s = 0;
        // ramp-up, up to the first 16-aligned step at which every row-holding lane is inside the matrix and past its first column
        const uint32_t s_fast = MULTI ? wave_max_u32(rows ? ((lane_last + 16u) & ~15u) : 0u) : ((lane_last + 16u) & ~15u);         // (strictly past every lane's first column: step() tests for that column only while PRED)
        const uint32_t fast_end = MULTI ? wave_min_u32(rows ? Ncols : 0xFFFFFFFFu) : Ncols;
        // (a wave that owns one job: the text group is wave-uniform, so the symbol -> selector arithmetic runs on the scalar unit)
        auto sel = [](const uint32_t g) { return 0x0C0C0100u + 0x0202u * g; };
        for (; s < n_steps && s < s_fast; ++s)
        {
            if ((s & 15u) == 0u && s < fetch_cols) grp = text_group(s);
            step<true>(s, sel((grp >> (2u * (s & 15u))) & 3u), s & 15u);
            if ((s & 15u) == 15u) fold(s - 15u);
        }
        // steady state: 16 unpredicated steps per text group (lanes past the last row compute harmlessly)
        for (; s + 16u < fast_end && s + 16u < n_steps; s += 16u)        // strict: the last column is always handled by the tail
        {
            grp = text_group(s);
            #pragma unroll
            for (int u = 0; u < 16; ++u) step<false>(s + u, sel((grp >> (2 * u)) & 3u), uint32_t(u));
            fold(s);
        }
        // tail and ramp-down
        for (; s < n_steps; ++s)
        {
            if ((s & 15u) == 0u && s < fetch_cols) grp = text_group(s);
            step<true>(s, sel((grp >> (2u * (s & 15u))) & 3u), s & 15u);
            if ((s & 15u) == 15u) fold(s - 15u);
        }
        if ((s & 15u) != 0u) fold(s & ~15u);

        // lane holding a job's last row / partner lane of step `off` of an all-reduce over the job's lanes (MULTI: a rotation inside the
        // segment -- max / min are idempotent, so windows of 1, 2, 4 ... lanes that wrap around cover any segment width)
        const int last_lane = int(MULTI ? seg_base + lane_last : lane_last);
        auto peer = [&](const int off) { return MULTI ? int(seg_base + (lane + uint32_t(off)) % seg_w) : int(wlane ^ uint32_t(off)); };
        SweepResult res;
        res.exit_col = uint32_t(__shfl(int32_t(exit_col), last_lane));
        res.score = -(1 << 30); res.sx = res.sy = 0xFFFFFFFFu;
        res.pb_exit_row = 0xFFFFFFFFu;
        const uint32_t BS = p.blk_log2, BLK = 1u << BS, KM = BLK * 64u * uint32_t(R);
        const bool PB = p.pattern_blocking != 0u;
        const uint32_t nvalid = lane > lane_last ? 0u : (lane == lane_last ? klast + 1u : uint32_t(R));
        if (pb_check)
        {
            // pattern blocking: after every non-final block of BLK pattern rows the reference tests
            // max_i H[i][block end] + (M - block end) * match < min_score (gotoh_inl.h:826-830, sw_inl.h:697-701)
            uint32_t first = 0xFFFFFFFFu;
            #pragma unroll
            for (int k = 0; k < R; ++k)
            {
                const uint32_t r = lane * R + k;
                if (uint32_t(k) < nvalid && ((r + 1u) & (BLK - 1u)) == 0u && r + 1u < M)
                {
                    const int32_t rm = (TYPE == NVBIO_HIP_LOCAL) ? int32_t(get_bk(k) >> 20) : int32_t(int16_t(rmax[PBX ? k : 0]));
                    if (rm + int32_t(M - (r + 1u)) * p.match < min_score) first = min(first, r);
                }
            }
            #pragma unroll
            for (int off = 32; off >= 1; off >>= 1) first = min(first, uint32_t(__shfl(int32_t(first), peer(off))));
            res.pb_exit_row = first;
        }
        if (TYPE == NVBIO_HIP_LOCAL)
        {
            // merge this lane's rows, then the lanes, with the reference's order key
            uint64_t b = 0; uint32_t hv = 0;
            #pragma unroll
            for (int k = 0; k < R; ++k)
            {
                if (uint32_t(k) < nvalid && Ncols > 0u) {
                    const uint32_t bkk = get_bk(k), hh = bkk >> 20, cc = bkk & 0xFFFFFu, r = lane * R + k;
                    // text blocking: block of columns -> row -> column in block; pattern blocking: block of rows -> column -> row in block
                    const uint32_t key = PB ? (((r >> BS) << (20u + BS)) | (cc << BS) | (r & (BLK - 1u)))
                                            : ((cc >> BS) * KM + r * BLK + (cc & (BLK - 1u)));
                    const uint64_t cand = (uint64_t(hh) << 32) | key;
                    if (!hv || cand > b) { b = cand; hv = 1u; }
                }
            }
            #pragma unroll
            for (int off = 32; off >= 1; off >>= 1)
            {
                const int pr = peer(off);
                const uint32_t olo = uint32_t(__shfl(int32_t(uint32_t(b)), pr));
                const uint32_t ohi = uint32_t(__shfl(int32_t(uint32_t(b >> 32)), pr));
                const uint32_t ohv = uint32_t(__shfl(int32_t(hv), pr));
                const uint64_t o = (uint64_t(ohi) << 32) | olo;
                if (ohv && (!hv || o > b)) { b = o; hv = 1u; }
            }
            if (hv) {
                const uint32_t key = uint32_t(b);
                const uint32_t col = PB ? ((key >> BS) & 0xFFFFFu) : ((key / KM) * BLK + (key & (BLK - 1u)));
                const uint32_t row = PB ? ((key >> (20u + BS)) * BLK + (key & (BLK - 1u))) : ((key % KM) >> BS);
                res.score = int32_t(uint32_t(b >> 32)); res.sx = col + 1u; res.sy = row + 1u;
            }
        }
        else
        {
            if (TYPE == NVBIO_HIP_SEMI_GLOBAL) sg_score = int32_t(int16_t(sg_hg16)) - Go;
            const int32_t  sc  = __shfl(sg_score, last_lane);
            const uint32_t col = uint32_t(__shfl(int32_t(sg_col), last_lane));
            const bool reported = (TYPE == NVBIO_HIP_SEMI_GLOBAL) ? (Ncols > 0u) : (Ncols == Nfull && Nfull > 0u);
            if (reported) { res.score = sc; res.sx = col + 1u; res.sy = M; }
        }
        return res;
    }
};

template <int TYPE, int R, bool CHECK, bool BKL = false>
__device__ __forceinline__ SweepResult sweep16(const FullParams& p, const uint64_t pb, const uint64_t tb,
                                               const uint32_t M, const uint32_t Ncols, const uint32_t Nfull, const int32_t min_score, uint32_t* bkl = nullptr)
{
    Sweep16<TYPE, R, CHECK, false, false, BKL> sw(p);
    sw.bkl = bkl;
    sw.init(pb, tb, M, Ncols, Nfull, min_score);
    return sw.run();
}
// pattern blocking with a min_score: one sweep that also yields the reference's exit row (0xFFFFFFFF = none)
template <int TYPE, int R>
__device__ __forceinline__ SweepResult sweep16_pb(const FullParams& p, const uint64_t pb, const uint64_t tb,
                                                  const uint32_t M, const uint32_t N, const int32_t min_score)
{
    Sweep16<TYPE, R, false, (TYPE != NVBIO_HIP_LOCAL)> sw(p);
    sw.init(pb, tb, M, N, N, min_score);
    sw.pb_check = true;
    return sw.run();
}

// MODE (FAST only; see full_gotoh_score_multi_kernel): 0 = no min_score, 1 = min_score with text blocking, 2 = min_score with pattern blocking
template <int TYPE, int R, bool TRUNC, bool FAST, int MODE = 0>
__global__ void __launch_bounds__(256)
full_gotoh_score_kernel(const FullParams p)
{
    const uint32_t entry = (blockIdx.x * 256u + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63u;
    if (entry >= (p.n_dev ? *p.n_dev : p.n) || gate_closed(p)) return;
    const uint32_t job  = p.job_index ? p.job_index[entry] : entry;
    const uint32_t M  = p.pat.length ? p.pat.length[job] : p.pat.fixed_length;
    const uint32_t N  = p.txt.length ? p.txt.length[job] : p.txt.fixed_length;
    const uint64_t pb = p.pat.begin[job], tb = p.txt.begin[job];
    const bool     check = p.min_score != nullptr;
    const int32_t  min_score = check ? p.min_score[job] : -(1 << 30);

    int32_t score = -(1 << 30); uint32_t sx = 0xFFFFFFFFu, sy = 0xFFFFFFFFu; uint32_t ok = 1u;
    if (M > p.max_m || N > p.max_n)
    {
        // a job longer than the caller stated: its rows would not fit the lanes' registers (or its columns the packed
        // keys).  Never a plausible number: the failed-alignment record with ok = 0.
        ok = 0u;
    }
    else if (M == 0u)
    {
        // no rows: only the row above the matrix is ever reported (:1203-1207, :1404-1421)
        const uint32_t xb = check ? empty_pattern_exit_block(N, 8u, p.match, min_score) : 0xFFFFFFFFu;
        const bool exits = xb != 0xFFFFFFFFu;
        if (exits) { ok = 0u; if (TYPE == NVBIO_HIP_SEMI_GLOBAL) { score = 0; sx = xb + 8u; sy = 0u; } }
        else if (N > 0u) {
            if (TYPE == NVBIO_HIP_SEMI_GLOBAL) { score = 0; sx = N; sy = 0u; }
            if (TYPE == NVBIO_HIP_GLOBAL)      { score = p.row_go + p.row_ge * int32_t(N - 1u); sx = N; sy = 0u; }
        }
    }
    else if constexpr (FAST && MODE == 2)
    {
        // PatternBlockingTag with a min_score: the exit test runs per block of pattern rows (gotoh_inl.h:826-830)
        const uint32_t BLK = 1u << p.blk_log2;
        const uint32_t n_blocks = (M + BLK - 1u) / BLK;
        if (N == 0u)
        {
            // no text row was visited: max_score is still its minimum, so the first non-final block exits
            if (n_blocks > 1u) ok = 0u;
            else if (TYPE == NVBIO_HIP_GLOBAL) { score = p.col_go + p.col_ge * int32_t(M - 1u); sx = 0u; sy = M; }
        }
        else
        {
            SweepResult r = sweep16_pb<TYPE, R>(p, pb, tb, M, N, min_score);
            if (r.pb_exit_row != 0xFFFFFFFFu)
            {
                ok = 0u;
                // the sink saw pattern rows [0, exit row] only: LOCAL reports of those blocks, nothing for the other types
                if (TYPE == NVBIO_HIP_LOCAL) { r = sweep16<TYPE, R, false>(p, pb, tb, r.pb_exit_row + 1u, N, N, min_score); score = r.score; sx = r.sx; sy = r.sy; }
            }
            else { score = r.score; sx = r.sx; sy = r.sy; }
        }
    }
    else
    {
        SweepResult r;
        if constexpr (!FAST)          r = sweep<TYPE, R, TRUNC>(p, pb, tb, M, N, N, check, min_score);
        else if constexpr (MODE == 1) r = sweep16<TYPE, R, true>(p, pb, tb, M, N, N, min_score);
        else                          r = sweep16<TYPE, R, false>(p, pb, tb, M, N, N, min_score);
        if (r.exit_col != 0xFFFFFFFFu)
        {
            // the reference returned false after this block: its sink saw columns [0, exit_col] only
            ok = 0u;
            if (!FAST)
                r = sweep<TYPE, R, TRUNC>(p, pb, tb, M, r.exit_col + 1u, N, false, min_score);
            else if (TYPE == NVBIO_HIP_GLOBAL)
                { r.score = -(1 << 30); r.sx = r.sy = 0xFFFFFFFFu; }            // the only report is at the last column
            else if (TYPE == NVBIO_HIP_LOCAL)
            {
                // if the best cell of the whole matrix lies at or before the exit column it is also the best of the
                // columns the reference visited (same total order); only otherwise sweep the truncated text again
                if (!(r.sx != 0xFFFFFFFFu && r.sx - 1u <= r.exit_col))
                    r = sweep16<TYPE, R, false>(p, pb, tb, M, r.exit_col + 1u, N, min_score);
            }
            // SEMI_GLOBAL: the last row's record was frozen at the exit column inside the sweep
        }
        score = r.score; sx = r.sx; sy = r.sy;
        // pattern blocking, GLOBAL, empty text: save_Mth reports the initial row (gotoh_inl.h:896-897)
        if (FAST && p.pattern_blocking != 0u && TYPE == NVBIO_HIP_GLOBAL && N == 0u) { score = p.col_go + p.col_ge * int32_t(M - 1u); sx = 0u; sy = M; }
    }
    if (lane == 0u)
    {
        p.out_score[job] = score;
        reinterpret_cast<uint2*>(p.out_sink)[job] = make_uint2(sx, sy);
        if (p.out_ok) p.out_ok[job] = uint8_t(ok);
    }
}

// ---------------------------------------------------------------------------------------------
// Several jobs per wave (Sweep16<..., MULTI>): short patterns leave most lanes of a 64-lane systolic sweep idle (150 rows at R = 3:
// 50 lanes; the four lane-to-lane moves of a step are paid for 3 cells), and their short texts make fill and drain a large share of the
// steps.  Here the wave is cut into n_seg segments of seg_w lanes (2 x 32, 3 x 21, 4 x 16), each sweeping its own job with R = 5 or 6
// rows per lane: 150-bp mates run two per wave on 30 + 30 lanes.  Every alignment type, on the 16-bit sweep.  A job whose early exit
// fires needs a second sweep over a prefix of its rows / columns (as in full_gotoh_score_kernel): the wave then runs that job alone
// on the single-job sweep, segment by segment, so every result is the one full_gotoh_score_kernel produces.
// ---------------------------------------------------------------------------------------------
// MODE: 0 = no min_score, 1 = min_score with text blocking (the sweep that watches the column maxima, CHECK), 2 = min_score with pattern
// blocking (the sweep that also keeps every row's maximum, PBX).  One kernel per job shape because a kernel's register allocation is the
// maximum over the sweeps compiled into it: with all three inside, every instance carried the CHECK sweep's budget (SEMI_GLOBAL at ten rows
// per lane: 256 VGPRs, two waves per SIMD); apart, the same instance takes 84 (no min_score: five waves) / 105 (PBX: four) and only the
// CHECK sweep keeps its 250 (capped at 168 it spills 428 bytes).  LOCAL at ten rows: 109 / 168 / 168.  profiles/r04/full_dp_rows.txt.
template <int TYPE, int R, int MODE>
__global__ void __launch_bounds__(256, ((R >= 8 && MODE == 1 && TYPE != NVBIO_HIP_LOCAL) ? 2 : 3))       // R <= 6: >= 3 waves per SIMD (left alone the SEMI_GLOBAL instance takes 176 VGPRs = 2 waves; 168 + 14 spilled: +8 %); deeper lanes hold more rows and get 256 VGPRs
full_gotoh_score_multi_kernel(const FullParams p, const uint32_t n_seg, const uint32_t seg_w)
{
    // LOCAL with eight or more rows per lane: the rows' records in LDS (Sweep16: BKL), which brings the instance from 176 to <= 168 VGPRs
    constexpr bool BKL = (TYPE == NVBIO_HIP_LOCAL && R >= 8);
    __shared__ uint32_t bk_sh[BKL ? 4 : 1][BKL ? R * 64 : 1];
    uint32_t* const bkl = BKL ? &bk_sh[threadIdx.x >> 6][0] : nullptr;
    const uint32_t wave = (blockIdx.x * 256u + threadIdx.x) >> 6;
    const uint32_t wl   = threadIdx.x & 63u;
    const uint32_t seg  = wl / seg_w, sl = wl - seg * seg_w;
    const uint32_t job  = wave * n_seg + seg;
    const bool has_job  = seg < n_seg && job < p.n;
    if (wave * n_seg >= p.n || gate_closed(p)) return;

    const uint32_t M  = has_job ? (p.pat.length ? p.pat.length[job] : p.pat.fixed_length) : 0u;
    const uint32_t N  = has_job ? (p.txt.length ? p.txt.length[job] : p.txt.fixed_length) : 0u;
    const uint64_t pb = has_job ? p.pat.begin[job] : 0ull, tb = has_job ? p.txt.begin[job] : 0ull;
    const bool     check = p.min_score != nullptr;
    const int32_t  min_score = (check && has_job) ? p.min_score[job] : -(1 << 30);
    const bool     PB = p.pattern_blocking != 0u;

    int32_t score = -(1 << 30); uint32_t sx = 0xFFFFFFFFu, sy = 0xFFFFFFFFu; uint32_t ok = 1u;
    // jobs the sweep does not take (see full_gotoh_score_kernel for each case): they sit out with an empty matrix
    bool swept = has_job;
    if (has_job)
    {
        if (M > p.max_m || N > p.max_n) { ok = 0u; swept = false; }
        else if (M == 0u)
        {
            swept = false;
            const uint32_t xb = check ? empty_pattern_exit_block(N, 8u, p.match, min_score) : 0xFFFFFFFFu;
            const bool exits = xb != 0xFFFFFFFFu;
            if (exits) { ok = 0u; if (TYPE == NVBIO_HIP_SEMI_GLOBAL) { score = 0; sx = xb + 8u; sy = 0u; } }
            else if (N > 0u && TYPE == NVBIO_HIP_SEMI_GLOBAL) { score = 0; sx = N; sy = 0u; }
            else if (N > 0u && TYPE == NVBIO_HIP_GLOBAL)      { score = p.row_go + p.row_ge * int32_t(N - 1u); sx = N; sy = 0u; }
        }
        else if (PB && check && N == 0u)
        {
            swept = false;
            const uint32_t BLK = 1u << p.blk_log2;
            if ((M + BLK - 1u) / BLK > 1u) ok = 0u;
            else if (TYPE == NVBIO_HIP_GLOBAL) { score = p.col_go + p.col_ge * int32_t(M - 1u); sx = 0u; sy = M; }
        }
    }
    const bool sweeps = swept;
    const uint32_t Ms = sweeps ? M : 1u, Ns = sweeps ? N : 0u;
    // 0 = nothing more to do, 1 = sweep rows [0, arg] again (pattern blocking, LOCAL), 2 = sweep columns [0, arg] again (text blocking, LOCAL)
    uint32_t redo = 0u, redo_arg = 0u;
    if constexpr (MODE == 2)
    {
        Sweep16<TYPE, R, false, (TYPE != NVBIO_HIP_LOCAL), true, BKL> sw(p);
        sw.bkl = bkl;
        sw.init(pb, tb, Ms, Ns, Ns, min_score, seg_w, sweeps);
        sw.pb_check = true;
        const SweepResult r = sw.run();
        if (sweeps)
        {
            if (r.pb_exit_row != 0xFFFFFFFFu) { ok = 0u; if (TYPE == NVBIO_HIP_LOCAL) { redo = 1u; redo_arg = r.pb_exit_row + 1u; } }
            else { score = r.score; sx = r.sx; sy = r.sy; }
        }
    }
    else
    {
        SweepResult r;
        if constexpr (MODE == 1) { Sweep16<TYPE, R, true,  false, true, BKL> sw(p); sw.bkl = bkl; sw.init(pb, tb, Ms, Ns, Ns, min_score, seg_w, sweeps); r = sw.run(); }
        else                     { Sweep16<TYPE, R, false, false, true, BKL> sw(p); sw.bkl = bkl; sw.init(pb, tb, Ms, Ns, Ns, min_score, seg_w, sweeps); r = sw.run(); }
        if (sweeps)
        {
            if (r.exit_col != 0xFFFFFFFFu)
            {
                ok = 0u;
                // LOCAL: the best cell of the whole matrix, if it lies at or before the exit column, is also the best of the columns the
                // reference visited; only otherwise sweep the truncated text again.  SEMI_GLOBAL: the record froze at the exit column.
                if (TYPE == NVBIO_HIP_LOCAL && !(r.sx != 0xFFFFFFFFu && r.sx - 1u <= r.exit_col)) { redo = 2u; redo_arg = r.exit_col + 1u; }
                else if (TYPE != NVBIO_HIP_GLOBAL) { score = r.score; sx = r.sx; sy = r.sy; }      // (GLOBAL: the only report is at the last column)
            }
            else { score = r.score; sx = r.sx; sy = r.sy; }
            // pattern blocking, GLOBAL, empty text: save_Mth reports the initial row (gotoh_inl.h:896-897)
            if (PB && TYPE == NVBIO_HIP_GLOBAL && N == 0u) { score = p.col_go + p.col_ge * int32_t(M - 1u); sx = 0u; sy = M; }
        }
    }
    // second sweeps, one job at a time on the whole wave (rare: a job whose early exit fired with its best cell beyond the exit)
    if (TYPE == NVBIO_HIP_LOCAL)
    {
        for (uint32_t g = 0; g < n_seg; ++g)
        {
            const int src = int(g * seg_w);                                   // the segment's first lane holds its job's values
            const uint32_t rd = uint32_t(__shfl(int32_t(redo), src));
            if (rd == 0u) continue;                                            // wave-uniform
            const uint32_t arg = uint32_t(__shfl(int32_t(redo_arg), src));
            const uint32_t gM = uint32_t(__shfl(int32_t(M), src)), gN = uint32_t(__shfl(int32_t(N), src));
            const uint64_t gpb = (uint64_t(uint32_t(__shfl(int32_t(uint32_t(pb >> 32)), src))) << 32) | uint32_t(__shfl(int32_t(uint32_t(pb)), src));
            const uint64_t gtb = (uint64_t(uint32_t(__shfl(int32_t(uint32_t(tb >> 32)), src))) << 32) | uint32_t(__shfl(int32_t(uint32_t(tb)), src));
            const int32_t  gms = __shfl(min_score, src);
            const SweepResult r2 = (rd == 1u) ? sweep16<TYPE, R, false, BKL>(p, gpb, gtb, arg, gN, gN, gms, bkl) : sweep16<TYPE, R, false, BKL>(p, gpb, gtb, gM, arg, gN, gms, bkl);
            if (seg == g) { score = r2.score; sx = r2.sx; sy = r2.sy; }
        }
    }
    if (has_job && sl == 0u)
    {
        p.out_score[job] = score;
        reinterpret_cast<uint2*>(p.out_sink)[job] = make_uint2(sx, sy);
        if (p.out_ok) p.out_ok[job] = uint8_t(ok);
    }
}

template <int R>
static hipError_t launch_full_multi(const FullParams& p, int type, uint32_t n_seg, hipStream_t s)
{
    const uint32_t seg_w = 64u / n_seg;
    const uint64_t waves = (uint64_t(p.n) + n_seg - 1u) / n_seg;
    const dim3 grid(uint32_t((waves * 64u + 255u) / 256u)), block(256);
    // (see the kernel: one instance per job shape, because their register budgets differ)
    const int mode = p.min_score == nullptr ? 0 : (p.pattern_blocking != 0u ? 2 : 1);
    #define NVB_LAUNCH_MULTI(T) do { \
        if (mode == 2)      hipLaunchKernelGGL((full_gotoh_score_multi_kernel<T, R, 2>), grid, block, 0, s, p, n_seg, seg_w); \
        else if (mode == 1) hipLaunchKernelGGL((full_gotoh_score_multi_kernel<T, R, 1>), grid, block, 0, s, p, n_seg, seg_w); \
        else                hipLaunchKernelGGL((full_gotoh_score_multi_kernel<T, R, 0>), grid, block, 0, s, p, n_seg, seg_w); } while (0)
    switch (type) {
    case NVBIO_HIP_LOCAL:       NVB_LAUNCH_MULTI(NVBIO_HIP_LOCAL); break;
    case NVBIO_HIP_SEMI_GLOBAL: NVB_LAUNCH_MULTI(NVBIO_HIP_SEMI_GLOBAL); break;
    case NVBIO_HIP_GLOBAL:      NVB_LAUNCH_MULTI(NVBIO_HIP_GLOBAL); break;
    default: return hipErrorInvalidValue;
    }
    #undef NVB_LAUNCH_MULTI
    return hipGetLastError();
}

template <int R, bool TRUNC, bool FAST>
static hipError_t launch_full(const FullParams& p, int type, hipStream_t s)
{
    const dim3 grid((uint64_t(p.n) * 64u + 255u) / 256u), block(256);
    // (the 16-bit sweep: one kernel per job shape, as for the multi-job kernel; the 32-bit sweep decides at run time)
    const int mode = !FAST ? 0 : (p.min_score == nullptr ? 0 : (p.pattern_blocking != 0u ? 2 : 1));
    #define NVB_LAUNCH_FULL(T) do { \
        if constexpr (FAST) { \
            if (mode == 2)      hipLaunchKernelGGL((full_gotoh_score_kernel<T, R, TRUNC, FAST, 2>), grid, block, 0, s, p); \
            else if (mode == 1) hipLaunchKernelGGL((full_gotoh_score_kernel<T, R, TRUNC, FAST, 1>), grid, block, 0, s, p); \
            else                hipLaunchKernelGGL((full_gotoh_score_kernel<T, R, TRUNC, FAST, 0>), grid, block, 0, s, p); \
        } else                  hipLaunchKernelGGL((full_gotoh_score_kernel<T, R, TRUNC, FAST, 0>), grid, block, 0, s, p); } while (0)
    switch (type) {
    case NVBIO_HIP_GLOBAL:      NVB_LAUNCH_FULL(NVBIO_HIP_GLOBAL); break;
    case NVBIO_HIP_LOCAL:       NVB_LAUNCH_FULL(NVBIO_HIP_LOCAL); break;
    case NVBIO_HIP_SEMI_GLOBAL: NVB_LAUNCH_FULL(NVBIO_HIP_SEMI_GLOBAL); break;
    default: return hipErrorInvalidValue;
    }
    #undef NVB_LAUNCH_FULL
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Edit distance (EditDistanceAligner: match 0, mismatch -1, deletion = insertion = -1; sw-benchmark's second leg,
// sw-benchmark.cu:641-657) without the matrix: with unit costs adjacent cells differ by -1, 0 or +1, so a column is two bit-vectors
// of vertical differences and advancing it by one text symbol is a dozen word operations per 64 pattern rows (Myers 1999, in
// the block form of Hyyro 2003: a carry h_in / h_out of -1, 0, +1 between the 64-row words).  What the reference reports is a
// function of the last row only -- SEMI_GLOBAL: every H(i, M) in text order into a BestSink (the last best column wins), GLOBAL:
// H(N, M) -- which the running value of the last row gives exactly; with no min_score the reference never exits early, and its
// int16 boundary column is exact in the range the host admits here.  One lane per job; 150 x 16 384: ~100 word operations per
// column instead of 150 cells.  LOCAL, min_score and other costs stay on the sweep.
// ---------------------------------------------------------------------------------------------
template <int TYPE, int W>       // W 64-row words: patterns to 64 * W
__global__ void __launch_bounds__(256) edit_distance_bitvector_kernel(const FullParams p)
{
    const uint32_t job = blockIdx.x * 256u + threadIdx.x;
    if (job >= p.n) return;
    const uint32_t M  = p.pat.length ? p.pat.length[job] : p.pat.fixed_length;
    const uint32_t N  = p.txt.length ? p.txt.length[job] : p.txt.fixed_length;
    const uint64_t pb = p.pat.begin[job], tb = p.txt.begin[job];
    int32_t score = -(1 << 30); uint32_t sx = 0xFFFFFFFFu, sy = 0xFFFFFFFFu; uint32_t ok = 1u;
    if (M > p.max_m || N > p.max_n) ok = 0u;
    else if (M == 0u)
    {
        if (N > 0u) {                                                    // as full_gotoh_score_kernel: only the row above the matrix is reported
            if (TYPE == NVBIO_HIP_SEMI_GLOBAL) { score = 0; sx = N; sy = 0u; }
            if (TYPE == NVBIO_HIP_GLOBAL)      { score = p.row_go + p.row_ge * int32_t(N - 1u); sx = N; sy = 0u; }
        }
    }
    else if (N == 0u)
    {
        if (p.pattern_blocking != 0u && TYPE == NVBIO_HIP_GLOBAL) { score = p.col_go + p.col_ge * int32_t(M - 1u); sx = 0u; sy = M; }
    }
    else
    {
        uint64_t Peq[4][W];
        #pragma unroll
        for (int w = 0; w < W; ++w) { Peq[0][w] = Peq[1][w] = Peq[2][w] = Peq[3][w] = 0ull; }
        #pragma unroll
        for (int w = 0; w < W; ++w)
        {
            for (uint32_t g0 = 0; g0 < 64u && uint32_t(w) * 64u + g0 < M; g0 += 16u)
            {
                const uint32_t r0 = uint32_t(w) * 64u + g0;
                const uint64_t grp = (p.pat.s.bits == 4) ? fetch16_4bit(p.pat.s, pb + r0) : expand_2to4(fetch16_2bit(p.pat.s, pb + r0));
                const uint32_t cnt = min(16u, M - r0);
                for (uint32_t k = 0; k < cnt; ++k)
                {
                    const uint32_t c = uint32_t(grp >> (4u * k)) & 15u;
                    const uint64_t bit = 1ull << (g0 + k);
                    Peq[0][w] |= (c == 0u) ? bit : 0ull; Peq[1][w] |= (c == 1u) ? bit : 0ull;
                    Peq[2][w] |= (c == 2u) ? bit : 0ull; Peq[3][w] |= (c == 3u) ? bit : 0ull;      // N codes match nothing
                }
            }
        }
        uint64_t Pv[W], Mv[W];
        #pragma unroll
        for (int w = 0; w < W; ++w) { Pv[w] = ~0ull; Mv[w] = 0ull; }
        const uint32_t lw = (M - 1u) >> 6, top = (M - 1u) & 63u;
        int32_t  d = int32_t(M);                                          // D(M, 0)
        int32_t  best_d = 0x7FFFFFFF; uint32_t best_i = 0u;
        for (uint32_t i0 = 0; i0 < N; i0 += 16u)
        {
            const uint32_t tg = fetch16_2bit(p.txt.s, tb + i0);
            const uint32_t cnt = min(16u, N - i0);
            for (uint32_t u = 0; u < cnt; ++u)
            {
                const uint32_t c = (tg >> (2u * u)) & 3u;
                int32_t hin = (TYPE == NVBIO_HIP_GLOBAL) ? 1 : 0;          // the row above the matrix: D(0, i) = i, or 0 with a free text start
                int32_t dd = 0;
                #pragma unroll
                for (int w = 0; w < W; ++w)
                {
                    uint64_t Eq = (c == 0u) ? Peq[0][w] : (c == 1u) ? Peq[1][w] : (c == 2u) ? Peq[2][w] : Peq[3][w];
                    const uint64_t Xv = Eq | Mv[w];
                    if (hin < 0) Eq |= 1ull;
                    const uint64_t Xh = (((Eq & Pv[w]) + Pv[w]) ^ Pv[w]) | Eq;
                    uint64_t Ph = Mv[w] | ~(Xh | Pv[w]);
                    uint64_t Mh = Pv[w] & Xh;
                    if (uint32_t(w) == lw) dd = int32_t((Ph >> top) & 1ull) - int32_t((Mh >> top) & 1ull);
                    const int32_t hout = int32_t(Ph >> 63) - int32_t(Mh >> 63);
                    Ph <<= 1; Mh <<= 1;
                    if (hin < 0) Mh |= 1ull; else if (hin > 0) Ph |= 1ull;
                    Pv[w] = Mh | ~(Xv | Ph);
                    Mv[w] = Ph & Xv;
                    hin = hout;
                }
                d += dd;
                if (TYPE == NVBIO_HIP_SEMI_GLOBAL) { if (d <= best_d) { best_d = d; best_i = i0 + u + 1u; } }
            }
        }
        if (TYPE == NVBIO_HIP_SEMI_GLOBAL) { score = -best_d; sx = best_i; sy = M; }
        else                               { score = -d;      sx = N;      sy = M; }
    }
    p.out_score[job] = score;
    reinterpret_cast<uint2*>(p.out_sink)[job] = make_uint2(sx, sy);
    if (p.out_ok) p.out_ok[job] = uint8_t(ok);
}

template <int W>
static hipError_t launch_ed(const FullParams& p, int type, hipStream_t s)
{
    const dim3 grid((p.n + 255u) / 256u), block(256);
    if (type == NVBIO_HIP_GLOBAL) hipLaunchKernelGGL((edit_distance_bitvector_kernel<NVBIO_HIP_GLOBAL, W>), grid, block, 0, s, p);
    else                          hipLaunchKernelGGL((edit_distance_bitvector_kernel<NVBIO_HIP_SEMI_GLOBAL, W>), grid, block, 0, s, p);
    return hipGetLastError();
}

} // namespace nvb

using namespace nvb;

// SimpleSmithWatermanScheme with deletion != insertion, full matrix (sw_inl.h:881-1222 text blocking, :417-760 pattern blocking): the move
// along the text costs `deletion`, the move down the pattern `insertion`; the column before the text is insertion * (r + 1), the row
// above the pattern deletion * (c + 1).  On the striped sweep, whatever the pattern length.
static int sw_asym_score(const nvbio_hip_sw_scheme* sw, int32_t type, uint32_t pattern_blocking,
                         const nvbio_hip_string_set* patterns, const nvbio_hip_string_set* texts, uint32_t max_pattern_len, uint32_t max_text_len,
                         uint32_t n, int32_t* out_score, uint32_t* out_sink, uint8_t* out_ok, void* stream)
{
    if (!patterns || !texts) return hipErrorInvalidValue;
    if (type < 0 || type > 2) return hipErrorInvalidValue;
    if (!(patterns->bits == 2 || patterns->bits == 4 || patterns->bits == 8) || texts->bits != 2) return hipErrorNotSupported;
    if (n == 0) return hipSuccess;
    if (!out_score || !out_sink || !patterns->words || !texts->words || !patterns->begin || !texts->begin ||
        patterns->n_words == 0 || texts->n_words == 0) return hipErrorInvalidValue;
    const uint32_t maxM = patterns->length ? max_pattern_len : patterns->fixed_length;
    const uint32_t maxN = texts->length ? max_text_len : texts->fixed_length;
    if (maxM == 0 || maxN == 0) return hipErrorNotSupported;
    auto iabs = [](int32_t v) { return v < 0 ? -int64_t(v) : int64_t(v); };
    const int64_t A = std::max(std::max(iabs(sw->match), iabs(sw->mismatch)), std::max(iabs(sw->deletion), iabs(sw->insertion)));
    const int64_t span = (type == NVBIO_HIP_GLOBAL) ? int64_t(maxM) + maxN + 4 : int64_t(maxM) + 4;
    const bool inside16 = sw->deletion <= 0 && sw->insertion <= 0 && span * A < 30000;
    // pattern blocking keeps its int16 line over the text, which the sweep does not model: admitted while no value can leave int16
    if (pattern_blocking && !inside16) return hipErrorNotSupported;
    StripeParams sp;
    sp.pat = make_string_set(patterns); sp.txt = make_string_set(texts);
    sp.match = sw->match; sp.mismatch = sw->mismatch;
    sp.e_go = sp.e_ge = sw->deletion; sp.f_go = sp.f_ge = sw->insertion;
    sp.col_go = sp.col_ge = sw->insertion; sp.row_go = sp.row_ge = sw->deletion;
    sp.infimum = -32768 - std::min(std::min(sw->deletion, sw->insertion), 0);
    sp.linear = 1u; sp.trunc = pattern_blocking ? 0u : 1u; sp.blk_log2 = 4u; sp.pattern_blocking = pattern_blocking;
    sp.min_score = nullptr; sp.n = n; sp.out_score = out_score; sp.out_sink = out_sink; sp.out_ok = out_ok; sp.max_n = maxN;
    g_last_kernel = "full_gotoh_striped_kernel<sw,asymmetric>";
    return launch_striped(sp, type, maxM, to_stream(stream));
}

struct QualPart { const uint8_t* quals; uint64_t n_quals; const int32_t* mismatch; int32_t text_gap_open, text_gap_ext; };

struct FullJobs { const uint32_t* n_dev; const uint32_t* job_index; const uint32_t* gate; uint32_t gate_limit; int wave_form; };

static int full_score_core(
    const nvbio_hip_gotoh_scheme* scheme, const QualPart* qual, int32_t type, uint32_t blk_log2, uint32_t pattern_blocking,
    const nvbio_hip_string_set* patterns, const nvbio_hip_string_set* texts,
    uint32_t max_pattern_len, uint32_t max_text_len, const int32_t* min_score,
    uint32_t n, int32_t* out_score, uint32_t* out_sink, uint8_t* out_ok, void* stream, const FullJobs* jobs = nullptr)
{
    if (!scheme || !patterns || !texts) return hipErrorInvalidValue;
    if (type < 0 || type > 2) return hipErrorInvalidValue;
    if (!(patterns->bits == 2 || patterns->bits == 4 || patterns->bits == 8) || texts->bits != 2) return hipErrorNotSupported;
    if (n == 0) return hipSuccess;
    if (!out_score || !out_sink || !patterns->words || !texts->words || !patterns->begin || !texts->begin ||
        patterns->n_words == 0 || texts->n_words == 0) return hipErrorInvalidValue;
    // the systolic mapping holds the whole pattern in one wave's registers: the caller states its bound
    const uint32_t maxM = patterns->length ? max_pattern_len : patterns->fixed_length;
    const uint32_t maxN = texts->length ? max_text_len : texts->fixed_length;
    if (maxM == 0 || maxN == 0) return hipErrorNotSupported;
    const bool striped = maxM > 1024u;                                  // beyond 64 lanes x 16 rows: stripes of 1 024 rows (full_gotoh_striped.hip)
    if (!striped && uint64_t(maxN) * 64u * (maxM <= 512u ? 8u : 16u) >= (1ull << 32)) return hipErrorNotSupported;    // LOCAL order keys are 32-bit

    FullParams p;
    p.pat = make_string_set(patterns); p.txt = make_string_set(texts);
    p.match = scheme->match; p.mismatch = scheme->mismatch; p.gap_open = scheme->gap_open; p.gap_ext = scheme->gap_ext;
    p.min_score = min_score; p.n = n; p.out_score = out_score; p.out_sink = out_sink; p.out_ok = out_ok;
    p.n_dev = nullptr; p.job_index = nullptr; p.gate = nullptr; p.gate_limit = 0u; p.gate_above = 1u;
    if (jobs)
    {
        if (jobs->wave_form) { p.n_dev = jobs->n_dev; p.job_index = jobs->job_index; }       // (the throughput dispatch walks the arrays as they are)
        p.gate = jobs->gate; p.gate_limit = jobs->gate_limit; p.gate_above = jobs->wave_form ? 0u : 1u;
    }
    p.blk_log2 = blk_log2; p.pattern_blocking = pattern_blocking;
    p.max_m = maxM; p.max_n = maxN;
    p.quals = qual ? qual->quals : nullptr; p.n_quals = qual ? qual->n_quals : 0;
    for (int i = 0; i < 256; ++i) p.mm_lut[i] = qual ? qual->mismatch[i] : scheme->mismatch;
    // which boundary line is initialised with which gap costs depends on the tag (gotoh_inl.h:82-88 vs :693-697 / :1171-1175)
    const int32_t tgo = qual ? qual->text_gap_open : scheme->gap_open, tge = qual ? qual->text_gap_ext : scheme->gap_ext;
    p.col_go = pattern_blocking ? scheme->gap_open : tgo; p.col_ge = pattern_blocking ? scheme->gap_ext : tge;
    p.row_go = pattern_blocking ? tgo : scheme->gap_open; p.row_ge = pattern_blocking ? tge : scheme->gap_ext;

    // can any H / E leave int16?  LOCAL: 0 <= H <= M*match, E/F a few gap costs below.  SEMI_GLOBAL (pattern
    // global, text free): every cell is reachable from the zero row above its column, so values stay within
    // (M+2) * max|cost| whatever the text length.  GLOBAL: the row above the matrix itself reaches G_o + G_e*N.
    auto iabs = [](int32_t v) { return v < 0 ? -int64_t(v) : int64_t(v); };
    int64_t A = std::max(std::max(iabs(scheme->match), iabs(scheme->mismatch)), std::max(iabs(scheme->gap_open), iabs(scheme->gap_ext)));
    if (qual) { for (int i = 0; i < 256; ++i) A = std::max(A, iabs(qual->mismatch[i])); A = std::max(A, std::max(iabs(qual->text_gap_open), iabs(qual->text_gap_ext))); }
    const int64_t span = (type == NVBIO_HIP_GLOBAL) ? int64_t(maxM) + maxN + 4 : int64_t(maxM) + 4;
    // the 16-bit sweep's sentinel and range reasoning assume that no gap move earns score, on either string
    const bool gaps_cost = scheme->gap_open <= 0 && scheme->gap_ext <= 0 && (!qual || (qual->text_gap_open <= 0 && qual->text_gap_ext <= 0));
    bool trunc = !(gaps_cost && span * A < 30000);
    if (trunc && gaps_cost && type == NVBIO_HIP_GLOBAL)
    {
        // GLOBAL, tighter: (M + N) * max|cost| overestimates badly when the expensive costs are not the gap extensions.  What the sweep
        // holds: the two boundary lines themselves, H(r,-1) = col_go + col_ge*r and H(-1,c) = row_go + row_ge*c; interior H, each at
        // least the value of reaching it by one gap run from either boundary line (H is a maximum over paths, those paths included) and
        // at most M times the best pair score; E, F, H + G_o and the diagonal sum within one gap open + one extension + one substitution
        // of an H.  All inside int16 => the 16-bit sweep and the reference's int16 boundary column are both exact.
        const int64_t row_line = iabs(p.row_go) + iabs(p.row_ge) * int64_t(maxN), col_line = iabs(p.col_go) + iabs(p.col_ge) * int64_t(maxM);
        const int64_t via_top  = row_line + iabs(scheme->gap_open) + iabs(scheme->gap_ext) * int64_t(maxM);
        const int64_t via_left = col_line + iabs(scheme->gap_open) + iabs(scheme->gap_ext) * int64_t(maxN);
        int64_t worst_sub = std::max(iabs(scheme->match), iabs(scheme->mismatch));
        if (qual) for (int i = 0; i < 256; ++i) worst_sub = std::max(worst_sub, iabs(qual->mismatch[i]));
        const int64_t low  = std::max(std::max(row_line, col_line), std::min(via_top, via_left)) + iabs(scheme->gap_open) + iabs(scheme->gap_ext) + worst_sub + 8;
        const int64_t high = int64_t(maxM) * std::max<int64_t>(0, std::max(scheme->match, scheme->mismatch)) + worst_sub + 8;
        if (low < 32000 && high < 32000) trunc = false;
    }
    if (striped)
    {
        // what the striped sweep does not model: qualities, job lists, and -- for pattern blocking -- thresholds and values beyond int16
        if (qual || jobs) return hipErrorNotSupported;
        if (pattern_blocking && (min_score != nullptr || trunc)) return hipErrorNotSupported;
        StripeParams sp;
        sp.pat = p.pat; sp.txt = p.txt; sp.match = p.match; sp.mismatch = p.mismatch;
        sp.e_go = sp.f_go = p.gap_open; sp.e_ge = sp.f_ge = p.gap_ext;
        sp.col_go = p.col_go; sp.col_ge = p.col_ge; sp.row_go = p.row_go; sp.row_ge = p.row_ge;
        sp.infimum = -32768 - std::min(p.gap_open, p.gap_ext);
        sp.linear = (blk_log2 == 4u) ? 1u : 0u; sp.trunc = trunc ? 1u : 0u; sp.blk_log2 = blk_log2; sp.pattern_blocking = pattern_blocking;
        sp.min_score = min_score; sp.n = n; sp.out_score = out_score; sp.out_sink = out_sink; sp.out_ok = out_ok; sp.max_n = maxN;
        g_last_kernel = "full_gotoh_striped_kernel";
        return launch_striped(sp, type, maxM, to_stream(stream));
    }
    // the largest score one aligned pair can add: LOCAL's H is bounded by M times it, not by M * match
    int32_t best_pair = std::max(scheme->match, scheme->mismatch);
    if (qual) for (int i = 0; i < 256; ++i) best_pair = std::max(best_pair, qual->mismatch[i]);
    hipStream_t s = to_stream(stream);
    g_last_kernel = "full_gotoh_score_kernel";
    const int R = maxM <= 64u ? 1 : maxM <= 128u ? 2 : maxM <= 192u ? 3 : maxM <= 256u ? 4 : maxM <= 512u ? 8 : 16;
    // the 16-bit sweep needs: values inside int16 (= !trunc), LOCAL scores < 2048 and columns < 2^20 for its packed row maxima
    // (LOCAL runs x16: scores below 2048 and every cost below 2048 / 3 keep H, E, F, H + G_o and the diagonal sum inside int16)
    const bool fast = !trunc && maxN < (1u << 20) && (type != NVBIO_HIP_LOCAL || (scheme->match >= 0 && int64_t(maxM) * best_pair < 2048 && A * 3 < 2000))
                      && test_switch(SW_FULL_GENERIC) != 1;
    // edit distance, no min_score, non-LOCAL: the bit-vector kernel (NVBIO_HIP_ED_SWEEP=1 keeps the sweep, for the tests)
    {
        const bool ed = !qual && blk_log2 == 4u && scheme->match == 0 && scheme->mismatch == -1 && scheme->gap_open == -1 && scheme->gap_ext == -1;
        if (ed && !trunc && type != NVBIO_HIP_LOCAL && min_score == nullptr && maxM <= 512u && test_switch(SW_ED_SWEEP) != 1 && !jobs && patterns->bits != 8u)
        {
            g_last_kernel = "edit_distance_bitvector_kernel";
            const uint32_t words = (maxM + 63u) / 64u;
            if (words <= 1u) return launch_ed<1>(p, type, s);
            if (words <= 2u) return launch_ed<2>(p, type, s);
            if (words <= 3u) return launch_ed<3>(p, type, s);
            if (words <= 4u) return launch_ed<4>(p, type, s);
            return launch_ed<8>(p, type, s);
        }
    }
    if (fast)
    {
        // several jobs per wave when that keeps more lanes busy: n_seg segments of 64 / n_seg lanes, R = 5 or 6 rows per lane.
        // Estimated cell throughput: busy lanes x (cell work) / (cell work + per-step overhead).
        const bool nomulti = test_switch(SW_FULL_SINGLE_JOB) == 1 || (jobs && jobs->wave_form);
        auto eff = [](const double lanes, const double rows, const double overhead) { return lanes / 64.0 * (28.0 * rows) / (28.0 * rows + overhead); };
        double best = eff(double((maxM + R - 1) / R), double(R), 24.0);
        uint32_t best_seg = 1u, best_r = 0u;
        // (rows per lane: 5 / 6 with two or three jobs per wave; 8 / 10 with four -- 150-bp mates: 4 x 15 lanes x 10 rows -- which halves the
        // per-step work that does not depend on the rows: the hand-off between lanes, the step's bookkeeping, the ramp in and out of a matrix)
        const uint32_t only_r = uint32_t(std::max(0, test_switch(SW_FULL_ROWS)));     // test switch: 5, 6, 8, 10 = only that depth
        const uint32_t depths[4] = { 5u, 6u, 8u, 10u };
        for (uint32_t ns = 2u; ns <= 4u; ++ns)
            for (uint32_t di = 0; di < 4u; ++di)
            {
                const uint32_t r = depths[di];
                if (only_r && r != only_r) continue;
                const uint32_t usable = 64u / ns;
                if (uint64_t(usable) * r < maxM) continue;
                double e = eff(double(ns * ((maxM + r - 1u) / r)), double(r), 24.0);
                // measured at 150 x 16 384 (profiles/r04/full_dp_rows.txt): ten rows per lane beat five by 7 % (SEMI_GLOBAL), 12 % (GLOBAL) and -- with
                // LOCAL's per-row records moved to LDS, which brings its instance from 176 to 168 VGPRs = three waves per SIMD -- 4 % (LOCAL); eight
                // rows (three jobs of 19 lanes) lose to five for every type
                if (type == NVBIO_HIP_LOCAL && r == 8u) e *= 0.90;
                if (e > best * 1.05) { best = e; best_seg = ns; best_r = r; }
            }
        if (best_seg > 1u && !nomulti && uint64_t(maxN) * 64u * 8u < (1ull << 32))
        {
            g_last_kernel = "full_gotoh_score_multi_kernel<16-bit>";
            switch (best_r) { case 5u: return launch_full_multi<5>(p, type, best_seg, s); case 6u: return launch_full_multi<6>(p, type, best_seg, s);
                              case 8u: return launch_full_multi<8>(p, type, best_seg, s); default: return launch_full_multi<10>(p, type, best_seg, s); }
        }
    }
    if (jobs && jobs->wave_form && !fast) return hipErrorNotSupported;         // the job list is a feature of the 16-bit one-job-per-wave sweep
    if (fast) {
        g_last_kernel = "full_gotoh_score_kernel<16-bit>";
        switch (R) { case 1: return launch_full<1, false, true>(p, type, s); case 2: return launch_full<2, false, true>(p, type, s);
                     case 3: return launch_full<3, false, true>(p, type, s); case 4: return launch_full<4, false, true>(p, type, s);
                     case 8: return launch_full<8, false, true>(p, type, s);
                     default: return launch_full<16, false, true>(p, type, s); }
    }
    if (trunc && blk_log2 != 3u) return hipErrorNotSupported;     // the int16 boundary column of the SW form is not modelled beyond its exact range
    if (pattern_blocking && !fast) return hipErrorNotSupported;   // pattern blocking is implemented on the 16-bit sweep only
    if (qual && !fast) return hipErrorNotSupported;               // so is the quality-aware scheme
    if (trunc) {
        switch (R) { case 1: return launch_full<1, true, false>(p, type, s); case 2: return launch_full<2, true, false>(p, type, s);
                     case 3: return launch_full<3, true, false>(p, type, s); case 4: return launch_full<4, true, false>(p, type, s);
                     case 8: return launch_full<8, true, false>(p, type, s);
                     default: return launch_full<16, true, false>(p, type, s); }
    } else {
        switch (R) { case 1: return launch_full<1, false, false>(p, type, s); case 2: return launch_full<2, false, false>(p, type, s);
                     case 3: return launch_full<3, false, false>(p, type, s); case 4: return launch_full<4, false, false>(p, type, s);
                     case 8: return launch_full<8, false, false>(p, type, s);
                     default: return launch_full<16, false, false>(p, type, s); }
    }
}

NVB_API int nvbio_hip_gotoh_score(
    const nvbio_hip_gotoh_scheme* scheme, int32_t type,
    const nvbio_hip_string_set* patterns, const nvbio_hip_string_set* texts,
    uint32_t max_pattern_len, uint32_t max_text_len, const int32_t* min_score,
    uint32_t n, int32_t* out_score, uint32_t* out_sink, uint8_t* out_ok, void* stream)
{
    return full_score_core(scheme, nullptr, type, 3u, 0u, patterns, texts, max_pattern_len, max_text_len, min_score, n, out_score, out_sink, out_ok, stream);
}

// SmithWatermanAligner / EditDistanceAligner, full matrix, text blocking (sw_inl.h:881-1222): linear gaps with
// deletion == insertion are Gotoh with gap_open == gap_ext cell for cell (H >= E, F always), the boundary column
// is exact while values fit int16, this variant never exits early, and its blocks are 16 columns wide.
NVB_API int nvbio_hip_sw_score(
    const nvbio_hip_sw_scheme* scheme, int32_t type,
    const nvbio_hip_string_set* patterns, const nvbio_hip_string_set* texts,
    uint32_t max_pattern_len, uint32_t max_text_len,
    uint32_t n, int32_t* out_score, uint32_t* out_sink, void* stream)
{
    if (!scheme) return hipErrorInvalidValue;
    if (scheme->deletion != scheme->insertion)
        return sw_asym_score(scheme, type, 0u, patterns, texts, max_pattern_len, max_text_len, n, out_score, out_sink, nullptr, stream);
    const nvbio_hip_gotoh_scheme g = { scheme->match, scheme->mismatch, scheme->deletion, scheme->deletion };
    const int e = full_score_core(&g, nullptr, type, 4u, 0u, patterns, texts, max_pattern_len, max_text_len, nullptr, n, out_score, out_sink, nullptr, stream);
    if (e == hipSuccess && n && g_last_kernel[0] != 'e' && !strstr(g_last_kernel, "striped")) g_last_kernel = "full_gotoh_score_kernel<16-bit,sw>";       // ('e': the edit-distance kernel ran)
    return e;
}

// The general full-matrix entry: aligner kind x algorithm tag.
NVB_API int nvbio_hip_alignment_score(
    int32_t aligner /* 0 Gotoh, 1 Smith-Waterman / edit distance */, int32_t algorithm /* 0 PatternBlockingTag, 1 TextBlockingTag */,
    const int32_t* scheme4, int32_t type,
    const nvbio_hip_string_set* patterns, const nvbio_hip_string_set* texts,
    uint32_t max_pattern_len, uint32_t max_text_len, const int32_t* min_score,
    uint32_t n, int32_t* out_score, uint32_t* out_sink, uint8_t* out_ok, void* stream)
{
    if (!scheme4 || aligner < 0 || aligner > 1 || algorithm < 0 || algorithm > 1) return hipErrorInvalidValue;
    if (aligner == 1 && scheme4[2] != scheme4[3])         // deletion != insertion
    {
        // pattern blocking tests min_score after each block of rows: only thresholds that cannot bind are admitted (NULL)
        if (algorithm == 0 && min_score != nullptr) return hipErrorNotSupported;
        const nvbio_hip_sw_scheme w = { scheme4[0], scheme4[1], scheme4[2], scheme4[3] };
        const int e = sw_asym_score(&w, type, algorithm == 0 ? 1u : 0u, patterns, texts, max_pattern_len, max_text_len, n, out_score, out_sink, nullptr, stream);
        if (e == hipSuccess && n && out_ok) return hipMemsetAsync(out_ok, 1, n, to_stream(stream));
        return e;
    }
    const nvbio_hip_gotoh_scheme g = { scheme4[0], scheme4[1], scheme4[2], aligner == 1 ? scheme4[2] : scheme4[3] };
    if (algorithm == 1 && aligner == 1) min_score = nullptr;                            // the text-blocking SW form never exits early
    const int e = full_score_core(&g, nullptr, type, aligner == 1 ? 4u : 3u, algorithm == 0 ? 1u : 0u, patterns, texts, max_pattern_len, max_text_len,
                                  min_score, n, out_score, out_sink, (algorithm == 1 && aligner == 1) ? nullptr : out_ok, stream);
    if (e == hipSuccess && n && algorithm == 1 && aligner == 1 && out_ok) return hipMemsetAsync(out_ok, 1, n, to_stream(stream));
    return e;
}

// GotohAligner<TYPE, SmithWatermanScoringScheme<...>> (nvBowtie's opposite-mate scoring, score_opposite_inl.h:266-269): per-symbol
// mismatch penalties from the read qualities, pattern gap costs in the recurrences, text gap costs on one boundary line.
NVB_API int nvbio_hip_alignment_score_qual(
    const nvbio_hip_gotoh_qual_scheme* scheme, int32_t algorithm, int32_t type,
    const nvbio_hip_string_set* patterns, const uint8_t* quals, uint64_t n_quals, const nvbio_hip_string_set* texts,
    uint32_t max_pattern_len, uint32_t max_text_len, const int32_t* min_score,
    uint32_t n, int32_t* out_score, uint32_t* out_sink, uint8_t* out_ok, void* stream)
{
    if (!scheme || algorithm < 0 || algorithm > 1) return hipErrorInvalidValue;
    if (n != 0 && (!quals || n_quals == 0)) return hipErrorInvalidValue;
    int32_t worst = 0;
    for (int i = 0; i < 256; ++i) worst = std::min(worst, scheme->mismatch[i]);
    const nvbio_hip_gotoh_scheme g = { scheme->match, worst, scheme->pattern_gap_open, scheme->pattern_gap_ext };
    const QualPart q = { quals, n_quals, scheme->mismatch, scheme->text_gap_open, scheme->text_gap_ext };
    return full_score_core(&g, &q, type, 3u, algorithm == 0 ? 1u : 0u, patterns, texts, max_pattern_len, max_text_len, min_score, n, out_score, out_sink, out_ok, stream);
}

// ... with job control: the throughput dispatch behind a device-side gate (wave_form = 0: runs when *gate > gate_limit), or one job per wave over a
// list of jobs (wave_form = 1: entries [0, *n_on_device) of job_index, run when *gate <= gate_limit) -- see nvbio_hip.h
NVB_API int nvbio_hip_alignment_score_qual_jobs(
    const nvbio_hip_gotoh_qual_scheme* scheme, int32_t algorithm, int32_t type,
    const nvbio_hip_string_set* patterns, const uint8_t* quals, uint64_t n_quals, const nvbio_hip_string_set* texts,
    uint32_t max_pattern_len, uint32_t max_text_len, const int32_t* min_score,
    uint32_t n, const uint32_t* n_on_device, const uint32_t* job_index, const uint32_t* gate, uint32_t gate_limit, int32_t wave_form,
    int32_t* out_score, uint32_t* out_sink, uint8_t* out_ok, void* stream)
{
    if (!scheme || algorithm < 0 || algorithm > 1) return hipErrorInvalidValue;
    if (n != 0 && (!quals || n_quals == 0)) return hipErrorInvalidValue;
    if (wave_form && (!n_on_device || !job_index)) return hipErrorInvalidValue;
    int32_t worst = 0;
    for (int i = 0; i < 256; ++i) worst = std::min(worst, scheme->mismatch[i]);
    const nvbio_hip_gotoh_scheme g = { scheme->match, worst, scheme->pattern_gap_open, scheme->pattern_gap_ext };
    const QualPart q = { quals, n_quals, scheme->mismatch, scheme->text_gap_open, scheme->text_gap_ext };
    const FullJobs j = { n_on_device, job_index, gate, gate_limit, wave_form ? 1 : 0 };
    return full_score_core(&g, &q, type, 3u, algorithm == 0 ? 1u : 0u, patterns, texts, max_pattern_len, max_text_len, min_score, n, out_score, out_sink, out_ok, stream, &j);
}
