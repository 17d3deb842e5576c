// full_traceback.hip -- batched full-matrix Gotoh traceback -> CIGAR, gfx950.
//
// Replaces, for GotohAligner<TYPE, SimpleGotohScheme>:
//   aln::BatchedAlignmentTraceback<CHECKPOINTS, stream>::enact (nvbio/alignment/batched.h:432-452) = per job
//   alignment_traceback (alignment_inl.h:365-480): the pattern-blocking score pass with checkpoints every CHECKPOINTS
//   pattern symbols (its sink decides where the walk starts), then per checkpoint the flow submatrix
//   (gotoh_inl.h:512-560, GotohSubmatrixContext::new_cell :407-425) and the walk (gotoh_inl.h:1806-1870), with
//   nvBowtie's CIGAR-forming backtracer (nvBowtie/bowtie2/cuda/alignment_utils.h:125-168).
//
// As in the banded kernel the checkpoint/recompute scheme is replaced by keeping the flow flags of the whole matrix in
// HBM: a lane walks the matrix exactly in the reference's pattern-blocking order (blocks of 8 pattern symbols in
// registers, the {H,E} boundary column over the text as int16 pairs in HBM), writes 8 flag nibbles = one dword per
// (block, text row), lane-interleaved, then walks them back.  150 x 650 (nvBowtie's opposite-mate window) is
// 49 KB of flags per alignment.  Exact while the DP values fit the reference's int16 columns; the host refuses
// schemes / lengths that could leave that range.
#include "common.h"
#include <algorithm>
#include <atomic>

namespace nvb {

enum : uint32_t { T_SUBSTITUTION = 0, T_INSERTION = 1, T_DELETION = 2, T_SINK = 3, T_INSERTION_EXT = 4, T_DELETION_EXT = 8 };

struct FullTbParams {
    StringSet pat, txt;
    int32_t   match, mismatch, gap_open, gap_ext;
    uint32_t  n, max_text_len;
    int32_t*  out_score; uint2* out_sink; uint2* out_source;
    uint16_t* out_cigar; uint32_t cigar_stride; uint32_t* out_cigar_len;
    uint32_t* flags;      // [block][text row][job]: 8 nibbles, pattern column (block*8 + k) at bits 4k
    uint32_t* column;     // [text row][job]: {int16 H, int16 E} of the last pattern column of the previous block
    const uint8_t* quals; uint64_t n_quals;      // quality-aware scheme: mismatch = mm_lut[quality of the pattern symbol]; nullptr = `mismatch`
    int32_t   mm_lut[256];
    int32_t   txt_gap_open, txt_gap_ext;         // the column before the pattern (GLOBAL) is initialised with the text gap costs (gotoh_inl.h:275-279)
    const uint32_t* pending;                     // nullable: the jobs this launch works on (slot -> job), *pending_count of them
    const uint32_t* pending_count;
    // with `pending`: |cheapest gap open|, |cheapest gap extension| (0 = keep the whole text): lets a queued job drop the text columns
    // no alignment ending at its sink with its score can reach (see the kernel)
    int32_t crop_open, crop_ext;
};

template <int TYPE, uint32_t BL>      // BL: pattern symbols per block of the reference's score pass (8 Gotoh, 16 SW / ED): fixes the sink's tie order
__global__ void __launch_bounds__(256) full_gotoh_traceback_kernel(const FullTbParams p)
{
    const uint32_t slot = blockIdx.x * 256u + threadIdx.x;          // where this lane's flags / boundary column live
    if (slot >= (p.pending ? *p.pending_count : p.n)) return;
    const uint32_t tid = p.pending ? p.pending[slot] : slot;         // the job
    const uint64_t pb = p.pat.begin[tid];
    uint64_t       tb = p.txt.begin[tid];
    const uint32_t M  = p.pat.length ? p.pat.length[tid] : p.pat.fixed_length;
    uint32_t       N  = p.txt.length ? p.txt.length[tid] : p.txt.fixed_length;
    const uint64_t n  = p.n;

    // A queued job knows its score and sink (the score pass ran over the whole text).  Any path that ends at the sink with that score
    // spends at most M * match - score on gaps, so it spans at most M + d text symbols, d = the text gaps that budget buys.  Every
    // cell the walk visits lies on such a path, and so does every path that ties with it at a visited cell: the values and directions
    // at the visited cells -- all the walk reads -- are the same whether or not the text left of sink.x - (M + d) is there.  One lane
    // sweeps the whole matrix here, so dropping those columns (opposite-mate windows are ~4 x the read) shortens the kernel's latency.
    uint32_t c0 = 0;
    if (p.pending && TYPE != NVBIO_HIP_GLOBAL && p.crop_ext > 0)
    {
        const int32_t  sc = p.out_score[tid];
        const uint32_t sx = p.out_sink[tid].x;
        const int64_t  budget = int64_t(M) * max(p.match, 0) - sc - p.crop_open;
        const uint32_t d = budget < 0 ? 0u : uint32_t(budget / p.crop_ext) + 1u;
        const uint32_t span = M + d + 2u;
        if (sx != 0xFFFFFFFFu && sx <= N && sx > span) c0 = sx - span;
    }
    tb += c0; N -= c0;

    int32_t  best = -(1 << 30);
    uint32_t bx = 0xFFFFFFFFu, by = 0xFFFFFFFFu;
    auto report = [&](const int32_t s, const uint32_t x, const uint32_t y) { if (best <= s) { best = s; bx = x; by = y; } };

    const int32_t G_o = p.gap_open, G_e = p.gap_ext;
    const int32_t infimum = -32768 - min(G_o, G_e);
    const uint32_t n_blocks = max(1u, (M + BL - 1u) / BL);
    int32_t H_band[BL + 1], F_band[BL + 1];
    uint32_t q_cache[BL];
    int32_t  x_cache[BL];
    #pragma unroll
    for (uint32_t t = 0; t < BL; ++t) { q_cache[t] = 255u; x_cache[t] = p.mismatch; }

    // ---- forward pass in the reference's visiting order (gotoh_inl.h:640-900), flags kept
    for (uint32_t blk = 0; blk < n_blocks; ++blk)
    {
        const uint32_t block = blk * BL;
        const bool last = (blk + 1u == n_blocks);
        #pragma unroll
        for (uint32_t t = 0; t < BL; ++t) if (block + t < M) {
            q_cache[t] = get_symbol(p.pat.s, pb + block + t);
            if (p.quals) x_cache[t] = p.mm_lut[p.quals[min(pb + block + t, p.n_quals - 1u)]];
        }
        #pragma unroll
        for (uint32_t j = 0; j <= BL; ++j) {
            H_band[j] = (TYPE != NVBIO_HIP_LOCAL) ? (block + j > 0u ? G_o + G_e * int32_t(block + j - 1u) : 0) : 0;
            F_band[j] = infimum;
        }
        int32_t temp_i = H_band[0];
        for (uint32_t i = 0; i < N; ++i)
        {
            const uint32_t r_i = get_symbol(p.txt.s, tb + i);
            int32_t H_diag = temp_i, E;
            if (blk == 0u) {        // context.init (:275-279)
                temp_i = (TYPE == NVBIO_HIP_GLOBAL) ? p.txt_gap_open + p.txt_gap_ext * int32_t(i) : 0;
                E      = (TYPE == NVBIO_HIP_LOCAL) ? 0 : infimum;
            } else {
                const uint32_t c = p.column[uint64_t(i) * n + slot];
                temp_i = int32_t(int16_t(c & 0xFFFFu));
                E      = int32_t(int16_t(c >> 16));
            }
            H_band[0] = temp_i;
            uint32_t word[BL / 8u];
            #pragma unroll
            for (uint32_t w8 = 0; w8 < BL / 8u; ++w8) word[w8] = 0;
            #pragma unroll
            for (uint32_t j = 1; j <= BL; ++j)
            {
                const int32_t ftop = F_band[j] + G_e, htop = H_band[j] + G_o;
                F_band[j] = max(ftop, htop);
                const uint32_t fdir = ftop > htop ? T_DELETION_EXT : T_SUBSTITUTION;
                const int32_t eleft = E + G_e, hleft = H_band[j - 1] + G_o;
                E = max(eleft, hleft);
                const uint32_t edir = eleft > hleft ? T_INSERTION_EXT : T_SUBSTITUTION;
                const int32_t diagonal = H_diag + (r_i == q_cache[j - 1] ? p.match : x_cache[j - 1]);
                const int32_t top = F_band[j], left = E;
                int32_t hi = max(max(left, top), diagonal);
                if (TYPE == NVBIO_HIP_LOCAL) hi = max(hi, 0);
                uint32_t hdir = top > left ? (top > diagonal ? T_DELETION : T_SUBSTITUTION) : (left > diagonal ? T_INSERTION : T_SUBSTITUTION);
                if (TYPE == NVBIO_HIP_LOCAL && hi == 0) hdir = T_SINK;
                H_diag = H_band[j];
                H_band[j] = hi;
                word[(j - 1u) >> 3] |= (hdir | edir | fdir) << (4u * ((j - 1u) & 7u));
                if (TYPE == NVBIO_HIP_LOCAL) { if (block + j <= M) report(hi, i + 1u, block + j); }
            }
            p.column[uint64_t(i) * n + slot] = (uint32_t(H_band[BL]) & 0xFFFFu) | (uint32_t(E) << 16);     // make_vector<short> (:565)
            #pragma unroll
            for (uint32_t w8 = 0; w8 < BL / 8u; ++w8)
                p.flags[(uint64_t(blk * (BL / 8u) + w8) * p.max_text_len + i) * n + slot] = word[w8];      // plain store: see banded_traceback.hip
            if (TYPE == NVBIO_HIP_SEMI_GLOBAL && last)
            {
                // save_boundary -> save_Mth: H[i][M] (utils_inl.h:206-226,279-299)
                const uint32_t jm = ((M - 1u) & (BL - 1u)) + 1u;
                int32_t v = 0;
                #pragma unroll
                for (uint32_t j = 1; j <= BL; ++j) if (j == jm) v = H_band[j];
                report(v, i + 1u, M);
            }
        }
    }
    if (TYPE == NVBIO_HIP_GLOBAL)
    {
        const uint32_t jm = ((M - 1u) & (BL - 1u)) + 1u;
        int32_t v = 0;
        #pragma unroll
        for (uint32_t j = 1; j <= BL; ++j) if (j == jm) v = H_band[j];
        report(v, N, M);
    }

    p.out_score[tid] = best;
    p.out_sink[tid]  = make_uint2(bx == 0xFFFFFFFFu ? bx : bx + c0, by);

    // ---- walk back: gotoh_inl.h:1806-1870 over all checkpoints, then alignment_inl.h:443-466; Backtracker::clip / push
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
    int32_t  row = int32_t(bx), col = int32_t(by) - 1;
    uint32_t state = 0;     // 0 = H, 1 = E, 2 = F
    while (row > 0 && col >= 0)
    {
        const uint32_t w  = p.flags[(uint64_t(uint32_t(col) >> 3) * p.max_text_len + uint32_t(row - 1)) * n + slot];
        const uint32_t op = (w >> ((uint32_t(col) & 7u) * 4u)) & 15u, h_op = op & 3u;
        if (TYPE == NVBIO_HIP_LOCAL && state == 0 && h_op == T_SINK) break;
        if (state == 1)      { if ((op & T_INSERTION_EXT) == 0u) state = 0; --col; push(T_INSERTION); }
        else if (state == 2) { if ((op & T_DELETION_EXT)  == 0u) state = 0; --row; push(T_DELETION); }
        else if (h_op == T_INSERTION) state = 1;
        else if (h_op == T_DELETION)  state = 2;
        else { --col; --row; push(T_SUBSTITUTION); }
    }
    uint32_t sx = uint32_t(row), sy = uint32_t(col + 1);
    if (TYPE == NVBIO_HIP_SEMI_GLOBAL || TYPE == NVBIO_HIP_GLOBAL) { if (sx == 0u) for (; sy > 0u; --sy) push(T_INSERTION); }
    if (TYPE == NVBIO_HIP_GLOBAL)                                   { if (sy == 0u) for (; sx > 0u; --sx) push(T_DELETION); }
    flush();
    clip(sy);
    p.out_source[tid]    = make_uint2(sx + c0, sy);
    p.out_cigar_len[tid] = size;
}

// ---- the same traceback with one job spread over the lanes of a wave.  A lane of the kernel above sweeps a whole matrix, so a launch
// lasts one job's latency (19 blocks x 650 rows of dependent loads for nvBowtie's opposite-mate windows) however few jobs there are.
// Here lane l of a job's segment owns pattern block l (BL symbols) and the blocks run as a systolic array: at step t lane l does
// text row t - l, taking the boundary column {H, E} of that row -- the reference's int16 pair -- and the row's text symbol from
// lane l-1, which finished that row one step earlier.  Every cell sees the operands it sees in the reference's block-by-block
// sweep, so values and flow flags are the same; what changes is the visiting order, which matters for LOCAL's sink only: the
// reference keeps the LAST best cell in (block, row, column) order, so each lane keeps its block's last best in (row, column)
// order and ties between lanes go to the higher block.  Flag words go to the job's own region [step][block] (a segment's lanes
// write consecutive dwords), the segment's first lane walks them back.  W = lanes per job (the batch's block count), 64 / W jobs
// share a wave.
template <int TYPE, uint32_t BL>
__global__ void __launch_bounds__(256) full_gotoh_traceback_wave_kernel(const FullTbParams p, const uint32_t W, uint32_t* __restrict__ regions, const uint64_t region_dwords)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (blockIdx.x * 256u + threadIdx.x) >> 6;
    const uint32_t jpw  = 64u / W;
    const uint32_t count = p.pending ? *p.pending_count : p.n;
    if (uint64_t(wave) * jpw >= count) return;                       // the whole wave
    const uint32_t seg = lane / W, l = lane - seg * W;
    const uint32_t slot = wave * jpw + seg;
    const bool have = seg < jpw && slot < count;
    const uint32_t tid = have ? (p.pending ? p.pending[slot] : slot) : 0u;
    const uint64_t pb = p.pat.begin[tid];
    uint64_t       tb = p.txt.begin[tid];
    const uint32_t M  = p.pat.length ? p.pat.length[tid] : p.pat.fixed_length;
    uint32_t       N  = p.txt.length ? p.txt.length[tid] : p.txt.fixed_length;

    uint32_t c0 = 0;                                                 // cropping of queued jobs: see the kernel above
    if (p.pending && TYPE != NVBIO_HIP_GLOBAL && p.crop_ext > 0)
    {
        const int32_t  sc = p.out_score[tid];
        const uint32_t sx = p.out_sink[tid].x;
        const int64_t  budget = int64_t(M) * max(p.match, 0) - sc - p.crop_open;
        const uint32_t d = budget < 0 ? 0u : uint32_t(budget / p.crop_ext) + 1u;
        const uint32_t span = M + d + 2u;
        if (sx != 0xFFFFFFFFu && sx <= N && sx > span) c0 = sx - span;
    }
    tb += c0; N -= c0;

    const uint32_t n_blocks = max(1u, (M + BL - 1u) / BL);
    const bool     mine  = have && l < n_blocks;                     // this lane owns block l of its job
    const uint32_t block = l * BL;
    const bool     last  = (l + 1u == n_blocks);
    uint32_t* fl = regions + uint64_t(slot) * region_dwords;         // [step][block][BL / 8]

    int32_t  best = -(1 << 30);
    uint32_t bx = 0xFFFFFFFFu, by = 0xFFFFFFFFu;
    auto report = [&](const int32_t s, const uint32_t x, const uint32_t y) { if (best <= s) { best = s; bx = x; by = y; } };

    const int32_t G_o = p.gap_open, G_e = p.gap_ext;
    const int32_t infimum = -32768 - min(G_o, G_e);
    int32_t  H_band[BL + 1], F_band[BL + 1];
    uint32_t q_cache[BL];
    int32_t  x_cache[BL];
    #pragma unroll
    for (uint32_t t = 0; t < BL; ++t) {
        q_cache[t] = 255u; x_cache[t] = p.mismatch;
        if (mine && block + t < M) {
            q_cache[t] = get_symbol(p.pat.s, pb + block + t);
            if (p.quals) x_cache[t] = p.mm_lut[p.quals[min(pb + block + t, p.n_quals - 1u)]];
        }
    }
    #pragma unroll
    for (uint32_t j = 0; j <= BL; ++j) {
        H_band[j] = (TYPE != NVBIO_HIP_LOCAL) ? (block + j > 0u ? G_o + G_e * int32_t(block + j - 1u) : 0) : 0;
        F_band[j] = infimum;
    }
    int32_t temp_i = H_band[0];

    // steps of the wave: the longest of its jobs
    uint32_t steps = mine ? N + n_blocks - 1u : 0u;
    if (N == 0u) steps = 0u;
    #pragma unroll
    for (uint32_t o = 32u; o >= 1u; o >>= 1) steps = max(steps, uint32_t(__shfl_xor(int(steps), int(o))));

    uint32_t pub_col = 0u, pub_sym = 0u;
    uint32_t sym_next = (mine && l == 0u && N > 0u) ? get_symbol(p.txt.s, tb) : 0u;      // block 0 reads the text, one row ahead
    for (uint32_t t = 0; t < steps; ++t)
    {
        const uint32_t in_col = uint32_t(__shfl_up(int(pub_col), 1));
        const uint32_t in_sym = uint32_t(__shfl_up(int(pub_sym), 1));
        const uint32_t i = t - l;                                    // wraps for t < l: fails the range test
        if (!(mine && i < N)) continue;
        uint32_t r_i;
        int32_t  H_diag = temp_i, E;
        if (l == 0u) {                                               // context.init (gotoh_inl.h:275-279)
            r_i = sym_next;
            if (i + 1u < N) sym_next = get_symbol(p.txt.s, tb + i + 1u);
            temp_i = (TYPE == NVBIO_HIP_GLOBAL) ? p.txt_gap_open + p.txt_gap_ext * int32_t(i) : 0;
            E      = (TYPE == NVBIO_HIP_LOCAL) ? 0 : infimum;
        } else {
            r_i = in_sym;
            temp_i = int32_t(int16_t(in_col & 0xFFFFu));
            E      = int32_t(int16_t(in_col >> 16));
        }
        H_band[0] = temp_i;
        uint32_t word[BL / 8u];
        #pragma unroll
        for (uint32_t w8 = 0; w8 < BL / 8u; ++w8) word[w8] = 0;
        #pragma unroll
        for (uint32_t j = 1; j <= BL; ++j)
        {
            const int32_t ftop = F_band[j] + G_e, htop = H_band[j] + G_o;
            F_band[j] = max(ftop, htop);
            const uint32_t fdir = ftop > htop ? T_DELETION_EXT : T_SUBSTITUTION;
            const int32_t eleft = E + G_e, hleft = H_band[j - 1] + G_o;
            E = max(eleft, hleft);
            const uint32_t edir = eleft > hleft ? T_INSERTION_EXT : T_SUBSTITUTION;
            const int32_t diagonal = H_diag + (r_i == q_cache[j - 1] ? p.match : x_cache[j - 1]);
            const int32_t top = F_band[j], left = E;
            int32_t hi = max(max(left, top), diagonal);
            if (TYPE == NVBIO_HIP_LOCAL) hi = max(hi, 0);
            uint32_t hdir = top > left ? (top > diagonal ? T_DELETION : T_SUBSTITUTION) : (left > diagonal ? T_INSERTION : T_SUBSTITUTION);
            if (TYPE == NVBIO_HIP_LOCAL && hi == 0) hdir = T_SINK;
            H_diag = H_band[j];
            H_band[j] = hi;
            word[(j - 1u) >> 3] |= (hdir | edir | fdir) << (4u * ((j - 1u) & 7u));
            if (TYPE == NVBIO_HIP_LOCAL) { if (block + j <= M) report(hi, i + 1u, block + j); }
        }
        pub_col = (uint32_t(H_band[BL]) & 0xFFFFu) | (uint32_t(E) << 16);        // make_vector<short> (gotoh_inl.h:565)
        pub_sym = r_i;
        #pragma unroll
        for (uint32_t w8 = 0; w8 < BL / 8u; ++w8)
            fl[(uint64_t(t) * n_blocks + l) * (BL / 8u) + w8] = word[w8];
        if (TYPE == NVBIO_HIP_SEMI_GLOBAL && last)
        {
            const uint32_t jm = ((M - 1u) & (BL - 1u)) + 1u;
            int32_t v = 0;
            #pragma unroll
            for (uint32_t j = 1; j <= BL; ++j) if (j == jm) v = H_band[j];
            report(v, i + 1u, M);
        }
    }
    if (TYPE == NVBIO_HIP_GLOBAL && mine && last)
    {
        const uint32_t jm = ((M - 1u) & (BL - 1u)) + 1u;
        int32_t v = 0;
        #pragma unroll
        for (uint32_t j = 1; j <= BL; ++j) if (j == jm) v = H_band[j];
        report(v, N, M);
    }

    // the job's sink: the best of its lanes, the higher block on ties (it is visited later)
    {
        int64_t key = (mine && bx != 0xFFFFFFFFu) ? int64_t(best) * 64 + int64_t(l) : INT64_MIN;
        int64_t top_key = INT64_MIN;
        for (uint32_t k = 0; k < W; ++k) {
            const int64_t c = __shfl(key, int(seg * W + k));
            top_key = max(top_key, c);
        }
        const uint32_t win = (top_key == INT64_MIN) ? seg * W : seg * W + uint32_t(top_key & 63);
        best = __shfl(best, int(win)); bx = uint32_t(__shfl(int(bx), int(win))); by = uint32_t(__shfl(int(by), int(win)));
    }
    __threadfence();                                                 // the walk reads what other lanes stored
    if (!(have && l == 0u)) return;

    p.out_score[tid] = best;
    p.out_sink[tid]  = make_uint2(bx == 0xFFFFFFFFu ? bx : bx + c0, by);

    // ---- walk back, as above
    uint16_t* cigar = p.out_cigar + uint64_t(tid) * p.cigar_stride;
    uint32_t  size = 0, run_type = 255u, run_len = 0;
    auto flush = [&]() { if (run_len) { if (size < p.cigar_stride) cigar[size] = uint16_t(run_type | (run_len << 2)); ++size; run_len = 0; } };
    auto clip  = [&](const uint32_t n) { if (n) { if (size < p.cigar_stride) cigar[size] = uint16_t(3u | (n << 2)); ++size; } };
    auto push  = [&](const uint32_t t) { if (t != run_type) { flush(); run_type = t; } ++run_len; };

    if (bx == 0xFFFFFFFFu || by == 0xFFFFFFFFu) {
        p.out_source[tid]    = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
        p.out_cigar_len[tid] = 0;
        return;
    }
    clip(M - by);
    int32_t  row = int32_t(bx), col = int32_t(by) - 1;
    uint32_t state = 0;     // 0 = H, 1 = E, 2 = F
    while (row > 0 && col >= 0)
    {
        const uint32_t blk = uint32_t(col) / BL, w8 = (uint32_t(col) % BL) >> 3;
        const uint32_t w  = __hip_atomic_load(fl + (uint64_t(uint32_t(row - 1) + blk) * n_blocks + blk) * (BL / 8u) + w8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t op = (w >> ((uint32_t(col) & 7u) * 4u)) & 15u, h_op = op & 3u;
        if (TYPE == NVBIO_HIP_LOCAL && state == 0 && h_op == T_SINK) break;
        if (state == 1)      { if ((op & T_INSERTION_EXT) == 0u) state = 0; --col; push(T_INSERTION); }
        else if (state == 2) { if ((op & T_DELETION_EXT)  == 0u) state = 0; --row; push(T_DELETION); }
        else if (h_op == T_INSERTION) state = 1;
        else if (h_op == T_DELETION)  state = 2;
        else { --col; --row; push(T_SUBSTITUTION); }
    }
    uint32_t sx = uint32_t(row), sy = uint32_t(col + 1);
    if (TYPE == NVBIO_HIP_SEMI_GLOBAL || TYPE == NVBIO_HIP_GLOBAL) { if (sx == 0u) for (; sy > 0u; --sy) push(T_INSERTION); }
    if (TYPE == NVBIO_HIP_GLOBAL)                                   { if (sy == 0u) for (; sx > 0u; --sx) push(T_DELETION); }
    flush();
    clip(sy);
    p.out_source[tid]    = make_uint2(sx + c0, sy);
    p.out_cigar_len[tid] = size;
}

// ---- the ungapped fast path (LOCAL / SEMI_GLOBAL): as in banded_traceback.hip.  Given score and sink from the score kernel (the same
// pattern-blocking DP, so the same sink: checked on tie-heavy batches), walk the diagonal up-left from the sink adding substitution
// scores; when the sum over k cells equals the score, every H on that segment equals its partial sum and each cell's direction is
// SUBSTITUTION, so the walk of the full kernel is exactly that diagonal: LOCAL stops at the first such k (H = 0 there: SINK, or the
// matrix edge), SEMI_GLOBAL needs the whole pattern (k = M, free text start).  Other jobs are queued for the full kernel.
template <int TYPE>
__global__ void __launch_bounds__(256) full_traceback_diagonal_kernel(const FullTbParams p, uint32_t* __restrict__ pending, uint32_t* __restrict__ pending_count)
{
    const uint32_t tid = blockIdx.x * 256u + threadIdx.x;
    if (tid >= p.n) return;
    const uint2 sink = p.out_sink[tid];
    if (sink.x == 0xFFFFFFFFu || sink.y == 0xFFFFFFFFu) {
        p.out_source[tid] = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
        p.out_cigar_len[tid] = 0;
        return;
    }
    const int32_t  best = p.out_score[tid];
    const uint64_t pb = p.pat.begin[tid], tb = p.txt.begin[tid];
    const uint32_t M  = p.pat.length ? p.pat.length[tid] : p.pat.fixed_length;
    const uint32_t bx = sink.x, by = sink.y;
    bool     found = false;
    uint32_t k_found = 0;
    const uint32_t kmax = min(bx, by);
    if (!(TYPE == NVBIO_HIP_LOCAL && best <= 0) && !(TYPE == NVBIO_HIP_SEMI_GLOBAL && bx < by))
    {
        int32_t c = 0;
        // cells k = 0 .. kmax-1: text symbol bx-1-k, pattern symbol by-1-k; 16 per fetch, walking down from the sink
        for (uint32_t k0 = 0; k0 < kmax && !found; k0 += 16u)
        {
            const uint32_t cnt = min(16u, kmax - k0);
            const uint32_t lo_p = by - k0 - cnt, lo_t = bx - k0 - cnt;                 // lowest symbol index of this group
            const uint64_t pq = (p.pat.s.bits == 2) ? expand_2to4(fetch16_2bit(p.pat.s, pb + lo_p)) : fetch16_4bit(p.pat.s, pb + lo_p);
            const uint64_t tg = expand_2to4(fetch16_2bit(p.txt.s, tb + lo_t));
            for (uint32_t u = 0; u < cnt; ++u)
            {
                const uint32_t m = cnt - 1u - u;                                       // nibble of cell k0 + u
                const uint32_t q = uint32_t(pq >> (4u * m)) & 15u, g = uint32_t(tg >> (4u * m)) & 15u;
                int32_t sc = p.match;
                if (g != q) sc = p.quals ? p.mm_lut[p.quals[min(pb + lo_p + m, p.n_quals - 1u)]] : p.mismatch;
                c += sc;
                if (TYPE == NVBIO_HIP_LOCAL && c == best) { found = true; k_found = k0 + u + 1u; break; }
            }
        }
        if (TYPE == NVBIO_HIP_SEMI_GLOBAL) { found = (c == best); k_found = by; }
    }
    {
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
    emit(3u, M - by);
    emit(T_SUBSTITUTION, k_found);
    emit(3u, by - k_found);
    p.out_source[tid]    = make_uint2(bx - k_found, by - k_found);
    p.out_cigar_len[tid] = size;
}

// ---- windows whose alignment is known to end at their last symbol with a known score (nvBowtie's opposite-mate tracebacks run over
// [alignment, alignment + sink), traceback_inl.h:833-905: the window ends where the scoring pass put the sink, and starts where the
// scoring window started -- several read lengths to the left).  An alignment of score S that ends at text row N spends at most
// M * match - S on gaps, so it starts no earlier than row N - (M + d) (d = the text gaps that budget buys, as in the queued-job
// cropping above), and the DP values on every path to it, and on every path tying with it, do not depend on the rows before that:
// the first N - span rows of the window are dropped before any kernel runs, and put back into sink.x / source.x at the end.
// Cell values can only shrink when paths are removed, S is the maximum of the whole window, and the cropped matrix keeps every cell
// that attains it, so score, sink, source and CIGAR are those of the whole window.
__global__ void __launch_bounds__(256)
crop_windows_kernel(const FullTbParams p, const int32_t* __restrict__ known_score, uint64_t* __restrict__ new_begin, uint32_t* __restrict__ new_len, uint32_t* __restrict__ c0_out)
{
    const uint32_t tid = blockIdx.x * 256u + threadIdx.x;
    if (tid >= p.n) return;
    const uint32_t M = p.pat.length ? p.pat.length[tid] : p.pat.fixed_length;
    const uint32_t N = p.txt.length ? p.txt.length[tid] : p.txt.fixed_length;
    const int64_t  budget = int64_t(M) * max(p.match, 0) - known_score[tid] - p.crop_open;
    const uint32_t d = budget < 0 ? 0u : uint32_t(min(budget / p.crop_ext, int64_t(1 << 20))) + 1u;
    const uint32_t span = M + d + 2u;
    const uint32_t c0 = N > span ? N - span : 0u;
    new_begin[tid] = p.txt.begin[tid] + c0;
    new_len[tid]   = N - c0;
    c0_out[tid]    = c0;
}
__global__ void __launch_bounds__(256)
uncrop_kernel(uint32_t n, const uint32_t* __restrict__ c0, uint2* __restrict__ sink, uint2* __restrict__ source)
{
    const uint32_t tid = blockIdx.x * 256u + threadIdx.x;
    if (tid >= n) return;
    if (sink[tid].x != 0xFFFFFFFFu)   sink[tid].x += c0[tid];
    if (source[tid].x != 0xFFFFFFFFu) source[tid].x += c0[tid];
}

// The premise of the _known_score forms, checked on what the cropped DP found (cropped coordinates): the best score is the one the
// caller announced and its alignment ends at the window's last row.  A wrong known_score (too high: the window was cut too short)
// or a window that does not end at its alignment shows up here; such jobs are queued and traced again over their whole window.
__global__ void __launch_bounds__(256)
verify_known_kernel(uint32_t n, const int32_t* __restrict__ known_score, const int32_t* __restrict__ score, const uint2* __restrict__ sink,
                    const uint32_t* __restrict__ new_len, const uint32_t* __restrict__ c0, uint32_t* __restrict__ bad, uint32_t* __restrict__ bad_count)
{
    const uint32_t tid = blockIdx.x * 256u + threadIdx.x;
    if (tid >= n) return;
    if (c0[tid] == 0u) return;                                                     // nothing was dropped: this IS the plain traceback
    if (score[tid] == known_score[tid] && sink[tid].x == new_len[tid]) return;
    bad[atomicAdd(bad_count, 1u)] = tid;
}
__global__ void __launch_bounds__(256)
gather_jobs_kernel(uint32_t m, const uint32_t* __restrict__ bad, const StringSet pat, const StringSet txt,
                   uint64_t* __restrict__ pb, uint32_t* __restrict__ pl, uint64_t* __restrict__ tb, uint32_t* __restrict__ tl)
{
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (k >= m) return;
    const uint32_t j = bad[k];
    pb[k] = pat.begin[j]; pl[k] = pat.length ? pat.length[j] : pat.fixed_length;
    tb[k] = txt.begin[j]; tl[k] = txt.length ? txt.length[j] : txt.fixed_length;
}
__global__ void __launch_bounds__(256)
scatter_tracebacks_kernel(uint32_t m, const uint32_t* __restrict__ bad, const int32_t* __restrict__ score, const uint2* __restrict__ sink, const uint2* __restrict__ source,
                          const uint16_t* __restrict__ cigar, const uint32_t* __restrict__ cigar_len, uint32_t cigar_stride,
                          int32_t* __restrict__ out_score, uint2* __restrict__ out_sink, uint2* __restrict__ out_source, uint16_t* __restrict__ out_cigar, uint32_t* __restrict__ out_cigar_len)
{
    const uint32_t k = blockIdx.x, j = bad[k];
    if (threadIdx.x == 0u) { out_score[j] = score[k]; out_sink[j] = sink[k]; out_source[j] = source[k]; out_cigar_len[j] = cigar_len[k]; }
    const uint32_t words = cigar_len[k] < cigar_stride ? cigar_len[k] : cigar_stride;
    for (uint32_t w = threadIdx.x; w < words; w += 256u) out_cigar[uint64_t(j) * cigar_stride + w] = cigar[uint64_t(k) * cigar_stride + w];
}

} // namespace nvb

using namespace nvb;

NVB_API uint64_t nvbio_hip_gotoh_traceback_temp_bytes(uint32_t max_pattern_len, uint32_t max_text_len, uint32_t n)
{
    const uint64_t blocks = 2u * std::max<uint64_t>(1u, (uint64_t(max_pattern_len) + 15u) / 16u);     // 8-column flag words, whole 16-column blocks
    // flags + the boundary column (or, per job of the wave kernel, [step][block] flag words: text rows + blocks steps) + the queue of gapped jobs
    // ... + the cropped windows of the _known_score forms (begin, length, dropped rows: 16 bytes per job) + their queue of jobs to redo
    return ((blocks + 1u) * uint64_t(max_text_len) + blocks * blocks) * uint64_t(n) * 4u + uint64_t(n) * 4u + 256u + 8u + uint64_t(n) * 16u + uint64_t(n) * 4u + 8u;
}

static std::atomic<uint64_t> g_known_score_redone{0};

struct TbQualPart { const uint8_t* quals; uint64_t n_quals; const int32_t* mismatch; int32_t text_gap_open, text_gap_ext; };

static int full_traceback_core(
    const nvbio_hip_gotoh_scheme* scheme, const TbQualPart* qual, int32_t type, uint32_t block_len,
    const nvbio_hip_string_set* patterns, const nvbio_hip_string_set* texts,
    uint32_t max_pattern_len, uint32_t max_text_len, uint32_t n,
    int32_t* out_score, uint32_t* out_sink, uint32_t* out_source,
    uint16_t* out_cigar, uint32_t cigar_stride, uint32_t* out_cigar_len,
    void* temp, uint64_t temp_bytes, void* stream, const int32_t* known_score = nullptr)
{
    if (!scheme || !patterns || !texts) return hipErrorInvalidValue;
    if (type < 0 || type > 2) return hipErrorInvalidValue;
    if (!(patterns->bits == 2 || patterns->bits == 4) || texts->bits != 2) return hipErrorNotSupported;
    if (n == 0) return hipSuccess;
    if (!out_score || !out_sink || !out_source || !out_cigar || !out_cigar_len || cigar_stride == 0) return hipErrorInvalidValue;
    if (!patterns->words || !texts->words || !patterns->begin || !texts->begin || patterns->n_words == 0 || texts->n_words == 0) return hipErrorInvalidValue;
    const uint32_t maxM = patterns->length ? max_pattern_len : patterns->fixed_length;
    const uint32_t maxN = texts->length ? max_text_len : texts->fixed_length;
    if (maxM == 0 || maxN == 0) return hipErrorInvalidValue;
    if (maxM >= (1u << 14) || maxN >= (1u << 14)) return hipErrorNotSupported;       // io::Cigar::m_len is 14 bits
    auto iabs = [](int32_t v) { return v < 0 ? -int64_t(v) : int64_t(v); };
    int64_t A = std::max(std::max(iabs(scheme->match), iabs(scheme->mismatch)), std::max(iabs(scheme->gap_open), iabs(scheme->gap_ext)));
    if (qual) { for (int i = 0; i < 256; ++i) A = std::max(A, iabs(qual->mismatch[i])); A = std::max(A, std::max(iabs(qual->text_gap_open), iabs(qual->text_gap_ext))); }
    const int64_t span = (type == NVBIO_HIP_GLOBAL) ? int64_t(maxM) + maxN + 4 : int64_t(maxM) + 4;
    const bool gaps_cost = scheme->gap_open <= 0 && scheme->gap_ext <= 0 && (!qual || (qual->text_gap_open <= 0 && qual->text_gap_ext <= 0));
    if (!(gaps_cost && span * A < 30000)) return hipErrorNotSupported;   // int16 checkpoints / columns exact only here (no gap move may earn score)
    const uint64_t need = nvbio_hip_gotoh_traceback_temp_bytes(maxM, maxN, n);
    if (!temp || temp_bytes < need) return hipErrorInvalidValue;

    FullTbParams p;
    p.pat = make_string_set(patterns); p.txt = make_string_set(texts);
    p.match = scheme->match; p.mismatch = scheme->mismatch; p.gap_open = scheme->gap_open; p.gap_ext = scheme->gap_ext;
    p.n = n; p.max_text_len = maxN;
    p.quals = qual ? qual->quals : nullptr; p.n_quals = qual ? qual->n_quals : 0;
    for (int i = 0; i < 256; ++i) p.mm_lut[i] = qual ? qual->mismatch[i] : scheme->mismatch;
    p.txt_gap_open = qual ? qual->text_gap_open : scheme->gap_open; p.txt_gap_ext = qual ? qual->text_gap_ext : scheme->gap_ext;
    p.out_score = out_score; p.out_sink = reinterpret_cast<uint2*>(out_sink); p.out_source = reinterpret_cast<uint2*>(out_source);
    p.out_cigar = out_cigar; p.cigar_stride = cigar_stride; p.out_cigar_len = out_cigar_len;
    p.column = static_cast<uint32_t*>(temp);
    p.flags  = p.column + uint64_t(maxN) * n;
    p.pending = nullptr; p.pending_count = nullptr;
    p.crop_open = int32_t(std::min(iabs(scheme->gap_open), qual ? iabs(qual->text_gap_open) : iabs(scheme->gap_open)));
    p.crop_ext  = int32_t(std::min(iabs(scheme->gap_ext),  qual ? iabs(qual->text_gap_ext)  : iabs(scheme->gap_ext)));
    const dim3 grid((n + 255u) / 256u), block(256);
    hipStream_t s = to_stream(stream);
    // temp: [job regions | queue of gapped jobs, its counter | cropped windows]
    const uint64_t redo_off = need - uint64_t(n) * 4u - 8u;
    const uint64_t queue_off = redo_off - uint64_t(n) * 16u - 8u - 256u - uint64_t(n) * 4u, crop_off = (queue_off + uint64_t(n) * 4u + 256u + 7u) & ~uint64_t(7);
    nvbio_hip_string_set cropped = *texts;
    const nvbio_hip_string_set* whole_texts = texts;
    uint32_t* c0 = nullptr;
    if (known_score && type != NVBIO_HIP_GLOBAL && p.crop_ext > 0)
    {
        uint64_t* nb = reinterpret_cast<uint64_t*>(static_cast<uint8_t*>(temp) + crop_off);
        uint32_t* nl = reinterpret_cast<uint32_t*>(nb + n);
        c0 = nl + n;
        hipLaunchKernelGGL(crop_windows_kernel, grid, block, 0, s, p, known_score, nb, nl, c0);
        if (hipError_t e = hipGetLastError()) return e;
        cropped.begin = nb; cropped.length = nl; cropped.fixed_length = 0;
        texts = &cropped;
        p.txt = make_string_set(texts);
    }
    const int rc = [&]() -> int {
    if (block_len == 8u && type != NVBIO_HIP_GLOBAL && maxM <= 512u)
    {
        // score + sink of every job from the (wave-per-alignment, 16-bit) pattern-blocking score kernel, CIGARs of the ungapped ones
        // from the diagonal check, the rest queued for the full kernel
        uint32_t* pending = reinterpret_cast<uint32_t*>(static_cast<uint8_t*>(temp) + queue_off);
        uint32_t* pending_count = pending + n;
        if (hipError_t e = hipMemsetAsync(pending_count, 0, 4, s)) return e;
        int err;
        if (qual) {
            nvbio_hip_gotoh_qual_scheme qs;
            qs.match = scheme->match; qs.pattern_gap_open = scheme->gap_open; qs.pattern_gap_ext = scheme->gap_ext;
            qs.text_gap_open = qual->text_gap_open; qs.text_gap_ext = qual->text_gap_ext;
            for (int i = 0; i < 256; ++i) qs.mismatch[i] = qual->mismatch[i];
            err = nvbio_hip_alignment_score_qual(&qs, NVBIO_HIP_PATTERN_BLOCKING, type, patterns, qual->quals, qual->n_quals, texts, maxM, maxN, nullptr, n,
                                                 out_score, out_sink, nullptr, stream);
        } else {
            const int32_t s4[4] = { scheme->match, scheme->mismatch, scheme->gap_open, scheme->gap_ext };
            err = nvbio_hip_alignment_score(NVBIO_HIP_GOTOH_ALIGNER, NVBIO_HIP_PATTERN_BLOCKING, s4, type, patterns, texts, maxM, maxN, nullptr, n,
                                            out_score, out_sink, nullptr, stream);
        }
        if (err == hipSuccess) {
            if (type == NVBIO_HIP_LOCAL) hipLaunchKernelGGL(full_traceback_diagonal_kernel<NVBIO_HIP_LOCAL>,       grid, block, 0, s, p, pending, pending_count);
            else                         hipLaunchKernelGGL(full_traceback_diagonal_kernel<NVBIO_HIP_SEMI_GLOBAL>, grid, block, 0, s, p, pending, pending_count);
            if (hipError_t e = hipGetLastError()) return e;
            p.pending = pending; p.pending_count = pending_count;
        } else if (err != hipErrorNotSupported) return err;          // a shape the score kernel does not take: every job goes to the full kernel
    }
    // one job per wave segment when its blocks fit a wave (patterns to 64 blocks); NVBIO_HIP_TRACEBACK_LANES=1 keeps one job per lane
    const uint32_t nb = std::max(1u, (maxM + block_len - 1u) / block_len);
    if (nb <= 64u && test_switch(SW_TRACEBACK_LANES) != 1)
    {
        const uint64_t blocks8 = 2u * std::max<uint64_t>(1u, (uint64_t(maxM) + 15u) / 16u);
        const uint64_t region = (blocks8 + 1u) * uint64_t(maxN) + blocks8 * blocks8;          // dwords per job (temp_bytes above)
        const uint32_t jpw = 64u / nb;
        const dim3 wgrid(uint32_t((uint64_t(n) + jpw * 4u - 1u) / (jpw * 4u)));
        uint32_t* regions = static_cast<uint32_t*>(temp);
        g_last_kernel = "full_gotoh_traceback_wave_kernel";
        if (block_len == 16u) {
            switch (type) {
            case NVBIO_HIP_LOCAL:       hipLaunchKernelGGL((full_gotoh_traceback_wave_kernel<NVBIO_HIP_LOCAL, 16u>),       wgrid, block, 0, s, p, nb, regions, region); break;
            case NVBIO_HIP_SEMI_GLOBAL: hipLaunchKernelGGL((full_gotoh_traceback_wave_kernel<NVBIO_HIP_SEMI_GLOBAL, 16u>), wgrid, block, 0, s, p, nb, regions, region); break;
            default:                    hipLaunchKernelGGL((full_gotoh_traceback_wave_kernel<NVBIO_HIP_GLOBAL, 16u>),      wgrid, block, 0, s, p, nb, regions, region); break;
            }
        } else {
            switch (type) {
            case NVBIO_HIP_LOCAL:       hipLaunchKernelGGL((full_gotoh_traceback_wave_kernel<NVBIO_HIP_LOCAL, 8u>),       wgrid, block, 0, s, p, nb, regions, region); break;
            case NVBIO_HIP_SEMI_GLOBAL: hipLaunchKernelGGL((full_gotoh_traceback_wave_kernel<NVBIO_HIP_SEMI_GLOBAL, 8u>), wgrid, block, 0, s, p, nb, regions, region); break;
            default:                    hipLaunchKernelGGL((full_gotoh_traceback_wave_kernel<NVBIO_HIP_GLOBAL, 8u>),      wgrid, block, 0, s, p, nb, regions, region); break;
            }
        }
        return hipGetLastError();
    }
    g_last_kernel = "full_gotoh_traceback_kernel";
    if (block_len == 16u) {
        switch (type) {
        case NVBIO_HIP_LOCAL:       hipLaunchKernelGGL((full_gotoh_traceback_kernel<NVBIO_HIP_LOCAL, 16u>),       grid, block, 0, s, p); break;
        case NVBIO_HIP_SEMI_GLOBAL: hipLaunchKernelGGL((full_gotoh_traceback_kernel<NVBIO_HIP_SEMI_GLOBAL, 16u>), grid, block, 0, s, p); break;
        default:                    hipLaunchKernelGGL((full_gotoh_traceback_kernel<NVBIO_HIP_GLOBAL, 16u>),      grid, block, 0, s, p); break;
        }
    } else {
        switch (type) {
        case NVBIO_HIP_LOCAL:       hipLaunchKernelGGL((full_gotoh_traceback_kernel<NVBIO_HIP_LOCAL, 8u>),       grid, block, 0, s, p); break;
        case NVBIO_HIP_SEMI_GLOBAL: hipLaunchKernelGGL((full_gotoh_traceback_kernel<NVBIO_HIP_SEMI_GLOBAL, 8u>), grid, block, 0, s, p); break;
        default:                    hipLaunchKernelGGL((full_gotoh_traceback_kernel<NVBIO_HIP_GLOBAL, 8u>),      grid, block, 0, s, p); break;
        }
    }
    return hipGetLastError();
    }();
    if (rc != hipSuccess || !c0) return rc;
    // the premise, checked per job on the cropped result; the (normally empty) list of jobs it failed for is traced again uncropped
    uint32_t* bad = reinterpret_cast<uint32_t*>(static_cast<uint8_t*>(temp) + redo_off);
    uint32_t* bad_count = bad + n;
    if (hipError_t e = hipMemsetAsync(bad_count, 0, 4, s)) return e;
    hipLaunchKernelGGL(verify_known_kernel, grid, block, 0, s, n, known_score, out_score, p.out_sink, cropped.length, c0, bad, bad_count);
    hipLaunchKernelGGL(uncrop_kernel, grid, block, 0, s, n, c0, p.out_sink, p.out_source);
    if (hipError_t e = hipGetLastError()) return e;
    uint32_t m = 0;
    if (hipError_t e = hipMemcpyAsync(&m, bad_count, 4, hipMemcpyDeviceToHost, s)) return e;
    if (hipError_t e = hipStreamSynchronize(s)) return e;
    if (m == 0) return hipSuccess;
    g_known_score_redone += m;
    const uint64_t sub_temp = nvbio_hip_gotoh_traceback_temp_bytes(maxM, maxN, m);
    const uint64_t row = 8u + 4u + 8u + 4u + 4u + 8u + 8u + 4u + uint64_t(cigar_stride) * 2u;
    uint8_t* blk = nullptr;
    if (hipError_t e = hipMalloc(reinterpret_cast<void**>(&blk), sub_temp + uint64_t(m) * row + 256u)) return e;
    uint8_t* q = blk + ((sub_temp + 15u) & ~uint64_t(15));
    uint64_t* pb = reinterpret_cast<uint64_t*>(q); q += uint64_t(m) * 8u;
    uint64_t* tb = reinterpret_cast<uint64_t*>(q); q += uint64_t(m) * 8u;
    uint2* r_sink = reinterpret_cast<uint2*>(q);   q += uint64_t(m) * 8u;
    uint2* r_src  = reinterpret_cast<uint2*>(q);   q += uint64_t(m) * 8u;
    uint32_t* pl = reinterpret_cast<uint32_t*>(q); q += uint64_t(m) * 4u;
    uint32_t* tl = reinterpret_cast<uint32_t*>(q); q += uint64_t(m) * 4u;
    int32_t* r_score = reinterpret_cast<int32_t*>(q); q += uint64_t(m) * 4u;
    uint32_t* r_len = reinterpret_cast<uint32_t*>(q); q += uint64_t(m) * 4u;
    uint16_t* r_cigar = reinterpret_cast<uint16_t*>(q);
    const dim3 mgrid((m + 255u) / 256u);
    hipLaunchKernelGGL(gather_jobs_kernel, mgrid, block, 0, s, m, bad, make_string_set(patterns), make_string_set(whole_texts), pb, pl, tb, tl);
    nvbio_hip_string_set sp = *patterns, st = *whole_texts;
    sp.begin = pb; sp.length = pl; sp.fixed_length = 0; st.begin = tb; st.length = tl; st.fixed_length = 0;
    int e2 = full_traceback_core(scheme, qual, type, block_len, &sp, &st, maxM, maxN, m, r_score, reinterpret_cast<uint32_t*>(r_sink), reinterpret_cast<uint32_t*>(r_src),
                                 r_cigar, cigar_stride, r_len, blk, sub_temp, stream);
    if (e2 == hipSuccess) {
        hipLaunchKernelGGL(scatter_tracebacks_kernel, dim3(m), block, 0, s, m, bad, r_score, r_sink, r_src, r_cigar, r_len, cigar_stride,
                           out_score, p.out_sink, p.out_source, out_cigar, out_cigar_len);
        e2 = hipGetLastError();
    }
    (void)hipStreamSynchronize(s);
    (void)hipFree(blk);
    return e2;
}

// jobs the _known_score forms had to trace again over their whole window since the library was loaded (a caller whose premise holds sees 0)
NVB_API uint64_t nvbio_hip_known_score_redone(void) { return g_known_score_redone; }

NVB_API int nvbio_hip_gotoh_traceback(
    const nvbio_hip_gotoh_scheme* scheme, int32_t type,
    const nvbio_hip_string_set* patterns, const nvbio_hip_string_set* texts,
    uint32_t max_pattern_len, uint32_t max_text_len, uint32_t n,
    int32_t* out_score, uint32_t* out_sink, uint32_t* out_source,
    uint16_t* out_cigar, uint32_t cigar_stride, uint32_t* out_cigar_len,
    void* temp, uint64_t temp_bytes, void* stream)
{
    return full_traceback_core(scheme, nullptr, type, 8u, patterns, texts, max_pattern_len, max_text_len, n, out_score, out_sink, out_source,
                               out_cigar, cigar_stride, out_cigar_len, temp, temp_bytes, stream);
}

// SmithWatermanAligner / EditDistanceAligner (sw_inl.h:389-396, 475-500, 1660-1700): with deletion == insertion the matrix, the
// flow directions and the walk are those of Gotoh with gap_open == gap_ext (the extension flags never fire); the sink comes
// from the 16-column pattern-blocking score pass.
NVB_API int nvbio_hip_sw_traceback(
    const nvbio_hip_sw_scheme* scheme, int32_t type,
    const nvbio_hip_string_set* patterns, const nvbio_hip_string_set* texts,
    uint32_t max_pattern_len, uint32_t max_text_len, uint32_t n,
    int32_t* out_score, uint32_t* out_sink, uint32_t* out_source,
    uint16_t* out_cigar, uint32_t cigar_stride, uint32_t* out_cigar_len,
    void* temp, uint64_t temp_bytes, void* stream)
{
    if (!scheme) return hipErrorInvalidValue;
    if (scheme->deletion != scheme->insertion) return hipErrorNotSupported;
    const nvbio_hip_gotoh_scheme g = { scheme->match, scheme->mismatch, scheme->deletion, scheme->deletion };
    return full_traceback_core(&g, nullptr, type, 16u, patterns, texts, max_pattern_len, max_text_len, n, out_score, out_sink, out_source,
                               out_cigar, cigar_stride, out_cigar_len, temp, temp_bytes, stream);
}

// nvBowtie's opposite-mate traceback: GotohAligner<TYPE, SmithWatermanScoringScheme<...>> over the full matrix
NVB_API int nvbio_hip_gotoh_traceback_qual(
    const nvbio_hip_gotoh_qual_scheme* scheme, int32_t type,
    const nvbio_hip_string_set* patterns, const uint8_t* quals, uint64_t n_quals, const nvbio_hip_string_set* texts,
    uint32_t max_pattern_len, uint32_t max_text_len, uint32_t n,
    int32_t* out_score, uint32_t* out_sink, uint32_t* out_source,
    uint16_t* out_cigar, uint32_t cigar_stride, uint32_t* out_cigar_len,
    void* temp, uint64_t temp_bytes, void* stream)
{
    if (!scheme) return hipErrorInvalidValue;
    if (n != 0 && (!quals || n_quals == 0)) return hipErrorInvalidValue;
    int32_t worst = 0;
    for (int i = 0; i < 256; ++i) worst = std::min(worst, scheme->mismatch[i]);
    const nvbio_hip_gotoh_scheme g = { scheme->match, worst, scheme->pattern_gap_open, scheme->pattern_gap_ext };
    const TbQualPart q = { quals, n_quals, scheme->mismatch, scheme->text_gap_open, scheme->text_gap_ext };
    return full_traceback_core(&g, &q, type, 8u, patterns, texts, max_pattern_len, max_text_len, n, out_score, out_sink, out_source,
                               out_cigar, cigar_stride, out_cigar_len, temp, temp_bytes, stream);
}

// The same two with the callers' knowledge that every job's best alignment has the given score and ends at the last symbol of its text
// (see crop_windows_kernel): nvBowtie's opposite-mate tracebacks.  Results are those of the plain forms; a job whose cropped DP does
// not reproduce the premise (another score, or a sink off the last row) is traced again over its whole window (verify_known_kernel).
NVB_API int nvbio_hip_gotoh_traceback_known_score(
    const nvbio_hip_gotoh_scheme* scheme, int32_t type,
    const nvbio_hip_string_set* patterns, const nvbio_hip_string_set* texts, const int32_t* known_score,
    uint32_t max_pattern_len, uint32_t max_text_len, uint32_t n,
    int32_t* out_score, uint32_t* out_sink, uint32_t* out_source,
    uint16_t* out_cigar, uint32_t cigar_stride, uint32_t* out_cigar_len,
    void* temp, uint64_t temp_bytes, void* stream)
{
    if (n != 0 && !known_score) return hipErrorInvalidValue;
    return full_traceback_core(scheme, nullptr, type, 8u, patterns, texts, max_pattern_len, max_text_len, n, out_score, out_sink, out_source,
                               out_cigar, cigar_stride, out_cigar_len, temp, temp_bytes, stream, known_score);
}

NVB_API int nvbio_hip_gotoh_traceback_qual_known_score(
    const nvbio_hip_gotoh_qual_scheme* scheme, int32_t type,
    const nvbio_hip_string_set* patterns, const uint8_t* quals, uint64_t n_quals, const nvbio_hip_string_set* texts, const int32_t* known_score,
    uint32_t max_pattern_len, uint32_t max_text_len, uint32_t n,
    int32_t* out_score, uint32_t* out_sink, uint32_t* out_source,
    uint16_t* out_cigar, uint32_t cigar_stride, uint32_t* out_cigar_len,
    void* temp, uint64_t temp_bytes, void* stream)
{
    if (!scheme) return hipErrorInvalidValue;
    if (n != 0 && (!quals || n_quals == 0 || !known_score)) return hipErrorInvalidValue;
    int32_t worst = 0;
    for (int i = 0; i < 256; ++i) worst = std::min(worst, scheme->mismatch[i]);
    const nvbio_hip_gotoh_scheme g = { scheme->match, worst, scheme->pattern_gap_open, scheme->pattern_gap_ext };
    const TbQualPart q = { quals, n_quals, scheme->mismatch, scheme->text_gap_open, scheme->text_gap_ext };
    return full_traceback_core(&g, &q, type, 8u, patterns, texts, max_pattern_len, max_text_len, n, out_score, out_sink, out_source,
                               out_cigar, cigar_stride, out_cigar_len, temp, temp_bytes, stream, known_score);
}
