// full_gotoh_striped.hip -- the full-matrix wave sweep for what the register-resident sweeps of full_gotoh.hip do not hold:
//
//  * patterns longer than 64 lanes x 16 rows.  The reference's text-blocking dispatch has no length limit
//    (nvbio/alignment/gotoh/gotoh_inl.h:969-1490, sw/sw_inl.h:881-1222: a boundary column of M int16 entries in the caller's temp
//    storage).  Here the matrix is cut into STRIPES of 1 024 pattern rows.  One wave sweeps a stripe over the whole text exactly as
//    full_gotoh_score_kernel does (lane l holds 16 rows, column c = step - l, values handed down with wave_shr:1 DPP moves); the row
//    below a stripe -- H and F of its last row and the running column maximum, per text column -- goes to HBM, 64 columns per
//    coalesced store, and comes back as "the row above" of the next stripe, 64 columns per coalesced load, one v_readlane per step.
//    A stripe reads column block b at step 64 b and writes it at step 64 b + 126, so one line of 3 x N words per wave is enough.
//    Waves are persistent (job = wave, wave + #waves, ...): the lines cost #waves x 12 N bytes whatever the batch size.
//  * direction-dependent linear gap costs (SimpleSmithWatermanScheme with deletion != insertion, alignment/utils.h:92-109):
//    the move along the text costs `deletion` (left + G, sw_inl.h:925-927), the move down the pattern `insertion` (top + I).
//
// 32-bit arithmetic, the reference's observable details as in full_gotoh.hip: LOCAL ties by the blocked visiting order (64-bit order
// keys: no bound on M x N), the int16 boundary column (H, E truncated where a column starts a block of 8 / 16, when the host cannot
// prove them inside int16), the Gotoh early exit after each full block of text columns (the column maximum crosses stripes with the
// data; when it fires every stripe runs again over the columns the reference visited).
#include "full_gotoh_striped.h"
#include <algorithm>
#include <stdlib.h>

namespace nvb {


__device__ __forceinline__ int32_t st_shr1(int32_t first_lane_value, int32_t x) { return __builtin_amdgcn_update_dpp(first_lane_value, x, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ int32_t st_sext16(int32_t v) { return int32_t(int16_t(v)); }

struct StripeResult { int32_t score; uint32_t sx, sy, exit_col; };

constexpr int SR = 16;                       // rows per lane
constexpr uint32_t STRIPE = 64u * SR;        // rows per stripe

template <int TYPE>
__device__ __forceinline__ StripeResult striped_sweep(const StripeParams& p, const uint64_t pb, const uint64_t tb, const uint32_t M, const uint32_t Ncols, const uint32_t Nfull,
                                                      const bool check, const int32_t min_score, int32_t* bnd)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t BS = p.blk_log2, BLK = 1u << BS;
    const bool trunc = p.trunc != 0u, linear = p.linear != 0u, PB = p.pattern_blocking != 0u;
    const int32_t infimum = p.infimum;

    // LOCAL: this lane's best cell over all stripes (score, order key, column, row)
    int32_t best_h = 0; uint64_t best_key = 0; uint32_t best_c = 0, best_r = 0; bool have = false;
    int32_t sg_score = -(1 << 30); uint32_t sg_col = 0, exit_col = 0xFFFFFFFFu;
    uint32_t last_lane_of_job = 0;

    for (uint32_t row0 = 0; row0 < M; row0 += STRIPE)
    {
        const uint32_t rows = min(M - row0, STRIPE);
        const uint32_t lane_last = (rows - 1u) / uint32_t(SR);
        const bool last = row0 + STRIPE >= M;
        const uint32_t first_row = row0 + lane * uint32_t(SR);
        if (last) last_lane_of_job = lane_last;

        uint32_t q[SR]; int32_t Hleft[SR], E[SR];
        #pragma unroll
        for (int k = 0; k < SR; ++k)
        {
            const uint32_t r = first_row + k;
            q[k] = r < M ? get_symbol(p.pat.s, pb + r) : 0xFFFFFFFFu;
            const int32_t h0 = (TYPE != NVBIO_HIP_LOCAL) ? p.col_go + p.col_ge * int32_t(r) : 0;
            const int32_t e0 = (TYPE == NVBIO_HIP_LOCAL) ? 0 : infimum;
            Hleft[k] = trunc ? st_sext16(h0) : h0;
            E[k]     = trunc ? st_sext16(e0) : e0;
        }
        int32_t out_h = 0, out_f = 0, out_ch = 0, out_cm = 0, prev_in_h = 0;
        int32_t bH = 0, bF = 0, bC = 0;            // the row above this stripe, columns [64 b, 64 b + 64) (lane = column in the block)
        int32_t oH = 0, oF = 0, oC = 0;            // the row below it, being collected
        uint32_t grp = 0;

        const uint32_t n_steps = Ncols + lane_last;
        for (uint32_t s = 0; s < n_steps; ++s)
        {
            if ((s & 15u) == 0u && s < Ncols) grp = fetch16_2bit(p.txt.s, tb + s);      // wave-uniform
            if (row0 != 0u && (s & 63u) == 0u && s < Ncols)
            {
                const uint32_t idx = min(s + lane, Ncols - 1u);
                bH = bnd[idx]; bF = bnd[p.bnd_stride + idx]; bC = bnd[2u * p.bnd_stride + idx];
            }
            const int32_t c_signed = int32_t(s) - int32_t(lane);
            const uint32_t c = uint32_t(c_signed);
            const bool active = c_signed >= 0 && c < Ncols && lane <= lane_last;

            // the row above lane 0: the boundary line of the matrix (first stripe) or the stripe above's last row
            int32_t top_h, top_f, top_cm;
            if (row0 == 0u) { top_h = (TYPE == NVBIO_HIP_GLOBAL) ? p.row_go + p.row_ge * int32_t(s) : 0; top_f = infimum; top_cm = -(1 << 30); }
            else            { top_h = __builtin_amdgcn_readlane(bH, int(s & 63u)); top_f = __builtin_amdgcn_readlane(bF, int(s & 63u)); top_cm = __builtin_amdgcn_readlane(bC, int(s & 63u)); }
            const int32_t ch0 = int32_t((grp >> (2u * (s & 15u))) & 3u);
            const int32_t in_h  = st_shr1(top_h, out_h);
            const int32_t in_f  = st_shr1(top_f, out_f);
            const int32_t in_ch = st_shr1(ch0, out_ch);
            const int32_t in_cm = st_shr1(top_cm, out_cm);

            // H(row above, c - 1): what came down one step ago; in a lane's first column the init column's entry of the row above (the corner: 0)
            int32_t diag = prev_in_h;
            if (c_signed == 0)
            {
                const int32_t h0 = (first_row == 0u || TYPE == NVBIO_HIP_LOCAL) ? 0 : p.col_go + p.col_ge * int32_t(first_row - 1u);
                diag = trunc ? st_sext16(h0) : h0;
            }
            const bool crossing = trunc && (c & (BLK - 1u)) == 0u && c_signed > 0;      // column c starts a block: the left values came through temp[]
            if (crossing && first_row != 0u) diag = st_sext16(diag);
            prev_in_h = in_h;

            int32_t habove = in_h, fabove = in_f, cm = in_cm;
            #pragma unroll
            for (int k = 0; k < SR; ++k)
            {
                const uint32_t r = first_row + k;
                int32_t hl = Hleft[k], e = E[k];
                if (crossing) { hl = st_sext16(hl); e = st_sext16(e); }
                const int32_t f = linear ? habove + p.f_go : max(fabove + p.f_ge, habove + p.f_go);
                e = linear ? hl + p.e_go : max(e + p.e_ge, hl + p.e_go);
                const int32_t d = diag + ((uint32_t(in_ch) == q[k]) ? p.match : p.mismatch);
                int32_t h = max(max(e, f), d);
                if (TYPE == NVBIO_HIP_LOCAL) h = max(h, 0);
                diag = hl;
                if (active)
                {
                    Hleft[k] = h; E[k] = e;
                    if (r < M)
                    {
                        cm = max(cm, h);
                        if (TYPE == NVBIO_HIP_LOCAL)
                        {
                            // text blocking: block of columns -> row -> column in block; pattern blocking: block of rows -> column -> row in block
                            const uint64_t key = PB ? ((uint64_t(r >> BS) * Ncols + c) << BS) | (r & (BLK - 1u))
                                                    : ((uint64_t(c >> BS) * M + r) << BS) | (c & (BLK - 1u));
                            if (!have || h > best_h || (h == best_h && key >= best_key)) { best_h = h; best_key = key; best_c = c; best_r = r; have = true; }
                        }
                    }
                }
                habove = h; fabove = f;
            }
            if (active) { out_h = habove; out_f = fabove; out_ch = in_ch; out_cm = cm; }

            if (last)
            {
                if (active && lane == lane_last)
                {
                    const uint32_t klast = (M - 1u - row0) - lane_last * uint32_t(SR);
                    int32_t hlast = Hleft[0];
                    #pragma unroll
                    for (int k = 1; k < SR; ++k) if (uint32_t(k) == klast) hlast = Hleft[k];
                    if (TYPE == NVBIO_HIP_SEMI_GLOBAL) { if (sg_score <= hlast) { sg_score = hlast; sg_col = c; } }
                    if (TYPE == NVBIO_HIP_GLOBAL && c + 1u == Nfull) { sg_score = hlast; sg_col = c; }
                    // the Gotoh early exit (gotoh_inl.h:1212-1214): after blocks that are not the last one
                    if (check && (c & (BLK - 1u)) == BLK - 1u && exit_col == 0xFFFFFFFFu)
                    {
                        const uint32_t nb = BLK * ((Nfull + BLK - 1u) / BLK);
                        const uint32_t end_block = nb > BLK ? nb : BLK;
                        const uint32_t block = c - (BLK - 1u);
                        if (block + BLK < end_block && cm + int32_t(Nfull - block - BLK) * p.match < min_score) exit_col = c;
                    }
                }
            }
            else if (s >= 63u && s - 63u < Ncols)
            {
                // lane 63 finished column s - 63 of the stripe's last row: collect it, store 64 columns at a time
                const uint32_t c63 = s - 63u;
                const bool mine = lane == (c63 & 63u);
                const int32_t vh = __builtin_amdgcn_readlane(out_h, 63), vf = __builtin_amdgcn_readlane(out_f, 63), vc = __builtin_amdgcn_readlane(out_cm, 63);
                oH = mine ? vh : oH; oF = mine ? vf : oF; oC = mine ? vc : oC;
                if ((c63 & 63u) == 63u || c63 + 1u == Ncols)
                {
                    const uint32_t idx = (c63 & ~63u) + lane;
                    if (idx <= c63) { bnd[idx] = oH; bnd[p.bnd_stride + idx] = oF; bnd[2u * p.bnd_stride + idx] = oC; }
                }
            }
        }
        // the next stripe's loads must see this stripe's stores (same wave, same addresses: program order suffices for the values the
        // wave itself wrote once the stores have left the wave)
        __builtin_amdgcn_s_waitcnt(0);
    }

    StripeResult res;
    res.exit_col = uint32_t(__shfl(int32_t(exit_col), int32_t(last_lane_of_job)));
    res.score = -(1 << 30); res.sx = res.sy = 0xFFFFFFFFu;
    if (TYPE == NVBIO_HIP_LOCAL)
    {
        uint32_t hv = have ? 1u : 0u;
        #pragma unroll
        for (int off = 32; off >= 1; off >>= 1)
        {
            const int32_t  oh  = __shfl_xor(best_h, off);
            const uint32_t klo = uint32_t(__shfl_xor(int32_t(uint32_t(best_key)), off)), khi = uint32_t(__shfl_xor(int32_t(uint32_t(best_key >> 32)), off));
            const uint32_t oc  = uint32_t(__shfl_xor(int32_t(best_c), off)), orr = uint32_t(__shfl_xor(int32_t(best_r), off));
            const uint32_t ohv = uint32_t(__shfl_xor(int32_t(hv), off));
            const uint64_t ok64 = (uint64_t(khi) << 32) | klo;
            if (ohv && (!hv || oh > best_h || (oh == best_h && ok64 > best_key))) { best_h = oh; best_key = ok64; best_c = oc; best_r = orr; hv = 1u; }
        }
        if (hv) { res.score = best_h; res.sx = best_c + 1u; res.sy = best_r + 1u; }
    }
    else
    {
        const int32_t  sc  = __shfl(sg_score, int32_t(last_lane_of_job));
        const uint32_t col = uint32_t(__shfl(int32_t(sg_col), int32_t(last_lane_of_job)));
        const bool reported = (TYPE == NVBIO_HIP_SEMI_GLOBAL) ? (Ncols > 0u) : (Ncols == Nfull && Nfull > 0u);
        if (reported) { res.score = sc; res.sx = col + 1u; res.sy = M; }
    }
    return res;
}

template <int TYPE>
__global__ void __launch_bounds__(256)
full_gotoh_striped_kernel(const StripeParams p)
{
    const uint32_t wave = (blockIdx.x * 256u + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t n_waves = gridDim.x * 4u;
    int32_t* const bnd = p.boundary ? p.boundary + uint64_t(wave) * 3u * p.bnd_stride : nullptr;
    const uint32_t BLK = 1u << p.blk_log2;

    for (uint32_t job = wave; job < p.n; job += n_waves)
    {
        const uint32_t M  = p.pat.length ? p.pat.length[job] : p.pat.fixed_length;
        const uint32_t N  = p.txt.length ? p.txt.length[job] : p.txt.fixed_length;
        const uint64_t pb = p.pat.begin[job], tb = p.txt.begin[job];
        const bool     check = p.min_score != nullptr;
        const int32_t  min_score = check ? p.min_score[job] : -(1 << 30);

        int32_t score = -(1 << 30); uint32_t sx = 0xFFFFFFFFu, sy = 0xFFFFFFFFu; uint32_t ok = 1u;
        if (N > p.max_n || (M > STRIPE && bnd == nullptr)) ok = 0u;         // longer than the caller stated: the failed-alignment record
        else if (M == 0u)
        {
            // no rows: only the row above the matrix is ever reported (see full_gotoh_score_kernel)
            const uint32_t xb = check ? empty_pattern_exit_block(N, BLK, p.match, min_score) : 0xFFFFFFFFu;
            const bool exits = xb != 0xFFFFFFFFu;
            if (exits) { ok = 0u; if (TYPE == NVBIO_HIP_SEMI_GLOBAL) { score = 0; sx = xb + BLK; sy = 0u; } }
            else if (N > 0u) {
                if (TYPE == NVBIO_HIP_SEMI_GLOBAL) { score = 0; sx = N; sy = 0u; }
                if (TYPE == NVBIO_HIP_GLOBAL)      { score = p.row_go + p.row_ge * int32_t(N - 1u); sx = N; sy = 0u; }
            }
        }
        else
        {
            StripeResult r = striped_sweep<TYPE>(p, pb, tb, M, N, N, check, min_score, bnd);
            if (r.exit_col != 0xFFFFFFFFu)
            {
                ok = 0u;       // the reference returned false after this block: its sink saw columns [0, exit_col] only
                r = striped_sweep<TYPE>(p, pb, tb, M, r.exit_col + 1u, N, false, min_score, bnd);
            }
            score = r.score; sx = r.sx; sy = r.sy;
            // pattern blocking, GLOBAL, empty text: save_Mth reports the initial row (gotoh_inl.h:896-897)
            if (p.pattern_blocking != 0u && TYPE == NVBIO_HIP_GLOBAL && N == 0u) { score = p.col_go + p.col_ge * int32_t(M - 1u); sx = 0u; sy = M; }
        }
        if (lane == 0u)
        {
            p.out_score[job] = score;
            reinterpret_cast<uint2*>(p.out_sink)[job] = make_uint2(sx, sy);
            if (p.out_ok) p.out_ok[job] = uint8_t(ok);
        }
    }
}

// the lines between stripes: one block per device, grown on demand, kept (a batch of long patterns is followed by another)
struct StripeLines { int32_t* ptr; uint64_t bytes; hipStream_t last; bool used; };
static thread_local StripeLines g_lines[64] = {};

hipError_t launch_striped(StripeParams& p, int type, uint32_t max_m, hipStream_t s)
{
    uint32_t waves = std::min<uint64_t>(p.n, 4096u);
    p.boundary = nullptr; p.bnd_stride = 0;
    if (max_m > STRIPE)
    {
        const uint32_t stride = (p.max_n + 63u) & ~63u;
        const uint64_t per_wave = uint64_t(stride) * 12u;
        const uint64_t budget = 2ull << 30;
        waves = uint32_t(std::max<uint64_t>(1u, std::min<uint64_t>(waves, budget / std::max<uint64_t>(per_wave, 1u))));
        waves = (waves + 3u) & ~3u;
        const uint64_t need = per_wave * waves;
        int dev = 0;
        if (hipError_t e = hipGetDevice(&dev)) return e;
        StripeLines& L = g_lines[dev & 63];
        // one set of lines per host thread and device: a launch on another stream than the last one waits for that one's sweep to finish
        if (L.used && L.last != s) { if (hipError_t e = hipDeviceSynchronize()) return e; }      // (the device, not L.last: that stream may be gone)
        L.last = s; L.used = true;
        if (L.bytes < need)
        {
            if (L.ptr) { if (hipError_t e = hipStreamSynchronize(s)) return e; nvbio_hip_device_free(L.ptr); L.ptr = nullptr; L.bytes = 0; }
            void* ptr = nullptr;
            if (int e = nvbio_hip_device_malloc(&ptr, need)) return hipError_t(e);
            L.ptr = static_cast<int32_t*>(ptr); L.bytes = need;
        }
        p.boundary = L.ptr; p.bnd_stride = stride;
    }
    const dim3 grid((waves + 3u) / 4u), block(256);
    switch (type) {
    case NVBIO_HIP_GLOBAL:      hipLaunchKernelGGL((full_gotoh_striped_kernel<NVBIO_HIP_GLOBAL>),      grid, block, 0, s, p); break;
    case NVBIO_HIP_LOCAL:       hipLaunchKernelGGL((full_gotoh_striped_kernel<NVBIO_HIP_LOCAL>),       grid, block, 0, s, p); break;
    case NVBIO_HIP_SEMI_GLOBAL: hipLaunchKernelGGL((full_gotoh_striped_kernel<NVBIO_HIP_SEMI_GLOBAL>), grid, block, 0, s, p); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

} // namespace nvb
