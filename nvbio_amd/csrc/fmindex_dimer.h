// fmindex_dimer.h -- the line-native FM-index of this build: one 128-byte record per 128 SA rows,
// holding a TWO-symbol BWT, so that a backward-search step consumes two pattern symbols -- and an
// LF step of locate walks two text positions -- per HBM line.
//
// Why: on MI355X every L2 miss is a 128-byte fabric request, whatever part of the line is used
// (profiles/r01/pmc_fm_rank.md), and random gathers top out at ~53 G lines/s.  The reference's
// 32-byte records (nvbio/io/fmindex/fmindex_impl.cu:305-322) therefore pay a whole line per
// symbol and per range end.  Results are the contract, the layout is not: this index returns,
// bit for bit, what match / rank / locate return on the reference layout
// (nvbio/fmindex/fmindex_inl.h:36-99, 307-341, 466-545), in about half the line fetches.
//
// Definitions.  Rows i = 0..n of the suffix array (row 0 = the empty '$' suffix, SA[primary] = 0).
// Row i stores the dimer (a_i, b_i) = (T[SA[i]-2], T[SA[i]-1]) as the nibble a*4+b; b_i is the
// ordinary BWT symbol.  Two rows have no complete dimer and hold fillers: row `primary` (SA = 0:
// neither symbol exists, stored AA) and row `p1` (SA = 1: a does not exist, stored (A, T[0])).
// Queries subtract the fillers again (two compares), so no row is ever shifted: unlike the
// reference's BWT, which drops the '$' row, all n+1 rows are addressed directly.
//
// Two structures live in the buffer, each shaped by what limits its consumer on this chip:
//
// (1) per-dimer rank arrays -- the backward-search step.  PMC on a first version that read the plane records
//     below (5 loads per range end) showed the kernel bound by the texture addresser, not by HBM: a vector load
//     whose 64 lanes touch 64 different lines occupies the CU's TA/L1 for ~64-90 cycles, whatever its width
//     (profiles/r02/pmc_dimer_match.txt: TA busy 96 %, 35 G lines/s).  So a step end must cost ONE load:
//       pd[ab][r] (uint4)  = { C2[ab] + #{ rows < 96r holding the dimer ab },  96-bit mask of rows 96r..96r+95 holding ab }
//     with C2[ab] = (first row whose suffix starts with "ab") - 1 folded in, fillers and rows past n never set.
//     D(e) = counter + popcount(mask below e - 96r) is one 16-byte load; a 128-byte line covers 768 rows of one
//     dimer, so the two ends of a range share a line from the sixth step of a 3 Gbp search on.  2.67 bytes per row.
//
// (2) plane records -- locate, single-symbol steps, rank4.  Record k (rows 128k .. 128k+127), 32 dwords:
//   [ 0..15]  cnt[b*4+a] = C2[ab] + #{ rows < 128k holding the nibble (a,b) }     (fillers counted)
//             The four counters of one b are one aligned uint4.
//   [16..31]  four bit-planes of the 128 nibbles (plane p = bit p of the nibble, row r of the
//             block at bit r&31 of dword r>>5): counting a dimer is and/andn over planes + popcount,
//             counting a single symbol uses the two b planes only.  Locate reads the row's own nibble from the
//             planes, then the one uint4 of counters its b selects (same line): 5 loads per two text positions.
// One extra record past the last row lets an exclusive end e = n+1 address block e>>7 (e/96) unconditionally.
// Backward step by a dimer, pattern "abP" from the inclusive range [x,y] of P:
//   x' = D(x) + 1,  y' = D(y+1),   D(e) = pd[ab][e/96].counter + popc(mask bits below e%96)
// and by one symbol c (odd lengths, N handling, the one-mismatch enumerations):
//   x' = R(x) + 1,  y' = R(y+1),   R(e) = S[c] + sum_a cnt[e>>7][a,c] + popc(b == c, rows < e) - [c == 0 and primary < e]
// with S[c] = L2[c] - sum_a C2[a,c] (mod 2^32; every true value fits 32 bits, so wrap-around is harmless).
// A non-empty SA range is unique, so a non-empty result of a dimer step IS the result of the two
// single steps.  An empty one is replayed with single steps, because the reference returns the raw
// (x,y) of the step at which the range became empty (fmindex_inl.h:326-341) and stops there.
#pragma once
#include "common.h"

namespace nvb {

enum { DIMER_MAGIC = 0x44694D32 };      // "DiM2"

// header line (32 dwords) at the start of the dimer buffer
struct DimerHeader {
    uint32_t magic, length, primary, p1;
    uint32_t fill1;          // nibble stored at row p1 (a = 0, b = T[0])
    uint32_t n_records;      // plane records
    uint32_t pd_stride;      // records per per-dimer array; the arrays start at dword 32 + 32 * n_records
    uint32_t L2_chk;         // L2[1] ^ rotl(L2[2], 11) of the index it was built from (identity check on attach)
    uint32_t S[4];           // L2[c] - sum_a C2[a,c]
    uint32_t T[4];           // -sum_a C2[a,c]          (rank4: counts without L2)
    uint32_t C2[16];         // slot b*4+a, as folded into the counters
};
static_assert(sizeof(DimerHeader) == 128, "dimer header is one line");

struct Dimer {
    const uint32_t* base;    // header line, then plane records, then the per-dimer arrays; nullptr = not attached
    const uint4*    pd;      // per-dimer arrays: pd[ab * pd_stride + r]
    uint32_t pd_stride;
    uint32_t primary, p1, fill1;
    uint32_t S[4], T[4];
    __device__ __forceinline__ const uint32_t* rec(uint32_t k) const { return base + 32u + 32ull * k; }
};
__device__ __forceinline__ uint32_t dm_sel4(const uint32_t (&v)[4], const uint32_t c)
{
    return c <= 1u ? (c == 0u ? v[0] : v[1]) : (c == 2u ? v[2] : v[3]);
}

struct Planes { uint64_t lo[4], hi[4]; };     // plane p: rows 0..63 in lo[p], 64..127 in hi[p]

__device__ __forceinline__ void dm_load_planes(const uint32_t* r, Planes& P, const int first, const int last)
{
    const uint4* q = reinterpret_cast<const uint4*>(r + 16);
    #pragma unroll
    for (int p = first; p <= last; ++p) {
        const uint4 v = q[p];
        P.lo[p] = (uint64_t(v.y) << 32) | v.x;
        P.hi[p] = (uint64_t(v.w) << 32) | v.z;
    }
}

// number of set bits among the first w (0..128) rows of a 128-bit row mask
__device__ __forceinline__ uint32_t dm_prefix_count(const uint64_t mlo, const uint64_t mhi, const uint32_t w)
{
    const uint64_t klo = w >= 64u ? ~0ull : ((1ull << w) - 1ull);
    const uint64_t khi = w <= 64u ? 0ull : (w >= 128u ? ~0ull : ((1ull << (w - 64u)) - 1ull));
    return __popcll(mlo & klo) + __popcll(mhi & khi);
}

// rows of the block whose b (planes 0,1) equals c
__device__ __forceinline__ void dm_match_b(const Planes& P, const uint32_t c, uint64_t& mlo, uint64_t& mhi)
{
    const uint64_t s0 = (c & 1u) ? 0ull : ~0ull, s1 = (c & 2u) ? 0ull : ~0ull;
    mlo = (P.lo[0] ^ s0) & (P.lo[1] ^ s1);
    mhi = (P.hi[0] ^ s0) & (P.hi[1] ^ s1);
}
// rows of the block whose nibble equals v = a*4+b
__device__ __forceinline__ void dm_match_ab(const Planes& P, const uint32_t v, uint64_t& mlo, uint64_t& mhi)
{
    const uint64_t s0 = (v & 1u) ? 0ull : ~0ull, s1 = (v & 2u) ? 0ull : ~0ull;
    const uint64_t s2 = (v & 4u) ? 0ull : ~0ull, s3 = (v & 8u) ? 0ull : ~0ull;
    mlo = (P.lo[0] ^ s0) & (P.lo[1] ^ s1) & (P.lo[2] ^ s2) & (P.lo[3] ^ s3);
    mhi = (P.hi[0] ^ s0) & (P.hi[1] ^ s1) & (P.hi[2] ^ s2) & (P.hi[3] ^ s3);
}

// ---------------------------------------------------------------------------- single-symbol counting
// R(e) - S[c] + filler term, i.e. the part that needs the record: sum of the 4 counters of c plus the
// in-block prefix count.  `e` is an exclusive row bound in [0, n+1].
__device__ __forceinline__ uint32_t dm_rec_count1(const uint32_t* r, const uint32_t w, const uint32_t c)
{
    const uint4 k = reinterpret_cast<const uint4*>(r)[c];
    Planes P;
    dm_load_planes(r, P, 0, 1);
    uint64_t mlo, mhi;
    dm_match_b(P, c, mlo, mhi);
    return k.x + k.y + k.z + k.w + dm_prefix_count(mlo, mhi, w);
}

// one backward-search step by symbol c: the reference's rank(fmi, (x-1, y), c) + L2(c) bookkeeping
// (fmindex_inl.h:333-339).  When every lane of the wave has both ends in one block (the narrow ranges of
// the late steps) the record is loaded once; the test is wave-uniform so a wave never runs both forms.
__device__ __forceinline__ uint2 dm_step1(const Dimer& d, const uint32_t x, const uint32_t y, const uint32_t c)
{
    const uint32_t ex = x, ey = y + 1u;
    const uint32_t s = dm_sel4(d.S, c);
    const uint32_t fx = (c == 0u && ex > d.primary) ? 1u : 0u;
    const uint32_t fy = (c == 0u && ey > d.primary) ? 1u : 0u;
    uint32_t rx, ry;
    if (__builtin_amdgcn_ballot_w64((ex >> 7) != (ey >> 7)) == 0ull)
    {
        const uint32_t* r = d.rec(ex >> 7);
        const uint4 k = reinterpret_cast<const uint4*>(r)[c];
        Planes P;
        dm_load_planes(r, P, 0, 1);
        uint64_t mlo, mhi;
        dm_match_b(P, c, mlo, mhi);
        const uint32_t base = k.x + k.y + k.z + k.w;
        rx = base + dm_prefix_count(mlo, mhi, ex & 127u);
        ry = base + dm_prefix_count(mlo, mhi, ey & 127u);
    }
    else
    {
        rx = dm_rec_count1(d.rec(ex >> 7), ex & 127u, c);
        ry = dm_rec_count1(d.rec(ey >> 7), ey & 127u, c);
    }
    return make_uint2(s + rx - fx + 1u, s + ry - fy);
}

// counts of all four symbols among rows < e, WITHOUT L2 (the reference's rank4, fmindex_inl.h:111-186)
__device__ __forceinline__ uint4 dm_rank4(const Dimer& d, const uint32_t e)
{
    const uint32_t* r = d.rec(e >> 7);
    const uint32_t w = e & 127u;
    const uint4* q = reinterpret_cast<const uint4*>(r);
    const uint4 k0 = q[0], k1 = q[1], k2 = q[2], k3 = q[3];
    Planes P;
    dm_load_planes(r, P, 0, 1);
    uint64_t mlo, mhi;
    uint4 o;
    dm_match_b(P, 0u, mlo, mhi); o.x = d.T[0] + k0.x + k0.y + k0.z + k0.w + dm_prefix_count(mlo, mhi, w) - (e > d.primary ? 1u : 0u);
    dm_match_b(P, 1u, mlo, mhi); o.y = d.T[1] + k1.x + k1.y + k1.z + k1.w + dm_prefix_count(mlo, mhi, w);
    dm_match_b(P, 2u, mlo, mhi); o.z = d.T[2] + k2.x + k2.y + k2.z + k2.w + dm_prefix_count(mlo, mhi, w);
    dm_match_b(P, 3u, mlo, mhi); o.w = d.T[3] + k3.x + k3.y + k3.z + k3.w + dm_prefix_count(mlo, mhi, w);
    return o;
}

// ---------------------------------------------------------------------------- dimer counting
__device__ __forceinline__ uint32_t dm_filler(const Dimer& d, const uint32_t e, const uint32_t v)
{
    return ((v == 0u && e > d.primary) ? 1u : 0u) + ((v == d.fill1 && e > d.p1) ? 1u : 0u);
}

// rows below w (0..95) set in the 96-bit mask {y,z,w} of a per-dimer record
__device__ __forceinline__ uint32_t pd_prefix_count(const uint4 r, const uint32_t w)
{
    const uint64_t lo = (uint64_t(r.z) << 32) | r.y;
    const uint64_t klo = w >= 64u ? ~0ull : ((1ull << w) - 1ull);
    const uint32_t khi = w <= 64u ? 0u : ((1u << (w - 64u)) - 1u);
    return __popcll(lo & klo) + __popc(r.w & khi);
}

// one backward-search step by a k-mer whose rank array is `arr` (records of {counter, 96-bit mask}): the range of "kP" from
// the range [x,y] of P.  One 16-byte load per range end (one in all when the wave's ranges are narrow enough to share a record).
__device__ __forceinline__ uint2 dm_step_array(const uint4* arr, const uint32_t x, const uint32_t y)
{
    const uint32_t ex = x, ey = y + 1u;
    const uint32_t qx = __umulhi(ex, 0xAAAAAAABu) >> 6, qy = __umulhi(ey, 0xAAAAAAABu) >> 6;       // e / 96
    uint32_t rx, ry;
    if (__builtin_amdgcn_ballot_w64(qx != qy) == 0ull)
    {
        const uint4 r = arr[qx];
        rx = r.x + pd_prefix_count(r, ex - qx * 96u);
        ry = r.x + pd_prefix_count(r, ey - qx * 96u);
    }
    else
    {
        const uint4 r0 = arr[qx], r1 = arr[qy];
        rx = r0.x + pd_prefix_count(r0, ex - qx * 96u);
        ry = r1.x + pd_prefix_count(r1, ey - qy * 96u);
    }
    return make_uint2(rx + 1u, ry);
}
// by the dimer (a,b): pattern "abP"
__device__ __forceinline__ uint2 dm_step2(const Dimer& d, const uint32_t x, const uint32_t y, const uint32_t a, const uint32_t b)
{
    return dm_step_array(d.pd + uint64_t(a * 4u + b) * d.pd_stride, x, y);
}

// ---------------------------------------------------------------------------- three symbols per step
// The same per-k-mer rank arrays for the 64 trimers (10.7 bytes per row: 32 GB at 3 Gbp -- a ninth of one MI355X's HBM): a
// 22-bp seed is 7 trimer steps + 1 symbol instead of 11 dimer steps, and what bounds the search is L2 misses per seed
// (fmindex_trimer.hip builds them; trimer (a,b,c) = pattern "abcP", code a*16 + b*4 + c, C3 folded into the counters).
struct Trimer {
    const uint4* pk;         // pk[code * stride + r]; nullptr = not attached
    uint32_t     stride;
};
__device__ __forceinline__ uint2 tm_step3(const Trimer& t, const uint32_t x, const uint32_t y, const uint32_t a, const uint32_t b, const uint32_t c)
{
    return dm_step_array(t.pk + uint64_t(a * 16u + b * 4u + c) * t.stride, x, y);
}

__device__ __forceinline__ uint32_t dm_sel4v(const uint4 q, const uint32_t i)
{
    return (i & 2u) ? ((i & 1u) ? q.w : q.z) : ((i & 1u) ? q.y : q.x);
}

// ---------------------------------------------------------------------------- locate
// locate_ssa_iterator (fmindex_inl.h:511-545): LF-walk from row j to the next sampled row, two text
// positions per record: the row's own nibble gives both symbols; the single step's landing row j1 is
// computed from the same record and tested against the sampling mask exactly where the reference tests
// it, so the iterator (row, steps) is the reference's.
// One iteration from an unsampled row j ((j & sa_mask) != 0): afterwards j is either sampled (the walk is over)
// or two text positions further.
__device__ __forceinline__ void dm_locate_step(const Dimer& d, uint32_t& j, uint32_t& t, const uint32_t sa_mask)
{
    if (j == d.primary) { j = 0u; ++t; return; }             // fmindex_inl.h:534-538: SA = 0 wraps to row 0 (always sampled)
    const uint32_t* r = d.rec(j >> 7);
    Planes P;
    dm_load_planes(r, P, 0, 3);
    const uint32_t rr = j & 127u;
    const uint32_t sh = rr & 63u;
    const bool up = rr >= 64u;
    const uint32_t b = uint32_t(((up ? P.hi[0] : P.lo[0]) >> sh) & 1u) | (uint32_t(((up ? P.hi[1] : P.lo[1]) >> sh) & 1u) << 1);
    const uint32_t a = uint32_t(((up ? P.hi[2] : P.lo[2]) >> sh) & 1u) | (uint32_t(((up ? P.hi[3] : P.lo[3]) >> sh) & 1u) << 1);
    // the four counters of b -- among them the dimer's own -- are one uint4 of the line just fetched
    // (loading all four counter groups up front, to spare the dependent load, is slower -- 12.1 vs 10.5 ms: the step is bound by
    // its count of scattered load instructions, profiles/r02/locate_refill.txt)
    const uint4 kb = reinterpret_cast<const uint4*>(r)[b];
    const uint32_t w = rr + 1u;                                // rows <= j of this block
    uint64_t mlo, mhi;
    // single step: j1 = L2[b] + #{rows <= j, != primary : b}
    dm_match_b(P, b, mlo, mhi);
    const uint32_t j1 = dm_sel4(d.S, b) + kb.x + kb.y + kb.z + kb.w + dm_prefix_count(mlo, mhi, w) - ((b == 0u && j > d.primary) ? 1u : 0u);
    if ((j1 & sa_mask) == 0u) { j = j1; ++t; return; }
    if (j1 == d.primary) { j = 0u; t += 2u; return; }        // SA[j] = 1: the next step is the wrap
    // second step folded in: j2 = C2[ab] + #{rows <= j holding (a,b)} - fillers
    const uint32_t v = a * 4u + b;
    dm_match_ab(P, v, mlo, mhi);
    j = dm_sel4v(kb, a) + dm_prefix_count(mlo, mhi, w) - dm_filler(d, j + 1u, v);
    t += 2u;
}

// (Two interleaved walks per lane were measured too: 13.6 ms against 11.7 ms for 50 M rows on 3 Gbp -- the walk is bound by the
// rate of line requests and of scattered load instructions, not by what is in flight; profiles/r02/locate_refill.txt.)

} // namespace nvb
