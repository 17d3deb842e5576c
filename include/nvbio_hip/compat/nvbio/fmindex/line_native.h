// compat/nvbio/fmindex/line_native.h -- the MI355X line-native FM-index under the reference's per-thread fm_index functions.
//
// nvBowtie's own kernels (map_queues_kernel, locate_init_kernel ...; mapping_inl.h:83-97, locate_inl.h:53-115) walk the index one symbol at a
// time, one lane per read, through rank(fmi, range, c) and locate_ssa_iterator(fmi, row).  On the reference layout every such step is a
// dependent gather of one or two 32-byte records -- a 128-byte fabric request each on this chip -- and the kernels are bound by the round trips.
// io::FMIndexDataDevice therefore also keeps the library's line-native index (nvbio_hip_fm_build_dimer_index, layout in
// nvbio_amd/csrc/fmindex_dimer.h: per 128 SA rows ONE 128-byte record of 16 two-symbol counters and four bit-planes of the rows' two-symbol
// BWT) and hands its address to the fm_index it gives out.  With it:
//   * rank(fmi, range, c) reads the one or two lines of the range ends and answers the call; from the SAME lines it also has the answer of the
//     NEXT backward-search step for each of the four symbols that may follow (a two-symbol count is a counter plus a popcount of four planes),
//     and leaves them in the index object the caller holds by value.  The next rank(fmi, range', c') whose range' is the one this step produced
//     returns from registers: every other step of a backward search costs no memory access.  The kept values are a pure function of
//     (range', c'), so whichever call finds them gets what the reference layout would have given.
//   * locate_ssa_iterator walks two text positions per line (the row's own two symbols and both counts sit in its record).
// Results are bit-identical to the reference-layout code of fmindex.h (tests/compat/fm_callers.hip runs both over the same queries).
// Record layout (dwords of record k = rows 128k .. 128k+127; header line first):
//   header[2] primary, [3] p1 (row whose suffix starts at text position 1), [4] fill1 (nibble stored there), [8..11] S[c], [12..15] T[c]
//   rec[0..15]   cnt[b*4 + a] = C2[ab] + #{rows < 128k holding the nibble (a,b)}, fillers counted; the four counters of one b are one uint4
//   rec[16..31]  plane p (bit p of the nibble a*4+b) of the 128 rows: dwords 16+4p .. 19+4p, row r at bit r&31 of dword r>>5
#pragma once
#include "../basic/types.h"

namespace nvbio {
namespace priv {

/// what an fm_index carries besides the reference's members: nothing, unless its coordinates are 32-bit
template <typename index_type> struct native_side
{
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE native_side() {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool attached() const { return false; }
};

#if defined(__HIPCC__)
struct native_planes { uint64 lo[4], hi[4]; };

NVBIO_FORCEINLINE __device__ uint32 native_prefix(const uint64 mlo, const uint64 mhi, const uint32 w)      // set bits among the first w (0..128) rows
{
    const uint64 klo = w >= 64u ? ~uint64(0) : ((uint64(1) << w) - 1u);
    const uint64 khi = w <= 64u ? uint64(0) : (w >= 128u ? ~uint64(0) : ((uint64(1) << (w - 64u)) - 1u));
    return uint32(__popcll(mlo & klo)) + uint32(__popcll(mhi & khi));
}
NVBIO_FORCEINLINE __device__ uint32 native_pick(const uint4 q, const uint32 i) { return (i & 2u) ? ((i & 1u) ? q.w : q.z) : ((i & 1u) ? q.y : q.x); }

/// one record, as far as a step needs it: the four counters of symbol b and the planes
struct native_record
{
    uint4         kb;
    native_planes P;
    NVBIO_FORCEINLINE __device__ void load(const uint32* base, const uint32 k, const uint32 b, const int n_planes)
    {
        const uint4* q = reinterpret_cast<const uint4*>(base + 32u + 32ull * k);
        kb = q[b];
        #pragma unroll
        for (int p = 0; p < 4; ++p)
            if (p < n_planes) { const uint4 v = q[4 + p]; P.lo[p] = (uint64(v.y) << 32) | v.x; P.hi[p] = (uint64(v.w) << 32) | v.z; }
    }
    /// rows below w whose second symbol is b
    NVBIO_FORCEINLINE __device__ uint32 count_b(const uint32 b, const uint32 w) const
    {
        const uint64 s0 = (b & 1u) ? uint64(0) : ~uint64(0), s1 = (b & 2u) ? uint64(0) : ~uint64(0);
        return native_prefix((P.lo[0] ^ s0) & (P.lo[1] ^ s1), (P.hi[0] ^ s0) & (P.hi[1] ^ s1), w);
    }
    /// rows below w holding the nibble v = a*4+b
    NVBIO_FORCEINLINE __device__ uint32 count_ab(const uint32 v, const uint32 w) const
    {
        const uint64 s0 = (v & 1u) ? uint64(0) : ~uint64(0), s1 = (v & 2u) ? uint64(0) : ~uint64(0);
        const uint64 s2 = (v & 4u) ? uint64(0) : ~uint64(0), s3 = (v & 8u) ? uint64(0) : ~uint64(0);
        return native_prefix((P.lo[0] ^ s0) & (P.lo[1] ^ s1) & (P.lo[2] ^ s2) & (P.lo[3] ^ s3),
                             (P.hi[0] ^ s0) & (P.hi[1] ^ s1) & (P.hi[2] ^ s2) & (P.hi[3] ^ s3), w);
    }
};
#endif

template <> struct native_side<uint32>
{
    static const uint32 NONE = 0xFFFFFFFEu;      // no caller passes -2 as a range end

    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE native_side() : base(NULL), full_sa(NULL), next_lo(NONE), next_hi(NONE) {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool attached() const { return base != NULL; }

    const uint32* base;                  ///< header line of the line-native buffer (device memory); NULL = reference layout only
    /// the whole suffix array in device memory (full_sa[row] = SA[row], full_sa[0] = 0xFFFFFFFF like ssa[0]; NULL = none): 4 bytes per row of the
    /// 288 GB (12 GB at 3 Gbp).  locate_ssa_iterator then answers with the iterator (0, SA[row] + 1) -- row 0 is sampled and holds -1, so
    /// lookup_ssa_iterator returns ssa[0] + SA[row] + 1 = SA[row], the value the LF walk to any other sampled row would have produced
    const uint32* full_sa;
    // the step kept for the next call: rank(fmi, (next_lo, next_hi), a) = (keep_lo[a] - L2[a], keep_hi[a] - L2[a]) wherever keep_lo[a] < keep_hi[a].
    // The reference's fm_index is a read-only view that several threads may share (in __shared__ memory, behind a pointer, in a struct a block
    // holds); this cache is written by rank(const fm_index&, ...).  It is therefore used ONLY when the object is the calling lane's own -- in
    // private memory, i.e. a by-value copy in registers / scratch, which is how nvBowtie's kernels hold theirs (thread_private()): an fm_index
    // in LDS or global memory never has these fields read or written, and every rank() on it reads its lines afresh.
    mutable uint32 next_lo, next_hi;
    mutable uint32 keep_lo[4], keep_hi[4];

#if defined(__HIPCC__)
    /// whether this object lives in the calling lane's private memory (folds to a constant wherever the compiler knows the object's address space,
    /// which it does for every by-value copy; a run-time aperture test for an object reached through a generic pointer)
#if defined(__HIP_DEVICE_COMPILE__)
    NVBIO_FORCEINLINE __device__ bool thread_private() const { return __builtin_amdgcn_is_private(static_cast<const void*>(this)); }
#else
    NVBIO_FORCEINLINE __device__ bool thread_private() const { return false; }          // (the host pass of a HIP translation unit only parses this)
#endif
    NVBIO_FORCEINLINE __device__ uint32 primary() const { return base[2]; }
    NVBIO_FORCEINLINE __device__ uint32 p1()      const { return base[3]; }
    NVBIO_FORCEINLINE __device__ uint32 fill1()   const { return base[4]; }
    NVBIO_FORCEINLINE __device__ uint32 S(const uint32 c) const { return base[8u + c]; }
    NVBIO_FORCEINLINE __device__ uint32 T(const uint32 c) const { return base[12u + c]; }
    NVBIO_FORCEINLINE __device__ uint32 filler(const uint32 e, const uint32 v) const
    { return ((v == 0u && e > primary()) ? 1u : 0u) + ((v == fill1() && e > p1()) ? 1u : 0u); }

    /// rank(fmi, (lo, hi), c) without its L2 terms: occurrences of c among SA rows [0, lo] and [0, hi] ('$' row not counted), lo / hi in [-1, n].
    /// Also keeps, for each symbol a, the counts the following step by a would need.
    NVBIO_FORCEINLINE __device__ uint2 step(const uint32 lo, const uint32 hi, const uint32 c, const uint32 L2c) const
    {
        const uint32 e0 = lo + 1u, e1 = hi + 1u;                 // exclusive row bounds
        native_record r0, r1;
        r0.load(base, e0 >> 7, c, 4);
        if ((e1 >> 7) != (e0 >> 7)) r1.load(base, e1 >> 7, c, 4); else r1 = r0;
        const uint32 w0 = e0 & 127u, w1 = e1 & 127u;
        const uint32 pr = primary();
        const uint32 t = T(c);
        const uint32 c0 = t + r0.kb.x + r0.kb.y + r0.kb.z + r0.kb.w + r0.count_b(c, w0) - ((c == 0u && e0 > pr) ? 1u : 0u);
        const uint32 c1 = t + r1.kb.x + r1.kb.y + r1.kb.z + r1.kb.w + r1.count_b(c, w1) - ((c == 0u && e1 > pr) ? 1u : 0u);
        // the range this step produces is [L2c + c0 + 1, L2c + c1]: the caller's next query, if it goes on, is (L2c + c0, L2c + c1)
        if (thread_private())
        {
            next_lo = L2c + c0; next_hi = L2c + c1;
            #pragma unroll
            for (uint32 a = 0; a < 4u; ++a)
            {
                const uint32 v = a * 4u + c;
                keep_lo[a] = native_pick(r0.kb, a) + r0.count_ab(v, w0) - filler(e0, v);
                keep_hi[a] = native_pick(r1.kb, a) + r1.count_ab(v, w1) - filler(e1, v);
            }
        }
        return make_uint2(c0, c1);
    }
    /// the kept step, if (lo, hi) is the range it was kept for and the two-symbol range is not empty (an empty one says nothing about the raw
    /// counts the reference returns, fmindex_dimer.h)
    NVBIO_FORCEINLINE __device__ bool kept(const uint32 lo, const uint32 hi, const uint32 a, const uint32 L2a, uint2& out) const
    {
        if (!thread_private()) return false;                     // a shared object: nothing was kept, nothing is read
        if (lo != next_lo || hi != next_hi) return false;
        const uint32 k0 = a <= 1u ? (a == 0u ? keep_lo[0] : keep_lo[1]) : (a == 2u ? keep_lo[2] : keep_lo[3]);
        const uint32 k1 = a <= 1u ? (a == 0u ? keep_hi[0] : keep_hi[1]) : (a == 2u ? keep_hi[2] : keep_hi[3]);
        next_lo = next_hi = NONE;                                // one use: the step after this one reads its own lines
        if (k0 + 1u > k1) return false;
        out = make_uint2(k0 - L2a, k1 - L2a);
        return true;
    }
    /// locate_ssa_iterator's walk from an unsampled row j: one record, up to two text positions (fmindex_inl.h:511-545 run twice).  `sampled(row)` is
    /// the suffix array's own test; `S_of(b)` = L2[b] - sum_a C2[a,b] comes from the header
    template <typename sampled_test>
    NVBIO_FORCEINLINE __device__ void locate_step(uint32& j, uint32& t, const sampled_test& sampled) const
    {
        const uint32 pr = primary();
        if (j == pr) { j = 0u; ++t; return; }                    // SA = 0 wraps to row 0 (fmindex_inl.h:534-538)
        const uint32* rec = base + 32u + 32ull * (j >> 7);
        const uint4* q = reinterpret_cast<const uint4*>(rec);
        native_record r;
        #pragma unroll
        for (int p = 0; p < 4; ++p) { const uint4 v = q[4 + p]; r.P.lo[p] = (uint64(v.y) << 32) | v.x; r.P.hi[p] = (uint64(v.w) << 32) | v.z; }
        const uint32 rr = j & 127u, sh = rr & 63u;
        const bool up = rr >= 64u;
        const uint32 b = uint32(((up ? r.P.hi[0] : r.P.lo[0]) >> sh) & 1u) | (uint32(((up ? r.P.hi[1] : r.P.lo[1]) >> sh) & 1u) << 1);
        const uint32 a = uint32(((up ? r.P.hi[2] : r.P.lo[2]) >> sh) & 1u) | (uint32(((up ? r.P.hi[3] : r.P.lo[3]) >> sh) & 1u) << 1);
        r.kb = q[b];
        const uint32 w = rr + 1u;                                // rows <= j of this block
        const uint32 j1 = S(b) + r.kb.x + r.kb.y + r.kb.z + r.kb.w + r.count_b(b, w) - ((b == 0u && j > pr) ? 1u : 0u);
        if (sampled(j1)) { j = j1; ++t; return; }
        if (j1 == pr) { j = 0u; t += 2u; return; }               // SA[j] = 1: the next step is the wrap
        const uint32 v = a * 4u + b;
        j = native_pick(r.kb, a) + r.count_ab(v, w) - filler(j + 1u, v);
        t += 2u;
    }
#endif
};

} // namespace priv
} // namespace nvbio
