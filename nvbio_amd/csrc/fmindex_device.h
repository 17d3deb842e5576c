// fmindex_device.h -- per-lane FM-index device functions (rank / range-rank / rank4 / match / locate
// iterator) shared by the batch kernels in fmindex.hip and mapping.hip.  See fmindex.hip for the
// layout and design notes.
#pragma once
#include "common.h"
#include "fmindex_dimer.h"

namespace nvb {

struct Fmi {
    uint32_t        length, primary;
    uint32_t        L2[5];
    uint32_t        sa_int;
    const uint4*    rec;        // 2 x uint4 per block: bwt words, occ counters
    const uint32_t* ssa;
    const uint2*    ktab;       // optional: match range of every ktab_k-mer
    uint32_t        ktab_k;
    Dimer           dm;         // optional: the line-native two-symbol index (fmindex_dimer.h); base == nullptr = none
    Trimer          tm;         // optional: the three-symbol rank arrays on top of it; pk == nullptr = none
};

inline Fmi make_fmi(const nvbio_hip_fmindex* h)
{
    Fmi f;
    f.length = h->length; f.primary = h->primary;
    for (int i = 0; i < 5; ++i) f.L2[i] = h->L2[i];
    f.sa_int = h->sa_int;
    f.rec = reinterpret_cast<const uint4*>(h->bwt_occ);
    f.ssa = h->ssa;
    f.ktab = reinterpret_cast<const uint2*>(h->ktab);
    f.ktab_k = h->ktab ? h->ktab_k : 0u;
    f.dm.base = h->dimer;
    {   // the per-dimer arrays follow the plane records (fmindex_dimer.h); both counts derive from the length
        const uint32_t n_rec = uint32_t((uint64_t(h->length) + 1u) >> 7) + 1u;
        f.dm.pd_stride = uint32_t((uint64_t(h->length) + 1u) / 96u) + 1u;
        f.dm.pd = h->dimer ? reinterpret_cast<const uint4*>(h->dimer + 32u + 32ull * n_rec) : nullptr;
    }
    f.dm.primary = h->primary; f.dm.p1 = h->dimer_p1; f.dm.fill1 = h->dimer_fill1;
    for (int i = 0; i < 4; ++i) { f.dm.S[i] = h->dimer_S[i]; f.dm.T[i] = h->dimer_T[i]; }
    // the trimer arrays ride on the dimer index (leftover symbols and empty-range replays use its steps)
    f.tm.pk = (h->dimer && h->trimer) ? reinterpret_cast<const uint4*>(h->trimer + 128u) : nullptr;      // past the 512-byte header
    f.tm.stride = f.dm.pd_stride;
    return f;
}

struct Record { uint4 bwt, occ; };

__device__ __forceinline__ Record load_record(const Fmi& f, uint32_t k)
{
    Record r;
    const uint4* p = f.rec + 2ull * k;
    r.bwt = p[0];
    r.occ = p[1];
    return r;
}

// bit-plane of "symbol == c" over 32 big-endian 2-bit symbols packed in 64 bits
// (hi word = first 16 symbols), one bit per symbol at the even position.
__device__ __forceinline__ uint64_t match_plane(uint64_t x, uint32_t c)
{
    const uint64_t hi = (c & 2u) ? x : ~x;
    const uint64_t lo = (c & 1u) ? x : ~x;
    return (hi >> 1) & lo & 0x5555555555555555ull;
}

// number of occurrences of c among the first `cnt` (1..64) symbols of the block
__device__ __forceinline__ uint32_t block_count(const uint4 bwt, uint32_t cnt, uint32_t c)
{
    const uint64_t a = (uint64_t(bwt.x) << 32) | bwt.y;      // symbols 0..31, symbol 0 at the top
    const uint64_t b = (uint64_t(bwt.z) << 32) | bwt.w;      // symbols 32..63
    const uint32_t ca = cnt < 32u ? cnt : 32u;
    const uint32_t cb = cnt - ca;
    // keep the top 2*ca (2*cb) bits
    const uint64_t ma = ca == 0 ? 0ull : (~0ull << (64u - 2u * ca));
    const uint64_t mb = cb == 0 ? 0ull : (~0ull << (64u - 2u * cb));
    return __popcll(match_plane(a, c) & ma) + __popcll(match_plane(b, c) & mb);
}
__device__ __forceinline__ uint32_t comp(const uint4 v, uint32_t c)
{
    return c <= 1 ? (c == 0 ? v.x : v.y) : (c == 2 ? v.z : v.w);
}

// rank_dictionary rank: occurrences of c in bwt[0..i]   (rank_dictionary_inl.h:502-513)
__device__ __forceinline__ uint32_t dict_rank(const Fmi& f, uint32_t i, uint32_t c)
{
    if (i == 0xFFFFFFFFu) return 0u;
    const Record r = load_record(f, i >> 6);
    return comp(r.occ, c) + block_count(r.bwt, (i & 63u) + 1u, c);
}

// fm_index rank  (fmindex_inl.h:36-57)
__device__ __forceinline__ uint32_t fm_rank(const Fmi& f, uint32_t k, uint32_t c)
{
    if (k == 0xFFFFFFFFu) return 0u;
    if (k == f.length)    return f.L2[c + 1] - f.L2[c];
    if (k >= f.primary) --k;
    return dict_rank(f, k, c);
}

// fm_index rank over a range  (fmindex_inl.h:66-99 over rank_dictionary_inl.h:515-538).
// Each end is resolved independently (the result of every branch of the reference equals the
// plain occurrence count of its end); the two record loads are issued together and shared
// when both ends fall in one block.
__device__ __forceinline__ uint2 fm_rank2(const Fmi& f, uint32_t x, uint32_t y, uint32_t c)
{
    const uint32_t cnt_all = f.L2[c + 1] - f.L2[c];
    // end -> (needs_record, adjusted index)
    bool nx = !(x == 0xFFFFFFFFu || x == f.length);
    bool ny = !(y == 0xFFFFFFFFu || y == f.length);
    uint32_t ax = x, ay = y;
    if (nx && ax >= f.primary) --ax;
    if (ny && ay >= f.primary) --ay;
    if (nx && ax == 0xFFFFFFFFu) nx = false;         // x == primary == 0 : nothing before it
    if (ny && ay == 0xFFFFFFFFu) ny = false;
    const uint32_t kx = ax >> 6, ky = ay >> 6;
    Record rx, ry;
    if (nx) rx = load_record(f, kx);
    if (ny) { if (nx && kx == ky) ry = rx; else ry = load_record(f, ky); }
    uint2 out;
    out.x = nx ? comp(rx.occ, c) + block_count(rx.bwt, (ax & 63u) + 1u, c) : (x == f.length ? cnt_all : 0u);
    out.y = ny ? comp(ry.occ, c) + block_count(ry.bwt, (ay & 63u) + 1u, c) : (y == f.length ? cnt_all : 0u);
    return out;
}

__device__ __forceinline__ uint4 fm_rank4(const Fmi& f, uint32_t k)
{
    if (k == 0xFFFFFFFFu) return make_uint4(0, 0, 0, 0);
    if (k == f.length)
        return make_uint4(f.L2[1] - f.L2[0], f.L2[2] - f.L2[1], f.L2[3] - f.L2[2], f.L2[4] - f.L2[3]);
    if (k >= f.primary) --k;
    const Record r = load_record(f, k >> 6);
    const uint32_t cnt = (k & 63u) + 1u;
    uint4 o = r.occ;
    o.x += block_count(r.bwt, cnt, 0);
    o.y += block_count(r.bwt, cnt, 1);
    o.z += block_count(r.bwt, cnt, 2);
    o.w = r.occ.w + cnt - (o.x - r.occ.x) - (o.y - r.occ.y) - (o.z - r.occ.z);
    return o;
}

// rank4 over a range (fmindex_inl.h:148-186): both ends resolved independently, one record load when
// they share a block
__device__ __forceinline__ void fm_rank4_range(const Fmi& f, uint32_t x, uint32_t y, uint4& lo, uint4& hi)
{
    const uint4 all = make_uint4(f.L2[1] - f.L2[0], f.L2[2] - f.L2[1], f.L2[3] - f.L2[2], f.L2[4] - f.L2[3]);
    bool nx = !(x == 0xFFFFFFFFu || x == f.length);
    bool ny = !(y == 0xFFFFFFFFu || y == f.length);
    uint32_t ax = x, ay = y;
    if (nx && ax >= f.primary) --ax;
    if (ny && ay >= f.primary) --ay;
    if (nx && ax == 0xFFFFFFFFu) nx = false;
    if (ny && ay == 0xFFFFFFFFu) ny = false;
    Record rx, ry;
    if (nx) rx = load_record(f, ax >> 6);
    if (ny) { if (nx && (ax >> 6) == (ay >> 6)) ry = rx; else ry = load_record(f, ay >> 6); }
    auto count4 = [](const Record& r, const uint32_t cnt) {
        uint4 o = r.occ;
        o.x += block_count(r.bwt, cnt, 0);
        o.y += block_count(r.bwt, cnt, 1);
        o.z += block_count(r.bwt, cnt, 2);
        o.w = r.occ.w + cnt - (o.x - r.occ.x) - (o.y - r.occ.y) - (o.z - r.occ.z);
        return o;
    };
    lo = nx ? count4(rx, (ax & 63u) + 1u) : (x == f.length ? all : make_uint4(0, 0, 0, 0));
    hi = ny ? count4(ry, (ay & 63u) + 1u) : (y == f.length ? all : make_uint4(0, 0, 0, 0));
}

// ------------------------------------------------------------------ backward search
// one step of the reference's loop body (fmindex_inl.h:333-339): the range of "cP" from the range [x,y] of P
__device__ __forceinline__ uint2 fm_step(const Fmi& f, const uint32_t x, const uint32_t y, const uint32_t c)
{
    if (f.dm.base) return dm_step1(f.dm, x, y, c);
    const uint2 r = fm_rank2(f, x - 1u, y, c);
    return make_uint2(f.L2[c] + r.x + 1u, f.L2[c] + r.y);
}
// rank4 of both ends of the step from [x,y] (counts without L2), as the one-mismatch mappers use it
__device__ __forceinline__ void fm_step4(const Fmi& f, const uint32_t x, const uint32_t y, uint4& lo, uint4& hi)
{
    if (f.dm.base) { lo = dm_rank4(f.dm, x); hi = dm_rank4(f.dm, y + 1u); }
    else fm_rank4_range(f, x - 1u, y, lo, hi);
}

// match (fmindex_inl.h:307-341) with nvBowtie's symbol test (mapping_inl.h:83-97).
// One lane = one seed; the seed is pulled 16 symbols per fetch.  With the two-symbol index attached the
// loop consumes symbol pairs: a non-empty range after a pair is the range after its two single steps; an
// empty one is replayed singly, because the reference returns the raw (x,y) of the step that emptied it.
__device__ __forceinline__ uint2 fm_match_from(const Fmi& f, const Stream& s, uint64_t begin, int32_t i, uint32_t x, uint32_t y)
{
    bool pairs = f.dm.base != nullptr;
    bool triples = pairs && f.tm.pk != nullptr;
    while (i >= 0 && x <= y)
    {
        // symbols [g0, g0+16) of the seed, g0 = 16-aligned group holding i
        const uint32_t g0 = uint32_t(i) & ~15u;
        uint64_t grp;   // 4 bits per symbol
        if (s.bits == 2)      grp = expand_2to4(fetch16_2bit(s, begin + g0));
        else                  grp = fetch16_4bit(s, begin + g0);
        while (i >= int32_t(g0) && x <= y)
        {
            const uint32_t k = uint32_t(i) - g0;
            const uint32_t c = uint32_t(grp >> (4u * k)) & 15u;
            if (c > 3u) return make_uint2(1u, 0u);
            if (pairs && k >= 1u)
            {
                const uint32_t b = uint32_t(grp >> (4u * (k - 1u))) & 15u;
                if (b <= 3u)
                {
                    if (triples && k >= 2u)
                    {
                        const uint32_t a = uint32_t(grp >> (4u * (k - 2u))) & 15u;
                        if (a <= 3u)
                        {
                            const uint2 r = tm_step3(f.tm, x, y, a, b, c);
                            if (r.x <= r.y) { x = r.x; y = r.y; i -= 3; continue; }
                            triples = false;          // empty: replay with smaller steps for the reference's raw values
                        }
                    }
                    const uint2 r = dm_step2(f.dm, x, y, b, c);
                    if (r.x <= r.y) { x = r.x; y = r.y; i -= 2; continue; }
                    pairs = false; triples = false;
                }
            }
            const uint2 r = fm_step(f, x, y, c);
            x = r.x; y = r.y; --i;
        }
    }
    return make_uint2(x, y);
}

// pack the low k nibbles (each <= 3) of v into k 2-bit fields
__device__ __forceinline__ uint32_t nibbles_to_2bit(uint64_t v)
{
    v = (v | (v >> 2)) & 0x0F0F0F0F0F0F0F0Full;
    v = (v | (v >> 4)) & 0x00FF00FF00FF00FFull;
    v = (v | (v >> 8)) & 0x0000FFFF0000FFFFull;
    v = (v | (v >> 16)) & 0x00000000FFFFFFFFull;
    return uint32_t(v);
}

__device__ __forceinline__ uint2 fm_match(const Fmi& f, const Stream& s, uint64_t begin, uint32_t len)
{
    const uint32_t k = f.ktab_k;
    if (k != 0u && len >= k)
    {
        // the last k symbols of the seed: one table lookup replaces the first k steps.  The table
        // entry IS match(k-mer), i.e. the state in which the reference's loop leaves those steps
        // (including an empty range at the step where it became empty).
        uint64_t tail = (s.bits == 2) ? expand_2to4(fetch16_2bit(s, begin + len - k)) : fetch16_4bit(s, begin + len - k);
        tail &= (k == 16u) ? ~0ull : ((1ull << (4u * k)) - 1ull);
        if ((tail & 0xCCCCCCCCCCCCCCCCull) == 0ull)          // no N among them (else: plain search)
        {
            const uint2 r = f.ktab[nibbles_to_2bit(tail)];
            if (r.x > r.y) return r;
            return fm_match_from(f, s, begin, int32_t(len - k) - 1, r.x, r.y);
        }
    }
    return fm_match_from(f, s, begin, int32_t(len) - 1, 0u, f.length);
}

// ------------------------------------------------------------------ locate
// One iteration of locate_ssa_iterator's LF walk (fmindex_inl.h:511-545) from an unsampled row j: one text
// position on the reference layout, up to two on the two-symbol index.
__device__ __forceinline__ void fm_locate_step(const Fmi& f, uint32_t& j, uint32_t& t)
{
    if (f.dm.base) { dm_locate_step(f.dm, j, t, f.sa_int - 1u); return; }
    if (j != f.primary)
    {
        // the BWT symbol of row j and its occurrence counters live in the same record
        const uint32_t k = (j < f.primary) ? j : j - 1u;
        const Record r = load_record(f, k >> 6);
        const uint32_t w = comp(r.bwt, (k & 63u) >> 4);
        const uint32_t c = (w >> (30u - ((k & 15u) << 1))) & 3u;
        j = f.L2[c] + comp(r.occ, c) + block_count(r.bwt, (k & 63u) + 1u, c);
    }
    else j = 0u;
    ++t;
}

__device__ __forceinline__ uint2 fm_locate_it(const Fmi& f, uint32_t j)
{
    uint32_t t = 0;
    const uint32_t mask = f.sa_int - 1u;
    while ((j & mask) != 0u) fm_locate_step(f, j, t);
    return make_uint2(j, t);
}

// (A lane-refill form of the batched walk -- a finished lane takes the wave's next row -- was measured and dropped:
// the reference-layout walk is already at the chip's random-line rate, and on the plane records it ran 12 % slower than
// one row per lane, profiles/r02/locate_refill.txt.)

} // namespace nvb
