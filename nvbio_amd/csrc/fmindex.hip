// fmindex.hip -- FM-index rank / backward search / locate for gfx950.
//
// Index layout (caller-owned HBM, the reference's production format, used in
// place): one 32-byte record per 64 BWT symbols = 4 words of big-endian 2-bit
// BWT + 4 occurrence counters (nvbio/io/fmindex/fmindex_impl.cu:305-327).  A
// record is exactly one 32-byte HBM sector, so a rank costs one sector, a
// range-rank one or two, and an LF step of locate one (the BWT symbol and the
// counters sit in the same record).
//
// What is computed is the reference's
//   rank / rank4 / rank(range)     nvbio/fmindex/fmindex_inl.h:36-186 over
//                                  rank_dictionary_inl.h:424-573 (uint4, K=64)
//   match                          nvbio/fmindex/fmindex_inl.h:307-341
//   locate / ssa iterators         nvbio/fmindex/fmindex_inl.h:466-569
//   FMIndexFilter rank / locate    nvbio/fmindex/filter_inl.h:268-402
//   build_occurrence_table<2,64>   nvbio/fmindex/rank_dictionary_inl.h:42-77
//
// The counting itself is re-derived: instead of the reference's per-word
// branches and its 256-entry byte LUT for rank4, a record's 128 BWT bits are
// handled as two 64-bit halves: symbol-match bit-planes by and/andn, a prefix
// mask built from the in-block offset, and v_bcnt popcounts -- branch-free, no
// table, no LDS.
//
#include "common.h"
#include <hipcub/hipcub.hpp>

namespace nvb {

struct Fmi {
    uint32_t        length, primary;
    uint32_t        L2[5];
    uint32_t        sa_int;
    const uint4*    rec;        // 2 x uint4 per block: bwt words, occ counters
    const uint32_t* ssa;
    const uint2*    ktab;       // optional: match range of every ktab_k-mer
    uint32_t        ktab_k;
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

// ------------------------------------------------------------------ point queries
__global__ void __launch_bounds__(256)
fm_rank_kernel(const Fmi f, const uint32_t* __restrict__ k, const uint8_t* __restrict__ c, uint32_t n, uint32_t* __restrict__ out)
{
    const uint32_t id = blockIdx.x * 256u + threadIdx.x;
    if (id >= n) return;
    out[id] = fm_rank(f, k[id], c[id] & 3u);
}
__global__ void __launch_bounds__(256)
fm_rank4_kernel(const Fmi f, const uint32_t* __restrict__ k, uint32_t n, uint4* __restrict__ out)
{
    const uint32_t id = blockIdx.x * 256u + threadIdx.x;
    if (id >= n) return;
    out[id] = fm_rank4(f, k[id]);
}
__global__ void __launch_bounds__(256)
fm_rank_range_kernel(const Fmi f, const uint2* __restrict__ range, const uint8_t* __restrict__ c, uint32_t n, uint2* __restrict__ out)
{
    const uint32_t id = blockIdx.x * 256u + threadIdx.x;
    if (id >= n) return;
    const uint2 r = range[id];
    out[id] = fm_rank2(f, r.x, r.y, c[id] & 3u);
}

// ------------------------------------------------------------------ backward search
// match (fmindex_inl.h:307-341) with nvBowtie's symbol test (mapping_inl.h:83-97).
// One lane = one seed; the seed is pulled 16 symbols per fetch.
__device__ __forceinline__ uint2 fm_match_from(const Fmi& f, const Stream& s, uint64_t begin, int32_t i, uint32_t x, uint32_t y)
{
    while (i >= 0 && x <= y)
    {
        // symbols [g0, g0+16) of the seed, g0 = 16-aligned group holding i
        const uint32_t g0 = uint32_t(i) & ~15u;
        uint64_t grp;   // 4 bits per symbol
        if (s.bits == 2)      grp = expand_2to4(fetch16_2bit(s, begin + g0));
        else                  grp = fetch16_4bit(s, begin + g0);
        for (; i >= int32_t(g0) && x <= y; --i)
        {
            const uint32_t c = uint32_t(grp >> (4u * (uint32_t(i) - g0))) & 15u;
            if (c > 3u) return make_uint2(1u, 0u);
            const uint2 r = fm_rank2(f, x - 1u, y, c);
            x = f.L2[c] + r.x + 1u;
            y = f.L2[c] + r.y;
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

// one lane = one k-mer code: its match range, by the plain search (the table is ignored)
__global__ void __launch_bounds__(256)
fm_build_ktab_kernel(const Fmi f, uint32_t k, uint32_t n_codes, uint2* __restrict__ out)
{
    const uint32_t code = blockIdx.x * 256u + threadIdx.x;
    if (code >= n_codes) return;
    uint32_t x = 0, y = f.length;
    for (int32_t t = int32_t(k) - 1; t >= 0 && x <= y; --t)
    {
        const uint32_t c = (code >> (2u * uint32_t(t))) & 3u;
        const uint2 r = fm_rank2(f, x - 1u, y, c);
        x = f.L2[c] + r.x + 1u;
        y = f.L2[c] + r.y;
    }
    out[code] = make_uint2(x, y);
}

__global__ void __launch_bounds__(256)
fm_match_kernel(const Fmi f, const StringSet seeds, uint32_t n, uint2* __restrict__ out)
{
    const uint32_t id = blockIdx.x * 256u + threadIdx.x;
    if (id >= n) return;
    const uint32_t len = seeds.length ? seeds.length[id] : seeds.fixed_length;
    out[id] = fm_match(f, seeds.s, seeds.begin[id], len);
}

// ------------------------------------------------------------------ locate
// locate_ssa_iterator (fmindex_inl.h:511-545): LF-walk to the next sampled row.
__device__ __forceinline__ uint2 fm_locate_it(const Fmi& f, uint32_t j)
{
    uint32_t t = 0;
    const uint32_t mask = f.sa_int - 1u;
    while ((j & mask) != 0u)
    {
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
    return make_uint2(j, t);
}

__global__ void __launch_bounds__(256)
fm_locate_kernel(const Fmi f, const uint32_t* __restrict__ rows, uint32_t n, uint32_t* __restrict__ out)
{
    const uint32_t id = blockIdx.x * 256u + threadIdx.x;
    if (id >= n) return;
    const uint2 it = fm_locate_it(f, rows[id]);
    out[id] = f.ssa[it.x / f.sa_int] + it.y;                      // fmindex_inl.h:500
}
__global__ void __launch_bounds__(256)
fm_locate_it_kernel(const Fmi f, const uint32_t* __restrict__ rows, uint32_t n, uint2* __restrict__ out)
{
    const uint32_t id = blockIdx.x * 256u + threadIdx.x;
    if (id >= n) return;
    out[id] = fm_locate_it(f, rows[id]);
}
__global__ void __launch_bounds__(256)
fm_lookup_it_kernel(const Fmi f, const uint2* __restrict__ it, uint32_t n, uint32_t* __restrict__ out)
{
    const uint32_t id = blockIdx.x * 256u + threadIdx.x;
    if (id >= n) return;
    const uint2 v = it[id];
    out[id] = f.ssa[v.x / f.sa_int] + v.y;                        // fmindex_inl.h:566-568
}

// ------------------------------------------------------------------ FMIndexFilter
__global__ void __launch_bounds__(256)
fm_filter_match_kernel(const Fmi f, const StringSet seeds, uint32_t n, uint2* __restrict__ out, uint64_t* __restrict__ sizes)
{
    const uint32_t id = blockIdx.x * 256u + threadIdx.x;
    if (id >= n) return;
    const uint32_t len = seeds.length ? seeds.length[id] : seeds.fixed_length;
    const uint2 r = fm_match(f, seeds.s, seeds.begin[id], len);
    out[id] = r;
    sizes[id] = uint64_t(uint32_t(1u + r.y - r.x));               // filter_inl.h:40-41
}

__global__ void __launch_bounds__(256)
fm_filter_locate_kernel(const Fmi f, const uint2* __restrict__ ranges, const uint64_t* __restrict__ slots,
                        uint32_t n_queries, uint64_t begin, uint64_t count, uint2* __restrict__ hits)
{
    const uint64_t id = uint64_t(blockIdx.x) * 256u + threadIdx.x;
    if (id >= count) return;
    const uint64_t h = begin + id;
    // upper_bound(slots, h)  (filter_inl.h:101-104)
    uint32_t lo = 0, hi = n_queries;
    while (lo < hi) { const uint32_t mid = lo + ((hi - lo) >> 1); if (slots[mid] <= h) lo = mid + 1; else hi = mid; }
    const uint32_t slot = lo;
    const uint64_t base = slot ? slots[slot - 1] : 0ull;
    const uint32_t row  = ranges[slot].x + uint32_t(h - base);
    const uint2 it = fm_locate_it(f, row);
    hits[id] = make_uint2(f.ssa[it.x / f.sa_int] + it.y, slot);
}

// ------------------------------------------------------------------ occurrence table
struct Count4 { uint32_t a, c, g, t; };
struct Count4Sum {
    __host__ __device__ __forceinline__ Count4 operator()(const Count4& x, const Count4& y) const
    { Count4 r; r.a = x.a + y.a; r.c = x.c + y.c; r.g = x.g + y.g; r.t = x.t + y.t; return r; }
};

// per block of 64 symbols: the 4 symbol counts (symbols past n in the last block are not counted)
__global__ void __launch_bounds__(256)
occ_count_kernel(uint32_t n, const uint4* __restrict__ bwt, uint32_t n_blocks, Count4* __restrict__ counts)
{
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (k >= n_blocks) return;
    const uint4 w = bwt[k];
    const uint32_t rem = n - k * 64u;
    const uint32_t cnt = rem < 64u ? rem : 64u;
    Count4 r;
    r.a = block_count(w, cnt, 0); r.c = block_count(w, cnt, 1); r.g = block_count(w, cnt, 2);
    r.t = cnt - r.a - r.c - r.g;
    counts[k] = r;
}
// interleave: record k = bwt words | exclusive prefix counts; the last block also emits L2
__global__ void __launch_bounds__(256)
occ_interleave_kernel(uint32_t n, const uint4* __restrict__ bwt, uint32_t n_blocks,
                      const Count4* __restrict__ excl, const Count4* __restrict__ counts,
                      uint4* __restrict__ rec, uint32_t* __restrict__ L2)
{
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (k >= n_blocks) return;
    const Count4 e = excl[k];
    rec[2ull * k]     = bwt[k];
    rec[2ull * k + 1] = make_uint4(e.a, e.c, e.g, e.t);
    if (k == n_blocks - 1)
    {
        const Count4 c = counts[k];
        const uint32_t ta = e.a + c.a, tc = e.c + c.c, tg = e.g + c.g, tt = e.t + c.t;
        L2[0] = 0; L2[1] = ta; L2[2] = ta + tc; L2[3] = ta + tc + tg; L2[4] = ta + tc + tg + tt;   // fmindex_impl.cu:324-327
    }
}

static inline dim3 grid_for(uint64_t n) { return dim3(uint32_t((n + 255u) / 256u)); }
static inline uint64_t align256(uint64_t x) { return (x + 255ull) & ~255ull; }

} // namespace nvb

using namespace nvb;

NVB_API int nvbio_hip_fm_rank(const nvbio_hip_fmindex* fmi, const uint32_t* k, const uint8_t* c,
                              uint32_t n, uint32_t* out, void* stream)
{
    if (n == 0 && fmi) return hipSuccess;
    if (!fmi || !fmi->bwt_occ || !k || !c || !out) return hipErrorInvalidValue;
    g_last_kernel = "fm_rank_kernel";
    hipLaunchKernelGGL(fm_rank_kernel, grid_for(n), dim3(256), 0, to_stream(stream), make_fmi(fmi), k, c, n, out);
    return hipGetLastError();
}

NVB_API int nvbio_hip_fm_rank4(const nvbio_hip_fmindex* fmi, const uint32_t* k, uint32_t n, uint32_t* out, void* stream)
{
    if (n == 0 && fmi) return hipSuccess;
    if (!fmi || !fmi->bwt_occ || !k || !out) return hipErrorInvalidValue;
    g_last_kernel = "fm_rank4_kernel";
    hipLaunchKernelGGL(fm_rank4_kernel, grid_for(n), dim3(256), 0, to_stream(stream), make_fmi(fmi), k, n, reinterpret_cast<uint4*>(out));
    return hipGetLastError();
}

NVB_API int nvbio_hip_fm_rank_range(const nvbio_hip_fmindex* fmi, const uint32_t* range, const uint8_t* c,
                                    uint32_t n, uint32_t* out, void* stream)
{
    if (n == 0 && fmi) return hipSuccess;
    if (!fmi || !fmi->bwt_occ || !range || !c || !out) return hipErrorInvalidValue;
    g_last_kernel = "fm_rank_range_kernel";
    hipLaunchKernelGGL(fm_rank_range_kernel, grid_for(n), dim3(256), 0, to_stream(stream), make_fmi(fmi),
                       reinterpret_cast<const uint2*>(range), c, n, reinterpret_cast<uint2*>(out));
    return hipGetLastError();
}

static int check_seeds(const nvbio_hip_string_set* s)
{
    if (!s || !s->words || !s->begin || s->n_words == 0) return hipErrorInvalidValue;
    if (!(s->bits == 2 || s->bits == 4)) return hipErrorNotSupported;
    return hipSuccess;
}

NVB_API int nvbio_hip_fm_match(const nvbio_hip_fmindex* fmi, const nvbio_hip_string_set* seeds,
                               uint32_t n, uint32_t* out_range, void* stream)
{
    if (n == 0 && fmi) return hipSuccess;
    if (!fmi || !fmi->bwt_occ || !out_range) return hipErrorInvalidValue;
    if (int e = check_seeds(seeds)) return e;
    g_last_kernel = "fm_match_kernel";
    hipLaunchKernelGGL(fm_match_kernel, grid_for(n), dim3(256), 0, to_stream(stream), make_fmi(fmi),
                       make_string_set(seeds), n, reinterpret_cast<uint2*>(out_range));
    return hipGetLastError();
}

NVB_API int nvbio_hip_fm_build_ktab(const nvbio_hip_fmindex* fmi, uint32_t k, uint32_t* out_ktab, void* stream)
{
    if (!fmi || !fmi->bwt_occ || !out_ktab || k < 1 || k > 14) return hipErrorInvalidValue;
    Fmi f = make_fmi(fmi);
    f.ktab = nullptr; f.ktab_k = 0;
    const uint32_t n_codes = 1u << (2u * k);
    g_last_kernel = "fm_build_ktab_kernel";
    hipLaunchKernelGGL(fm_build_ktab_kernel, grid_for(n_codes), dim3(256), 0, to_stream(stream), f, k, n_codes, reinterpret_cast<uint2*>(out_ktab));
    return hipGetLastError();
}

static int check_locate(const nvbio_hip_fmindex* fmi)
{
    if (!fmi || !fmi->bwt_occ || !fmi->ssa) return hipErrorInvalidValue;
    if (fmi->sa_int == 0 || (fmi->sa_int & (fmi->sa_int - 1)) != 0) return hipErrorInvalidValue;
    return hipSuccess;
}

NVB_API int nvbio_hip_fm_locate(const nvbio_hip_fmindex* fmi, const uint32_t* sa_rows, uint32_t n, uint32_t* out_pos, void* stream)
{
    if (n == 0 && fmi) return hipSuccess;
    if (int e = check_locate(fmi)) return e;
    if (!sa_rows || !out_pos) return hipErrorInvalidValue;
    g_last_kernel = "fm_locate_kernel";
    hipLaunchKernelGGL(fm_locate_kernel, grid_for(n), dim3(256), 0, to_stream(stream), make_fmi(fmi), sa_rows, n, out_pos);
    return hipGetLastError();
}

NVB_API int nvbio_hip_fm_locate_ssa_iterator(const nvbio_hip_fmindex* fmi, const uint32_t* sa_rows, uint32_t n, uint32_t* out_it, void* stream)
{
    if (!fmi || !fmi->bwt_occ || !sa_rows || !out_it) return hipErrorInvalidValue;
    if (fmi->sa_int == 0 || (fmi->sa_int & (fmi->sa_int - 1)) != 0) return hipErrorInvalidValue;
    if (n == 0) return hipSuccess;
    g_last_kernel = "fm_locate_it_kernel";
    hipLaunchKernelGGL(fm_locate_it_kernel, grid_for(n), dim3(256), 0, to_stream(stream), make_fmi(fmi), sa_rows, n, reinterpret_cast<uint2*>(out_it));
    return hipGetLastError();
}

NVB_API int nvbio_hip_fm_lookup_ssa_iterator(const nvbio_hip_fmindex* fmi, const uint32_t* it, uint32_t n, uint32_t* out_pos, void* stream)
{
    if (int e = check_locate(fmi)) return e;
    if (!it || !out_pos) return hipErrorInvalidValue;
    if (n == 0) return hipSuccess;
    g_last_kernel = "fm_lookup_it_kernel";
    hipLaunchKernelGGL(fm_lookup_it_kernel, grid_for(n), dim3(256), 0, to_stream(stream), make_fmi(fmi), reinterpret_cast<const uint2*>(it), n, out_pos);
    return hipGetLastError();
}

NVB_API uint64_t nvbio_hip_fm_filter_temp_bytes(uint32_t n)
{
    size_t scan = 0;
    hipcub::DeviceScan::InclusiveSum(nullptr, scan, (const uint64_t*)nullptr, (uint64_t*)nullptr, int(n));
    return align256(uint64_t(n) * 8u) + align256(scan) + 256u;
}

NVB_API int nvbio_hip_fm_filter_rank(const nvbio_hip_fmindex* fmi, const nvbio_hip_string_set* seeds,
                                     uint32_t n, uint32_t* out_range, uint64_t* out_slots,
                                     void* temp, uint64_t temp_bytes, void* stream)
{
    if (!fmi || !fmi->bwt_occ || !out_range || !out_slots) return hipErrorInvalidValue;
    if (n == 0) return hipSuccess;
    if (int e = check_seeds(seeds)) return e;
    if (!temp || temp_bytes < nvbio_hip_fm_filter_temp_bytes(n)) return hipErrorInvalidValue;
    uint64_t* sizes = reinterpret_cast<uint64_t*>(temp);
    void*  scan_tmp = reinterpret_cast<uint8_t*>(temp) + align256(uint64_t(n) * 8u);
    size_t scan_bytes = size_t(temp_bytes - align256(uint64_t(n) * 8u));
    g_last_kernel = "fm_filter_match_kernel";
    hipLaunchKernelGGL(fm_filter_match_kernel, grid_for(n), dim3(256), 0, to_stream(stream), make_fmi(fmi),
                       make_string_set(seeds), n, reinterpret_cast<uint2*>(out_range), sizes);
    if (hipError_t e = hipGetLastError()) return e;
    return hipcub::DeviceScan::InclusiveSum(scan_tmp, scan_bytes, sizes, out_slots, int(n), to_stream(stream));
}

NVB_API int nvbio_hip_fm_filter_locate(const nvbio_hip_fmindex* fmi, const uint32_t* range, const uint64_t* slots,
                                       uint32_t n_queries, uint64_t begin, uint64_t end,
                                       uint32_t* out_hits, void* stream)
{
    if (int e = check_locate(fmi)) return e;
    if (!range || !slots || !out_hits || end < begin) return hipErrorInvalidValue;
    if (end == begin) return hipSuccess;
    if (end - begin > 0xFFFFFF00ull * 256ull) return hipErrorInvalidValue;
    g_last_kernel = "fm_filter_locate_kernel";
    hipLaunchKernelGGL(fm_filter_locate_kernel, grid_for(end - begin), dim3(256), 0, to_stream(stream), make_fmi(fmi),
                       reinterpret_cast<const uint2*>(range), slots, n_queries, begin, end - begin, reinterpret_cast<uint2*>(out_hits));
    return hipGetLastError();
}

NVB_API uint64_t nvbio_hip_build_bwt_occ_temp_bytes(uint32_t n)
{
    const uint64_t n_blocks = (uint64_t(n) + 63u) / 64u;
    size_t scan = 0;
    hipcub::DeviceScan::ExclusiveScan(nullptr, scan, (const Count4*)nullptr, (Count4*)nullptr, Count4Sum(), Count4{0, 0, 0, 0}, int(n_blocks));
    return 2u * align256(n_blocks * 16u) + align256(scan) + 256u;
}

NVB_API int nvbio_hip_build_bwt_occ(uint32_t n, const uint32_t* bwt_words, uint32_t* out_bwt_occ, uint32_t* out_L2,
                                    void* temp, uint64_t temp_bytes, void* stream)
{
    if (!bwt_words || !out_bwt_occ || !out_L2 || n == 0) return hipErrorInvalidValue;
    if (!temp || temp_bytes < nvbio_hip_build_bwt_occ_temp_bytes(n)) return hipErrorInvalidValue;
    const uint32_t n_blocks = uint32_t((uint64_t(n) + 63u) / 64u);
    uint8_t* t = reinterpret_cast<uint8_t*>(temp);
    Count4* counts = reinterpret_cast<Count4*>(t);
    Count4* excl   = reinterpret_cast<Count4*>(t + align256(uint64_t(n_blocks) * 16u));
    void*   scan_tmp = t + 2u * align256(uint64_t(n_blocks) * 16u);
    size_t  scan_bytes = size_t(temp_bytes - 2u * align256(uint64_t(n_blocks) * 16u));
    hipStream_t s = to_stream(stream);
    g_last_kernel = "occ_count_kernel";
    hipLaunchKernelGGL(occ_count_kernel, grid_for(n_blocks), dim3(256), 0, s, n, reinterpret_cast<const uint4*>(bwt_words), n_blocks, counts);
    if (hipError_t e = hipGetLastError()) return e;
    if (hipError_t e = hipcub::DeviceScan::ExclusiveScan(scan_tmp, scan_bytes, counts, excl, Count4Sum(), Count4{0, 0, 0, 0}, int(n_blocks), s)) return e;
    hipLaunchKernelGGL(occ_interleave_kernel, grid_for(n_blocks), dim3(256), 0, s, n, reinterpret_cast<const uint4*>(bwt_words), n_blocks,
                       excl, counts, reinterpret_cast<uint4*>(out_bwt_occ), out_L2);
    return hipGetLastError();
}
