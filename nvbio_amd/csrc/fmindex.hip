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
#include "fmindex_device.h"
#include <hipcub/hipcub.hpp>

namespace nvb {

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

// one lane = one k-mer code: its match range, by the plain search (the table is ignored)
__global__ void __launch_bounds__(256)
fm_build_ktab_kernel(const Fmi f, uint32_t k, uint64_t n_codes, uint2* __restrict__ out)
{
    // (grid-stride: k = 16 has 2^32 codes, more than one launch may hold threads)
    for (uint64_t code64 = uint64_t(blockIdx.x) * 256u + threadIdx.x; code64 < n_codes; code64 += uint64_t(gridDim.x) * 256u)
    {
        const uint32_t code = uint32_t(code64);
        uint32_t x = 0, y = f.length;
        for (int32_t t = int32_t(k) - 1; t >= 0 && x <= y; --t)
        {
            const uint32_t c = (code >> (2u * uint32_t(t))) & 3u;
            const uint2 r = fm_rank2(f, x - 1u, y, c);
            x = f.L2[c] + r.x + 1u;
            y = f.L2[c] + r.y;
        }
        out[code64] = make_uint2(x, y);
    }
}

__global__ void __launch_bounds__(256)
fm_match_kernel(const Fmi f, const StringSet seeds, uint32_t n, uint2* __restrict__ out)
{
    const uint32_t id = blockIdx.x * 256u + threadIdx.x;
    if (id >= n) return;
    const uint32_t len = seeds.length ? seeds.length[id] : seeds.fixed_length;
    out[id] = fm_match(f, seeds.s, seeds.begin[id], len);
}

__global__ void __launch_bounds__(256)
fm_locate_kernel(const Fmi f, const uint32_t* __restrict__ rows, uint32_t n, uint32_t* __restrict__ out)
{
    const uint32_t id = blockIdx.x * 256u + threadIdx.x;
    if (id >= n) return;
    const uint2 it = fm_locate_it(f, rows[id]);
    out[id] = f.ssa[it.x / f.sa_int] + it.y;                      // fmindex_inl.h:500
}
// the suffix array sampled every `step` rows: out[k] = locate(k * step) through the index's own (sparser) SSA
__global__ void __launch_bounds__(256)
fm_dense_ssa_kernel(const Fmi f, uint32_t step, uint64_t n_out, uint32_t* __restrict__ out)
{
    const uint64_t id = uint64_t(blockIdx.x) * 256u + threadIdx.x;
    if (id >= n_out) return;
    const uint2 it = fm_locate_it(f, uint32_t(id * step));
    out[id] = f.ssa[it.x / f.sa_int] + it.y;
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
    if (!fmi || !fmi->bwt_occ || !out_ktab || k < 1 || k > 16) return hipErrorInvalidValue;      // 4^15 entries = 8.6 GB, 4^16 = 34 GB; a code is a uint32
    Fmi f = make_fmi(fmi);
    f.ktab = nullptr; f.ktab_k = 0;
    const uint64_t n_codes = 1ull << (2u * k);
    g_last_kernel = "fm_build_ktab_kernel";
    hipLaunchKernelGGL(fm_build_ktab_kernel, grid_for(std::min<uint64_t>(n_codes, 1ull << 30)), dim3(256), 0, to_stream(stream), f, k, n_codes, reinterpret_cast<uint2*>(out_ktab));
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

NVB_API uint64_t nvbio_hip_fm_dense_ssa_entries(uint32_t length, uint32_t sa_int) { return sa_int ? (uint64_t(length) + sa_int) / sa_int : 0ull; }
NVB_API int nvbio_hip_fm_build_dense_ssa(const nvbio_hip_fmindex* fmi, uint32_t sa_int_out, uint32_t* out_ssa, void* stream)
{
    if (int e = check_locate(fmi)) return e;
    if (!out_ssa || sa_int_out == 0 || (sa_int_out & (sa_int_out - 1)) != 0 || sa_int_out > fmi->sa_int) return hipErrorInvalidValue;
    const uint64_t n_out = nvbio_hip_fm_dense_ssa_entries(fmi->length, sa_int_out);
    g_last_kernel = "fm_dense_ssa_kernel";
    hipLaunchKernelGGL(fm_dense_ssa_kernel, grid_for(n_out), dim3(256), 0, to_stream(stream), make_fmi(fmi), sa_int_out, n_out, out_ssa);
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
    (void)hipcub::DeviceScan::InclusiveSum(nullptr, scan, (const uint64_t*)nullptr, (uint64_t*)nullptr, int(n));     // size query only
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
    (void)hipcub::DeviceScan::ExclusiveScan(nullptr, scan, (const Count4*)nullptr, (Count4*)nullptr, Count4Sum(), Count4{0, 0, 0, 0}, int(n_blocks));     // size query only
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
