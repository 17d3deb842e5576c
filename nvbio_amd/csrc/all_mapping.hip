// all_mapping.hip -- the device stages of nvBowtie's all-mapping mode (Aligner::all / score_all,
// nvBowtie/bowtie2/cuda/aligner_all.h:47-694): every row of every seed hit range of every read is located, de-duplicated,
// extended and -- if its score reaches the scheme's threshold -- traced back and reported.
//
//   gather_ranges_kernel     mapping.cu:39-67       the size of every SA range of every read, in deque array order
//   select_all_kernel        select.cu:175-219      global hit number -> (read, range, row) by two binary searches
//   mark_straddling_kernel   locate_inl.h:213-244   hits whose seed crosses a reference sequence boundary
//   all_window_kernel        score_all_inl.h:99-127 / traceback_inl.h:361-385 (the same window rule for scoring and traceback)
//   score_all_output_kernel  score_all_inl.h:131-153 accepted hits become alignments (the reference appends them to a ring
//                            buffer with an atomic counter; here every job gets a flag and the host compacts in job order)
//
// One lane per hit everywhere: these are streaming passes over the hit arrays (HBM bound, a few bytes per hit); the
// binary searches run over scans that stay in L2.
#include "common.h"
#include <hipcub/hipcub.hpp>
#include <algorithm>

namespace nvb {

// first index i in [0, n) with a[i] > key (n if none): nvbio::upper_bound
template <typename K, typename T>
__device__ __forceinline__ uint32_t upper_bound_index(const K key, const T* __restrict__ a, const uint32_t n)
{
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (K(a[mid]) > key) hi = mid; else lo = mid + 1u;
    }
    return lo;
}

__global__ void __launch_bounds__(256)
gather_ranges_kernel(uint32_t n_ranges, uint32_t n_reads, const uint2* __restrict__ hits, uint32_t hits_stride,
                     const uint32_t* __restrict__ count_scan, uint64_t* __restrict__ out_ranges)
{
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= n_ranges) return;
    const uint32_t read = upper_bound_index<uint32_t>(t, count_scan, n_reads);
    const uint32_t k = t - (read ? count_scan[read - 1u] : 0u);
    out_ranges[t] = hits[uint64_t(read) * hits_stride + k].y & 0xFFFFFu;                  // range.y - range.x
}

__global__ void __launch_bounds__(256)
select_all_kernel(uint64_t begin, uint32_t count, uint32_t n_reads, uint32_t n_ranges, const uint2* __restrict__ hits, uint32_t hits_stride,
                  const uint32_t* __restrict__ count_scan, const uint64_t* __restrict__ range_scan,
                  uint32_t* __restrict__ out_loc, uint32_t* __restrict__ out_seed, uint32_t* __restrict__ out_read)
{
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= count) return;
    const uint64_t g = begin + t;
    const uint32_t range = upper_bound_index<uint64_t>(g, range_scan, n_ranges);
    const uint32_t read = upper_bound_index<uint32_t>(range, count_scan, n_reads);
    const uint32_t k = range - (read ? count_scan[read - 1u] : 0u);
    const uint2 h = hits[uint64_t(read) * hits_stride + k];
    const uint32_t row = uint32_t(g - (range ? range_scan[range - 1u] : 0ull));
    out_loc[t] = h.x + row;                                                                 // hit->front() + hit_id
    // packed_seed( pos_in_read, index_dir, rc, top_flag = 0 )
    out_seed[t] = ((h.y >> 20) & 0x3FFu) | (((h.y >> 31) & 1u) << 12) | (((h.y >> 30) & 1u) << 13);
    out_read[t] = read;
}

// NOTE the indexing, kept from the reference: lane t examines hit idx_queue[t] -- the order the hits were located in -- but
// clears flags[t], which the caller reads in the (read, strand, position) sort order.
__global__ void __launch_bounds__(256)
mark_straddling_kernel(uint32_t n, const uint32_t* __restrict__ idx_queue, uint32_t n_seqs, const uint32_t* __restrict__ seq_index,
                       const uint32_t* __restrict__ hit_loc, uint32_t seed_len, uint8_t* __restrict__ flags)
{
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= n) return;
    const uint32_t g = hit_loc[idx_queue[t]];
    const uint32_t s0 = upper_bound_index<uint32_t>(g, seq_index, n_seqs + 1u) - 1u;
    const uint32_t s1 = upper_bound_index<uint32_t>(g + seed_len, seq_index, n_seqs + 1u) - 1u;
    if (s0 != s1) flags[t] = 0u;
}

// job i: hit j = idx ? idx[i] : i at (hit_read[j], hit_loc[j], strand bit 13 of hit_seed[j]) -- or, with `alignments`, alignment i of
// read hit_read[i] -- gets the window [pos - band/2 (clamped at 0), + band + read_len (clamped at the genome's end)) and its pattern
__global__ void __launch_bounds__(256)
all_window_kernel(uint32_t n, const uint32_t* __restrict__ idx, const uint32_t* __restrict__ hit_read, const uint32_t* __restrict__ hit_loc,
                  const uint32_t* __restrict__ hit_seed, const uint2* __restrict__ alignments,
                  const uint64_t* __restrict__ read_begin, const uint32_t* __restrict__ read_len, uint32_t fixed_len, uint64_t rc_offset,
                  uint32_t band_len, uint32_t genome_len,
                  uint64_t* __restrict__ pat_begin, uint32_t* __restrict__ pat_len, uint64_t* __restrict__ text_begin, uint32_t* __restrict__ text_len)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    uint32_t r, pos, rc;
    if (alignments) { const uint2 a = alignments[i]; r = hit_read[i]; pos = a.y; rc = (a.x >> 28) & 1u; }
    else            { const uint32_t j = idx ? idx[i] : i; r = hit_read[j]; pos = hit_loc[j]; rc = (hit_seed[j] >> 13) & 1u; }
    const uint32_t len = read_len ? read_len[r] : fixed_len;
    const uint32_t gb = pos > band_len / 2u ? pos - band_len / 2u : 0u;
    const uint32_t sum = gb + band_len + len;                                                // (uint32 arithmetic, as the reference's)
    const uint32_t ge = sum < genome_len ? sum : genome_len;
    text_begin[i] = gb;
    text_len[i] = ge > gb ? ge - gb : 0u;
    pat_begin[i] = (read_begin ? read_begin[r] : uint64_t(r) * fixed_len) + (rc ? rc_offset : 0ull);
    if (pat_len) pat_len[i] = len;
}

__global__ void __launch_bounds__(256)
score_all_output_kernel(uint32_t n, const uint32_t* __restrict__ idx, const uint32_t* __restrict__ hit_read, const uint32_t* __restrict__ hit_loc,
                        const uint32_t* __restrict__ hit_seed, const int32_t* __restrict__ score, const int32_t* __restrict__ min_score_by_len,
                        const uint32_t* __restrict__ read_len, uint32_t fixed_len,
                        uint8_t* __restrict__ out_flags, uint2* __restrict__ out_alignments, uint32_t* __restrict__ out_read)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t j = idx ? idx[i] : i;
    const uint32_t r = hit_read[j];
    const int32_t s = score[i];
    const bool ok = s >= min_score_by_len[read_len ? read_len[r] : fixed_len];
    out_flags[i] = ok ? 1u : 0u;
    // io::Alignment( hit.loc, 0u, sink.score, hit.seed.rc )
    const uint32_t mag = uint32_t(s < 0 ? -s : s);
    out_alignments[i] = make_uint2((s < 0 ? 1u : 0u) | ((mag & 0x1FFFFu) << 1) | (((hit_seed[j] >> 13) & 1u) << 28), hit_loc[j]);
    out_read[i] = r;
}

// ---- the library primitives Aligner::score_all calls between its kernels (thrust::inclusive_scan, sort_enactor.sort over
// SortBuffers -- aligner_sort.cu:39-92 -- and the dedup transform, aligner_all.h:498-509), on hipCUB
__global__ void __launch_bounds__(256) iota_kernel(uint32_t n, uint32_t* __restrict__ out)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) out[i] = i;
}
__global__ void __launch_bounds__(256) hi_bits_kernel(uint32_t n, const uint32_t* __restrict__ keys, uint32_t* __restrict__ hi, uint32_t* __restrict__ idx)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) { hi[i] = keys[i] >> 16; idx[i] = i; }                                    // hi_bits_functor<uint16,uint32>
}
__global__ void __launch_bounds__(256) hi_bits16_kernel(uint32_t n, const uint32_t* __restrict__ keys, uint16_t* __restrict__ hi, uint32_t* __restrict__ idx)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) { hi[i] = uint16_t(keys[i] >> 16); idx[i] = i; }
}
// SortingKeys (aligner_all.h:229-247): loc + (read_id << 33) + (rc << 32)
__global__ void __launch_bounds__(256) hit_keys_kernel(uint32_t n, const uint32_t* __restrict__ hit_read, const uint32_t* __restrict__ hit_loc,
                                                       const uint32_t* __restrict__ hit_seed, uint64_t* __restrict__ keys, uint32_t* __restrict__ idx)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) { keys[i] = uint64_t(hit_loc[i]) + (uint64_t(hit_read[i]) << 33) + (uint64_t((hit_seed[i] >> 13) & 1u) << 32); idx[i] = i; }
}
__global__ void __launch_bounds__(256) first_of_run_kernel(uint32_t n, const uint64_t* __restrict__ sorted_keys, uint8_t* __restrict__ flags)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) flags[i] = (i == 0u || sorted_keys[i] != sorted_keys[i - 1u]) ? 1u : 0u;
}

static inline uint64_t align256(uint64_t x) { return (x + 255u) & ~uint64_t(255); }

} // namespace nvb

using namespace nvb;

static inline dim3 grid_for(uint64_t n) { return dim3(uint32_t((n + 255u) / 256u)); }

NVB_API int nvbio_hip_gather_ranges(uint32_t n_ranges, uint32_t n_reads, const uint64_t* hits, uint32_t hits_stride, const uint32_t* hit_count_scan,
                                    uint64_t* out_ranges, void* stream)
{
    if (n_ranges == 0) return hipSuccess;
    if (!hits || !hit_count_scan || !out_ranges || hits_stride == 0 || n_reads == 0) return hipErrorInvalidValue;
    g_last_kernel = "gather_ranges_kernel";
    hipLaunchKernelGGL(gather_ranges_kernel, grid_for(n_ranges), dim3(256), 0, to_stream(stream), n_ranges, n_reads, reinterpret_cast<const uint2*>(hits),
                       hits_stride, hit_count_scan, out_ranges);
    return hipGetLastError();
}

NVB_API int nvbio_hip_select_all(uint64_t begin, uint32_t count, uint32_t n_reads, uint32_t n_ranges, const uint64_t* hits, uint32_t hits_stride,
                                 const uint32_t* hit_count_scan, const uint64_t* hit_range_scan, uint32_t* out_loc, uint32_t* out_seed, uint32_t* out_read_id,
                                 void* stream)
{
    if (count == 0) return hipSuccess;
    if (!hits || !hit_count_scan || !hit_range_scan || !out_loc || !out_seed || !out_read_id || hits_stride == 0 || n_reads == 0 || n_ranges == 0)
        return hipErrorInvalidValue;
    g_last_kernel = "select_all_kernel";
    hipLaunchKernelGGL(select_all_kernel, grid_for(count), dim3(256), 0, to_stream(stream), begin, count, n_reads, n_ranges, reinterpret_cast<const uint2*>(hits),
                       hits_stride, hit_count_scan, hit_range_scan, out_loc, out_seed, out_read_id);
    return hipGetLastError();
}

NVB_API int nvbio_hip_mark_straddling(uint32_t n, const uint32_t* idx_queue, uint32_t n_sequences, const uint32_t* sequence_index, const uint32_t* hit_loc,
                                      uint32_t seed_len, uint8_t* flags, void* stream)
{
    if (n == 0) return hipSuccess;
    if (!idx_queue || !sequence_index || !hit_loc || !flags || n_sequences == 0) return hipErrorInvalidValue;
    g_last_kernel = "mark_straddling_kernel";
    hipLaunchKernelGGL(mark_straddling_kernel, grid_for(n), dim3(256), 0, to_stream(stream), n, idx_queue, n_sequences, sequence_index, hit_loc, seed_len, flags);
    return hipGetLastError();
}

NVB_API int nvbio_hip_score_all_setup(uint32_t n, const uint32_t* idx, const uint32_t* hit_read_id, const uint32_t* hit_loc, const uint32_t* hit_seed,
                                      const uint64_t* read_begin, const uint32_t* read_len, uint32_t fixed_read_len, uint64_t rc_offset,
                                      uint32_t band_len, uint32_t genome_length,
                                      uint64_t* pattern_begin, uint32_t* pattern_len, uint64_t* text_begin, uint32_t* text_len, void* stream)
{
    if (n == 0) return hipSuccess;
    if (!hit_read_id || !hit_loc || !hit_seed || !pattern_begin || !text_begin || !text_len) return hipErrorInvalidValue;
    if ((!read_len && fixed_read_len == 0) || (read_len && !pattern_len)) return hipErrorInvalidValue;
    g_last_kernel = "all_window_kernel";
    hipLaunchKernelGGL(all_window_kernel, grid_for(n), dim3(256), 0, to_stream(stream), n, idx, hit_read_id, hit_loc, hit_seed, static_cast<const uint2*>(nullptr),
                       read_begin, read_len, fixed_read_len, rc_offset, band_len, genome_length, pattern_begin, pattern_len, text_begin, text_len);
    return hipGetLastError();
}

NVB_API int nvbio_hip_score_all_output(uint32_t n, const uint32_t* idx, const uint32_t* hit_read_id, const uint32_t* hit_loc, const uint32_t* hit_seed,
                                       const int32_t* score, const int32_t* min_score_by_len, const uint32_t* read_len, uint32_t fixed_read_len,
                                       uint8_t* out_flags, uint64_t* out_alignments, uint32_t* out_read_id, void* stream)
{
    if (n == 0) return hipSuccess;
    if (!hit_read_id || !hit_loc || !hit_seed || !score || !min_score_by_len || !out_flags || !out_alignments || !out_read_id) return hipErrorInvalidValue;
    if (!read_len && fixed_read_len == 0) return hipErrorInvalidValue;
    g_last_kernel = "score_all_output_kernel";
    hipLaunchKernelGGL(score_all_output_kernel, grid_for(n), dim3(256), 0, to_stream(stream), n, idx, hit_read_id, hit_loc, hit_seed, score, min_score_by_len,
                       read_len, fixed_read_len, out_flags, reinterpret_cast<uint2*>(out_alignments), out_read_id);
    return hipGetLastError();
}

NVB_API int nvbio_hip_traceback_all_setup(uint32_t n, const uint64_t* alignments, const uint32_t* read_id,
                                          const uint64_t* read_begin, const uint32_t* read_len, uint32_t fixed_read_len, uint64_t rc_offset,
                                          uint32_t band_len, uint32_t genome_length,
                                          uint64_t* pattern_begin, uint32_t* pattern_len, uint64_t* text_begin, uint32_t* text_len, void* stream)
{
    if (n == 0) return hipSuccess;
    if (!alignments || !read_id || !pattern_begin || !text_begin || !text_len) return hipErrorInvalidValue;
    if ((!read_len && fixed_read_len == 0) || (read_len && !pattern_len)) return hipErrorInvalidValue;
    g_last_kernel = "all_window_kernel";
    hipLaunchKernelGGL(all_window_kernel, grid_for(n), dim3(256), 0, to_stream(stream), n, static_cast<const uint32_t*>(nullptr), read_id,
                       static_cast<const uint32_t*>(nullptr), static_cast<const uint32_t*>(nullptr), reinterpret_cast<const uint2*>(alignments),
                       read_begin, read_len, fixed_read_len, rc_offset, band_len, genome_length, pattern_begin, pattern_len, text_begin, text_len);
    return hipGetLastError();
}

// ------------------------------------------------------------------ scans and sorts
// One scratch size serves every call below over n items: key / value ping-pong buffers + the hipCUB workspace.
NVB_API uint64_t nvbio_hip_all_mapping_temp_bytes(uint32_t n)
{
    size_t a = 0, b = 0, c = 0, d = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, a, (const uint64_t*)nullptr, (uint64_t*)nullptr, (const uint32_t*)nullptr, (uint32_t*)nullptr, int(n), 0, 64);
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, b, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint32_t*)nullptr, (uint32_t*)nullptr, int(n), 0, 16);
    (void)hipcub::DeviceScan::InclusiveSum(nullptr, c, (const uint64_t*)nullptr, (uint64_t*)nullptr, int(n));
    (void)hipcub::DeviceScan::InclusiveSum(nullptr, d, (const uint32_t*)nullptr, (uint32_t*)nullptr, int(n));
    const size_t w = std::max(std::max(a, b), std::max(c, d));
    return align256(w) + 2u * align256(uint64_t(n) * 8u) + align256(uint64_t(n) * 4u) + 2u * align256(uint64_t(n) * 4u) + 256u;      // (+ the ping-pong index halves of nvbio_hip_sort_hits_pingpong)
}

namespace {
struct Scratch { uint8_t* work; size_t work_bytes; uint64_t* k0; uint64_t* k1; uint32_t* v0; uint32_t* pp0; uint32_t* pp1; };
inline bool carve(void* temp, uint64_t temp_bytes, uint32_t n, Scratch& s)
{
    if (!temp || temp_bytes < nvbio_hip_all_mapping_temp_bytes(n)) return false;
    uint8_t* p = reinterpret_cast<uint8_t*>((reinterpret_cast<uintptr_t>(temp) + 255u) & ~uintptr_t(255));
    s.k0 = reinterpret_cast<uint64_t*>(p); p += align256(uint64_t(n) * 8u);
    s.k1 = reinterpret_cast<uint64_t*>(p); p += align256(uint64_t(n) * 8u);
    s.v0 = reinterpret_cast<uint32_t*>(p); p += align256(uint64_t(n) * 4u);
    s.pp0 = reinterpret_cast<uint32_t*>(p); p += align256(uint64_t(n) * 4u);
    s.pp1 = reinterpret_cast<uint32_t*>(p); p += align256(uint64_t(n) * 4u);
    s.work = p; s.work_bytes = size_t(temp_bytes - 256u - 2u * align256(uint64_t(n) * 8u) - 3u * align256(uint64_t(n) * 4u));
    return true;
}
}

NVB_API int nvbio_hip_inclusive_scan_u32(uint32_t n, const uint32_t* in, uint32_t* out, void* temp, uint64_t temp_bytes, void* stream)
{
    if (n == 0) return hipSuccess;
    Scratch s;
    if (!in || !out || !carve(temp, temp_bytes, n, s)) return hipErrorInvalidValue;
    g_last_kernel = "hipcub::DeviceScan::InclusiveSum";
    return hipcub::DeviceScan::InclusiveSum(s.work, s.work_bytes, in, out, int(n), to_stream(stream));
}

NVB_API int nvbio_hip_inclusive_scan_u64(uint32_t n, const uint64_t* in, uint64_t* out, void* temp, uint64_t temp_bytes, void* stream)
{
    if (n == 0) return hipSuccess;
    Scratch s;
    if (!in || !out || !carve(temp, temp_bytes, n, s)) return hipErrorInvalidValue;
    g_last_kernel = "hipcub::DeviceScan::InclusiveSum";
    return hipcub::DeviceScan::InclusiveSum(s.work, s.work_bytes, in, out, int(n), to_stream(stream));
}

NVB_API int nvbio_hip_sort_hi_bits(uint32_t n, const uint32_t* keys, uint32_t* out_idx, void* temp, uint64_t temp_bytes, void* stream)
{
    if (n == 0) return hipSuccess;
    Scratch s;
    if (!keys || !out_idx || !carve(temp, temp_bytes, n, s)) return hipErrorInvalidValue;
    uint32_t* h0 = reinterpret_cast<uint32_t*>(s.k0); uint32_t* h1 = reinterpret_cast<uint32_t*>(s.k1);
    hipLaunchKernelGGL(hi_bits_kernel, grid_for(n), dim3(256), 0, to_stream(stream), n, keys, h0, s.v0);
    g_last_kernel = "hipcub::DeviceRadixSort::SortPairs";
    return hipcub::DeviceRadixSort::SortPairs(s.work, s.work_bytes, h0, h1, s.v0, out_idx, int(n), 0, 16, to_stream(stream));       // stable
}

NVB_API int nvbio_hip_sort_hits(uint32_t n, const uint32_t* hit_read_id, const uint32_t* hit_loc, const uint32_t* hit_seed, uint32_t* out_idx, uint8_t* out_first,
                                void* temp, uint64_t temp_bytes, void* stream)
{
    if (n == 0) return hipSuccess;
    Scratch s;
    if (!hit_read_id || !hit_loc || !hit_seed || !out_idx || !out_first || !carve(temp, temp_bytes, n, s)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(hit_keys_kernel, grid_for(n), dim3(256), 0, to_stream(stream), n, hit_read_id, hit_loc, hit_seed, s.k0, s.v0);
    g_last_kernel = "hipcub::DeviceRadixSort::SortPairs";
    const hipError_t e = hipcub::DeviceRadixSort::SortPairs(s.work, s.work_bytes, s.k0, s.k1, s.v0, out_idx, int(n), 0, 64, to_stream(stream));   // stable
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(first_of_run_kernel, grid_for(n), dim3(256), 0, to_stream(stream), n, s.k1, out_first);
    return hipGetLastError();
}

// Aligner::all hands mark_straddling `pipeline.idx_queue` (aligner_all.h:520): the pointer sort_hi_bits returned (:465, aligner_sort.cu:37-62), i.e.
// one half of the ping-pong index buffer -- which sort_64_bits (:500, aligner_sort.cu:66-86) has meanwhile refilled with 0 .. n-1 and sorted through.
// What that half holds afterwards is the final index when both sorts end in the same half, and whatever the last pass but one left when they do not:
// decided by the sort library's pass structure for (n, key type), not by the data of the first sort.  This entry replays the two calls as the
// reference makes them -- DoubleBuffer<uint16> / <uint32> over bits 0..16, then DoubleBuffer<uint64> / <uint32> over bits 0..64 on the SAME index
// halves, 0 .. n-1 written into half 0 before each -- and returns, besides sort_hits' results, the half the first sort ended in: out_stale.
NVB_API int nvbio_hip_sort_hits_pingpong(uint32_t n, const uint32_t* hit_read_id, const uint32_t* hit_loc, const uint32_t* hit_seed, uint32_t* out_idx, uint8_t* out_first,
                                         uint32_t* out_stale, void* temp, uint64_t temp_bytes, void* stream)
{
    if (n == 0) return hipSuccess;
    Scratch s;
    if (!hit_read_id || !hit_loc || !hit_seed || !out_idx || !out_first || !out_stale || !carve(temp, temp_bytes, n, s)) return hipErrorInvalidValue;
    hipStream_t st = to_stream(stream);
    g_last_kernel = "hipcub::DeviceRadixSort::SortPairs";
    // sort_hi_bits: only the half it ends in matters here (its indices are overwritten below), so any 16-bit keys do
    uint16_t* h0 = reinterpret_cast<uint16_t*>(s.k0); uint16_t* h1 = reinterpret_cast<uint16_t*>(s.k1);
    hipLaunchKernelGGL(hi_bits16_kernel, grid_for(n), dim3(256), 0, st, n, hit_loc, h0, s.pp0);
    uint32_t first_half;
    {
        hipcub::DoubleBuffer<uint16_t> k(h0, h1);
        hipcub::DoubleBuffer<uint32_t> v(s.pp0, s.pp1);
        size_t bytes = s.work_bytes;
        if (const hipError_t e = hipcub::DeviceRadixSort::SortPairs(s.work, bytes, k, v, int(n), 0, 16, st)) return e;
        first_half = uint32_t(v.selector);
    }
    // sort_64_bits
    hipLaunchKernelGGL(hit_keys_kernel, grid_for(n), dim3(256), 0, st, n, hit_read_id, hit_loc, hit_seed, s.k0, s.pp0);
    hipcub::DoubleBuffer<uint64_t> k(s.k0, s.k1);
    hipcub::DoubleBuffer<uint32_t> v(s.pp0, s.pp1);
    size_t bytes = s.work_bytes;
    if (const hipError_t e = hipcub::DeviceRadixSort::SortPairs(s.work, bytes, k, v, int(n), 0, 64, st)) return e;
    if (const hipError_t e = hipMemcpyAsync(out_idx, v.Current(), uint64_t(n) * 4u, hipMemcpyDeviceToDevice, st)) return e;
    if (const hipError_t e = hipMemcpyAsync(out_stale, first_half ? s.pp1 : s.pp0, uint64_t(n) * 4u, hipMemcpyDeviceToDevice, st)) return e;
    hipLaunchKernelGGL(first_of_run_kernel, grid_for(n), dim3(256), 0, st, n, k.Current(), out_first);
    return hipGetLastError();
}
