// compat/nvbio/fmindex/filter.h -- FMIndexFilter<system_tag, fm_index_type> (nvbio/fmindex/filter.h:74-203, filter_inl.h:200-402)
// behind the reference's own template: rank(index, string_set) matches every string of the set in the FM-index and ranks the
// ranges (inclusive scan of their sizes, uint64), locate(begin, end, hits) enumerates the global hits [begin, end) as
// (text position, string id) pairs.  FMIndexFilterHost / FMIndexFilterDevice are the two aliases callers name
// (examples/fmmap/fmmap.cu:92-97).
//
// The device filter picks one of two executions, with identical results (last_path() says which):
//   * tuned   -- the fm_index is the production layout (nvbio/io/fmindex/fmindex.h:159-174: one uint4 of big-endian 2-bit BWT and
//                one uint4 of counters per 64 symbols, seen through deinterleaved_iterator<2,0|1,.>, 32-bit coordinates, an
//                SSA_index_multiple_context over device words) and the strings are windows of packed words in device memory:
//                the index is described to the C-ABI (include/nvbio_hip.h) in place and nvbio_hip_fm_filter_rank / _locate run
//                the gfx950 kernels.  With 288 GB of HBM per GPU the filter also keeps, per index it has seen, the line-native
//                two-symbol index those kernels prefer (nvbio_hip_fm_build_dimer_index: 3.7 B per row, built on the device in
//                tens of milliseconds) -- set_line_native(false) keeps it on the reference layout;
//   * generic -- any other fm_index<> / string-set (64-bit coordinates, separate bwt / occ arrays, 8-bit strings ...): one lane
//                per query runs match() from fmindex.h, hipCUB scans the sizes, one lane per hit runs upper_bound + locate().
// The host filter runs the same templates under OpenMP.
#pragma once
#include "fmindex.h"
#include "../basic/packed_view.h"
#include "../basic/deinterleaved_iterator.h"
#include "../basic/vector.h"
#include "../basic/cuda/primitives.h"
#include "../strings/string_set.h"
#include <stdexcept>
#include <string>
#include <vector>
#if defined(_OPENMP)
#include <omp.h>
#endif
#if defined(__HIPCC__)
#include <hipcub/hipcub.hpp>
#if !defined(NVBIO_HIP_COMPAT_NO_TUNED)
#include "../../../../nvbio_hip.h"
#define NVBIO_HIP_COMPAT_FILTER_TUNED 1
#endif
#endif

namespace nvbio {

namespace fmindex {

/// rows of an inclusive SA range (filter_inl.h:36-43); an emptied range (l = r + 1) has none
template <typename range_type>
struct range_size
{
    typedef range_type argument_type;
    typedef uint64     result_type;
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint64 operator()(const range_type range) const
    {
        typedef typename vector_traits<range_type>::value_type coord_type;
        return uint64(coord_type(coord_type(1) + range.y - range.x));
    }
};

/// first element of a sorted array greater than x (upper_bound, nvbio/basic/algorithms.h)
template <typename T>
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 upper_bound_index(const T x, const T* a, const uint32 n)
{
    uint32 lo = 0, hi = n;
    while (lo < hi) { const uint32 mid = lo + (hi - lo) / 2u; if (a[mid] <= x) lo = mid + 1u; else hi = mid; }
    return lo;
}

/// global hit index -> (SA row, string id)   (filter_inl.h:77-121)
template <typename range_type>
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE range_type filter_hit(const uint64 output_index, const uint32 n_queries, const uint64* slots, const range_type* ranges)
{
    typedef typename vector_traits<range_type>::value_type coord_type;
    const uint32 slot = upper_bound_index(output_index, slots, n_queries);
    const uint64 base = slot ? slots[slot - 1u] : 0u;
    return make_vector(coord_type(ranges[slot].x + coord_type(output_index - base)), coord_type(slot));
}

struct hip_error : public std::runtime_error {
    int code;
    hip_error(const char* what, int c) : std::runtime_error(std::string(what) + " failed, hipError " + std::to_string(c)), code(c) {}
};
inline void check(const int err, const char* what) { if (err != 0) throw hip_error(what, err); }

#if defined(__HIPCC__)
template <typename index_type, typename string_set_type>
__global__ void __launch_bounds__(128) filter_rank_kernel(const uint32 n, const index_type index, const string_set_type string_set,
                                                          typename index_type::range_type* ranges, uint64* sizes)
{
    const uint32 q = blockIdx.x * 128u + threadIdx.x;
    if (q >= n) return;
    const typename string_set_type::string_type s = string_set[q];
    const typename index_type::range_type r = match(index, s, length(s));
    ranges[q] = r;
    sizes[q]  = range_size<typename index_type::range_type>()(r);
}
template <typename index_type, typename hits_iterator>
__global__ void __launch_bounds__(128) filter_locate_kernel(const uint64 begin, const uint32 n_hits, const index_type index, const uint32 n_queries,
                                                            const uint64* slots, const typename index_type::range_type* ranges, hits_iterator hits)
{
    const uint32 h = blockIdx.x * 128u + threadIdx.x;
    if (h >= n_hits) return;
    const typename index_type::range_type pair = filter_hit(begin + h, n_queries, slots, ranges);
    hits[h] = make_vector(locate(index, pair.x), pair.y);
}

/// growable device storage owned by a filter (the reference keeps thrust::device_vectors there)
template <typename T>
struct device_array
{
    device_array() : ptr(nullptr), cap(0) {}
    ~device_array() { if (ptr) (void)hipFree(ptr); }
    device_array(const device_array&) = delete;
    device_array& operator=(const device_array&) = delete;
    T* reserve(const uint64 n)
    {
        if (n > cap) { if (ptr) (void)hipFree(ptr); ptr = nullptr; cap = 0; check(hipMalloc(reinterpret_cast<void**>(&ptr), (n ? n : 1u) * sizeof(T)), "hipMalloc"); cap = n; }
        return ptr;
    }
    T* ptr; uint64 cap;
};
#endif

// ---------------------------------------------------------------------------------------- recognition of the production layout
template <typename It> struct uint4_pointer { static const bool ok = false; };
template <> struct uint4_pointer<const uint4*> { static const bool ok = true; static const uint4* get(const uint4* p) { return p; } };
template <> struct uint4_pointer<uint4*>       { static const bool ok = true; static const uint4* get(uint4* p) { return p; } };
template <> struct uint4_pointer< cuda::ldg_pointer<uint4> > { static const bool ok = true; static const uint4* get(cuda::ldg_pointer<uint4> p) { return p.base; } };
template <typename It> struct uint4_pointer< const_cached_iterator<It> > { static const bool ok = uint4_pointer<It>::ok;
    static const uint4* get(const_cached_iterator<It> p) { return uint4_pointer<It>::get(p.base()); } };

template <typename L> struct l2_pointer { static const bool ok = false; };
template <> struct l2_pointer<const uint32*> { static const bool ok = true; static const uint32* get(const uint32* p) { return p; } };
template <> struct l2_pointer<uint32*>       { static const bool ok = true; static const uint32* get(uint32* p) { return p; } };
template <> struct l2_pointer< cuda::ldg_pointer<uint32> > { static const bool ok = true; static const uint32* get(cuda::ldg_pointer<uint32> p) { return p.base; } };

template <typename fm_index_type> struct production_layout { static const bool ok = false; };
template <typename B0, typename B1, typename X, typename CT, uint32 K, typename SI, typename L>
struct production_layout< fm_index< rank_dictionary<2, 64, PackedStream<deinterleaved_iterator<2, 0, B0>, uint8, 2, true, X>, deinterleaved_iterator<2, 1, B1>, CT>,
                                    SSA_index_multiple_context<K, SI>, L> >
{
    typedef fm_index< rank_dictionary<2, 64, PackedStream<deinterleaved_iterator<2, 0, B0>, uint8, 2, true, X>, deinterleaved_iterator<2, 1, B1>, CT>,
                      SSA_index_multiple_context<K, SI>, L> index_type;
    typedef typename index_type::L2_iterator L2_iterator;
    static const bool ok = uint4_pointer<B0>::ok && uint4_pointer<B1>::ok && nvbio::priv::word_pointer<SI>::ok && l2_pointer<L2_iterator>::ok &&
                           (K & (K - 1u)) == 0u && same_type<typename index_type::index_type, uint32>::pred;
#if defined(NVBIO_HIP_COMPAT_FILTER_TUNED)
    /// fill the C-ABI description; false when the two halves are not one interleaved array starting at its first record
    static bool describe(const index_type& f, nvbio_hip_fmindex& m)
    {
        const uint4* b0 = uint4_pointer<B0>::get(f.m_rank_dict.m_text.stream().base());
        const uint4* b1 = uint4_pointer<B1>::get(f.m_rank_dict.m_occ.base());
        if (b0 != b1 || b0 == nullptr || f.m_rank_dict.m_text.index() != 0u || f.m_rank_dict.m_text.stream().m_off != 0u || f.m_rank_dict.m_occ.m_off != 0u) return false;
        m.length = f.length(); m.primary = f.primary(); m.sa_int = K;
        // L2 lives wherever the caller keeps it (device memory for a device index)
        if (hipMemcpy(m.L2, l2_pointer<L2_iterator>::get(f.m_L2), 5u * sizeof(uint32), hipMemcpyDefault) != hipSuccess) { (void)hipGetLastError(); return false; }
        m.bwt_occ = reinterpret_cast<const uint32*>(b0);
        m.ssa = nvbio::priv::word_pointer<SI>::get(f.m_sa.m_ssa);
        m.ktab = nullptr; m.ktab_k = 0; m._pad = 0; m.dimer = nullptr; m.trimer = nullptr; m.dimer_p1 = m.dimer_fill1 = 0;
        for (int i = 0; i < 4; ++i) m.dimer_S[i] = m.dimer_T[i] = 0;
        return true;
    }
#endif
};

#if defined(NVBIO_HIP_COMPAT_FILTER_TUNED)
__device__ __forceinline__ unsigned long long filter_wave_min(unsigned long long v)
{ for (int o = 32; o > 0; o >>= 1) { const unsigned long long w = __shfl_xor(v, o, 64); v = w < v ? w : v; } return v; }
__device__ __forceinline__ unsigned long long filter_wave_max(unsigned long long v)
{ for (int o = 32; o > 0; o >>= 1) { const unsigned long long w = __shfl_xor(v, o, 64); v = w > v ? w : v; } return v; }

static __global__ void filter_init_bounds_kernel(unsigned long long* b) { if (threadIdx.x < 2u) b[threadIdx.x] = threadIdx.x ? 0ull : ~0ull; }
/// where each string of the set lives: absolute symbol offset (from address 0), length; lowest / highest word over the set
template <typename string_set_type>
__global__ void __launch_bounds__(128) filter_describe_kernel(const uint32 n, const string_set_type string_set, uint64* begin, uint32* len, unsigned long long* bounds)
{
    typedef typename string_set_type::string_type string_type;
    typedef nvbio::priv::packed_view<string_type> view;
    const uint32 q = blockIdx.x * 128u + threadIdx.x;
    unsigned long long lo = ~0ull, hi = 0ull;
    if (q < n)
    {
        const string_type s = string_set[q];
        uint64 w0; uint32 first;
        view::where(s, w0, first);
        const uint32 per = 32u / view::BITS;
        begin[q] = w0 * per + first; len[q] = uint32(s.length());
        lo = w0; hi = w0 + (first + uint32(s.length()) + per - 1u) / per;        // exactly the words the string occupies: the kernels clamp their look-ahead to it
    }
    lo = filter_wave_min(lo); hi = filter_wave_max(hi);
    if ((threadIdx.x & 63u) == 0u && hi) { atomicMin(&bounds[0], lo); atomicMax(&bounds[1], hi); }
}
static __global__ void __launch_bounds__(256) filter_rebase_kernel(const uint32 n, uint64* begin, const uint64 delta)
{
    const uint32 q = blockIdx.x * 256u + threadIdx.x;
    if (q < n) begin[q] -= delta;
}
#endif

} // namespace fmindex

template <typename system_tag, typename fm_index_type> struct FMIndexFilter {};

// ---------------------------------------------------------------------------------------- host
template <typename fm_index_type>
struct FMIndexFilter<host_tag, fm_index_type>
{
    typedef host_tag                                        system_tag;
    typedef fm_index_type                                   index_type;
    typedef typename index_type::index_type                 coord_type;
    static const uint32                                     coord_dim = vector_traits<coord_type>::DIM;
    typedef typename vector_type<coord_type, 2>::type       range_type;
    static const uint32                                     hit_dim = coord_dim * 2;
    typedef typename vector_type<coord_type, hit_dim>::type hit_type;

    FMIndexFilter() : m_n_queries(0), m_n_occurrences(0) {}

    /// filter.h:91-101: match every string of the set, rank the ranges; returns the total number of hits
    template <typename string_set_type>
    uint64 rank(const fm_index_type& index, const string_set_type& string_set)
    {
        m_n_queries = string_set.size();
        m_index     = index;
        m_ranges.resize(m_n_queries);
        m_slots.resize(m_n_queries);
        const int64 n = int64(m_n_queries);
        #pragma omp parallel for schedule(dynamic, 1024)
        for (int64 q = 0; q < n; ++q)
        {
            const typename string_set_type::string_type s = string_set[uint32(q)];
            m_ranges[q] = match(m_index, s, length(s));
        }
        uint64 sum = 0;
        for (int64 q = 0; q < n; ++q) { sum += fmindex::range_size<range_type>()(m_ranges[q]); m_slots[q] = sum; }
        m_n_occurrences = sum;
        return m_n_occurrences;
    }
    /// filter.h:103-114: hits[h - begin] = (text position, string id) of global hit h, begin <= h < end
    template <typename hits_iterator>
    void locate(const uint64 begin, const uint64 end, hits_iterator hits)
    {
        const int64 n = int64(end - begin);
        #pragma omp parallel for schedule(dynamic, 1024)
        for (int64 h = 0; h < n; ++h)
        {
            const range_type pair = fmindex::filter_hit(begin + uint64(h), m_n_queries, m_slots.data(), m_ranges.data());
            hits[h] = make_vector(nvbio::locate(m_index, pair.x), pair.y);
        }
    }
    uint64            n_hits() const { return m_n_occurrences; }
    const range_type* ranges() const { return m_ranges.data(); }
    const uint64*     ranks()  const { return m_slots.data(); }
    const char*       last_path() const { return "host"; }

    uint32                  m_n_queries;
    index_type              m_index;
    uint64                  m_n_occurrences;
    std::vector<range_type> m_ranges;
    std::vector<uint64>     m_slots;
};

// ---------------------------------------------------------------------------------------- device
#if defined(__HIPCC__)
template <typename fm_index_type>
struct FMIndexFilter<device_tag, fm_index_type>
{
    typedef device_tag                                      system_tag;
    typedef fm_index_type                                   index_type;
    typedef typename index_type::index_type                 coord_type;
    static const uint32                                     coord_dim = vector_traits<coord_type>::DIM;
    typedef typename vector_type<coord_type, 2>::type       range_type;
    static const uint32                                     hit_dim = coord_dim * 2;
    typedef typename vector_type<coord_type, hit_dim>::type hit_type;

    FMIndexFilter() : m_n_queries(0), m_n_occurrences(0), m_path("none"), m_stream(0), m_tuned(false), m_line_native(true), m_dimer_for(nullptr), m_dimer_len(0), m_dimer_primary(0) {}

    /// the HIP stream rank() / locate() queue their work on (the reference uses the default stream of the current device)
    void set_stream(hipStream_t s) { m_stream = s; }
    /// whether the tuned execution may build and keep the line-native two-symbol index (default: yes; 3.7 bytes per SA row)
    void set_line_native(const bool on) { m_line_native = on; }
    const char* last_path() const { return m_path; }

    /// filter.h:139-170, filter_inl.h:268-300
    template <typename string_set_type>
    uint64 rank(const fm_index_type& index, const string_set_type& string_set)
    {
        m_n_queries = string_set.size();
        m_index     = index;
        m_n_occurrences = 0;
        m_tuned = false;
        if (m_n_queries == 0) { m_path = "none"; return 0; }
        range_type* ranges = m_ranges.reserve(m_n_queries);
        uint64*     slots  = m_slots.reserve(m_n_queries);
        typedef typename string_set_type::string_type string_type;
        if (!rank_tuned(string_set, ranges, slots,
                        std::integral_constant<bool, fmindex::production_layout<fm_index_type>::ok && nvbio::priv::packed_view<string_type>::ok>()))
        {
            const uint32 n = m_n_queries;
            hipLaunchKernelGGL((fmindex::filter_rank_kernel<fm_index_type, string_set_type>), dim3((n + 127u) / 128u), dim3(128), 0, m_stream, n, m_index, string_set, ranges, slots);
            fmindex::check(hipGetLastError(), "filter_rank_kernel");
            size_t tb = 0;
            fmindex::check(hipcub::DeviceScan::InclusiveSum(nullptr, tb, slots, slots, int(n), m_stream), "hipcub::DeviceScan::InclusiveSum");
            uint8* temp = m_temp.reserve(tb + 16u);
            fmindex::check(hipcub::DeviceScan::InclusiveSum(temp, tb, slots, slots, int(n), m_stream), "hipcub::DeviceScan::InclusiveSum");
            m_path = "generic";
        }
        fmindex::check(hipMemcpyAsync(&m_n_occurrences, slots + (m_n_queries - 1u), sizeof(uint64), hipMemcpyDeviceToHost, m_stream), "hipMemcpyAsync");
        fmindex::check(hipStreamSynchronize(m_stream), "hipStreamSynchronize");
        return m_n_occurrences;
    }

    /// filter.h:172-183, filter_inl.h:306-402.  hits: any device iterator of hit_type (a raw pointer, thrust::device_ptr, a
    /// device_vector's begin()).  The call returns after queueing the work, as the reference's does.
    template <typename hits_iterator>
    void locate(const uint64 begin, const uint64 end, hits_iterator hits)
    {
        if (end <= begin || m_n_queries == 0) return;
        if (end - begin > 0xFFFFFFFFull) throw std::runtime_error("FMIndexFilter::locate: more than 2^32 - 1 hits in one call");
        const uint32 n_hits = uint32(end - begin);
        if (locate_tuned(begin, end, hits, std::integral_constant<bool, fmindex::production_layout<fm_index_type>::ok && nvbio::priv::plain_iterator<hits_iterator>::ok>())) return;
        hipLaunchKernelGGL((fmindex::filter_locate_kernel<fm_index_type, hits_iterator>), dim3((n_hits + 127u) / 128u), dim3(128), 0, m_stream,
                           begin, n_hits, m_index, m_n_queries, m_slots.ptr, m_ranges.ptr, hits);
        fmindex::check(hipGetLastError(), "filter_locate_kernel");
        m_path = "generic";
    }

    uint64            n_hits() const { return m_n_occurrences; }
    const range_type* ranges() const { return m_ranges.ptr; }       ///< device memory
    const uint64*     ranks()  const { return m_slots.ptr; }        ///< device memory

private:
    template <typename string_set_type>
    bool rank_tuned(const string_set_type&, range_type*, uint64*, std::false_type) { return false; }
    template <typename hits_iterator>
    bool locate_tuned(const uint64, const uint64, hits_iterator, std::false_type) { return false; }

    template <typename string_set_type>
    bool rank_tuned(const string_set_type& string_set, range_type* ranges, uint64* slots, std::true_type)
    {
#if defined(NVBIO_HIP_COMPAT_FILTER_TUNED)
        typedef typename string_set_type::string_type string_type;
        typedef nvbio::priv::packed_view<string_type> view;
        if (!fmindex::production_layout<fm_index_type>::describe(m_index, m_fmi)) return false;
        attach_line_native();
        const uint32 n = m_n_queries;
        // the strings in place: a job table of (offset, length) into the caller's own words
        uint8* base = m_jobs.reserve(64u + uint64(n) * 12u + 16u);
        unsigned long long* bounds = reinterpret_cast<unsigned long long*>(base);
        uint64* begin = reinterpret_cast<uint64*>(base + 64u);
        uint32* len   = reinterpret_cast<uint32*>(begin + n);
        hipLaunchKernelGGL(fmindex::filter_init_bounds_kernel, dim3(1), dim3(64), 0, m_stream, bounds);
        hipLaunchKernelGGL((fmindex::filter_describe_kernel<string_set_type>), dim3((n + 127u) / 128u), dim3(128), 0, m_stream, n, string_set, begin, len, bounds);
        unsigned long long b[2];
        fmindex::check(hipMemcpyAsync(b, bounds, sizeof(b), hipMemcpyDeviceToHost, m_stream), "hipMemcpyAsync");
        fmindex::check(hipStreamSynchronize(m_stream), "hipStreamSynchronize");
        if (b[1] == 0ull) { b[0] = 0ull; b[1] = 1ull; }
        hipLaunchKernelGGL(fmindex::filter_rebase_kernel, dim3((n + 255u) / 256u), dim3(256), 0, m_stream, n, begin, uint64(b[0]) * (32u / view::BITS));
        nvbio_hip_string_set seeds;
        seeds.words = reinterpret_cast<const uint32*>(uintptr_t(b[0]) * 4u); seeds.n_words = b[1] - b[0];
        seeds.bits = view::BITS; seeds.big_endian = view::BE ? 1u : 0u;
        seeds.begin = begin; seeds.length = len; seeds.fixed_length = 0; seeds._pad = 0;
        const uint64 tb = nvbio_hip_fm_filter_temp_bytes(n);
        uint8* temp = m_temp.reserve(tb + 16u);
        const int err = nvbio_hip_fm_filter_rank(&m_fmi, &seeds, n, reinterpret_cast<uint32*>(ranges), slots, temp, tb, m_stream);
        if (err == 801) return false;
        fmindex::check(err, "nvbio_hip_fm_filter_rank");
        m_tuned = true; m_path = "tuned";
        return true;
#else
        (void)string_set; (void)ranges; (void)slots; return false;
#endif
    }
    template <typename hits_iterator>
    bool locate_tuned(const uint64 begin, const uint64 end, hits_iterator hits, std::true_type)
    {
#if defined(NVBIO_HIP_COMPAT_FILTER_TUNED)
        // the C-ABI description of the index is valid whenever describe() succeeded for the last rank(), whichever execution ranked
        if (!m_tuned && !fmindex::production_layout<fm_index_type>::describe(m_index, m_fmi)) return false;
        if (!m_tuned) attach_line_native();
        static_assert(sizeof(hit_type) == 8u, "32-bit coordinates");
        const int err = nvbio_hip_fm_filter_locate(&m_fmi, reinterpret_cast<const uint32*>(m_ranges.ptr), m_slots.ptr, m_n_queries, begin, end,
                                                   reinterpret_cast<uint32*>(nvbio::priv::plain_iterator<hits_iterator>::get(hits)), m_stream);
        if (err == 801) return false;
        fmindex::check(err, "nvbio_hip_fm_filter_locate");
        m_path = "tuned";
        return true;
#else
        (void)begin; (void)end; (void)hits; return false;
#endif
    }
#if defined(NVBIO_HIP_COMPAT_FILTER_TUNED)
    /// build (once per index seen) and attach the line-native two-symbol index; failures leave the reference layout in use
    void attach_line_native()
    {
        if (!m_line_native) return;
        if (m_dimer_for != m_fmi.bwt_occ || m_dimer_len != m_fmi.length || m_dimer_primary != m_fmi.primary ||
            m_dimer_L2[0] != m_fmi.L2[1] || m_dimer_L2[1] != m_fmi.L2[2] || m_dimer_L2[2] != m_fmi.L2[3])      // a different index rebuilt at the same address
        {
            m_dimer_for = nullptr;
            const uint64 bytes = nvbio_hip_fm_dimer_index_bytes(m_fmi.length), tb = nvbio_hip_fm_build_dimer_index_temp_bytes(m_fmi.length);
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || uint64(free_b) < 2u * (bytes + tb)) { m_line_native = false; return; }
            try {
                uint32* d = m_dimer.reserve(bytes / 4u + 64u);
                uint8*  t = m_temp.reserve(tb + 16u);
                if (nvbio_hip_fm_build_dimer_index(&m_fmi, d, t, tb, m_stream) != 0) { m_line_native = false; return; }
            } catch (const fmindex::hip_error&) { m_line_native = false; return; }
            m_dimer_for = m_fmi.bwt_occ; m_dimer_len = m_fmi.length; m_dimer_primary = m_fmi.primary;
            m_dimer_L2[0] = m_fmi.L2[1]; m_dimer_L2[1] = m_fmi.L2[2]; m_dimer_L2[2] = m_fmi.L2[3];
        }
        if (nvbio_hip_fm_attach_dimer_index(&m_fmi, m_dimer.ptr, m_stream) != 0) m_fmi.dimer = nullptr;
    }
    nvbio_hip_fmindex               m_fmi;
#endif
public:
    uint32                          m_n_queries;
    index_type                      m_index;
    uint64                          m_n_occurrences;
private:
    fmindex::device_array<range_type> m_ranges;
    fmindex::device_array<uint64>     m_slots;
    fmindex::device_array<uint8>      m_temp, m_jobs;
    fmindex::device_array<uint32>     m_dimer;
    const char*                       m_path;
    hipStream_t                       m_stream;
    bool                              m_tuned, m_line_native;
    const uint32*                     m_dimer_for; uint32 m_dimer_len, m_dimer_primary, m_dimer_L2[3] = { 0u, 0u, 0u };
};
#endif // __HIPCC__

/// filter.h:215-260: the two names callers use
template <typename fm_index_type> struct FMIndexFilterHost : public FMIndexFilter<host_tag, fm_index_type>
{
    typedef FMIndexFilter<host_tag, fm_index_type>  core_type;
    typedef typename core_type::system_tag          system_tag;
    typedef typename core_type::index_type          index_type;
    typedef typename core_type::coord_type          coord_type;
    typedef typename core_type::range_type          range_type;
    typedef typename core_type::hit_type            hit_type;
};
#if defined(__HIPCC__)
template <typename fm_index_type> struct FMIndexFilterDevice : public FMIndexFilter<device_tag, fm_index_type>
{
    typedef FMIndexFilter<device_tag, fm_index_type> core_type;
    typedef typename core_type::system_tag           system_tag;
    typedef typename core_type::index_type           index_type;
    typedef typename core_type::coord_type           coord_type;
    typedef typename core_type::range_type           range_type;
    typedef typename core_type::hit_type             hit_type;
};
#endif

} // namespace nvbio
