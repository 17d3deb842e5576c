// compat/nvbio/fmindex/ssa.h -- the sampled suffix array of locate (nvbio/fmindex/ssa.h:219-345, ssa_inl.h:263-504):
// SSA_index_multiple<K> keeps SA[k*K] for every K-th ROW (row 0, the '$' suffix, stored as -1);
// SSA_index_multiple_context<K,Iterator> is the view fm_index carries: fetch(i, r) / has(i) succeed on rows i % K == 0;
// SSA_index_multiple_device<K> is the same array in device memory, copied from the host one or built on the device from an
// FM-index alone.
//
// Building from an FM-index: the LF-mapping visits the rows in decreasing text order starting from row 0 (suffix n), so one
// walk of n steps meets every sampled row with its suffix known.  The device builder cuts that walk at the sampled rows: one
// lane per sampled row walks to the NEXT sampled row on its path (next[k], hops[k]) -- independent, at most a few K steps on
// random text -- and the host then follows the n/K links once, subtracting hops.
#pragma once
#include "../basic/types.h"
#include <vector>
#include <stdexcept>
#if defined(__HIPCC__)
#include <thrust/device_vector.h>
#include <thrust/host_vector.h>
#endif

namespace nvbio {

// the builders below walk an fm_index (fmindex.h, which includes this header)
template <typename TRankDictionary, typename TSuffixArray, typename TL2 = null_type> struct fm_index;
template <typename R, typename S, typename L> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE
typename fm_index<R, S, L>::index_type basic_inv_psi(const fm_index<R, S, L>& fmi, const typename fm_index<R, S, L>::index_type i);

template <uint32 K, typename Iterator = const uint32*>
struct SSA_index_multiple_context
{
    typedef typename std::iterator_traits<Iterator>::value_type index_type;
    typedef index_type                                          value_type;
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE SSA_index_multiple_context() {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE SSA_index_multiple_context(const Iterator ssa) : m_ssa(ssa) {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool fetch(const index_type i, index_type& r) const
    {
        if ((i & index_type(K - 1u)) != 0) return false;
        r = m_ssa[i / K];
        return true;
    }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool has(const index_type i) const { return (i & index_type(K - 1u)) == 0; }
    Iterator m_ssa;
};

template <uint32 K, typename index_type> struct SSA_index_multiple_device;

/// host storage, built from a full suffix array (rows 0..n, SA[0] = n for the '$' row) or by walking an FM-index
template <uint32 K, typename index_type = uint32>
struct SSA_index_multiple
{
    typedef index_type                                              value_type;
    typedef SSA_index_multiple_context<K, const index_type*>        context_type;
    typedef SSA_index_multiple_device<K, index_type>                device_type;
    typedef context_type                                            device_view_type;
    typedef context_type                                            plain_view_type;

    SSA_index_multiple() : m_n(0) {}
    /// ssa_inl.h:263-276: from the suffix array of a text of n symbols (sa has n+1 entries)
    template <typename SAIterator>
    SSA_index_multiple(const index_type n, const SAIterator sa) : m_n(n), m_ssa((uint64(n) + K) / K)
    {
        for (uint64 k = 0; k < m_ssa.size(); ++k) m_ssa[k] = index_type(sa[k * K]);
        m_ssa[0] = index_type(-1);
    }
    /// ssa_inl.h:279-309: from an FM-index whose own suffix array is not used (any TSuffixArray): one LF walk over the text
    template <typename R, typename S, typename L>
    SSA_index_multiple(const fm_index<R, S, L>& fmi) : m_n(fmi.length()), m_ssa((uint64(fmi.length()) + K) / K)
    {
        index_type row = 0, suffix = fmi.length();
        while (suffix > 0)
        {
            row = basic_inv_psi(fmi, row);
            --suffix;
            if ((row & index_type(K - 1u)) == 0) m_ssa[row / K] = suffix;
        }
        m_ssa[0] = index_type(-1);
    }
#if defined(__HIPCC__)
    SSA_index_multiple(const SSA_index_multiple_device<K, index_type>& ssa) { *this = ssa; }
    SSA_index_multiple& operator=(const SSA_index_multiple_device<K, index_type>& ssa)
    {
        m_n = ssa.m_n;
        thrust::host_vector<index_type> h = ssa.m_ssa;
        m_ssa.assign(h.begin(), h.end());
        return *this;
    }
#endif
    context_type get_context() const { return context_type(m_ssa.data()); }
    index_type              m_n;
    std::vector<index_type> m_ssa;
};

#if defined(__HIPCC__)
namespace priv {
/// for sampled row k*K: the next sampled row on its LF path and the number of steps to it
template <uint32 K, typename index_type, typename FMIndexType>
__global__ void ssa_links_kernel(const uint32 n_items, const FMIndexType fmi, index_type* next, index_type* hops)
{
    const uint32 k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_items) return;
    index_type row = index_type(k) * K, t = 0;
    do { row = basic_inv_psi(fmi, row); ++t; } while ((row & index_type(K - 1u)) != 0);
    next[k] = row / K;
    hops[k] = t;
}
} // namespace priv

template <uint32 K, typename index_type = uint32>
struct SSA_index_multiple_device
{
    typedef index_type                                              value_type;
    typedef SSA_index_multiple_context<K, const index_type*>        context_type;
    typedef context_type                                            device_view_type;
    typedef context_type                                            plain_view_type;

    SSA_index_multiple_device() : m_n(0) {}
    SSA_index_multiple_device(const SSA_index_multiple<K, index_type>& ssa) : m_n(ssa.m_n), m_ssa(ssa.m_ssa.begin(), ssa.m_ssa.end()) {}
    /// ssa_inl.h:375-470: from a device-resident FM-index
    template <typename R, typename S, typename L>
    SSA_index_multiple_device(const fm_index<R, S, L>& fmi) : m_n(0) { init(fmi); }

    template <typename FMIndexType>
    void init(const FMIndexType& fmi)
    {
        m_n = fmi.length();
        const uint32 n_items = uint32((uint64(m_n) + K) / K);
        thrust::device_vector<index_type> d_next(n_items), d_hops(n_items);
        hipLaunchKernelGGL((priv::ssa_links_kernel<K, index_type, FMIndexType>), dim3((n_items + 255u) / 256u), dim3(256), 0, 0,
                           n_items, fmi, thrust::raw_pointer_cast(d_next.data()), thrust::raw_pointer_cast(d_hops.data()));
        if (hipDeviceSynchronize() != hipSuccess) throw std::runtime_error("SSA_index_multiple_device: link kernel failed");
        const thrust::host_vector<index_type> next = d_next, hops = d_hops;
        thrust::host_vector<index_type> ssa(n_items, index_type(0));
        // row 0 holds suffix n; every link moves `hops` positions towards the start of the text.  The walk is a cycle of n + 1 rows
        // (the row of suffix 0 maps back to row 0), so the chain ends at the first link longer than what is left of the text.
        index_type k = 0, suffix = m_n;
        for (uint32 links = 0; hops[k] <= suffix; ++links)
        {
            if (links > n_items) throw std::runtime_error("SSA_index_multiple_device: index out of bounds");
            suffix -= hops[k];
            k = next[k];
            ssa[k] = suffix;
        }
        ssa[0] = index_type(-1);
        m_ssa = ssa;
    }
    context_type get_context() const { return context_type(thrust::raw_pointer_cast(m_ssa.data())); }
    index_type                        m_n;
    thrust::device_vector<index_type> m_ssa;
};
#endif

template <uint32 K, typename index_type>
typename SSA_index_multiple<K, index_type>::plain_view_type plain_view(const SSA_index_multiple<K, index_type>& v) { return v.get_context(); }
#if defined(__HIPCC__)
template <uint32 K, typename index_type>
typename SSA_index_multiple_device<K, index_type>::plain_view_type plain_view(const SSA_index_multiple_device<K, index_type>& v) { return v.get_context(); }
#endif

} // namespace nvbio
