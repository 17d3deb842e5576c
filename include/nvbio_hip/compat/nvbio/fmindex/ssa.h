// compat/nvbio/fmindex/ssa.h -- the sampled suffix array of locate (nvbio/fmindex/ssa.h, ssa_inl.h:263-309, 486-504):
// SSA_index_multiple<K> keeps SA[k*K] for every K-th ROW (row 0, the '$' suffix, stored as -1);
// SSA_index_multiple_context<K,Iterator> is the view fm_index carries: fetch(i, r) / has(i) succeed on rows i % K == 0.
#pragma once
#include "../basic/types.h"
#include <vector>

namespace nvbio {

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

/// host storage, built from a full suffix array (rows 0..n, SA[0] = n for the '$' row) or by walking an FM-index
template <uint32 K, typename index_type = uint32>
struct SSA_index_multiple
{
    typedef index_type                                              value_type;
    typedef SSA_index_multiple_context<K, const index_type*>        context_type;
    SSA_index_multiple() : m_n(0) {}
    /// ssa_inl.h:263-276: from the suffix array of a text of n symbols (sa has n+1 entries)
    template <typename SAIterator>
    SSA_index_multiple(const index_type n, const SAIterator sa) : m_n(n), m_ssa((uint64(n) + K) / K)
    {
        for (uint64 k = 0; k < m_ssa.size(); ++k) m_ssa[k] = index_type(sa[k * K]);
        m_ssa[0] = index_type(-1);
    }
    context_type get_context() const { return context_type(m_ssa.data()); }
    index_type              m_n;
    std::vector<index_type> m_ssa;
};

} // namespace nvbio
