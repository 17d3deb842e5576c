// compat/nvbio/basic/cached_iterator.h -- const_cached_iterator<It> (nvbio/basic/cached_iterator.h).  The reference keeps
// the last fetched word in a register to save texture fetches; on gfx950 the L1/K$ serve repeated word reads, so the
// wrapper only forwards (same values, same interface).
#pragma once
#include "types.h"
#include "iterator.h"

namespace nvbio {

template <typename InputStream>
struct const_cached_iterator
{
    typedef typename std::iterator_traits<InputStream>::value_type       value_type;
    typedef value_type                                                    reference;
    typedef const value_type*                                             pointer;
    typedef typename std::iterator_traits<InputStream>::difference_type  difference_type;
    typedef std::random_access_iterator_tag                               iterator_category;

    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE const_cached_iterator() {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE const_cached_iterator(InputStream stream) : m_stream(stream) {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE value_type operator[](const uint64 i) const { return m_stream[i]; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE value_type operator*() const { return *m_stream; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE const_cached_iterator operator+(const difference_type d) const { return const_cached_iterator(m_stream + d); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE const_cached_iterator operator-(const difference_type d) const { return const_cached_iterator(m_stream - d); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE difference_type operator-(const const_cached_iterator o) const { return m_stream - o.m_stream; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE InputStream base() const { return m_stream; }
    InputStream m_stream;
};
template <typename It> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE const_cached_iterator<It> make_const_cached_iterator(It it) { return const_cached_iterator<It>(it); }

} // namespace nvbio
