// compat/nvbio/strings/string_set.h -- the two string-set views the batch functions are called with
// (nvbio/strings/string_set.h:380-560): strings inside one symbol stream, addressed by n+1 offsets
// (ConcatenatedStringSet) or by n (begin,end) ranges (SparseStringSet).  operator[] yields a vector_view.
#pragma once
#include "../basic/vector_view.h"

namespace nvbio {

struct concatenated_string_set_tag {};
struct sparse_string_set_tag {};

template <typename StringIterator, typename OffsetIterator>
struct ConcatenatedStringSet
{
    typedef concatenated_string_set_tag                                 string_set_tag;
    typedef typename std::iterator_traits<StringIterator>::value_type   symbol_type;
    typedef vector_view<StringIterator>                                 string_type;
    typedef StringIterator                                              symbol_iterator;
    typedef OffsetIterator                                              offset_iterator;

    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE ConcatenatedStringSet() : m_size(0) {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE ConcatenatedStringSet(const uint32 size, const StringIterator string, const OffsetIterator offsets)
        : m_size(size), m_string(string), m_offsets(offsets) {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 size() const { return m_size; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE string_type operator[](const uint32 i) const
    {
        const typename std::iterator_traits<OffsetIterator>::value_type o = m_offsets[i];
        return string_type(uint32(m_offsets[i + 1] - o), m_string + o);
    }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE symbol_iterator base_string() const { return m_string; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE offset_iterator offsets() const { return m_offsets; }

    uint32         m_size;
    StringIterator m_string;
    OffsetIterator m_offsets;
};
template <typename StringIterator, typename OffsetIterator>
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE ConcatenatedStringSet<StringIterator, OffsetIterator>
make_concatenated_string_set(const uint32 size, const StringIterator string, const OffsetIterator offsets)
{ return ConcatenatedStringSet<StringIterator, OffsetIterator>(size, string, offsets); }

template <typename StringIterator, typename RangeIterator>
struct SparseStringSet
{
    typedef sparse_string_set_tag                                       string_set_tag;
    typedef typename std::iterator_traits<StringIterator>::value_type   symbol_type;
    typedef vector_view<StringIterator>                                 string_type;
    typedef StringIterator                                              symbol_iterator;
    typedef RangeIterator                                               range_iterator;

    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE SparseStringSet() : m_size(0) {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE SparseStringSet(const uint32 size, const StringIterator string, const RangeIterator ranges)
        : m_size(size), m_string(string), m_ranges(ranges) {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 size() const { return m_size; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE string_type operator[](const uint32 i) const
    {
        const uint2 r = m_ranges[i];
        return string_type(r.y - r.x, m_string + r.x);
    }
    uint32         m_size;
    StringIterator m_string;
    RangeIterator  m_ranges;
};
template <typename StringIterator, typename RangeIterator>
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE SparseStringSet<StringIterator, RangeIterator>
make_sparse_string_set(const uint32 size, const StringIterator string, const RangeIterator ranges)
{ return SparseStringSet<StringIterator, RangeIterator>(size, string, ranges); }

} // namespace nvbio
