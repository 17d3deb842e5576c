// compat/nvbio/basic/index_transform_iterator.h -- index_transform_iterator<Iterator,IndexFunctor>
// (nvbio/basic/index_transform_iterator.h:46-245): element i is base[f(position + i)]; nvBowtie reads a seed forwards
// (OffsetXform) or backwards (ReverseXform) through it (mapping_inl.h:249-256).
#pragma once
#include "types.h"
#include "iterator.h"

namespace nvbio {

template <typename T, typename Transform>
struct index_transform_iterator
{
    typedef index_transform_iterator<T, Transform>              this_type;
    typedef typename std::iterator_traits<T>::value_type        value_type;
    typedef value_type                                          reference;
    typedef value_type                                          const_reference;
    typedef const value_type*                                   pointer;
    typedef typename std::iterator_traits<T>::difference_type   difference_type;
    typedef std::random_access_iterator_tag                     iterator_category;

    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE index_transform_iterator() : m_index(0) {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE index_transform_iterator(const T base, const Transform f, const difference_type i = 0) : m_base(base), m_f(f), m_index(i) {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE index_transform_iterator(const index_transform_iterator& o) : m_base(o.m_base), m_f(o.m_f), m_index(o.m_index) {}
    /// the functors carry const members, so assignment re-creates the object in place
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE index_transform_iterator& operator=(const index_transform_iterator& o)
    { if (this != &o) { this->~index_transform_iterator(); new (this) index_transform_iterator(o); } return *this; }

    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE value_type operator[](const uint32 i) const { return m_base[m_f(m_index + i)]; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE value_type operator*() const { return m_base[m_f(m_index)]; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE void       set(const value_type v) { m_base[m_f(m_index)] = v; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE this_type& operator++()    { ++m_index; return *this; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE this_type  operator++(int) { this_type r(*this); ++m_index; return r; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE this_type& operator--()    { --m_index; return *this; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE this_type  operator--(int) { this_type r(*this); --m_index; return r; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE this_type& operator+=(const difference_type d) { m_index += d; return *this; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE this_type& operator-=(const difference_type d) { m_index -= d; return *this; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE this_type  operator+(const difference_type d) const { return this_type(m_base, m_f, m_index + d); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE this_type  operator-(const difference_type d) const { return this_type(m_base, m_f, m_index - d); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE difference_type operator-(const this_type o) const { return m_index - o.m_index; }

    T               m_base;
    Transform       m_f;
    difference_type m_index;
};
template <typename T, typename F> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE index_transform_iterator<T, F> make_index_transform_iterator(const T it, const F f) { return index_transform_iterator<T, F>(it, f); }
template <typename T, typename F> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool operator==(const index_transform_iterator<T, F> a, const index_transform_iterator<T, F> b) { return a.m_index == b.m_index; }
template <typename T, typename F> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool operator!=(const index_transform_iterator<T, F> a, const index_transform_iterator<T, F> b) { return a.m_index != b.m_index; }
template <typename T, typename F> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool operator< (const index_transform_iterator<T, F> a, const index_transform_iterator<T, F> b) { return a.m_index <  b.m_index; }
template <typename T, typename F> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool operator> (const index_transform_iterator<T, F> a, const index_transform_iterator<T, F> b) { return a.m_index >  b.m_index; }

} // namespace nvbio
