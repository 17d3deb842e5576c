// compat/nvbio/basic/transform_iterator.h -- transform_iterator<Iterator,Functor> (nvbio/basic/transform_iterator.h:45-230):
// a read-only iterator yielding f(base[i]); nvBowtie complements seeds with it (mapping_inl.h:276,298,361).
#pragma once
#include "types.h"
#include "iterator.h"

namespace nvbio {

template <typename T, typename Transform>
struct transform_iterator
{
    typedef typename Transform::result_type                     value_type;
    typedef value_type                                          reference;
    typedef value_type                                          const_reference;
    typedef const value_type*                                   pointer;
    typedef typename std::iterator_traits<T>::difference_type   difference_type;
    typedef typename std::iterator_traits<T>::iterator_category iterator_category;

    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE transform_iterator() {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE transform_iterator(const T base, const Transform f) : m_base(base), m_f(f) {}

    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE value_type operator[](const uint32 i) const { return m_f(m_base[i]); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE value_type operator*() const { return m_f(m_base[0]); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE transform_iterator& operator++()    { ++m_base; return *this; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE transform_iterator  operator++(int) { transform_iterator r(*this); ++m_base; return r; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE transform_iterator& operator--()    { --m_base; return *this; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE transform_iterator  operator--(int) { transform_iterator r(*this); --m_base; return r; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE transform_iterator& operator+=(const difference_type d) { m_base = m_base + d; return *this; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE transform_iterator& operator-=(const difference_type d) { m_base = m_base - d; return *this; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE transform_iterator  operator+(const difference_type d) const { return transform_iterator(m_base + d, m_f); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE transform_iterator  operator-(const difference_type d) const { return transform_iterator(m_base - d, m_f); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE difference_type     operator-(const transform_iterator o) const { return m_base - o.m_base; }

    T         m_base;
    Transform m_f;
};
template <typename T, typename F> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE transform_iterator<T, F> make_transform_iterator(const T it, const F f) { return transform_iterator<T, F>(it, f); }
template <typename T, typename F> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool operator==(const transform_iterator<T, F> a, const transform_iterator<T, F> b) { return a.m_base == b.m_base; }
template <typename T, typename F> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool operator!=(const transform_iterator<T, F> a, const transform_iterator<T, F> b) { return a.m_base != b.m_base; }
template <typename T, typename F> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool operator< (const transform_iterator<T, F> a, const transform_iterator<T, F> b) { return a.m_base <  b.m_base; }
template <typename T, typename F> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool operator> (const transform_iterator<T, F> a, const transform_iterator<T, F> b) { return a.m_base >  b.m_base; }

} // namespace nvbio
