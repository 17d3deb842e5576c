// compat/nvbio/basic/strided_iterator.h -- strided_iterator<T> (every stride-th element of a base iterator) and
// block_strided_iterator<BLOCKSIZE,T,LAYOUT> (blocks of BLOCKSIZE elements a stride apart, row- or column-major)
// (nvbio/basic/strided_iterator.h:59-290).  nvBowtie lays its per-read queues and hit links out through these.
#pragma once
#include "types.h"
#include "iterator.h"

namespace nvbio {

template <typename T>
struct strided_iterator
{
    typedef typename std::iterator_traits<T>::value_type         value_type;
    typedef typename std::iterator_traits<T>::reference          reference;
    typedef typename to_const<reference>::type                   const_reference;
    typedef typename std::iterator_traits<T>::pointer            pointer;
    typedef typename std::iterator_traits<T>::difference_type    difference_type;
    typedef typename std::iterator_traits<T>::iterator_category  iterator_category;

    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE strided_iterator() : m_vec(), m_stride(0) {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE strided_iterator(T vec, const uint32 stride) : m_vec(vec), m_stride(stride) {}

    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE const_reference operator*() const { return *m_vec; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE reference       operator*()       { return *m_vec; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE const_reference operator[](const uint32 i) const { return m_vec[i * m_stride]; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE reference       operator[](const uint32 i)       { return m_vec[i * m_stride]; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE strided_iterator  operator+(const uint32 i) const { return strided_iterator(m_vec + i * m_stride, m_stride); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE difference_type   operator-(const strided_iterator it) const { return (m_vec - it.m_vec) / difference_type(m_stride); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE strided_iterator& operator++() { m_vec += m_stride; return *this; }

    T      m_vec;
    uint32 m_stride;
};
template <typename T> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE strided_iterator<T> make_strided_iterator(T it, const uint32 stride) { return strided_iterator<T>(it, stride); }
template <typename T> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool operator==(const strided_iterator<T> a, const strided_iterator<T> b) { return a.m_vec == b.m_vec && a.m_stride == b.m_stride; }
template <typename T> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool operator!=(const strided_iterator<T> a, const strided_iterator<T> b) { return !(a == b); }
template <typename T> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool operator< (const strided_iterator<T> a, const strided_iterator<T> b) { return a.m_vec <  b.m_vec; }
template <typename T> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool operator<=(const strided_iterator<T> a, const strided_iterator<T> b) { return a.m_vec <= b.m_vec; }
template <typename T> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool operator> (const strided_iterator<T> a, const strided_iterator<T> b) { return a.m_vec >  b.m_vec; }
template <typename T> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool operator>=(const strided_iterator<T> a, const strided_iterator<T> b) { return a.m_vec >= b.m_vec; }

enum block_strided_layout { ROW_MAJOR_LAYOUT = 0u, COLUMN_MAJOR_LAYOUT = 1u };

/// it[j] = base[(j / BLOCKSIZE) * stride + j % BLOCKSIZE] (row-major) or base[(j % BLOCKSIZE) * stride + j / BLOCKSIZE] (column-major)
template <uint32 BLOCKSIZE, typename T, block_strided_layout LAYOUT = ROW_MAJOR_LAYOUT>
struct block_strided_iterator
{
    typedef typename std::iterator_traits<T>::value_type         value_type;
    typedef typename std::iterator_traits<T>::reference          reference;
    typedef typename to_const<reference>::type                   const_reference;
    typedef typename std::iterator_traits<T>::pointer            pointer;
    typedef typename std::iterator_traits<T>::difference_type    difference_type;
    typedef typename std::iterator_traits<T>::iterator_category  iterator_category;

    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE block_strided_iterator() : m_vec(), m_offset(0), m_stride(0) {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE block_strided_iterator(T vec, const uint32 stride, const uint32 offset = 0) : m_vec(vec), m_offset(offset), m_stride(stride) {}

    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 slot(const uint32 i) const
    {
        const uint32 j = i + m_offset;
        return LAYOUT == ROW_MAJOR_LAYOUT ? (j / BLOCKSIZE) * m_stride + (j % BLOCKSIZE) : (j % BLOCKSIZE) * m_stride + (j / BLOCKSIZE);
    }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE const_reference operator*() const { return m_vec[m_offset]; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE const_reference operator[](const uint32 i) const { return m_vec[slot(i)]; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE reference       operator[](const uint32 i)       { return m_vec[slot(i)]; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE block_strided_iterator  operator+(const uint32 i) const { return block_strided_iterator(m_vec, m_stride, m_offset + i); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE difference_type         operator-(const block_strided_iterator it) const { return (m_vec + m_offset) - (it.m_vec + it.m_offset); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE block_strided_iterator& operator++() { ++m_offset; return *this; }

    T      m_vec;
    uint32 m_offset;
    uint32 m_stride;
};
template <uint32 B, typename T, block_strided_layout L> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE
bool operator==(const block_strided_iterator<B, T, L> a, const block_strided_iterator<B, T, L> b) { return a.m_vec == b.m_vec && a.m_offset == b.m_offset && a.m_stride == b.m_stride; }
template <uint32 B, typename T, block_strided_layout L> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE
bool operator!=(const block_strided_iterator<B, T, L> a, const block_strided_iterator<B, T, L> b) { return !(a == b); }

} // namespace nvbio
