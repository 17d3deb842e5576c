// compat/nvbio/basic/deinterleaved_iterator.h -- deinterleaved_iterator<STRIDE,WHICH,BaseIterator>
// (nvbio/basic/deinterleaved_iterator.h): element i of the WHICH-th of STRIDE interleaved streams.  With a uint4 base
// it yields the 32-bit words of the chosen uint4s: word w of stream WHICH = base[(w/4)*STRIDE + WHICH].{x,y,z,w}[w%4] --
// the form nvBowtie uses to see the bwt (WHICH 0) and occ (WHICH 1) halves of the interleaved index
// (nvbio/io/fmindex/fmindex.h:159-174).
#pragma once
#include "types.h"
#include "iterator.h"

namespace nvbio {

template <uint32 STRIDE, uint32 WHICH, typename BaseIterator>
struct deinterleaved_iterator
{
    typedef typename std::iterator_traits<BaseIterator>::value_type  vector_type_;
    typedef priv::vec_comp<vector_type_>                             comp;
    typedef typename comp::type                                      value_type;
    typedef value_type                                               reference;
    typedef const value_type*                                        pointer;
    typedef int64                                                    difference_type;
    typedef std::random_access_iterator_tag                          iterator_category;

    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE deinterleaved_iterator() : m_off(0) {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE deinterleaved_iterator(const BaseIterator it, const uint64 off = 0) : m_it(it), m_off(off) {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE value_type operator[](const uint64 i) const
    {
        const uint64 k = m_off + i;
        return comp::get(m_it[(k / comp::N) * STRIDE + WHICH], uint32(k % comp::N));
    }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE value_type operator*() const { return (*this)[0]; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE deinterleaved_iterator operator+(const difference_type d) const { return deinterleaved_iterator(m_it, uint64(int64(m_off) + d)); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE deinterleaved_iterator operator-(const difference_type d) const { return deinterleaved_iterator(m_it, uint64(int64(m_off) - d)); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE difference_type operator-(const deinterleaved_iterator o) const { return int64(m_off) - int64(o.m_off); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE BaseIterator base() const { return m_it; }
    BaseIterator m_it;
    uint64       m_off;
};

} // namespace nvbio
