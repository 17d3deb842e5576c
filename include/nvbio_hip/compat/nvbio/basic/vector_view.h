// compat/nvbio/basic/vector_view.h -- vector_view<Iterator,IndexType> (nvbio/basic/vector_view.h:80-230): a sized
// view over any random-access iterator; the string type the alignment functions take.
#pragma once
#include "types.h"

namespace nvbio {

namespace priv {
/// what a const view hands out: `const T&` into the storage when the iterator refers to real objects (vector_view.h:96: to_const<reference>
/// -- nvBowtie's non-randomized selection keeps `&deque.top()` and pops SA rows off that hit in place, select_inl.h:108-127, so a copy
/// here is a dangling pointer there), the value where the iterator's reference is a proxy (packed streams)
template <typename R, typename V> struct view_const_reference            { typedef V        type; };
template <typename T, typename V> struct view_const_reference<T&, V>       { typedef const T& type; };
template <typename T, typename V> struct view_const_reference<const T&, V> { typedef const T& type; };
} // namespace priv

template <typename Iterator, typename IndexType = uint32>
struct vector_view
{
    typedef Iterator                                                iterator;
    typedef Iterator                                                const_iterator;
    typedef typename std::iterator_traits<Iterator>::value_type     value_type;
    typedef typename std::iterator_traits<Iterator>::reference      reference;
    typedef typename priv::view_const_reference<reference, value_type>::type const_reference;
    typedef IndexType                                               index_type;
    typedef IndexType                                               size_type;
    typedef typename std::iterator_traits<Iterator>::pointer         pointer;
    typedef typename std::iterator_traits<Iterator>::difference_type difference_type;
    typedef std::random_access_iterator_tag                         iterator_category;
    typedef Iterator                                                forward_iterator;
    typedef vector_view<Iterator, IndexType>                        plain_view_type;
    typedef vector_view<Iterator, IndexType>                        const_plain_view_type;

    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE vector_view() : m_size(0) {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE vector_view(const IndexType size, Iterator vec) : m_size(size), m_vec(vec) {}

    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE void      resize(const uint32 sz) { m_size = sz; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE void      clear() { m_size = 0; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE void      push_back(const value_type& v) { m_vec[m_size++] = v; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE void      pop_back() { --m_size; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE IndexType size() const { return m_size; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE IndexType length() const { return m_size; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool      empty() const { return m_size == 0; }

    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE const_reference operator[](const IndexType i) const { return m_vec[i]; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE reference       operator[](const IndexType i)       { return m_vec[i]; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE const_reference front() const { return m_vec[0]; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE const_reference back()  const { return m_vec[m_size - 1]; }

    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE Iterator begin() const { return m_vec; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE Iterator end()   const { return m_vec + m_size; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE Iterator base()  const { return m_vec; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE reference operator*() const { return *m_vec; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE operator Iterator() const { return m_vec; }            // a view decays to its iterator (vector_view.h:199)

    IndexType m_size;
    Iterator  m_vec;
};

template <typename Iterator, typename I> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 length(const vector_view<Iterator, I>& v) { return uint32(v.length()); }
template <typename T> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 length(const T& s) { return uint32(s.length()); }
template <typename T>             NVBIO_FORCEINLINE NVBIO_HOST_DEVICE T raw_pointer(const vector_view<T>& v) { return v.base(); }
template <typename T, typename I> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE T begin(const vector_view<T, I>& v) { return v.begin(); }
template <typename T> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE T*       begin(T* v)       { return v; }
template <typename T> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE const T* begin(const T* v) { return v; }
template <typename Iterator, typename I> struct string_traits< vector_view<Iterator, I> > {
    typedef typename vector_view<Iterator, I>::value_type value_type; typedef I index_type;
};

} // namespace nvbio
