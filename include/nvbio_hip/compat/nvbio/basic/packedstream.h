// compat/nvbio/basic/packedstream.h -- PackedStream<InputStream,Symbol,SYMBOL_SIZE,BIG_ENDIAN,IndexType>
// (nvbio/basic/packedstream.h:186-330, packedstream_inl.h:336-400): a random-access iterator over symbols of
// SYMBOL_SIZE bits packed in the words of an underlying word iterator.  Symbol s lives in word s / per_word at bit
//   big-endian   : word_bits - SYMBOL_SIZE * (1 + s % per_word)
//   little-endian: SYMBOL_SIZE * (s % per_word)
// which is the layout every packed string of the hot path uses (reads 4-bit BE, genome / BWT 2-bit BE,
// sw-benchmark's references 2-bit LE).
#pragma once
#include "types.h"
#include "strided_iterator.h"     // (the reference's packedstream.h brings it in, and callers rely on that: persist.cu:215)
#include "numbers.h"
#include "iterator.h"
// <endian.h> defines BIG_ENDIAN as a macro; the reference drops it (packedstream.h:38-40) because PackedStream has a member of that name
#if defined(BIG_ENDIAN)
#undef BIG_ENDIAN
#endif

namespace nvbio {

template <typename Stream> struct PackedStreamRef;

template <typename InputStream, typename Symbol, uint32 SYMBOL_SIZE_T, bool BIG_ENDIAN_T, typename IndexType = uint32>
struct PackedStream
{
    typedef PackedStream<InputStream, Symbol, SYMBOL_SIZE_T, BIG_ENDIAN_T, IndexType>  This;
    static const uint32 SYMBOL_SIZE       = SYMBOL_SIZE_T;
    static const uint32 SYMBOL_COUNT      = 1u << SYMBOL_SIZE_T;
    static const uint32 SYMBOL_MASK       = SYMBOL_COUNT - 1u;
    static const uint32 BIG_ENDIAN        = BIG_ENDIAN_T;            ///< the reference's name for it (packedstream.h:202)
    static const bool   IS_BIG_ENDIAN     = BIG_ENDIAN_T;
    static const uint32 ALPHABET_SIZE     = SYMBOL_COUNT;

    typedef typename unsigned_type<IndexType>::type                          index_type;
    typedef typename signed_type<IndexType>::type                            sindex_type;
    typedef typename std::iterator_traits<InputStream>::value_type           storage_type;
    typedef priv::vec_comp<storage_type>                                     storage_comp;        ///< uint4 / uint2 storage is seen as its scalar words
    typedef typename storage_comp::type                                      word_type;
    static const uint32 WORD_SIZE         = uint32(8u * sizeof(word_type));
    static const uint32 SYMBOLS_PER_WORD  = WORD_SIZE / SYMBOL_SIZE_T;

    typedef InputStream                      stream_type;
    typedef InputStream                      storage_iterator;
    typedef Symbol                           symbol_type;
    typedef Symbol                           value_type;
    typedef PackedStreamRef<This>            reference;
    typedef Symbol                           const_reference;
    typedef reference*                       pointer;
    typedef std::random_access_iterator_tag  iterator_category;
    typedef sindex_type                      difference_type;
    typedef sindex_type                      distance_type;
    typedef This                             iterator;
    typedef This                             const_iterator;
    typedef This                             forward_iterator;

    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE PackedStream() : m_index(0) {}
    template <typename UInputStream>
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE explicit PackedStream(const UInputStream stream, const index_type index = 0) : m_stream(static_cast<InputStream>(stream)), m_index(index) {}
    template <typename UInputStream, typename USymbol>
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE PackedStream(const PackedStream<UInputStream, USymbol, SYMBOL_SIZE_T, BIG_ENDIAN_T, IndexType>& other)
        : m_stream(static_cast<InputStream>(other.stream())), m_index(other.index()) {}

    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE Symbol get(const index_type i) const
    {
        const index_type s = m_index + i;
        const word_type w = word(s / SYMBOLS_PER_WORD);
        const uint32 k = uint32(s % SYMBOLS_PER_WORD);
        const uint32 sh = BIG_ENDIAN_T ? (WORD_SIZE - SYMBOL_SIZE_T * (k + 1u)) : (SYMBOL_SIZE_T * k);
        return Symbol((w >> sh) & word_type(SYMBOL_MASK));
    }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE void set(const index_type i, const Symbol v)
    {
        const index_type s = m_index + i;
        const uint32 k = uint32(s % SYMBOLS_PER_WORD);
        const uint32 sh = BIG_ENDIAN_T ? (WORD_SIZE - SYMBOL_SIZE_T * (k + 1u)) : (SYMBOL_SIZE_T * k);
        word_type w = word(s / SYMBOLS_PER_WORD);
        w = (w & ~(word_type(SYMBOL_MASK) << sh)) | (word_type(uint32(v) & SYMBOL_MASK) << sh);
        set_word(s / SYMBOLS_PER_WORD, w);
    }
    /// scalar word j of the underlying storage (absolute: the stream's own symbol offset is not applied)
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE word_type word(const index_type j) const
    {
        if (storage_comp::N == 1u) return storage_comp::get(m_stream[j], 0u);
        return storage_comp::get(m_stream[j / storage_comp::N], uint32(j % storage_comp::N));
    }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE void set_word(const index_type j, const word_type w)
    {
        storage_type e = m_stream[j / storage_comp::N];
        storage_comp::put(e, uint32(j % storage_comp::N), w);
        m_stream[j / storage_comp::N] = e;
    }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE Symbol get() const { return get(0); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE void   set(const Symbol v) { set(0, v); }

    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE Symbol    operator[](const index_type i) const { return get(i); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE reference operator[](const index_type i)       { return reference(*this + sindex_type(i)); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE reference operator*() const { return reference(*this); }

    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE iterator    begin()  const { return *this; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE InputStream stream() const { return m_stream; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE index_type  index()  const { return m_index; }

    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE This& operator++()    { ++m_index; return *this; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE This  operator++(int) { This r(*this); ++m_index; return r; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE This& operator--()    { --m_index; return *this; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE This  operator--(int) { This r(*this); --m_index; return r; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE This& operator+=(const sindex_type d) { m_index = index_type(sindex_type(m_index) + d); return *this; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE This& operator-=(const sindex_type d) { m_index = index_type(sindex_type(m_index) - d); return *this; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE This  operator+(const sindex_type d) const { This r(*this); r += d; return r; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE This  operator-(const sindex_type d) const { This r(*this); r -= d; return r; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE sindex_type operator-(const This o) const { return sindex_type(m_index) - sindex_type(o.m_index); }

    InputStream m_stream;
    index_type  m_index;
};

template <typename I, typename S, uint32 B, bool E, typename X> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool operator< (const PackedStream<I,S,B,E,X>& a, const PackedStream<I,S,B,E,X>& b) { return a.index() <  b.index(); }
template <typename I, typename S, uint32 B, bool E, typename X> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool operator> (const PackedStream<I,S,B,E,X>& a, const PackedStream<I,S,B,E,X>& b) { return a.index() >  b.index(); }
template <typename I, typename S, uint32 B, bool E, typename X> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool operator==(const PackedStream<I,S,B,E,X>& a, const PackedStream<I,S,B,E,X>& b) { return a.index() == b.index(); }
template <typename I, typename S, uint32 B, bool E, typename X> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool operator!=(const PackedStream<I,S,B,E,X>& a, const PackedStream<I,S,B,E,X>& b) { return a.index() != b.index(); }

/// the assignable reference operator* / operator[] hand out
template <typename Stream>
struct PackedStreamRef
{
    typedef typename Stream::symbol_type Symbol;
    typedef Symbol symbol_type;
    typedef Symbol value_type;
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE PackedStreamRef(Stream stream) : m_stream(stream) {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE PackedStreamRef& operator=(const PackedStreamRef& ref) { m_stream.set(ref.m_stream.get()); return *this; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE PackedStreamRef& operator=(const Symbol s) { m_stream.set(s); return *this; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE operator Symbol() const { return m_stream.get(); }
    Stream m_stream;
};

template <typename T> struct stream_traits { typedef uint32 index_type; typedef char symbol_type; static const uint32 SYMBOL_SIZE = 8u; static const uint32 SYMBOL_COUNT = 256u; };
template <typename T> struct stream_traits<T*> { typedef uint32 index_type; typedef T symbol_type; static const uint32 SYMBOL_SIZE = uint32(8u * sizeof(T)); static const uint32 SYMBOL_COUNT = 256u; };
template <typename T> struct stream_traits<const T*> { typedef uint32 index_type; typedef T symbol_type; static const uint32 SYMBOL_SIZE = uint32(8u * sizeof(T)); static const uint32 SYMBOL_COUNT = 256u; };
template <typename I, typename S, uint32 B, bool E, typename X>
struct stream_traits< PackedStream<I, S, B, E, X> > { typedef X index_type; typedef S symbol_type; static const uint32 SYMBOL_SIZE = B; static const uint32 SYMBOL_COUNT = 1u << B; };
template <typename I, typename S, uint32 B, bool E, typename X>
struct string_traits< PackedStream<I, S, B, E, X> > { typedef S value_type; typedef X index_type; };

/// an iterator over uint4 words read as a stream of uint32 components (packedstream.h:610-635)
template <typename IteratorType>
struct uint4_as_uint32_iterator
{
    typedef uint32                                                          value_type;
    typedef value_type*                                                     pointer;
    typedef value_type                                                      reference;
    typedef typename std::iterator_traits<IteratorType>::difference_type    difference_type;
    typedef typename std::iterator_traits<IteratorType>::iterator_category  iterator_category;
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint4_as_uint32_iterator() {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint4_as_uint32_iterator(const IteratorType it) : m_it(it) {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE value_type operator[](const uint32 i) const { const uint4 w = m_it[i >> 2]; const uint32 c = i & 3u; return c == 0 ? w.x : c == 1 ? w.y : c == 2 ? w.z : w.w; }
    IteratorType m_it;
};

/// pack a symbol range into a stream (assign(), packedstream_inl.h)
template <typename InputIterator, typename I, typename S, uint32 B, bool E, typename X>
NVBIO_HOST_DEVICE inline void assign(const X n, InputIterator in, PackedStream<I, S, B, E, X> out)
{
    for (X i = 0; i < n; ++i) out.set(i, S(in[i]));
}

} // namespace nvbio

namespace std {
template <typename I, typename S, nvbio::uint32 B, bool E, typename X>
struct iterator_traits< nvbio::PackedStream<I, S, B, E, X> > {
    typedef typename nvbio::PackedStream<I, S, B, E, X>::iterator_category iterator_category;
    typedef S value_type;
    typedef typename nvbio::PackedStream<I, S, B, E, X>::difference_type difference_type;
    typedef typename nvbio::PackedStream<I, S, B, E, X>::pointer pointer;
    typedef typename nvbio::PackedStream<I, S, B, E, X>::reference reference;
};
} // namespace std
