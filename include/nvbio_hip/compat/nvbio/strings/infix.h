// compat/nvbio/strings/infix.h -- windows of strings addressed by coordinates (nvbio/strings/infix.h:43-597): an Infix is a
// string plus a (begin,end) pair -- or a (string id, begin, end, -) quadruple when it was cut out of a string SET --, an InfixSet
// is an array of such coordinates over one string (2 coordinates) or over a string set (4 coordinates).  examples/fmmap/fmmap.cu:112-149
// builds its seeds as an InfixSet over the read string-set and hands it to FMIndexFilter::rank.
//
// Own layout: one Infix template holding (string, first, last, id); the coordinate dimension only decides how the four numbers are
// read out of / written back into the caller's coordinate type (priv::infix_coords<>).  An Infix over a packed-word string is itself
// recognised as a packed window (priv::packed_view, basic/packed_view.h), so seed sets reach the tuned FM-index kernels in place.
#pragma once
#include "../basic/types.h"
#include "../basic/vector_view.h"
#include "../basic/packed_view.h"
#include "../fmindex/rank_dictionary.h"     // vector_traits
#include "string_set.h"
#include <utility>

namespace nvbio {

typedef uint2       uint32_2;
typedef uint4       uint32_4;

typedef uint32_2    string_infix_coord_type;            ///< (begin, end)
typedef uint64_2    long_string_infix_coord_type;
typedef uint32_4    string_set_infix_coord_type;        ///< (string id, begin, end, unused)
typedef uint64_4    long_string_set_infix_coord_type;

namespace priv {
/// how a coordinate type spells (id, begin, end)
template <typename C, uint32 DIM = vector_traits<C>::DIM> struct infix_coords {};
template <typename C> struct infix_coords<C, 2u>
{
    typedef typename vector_traits<C>::value_type index_type;
    static const bool in_set = false;
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE static index_type id(const C&)      { return index_type(0); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE static index_type first(const C& c) { return c.x; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE static index_type last(const C& c)  { return c.y; }
};
template <typename C> struct infix_coords<C, 4u>
{
    typedef typename vector_traits<C>::value_type index_type;
    static const bool in_set = true;
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE static index_type id(const C& c)    { return c.x; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE static index_type first(const C& c) { return c.y; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE static index_type last(const C& c)  { return c.z; }
};
} // namespace priv

// the accessors on bare coordinates (infix.h:429-495)
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 infix_begin(const uint32_2& c) { return c.x; }
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 infix_end(const uint32_2& c)   { return c.y; }
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 length(const uint32_2& c)      { return c.y - c.x; }
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint64 infix_begin(const uint64_2& c) { return c.x; }
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint64 infix_end(const uint64_2& c)   { return c.y; }
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint64 length(const uint64_2& c)      { return c.y - c.x; }
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 string_id(const uint32_4& c)   { return c.x; }
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 infix_begin(const uint32_4& c) { return c.y; }
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 infix_end(const uint32_4& c)   { return c.z; }
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 length(const uint32_4& c)      { return c.z - c.y; }
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint64 string_id(const uint64_4& c)   { return c.x; }
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint64 infix_begin(const uint64_4& c) { return c.y; }
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint64 infix_end(const uint64_4& c)   { return c.z; }
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint64 length(const uint64_4& c)      { return c.z - c.y; }

/// symbols [first, last) of a string, remembering the coordinates it was cut with (infix.h:259-296)
template <typename StringType, typename CoordType>
struct Infix
{
    typedef priv::infix_coords<CoordType>                                   coord_access;
    typedef StringType                                                      string_type;
    typedef CoordType                                                       coord_type;
    typedef typename coord_access::index_type                               index_type;
    typedef typename vector_type<index_type, 2>::type                       range_type;
    typedef typename std::iterator_traits<StringType>::value_type           symbol_type;
    typedef symbol_type                                                     value_type;
    typedef typename std::iterator_traits<StringType>::reference            reference;
    typedef decltype(nvbio::begin(std::declval<const StringType&>()))       iterator;      ///< the underlying string's own iterator
    typedef iterator                                                        const_iterator;
    typedef iterator                                                        forward_iterator;

    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE Infix() {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE Infix(const StringType string, const CoordType c) : m_string(string), m_coords(c) {}

    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32     size()      const { return uint32(coord_access::last(m_coords) - coord_access::first(m_coords)); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32     length()    const { return size(); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE index_type string_id() const { return coord_access::id(m_coords); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE range_type range()     const { return make_vector(coord_access::first(m_coords), coord_access::last(m_coords)); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE coord_type coords()    const { return m_coords; }

    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE symbol_type operator[](const uint32 i) const { return m_string[coord_access::first(m_coords) + i]; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE reference   operator[](const uint32 i)       { return m_string[coord_access::first(m_coords) + i]; }

    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE iterator begin() const { return nvbio::begin(m_string) + coord_access::first(m_coords); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE iterator end()   const { return nvbio::begin(m_string) + coord_access::last(m_coords); }

    StringType m_string;
    CoordType  m_coords;
};
template <typename StringType, typename CoordType>
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE Infix<StringType, CoordType> make_infix(const StringType string, const CoordType c) { return Infix<StringType, CoordType>(string, c); }

template <typename S, typename C> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE typename Infix<S, C>::index_type infix_begin(const Infix<S, C>& i) { return i.range().x; }
template <typename S, typename C> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE typename Infix<S, C>::index_type infix_end(const Infix<S, C>& i)   { return i.range().y; }
template <typename S, typename C> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE typename Infix<S, C>::index_type string_id(const Infix<S, C>& i)   { return i.string_id(); }
template <typename S, typename C> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 length(const Infix<S, C>& i) { return i.length(); }
template <typename S, typename C> struct string_traits< Infix<S, C> > { typedef typename Infix<S, C>::value_type value_type; typedef typename Infix<S, C>::index_type index_type; };

namespace priv {
/// the string an infix set cuts: the sequence itself for (begin,end) coordinates, member `id` of a string set for quadruples
template <typename SequenceType, typename CoordType, bool IN_SET> struct infix_source
{
    typedef SequenceType base_string_type;
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE static base_string_type get(const SequenceType& s, const CoordType&) { return s; }
};
template <typename SequenceType, typename CoordType> struct infix_source<SequenceType, CoordType, true>
{
    typedef typename SequenceType::string_type base_string_type;
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE static base_string_type get(const SequenceType& s, const CoordType& c) { return s[uint32(c.x)]; }
};
} // namespace priv

/// a set of infixes over one string or over a string set (infix.h:538-592); operator[] yields the Infix
template <typename SequenceType, typename InfixIterator>
struct InfixSet
{
    typedef SequenceType                                                    sequence_type;
    typedef InfixIterator                                                   infix_iterator;
    typedef typename std::iterator_traits<InfixIterator>::value_type        coord_type;
    typedef priv::infix_source<SequenceType, coord_type, priv::infix_coords<coord_type>::in_set> source;
    typedef typename source::base_string_type                               base_string_type;
    typedef Infix<base_string_type, coord_type>                             string_type;
    typedef typename string_type::symbol_type                               symbol_type;

    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE InfixSet() : m_size(0) {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE InfixSet(const uint32 size, const SequenceType sequence, const InfixIterator infixes)
        : m_size(size), m_sequence(sequence), m_infixes(infixes) {}

    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 size() const { return m_size; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE string_type operator[](const uint32 i) const
    {
        const coord_type c = m_infixes[i];
        return string_type(source::get(m_sequence, c), c);
    }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE const SequenceType& sequence() const { return m_sequence; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE InfixIterator       infixes()  const { return m_infixes; }

    uint32        m_size;
    SequenceType  m_sequence;
    InfixIterator m_infixes;
};

namespace priv {
/// an infix of a packed window is a packed window
template <typename S, typename C>
struct packed_view< Infix<S, C> >
{
    typedef packed_view<S> base;
    static const bool   ok   = base::ok;
    static const uint32 BITS = base::BITS;
    static const bool   BE   = base::BE;
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE static void where(const Infix<S, C>& v, uint64& word0, uint32& first)
    {
        base::stream_where::where(v.begin(), word0, first);
    }
};
} // namespace priv

} // namespace nvbio
