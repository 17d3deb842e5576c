// compat/nvbio/fmindex/rank_dictionary.h -- rank_dictionary<SYMBOL_SIZE,K,TextString,OccIterator,CountTable> and its rank
// queries (nvbio/fmindex/rank_dictionary.h:60-134, rank_dictionary_inl.h:243-585) as host-device templates over any
// word iterator: separate bwt / occ arrays (as the reference's tests build them), the interleaved uint4 production layout
// seen through deinterleaved_iterator, 32- or 64-bit indices.
//   occ[k * SYMBOL_COUNT + c] = #c in text[0, k*K);   rank(dict, i, c) = #c in text[0 .. i]   (i == -1 -> 0)
// The count is re-derived: whole words of the block by symbol-match bit-planes + popcount, the last word under a prefix
// mask -- no per-symbol loop for 2-bit texts, no lookup table (the CountTable argument is accepted and unused).
#pragma once
#include "../basic/types.h"
#include "../basic/packedstream.h"

namespace nvbio {

template <typename T> struct vector_traits { typedef T value_type; static const uint32 DIM = 1; };
template <> struct vector_traits<uint2> { typedef uint32 value_type; static const uint32 DIM = 2; };
template <> struct vector_traits<uint4> { typedef uint32 value_type; static const uint32 DIM = 4; };
template <> struct vector_traits<ulonglong2> { typedef uint64 value_type; static const uint32 DIM = 2; };
template <> struct vector_traits<ulonglong4> { typedef uint64 value_type; static const uint32 DIM = 4; };

/// a tiny fixed-size vector (StaticVector, nvbio/basic/static_vector.h) for the all-symbol queries
template <typename T, uint32 N> struct StaticVector {
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE T  operator[](const uint32 i) const { return data[i]; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE T& operator[](const uint32 i)       { return data[i]; }
    T data[N];
};

template <uint32 SYMBOL_SIZE_T, uint32 K, typename TextString, typename OccIterator, typename CountTable = null_type>
struct rank_dictionary
{
    static const uint32 BLOCK_INTERVAL = K;
    static const uint32 SYMBOL_SIZE    = SYMBOL_SIZE_T;
    static const uint32 SYMBOL_COUNT   = 1u << SYMBOL_SIZE_T;

    typedef TextString   text_type;
    typedef OccIterator  occ_iterator;
    typedef CountTable   count_table_type;
    typedef typename vector_traits<typename std::iterator_traits<OccIterator>::value_type>::value_type index_type;
    typedef typename vector_type<index_type, 2>::type   range_type;
    typedef typename vector_type<index_type, 2>::type   vec2_type;
    typedef typename vector_type<index_type, 4>::type   vec4_type;
    typedef StaticVector<index_type, SYMBOL_COUNT>      vector_type;

    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE rank_dictionary() {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE rank_dictionary(const TextString _text, const OccIterator _occ, const CountTable _count_table)
        : m_text(_text), m_occ(_occ), m_count_table(_count_table) {}

    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 symbol_count() const { return SYMBOL_COUNT; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 symbol_size()  const { return SYMBOL_SIZE_T; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE text_type        text() const { return m_text; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE occ_iterator     occ() const { return m_occ; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE count_table_type count_table() const { return m_count_table; }

    TextString  m_text;
    OccIterator m_occ;
    CountTable  m_count_table;
};

namespace priv {

/// occurrences of c among the first `cnt` symbols (0 < cnt <= per word) of one big-endian packed word
template <uint32 BITS, typename word_type>
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 word_count(const word_type w, const uint32 c, const uint32 cnt)
{
    const uint32 W = uint32(8u * sizeof(word_type)), PER = W / BITS;
    if (BITS == 2u)
    {
        // bit-plane of "symbol == c" at the even bit of every symbol, then the leading cnt symbols
        const word_type even = word_type(~word_type(0)) / 3u;                 // 0x5555...
        const word_type hi = (c & 2u) ? w : word_type(~w), lo = (c & 1u) ? w : word_type(~w);
        word_type m = word_type(hi >> 1) & lo & even;
        if (cnt < PER) m &= word_type(~word_type(0)) << (W - 2u * cnt);
        return popc(uint64(m));
    }
    uint32 r = 0;
    for (uint32 s = 0; s < cnt; ++s) r += (uint32((w >> (W - BITS * (s + 1u))) & word_type((1u << BITS) - 1u)) == c) ? 1u : 0u;
    return r;
}

/// occurrences of c in symbols [block*K, i] of the dictionary's text (K a multiple of the symbols per word)
template <uint32 BITS, uint32 K, typename TextString, typename index_type>
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 block_count(const TextString& text, const index_type i, const uint32 c)
{
    typedef typename TextString::storage_type word_type;
    const uint32 PER = uint32(8u * sizeof(word_type)) / BITS;
    const typename TextString::storage_iterator words = text.stream();
    const index_type first = (i / K) * K;
    const uint64 w0 = uint64(text.index() + first) / PER;                    // text.index() is word aligned for index streams
    const uint32 n = uint32(i - first) + 1u;                                  // symbols to count
    uint32 r = 0;
    uint32 w = 0;
    for (; (w + 1u) * PER <= n; ++w) r += word_count<BITS>(word_type(words[w0 + w]), c, PER);
    if (w * PER < n) r += word_count<BITS>(word_type(words[w0 + w]), c, n - w * PER);
    return r;
}

} // namespace priv

/// rank(dict, i, c): occurrences of c in text[0 .. i]   (rank_dictionary_inl.h:305-322 / 502-513)
template <uint32 B, uint32 K, typename T, typename O, typename C>
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE
typename rank_dictionary<B, K, T, O, C>::index_type rank(const rank_dictionary<B, K, T, O, C>& dict, const typename rank_dictionary<B, K, T, O, C>::index_type i, const uint32 c)
{
    typedef typename rank_dictionary<B, K, T, O, C>::index_type index_type;
    if (i == index_type(-1)) return 0u;
    return index_type(dict.m_occ[(i / K) * (1u << B) + c]) + priv::block_count<B, K>(dict.m_text, i, c);
}
/// rank(dict, (l,r), c): both ends at once   (rank_dictionary_inl.h:515-538: l == -1 -> 0)
template <uint32 B, uint32 K, typename T, typename O, typename C>
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE
typename rank_dictionary<B, K, T, O, C>::range_type rank(const rank_dictionary<B, K, T, O, C>& dict, const typename rank_dictionary<B, K, T, O, C>::range_type range, const uint32 c)
{
    return make_vector(rank(dict, range.x, c), rank(dict, range.y, c));
}
/// rank4 / rank_all: all four symbols of a 2-bit text
template <uint32 K, typename T, typename O, typename C>
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE
typename rank_dictionary<2, K, T, O, C>::vec4_type rank4(const rank_dictionary<2, K, T, O, C>& dict, const typename rank_dictionary<2, K, T, O, C>::index_type i)
{
    return make_vector(rank(dict, i, 0u), rank(dict, i, 1u), rank(dict, i, 2u), rank(dict, i, 3u));
}
template <uint32 B, uint32 K, typename T, typename O, typename C>
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE
void rank_all(const rank_dictionary<B, K, T, O, C>& dict, const typename rank_dictionary<B, K, T, O, C>::index_type i, typename rank_dictionary<B, K, T, O, C>::vector_type* out)
{
    for (uint32 c = 0; c < (1u << B); ++c) (*out)[c] = rank(dict, i, c);
}

/// build_occurrence_table<SYMBOL_SIZE,K>(begin, end, occ, cnt) on the host (rank_dictionary_inl.h:42-77): occ[k*S + c] = #c before block k
template <uint32 SYMBOL_SIZE, uint32 K, typename SymbolIterator, typename IndexType>
inline void build_occurrence_table(SymbolIterator begin, SymbolIterator end, IndexType* occ, IndexType* cnt)
{
    const uint32 S = 1u << SYMBOL_SIZE;
    IndexType run[S];
    for (uint32 c = 0; c < S; ++c) run[c] = 0;
    uint64 i = 0;
    for (SymbolIterator it = begin; it != end; ++it, ++i)
    {
        if (i % K == 0) for (uint32 c = 0; c < S; ++c) occ[(i / K) * S + c] = run[c];
        ++run[uint32(*it) & (S - 1u)];
    }
    if (i % K == 0) for (uint32 c = 0; c < S; ++c) occ[(i / K) * S + c] = run[c];
    for (uint32 c = 0; c < S; ++c) cnt[c] = run[c];
}

} // namespace nvbio
