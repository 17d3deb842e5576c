// compat/nvbio/fmindex/rank_dictionary.h -- rank_dictionary<SYMBOL_SIZE,K,TextString,OccIterator,CountTable> and its rank
// queries (nvbio/fmindex/rank_dictionary.h:60-280, rank_dictionary_inl.h:243-585) as host-device templates over any
// word iterator: separate bwt / occ arrays of uint32 / uint64 / uint4 (as the reference's tests build them,
// nvbio-test/rank_test.cu:139-232), the interleaved uint4 production layout seen through deinterleaved_iterator
// (nvbio/io/fmindex/fmindex.h:159-174), 32- or 64-bit indices.
//   occ[k * SYMBOL_COUNT + c] = #c in text[0, k*K);   rank(dict, i, c) = #c in text[0 .. i]   (i == -1 -> 0)
// The count is re-derived: whole words of the block by symbol-match bit-planes + popcount, the last word under a prefix
// mask -- no per-symbol loop for 2-bit texts, no lookup table (the CountTable argument is accepted and unused).
//
// Two execution forms, chosen at compile time:
//   * 2-bit text in 32-bit words with K = 64 (every index of the hot path): one block = four words + four counters.  When
//     the iterators hand out uint4s (production layout, or uint4 test arrays) each is ONE 16-byte load, the block is counted
//     as two 64-bit planes, and the range forms -- rank(dict, (l,r), c), rank4 / rank_all(dict, (l,r), &lo, &hi) -- issue the
//     loads of both ends together and share them when l and r fall in one block: the same two-ended read the batch kernels
//     make (nvbio_amd/csrc/fmindex_device.h, fm_rank2 / fm_rank4_range).
//   * anything else (64-bit words, other K, other symbol sizes): a word loop.
#pragma once
#include "../basic/types.h"
#include "../basic/numbers.h"
#include "../basic/static_vector.h"
#include "../basic/packedstream.h"

namespace nvbio {

template <typename T> struct vector_traits { typedef T value_type; static const uint32 DIM = 1; };
template <> struct vector_traits<uint2> { typedef uint32 value_type; static const uint32 DIM = 2; };
template <> struct vector_traits<uint4> { typedef uint32 value_type; static const uint32 DIM = 4; };
template <> struct vector_traits<ulonglong2> { typedef uint64 value_type; static const uint32 DIM = 2; };
template <> struct vector_traits<ulonglong4> { typedef uint64 value_type; static const uint32 DIM = 4; };

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

/// counter c of block k, whatever the occurrence iterator hands out (scalars, uint4s, ...)
template <uint32 S, typename OccIterator>
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE
typename vec_comp<typename std::iterator_traits<OccIterator>::value_type>::type occ_at(const OccIterator& occ, const uint64 k, const uint32 c)
{
    typedef vec_comp<typename std::iterator_traits<OccIterator>::value_type> vc;
    const uint64 j = k * S + c;
    if (vc::N == 1u) return vc::get(occ[j], 0u);
    return vc::get(occ[j / vc::N], uint32(j % vc::N));
}

/// occurrences of c in symbols [block*K, i] of the dictionary's text (K a multiple of the symbols per word)
template <uint32 BITS, uint32 K, typename TextString, typename index_type>
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 block_count(const TextString& text, const index_type i, const uint32 c)
{
    typedef typename TextString::word_type word_type;
    const uint32 PER = uint32(8u * sizeof(word_type)) / BITS;
    const index_type first = (i / K) * K;
    const uint64 w0 = uint64(text.index() + first) / PER;                    // text.index() is word aligned for index streams
    const uint32 n = uint32(i - first) + 1u;                                  // symbols to count
    uint32 r = 0;
    uint32 w = 0;
    for (; (w + 1u) * PER <= n; ++w) r += word_count<BITS>(text.word(w0 + w), c, PER);
    if (w * PER < n) r += word_count<BITS>(text.word(w0 + w), c, n - w * PER);
    return r;
}

// ---- the 2-bit / 32-bit word / K = 64 block: four text words + four counters ------------------------------------------------
template <uint32 B, uint32 K, typename T, typename O> struct block64
{
    static const bool value = false;
};
template <typename I, typename S, bool E, typename X, typename O>
struct block64<2u, 64u, PackedStream<I, S, 2u, E, X>, O>
{
    typedef PackedStream<I, S, 2u, E, X> text_type;
    static const bool value = E && sizeof(typename text_type::word_type) == 4 &&
                              sizeof(typename vec_comp<typename std::iterator_traits<O>::value_type>::type) == 4;
};

/// the block's four words / counters as a uint4: one load when the iterator yields uint4s and the block is 16-byte aligned in it
template <typename TextString>
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint4 block_words(const TextString& text, const uint64 k)
{
    typedef typename TextString::storage_comp sc;
    const uint64 w0 = uint64(text.index()) / 16u + 4u * k;
    if (sc::N == 4u && (w0 & 3u) == 0u)
    {
        const typename TextString::storage_type v = text.stream()[w0 / 4u];
        return make_uint4(uint32(sc::get(v, 0u)), uint32(sc::get(v, 1u)), uint32(sc::get(v, 2u)), uint32(sc::get(v, 3u)));
    }
    return make_uint4(uint32(text.word(w0)), uint32(text.word(w0 + 1u)), uint32(text.word(w0 + 2u)), uint32(text.word(w0 + 3u)));
}
template <typename OccIterator>
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint4 block_counters(const OccIterator& occ, const uint64 k)
{
    typedef typename std::iterator_traits<OccIterator>::value_type value_type;
    typedef vec_comp<value_type> vc;
    if (vc::N == 4u)
    {
        const value_type v = occ[k];
        return make_uint4(uint32(vc::get(v, 0u)), uint32(vc::get(v, 1u)), uint32(vc::get(v, 2u)), uint32(vc::get(v, 3u)));
    }
    return make_uint4(uint32(occ_at<4u>(occ, k, 0u)), uint32(occ_at<4u>(occ, k, 1u)), uint32(occ_at<4u>(occ, k, 2u)), uint32(occ_at<4u>(occ, k, 3u)));
}
/// "symbol == c" over 32 big-endian symbols in 64 bits, one bit per symbol
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint64 plane64(const uint64 x, const uint32 c)
{
    const uint64 hi = (c & 2u) ? x : ~x, lo = (c & 1u) ? x : ~x;
    return (hi >> 1) & lo & 0x5555555555555555ull;
}
/// occurrences of c among the first cnt (1..64) symbols of a block
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 count64(const uint4 w, const uint32 cnt, const uint32 c)
{
    const uint64 a = (uint64(w.x) << 32) | w.y, b = (uint64(w.z) << 32) | w.w;
    const uint32 ca = cnt < 32u ? cnt : 32u, cb = cnt - ca;
    const uint64 ma = ca ? (~uint64(0) << (64u - 2u * ca)) : 0ull, mb = cb ? (~uint64(0) << (64u - 2u * cb)) : 0ull;
    return popc(plane64(a, c) & ma) + popc(plane64(b, c) & mb);
}
/// ... of all four symbols (the fourth by difference)
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint4 count64x4(const uint4 w, const uint4 o, const uint32 cnt)
{
    const uint32 a = count64(w, cnt, 0u), c = count64(w, cnt, 1u), g = count64(w, cnt, 2u);
    return make_uint4(o.x + a, o.y + c, o.z + g, o.w + (cnt - a - c - g));
}

template <typename index_type> struct vec_of { };
template <> struct vec_of<uint32> { NVBIO_FORCEINLINE NVBIO_HOST_DEVICE static uint4 make(const uint4 v) { return v; } };
template <> struct vec_of<uint64> { NVBIO_FORCEINLINE NVBIO_HOST_DEVICE static ulonglong4 make(const uint4 v) { return make_ulonglong4(v.x, v.y, v.z, v.w); } };

} // namespace priv

#define NVBIO_RD_T template <uint32 B, uint32 K, typename T, typename O, typename C>
#define NVBIO_RD   rank_dictionary<B, K, T, O, C>

/// rank(dict, i, c): occurrences of c in text[0 .. i]   (rank_dictionary_inl.h:305-322 / 502-513)
NVBIO_RD_T NVBIO_FORCEINLINE NVBIO_HOST_DEVICE
typename NVBIO_RD::index_type rank(const NVBIO_RD& dict, const typename NVBIO_RD::index_type i, const uint32 c)
{
    typedef typename NVBIO_RD::index_type index_type;
    if (i == index_type(-1)) return 0u;
    if (priv::block64<B, K, T, O>::value)
    {
        const uint64 k = uint64(i) >> 6;
        return index_type(priv::occ_at<4u>(dict.m_occ, k, c)) + priv::count64(priv::block_words(dict.m_text, k), (uint32(i) & 63u) + 1u, c);
    }
    return index_type(priv::occ_at<(1u << B)>(dict.m_occ, uint64(i / K), c)) + priv::block_count<B, K>(dict.m_text, i, c);
}
/// rank(dict, (l,r), c): both ends at once   (rank_dictionary_inl.h:326-345 / 515-538).  Each end is the plain count of its own
/// prefix (-1 -> 0); the reference's generic form answers (r,r) for l == -1 != r where its uint4 form answers (0,r) -- a case
/// fm_index::rank never forwards (fmindex_inl.h:83-87); this is the uint4 form's answer.
NVBIO_RD_T NVBIO_FORCEINLINE NVBIO_HOST_DEVICE
typename NVBIO_RD::range_type rank(const NVBIO_RD& dict, const typename NVBIO_RD::range_type range, const uint32 c)
{
    typedef typename NVBIO_RD::index_type index_type;
    if (priv::block64<B, K, T, O>::value && range.x != index_type(-1) && range.y != index_type(-1))
    {
        const uint64 kx = uint64(range.x) >> 6, ky = uint64(range.y) >> 6;
        const uint4 wx = priv::block_words(dict.m_text, kx);
        const index_type ox = index_type(priv::occ_at<4u>(dict.m_occ, kx, c));
        const uint4 wy = kx == ky ? wx : priv::block_words(dict.m_text, ky);
        const index_type oy = kx == ky ? ox : index_type(priv::occ_at<4u>(dict.m_occ, ky, c));
        return make_vector(index_type(ox + priv::count64(wx, (uint32(range.x) & 63u) + 1u, c)),
                           index_type(oy + priv::count64(wy, (uint32(range.y) & 63u) + 1u, c)));
    }
    return make_vector(rank(dict, range.x, c), rank(dict, range.y, c));
}

/// rank4(dict, i): all four symbols of a 2-bit text   (rank_dictionary_inl.h:347-372 / 540-573)
template <uint32 K, typename T, typename O, typename C>
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE
typename rank_dictionary<2, K, T, O, C>::vec4_type rank4(const rank_dictionary<2, K, T, O, C>& dict, const typename rank_dictionary<2, K, T, O, C>::index_type i)
{
    typedef typename rank_dictionary<2, K, T, O, C>::index_type index_type;
    if (i == index_type(-1)) return make_vector(index_type(0), index_type(0), index_type(0), index_type(0));
    if (priv::block64<2u, K, T, O>::value)
    {
        const uint64 k = uint64(i) >> 6;
        return priv::vec_of<index_type>::make(priv::count64x4(priv::block_words(dict.m_text, k), priv::block_counters(dict.m_occ, k), (uint32(i) & 63u) + 1u));
    }
    return make_vector(rank(dict, i, 0u), rank(dict, i, 1u), rank(dict, i, 2u), rank(dict, i, 3u));
}
/// rank4(dict, (l,r), &lo, &hi): both ends, sharing the block when they meet in one   (rank_dictionary.h:216-232)
template <uint32 K, typename T, typename O, typename C>
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE
void rank4(const rank_dictionary<2, K, T, O, C>& dict, const typename rank_dictionary<2, K, T, O, C>::range_type range,
           typename rank_dictionary<2, K, T, O, C>::vec4_type* outl, typename rank_dictionary<2, K, T, O, C>::vec4_type* outh)
{
    typedef typename rank_dictionary<2, K, T, O, C>::index_type index_type;
    if (priv::block64<2u, K, T, O>::value && range.x != index_type(-1) && range.y != index_type(-1))
    {
        const uint64 kx = uint64(range.x) >> 6, ky = uint64(range.y) >> 6;
        const uint4 wx = priv::block_words(dict.m_text, kx), ox = priv::block_counters(dict.m_occ, kx);
        const uint4 wy = kx == ky ? wx : priv::block_words(dict.m_text, ky);
        const uint4 oy = kx == ky ? ox : priv::block_counters(dict.m_occ, ky);
        *outl = priv::vec_of<index_type>::make(priv::count64x4(wx, ox, (uint32(range.x) & 63u) + 1u));
        *outh = priv::vec_of<index_type>::make(priv::count64x4(wy, oy, (uint32(range.y) & 63u) + 1u));
        return;
    }
    *outl = rank4(dict, range.x);
    *outh = rank4(dict, range.y);
}

/// rank_all(dict, i [, &out]): every symbol of the alphabet   (rank_dictionary.h:240-262)
NVBIO_RD_T NVBIO_FORCEINLINE NVBIO_HOST_DEVICE
void rank_all(const NVBIO_RD& dict, const typename NVBIO_RD::index_type i, typename NVBIO_RD::vector_type* out)
{
    typedef typename NVBIO_RD::index_type index_type;
    if (B == 2u && priv::block64<B, K, T, O>::value && i != index_type(-1))
    {
        const uint64 k = uint64(i) >> 6;
        const uint4 r = priv::count64x4(priv::block_words(dict.m_text, k), priv::block_counters(dict.m_occ, k), (uint32(i) & 63u) + 1u);
        (*out)[0] = r.x; (*out)[1] = r.y; (*out)[2] = r.z; (*out)[3] = r.w;
        return;
    }
    for (uint32 c = 0; c < (1u << B); ++c) (*out)[c] = rank(dict, i, c);
}
NVBIO_RD_T NVBIO_FORCEINLINE NVBIO_HOST_DEVICE
typename NVBIO_RD::vector_type rank_all(const NVBIO_RD& dict, const typename NVBIO_RD::index_type i)
{
    typename NVBIO_RD::vector_type r;
    rank_all(dict, i, &r);
    return r;
}
/// rank_all(dict, (l,r), &lo, &hi)   (rank_dictionary.h:270-280)
NVBIO_RD_T NVBIO_FORCEINLINE NVBIO_HOST_DEVICE
void rank_all(const NVBIO_RD& dict, const typename NVBIO_RD::range_type range, typename NVBIO_RD::vector_type* outl, typename NVBIO_RD::vector_type* outh)
{
    typedef typename NVBIO_RD::index_type index_type;
    if (B == 2u && priv::block64<B, K, T, O>::value && range.x != index_type(-1) && range.y != index_type(-1))
    {
        const uint64 kx = uint64(range.x) >> 6, ky = uint64(range.y) >> 6;
        const uint4 wx = priv::block_words(dict.m_text, kx), ox = priv::block_counters(dict.m_occ, kx);
        const uint4 wy = kx == ky ? wx : priv::block_words(dict.m_text, ky);
        const uint4 oy = kx == ky ? ox : priv::block_counters(dict.m_occ, ky);
        const uint4 lo = priv::count64x4(wx, ox, (uint32(range.x) & 63u) + 1u), hi = priv::count64x4(wy, oy, (uint32(range.y) & 63u) + 1u);
        (*outl)[0] = lo.x; (*outl)[1] = lo.y; (*outl)[2] = lo.z; (*outl)[3] = lo.w;
        (*outh)[0] = hi.x; (*outh)[1] = hi.y; (*outh)[2] = hi.z; (*outh)[3] = hi.w;
        return;
    }
    rank_all(dict, range.x, outl);
    rank_all(dict, range.y, outh);
}

#undef NVBIO_RD_T
#undef NVBIO_RD

/// build_occurrence_table<SYMBOL_SIZE,K>(begin, end, occ, cnt) on the host (rank_dictionary_inl.h:42-77): occ[k*S + c] = #c before
/// block k, for every block that holds at least one symbol (a text of exactly m*K symbols gets m blocks, not m+1); cnt, when not
/// NULL, receives the totals
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
    if (cnt) for (uint32 c = 0; c < S; ++c) cnt[c] = run[c];
}

} // namespace nvbio
