// compat/nvbio/fmindex/fmindex.h -- fm_index<TRankDictionary,TSuffixArray,TL2> and its free functions
// (nvbio/fmindex/fmindex.h:330-390, fmindex_inl.h:36-570) as host-device templates a caller kernel can invoke per thread:
// rank / rank4, match / match_reverse (backward search), basic_inv_psi, locate and the two-pass ssa iterators.
// Conventions kept from the reference: the '$' row `primary` is not stored in the BWT (rows >= primary read k-1),
// rank(-1) = 0, rank(length) = count, match returns the raw (l,r) of the step that emptied the range, and (1,0) on a
// symbol outside the alphabet.  (The reference tests `c > symbol_count()`, which lets the symbol equal to the alphabet size
// -- N = 4 over DNA -- through to L2[4] and a counter of the next block; here that symbol is "no match" too, as in
// nvBowtie's own match_range, mapping_inl.h:90.)
#pragma once
#include "rank_dictionary.h"
#include "line_native.h"
#include "ssa.h"

namespace nvbio {

template <typename A, typename B, typename T, typename F> struct if_equal { typedef F type; };
template <typename A, typename T, typename F> struct if_equal<A, A, T, F> { typedef T type; };

template <typename TRankDictionary, typename TSuffixArray, typename TL2>
struct fm_index
{
    typedef TRankDictionary                         rank_dictionary_type;
    typedef typename TRankDictionary::text_type     bwt_type;
    typedef TSuffixArray                            suffix_array_type;
    typedef typename TRankDictionary::index_type    index_type;
    typedef typename TRankDictionary::range_type    range_type;
    typedef typename TRankDictionary::vector_type   vector_type;
    typedef typename if_equal<TL2, null_type, const index_type*, TL2>::type L2_iterator;

    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE index_type      length() const { return m_length; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE index_type      primary() const { return m_primary; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE index_type      count(const uint32 c) const { return m_L2[c + 1] - m_L2[c]; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE index_type      L2(const uint32 c) const { return m_L2[c]; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE TRankDictionary rank_dict() const { return m_rank_dict; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE TSuffixArray    sa() const { return m_sa; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bwt_type        bwt() const { return m_rank_dict.text(); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32          symbol_count() const { return m_rank_dict.symbol_count(); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32          symbol_size()  const { return m_rank_dict.symbol_size(); }

    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE fm_index() {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE fm_index(const index_type length, const index_type primary, const L2_iterator L2,
                                                 const TRankDictionary rank_dict, const TSuffixArray sa)
        : m_length(length), m_primary(primary), m_L2(L2), m_rank_dict(rank_dict), m_sa(sa) {}

    /// the line-native index of this FM-index in device memory (line_native.h; io::FMIndexDataDevice attaches it): the per-thread device
    /// functions below then read one 128-byte record per step end instead of the 32-byte records of the reference layout
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE void set_line_native(const uint32* base) { m_side.base = base; }
    /// the whole suffix array of this FM-index in device memory (line_native.h: full_sa; io::FMIndexDataDevice builds and attaches it when the
    /// device has the room): locate_ssa_iterator becomes one load, the positions stay what the sampled array gives
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE void set_full_sa(const uint32* sa) { m_side.full_sa = sa; }

    index_type      m_length;
    index_type      m_primary;
    L2_iterator     m_L2;
    TRankDictionary m_rank_dict;
    TSuffixArray    m_sa;
    priv::native_side<index_type> m_side;
};

#define NVBIO_FMI_T template <typename R, typename S, typename L>
#define NVBIO_FMI   fm_index<R, S, L>

#if defined(__HIPCC__)
namespace priv {
/// rank over both ends of a range on the line-native records: the step kept by the previous call, or one or two fresh lines
template <typename F, typename side_type, typename range_type>
NVBIO_FORCEINLINE __device__ range_type native_rank(const F&, const side_type&, const range_type range, const uint8) { return range; }     // (never attached)
template <typename F>
NVBIO_FORCEINLINE __device__ uint2 native_rank(const F& fmi, const native_side<uint32>& side, const uint2 range, const uint8 c)
{
    const uint32 L2c = fmi.L2(c);
    uint2 out;
    if (side.kept(range.x, range.y, c, L2c, out)) return out;
    return side.step(range.x, range.y, c, L2c);
}
/// the attached whole suffix array, NULL for index types that cannot carry one
template <typename side_type> NVBIO_FORCEINLINE __device__ const uint32* full_sa_of(const side_type&) { return NULL; }
NVBIO_FORCEINLINE __device__ const uint32* full_sa_of(const native_side<uint32>& side) { return side.full_sa; }
template <typename sa_type> struct native_sampled
{
    NVBIO_FORCEINLINE __device__ native_sampled(const sa_type& sa) : m_sa(sa) {}
    NVBIO_FORCEINLINE __device__ bool operator()(const uint32 row) const { return m_sa.has(row); }
    const sa_type& m_sa;
};
template <typename F, typename side_type, typename index_type>
NVBIO_FORCEINLINE __device__ void native_locate_step(const F&, const side_type&, index_type&, index_type&) {}
template <typename F>
NVBIO_FORCEINLINE __device__ void native_locate_step(const F& fmi, const native_side<uint32>& side, uint32& j, uint32& t)
{ side.locate_step(j, t, native_sampled<typename F::suffix_array_type>(fmi.m_sa)); }
} // namespace priv
#endif

/// occurrences of c in BWT rows [0, k]   (fmindex_inl.h:36-57)
NVBIO_FMI_T NVBIO_FORCEINLINE NVBIO_HOST_DEVICE
typename NVBIO_FMI::index_type rank(const NVBIO_FMI& fmi, typename NVBIO_FMI::index_type k, uint8 c)
{
    typedef typename NVBIO_FMI::index_type index_type;
    if (k == index_type(-1)) return 0;
    if (k == fmi.length())   return fmi.count(c);
    if (k >= fmi.primary())  --k;
    return rank(fmi.m_rank_dict, k, uint32(c));
}
/// ... for both ends of a range   (fmindex_inl.h:66-99)
NVBIO_FMI_T NVBIO_FORCEINLINE NVBIO_HOST_DEVICE
typename NVBIO_FMI::range_type rank(const NVBIO_FMI& fmi, typename NVBIO_FMI::range_type range, uint8 c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if (fmi.m_side.attached()) return priv::native_rank(fmi, fmi.m_side, range, c);
#endif
    return make_vector(rank(fmi, range.x, c), rank(fmi, range.y, c));
}
/// all four symbols at once   (fmindex_inl.h:111-135)
NVBIO_FMI_T NVBIO_FORCEINLINE NVBIO_HOST_DEVICE
typename R::vec4_type rank4(const NVBIO_FMI& fmi, typename NVBIO_FMI::index_type k)
{
    typedef typename NVBIO_FMI::index_type index_type;
    if (k == index_type(-1)) return make_vector(index_type(0), index_type(0), index_type(0), index_type(0));
    if (k == fmi.length())   return make_vector(fmi.count(0), fmi.count(1), fmi.count(2), fmi.count(3));
    if (k >= fmi.primary())  --k;
    return rank4(fmi.m_rank_dict, k);
}

/// ... of both ends of a range, the form nvBowtie's 1-mismatch map<> steps with (fmindex_inl.h:138-186, mapping_inl.h:178).
/// The '$' adjustments are applied per end and the two ends go to the dictionary together, so that on the production layout
/// they share one 32-byte record when they fall in the same block.
NVBIO_FMI_T NVBIO_FORCEINLINE NVBIO_HOST_DEVICE
void rank4(const NVBIO_FMI& fmi, typename NVBIO_FMI::range_type range, typename R::vec4_type* outl, typename R::vec4_type* outh)
{
    typedef typename NVBIO_FMI::index_type index_type;
    if (range.x == range.y) { *outl = rank4(fmi, range.x); *outh = *outl; return; }
    if (range.x == index_type(-1) || range.y == fmi.length() || range.y == index_type(-1) || range.x == fmi.length())
    { *outl = rank4(fmi, range.x); *outh = rank4(fmi, range.y); return; }
    if (range.x >= fmi.primary()) --range.x;
    if (range.y >= fmi.primary()) --range.y;
    rank4(fmi.m_rank_dict, range, outl, outh);
}
/// every symbol of the alphabet (fmindex_inl.h:188-262).  The reference's single-ended form falls through its k == -1 and
/// k == length branches into the dictionary query (:208-222, reading out of bounds); here those branches return.
NVBIO_FMI_T NVBIO_FORCEINLINE NVBIO_HOST_DEVICE
void rank_all(const NVBIO_FMI& fmi, typename NVBIO_FMI::index_type k, typename NVBIO_FMI::vector_type* out)
{
    typedef typename NVBIO_FMI::index_type index_type;
    if (k == index_type(-1)) { for (uint32 c = 0; c < fmi.symbol_count(); ++c) (*out)[c] = 0; return; }
    if (k == fmi.length())   { for (uint32 c = 0; c < fmi.symbol_count(); ++c) (*out)[c] = fmi.count(c); return; }
    if (k >= fmi.primary()) --k;
    rank_all(fmi.m_rank_dict, k, out);
}
NVBIO_FMI_T NVBIO_FORCEINLINE NVBIO_HOST_DEVICE
void rank_all(const NVBIO_FMI& fmi, typename NVBIO_FMI::range_type range, typename NVBIO_FMI::vector_type* outl, typename NVBIO_FMI::vector_type* outh)
{
    typedef typename NVBIO_FMI::index_type index_type;
    if (range.x == range.y) { rank_all(fmi, range.x, outl); *outh = *outl; return; }
    if (range.x == index_type(-1) || range.y == fmi.length() || range.y == index_type(-1) || range.x == fmi.length())
    { rank_all(fmi, range.x, outl); rank_all(fmi, range.y, outh); return; }
    if (range.x >= fmi.primary()) --range.x;
    if (range.y >= fmi.primary()) --range.y;
    rank_all(fmi.m_rank_dict, range, outl, outh);
}

/// backward search from a given range   (fmindex_inl.h:307-341)
template <typename R, typename S, typename L, typename Iterator> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE
typename NVBIO_FMI::range_type match(const NVBIO_FMI& fmi, const Iterator pattern, const uint32 pattern_len, const typename NVBIO_FMI::range_type in_range)
{
    typedef typename NVBIO_FMI::index_type index_type;
    typename NVBIO_FMI::range_type range = in_range;
    for (int32 i = int32(pattern_len) - 1; i >= 0 && range.x <= range.y; --i)
    {
        const uint32 c = uint32(pattern[i]);
        if (c >= fmi.symbol_count()) return make_vector(index_type(1), index_type(0));
        const index_type lo = rank(fmi, index_type(range.x - 1), uint8(c)), hi = rank(fmi, range.y, uint8(c));
        range.x = fmi.L2(c) + lo + 1;
        range.y = fmi.L2(c) + hi;
    }
    return range;
}
template <typename R, typename S, typename L, typename Iterator> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE
typename NVBIO_FMI::range_type match(const NVBIO_FMI& fmi, const Iterator pattern, const uint32 pattern_len)
{
    typedef typename NVBIO_FMI::index_type index_type;
    return match(fmi, pattern, pattern_len, make_vector(index_type(0), fmi.length()));
}
/// the pattern walked front to back, i.e. a search for its reverse   (fmindex_inl.h:349-382)
template <typename R, typename S, typename L, typename Iterator> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE
typename NVBIO_FMI::range_type match_reverse(const NVBIO_FMI& fmi, const Iterator pattern, const uint32 pattern_len)
{
    typedef typename NVBIO_FMI::index_type index_type;
    typename NVBIO_FMI::range_type range = make_vector(index_type(0), fmi.length());
    for (uint32 i = 0; i < pattern_len && range.x <= range.y; ++i)
    {
        const uint32 c = uint32(pattern[i]);
        if (c >= fmi.symbol_count()) return make_vector(index_type(1), index_type(0));
        const index_type lo = rank(fmi, index_type(range.x - 1), uint8(c)), hi = rank(fmi, range.y, uint8(c));
        range.x = fmi.L2(c) + lo + 1;
        range.y = fmi.L2(c) + hi;
    }
    return range;
}

/// one LF step (fmindex_inl.h:390-414): the row of the suffix one text position earlier; the SA = 0 row wraps to row 0
NVBIO_FMI_T NVBIO_FORCEINLINE NVBIO_HOST_DEVICE
typename NVBIO_FMI::index_type basic_inv_psi(const NVBIO_FMI& fmi, const typename NVBIO_FMI::index_type i)
{
    typedef typename NVBIO_FMI::index_type index_type;
    if (i == fmi.primary()) return 0;
    const index_type k = i < fmi.primary() ? i : i - 1;
    const uint8 c = fmi.m_rank_dict.m_text[k];
    return fmi.L2(c) + rank(fmi.m_rank_dict, k, uint32(c));
}

/// LF-walk until the suffix array knows the row: (row, steps)   (fmindex_inl.h:416-456)
NVBIO_FMI_T NVBIO_FORCEINLINE NVBIO_HOST_DEVICE
typename NVBIO_FMI::range_type inv_psi(const NVBIO_FMI& fmi, const typename NVBIO_FMI::index_type i)
{
    typedef typename NVBIO_FMI::index_type index_type;
    index_type j = i, t = 0, suffix;
    while (!fmi.m_sa.fetch(j, suffix)) { j = basic_inv_psi(fmi, j); ++t; }
    return make_vector(j, t);
}
/// LF-walk to the next sampled row: (row, steps)   (fmindex_inl.h:511-545)
NVBIO_FMI_T NVBIO_FORCEINLINE NVBIO_HOST_DEVICE
typename NVBIO_FMI::range_type locate_ssa_iterator(const NVBIO_FMI& fmi, const typename NVBIO_FMI::index_type i)
{
    typedef typename NVBIO_FMI::index_type index_type;
    index_type j = i, t = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    if (priv::full_sa_of(fmi.m_side)) return make_vector(index_type(0), index_type(priv::full_sa_of(fmi.m_side)[i] + 1u));      // (0, SA[i] + 1): ssa[0] = -1
    if (fmi.m_side.attached())
    {
        while (!fmi.m_sa.has(j)) priv::native_locate_step(fmi, fmi.m_side, j, t);        // up to two text positions per record
        return make_vector(j, t);
    }
#endif
    while (!fmi.m_sa.has(j)) { j = basic_inv_psi(fmi, j); ++t; }
    return make_vector(j, t);
}
/// the sampled value of an iterator plus its steps   (fmindex_inl.h:553-569)
NVBIO_FMI_T NVBIO_FORCEINLINE NVBIO_HOST_DEVICE
typename NVBIO_FMI::index_type lookup_ssa_iterator(const NVBIO_FMI& fmi, const typename NVBIO_FMI::range_type it)
{
    typename NVBIO_FMI::index_type suffix = 0;
    fmi.m_sa.fetch(it.x, suffix);
    return suffix + it.y;
}
/// text position of SA row i   (fmindex_inl.h:466-501)
NVBIO_FMI_T NVBIO_FORCEINLINE NVBIO_HOST_DEVICE
typename NVBIO_FMI::index_type locate(const NVBIO_FMI& fmi, const typename NVBIO_FMI::index_type i)
{
    return lookup_ssa_iterator(fmi, locate_ssa_iterator(fmi, i));
}

#undef NVBIO_FMI_T
#undef NVBIO_FMI

} // namespace nvbio
