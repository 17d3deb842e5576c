// compat/nvbio/alignment/alignment.h -- the per-thread scoring functions of nvbio::aln as host-device templates over ANY
// string iterator, scheme and sink:
//     banded_alignment_score<BAND_LEN>(aligner, pattern, quals, text, min_score, sink)      nvbio/alignment/alignment.h:257-300
//     alignment_score(aligner, pattern, quals, text, min_score, sink, column)               nvbio/alignment/alignment.h:455-540
// They are the generic half of the drop-in layer: BatchedBandedAlignmentScore / BatchedAlignmentScore (batched.h) run them
// one lane per job for stream types the tuned gfx950 kernels do not recognise (8-bit or user-defined strings, user
// schemes, uint64 streams, asymmetric linear gaps, patterns beyond the tuned kernels' limits), and on the host under
// HostThreadScheduler.  What they compute is the reference's result, observable quirks included:
//   banded  gotoh/gotoh_banded_inl.h:415-658, sw/sw_banded_inl.h:340-520 (edit distance = the SW code with
//           EditDistanceSWScheme): band cell j of row i <-> (pattern i, text i+j); BestSink's last-maximum rule;
//           the short-based infimum; a text window cache that keeps only 2 bits per symbol for bands other than
//           3, 5, 7, 15 (alignment_base_inl.h:75-98)
//   full    gotoh/gotoh_inl.h:459-1490, sw/sw_inl.h:417-1222: text- and pattern-blocking sweeps, whose block order
//           decides LOCAL ties, whose boundary column is cut to int16, and whose early exit (best reachable score below
//           min_score) leaves the sink as reported so far
#pragma once
#include "alignment_base.h"

namespace nvbio {
namespace aln {

namespace priv {

template <uint32 BAND_LEN> struct wide_text_cache { static const bool value = (BAND_LEN == 3u || BAND_LEN == 5u || BAND_LEN == 7u || BAND_LEN == 15u); };
/// what the reference's window cache returns for a stored symbol
template <uint32 BAND_LEN> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint8 cached_symbol(const uint8 g) { return wide_text_cache<BAND_LEN>::value ? g : uint8(g & 3u); }

// ------------------------------------------------------------------ banded, affine gaps
template <uint32 BAND_LEN, AlignmentType TYPE, typename scheme_type, typename pattern_string, typename qual_string, typename text_string, typename sink_type>
NVBIO_HOST_DEVICE inline
bool banded_gotoh_score(const scheme_type& scoring, const pattern_string pattern, const qual_string quals, const text_string text, sink_type& sink)
{
    const uint32 M = pattern.length(), N = text.length();
    if (N < M) return false;

    const int32 Go = scoring.pattern_gap_open(), Ge = scoring.pattern_gap_extension();
    const int32 floor_ = int32(Field_traits<int16>::min()) - nvbio::max(nvbio::max(Go, Ge), nvbio::max(scoring.text_gap_open(), scoring.text_gap_extension()));

    int32 H[BAND_LEN], F[BAND_LEN];
    uint8 win[BAND_LEN];                                   // win[j] = text symbol under band cell j of the current row
    for (uint32 j = 0; j + 1 < BAND_LEN; ++j) win[j] = cached_symbol<BAND_LEN>(uint8(text[j]));
    for (uint32 j = 0; j < BAND_LEN; ++j)
    {
        H[j] = (TYPE == GLOBAL && j > 0) ? scoring.text_gap_open() + int32(j - 1u) * scoring.text_gap_extension() : 0;
        F[j] = floor_;
    }
    for (uint32 i = 0; i < M; ++i)
    {
        const uint8 q = uint8(pattern[i]), qq = uint8(quals[i]);
        // the symbol entering the band at its last cell: read unchecked past the window only through this test
        const uint8 g_in = (i + BAND_LEN - 1u < N) ? uint8(text[i + BAND_LEN - 1u]) : uint8(255u);
        int32 E = 0;
        #pragma unroll
        for (uint32 j = 0; j < BAND_LEN; ++j)
        {
            const uint8 g = (j + 1 < BAND_LEN) ? win[j] : g_in;
            const int32 diag = H[j] + scoring.substitution(i + j, i, g, q, qq);
            int32 h;
            if (j + 1 < BAND_LEN)
            {
                F[j] = nvbio::max(F[j + 1] + Ge, H[j + 1] + Go);           // previous row, one text position further
                h = (j == 0) ? nvbio::max(F[j], diag) : nvbio::max3(F[j], E, diag);
            }
            else { F[j] = floor_; h = nvbio::max(E, diag); }
            if (TYPE == LOCAL) { h = nvbio::max(h, int32(0)); sink.report(h, make_uint2(i + j + 1u, i + 1u)); }
            H[j] = h;
            E = (j == 0) ? h + Go : nvbio::max(h + Go, E + Ge);
        }
        // slide the window: cell j of the next row sees what cell j+1 saw
        #pragma unroll
        for (uint32 j = 0; j + 2 < BAND_LEN; ++j) win[j] = win[j + 1];
        if (BAND_LEN >= 2) win[BAND_LEN - 2] = cached_symbol<BAND_LEN>(g_in);
    }
    if (TYPE == GLOBAL) sink.report(H[BAND_LEN - 1], make_uint2(M + BAND_LEN - 1u, M));
    else if (TYPE == SEMI_GLOBAL)
    {
        const uint32 m = nvbio::min(M + BAND_LEN - 1u, N) - (M - 1u);
        for (uint32 j = 0; j < BAND_LEN; ++j)
            if (j == 0 || j < m) sink.report(H[j], make_uint2(M + j, M));
    }
    return true;
}

// ------------------------------------------------------------------ banded, linear gaps (Smith-Waterman / edit distance)
// vertical moves (a pattern symbol against no text) cost deletion(), horizontal ones insertion()  (sw_banded_inl.h:384-441)
template <uint32 BAND_LEN, AlignmentType TYPE, typename scheme_type, typename pattern_string, typename qual_string, typename text_string, typename sink_type>
NVBIO_HOST_DEVICE inline
bool banded_sw_score(const scheme_type& scoring, const pattern_string pattern, const qual_string quals, const text_string text, sink_type& sink)
{
    const uint32 M = pattern.length(), N = text.length();
    if (N < M) return false;
    const int32 G = scoring.deletion(), I = scoring.insertion();
    int32 B[BAND_LEN];
    uint8 win[BAND_LEN];
    for (uint32 j = 0; j + 1 < BAND_LEN; ++j) win[j] = cached_symbol<BAND_LEN>(uint8(text[j]));
    for (uint32 j = 0; j < BAND_LEN; ++j) B[j] = (TYPE == GLOBAL) ? int32(j) * G : 0;                    // sw_banded_inl.h:47-58
    for (uint32 i = 0; i < M; ++i)
    {
        const uint8 q = uint8(pattern[i]), qq = uint8(quals[i]);
        const int32 V = scoring.match(qq);
        const uint8 g_in = (i + BAND_LEN - 1u < N) ? uint8(text[i + BAND_LEN - 1u]) : uint8(255u);
        #pragma unroll
        for (uint32 j = 0; j < BAND_LEN; ++j)
        {
            const uint8 g = (j + 1 < BAND_LEN) ? win[j] : g_in;
            int32 h = B[j] + (g == q ? V : scoring.mismatch(g, q, qq));
            if (j + 1 < BAND_LEN) h = nvbio::max(h, B[j + 1] + G);
            if (j > 0)            h = nvbio::max(h, B[j - 1] + I);
            if (TYPE == LOCAL) { h = nvbio::max(h, int32(0)); sink.report(h, make_uint2(i + j + 1u, i + 1u)); }
            B[j] = h;
        }
        #pragma unroll
        for (uint32 j = 0; j + 2 < BAND_LEN; ++j) win[j] = win[j + 1];
        if (BAND_LEN >= 2) win[BAND_LEN - 2] = cached_symbol<BAND_LEN>(g_in);
    }
    if (TYPE == GLOBAL) sink.report(B[BAND_LEN - 1], make_uint2(M + BAND_LEN - 1u, M));
    else if (TYPE == SEMI_GLOBAL)
    {
        const uint32 m = nvbio::min(M + BAND_LEN - 1u, N) - (M - 1u);
        for (uint32 j = 0; j < BAND_LEN; ++j)
            if (j == 0 || j < m) sink.report(B[j], make_uint2(M + j, M));
    }
    return true;
}

// dispatch on the aligner kind
template <uint32 BAND_LEN, AlignmentType TYPE, typename S, typename A, typename P, typename Q, typename T, typename K>
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool banded_score(const GotohAligner<TYPE, S, A>& al, const P p, const Q q, const T t, K& sink)
{ return banded_gotoh_score<BAND_LEN, TYPE>(al.scheme, p, q, t, sink); }
template <uint32 BAND_LEN, AlignmentType TYPE, typename S, typename A, typename P, typename Q, typename T, typename K>
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool banded_score(const SmithWatermanAligner<TYPE, S, A>& al, const P p, const Q q, const T t, K& sink)
{ return banded_sw_score<BAND_LEN, TYPE>(al.scheme, p, q, t, sink); }
template <uint32 BAND_LEN, AlignmentType TYPE, typename A, typename P, typename Q, typename T, typename K>
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool banded_score(const EditDistanceAligner<TYPE, A>&, const P p, const Q q, const T t, K& sink)
{ return banded_sw_score<BAND_LEN, TYPE>(EditDistanceSWScheme(), p, q, t, sink); }

// ------------------------------------------------------------------ banded, bit-vector edit distance (MyersTag)
// EditDistanceAligner<TYPE, MyersTag<A>> selects Hyyro's banded form of Myers' bit-vector algorithm (myers_banded_inl.h:236-291 with no
// pre-loaded pattern symbols): one 32-bit word holds the vertical deltas of the BAND_LEN cells of a column, the band slides one pattern
// row per text column while the pattern lasts ("diagonal" columns: the tracked cell is the band's lowest), then stays put ("horizontal"
// columns: the tracked cell is the pattern's last row).  What distinguishes its results from the banded DP above, all kept:
//   * the threshold is taken as int16 (a -2^30 "no threshold" becomes 0: only exact occurrences report);
//   * scores are reported only from the horizontal columns, in text order (a BestSink keeps the last of equal scores);
//   * a text as long as the pattern reads delta bit BAND_LEN of the horizontal word in its single horizontal column.
template <uint32 ALPHABET_SIZE>
struct symbol_masks
{
    uint32 eq[ALPHABET_SIZE];
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE symbol_masks() { for (uint32 c = 0; c < ALPHABET_SIZE; ++c) eq[c] = 0u; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE static uint32 slot(const uint8 c) { return (ALPHABET_SIZE & (ALPHABET_SIZE - 1u)) ? uint32(c) : uint32(c) & (ALPHABET_SIZE - 1u); }
    /// the band moves down one pattern row: every mask loses its top row, `c` enters at the bottom (bit `bit`)
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE void slide() { for (uint32 c = 0; c < ALPHABET_SIZE; ++c) eq[c] >>= 1; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE void enter(const uint8 c, const uint32 bit)
    { const uint32 k = slot(c); for (uint32 a = 0; a < ALPHABET_SIZE; ++a) eq[a] |= (a == k) ? bit : 0u; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 of(const uint8 c) const
    { const uint32 k = slot(c); uint32 r = 0u; for (uint32 a = 0; a < ALPHABET_SIZE; ++a) r = (a == k) ? eq[a] : r; return r; }
};
/// one text column of the band: vertical deltas (plus / minus) in, out one row lower; returns the diagonal-zero and horizontal words
struct band_deltas
{
    uint32 plus, minus;
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE band_deltas() : plus(0xFFFFFFFFu), minus(0u) {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE void column(const uint32 eq, uint32& d0, uint32& h_plus, uint32& h_minus)
    {
        const uint32 x = eq | minus;
        d0      = ((plus + (x & plus)) ^ plus) | x;
        h_minus = plus & d0;
        h_plus  = minus | ~(plus | d0);
        const uint32 up = d0 >> 1;
        minus = up & h_plus;
        plus  = h_minus | ~(up | h_plus);
    }
};
template <uint32 BAND_LEN, AlignmentType TYPE, uint32 ALPHABET_SIZE, typename pattern_string, typename text_string, typename sink_type>
NVBIO_HOST_DEVICE inline
bool banded_bitvector_score(const pattern_string pattern, const text_string text, const int16 min_score, sink_type& sink)
{
    const uint32 M = pattern.length(), N = text.length();
    if (N < M) return false;
    symbol_masks<ALPHABET_SIZE> masks;
    band_deltas v;
    int32 score = 0;
    const uint32 bottom = 1u << (BAND_LEN - 1u);
    const uint32 n_diag = nvbio::min(N - 1u, M);
    uint32 d0, hp, hn;
    for (uint32 i = 0; i < n_diag; ++i)
    {
        masks.slide(); masks.enter(uint8(pattern[i]), bottom);
        v.column(masks.of(uint8(text[i])), d0, hp, hn);
        score -= (d0 & bottom) ? 0 : 1;
    }
    int32 row = int32(BAND_LEN - 1u + M - n_diag);            // bit of the pattern's last row in the column about to be computed
    for (uint32 i = n_diag; i < N && row >= 0; ++i, --row)
    {
        masks.slide();
        v.column(masks.of(uint8(text[i])), d0, hp, hn);
        score -= int32((hp >> row) & 1u) - int32((hn >> row) & 1u);
        if (TYPE == SEMI_GLOBAL && score >= min_score) sink.report(score, make_uint2(i + 1u, M));
    }
    if (TYPE == GLOBAL && score >= min_score) sink.report(score, make_uint2(N, M));
    return true;
}
template <uint32 BAND_LEN, AlignmentType TYPE, uint32 A, typename P, typename Q, typename T, typename K>
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool banded_score(const EditDistanceAligner<TYPE, MyersTag<A> >&, const P p, const Q, const T t, const int32 min_score, K& sink)
{ return banded_bitvector_score<BAND_LEN, TYPE, A>(p, t, int16(min_score), sink); }
/// every other aligner: the whole-pattern banded DP has no use for the threshold
template <uint32 BAND_LEN, typename aligner_type, typename P, typename Q, typename T, typename K>
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool banded_score(const aligner_type& al, const P p, const Q q, const T t, const int32, K& sink)
{ return banded_score<BAND_LEN>(al, p, q, t, sink); }

// ------------------------------------------------------------------ full matrix
// One template covers the four sweeps.  `outer` is the string walked in blocks of BL symbols (the text for
// TextBlockingTag, the pattern for PatternBlockingTag), `inner` the other one; the boundary between consecutive blocks
// is a column of int16 {H[,E]} per inner symbol.  LINEAR selects the Smith-Waterman recurrence.
template <typename scheme_type, bool LINEAR> struct full_costs {};
template <typename scheme_type> struct full_costs<scheme_type, false>
{
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE full_costs(const scheme_type& s) : sc(s) {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE int32 open() const { return sc.pattern_gap_open(); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE int32 ext()  const { return sc.pattern_gap_extension(); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE int32 text_open() const { return sc.text_gap_open(); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE int32 text_ext()  const { return sc.text_gap_extension(); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE int32 sub(const uint32 ti, const uint32 pi, const uint8 r, const uint8 q, const uint8 qq) const { return sc.substitution(ti, pi, r, q, qq); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE int32 best_match() const { return sc.match(255); }
    const scheme_type& sc;
};
template <typename scheme_type> struct full_costs<scheme_type, true>
{
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE full_costs(const scheme_type& s) : sc(s) {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE int32 del() const { return sc.deletion(); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE int32 ins() const { return sc.insertion(); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE int32 sub(const uint32, const uint32, const uint8 r, const uint8 q, const uint8 qq) const { return r == q ? sc.match(qq) : sc.mismatch(r, q, qq); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE int32 best_match() const { return sc.match(255); }
    const scheme_type& sc;
};

// text blocking, affine gaps (gotoh_inl.h:969-1489): blocks of 8 text columns, column over the pattern
template <AlignmentType TYPE, typename scheme_type, typename pattern_string, typename qual_string, typename text_string, typename sink_type, typename column_type>
NVBIO_HOST_DEVICE inline
bool gotoh_score_text_blocking(const scheme_type& scoring, const pattern_string pattern, const qual_string quals, const text_string text,
                               const int32 min_score, sink_type& sink, column_type column)
{
    const uint32 BL = 8u;
    const uint32 M = pattern.length(), N = text.length();
    const int32 Go = scoring.pattern_gap_open(), Ge = scoring.pattern_gap_extension();
    const int32 floor_ = int32(Field_traits<int16>::min()) - nvbio::min(Go, Ge);
    for (uint32 i = 0; i < M; ++i)
    {
        column[2u * i]      = int16(TYPE != LOCAL ? scoring.text_gap_open() + scoring.text_gap_extension() * int32(i) : 0);
        column[2u * i + 1u] = int16(TYPE == LOCAL ? 0 : floor_);
    }
    int32 H[BL + 1], F[BL + 1];
    uint8 r[BL];
    for (uint32 t = 0; t < BL; ++t) r[t] = 0;
    const uint32 padded = BL * ((N + BL - 1u) / BL);
    const uint32 end_block = padded > BL ? padded : BL;
    for (uint32 block = 0; block + BL <= end_block; block += BL)
    {
        const bool last = (block + BL == end_block);
        for (uint32 t = 0; t < BL; ++t)
            if (block + t < N) r[t] = uint8(text[block + t]);
        for (uint32 j = 0; j <= BL; ++j)
        {
            H[j] = (TYPE == GLOBAL) ? (block + j > 0 ? Go + Ge * int32(block + j - 1u) : 0) : 0;
            F[j] = floor_;
        }
        int32 best_edge = -(1 << 30);
        int32 carry = H[0];
        for (uint32 i = 0; i < M; ++i)
        {
            const uint8 q = uint8(pattern[i]), qq = uint8(quals[i]);
            int32 diag = carry;
            H[0] = carry = column[2u * i];
            int32 E = column[2u * i + 1u];
            for (uint32 j = 1; j <= BL; ++j)
            {
                F[j] = nvbio::max(F[j] + Ge, H[j] + Go);
                E    = nvbio::max(E + Ge, H[j - 1] + Go);
                int32 h = nvbio::max3(E, F[j], diag + scoring.substitution(block + j - 1u, i, r[j - 1], q, qq));
                if (TYPE == LOCAL) h = nvbio::max(h, int32(0));
                diag = H[j];
                H[j] = h;
            }
            column[2u * i] = int16(H[BL]); column[2u * i + 1u] = int16(E);            // the boundary is kept as shorts
            best_edge = nvbio::max(best_edge, H[BL]);
            if (TYPE == LOCAL)
                for (uint32 j = 1; j <= BL; ++j)
                    if (!last || block + j <= N) sink.report(H[j], make_uint2(block + j, i + 1u));
        }
        if (!last)
        {
            if (TYPE == SEMI_GLOBAL)
                for (uint32 j = 1; j <= BL; ++j) sink.report(H[j], make_uint2(block + j, M));
            if (best_edge + int32(N - block - BL) * scoring.match(255) < min_score) return false;
        }
        else if (TYPE == SEMI_GLOBAL) { for (uint32 j = 1; j <= BL; ++j) if (block + j <= N) sink.report(H[j], make_uint2(block + j, M)); }
        else if (TYPE == GLOBAL)      { for (uint32 j = 1; j <= BL; ++j) if (block + j == N) sink.report(H[j], make_uint2(block + j, M)); }
    }
    return true;
}

// text blocking, linear gaps (sw_inl.h:881-1222): blocks of 16 text columns, no early exit in this form
template <AlignmentType TYPE, typename scheme_type, typename pattern_string, typename qual_string, typename text_string, typename sink_type, typename column_type>
NVBIO_HOST_DEVICE inline
bool sw_score_text_blocking(const scheme_type& scoring, const pattern_string pattern, const qual_string quals, const text_string text,
                            const int32, sink_type& sink, column_type column)
{
    const uint32 BL = 16u;
    const uint32 M = pattern.length(), N = text.length();
    const int32 G = scoring.deletion(), I = scoring.insertion();
    for (uint32 i = 0; i < M; ++i) column[i] = int16(TYPE != LOCAL ? I * int32(i + 1u) : 0);
    int32 B[BL + 1];
    uint8 r[BL];
    for (uint32 t = 0; t < BL; ++t) r[t] = 0;
    const uint32 padded = BL * ((N + BL - 1u) / BL);
    const uint32 end_block = padded > BL ? padded : BL;
    for (uint32 block = 0; block + BL <= end_block; block += BL)
    {
        const bool last = (block + BL == end_block);
        for (uint32 j = 0; j <= BL; ++j) B[j] = (TYPE == GLOBAL) ? G * int32(block + j) : 0;
        for (uint32 t = 0; t < BL; ++t)
            if (block + t < N) r[t] = uint8(text[block + t]);
        int32 carry = B[0];
        for (uint32 i = 0; i < M; ++i)
        {
            const uint8 q = uint8(pattern[i]), qq = uint8(quals[i]);
            const int32 V = scoring.match(qq);
            int32 diag = carry;
            B[0] = carry = column[i];
            for (uint32 j = 1; j <= BL; ++j)
            {
                int32 h = nvbio::max3(B[j] + I, B[j - 1] + G, diag + (r[j - 1] == q ? V : scoring.mismatch(r[j - 1], q, qq)));
                if (TYPE == LOCAL) h = nvbio::max(h, int32(0));
                diag = B[j];
                B[j] = h;
            }
            column[i] = int16(B[BL]);
            if (TYPE == LOCAL)
                for (uint32 j = 1; j <= BL; ++j)
                    if (!last || block + j <= N) sink.report(B[j], make_uint2(block + j, i + 1u));
        }
        if (TYPE == SEMI_GLOBAL)         { for (uint32 j = 1; j <= BL; ++j) if (!last || block + j <= N) sink.report(B[j], make_uint2(block + j, M)); }
        else if (TYPE == GLOBAL && last) { for (uint32 j = 1; j <= BL; ++j) if (block + j == N) sink.report(B[j], make_uint2(block + j, M)); }
    }
    return true;
}

// pattern blocking, both recurrences (gotoh_inl.h:459-900 with 8-symbol blocks, sw_inl.h:417-760 with 16): column over the text
template <bool LINEAR, AlignmentType TYPE, typename scheme_type, typename pattern_string, typename qual_string, typename text_string, typename sink_type, typename column_type>
NVBIO_HOST_DEVICE inline
bool score_pattern_blocking(const scheme_type& scoring, const pattern_string pattern, const qual_string quals, const text_string text,
                            const int32 min_score, sink_type& sink, column_type column)
{
    const uint32 BL = LINEAR ? 16u : 8u;
    const uint32 M = pattern.length(), N = text.length();
    const full_costs<scheme_type, LINEAR> c(scoring);
    int32 Go = 0, Ge = 0, G = 0, I = 0, floor_ = 0;
    if constexpr (LINEAR) { G = c.del(); I = c.ins(); }
    else { Go = c.open(); Ge = c.ext(); floor_ = int32(Field_traits<int16>::min()) - nvbio::min(Go, Ge); }
    for (uint32 i = 0; i < N; ++i)
    {
        if constexpr (LINEAR) column[i] = int16(TYPE == GLOBAL ? G * int32(i + 1u) : 0);
        else { column[2u * i] = int16(TYPE == GLOBAL ? c.text_open() + c.text_ext() * int32(i) : 0); column[2u * i + 1u] = int16(TYPE == LOCAL ? 0 : floor_); }
    }
    int32 H[17], F[17];
    uint8 q[16], qq[16];
    for (uint32 t = 0; t < 16u; ++t) { q[t] = 0; qq[t] = 0; }
    const uint32 padded = BL * ((M + BL - 1u) / BL);
    const uint32 end_block = padded > BL ? padded : BL;
    for (uint32 block = 0; block + BL <= end_block; block += BL)
    {
        const bool last = (block + BL == end_block);
        for (uint32 t = 0; t < BL; ++t)
            if (block + t < M) { q[t] = uint8(pattern[block + t]); qq[t] = uint8(quals[block + t]); }
        for (uint32 j = 0; j <= BL; ++j)
        {
            if constexpr (LINEAR) H[j] = (TYPE != LOCAL) ? I * int32(block + j) : 0;
            else { H[j] = (TYPE != LOCAL) ? (block + j > 0 ? Go + Ge * int32(block + j - 1u) : 0) : 0; F[j] = floor_; }
        }
        int32 best_edge = int32(-2147483647 - 1);
        int32 carry = H[0];
        for (uint32 i = 0; i < N; ++i)
        {
            const uint8 r = uint8(text[i]);
            int32 diag = carry, E = 0;
            if constexpr (LINEAR) { H[0] = carry = column[i]; }
            else                  { H[0] = carry = column[2u * i]; E = column[2u * i + 1u]; }
            for (uint32 j = 1; j <= BL; ++j)
            {
                int32 h;
                if constexpr (LINEAR) h = nvbio::max3(H[j] + G, H[j - 1] + I, diag + c.sub(i, block + j - 1u, r, q[j - 1], qq[j - 1]));
                else
                {
                    F[j] = nvbio::max(F[j] + Ge, H[j] + Go);
                    E    = nvbio::max(E + Ge, H[j - 1] + Go);
                    h = nvbio::max3(E, F[j], diag + c.sub(i, block + j - 1u, r, q[j - 1], qq[j - 1]));
                }
                if (TYPE == LOCAL) h = nvbio::max(h, int32(0));
                diag = H[j];
                H[j] = h;
            }
            if constexpr (LINEAR) column[i] = int16(H[BL]);
            else { column[2u * i] = int16(H[BL]); column[2u * i + 1u] = int16(E); }
            best_edge = nvbio::max(best_edge, H[BL]);
            if (TYPE == LOCAL)
            {
                for (uint32 j = 1; j <= BL; ++j)
                    if (!last || block + j <= M) sink.report(H[j], make_uint2(i + 1u, block + j));
            }
            else if (last && TYPE == SEMI_GLOBAL) sink.report(H[((M - 1u) & (BL - 1u)) + 1u], make_uint2(i + 1u, M));
        }
        if (!last && int64(best_edge) + int64(M - block - BL) * c.best_match() < int64(min_score)) return false;
    }
    if (TYPE == GLOBAL) sink.report(H[((M - 1u) & (BL - 1u)) + 1u], make_uint2(N, M));
    return true;
}

// ungapped ("Hamming") scoring (hamming/hamming_inl.h:413-1270): the linear-gap sweeps with the gap terms removed and every
// border zero -- H(i,j) = H(i-1,j-1) + S, clamped at zero for LOCAL; 16-symbol blocks, the same report order as the sweeps above.
// TEXT_BLOCKING walks the text in blocks with a column over the pattern (no early exit), otherwise the pattern in blocks with
// a column over the text (early exit on the block's right edge, :654-656).
template <bool TEXT_BLOCKING, AlignmentType TYPE, typename scheme_type, typename pattern_string, typename qual_string, typename text_string, typename sink_type, typename column_type>
NVBIO_HOST_DEVICE inline
bool hamming_score(const scheme_type& scoring, const pattern_string pattern, const qual_string quals, const text_string text,
                   const int32 min_score, sink_type& sink, column_type column)
{
    const uint32 BL = 16u;
    const uint32 M = pattern.length(), N = text.length();
    const uint32 OUTER = TEXT_BLOCKING ? N : M, INNER = TEXT_BLOCKING ? M : N;     // blocked-over string, column string
    for (uint32 i = 0; i < INNER; ++i) column[i] = int16(0);
    int32 H[BL + 1];
    uint8 sym[BL], ql[BL];
    for (uint32 t = 0; t < BL; ++t) { sym[t] = 0; ql[t] = 0; }
    const uint32 padded = BL * ((OUTER + BL - 1u) / BL);
    const uint32 end_block = padded > BL ? padded : BL;
    for (uint32 block = 0; block + BL <= end_block; block += BL)
    {
        const bool last = (block + BL == end_block);
        for (uint32 t = 0; t < BL; ++t)
            if (block + t < OUTER)
            {
                if (TEXT_BLOCKING) sym[t] = uint8(text[block + t]);
                else { sym[t] = uint8(pattern[block + t]); ql[t] = uint8(quals[block + t]); }
            }
        for (uint32 j = 0; j <= BL; ++j) H[j] = 0;
        int32 best_edge = int32(-2147483647 - 1);
        int32 carry = 0;
        for (uint32 i = 0; i < INNER; ++i)
        {
            const uint8 o = TEXT_BLOCKING ? uint8(pattern[i]) : uint8(text[i]);
            const uint8 oq = TEXT_BLOCKING ? uint8(quals[i]) : uint8(0);
            int32 diag = carry;
            H[0] = carry = column[i];
            for (uint32 j = 1; j <= BL; ++j)
            {
                const uint8 r = TEXT_BLOCKING ? sym[j - 1] : o, q = TEXT_BLOCKING ? o : sym[j - 1], qq = TEXT_BLOCKING ? oq : ql[j - 1];
                int32 h = diag + (r == q ? scoring.match(qq) : scoring.mismatch(r, q, qq));
                if (TYPE == LOCAL) h = nvbio::max(h, int32(0));
                diag = H[j];
                H[j] = h;
            }
            column[i] = int16(H[BL]);
            best_edge = nvbio::max(best_edge, H[BL]);
            if (TYPE == LOCAL)
            {
                for (uint32 j = 1; j <= BL; ++j)
                    if (!last || block + j <= OUTER) sink.report(H[j], TEXT_BLOCKING ? make_uint2(block + j, i + 1u) : make_uint2(i + 1u, block + j));
            }
            else if (!TEXT_BLOCKING && last && TYPE == SEMI_GLOBAL) sink.report(H[((M - 1u) & (BL - 1u)) + 1u], make_uint2(i + 1u, M));
        }
        if (TEXT_BLOCKING)
        {
            if (TYPE == SEMI_GLOBAL)         { for (uint32 j = 1; j <= BL; ++j) if (!last || block + j <= N) sink.report(H[j], make_uint2(block + j, M)); }
            else if (TYPE == GLOBAL && last) { for (uint32 j = 1; j <= BL; ++j) if (block + j == N) sink.report(H[j], make_uint2(block + j, M)); }
        }
        else if (!last && int64(best_edge) + int64(M - block - BL) * scoring.match(255) < int64(min_score)) return false;
    }
    if (!TEXT_BLOCKING && TYPE == GLOBAL) sink.report(H[((M - 1u) & (BL - 1u)) + 1u], make_uint2(N, M));
    return true;
}

/// the boundary column as the int16 entries the sweeps index: callers hand in a pointer to the reference's
/// column_storage_type<aligner>::type (short2 per symbol for Gotoh, int16 otherwise) or to plain int16s
template <typename T> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE T      column_words(T c)       { return c; }
#if defined(__HIPCC__)
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE int16* column_words(short2* c) { return reinterpret_cast<int16*>(c); }
#endif

template <AlignmentType TYPE, typename S, typename P, typename Q, typename T, typename K, typename C>
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool full_score(const GotohAligner<TYPE, S, TextBlockingTag>& al, const P p, const Q q, const T t, const int32 ms, K& sink, C col)
{ return gotoh_score_text_blocking<TYPE>(al.scheme, p, q, t, ms, sink, col); }
template <AlignmentType TYPE, typename S, typename P, typename Q, typename T, typename K, typename C>
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool full_score(const GotohAligner<TYPE, S, PatternBlockingTag>& al, const P p, const Q q, const T t, const int32 ms, K& sink, C col)
{ return score_pattern_blocking<false, TYPE>(al.scheme, p, q, t, ms, sink, col); }
template <AlignmentType TYPE, typename S, typename P, typename Q, typename T, typename K, typename C>
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool full_score(const SmithWatermanAligner<TYPE, S, TextBlockingTag>& al, const P p, const Q q, const T t, const int32 ms, K& sink, C col)
{ return sw_score_text_blocking<TYPE>(al.scheme, p, q, t, ms, sink, col); }
template <AlignmentType TYPE, typename S, typename P, typename Q, typename T, typename K, typename C>
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool full_score(const SmithWatermanAligner<TYPE, S, PatternBlockingTag>& al, const P p, const Q q, const T t, const int32 ms, K& sink, C col)
{ return score_pattern_blocking<true, TYPE>(al.scheme, p, q, t, ms, sink, col); }
template <AlignmentType TYPE, typename P, typename Q, typename T, typename K, typename C>
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool full_score(const EditDistanceAligner<TYPE, TextBlockingTag>&, const P p, const Q q, const T t, const int32 ms, K& sink, C col)
{ return sw_score_text_blocking<TYPE>(EditDistanceSWScheme(), p, q, t, ms, sink, col); }
template <AlignmentType TYPE, typename P, typename Q, typename T, typename K, typename C>
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool full_score(const EditDistanceAligner<TYPE, PatternBlockingTag>&, const P p, const Q q, const T t, const int32 ms, K& sink, C col)
{ return score_pattern_blocking<true, TYPE>(EditDistanceSWScheme(), p, q, t, ms, sink, col); }

template <AlignmentType TYPE, typename S, typename P, typename Q, typename T, typename K, typename C>
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool full_score(const HammingDistanceAligner<TYPE, S, TextBlockingTag>& al, const P p, const Q q, const T t, const int32 ms, K& sink, C col)
{ return hamming_score<true, TYPE>(al.scheme, p, q, t, ms, sink, col); }
template <AlignmentType TYPE, typename S, typename P, typename Q, typename T, typename K, typename C>
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool full_score(const HammingDistanceAligner<TYPE, S, PatternBlockingTag>& al, const P p, const Q q, const T t, const int32 ms, K& sink, C col)
{ return hamming_score<false, TYPE>(al.scheme, p, q, t, ms, sink, col); }

/// int16 entries of boundary column an aligner needs for a (pattern, text) pair: per symbol of the string it does NOT block over
template <typename aligner_type> struct column_entries {};
template <AlignmentType T, typename S> struct column_entries< GotohAligner<T, S, TextBlockingTag> >          { NVBIO_HOST_DEVICE static uint32 get(uint32 M, uint32)   { return 2u * M; } };
template <AlignmentType T, typename S> struct column_entries< GotohAligner<T, S, PatternBlockingTag> >       { NVBIO_HOST_DEVICE static uint32 get(uint32, uint32 N)   { return 2u * N; } };
template <AlignmentType T, typename S> struct column_entries< SmithWatermanAligner<T, S, TextBlockingTag> >    { NVBIO_HOST_DEVICE static uint32 get(uint32 M, uint32) { return M; } };
template <AlignmentType T, typename S> struct column_entries< SmithWatermanAligner<T, S, PatternBlockingTag> > { NVBIO_HOST_DEVICE static uint32 get(uint32, uint32 N) { return N; } };
template <AlignmentType T> struct column_entries< EditDistanceAligner<T, TextBlockingTag> >                  { NVBIO_HOST_DEVICE static uint32 get(uint32 M, uint32)   { return M; } };
template <AlignmentType T> struct column_entries< EditDistanceAligner<T, PatternBlockingTag> >               { NVBIO_HOST_DEVICE static uint32 get(uint32, uint32 N)   { return N; } };
template <AlignmentType T, typename S> struct column_entries< HammingDistanceAligner<T, S, TextBlockingTag> >    { NVBIO_HOST_DEVICE static uint32 get(uint32 M, uint32) { return M; } };
template <AlignmentType T, typename S> struct column_entries< HammingDistanceAligner<T, S, PatternBlockingTag> > { NVBIO_HOST_DEVICE static uint32 get(uint32, uint32 N) { return N; } };

} // namespace priv

// ---------------------------------------------------------------------------------------- public per-thread functions
/// banded_alignment_score<BAND_LEN>(aligner, pattern, quals, text, min_score, sink)   (alignment.h:257-272).
/// min_score only matters to the reference's windowed (staged) form and to the bit-vector edit distance; the whole-pattern banded DP ignores it.
template <uint32 BAND_LEN, typename aligner_type, typename pattern_string, typename qual_string, typename text_string, typename sink_type>
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE
bool banded_alignment_score(const aligner_type aligner, const pattern_string pattern, const qual_string quals, const text_string text,
                            const int32 min_score, sink_type& sink)
{
    return priv::banded_score<BAND_LEN>(aligner, pattern, quals, text, min_score, sink);
}
/// ... without qualities (alignment.h:284-298)
template <uint32 BAND_LEN, typename aligner_type, typename pattern_string, typename text_string, typename sink_type>
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE
bool banded_alignment_score(const aligner_type aligner, const pattern_string pattern, const text_string text, const int32 min_score, sink_type& sink)
{
    return banded_alignment_score<BAND_LEN>(aligner, pattern, trivial_quality_string(), text, min_score, sink);
}
/// ... returning the best score (alignment.h:310-352)
template <uint32 BAND_LEN, typename aligner_type, typename pattern_string, typename qual_string, typename text_string>
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE
int32 banded_alignment_score(const aligner_type aligner, const pattern_string pattern, const qual_string quals, const text_string text, const int32 min_score)
{
    BestSink<int32> sink;
    banded_alignment_score<BAND_LEN>(aligner, pattern, quals, text, min_score, sink);
    return sink.score;
}
template <uint32 BAND_LEN, typename aligner_type, typename pattern_string, typename text_string>
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE
int32 banded_alignment_score(const aligner_type aligner, const pattern_string pattern, const text_string text, const int32 min_score)
{
    BestSink<int32> sink;
    banded_alignment_score<BAND_LEN>(aligner, pattern, trivial_quality_string(), text, min_score, sink);
    return sink.score;
}

/// alignment_score(aligner, pattern, quals, text, min_score, sink, column)   (alignment.h:455-472): the low-level full DP;
/// `column` is caller storage of int16 entries, priv::column_entries<aligner>::get(M, N) of them.
template <typename aligner_type, typename pattern_string, typename qual_string, typename text_string, typename sink_type, typename column_type>
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE
bool alignment_score(const aligner_type aligner, const pattern_string pattern, const qual_string quals, const text_string text,
                     const int32 min_score, sink_type& sink, column_type column)
{
    return priv::full_score(aligner, pattern, quals, text, min_score, sink, priv::column_words(column));
}
/// ... with the column in local storage sized by a compile-time bound on the blocked-over string (alignment.h:510-527)
template <uint32 MAX_LEN, typename aligner_type, typename pattern_string, typename qual_string, typename text_string, typename sink_type>
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE
bool alignment_score(const aligner_type aligner, const pattern_string pattern, const qual_string quals, const text_string text,
                     const int32 min_score, sink_type& sink)
{
    int16 column[2u * MAX_LEN];
    return priv::full_score(aligner, pattern, quals, text, min_score, sink, &column[0]);
}

} // namespace aln
} // namespace nvbio

#include "batched.h"
