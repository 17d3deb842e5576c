// compat/nvbio/alignment/traceback.h -- the per-thread tracebacks of nvbio::aln as host-device templates over ANY string
// iterator, scheme and backtracer -- the generic half of BatchedBandedAlignmentTraceback / BatchedAlignmentTraceback (batched.h),
// run one lane per job for the streams the tuned kernels do not take (8-bit strings, user schemes, asymmetric linear gaps) and under
// HostThreadScheduler.  What a job computes is what the reference's drivers compute:
//   banded   banded_alignment_traceback (nvbio/alignment/banded_inl.h:352-489): a scoring pass over the band that records, per
//            cell, the flow directions new_cell() is handed (gotoh_banded_inl.h:479-614, sw_banded_inl.h:405-470), and the walk back
//            from the best sink (gotoh_banded_inl.h:878-960, sw_banded_inl.h:748-800);
//   full     alignment_traceback (alignment_inl.h:365-480): the sink of the pattern-blocking score pass, the flow flags of
//            gotoh_inl.h:512-560 / sw_inl.h:475-500, the walk of gotoh_inl.h:1806-1870 / sw_inl.h:1660-1700 and the completion
//            along the first row / column for GLOBAL and SEMI_GLOBAL (alignment_inl.h:443-466).
// The reference recomputes the flags window by window from int16 checkpoints; one dense pass stores the same flags while the DP
// values fit int16, which is the range the reference itself is exact in.  The caller provides the flag storage: one byte per cell
// (pattern length x BAND_LEN, or text length x pattern length), plus -- full matrix -- two rows of pattern length + 1 int32 and
// the boundary column of the score pass (2 x text length int16; text length for linear gaps).
// The backtracer receives clip(pattern length - sink.y), push(op) for every step from the END of the alignment backwards, and
// clip(source.y), exactly the calls of banded_inl.h:383-426.
#pragma once
#include "alignment.h"

namespace nvbio {
namespace aln {
namespace priv {

enum { FLOW_H_MASK = 3u, FLOW_INSERTION_EXT = 4u, FLOW_DELETION_EXT = 8u };

// ------------------------------------------------------------------ banded, affine gaps
template <uint32 BAND_LEN, AlignmentType TYPE, typename scheme_type, typename pattern_string, typename qual_string, typename text_string, typename backtracer_type>
NVBIO_HOST_DEVICE inline
Alignment<int32> banded_gotoh_traceback(const scheme_type& scoring, const pattern_string pattern, const qual_string quals, const text_string text,
                                        backtracer_type& backtracer, uint8* flags)
{
    const uint2 none = make_uint2(uint32(-1), uint32(-1));
    BestSink<int32> best;
    const uint32 M = pattern.length(), N = text.length();
    if (N < M) return Alignment<int32>(best.score, none, none);
    const int32 Go = scoring.pattern_gap_open(), Ge = scoring.pattern_gap_extension();
    const int32 floor_ = int32(Field_traits<int16>::min()) - nvbio::max(nvbio::max(Go, Ge), nvbio::max(scoring.text_gap_open(), scoring.text_gap_extension()));
    int32 H[BAND_LEN], F[BAND_LEN];
    uint8 win[BAND_LEN];
    for (uint32 j = 0; j + 1 < BAND_LEN; ++j) win[j] = cached_symbol<BAND_LEN>(uint8(text[j]));
    for (uint32 j = 0; j < BAND_LEN; ++j) { H[j] = (TYPE == GLOBAL && j > 0) ? scoring.text_gap_open() + int32(j - 1u) * scoring.text_gap_extension() : 0; F[j] = floor_; }
    for (uint32 i = 0; i < M; ++i)
    {
        const uint8 q = uint8(pattern[i]), qq = uint8(quals[i]);
        const uint8 g_in = (i + BAND_LEN - 1u < N) ? uint8(text[i + BAND_LEN - 1u]) : uint8(255u);
        uint8* row = flags + uint64(i) * BAND_LEN;
        int32 E = 0;
        uint8 edir = SUBSTITUTION;
        for (uint32 j = 0; j < BAND_LEN; ++j)
        {
            const uint8 g = (j + 1 < BAND_LEN) ? win[j] : g_in;
            const int32 diag = H[j] + scoring.substitution(i + j, i, g, q, qq);
            uint8 fdir = SUBSTITUTION, hdir;
            int32 h;
            if (j + 1 < BAND_LEN)
            {
                const int32 ftop = F[j + 1] + Ge, htop = H[j + 1] + Go;
                F[j] = nvbio::max(ftop, htop);
                fdir = ftop > htop ? uint8(FLOW_DELETION_EXT) : uint8(SUBSTITUTION);
                if (j == 0) { h = nvbio::max(F[0], diag); hdir = F[0] > diag ? uint8(INSERTION) : uint8(SUBSTITUTION); }
                else
                {
                    h = nvbio::max3(F[j], E, diag);
                    hdir = F[j] > E ? (F[j] > diag ? uint8(INSERTION) : uint8(SUBSTITUTION)) : (E > diag ? uint8(DELETION) : uint8(SUBSTITUTION));
                }
            }
            else { F[j] = floor_; h = nvbio::max(E, diag); hdir = E > diag ? uint8(DELETION) : uint8(SUBSTITUTION); }
            if (TYPE == LOCAL) { h = nvbio::max(h, int32(0)); if (h == 0) hdir = uint8(SINK); best.report(h, make_uint2(i + j + 1u, i + 1u)); }
            H[j] = h;
            row[j] = uint8(hdir | (j == 0 ? uint8(SUBSTITUTION) : edir) | fdir);
            if (j == 0) E = h + Go;
            else
            {
                const int32 eleft = E + Ge, ediag = h + Go;
                edir = eleft > ediag ? uint8(FLOW_INSERTION_EXT) : uint8(SUBSTITUTION);
                E = nvbio::max(ediag, eleft);
            }
        }
        for (uint32 j = 0; j + 2 < BAND_LEN; ++j) win[j] = win[j + 1];
        if (BAND_LEN >= 2) win[BAND_LEN - 2] = cached_symbol<BAND_LEN>(g_in);
    }
    if (TYPE == GLOBAL) best.report(H[BAND_LEN - 1], make_uint2(M + BAND_LEN - 1u, M));
    else if (TYPE == SEMI_GLOBAL)
    {
        const uint32 m = nvbio::min(M + BAND_LEN - 1u, N) - (M - 1u);
        for (uint32 j = 0; j < BAND_LEN; ++j) if (j == 0 || j < m) best.report(H[j], make_uint2(M + j, M));
    }
    if (best.sink.x == uint32(-1) || best.sink.y == uint32(-1)) return Alignment<int32>(best.score, none, none);

    backtracer.clip(M - best.sink.y);
    int32 entry = int32(best.sink.x - best.sink.y), r = int32(best.sink.y) - 1;
    uint32 state = 0;               // 0 = H, 1 = E (text gap being extended), 2 = F
    uint2 source = make_uint2(0u, 0u);
    bool stopped = false;
    while (r >= 0)
    {
        const uint8 op = flags[uint64(r) * BAND_LEN + uint32(entry)], h_op = op & FLOW_H_MASK;
        if (TYPE == LOCAL && state == 0 && h_op == SINK) { source.y = uint32(r) + 1u; source.x = uint32(entry) + source.y; stopped = true; break; }
        if (state == 1)      { if ((op & FLOW_INSERTION_EXT) == 0u) state = 0; --entry; backtracer.push(DELETION); }
        else if (state == 2) { if ((op & FLOW_DELETION_EXT) == 0u) state = 0; ++entry; --r; backtracer.push(INSERTION); }
        else if (h_op == DELETION)  state = 1;
        else if (h_op == INSERTION) state = 2;
        else { --r; backtracer.push(SUBSTITUTION); }
    }
    if (!stopped) source = make_uint2(uint32(entry), 0u);
    backtracer.clip(source.y);
    return Alignment<int32>(best.score, source, best.sink);
}

// ------------------------------------------------------------------ banded, linear gaps
template <uint32 BAND_LEN, AlignmentType TYPE, typename scheme_type, typename pattern_string, typename qual_string, typename text_string, typename backtracer_type>
NVBIO_HOST_DEVICE inline
Alignment<int32> banded_sw_traceback(const scheme_type& scoring, const pattern_string pattern, const qual_string quals, const text_string text,
                                     backtracer_type& backtracer, uint8* flags)
{
    const uint2 none = make_uint2(uint32(-1), uint32(-1));
    BestSink<int32> best;
    const uint32 M = pattern.length(), N = text.length();
    if (N < M) return Alignment<int32>(best.score, none, none);
    const int32 G = scoring.deletion(), I = scoring.insertion();
    int32 B[BAND_LEN];
    uint8 win[BAND_LEN];
    for (uint32 j = 0; j + 1 < BAND_LEN; ++j) win[j] = cached_symbol<BAND_LEN>(uint8(text[j]));
    for (uint32 j = 0; j < BAND_LEN; ++j) B[j] = (TYPE == GLOBAL) ? int32(j) * G : 0;
    for (uint32 i = 0; i < M; ++i)
    {
        const uint8 q = uint8(pattern[i]), qq = uint8(quals[i]);
        const int32 V = scoring.match(qq);
        const uint8 g_in = (i + BAND_LEN - 1u < N) ? uint8(text[i + BAND_LEN - 1u]) : uint8(255u);
        uint8* row = flags + uint64(i) * BAND_LEN;
        for (uint32 j = 0; j < BAND_LEN; ++j)
        {
            const uint8 g = (j + 1 < BAND_LEN) ? win[j] : g_in;
            const int32 diag = B[j] + (g == q ? V : scoring.mismatch(g, q, qq));
            int32 h; uint8 dir;
            if (j == 0 && BAND_LEN > 1)   { const int32 top = B[1] + G; h = nvbio::max(top, diag); dir = top > diag ? uint8(INSERTION) : uint8(SUBSTITUTION); }
            else if (j + 1 < BAND_LEN)
            {
                const int32 top = B[j + 1] + G, left = B[j - 1] + I;
                h = nvbio::max3(top, left, diag);
                dir = top > left ? (top > diag ? uint8(INSERTION) : uint8(SUBSTITUTION)) : (left > diag ? uint8(DELETION) : uint8(SUBSTITUTION));
            }
            else { const int32 left = (j > 0) ? B[j - 1] + I : diag; h = nvbio::max(left, diag); dir = left > diag ? uint8(DELETION) : uint8(SUBSTITUTION); }
            if (TYPE == LOCAL) { h = nvbio::max(h, int32(0)); best.report(h, make_uint2(i + j + 1u, i + 1u)); }
            B[j] = h; row[j] = dir;               // the banded context keeps `dir` as is: no SINK marks, the LOCAL walk ends at row zero
        }
        for (uint32 j = 0; j + 2 < BAND_LEN; ++j) win[j] = win[j + 1];
        if (BAND_LEN >= 2) win[BAND_LEN - 2] = cached_symbol<BAND_LEN>(g_in);
    }
    if (TYPE == GLOBAL) best.report(B[BAND_LEN - 1], make_uint2(M + BAND_LEN - 1u, M));
    else if (TYPE == SEMI_GLOBAL)
    {
        const uint32 m = nvbio::min(M + BAND_LEN - 1u, N) - (M - 1u);
        for (uint32 j = 0; j < BAND_LEN; ++j) if (j == 0 || j < m) best.report(B[j], make_uint2(M + j, M));
    }
    if (best.sink.x == uint32(-1) || best.sink.y == uint32(-1)) return Alignment<int32>(best.score, none, none);
    backtracer.clip(M - best.sink.y);
    int32 entry = int32(best.sink.x - best.sink.y), r = int32(best.sink.y) - 1;
    while (r >= 0)
    {
        const uint8 op = flags[uint64(r) * BAND_LEN + uint32(entry)];
        if (op == DELETION)       { --entry; backtracer.push(DELETION); }
        else if (op == INSERTION) { ++entry; --r; backtracer.push(INSERTION); }
        else                      { --r; backtracer.push(SUBSTITUTION); }
    }
    backtracer.clip(0u);
    return Alignment<int32>(best.score, make_uint2(uint32(entry), 0u), best.sink);
}

// ------------------------------------------------------------------ full matrix, both recurrences
// rows: 2 x (M + 1) int32; column: the score pass's boundary (2N int16 affine, N linear); flags: N x M bytes, cell (text row i, pattern column j)
template <bool LINEAR, AlignmentType TYPE, typename scheme_type, typename pattern_string, typename qual_string, typename text_string, typename backtracer_type>
NVBIO_HOST_DEVICE inline
Alignment<int32> full_traceback(const scheme_type& scoring, const pattern_string pattern, const qual_string quals, const text_string text,
                                backtracer_type& backtracer, uint8* flags, int32* rows, int16* column)
{
    const uint2 none = make_uint2(uint32(-1), uint32(-1));
    BestSink<int32> best;
    const uint32 M = pattern.length(), N = text.length();
    score_pattern_blocking<LINEAR, TYPE>(scoring, pattern, quals, text, int32(-2147483647 - 1), best, column);
    if (best.sink.x == uint32(-1) || best.sink.y == uint32(-1)) return Alignment<int32>(best.score, none, none);
    const full_costs<scheme_type, LINEAR> c(scoring);
    int32* hrow = rows; int32* frow = rows + (M + 1u);
    int32 Go = 0, Ge = 0, G = 0, I = 0, floor_ = 0;
    if constexpr (LINEAR) { G = c.del(); I = c.ins(); }
    else { Go = c.open(); Ge = c.ext(); floor_ = int32(Field_traits<int16>::min()) - nvbio::min(Go, Ge); }
    for (uint32 j = 0; j <= M; ++j)
    {
        if constexpr (LINEAR) hrow[j] = (TYPE != LOCAL) ? I * int32(j) : 0;
        else { hrow[j] = (TYPE != LOCAL) ? (j > 0 ? Go + Ge * int32(j - 1u) : 0) : 0; frow[j] = floor_; }
    }
    for (uint32 i = 0; i < N; ++i)
    {
        const uint8 r_i = uint8(text[i]);
        int32 diag_h = hrow[0];
        if constexpr (LINEAR) hrow[0] = (TYPE == GLOBAL) ? G * int32(i + 1u) : 0;
        else                  hrow[0] = (TYPE == GLOBAL) ? c.text_open() + c.text_ext() * int32(i) : 0;
        int32 E = (TYPE == LOCAL) ? 0 : floor_;
        for (uint32 j = 1; j <= M; ++j)
        {
            const uint8 q_j = uint8(pattern[j - 1u]), qq = uint8(quals[j - 1u]);
            int32 h; uint8 flag;
            if constexpr (LINEAR)
            {
                const int32 diagonal = diag_h + c.sub(i, j - 1u, r_i, q_j, qq), top = hrow[j] + G, left = hrow[j - 1] + I;
                h = nvbio::max3(top, left, diagonal);
                if (TYPE == LOCAL) h = nvbio::max(h, int32(0));
                flag = top > left ? (top > diagonal ? uint8(DELETION) : uint8(SUBSTITUTION)) : (left > diagonal ? uint8(INSERTION) : uint8(SUBSTITUTION));
                if (TYPE == LOCAL && h == 0) flag = uint8(SINK);
            }
            else
            {
                const int32 ftop = frow[j] + Ge, htop = hrow[j] + Go;
                frow[j] = nvbio::max(ftop, htop);
                const uint8 fdir = ftop > htop ? uint8(FLOW_DELETION_EXT) : uint8(SUBSTITUTION);
                const int32 eleft = E + Ge, hleft = hrow[j - 1] + Go;
                E = nvbio::max(eleft, hleft);
                const uint8 edir = eleft > hleft ? uint8(FLOW_INSERTION_EXT) : uint8(SUBSTITUTION);
                const int32 diagonal = diag_h + c.sub(i, j - 1u, r_i, q_j, qq), top = frow[j], left = E;
                h = nvbio::max3(left, top, diagonal);
                if (TYPE == LOCAL) h = nvbio::max(h, int32(0));
                uint8 hdir = top > left ? (top > diagonal ? uint8(DELETION) : uint8(SUBSTITUTION)) : (left > diagonal ? uint8(INSERTION) : uint8(SUBSTITUTION));
                if (TYPE == LOCAL && h == 0) hdir = uint8(SINK);
                flag = uint8(hdir | edir | fdir);
            }
            diag_h = hrow[j];
            hrow[j] = h;
            flags[uint64(i) * M + (j - 1u)] = flag;
        }
    }
    backtracer.clip(M - best.sink.y);
    int32 r = int32(best.sink.x), col = int32(best.sink.y) - 1;
    uint32 state = 0;
    while (r > 0 && col >= 0)
    {
        const uint8 op = flags[uint64(r - 1) * M + uint32(col)], h_op = op & FLOW_H_MASK;
        if constexpr (LINEAR)
        {
            if (TYPE == LOCAL && op == SINK) break;
            if (op != DELETION)  --col;
            if (op != INSERTION) --r;
            backtracer.push(DirectionVector(op));
        }
        else
        {
            if (TYPE == LOCAL && state == 0 && h_op == SINK) break;
            if (state == 1)      { if ((op & FLOW_INSERTION_EXT) == 0u) state = 0; --col; backtracer.push(INSERTION); }
            else if (state == 2) { if ((op & FLOW_DELETION_EXT) == 0u) state = 0; --r; backtracer.push(DELETION); }
            else if (h_op == INSERTION) state = 1;
            else if (h_op == DELETION)  state = 2;
            else { --col; --r; backtracer.push(SUBSTITUTION); }
        }
    }
    uint32 sx = uint32(r), sy = uint32(col + 1);
    if (TYPE == SEMI_GLOBAL || TYPE == GLOBAL) { if (sx == 0) for (; sy > 0; --sy) backtracer.push(INSERTION); }
    if (TYPE == GLOBAL)                        { if (sy == 0) for (; sx > 0; --sx) backtracer.push(DELETION); }
    backtracer.clip(sy);
    return Alignment<int32>(best.score, make_uint2(sx, sy), best.sink);
}

// dispatch on the aligner kind
template <uint32 BAND_LEN, AlignmentType TYPE, typename S, typename A, typename P, typename Q, typename T, typename B>
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE Alignment<int32> banded_traceback(const GotohAligner<TYPE, S, A>& al, const P p, const Q q, const T t, B& bt, uint8* flags)
{ return banded_gotoh_traceback<BAND_LEN, TYPE>(al.scheme, p, q, t, bt, flags); }
template <uint32 BAND_LEN, AlignmentType TYPE, typename S, typename A, typename P, typename Q, typename T, typename B>
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE Alignment<int32> banded_traceback(const SmithWatermanAligner<TYPE, S, A>& al, const P p, const Q q, const T t, B& bt, uint8* flags)
{ return banded_sw_traceback<BAND_LEN, TYPE>(al.scheme, p, q, t, bt, flags); }
template <uint32 BAND_LEN, AlignmentType TYPE, typename A, typename P, typename Q, typename T, typename B>
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE Alignment<int32> banded_traceback(const EditDistanceAligner<TYPE, A>&, const P p, const Q q, const T t, B& bt, uint8* flags)
{ return banded_sw_traceback<BAND_LEN, TYPE>(EditDistanceSWScheme(), p, q, t, bt, flags); }

template <AlignmentType TYPE, typename S, typename A, typename P, typename Q, typename T, typename B>
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE Alignment<int32> matrix_traceback(const GotohAligner<TYPE, S, A>& al, const P p, const Q q, const T t, B& bt, uint8* flags, int32* rows, int16* column)
{ return full_traceback<false, TYPE>(al.scheme, p, q, t, bt, flags, rows, column); }
template <AlignmentType TYPE, typename S, typename A, typename P, typename Q, typename T, typename B>
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE Alignment<int32> matrix_traceback(const SmithWatermanAligner<TYPE, S, A>& al, const P p, const Q q, const T t, B& bt, uint8* flags, int32* rows, int16* column)
{ return full_traceback<true, TYPE>(al.scheme, p, q, t, bt, flags, rows, column); }
template <AlignmentType TYPE, typename A, typename P, typename Q, typename T, typename B>
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE Alignment<int32> matrix_traceback(const EditDistanceAligner<TYPE, A>&, const P p, const Q q, const T t, B& bt, uint8* flags, int32* rows, int16* column)
{ return full_traceback<true, TYPE>(EditDistanceSWScheme(), p, q, t, bt, flags, rows, column); }

/// bytes of per-job scratch the generic tracebacks need (flags, then 8-byte aligned rows, then the score pass's column)
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint64 banded_traceback_scratch(const uint32 band, const uint32 maxP) { return (uint64(maxP) * band + 15u) & ~uint64(15); }
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint64 full_traceback_flag_bytes(const uint32 maxP, const uint32 maxT) { return (uint64(maxP) * maxT + 15u) & ~uint64(15); }
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint64 full_traceback_row_bytes(const uint32 maxP) { return (uint64(maxP + 1u) * 8u + 15u) & ~uint64(15); }
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint64 full_traceback_column_bytes(const uint32 maxT) { return (uint64(maxT) * 4u + 16u + 15u) & ~uint64(15); }
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint64 full_traceback_scratch(const uint32 maxP, const uint32 maxT)
{ return full_traceback_flag_bytes(maxP, maxT) + full_traceback_row_bytes(maxP) + full_traceback_column_bytes(maxT); }

} // namespace priv

// ---------------------------------------------------------------------------------------- public per-thread functions
// The storage-less forms (banded_inl.h:450-489, alignment_inl.h:482-560): the reference sizes checkpoint + sub-matrix scratch
// from its template bounds and keeps it in local memory; here the scratch is the dense flag matrix of the generic tracebacks
// above -- MAX_PATTERN_LEN x BAND_LEN (banded) or MAX_PATTERN_LEN x MAX_TEXT_LEN (full) bytes -- taken from the heap in host
// code and from the lane's private memory in device code (CHECKPOINTS is accepted and unused: nothing is recomputed).
namespace priv {
template <typename T, uint64 N> struct scratch
{
#if defined(NVBIO_DEVICE_COMPILATION)
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE T* get() { return m; }
    T m[N];
#else
    scratch() : m(new T[N]) {}
    ~scratch() { delete[] m; }
    T* get() { return m; }
    T* m;
#endif
};
} // namespace priv

template <uint32 BAND_LEN, uint32 MAX_PATTERN_LEN, uint32 CHECKPOINTS, typename aligner_type, typename pattern_string, typename qual_string, typename text_string, typename backtracer_type>
NVBIO_HOST_DEVICE inline
Alignment<int32> banded_alignment_traceback(const aligner_type aligner, const pattern_string pattern, const qual_string quals, const text_string text,
                                            const int32 /*min_score*/, backtracer_type& backtracer)
{
    priv::scratch<uint8, uint64(MAX_PATTERN_LEN) * BAND_LEN> flags;
    return priv::banded_traceback<BAND_LEN>(aligner, pattern, quals, text, backtracer, flags.get());
}
template <uint32 MAX_PATTERN_LEN, uint32 MAX_TEXT_LEN, uint32 CHECKPOINTS, typename aligner_type, typename pattern_string, typename qual_string, typename text_string, typename backtracer_type>
NVBIO_HOST_DEVICE inline
Alignment<int32> alignment_traceback(const aligner_type aligner, const pattern_string pattern, const qual_string quals, const text_string text,
                                     const int32 /*min_score*/, backtracer_type& backtracer)
{
    priv::scratch<uint8, uint64(MAX_PATTERN_LEN) * MAX_TEXT_LEN> flags;
    priv::scratch<int32, 2ull * (MAX_PATTERN_LEN + 1u)>          rows;
    priv::scratch<int16, 2ull * MAX_TEXT_LEN + 8u>               column;
    return priv::matrix_traceback(aligner, pattern, quals, text, backtracer, flags.get(), rows.get(), column.get());
}

} // namespace aln
} // namespace nvbio
