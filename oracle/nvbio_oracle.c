/*
 * nvbio_oracle.c -- CPU restatement of the nvbio seed-and-extend hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity checker for the HIP
 * product path in nvbio_amd/csrc.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it; the product library never links,
 * calls or falls back to anything in oracle/.
 *
 * It restates, in plain scalar C, the arithmetic of the reference's *host*
 * path (the same template source the reference compiles for host and device):
 *
 *   banded Gotoh score      nvbio/alignment/gotoh/gotoh_banded_inl.h:415-658
 *     row-zero init         nvbio/alignment/gotoh/gotoh_banded_inl.h:46-77
 *     text register cache   nvbio/alignment/alignment_base_inl.h:75-98
 *     BestSink              nvbio/alignment/sink_inl.h:38-68
 *     SimpleGotohScheme     nvbio/alignment/utils.h:114-134
 *     host batch scheduler  nvbio/alignment/batched_banded_inl.h:97-128
 *   packed streams          nvbio/basic/packedstream_inl.h:37-75,336-400
 *   occurrence table        nvbio/fmindex/rank_dictionary_inl.h:42-77
 *   rank / rank4 (uint4,K=64) nvbio/fmindex/rank_dictionary_inl.h:424-573
 *   2-bit popcounts         nvbio/basic/popcount_inl.h:239-362,484-493
 *   count table             nvbio/fmindex/bwt.h:77-88
 *   fm_index rank/rank4     nvbio/fmindex/fmindex_inl.h:36-186
 *   match (backward search) nvbio/fmindex/fmindex_inl.h:307-341
 *   locate / ssa iterators  nvbio/fmindex/fmindex_inl.h:466-569
 *   sampled SA              nvbio/fmindex/ssa_inl.h:263-309,486-504
 *   BWT from SA             nvbio/fmindex/bwt.h:47-60
 *   interleaved bwt|occ     nvbio/io/fmindex/fmindex_impl.cu:305-327
 *   FMIndexFilter (host)    nvbio/fmindex/filter_inl.h:200-259
 *
 * plus the independent plain-loop checker the reference's own test-suite uses
 * as a differential oracle:
 *
 *   ref_banded_sw (Gotoh)   nvbio-test/alignment_test_utils.h:314-460
 *
 * Parity pinning: the reference itself cannot be compiled in this image
 * without writing stand-ins for CUDA headers it includes unconditionally
 * (vector_types.h, cuda_runtime.h) and its CUB submodule is not vendored, so
 * no oracle/_ref build exists.  The restatement is pinned against the known
 * answers held by the reference's own tests (tests/golden/kat.json, see
 * tests/test_oracle_kat.py) and against the values SURVEY.md 8(c) recorded
 * from the reference's compiled host path.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#if defined(_OPENMP)
#include <omp.h>
#endif

#define ORACLE_API __attribute__((visibility("default")))

enum { ALN_GLOBAL = 0, ALN_LOCAL = 1, ALN_SEMI_GLOBAL = 2 }; /* alignment_base.h:54 */

/* ------------------------------------------------------------------------ */
/* packed streams: packedstream_inl.h:37-75 (generic pow2), 336-371 (2-bit), */
/* 373-400 (4-bit). 8-bit big-endian follows the generic pow2 formula.       */
/* ------------------------------------------------------------------------ */
static inline uint32_t ps_get(const uint32_t* w, uint32_t bits, uint32_t big_endian, uint64_t idx)
{
    if (bits == 2) {
        const uint32_t word = w[idx >> 4];
        const uint32_t off  = big_endian ? (30u - ((uint32_t)(idx & 15u) << 1)) : ((uint32_t)(idx & 15u) << 1);
        return (word >> off) & 3u;
    } else if (bits == 4) {
        const uint32_t word = w[idx >> 3];
        const uint32_t off  = big_endian ? (28u - ((uint32_t)(idx & 7u) << 2)) : ((uint32_t)(idx & 7u) << 2);
        return (word >> off) & 15u;
    } else { /* 8 */
        const uint32_t word = w[idx >> 2];
        const uint32_t bit  = (uint32_t)(idx & 3u) << 3;
        const uint32_t off  = big_endian ? (24u - bit) : bit;
        return (word >> off) & 255u;
    }
}

static inline void ps_set(uint32_t* w, uint32_t bits, uint32_t big_endian, uint64_t idx, uint32_t sym)
{
    const uint32_t per  = 32u / bits;
    const uint32_t mask = (bits == 32) ? 0xFFFFFFFFu : ((1u << bits) - 1u);
    const uint32_t k    = (uint32_t)(idx % per);
    const uint32_t off  = big_endian ? (32u - bits - k * bits) : (k * bits);
    uint32_t word = w[idx / per];
    word &= ~(mask << off);
    word |= (sym & mask) << off;
    w[idx / per] = word;
}

ORACLE_API void oracle_pack(const uint8_t* sym, uint64_t n, uint32_t bits, uint32_t big_endian, uint32_t* words)
{
    for (uint64_t i = 0; i < n; ++i) ps_set(words, bits, big_endian, i, sym[i]);
}
ORACLE_API void oracle_unpack(const uint32_t* words, uint64_t begin, uint64_t n, uint32_t bits, uint32_t big_endian, uint8_t* sym)
{
    for (uint64_t i = 0; i < n; ++i) sym[i] = (uint8_t)ps_get(words, bits, big_endian, begin + i);
}

/* ------------------------------------------------------------------------ */
/* Banded Gotoh score: gotoh_banded_inl.h:415-658                            */
/* ------------------------------------------------------------------------ */
typedef struct { int32_t score; uint32_t sink_x, sink_y; } best_sink_t;

static inline void sink_init(best_sink_t* s)
{   /* sink_inl.h:38-40, numbers.h:832-835 */
    s->score = -(1 << 30); s->sink_x = 0xFFFFFFFFu; s->sink_y = 0xFFFFFFFFu;
}
static inline void sink_report(best_sink_t* s, int32_t score, uint32_t x, uint32_t y)
{   /* sink_inl.h:57-68 : '<=' so that the last report wins ties */
    if (s->score <= score) { s->score = score; s->sink_x = x; s->sink_y = y; }
}
static inline int32_t imax(int32_t a, int32_t b) { return a > b ? a : b; }
/* DirectionVector */
enum { DIR_SUBSTITUTION = 0, DIR_INSERTION = 1, DIR_DELETION = 2, DIR_SINK = 3, DIR_INSERTION_EXT = 4, DIR_DELETION_EXT = 8 };   /* alignment_base.h:139-150 */

#define MAX_BAND 64

/* The reference keeps the text window in Reference_cache<BAND_LEN>
 * (alignment_base_inl.h:75-98): plain uint32 registers for BAND_LEN in
 * {3,5,7,15}, a 2-bit PackedStream otherwise -- whose set() masks the stored
 * symbol to 2 bits (packedstream_inl.h:352-369).  So for e.g. BAND_LEN=31 a
 * text symbol re-read from the cache is (symbol & 3), and in particular the
 * out-of-range marker 255 comes back as 3. */
static inline uint32_t cache_store(uint32_t band, uint32_t g)
{
    return (band == 3 || band == 5 || band == 7 || band == 15) ? g : (g & 3u);
}


/* Scoring scheme as the DP sees it (the GotohAligner interface, nvbio/alignment/utils.h:114-134 and
 * nvBowtie/bowtie2/cuda/scoring.h:283-293): substitution(r,q,qq) = (r == q) ? match : mismatch(qq);
 * mm_lut == NULL: SimpleGotohScheme (constant mismatch).  mm_lut != NULL: nvBowtie's
 * SmithWatermanScoringScheme, mismatch(qq) = -m_mmp(qq) tabulated for the 256 quality bytes. */
typedef struct {
    int32_t match, mismatch;
    int32_t pat_gap_open, pat_gap_ext, txt_gap_open, txt_gap_ext;
    const int32_t* mm_lut;
    const uint8_t* quals;          /* quality of pattern symbol at stream index pat_begin + i */
} scheme_t;

static int banded_gotoh_score_x(uint32_t band, int type, const scheme_t* sc,
    const uint32_t* pat_w, uint32_t pat_bits, uint32_t pat_be, uint64_t pat_begin, uint32_t pattern_len,
    const uint32_t* txt_w, uint32_t txt_bits, uint32_t txt_be, uint64_t txt_begin, uint32_t text_len,
    best_sink_t* sink);

static int banded_gotoh_score(
    uint32_t band, int type,
    int32_t s_match, int32_t s_mismatch, int32_t s_gap_open, int32_t s_gap_ext,
    const uint32_t* pat_w, uint32_t pat_bits, uint32_t pat_be, uint64_t pat_begin, uint32_t pattern_len,
    const uint32_t* txt_w, uint32_t txt_bits, uint32_t txt_be, uint64_t txt_begin, uint32_t text_len,
    best_sink_t* sink)
{
    scheme_t sc = { s_match, s_mismatch, s_gap_open, s_gap_ext, s_gap_open, s_gap_ext, NULL, NULL };
    return banded_gotoh_score_x(band, type, &sc, pat_w, pat_bits, pat_be, pat_begin, pattern_len,
                                txt_w, txt_bits, txt_be, txt_begin, text_len, sink);
}

static inline int32_t subst(const scheme_t* sc, uint8_t r, uint8_t q, uint8_t qq)
{
    return q == r ? sc->match : (sc->mm_lut ? sc->mm_lut[qq] : sc->mismatch);
}

static int banded_gotoh_score_x(uint32_t band, int type, const scheme_t* sc,
    const uint32_t* pat_w, uint32_t pat_bits, uint32_t pat_be, uint64_t pat_begin, uint32_t pattern_len,
    const uint32_t* txt_w, uint32_t txt_bits, uint32_t txt_be, uint64_t txt_begin, uint32_t text_len,
    best_sink_t* sink)
{
    if (text_len < pattern_len) return 0;                       /* :431-432 */

    uint32_t text_cache[MAX_BAND];
    int32_t  H_band[MAX_BAND], F_band[MAX_BAND];

    /* load first band of text (:441-442) -- no bounds check in the reference */
    for (uint32_t j = 0; j + 1 < band; ++j)
        text_cache[j] = cache_store(band, ps_get(txt_w, txt_bits, txt_be, txt_begin + j));

    const int32_t G_o = sc->pat_gap_open;                        /* pattern_gap_open      :444 */
    const int32_t G_e = sc->pat_gap_ext;                         /* pattern_gap_extension :445 */
    const int32_t infimum = -32768 - imax(imax(G_o, G_e), imax(sc->txt_gap_open, sc->txt_gap_ext)); /* :446-448 */

    /* init_row_zero :46-77 */
    H_band[0] = 0;
    for (uint32_t j = 1; j < band; ++j)
        H_band[j] = (type == ALN_GLOBAL) ? sc->txt_gap_open + (int32_t)(j - 1) * sc->txt_gap_ext : 0;   /* :57 */
    for (uint32_t j = 0; j < band; ++j)
        F_band[j] = infimum;

    for (uint32_t i = 0; i < pattern_len; ++i)                  /* :463 */
    {
        const uint8_t q  = (uint8_t)ps_get(pat_w, pat_bits, pat_be, pat_begin + i);
        const uint8_t qq = sc->quals ? sc->quals[pat_begin + i] : 0;       /* :470 */

        /* j == 0 (:483-515) */
        {
            const int32_t ftop = F_band[1] + G_e;
            const int32_t htop = H_band[1] + G_o;
            F_band[0] = imax(ftop, htop);
            const uint8_t g = (uint8_t)text_cache[0];
            const int32_t S_ij     = subst(sc, g, q, qq);
            const int32_t diagonal = H_band[0] + S_ij;
            const int32_t top      = F_band[0];
            int32_t hi = imax(top, diagonal);
            if (type == ALN_LOCAL) {
                hi = imax(hi, 0);
                sink_report(sink, hi, i + 1, i + 1);
            }
            H_band[0] = hi;
        }
        int32_t E_j = H_band[0] + G_o;                           /* :517 */

        for (uint32_t j = 1; j + 1 < band; ++j)                  /* :520-577 */
        {
            const int32_t ftop = F_band[j + 1] + G_e;
            const int32_t htop = H_band[j + 1] + G_o;
            F_band[j] = imax(ftop, htop);

            const uint32_t g = text_cache[j]; text_cache[j - 1] = g;      /* :542 */
            const int32_t S_ij     = subst(sc, (uint8_t)g, q, qq);
            const int32_t diagonal = H_band[j] + S_ij;
            const int32_t top      = F_band[j];
            const int32_t left     = E_j;
            int32_t hi = imax(imax(top, left), diagonal);
            if (type == ALN_LOCAL) {
                hi = imax(hi, 0);
                sink_report(sink, hi, i + j + 1, i + 1);
            }
            H_band[j] = hi;
            const int32_t eleft     = E_j + G_e;
            const int32_t ediagonal = hi + G_o;
            E_j = imax(ediagonal, eleft);
        }

        /* load the new text character (:580-581) */
        const uint8_t g = (i + band - 1 < text_len)
            ? (uint8_t)ps_get(txt_w, txt_bits, txt_be, txt_begin + i + band - 1) : 255u;
        text_cache[band - 2] = cache_store(band, g);

        /* j == BAND_LEN-1 (:584-614) -- uses the raw g, not the cached copy */
        {
            F_band[band - 1] = infimum;
            const int32_t S_ij     = subst(sc, g, q, qq);
            const int32_t diagonal = H_band[band - 1] + S_ij;
            const int32_t left     = E_j;
            int32_t hi = imax(left, diagonal);
            if (type == ALN_LOCAL) {
                hi = imax(hi, 0);
                sink_report(sink, hi, i + band, i + 1);
            }
            H_band[band - 1] = hi;
        }
    }

    /* window_end == pattern_len always for the non-windowed entry point (:680-700) */
    if (type == ALN_GLOBAL)
        sink_report(sink, H_band[band - 1], pattern_len + band - 1, pattern_len);   /* :641-642 */
    else if (type == ALN_SEMI_GLOBAL)
    {
        const uint32_t a = pattern_len + band - 1u;
        const uint32_t m = (a < text_len ? a : text_len) - (pattern_len - 1u);       /* :645 */
        sink_report(sink, H_band[0], pattern_len + 0, pattern_len);
        for (uint32_t j = 1; j < band; ++j)
            if (j < m) sink_report(sink, H_band[j], pattern_len + j, pattern_len);
    }
    return 1;
}

/* One alignment; strings given as packed streams.  Returns the reference's
 * bool; sink_out = {score, sink.x, sink.y} of a fresh BestSink<int32>. */
ORACLE_API int oracle_banded_gotoh_score(
    uint32_t band, int type, const int32_t* scheme /* match,mismatch,gap_open,gap_ext */,
    const uint32_t* pat_w, uint32_t pat_bits, uint32_t pat_be, uint64_t pat_begin, uint32_t pat_len,
    const uint32_t* txt_w, uint32_t txt_bits, uint32_t txt_be, uint64_t txt_begin, uint32_t txt_len,
    int32_t* sink_out)
{
    best_sink_t s; sink_init(&s);
    const int r = banded_gotoh_score(band, type, scheme[0], scheme[1], scheme[2], scheme[3],
        pat_w, pat_bits, pat_be, pat_begin, pat_len, txt_w, txt_bits, txt_be, txt_begin, txt_len, &s);
    sink_out[0] = s.score; sink_out[1] = (int32_t)s.sink_x; sink_out[2] = (int32_t)s.sink_y;
    return r;
}

/* BatchedBandedAlignmentScore<BAND,stream,HostThreadScheduler>::enact
 * (batched_banded_inl.h:121-128): an OpenMP parallel-for over independent jobs,
 * each = init_context -> load_strings -> banded_alignment_score -> output
 * (batched_banded_inl.h:43-76). */
ORACLE_API void oracle_batch_banded_gotoh_score(
    uint32_t band, int type, const int32_t* scheme,
    const uint32_t* pat_w, uint32_t pat_bits, uint32_t pat_be, const uint64_t* pat_begin, const uint32_t* pat_len,
    const uint32_t* txt_w, uint32_t txt_bits, uint32_t txt_be, const uint64_t* txt_begin, const uint32_t* txt_len,
    uint32_t n, int32_t* out_score, uint32_t* out_sink /* 2 per job */, int n_threads)
{
#if defined(_OPENMP)
    if (n_threads > 0) omp_set_num_threads(n_threads);
    #pragma omp parallel for schedule(static)
#endif
    for (int64_t i = 0; i < (int64_t)n; ++i)
    {
        best_sink_t s; sink_init(&s);
        banded_gotoh_score(band, type, scheme[0], scheme[1], scheme[2], scheme[3],
            pat_w, pat_bits, pat_be, pat_begin[i], pat_len[i],
            txt_w, txt_bits, txt_be, txt_begin[i], txt_len[i], &s);
        out_score[i] = s.score; out_sink[2 * i] = s.sink_x; out_sink[2 * i + 1] = s.sink_y;
    }
}

/* The same batch with nvBowtie's quality-aware scheme (scoring.h:283-293): scheme6 = {match,
 * pattern_gap_open, pattern_gap_ext, text_gap_open, text_gap_ext, unused}; mm_lut[256] = mismatch(q);
 * quals[] indexed like the pattern stream's symbols. */
ORACLE_API void oracle_batch_banded_gotoh_score_qual(
    uint32_t band, int type, const int32_t* scheme6, const int32_t* mm_lut, const uint8_t* quals,
    const uint32_t* pat_w, uint32_t pat_bits, uint32_t pat_be, const uint64_t* pat_begin, const uint32_t* pat_len,
    const uint32_t* txt_w, uint32_t txt_bits, uint32_t txt_be, const uint64_t* txt_begin, const uint32_t* txt_len,
    uint32_t n, int32_t* out_score, uint32_t* out_sink, int n_threads)
{
    scheme_t sc = { scheme6[0], 0, scheme6[1], scheme6[2], scheme6[3], scheme6[4], mm_lut, quals };
#if defined(_OPENMP)
    if (n_threads > 0) omp_set_num_threads(n_threads);
    #pragma omp parallel for schedule(static)
#endif
    for (int64_t i = 0; i < (int64_t)n; ++i)
    {
        best_sink_t s; sink_init(&s);
        banded_gotoh_score_x(band, type, &sc, pat_w, pat_bits, pat_be, pat_begin[i], pat_len[i],
            txt_w, txt_bits, txt_be, txt_begin[i], txt_len[i], &s);
        out_score[i] = s.score; out_sink[2 * i] = s.sink_x; out_sink[2 * i + 1] = s.sink_y;
    }
}

/* QualCost<int>::operator() (nvBowtie/bowtie2/cuda/scoring.h:86-104): the float -> int truncation
 * is part of the reference's arithmetic, so the table is produced here exactly as written there. */
ORACLE_API void oracle_qual_cost_lut(int32_t min_val, int32_t max_val, int32_t* lut256)
{
    for (int i = 0; i < 256; ++i) {
        const float frac = (float)((i < 40 ? i : 40) / 40.0f);
        lut256[i] = min_val + (int32_t)(frac * (max_val - min_val));
    }
}

/* ------------------------------------------------------------------------ */
/* Full-matrix Gotoh score, text-blocking: gotoh_inl.h:969-1489              */
/*   (the form sw-benchmark instantiates: make_gotoh_aligner<TYPE,           */
/*    TextBlockingTag>, sw-benchmark.cu:604-631; BAND_LEN = 8 text columns   */
/*    per block, gotoh_inl.h:1491-1495; boundary column kept as short2       */
/*    {H,E} per pattern row, gotoh_inl.h:1476-1477)                          */
/* ------------------------------------------------------------------------ */
#define FULL_BAND 8
static int gotoh_score_text_blocking(int type, const scheme_t* sc,
    const uint32_t* pat_w, uint32_t pat_bits, uint32_t pat_be, uint64_t pat_begin, uint32_t M,
    const uint32_t* txt_w, uint32_t txt_bits, uint32_t txt_be, uint64_t txt_begin, uint32_t N,
    int32_t min_score, best_sink_t* sink, int16_t* temp /* 2*M scratch */)
{
    int32_t H_band[FULL_BAND + 1], F_band[FULL_BAND + 1];
    uint8_t r_cache[FULL_BAND];
    memset(r_cache, 0, sizeof r_cache);
    const int32_t G_o = sc->pat_gap_open, G_e = sc->pat_gap_ext, zero = 0;
    const int32_t infimum = -32768 - (G_o < G_e ? G_o : G_e);                    /* :1153 */
    /* context.init, GotohScoringContext (:69-93), TextBlockingTag branch */
    for (uint32_t i = 0; i < M; ++i) {
        temp[2 * i]     = (int16_t)(type != ALN_LOCAL ? sc->txt_gap_open + sc->txt_gap_ext * (int32_t)i : zero);
        temp[2 * i + 1] = (int16_t)(type == ALN_LOCAL ? zero : infimum);
    }
    const uint32_t nb = FULL_BAND * ((N + FULL_BAND - 1) / FULL_BAND);
    const uint32_t end_block = nb > FULL_BAND ? nb : FULL_BAND;                  /* :1160-1162, window_end == N */
    for (uint32_t block = 0; block + FULL_BAND <= end_block; block += FULL_BAND)
    {
        const int last = (block + FULL_BAND == end_block);                       /* the CHECK_N block (:1219-) */
        const uint32_t block_end = (block + FULL_BAND < N) ? block + FULL_BAND : N;
        for (uint32_t t = 0; t < FULL_BAND; ++t)
            if (!last || block + t < block_end) r_cache[t] = (uint8_t)ps_get(txt_w, txt_bits, txt_be, txt_begin + block + t);
        for (uint32_t j = 0; j <= FULL_BAND; ++j) {
            H_band[j] = (type == ALN_GLOBAL) ? (block + j > 0 ? G_o + G_e * (int32_t)(block + j - 1u) : zero) : zero;
            F_band[j] = infimum;
        }
        int32_t max_score = -(1 << 30);
        int32_t temp_i = H_band[0];
        for (uint32_t i = 0; i < M; ++i)
        {
            const uint8_t q_i  = (uint8_t)ps_get(pat_w, pat_bits, pat_be, pat_begin + i);
            const uint8_t qq_i = sc->quals ? sc->quals[pat_begin + i] : 0;
            /* update_row (:974-1098) */
            int32_t H_diag = temp_i;
            H_band[0] = temp_i = temp[2 * i];
            int32_t E = temp[2 * i + 1];
            for (uint32_t j = 1; j <= FULL_BAND; ++j)
            {
                F_band[j] = imax(F_band[j] + G_e, H_band[j] + G_o);
                E = imax(E + G_e, H_band[j - 1] + G_o);
                const int32_t diagonal = H_diag + subst(sc, r_cache[j - 1], q_i, qq_i);
                int32_t hi = imax(imax(E, F_band[j]), diagonal);
                if (type == ALN_LOCAL) hi = imax(hi, zero);
                H_diag = H_band[j];
                H_band[j] = hi;
            }
            temp[2 * i] = (int16_t)H_band[FULL_BAND]; temp[2 * i + 1] = (int16_t)E;      /* make_vector<short> :1065 */
            max_score = imax(max_score, H_band[FULL_BAND]);
            if (type == ALN_LOCAL)
                for (uint32_t j = 1; j <= FULL_BAND; ++j)
                    if (!last || block + j <= N) sink_report(sink, H_band[j], block + j, i + 1);
        }
        if (!last)
        {
            if (type == ALN_SEMI_GLOBAL)
                for (uint32_t j = 1; j <= FULL_BAND; ++j) sink_report(sink, H_band[j], block + j, M);
            const int32_t missing_cols = (int32_t)(N - block - FULL_BAND);
            if (max_score + missing_cols * sc->match < min_score) return 0;         /* :1212-1214 (match(255)) */
        }
        else
        {
            if (type == ALN_SEMI_GLOBAL) {
                for (uint32_t j = 1; j <= FULL_BAND; ++j) if (block + j <= N) sink_report(sink, H_band[j], block + j, M);
            } else if (type == ALN_GLOBAL) {
                for (uint32_t j = 1; j <= FULL_BAND; ++j) if (block + j == N) sink_report(sink, H_band[j], block + j, M);
            }
        }
    }
    return 1;
}

/* BatchedAlignmentScore<stream,HostThreadScheduler> over GotohAligner<TYPE,SimpleGotohScheme,TextBlockingTag>
 * (nvbio/alignment/batched_inl.h:236-300): out_ok[i] = the per-job bool (0 = early-exited). */
ORACLE_API void oracle_batch_gotoh_score(
    int type, const int32_t* scheme,
    const uint32_t* pat_w, uint32_t pat_bits, uint32_t pat_be, const uint64_t* pat_begin, const uint32_t* pat_len,
    const uint32_t* txt_w, uint32_t txt_bits, uint32_t txt_be, const uint64_t* txt_begin, const uint32_t* txt_len,
    const int32_t* min_score /* nullable: -(1<<30) */, uint32_t n, int32_t* out_score, uint32_t* out_sink, uint8_t* out_ok, int n_threads)
{
    scheme_t sc = { scheme[0], scheme[1], scheme[2], scheme[3], scheme[2], scheme[3], NULL, NULL };
#if defined(_OPENMP)
    if (n_threads > 0) omp_set_num_threads(n_threads);
    #pragma omp parallel for schedule(dynamic, 16)
#endif
    for (int64_t i = 0; i < (int64_t)n; ++i)
    {
        best_sink_t s; sink_init(&s);
        int16_t* temp = (int16_t*)malloc(sizeof(int16_t) * 2 * (size_t)(pat_len[i] + 1));
        const int ok = gotoh_score_text_blocking(type, &sc, pat_w, pat_bits, pat_be, pat_begin[i], pat_len[i],
            txt_w, txt_bits, txt_be, txt_begin[i], txt_len[i], min_score ? min_score[i] : -(1 << 30), &s, temp);
        free(temp);
        out_score[i] = s.score; out_sink[2 * i] = s.sink_x; out_sink[2 * i + 1] = s.sink_y;
        if (out_ok) out_ok[i] = (uint8_t)ok;
    }
}

/* ------------------------------------------------------------------------ */
/* Smith-Waterman (linear gaps) and edit-distance aligners                    */
/*   banded score     nvbio/alignment/sw/sw_banded_inl.h:47-58, 340-520       */
/*   full, text-block nvbio/alignment/sw/sw_inl.h:55-81 (init), 881-1222      */
/*                    BAND_LEN = 16 columns per block (:1330-1334), boundary   */
/*                    column kept as int16, no early exit in this variant     */
/*   edit distance    = the same code with EditDistanceSWScheme (0,-1,-1,-1): */
/*                    ed/ed_utils.h:44-51, ed/ed_inl.h:85-99, ed_banded_inl.h */
/* scheme = {match, mismatch, deletion, insertion}                            */
/* ------------------------------------------------------------------------ */
static int banded_sw_score_x(uint32_t band, int type, const int32_t* sw,
    const uint32_t* pat_w, uint32_t pat_bits, uint32_t pat_be, uint64_t pat_begin, uint32_t pattern_len,
    const uint32_t* txt_w, uint32_t txt_bits, uint32_t txt_be, uint64_t txt_begin, uint32_t text_len,
    best_sink_t* sink)
{
    if (text_len < pattern_len) return 0;
    uint32_t text_cache[MAX_BAND];
    int32_t  B[MAX_BAND];
    for (uint32_t j = 0; j + 1 < band; ++j)
        text_cache[j] = cache_store(band, ps_get(txt_w, txt_bits, txt_be, txt_begin + j));
    const int32_t V = sw[0], X = sw[1], G = sw[2], I = sw[3];
    for (uint32_t j = 0; j < band; ++j) B[j] = (type == ALN_GLOBAL) ? (int32_t)j * G : 0;       /* init_row_zero :47-58 */
    for (uint32_t i = 0; i < pattern_len; ++i)
    {
        const uint8_t q = (uint8_t)ps_get(pat_w, pat_bits, pat_be, pat_begin + i);
        {
            const uint8_t g = (uint8_t)text_cache[0];
            const int32_t diagonal = B[0] + (g == q ? V : X), top = B[1] + G;
            int32_t hi = imax(top, diagonal);
            if (type == ALN_LOCAL) { hi = imax(hi, 0); sink_report(sink, hi, i + 1, i + 1); }
            B[0] = hi;
        }
        for (uint32_t j = 1; j + 1 < band; ++j)
        {
            const uint32_t g = text_cache[j]; text_cache[j - 1] = g;
            const int32_t diagonal = B[j] + ((uint8_t)g == q ? V : X), top = B[j + 1] + G, left = B[j - 1] + I;
            int32_t hi = imax(imax(top, left), diagonal);
            if (type == ALN_LOCAL) { hi = imax(hi, 0); sink_report(sink, hi, i + j + 1, i + 1); }
            B[j] = hi;
        }
        const uint8_t g = (i + band - 1 < text_len) ? (uint8_t)ps_get(txt_w, txt_bits, txt_be, txt_begin + i + band - 1) : 255u;
        text_cache[band - 2] = cache_store(band, g);
        {
            const int32_t diagonal = B[band - 1] + (g == q ? V : X), left = B[band - 2] + I;
            int32_t hi = imax(left, diagonal);
            if (type == ALN_LOCAL) { hi = imax(hi, 0); sink_report(sink, hi, i + band, i + 1); }
            B[band - 1] = hi;
        }
    }
    if (type == ALN_GLOBAL)
        sink_report(sink, B[band - 1], pattern_len + band - 1, pattern_len);
    else if (type == ALN_SEMI_GLOBAL) {
        const uint32_t a = pattern_len + band - 1u;
        const uint32_t m = (a < text_len ? a : text_len) - (pattern_len - 1u);
        sink_report(sink, B[0], pattern_len, pattern_len);
        for (uint32_t j = 1; j < band; ++j) if (j < m) sink_report(sink, B[j], pattern_len + j, pattern_len);
    }
    return 1;
}

#define SW_FULL_BAND 16
static int sw_score_text_blocking(int type, const int32_t* sw,
    const uint32_t* pat_w, uint32_t pat_bits, uint32_t pat_be, uint64_t pat_begin, uint32_t M,
    const uint32_t* txt_w, uint32_t txt_bits, uint32_t txt_be, uint64_t txt_begin, uint32_t N,
    best_sink_t* sink, int16_t* temp /* M scratch */)
{
    int32_t band[SW_FULL_BAND + 1];
    uint8_t r_cache[SW_FULL_BAND];
    memset(r_cache, 0, sizeof r_cache);
    const int32_t V = sw[0], X = sw[1], G = sw[2], I = sw[3], zero = 0;
    for (uint32_t i = 0; i < M; ++i) temp[i] = (int16_t)(type != ALN_LOCAL ? I * (int32_t)(i + 1) : zero);     /* SWScoringContext::init :67-81 */
    const uint32_t nb = SW_FULL_BAND * ((N + SW_FULL_BAND - 1) / SW_FULL_BAND);
    const uint32_t end_block = nb > SW_FULL_BAND ? nb : SW_FULL_BAND;
    for (uint32_t block = 0; block + SW_FULL_BAND <= end_block; block += SW_FULL_BAND)
    {
        const int last = (block + SW_FULL_BAND == end_block);
        const uint32_t block_end = (block + SW_FULL_BAND < N) ? block + SW_FULL_BAND : N;
        for (uint32_t j = 0; j <= SW_FULL_BAND; ++j) band[j] = (type == ALN_GLOBAL) ? G * (int32_t)(block + j) : zero;
        for (uint32_t t = 0; t < SW_FULL_BAND; ++t)
            if (!last || block + t < block_end) r_cache[t] = (uint8_t)ps_get(txt_w, txt_bits, txt_be, txt_begin + block + t);
        int32_t temp_i = band[0];
        for (uint32_t i = 0; i < M; ++i)
        {
            const uint8_t q_i = (uint8_t)ps_get(pat_w, pat_bits, pat_be, pat_begin + i);
            int32_t prev = temp_i;                                      /* update_row :884-960 */
            band[0] = temp_i = temp[i];
            for (uint32_t j = 1; j <= SW_FULL_BAND; ++j)
            {
                const int32_t diagonal = prev + (r_cache[j - 1] == q_i ? V : X);
                const int32_t top = band[j] + I, left = band[j - 1] + G;
                int32_t hi = imax(imax(top, left), diagonal);
                if (type == ALN_LOCAL) hi = imax(hi, zero);
                prev = band[j];
                band[j] = hi;
            }
            temp[i] = (int16_t)band[SW_FULL_BAND];
            if (type == ALN_LOCAL)
                for (uint32_t j = 1; j <= SW_FULL_BAND; ++j)
                    if (!last || block + j <= N) sink_report(sink, band[j], block + j, i + 1);
        }
        if (type == ALN_SEMI_GLOBAL) {
            for (uint32_t j = 1; j <= SW_FULL_BAND; ++j) if (!last || block + j <= N) sink_report(sink, band[j], block + j, M);
        } else if (type == ALN_GLOBAL && last) {
            for (uint32_t j = 1; j <= SW_FULL_BAND; ++j) if (block + j == N) sink_report(sink, band[j], block + j, M);
        }
    }
    return 1;
}

/* ------------------------------------------------------------------------ */
/* Full-matrix scores, PATTERN blocking (the default algorithm tag):          */
/*   Gotoh  gotoh_inl.h:459-900 (8 pattern symbols per block, short2 column   */
/*          over the text), init :69-93, save_boundary/save_Mth utils_inl.h   */
/*          :206-226,279-299                                                  */
/*   SW/ED  sw_inl.h:417-760 (16 pattern symbols per block, int16 column)     */
/* Same DP as the text-blocking forms; what differs is the visiting order     */
/* (block of pattern symbols -> text position -> symbol in block: it decides   */
/* LOCAL ties), where the int16 column cuts, and the early exit: after each    */
/* non-final block, max_i H[i][block end] + (M - block end) * match < min_score */
/* returns false (sink as reported so far).  M >= 1 (M == 0 reads              */
/* uninitialised cells in the reference).                                     */
/* linear != 0: SW recurrences with scheme {match, mismatch, deletion,        */
/* insertion}; else Gotoh with scheme_t.                                      */
/* ------------------------------------------------------------------------ */
static int score_pattern_blocking(int linear, int type, const scheme_t* sc, const int32_t* sw,
    const uint32_t* pat_w, uint32_t pat_bits, uint32_t pat_be, uint64_t pat_begin, uint32_t M,
    const uint32_t* txt_w, uint32_t txt_bits, uint32_t txt_be, uint64_t txt_begin, uint32_t N,
    int32_t min_score, best_sink_t* sink, int16_t* temp /* 2*N scratch */)
{
    const uint32_t BL = linear ? 16u : 8u;
    int32_t H_band[17], F_band[17];
    uint8_t q_cache[16], qq_cache[16];
    memset(q_cache, 0, sizeof q_cache); memset(qq_cache, 0, sizeof qq_cache);
    const int32_t zero = 0;
    const int32_t G_o = linear ? 0 : sc->pat_gap_open, G_e = linear ? 0 : sc->pat_gap_ext;
    const int32_t V = linear ? sw[0] : 0, X = linear ? sw[1] : 0, G = linear ? sw[2] : 0, I = linear ? sw[3] : 0;
    const int32_t infimum = -32768 - (G_o < G_e ? G_o : G_e);
    const int32_t match255 = linear ? V : sc->match;
    for (uint32_t i = 0; i < N; ++i) {                       /* context.init, PatternBlockingTag branch */
        if (linear) temp[i] = (int16_t)(type == ALN_GLOBAL ? G * (int32_t)(i + 1) : zero);          /* sw_inl.h:76-79 */
        else { temp[2 * i] = (int16_t)(type == ALN_GLOBAL ? sc->txt_gap_open + sc->txt_gap_ext * (int32_t)i : zero);
               temp[2 * i + 1] = (int16_t)(type == ALN_LOCAL ? zero : infimum); }                   /* gotoh_inl.h:82-88 */
    }
    const uint32_t nb = BL * ((M + BL - 1) / BL);
    const uint32_t end_block = nb > BL ? nb : BL;
    for (uint32_t block = 0; block + BL <= end_block; block += BL)
    {
        const int last = (block + BL == end_block);
        const uint32_t block_end = (block + BL < M) ? block + BL : M;
        for (uint32_t t = 0; t < BL; ++t)
            if (!last || block + t < block_end) {
                q_cache[t]  = (uint8_t)ps_get(pat_w, pat_bits, pat_be, pat_begin + block + t);
                qq_cache[t] = (!linear && sc->quals) ? sc->quals[pat_begin + block + t] : 0;
            }
        for (uint32_t j = 0; j <= BL; ++j) {
            if (linear) H_band[j] = (type != ALN_LOCAL) ? I * (int32_t)(block + j) : zero;
            else { H_band[j] = (type != ALN_LOCAL) ? (block + j > 0 ? G_o + G_e * (int32_t)(block + j - 1u) : zero) : zero; F_band[j] = infimum; }
        }
        int32_t max_score = (-2147483647 - 1);               /* Field_traits<int32>::min() */
        int32_t temp_i = H_band[0];
        for (uint32_t i = 0; i < N; ++i)
        {
            const uint8_t r_i = (uint8_t)ps_get(txt_w, txt_bits, txt_be, txt_begin + i);
            int32_t H_diag = temp_i, E = 0;
            if (linear) { H_band[0] = temp_i = temp[i]; }
            else        { H_band[0] = temp_i = temp[2 * i]; E = temp[2 * i + 1]; }
            for (uint32_t j = 1; j <= BL; ++j)
            {
                int32_t hi;
                if (linear) {
                    const int32_t diagonal = H_diag + (r_i == q_cache[j - 1] ? V : X);
                    const int32_t top = H_band[j] + G, left = H_band[j - 1] + I;
                    hi = imax(imax(top, left), diagonal);
                } else {
                    F_band[j] = imax(F_band[j] + G_e, H_band[j] + G_o);
                    E = imax(E + G_e, H_band[j - 1] + G_o);
                    const int32_t diagonal = H_diag + subst(sc, r_i, q_cache[j - 1], qq_cache[j - 1]);
                    hi = imax(imax(E, F_band[j]), diagonal);
                }
                if (type == ALN_LOCAL) hi = imax(hi, zero);
                H_diag = H_band[j];
                H_band[j] = hi;
            }
            if (linear) temp[i] = (int16_t)H_band[BL];
            else { temp[2 * i] = (int16_t)H_band[BL]; temp[2 * i + 1] = (int16_t)E; }
            max_score = imax(max_score, H_band[BL]);
            if (type == ALN_LOCAL) {
                for (uint32_t j = 1; j <= BL; ++j)
                    if (!last || block + j <= M) sink_report(sink, H_band[j], i + 1, block + j);
            } else if (last && type == ALN_SEMI_GLOBAL) {
                if (block + BL >= M) sink_report(sink, H_band[((M - 1) & (BL - 1)) + 1], i + 1, M);     /* save_boundary -> save_Mth */
            }
        }
        if (!last) {
            const int32_t missing_cols = (int32_t)(M - block - BL);
            if ((int64_t)max_score + (int64_t)missing_cols * match255 < (int64_t)min_score) return 0;
        }
    }
    if (type == ALN_GLOBAL)
        sink_report(sink, H_band[((M - 1) & (BL - 1)) + 1], N, M);          /* save_Mth(M, band, N-1, sink): (i+1, M) with i = N-1 */
    return 1;
}

/* kind: 0 Gotoh {match,mismatch,gap_open,gap_ext}, 1 SW / ED {match,mismatch,deletion,insertion} */
ORACLE_API void oracle_batch_score_pattern_blocking(
    int kind, int type, const int32_t* scheme,
    const uint32_t* pat_w, uint32_t pat_bits, uint32_t pat_be, const uint64_t* pat_begin, const uint32_t* pat_len,
    const uint32_t* txt_w, uint32_t txt_bits, uint32_t txt_be, const uint64_t* txt_begin, const uint32_t* txt_len,
    const int32_t* min_score /* nullable */, uint32_t n, int32_t* out_score, uint32_t* out_sink, uint8_t* out_ok, int n_threads)
{
    scheme_t sc = { scheme[0], scheme[1], scheme[2], scheme[3], scheme[2], scheme[3], NULL, NULL };
#if defined(_OPENMP)
    if (n_threads > 0) omp_set_num_threads(n_threads);
    #pragma omp parallel for schedule(dynamic, 16)
#endif
    for (int64_t i = 0; i < (int64_t)n; ++i)
    {
        best_sink_t s; sink_init(&s);
        int16_t* temp = (int16_t*)malloc(sizeof(int16_t) * 2 * (size_t)(txt_len[i] + 1));
        const int ok = score_pattern_blocking(kind, type, &sc, scheme, pat_w, pat_bits, pat_be, pat_begin[i], pat_len[i],
            txt_w, txt_bits, txt_be, txt_begin[i], txt_len[i], min_score ? min_score[i] : (-2147483647 - 1) /* never exits */, &s, temp);
        free(temp);
        out_score[i] = s.score; out_sink[2 * i] = s.sink_x; out_sink[2 * i + 1] = s.sink_y;
        if (out_ok) out_ok[i] = (uint8_t)ok;
    }
}

/* the full-matrix score for nvBowtie's quality-aware scheme: qscheme = {match, pattern_gap_open, pattern_gap_ext,
 * text_gap_open, text_gap_ext}; algorithm 0 = pattern blocking, 1 = text blocking */
ORACLE_API void oracle_batch_gotoh_score_qual(
    int algorithm, int type, const int32_t* qscheme, const int32_t* mm_lut, const uint8_t* quals,
    const uint32_t* pat_w, uint32_t pat_bits, uint32_t pat_be, const uint64_t* pat_begin, const uint32_t* pat_len,
    const uint32_t* txt_w, uint32_t txt_bits, uint32_t txt_be, const uint64_t* txt_begin, const uint32_t* txt_len,
    const int32_t* min_score /* nullable */, uint32_t n, int32_t* out_score, uint32_t* out_sink, uint8_t* out_ok, int n_threads)
{
    const scheme_t sc = { qscheme[0], 0, qscheme[1], qscheme[2], qscheme[3], qscheme[4], mm_lut, quals };
#if defined(_OPENMP)
    if (n_threads > 0) omp_set_num_threads(n_threads);
    #pragma omp parallel for schedule(dynamic, 16)
#endif
    for (int64_t i = 0; i < (int64_t)n; ++i)
    {
        best_sink_t s; sink_init(&s);
        const size_t len = (algorithm ? pat_len[i] : txt_len[i]) + 1;
        int16_t* temp = (int16_t*)malloc(sizeof(int16_t) * 2 * len);
        const int ok = algorithm
            ? gotoh_score_text_blocking(type, &sc, pat_w, pat_bits, pat_be, pat_begin[i], pat_len[i], txt_w, txt_bits, txt_be, txt_begin[i], txt_len[i],
                                        min_score ? min_score[i] : -(1 << 30), &s, temp)
            : score_pattern_blocking(0, type, &sc, NULL, pat_w, pat_bits, pat_be, pat_begin[i], pat_len[i], txt_w, txt_bits, txt_be, txt_begin[i], txt_len[i],
                                     min_score ? min_score[i] : (-2147483647 - 1), &s, temp);
        free(temp);
        out_score[i] = s.score; out_sink[2 * i] = s.sink_x; out_sink[2 * i + 1] = s.sink_y;
        if (out_ok) out_ok[i] = (uint8_t)ok;
    }
}

/* ------------------------------------------------------------------------ */
/* Full-matrix Gotoh traceback                                                */
/*   driver      nvbio/alignment/alignment_inl.h:365-480 (score with           */
/*               checkpoints every CHECKPOINTS pattern symbols, then per       */
/*               checkpoint: recompute the flow submatrix, walk it back)       */
/*   flow flags  gotoh_inl.h:512-560 (update_row's new_cell arguments) and     */
/*               GotohSubmatrixContext::new_cell (:407-425)                    */
/*   walk        gotoh_inl.h:1806-1870                                         */
/* The sink is the one of the pattern-blocking score pass; the flags are        */
/* restated densely (the checkpoints are exact while values fit int16).        */
/* res: [0] score, [1..2] source (x = text, y = pattern), [3..4] sink,          */
/*      [5] ops pushed, [6] clip pushed first (M - sink.y), [7] clip pushed    */
/*      last (source.y);  ops: 0 = M, 1 = I, 2 = D, end of the alignment first */
/* flags: scratch of M*N bytes, hrow / frow: scratch of M+1 int32 each.        */
/* ------------------------------------------------------------------------ */
static void gotoh_traceback_x(int type, const scheme_t* scp,
    const uint32_t* pat_w, uint32_t pat_bits, uint32_t pat_be, uint64_t pat_begin, uint32_t M,
    const uint32_t* txt_w, uint32_t txt_bits, uint32_t txt_be, uint64_t txt_begin, uint32_t N,
    int32_t* res, uint8_t* ops, uint32_t ops_capacity, uint8_t* flags, int32_t* hrow, int32_t* frow)
{
    const scheme_t sc = *scp;
    best_sink_t best; sink_init(&best);
    {
        int16_t* temp = (int16_t*)malloc(sizeof(int16_t) * 2 * (size_t)(N + 1));
        score_pattern_blocking(0, type, &sc, NULL, pat_w, pat_bits, pat_be, pat_begin, M, txt_w, txt_bits, txt_be, txt_begin, N, (-2147483647 - 1), &best, temp);
        free(temp);
    }
    res[0] = best.score; res[3] = (int32_t)best.sink_x; res[4] = (int32_t)best.sink_y; res[5] = res[6] = res[7] = 0;
    if (best.sink_x == 0xFFFFFFFFu || best.sink_y == 0xFFFFFFFFu) { res[1] = res[2] = -1; return; }
    /* dense flow flags: cell (i, j) = text row i, pattern column j */
    const int32_t G_o = sc.pat_gap_open, G_e = sc.pat_gap_ext;
    const int32_t infimum = -32768 - (G_o < G_e ? G_o : G_e);
    for (uint32_t j = 0; j <= M; ++j) {                      /* the row above the text: H_band init (:693-697), F = infimum */
        hrow[j] = (type != ALN_LOCAL) ? (j > 0 ? G_o + G_e * (int32_t)(j - 1u) : 0) : 0;
        frow[j] = infimum;
    }
    for (uint32_t i = 0; i < N; ++i)
    {
        const uint8_t r_i = (uint8_t)ps_get(txt_w, txt_bits, txt_be, txt_begin + i);
        int32_t H_diag = hrow[0];
        hrow[0] = (type == ALN_GLOBAL) ? sc.txt_gap_open + sc.txt_gap_ext * (int32_t)i : 0;       /* context.init (:275-279) */
        int32_t E = (type == ALN_LOCAL) ? 0 : infimum;
        for (uint32_t j = 1; j <= M; ++j)
        {
            const int32_t ftop = frow[j] + G_e, htop = hrow[j] + G_o;
            frow[j] = imax(ftop, htop);
            const uint8_t fdir = ftop > htop ? DIR_DELETION_EXT : DIR_SUBSTITUTION;
            const int32_t eleft = E + G_e, hleft = hrow[j - 1] + G_o;
            E = imax(eleft, hleft);
            const uint8_t edir = eleft > hleft ? DIR_INSERTION_EXT : DIR_SUBSTITUTION;
            const uint8_t q_j = (uint8_t)ps_get(pat_w, pat_bits, pat_be, pat_begin + j - 1);
            const int32_t diagonal = H_diag + subst(&sc, r_i, q_j, sc.quals ? sc.quals[pat_begin + j - 1] : 0);
            const int32_t top = frow[j], left = E;
            int32_t hi = imax(imax(left, top), diagonal);
            if (type == ALN_LOCAL) hi = imax(hi, 0);
            uint8_t hdir = top > left ? (top > diagonal ? DIR_DELETION : DIR_SUBSTITUTION) : (left > diagonal ? DIR_INSERTION : DIR_SUBSTITUTION);
            if (type == ALN_LOCAL && hi == 0) hdir = DIR_SINK;
            H_diag = hrow[j];
            hrow[j] = hi;
            flags[(size_t)i * M + (j - 1)] = (uint8_t)(hdir | edir | fdir);
        }
    }
    /* walk back (:1806-1870 across all checkpoints, then alignment_inl.h:443-466) */
    uint32_t n = 0;
    res[6] = (int32_t)(M - best.sink_y);
    int32_t row = (int32_t)best.sink_x, col = (int32_t)best.sink_y - 1;
    uint8_t state = 0;     /* HSTATE */
    int found = 0;
    #define PUSH(o) do { if (n < ops_capacity) ops[n] = (o); ++n; } while (0)
    while (row > 0 && col >= 0)
    {
        const uint8_t op = flags[(size_t)(row - 1) * M + col], h_op = op & 3u;
        if (type == ALN_LOCAL && state == 0 && h_op == DIR_SINK) { found = 1; break; }
        if (state == 1)      { if ((op & DIR_INSERTION_EXT) == 0u) state = 0; --col; PUSH(DIR_INSERTION); }
        else if (state == 2) { if ((op & DIR_DELETION_EXT)  == 0u) state = 0; --row; PUSH(DIR_DELETION); }
        else if (h_op == DIR_INSERTION) state = 1;
        else if (h_op == DIR_DELETION)  state = 2;
        else { --col; --row; PUSH(DIR_SUBSTITUTION); }
    }
    (void)found;
    uint32_t sx = (uint32_t)row, sy = (uint32_t)(col + 1);
    if (type == ALN_SEMI_GLOBAL || type == ALN_GLOBAL) { if (sx == 0) for (; sy > 0; --sy) PUSH(DIR_INSERTION); }
    if (type == ALN_GLOBAL)                            { if (sy == 0) for (; sx > 0; --sx) PUSH(DIR_DELETION); }
    #undef PUSH
    res[1] = (int32_t)sx; res[2] = (int32_t)sy; res[5] = (int32_t)n; res[7] = (int32_t)sy;
}

ORACLE_API void oracle_gotoh_traceback(int type, const int32_t* scheme,
    const uint32_t* pat_w, uint32_t pat_bits, uint32_t pat_be, uint64_t pat_begin, uint32_t M,
    const uint32_t* txt_w, uint32_t txt_bits, uint32_t txt_be, uint64_t txt_begin, uint32_t N,
    int32_t* res, uint8_t* ops, uint32_t ops_capacity, uint8_t* flags, int32_t* hrow, int32_t* frow)
{
    const scheme_t sc = { scheme[0], scheme[1], scheme[2], scheme[3], scheme[2], scheme[3], NULL, NULL };
    gotoh_traceback_x(type, &sc, pat_w, pat_bits, pat_be, pat_begin, M, txt_w, txt_bits, txt_be, txt_begin, N, res, ops, ops_capacity, flags, hrow, frow);
}
/* quality-aware scheme: qscheme = {match, pattern_gap_open, pattern_gap_ext, text_gap_open, text_gap_ext} */
ORACLE_API void oracle_gotoh_traceback_qual(int type, const int32_t* qscheme, const int32_t* mm_lut, const uint8_t* quals,
    const uint32_t* pat_w, uint32_t pat_bits, uint32_t pat_be, uint64_t pat_begin, uint32_t M,
    const uint32_t* txt_w, uint32_t txt_bits, uint32_t txt_be, uint64_t txt_begin, uint32_t N,
    int32_t* res, uint8_t* ops, uint32_t ops_capacity, uint8_t* flags, int32_t* hrow, int32_t* frow)
{
    const scheme_t sc = { qscheme[0], 0, qscheme[1], qscheme[2], qscheme[3], qscheme[4], mm_lut, quals };
    gotoh_traceback_x(type, &sc, pat_w, pat_bits, pat_be, pat_begin, M, txt_w, txt_bits, txt_be, txt_begin, N, res, ops, ops_capacity, flags, hrow, frow);
}

/* ------------------------------------------------------------------------ */
/* Smith-Waterman / edit-distance tracebacks (linear gaps)                    */
/*   banded  flags: sw_banded_inl.h:405-470 (new_cell's dir) stored as is by  */
/*           SmithWatermanSubmatrixContext (:269-279) -- NOTE no SINK marking */
/*           even for LOCAL, so its walk (:748-800) only ends at row zero;    */
/*           driver banded_inl.h:352-423                                      */
/*   full    flags: sw_inl.h:475-500 with `score ? dir : SINK` for LOCAL      */
/*           (:389-396); walk :1660-1700; driver alignment_inl.h:365-480;     */
/*           sink = the 16-column pattern-blocking score pass                 */
/* scheme = {match, mismatch, deletion, insertion}; res / ops as the Gotoh    */
/* tracebacks; flags: M*band (banded) or M*N (full) bytes; row: M+1 int32.     */
/* ------------------------------------------------------------------------ */
ORACLE_API void oracle_sw_traceback(uint32_t band /* 0 = full matrix */, int type, const int32_t* sw,
    const uint32_t* pat_w, uint32_t pat_bits, uint32_t pat_be, uint64_t pat_begin, uint32_t M,
    const uint32_t* txt_w, uint32_t txt_bits, uint32_t txt_be, uint64_t txt_begin, uint32_t N,
    int32_t* res, uint8_t* ops, uint32_t ops_capacity, uint8_t* flags, int32_t* row)
{
    const int32_t V = sw[0], X = sw[1], G = sw[2], I = sw[3];
    best_sink_t best; sink_init(&best);
    uint32_t n = 0;
    #define PUSH(o) do { if (n < ops_capacity) ops[n] = (o); ++n; } while (0)
    res[5] = res[6] = res[7] = 0;
    if (band)
    {
        /* forward pass = banded_sw_score_x with the direction of every cell kept */
        if (N >= M)
        {
            uint32_t text_cache[MAX_BAND]; int32_t B[MAX_BAND];
            for (uint32_t j = 0; j + 1 < band; ++j) text_cache[j] = cache_store(band, ps_get(txt_w, txt_bits, txt_be, txt_begin + j));
            for (uint32_t j = 0; j < band; ++j) B[j] = (type == ALN_GLOBAL) ? (int32_t)j * G : 0;
            for (uint32_t i = 0; i < M; ++i)
            {
                const uint8_t q = (uint8_t)ps_get(pat_w, pat_bits, pat_be, pat_begin + i);
                uint8_t* fr = flags + (size_t)i * band;
                {
                    const int32_t diagonal = B[0] + ((uint8_t)text_cache[0] == q ? V : X), top = B[1] + G;
                    int32_t hi = imax(top, diagonal);
                    if (type == ALN_LOCAL) { hi = imax(hi, 0); sink_report(&best, hi, i + 1, i + 1); }
                    B[0] = hi; fr[0] = top > diagonal ? DIR_INSERTION : DIR_SUBSTITUTION;
                }
                for (uint32_t j = 1; j + 1 < band; ++j)
                {
                    const uint32_t g = text_cache[j]; text_cache[j - 1] = g;
                    const int32_t diagonal = B[j] + ((uint8_t)g == q ? V : X), top = B[j + 1] + G, left = B[j - 1] + I;
                    int32_t hi = imax(imax(top, left), diagonal);
                    if (type == ALN_LOCAL) { hi = imax(hi, 0); sink_report(&best, hi, i + j + 1, i + 1); }
                    B[j] = hi;
                    fr[j] = top > left ? (top > diagonal ? DIR_INSERTION : DIR_SUBSTITUTION) : (left > diagonal ? DIR_DELETION : DIR_SUBSTITUTION);
                }
                const uint8_t g = (i + band - 1 < N) ? (uint8_t)ps_get(txt_w, txt_bits, txt_be, txt_begin + i + band - 1) : 255u;
                text_cache[band - 2] = cache_store(band, g);
                {
                    const int32_t diagonal = B[band - 1] + (g == q ? V : X), left = B[band - 2] + I;
                    int32_t hi = imax(left, diagonal);
                    if (type == ALN_LOCAL) { hi = imax(hi, 0); sink_report(&best, hi, i + band, i + 1); }
                    B[band - 1] = hi; fr[band - 1] = left > diagonal ? DIR_DELETION : DIR_SUBSTITUTION;
                }
            }
            if (type == ALN_GLOBAL) sink_report(&best, B[band - 1], M + band - 1, M);
            else if (type == ALN_SEMI_GLOBAL) {
                const uint32_t a = M + band - 1u, m = (a < N ? a : N) - (M - 1u);
                sink_report(&best, B[0], M, M);
                for (uint32_t j = 1; j < band; ++j) if (j < m) sink_report(&best, B[j], M + j, M);
            }
        }
        res[0] = best.score; res[3] = (int32_t)best.sink_x; res[4] = (int32_t)best.sink_y;
        if (best.sink_x == 0xFFFFFFFFu || best.sink_y == 0xFFFFFFFFu) { res[1] = res[2] = -1; return; }
        res[6] = (int32_t)(M - best.sink_y);
        int32_t entry = (int32_t)(best.sink_x - best.sink_y), r = (int32_t)best.sink_y - 1;
        while (r >= 0)
        {
            const uint8_t op = flags[(size_t)r * band + entry];
            /* (TYPE == LOCAL && op == SINK) can never hold: the context stores `dir` only */
            if (op == DIR_DELETION)       { --entry; PUSH(DIR_DELETION); }
            else if (op == DIR_INSERTION) { ++entry; --r; PUSH(DIR_INSERTION); }
            else                          { --r; PUSH(DIR_SUBSTITUTION); }
        }
        res[1] = entry; res[2] = 0; res[5] = (int32_t)n; res[7] = 0;
        #undef PUSH
        return;
    }
    /* full matrix */
    {
        int16_t* temp = (int16_t*)malloc(sizeof(int16_t) * 2 * (size_t)(N + 1));
        const scheme_t dummy = { 0, 0, 0, 0, 0, 0, NULL, NULL };
        score_pattern_blocking(1, type, &dummy, sw, pat_w, pat_bits, pat_be, pat_begin, M, txt_w, txt_bits, txt_be, txt_begin, N, (-2147483647 - 1), &best, temp);
        free(temp);
    }
    res[0] = best.score; res[3] = (int32_t)best.sink_x; res[4] = (int32_t)best.sink_y;
    if (best.sink_x == 0xFFFFFFFFu || best.sink_y == 0xFFFFFFFFu) { res[1] = res[2] = -1; return; }
    for (uint32_t j = 0; j <= M; ++j) row[j] = (type != ALN_LOCAL) ? I * (int32_t)j : 0;          /* band init: sw_inl.h:648-649 */
    for (uint32_t i = 0; i < N; ++i)
    {
        const uint8_t r_i = (uint8_t)ps_get(txt_w, txt_bits, txt_be, txt_begin + i);
        int32_t prev = row[0];
        row[0] = (type == ALN_GLOBAL) ? G * (int32_t)(i + 1) : 0;                                  /* context.init: :76-79 */
        for (uint32_t j = 1; j <= M; ++j)
        {
            const uint8_t q_j = (uint8_t)ps_get(pat_w, pat_bits, pat_be, pat_begin + j - 1);
            const int32_t diagonal = prev + (r_i == q_j ? V : X), top = row[j] + G, left = row[j - 1] + I;
            int32_t hi = imax(imax(top, left), diagonal);
            if (type == ALN_LOCAL) hi = imax(hi, 0);
            uint8_t dir = top > left ? (top > diagonal ? DIR_DELETION : DIR_SUBSTITUTION) : (left > diagonal ? DIR_INSERTION : DIR_SUBSTITUTION);
            if (type == ALN_LOCAL && hi == 0) dir = DIR_SINK;
            prev = row[j]; row[j] = hi;
            flags[(size_t)i * M + (j - 1)] = dir;
        }
    }
    #define PUSH(o) do { if (n < ops_capacity) ops[n] = (o); ++n; } while (0)
    res[6] = (int32_t)(M - best.sink_y);
    int32_t r = (int32_t)best.sink_x, c = (int32_t)best.sink_y - 1;
    while (r > 0 && c >= 0)
    {
        const uint8_t op = flags[(size_t)(r - 1) * M + c];
        if (type == ALN_LOCAL && op == DIR_SINK) break;
        if (op != DIR_DELETION)  --c;
        if (op != DIR_INSERTION) --r;
        PUSH(op);
    }
    uint32_t sx = (uint32_t)r, sy = (uint32_t)(c + 1);
    if (type == ALN_SEMI_GLOBAL || type == ALN_GLOBAL) { if (sx == 0) for (; sy > 0; --sy) PUSH(DIR_INSERTION); }
    if (type == ALN_GLOBAL)                            { if (sy == 0) for (; sx > 0; --sx) PUSH(DIR_DELETION); }
    #undef PUSH
    res[1] = (int32_t)sx; res[2] = (int32_t)sy; res[5] = (int32_t)n; res[7] = (int32_t)sy;
}

/* banded: BatchedBandedAlignmentScore over SmithWatermanAligner / EditDistanceAligner;
 * full (band == 0): BatchedAlignmentScore over the TextBlockingTag forms */
ORACLE_API void oracle_batch_sw_score(
    uint32_t band /* 0 = full matrix */, int type, const int32_t* scheme,
    const uint32_t* pat_w, uint32_t pat_bits, uint32_t pat_be, const uint64_t* pat_begin, const uint32_t* pat_len,
    const uint32_t* txt_w, uint32_t txt_bits, uint32_t txt_be, const uint64_t* txt_begin, const uint32_t* txt_len,
    uint32_t n, int32_t* out_score, uint32_t* out_sink, int n_threads)
{
#if defined(_OPENMP)
    if (n_threads > 0) omp_set_num_threads(n_threads);
    #pragma omp parallel for schedule(dynamic, 16)
#endif
    for (int64_t i = 0; i < (int64_t)n; ++i)
    {
        best_sink_t s; sink_init(&s);
        if (band) banded_sw_score_x(band, type, scheme, pat_w, pat_bits, pat_be, pat_begin[i], pat_len[i], txt_w, txt_bits, txt_be, txt_begin[i], txt_len[i], &s);
        else {
            int16_t* temp = (int16_t*)malloc(sizeof(int16_t) * (size_t)(pat_len[i] + 1));
            sw_score_text_blocking(type, scheme, pat_w, pat_bits, pat_be, pat_begin[i], pat_len[i], txt_w, txt_bits, txt_be, txt_begin[i], txt_len[i], &s, temp);
            free(temp);
        }
        out_score[i] = s.score; out_sink[2 * i] = s.sink_x; out_sink[2 * i + 1] = s.sink_y;
    }
}

/* ------------------------------------------------------------------------ */
/* Banded bit-vector edit distance: EditDistanceAligner<TYPE, MyersTag<A>>      */
/* banded_myers<BAND_WIDTH, C = 0, TYPE, ALPHABET_SIZE> (alignment/myers/myers_banded_inl.h:236-291, reached through */
/* banded_alignment_score :306-329), with MyersBitVectors<A> :44-196, diagonal_column :198-212, horizontal_column :214-228. */
/* The threshold parameter is an int16 (:243): the caller's int32 is narrowed on the way in, and the sink's score type */
/* narrows the reported distance (BestSink<int16> in examples/fmmap/fmmap.cu:306; sink_bits says which).  A = 5 leaves the */
/* fifth vector uninitialised in the reference (:160-165 clears four); texts of the callers here hold symbols 0..3, so it is */
/* never read -- zero here.                                                                                                  */
/* ------------------------------------------------------------------------ */
static inline uint32_t myers_slot(uint32_t A, uint32_t c) { return A == 4 ? (c & 3u) : A == 2 ? (c & 1u) : c; }
static int banded_myers_x(uint32_t BW, int type, uint32_t A,
    const uint32_t* pat_w, uint32_t pat_bits, uint32_t pat_be, uint64_t pat_begin, uint32_t pattern_len,
    const uint32_t* txt_w, uint32_t txt_bits, uint32_t txt_be, uint64_t txt_begin, uint32_t text_len,
    int32_t min_score32, best_sink_t* sink)
{
    const int16_t min_score = (int16_t)min_score32;                        /* :243 */
    if (text_len < pattern_len) return 0;                                  /* :250-251 */
    uint32_t B[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    uint32_t VP = 0xFFFFFFFFu, VN = 0u;                                    /* :255-256 */
    int dist = 0;                                                          /* :258, C = 0 */
    const uint32_t last = (text_len - 1u < pattern_len) ? text_len - 1u : pattern_len;     /* :266 */
    for (uint32_t i = 0; i < last; ++i)                                    /* phase 1 (diagonal) :267-274 */
    {
        for (uint32_t c = 0; c < A; ++c) B[c] >>= 1;
        B[myers_slot(A, ps_get(pat_w, pat_bits, pat_be, pat_begin + i))] |= 1u << (BW - 1);
        const uint32_t Bc = B[myers_slot(A, ps_get(txt_w, txt_bits, txt_be, txt_begin + i))];
        uint32_t X = Bc | VN;                                              /* diagonal_column :200-211 */
        const uint32_t D0 = ((VP + (X & VP)) ^ VP) | X;
        const uint32_t HN = VP & D0;
        const uint32_t HP = VN | ~(VP | D0);
        X = D0 >> 1; VN = X & HP; VP = HN | ~(X | HP);
        dist -= 1 - (int)((D0 >> (BW - 1)) & 1u);
    }
    int s = (int)(BW - 1u + pattern_len - last);                           /* :277, C = 0 */
    for (uint32_t i = last; i < text_len && s >= 0; ++i)                   /* phase 2 (horizontal) :278-289 */
    {
        for (uint32_t c = 0; c < A; ++c) B[c] >>= 1;
        const uint32_t Bc = B[myers_slot(A, ps_get(txt_w, txt_bits, txt_be, txt_begin + i))];
        uint32_t X = Bc | VN;                                              /* horizontal_column :216-227 */
        const uint32_t D0 = ((VP + (X & VP)) ^ VP) | X;
        const uint32_t HN = VP & D0;
        const uint32_t HP = VN | ~(VP | D0);
        X = D0 >> 1; VN = X & HP; VP = HN | ~(X | HP);
        dist -= (int)((HP >> s) & 1u) - (int)((HN >> s) & 1u);
        if (type == ALN_SEMI_GLOBAL && dist >= min_score) sink_report(sink, dist, i + 1, pattern_len);
        --s;
    }
    if (type == ALN_GLOBAL && dist >= min_score) sink_report(sink, dist, text_len, pattern_len);
    return 1;
}
/* BatchedBandedAlignmentScore over the bit-vector edit-distance aligner; sink_bits = 16 narrows the sink as BestSink<int16> does
 * (initial score -32768, sink_inl.h:38-46 over Field_traits<int16>::min()) */
ORACLE_API void oracle_batch_banded_myers_score(
    uint32_t band, int type, uint32_t alphabet, int32_t min_score, uint32_t sink_bits,
    const uint32_t* pat_w, uint32_t pat_bits, uint32_t pat_be, const uint64_t* pat_begin, const uint32_t* pat_len,
    const uint32_t* txt_w, uint32_t txt_bits, uint32_t txt_be, const uint64_t* txt_begin, const uint32_t* txt_len,
    uint32_t n, int32_t* out_score, uint32_t* out_sink, int n_threads)
{
#if defined(_OPENMP)
    if (n_threads > 0) omp_set_num_threads(n_threads);
    #pragma omp parallel for schedule(dynamic, 16)
#endif
    for (int64_t i = 0; i < (int64_t)n; ++i)
    {
        best_sink_t s; sink_init(&s);
        if (sink_bits == 16) s.score = -32768;
        banded_myers_x(band, type, alphabet, pat_w, pat_bits, pat_be, pat_begin[i], pat_len[i], txt_w, txt_bits, txt_be, txt_begin[i], txt_len[i], min_score, &s);
        out_score[i] = s.score; out_sink[2 * i] = s.sink_x; out_sink[2 * i + 1] = s.sink_y;
    }
}

/* ref_sw for the Gotoh aligner: nvbio-test/alignment_test_utils.h:536-624, the independent
 * full-matrix checker the reference's alignment test compares alignment_score() with
 * (alignment_test.cu:247-265).  i runs over the text, j over the pattern. */
ORACLE_API int32_t oracle_ref_sw_gotoh(int type, const int32_t* scheme, const uint8_t* str, uint32_t M, const uint8_t* ref, uint32_t N)
{
    const int32_t V = scheme[0], S = scheme[1], G_o = scheme[2], G_e = scheme[3];
    const size_t W = (size_t)M + 1;
    int32_t* H = (int32_t*)malloc(sizeof(int32_t) * 3 * W * ((size_t)N + 1));
    int32_t* E = H + W * ((size_t)N + 1);
    int32_t* F = E + W * ((size_t)N + 1);
    const int32_t ninf = (type != ALN_LOCAL) ? -100000 : 0;
    H[0] = 0; E[0] = F[0] = ninf;
    for (uint32_t j = 1; j <= M; ++j) { H[j] = (type != ALN_LOCAL) ? G_o + G_e * (int32_t)(j - 1) : 0; E[j] = F[j] = ninf; }
    for (uint32_t i = 1; i <= N; ++i) { H[i * W] = (type == ALN_GLOBAL) ? G_o + G_e * (int32_t)(i - 1) : 0; E[i * W] = F[i * W] = ninf; }
    int32_t best = -(1 << 30);
    for (uint32_t i = 1; i <= N; ++i) {
        for (uint32_t j = 1; j <= M; ++j) {
            const int32_t S_ij = (ref[i - 1] == str[j - 1]) ? V : S;
            E[i * W + j] = imax(E[i * W + j - 1] + G_e, H[i * W + j - 1] + G_o);
            F[i * W + j] = imax(F[(i - 1) * W + j] + G_e, H[(i - 1) * W + j] + G_o);
            int32_t h = imax(imax(H[(i - 1) * W + j - 1] + S_ij, E[i * W + j]), F[i * W + j]);
            if (type == ALN_LOCAL) { h = imax(h, 0); if (best < h) best = h; }
            H[i * W + j] = h;
        }
        if (type == ALN_SEMI_GLOBAL && best < H[i * W + M]) best = H[i * W + M];
    }
    const int32_t r = (type == ALN_GLOBAL) ? H[(size_t)N * W + M] : best;
    free(H);
    return r;
}

/* ------------------------------------------------------------------------ */
/* Banded Gotoh traceback                                                     */
/*   driver            nvbio/alignment/banded_inl.h:352-489                   */
/*   flow flags        gotoh_banded_inl.h:479-614 (the dir/edir/fdir handed to */
/*                     new_cell) and :323-336 (GotohSubmatrixContext)         */
/*   backtracking      gotoh_banded_inl.h:878-960                             */
/* The reference recomputes the flags window by window from short2 checkpoints */
/* (:205-262); whenever the DP values fit int16 that is the same as one dense  */
/* forward pass, which is what is restated here.                              */
/* ------------------------------------------------------------------------ */

/* forward pass of gotoh_alignment_score_dispatch::run recording new_cell()'s cdir per cell; flags: M x band bytes */
static int banded_gotoh_flow(uint32_t band, int type, const scheme_t* sc,
    const uint32_t* pat_w, uint32_t pat_bits, uint32_t pat_be, uint64_t pat_begin, uint32_t pattern_len,
    const uint32_t* txt_w, uint32_t txt_bits, uint32_t txt_be, uint64_t txt_begin, uint32_t text_len,
    best_sink_t* sink, uint8_t* flags)
{
    if (text_len < pattern_len) return 0;
    uint32_t text_cache[MAX_BAND];
    int32_t  H_band[MAX_BAND], F_band[MAX_BAND];
    for (uint32_t j = 0; j + 1 < band; ++j)
        text_cache[j] = cache_store(band, ps_get(txt_w, txt_bits, txt_be, txt_begin + j));
    const int32_t G_o = sc->pat_gap_open, G_e = sc->pat_gap_ext;
    const int32_t infimum = -32768 - imax(imax(G_o, G_e), imax(sc->txt_gap_open, sc->txt_gap_ext));
    H_band[0] = 0;
    for (uint32_t j = 1; j < band; ++j) H_band[j] = (type == ALN_GLOBAL) ? sc->txt_gap_open + (int32_t)(j - 1) * sc->txt_gap_ext : 0;
    for (uint32_t j = 0; j < band; ++j) F_band[j] = infimum;
    for (uint32_t i = 0; i < pattern_len; ++i)
    {
        const uint8_t q  = (uint8_t)ps_get(pat_w, pat_bits, pat_be, pat_begin + i);
        const uint8_t qq = sc->quals ? sc->quals[pat_begin + i] : 0;
        uint8_t* row = flags + (size_t)i * band;
        uint8_t edir = DIR_SUBSTITUTION;
        {   /* j == 0 (:483-515) */
            const int32_t ftop = F_band[1] + G_e, htop = H_band[1] + G_o;
            F_band[0] = imax(ftop, htop);
            const uint8_t fdir = ftop > htop ? DIR_DELETION_EXT : DIR_SUBSTITUTION;
            const int32_t diagonal = H_band[0] + subst(sc, (uint8_t)text_cache[0], q, qq);
            const int32_t top = F_band[0];
            int32_t hi = imax(top, diagonal);
            uint8_t hdir = (top > diagonal) ? DIR_INSERTION : DIR_SUBSTITUTION;
            if (type == ALN_LOCAL) { hi = imax(hi, 0); if (hi == 0) hdir = DIR_SINK; sink_report(sink, hi, i + 1, i + 1); }
            H_band[0] = hi;
            row[0] = (uint8_t)(hdir | DIR_SUBSTITUTION | fdir);
        }
        int32_t E_j = H_band[0] + G_o;
        for (uint32_t j = 1; j + 1 < band; ++j)
        {
            const int32_t ftop = F_band[j + 1] + G_e, htop = H_band[j + 1] + G_o;
            F_band[j] = imax(ftop, htop);
            const uint8_t fdir = ftop > htop ? DIR_DELETION_EXT : DIR_SUBSTITUTION;
            const uint32_t g = text_cache[j]; text_cache[j - 1] = g;
            const int32_t diagonal = H_band[j] + subst(sc, (uint8_t)g, q, qq);
            const int32_t top = F_band[j], left = E_j;
            int32_t hi = imax(imax(top, left), diagonal);
            uint8_t hdir = top > left ? (top > diagonal ? DIR_INSERTION : DIR_SUBSTITUTION) : (left > diagonal ? DIR_DELETION : DIR_SUBSTITUTION);
            if (type == ALN_LOCAL) { hi = imax(hi, 0); if (hi == 0) hdir = DIR_SINK; sink_report(sink, hi, i + j + 1, i + 1); }
            H_band[j] = hi;
            row[j] = (uint8_t)(hdir | edir | fdir);
            const int32_t eleft = E_j + G_e, ediagonal = hi + G_o;
            edir = (eleft > ediagonal) ? DIR_INSERTION_EXT : DIR_SUBSTITUTION;
            E_j = imax(ediagonal, eleft);
        }
        const uint8_t g = (i + band - 1 < text_len) ? (uint8_t)ps_get(txt_w, txt_bits, txt_be, txt_begin + i + band - 1) : 255u;
        text_cache[band - 2] = cache_store(band, g);
        {   /* j == BAND_LEN-1 (:584-614) */
            F_band[band - 1] = infimum;
            const int32_t diagonal = H_band[band - 1] + subst(sc, g, q, qq);
            const int32_t left = E_j;
            int32_t hi = imax(left, diagonal);
            uint8_t hdir = (left > diagonal) ? DIR_DELETION : DIR_SUBSTITUTION;
            if (type == ALN_LOCAL) { hi = imax(hi, 0); if (hi == 0) hdir = DIR_SINK; sink_report(sink, hi, i + band, i + 1); }
            H_band[band - 1] = hi;
            row[band - 1] = (uint8_t)(hdir | edir | DIR_SUBSTITUTION);
        }
    }
    if (type == ALN_GLOBAL)
        sink_report(sink, H_band[band - 1], pattern_len + band - 1, pattern_len);
    else if (type == ALN_SEMI_GLOBAL) {
        const uint32_t a = pattern_len + band - 1u;
        const uint32_t m = (a < text_len ? a : text_len) - (pattern_len - 1u);
        sink_report(sink, H_band[0], pattern_len, pattern_len);
        for (uint32_t j = 1; j < band; ++j) if (j < m) sink_report(sink, H_band[j], pattern_len + j, pattern_len);
    }
    return 1;
}

/* banded_alignment_traceback (banded_inl.h:352-423 + gotoh_banded_inl.h:878-960).
 * out: res[0] = score, res[1..2] = source (x,y), res[3..4] = sink (x,y), res[5] = number of ops,
 *      res[6] = clip pushed first (pattern_len - sink.y), res[7] = clip pushed last (source.y);
 * ops[] = the backtracer's pushes in order (end of the alignment first): 0 = M, 1 = I, 2 = D.
 * flags: scratch of pattern_len * band bytes. */
static void banded_gotoh_traceback_x(
    uint32_t band, int type, const scheme_t* scp,
    const uint32_t* pat_w, uint32_t pat_bits, uint32_t pat_be, uint64_t pat_begin, uint32_t pat_len,
    const uint32_t* txt_w, uint32_t txt_bits, uint32_t txt_be, uint64_t txt_begin, uint32_t txt_len,
    int32_t* res, uint8_t* ops, uint32_t ops_capacity, uint8_t* flags)
{
    const scheme_t sc = *scp;
    best_sink_t best; sink_init(&best);
    banded_gotoh_flow(band, type, &sc, pat_w, pat_bits, pat_be, pat_begin, pat_len, txt_w, txt_bits, txt_be, txt_begin, txt_len, &best, flags);
    res[0] = best.score; res[3] = (int32_t)best.sink_x; res[4] = (int32_t)best.sink_y; res[5] = 0; res[6] = res[7] = 0;
    if (best.sink_x == 0xFFFFFFFFu || best.sink_y == 0xFFFFFFFFu) { res[1] = res[2] = -1; return; }      /* :383-385 */
    uint32_t n = 0;
    res[6] = (int32_t)(pat_len - best.sink_y);                                                          /* clip :387 */
    int32_t entry = (int32_t)(best.sink_x - best.sink_y), row = (int32_t)best.sink_y - 1;
    uint8_t state = 0;   /* HSTATE */
    uint32_t sx, sy;
    int stopped = 0;
    while (row >= 0)
    {
        const uint8_t op = flags[(size_t)row * band + entry], h_op = op & 3u;
        if (type == ALN_LOCAL && state == 0 && h_op == DIR_SINK) { sy = (uint32_t)row + 1u; sx = (uint32_t)entry + sy; stopped = 1; break; }
        if (state == 1) {            /* ESTATE */
            if ((op & DIR_INSERTION_EXT) == 0u) state = 0;
            --entry; if (n < ops_capacity) ops[n] = DIR_DELETION; ++n;
        } else if (state == 2) {     /* FSTATE */
            if ((op & DIR_DELETION_EXT) == 0u) state = 0;
            ++entry; --row; if (n < ops_capacity) ops[n] = DIR_INSERTION; ++n;
        } else {
            if (h_op == DIR_DELETION) state = 1;
            else if (h_op == DIR_INSERTION) state = 2;
            else { --row; if (n < ops_capacity) ops[n] = DIR_SUBSTITUTION; ++n; }
        }
    }
    if (!stopped) { sy = 0; sx = (uint32_t)entry; }
    res[1] = (int32_t)sx; res[2] = (int32_t)sy; res[5] = (int32_t)n; res[7] = (int32_t)sy;              /* clip :418 */
}

ORACLE_API void oracle_banded_gotoh_traceback(
    uint32_t band, int type, const int32_t* scheme,
    const uint32_t* pat_w, uint32_t pat_bits, uint32_t pat_be, uint64_t pat_begin, uint32_t pat_len,
    const uint32_t* txt_w, uint32_t txt_bits, uint32_t txt_be, uint64_t txt_begin, uint32_t txt_len,
    int32_t* res, uint8_t* ops, uint32_t ops_capacity, uint8_t* flags)
{
    const scheme_t sc = { scheme[0], scheme[1], scheme[2], scheme[3], scheme[2], scheme[3], NULL, NULL };
    banded_gotoh_traceback_x(band, type, &sc, pat_w, pat_bits, pat_be, pat_begin, pat_len, txt_w, txt_bits, txt_be, txt_begin, txt_len, res, ops, ops_capacity, flags);
}
/* quality-aware scheme: qscheme = {match, pattern_gap_open, pattern_gap_ext, text_gap_open, text_gap_ext} */
ORACLE_API void oracle_banded_gotoh_traceback_qual(
    uint32_t band, int type, const int32_t* qscheme, const int32_t* mm_lut, const uint8_t* quals,
    const uint32_t* pat_w, uint32_t pat_bits, uint32_t pat_be, uint64_t pat_begin, uint32_t pat_len,
    const uint32_t* txt_w, uint32_t txt_bits, uint32_t txt_be, uint64_t txt_begin, uint32_t txt_len,
    int32_t* res, uint8_t* ops, uint32_t ops_capacity, uint8_t* flags)
{
    const scheme_t sc = { qscheme[0], 0, qscheme[1], qscheme[2], qscheme[3], qscheme[4], mm_lut, quals };
    banded_gotoh_traceback_x(band, type, &sc, pat_w, pat_bits, pat_be, pat_begin, pat_len, txt_w, txt_bits, txt_be, txt_begin, txt_len, res, ops, ops_capacity, flags);
}

/* ref_banded_sw: nvbio-test/alignment_test_utils.h:314-460.  The reference's
 * test-suite asserts  banded_alignment_score(...) == ref_banded_sw(...)
 * (alignment_test.cu:310-326); restated here so our tests can make the same
 * assertion.  Strings are plain uint8 arrays as in the test; needs
 * N >= M + BAND - 1 (it reads text[i+BAND-1] unchecked). Returns the score only. */
ORACLE_API int32_t oracle_ref_banded_sw(
    uint32_t band, int type, const int32_t* scheme,
    const uint8_t* pattern, uint32_t M, const uint8_t* text, uint32_t pos)
{
    const int32_t V = scheme[0], S = scheme[1], G_o = scheme[2], G_e = scheme[3];
    int32_t best_score = -(1 << 30);
    const int32_t infimum = -(1 << 30) - G_e;
    int32_t H_band[MAX_BAND], F_band[MAX_BAND];

    H_band[0] = 0;
    for (uint32_t j = 1; j < band; ++j)
        H_band[j] = (type == ALN_GLOBAL) ? scheme[2] + (int32_t)(j - 1) * scheme[3] : 0;
    for (uint32_t j = 0; j < band; ++j) F_band[j] = infimum;

    for (uint32_t i = 0; i < M; ++i)
    {
        const uint8_t q = pattern[i];
        for (uint32_t j = 0; j + 1 < band; ++j)
            F_band[j] = imax(F_band[j + 1] + G_e, H_band[j + 1] + G_o);
        F_band[band - 1] = infimum;
        {
            const int32_t S_ij = (text[pos + i] == q) ? V : S;
            int32_t hi = imax(F_band[0], H_band[0] + S_ij);
            if (type == ALN_LOCAL) { hi = imax(hi, 0); best_score = imax(best_score, hi); }
            H_band[0] = hi;
        }
        int32_t E_j = H_band[0] + G_o;
        for (uint32_t j = 1; j + 1 < band; ++j)
        {
            const uint32_t g = text[pos + i + j];
            const int32_t S_ij = (g == q) ? V : S;
            int32_t hi = imax(imax(F_band[j], E_j), H_band[j] + S_ij);
            if (type == ALN_LOCAL) { hi = imax(hi, 0); best_score = imax(best_score, hi); }
            H_band[j] = hi;
            E_j = imax(hi + G_o, E_j + G_e);
        }
        {
            const uint8_t g = text[pos + i + band - 1];
            const int32_t S_ij = (g == q) ? V : S;
            int32_t hi = imax(E_j, H_band[band - 1] + S_ij);
            if (type == ALN_LOCAL) { hi = imax(hi, 0); best_score = imax(best_score, hi); }
            H_band[band - 1] = hi;
        }
    }
    if (type == ALN_GLOBAL) best_score = H_band[band - 1];
    else if (type == ALN_SEMI_GLOBAL) {
        best_score = H_band[0];
        for (uint32_t j = 1; j < band; ++j) best_score = imax(best_score, H_band[j]);
    }
    return best_score;
}

/* ------------------------------------------------------------------------ */
/* FM-index                                                                  */
/* ------------------------------------------------------------------------ */
typedef struct {
    uint32_t        length;     /* n: number of text symbols; SA has n+1 rows     */
    uint32_t        primary;    /* row of the '$' suffix  (bwt.h:47-60)            */
    uint32_t        L2[5];      /* fmindex_impl.cu:324-327                         */
    const uint32_t* bwt_occ;    /* interleaved records, 8 words / 64 symbols       */
    const uint32_t* ssa;        /* ssa[k] = SA[k*sa_int]  (ssa_inl.h:263-276)      */
    uint32_t        sa_int;     /* 16 in production (SSA_index_multiple<16>)       */
} oracle_fmi_t;

static inline uint32_t popc32(uint32_t x) { return (uint32_t)__builtin_popcount(x); }

/* popcount_inl.h:239-245 */
static inline uint32_t popc_2bit(uint32_t x, uint32_t c)
{
    const uint32_t odd  = ((c & 2) ? x : ~x) >> 1;
    const uint32_t even = ((c & 1) ? x : ~x);
    return popc32(odd & even & 0x55555555u);
}
/* popcount_inl.h:327-330 */
static inline uint32_t hibits_2bit(uint32_t mask, uint32_t i) { return mask & ~((1u << (i << 1)) - 1u); }
/* popcount_inl.h:343-350 */
static inline uint32_t popc_2bit_i(uint32_t mask, uint32_t c, uint32_t i)
{
    const uint32_t r = popc_2bit(hibits_2bit(mask, i), c);
    return (c == 0) ? r - i : r;
}
/* bwt.h:77-88 */
static uint32_t g_count_table[256];
static int      g_count_table_ready = 0;
static void count_table_init(void)
{
    if (g_count_table_ready) return;
    for (int i = 0; i != 256; ++i) {
        uint32_t x = 0;
        for (int j = 0; j != 4; ++j)
            x |= (uint32_t)(((i & 3) == j) + ((i >> 2 & 3) == j) + ((i >> 4 & 3) == j) + ((i >> 6) == j)) << (j << 3);
        g_count_table[i] = x;
    }
    g_count_table_ready = 1;
}
/* popcount_inl.h:264-272, 486-493 */
static inline uint32_t popc_2bit_all(uint32_t b)
{
    return g_count_table[b & 0xff] + g_count_table[(b >> 8) & 0xff] +
           g_count_table[(b >> 16) & 0xff] + g_count_table[b >> 24];
}
static inline uint32_t popc_2bit_all_i(uint32_t mask, uint32_t i) { return popc_2bit_all(hibits_2bit(mask, i)) - i; }

/* rank_dictionary_inl.h:502-513 with popc :441-460 */
static uint32_t dict_rank(const uint32_t* bwt_occ, uint32_t i, uint32_t c)
{
    if (i == 0xFFFFFFFFu) return 0u;
    const uint32_t k = i >> 6;
    const uint32_t* rec = bwt_occ + (uint64_t)k * 8u;
    const uint32_t out = rec[4 + c];
    const uint32_t m = (i - k * 64u) >> 4;
    const uint32_t i_16 = ~i & 15u;
    uint32_t x = 0;
    if (m > 0) x += popc_2bit(rec[0], c);
    if (m > 1) x += popc_2bit(rec[1], c);
    if (m > 2) x += popc_2bit(rec[2], c);
    return out + x + popc_2bit_i(rec[m], c, i_16);
}
/* rank_dictionary_inl.h:515-538 with popc2 :465-499 */
static void dict_rank2(const uint32_t* bwt_occ, uint32_t rx, uint32_t ry, uint32_t c, uint32_t* ox, uint32_t* oy)
{
    if (rx == 0xFFFFFFFFu && ry == 0xFFFFFFFFu) { *ox = 0; *oy = 0; return; }
    if (rx == 0xFFFFFFFFu || rx == ry) {
        const uint32_t r = dict_rank(bwt_occ, ry, c);
        *ox = (rx == 0xFFFFFFFFu) ? 0u : r; *oy = r; return;
    }
    const uint32_t kl = rx >> 6, kh = ry >> 6;
    const uint32_t* rl = bwt_occ + (uint64_t)kl * 8u;
    const uint32_t* rh = bwt_occ + (uint64_t)kh * 8u;
    const uint32_t outl = rl[4 + c];
    const uint32_t outh = (kl == kh) ? outl : rh[4 + c];
    const uint32_t ml = (rx - kl * 64u) >> 4, mh = (ry - kh * 64u) >> 4;
    const uint32_t l_16 = ~rx & 15u, h_16 = ~ry & 15u;
    uint32_t xl = 0;
    if (ml > 0) xl += popc_2bit(rl[0], c);
    if (ml > 1) xl += popc_2bit(rl[1], c);
    if (ml > 2) xl += popc_2bit(rl[2], c);
    uint32_t xh = (kl == kh) ? xl : 0u;
    const uint32_t startm = (kl == kh) ? ml : 0u;
    if (mh > 0 && startm == 0) xh += popc_2bit(rh[0], c);
    if (mh > 1 && startm <= 1) xh += popc_2bit(rh[1], c);
    if (mh > 2 && startm <= 2) xh += popc_2bit(rh[2], c);
    xl += popc_2bit_i(rl[ml], c, l_16);
    xh += popc_2bit_i(rh[mh], c, h_16);
    *ox = outl + xl; *oy = outh + xh;
}
/* rank_dictionary_inl.h:540-553 with unpack_add :96-102 */
static void dict_rank4(const uint32_t* bwt_occ, uint32_t i, uint32_t* out4)
{
    count_table_init();
    const uint32_t k = i >> 6;
    const uint32_t* rec = bwt_occ + (uint64_t)k * 8u;
    const uint32_t m = (i - k * 64u) >> 4;
    const uint32_t i_16 = ~i & 15u;
    uint32_t x = 0;
    if (m > 0) x += popc_2bit_all(rec[0]);
    if (m > 1) x += popc_2bit_all(rec[1]);
    if (m > 2) x += popc_2bit_all(rec[2]);
    x += popc_2bit_all_i(rec[m], i_16);
    out4[0] = rec[4] + (x & 0xff);
    out4[1] = rec[5] + (x >> 8 & 0xff);
    out4[2] = rec[6] + (x >> 16 & 0xff);
    out4[3] = rec[7] + (x >> 24);
}

/* fmindex_inl.h:36-57 */
static uint32_t fm_rank(const oracle_fmi_t* f, uint32_t k, uint32_t c)
{
    if (k == 0xFFFFFFFFu) return 0;
    if (k == f->length)   return f->L2[c + 1] - f->L2[c];
    if (k >= f->primary) --k;
    return dict_rank(f->bwt_occ, k, c);
}
/* fmindex_inl.h:66-99 */
static void fm_rank2(const oracle_fmi_t* f, uint32_t rx, uint32_t ry, uint32_t c, uint32_t* ox, uint32_t* oy)
{
    if (rx == ry) { const uint32_t r = fm_rank(f, rx, c); *ox = r; *oy = r; return; }
    else if (rx == 0xFFFFFFFFu) { *ox = 0; *oy = fm_rank(f, ry, c); return; }
    if (ry == f->length) { *ox = fm_rank(f, rx, c); *oy = f->L2[c + 1] - f->L2[c]; return; }
    if (rx >= f->primary) --rx;
    if (ry >= f->primary) --ry;
    dict_rank2(f->bwt_occ, rx, ry, c, ox, oy);
}
/* fmindex_inl.h:111-135 */
static void fm_rank4(const oracle_fmi_t* f, uint32_t k, uint32_t* out4)
{
    if (k == 0xFFFFFFFFu) { out4[0] = out4[1] = out4[2] = out4[3] = 0; return; }
    if (k == f->length) { for (int c = 0; c < 4; ++c) out4[c] = f->L2[c + 1] - f->L2[c]; return; }
    if (k >= f->primary) --k;
    dict_rank4(f->bwt_occ, k, out4);
}
static inline uint32_t fm_bwt(const oracle_fmi_t* f, uint32_t k)
{   /* PackedStream<..,2,true> over the deinterleaved bwt words */
    const uint32_t word = f->bwt_occ[(uint64_t)(k >> 6) * 8u + ((k & 63u) >> 4)];
    return (word >> (30u - ((k & 15u) << 1))) & 3u;
}

ORACLE_API void oracle_fm_rank(const oracle_fmi_t* f, const uint32_t* k, const uint8_t* c, uint32_t n, uint32_t* out)
{
    for (uint32_t i = 0; i < n; ++i) out[i] = fm_rank(f, k[i], c[i]);
}
ORACLE_API void oracle_fm_rank4(const oracle_fmi_t* f, const uint32_t* k, uint32_t n, uint32_t* out /* 4n */)
{
    for (uint32_t i = 0; i < n; ++i) fm_rank4(f, k[i], out + 4 * (size_t)i);
}
ORACLE_API void oracle_fm_rank_range(const oracle_fmi_t* f, const uint32_t* range /* 2n */, const uint8_t* c, uint32_t n, uint32_t* out /* 2n */)
{
    for (uint32_t i = 0; i < n; ++i) fm_rank2(f, range[2 * i], range[2 * i + 1], c[i], &out[2 * i], &out[2 * i + 1]);
}

/* match: fmindex_inl.h:307-341.  The reference tests `c > symbol_count()`
 * (i.e. c > 4) in the library version and `c > 3` in nvBowtie's match_range
 * (nvBowtie/bowtie2/cuda/mapping_inl.h:83-97); c == 4 in the library version
 * indexes L2[5] out of bounds, so this restatement (and the product) use the
 * well-defined nvBowtie test: any symbol > 3 yields the empty range (1,0).
 * bytes_out (nullable) accumulates the algorithmic index bytes SURVEY.md 8(d)
 * defines: 32 B per distinct record touched per step. */
static void fm_match(const oracle_fmi_t* f,
    const uint32_t* w, uint32_t bits, uint32_t be, uint64_t begin, uint32_t len,
    uint32_t* ox, uint32_t* oy, uint64_t* bytes_out)
{
    uint32_t rx = 0, ry = f->length;
    uint64_t bytes = 0;
    for (int32_t i = (int32_t)len - 1; i >= 0 && rx <= ry; --i)
    {
        const uint32_t c = ps_get(w, bits, be, begin + (uint32_t)i);
        if (c > 3) { rx = 1; ry = 0; break; }
        uint32_t a, b;
        {   /* accounting only */
            uint32_t lx = rx - 1, ly = ry; int nrec = 0;
            if (lx == ly) nrec = (lx == 0xFFFFFFFFu || lx == f->length) ? 0 : 1;
            else {
                uint32_t ax = lx, ay = ly;
                int hx = (ax != 0xFFFFFFFFu), hy = (ay != f->length);
                if (hx && ax >= f->primary) --ax;
                if (hy && ay >= f->primary) --ay;
                if (hx && ax == 0xFFFFFFFFu) hx = 0;
                if (hx && hy) nrec = ((ax >> 6) == (ay >> 6)) ? 1 : 2; else nrec = hx + hy;
            }
            bytes += 32u * (uint64_t)nrec;
        }
        fm_rank2(f, rx - 1, ry, c, &a, &b);
        rx = f->L2[c] + a + 1;
        ry = f->L2[c] + b;
    }
    *ox = rx; *oy = ry;
    if (bytes_out) *bytes_out += bytes;
}

ORACLE_API void oracle_fm_match(const oracle_fmi_t* f,
    const uint32_t* w, uint32_t bits, uint32_t be, const uint64_t* begin, const uint32_t* len,
    uint32_t n, uint32_t* out_range /* 2n */, uint64_t* algo_bytes /* nullable, 1 */, int n_threads)
{
    uint64_t total = 0;
#if defined(_OPENMP)
    if (n_threads > 0) omp_set_num_threads(n_threads);
    #pragma omp parallel for schedule(static) reduction(+:total)
#endif
    for (int64_t i = 0; i < (int64_t)n; ++i) {
        uint64_t b = 0;
        fm_match(f, w, bits, be, begin[i], len[i], &out_range[2 * i], &out_range[2 * i + 1], &b);
        total += b;
    }
    if (algo_bytes) *algo_bytes = total;
}

/* locate: fmindex_inl.h:466-501 with SSA_index_multiple_context::fetch (ssa_inl.h:486-497) */
static uint32_t fm_locate(const oracle_fmi_t* f, uint32_t i, uint32_t* steps)
{
    uint32_t j = i, t = 0;
    while ((j & (f->sa_int - 1)) != 0)
    {
        if (j != f->primary) {
            const uint32_t c = (j < f->primary) ? fm_bwt(f, j) : fm_bwt(f, j - 1);
            j = f->L2[c] + fm_rank(f, j, c);
        } else
            j = 0;
        ++t;
    }
    if (steps) *steps = t;
    return f->ssa[j / f->sa_int] + t;
}
ORACLE_API void oracle_fm_locate(const oracle_fmi_t* f, const uint32_t* rows, uint32_t n, uint32_t* out_pos,
    uint64_t* total_steps /* nullable */, int n_threads)
{
    uint64_t total = 0;
#if defined(_OPENMP)
    if (n_threads > 0) omp_set_num_threads(n_threads);
    #pragma omp parallel for schedule(static) reduction(+:total)
#endif
    for (int64_t i = 0; i < (int64_t)n; ++i) {
        uint32_t t = 0;
        out_pos[i] = fm_locate(f, rows[i], &t);
        total += t;
    }
    if (total_steps) *total_steps = total;
}
/* locate_ssa_iterator / lookup_ssa_iterator: fmindex_inl.h:511-569 */
ORACLE_API void oracle_fm_locate_ssa_iterator(const oracle_fmi_t* f, const uint32_t* rows, uint32_t n, uint32_t* out_it /* 2n */)
{
    for (uint32_t q = 0; q < n; ++q) {
        uint32_t j = rows[q], t = 0;
        while ((j & (f->sa_int - 1)) != 0) {
            if (j != f->primary) {
                const uint32_t c = (j < f->primary) ? fm_bwt(f, j) : fm_bwt(f, j - 1);
                j = f->L2[c] + fm_rank(f, j, c);
            } else j = 0;
            ++t;
        }
        out_it[2 * q] = j; out_it[2 * q + 1] = t;
    }
}
ORACLE_API void oracle_fm_lookup_ssa_iterator(const oracle_fmi_t* f, const uint32_t* it /* 2n */, uint32_t n, uint32_t* out_pos)
{
    for (uint32_t q = 0; q < n; ++q) out_pos[q] = f->ssa[it[2 * q] / f->sa_int] + it[2 * q + 1];
}

/* ------------------------------------------------------------------------ */
/* Index construction helpers (host, small sizes; test tooling)              */
/* ------------------------------------------------------------------------ */

/* gen_bwt_from_sa: bwt.h:47-60.  T: n symbols (uint8), SA: n+1 rows with
 * SA[0] = n.  bwt: n symbols out.  Returns primary. */
ORACLE_API uint32_t oracle_bwt_from_sa(uint32_t n, const uint8_t* T, const uint32_t* SA, uint8_t* bwt /* n+1 scratch */)
{
    uint32_t i, primary = 0;
    for (i = 0; i <= n; ++i) {
        if (SA[i] == 0) primary = i;
        else bwt[i] = T[SA[i] - 1];
    }
    for (i = primary; i < n; ++i) bwt[i] = bwt[i + 1];
    return primary;
}

/* build_occurrence_table<2,64> (rank_dictionary_inl.h:42-77) over the
 * big-endian 2-bit packed BWT, fused with the interleave of
 * fmindex_impl.cu:305-327.  bwt_words: ceil(n/64)*4 words (zero padded).
 * out: ceil(n/64)*8 words.  L2: 5 entries. */
ORACLE_API void oracle_build_bwt_occ(uint32_t n, const uint32_t* bwt_words, uint32_t* bwt_occ, uint32_t* L2)
{
    uint32_t counters[4] = { 0, 0, 0, 0 };
    const uint32_t n_blocks = (n + 63u) / 64u;
    for (uint32_t k = 0; k < n_blocks; ++k) {
        uint32_t* rec = bwt_occ + (uint64_t)k * 8u;
        for (int w = 0; w < 4; ++w) rec[w] = bwt_words[(uint64_t)k * 4u + w];
        for (int c = 0; c < 4; ++c) rec[4 + c] = counters[c];
        for (uint32_t s = 0; s < 64u && k * 64u + s < n; ++s)
            ++counters[ps_get(bwt_words, 2, 1, (uint64_t)k * 64u + s)];
    }
    L2[0] = 0;
    for (int c = 0; c < 4; ++c) L2[c + 1] = L2[c] + counters[c];
}

/* SSA_index_multiple<K>(n, sa): ssa_inl.h:263-276, then ssa[0] = -1 as the
 * FM-index-driven constructor does (ssa_inl.h:308) -- the convention the
 * production loader hands to the kernels. */
ORACLE_API void oracle_build_ssa(uint32_t n, const uint32_t* SA, uint32_t K, uint32_t* ssa)
{
    const uint32_t n_items = (n + 1 + K - 1) / K;
    for (uint32_t i = 0; i < n_items; ++i) ssa[i] = SA[(uint64_t)i * K];
    ssa[0] = 0xFFFFFFFFu;
}

/* FMIndexFilter<host_tag>::rank (filter_inl.h:200-231): ranges[q] = match(seed q);
 * slots = inclusive_scan(1 + y - x) with empty ranges contributing 0. Returns n_hits. */
ORACLE_API uint64_t oracle_filter_rank(const oracle_fmi_t* f,
    const uint32_t* w, uint32_t bits, uint32_t be, const uint64_t* begin, const uint32_t* len,
    uint32_t n, uint32_t* ranges /* 2n */, uint64_t* slots /* n */)
{
    uint64_t acc = 0;
    for (uint32_t q = 0; q < n; ++q) {
        fm_match(f, w, bits, be, begin[q], len[q], &ranges[2 * q], &ranges[2 * q + 1], NULL);
        const uint32_t x = ranges[2 * q], y = ranges[2 * q + 1];
        acc += (uint64_t)(uint32_t)(1u + y - x);           /* filter_inl.h:40-41 range_size (uint32 arithmetic) */
        slots[q] = acc;
    }
    return acc;
}
/* FMIndexFilter<host_tag>::locate (filter_inl.h:233-259): hit h in [begin,end):
 * slot = upper_bound(slots, h); row = ranges[slot].x + (h - base); out = (locate(row), slot) */
ORACLE_API void oracle_filter_locate(const oracle_fmi_t* f,
    const uint32_t* ranges, const uint64_t* slots, uint32_t n_queries,
    uint64_t begin, uint64_t end, uint32_t* hits /* 2 per hit: text_pos, seed id */)
{
    for (uint64_t h = begin; h < end; ++h) {
        uint32_t lo = 0, hi = n_queries;
        while (lo < hi) { const uint32_t mid = (lo + hi) / 2; if (slots[mid] <= h) lo = mid + 1; else hi = mid; }
        const uint32_t slot = lo;
        const uint64_t base = slot ? slots[slot - 1] : 0u;
        const uint32_t row  = ranges[2 * slot] + (uint32_t)(h - base);
        hits[2 * (h - begin)]     = fm_locate(f, row, NULL);
        hits[2 * (h - begin) + 1] = slot;
    }
}

/* ------------------------------------------------------------------------ */
/* nvBowtie exact seed mapping: what a "seed hit set" is    */
/*   match_range                nvBowtie/bowtie2/cuda/mapping_inl.h:83-97    */
/*   seed_mapper<EXACT_MAPPING> mapping_inl.h:229-312 (USE_REVERSE_INDEX 0)  */
/*   map_queues_kernel          mapping_inl.h:511-592                        */
/*   SeedHit                    seed_hit.h:54-223                            */
/*   complement_functor<4>      nvbio/basic/numbers.h:1338-1352              */
/* ------------------------------------------------------------------------ */
typedef struct {
    uint32_t seed_len, min_read_len, max_hits, max_reseed, retry, rep_seeds, fw, rc;
} oracle_map_params_t;

/* SeedHit: {uint32 range_begin; uint32 range_delta:20, pos:10, rc:1, indexdir:1}, exclusive range */
static inline uint64_t seed_hit_pack(uint32_t begin, uint32_t delta, uint32_t pos, uint32_t rc, uint32_t indexdir)
{
    const uint32_t w1 = (delta & 0xFFFFFu) | ((pos & 0x3FFu) << 20) | ((rc & 1u) << 30) | ((indexdir & 1u) << 31);
    return ((uint64_t)w1 << 32) | begin;
}

/* ------------------------------------------------------------------------ */
/* The seed-hit deque: priority_deque<SeedHit, vector_view, hit_compare>       */
/*   nvbio/basic/priority_deque.h:329-421 over the interval heap of            */
/*   nvbio/basic/interval_heap.h:195-260,389-533 (N. McClatchey's, Boost SL),  */
/*   hit_compare nvBowtie/bowtie2/cuda/seed_hit.h:235-244 (larger range first) */
/* Even slots hold interval lower bounds, odd slots upper bounds; slot 0 is a  */
/* hit of LARGEST range (pop_bottom drops it), slot 1 (slot 0 when alone) one   */
/* of SMALLEST range (top()).  The array order decides which of several         */
/* equal-sized hits the selection stage meets first, so every swap is the       */
/* reference's.  Pinned against the reference header itself (oracle/_ref).      */
/* ------------------------------------------------------------------------ */
static inline int hit_before(uint64_t f, uint64_t s) { return ((f >> 32) & 0xFFFFFu) > ((s >> 32) & 0xFFFFFu); }
static inline void hit_swap(uint64_t* a, long i, long j) { const uint64_t t = a[i]; a[i] = a[j]; a[j] = t; }

/* interval_heap.h:389-409 */
static void ih_sift_up(uint64_t* a, long i, int lower, long limit)
{
    while (i >= limit) {
        const long parent = ((i / 2 - 1) | 1) ^ (lower ? 1 : 0);
        if (!(lower ? hit_before(a[i], a[parent]) : hit_before(a[parent], a[i]))) break;
        hit_swap(a, i, parent);
        i = parent;
    }
}
/* :412-429 */
static void ih_leaf_upper(uint64_t* a, long n, long i, long limit)
{
    const long co = (i * 2 < n) ? i * 2 : (i ^ 1);
    if (hit_before(a[i], a[co])) { hit_swap(a, i, co); ih_sift_up(a, co, 1, limit); }
    else ih_sift_up(a, i, 0, limit);
}
/* :432-455 */
static void ih_leaf_lower(uint64_t* a, long n, long i, long limit)
{
    long co = i | 1;
    if (co >= n) { if (co == 1) return; co = (co / 2 - 1) | 1; }
    if (hit_before(a[co], a[i])) { hit_swap(a, i, co); ih_sift_up(a, co, 0, limit); }
    else ih_sift_up(a, i, 1, limit);
}
/* :458-517 */
static void ih_sift_down(uint64_t* a, long n, long i, int lower, long limit)
{
    const long end_parent = n / 2 - ((lower && (n & 3) == 0) ? 2 : 1);
    while (i < end_parent) {
        long child = i * 2 + (lower ? 2 : 1);
        if (lower ? hit_before(a[child + 2], a[child]) : hit_before(a[child], a[child + 2])) child += 2;
        hit_swap(a, i, child);
        i = child;
    }
    if (i <= end_parent + (lower ? 0 : 1)) {
        long child = i * 2 + (lower ? 2 : 1);
        if (child < n) {
            if (!lower && child + 1 < n && hit_before(a[child], a[child + 1])) {
                ++child;
                hit_swap(a, i, child);
                ih_leaf_lower(a, n, child, limit);
                return;
            }
            hit_swap(a, i, child);
            i = child;
        }
    }
    if (lower) ih_leaf_lower(a, n, i, limit); else ih_leaf_upper(a, n, i, limit);
}
/* push (priority_deque.h:354-357): a[n-1] is the new element */
static void ih_push(uint64_t* a, long n)
{
    if ((n - 1) & 1) ih_leaf_upper(a, n, n - 1, 2); else ih_leaf_lower(a, n, n - 1, 2);
}
/* pop_bottom (:397-402): afterwards the heap is a[0..n-1) */
static void ih_pop_bottom(uint64_t* a, long n) { hit_swap(a, 0, n - 1); ih_sift_down(a, n - 1, 0, 1, 2); }
/* pop_top (:412-417) */
static void ih_pop_top(uint64_t* a, long n)
{
    if (n <= 2) return;
    hit_swap(a, 1, n - 1);
    ih_sift_down(a, n - 1, 1, 0, 2);
}
/* make_interval_heap (interval_heap.h:356-384): what priority_deque(seq, constructed = false) runs (priority_deque.h:320-325) -- and
 * SeedHitDequeArrayDeviceView::get_deque(read_id, build_heap = false) passes that false as `constructed`
 * (seed_hit_deque_array_inl.h:104-109), so EVERY hits[read_id] the selection kernels take rebuilds the read's heap by the hits'
 * current range sizes: the hits change slots between rounds while the probability-tree leaves stay where select_init put them. */
static void ih_make(uint64_t* a, long n)
{
    if (n <= 1) return;
    const long end_parent = n / 2 - 1;
    long i = n ^ (n & 1);                          /* a trailing singleton interval is skipped */
    do {
        i -= 2;
        const long stop = (i <= end_parent) ? (i * 2 + 2) : n;
        if (hit_before(a[i + 1], a[i])) hit_swap(a, i + 1, i);
        ih_sift_down(a, n, i + 1, 0, stop);
        ih_sift_down(a, n, i, 1, stop);
    } while (i >= 2);
}
ORACLE_API void oracle_hit_deque_make(uint64_t* a, uint32_t n)       { ih_make(a, n); }
ORACLE_API void oracle_hit_deque_push(uint64_t* a, uint32_t n)       { ih_push(a, n); }
ORACLE_API void oracle_hit_deque_pop_bottom(uint64_t* a, uint32_t n) { ih_pop_bottom(a, n); }
ORACLE_API void oracle_hit_deque_pop_top(uint64_t* a, uint32_t n)    { ih_pop_top(a, n); }

/* match_range over a transformed seed: symbol t of the scan = comp(seed[reverse ? len-1-t : t]) */
static void match_range_x(const oracle_fmi_t* f, const uint32_t* w, uint32_t bits, uint32_t be, uint64_t begin, uint32_t len,
                          int reverse, int complement, uint32_t* ox, uint32_t* oy)
{
    uint32_t rx = 0, ry = f->length;
    for (uint32_t t = 0; t < len && rx <= ry; ++t) {
        uint32_t c = ps_get(w, bits, be, begin + (reverse ? len - 1 - t : t));
        if (complement) c = (c >= 4) ? c : 3u - c;
        if (c > 3) { rx = 1; ry = 0; break; }
        uint32_t a, b;
        fm_rank2(f, rx - 1, ry, c, &a, &b);
        rx = f->L2[c] + a + 1;
        ry = f->L2[c] + b;
    }
    *ox = rx; *oy = ry;
}

/* one read -> its seed hits, laid out as the reference's hit deque (interval heap, see above); when a hit
 * arrives with max_hits already held, the deque's bottom (a hit of largest range size) is dropped first.
 * Returns the number of hits kept; *reseed as map_queues_kernel computes it. */
ORACLE_API void oracle_map_exact(const oracle_fmi_t* f,
    const uint32_t* w, uint32_t bits, uint32_t be, const uint64_t* read_begin, const uint32_t* read_len,
    const uint32_t* in_queue /* nullable */, uint32_t n, const oracle_map_params_t* p, const uint32_t* seed_freq_by_len,
    uint64_t* out_hits, uint32_t hits_stride, uint32_t* out_counts, uint8_t* out_reseed /* nullable */)
{
    for (uint32_t id = 0; id < n; ++id)
    {
        const uint32_t read_id = in_queue ? in_queue[id] : id;
        const uint64_t rx = read_begin[read_id];
        const uint32_t rlen = read_len[read_id];
        uint64_t* hits = out_hits + (uint64_t)read_id * hits_stride;
        uint32_t nh = 0, range_sum = 0, range_count = 0;
        if (rlen < p->min_read_len) { out_counts[read_id] = 0; continue; }   /* reseed[id] is left untouched, as in :542-546 */
        const uint32_t seed_len = p->seed_len < rlen ? p->seed_len : rlen;
        const uint32_t seed_freq = seed_freq_by_len[rlen];
        const uint32_t retry_stride = seed_freq / (p->max_reseed + 1);
        for (uint64_t pos = rx + (uint64_t)p->retry * retry_stride; pos + seed_len <= rx + rlen; pos += seed_freq)
        {
            int has_n = 0;
            for (uint32_t i = 0; i < seed_len; ++i) has_n |= (ps_get(w, bits, be, pos + i) == 4u);   /* count_occurrences(...,4u,1u) :258 */
            if (has_n) continue;
            for (int strand = 0; strand < 2; ++strand)
            {
                if (strand == 0 ? !p->fw : !p->rc) continue;
                uint32_t x, y;
                match_range_x(f, w, bits, be, pos, seed_len, strand, strand, &x, &y);
                if (x > y) continue;
                const uint32_t pir = strand == 0 ? (uint32_t)(rx + rlen - pos - seed_len) : (uint32_t)(pos - rx);
                if (nh == p->max_hits || nh == hits_stride) { ih_pop_bottom(hits, nh); --nh; }     /* :268-270 */
                hits[nh++] = seed_hit_pack(x, y + 1u - x, pir, strand, 0);
                ih_push(hits, nh);
                range_sum += y - x + 1u; range_count++;
            }
        }
        out_counts[read_id] = nh;
        if (out_reseed) out_reseed[id] = (range_count == 0 || range_sum >= p->rep_seeds * range_count);
    }
}

/* ------------------------------------------------------------------------ */
/* nvBowtie seed mapping with one mismatch                                    */
/*   map<find_exact>                   mapping_inl.h:124-223                  */
/*   seed_mapper<APPROX_MAPPING>       mapping_inl.h:318-365                  */
/*   seed_mapper<CASE_PRUNING_MAPPING> mapping_inl.h:372-428                  */
/*   map_queues_kernel<ALGO>           mapping_inl.h:511-592                  */
/*   map_t (algorithm choice)          mapping_inl.h:809-843                  */
/* ------------------------------------------------------------------------ */
typedef struct { const uint32_t* w; uint32_t bits, be; uint64_t begin; uint32_t len; int reverse, complement; } seed_reader_t;

static inline uint32_t seed_sym(const seed_reader_t* r, uint32_t t)
{
    uint32_t c = ps_get(r->w, r->bits, r->be, r->begin + (r->reverse ? r->len - 1 - t : t));
    if (r->complement) c = (c >= 4) ? c : 3u - c;            /* complement_functor<4> */
    return c;
}

/* match_range (mapping_inl.h:80-96): inclusive range in/out */
static void match_range_span(const oracle_fmi_t* f, const seed_reader_t* q, uint32_t begin, uint32_t end, uint32_t* rx, uint32_t* ry)
{
    for (uint32_t i = begin; i < end && *rx <= *ry; ++i) {
        const uint32_t c = seed_sym(q, i);
        if (c > 3) { *rx = 1; *ry = 0; return; }
        uint32_t a, b;
        fm_rank2(f, *rx - 1, *ry, c, &a, &b);
        *rx = f->L2[c] + a + 1;
        *ry = f->L2[c] + b;
    }
}

typedef struct { uint64_t* hits; uint32_t nh, cap, range_sum, range_count; } hit_heap_t;

static void heap_push(hit_heap_t* h, uint32_t x, uint32_t y /* inclusive */, uint32_t pos, uint32_t rc, uint32_t indexdir)
{
    if (h->nh == h->cap) { ih_pop_bottom(h->hits, h->nh); --h->nh; }
    h->hits[h->nh++] = seed_hit_pack(x, y + 1u - x, pos, rc, indexdir);
    ih_push(h->hits, h->nh);
    h->range_sum += y - x + 1u; h->range_count++;
}

static void map_one_mismatch(int find_exact, const seed_reader_t* q, uint32_t len1, uint32_t len2, const oracle_fmi_t* f,
                             uint32_t pos, uint32_t rc, uint32_t indexdir, hit_heap_t* heap)
{
    /* a read shorter than subseed_len makes the reference index outside its seed (query[i], i >= len2, and
     * ReverseXform wraps below zero): undefined there, defined here as an all-exact search of the seed */
    if (len1 > len2) len1 = len2;
    uint32_t N_pos = 0, N_cnt = 0;
    for (uint32_t i = 0; i < len2; ++i)
        if (seed_sym(q, i) > 3) { if (i < len1 || N_cnt) return; N_pos = i; N_cnt++; }
    len1 = N_cnt ? N_pos : len1;
    uint32_t bx = 0, by = f->length;
    match_range_span(f, q, 0, len1, &bx, &by);
    for (uint32_t i = len1; i < len2 && bx <= by; ++i)
    {
        const uint32_t c = seed_sym(q, i);
        uint32_t lo[4], hi[4];
        fm_rank4(f, bx - 1, lo); fm_rank4(f, by, hi);
        for (uint32_t sub = 0; sub < 4; ++sub)
            if (sub != c && hi[sub] > lo[sub]) {
                uint32_t x = f->L2[sub] + lo[sub] + 1, y = f->L2[sub] + hi[sub];
                match_range_span(f, q, i + 1, len2, &x, &y);
                if (x <= y) heap_push(heap, x, y, pos, rc, indexdir);
            }
        if (c < 4) { bx = f->L2[c] + lo[c] + 1; by = f->L2[c] + hi[c]; }
        else { bx = 1; by = 0; break; }
    }
    if (find_exact && bx <= by) heap_push(heap, bx, by, pos, rc, indexdir);
}

/* algorithm: 0 exact, 1 approx (subseed_len exact + 1 mismatch in the rest), 2 case pruning (needs rf = index of the reversed text) */
ORACLE_API void oracle_map(int algorithm, uint32_t subseed_len, const oracle_fmi_t* f, const oracle_fmi_t* rf,
    const uint32_t* w, uint32_t bits, uint32_t be, const uint64_t* read_begin, const uint32_t* read_len,
    const uint32_t* in_queue /* nullable */, uint32_t n, const oracle_map_params_t* p, const uint32_t* seed_freq_by_len,
    uint64_t* out_hits, uint32_t hits_stride, uint32_t* out_counts, uint8_t* out_reseed /* nullable */)
{
    if (algorithm == 0) { oracle_map_exact(f, w, bits, be, read_begin, read_len, in_queue, n, p, seed_freq_by_len, out_hits, hits_stride, out_counts, out_reseed); return; }
    for (uint32_t id = 0; id < n; ++id)
    {
        const uint32_t read_id = in_queue ? in_queue[id] : id;
        const uint64_t rx = read_begin[read_id];
        const uint32_t rlen = read_len[read_id];
        hit_heap_t heap = { out_hits + (uint64_t)read_id * hits_stride, 0, p->max_hits < hits_stride ? p->max_hits : hits_stride, 0, 0 };
        if (rlen < p->min_read_len) { out_counts[read_id] = 0; continue; }
        const uint32_t seed_len = p->seed_len < rlen ? p->seed_len : rlen;
        const uint32_t seed_freq = seed_freq_by_len[rlen];
        const uint32_t retry_stride = seed_freq / (p->max_reseed + 1);
        for (uint64_t pos = rx + (uint64_t)p->retry * retry_stride; pos + seed_len <= rx + rlen; pos += seed_freq)
        {
            const seed_reader_t fr  = { w, bits, be, pos, seed_len, 0, 0 }, rr  = { w, bits, be, pos, seed_len, 1, 0 };
            const seed_reader_t cfr = { w, bits, be, pos, seed_len, 0, 1 }, crr = { w, bits, be, pos, seed_len, 1, 1 };
            const uint32_t rel = (uint32_t)(pos - rx);
            if (algorithm == 1) {
                uint32_t nN = 0;
                for (uint32_t i = 0; i < seed_len; ++i) nN += (seed_sym(&fr, i) == 4u);
                if (nN >= 2) continue;                                                           /* :346 */
                if (p->fw) map_one_mismatch(1, &fr,  subseed_len, seed_len, f, rlen - rel - seed_len, 0, 0, &heap);
                if (p->rc) map_one_mismatch(0, &crr, subseed_len, seed_len, f, rel,                   1, 0, &heap);
            } else {
                if (p->fw) map_one_mismatch(1, &fr,  seed_len / 2,       seed_len, f,  rlen - rel - seed_len, 0, 0, &heap);
                if (p->fw) map_one_mismatch(0, &rr,  (seed_len + 1) / 2, seed_len, rf, rlen - rel - 1,        0, 1, &heap);
                if (p->rc) map_one_mismatch(1, &cfr, seed_len / 2,       seed_len, rf, rel + seed_len - 1,    1, 1, &heap);
                if (p->rc) map_one_mismatch(0, &crr, (seed_len + 1) / 2, seed_len, f,  rel,                   1, 0, &heap);
            }
        }
        out_counts[read_id] = heap.nh;
        if (out_reseed) out_reseed[id] = (heap.range_count == 0 || heap.range_sum >= p->rep_seeds * heap.range_count);
    }
}

/* SimpleFunc (nvBowtie/bowtie2/cuda/func.h:39-70) tabulated for x in [0,n): type 0 linear, 1 log, 2 sqrt */
#include <math.h>
ORACLE_API void oracle_simple_func_table(int type, float k, float m, uint32_t n, uint32_t* out)
{
    for (uint32_t x = 0; x < n; ++x)
        out[x] = (uint32_t)(int32_t)(k + m * (type == 1 ? logf((float)x) : type == 2 ? sqrtf((float)x) : (float)x));
}

/* ------------------------------------------------------------------------ */
/* nvBowtie score reduction and mapping quality                               */
/*   io::Alignment / BestAlignments   nvbio/io/alignments.h:80-176            */
/*   io::distinct_alignments          nvbio/io/alignments_inl.h:35-47         */
/*   score_reduce_kernel              nvBowtie/bowtie2/cuda/reduce_inl.h:71-160 */
/*   BowtieMapq2 / BowtieMapq3        nvBowtie/bowtie2/cuda/mapq.h:42-330     */
/* io::Alignment is two words: {score_sgn:1, score:17, ed:10, rc:1, mate:1,    */
/* paired:1, discordant:1} (first bit-field in the low bits) and m_align.      */
/* ------------------------------------------------------------------------ */
static int32_t simple_func(int type, float k, float m, int32_t x);
typedef struct { uint32_t w, align; } io_aln_t;
static inline io_aln_t io_aln_make(uint32_t pos, uint32_t ed, int32_t score, uint32_t rc)
{
    io_aln_t a;
    const uint32_t mag = score < 0 ? (uint32_t)(-score) : (uint32_t)score;
    a.w = (score < 0 ? 1u : 0u) | ((mag & 0x1FFFFu) << 1) | ((ed & 0x3FFu) << 18) | ((rc & 1u) << 28);
    a.align = pos;
    return a;
}
static inline int32_t  io_aln_score(io_aln_t a) { const int32_t m = (int32_t)((a.w >> 1) & 0x1FFFFu); return (a.w & 1u) ? -m : m; }
static inline uint32_t io_aln_rc(io_aln_t a) { return (a.w >> 28) & 1u; }
static inline int      io_aln_aligned(io_aln_t a) { return a.align != 0xFFFFFFFFu; }
ORACLE_API uint64_t oracle_alignment_invalid(void)
{   /* Alignment::invalid() = Alignment(uint32(-1), max_ed() = 255, max_score() = 2^17-1, 0) */
    const io_aln_t a = io_aln_make(0xFFFFFFFFu, 255u, (1 << 17) - 1, 0u);
    return ((uint64_t)a.align << 32) | a.w;
}
static inline int distinct_alignments(uint32_t pos1, uint32_t rc1, uint32_t pos2, uint32_t rc2, uint32_t dist)
{
    if (rc1 != rc2) return 1;
    return (pos1 >= pos2 - (pos2 < dist ? pos2 : dist) && pos1 <= pos2 + dist) ? 0 : 1;
}

/* init_alignments_kernel (aligner.h:323-346); `mate` lands in the constructor's rc slot there */
ORACLE_API void oracle_init_alignments(uint32_t n_reads, const uint32_t* read_len, int min_type, float min_k, float min_m, uint32_t mate,
    uint64_t* best, uint32_t best_stride)
{
    for (uint32_t r = 0; r < n_reads; ++r) {
        const io_aln_t a = io_aln_make(0xFFFFFFFFu, 255u, simple_func(min_type, min_k, min_m, (int32_t)read_len[r]), mate);
        best[r] = best[r + best_stride] = ((uint64_t)a.align << 32) | a.w;
    }
}

/* best: [2][best_stride] words pairs {w, align} (best at read_id, second best at read_id + best_stride);
 * hits of active read t: [hit_begin[t], hit_begin[t+1]) in extension order */
ORACLE_API void oracle_score_reduce(uint32_t n_active, const uint32_t* read_ids /* nullable */, const uint64_t* hit_begin,
    const int32_t* hit_score, const uint32_t* hit_loc, const uint8_t* hit_rc, const uint32_t* read_len /* by read id */,
    uint64_t* best, uint32_t best_stride)
{
    for (uint32_t t = 0; t < n_active; ++t)
    {
        const uint32_t read_id = read_ids ? read_ids[t] : t;
        io_aln_t a1 = { (uint32_t)best[read_id], (uint32_t)(best[read_id] >> 32) };
        io_aln_t a2 = { (uint32_t)best[read_id + best_stride], (uint32_t)(best[read_id + best_stride] >> 32) };
        const uint32_t len = read_len[read_id];
        for (uint64_t i = hit_begin[t]; i < hit_begin[t + 1]; ++i)
        {
            const int32_t score = hit_score[i]; const uint32_t g_pos = hit_loc[i], rc = hit_rc[i];
            if ((rc == io_aln_rc(a1) && g_pos == a1.align) || (rc == io_aln_rc(a2) && g_pos == a2.align)) continue;
            if (score > io_aln_score(a1)) { a2 = a1; a1 = io_aln_make(g_pos, 0u, score, rc); }
            else if (score > io_aln_score(a2) && distinct_alignments(a1.align, io_aln_rc(a1), g_pos, rc, len / 2)) a2 = io_aln_make(g_pos, 0u, score, rc);
        }
        best[read_id] = ((uint64_t)a1.align << 32) | a1.w;
        best[read_id + best_stride] = ((uint64_t)a2.align << 32) | a2.w;
    }
}

/* ------------------------------------------------------------------------ */
/* nvBowtie hit selection and the best-approx extension loop's per-round stages */
/*   select_init_kernel            nvBowtie/bowtie2/cuda/select.cu:36-103        */
/*   select_kernel / rand_select_kernel / select_multi_kernel /                 */
/*   rand_select_multi_kernel, randomized_select   select_inl.h:74-607          */
/*   SumTree<float*>, sample()     nvbio/basic/sum_tree_inl.h:38-178             */
/*   SeedHit::pop_front / empty    seed_hit.h:136-149                            */
/*   packed_read / packed_seed     defs.h:152-181                                */
/*   locate (index direction)      locate_inl.h:53-143                           */
/*   BestScoreStream               score_best_inl.h:54-148                       */
/*   ReduceBestApproxContext       reduce.h:63-105 ; score_reduce_kernel         */
/*                                 reduce_inl.h:71-160                           */
/* Layout here: a read's deque lives at hits[read_id * hits_stride], its size in */
/* counts[read_id], its probability tree at probs[read_id * probs_stride].       */
/* Active reads are packed_read words (read_id:31, top_flag:1); a selected hit   */
/* carries {read_id, loc (SA row), seed = packed_seed word (pos_in_read:12,      */
/* index_dir:1, rc:1, top_flag:1)}.  The reference allocates output slots with   */
/* atomics (run-dependent order); here outputs follow the input queue's order.   */
/* Single precision, every operation rounded on its own (host arithmetic).       */
/* ------------------------------------------------------------------------ */
static inline uint32_t ilog2_u32(uint32_t n)       /* nvbio::log2, numbers.h:516-524 */
{
    uint32_t c = 0;
    if (n & 0xffff0000u) { n >>= 16; c |= 16; }
    if (n & 0xff00) { n >>= 8; c |= 8; }
    if (n & 0xf0) { n >>= 4; c |= 4; }
    if (n & 0xc) { n >>= 2; c |= 2; }
    if (n & 0x2) c |= 1;
    return c;
}
static inline uint32_t st_padded(uint32_t size) { const uint32_t l = ilog2_u32(size); return (1u << l) < size ? 1u << (l + 1u) : 1u << l; }
ORACLE_API uint32_t oracle_sum_tree_node_count(uint32_t size) { return st_padded(size) * 2u - 1u; }
static void st_setup(float* c, uint32_t size, uint32_t padded)
{
    for (uint32_t i = size; i < padded; ++i) c[i] = 0.0f;
    uint32_t src = 0;
    for (uint32_t n = padded; n >= 2; n >>= 1) {
        const uint32_t dst = src + n, m = n >> 1;
        for (uint32_t i = 0; i < m; ++i) c[dst + i] = c[src + i * 2] + c[src + i * 2 + 1u];
        src += n;
    }
}
static void st_set(float* c, uint32_t padded, uint32_t i, float v)
{
    c[i] = v;
    uint32_t prev = 0u, base = padded, parent = i >> 1;
    for (uint32_t m = padded >> 1; base + parent < padded * 2u - 1u; m >>= 1) {
        c[base + parent] = c[prev + parent * 2] + c[prev + parent * 2 + 1];
        prev = base; base += m; parent >>= 1;
    }
}
static inline float st_sum(const float* c, uint32_t padded) { return c[padded * 2u - 2u]; }
static inline float fmin1(float a) { return a < 1.0f ? a : 1.0f; }          /* nvbio::min(a, 1.0f) */
static uint32_t st_sample(const float* c, uint32_t size, uint32_t padded, float value)
{
    uint32_t base = padded * 2u - 4u, node = 0;
    float v = value;
    for (uint32_t m = 2; m < padded; m *= 2) {
        const float l = c[base + node], r = c[base + node + 1u];
        const float sum = l + r;
        if (sum == 0.0f) node *= 2;
        else {
            const float vs = v * sum;
            if (vs < l || r == 0.0f) { node = node * 2u; v = fmin1(vs / l); }
            else { node = (node + 1u) * 2u; const float d = vs - l; v = fmin1(d / r); }
        }
        base -= m * 2;
    }
    {
        const float l = node < size ? c[node] : 0.0f, r = node + 1u < size ? c[node + 1u] : 0.0f;
        const float sum = l + r;
        const float vs = v * sum;
        node = (vs < l || r == 0.0f) ? node : node + 1u;
    }
    return node < size ? node : size - 1u;
}
/* exposed so the tree can be exercised on its own */
ORACLE_API void oracle_sum_tree_setup(float* c, uint32_t size) { st_setup(c, size, st_padded(size)); }
ORACLE_API void oracle_sum_tree_set(float* c, uint32_t size, uint32_t i, float v) { st_set(c, st_padded(size), i, v); }
ORACLE_API uint32_t oracle_sum_tree_sample(const float* c, uint32_t size, float v) { return st_sample(c, size, st_padded(size), v); }

static inline uint32_t hit_delta(uint64_t h) { return (uint32_t)(h >> 32) & 0xFFFFFu; }
static inline uint32_t hit_pop_front(uint64_t* h)      /* SeedHit::pop_front: ++range_begin, --range_delta (20-bit field) */
{
    const uint32_t r = (uint32_t)*h;
    const uint32_t hi = (uint32_t)(*h >> 32);
    const uint32_t nhi = (hi & ~0xFFFFFu) | ((hi - 1u) & 0xFFFFFu);
    *h = ((uint64_t)nhi << 32) | (uint32_t)(r + 1u);
    return r;
}
static inline uint32_t packed_seed_of(uint64_t h, uint32_t top_flag)
{
    const uint32_t hi = (uint32_t)(h >> 32);
    return ((hi >> 20) & 0x3FFu) | (((hi >> 31) & 1u) << 12) | (((hi >> 30) & 1u) << 13) | ((top_flag & 1u) << 14);
}

ORACLE_API void oracle_select_init(uint32_t n_reads, const char* names /* nullable */, const uint32_t* names_idx,
    const uint64_t* hits, uint32_t hits_stride, const uint32_t* counts, float* probs, uint32_t probs_stride,
    uint32_t* trys /* nullable */, uint32_t* rseeds, uint32_t max_effort_init, int randomized, int top_seed)
{
    for (uint32_t r = 0; r < n_reads; ++r)
    {
        if (trys) trys[r] = max_effort_init;
        if (!randomized) continue;
        if (names) {                                   /* djb2 (xor form) of the read name, select.cu:64-76 */
            const uint32_t off = names_idx[r], len = names_idx[r + 1] - off;
            uint32_t hash = 5381;
            for (uint32_t i = 0; i < len && names[off + i]; ++i) hash = ((hash << 5) + hash) ^ (uint32_t)(int32_t)names[off + i];
            rseeds[r] = hash;
        }
        const uint32_t n = counts[r];
        if (n == 0) continue;
        float* pr = probs + (uint64_t)r * probs_stride;
        const uint64_t* h = hits + (uint64_t)r * hits_stride;
        for (uint32_t i = 0; i < n; ++i) { const float d = (float)hit_delta(h[i]); pr[i] = 1.0f / (d * d); }
        if (top_seed) pr[0] = 0.0f;
        st_setup(pr, n, st_padded(n));
    }
}

static uint32_t randomized_select(const float* pr, uint32_t n, uint32_t padded, const uint64_t* h, uint32_t* rseed)
{
    for (uint32_t i = 0; i < 10; ++i) {
        const uint32_t ri = 1664525u * *rseed + 1013904223u;
        *rseed = ri;
        const float rf = (float)ri / (float)0xFFFFFFFFu;
        const uint32_t id = st_sample(pr, n, padded, rf);
        if (hit_delta(h[id]) != 0) return id;
    }
    return 0;
}

/* one selection round.  out arrays sized for n_active reads / n_active * n_multi hits; out_sizes = {reads kept, hits}.
 * NOTE erase() inside the reference's selection kernels is undone by ~SeedHitDequeReference (seed_hit_deque_array.h:
 * 263-268 writes the size captured at construction back after hits.erase()), so a read's count survives; the read
 * still leaves the active queue because nothing is emitted for it. */
ORACLE_API void oracle_select(int randomized, uint32_t n_multi, const uint32_t* active_in, uint32_t n_active,
    uint64_t* hits, uint32_t hits_stride, uint32_t* counts, float* probs, uint32_t probs_stride, uint32_t* rseeds, const uint32_t* trys,
    uint32_t* active_out, uint64_t* hit_begin, uint32_t* hit_read_id, uint32_t* hit_loc, uint32_t* hit_seed, uint32_t* out_sizes)
{
    uint32_t n_out = 0; uint64_t n_hits = 0;
    for (uint32_t t = 0; t < n_active; ++t)
    {
        const uint32_t read_id = active_in[t] & 0x7FFFFFFFu;
        uint32_t top_flag = active_in[t] >> 31;
        if (trys[read_id] == 0) continue;                                      /* SelectBestApproxContext::stop */
        uint32_t n = counts[read_id];
        if (n == 0) continue;
        uint64_t* h = hits + (uint64_t)read_id * hits_stride;
        const uint64_t first = n_hits;
        ih_make(h, n);                                                         /* hits[ read_id ] (select_inl.h:102,212,318,519) */
        if (!randomized)
        {
            /* select_kernel (:74-137) and select_multi_kernel (:281-480): walk the deque from its top */
            for (uint32_t i = 0; i < n_multi; ++i)
            {
                uint32_t top = (n == 1) ? 0u : 1u;
                if (hit_delta(h[top]) == 0) {
                    ih_pop_top(h, n); --n;
                    if (n == 0) break;
                    top = (n == 1) ? 0u : 1u;
                    top_flag = 0u;
                }
                const uint32_t sa_pos = hit_pop_front(&h[top]);
                hit_read_id[n_hits] = read_id; hit_loc[n_hits] = sa_pos; hit_seed[n_hits] = packed_seed_of(h[top], top_flag);
                ++n_hits;
            }
            counts[read_id] = n;
        }
        else
        {
            /* rand_select_kernel (:176-264) and rand_select_multi_kernel (:496-607) */
            float* pr = probs + (uint64_t)read_id * probs_stride;
            const uint32_t padded = st_padded(n);
            for (uint32_t i = 0; i < n_multi; ++i)
            {
                if (st_sum(pr, padded) <= 0.0f) break;
                if (top_flag && hit_delta(h[0]) == 0) top_flag = 0;
                const uint32_t id = top_flag ? 0u : randomized_select(pr, n, padded, h, &rseeds[read_id]);
                if (hit_delta(h[id]) == 0) { if (n_multi > 1) continue; else break; }
                const uint32_t sa_pos = hit_pop_front(&h[id]);
                if (hit_delta(h[id]) == 0) st_set(pr, padded, id, 0.0f);
                hit_read_id[n_hits] = read_id; hit_loc[n_hits] = sa_pos; hit_seed[n_hits] = packed_seed_of(h[id], top_flag);
                ++n_hits;
            }
        }
        if (n_hits > first) {
            hit_begin[n_out] = first;
            active_out[n_out] = read_id | (top_flag << 31);
            ++n_out;
        }
    }
    hit_begin[n_out] = n_hits;
    out_sizes[0] = n_out; out_sizes[1] = (uint32_t)n_hits;
}

/* locate_kernel (locate_inl.h:122-143): SA row -> read start in genome coordinates (may wrap below zero) */
ORACLE_API void oracle_locate_hits(const oracle_fmi_t* f, const oracle_fmi_t* rf, uint32_t n, uint32_t* hit_loc, const uint32_t* hit_seed)
{
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t dir = (hit_seed[i] >> 12) & 1u, pir = hit_seed[i] & 0xFFFu;
        const uint32_t g = dir ? rf->length - 1u - fm_locate(rf, hit_loc[i], NULL) : fm_locate(f, hit_loc[i], NULL);
        hit_loc[i] = g - pir;
    }
}

/* BestScoreStream::init_context (score_best_inl.h:95-126): the genome window and the score threshold of every hit.
 * A window that starts at or beyond its end (a wrapped read start) is given length 0 -- the reference would take the
 * wrapped difference as a length and read out of bounds. */
ORACLE_API void oracle_score_best_setup(uint32_t n, const uint32_t* hit_read_id, const uint32_t* hit_loc,
    const uint32_t* read_len, uint32_t band_len, uint32_t genome_len, const uint64_t* best, uint32_t best_stride, int32_t score_limit,
    uint64_t* text_begin, uint32_t* text_len, int32_t* min_score)
{
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t r = hit_read_id[i], g = hit_loc[i];
        const uint32_t gb = g > band_len / 2 ? g - band_len / 2 : 0u;
        const uint32_t sum = gb + band_len + read_len[r];                      /* uint32 arithmetic as in the reference */
        const uint32_t ge = sum < genome_len ? sum : genome_len;
        text_begin[i] = gb; text_len[i] = ge > gb ? ge - gb : 0u;
        const io_aln_t a2 = { (uint32_t)best[r + best_stride], (uint32_t)(best[r + best_stride] >> 32) };
        const int32_t s2 = io_aln_score(a2);
        min_score[i] = s2 > score_limit ? s2 : score_limit;
    }
}

/* score_reduce_kernel with ReduceBestApproxContext.  hit_score is the raw DP result: BestScoreStream::output clamps it to
 * worst_score first (score_best_inl.h:139).  active = packed_read words. */
ORACLE_API void oracle_score_reduce_best_approx(uint32_t n_active, const uint32_t* active, const uint64_t* hit_begin,
    const int32_t* hit_score, const uint32_t* hit_loc, const uint32_t* hit_seed, const uint32_t* read_len,
    uint64_t* best, uint32_t best_stride, int32_t worst_score,
    uint32_t* trys, uint32_t* counts, uint32_t n_ext, uint32_t min_ext, uint32_t max_ext, uint32_t max_effort)
{
    for (uint32_t t = 0; t < n_active; ++t)
    {
        const uint32_t read_id = active[t] & 0x7FFFFFFFu;
        io_aln_t a1 = { (uint32_t)best[read_id], (uint32_t)(best[read_id] >> 32) };
        io_aln_t a2 = { (uint32_t)best[read_id + best_stride], (uint32_t)(best[read_id + best_stride] >> 32) };
        const uint32_t len = read_len[read_id];
        const uint64_t hb = hit_begin[t];
        for (uint64_t i = hb; i < hit_begin[t + 1]; ++i)
        {
            const int32_t score = hit_score[i] > worst_score ? hit_score[i] : worst_score;
            const uint32_t g_pos = hit_loc[i], rc = (hit_seed[i] >> 13) & 1u, top_flag = (hit_seed[i] >> 14) & 1u;
            if ((rc == io_aln_rc(a1) && g_pos == a1.align) || (rc == io_aln_rc(a2) && g_pos == a2.align)) continue;
            if (score > io_aln_score(a1)) { trys[read_id] = max_effort; a2 = a1; a1 = io_aln_make(g_pos, 0u, score, rc); }
            else if (score > io_aln_score(a2) && distinct_alignments(a1.align, io_aln_rc(a1), g_pos, rc, len / 2)) {
                trys[read_id] = max_effort; a2 = io_aln_make(g_pos, 0u, score, rc);
            }
            else if (trys[read_id] > 0) {                                       /* ReduceBestApproxContext::failure, reduce.h:90-100 */
                const uint32_t idx = (uint32_t)(i - hb);
                if (((n_ext + idx >= min_ext) && top_flag == 0 && --trys[read_id] == 0) || (n_ext + idx >= max_ext))
                    counts[read_id] = 0;                                        /* hits.erase(read_id) */
            }
        }
        best[read_id] = ((uint64_t)a1.align << 32) | a1.w;
        best[read_id + best_stride] = ((uint64_t)a2.align << 32) | a2.w;
    }
}

/* ------------------------------------------------------------------------ */
/* paired-end reduction: score_reduce_paired_kernel (reduce_inl.h:355-500),   */
/* try_update / update_best / replace_best / update_second (:160-350),        */
/* frame_opposite_mate (alignment_utils.h:61-98), io::PairedAlignments /      */
/* BestPairedAlignments (alignments.h:180-330), the paired                    */
/* distinct_alignments (alignments_inl.h:66-80,121-136)                       */
/* ------------------------------------------------------------------------ */
static inline io_aln_t io_aln_make_full(uint32_t pos, uint32_t ed, int32_t score, uint32_t rc, uint32_t mate, int paired)
{
    io_aln_t a = io_aln_make(pos, ed, score, rc);
    a.w |= ((mate & 1u) << 29) | ((paired ? 1u : 0u) << 30);
    return a;
}
static inline uint32_t io_aln_sink(io_aln_t a)   { return (a.w >> 18) & 0x3FFu; }
static inline uint32_t io_aln_mate(io_aln_t a)   { return (a.w >> 29) & 1u; }
static inline int      io_aln_paired(io_aln_t a) { return ((a.w >> 30) & 1u) && io_aln_aligned(a); }       /* is_paired() */
static inline int      io_aln_unpaired(io_aln_t a) { return !((a.w >> 30) & 1u) && io_aln_aligned(a); }
typedef struct { io_aln_t a, o; } io_pair_t;
typedef struct { io_aln_t a1, a2, o1, o2; } io_best_pairs_t;
static inline io_aln_t pair_mate(const io_pair_t* p, uint32_t m) { return m == io_aln_mate(p->a) ? p->a : p->o; }
static inline int32_t  pair_score(const io_pair_t* p) { return io_aln_score(p->a) + io_aln_score(p->o); }
static inline int      bp_is_paired(const io_best_pairs_t* b) { return io_aln_paired(b->a1); }
static inline int      bp_has_second_paired(const io_best_pairs_t* b) { return io_aln_paired(b->a2); }
static inline int32_t  bp_best_score(const io_best_pairs_t* b) { return io_aln_score(b->a1) + (bp_is_paired(b) ? io_aln_score(b->o1) : 0); }
static inline int32_t  bp_second_score(const io_best_pairs_t* b) { return io_aln_score(b->a2) + (bp_has_second_paired(b) ? io_aln_score(b->o2) : 0); }
static int distinct_pairs(const io_pair_t* p1, const io_pair_t* p2, uint32_t dist)
{
    const io_aln_t a1 = pair_mate(p1, 0), o1 = pair_mate(p1, 1), a2 = pair_mate(p2, 0), o2 = pair_mate(p2, 1);
    const uint32_t apos1 = a1.align + io_aln_sink(a1), opos1 = o1.align + io_aln_sink(o1);
    const uint32_t apos2 = a2.align + io_aln_sink(a2), opos2 = o2.align + io_aln_sink(o2);
    if (io_aln_rc(a1) != io_aln_rc(a2) || io_aln_rc(o1) != io_aln_rc(o2)) return 1;
    return ((apos1 >= apos2 - (apos2 < dist ? apos2 : dist) && apos1 <= apos2 + dist) &&
            (opos1 >= opos2 - (opos2 < dist ? opos2 : dist) && opos1 <= opos2 + dist)) ? 0 : 1;
}
static int distinct_alns(io_aln_t p1, io_aln_t p2, uint32_t dist)
{
    return distinct_alignments(p1.align + io_aln_sink(p1), io_aln_rc(p1), p2.align + io_aln_sink(p2), io_aln_rc(p2), dist);
}
/* tr (nullable): the read's try counter -- ReduceBestApproxContext::best_score / second_score refill it on every update */
static int try_update_pair_ctx(io_best_pairs_t* b, const io_pair_t* pair, uint32_t min_distance, uint32_t* tr, uint32_t max_effort)
{
    const int32_t score = pair_score(pair);
    const io_pair_t p0 = { b->a1, b->o1 }, p1 = { b->a2, b->o2 };
    #define HOOK() do { if (tr) *tr = max_effort; } while (0)
    if (!distinct_pairs(&p0, pair, min_distance)) {
        if (score > bp_best_score(b)) { HOOK(); b->a1 = pair->a; b->o1 = pair->o; }                         /* replace_best */
        return 1;
    } else if (!distinct_pairs(&p1, pair, min_distance)) {
        if (score > bp_best_score(b)) { HOOK(); b->a2 = b->a1; b->o2 = b->o1; b->a1 = pair->a; b->o1 = pair->o; }     /* update_best */
        else if (score > bp_second_score(b)) { HOOK(); b->a2 = pair->a; b->o2 = pair->o; }                  /* update_second */
        return 1;
    } else if (!bp_is_paired(b) || score > bp_best_score(b)) {
        HOOK(); b->a2 = b->a1; b->o2 = b->o1; b->a1 = pair->a; b->o1 = pair->o; return 1;
    } else if (!bp_has_second_paired(b) || score > bp_second_score(b)) {
        HOOK(); b->a2 = pair->a; b->o2 = pair->o; return 1;
    }
    return 0;
}
static int try_update_single_ctx(io_aln_t* a1, io_aln_t* a2, io_aln_t a, uint32_t min_distance, uint32_t* tr, uint32_t max_effort)
{
    if (!distinct_alns(*a1, a, min_distance)) { if (io_aln_score(a) > io_aln_score(*a1)) { HOOK(); *a1 = a; } return 1; }
    else if (!distinct_alns(*a2, a, min_distance)) {
        if (io_aln_score(a) > io_aln_score(*a1)) { HOOK(); *a2 = *a1; *a1 = a; }
        else if (io_aln_score(a) > io_aln_score(*a2)) { HOOK(); *a2 = a; }
        return 1;
    }
    else if (io_aln_score(a) > io_aln_score(*a1)) { HOOK(); *a2 = *a1; *a1 = a; return 1; }
    else if (io_aln_score(a) > io_aln_score(*a2)) { HOOK(); *a2 = a; return 1; }
    return 0;
    #undef HOOK
}
static int try_update_pair(io_best_pairs_t* b, const io_pair_t* pair, uint32_t min_distance) { return try_update_pair_ctx(b, pair, min_distance, NULL, 0); }
static int try_update_single(io_aln_t* a1, io_aln_t* a2, io_aln_t a, uint32_t min_distance) { return try_update_single_ctx(a1, a2, a, min_distance, NULL, 0); }
static void frame_opposite_mate(int policy, uint32_t anchor, int anchor_fw, int* left, int* fw)
{   /* io::PairedEndPolicy: FF = 0, FR = 1, RF = 2, RR = 3 (nvbio/io/sequence/sequence.h:190-196) */
    const int anchor_1 = (anchor == 0);
    switch (policy) {
    case 0:  *left = (anchor_1 != anchor_fw); *fw = anchor_fw;  break;     /* FF */
    case 3:  *left = (anchor_1 == anchor_fw); *fw = anchor_fw;  break;     /* RR */
    case 1:  *left = !anchor_fw;              *fw = !anchor_fw; break;     /* FR */
    default: *left = anchor_fw;               *fw = !anchor_fw; break;     /* RF */
    }
}
/* hit arrays per extension result: loc, sink (genome end of the anchor), score, rc, opposite_loc, opposite_sink,
 * opposite_sink2, opposite_score, opposite_score2; best / best_o: [2][best_stride] io::Alignment words */
ORACLE_API void oracle_score_reduce_paired(uint32_t n_active, const uint32_t* read_ids, const uint64_t* hit_begin,
    const uint32_t* hit_loc, const uint32_t* hit_sink, const int32_t* hit_score, const uint8_t* hit_rc,
    const uint32_t* o_loc, const uint32_t* o_sink, const uint32_t* o_sink2, const int32_t* o_score, const int32_t* o_score2,
    const uint32_t* read_len, uint32_t anchor, int pe_policy, int pe_unpaired, int32_t score_limit,
    uint64_t* best, uint64_t* best_o, uint32_t best_stride)
{
    #define LD(p, i) ((io_aln_t){ (uint32_t)(p)[i], (uint32_t)((p)[i] >> 32) })
    #define ST(p, i, a) ((p)[i] = ((uint64_t)(a).align << 32) | (a).w)
    for (uint32_t t = 0; t < n_active; ++t)
    {
        const uint32_t read_id = read_ids ? read_ids[t] : t;
        io_best_pairs_t b = { LD(best, read_id), LD(best, read_id + best_stride), LD(best_o, read_id), LD(best_o, read_id + best_stride) };
        const uint32_t min_distance = read_len[read_id] / 4;
        for (uint64_t i = hit_begin[t]; i < hit_begin[t + 1]; ++i)
        {
            const uint32_t rc = hit_rc[i];
            int o_left, o_fw;
            frame_opposite_mate(pe_policy, anchor, !rc, &o_left, &o_fw);
            const uint32_t o_rc = !o_fw;
            const io_pair_t pair  = { io_aln_make_full(hit_loc[i], hit_sink[i] - hit_loc[i], hit_score[i], rc, anchor, o_score[i] > score_limit),
                                      io_aln_make_full(o_loc[i], o_sink[i] - o_loc[i], o_score[i], o_rc, !anchor, o_score[i] > score_limit) };
            const io_pair_t pair2 = { io_aln_make_full(hit_loc[i], hit_sink[i] - hit_loc[i], hit_score[i], rc, anchor, o_score2[i] > score_limit),
                                      io_aln_make_full(o_loc[i], o_sink2[i] - o_loc[i], o_score2[i], o_rc, !anchor, o_score2[i] > score_limit) };
            if (io_aln_paired(pair.a)) {
                try_update_pair(&b, &pair, min_distance);
                if (io_aln_paired(pair2.a)) try_update_pair(&b, &pair2, min_distance);
            } else if (pe_unpaired && !bp_is_paired(&b)) {
                if (anchor) try_update_single(&b.o1, &b.o2, pair.a, min_distance);
                else        try_update_single(&b.a1, &b.a2, pair.a, min_distance);
            }
        }
        ST(best, read_id, b.a1); ST(best, read_id + best_stride, b.a2); ST(best_o, read_id, b.o1); ST(best_o, read_id + best_stride, b.o2);
    }
    #undef LD
    #undef ST
}

/* BestOppositeScoreStream::init_context (score_opposite_inl.h:92-200): for every anchor hit, the opposite mate's strand,
 * genome window and score threshold; compute_target_score (alignment_utils.h:100-111); max_text_gaps (utils_inl.h:181-204).
 * limits: {match, min_type, text_gap_open, text_gap_ext}, min_k / min_m the threshold function's coefficients */
typedef struct { int32_t pe_policy, min_frag_len, max_frag_len, pe_overlap, score_limit; uint32_t anchor, genome_length; } oracle_pe_params_t;
ORACLE_API void oracle_opposite_windows(uint32_t n_hits, const uint32_t* hit_read_id, const uint8_t* hit_rc, const uint32_t* hit_loc, const int32_t* hit_score,
    const uint32_t* a_read_len, const uint32_t* o_read_len, const uint64_t* best, const uint64_t* best_o, uint32_t best_stride,
    int32_t match, int min_type, float min_k, float min_m, int32_t text_gap_open, int32_t text_gap_ext, const oracle_pe_params_t* pp,
    uint8_t* out_valid, int32_t* out_min_score, uint8_t* out_read_rc, uint32_t* out_genome_begin, uint32_t* out_genome_end)
{
    for (uint32_t i = 0; i < n_hits; ++i)
    {
        const uint32_t read_rc = hit_rc[i], read_id = hit_read_id[i], g_pos = hit_loc[i];
        const uint32_t a_len = a_read_len[read_id], o_len = o_read_len[read_id];
        const int32_t a_optimal = (int32_t)a_len * match, a_worst = simple_func(min_type, min_k, min_m, (int32_t)a_len);
        const int32_t o_optimal = (int32_t)o_len * match, o_worst = simple_func(min_type, min_k, min_m, (int32_t)o_len);
        const io_best_pairs_t b = { { (uint32_t)best[read_id], (uint32_t)(best[read_id] >> 32) }, { (uint32_t)best[read_id + best_stride], (uint32_t)(best[read_id + best_stride] >> 32) },
                                    { (uint32_t)best_o[read_id], (uint32_t)(best_o[read_id] >> 32) }, { (uint32_t)best_o[read_id + best_stride], (uint32_t)(best_o[read_id + best_stride] >> 32) } };
        int32_t target;                                                         /* compute_target_score */
        if (!bp_has_second_paired(&b)) target = a_worst + o_worst;
        else { const int32_t delta = bp_best_score(&b) - bp_second_score(&b); target = bp_second_score(&b) + (delta * 3) / 4; }
        int32_t target_pair = target + 1; if (a_optimal + o_optimal < target_pair) target_pair = a_optimal + o_optimal;
        int32_t target_mate = target_pair - hit_score[i];
        if (target_mate < o_worst) target_mate = o_worst;
        const int32_t min_score = target_mate > pp->score_limit ? target_mate : pp->score_limit;
        out_min_score[i] = min_score;
        out_valid[i] = 0; out_read_rc[i] = 0; out_genome_begin[i] = out_genome_end[i] = 0;
        if (min_score > o_optimal) continue;
        int o_left, o_fw;
        frame_opposite_mate(pp->pe_policy, pp->anchor, !read_rc, &o_left, &o_fw);
        out_read_rc[i] = (uint8_t)!o_fw;
        int32_t max_ref_gaps;                                                   /* max_text_gaps, uint32 result taken as int32 */
        {
            int32_t score = (int32_t)o_len * match;
            if (score < min_score) max_ref_gaps = 0;
            else {
                score += text_gap_open;
                uint32_t n = 0;
                while (score >= min_score && n < o_len) { score += text_gap_ext; ++n; }
                max_ref_gaps = (int32_t)(n - 1u);
            }
        }
        const uint32_t o_gapped_len = o_len + (uint32_t)max_ref_gaps;
        const uint32_t min_frag = (uint32_t)pp->min_frag_len, max_frag = (uint32_t)pp->max_frag_len;
        uint32_t gb, ge;
        if (o_left) {
            const uint32_t max_end = g_pos + a_len + o_gapped_len > min_frag ? g_pos + a_len + o_gapped_len - min_frag : 0u;
            gb = g_pos + a_len > max_frag ? (g_pos + a_len) - max_frag : 0u;
            ge = pp->pe_overlap ? g_pos + a_len : g_pos;
            if (max_end < ge) ge = max_end;
        } else {
            const uint32_t min_begin = g_pos + min_frag > o_gapped_len ? g_pos + min_frag - o_gapped_len : 0u;
            ge = g_pos + max_frag;
            gb = pp->pe_overlap ? g_pos : g_pos + a_len;
            if (min_begin > gb) gb = min_begin;
        }
        if (ge > pp->genome_length) ge = pp->genome_length;
        out_genome_begin[i] = gb; out_genome_end[i] = ge;
        if (gb >= pp->genome_length) continue;
        const uint32_t mate = pp->anchor ? 0u : 1u, rrc = (uint32_t)!o_fw;
        const int skip = (mate == io_aln_mate(b.a1) && rrc == io_aln_rc(b.a1) && g_pos == b.a1.align) ||
                         (mate == io_aln_mate(b.o1) && rrc == io_aln_rc(b.o1) && g_pos == b.o1.align) ||
                         (mate == io_aln_mate(b.a2) && rrc == io_aln_rc(b.a2) && g_pos == b.a2.align) ||
                         (mate == io_aln_mate(b.o2) && rrc == io_aln_rc(b.o2) && g_pos == b.o2.align) || (gb == ge);
        out_valid[i] = (uint8_t)!skip;
    }
}

static int32_t simple_func(int type, float k, float m, int32_t x)
{
    return (int32_t)(k + m * (type == 1 ? logf((float)x) : type == 2 ? sqrtf((float)x) : (float)x));
}
static inline int clamp10(int v) { return v < 0 ? 0 : v > 10 ? 10 : v; }

static uint32_t mapq_v3(int32_t best_score, int has_second, int32_t second_score, float max_score, float min_score, int is_paired)
{
    if (is_paired) return 44;        /* paired_one_perfect: mapq.h:96-102 (before any threshold test) */
    static const int unpaired_one[11] = { 43, 42, 41, 36, 32, 27, 20, 11, 4, 1, 0 };
    static const int unpaired_two_perfect[11] = { 2, 16, 23, 30, 31, 32, 34, 36, 38, 40, 42 };
    static const int unpaired_two[11][11] = {
        {  2,  2,  2,  1,  1, 0, 0, 0, 0, 0, 0 }, { 20, 14,  7,  3,  2, 1, 0, 0, 0, 0, 0 }, { 20, 16, 10,  6,  3, 1, 0, 0, 0, 0, 0 },
        { 20, 17, 13,  9,  3, 1, 1, 0, 0, 0, 0 }, { 21, 19, 15,  9,  5, 2, 2, 0, 0, 0, 0 }, { 22, 21, 16, 11, 10, 5, 0, 0, 0, 0, 0 },
        { 23, 22, 19, 16, 11, 0, 0, 0, 0, 0, 0 }, { 24, 25, 21, 30,  0, 0, 0, 0, 0, 0, 0 }, { 30, 26, 29,  0,  0, 0, 0, 0, 0, 0, 0 },
        { 30, 27,  0,  0,  0, 0, 0, 0, 0, 0, 0 }, { 30,  0,  0,  0,  0, 0, 0, 0, 0, 0, 0 } };
    const float norm_factor = 10.0f / (max_score - min_score);
    if ((float)best_score < min_score) return 0;
    const int best = ((int)max_score - best_score) > 0 ? ((int)max_score - best_score) : 0;
    const int best_bin = clamp10((int)((float)best * norm_factor + 0.5f));
    if (has_second) {
        const int diff = best_score - second_score;
        const int diff_bin = clamp10((int)((float)diff * norm_factor + 0.5f));
        return ((float)best == max_score) ? (uint32_t)unpaired_two_perfect[best_bin] : (uint32_t)unpaired_two[diff_bin][best_bin];
    }
    return ((float)best == max_score) ? 44u : (uint32_t)unpaired_one[best_bin];
}
static uint32_t mapq_v2(int32_t best_score, int has_second, int32_t second_score, float max_score, float min_score, int monotone)
{
    const float diff = max_score - min_score, best = (float)best_score;
    if (best < min_score) return 0;
    const float best_over = best - min_score;
    if (monotone) {
        if (!has_second) {
            if (best_over >= diff * 0.8f) return 42; if (best_over >= diff * 0.7f) return 40; if (best_over >= diff * 0.6f) return 24;
            if (best_over >= diff * 0.5f) return 23; if (best_over >= diff * 0.4f) return 8;  if (best_over >= diff * 0.3f) return 3;
            return 0;
        }
        const float best_diff = fabsf(fabsf(best) - fabsf((float)second_score));
        if (best_diff >= diff * 0.9f) return (best_over == diff) ? 39 : 33;
        if (best_diff >= diff * 0.8f) return (best_over == diff) ? 38 : 27;
        if (best_diff >= diff * 0.7f) return (best_over == diff) ? 37 : 26;
        if (best_diff >= diff * 0.6f) return (best_over == diff) ? 36 : 22;
        if (best_diff >= diff * 0.5f) { if (best_over == diff) return 35; if (best_over >= diff * 0.84f) return 25; if (best_over >= diff * 0.68f) return 16; return 5; }
        if (best_diff >= diff * 0.4f) { if (best_over == diff) return 34; if (best_over >= diff * 0.84f) return 21; if (best_over >= diff * 0.68f) return 14; return 4; }
        if (best_diff >= diff * 0.3f) { if (best_over == diff) return 32; if (best_over >= diff * 0.88f) return 18; if (best_over >= diff * 0.67f) return 15; return 3; }
        if (best_diff >= diff * 0.2f) { if (best_over == diff) return 31; if (best_over >= diff * 0.88f) return 17; if (best_over >= diff * 0.67f) return 11; return 0; }
        if (best_diff >= diff * 0.1f) { if (best_over == diff) return 30; if (best_over >= diff * 0.88f) return 12; if (best_over >= diff * 0.67f) return 7;  return 0; }
        if (best_diff > 0) return (best_over >= diff * 0.67f) ? 6 : 2;
        return (best_over >= diff * 0.67f) ? 1 : 0;
    }
    if (!has_second) {
        if (best_over >= diff * 0.8f) return 44; if (best_over >= diff * 0.7f) return 42; if (best_over >= diff * 0.6f) return 41;
        if (best_over >= diff * 0.5f) return 36; if (best_over >= diff * 0.4f) return 28; if (best_over >= diff * 0.3f) return 24;
        return 22;
    }
    const float best_diff = fabsf(fabsf(best) - fabsf((float)second_score));
    if (best_diff >= diff * 0.9f) return 40; if (best_diff >= diff * 0.8f) return 39; if (best_diff >= diff * 0.7f) return 38; if (best_diff >= diff * 0.6f) return 37;
    if (best_diff >= diff * 0.5f) { if (best_over == diff) return 35; return (best_over >= diff * 0.50f) ? 25 : 20; }
    if (best_diff >= diff * 0.4f) { if (best_over == diff) return 34; return (best_over >= diff * 0.50f) ? 21 : 19; }
    if (best_diff >= diff * 0.3f) { if (best_over == diff) return 33; return (best_over >= diff * 0.5f) ? 18 : 16; }
    if (best_diff >= diff * 0.2f) { if (best_over == diff) return 32; return (best_over >= diff * 0.5f) ? 17 : 12; }
    if (best_diff >= diff * 0.1f) { if (best_over == diff) return 31; return (best_over >= diff * 0.5f) ? 14 : 9; }
    if (best_diff > 0) return (best_over >= diff * 0.5f) ? 11 : 2;
    return (best_over >= diff * 0.5f) ? 1 : 0;
}
/* single-end reads: out[r] = mapq(BestPairedAlignments(best of read r), read_len[r]) for EVERY read, aligned or not, as
 * MapqFunctorSE does (aligner_best_approx.h:62-76): an unaligned read carries Alignment::invalid() (score 2^17 - 1), for which the
 * calculator returns its top no-second value; the reference's writers zero it later (output_sam.cpp:462, output_bam.cpp:317).
 * Pinned against the reference's compiled mapq.h / functors (oracle/_ref/libref_mapq.so, tests/test_ref_mapq.py).
 * scheme: match bonus, min-score SimpleFunc (type,k,m), monotone flag (scoring.h:272-281,347) */
ORACLE_API void oracle_mapq(int version, int32_t match, int min_type, float min_k, float min_m, int monotone,
    uint32_t n_reads, const uint64_t* best, uint32_t best_stride, const uint32_t* read_len, uint8_t* out)
{
    for (uint32_t r = 0; r < n_reads; ++r)
    {
        const io_aln_t a1 = { (uint32_t)best[r], (uint32_t)(best[r] >> 32) }, a2 = { (uint32_t)best[r + best_stride], (uint32_t)(best[r + best_stride] >> 32) };
        const float max_score = (float)((int32_t)read_len[r] * match), min_score = (float)simple_func(min_type, min_k, min_m, (int32_t)read_len[r]);
        out[r] = (uint8_t)(version == 3 ? mapq_v3(io_aln_score(a1), io_aln_aligned(a2), io_aln_score(a2), max_score, min_score, 0)
                                        : mapq_v2(io_aln_score(a1), io_aln_aligned(a2), io_aln_score(a2), max_score, min_score, monotone));
    }
}

/* ------------------------------------------------------------------------ */
/* The per-round stages of nvBowtie's PAIRED best-approx loop                  */
/*   (aligner_best_approx_paired.h:455-700)                                   */
/*   BestAnchorScoreStream          score_paired_inl.h:54-150                 */
/*   score_reduce_paired_kernel + ReduceBestApproxContext  reduce_inl.h:355-480 */
/*   mark_discordant_kernel         aligner_init.cu:457-480                   */
/* ------------------------------------------------------------------------ */
static int32_t target_pair_score(const io_best_pairs_t* b, int32_t a_worst, int32_t o_worst, int32_t a_optimal, int32_t o_optimal)
{
    int32_t target;                                                             /* compute_target_score, alignment_utils.h:100-111 */
    if (!bp_has_second_paired(b)) target = a_worst + o_worst;
    else { const int32_t delta = bp_best_score(b) - bp_second_score(b); target = bp_second_score(b) + (delta * 3) / 4; }
    const int32_t t1 = target + 1, t2 = a_optimal + o_optimal;
    return t1 < t2 ? t1 : t2;
}
/* BestAnchorScoreStream::init_context: the anchor hit's genome window and score threshold.  A hit at a location already
 * recorded in the read's best pairs is skipped: threshold INT32_MAX and (here) an empty window, so that the DP fails and
 * the hit scores worst_score.  The reference's skip test also reads context->min_score before it is set (an
 * uninitialised read, score_paired_inl.h:128); that term is taken as false. */
ORACLE_API void oracle_anchor_score_setup(uint32_t n, const uint32_t* hit_read_id, const uint32_t* hit_loc, const uint32_t* hit_seed,
    const uint32_t* a_read_len, const uint32_t* o_read_len, uint32_t band_len, uint32_t genome_len,
    const uint64_t* best, const uint64_t* best_o, uint32_t best_stride, int32_t match, int min_type, float min_k, float min_m,
    int32_t score_limit, uint32_t anchor, uint64_t* text_begin, uint32_t* text_len, int32_t* min_score)
{
    for (uint32_t i = 0; i < n; ++i)
    {
        const uint32_t read_id = hit_read_id[i], g_pos = hit_loc[i], read_rc = (hit_seed[i] >> 13) & 1u;
        const uint32_t a_len = a_read_len[read_id], o_len = o_read_len[read_id];
        const int32_t a_optimal = (int32_t)a_len * match, a_worst = simple_func(min_type, min_k, min_m, (int32_t)a_len);
        const int32_t o_optimal = (int32_t)o_len * match, o_worst = simple_func(min_type, min_k, min_m, (int32_t)o_len);
        const io_best_pairs_t b = { { (uint32_t)best[read_id], (uint32_t)(best[read_id] >> 32) }, { (uint32_t)best[read_id + best_stride], (uint32_t)(best[read_id + best_stride] >> 32) },
                                    { (uint32_t)best_o[read_id], (uint32_t)(best_o[read_id] >> 32) }, { (uint32_t)best_o[read_id + best_stride], (uint32_t)(best_o[read_id + best_stride] >> 32) } };
        int32_t target_mate = target_pair_score(&b, a_worst, o_worst, a_optimal, o_optimal) - o_optimal;
        if (target_mate < a_worst) target_mate = a_worst;
        const uint32_t gb = g_pos > band_len / 2 ? g_pos - band_len / 2 : 0u;
        const uint32_t sum = gb + band_len + a_len;
        const uint32_t ge = sum < genome_len ? sum : genome_len;
        const int skip = (anchor == io_aln_mate(b.a1) && read_rc == io_aln_rc(b.a1) && g_pos == b.a1.align) ||
                         (anchor == io_aln_mate(b.o1) && read_rc == io_aln_rc(b.o1) && g_pos == b.o1.align) ||
                         (anchor == io_aln_mate(b.a2) && read_rc == io_aln_rc(b.a2) && g_pos == b.a2.align) ||
                         (anchor == io_aln_mate(b.o2) && read_rc == io_aln_rc(b.o2) && g_pos == b.o2.align);
        text_begin[i] = gb;
        text_len[i] = (skip || ge <= gb) ? 0u : ge - gb;
        min_score[i] = skip ? INT32_MAX : (target_mate > score_limit ? target_mate : score_limit);
    }
}
/* BestAnchorScoreStream::output */
ORACLE_API void oracle_anchor_score_finish(uint32_t n, const int32_t* raw_score, const uint32_t* raw_sink /* 2n */, const uint64_t* text_begin,
    const int32_t* min_score, int32_t worst_score, int32_t* hit_score, uint32_t* hit_sink)
{
    for (uint32_t i = 0; i < n; ++i) {
        hit_score[i] = raw_score[i] >= min_score[i] ? raw_score[i] : worst_score;
        hit_sink[i] = (uint32_t)text_begin[i] + raw_sink[2 * i];
    }
}

ORACLE_API void oracle_score_reduce_paired_best_approx(uint32_t n_active, const uint32_t* active, const uint64_t* hit_begin,
    const uint32_t* hit_loc, const uint32_t* hit_sink, const int32_t* hit_score, const uint32_t* hit_seed,
    const uint32_t* o_loc, const uint32_t* o_sink, const uint32_t* o_sink2, const int32_t* o_score, const int32_t* o_score2,
    const uint32_t* read_len, uint32_t anchor, int pe_policy, int pe_unpaired, int32_t score_limit,
    uint64_t* best, uint64_t* best_o, uint32_t best_stride,
    uint32_t* trys, uint32_t* counts, uint32_t n_ext, uint32_t min_ext, uint32_t max_ext, uint32_t max_effort)
{
    #define LD(p, i) ((io_aln_t){ (uint32_t)(p)[i], (uint32_t)((p)[i] >> 32) })
    #define ST(p, i, a) ((p)[i] = ((uint64_t)(a).align << 32) | (a).w)
    for (uint32_t t = 0; t < n_active; ++t)
    {
        const uint32_t read_id = active[t] & 0x7FFFFFFFu;
        io_best_pairs_t b = { LD(best, read_id), LD(best, read_id + best_stride), LD(best_o, read_id), LD(best_o, read_id + best_stride) };
        const uint32_t min_distance = read_len[read_id] / 4;
        const uint64_t hb = hit_begin[t];
        for (uint64_t i = hb; i < hit_begin[t + 1]; ++i)
        {
            const uint32_t rc = (hit_seed[i] >> 13) & 1u, top_flag = (hit_seed[i] >> 14) & 1u;
            int o_left, o_fw;
            frame_opposite_mate(pe_policy, anchor, !rc, &o_left, &o_fw);
            const uint32_t o_rc = !o_fw;
            const io_pair_t pair  = { io_aln_make_full(hit_loc[i], hit_sink[i] - hit_loc[i], hit_score[i], rc, anchor, o_score[i] > score_limit),
                                      io_aln_make_full(o_loc[i], o_sink[i] - o_loc[i], o_score[i], o_rc, !anchor, o_score[i] > score_limit) };
            const io_pair_t pair2 = { io_aln_make_full(hit_loc[i], hit_sink[i] - hit_loc[i], hit_score[i], rc, anchor, o_score2[i] > score_limit),
                                      io_aln_make_full(o_loc[i], o_sink2[i] - o_loc[i], o_score2[i], o_rc, !anchor, o_score2[i] > score_limit) };
            int updated = 0;
            if (io_aln_paired(pair.a)) {
                if (try_update_pair_ctx(&b, &pair, min_distance, &trys[read_id], max_effort)) updated = 1;
                if (io_aln_paired(pair2.a)) { if (try_update_pair_ctx(&b, &pair2, min_distance, &trys[read_id], max_effort)) updated = 1; }
            } else if (pe_unpaired && !bp_is_paired(&b)) {
                if (anchor ? try_update_single_ctx(&b.o1, &b.o2, pair.a, min_distance, &trys[read_id], max_effort)
                           : try_update_single_ctx(&b.a1, &b.a2, pair.a, min_distance, &trys[read_id], max_effort)) updated = 1;
            }
            if (!updated && trys[read_id] > 0) {
                const uint32_t idx = (uint32_t)(i - hb);
                if (((n_ext + idx >= min_ext) && top_flag == 0 && --trys[read_id] == 0) || (n_ext + idx >= max_ext)) counts[read_id] = 0;
            }
        }
        ST(best, read_id, b.a1); ST(best, read_id + best_stride, b.a2); ST(best_o, read_id, b.o1); ST(best_o, read_id + best_stride, b.o2);
    }
    #undef LD
    #undef ST
}

ORACLE_API void oracle_mark_discordant(uint32_t n_reads, uint64_t* best, uint64_t* best_o, uint32_t stride)
{
    for (uint32_t r = 0; r < n_reads; ++r) {
        const uint32_t w = (uint32_t)best[r];
        const int concordant = ((w >> 30) & 1u) && !((w >> 31) & 1u);
        const int aligned = (best[r] >> 32) != 0xFFFFFFFFu, second = (best[r + stride] >> 32) != 0xFFFFFFFFu;
        const int o_aligned = (best_o[r] >> 32) != 0xFFFFFFFFu, o_second = (best_o[r + stride] >> 32) != 0xFFFFFFFFu;
        if (!concordant && aligned && !second && o_aligned && !o_second) { best[r] |= 0xC0000000ull; best_o[r] |= 0xC0000000ull; }
    }
}

/* ------------------------------------------------------------------------ */
/* finish_alignment_kernel (nvBowtie/bowtie2/cuda/traceback_inl.h:523-722):     */
/* from a best alignment's CIGAR, its MD string in nvbio's byte-coded form       */
/* (io::MDS_OP, nvbio/io/alignments.h:46-52: [len lo, len hi] then {MATCH, run   */
/* <= 255} / {MISMATCH, read symbol} / {INSERTION|DELETION, length byte, symbols */
/* ...}), its edit distance (mismatches + inserted / deleted symbols, clips not   */
/* counted) and its final score -- the sum of scoring_scheme.score() over the     */
/* SUBSTITUTION columns (scoring.h:301-311: N penalty if the read symbol is > 3,  */
/* else match bonus or the quality's mismatch penalty) MINUS, per INSERTION run   */
/* of l read symbols, cumulative_deletion(l) = ref_gap_const + ref_gap_coeff * l, */
/* and per DELETION run cumulative_insertion(l) (traceback_inl.h:664-665; round 4 */
/* found the two subtractions missing here by running nvBowtie's own compiled     */
/* finish_alignment_kernel beside this restatement) -- and the alignment          */
/* rewritten by BestTracebackStream::finish (:177-189): m_align = window begin,   */
/* m_ed, m_score.  CIGAR words are stored end first.  A CIGAR that overflowed its */
/* slots cannot be replayed: the job is skipped (mds_len 0, alignment untouched). */
/* ------------------------------------------------------------------------ */
ORACLE_API void oracle_finish_alignment(uint32_t n, const uint8_t* valid,
    const uint32_t* pw, uint32_t pbits, uint32_t pbe, const uint64_t* pbegin, const uint32_t* plen, const uint8_t* quals, uint64_t n_quals,
    const uint32_t* tw, uint32_t tbe, const uint64_t* tbegin, const uint32_t* tlen,
    const uint16_t* cigar, uint32_t cigar_stride, const uint32_t* cigar_len, const uint32_t* cigar_source /* 2n */,
    int32_t match, const int32_t* mismatch_lut, int32_t n_penalty, const int32_t* gap_costs /* pattern open, ext, text open, ext (<= 0) */,
    const uint32_t* idx /* nullable */, uint64_t* best, uint8_t* out_mds, uint32_t mds_stride, uint32_t* out_mds_len)
{
    for (uint32_t w = 0; w < n; ++w)
    {
        out_mds_len[w] = 0;
        if (!valid[w] || cigar_len[w] == 0 || cigar_len[w] > cigar_stride) continue;
        const uint16_t* cv = cigar + (uint64_t)w * cigar_stride;
        uint8_t* mds = out_mds + (uint64_t)w * mds_stride;
        #define MDS_PUSH(v) do { if (mds_len < mds_stride) mds[mds_len] = (uint8_t)(v); ++mds_len; } while (0)
        uint32_t mds_len = 2, mds_op = 4 /* MDS_INVALID */, ed = 0, last_run = 0 /* index of the open MATCH run's counter */;
        int32_t score = 0;
        uint32_t j = 0, k = cigar_source[2 * w] & 0xFFFFu;
        for (uint32_t i = 0; i < cigar_len[w]; ++i)
        {
            const uint32_t word = cv[cigar_len[w] - i - 1u], t = word & 3u, l = word >> 2;
            if (t != 0u) {                                   /* INSERTION / DELETION / SOFT_CLIPPING header */
                mds_op = (t == 2u) ? 3u : 2u;
                MDS_PUSH(mds_op); MDS_PUSH(l);
            }
            for (uint32_t x = 0; x < l; ++x)
            {
                j += (t != 2u) ? 1u : 0u;
                k += (t == 0u || t == 2u) ? 1u : 0u;
                const uint32_t readc = j > 0 ? ps_get(pw, pbits, pbe, pbegin[w] + j - 1) : 255u;
                const uint32_t refc  = k > 0 ? ps_get(tw, 2, tbe, tbegin[w] + k - 1) : 255u;
                if (t == 0u) {
                    if (readc == refc) {
                        if (mds_op == 0u && last_run < mds_stride && mds[last_run] < 255) mds[last_run]++;
                        else { mds_op = 0u; MDS_PUSH(0u); last_run = mds_len; MDS_PUSH(1u); }
                    } else { mds_op = 1u; MDS_PUSH(1u); MDS_PUSH(readc); ++ed; }
                    const uint32_t q = quals ? quals[(pbegin[w] + j - 1 < n_quals) ? pbegin[w] + j - 1 : n_quals - 1] : 0u;
                    const uint32_t ref_mask = (1u << (refc & 31u)) & 0xFFu;
                    score += (readc > 3u || ref_mask > 15u) ? -n_penalty : ((ref_mask & (1u << readc)) ? match : mismatch_lut[q]);
                } else {
                    MDS_PUSH(t == 2u ? refc : readc);
                    if (t != 3u) ++ed;
                }
            }
            if (t == 1u && l)      score += gap_costs[2] + (int32_t)(l - 1u) * gap_costs[3];      /* -cumulative_deletion(l): the reference gap (text) costs */
            else if (t == 2u && l) score += gap_costs[0] + (int32_t)(l - 1u) * gap_costs[1];      /* -cumulative_insertion(l): the read gap (pattern) costs */
        }
        #undef MDS_PUSH
        if (mds_stride >= 2) { mds[0] = (uint8_t)(mds_len & 0xFF); mds[1] = (uint8_t)(mds_len >> 8); }
        out_mds_len[w] = mds_len;
        const uint32_t r = idx ? idx[w] : w;
        uint32_t a = (uint32_t)best[r];
        const uint32_t mag = score < 0 ? (uint32_t)(-score) : (uint32_t)score;
        a = (a & 0xF0000000u) | (score < 0 ? 1u : 0u) | ((mag & 0x1FFFFu) << 1) | ((ed & 0x3FFu) << 18);
        best[r] = ((uint64_t)(uint32_t)tbegin[w] << 32) | a;
    }
}

/* paired-end reads: mapq(BestPairedAlignments(anchor best pair, opposite best pair), read_len, o_read_len) (mapq.h:56-58,150-166) */
ORACLE_API void oracle_mapq_paired(int version, int32_t match, int min_type, float min_k, float min_m, int monotone,
    uint32_t n_reads, const uint64_t* best, const uint64_t* best_o, uint32_t best_stride, const uint32_t* read_len, const uint32_t* o_read_len, uint8_t* out)
{
    for (uint32_t r = 0; r < n_reads; ++r)
    {
        const io_best_pairs_t b = { { (uint32_t)best[r], (uint32_t)(best[r] >> 32) }, { (uint32_t)best[r + best_stride], (uint32_t)(best[r + best_stride] >> 32) },
                                    { (uint32_t)best_o[r], (uint32_t)(best_o[r] >> 32) }, { (uint32_t)best_o[r + best_stride], (uint32_t)(best_o[r + best_stride] >> 32) } };
        const int paired = bp_is_paired(&b);
        const int has_second = paired ? bp_has_second_paired(&b) : io_aln_aligned(b.a2);
        if (version == 3) {
            const float max_score = (float)((int32_t)read_len[r] * match), min_score = (float)simple_func(min_type, min_k, min_m, (int32_t)read_len[r]);
            out[r] = (uint8_t)mapq_v3(bp_best_score(&b), has_second, bp_second_score(&b), max_score, min_score, paired);
        } else {
            const float max_score = (float)((int32_t)read_len[r] * match) + (paired ? (float)((int32_t)o_read_len[r] * match) : 0.0f);
            const float min_score = (float)simple_func(min_type, min_k, min_m, (int32_t)read_len[r]) + (paired ? (float)simple_func(min_type, min_k, min_m, (int32_t)o_read_len[r]) : 0.0f);
            out[r] = (uint8_t)mapq_v2(bp_best_score(&b), has_second, bp_second_score(&b), max_score, min_score, monotone);
        }
    }
}

ORACLE_API int oracle_num_threads(void)
{
#if defined(_OPENMP)
    return omp_get_max_threads();
#else
    return 1;
#endif
}
