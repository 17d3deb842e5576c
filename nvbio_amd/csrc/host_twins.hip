// host_twins.hip -- the `_host` twins of the C-ABI: the reference's HostThreadScheduler / host_tag paths
// (nvbio/alignment/batched_banded_inl.h:97-128, batched_inl.h:236-300, nvbio/fmindex/filter_inl.h:200-259) for callers
// that cannot instantiate templates.  Same argument lists as the device entry points with HOST pointers everywhere and a
// thread count instead of a stream; OpenMP over independent jobs.  They instantiate the very templates the drop-in layer
// hands to hipcc callers (include/nvbio_hip/compat: the generic per-job DP and fm_index functions), on runtime-typed packed
// strings.  These are explicit entry points -- nothing in the device path ever falls back to them.
#define NVBIO_HIP_COMPAT_NO_TUNED 1
#include "common.h"
#include "../../include/nvbio_hip/compat/nvbio/alignment/alignment.h"
#include "../../include/nvbio_hip/compat/nvbio/fmindex/fmindex.h"
#include "../../include/nvbio_hip/compat/nvbio/basic/deinterleaved_iterator.h"
#if defined(_OPENMP)
#include <omp.h>
#endif

namespace {

using namespace nvbio;

/// a packed string whose symbol width and endianness are runtime values (the C-ABI's nvbio_hip_string_set)
struct RtString
{
    typedef uint8 value_type;
    const uint32* words; uint64 begin; uint32 len; uint32 bits; uint32 be;
    uint32 length() const { return len; }
    uint8 operator[](const uint32 i) const
    {
        const uint64 s = begin + i;
        const uint32 per = 32u / bits, k = uint32(s % per);
        const uint32 sh = be ? (32u - bits * (k + 1u)) : (bits * k);
        return uint8((words[s / per] >> sh) & ((1u << bits) - 1u));
    }
};
inline RtString string_of(const nvbio_hip_string_set* s, const uint32 i)
{
    const RtString r = { s->words, s->begin[i], s->length ? s->length[i] : s->fixed_length, s->bits, s->big_endian };
    return r;
}
inline bool valid_set(const nvbio_hip_string_set* s) { return s && s->words && s->begin && (s->bits == 2 || s->bits == 4 || s->bits == 8); }

template <typename F> inline int with_band(const uint32 band, F f)
{
    switch (band) {
        case 3:  return f(std::integral_constant<uint32, 3>());
        case 5:  return f(std::integral_constant<uint32, 5>());
        case 7:  return f(std::integral_constant<uint32, 7>());
        case 15: return f(std::integral_constant<uint32, 15>());
        case 31: return f(std::integral_constant<uint32, 31>());
    }
    return hipErrorNotSupported;
}
template <typename F> inline int with_type(const int32 type, F f)
{
    if (type == 0) return f(std::integral_constant<aln::AlignmentType, aln::GLOBAL>());
    if (type == 1) return f(std::integral_constant<aln::AlignmentType, aln::LOCAL>());
    if (type == 2) return f(std::integral_constant<aln::AlignmentType, aln::SEMI_GLOBAL>());
    return hipErrorInvalidValue;
}

template <typename make_aligner>
int banded_host(make_aligner make, const int32 type, const uint32 band, const nvbio_hip_string_set* patterns, const nvbio_hip_string_set* texts,
                const uint32 n, int32* out_score, uint32* out_sink, const int n_threads)
{
    if (!valid_set(patterns) || !valid_set(texts)) return hipErrorInvalidValue;
    if (n && (!out_score || !out_sink)) return hipErrorInvalidValue;
#if defined(_OPENMP)
    const int threads = n_threads > 0 ? n_threads : omp_get_max_threads();
#endif
    return with_band(band, [&](auto B) { return with_type(type, [&](auto T) {
        const auto aligner = make(T);
        #pragma omp parallel for schedule(dynamic, 256) num_threads(threads)
        for (int64 i = 0; i < int64(n); ++i)
        {
            aln::BestSink<int32> sink;
            aln::banded_alignment_score<decltype(B)::value>(aligner, string_of(patterns, uint32(i)), aln::trivial_quality_string(), string_of(texts, uint32(i)),
                                                            Field_traits<int32>::min(), sink);
            out_score[i] = sink.score; out_sink[2 * i] = sink.sink.x; out_sink[2 * i + 1] = sink.sink.y;
        }
        return int(hipSuccess);
    }); });
}

// the production index layout seen through the reference's own types (nvbio/io/fmindex/fmindex.h:159-174)
typedef deinterleaved_iterator<2, 0, const uint4*>               bwt_words_t;
typedef deinterleaved_iterator<2, 1, const uint4*>               occ_t;
typedef PackedStream<bwt_words_t, uint8, 2, true>                bwt_t;
typedef rank_dictionary<2, 64, bwt_t, occ_t, null_type>          dict_t;

/// SSA sampled every sa_int rows, sa_int a runtime power of two
struct RtSSA
{
    const uint32* ssa; uint32 sa_int;
    bool fetch(const uint32 i, uint32& r) const { if (i & (sa_int - 1u)) return false; r = ssa[i / sa_int]; return true; }
    bool has(const uint32 i) const { return (i & (sa_int - 1u)) == 0u; }
};
typedef fm_index<dict_t, RtSSA> fmi_t;
inline fmi_t host_index(const nvbio_hip_fmindex* f)
{
    const uint4* base = reinterpret_cast<const uint4*>(f->bwt_occ);
    const RtSSA ssa = { f->ssa, f->sa_int };
    return fmi_t(f->length, f->primary, f->L2, dict_t(bwt_t(bwt_words_t(base)), occ_t(base), null_type()), ssa);
}

} // namespace

NVB_API int nvbio_hip_banded_gotoh_score_host(const nvbio_hip_gotoh_scheme* scheme, int32_t type, uint32_t band_len,
                                              const nvbio_hip_string_set* patterns, const nvbio_hip_string_set* texts,
                                              uint32_t n, int32_t* out_score, uint32_t* out_sink, int32_t n_threads)
{
    if (!scheme) return hipErrorInvalidValue;
    const aln::SimpleGotohScheme sc(scheme->match, scheme->mismatch, scheme->gap_open, scheme->gap_ext);
    return banded_host([&](auto T) { return aln::make_gotoh_aligner<decltype(T)::value>(sc); }, type, band_len, patterns, texts, n, out_score, out_sink, n_threads);
}

NVB_API int nvbio_hip_banded_sw_score_host(const nvbio_hip_sw_scheme* scheme, int32_t type, uint32_t band_len,
                                           const nvbio_hip_string_set* patterns, const nvbio_hip_string_set* texts,
                                           uint32_t n, int32_t* out_score, uint32_t* out_sink, int32_t n_threads)
{
    if (!scheme) return hipErrorInvalidValue;
    const aln::SimpleSmithWatermanScheme sc(scheme->match, scheme->mismatch, scheme->deletion, scheme->insertion);
    return banded_host([&](auto T) { return aln::make_smith_waterman_aligner<decltype(T)::value>(sc); }, type, band_len, patterns, texts, n, out_score, out_sink, n_threads);
}

NVB_API int nvbio_hip_alignment_score_host(int32_t aligner, int32_t algorithm, const int32_t* scheme4, int32_t type,
                                           const nvbio_hip_string_set* patterns, const nvbio_hip_string_set* texts, const int32_t* min_score,
                                           uint32_t n, int32_t* out_score, uint32_t* out_sink, uint8_t* out_ok, int32_t n_threads)
{
    if (!scheme4 || !valid_set(patterns) || !valid_set(texts)) return hipErrorInvalidValue;
    if (n && (!out_score || !out_sink)) return hipErrorInvalidValue;
    if (aligner != NVBIO_HIP_GOTOH_ALIGNER && aligner != NVBIO_HIP_SW_ALIGNER) return hipErrorInvalidValue;
#if defined(_OPENMP)
    const int threads = n_threads > 0 ? n_threads : omp_get_max_threads();
#endif
    auto run = [&](const auto al) {
        #pragma omp parallel num_threads(threads)
        {
            std::vector<int16> column;
            #pragma omp for schedule(dynamic, 64)
            for (int64 i = 0; i < int64(n); ++i)
            {
                const RtString p = string_of(patterns, uint32(i)), t = string_of(texts, uint32(i));
                column.resize(2u * size_t(p.len > t.len ? p.len : t.len) + 8u);
                aln::BestSink<int32> sink;
                const bool ok = aln::alignment_score(al, p, aln::trivial_quality_string(), t, min_score ? min_score[i] : Field_traits<int32>::min(), sink, column.data());
                out_score[i] = sink.score; out_sink[2 * i] = sink.sink.x; out_sink[2 * i + 1] = sink.sink.y;
                if (out_ok) out_ok[i] = ok ? 1 : 0;
            }
        }
        return int(hipSuccess);
    };
    return with_type(type, [&](auto T) {
        const aln::AlignmentType TYPE = decltype(T)::value;
        if (aligner == NVBIO_HIP_GOTOH_ALIGNER) {
            const aln::SimpleGotohScheme sc(scheme4[0], scheme4[1], scheme4[2], scheme4[3]);
            return algorithm == NVBIO_HIP_TEXT_BLOCKING ? run(aln::make_gotoh_aligner<TYPE, aln::TextBlockingTag>(sc)) : run(aln::make_gotoh_aligner<TYPE, aln::PatternBlockingTag>(sc));
        }
        const aln::SimpleSmithWatermanScheme sc(scheme4[0], scheme4[1], scheme4[2], scheme4[3]);
        return algorithm == NVBIO_HIP_TEXT_BLOCKING ? run(aln::make_smith_waterman_aligner<TYPE, aln::TextBlockingTag>(sc)) : run(aln::make_smith_waterman_aligner<TYPE, aln::PatternBlockingTag>(sc));
    });
}

NVB_API int nvbio_hip_fm_rank_host(const nvbio_hip_fmindex* fmi, const uint32_t* k, const uint8_t* c, uint32_t n, uint32_t* out, int32_t n_threads)
{
    if (!fmi || !fmi->bwt_occ || (n && (!k || !c || !out))) return hipErrorInvalidValue;
    const fmi_t f = host_index(fmi);
#if defined(_OPENMP)
    const int threads = n_threads > 0 ? n_threads : omp_get_max_threads();
#endif
    #pragma omp parallel for schedule(static) num_threads(threads)
    for (int64 i = 0; i < int64(n); ++i) out[i] = rank(f, k[i], uint8(c[i] & 3u));
    return hipSuccess;
}

NVB_API int nvbio_hip_fm_match_host(const nvbio_hip_fmindex* fmi, const nvbio_hip_string_set* seeds, uint32_t n, uint32_t* out_range, int32_t n_threads)
{
    if (!fmi || !fmi->bwt_occ || !valid_set(seeds) || (n && !out_range)) return hipErrorInvalidValue;
    const fmi_t f = host_index(fmi);
#if defined(_OPENMP)
    const int threads = n_threads > 0 ? n_threads : omp_get_max_threads();
#endif
    #pragma omp parallel for schedule(dynamic, 1024) num_threads(threads)
    for (int64 i = 0; i < int64(n); ++i)
    {
        const RtString s = string_of(seeds, uint32(i));
        const uint2 r = match(f, s, s.len);
        out_range[2 * i] = r.x; out_range[2 * i + 1] = r.y;
    }
    return hipSuccess;
}

NVB_API int nvbio_hip_fm_locate_host(const nvbio_hip_fmindex* fmi, const uint32_t* sa_rows, uint32_t n, uint32_t* out_pos, int32_t n_threads)
{
    if (!fmi || !fmi->bwt_occ || !fmi->ssa || fmi->sa_int == 0 || (fmi->sa_int & (fmi->sa_int - 1)) != 0 || (n && (!sa_rows || !out_pos))) return hipErrorInvalidValue;
    const fmi_t f = host_index(fmi);
#if defined(_OPENMP)
    const int threads = n_threads > 0 ? n_threads : omp_get_max_threads();
#endif
    #pragma omp parallel for schedule(dynamic, 1024) num_threads(threads)
    for (int64 i = 0; i < int64(n); ++i) out_pos[i] = locate(f, sa_rows[i]);
    return hipSuccess;
}
