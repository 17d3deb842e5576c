// nvbio_hip_test.cpp -- C++ parity driver for the hot path, written against the host layer in
// include/nvbio_hip/ the way the reference's own suites are written against nvbio
// (nvbio-test/alignment_test.cu, rank_test.cu, fmindex_test.cu): synthetic data, run the device
// path, compare with an independent host computation, exit(1) on the first mismatch.
// The host computation is the CPU oracle (oracle/nvbio_oracle.c), linked here as the checker.
//
//   nvbio_hip_test [-aln] [-rank] [-fm-index]      (no flag = all)
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <nvbio_hip/alignment.h>
#include <nvbio_hip/fmindex.h>
#include <nvbio_hip/mapping.h>
#include <nvbio_hip/io.h>
#include <nvbio_hip/reduce.h>
#include <nvbio_hip/select.h>
#include <unistd.h>

using namespace nvbio;

// ---- the oracle's C entry points (oracle/nvbio_oracle.c) --------------------------------------
extern "C" {
typedef struct { uint32_t length, primary, L2[5]; const uint32_t* bwt_occ; const uint32_t* ssa; uint32_t sa_int; } oracle_fmi_t;
void oracle_batch_gotoh_score(int type, const int32_t* scheme,
    const uint32_t* pat_w, uint32_t pat_bits, uint32_t pat_be, const uint64_t* pat_begin, const uint32_t* pat_len,
    const uint32_t* txt_w, uint32_t txt_bits, uint32_t txt_be, const uint64_t* txt_begin, const uint32_t* txt_len,
    const int32_t* min_score, uint32_t n, int32_t* out_score, uint32_t* out_sink, uint8_t* out_ok, int n_threads);
void oracle_batch_banded_gotoh_score(uint32_t band, int type, const int32_t* scheme,
    const uint32_t* pat_w, uint32_t pat_bits, uint32_t pat_be, const uint64_t* pat_begin, const uint32_t* pat_len,
    const uint32_t* txt_w, uint32_t txt_bits, uint32_t txt_be, const uint64_t* txt_begin, const uint32_t* txt_len,
    uint32_t n, int32_t* out_score, uint32_t* out_sink, int n_threads);
void oracle_banded_gotoh_traceback(uint32_t band, int type, const int32_t* scheme,
    const uint32_t* pat_w, uint32_t pat_bits, uint32_t pat_be, uint64_t pat_begin, uint32_t pat_len,
    const uint32_t* txt_w, uint32_t txt_bits, uint32_t txt_be, uint64_t txt_begin, uint32_t txt_len,
    int32_t* res, uint8_t* ops, uint32_t ops_capacity, uint8_t* flags);
void oracle_batch_score_pattern_blocking(int kind, int type, const int32_t* scheme,
    const uint32_t* pat_w, uint32_t pat_bits, uint32_t pat_be, const uint64_t* pat_begin, const uint32_t* pat_len,
    const uint32_t* txt_w, uint32_t txt_bits, uint32_t txt_be, const uint64_t* txt_begin, const uint32_t* txt_len,
    const int32_t* min_score, uint32_t n, int32_t* out_score, uint32_t* out_sink, uint8_t* out_ok, int n_threads);
void oracle_batch_sw_score(uint32_t band, int type, const int32_t* scheme,
    const uint32_t* pat_w, uint32_t pat_bits, uint32_t pat_be, const uint64_t* pat_begin, const uint32_t* pat_len,
    const uint32_t* txt_w, uint32_t txt_bits, uint32_t txt_be, const uint64_t* txt_begin, const uint32_t* txt_len,
    uint32_t n, int32_t* out_score, uint32_t* out_sink, int n_threads);
uint32_t oracle_bwt_from_sa(uint32_t n, const uint8_t* T, const uint32_t* SA, uint8_t* bwt);
void oracle_build_bwt_occ(uint32_t n, const uint32_t* bwt_words, uint32_t* bwt_occ, uint32_t* L2);
void oracle_build_ssa(uint32_t n, const uint32_t* SA, uint32_t K, uint32_t* ssa);
void oracle_fm_rank(const oracle_fmi_t* f, const uint32_t* k, const uint8_t* c, uint32_t n, uint32_t* out);
void oracle_fm_rank4(const oracle_fmi_t* f, const uint32_t* k, uint32_t n, uint32_t* out);
void oracle_fm_match(const oracle_fmi_t* f, const uint32_t* w, uint32_t bits, uint32_t be, const uint64_t* begin, const uint32_t* len,
    uint32_t n, uint32_t* out_range, uint64_t* algo_bytes, int n_threads);
void oracle_fm_locate(const oracle_fmi_t* f, const uint32_t* rows, uint32_t n, uint32_t* out_pos, uint64_t* total_steps, int n_threads);
typedef struct { uint32_t seed_len, min_read_len, max_hits, max_reseed, retry, rep_seeds, fw, rc; } oracle_map_params_t;
void oracle_map(int algorithm, uint32_t subseed_len, const oracle_fmi_t* f, const oracle_fmi_t* rf,
    const uint32_t* w, uint32_t bits, uint32_t be, const uint64_t* read_begin, const uint32_t* read_len,
    const uint32_t* in_queue, uint32_t n, const oracle_map_params_t* p, const uint32_t* seed_freq_by_len,
    uint64_t* out_hits, uint32_t hits_stride, uint32_t* out_counts, uint8_t* out_reseed);
uint32_t oracle_sum_tree_node_count(uint32_t size);
void oracle_select_init(uint32_t n_reads, const char* names, const uint32_t* names_idx, const uint64_t* hits, uint32_t hits_stride, const uint32_t* counts,
    float* probs, uint32_t probs_stride, uint32_t* trys, uint32_t* rseeds, uint32_t max_effort_init, int randomized, int top_seed);
void oracle_select(int randomized, uint32_t n_multi, const uint32_t* active_in, uint32_t n_active, uint64_t* hits, uint32_t hits_stride, uint32_t* counts,
    float* probs, uint32_t probs_stride, uint32_t* rseeds, const uint32_t* trys, uint32_t* active_out, uint64_t* hit_begin, uint32_t* hit_read_id,
    uint32_t* hit_loc, uint32_t* hit_seed, uint32_t* out_sizes);
void oracle_locate_hits(const oracle_fmi_t* f, const oracle_fmi_t* rf, uint32_t n, uint32_t* hit_loc, const uint32_t* hit_seed);
uint64_t oracle_filter_rank(const oracle_fmi_t* f, const uint32_t* w, uint32_t bits, uint32_t be, const uint64_t* begin, const uint32_t* len,
    uint32_t n, uint32_t* ranges, uint64_t* slots);
void oracle_filter_locate(const oracle_fmi_t* f, const uint32_t* ranges, const uint64_t* slots, uint32_t n_queries,
    uint64_t begin, uint64_t end, uint32_t* hits);
}

#define FAIL(...) do { fprintf(stderr, "  error: " __VA_ARGS__); fprintf(stderr, "\n"); exit(1); } while (0)

// the reference's LCG (nvbio/basic/numbers.h:610-621); its low bits are periodic, so symbols are
// drawn from the high bits here
struct LCG_random { uint32 m_s; LCG_random(uint32 s = 0) : m_s(s) {} uint32 next() { m_s = m_s * 1664525u + 1013904223u; return m_s; } uint32 sym() { return next() >> 30; } };

static std::vector<uint8> string_to_dna(const char* s) { std::vector<uint8> r; for (; *s; ++s) r.push_back(*s == 'A' ? 0 : *s == 'C' ? 1 : *s == 'G' ? 2 : 3); return r; }

// ----------------------------------------------------------------------------------- alignment
template <uint32 BAND_LEN, aln::AlignmentType TYPE, uint32 PBITS, bool PBE, bool TBE>
static void run_batch(const char* name, const aln::SimpleGotohScheme scoring,
                      const std::vector<std::vector<uint8> >& patterns, const std::vector<std::vector<uint8> >& texts,
                      std::vector<int32>* out_score = nullptr, std::vector<uint32>* out_sink = nullptr)
{
    const uint32 n = uint32(patterns.size());
    PackedStringSetDevice<PBITS, PBE> d_patterns(patterns);
    PackedStringSetDevice<2, TBE>     d_texts(texts);
    hip::device_vector<int32>  d_score(n);
    hip::device_vector<uint32> d_sink(2 * size_t(n));
    aln::BestSinkArrays sinks = { d_score.data(), d_sink.data() };

    aln::batch_banded_alignment_score<BAND_LEN>(
        aln::make_gotoh_aligner<TYPE>(scoring), d_patterns.view(), d_texts.view(), sinks,
        aln::DeviceThreadScheduler(), 0u, 0u);
    hip::synchronize();
    const std::vector<int32>  score = d_score.to_host();
    const std::vector<uint32> sink  = d_sink.to_host();

    // host reference: HostThreadScheduler semantics
    std::vector<uint8> pc, tc; std::vector<uint64> pb(n), tb(n); std::vector<uint32> pl(n), tl(n);
    for (uint32 i = 0; i < n; ++i) { pb[i] = pc.size(); pl[i] = uint32(patterns[i].size()); pc.insert(pc.end(), patterns[i].begin(), patterns[i].end());
                                     tb[i] = tc.size(); tl[i] = uint32(texts[i].size());    tc.insert(tc.end(), texts[i].begin(), texts[i].end()); }
    const std::vector<uint32> pw = pack_symbols<PBITS, PBE>(pc.data(), pc.size()), tw = pack_symbols<2, TBE>(tc.data(), tc.size());
    std::vector<int32> hs(n); std::vector<uint32> hk(2 * size_t(n));
    const int32 sc[4] = { scoring.m_match, scoring.m_mismatch, scoring.m_gap_open, scoring.m_gap_ext };
    oracle_batch_banded_gotoh_score(BAND_LEN, int(TYPE), sc, pw.data(), PBITS, PBE, pb.data(), pl.data(), tw.data(), 2, TBE, tb.data(), tl.data(), n, hs.data(), hk.data(), 0);
    for (uint32 i = 0; i < n; ++i)
        if (score[i] != hs[i] || sink[2 * i] != hk[2 * i] || sink[2 * i + 1] != hk[2 * i + 1])
            FAIL("%s: job %u: device (%d, %u,%u) != host (%d, %u,%u)", name, i, score[i], sink[2 * i], sink[2 * i + 1], hs[i], hk[2 * i], hk[2 * i + 1]);
    if (out_score) *out_score = score;
    if (out_sink)  *out_sink = sink;
    fprintf(stderr, "    %-44s : %u jobs ok\n", name, n);
}

// full-matrix Gotoh through batch_alignment_score (sw-benchmark's instantiation), device vs host
template <aln::AlignmentType TYPE>
static void run_full_batch(const char* name, const aln::SimpleGotohScheme scoring,
                           const std::vector<std::vector<uint8> >& patterns, const std::vector<std::vector<uint8> >& texts)
{
    const uint32 n = uint32(patterns.size());
    PackedStringSetDevice<4, true>  d_patterns(patterns);
    PackedStringSetDevice<2, false> d_texts(texts);
    hip::device_vector<int32>  d_score(n);
    hip::device_vector<uint32> d_sink(2 * size_t(n));
    aln::BestSinkArrays sinks = { d_score.data(), d_sink.data() };
    uint32 maxP = 1, maxT = 1;
    for (uint32 i = 0; i < n; ++i) { maxP = std::max(maxP, uint32(patterns[i].size())); maxT = std::max(maxT, uint32(texts[i].size())); }
    aln::batch_alignment_score(aln::make_gotoh_aligner<TYPE, aln::TextBlockingTag>(scoring), d_patterns.view(), d_texts.view(), sinks, aln::DeviceThreadScheduler(), maxP, maxT);
    hip::synchronize();
    const std::vector<int32> score = d_score.to_host(); const std::vector<uint32> sink = d_sink.to_host();
    std::vector<uint8> pc, tc; std::vector<uint64> pb(n), tb(n); std::vector<uint32> pl(n), tl(n);
    for (uint32 i = 0; i < n; ++i) { pb[i] = pc.size(); pl[i] = uint32(patterns[i].size()); pc.insert(pc.end(), patterns[i].begin(), patterns[i].end());
                                     tb[i] = tc.size(); tl[i] = uint32(texts[i].size());    tc.insert(tc.end(), texts[i].begin(), texts[i].end()); }
    const std::vector<uint32> pw = pack_symbols<4, true>(pc.data(), pc.size()), tw = pack_symbols<2, false>(tc.data(), tc.size());
    std::vector<int32> hs(n); std::vector<uint32> hk(2 * size_t(n));
    const int32 sc[4] = { scoring.m_match, scoring.m_mismatch, scoring.m_gap_open, scoring.m_gap_ext };
    oracle_batch_gotoh_score(int(TYPE), sc, pw.data(), 4, 1, pb.data(), pl.data(), tw.data(), 2, 0, tb.data(), tl.data(), nullptr, n, hs.data(), hk.data(), nullptr, 0);
    for (uint32 i = 0; i < n; ++i)
        if (score[i] != hs[i] || sink[2 * i] != hk[2 * i] || sink[2 * i + 1] != hk[2 * i + 1])
            FAIL("%s: job %u: device (%d, %u,%u) != host (%d, %u,%u)", name, i, score[i], sink[2 * i], sink[2 * i + 1], hs[i], hk[2 * i], hk[2 * i + 1]);
    fprintf(stderr, "    %-44s : %u jobs ok\n", name, n);
}

// SmithWatermanAligner / EditDistanceAligner batches (banded: BAND_LEN > 0; full matrix, text blocking: BAND_LEN == 0)
template <uint32 BAND_LEN, typename aligner_type>
static void run_sw_batch(const char* name, const aligner_type aligner,
                         const std::vector<std::vector<uint8> >& patterns, const std::vector<std::vector<uint8> >& texts)
{
    const uint32 n = uint32(patterns.size());
    uint32 maxP = 0, maxT = 0;
    for (uint32 i = 0; i < n; ++i) { maxP = std::max(maxP, uint32(patterns[i].size())); maxT = std::max(maxT, uint32(texts[i].size())); }
    PackedStringSetDevice<4, true>  d_patterns(patterns);
    PackedStringSetDevice<2, false> d_texts(texts);
    hip::device_vector<int32>  d_score(n);
    hip::device_vector<uint32> d_sink(2 * size_t(n));
    aln::BestSinkArrays sinks = { d_score.data(), d_sink.data() };
    constexpr bool PATTERN_BLOCKING = !aln::priv::same_type<typename aligner_type::algorithm_type, aln::TextBlockingTag>::value;
    if constexpr (BAND_LEN != 0) aln::batch_banded_alignment_score<BAND_LEN ? BAND_LEN : 15>(aligner, d_patterns.view(), d_texts.view(), sinks, aln::DeviceThreadScheduler(), maxP, maxT);
    else          aln::batch_alignment_score(aligner, d_patterns.view(), d_texts.view(), sinks, aln::DeviceThreadScheduler(), maxP, maxT);
    hip::synchronize();
    const std::vector<int32> score = d_score.to_host(); const std::vector<uint32> sink = d_sink.to_host();
    std::vector<uint8> pc, tc; std::vector<uint64> pb(n), tb(n); std::vector<uint32> pl(n), tl(n);
    for (uint32 i = 0; i < n; ++i) { pb[i] = pc.size(); pl[i] = uint32(patterns[i].size()); pc.insert(pc.end(), patterns[i].begin(), patterns[i].end());
                                     tb[i] = tc.size(); tl[i] = uint32(texts[i].size());    tc.insert(tc.end(), texts[i].begin(), texts[i].end()); }
    tc.insert(tc.end(), 64, 0);
    const std::vector<uint32> pw = pack_symbols<4, true>(pc.data(), pc.size()), tw = pack_symbols<2, false>(tc.data(), tc.size());
    std::vector<int32> hs(n); std::vector<uint32> hk(2 * size_t(n));
    const int32 sc[4] = { aligner.scheme.m_match, aligner.scheme.m_mismatch, aligner.scheme.m_deletion, aligner.scheme.m_insertion };
    if (BAND_LEN == 0 && PATTERN_BLOCKING)     // the reference's default algorithm tag
        oracle_batch_score_pattern_blocking(1, int(aligner_type::TYPE), sc, pw.data(), 4, 1, pb.data(), pl.data(), tw.data(), 2, 0, tb.data(), tl.data(), nullptr, n, hs.data(), hk.data(), nullptr, 0);
    else
        oracle_batch_sw_score(BAND_LEN, int(aligner_type::TYPE), sc, pw.data(), 4, 1, pb.data(), pl.data(), tw.data(), 2, 0, tb.data(), tl.data(), n, hs.data(), hk.data(), 0);
    for (uint32 i = 0; i < n; ++i)
        if (score[i] != hs[i] || sink[2 * i] != hk[2 * i] || sink[2 * i + 1] != hk[2 * i + 1])
            FAIL("%s: job %u: device (%d, %u,%u) != host (%d, %u,%u)", name, i, score[i], sink[2 * i], sink[2 * i + 1], hs[i], hk[2 * i], hk[2 * i + 1]);
    fprintf(stderr, "    %-44s : %u jobs ok\n", name, n);
}

// run-length string of a backwards io::Cigar vector, as the reference test prints it (rle(backtracker.aln))
static std::string cigar_string(const io::Cigar* c, const uint32 n)
{
    std::string r; char buf[32];
    for (uint32 i = 0; i < n; ++i) { snprintf(buf, sizeof(buf), "%u%c", uint32(c[i].m_len), "MIDS"[c[i].m_type]); r += buf; }
    return r;
}

// BatchedBandedAlignmentTraceback vs the oracle's banded_alignment_traceback, job by job
template <uint32 BAND_LEN, aln::AlignmentType TYPE>
static void run_traceback_batch(const char* name, const aln::SimpleGotohScheme scoring,
                                const std::vector<std::vector<uint8> >& patterns, const std::vector<std::vector<uint8> >& texts,
                                const char* expect_cigar = nullptr)
{
    const uint32 n = uint32(patterns.size()), STRIDE = 48;
    uint32 maxP = 0; for (uint32 i = 0; i < n; ++i) maxP = std::max(maxP, uint32(patterns[i].size()));
    PackedStringSetDevice<4, true> d_patterns(patterns);
    PackedStringSetDevice<2, true> d_texts(texts);
    hip::device_vector<int32>  d_score(n);
    hip::device_vector<uint32> d_sink(2 * size_t(n)), d_source(2 * size_t(n)), d_len(n);
    hip::device_vector<io::Cigar> d_cigar(size_t(n) * STRIDE);
    typedef aln::GotohAligner<TYPE, aln::SimpleGotohScheme> aligner_type;
    typedef aln::PackedTracebackStream<aligner_type, PackedStringSetView<4, true>, PackedStringSetView<2, true> > stream_type;
    typedef aln::BatchedBandedAlignmentTraceback<BAND_LEN, 32u, stream_type> batch_type;
    const uint64 temp_size = batch_type::max_temp_storage(maxP, maxP + BAND_LEN, n);
    hip::device_vector<uint8> d_temp(temp_size ? temp_size : 1);
    aln::AlignmentArrays alns = { d_score.data(), d_source.data(), d_sink.data() };
    aln::CigarArrays cigs = { d_cigar.data(), STRIDE, d_len.data() };
    batch_type batch;
    batch.enact(stream_type(aln::make_gotoh_aligner<TYPE>(scoring), d_patterns.view(), d_texts.view(), alns, cigs, maxP, maxP + BAND_LEN), temp_size, d_temp.data());
    hip::synchronize();
    const std::vector<int32> score = d_score.to_host();
    const std::vector<uint32> sink = d_sink.to_host(), source = d_source.to_host(), len = d_len.to_host();
    const std::vector<io::Cigar> cigar = d_cigar.to_host();

    std::vector<uint8> pc, tc; std::vector<uint64> pb(n), tb(n);
    for (uint32 i = 0; i < n; ++i) { pb[i] = pc.size(); pc.insert(pc.end(), patterns[i].begin(), patterns[i].end());
                                     tb[i] = tc.size(); tc.insert(tc.end(), texts[i].begin(), texts[i].end()); }
    tc.insert(tc.end(), 64, 0);            // the device set is zero padded past its last string as well
    const std::vector<uint32> pw = pack_symbols<4, true>(pc.data(), pc.size()), tw = pack_symbols<2, true>(tc.data(), tc.size());
    const int32 sc[4] = { scoring.m_match, scoring.m_mismatch, scoring.m_gap_open, scoring.m_gap_ext };
    std::vector<uint8> ops(2 * maxP + BAND_LEN + 8), flags(size_t(maxP + 1) * BAND_LEN);
    for (uint32 i = 0; i < n; ++i)
    {
        int32 res[8];
        oracle_banded_gotoh_traceback(BAND_LEN, int(TYPE), sc, pw.data(), 4, 1, pb[i], uint32(patterns[i].size()),
                                      tw.data(), 2, 1, tb[i], uint32(texts[i].size()), res, ops.data(), uint32(ops.size()), flags.data());
        std::vector<io::Cigar> h;
        if (res[1] != -1 || res[2] != -1) {
            if (res[6]) h.push_back(io::Cigar(io::Cigar::SOFT_CLIPPING, uint16(res[6])));
            for (int32 k = 0; k < res[5]; ) { int32 e = k; while (e < res[5] && ops[e] == ops[k]) ++e; h.push_back(io::Cigar(ops[k], uint16(e - k))); k = e; }
            if (res[7]) h.push_back(io::Cigar(io::Cigar::SOFT_CLIPPING, uint16(res[7])));
        }
        bool ok = score[i] == res[0] && source[2 * i] == uint32(res[1]) && source[2 * i + 1] == uint32(res[2]) &&
                  sink[2 * i] == uint32(res[3]) && sink[2 * i + 1] == uint32(res[4]) && len[i] == h.size();
        for (uint32 k = 0; ok && k < std::min<uint32>(len[i], STRIDE); ++k)
            ok = cigar[size_t(i) * STRIDE + k].m_type == h[k].m_type && cigar[size_t(i) * STRIDE + k].m_len == h[k].m_len;
        if (!ok)
            FAIL("%s: job %u: device %d [%u,%u]->[%u,%u] %s != host %d [%u,%u]->[%u,%u] %s", name, i,
                 score[i], source[2 * i], source[2 * i + 1], sink[2 * i], sink[2 * i + 1], cigar_string(&cigar[size_t(i) * STRIDE], std::min<uint32>(len[i], STRIDE)).c_str(),
                 res[0], uint32(res[1]), uint32(res[2]), uint32(res[3]), uint32(res[4]), cigar_string(h.data(), uint32(h.size())).c_str());
    }
    if (expect_cigar && cigar_string(&cigar[0], len[0]) != expect_cigar)
        FAIL("%s: expected %s, got %s", name, expect_cigar, cigar_string(&cigar[0], len[0]).c_str());
    fprintf(stderr, "    %-44s : %u jobs ok%s%s\n", name, n, expect_cigar ? "  " : "", expect_cigar ? expect_cigar : "");
}

template <uint32 BAND_LEN, aln::AlignmentType TYPE>
static void expect_single(const char* name, aln::SimpleGotohScheme sc, const char* p, const char* t, int32 score, uint32 sx, uint32 sy)
{
    std::vector<int32> s; std::vector<uint32> k;
    run_batch<BAND_LEN, TYPE, 4, true, false>(name, sc, { string_to_dna(p) }, { string_to_dna(t) }, &s, &k);
    if (s[0] != score || k[0] != sx || k[1] != sy) FAIL("%s: expected %d @ (%u,%u), got %d @ (%u,%u)", name, score, sx, sy, s[0], k[0], k[1]);
}

static int alignment_test()
{
    fprintf(stderr, "testing alignment... started\n");
    // known answers (tests/golden/kat.json; strings of nvbio-test/alignment_test.cu:749-826)
    const char* P1 = "ACAACTA"; const char* T1 = "AAACACCCTAACACACTAAA";
    expect_single<7,  aln::SEMI_GLOBAL>("kat banded-semi-global 7", aln::SimpleGotohScheme(2, -1, -1, -1), P1, T1, 10, 10, 7);
    expect_single<7,  aln::LOCAL>      ("kat banded-local 7",       aln::SimpleGotohScheme(2, -1, -1, -1), P1, T1, 10, 10, 7);
    expect_single<7,  aln::GLOBAL>     ("kat banded-global 7",      aln::SimpleGotohScheme(2, -1, -1, -1), P1, T1, 5, 13, 7);
    expect_single<15, aln::SEMI_GLOBAL>("kat banded-semi-global 15",aln::SimpleGotohScheme(2, -1, -1, -1), P1, T1, 13, 18, 7);
    const char* P2 = "TTATGTAGGTGGTCTGGTTTTTGCCTTTTAAGCTTCTGCAAAAAACAACAACAAACTTGTGGTATTACACTGACTCTACAGATCAATTTGGGGACAACTTCCATGTGTTCCACCACCAATACTGAATCTTTCAATCGACTGACGTGGTAT";
    const char* T2 = "ATCGGATTCTTTCTTACTTGTAGGTGGTCTGGTTTTTGCCTTTTAAGCTTCTGCAAAAAACAACAACAAACTTGTGGTATTACACTGACTCTACAGATCAATTTGGGGACAACTTCCATGTGTTCCACCACCAATACTGAATCTTTCAATCGACTGACGTGGTATCTCTCTCTCCATCTAT";
    expect_single<31, aln::SEMI_GLOBAL>("kat real banded Gotoh 31", aln::SimpleGotohScheme(0, -5, -8, -3), P2, T2, -11, 165, 150);
    expect_single<15, aln::SEMI_GLOBAL>("kat real banded Gotoh 15", aln::SimpleGotohScheme(0, -5, -8, -3), P2, T2, -403, 160, 150);
    expect_single<31, aln::LOCAL>      ("kat real banded local 31", aln::SimpleGotohScheme(2, -1, -2, -1), P2, T2, 297, 165, 150);

    // the CIGAR literals of the reference's banded traceback tests (alignment_test.cu:793, :825)
    run_traceback_batch<7,  aln::SEMI_GLOBAL>("kat traceback banded-semi-global 7", aln::SimpleGotohScheme(2, -1, -1, -1), { string_to_dna(P1) }, { string_to_dna(T1) }, "4M1D3M");
    run_traceback_batch<31, aln::SEMI_GLOBAL>("kat traceback real banded Gotoh 31", aln::SimpleGotohScheme(0, -5, -8, -3), { string_to_dna(P2) }, { string_to_dna(T2) }, "147M2D3M");

    // the full-matrix Gotoh CIGARs of the same functional test (alignment_test.cu:788-792)
    {
        const struct { aln::AlignmentType type; const char* cigar; } cases[3] = { { aln::GLOBAL, "1M2D3M1D3M10D" }, { aln::LOCAL, "4M1D3M" }, { aln::SEMI_GLOBAL, "4M1D3M" } };
        PackedStringSetDevice<4, true> d_p(std::vector<std::vector<uint8> >(1, string_to_dna(P1)));
        PackedStringSetDevice<2, true> d_t(std::vector<std::vector<uint8> >(1, string_to_dna(T1)));
        hip::device_vector<int32> d_score(1); hip::device_vector<uint32> d_sink(2), d_source(2), d_len(1); hip::device_vector<io::Cigar> d_cigar(32);
        const uint64 tsz = nvbio_hip_gotoh_traceback_temp_bytes(7, 20, 1);
        hip::device_vector<uint8> d_temp(tsz);
        aln::AlignmentArrays alns = { d_score.data(), d_source.data(), d_sink.data() };
        aln::CigarArrays cigs = { d_cigar.data(), 32u, d_len.data() };
        for (int c = 0; c < 3; ++c) {
            const nvbio_hip_gotoh_scheme sc = { 2, -1, -1, -1 };
            const nvbio_hip_string_set p = d_p.view().abi(), t = d_t.view().abi();
            hip_check(nvbio_hip_gotoh_traceback(&sc, int32(cases[c].type), &p, &t, 7, 20, 1, alns.score, alns.sink, alns.source,
                                                reinterpret_cast<uint16*>(cigs.cigar), 32u, cigs.cigar_len, d_temp.data(), tsz, nullptr), "nvbio_hip_gotoh_traceback");
            hip::synchronize();
            const std::vector<io::Cigar> cg = d_cigar.to_host(); const uint32 len = d_len.to_host()[0];
            if (cigar_string(cg.data(), len) != cases[c].cigar) FAIL("full traceback type %d: expected %s, got %s", int(cases[c].type), cases[c].cigar, cigar_string(cg.data(), len).c_str());
        }
        fprintf(stderr, "    %-44s : 1M2D3M1D3M10D / 4M1D3M / 4M1D3M ok\n", "kat full-matrix traceback (global/local/semi)");
    }
    // the throughput configuration of alignment_test.cu:1071-1194 (BAND=15, M=150, N=M+15), with
    // planted substitutions so that scores are not trivial; reads 4-bit LE, refs 2-bit LE as there
    const uint32 N_TASKS = 32768, M = 150, N = M + 15;
    LCG_random rnd(7);
    std::vector<std::vector<uint8> > pats(N_TASKS), txts(N_TASKS);
    for (uint32 i = 0; i < N_TASKS; ++i) {
        txts[i].resize(N); for (uint32 j = 0; j < N; ++j) txts[i][j] = uint8(rnd.sym());
        pats[i].assign(txts[i].begin() + 7, txts[i].begin() + 7 + M);
        for (uint32 j = 0; j < M; ++j) if ((rnd.next() >> 16) % 100 < 5) pats[i][j] = uint8(rnd.sym());
        if (i % 3 == 0) pats[i].erase(pats[i].begin() + 40 + (i % 50), pats[i].begin() + 42 + (i % 50));    // ragged: a deletion
        if (i % 97 == 0) pats[i][i % pats[i].size()] = 4;                                                   // an N
        if (i % 211 == 0) txts[i].resize(pats[i].size() - 3);                                               // text shorter than pattern
    }
    const aln::SimpleGotohScheme s1(2, -1, -1, -1), s2(2, -1, -2, -1), s3(0, -5, -8, -3);
    run_batch<15, aln::GLOBAL,      4, false, false>("batch gotoh-banded global 15",      s1, pats, txts);
    run_batch<15, aln::SEMI_GLOBAL, 4, false, false>("batch gotoh-banded semi-global 15", s1, pats, txts);
    run_batch<15, aln::LOCAL,       4, false, false>("batch gotoh-banded local 15",       s1, pats, txts);
    run_batch<15, aln::LOCAL,       4, true,  false>("batch local 15, BE reads (sw-benchmark fmt)", s2, pats, txts);
    run_batch<15, aln::LOCAL,       4, true,  true >("batch local 15, BE reads + BE genome (nvBowtie fmt)", s2, pats, txts);
    run_batch<31, aln::SEMI_GLOBAL, 4, true,  true >("batch semi-global 31",              s3, pats, txts);
    run_batch<31, aln::LOCAL,       4, true,  false>("batch local 31",                    s2, pats, txts);
    run_batch<7,  aln::SEMI_GLOBAL, 4, true,  false>("batch semi-global 7",               s3, pats, txts);
    run_batch<3,  aln::LOCAL,       4, true,  false>("batch local 3",                     s2, pats, txts);
    run_batch<5,  aln::GLOBAL,      4, true,  false>("batch global 5",                    s2, pats, txts);
    // linear-gap aligners in the band (nvbio-test's ed-banded / sw-banded cases)
    run_sw_batch<15>("batch ed-banded semi-global 15", aln::make_edit_distance_aligner<aln::SEMI_GLOBAL>(), pats, txts);
    run_sw_batch<15>("batch sw-banded local 15",       aln::make_smith_waterman_aligner<aln::LOCAL>(aln::SimpleSmithWatermanScheme(2, -1, -1, -1)), pats, txts);
    run_sw_batch<7> ("batch sw-banded global 7",       aln::make_smith_waterman_aligner<aln::GLOBAL>(aln::SimpleSmithWatermanScheme(2, -1, -1, -1)), pats, txts);
    // banded traceback -> CIGAR over the same ragged batch (nvBowtie's traceback stage shape)
    run_traceback_batch<15, aln::LOCAL>      ("batch traceback local 15",       s2, pats, txts);
    run_traceback_batch<15, aln::SEMI_GLOBAL>("batch traceback semi-global 15", s3, pats, txts);
    run_traceback_batch<31, aln::GLOBAL>     ("batch traceback global 31",      s1, pats, txts);
    run_traceback_batch<7,  aln::LOCAL>      ("batch traceback local 7",        s1, pats, txts);
    // full-matrix Gotoh (the sw-benchmark instantiation): reads against longer references
    {
        std::vector<std::vector<uint8> > fp(2048), ft(2048);
        for (uint32 i = 0; i < 2048; ++i) {
            ft[i].resize(300 + i % 200); for (size_t j = 0; j < ft[i].size(); ++j) ft[i][j] = uint8(rnd.sym());
            const uint32 L = 30 + i % 120, off = i % 100;
            fp[i].assign(ft[i].begin() + off, ft[i].begin() + off + L);
            for (uint32 j = 0; j < L; ++j) if ((rnd.next() >> 16) % 100 < 6) fp[i][j] = uint8(rnd.sym());
        }
        run_sw_batch<0>("batch ed full semi-global (sw-benchmark leg)", aln::make_edit_distance_aligner<aln::SEMI_GLOBAL, aln::TextBlockingTag>(), fp, ft);
        run_sw_batch<0>("batch sw full local", aln::make_smith_waterman_aligner<aln::LOCAL, aln::TextBlockingTag>(aln::SimpleSmithWatermanScheme(2, -1, -1, -1)), fp, ft);
        run_sw_batch<0>("batch sw full local (default tag: pattern blocking)", aln::make_smith_waterman_aligner<aln::LOCAL>(aln::SimpleSmithWatermanScheme(2, -1, -1, -1)), fp, ft);
        run_sw_batch<0>("batch ed full global (default tag)", aln::make_edit_distance_aligner<aln::GLOBAL>(), fp, ft);
        run_full_batch<aln::LOCAL>      ("batch gotoh full local",       s2, fp, ft);
        run_full_batch<aln::SEMI_GLOBAL>("batch gotoh full semi-global", s2, fp, ft);
        run_full_batch<aln::GLOBAL>     ("batch gotoh full global",      s3, fp, ft);
    }
    fprintf(stderr, "testing alignment... done\n");
    return 0;
}

// ----------------------------------------------------------------------------------- FM-index
struct HostIndex {
    uint32 n, primary; uint32 L2[5];
    std::vector<uint8> text, bwt; std::vector<uint32> sa, bwt_words, bwt_occ, ssa;
    oracle_fmi_t ofmi;
};

static void build_host_index(HostIndex& h, uint32 n, uint32 seed, bool reversed = false)
{
    h.n = n; h.text.resize(n);
    LCG_random rnd(seed);
    for (uint32 i = 0; i < n; ++i) h.text[i] = uint8(rnd.sym());
    if (reversed) std::reverse(h.text.begin(), h.text.end());      // the index of the reversed genome (rfmi)
    // suffix array by prefix doubling; SA[0] = n (the '$' suffix) as gen_sa pads it (bwt.h:36-45)
    std::vector<uint32> sa(n + 1), rk(n + 1), tmp(n + 1);
    std::iota(sa.begin(), sa.end(), 0u);
    for (uint32 i = 0; i <= n; ++i) rk[i] = i < n ? h.text[i] + 1u : 0u;
    for (uint32 k = 1;; k <<= 1) {
        auto key = [&](uint32 i) { return std::make_pair(rk[i], i + k <= n ? rk[i + k] + 1u : 0u); };
        std::sort(sa.begin(), sa.end(), [&](uint32 a, uint32 b) { return key(a) < key(b); });
        tmp[sa[0]] = 0;
        for (uint32 i = 1; i <= n; ++i) tmp[sa[i]] = tmp[sa[i - 1]] + (key(sa[i - 1]) < key(sa[i]) ? 1u : 0u);
        rk = tmp;
        if (rk[sa[n]] == n) break;
    }
    h.sa = sa;
    h.bwt.resize(n + 1);
    h.primary = oracle_bwt_from_sa(n, h.text.data(), h.sa.data(), h.bwt.data());
    const uint32 n_blocks = (n + 63) / 64;
    h.bwt_words.assign(size_t(n_blocks) * 4, 0u);
    { const std::vector<uint32> w = pack_symbols<2, true>(h.bwt.data(), n, 0); std::copy(w.begin(), w.end(), h.bwt_words.begin()); }
    h.bwt_occ.resize(size_t(n_blocks) * 8);
    oracle_build_bwt_occ(n, h.bwt_words.data(), h.bwt_occ.data(), h.L2);
    h.ssa.resize((n + 16) / 16);
    oracle_build_ssa(n, h.sa.data(), 16, h.ssa.data());
    h.ofmi.length = n; h.ofmi.primary = h.primary; memcpy(h.ofmi.L2, h.L2, sizeof h.L2);
    h.ofmi.bwt_occ = h.bwt_occ.data(); h.ofmi.ssa = h.ssa.data(); h.ofmi.sa_int = 16;
}

static int rank_test()
{
    fprintf(stderr, "rank test... started\n");
    HostIndex h; build_host_index(h, 100000, 3);
    // device-built occurrence table == host build_occurrence_table<2,64> + interleave
    hip::device_vector<uint32> d_bwt(h.bwt_words), d_bwt_occ(h.bwt_occ.size());
    uint32 L2[5];
    build_bwt_occ(h.n, d_bwt.data(), d_bwt_occ.data(), L2);
    if (d_bwt_occ.to_host() != h.bwt_occ || memcmp(L2, h.L2, sizeof L2)) FAIL("device occurrence table differs from the host one");
    // rank(dict, i, c) against a running naive count, every i and c (rank_test.cu:55-86).
    // primary = n keeps the fm_index '$' adjustment out of the way: this is the rank dictionary alone.
    fm_index_device fmi(h.n, h.n, h.L2, d_bwt_occ.data(), nullptr);
    std::vector<uint32> k(h.n); std::iota(k.begin(), k.end(), 0u);
    hip::device_vector<uint32> d_k(k); hip::device_vector<uint4> d_r4(h.n);
    rank4(fmi, h.n, d_k.data(), d_r4.data());
    const std::vector<uint4> r4 = d_r4.to_host();
    uint32 cnt[4] = { 0, 0, 0, 0 };
    for (uint32 i = 0; i < h.n; ++i) {
        ++cnt[h.bwt[i]];
        if (r4[i].x != cnt[0] || r4[i].y != cnt[1] || r4[i].z != cnt[2] || r4[i].w != cnt[3]) FAIL("rank4 mismatch at %u", i);
    }
    for (uint32 c = 0; c < 4; ++c) {
        std::vector<uint8> cc(h.n, uint8(c)); hip::device_vector<uint8> d_c(cc); hip::device_vector<uint32> d_r(h.n);
        rank(fmi, h.n, d_k.data(), d_c.data(), d_r.data());
        const std::vector<uint32> r = d_r.to_host();
        uint32 run = 0;
        for (uint32 i = 0; i < h.n; ++i) { run += (h.bwt[i] == c); if (r[i] != run) FAIL("rank mismatch at %u, c=%u: %u != %u", i, c, r[i], run); }
    }
    fprintf(stderr, "rank test... done\n");
    return 0;
}

static int fmindex_test()
{
    fprintf(stderr, "FM-index test... started\n");
    HostIndex h; build_host_index(h, 1u << 18, 11);
    hip::device_vector<uint32> d_bwt_occ(h.bwt_occ), d_ssa(h.ssa);
    fm_index_device fmi(h.n, h.primary, h.L2, d_bwt_occ.data(), d_ssa.data(), 16);

    // ssa check (fmindex_test.cu:582-592): locate of every row == SA
    {
        std::vector<uint32> rows(h.n); std::iota(rows.begin(), rows.end(), 1u);
        hip::device_vector<uint32> d_rows(rows), d_pos(h.n);
        locate(fmi, h.n, d_rows.data(), d_pos.data());
        const std::vector<uint32> pos = d_pos.to_host();
        for (uint32 i = 0; i < h.n; ++i) if (pos[i] != h.sa[i + 1]) FAIL("locate(%u) = %u != SA = %u", i + 1, pos[i], h.sa[i + 1]);
    }
    // match 8-mers, then locate every row of the range and compare the text there (fmindex_test.cu:610-664)
    LCG_random rnd(5);
    const uint32 Q = 65536, LEN = 8;
    std::vector<std::vector<uint8> > seeds(Q);
    for (uint32 q = 0; q < Q; ++q) { const uint32 p = rnd.next() % (h.n - LEN); seeds[q].assign(h.text.begin() + p, h.text.begin() + p + LEN); if (q % 7 == 0) seeds[q][q % LEN] = uint8(rnd.sym()); }
    for (int shuffled = 0; shuffled < 2; ++shuffled) {       // sorted and shuffled orders (fmindex_test.cu:666-716)
        if (!shuffled) std::sort(seeds.begin(), seeds.end());
        else for (uint32 q = Q - 1; q > 0; --q) std::swap(seeds[q], seeds[rnd.next() % (q + 1)]);
        PackedStringSetDevice<2, true> d_seeds(seeds);
        hip::device_vector<uint2> d_ranges(Q);
        match(fmi, d_seeds.view(), d_ranges.data());
        const std::vector<uint2> ranges = d_ranges.to_host();
        // host oracle
        std::vector<uint8> cat; std::vector<uint64> b(Q); std::vector<uint32> l(Q, LEN);
        for (uint32 q = 0; q < Q; ++q) { b[q] = cat.size(); cat.insert(cat.end(), seeds[q].begin(), seeds[q].end()); }
        const std::vector<uint32> w = pack_symbols<2, true>(cat.data(), cat.size());
        std::vector<uint32> hr(2 * size_t(Q));
        oracle_fm_match(&h.ofmi, w.data(), 2, 1, b.data(), l.data(), Q, hr.data(), nullptr, 0);
        for (uint32 q = 0; q < Q; ++q) if (ranges[q].x != hr[2 * q] || ranges[q].y != hr[2 * q + 1]) FAIL("match %u: device (%u,%u) host (%u,%u)", q, ranges[q].x, ranges[q].y, hr[2 * q], hr[2 * q + 1]);
        // the filter: rank + locate, every hit must hold its seed in the text
        FMIndexFilterDevice filter;
        const uint64 n_hits = filter.rank(fmi, d_seeds.view());
        hip::device_vector<uint2> d_hits(n_hits);
        filter.locate(0, n_hits, d_hits.data());
        const std::vector<uint2> hits = d_hits.to_host();
        uint64 expect = 0;
        for (uint32 q = 0; q < Q; ++q) expect += uint32(1u + hr[2 * q + 1] - hr[2 * q]);
        if (expect != n_hits) FAIL("filter.rank: %llu hits, expected %llu", (unsigned long long)n_hits, (unsigned long long)expect);
        for (uint64 i = 0; i < n_hits; ++i)
            if (memcmp(h.text.data() + hits[i].x, seeds[hits[i].y].data(), LEN)) FAIL("hit %llu: text at %u does not hold seed %u", (unsigned long long)i, hits[i].x, hits[i].y);
    }
    // on-disk formats: write <prefix>.bwt/.sa the way nvBWT does, load them through io::FMIndexDataHost /
    // io::FMIndexDataDevice (occurrence table built on the device) and compare with the index built in memory
    {
        char prefix[256]; snprintf(prefix, sizeof(prefix), "/tmp/nvbio_hip_test_%d", int(getpid()));
        const std::string bwt_name = std::string(prefix) + ".bwt", sa_name = std::string(prefix) + ".sa";
        uint32 cum[4] = { h.L2[1], h.L2[2], h.L2[3], h.L2[4] };
        FILE* f = fopen(bwt_name.c_str(), "wb");
        fwrite(&h.primary, 4, 1, f); fwrite(cum, 4, 4, f); fwrite(h.bwt_words.data(), 4, (h.n + 15) / 16, f); fclose(f);
        f = fopen(sa_name.c_str(), "wb");
        const uint32 sa_int = 16;
        fwrite(&h.primary, 4, 1, f); fwrite(cum, 4, 4, f); fwrite(&sa_int, 4, 1, f); fwrite(&h.n, 4, 1, f); fwrite(h.ssa.data() + 1, 4, h.ssa.size() - 1, f); fclose(f);
        io::FMIndexDataHost host_data;
        if (!host_data.load(prefix, io::FMIndexDataCore::FORWARD | io::FMIndexDataCore::SA)) FAIL("FMIndexDataHost::load failed");
        io::FMIndexDataDevice dev_data(host_data, io::FMIndexDataCore::FORWARD);
        const fm_index_device& lf = dev_data.index();
        if (lf.length() != h.n || lf.primary() != h.primary) FAIL("loaded index: length/primary mismatch");
        for (int c = 0; c < 5; ++c) if (lf.m.L2[c] != h.L2[c]) FAIL("loaded index: L2[%d] = %u != %u", c, lf.m.L2[c], h.L2[c]);
        std::vector<uint32> got(h.bwt_occ.size());
        hip_check(nvbio_hip_memcpy(got.data(), lf.m.bwt_occ, got.size() * 4, 2, nullptr), "memcpy");
        if (got != h.bwt_occ) FAIL("loaded index: bwt_occ differs from the host-built table");
        std::vector<uint32> rows(1000), pos0, pos1; for (uint32 i = 0; i < 1000; ++i) rows[i] = 1u + (i * 257u) % h.n;
        hip::device_vector<uint32> d_rows(rows), d_pos(1000);
        locate(lf, 1000, d_rows.data(), d_pos.data());
        pos0 = d_pos.to_host();
        for (uint32 i = 0; i < 1000; ++i) if (pos0[i] != h.sa[rows[i]]) FAIL("loaded index: locate(%u) = %u != %u", rows[i], pos0[i], h.sa[rows[i]]);
        remove(bwt_name.c_str()); remove(sa_name.c_str());
        fprintf(stderr, "    %-44s : ok\n", "io::FMIndexDataHost/Device (.bwt/.sa)");
    }
    // nvBowtie's seed mapping stage: the three algorithms of map_t vs the oracle, hit deques compared verbatim
    {
        HostIndex rh; build_host_index(rh, h.n, 11, /*reversed=*/true);
        hip::device_vector<uint32> d_rbwt_occ(rh.bwt_occ), d_rssa(rh.ssa);
        fm_index_device rfmi(rh.n, rh.primary, rh.L2, d_rbwt_occ.data(), d_rssa.data(), 16);
        const uint32 R = 8192, STRIDE = 128;
        std::vector<std::vector<uint8> > reads(R);
        for (uint32 r = 0; r < R; ++r) {
            const uint32 L = 30 + r % 100, p = rnd.next() % (h.n - L);
            reads[r].assign(h.text.begin() + p, h.text.begin() + p + L);
            if (r & 1) { std::reverse(reads[r].begin(), reads[r].end()); for (auto& c : reads[r]) c = uint8(3 - c); }
            for (uint32 j = 0; j < L; ++j) if ((rnd.next() >> 16) % 100 < 3) reads[r][j] = uint8(rnd.sym());
            if (r % 19 == 0) reads[r][r % L] = 4;
            std::reverse(reads[r].begin(), reads[r].end());          // io::REVERSE storage
        }
        PackedStringSetDevice<4, true> d_reads(reads);
        std::vector<uint8> cat; std::vector<uint64> rb(R); std::vector<uint32> rl(R);
        for (uint32 r = 0; r < R; ++r) { rb[r] = cat.size(); rl[r] = uint32(reads[r].size()); cat.insert(cat.end(), reads[r].begin(), reads[r].end()); }
        const std::vector<uint32> rw = pack_symbols<4, true>(cat.data(), cat.size());
        for (int mode = 0; mode < 3; ++mode) {
            bowtie2::cuda::ParamsPOD params;
            params.allow_sub = mode > 0; params.subseed_len = mode == 1 ? 12u : 0u; params.max_hits = 120;
            const std::vector<uint32> sf = params.seed_freq_table(130);
            hip::device_vector<uint32> d_sf(sf), d_counts(R);
            hip::device_vector<bowtie2::cuda::SeedHit> d_hits(size_t(R) * STRIDE);
            hip::device_vector<uint8> d_reseed(R);
            bowtie2::cuda::SeedHitDequeArrayDeviceView hits = { d_hits.data(), STRIDE, d_counts.data() };
            const bowtie2::cuda::PingPongQueuesView queues = { R, nullptr };
            bowtie2::cuda::map(d_reads.view(), fmi, rfmi, 0u, queues, d_reseed.data(), hits, params, d_sf.data(), true, true);
            hip::synchronize();
            const std::vector<bowtie2::cuda::SeedHit> gh = d_hits.to_host(); const std::vector<uint32> gc = d_counts.to_host(); const std::vector<uint8> gr = d_reseed.to_host();
            const oracle_map_params_t op = { params.seed_len, params.min_read_len, params.max_hits, params.max_reseed, 0u, params.rep_seeds, 1u, 1u };
            std::vector<uint64> eh(size_t(R) * STRIDE); std::vector<uint32> ec(R); std::vector<uint8> er(R);
            oracle_map(mode, params.subseed_len, &h.ofmi, &rh.ofmi, rw.data(), 4, 1, rb.data(), rl.data(), nullptr, R, &op, sf.data(), eh.data(), STRIDE, ec.data(), er.data());
            uint64 total = 0;
            for (uint32 r = 0; r < R; ++r) {
                if (gc[r] != ec[r] || gr[r] != er[r]) FAIL("map mode %d read %u: %u hits (reseed %u), expected %u (%u)", mode, r, gc[r], gr[r], ec[r], er[r]);
                std::vector<uint64> a(gc[r]), b(eh.begin() + size_t(r) * STRIDE, eh.begin() + size_t(r) * STRIDE + ec[r]);
                memcpy(a.data(), &gh[size_t(r) * STRIDE], sizeof(uint64) * gc[r]);
                if (a != b) FAIL("map mode %d read %u: hit deques differ", mode, r);          // array order of the reference's deque
                total += gc[r];
            }
            // the selection stage on these deques, through select.h: select_init + rounds of select / locate vs the oracle
            for (int randomized = 0; randomized < 2; ++randomized) {
                bowtie2::cuda::SelectParamsPOD sp; sp.randomized = randomized != 0;
                std::string names; std::vector<uint32> names_idx(R + 1u, 0u);
                for (uint32 r = 0; r < R; ++r) { char nm[32]; snprintf(nm, sizeof(nm), "read.%u", r * 31u); names += nm; names.push_back('\0'); names_idx[r + 1] = uint32(names.size()); }
                hip::device_vector<char> d_names(std::vector<char>(names.begin(), names.end())); hip::device_vector<uint32> d_names_idx(names_idx);
                hip::device_vector<bowtie2::cuda::SeedHit> d_hits2(gh); hip::device_vector<uint32> d_counts2(gc);
                bowtie2::cuda::SeedHitDequeArrayDeviceView hits2 = { d_hits2.data(), STRIDE, d_counts2.data() };
                bowtie2::cuda::SelectState state(R, STRIDE);
                bowtie2::cuda::select_init(R, d_names.data(), d_names_idx.data(), hits2, state, sp);
                const uint32 ps = oracle_sum_tree_node_count(STRIDE);
                std::vector<uint64> oh(eh); std::vector<uint32> oc(ec), otrys(R), orseeds(R); std::vector<float> oprobs(size_t(R) * ps);
                oracle_select_init(R, names.data(), names_idx.data(), oh.data(), STRIDE, oc.data(), oprobs.data(), ps, otrys.data(), orseeds.data(), sp.max_effort_init, randomized, 0);
                const uint32 N_MULTI = 3;
                bowtie2::cuda::ScoringQueues queues(R, R * N_MULTI);
                std::vector<bowtie2::cuda::packed_read> q0(R); std::vector<uint32> oq(R);
                for (uint32 r = 0; r < R; ++r) { q0[r] = bowtie2::cuda::packed_read(r, 0u); oq[r] = r; }
                queues.active_in.assign(q0.data(), R); queues.in_size = R;
                uint64 selected = 0;
                for (int round = 0; round < 4 && queues.in_size; ++round) {
                    bowtie2::cuda::select(hits2, state, queues, N_MULTI, sp);
                    std::vector<uint32> oout(oq.size()), orid(oq.size() * N_MULTI), oloc(oq.size() * N_MULTI), oseed(oq.size() * N_MULTI), osz(2);
                    std::vector<uint64> ohb(oq.size() + 1);
                    oracle_select(randomized, N_MULTI, oq.data(), uint32(oq.size()), oh.data(), STRIDE, oc.data(), oprobs.data(), ps, orseeds.data(), otrys.data(),
                                  oout.data(), ohb.data(), orid.data(), oloc.data(), oseed.data(), osz.data());
                    if (queues.in_size != osz[0] || queues.hits_size != osz[1]) FAIL("select (rand %d) round %d: %u reads / %u hits, expected %u / %u", randomized, round, queues.in_size, queues.hits_size, osz[0], osz[1]);
                    bowtie2::cuda::locate(fmi, rfmi, queues);
                    oracle_locate_hits(&h.ofmi, &rh.ofmi, osz[1], oloc.data(), oseed.data());
                    hip::synchronize();
                    const std::vector<bowtie2::cuda::packed_read> ga = queues.active_in.to_host();
                    const std::vector<uint32> gl = queues.hit_loc.to_host(), gi = queues.hit_read_id.to_host();
                    const std::vector<bowtie2::cuda::packed_seed> gs = queues.hit_seed.to_host();
                    const std::vector<uint64> ghb = queues.hit_begin.to_host();
                    for (uint32 t = 0; t < osz[0]; ++t) { uint32 w; memcpy(&w, &ga[t], 4); if (w != oout[t] || ghb[t] != ohb[t]) FAIL("select (rand %d) round %d: active read %u differs", randomized, round, t); }
                    for (uint32 i = 0; i < osz[1]; ++i) { uint32 w; memcpy(&w, &gs[i], 4); if (gl[i] != oloc[i] || gi[i] != orid[i] || (w & 0x7FFFu) != oseed[i]) FAIL("select (rand %d) round %d: hit %u differs", randomized, round, i); }
                    oq.assign(oout.begin(), oout.begin() + osz[0]);
                    selected += osz[1];
                }
                fprintf(stderr, "    %-44s : %llu hits selected + located ok\n", mode == 0 ? (randomized ? "  select (randomized, 3 hits/read)" : "  select (top of deque, 3 hits/read)") : "  select", (unsigned long long)selected);
            }
            fprintf(stderr, "    %-44s : %u reads, %llu seed hits ok\n", mode == 0 ? "map exact" : mode == 1 ? "map approx (subseed 12)" : "map case-pruning (fwd + rev index)", R, (unsigned long long)total);
        }
    }
    // io::Alignment's bit layout as the kernels write it: invalid() == nvbio_hip_alignment_invalid(), and a reduce + mapq smoke
    {
        const io::Alignment inv = io::Alignment::invalid(); uint64 w; memcpy(&w, &inv, 8);
        if (w != nvbio_hip_alignment_invalid()) FAIL("io::Alignment layout: invalid() = %llx, library %llx", (unsigned long long)w, (unsigned long long)nvbio_hip_alignment_invalid());
        const uint32 R = 4, L = 100;
        bowtie2::cuda::ScoreLimits sc;                         // end-to-end defaults
        hip::device_vector<int32> d_tab(sc.min_score_table(L));
        hip::device_vector<io::Alignment> d_best(2 * R);
        bowtie2::cuda::init_alignments(R, nullptr, L, d_tab.data(), d_best.data(), R);
        // read 0: two distinct hits; read 1: the same location twice; read 2: below the threshold; read 3: none
        const std::vector<uint64> hb = { 0, 2, 4, 5, 5 };
        const std::vector<int32>  hs = { -12, -30, -6, -6, -200 };
        const std::vector<uint32> hl = { 1000, 5000, 777, 777, 42 };
        const std::vector<uint8>  hr = { 0, 1, 0, 0, 0 };
        hip::device_vector<uint64> d_hb(hb); hip::device_vector<int32> d_hs(hs); hip::device_vector<uint32> d_hl(hl); hip::device_vector<uint8> d_hr(hr), d_mq(R);
        bowtie2::cuda::score_reduce(R, nullptr, d_hb.data(), d_hs.data(), d_hl.data(), d_hr.data(), nullptr, L, d_best.data(), R);
        bowtie2::cuda::mapq(2, sc, d_tab.data(), R, d_best.data(), R, nullptr, L, d_mq.data());
        hip::synchronize();
        const std::vector<io::Alignment> b = d_best.to_host(); const std::vector<uint8> mq = d_mq.to_host();
        if (!(b[0].score() == -12 && b[0].alignment() == 1000 && b[R + 0].score() == -30 && b[R + 0].is_rc())) FAIL("score_reduce: read 0");
        if (!(b[1].score() == -6 && !b[R + 1].is_aligned())) FAIL("score_reduce: read 1 (a revisited location must not become the second best)");
        if (b[2].is_aligned() || b[3].is_aligned()) FAIL("score_reduce: reads 2/3 must stay unaligned");
        if (!(mq[1] == 42 && mq[2] == 0 && mq[3] == 0 && mq[0] > 0 && mq[0] < 42)) FAIL("mapq: got %u %u %u %u", mq[0], mq[1], mq[2], mq[3]);
        fprintf(stderr, "    %-44s : ok (mapq %u %u %u %u)\n", "init_alignments / score_reduce / mapq", mq[0], mq[1], mq[2], mq[3]);
    }
    fprintf(stderr, "FM-index test... done\n");
    return 0;
}

int main(int argc, char** argv)
{
    bool aln_t = false, rank_t = false, fm_t = false;
    for (int i = 1; i < argc; ++i) { aln_t |= !strcmp(argv[i], "-aln"); rank_t |= !strcmp(argv[i], "-rank"); fm_t |= !strcmp(argv[i], "-fm-index"); }
    if (!aln_t && !rank_t && !fm_t) aln_t = rank_t = fm_t = true;
    try {
        if (aln_t)  alignment_test();
        if (rank_t) rank_test();
        if (fm_t)   fmindex_test();
    } catch (const nvbio::hip_error& e) { fprintf(stderr, "caught a nvbio::hip_error exception:\n  %s\n", e.what()); return 1; }
    fprintf(stderr, "nvbio_hip_test: all passed\n");
    return 0;
}
