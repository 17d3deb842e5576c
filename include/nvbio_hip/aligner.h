// nvbio_hip/aligner.h -- nvBowtie's single-end best-mapping driver over libnvbio_hip.so, in the reference's language.
// Mirrors nvBowtie/bowtie2/cuda/aligner.h (struct Aligner: BATCH_SIZE, best_data_dvec, mapq_dvec, cigar storage,
// band_length) and aligner_best_approx.h (Aligner::best_approx :85-520, Aligner::best_approx_score :522-840): the same
// control flow over the C-ABI stages -- init_alignments, per seeding pass map -> select_init -> extension rounds of
// {select, locate, score_best, score_reduce}, mark_unaligned + copy_flagged into the re-seed queue, BowtieMapq2,
// banded_traceback_best.  Inputs are device resident (the reference's io::SequenceDataDevice / FMIndexDataDevice).
#pragma once
#include <mutex>
#include <algorithm>
#include <chrono>
#include <map>
#include <numeric>
#include <string>
#include <vector>
#include "alignment.h"
#include "mapping.h"
#include "reduce.h"
#include "select.h"

namespace nvbio {
namespace bowtie2 {
namespace cuda {

enum AlignmentTypeMode { EndToEndAlignment = 0, LocalAlignment = 1 };           // params.h
enum ScoringMode       { EditDistanceMode = 0, SmithWatermanMode = 1 };         // params.h:47-51

/// the fields of nvBowtie's Params the driver reads, with its defaults (params.cpp:116-197)
struct Params : public ParamsPOD
{
    Params() : max_dist(15), alignment_type(EndToEndAlignment), no_multi_hits(false), fw(true), rc(true), hits_stride(0), finish_alignments(true),
               scoring_mode(SmithWatermanMode) {}
    SelectParamsPOD select;
    uint32 max_dist; AlignmentTypeMode alignment_type; bool no_multi_hits, fw, rc; uint32 hits_stride;
    bool   finish_alignments;      ///< run finish_alignment_best (MD strings, edit distances, final scores) as the reference always does
    /// --scoring ed|sw (params.cpp:117, compute_thread.cu:296): in edit-distance mode hits are extended, reduced and traced with the
    /// edit-distance aligner against score-min = -max_dist (params.cpp:203-204); MAPQ and the final scores of finish_alignment still come
    /// from the Smith-Waterman scheme the caller passes (aligner_best_approx.h:294-296, traceback.cu:146-160), as in the reference
    ScoringMode scoring_mode;
    /// the search scheme and thresholds of the mode
    aln::SmithWatermanScoringScheme search_scheme(const aln::SmithWatermanScoringScheme& sw) const { return scoring_mode == EditDistanceMode ? aln::SmithWatermanScoringScheme::edit_distance() : sw; }
    ScoreLimits search_limits(const ScoreLimits& sw) const { return scoring_mode == EditDistanceMode ? ScoreLimits(0, SimpleFunc(SimpleFunc::LinearFunc, -float(max_dist), 0.0f)) : sw; }

    /// Slots of a read's hit deque when the caller names none (hits_stride == 0).  The reference's deques hold max_hits (100) ranges
    /// (seed_hit_deque_array.h); a read of length L yields at most one range per seed and strand under exact seeding -- 2 * ((L - seed_len) /
    /// seed_freq(L) + 1): 14 at 100 bp, 28 at 150 bp with --local -- so a row of 16 or 32 slots holds every range the reference's would, and the
    /// selection stage works on 128- or 256-byte rows with its tree in LDS instead of 100-slot rows with the tree in memory (select.hip).
    /// With one-mismatch seeding (allow_sub) a seed can yield several ranges: the reference's capacity stands.
    uint32 resolved_hits_stride(const uint32 max_read_len) const
    {
        if (hits_stride) return hits_stride;
        const uint32 cap = std::min(max_hits, 128u);
        if (allow_sub) return cap;
        uint32 most = 0;
        for (uint32 L = std::max(min_read_len, 1u); L <= max_read_len; ++L)
        {
            const int32 f = seed_freq(int32(L));
            if (f <= 0) continue;
            most = std::max(most, 2u * ((L - std::min(seed_len, L)) / uint32(f) + 1u));
        }
        const uint32 rows = most <= 16u ? 16u : most <= 32u ? 32u : most;
        return std::min(cap, rows);
    }

    /// switch between end-to-end and local alignment the way nvBowtie's option parser does (params.cpp:156-160): the alignment type
    /// also moves the seeding defaults -- 22-bp seeds every 1 + 1.15 sqrt(L) end-to-end, 20-bp seeds every 1 + 0.75 sqrt(L) local
    void set_alignment_type(const AlignmentTypeMode type)
    {
        alignment_type = type;
        seed_len  = type == LocalAlignment ? 20u : 22u;
        seed_freq = SimpleFunc(SimpleFunc::SqrtFunc, 1.0f, type == LocalAlignment ? 0.75f : 1.15f);
    }
};

/// per-stage device times (the reference's stats.map / select / locate / score ... timers, aligner_best_approx.h): off by default,
/// since a stage boundary costs two stream synchronisations
struct StageClock
{
    StageClock() : enabled(false) {}
    bool enabled;
    std::map<std::string, double> ms;
    std::chrono::steady_clock::time_point t0;
    void begin(const char*, void* s) { if (!enabled) return; hip::synchronize(s); t0 = std::chrono::steady_clock::now(); }
    void end(const char* name, void* s) { if (!enabled) return; hip::synchronize(s); ms[name] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
    template <typename F> void run(const char* name, void* s, F f) { begin(name, s); f(); end(name, s); }
};

struct Stats { uint64 extensions, dp_jobs, hits, ranges, unique; uint32 rounds, seeding_passes; std::vector<uint32> queue; StageClock clock;
               Stats() : extensions(0), dp_jobs(0), hits(0), ranges(0), unique(0), rounds(0), seeding_passes(0) {} };

/// A batch of equal-length reads on the device in the layouts the stages read (io::SequenceDataDevice's role): the reads
/// stored reversed (io::REVERSE, what the mappers scan), their forward copies followed rc_offset symbols later by their
/// reverse complements (the extension / traceback patterns), one quality byte per pattern symbol, the read names.
struct ReadBatch
{
    ReadBatch() : n(0), len(0), fw_rc_words(nullptr), fw_rc_n_words(0), rc_offset(0), quals(nullptr), n_quals(0), names(nullptr), names_idx(nullptr),
                  read_begin(nullptr), read_len(nullptr) {}
    uint32 n, len;                  ///< len: every read's length, or -- with read_len -- the longest
    PackedStringSetView<4, true> reversed;
    const uint32* fw_rc_words; uint64 fw_rc_n_words; uint64 rc_offset;
    const uint8*  quals;       uint64 n_quals;
    const char*   names;       const uint32* names_idx;
    /// reads of their own lengths (device arrays, NULL for an equal-length batch): read r's forward copy starts read_begin[r] symbols into
    /// fw_rc_words and is read_len[r] long, its reverse complement rc_offset symbols further; `reversed` carries the same lengths
    const uint64* read_begin;  const uint32* read_len;
    uint32 fixed() const { return read_len ? 0u : len; }
};

/// the paired-end fields of Params (params.cpp:165-172; io::PairedEndPolicy FF 0, FR 1, RF 2, RR 3)
struct PairedParams
{
    PairedParams() : pe_policy(1), pe_overlap(true), pe_unpaired(true), pe_discordant(true), min_frag_len(0), max_frag_len(500) {}
    int32 pe_policy; bool pe_overlap, pe_unpaired, pe_discordant; uint32 min_frag_len, max_frag_len;
};

/// Two mate batches of equal size and read length.  The tracebacks pick a read by the alignment's mate bit, so they read ONE pattern
/// stream holding both mates: mate m's fw + rc copies start m * mate_offset symbols into `both_words` (qualities likewise).
struct PairedReadBatch
{
    ReadBatch     mate[2];
    const uint32* both_words; uint64 both_n_words; uint64 mate_offset;
    const uint8*  both_quals; uint64 both_n_quals;
};

struct Aligner
{
    static const int32 worst_score = -(1 << 16);         // SmithWatermanScoringScheme::worst_score (scoring.h:226-227)

    uint32                            BATCH_SIZE;        // capacity, and the stride of best_data
    uint32                            SCORING_BATCH;     // the batch size the hits-per-read rule reasons with: BATCH_SIZE in the reference
                                                         // (kept apart so that small test batches can reach the one-hit-per-round regime)
    hip::device_vector<io::Alignment> best_data_dvec;    // [2][BATCH_SIZE]
    hip::device_vector<uint8>         mapq_dvec;
    hip::device_vector<io::Cigar>     cigar;             // [BATCH_SIZE][cigar_stride], end of the alignment first
    hip::device_vector<uint32>        cigar_len, cigar_source, cigar_sink;   // cigar_coords + sinks of the tracebacks
    hip::device_vector<int32>         traceback_score;
    uint32                            cigar_stride;
    hip::device_vector<uint8>         mds;               // [BATCH_SIZE][mds_stride]: MD strings in nvbio's byte code (finish_alignment)
    hip::device_vector<uint32>        mds_len;
    uint32                            mds_stride;
    // paired-end: the opposite slot set (best_data_dvec_o of the reference) and mate 2's MAPQ
    hip::device_vector<io::Alignment> best_data_dvec_o;
    hip::device_vector<uint8>         mapq_dvec_o;
    hip::device_vector<io::Cigar>     cigar_o;
    hip::device_vector<uint32>        cigar_len_o, cigar_source_o, cigar_sink_o, mds_len_o;
    hip::device_vector<int32>         traceback_score_o;
    hip::device_vector<uint8>         mds_o;

    // all-mapping (Aligner::all): the reported alignments, in batch order, sorted by (read, strand, position) inside a batch;
    // their CIGARs / MD strings land in cigar, cigar_len, cigar_source, cigar_sink, traceback_score, mds, mds_len (resized to fit)
    uint64                            n_alignments;
    hip::device_vector<uint32>        output_read_info_dvec;     // read id of every alignment
    hip::device_vector<io::Alignment> output_alignments_dvec;    // finished alignments (window begin, edit distance, final score)
    hip::device_vector<io::Alignment> scored_alignments_dvec;    // as accepted (read start, extension score)

    hip::device_arena                 workspace;                 // the per-batch queues and temporaries of the best-mapping drivers

    /// Several Aligner objects, one per host thread and HIP stream, may share one device (include/nvbio_hip.h, "Streams"): seeding
    /// (map / locate) is bound by the fabric's random-line rate with idle VALUs, everything else is VALU-bound with an idle fabric, so two
    /// batches in flight overlap the two kinds (+7-10 %).  Nothing else is needed for that: the seeding stages are queued on the batch's own
    /// stream like every other stage (a token that serialised the seeding stages of co-resident Aligners, and a CU-masked seeding stream,
    /// were tried in round 3 and removed -- neither beat plain streams).
    Aligner() : BATCH_SIZE(0), SCORING_BATCH(0), cigar_stride(64), mds_stride(256), n_alignments(0) { check_abi(); }

    /// run a seeding stage: f(stream) queues its kernels on the stream it is given
    template <typename F> void fabric_bound(void* hip_stream, F f) { f(hip_stream); }

    /// Aligner::band_length (aligner.h:165-174)
    static uint32 band_length(const uint32 max_dist)
    {
        uint32 band_len = 4;
        while (band_len - 1 < max_dist * 2 + 1) band_len *= 2;
        return band_len - 1;
    }

    bool init(const uint32 batch_size, const uint32 scoring_batch = 0)
    {
        BATCH_SIZE = batch_size; SCORING_BATCH = scoring_batch ? scoring_batch : batch_size;
        best_data_dvec.resize(size_t(batch_size) * 2u); mapq_dvec.resize(batch_size);
        cigar.resize(size_t(batch_size) * cigar_stride); cigar_len.resize(batch_size);
        cigar_source.resize(size_t(batch_size) * 2u); cigar_sink.resize(size_t(batch_size) * 2u); traceback_score.resize(batch_size);
        mds.resize(size_t(batch_size) * mds_stride); mds_len.resize(batch_size);
        return true;
    }

    /// the additional storage of paired-end runs (Aligner::init with EndType PAIRED_END)
    bool init_paired()
    {
        const uint32 b = BATCH_SIZE;
        best_data_dvec_o.resize(size_t(b) * 2u); mapq_dvec_o.resize(b); cigar_o.resize(size_t(b) * cigar_stride); cigar_len_o.resize(b);
        cigar_source_o.resize(size_t(b) * 2u); cigar_sink_o.resize(size_t(b) * 2u); traceback_score_o.resize(b); mds_o.resize(size_t(b) * mds_stride); mds_len_o.resize(b);
        return true;
    }

    /// Aligner::best_approx for read pairs (aligner_best_approx_paired.h:95-453)
    void best_approx(const Params& params, const PairedParams& pe, const fm_index_device& fmi, const fm_index_device& rfmi,
                     const aln::SmithWatermanScoringScheme& scoring_scheme, const ScoreLimits& limits,
                     const uint32* genome_words, const uint64 genome_n_words, const uint32 genome_len, const PairedReadBatch& reads, Stats& stats, void* hip_stream = nullptr)
    {
        if (params.alignment_type == LocalAlignment) best_approx_paired_t<aln::LOCAL>(params, pe, fmi, rfmi, scoring_scheme, limits, genome_words, genome_n_words, genome_len, reads, stats, hip_stream);
        else                                         best_approx_paired_t<aln::SEMI_GLOBAL>(params, pe, fmi, rfmi, scoring_scheme, limits, genome_words, genome_n_words, genome_len, reads, stats, hip_stream);
    }

    /// Aligner::best_approx (aligner_best_approx.h:85-520)
    void best_approx(const Params& params, const fm_index_device& fmi, const fm_index_device& rfmi, const aln::SmithWatermanScoringScheme& scoring_scheme,
                     const ScoreLimits& limits, const uint32* genome_words, const uint64 genome_n_words, const uint32 genome_len,
                     const ReadBatch& reads, Stats& stats, void* hip_stream = nullptr)
    {
        if (params.alignment_type == LocalAlignment) best_approx_t<aln::LOCAL>(params, fmi, rfmi, scoring_scheme, limits, genome_words, genome_n_words, genome_len, reads, stats, hip_stream);
        else                                         best_approx_t<aln::SEMI_GLOBAL>(params, fmi, rfmi, scoring_scheme, limits, genome_words, genome_n_words, genome_len, reads, stats, hip_stream);
    }

    /// Aligner::all + score_all (aligner_all.h:47-227, :264-694): every placement of every read that reaches min_score(read_len).
    /// sequence_index: the offsets of the reference's sequences (n_sequences + 1 entries; {0, genome_len} for one sequence).
    void all(const Params& params, const fm_index_device& fmi, const fm_index_device& rfmi, const aln::SmithWatermanScoringScheme& scoring_scheme,
             const ScoreLimits& limits, const uint32* genome_words, const uint64 genome_n_words, const uint32 genome_len,
             const std::vector<uint32>& sequence_index, const ReadBatch& reads, Stats& stats, void* hip_stream = nullptr)
    {
        if (params.alignment_type == LocalAlignment) all_t<aln::LOCAL>(params, fmi, rfmi, scoring_scheme, limits, genome_words, genome_n_words, genome_len, sequence_index, reads, stats, hip_stream);
        else                                         all_t<aln::SEMI_GLOBAL>(params, fmi, rfmi, scoring_scheme, limits, genome_words, genome_n_words, genome_len, sequence_index, reads, stats, hip_stream);
    }

private:
    template <aln::AlignmentType TYPE>
    void all_t(const Params& params, const fm_index_device& fmi, const fm_index_device& rfmi, const aln::SmithWatermanScoringScheme& scoring_scheme,
               const ScoreLimits& limits, const uint32* genome_words, const uint64 genome_n_words, const uint32 genome_len,
               const std::vector<uint32>& sequence_index, const ReadBatch& reads, Stats& stats, void* hip_stream)
    {
        const uint32 count = reads.n, L = reads.len, B = SCORING_BATCH;
        const uint32 band_len = band_length(params.max_dist);
        const uint32 hits_stride = params.resolved_hits_stride(L);
        // (Params::scoring_mode: hits scored and traced with the edit-distance aligner, finished with the caller's scheme -- see best_approx_t)
        const bool ed_mode = params.scoring_mode == EditDistanceMode;
        const aln::GotohAligner<TYPE, aln::SmithWatermanScoringScheme> aligner(params.search_scheme(scoring_scheme));
        n_alignments = 0;
        if (count == 0) return;

        // map_kernel searches seeds 0 .. max_seeds-1 (aligner_all.h:93-95, mapping_inl.h:659); map(retry 0) searches every seed that
        // fits: the same set unless the cap binds, which is refused rather than mapped differently
        {
            const int32 f = params.seed_freq(int32(L));
            const uint32 max_seeds = f > 0 ? L / uint32(f) : 0u;
            if (L >= params.min_read_len && f > 0 && (L - std::min(params.seed_len, L)) / uint32(f) + 1u > max_seeds)
                throw std::runtime_error("Aligner::all: the reference's seed cap would drop seeds at this seeding interval (unsupported)");
        }
        hip::device_vector<int32>   min_score_table(params.search_limits(limits).min_score_table(L));
        std::vector<uint32> iota(std::max(count, B)); std::iota(iota.begin(), iota.end(), 0u);
        hip::device_vector<uint32>  d_iota(iota);
        hip::device_vector<SeedHit> hit_data(size_t(count) * hits_stride);
        hip::device_vector<uint32>  hit_counts(count), hit_count_scan(count);
        hip::device_vector<uint8>   reseed(count);
        hip::device_vector<uint32>  seed_freq(params.seed_freq_table(L));
        SeedHitDequeArrayDeviceView hits = { hit_data.data(), hits_stride, hit_counts.data() };
        hip_check(nvbio_hip_memset(hit_counts.data(), 0, uint64(count) * 4u, hip_stream), "nvbio_hip_memset");
        const PingPongQueuesView seed_queues = { count, d_iota.data() };
        map_all(reads.reversed, fmi, rfmi, seed_queues, reseed.data(), hits, params, seed_freq.data(), params.fw, params.rc, hip_stream);

        // scan the deque sizes, gather and scan the range sizes
        uint32 n_ranges = 0;
        {
            hip::device_vector<uint8> temp(nvbio_hip_all_mapping_temp_bytes(count));
            hip_check(nvbio_hip_inclusive_scan_u32(count, hit_counts.data(), hit_count_scan.data(), temp.data(), temp.size(), hip_stream), "nvbio_hip_inclusive_scan_u32");
            hip::synchronize(hip_stream);
            hip_check(nvbio_hip_memcpy(&n_ranges, hit_count_scan.data() + (count - 1u), 4u, 2, hip_stream), "nvbio_hip_memcpy(d2h)");
        }
        stats.ranges = n_ranges;
        if (n_ranges == 0) return;
        hip::device_vector<uint64> hit_range_scan(n_ranges);
        uint64 n_hits = 0;
        {
            hip::device_vector<uint64> ranges(n_ranges);
            hip::device_vector<uint8>  temp(nvbio_hip_all_mapping_temp_bytes(n_ranges));
            hip_check(nvbio_hip_gather_ranges(n_ranges, count, reinterpret_cast<const uint64*>(hit_data.data()), hits_stride, hit_count_scan.data(), ranges.data(), hip_stream), "nvbio_hip_gather_ranges");
            hip_check(nvbio_hip_inclusive_scan_u64(n_ranges, ranges.data(), hit_range_scan.data(), temp.data(), temp.size(), hip_stream), "nvbio_hip_inclusive_scan_u64");
            hip::synchronize(hip_stream);
            hip_check(nvbio_hip_memcpy(&n_hits, hit_range_scan.data() + (n_ranges - 1u), 8u, 2, hip_stream), "nvbio_hip_memcpy(d2h)");
        }
        stats.hits = n_hits;

        // the hit queues of a batch and the alignment buffer (every row can yield at most one alignment per batch)
        const uint32 cap = uint32(std::min<uint64>(n_hits, B));
        hip::device_vector<uint32> hit_loc(cap), hit_seed(cap), hit_read(cap), idx_queue(cap), sort_idx(cap), queue(cap), accepted(cap), counter(1);
        hip::device_vector<uint8>  flags(cap), sort_temp(nvbio_hip_all_mapping_temp_bytes(cap)), flag_temp(nvbio_hip_copy_flagged_temp_bytes(cap));
        hip::device_vector<uint64> pat_begin(cap), txt_begin(cap);
        hip::device_vector<uint32> txt_len(cap), sinks(size_t(cap) * 2u), job_read(cap);
        hip::device_vector<int32>  hit_score(cap);
        hip::device_vector<io::Alignment> job_aln(cap);
        hip::device_vector<uint32> d_seq_index(sequence_index);
        // the alignment buffer grows as batches report (the reference streams them out through a 2 x BATCH_SIZE ring instead)
        uint64 capacity = 0;
        auto reserve = [&](const uint64 need) {
            if (need <= capacity) return;
            const uint64 grown = std::max<uint64>(need, std::max<uint64>(2u * capacity, 2u * uint64(B)));
            hip::device_vector<io::Alignment> a(grown); hip::device_vector<uint32> r(grown);
            if (n_alignments) {
                hip_check(nvbio_hip_memcpy(a.data(), scored_alignments_dvec.data(), n_alignments * 8u, 3, hip_stream), "nvbio_hip_memcpy(d2d)");
                hip_check(nvbio_hip_memcpy(r.data(), output_read_info_dvec.data(), n_alignments * 4u, 3, hip_stream), "nvbio_hip_memcpy(d2d)");
                hip::synchronize(hip_stream);
            }
            std::swap(a.m_ptr, scored_alignments_dvec.m_ptr); std::swap(a.m_size, scored_alignments_dvec.m_size);
            std::swap(r.m_ptr, output_read_info_dvec.m_ptr);  std::swap(r.m_size, output_read_info_dvec.m_size);
            capacity = grown;
        };

        for (uint64 hit_offset = 0; hit_offset < n_hits; hit_offset += B)
        {
            const uint32 hit_count = uint32(std::min<uint64>(n_hits - hit_offset, B));
            hip_check(nvbio_hip_select_all(hit_offset, hit_count, count, n_ranges, reinterpret_cast<const uint64*>(hit_data.data()), hits_stride, hit_count_scan.data(),
                                           hit_range_scan.data(), hit_loc.data(), hit_seed.data(), hit_read.data(), hip_stream), "nvbio_hip_select_all");
            // sort_hi_bits, locate, sort by (read, strand, position), dedup, straddling marks, compaction
            hip_check(nvbio_hip_locate_hits(&fmi.m, &rfmi.m, hit_count, hit_loc.data(), hit_seed.data(), hip_stream), "nvbio_hip_locate_hits");
            // (idx_queue: what the reference's stale `pipeline.idx_queue` holds when mark_straddling reads it -- the half of the ping-pong index buffer
            // sort_hi_bits ended in, as sort_64_bits left it; nvbio_hip.h)
            hip_check(nvbio_hip_sort_hits_pingpong(hit_count, hit_read.data(), hit_loc.data(), hit_seed.data(), sort_idx.data(), flags.data(), idx_queue.data(),
                                                   sort_temp.data(), sort_temp.size(), hip_stream), "nvbio_hip_sort_hits_pingpong");
            hip_check(nvbio_hip_mark_straddling(hit_count, idx_queue.data(), uint32(sequence_index.size() - 1u), d_seq_index.data(), hit_loc.data(), params.seed_len, flags.data(), hip_stream),
                      "nvbio_hip_mark_straddling");
            hip_check(nvbio_hip_copy_flagged(hit_count, sort_idx.data(), flags.data(), queue.data(), counter.data(), flag_temp.data(), flag_temp.size(), hip_stream), "nvbio_hip_copy_flagged");
            hip::synchronize(hip_stream);
            const uint32 queue_size = counter.to_host(hip_stream)[0];
            stats.unique += queue_size;
            if (queue_size == 0) continue;

            // score_all: windows, the banded scorer, acceptance at min_score(read_len)
            hip::device_vector<uint32> pat_len(reads.read_len ? queue_size : 0u);                 // reads of their own lengths: the jobs' pattern lengths
            uint32* const pat_len_ptr = reads.read_len ? pat_len.data() : nullptr;
            hip_check(nvbio_hip_score_all_setup(queue_size, queue.data(), hit_read.data(), hit_loc.data(), hit_seed.data(), reads.read_begin, reads.read_len, reads.fixed(), reads.rc_offset, band_len, genome_len,
                                                pat_begin.data(), pat_len_ptr, txt_begin.data(), txt_len.data(), hip_stream), "nvbio_hip_score_all_setup");
            {
                const PackedStringSetView<4, true> patterns(queue_size, reads.fw_rc_words, reads.fw_rc_n_words, pat_begin.data(), pat_len_ptr, reads.fixed());
                const PackedStringSetView<2, true> texts(queue_size, genome_words, genome_n_words, txt_begin.data(), txt_len.data(), 0u);
                const aln::BestSinkArrays sink_arrays = { hit_score.data(), sinks.data() };
                dispatch_band(band_len, [&](auto band) {
                    aln::batch_banded_alignment_score<decltype(band)::value>(aligner, patterns, reads.quals, reads.n_quals, texts, sink_arrays, L, L + band_len, hip_stream);
                });
            }
            hip_check(nvbio_hip_score_all_output(queue_size, queue.data(), hit_read.data(), hit_loc.data(), hit_seed.data(), hit_score.data(), min_score_table.data(), reads.read_len, reads.fixed(),
                                                 flags.data(), reinterpret_cast<uint64*>(job_aln.data()), job_read.data(), hip_stream), "nvbio_hip_score_all_output");
            hip_check(nvbio_hip_copy_flagged(queue_size, d_iota.data(), flags.data(), accepted.data(), counter.data(), flag_temp.data(), flag_temp.size(), hip_stream), "nvbio_hip_copy_flagged");
            hip::synchronize(hip_stream);
            const uint32 n_accepted = counter.to_host(hip_stream)[0];
            reserve(n_alignments + n_accepted);
            hip_check(nvbio_hip_gather_rows(n_accepted, accepted.data(), job_aln.data(), scored_alignments_dvec.data() + n_alignments, 8u, hip_stream), "nvbio_hip_gather_rows");
            hip_check(nvbio_hip_gather_rows(n_accepted, accepted.data(), job_read.data(), output_read_info_dvec.data() + n_alignments, 4u, hip_stream), "nvbio_hip_gather_rows");
            n_alignments += n_accepted;
        }
        hip::synchronize(hip_stream);
        if (n_alignments == 0) return;

        // banded_traceback_all + finish_alignment_all, BATCH_SIZE alignments at a time
        const uint64 m = n_alignments;
        output_alignments_dvec.resize(m);
        hip_check(nvbio_hip_memcpy(output_alignments_dvec.data(), scored_alignments_dvec.data(), m * 8u, 3, hip_stream), "nvbio_hip_memcpy(d2d)");
        cigar.resize(size_t(m) * cigar_stride); cigar_len.resize(m); cigar_source.resize(size_t(m) * 2u); cigar_sink.resize(size_t(m) * 2u); traceback_score.resize(m);
        mds.resize(size_t(m) * mds_stride); mds_len.resize(m);
        hip_check(nvbio_hip_memset(cigar.data(), 0, m * cigar_stride * sizeof(io::Cigar), hip_stream), "nvbio_hip_memset");
        hip_check(nvbio_hip_memset(mds.data(), 0, m * mds_stride, hip_stream), "nvbio_hip_memset");
        const uint32 tcap = uint32(std::min<uint64>(m, B));
        hip::device_vector<uint64> tb_pat(tcap), tb_txt(tcap);
        hip::device_vector<uint32> tb_len(tcap);
        hip::device_vector<uint8>  valid(std::vector<uint8>(tcap, 1u));
        const nvbio_hip_gotoh_qual_scheme sc = scoring_scheme.abi();
        for (uint64 off = 0; off < m; off += B)
        {
            const uint32 nb = uint32(std::min<uint64>(m - off, B));
            uint64* a = reinterpret_cast<uint64*>(output_alignments_dvec.data() + off);
            hip::device_vector<uint32> tb_plen(reads.read_len ? nb : 0u);
            uint32* const tb_plen_ptr = reads.read_len ? tb_plen.data() : nullptr;
            hip_check(nvbio_hip_traceback_all_setup(nb, a, output_read_info_dvec.data() + off, reads.read_begin, reads.read_len, reads.fixed(), reads.rc_offset, band_len, genome_len,
                                                    tb_pat.data(), tb_plen_ptr, tb_txt.data(), tb_len.data(), hip_stream), "nvbio_hip_traceback_all_setup");
            const PackedStringSetView<4, true>  patterns(nb, reads.fw_rc_words, reads.fw_rc_n_words, tb_pat.data(), tb_plen_ptr, reads.fixed());
            const PackedStringSetView<2, true>  texts(nb, genome_words, genome_n_words, tb_txt.data(), tb_len.data(), 0u);
            const aln::AlignmentArrays alignments = { traceback_score.data() + off, cigar_source.data() + 2u * off, cigar_sink.data() + 2u * off };
            const aln::CigarArrays     cigars     = { cigar.data() + off * cigar_stride, cigar_stride, cigar_len.data() + off };
            if (ed_mode)
            {
                const nvbio_hip_sw_scheme w = { 0, -1, -1, -1 };
                const nvbio_hip_string_set p = patterns.abi(), t = texts.abi();
                const uint32 bl = band_len < 4 ? 3u : band_len < 8 ? 7u : band_len < 16 ? 15u : 31u;
                hip::device_vector<uint8> temp(nvbio_hip_banded_gotoh_traceback_temp_bytes(bl, L, nb));
                hip_check(nvbio_hip_banded_sw_traceback(&w, int32(TYPE), bl, &p, &t, L, L + band_len, nb, alignments.score, alignments.sink, alignments.source,
                                                        reinterpret_cast<uint16*>(cigars.cigar), cigar_stride, cigars.cigar_len, temp.data(), temp.size(), hip_stream),
                          "nvbio_hip_banded_sw_traceback");
                hip::synchronize(hip_stream);
            }
            else
            dispatch_band(band_len, [&](auto band) {
                typedef aln::PackedTracebackStream<aln::GotohAligner<TYPE, aln::SmithWatermanScoringScheme>, PackedStringSetView<4, true>, PackedStringSetView<2, true> > stream_type;
                const stream_type stream(aligner, patterns, texts, alignments, cigars, L, L + band_len, reads.quals, reads.n_quals);
                typedef aln::BatchedBandedAlignmentTraceback<decltype(band)::value, 32u, stream_type> batch_type;
                hip::device_vector<uint8> temp(batch_type::min_temp_storage(L, L + band_len, nb));
                batch_type().enact(stream, temp.size(), temp.data(), hip_stream);
                hip::synchronize(hip_stream);
            });
            const nvbio_hip_string_set p = patterns.abi(), t = texts.abi();
            hip_check(nvbio_hip_finish_alignment(nb, valid.data(), &p, reads.quals, reads.n_quals, &t, reinterpret_cast<const uint16*>(cigar.data() + off * cigar_stride), cigar_stride,
                                                 cigar_len.data() + off, cigar_source.data() + 2u * off, sc.match, sc.mismatch, 1, &sc.pattern_gap_open, nullptr, a,
                                                 mds.data() + off * mds_stride, mds_stride, mds_len.data() + off, hip_stream), "nvbio_hip_finish_alignment");
            hip::synchronize(hip_stream);
        }
    }

    template <aln::AlignmentType TYPE>
    void best_approx_t(const Params& params, const fm_index_device& fmi, const fm_index_device& rfmi, const aln::SmithWatermanScoringScheme& scoring_scheme,
                       const ScoreLimits& limits, const uint32* genome_words, const uint64 genome_n_words, const uint32 genome_len,
                       const ReadBatch& reads, Stats& stats, void* hip_stream)
    {
        const uint32 count = reads.n, L = reads.len;
        const uint32 band_len = band_length(params.max_dist);
        const uint32 hits_stride = params.resolved_hits_stride(L);
        // the scheme hits are extended with: the caller's, or the edit-distance costs (Params::scoring_mode)
        const bool ed_mode = params.scoring_mode == EditDistanceMode;
        const aln::SmithWatermanScoringScheme search_scheme = params.search_scheme(scoring_scheme);
        const ScoreLimits search_limits = params.search_limits(limits);
        const aln::GotohAligner<TYPE, aln::SmithWatermanScoringScheme> aligner(search_scheme);
        hip::synchronize(hip_stream);                                   // nothing of the previous batch may still read the workspace
        const hip::arena_scope scope(workspace);                        // every vector below lives in the Aligner's workspace

        // initialize best-alignments with the threshold score
        hip::device_vector<int32> min_score_table(search_limits.min_score_table(L));
        hip::device_vector<int32> mapq_score_table(limits.min_score_table(L));        // MAPQ reads the Smith-Waterman scheme in either mode
        init_alignments(count, reads.read_len, reads.fixed(), min_score_table.data(), best_data_dvec.data(), BATCH_SIZE, 0u, hip_stream);

        // the seed queue, hit deques, selection state and scoring queues of the pipeline
        hip::device_vector<uint32> seed_queue_in(count), seed_queue_out(count), queue_count(1);
        hip_check(nvbio_hip_pack_read_queue(count, nullptr, 0u, seed_queue_in.data(), hip_stream), "nvbio_hip_pack_read_queue");      // 0 .. count-1
        uint32 seed_queue_size = count;
        hip::device_vector<SeedHit> hit_data(size_t(count) * hits_stride);
        hip::device_vector<uint32>  hit_counts(count);
        hip::device_vector<uint8>   reseed(count);
        hip::device_vector<uint32>  seed_freq(params.seed_freq_table(L));
        SelectState   state(count, hits_stride);
        const uint32  max_hits_per_round = std::max(SCORING_BATCH, count);
        ScoringQueues queues(count, max_hits_per_round);
        hip::device_vector<uint64> pat_begin(max_hits_per_round), txt_begin(max_hits_per_round);
        hip::device_vector<uint32> txt_len(max_hits_per_round), sinks(size_t(max_hits_per_round) * 2u);
        hip::device_vector<int32>  min_score(max_hits_per_round), hit_score(max_hits_per_round);
        hip::device_vector<uint8>  flag_temp(nvbio_hip_copy_flagged_temp_bytes(count));
        SeedHitDequeArrayDeviceView hits = { hit_data.data(), hits_stride, hit_counts.data() };
        // the DP sink of every read's best alignment, kept by the reduction: the traceback re-scores the very job the extension scored,
        // so it starts from that score and sink instead (nvbio_hip.h)
        hip::device_vector<uint32> best_sink(size_t(count) * 2u);
        hip_check(nvbio_hip_memset(best_sink.data(), 0xFF, uint64(count) * 8u, hip_stream), "nvbio_hip_memset");

        for (uint32 seeding_pass = 0; seeding_pass < params.max_reseed + 1; ++seeding_pass)
        {
            if (seed_queue_size == 0) break;
            stats.queue.push_back(seed_queue_size); ++stats.seeding_passes;

            // hit_deques.clear_deques() + map
            hip_check(nvbio_hip_memset(hit_counts.data(), 0, uint64(count) * 4u, hip_stream), "nvbio_hip_memset");
            const PingPongQueuesView seed_queues = { seed_queue_size, seed_queue_in.data() };
            stats.clock.run("map", hip_stream, [&] { fabric_bound(hip_stream, [&](void* ss) { map(reads.reversed, fmi, rfmi, seeding_pass, seed_queues, reseed.data(), hits, params, seed_freq.data(), params.fw, params.rc, ss); }); });

            best_approx_score<TYPE>(params, fmi, rfmi, aligner, genome_words, genome_n_words, genome_len, reads, band_len, seed_queue_size, seed_queue_in.data(),
                                    hits, state, queues, pat_begin, txt_begin, txt_len, sinks, min_score, hit_score, stats, hip_stream, best_sink.data());

            // mark unaligned reads, copy the reads that need reseeding, swap the queues
            hip_check(nvbio_hip_mark_unaligned(seed_queue_size, seed_queue_in.data(), reinterpret_cast<const uint64*>(best_data_dvec.data()), reseed.data(), hip_stream), "nvbio_hip_mark_unaligned");
            hip_check(nvbio_hip_copy_flagged(seed_queue_size, seed_queue_in.data(), reseed.data(), seed_queue_out.data(), queue_count.data(),
                                             flag_temp.data(), flag_temp.size(), hip_stream), "nvbio_hip_copy_flagged");
            hip::synchronize(hip_stream);
            seed_queue_size = queue_count.to_host(hip_stream)[0];
            std::swap(seed_queue_in.m_ptr, seed_queue_out.m_ptr);
        }

        // compute mapq (BowtieMapq2)
        stats.clock.run("mapq", hip_stream, [&] { mapq(2, limits, mapq_score_table.data(), count, best_data_dvec.data(), BATCH_SIZE, reads.read_len, reads.fixed(), mapq_dvec.data(), hip_stream); });

        // banded_traceback_best over every read (unaligned ones get an empty window and no CIGAR)
        stats.clock.begin("traceback", hip_stream);
        {
            hip::device_vector<uint8>  valid(count);
            hip::device_vector<uint64> tb_pat(count), tb_txt(count);
            hip::device_vector<uint32> tb_len(count);
            hip_check(nvbio_hip_memset(cigar.data(), 0, uint64(count) * cigar_stride * sizeof(io::Cigar), hip_stream), "nvbio_hip_memset");
            hip::device_vector<uint32> tb_plen(reads.read_len ? count : 0u);
            uint32* const tb_plen_ptr = reads.read_len ? tb_plen.data() : nullptr;
            hip_check(nvbio_hip_traceback_best_setup(count, nullptr, reinterpret_cast<const uint64*>(best_data_dvec.data()), band_len, genome_len, reads.read_begin, reads.read_len, reads.fixed(),
                                                     reads.rc_offset, 0u, 0, valid.data(), tb_pat.data(), tb_plen_ptr, tb_txt.data(), tb_len.data(), hip_stream),
                      "nvbio_hip_traceback_best_setup");
            const PackedStringSetView<4, true>  patterns(count, reads.fw_rc_words, reads.fw_rc_n_words, tb_pat.data(), tb_plen_ptr, reads.fixed());
            const PackedStringSetView<2, true>  texts(count, genome_words, genome_n_words, tb_txt.data(), tb_len.data(), 0u);
            const aln::AlignmentArrays alignments = { traceback_score.data(), cigar_source.data(), cigar_sink.data() };
            const aln::CigarArrays     cigars     = { cigar.data(), cigar_stride, cigar_len.data() };
            hip_check(nvbio_hip_traceback_best_known(count, nullptr, reinterpret_cast<const uint64*>(best_data_dvec.data()), best_sink.data(), traceback_score.data(),
                                                     cigar_sink.data(), hip_stream), "nvbio_hip_traceback_best_known");
            if (ed_mode)
            {
                // the edit-distance aligner's own walk (sw_banded_inl.h:405-470: among equal moves it does not choose as the Gotoh walk does)
                const nvbio_hip_sw_scheme w = { 0, -1, -1, -1 };
                const nvbio_hip_string_set p = patterns.abi(), t = texts.abi();
                hip::device_vector<uint8> temp(nvbio_hip_banded_gotoh_traceback_temp_bytes(band_len, L, count));
                hip_check(nvbio_hip_banded_sw_traceback(&w, int32(TYPE), band_len, &p, &t, L, L + band_len, count, traceback_score.data(), cigar_sink.data(), cigar_source.data(),
                                                        reinterpret_cast<uint16*>(cigar.data()), cigar_stride, cigar_len.data(), temp.data(), temp.size(), hip_stream),
                          "nvbio_hip_banded_sw_traceback");
                hip::synchronize(hip_stream);
            }
            else
            dispatch_band(band_len, [&](auto band) {
                typedef aln::PackedTracebackStream<aln::GotohAligner<TYPE, aln::SmithWatermanScoringScheme>, PackedStringSetView<4, true>, PackedStringSetView<2, true> > stream_type;
                const stream_type stream(aligner, patterns, texts, alignments, cigars, L, L + band_len, reads.quals, reads.n_quals);
                typedef aln::BatchedBandedAlignmentTraceback<decltype(band)::value, 32u, stream_type> batch_type;
                hip::device_vector<uint8> temp(batch_type::min_temp_storage(L, L + band_len, count));
                batch_type batch; batch.known_sinks = true;       // traceback_score / cigar_sink hold the extension's results (below)
                batch.enact(stream, temp.size(), temp.data(), hip_stream);
                hip::synchronize(hip_stream);                     // temp is released on scope exit
            });
            stats.clock.end("traceback", hip_stream);
            // finish_alignment_best: MD strings, edit distances, final scores; best_data then holds what the output stage reads
            if (params.finish_alignments)
            {
                stats.clock.begin("finish", hip_stream);
                const nvbio_hip_gotoh_qual_scheme sc = scoring_scheme.abi();
                const nvbio_hip_string_set p = patterns.abi(), t = texts.abi();
                hip_check(nvbio_hip_finish_alignment(count, valid.data(), &p, reads.quals, reads.n_quals, &t, reinterpret_cast<const uint16*>(cigar.data()), cigar_stride,
                                                     cigar_len.data(), cigar_source.data(), sc.match, sc.mismatch, 1 /* m_np = ConstantCost(1,1) */, &sc.pattern_gap_open, nullptr,
                                                     reinterpret_cast<uint64*>(best_data_dvec.data()), mds.data(), mds_stride, mds_len.data(), hip_stream),
                          "nvbio_hip_finish_alignment");
                hip::synchronize(hip_stream);
                stats.clock.end("finish", hip_stream);
            }
        }
        hip::synchronize(hip_stream);
    }

    template <aln::AlignmentType TYPE>
    void best_approx_paired_t(const Params& params, const PairedParams& pe, const fm_index_device& fmi, const fm_index_device& rfmi,
                              const aln::SmithWatermanScoringScheme& scoring_scheme, const ScoreLimits& limits,
                              const uint32* genome_words, const uint64 genome_n_words, const uint32 genome_len, const PairedReadBatch& reads, Stats& stats, void* hip_stream)
    {
        const uint32 count = reads.mate[0].n, L = std::max(reads.mate[0].len, reads.mate[1].len);      // L: the longest read of either mate
        const uint32 band_len = band_length(params.max_dist);
        const uint32 hits_stride = params.resolved_hits_stride(L);
        // the scheme hits are extended with (Params::scoring_mode): `sc`; the scheme of MAPQ and finish_alignment is the caller's: `fsc`
        const bool ed_mode = params.scoring_mode == EditDistanceMode;
        const aln::SmithWatermanScoringScheme search_scheme = params.search_scheme(scoring_scheme);
        const ScoreLimits search_limits = params.search_limits(limits);
        const aln::GotohAligner<TYPE, aln::SmithWatermanScoringScheme> aligner(search_scheme);
        const nvbio_hip_gotoh_qual_scheme sc = search_scheme.abi(), fsc = scoring_scheme.abi();
        const nvbio_hip_sw_scheme ed_costs = { 0, -1, -1, -1 };
        uint64* best   = reinterpret_cast<uint64*>(best_data_dvec.data());
        uint64* best_o = reinterpret_cast<uint64*>(best_data_dvec_o.data());
        hip::synchronize(hip_stream);
        const hip::arena_scope scope(workspace);                        // the per-batch vectors below live in the Aligner's workspace

        hip::device_vector<int32> min_score_table(search_limits.min_score_table(L)), mapq_score_table(limits.min_score_table(L));
        init_alignments(count, reads.mate[0].read_len, reads.mate[0].fixed(), min_score_table.data(), best_data_dvec.data(),   BATCH_SIZE, 0u, hip_stream);
        init_alignments(count, reads.mate[1].read_len, reads.mate[1].fixed(), min_score_table.data(), best_data_dvec_o.data(), BATCH_SIZE, 1u, hip_stream);

        std::vector<uint32> iota(count); std::iota(iota.begin(), iota.end(), 0u);
        hip::device_vector<uint32>  d_iota(iota), seed_queue_in(count), seed_queue_out(count), queue_count(1);
        hip::device_vector<SeedHit> hit_data(size_t(count) * hits_stride);
        hip::device_vector<uint32>  hit_counts(count);
        hip::device_vector<uint8>   reseed(count);
        hip::device_vector<uint32>  seed_freq(params.seed_freq_table(L));
        SelectState   state(count, hits_stride);
        const uint32  max_hits_per_round = std::max(SCORING_BATCH, count);
        ScoringQueues queues(count, max_hits_per_round);
        hip::device_vector<uint64> pat_begin(max_hits_per_round), txt_begin(max_hits_per_round);
        hip::device_vector<uint32> pat_len((reads.mate[0].read_len || reads.mate[1].read_len) ? max_hits_per_round : 0u);      // mates of their own lengths: a job's pattern length
        hip::device_vector<uint32> txt_len(max_hits_per_round), sinks(size_t(max_hits_per_round) * 2u), hit_sink(max_hits_per_round);
        hip::device_vector<int32>  min_score(max_hits_per_round), raw_score(max_hits_per_round), hit_score(max_hits_per_round);
        hip::device_vector<uint8>  o_valid(max_hits_per_round), o_rc(max_hits_per_round);
        hip::device_vector<uint32> o_gbegin(max_hits_per_round), o_gend(max_hits_per_round), o_loc(max_hits_per_round), o_sink(max_hits_per_round), o_sink2(max_hits_per_round);
        hip::device_vector<int32>  o_score(max_hits_per_round), o_score2(max_hits_per_round);
        hip::device_vector<uint8>  flag_temp(nvbio_hip_copy_flagged_temp_bytes(count));
        SeedHitDequeArrayDeviceView hits = { hit_data.data(), hits_stride, hit_counts.data() };
        const nvbio_hip_pe_params pp = { pe.pe_policy, int32(pe.min_frag_len), int32(pe.max_frag_len), pe.pe_overlap ? 1 : 0, worst_score, 0u, genome_len };
        hip::device_vector<uint32> memo(size_t(count) * 6u);                  // the opposite-mate memo, empty
        hip_check(nvbio_hip_memset(memo.data(), 0, uint64(count) * 24u, hip_stream), "nvbio_hip_memset");
        hip::device_vector<uint32> a_memo(size_t(count) * 6u);                // the anchor memo, empty (nvbio_hip.h: an anchor DP already run is not run again)
        hip_check(nvbio_hip_memset(a_memo.data(), 0, uint64(count) * 24u, hip_stream), "nvbio_hip_memset");
        hip::device_vector<uint8>  a_from_memo(max_hits_per_round);
        hip::device_vector<uint32> a_txt_len(max_hits_per_round), a_live_idx(max_hits_per_round), a_live_count(1);

        for (uint32 anchor = 0; anchor < 2; ++anchor)
        {
            const ReadBatch& a_reads = reads.mate[anchor];
            const ReadBatch& o_reads = reads.mate[1u - anchor];
            nvbio_hip_pe_params app = pp; app.anchor = anchor;
            seed_queue_in.assign(iota.data(), count, hip_stream);
            uint32 seed_queue_size = count;
            const bool fw_strand = (anchor == 0) ? (pe.pe_policy == 0 || pe.pe_policy == 1) : (pe.pe_policy == 0 || pe.pe_policy == 2);     // :168-176
            const bool fw = fw_strand ? params.fw : params.rc, rc = fw_strand ? params.rc : params.fw;

            for (uint32 seeding_pass = 0; seeding_pass < params.max_reseed + 1; ++seeding_pass)
            {
                if (seed_queue_size == 0) break;
                stats.queue.push_back(seed_queue_size); ++stats.seeding_passes;
                hip_check(nvbio_hip_memset(hit_counts.data(), 0, uint64(count) * 4u, hip_stream), "nvbio_hip_memset");
                const PingPongQueuesView seed_queues = { seed_queue_size, seed_queue_in.data() };
                stats.clock.run("map", hip_stream, [&] { fabric_bound(hip_stream, [&](void* ss) { map(a_reads.reversed, fmi, rfmi, seeding_pass, seed_queues, reseed.data(), hits, params, seed_freq.data(), fw, rc, ss); }); });

                // best_approx_score (:455-700)
                hip_check(nvbio_hip_pack_read_queue(seed_queue_size, seed_queue_in.data(), params.select.top_seed & 1u, reinterpret_cast<uint32*>(queues.active_in.data()), hip_stream),
                          "nvbio_hip_pack_read_queue");
                queues.in_size = seed_queue_size;
                // (a re-seeding pass: only its queue's reads have hits and are selected from -- select.h)
                stats.clock.run("select_init", hip_stream, [&] {
                    if (seed_queue_size < count) select_init(seed_queue_size, seed_queue_in.data(), a_reads.names, a_reads.names_idx, hits, state, params.select, hip_stream);
                    else                         select_init(count, a_reads.names, a_reads.names_idx, hits, state, params.select, hip_stream); });
                uint32 n_ext = 0;
                while (queues.in_size && n_ext < params.select.max_ext)
                {
                    uint32 n_hits_per_read = 1;
                    if (queues.in_size <= SCORING_BATCH / 2 && !params.no_multi_hits)
                        n_hits_per_read = std::min(SCORING_BATCH / queues.in_size, std::min(4096u, params.select.max_ext - n_ext));
                    stats.clock.run("select", hip_stream, [&] { select(hits, state, queues, n_hits_per_read, params.select, hip_stream); });
                    if (queues.in_size == 0) break;
                    if (queues.hits_size == 0) continue;
                    stats.clock.run("locate", hip_stream, [&] { fabric_bound(hip_stream, [&](void* ss) { locate(fmi, rfmi, queues, ss); }); });
                    const uint32 nh = queues.hits_size;
                    const uint32* hit_seed = reinterpret_cast<const uint32*>(queues.hit_seed.data());

                    // anchor_score_best
                    stats.clock.begin("anchor_score", hip_stream);
                    hip_check(nvbio_hip_anchor_score_setup(nh, queues.hit_read_id.data(), queues.hit_loc.data(), hit_seed, a_reads.read_begin, a_reads.read_len, o_reads.read_len, a_reads.fixed(), o_reads.fixed(), a_reads.rc_offset,
                                                           band_len, genome_len, best, best_o, BATCH_SIZE, sc.match, min_score_table.data(), worst_score, anchor,
                                                           pat_begin.data(), a_reads.read_len ? pat_len.data() : nullptr, txt_begin.data(), txt_len.data(), min_score.data(), hip_stream), "nvbio_hip_anchor_score_setup");
                    hip_check(nvbio_hip_anchor_memo_mark(nh, queues.hit_read_id.data(), hit_seed, txt_begin.data(), txt_len.data(), anchor, a_memo.data(), a_from_memo.data(),
                                                         a_txt_len.data(), a_live_count.data(), a_live_idx.data(), hip_stream), "nvbio_hip_anchor_memo_mark");
                    {
                        // the hits left with a window: one lane per job over all hits when there are many (the answered ones return at once), one wave per
                        // job over their list when there are few -- both queued, the device runs the one the count calls for
                        const PackedStringSetView<4, true> patterns(nh, a_reads.fw_rc_words, a_reads.fw_rc_n_words, pat_begin.data(), a_reads.read_len ? pat_len.data() : nullptr, a_reads.fixed());
                        const PackedStringSetView<2, true> texts(nh, genome_words, genome_n_words, txt_begin.data(), a_txt_len.data(), 0u);
                        const nvbio_hip_string_set ap = patterns.abi(), at = texts.abi();
                        const uint32 wave_up_to = L <= 512u ? wave_form_up_to() : 0u;
                        const uint32 bl = band_len < 4 ? 3u : band_len < 8 ? 7u : band_len < 16 ? 15u : 31u;
                        hip_check(nvbio_hip_banded_gotoh_score_qual_bounded(&sc, int32(TYPE), bl, &ap, a_reads.quals, a_reads.n_quals, nullptr, &at, L, L + band_len, nh, nullptr, nullptr,
                                                                            nullptr, nullptr, wave_up_to ? a_live_count.data() : nullptr, wave_up_to, raw_score.data(), sinks.data(), hip_stream),
                                  "nvbio_hip_banded_gotoh_score_qual_bounded");
                        if (wave_up_to)
                            hip_check(nvbio_hip_banded_gotoh_score_qual_wave(&sc, int32(TYPE), bl, &ap, a_reads.quals, a_reads.n_quals, &at, L, std::min(nh, wave_up_to), a_live_count.data(),
                                                                             a_live_idx.data(), nullptr, a_live_count.data(), wave_up_to, raw_score.data(), sinks.data(), hip_stream),
                                      "nvbio_hip_banded_gotoh_score_qual_wave");
                    }
                    hip_check(nvbio_hip_anchor_score_finish_memo(nh, raw_score.data(), sinks.data(), txt_begin.data(), min_score.data(), worst_score, a_from_memo.data(),
                                                                 queues.hit_read_id.data(), a_memo.data(), hit_score.data(), hit_sink.data(), hip_stream), "nvbio_hip_anchor_score_finish_memo");
                    hip_check(nvbio_hip_anchor_memo_update(queues.in_size, reinterpret_cast<const uint32*>(queues.active_in.data()), queues.hit_begin.data(), queues.hit_read_id.data(), hit_seed,
                                                           txt_begin.data(), txt_len.data(), a_from_memo.data(), raw_score.data(), sinks.data(), anchor, a_memo.data(), hip_stream),
                              "nvbio_hip_anchor_memo_update");
                    stats.clock.end("anchor_score", hip_stream);

                    // opposite_score_best over the hits whose anchor scored: every hit gets a job, the invalid ones an empty text
                    stats.clock.begin("opposite_score", hip_stream);
                    hip_check(nvbio_hip_opposite_score_setup(nh, queues.hit_read_id.data(), hit_seed, queues.hit_loc.data(), hit_score.data(), worst_score, a_reads.read_len, o_reads.read_len, a_reads.fixed(), o_reads.fixed(),
                                                             best, best_o, BATCH_SIZE, sc.match, min_score_table.data(), sc.text_gap_open, sc.text_gap_ext, &app,
                                                             o_valid.data(), min_score.data(), o_rc.data(), o_gbegin.data(), o_gend.data(),
                                                             o_reads.read_begin, o_reads.rc_offset, pat_begin.data(), txt_begin.data(), txt_len.data(), hip_stream), "nvbio_hip_opposite_score_setup");
                    if (o_reads.read_len)           // mates of their own lengths: each job's pattern length is its read's
                        hip_check(nvbio_hip_gather_rows(nh, queues.hit_read_id.data(), o_reads.read_len, pat_len.data(), 4u, hip_stream), "nvbio_hip_gather_rows");
                    // jobs equal to the pair's last scored job are answered from the memo (nvbio_hip.h: the reference re-runs them)
                    hip_check(nvbio_hip_opposite_memo_lookup(nh, queues.hit_read_id.data(), o_valid.data(), o_rc.data(), o_gbegin.data(), o_gend.data(), min_score.data(), anchor,
                                                             memo.data(), worst_score, o_score.data(), o_score2.data(), o_loc.data(), o_sink.data(), o_sink2.data(), txt_len.data(),
                                                             hip_stream), "nvbio_hip_opposite_memo_lookup");
                    {
                        // the windows left to score (valid == 1 after the memo): the throughput kernels over all hits when there are many, one job per
                        // wave over their list when there are few -- both queued, the device runs the one the count calls for (nvbio_hip.h)
                        const PackedStringSetView<4, true> patterns(nh, o_reads.fw_rc_words, o_reads.fw_rc_n_words, pat_begin.data(), o_reads.read_len ? pat_len.data() : nullptr, o_reads.fixed());
                        const PackedStringSetView<2, true> texts(nh, genome_words, genome_n_words, txt_begin.data(), txt_len.data(), 0u);
                        const nvbio_hip_string_set p = patterns.abi(), t = texts.abi();
                        const uint32 wave_up_to = wave_form_full_up_to();
                        if (wave_up_to)
                        {
                            hip_check(nvbio_hip_list_flagged(nh, o_valid.data(), 1u, a_live_count.data(), a_live_idx.data(), hip_stream), "nvbio_hip_list_flagged");
                            hip_check(nvbio_hip_alignment_score_qual_jobs(&sc, NVBIO_HIP_PATTERN_BLOCKING, int32(TYPE), &p, o_reads.quals, o_reads.n_quals, &t, L, pe.max_frag_len + L,
                                                                          min_score.data(), nh, nullptr, nullptr, a_live_count.data(), wave_up_to, 0, raw_score.data(), sinks.data(), nullptr,
                                                                          hip_stream), "nvbio_hip_alignment_score_qual_jobs");
                            hip_check(nvbio_hip_alignment_score_qual_jobs(&sc, NVBIO_HIP_PATTERN_BLOCKING, int32(TYPE), &p, o_reads.quals, o_reads.n_quals, &t, L, pe.max_frag_len + L,
                                                                          min_score.data(), std::min(nh, wave_up_to), a_live_count.data(), a_live_idx.data(), a_live_count.data(), wave_up_to, 1,
                                                                          raw_score.data(), sinks.data(), nullptr, hip_stream), "nvbio_hip_alignment_score_qual_jobs");
                        }
                        else
                            hip_check(nvbio_hip_alignment_score_qual(&sc, NVBIO_HIP_PATTERN_BLOCKING, int32(TYPE), &p, o_reads.quals, o_reads.n_quals, &t, L, pe.max_frag_len + L,
                                                                     min_score.data(), nh, raw_score.data(), sinks.data(), nullptr, hip_stream), "nvbio_hip_alignment_score_qual");
                    }
                    hip_check(nvbio_hip_opposite_score_finish(nh, nullptr, o_valid.data(), raw_score.data(), sinks.data(), min_score.data(), o_gbegin.data(), worst_score,
                                                              o_score.data(), o_score2.data(), o_loc.data(), o_sink.data(), o_sink2.data(), hip_stream), "nvbio_hip_opposite_score_finish");
                    hip_check(nvbio_hip_opposite_memo_update(queues.in_size, reinterpret_cast<const uint32*>(queues.active_in.data()), queues.hit_begin.data(), o_valid.data(), o_rc.data(),
                                                             o_gbegin.data(), o_gend.data(), min_score.data(), o_score.data(), o_sink.data(), anchor, memo.data(), hip_stream),
                              "nvbio_hip_opposite_memo_update");
                    stats.clock.end("opposite_score", hip_stream);

                    // score_reduce_paired with the give-up counters
                    stats.clock.begin("reduce", hip_stream);
                    hip_check(nvbio_hip_score_reduce_paired_best_approx(queues.in_size, reinterpret_cast<const uint32*>(queues.active_in.data()), queues.hit_begin.data(),
                                  queues.hit_loc.data(), hit_sink.data(), hit_score.data(), hit_seed, o_loc.data(), o_sink.data(), o_sink2.data(), o_score.data(), o_score2.data(),
                                  a_reads.read_len, a_reads.fixed(), anchor, pe.pe_policy, pe.pe_unpaired ? 1 : 0, worst_score, best, best_o, BATCH_SIZE,
                                  state.trys.data(), hit_counts.data(), n_ext, params.select.min_ext, params.select.max_ext, params.select.max_effort, hip_stream),
                              "nvbio_hip_score_reduce_paired_best_approx");
                    stats.clock.end("reduce", hip_stream);
                    stats.extensions += nh; ++stats.rounds;
                    n_ext += n_hits_per_read;
                }

                // copy the reads that need reseeding (no mark_unaligned in the paired driver)
                hip_check(nvbio_hip_copy_flagged(seed_queue_size, seed_queue_in.data(), reseed.data(), seed_queue_out.data(), queue_count.data(),
                                                 flag_temp.data(), flag_temp.size(), hip_stream), "nvbio_hip_copy_flagged");
                hip::synchronize(hip_stream);
                seed_queue_size = queue_count.to_host(hip_stream)[0];
                std::swap(seed_queue_in.m_ptr, seed_queue_out.m_ptr);
            }
        }

        if (pe.pe_discordant)
            hip_check(nvbio_hip_mark_discordant(count, best, best_o, BATCH_SIZE, hip_stream), "nvbio_hip_mark_discordant");
        // mate 1's MAPQ functor
        hip_check(nvbio_hip_mapq_paired(2, limits.match, limits.monotone ? 1 : 0, mapq_score_table.data(), count, best, best_o, BATCH_SIZE, reads.mate[0].read_len, reads.mate[1].read_len, reads.mate[0].fixed(), reads.mate[1].fixed(),
                                        mapq_dvec.data(), hip_stream), "nvbio_hip_mapq_paired");

        // tracebacks + finish: anchor slots (banded), opposite slots (full matrix for the concordant ones, banded for the others)
        stats.clock.begin("traceback", hip_stream);
        hip::device_vector<uint8>  valid(count), valid_c(count);
        hip::device_vector<uint64> tb_pat(count), tb_txt(count);
        hip::device_vector<uint32> tb_len(count), tb_plen(count), idx_c(count);
        // a slot's pattern comes from its mate's own half of the stream, at that mate's begins / lengths (nvbio_hip_traceback_best_setup_mates)
        const uint64* const m_begin[2] = { reads.mate[0].read_begin, reads.mate[1].read_begin };
        const uint32* const m_len[2]   = { reads.mate[0].read_len,   reads.mate[1].read_len };
        const uint32        m_fixed[2] = { reads.mate[0].fixed(),    reads.mate[1].fixed() };
        const uint64        m_rc[2]    = { reads.mate[0].rc_offset,  reads.mate[1].rc_offset };
        auto banded_tb = [&](const uint64* slots, const int32 want, hip::device_vector<uint8>& v, io::Cigar* cg, uint32* cg_len, uint32* src, uint32* snk, int32* score) {
            hip_check(nvbio_hip_memset(cg, 0, uint64(count) * cigar_stride * sizeof(io::Cigar), hip_stream), "nvbio_hip_memset");
            hip_check(nvbio_hip_traceback_best_setup_mates(count, nullptr, slots, band_len, genome_len, m_begin, m_len, m_fixed, m_rc, reads.mate_offset, want,
                                                           v.data(), tb_pat.data(), tb_plen.data(), tb_txt.data(), tb_len.data(), hip_stream), "nvbio_hip_traceback_best_setup_mates");
            const PackedStringSetView<4, true> patterns(count, reads.both_words, reads.both_n_words, tb_pat.data(), tb_plen.data(), 0u);
            const PackedStringSetView<2, true> texts(count, genome_words, genome_n_words, tb_txt.data(), tb_len.data(), 0u);
            const nvbio_hip_string_set p = patterns.abi(), t = texts.abi();
            hip::device_vector<uint8> temp(nvbio_hip_banded_gotoh_traceback_temp_bytes(band_len < 4 ? 3u : band_len < 8 ? 7u : band_len < 16 ? 15u : 31u, L, count));
            if (ed_mode)      // the edit-distance aligner's own walk (sw_banded_inl.h:405-470)
                hip_check(nvbio_hip_banded_sw_traceback(&ed_costs, int32(TYPE), band_len < 4 ? 3u : band_len < 8 ? 7u : band_len < 16 ? 15u : 31u, &p, &t, L, L + band_len, count, score, snk, src,
                                                        reinterpret_cast<uint16*>(cg), cigar_stride, cg_len, temp.data(), temp.size(), hip_stream), "nvbio_hip_banded_sw_traceback");
            else
            hip_check(nvbio_hip_banded_gotoh_traceback_qual(&sc, int32(TYPE), band_len < 4 ? 3u : band_len < 8 ? 7u : band_len < 16 ? 15u : 31u, &p, reads.both_quals, reads.both_n_quals, &t,
                                                            L, L + band_len, count, score, snk, src, reinterpret_cast<uint16*>(cg), cigar_stride, cg_len, temp.data(), temp.size(), hip_stream),
                      "nvbio_hip_banded_gotoh_traceback_qual");
            hip::synchronize(hip_stream);
        };
        auto finish = [&](const uint32 n_jobs, const uint8* v, const uint32* idx, uint64* slots, const io::Cigar* cg, const uint32* cg_len, const uint32* src, uint8* md, uint32* md_len) {
            const PackedStringSetView<4, true> patterns(n_jobs, reads.both_words, reads.both_n_words, tb_pat.data(), tb_plen.data(), 0u);
            const PackedStringSetView<2, true> texts(n_jobs, genome_words, genome_n_words, tb_txt.data(), tb_len.data(), 0u);
            const nvbio_hip_string_set p = patterns.abi(), t = texts.abi();
            hip_check(nvbio_hip_finish_alignment(n_jobs, v, &p, reads.both_quals, reads.both_n_quals, &t, reinterpret_cast<const uint16*>(cg), cigar_stride, cg_len, src,
                                                 fsc.match, fsc.mismatch, 1, &fsc.pattern_gap_open, idx, slots, md, mds_stride, md_len, hip_stream), "nvbio_hip_finish_alignment");
        };

        banded_tb(best, 0, valid, cigar.data(), cigar_len.data(), cigar_source.data(), cigar_sink.data(), traceback_score.data());
        if (params.finish_alignments) finish(count, valid.data(), nullptr, best, cigar.data(), cigar_len.data(), cigar_source.data(), mds.data(), mds_len.data());
        // mate 2's MAPQ functor: after the anchor slots were finished, before the opposite ones are (:308-323)
        hip_check(nvbio_hip_mapq_paired(2, limits.match, limits.monotone ? 1 : 0, mapq_score_table.data(), count, best_o, best, BATCH_SIZE, reads.mate[1].read_len, reads.mate[0].read_len, reads.mate[1].fixed(), reads.mate[0].fixed(),
                                        mapq_dvec_o.data(), hip_stream), "nvbio_hip_mapq_paired");

        // which opposite slots are concordant (their tracebacks run over the full matrix of [alignment, alignment + sink))
        hip_check(nvbio_hip_traceback_best_setup_mates(count, nullptr, best_o, band_len, genome_len, m_begin, m_len, m_fixed, m_rc, reads.mate_offset, 1,
                                                       valid_c.data(), tb_pat.data(), tb_plen.data(), tb_txt.data(), tb_len.data(), hip_stream), "nvbio_hip_traceback_best_setup_mates");
        hip_check(nvbio_hip_copy_flagged(count, d_iota.data(), valid_c.data(), idx_c.data(), queue_count.data(), flag_temp.data(), flag_temp.size(), hip_stream), "nvbio_hip_copy_flagged");
        hip::synchronize(hip_stream);
        const uint32 n_conc = queue_count.to_host(hip_stream)[0];

        banded_tb(best_o, 2, valid, cigar_o.data(), cigar_len_o.data(), cigar_source_o.data(), cigar_sink_o.data(), traceback_score_o.data());
        if (params.finish_alignments) finish(count, valid.data(), nullptr, best_o, cigar_o.data(), cigar_len_o.data(), cigar_source_o.data(), mds_o.data(), mds_len_o.data());
        if (n_conc)
        {
            hip::device_vector<uint8>     v(n_conc), md(size_t(n_conc) * mds_stride);
            hip::device_vector<io::Cigar> cg(size_t(n_conc) * cigar_stride);
            hip::device_vector<uint32>    cg_len(n_conc), src(size_t(n_conc) * 2u), snk(size_t(n_conc) * 2u), md_len(n_conc);
            hip::device_vector<int32>     score(n_conc);
            hip_check(nvbio_hip_memset(cg.data(), 0, uint64(n_conc) * cigar_stride * sizeof(io::Cigar), hip_stream), "nvbio_hip_memset");
            hip_check(nvbio_hip_traceback_best_setup_mates(n_conc, idx_c.data(), best_o, band_len, genome_len, m_begin, m_len, m_fixed, m_rc, reads.mate_offset, 1,
                                                           v.data(), tb_pat.data(), tb_plen.data(), tb_txt.data(), tb_len.data(), hip_stream), "nvbio_hip_traceback_best_setup_mates");
            const PackedStringSetView<4, true> patterns(n_conc, reads.both_words, reads.both_n_words, tb_pat.data(), tb_plen.data(), 0u);
            const PackedStringSetView<2, true> texts(n_conc, genome_words, genome_n_words, tb_txt.data(), tb_len.data(), 0u);
            const nvbio_hip_string_set p = patterns.abi(), t = texts.abi();
            hip::device_vector<uint8> temp(nvbio_hip_gotoh_traceback_temp_bytes(L, 1024u, n_conc));
            // each window ends at the sink of the scoring pass whose score the slot holds: the rows no alignment of that score can reach are dropped
            hip::device_vector<int32> known(n_conc);
            hip_check(nvbio_hip_traceback_best_known(n_conc, idx_c.data(), best_o, nullptr, known.data(), nullptr, hip_stream), "nvbio_hip_traceback_best_known");
            if (ed_mode)      // (sw_inl.h:1660-1700)
                hip_check(nvbio_hip_sw_traceback(&ed_costs, int32(TYPE), &p, &t, L, 1024u, n_conc, score.data(), snk.data(), src.data(), reinterpret_cast<uint16*>(cg.data()), cigar_stride,
                                                 cg_len.data(), temp.data(), temp.size(), hip_stream), "nvbio_hip_sw_traceback");
            else
            hip_check(nvbio_hip_gotoh_traceback_qual_known_score(&sc, int32(TYPE), &p, reads.both_quals, reads.both_n_quals, &t, known.data(), L, 1024u, n_conc, score.data(), snk.data(),
                                                                 src.data(), reinterpret_cast<uint16*>(cg.data()), cigar_stride, cg_len.data(), temp.data(), temp.size(), hip_stream),
                      "nvbio_hip_gotoh_traceback_qual_known_score");
            if (params.finish_alignments) finish(n_conc, v.data(), idx_c.data(), best_o, cg.data(), cg_len.data(), src.data(), md.data(), md_len.data());
            // put the concordant mates' results at their reads
            hip_check(nvbio_hip_scatter_rows(n_conc, idx_c.data(), cg.data(),     cigar_o.data(),           cigar_stride * 2u, hip_stream), "nvbio_hip_scatter_rows");
            hip_check(nvbio_hip_scatter_rows(n_conc, idx_c.data(), cg_len.data(), cigar_len_o.data(),       4u, hip_stream), "nvbio_hip_scatter_rows");
            hip_check(nvbio_hip_scatter_rows(n_conc, idx_c.data(), src.data(),    cigar_source_o.data(),    8u, hip_stream), "nvbio_hip_scatter_rows");
            hip_check(nvbio_hip_scatter_rows(n_conc, idx_c.data(), snk.data(),    cigar_sink_o.data(),      8u, hip_stream), "nvbio_hip_scatter_rows");
            hip_check(nvbio_hip_scatter_rows(n_conc, idx_c.data(), score.data(),  traceback_score_o.data(), 4u, hip_stream), "nvbio_hip_scatter_rows");
            if (params.finish_alignments) {
                hip_check(nvbio_hip_scatter_rows(n_conc, idx_c.data(), md.data(),     mds_o.data(),     mds_stride, hip_stream), "nvbio_hip_scatter_rows");
                hip_check(nvbio_hip_scatter_rows(n_conc, idx_c.data(), md_len.data(), mds_len_o.data(), 4u, hip_stream), "nvbio_hip_scatter_rows");
            }
            hip::synchronize(hip_stream);
        }
        hip::synchronize(hip_stream);
        stats.clock.end("traceback", hip_stream);
    }

    /// whether the extension rounds hand the scorer their thresholds (NVBIO_HIP_BOUNDED_DP=1, read once; off by default)
    static bool bounded_dp() { static const bool on = [] { const char* e = getenv("NVBIO_HIP_BOUNDED_DP"); return e && atoi(e) == 1; }(); return on; }
    /// the largest batch of DP jobs that runs one wave per job (banded_gotoh_wave.hip) instead of one lane per job: NVBIO_HIP_WAVE_JOBS (read once;
    /// default 24576, 0 = never).  A round's jobs are counted on the device and the device picks the form (alignment.h).
    static uint32 wave_form_up_to() { static const uint32 v = [] { const char* e = getenv("NVBIO_HIP_WAVE_JOBS"); return e ? uint32(atoi(e)) : 24576u; }(); return v; }
    /// ... and the same for the opposite mate's full-matrix DP: NVBIO_HIP_WAVE_JOBS_FULL (default 6144)
    static uint32 wave_form_full_up_to() { static const uint32 v = [] { const char* e = getenv("NVBIO_HIP_WAVE_JOBS_FULL"); return e ? uint32(atoi(e)) : 6144u; }(); return v; }
    /// NVBIO_HIP_TRACE_ROUNDS=1 (read once): one line per extension round on stderr -- the queue sizes the hits-per-read rule saw
    static bool trace_rounds() { static const bool on = [] { const char* e = getenv("NVBIO_HIP_TRACE_ROUNDS"); return e && atoi(e) == 1; }(); return on; }
    bool count_jobs = false;          ///< fill Stats::dp_jobs (one host round trip per extension round; the stage clock does it too)

    /// the static band of banded_score_best / banded_traceback_best (score_best_inl.h:160-164)
    template <typename F>
    static void dispatch_band(const uint32 band_len, F f)
    {
        if      (band_len < 4)  f(std::integral_constant<uint32, 3u>());
        else if (band_len < 8)  f(std::integral_constant<uint32, 7u>());
        else if (band_len < 16) f(std::integral_constant<uint32, 15u>());
        else                    f(std::integral_constant<uint32, 31u>());
    }

    /// Aligner::best_approx_score (aligner_best_approx.h:522-840): the extension rounds of one seeding pass
    template <aln::AlignmentType TYPE>
    void best_approx_score(const Params& params, const fm_index_device& fmi, const fm_index_device& rfmi,
                           const aln::GotohAligner<TYPE, aln::SmithWatermanScoringScheme>& aligner,
                           const uint32* genome_words, const uint64 genome_n_words, const uint32 genome_len, const ReadBatch& reads, const uint32 band_len,
                           const uint32 seed_queue_size, const uint32* seed_queue,
                           SeedHitDequeArrayDeviceView hits, SelectState& state, ScoringQueues& queues,
                           hip::device_vector<uint64>& pat_begin, hip::device_vector<uint64>& txt_begin, hip::device_vector<uint32>& txt_len,
                           hip::device_vector<uint32>& sinks, hip::device_vector<int32>& min_score, hip::device_vector<int32>& hit_score,
                           Stats& stats, void* hip_stream, uint32* best_sink = nullptr)
    {
        const uint32 L = reads.len;
        hip::device_vector<int32> known_score(pat_begin.size());
        hip::device_vector<uint32> hit_sink(best_sink ? pat_begin.size() * 2u : 0u);       // the DP sinks, per hit (kept for the traceback)
        hip::device_vector<uint32> job_hit(pat_begin.size()), job_count(1), work_counter(1);
        hip::device_vector<uint32> pat_len(reads.read_len ? pat_begin.size() : 0u);          // reads of their own lengths: the jobs' pattern lengths
        // active_read_queues.in_queue = pack_read( params.top_seed ) of the seed queue
        hip_check(nvbio_hip_pack_read_queue(seed_queue_size, seed_queue, params.select.top_seed & 1u, reinterpret_cast<uint32*>(queues.active_in.data()), hip_stream),
                  "nvbio_hip_pack_read_queue");
        queues.in_size = seed_queue_size;
        // (a re-seeding pass: only its queue's reads have hits and are selected from -- select.h)
        stats.clock.run("select_init", hip_stream, [&] {
            if (seed_queue_size < reads.n) select_init(seed_queue_size, seed_queue, reads.names, reads.names_idx, hits, state, params.select, hip_stream);
            else                           select_init(reads.n, reads.names, reads.names_idx, hits, state, params.select, hip_stream); });

        uint32 n_ext = 0;
        while (queues.in_size && n_ext < params.select.max_ext)
        {
            // how many hits per read this round (:627-650)
            uint32 n_hits_per_read = 1;
            if (queues.in_size <= SCORING_BATCH / 2 && !params.no_multi_hits)
                n_hits_per_read = std::min(SCORING_BATCH / queues.in_size, std::min(4096u, params.select.max_ext - n_ext));

            const uint32 in_before = queues.in_size;
            stats.clock.run("select", hip_stream, [&] { select(hits, state, queues, n_hits_per_read, params.select, hip_stream); });
            if (trace_rounds()) fprintf(stderr, "round %u: n_ext %u, active in %u -> out %u, hits per read %u, hits %u\n", stats.rounds, n_ext, in_before, queues.in_size, n_hits_per_read, queues.hits_size);
            if (queues.in_size == 0) break;
            if (queues.hits_size == 0) continue;
            stats.clock.run("locate", hip_stream, [&] { fabric_bound(hip_stream, [&](void* ss) { locate(fmi, rfmi, queues, ss); }); });
            stats.clock.begin("score", hip_stream);

            // score_best: BestScoreStream's windows, then the banded scorer in nvBowtie's quality-aware scheme
            // Hits at a placement the read already recorded keep the recorded score (known_score, see nvbio_hip.h); only the others
            // become DP jobs, compacted, their scores scattered back at their hits.
            uint32* const pat_len_ptr = reads.read_len ? pat_len.data() : nullptr;
            score_best_setup(queues, reads.read_begin, reads.read_len, reads.fixed(), reads.rc_offset, band_len, genome_len, best_data_dvec.data(), BATCH_SIZE, worst_score,
                             pat_begin.data(), pat_len_ptr, txt_begin.data(), txt_len.data(), min_score.data(), known_score.data(),
                             job_count.data(), job_hit.data(), hip_stream);
            // The DP jobs.  job_count stays on the device -- the scorer reads it there, so the host does not wait for the set-up kernel between
            // the two -- and every job's score and sink are written straight back at its hit (job_hit): scores into known_score, sinks into
            // hit_sink.  NVBIO_HIP_BOUNDED_DP=1 also hands the scorer every job's min_score (the read's second-best score, as
            // BestScoreStream::init_context sets it): jobs that cannot beat it are given up part way by persistent waves
            // (nvbio_amd/csrc/banded_gotoh_bounded.h) -- same records, measured slower than the plain kernel on this hardware.
            uint32 n_jobs = 0;
            {
                const uint32 nh = queues.hits_size;
                const PackedStringSetView<4, true> patterns(nh, reads.fw_rc_words, reads.fw_rc_n_words, pat_begin.data(), pat_len_ptr, reads.fixed());
                const PackedStringSetView<2, true> texts(nh, genome_words, genome_n_words, txt_begin.data(), txt_len.data(), 0u);
                const aln::BestSinkArrays sink_arrays = { known_score.data(), best_sink ? hit_sink.data() : sinks.data() };
                dispatch_band(band_len, [&](auto band) {
                    aln::batch_banded_alignment_score<decltype(band)::value>(aligner, patterns, reads.quals, reads.n_quals, texts, bounded_dp() ? min_score.data() : nullptr,
                                                                             job_count.data(), work_counter.data(), job_hit.data(), sink_arrays, L, L + band_len, hip_stream,
                                                                             wave_form_up_to());
                });
                if (stats.clock.enabled || count_jobs) { hip::synchronize(hip_stream); hip_check(nvbio_hip_memcpy(&n_jobs, job_count.data(), 4u, 2, hip_stream), "nvbio_hip_memcpy(d2h)"); }
            }
            stats.dp_jobs += n_jobs;
            stats.clock.end("score", hip_stream);

            // score_reduce with the give-up counters
            stats.clock.run("reduce", hip_stream, [&] {
                score_reduce(ReduceBestApproxContext(state.trys.data(), n_ext), hits, queues, known_score.data(), reads.read_len, reads.fixed(), best_data_dvec.data(), BATCH_SIZE,
                             worst_score, params.select, nullptr, hip_stream, best_sink ? hit_sink.data() : nullptr, best_sink);
            });
            stats.extensions += queues.hits_size; ++stats.rounds;
            n_ext += n_hits_per_read;
        }
    }
};

} // namespace cuda
} // namespace bowtie2
} // namespace nvbio
