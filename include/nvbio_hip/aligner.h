// nvbio_hip/aligner.h -- nvBowtie's single-end best-mapping driver over libnvbio_hip.so, in the reference's language.
// Mirrors nvBowtie/bowtie2/cuda/aligner.h (struct Aligner: BATCH_SIZE, best_data_dvec, mapq_dvec, cigar storage,
// band_length) and aligner_best_approx.h (Aligner::best_approx :85-520, Aligner::best_approx_score :522-840): the same
// control flow over the C-ABI stages -- init_alignments, per seeding pass map -> select_init -> extension rounds of
// {select, locate, score_best, score_reduce}, mark_unaligned + copy_flagged into the re-seed queue, BowtieMapq2,
// banded_traceback_best.  Inputs are device resident (the reference's io::SequenceDataDevice / FMIndexDataDevice).
#pragma once
#include <algorithm>
#include <numeric>
#include <vector>
#include "alignment.h"
#include "mapping.h"
#include "reduce.h"
#include "select.h"

namespace nvbio {
namespace bowtie2 {
namespace cuda {

enum AlignmentTypeMode { EndToEndAlignment = 0, LocalAlignment = 1 };           // params.h

/// the fields of nvBowtie's Params the driver reads, with its defaults (params.cpp:116-197)
struct Params : public ParamsPOD
{
    Params() : max_dist(15), alignment_type(EndToEndAlignment), no_multi_hits(false), fw(true), rc(true), hits_stride(0), finish_alignments(true) {}
    SelectParamsPOD select;
    uint32 max_dist; AlignmentTypeMode alignment_type; bool no_multi_hits, fw, rc; uint32 hits_stride;
    bool   finish_alignments;      ///< run finish_alignment_best (MD strings, edit distances, final scores) as the reference always does
};

struct Stats { uint64 extensions; uint32 rounds, seeding_passes; std::vector<uint32> queue; Stats() : extensions(0), rounds(0), seeding_passes(0) {} };

/// A batch of equal-length reads on the device in the layouts the stages read (io::SequenceDataDevice's role): the reads
/// stored reversed (io::REVERSE, what the mappers scan), their forward copies followed rc_offset symbols later by their
/// reverse complements (the extension / traceback patterns), one quality byte per pattern symbol, the read names.
struct ReadBatch
{
    uint32 n, len;
    PackedStringSetView<4, true> reversed;
    const uint32* fw_rc_words; uint64 fw_rc_n_words; uint64 rc_offset;
    const uint8*  quals;       uint64 n_quals;
    const char*   names;       const uint32* names_idx;
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

    Aligner() : BATCH_SIZE(0), SCORING_BATCH(0), cigar_stride(64), mds_stride(256) {}

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

    /// Aligner::best_approx (aligner_best_approx.h:85-520)
    void best_approx(const Params& params, const fm_index_device& fmi, const fm_index_device& rfmi, const aln::SmithWatermanScoringScheme& scoring_scheme,
                     const ScoreLimits& limits, const uint32* genome_words, const uint64 genome_n_words, const uint32 genome_len,
                     const ReadBatch& reads, Stats& stats, void* hip_stream = nullptr)
    {
        if (params.alignment_type == LocalAlignment) best_approx_t<aln::LOCAL>(params, fmi, rfmi, scoring_scheme, limits, genome_words, genome_n_words, genome_len, reads, stats, hip_stream);
        else                                         best_approx_t<aln::SEMI_GLOBAL>(params, fmi, rfmi, scoring_scheme, limits, genome_words, genome_n_words, genome_len, reads, stats, hip_stream);
    }

private:
    template <aln::AlignmentType TYPE>
    void best_approx_t(const Params& params, const fm_index_device& fmi, const fm_index_device& rfmi, const aln::SmithWatermanScoringScheme& scoring_scheme,
                       const ScoreLimits& limits, const uint32* genome_words, const uint64 genome_n_words, const uint32 genome_len,
                       const ReadBatch& reads, Stats& stats, void* hip_stream)
    {
        const uint32 count = reads.n, L = reads.len;
        const uint32 band_len = band_length(params.max_dist);
        const uint32 hits_stride = params.hits_stride ? params.hits_stride : std::min(params.max_hits, 128u);
        const aln::GotohAligner<TYPE, aln::SmithWatermanScoringScheme> aligner(scoring_scheme);

        // initialize best-alignments with the threshold score
        hip::device_vector<int32> min_score_table(limits.min_score_table(L));
        init_alignments(count, nullptr, L, min_score_table.data(), best_data_dvec.data(), BATCH_SIZE, 0u, hip_stream);

        // the seed queue, hit deques, selection state and scoring queues of the pipeline
        std::vector<uint32> iota(count); std::iota(iota.begin(), iota.end(), 0u);
        hip::device_vector<uint32> seed_queue_in(iota), seed_queue_out(count), queue_count(1);
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

        for (uint32 seeding_pass = 0; seeding_pass < params.max_reseed + 1; ++seeding_pass)
        {
            if (seed_queue_size == 0) break;
            stats.queue.push_back(seed_queue_size); ++stats.seeding_passes;

            // hit_deques.clear_deques() + map
            hip_check(nvbio_hip_memset(hit_counts.data(), 0, uint64(count) * 4u, hip_stream), "nvbio_hip_memset");
            const PingPongQueuesView seed_queues = { seed_queue_size, seed_queue_in.data() };
            map(reads.reversed, fmi, rfmi, seeding_pass, seed_queues, reseed.data(), hits, params, seed_freq.data(), params.fw, params.rc, hip_stream);

            best_approx_score<TYPE>(params, fmi, rfmi, aligner, genome_words, genome_n_words, genome_len, reads, band_len, seed_queue_size, seed_queue_in.data(),
                                    hits, state, queues, pat_begin, txt_begin, txt_len, sinks, min_score, hit_score, stats, hip_stream);

            // mark unaligned reads, copy the reads that need reseeding, swap the queues
            hip_check(nvbio_hip_mark_unaligned(seed_queue_size, seed_queue_in.data(), reinterpret_cast<const uint64*>(best_data_dvec.data()), reseed.data(), hip_stream), "nvbio_hip_mark_unaligned");
            hip_check(nvbio_hip_copy_flagged(seed_queue_size, seed_queue_in.data(), reseed.data(), seed_queue_out.data(), queue_count.data(),
                                             flag_temp.data(), flag_temp.size(), hip_stream), "nvbio_hip_copy_flagged");
            hip::synchronize(hip_stream);
            seed_queue_size = queue_count.to_host()[0];
            std::swap(seed_queue_in.m_ptr, seed_queue_out.m_ptr);
        }

        // compute mapq (BowtieMapq2)
        mapq(2, limits, min_score_table.data(), count, best_data_dvec.data(), BATCH_SIZE, nullptr, L, mapq_dvec.data(), hip_stream);

        // banded_traceback_best over every read (unaligned ones get an empty window and no CIGAR)
        {
            hip::device_vector<uint8>  valid(count);
            hip::device_vector<uint64> tb_pat(count), tb_txt(count);
            hip::device_vector<uint32> tb_len(count);
            hip_check(nvbio_hip_memset(cigar.data(), 0, uint64(count) * cigar_stride * sizeof(io::Cigar), hip_stream), "nvbio_hip_memset");
            hip_check(nvbio_hip_traceback_best_setup(count, nullptr, reinterpret_cast<const uint64*>(best_data_dvec.data()), band_len, genome_len, nullptr, nullptr, L,
                                                     reads.rc_offset, 0u, 0, valid.data(), tb_pat.data(), nullptr, tb_txt.data(), tb_len.data(), hip_stream),
                      "nvbio_hip_traceback_best_setup");
            const PackedStringSetView<4, true>  patterns(count, reads.fw_rc_words, reads.fw_rc_n_words, tb_pat.data(), nullptr, L);
            const PackedStringSetView<2, true>  texts(count, genome_words, genome_n_words, tb_txt.data(), tb_len.data(), 0u);
            const aln::AlignmentArrays alignments = { traceback_score.data(), cigar_source.data(), cigar_sink.data() };
            const aln::CigarArrays     cigars     = { cigar.data(), cigar_stride, cigar_len.data() };
            dispatch_band(band_len, [&](auto band) {
                typedef aln::PackedTracebackStream<aln::GotohAligner<TYPE, aln::SmithWatermanScoringScheme>, PackedStringSetView<4, true>, PackedStringSetView<2, true> > stream_type;
                const stream_type stream(aligner, patterns, texts, alignments, cigars, L, L + band_len, reads.quals, reads.n_quals);
                typedef aln::BatchedBandedAlignmentTraceback<decltype(band)::value, 32u, stream_type> batch_type;
                hip::device_vector<uint8> temp(batch_type::min_temp_storage(L, L + band_len, count));
                batch_type().enact(stream, temp.size(), temp.data(), hip_stream);
                hip::synchronize(hip_stream);                     // temp is released on scope exit
            });
            // finish_alignment_best: MD strings, edit distances, final scores; best_data then holds what the output stage reads
            if (params.finish_alignments)
            {
                const nvbio_hip_gotoh_qual_scheme sc = scoring_scheme.abi();
                const nvbio_hip_string_set p = patterns.abi(), t = texts.abi();
                hip_check(nvbio_hip_finish_alignment(count, valid.data(), &p, reads.quals, reads.n_quals, &t, reinterpret_cast<const uint16*>(cigar.data()), cigar_stride,
                                                     cigar_len.data(), cigar_source.data(), sc.match, sc.mismatch, 1 /* m_np = ConstantCost(1,1) */, nullptr,
                                                     reinterpret_cast<uint64*>(best_data_dvec.data()), mds.data(), mds_stride, mds_len.data(), hip_stream),
                          "nvbio_hip_finish_alignment");
                hip::synchronize(hip_stream);
            }
        }
        hip::synchronize(hip_stream);
    }

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
                           Stats& stats, void* hip_stream)
    {
        const uint32 L = reads.len;
        // active_read_queues.in_queue = pack_read( params.top_seed ) of the seed queue
        {
            hip::synchronize(hip_stream);
            std::vector<uint32> q(seed_queue_size);
            hip_check(nvbio_hip_memcpy(q.data(), seed_queue, uint64(seed_queue_size) * 4u, 2, nullptr), "nvbio_hip_memcpy(d2h)");
            std::vector<packed_read> packed(seed_queue_size);
            for (uint32 i = 0; i < seed_queue_size; ++i) packed[i] = packed_read(q[i], params.select.top_seed & 1u);
            hip_check(nvbio_hip_memcpy(queues.active_in.data(), packed.data(), uint64(seed_queue_size) * 4u, 1, nullptr), "nvbio_hip_memcpy(h2d)");
            queues.in_size = seed_queue_size;
        }
        select_init(reads.n, reads.names, reads.names_idx, hits, state, params.select, hip_stream);

        uint32 n_ext = 0;
        while (queues.in_size && n_ext < params.select.max_ext)
        {
            // how many hits per read this round (:627-650)
            uint32 n_hits_per_read = 1;
            if (queues.in_size <= SCORING_BATCH / 2 && !params.no_multi_hits)
                n_hits_per_read = std::min(SCORING_BATCH / queues.in_size, std::min(4096u, params.select.max_ext - n_ext));

            select(hits, state, queues, n_hits_per_read, params.select, hip_stream);
            if (queues.in_size == 0) break;
            if (queues.hits_size == 0) continue;
            locate(fmi, rfmi, queues, hip_stream);

            // score_best: BestScoreStream's windows, then the banded scorer in nvBowtie's quality-aware scheme
            score_best_setup(queues, nullptr, nullptr, L, reads.rc_offset, band_len, genome_len, best_data_dvec.data(), BATCH_SIZE, worst_score,
                             pat_begin.data(), nullptr, txt_begin.data(), txt_len.data(), min_score.data(), hip_stream);
            const PackedStringSetView<4, true> patterns(queues.hits_size, reads.fw_rc_words, reads.fw_rc_n_words, pat_begin.data(), nullptr, L);
            const PackedStringSetView<2, true> texts(queues.hits_size, genome_words, genome_n_words, txt_begin.data(), txt_len.data(), 0u);
            const aln::BestSinkArrays sink_arrays = { hit_score.data(), sinks.data() };
            dispatch_band(band_len, [&](auto band) {
                aln::batch_banded_alignment_score<decltype(band)::value>(aligner, patterns, reads.quals, reads.n_quals, texts, sink_arrays, L, L + band_len, hip_stream);
            });

            // score_reduce with the give-up counters
            score_reduce(ReduceBestApproxContext(state.trys.data(), n_ext), hits, queues, hit_score.data(), nullptr, L, best_data_dvec.data(), BATCH_SIZE,
                         worst_score, params.select, hip_stream);
            stats.extensions += queues.hits_size; ++stats.rounds;
            n_ext += n_hits_per_read;
        }
    }
};

} // namespace cuda
} // namespace bowtie2
} // namespace nvbio
