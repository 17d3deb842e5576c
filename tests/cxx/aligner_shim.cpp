// aligner_shim.cpp -- a C entry point around the C++ host driver nvbio::bowtie2::cuda::Aligner (include/nvbio_hip/aligner.h)
// so that the Python GPU tests can run it on the same device-resident inputs as the Python driver and compare both with the
// oracle driver.  Test infrastructure: built by __graft_entry__.build() into tests/cxx/libaligner_shim.so.
#include <chrono>
#include <cstring>
#include <nvbio_hip/aligner.h>

using namespace nvbio;
using namespace nvbio::bowtie2::cuda;


struct shim_params
{
    uint32_t local, randomized, top_seed, max_effort_init, max_effort, min_ext, max_ext, max_reseed, rep_seeds, max_hits, allow_sub, subseed_len,
             seed_len, seed_freq_type, min_read_len, max_dist, no_multi_hits, batch_size, hits_stride;
    float    seed_freq_k, seed_freq_m;
    int32_t  match, score_min_type; float score_min_k, score_min_m;
    uint32_t finish;
    uint32_t edit_distance;        // 1: --scoring ed (params.h:47-51), 0: Smith-Waterman
};

// Reads of their own lengths: the next nvbio_aligner_best_approx / _paired / _paired_quals call of this thread takes mate m's reads at d_read_begin[r]
// (symbols into its forward copies, the reversed stream alike), d_read_len[r] long, reverse complements rc_offset further; L is then the
// longest read.  d_read_len == NULL clears it.
struct ragged_layout { const uint64_t* begin; const uint32_t* len; uint64_t rc_offset; };
static thread_local ragged_layout g_ragged[2] = { { nullptr, nullptr, 0 }, { nullptr, nullptr, 0 } };
extern "C" __attribute__((visibility("default")))
void nvbio_aligner_set_ragged(int mate, const uint64_t* d_read_begin, const uint32_t* d_read_len, uint64_t rc_offset)
{ if (mate >= 0 && mate < 2) { g_ragged[mate].begin = d_read_begin; g_ragged[mate].len = d_read_len; g_ragged[mate].rc_offset = rc_offset; } }
static void apply_ragged(ReadBatch& b, const int mate, const uint32_t* rev_words, const uint64_t rev_n_words, const uint64_t* rev_begin)
{
    const ragged_layout& r = g_ragged[mate];
    if (!r.len) return;
    b.read_begin = r.begin; b.read_len = r.len; b.rc_offset = r.rc_offset;
    b.reversed = PackedStringSetView<4, true>(b.n, rev_words, rev_n_words, rev_begin, r.len, 0u);
}

extern "C" __attribute__((visibility("default")))
int nvbio_aligner_best_approx(const nvbio_hip_fmindex* fmi, const nvbio_hip_fmindex* rfmi, uint32_t n, uint32_t L,
                              const uint32_t* d_rev_words, uint64_t rev_n_words, const uint64_t* d_rev_begin,
                              const uint32_t* d_fwrc_words, uint64_t fwrc_n_words, const uint8_t* d_quals, uint64_t n_quals,
                              const char* d_names, const uint32_t* d_names_idx,
                              const uint32_t* d_genome_words, uint64_t genome_n_words, uint32_t genome_len, const shim_params* sp,
                              uint64_t* h_best /* 2n */, uint8_t* h_mapq, uint16_t* h_cigar /* n*64 */, uint32_t* h_cigar_len, uint32_t* h_source, uint32_t* h_sink,
                              int32_t* h_tb_score, uint64_t* h_stats /* extensions, rounds, seeding_passes, n_queue, queue[8] */,
                              uint8_t* h_mds /* n*256 */, uint32_t* h_mds_len)
{
    try {
        Params params;
        params.seed_len = sp->seed_len; params.seed_freq = SimpleFunc(SimpleFunc::Type(sp->seed_freq_type), sp->seed_freq_k, sp->seed_freq_m);
        params.min_read_len = sp->min_read_len; params.max_hits = sp->max_hits; params.max_reseed = sp->max_reseed; params.rep_seeds = sp->rep_seeds;
        params.allow_sub = sp->allow_sub; params.subseed_len = sp->subseed_len;
        params.select.randomized = sp->randomized != 0; params.select.top_seed = sp->top_seed; params.select.max_effort_init = sp->max_effort_init;
        params.select.max_effort = sp->max_effort; params.select.min_ext = sp->min_ext; params.select.max_ext = sp->max_ext;
        params.max_dist = sp->max_dist; params.alignment_type = sp->local ? LocalAlignment : EndToEndAlignment;
        params.no_multi_hits = sp->no_multi_hits != 0; params.hits_stride = sp->hits_stride; params.finish_alignments = sp->finish != 0; params.scoring_mode = sp->edit_distance ? EditDistanceMode : SmithWatermanMode;

        aln::SmithWatermanScoringScheme scheme = sp->local ? aln::SmithWatermanScoringScheme::local() : aln::SmithWatermanScoringScheme();
        scheme.m_match = sp->match;
        const ScoreLimits limits(sp->match, SimpleFunc(SimpleFunc::Type(sp->score_min_type), sp->score_min_k, sp->score_min_m));

        fm_index_device f, rf; f.m = *fmi; rf.m = rfmi ? *rfmi : *fmi;
        ReadBatch reads;
        reads.n = n; reads.len = L;
        reads.reversed = PackedStringSetView<4, true>(n, d_rev_words, rev_n_words, d_rev_begin, nullptr, L);
        reads.fw_rc_words = d_fwrc_words; reads.fw_rc_n_words = fwrc_n_words; reads.rc_offset = uint64_t(n) * L;
        reads.quals = d_quals; reads.n_quals = n_quals; reads.names = d_names; reads.names_idx = d_names_idx;
        apply_ragged(reads, 0, d_rev_words, rev_n_words, d_rev_begin);

        Aligner aligner;
        aligner.init(std::max(sp->batch_size, n), sp->batch_size);
        Stats stats;
        aligner.best_approx(params, f, rf, scheme, limits, d_genome_words, genome_n_words, genome_len, reads, stats);

        const std::vector<io::Alignment> best = aligner.best_data_dvec.to_host();
        for (uint32_t i = 0; i < n; ++i) { memcpy(&h_best[i], &best[i], 8); memcpy(&h_best[n + i], &best[aligner.BATCH_SIZE + i], 8); }
        const std::vector<uint8> mq = aligner.mapq_dvec.to_host();         memcpy(h_mapq, mq.data(), n);
        const std::vector<io::Cigar> cg = aligner.cigar.to_host();         memcpy(h_cigar, cg.data(), size_t(n) * aligner.cigar_stride * 2u);
        const std::vector<uint32> cl = aligner.cigar_len.to_host();        memcpy(h_cigar_len, cl.data(), size_t(n) * 4u);
        const std::vector<uint32> so = aligner.cigar_source.to_host();     memcpy(h_source, so.data(), size_t(n) * 8u);
        const std::vector<uint32> si = aligner.cigar_sink.to_host();       memcpy(h_sink, si.data(), size_t(n) * 8u);
        const std::vector<int32> ts = aligner.traceback_score.to_host();   memcpy(h_tb_score, ts.data(), size_t(n) * 4u);
        const std::vector<uint8> md = aligner.mds.to_host();               memcpy(h_mds, md.data(), size_t(n) * aligner.mds_stride);
        const std::vector<uint32> ml = aligner.mds_len.to_host();          memcpy(h_mds_len, ml.data(), size_t(n) * 4u);
        h_stats[0] = stats.extensions; h_stats[1] = stats.rounds; h_stats[2] = stats.seeding_passes; h_stats[3] = stats.queue.size();
        for (size_t k = 0; k < stats.queue.size() && k < 8; ++k) h_stats[4 + k] = stats.queue[k];
        return 0;
    } catch (const nvbio::hip_error& e) { fprintf(stderr, "aligner_shim: %s\n", e.what()); return 1; }
}

// The same driver for timing (bench.py's e2e_leg.cxx_best_approx): one Aligner kept across `reps` batches, wall time per batch
// between device synchronisations, then one more batch with the stage clock on.  out_ms[0] = mean ms per batch,
// out_stage_ms = {map, select_init, select, locate, score, reduce, mapq, traceback, finish}; h_best / h_mapq as above.
extern "C" __attribute__((visibility("default")))
int nvbio_aligner_best_approx_timed(const nvbio_hip_fmindex* fmi, const nvbio_hip_fmindex* rfmi, uint32_t n, uint32_t L,
                                    const uint32_t* d_rev_words, uint64_t rev_n_words, const uint64_t* d_rev_begin,
                                    const uint32_t* d_fwrc_words, uint64_t fwrc_n_words, const uint8_t* d_quals, uint64_t n_quals,
                                    const char* d_names, const uint32_t* d_names_idx,
                                    const uint32_t* d_genome_words, uint64_t genome_n_words, uint32_t genome_len, const shim_params* sp,
                                    uint32_t reps, double* out_ms, double* out_stage_ms /* 9 */, uint64_t* d_best /* device, 2n */, uint8_t* d_mapq /* device */,
                                    uint64_t* h_stats)
{
    try {
        Params params;
        params.seed_len = sp->seed_len; params.seed_freq = SimpleFunc(SimpleFunc::Type(sp->seed_freq_type), sp->seed_freq_k, sp->seed_freq_m);
        params.min_read_len = sp->min_read_len; params.max_hits = sp->max_hits; params.max_reseed = sp->max_reseed; params.rep_seeds = sp->rep_seeds;
        params.allow_sub = sp->allow_sub; params.subseed_len = sp->subseed_len;
        params.select.randomized = sp->randomized != 0; params.select.top_seed = sp->top_seed; params.select.max_effort_init = sp->max_effort_init;
        params.select.max_effort = sp->max_effort; params.select.min_ext = sp->min_ext; params.select.max_ext = sp->max_ext;
        params.max_dist = sp->max_dist; params.alignment_type = sp->local ? LocalAlignment : EndToEndAlignment;
        params.no_multi_hits = sp->no_multi_hits != 0; params.hits_stride = sp->hits_stride; params.finish_alignments = sp->finish != 0; params.scoring_mode = sp->edit_distance ? EditDistanceMode : SmithWatermanMode;
        aln::SmithWatermanScoringScheme scheme = sp->local ? aln::SmithWatermanScoringScheme::local() : aln::SmithWatermanScoringScheme();
        scheme.m_match = sp->match;
        const ScoreLimits limits(sp->match, SimpleFunc(SimpleFunc::Type(sp->score_min_type), sp->score_min_k, sp->score_min_m));
        fm_index_device f, rf; f.m = *fmi; rf.m = rfmi ? *rfmi : *fmi;
        ReadBatch reads;
        reads.n = n; reads.len = L;
        reads.reversed = PackedStringSetView<4, true>(n, d_rev_words, rev_n_words, d_rev_begin, nullptr, L);
        reads.fw_rc_words = d_fwrc_words; reads.fw_rc_n_words = fwrc_n_words; reads.rc_offset = uint64_t(n) * L;
        reads.quals = d_quals; reads.n_quals = n_quals; reads.names = d_names; reads.names_idx = d_names_idx;

        Aligner aligner;
        aligner.init(std::max(sp->batch_size, n), sp->batch_size);
        // two warm-up batches: the first sizes the Aligner's workspace, the second finds it in one block
        for (int w = 0; w < 2; ++w) { Stats warm; aligner.best_approx(params, f, rf, scheme, limits, d_genome_words, genome_n_words, genome_len, reads, warm); }
        hip::synchronize();
        double total = 0.0;
        Stats stats;
        for (uint32_t r = 0; r < reps; ++r)
        {
            stats = Stats();
            const auto t0 = std::chrono::steady_clock::now();
            aligner.best_approx(params, f, rf, scheme, limits, d_genome_words, genome_n_words, genome_len, reads, stats);
            hip::synchronize();
            total += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        }
        out_ms[0] = reps ? total / reps : 0.0;
        if (getenv("NVBIO_SHIM_TRACE")) {
            fprintf(stderr, "aligner_shim: %.1f ms per batch (%u batches incl. warm-up)\n", out_ms[0], reps + 2u);
        }
        Stats timed; timed.clock.enabled = true;
        aligner.best_approx(params, f, rf, scheme, limits, d_genome_words, genome_n_words, genome_len, reads, timed);
        hip::synchronize();
        const char* names[9] = { "map", "select_init", "select", "locate", "score", "reduce", "mapq", "traceback", "finish" };
        for (int k = 0; k < 9; ++k) out_stage_ms[k] = timed.clock.ms.count(names[k]) ? timed.clock.ms[names[k]] : 0.0;
        hip_check(nvbio_hip_memcpy(d_best, aligner.best_data_dvec.data(), uint64_t(n) * 8u, 3, nullptr), "d2d");
        hip_check(nvbio_hip_memcpy(d_best + n, aligner.best_data_dvec.data() + aligner.BATCH_SIZE, uint64_t(n) * 8u, 3, nullptr), "d2d");
        hip_check(nvbio_hip_memcpy(d_mapq, aligner.mapq_dvec.data(), n, 3, nullptr), "d2d");
        hip::synchronize();
        h_stats[0] = stats.extensions; h_stats[1] = stats.rounds; h_stats[2] = stats.seeding_passes; h_stats[3] = timed.dp_jobs;      // (the DP jobs are only counted under the stage clock)
        return 0;
    } catch (const std::exception& e) { fprintf(stderr, "aligner_shim: %s\n", e.what()); return 1; }
}

// Co-scheduling probe / pipelined driver (bench.py's config-4 leg, tools/cosched_probe.py): `n_batches` batches through the single-end
// driver with `n_workers` host threads, each owning one Aligner and one non-blocking HIP stream (worker w takes batches w, w + n_workers,
// ...), the way the reference runs one host thread per device -- here several per device, so that one batch's fabric-bound seeding
// overlaps another's VALU-bound extension.
// batch b's reads: d_rev_words[b] / d_rev_begin[b] / d_fwrc_words[b] (device pointers, one set per batch).
// out_wall_ms = wall time of all batches between two device synchronisations (after one untimed warm-up pass that sizes every worker's
// workspace); d_best[b] (device, 2n) / d_mapq[b] receive every batch's results.
#include <thread>
#include <mutex>
extern "C" __attribute__((visibility("default")))
int nvbio_aligner_best_approx_pipelined_names(const nvbio_hip_fmindex* fmi, const nvbio_hip_fmindex* rfmi, uint32_t n, uint32_t L, uint32_t n_batches,
                                        const uint32_t* const* d_rev_words, uint64_t rev_n_words, const uint64_t* const* d_rev_begin,
                                        const uint32_t* const* d_fwrc_words, uint64_t fwrc_n_words, const uint8_t* const* d_quals, uint64_t n_quals,
                                        const char* const* d_names, const uint32_t* const* d_names_idx,
                                        const uint32_t* d_genome_words, uint64_t genome_n_words, uint32_t genome_len, const shim_params* sp,
                                        uint32_t n_workers, uint32_t reps, double* out_wall_ms,
                                        uint64_t* const* d_best, uint8_t* const* d_mapq)
{
    try {
        Params params;
        params.seed_len = sp->seed_len; params.seed_freq = SimpleFunc(SimpleFunc::Type(sp->seed_freq_type), sp->seed_freq_k, sp->seed_freq_m);
        params.min_read_len = sp->min_read_len; params.max_hits = sp->max_hits; params.max_reseed = sp->max_reseed; params.rep_seeds = sp->rep_seeds;
        params.allow_sub = sp->allow_sub; params.subseed_len = sp->subseed_len;
        params.select.randomized = sp->randomized != 0; params.select.top_seed = sp->top_seed; params.select.max_effort_init = sp->max_effort_init;
        params.select.max_effort = sp->max_effort; params.select.min_ext = sp->min_ext; params.select.max_ext = sp->max_ext;
        params.max_dist = sp->max_dist; params.alignment_type = sp->local ? LocalAlignment : EndToEndAlignment;
        params.no_multi_hits = sp->no_multi_hits != 0; params.hits_stride = sp->hits_stride; params.finish_alignments = sp->finish != 0; params.scoring_mode = sp->edit_distance ? EditDistanceMode : SmithWatermanMode;
        aln::SmithWatermanScoringScheme scheme = sp->local ? aln::SmithWatermanScoringScheme::local() : aln::SmithWatermanScoringScheme();
        scheme.m_match = sp->match;
        const ScoreLimits limits(sp->match, SimpleFunc(SimpleFunc::Type(sp->score_min_type), sp->score_min_k, sp->score_min_m));
        fm_index_device f, rf; f.m = *fmi; rf.m = rfmi ? *rfmi : *fmi;
        if (n_workers == 0) n_workers = 1;

        std::vector<Aligner*> aligners(n_workers, nullptr);
        std::vector<void*>    streams(n_workers, nullptr);
        for (uint32_t w = 0; w < n_workers; ++w)
        {
            aligners[w] = new Aligner();
            aligners[w]->init(std::max(sp->batch_size, n), sp->batch_size);
            if (n_workers > 1) hip_check(nvbio_hip_stream_create(&streams[w], 1u), "nvbio_hip_stream_create");
        }
        std::vector<int> failed(n_workers, 0);
        auto worker = [&](const uint32_t w)
        {
            try {
                for (uint32_t b = w; b < n_batches; b += n_workers)
                {
                    ReadBatch reads;
                    reads.n = n; reads.len = L;
                    reads.reversed = PackedStringSetView<4, true>(n, d_rev_words[b], rev_n_words, d_rev_begin[b], nullptr, L);
                    reads.fw_rc_words = d_fwrc_words[b]; reads.fw_rc_n_words = fwrc_n_words; reads.rc_offset = uint64_t(n) * L;
                    reads.quals = d_quals[b]; reads.n_quals = n_quals; reads.names = d_names[b]; reads.names_idx = d_names_idx[b];
                    Stats stats;
                    aligners[w]->best_approx(params, f, rf, scheme, limits, d_genome_words, genome_n_words, genome_len, reads, stats, streams[w]);
                    hip_check(nvbio_hip_memcpy(d_best[b], aligners[w]->best_data_dvec.data(), uint64_t(n) * 8u, 3, streams[w]), "d2d");
                    hip_check(nvbio_hip_memcpy(d_best[b] + n, aligners[w]->best_data_dvec.data() + aligners[w]->BATCH_SIZE, uint64_t(n) * 8u, 3, streams[w]), "d2d");
                    hip_check(nvbio_hip_memcpy(d_mapq[b], aligners[w]->mapq_dvec.data(), n, 3, streams[w]), "d2d");
                    hip::synchronize(streams[w]);
                }
            } catch (const std::exception& e) { fprintf(stderr, "aligner_shim worker %u: %s\n", w, e.what()); failed[w] = 1; }
        };
        auto pass = [&]()
        {
            if (n_workers == 1) { worker(0); return; }
            std::vector<std::thread> th;
            for (uint32_t w = 0; w < n_workers; ++w) th.emplace_back(worker, w);
            for (auto& t : th) t.join();
        };
        pass();                                            // warm-up: every worker's workspace reaches its final size ...
        if (n_batches < 2u * n_workers) pass();            // ... and is merged into one block, which happens at a worker's SECOND batch
        hip::synchronize();
        double total = 0.0;
        for (uint32_t r = 0; r < reps; ++r)
        {
            const auto t0 = std::chrono::steady_clock::now();
            pass();
            hip::synchronize();
            total += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        }
        out_wall_ms[0] = reps ? total / reps : 0.0;
        for (uint32_t w = 0; w < n_workers; ++w) { delete aligners[w]; if (streams[w]) nvbio_hip_stream_destroy(streams[w]); }
        for (uint32_t w = 0; w < n_workers; ++w) if (failed[w]) return 1;
        return 0;
    } catch (const std::exception& e) { fprintf(stderr, "aligner_shim: %s\n", e.what()); return 1; }
}

// the same with one quality stream and one name arena shared by every batch (bench.py's config-4 leg: constant qualities)
extern "C" __attribute__((visibility("default")))
int nvbio_aligner_best_approx_pipelined(const nvbio_hip_fmindex* fmi, const nvbio_hip_fmindex* rfmi, uint32_t n, uint32_t L, uint32_t n_batches,
                                        const uint32_t* const* d_rev_words, uint64_t rev_n_words, const uint64_t* const* d_rev_begin,
                                        const uint32_t* const* d_fwrc_words, uint64_t fwrc_n_words, const uint8_t* d_quals, uint64_t n_quals,
                                        const char* d_names, const uint32_t* d_names_idx,
                                        const uint32_t* d_genome_words, uint64_t genome_n_words, uint32_t genome_len, const shim_params* sp,
                                        uint32_t n_workers, uint32_t reps, double* out_wall_ms,
                                        uint64_t* const* d_best, uint8_t* const* d_mapq)
{
    std::vector<const uint8_t*> q(n_batches, d_quals); std::vector<const char*> nm(n_batches, d_names); std::vector<const uint32_t*> ni(n_batches, d_names_idx);
    return nvbio_aligner_best_approx_pipelined_names(fmi, rfmi, n, L, n_batches, d_rev_words, rev_n_words, d_rev_begin, d_fwrc_words, fwrc_n_words, q.data(), n_quals,
                                                     nm.data(), ni.data(), d_genome_words, genome_n_words, genome_len, sp, n_workers, reps, out_wall_ms, d_best, d_mapq);
}

struct shim_pe_params { int32_t pe_policy; uint32_t pe_overlap, pe_unpaired, pe_discordant, min_frag_len, max_frag_len; };

// the paired-end driver: per mate {reversed words, begin, fw+rc words, quals, names}, the joint pattern stream of the tracebacks;
// outputs per slot set: best[2n], mapq[n], cigar[n*64], cigar_len[n], source[2n], sink[2n], mds[n*256], mds_len[n]
static int paired_impl(const nvbio_hip_fmindex* fmi, const nvbio_hip_fmindex* rfmi, uint32_t n, uint32_t L,
                                     const uint32_t* const* d_rev_words, const uint64_t* rev_n_words, const uint64_t* const* d_rev_begin,
                                     const uint32_t* const* d_fwrc_words, const uint64_t* fwrc_n_words, const uint8_t* d_quals, uint64_t n_quals,
                                     const char* d_names, const uint32_t* d_names_idx,
                                     const uint32_t* d_both_words, uint64_t both_n_words, uint64_t mate_offset, const uint8_t* d_both_quals, uint64_t both_n_quals,
                                     const uint32_t* d_genome_words, uint64_t genome_n_words, uint32_t genome_len, const shim_params* sp, const shim_pe_params* pp,
                                     uint64_t* const* h_best, uint8_t* const* h_mapq, uint16_t* const* h_cigar, uint32_t* const* h_cigar_len, uint32_t* const* h_source,
                                     uint32_t* const* h_sink, uint8_t* const* h_mds, uint32_t* const* h_mds_len, uint64_t* h_stats,
                                     uint32_t reps, double* out_ms, double* out_stage_ms /* 10 */, const uint8_t* const* d_mate_quals = nullptr /* 2: each mate's own stream */)
{
    try {
        Params params;
        params.seed_len = sp->seed_len; params.seed_freq = SimpleFunc(SimpleFunc::Type(sp->seed_freq_type), sp->seed_freq_k, sp->seed_freq_m);
        params.min_read_len = sp->min_read_len; params.max_hits = sp->max_hits; params.max_reseed = sp->max_reseed; params.rep_seeds = sp->rep_seeds;
        params.allow_sub = sp->allow_sub; params.subseed_len = sp->subseed_len;
        params.select.randomized = sp->randomized != 0; params.select.top_seed = sp->top_seed; params.select.max_effort_init = sp->max_effort_init;
        params.select.max_effort = sp->max_effort; params.select.min_ext = sp->min_ext; params.select.max_ext = sp->max_ext;
        params.max_dist = sp->max_dist; params.alignment_type = sp->local ? LocalAlignment : EndToEndAlignment;
        params.no_multi_hits = sp->no_multi_hits != 0; params.hits_stride = sp->hits_stride; params.finish_alignments = sp->finish != 0; params.scoring_mode = sp->edit_distance ? EditDistanceMode : SmithWatermanMode;
        PairedParams pe;
        pe.pe_policy = pp->pe_policy; pe.pe_overlap = pp->pe_overlap != 0; pe.pe_unpaired = pp->pe_unpaired != 0; pe.pe_discordant = pp->pe_discordant != 0;
        pe.min_frag_len = pp->min_frag_len; pe.max_frag_len = pp->max_frag_len;

        aln::SmithWatermanScoringScheme scheme = sp->local ? aln::SmithWatermanScoringScheme::local() : aln::SmithWatermanScoringScheme();
        scheme.m_match = sp->match;
        const ScoreLimits limits(sp->match, SimpleFunc(SimpleFunc::Type(sp->score_min_type), sp->score_min_k, sp->score_min_m));
        fm_index_device f, rf; f.m = *fmi; rf.m = rfmi ? *rfmi : *fmi;

        PairedReadBatch reads;
        for (int m = 0; m < 2; ++m) {
            ReadBatch& b = reads.mate[m];
            b.n = n; b.len = L;
            b.reversed = PackedStringSetView<4, true>(n, d_rev_words[m], rev_n_words[m], d_rev_begin[m], nullptr, L);
            b.fw_rc_words = d_fwrc_words[m]; b.fw_rc_n_words = fwrc_n_words[m]; b.rc_offset = uint64_t(n) * L;
            b.quals = d_mate_quals ? d_mate_quals[m] : d_quals; b.n_quals = n_quals; b.names = d_names; b.names_idx = d_names_idx;
            apply_ragged(b, m, d_rev_words[m], rev_n_words[m], d_rev_begin[m]);
        }
        reads.both_words = d_both_words; reads.both_n_words = both_n_words; reads.mate_offset = mate_offset;
        reads.both_quals = d_both_quals; reads.both_n_quals = both_n_quals;

        Aligner aligner;
        aligner.init(std::max(sp->batch_size, n), sp->batch_size);
        aligner.init_paired();
        Stats stats;
        aligner.best_approx(params, pe, f, rf, scheme, limits, d_genome_words, genome_n_words, genome_len, reads, stats);
        if (reps)
        {
            // timed mode: the batch above sized the workspace; one more warm-up, then `reps` batches on the wall clock, then one with the stage clock
            { Stats warm; aligner.best_approx(params, pe, f, rf, scheme, limits, d_genome_words, genome_n_words, genome_len, reads, warm); }
            hip::synchronize();
            double total = 0.0;
            for (uint32_t r = 0; r < reps; ++r)
            {
                stats = Stats();
                const auto t0 = std::chrono::steady_clock::now();
                aligner.best_approx(params, pe, f, rf, scheme, limits, d_genome_words, genome_n_words, genome_len, reads, stats);
                hip::synchronize();
                total += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            }
            out_ms[0] = total / reps;
            Stats timed; timed.clock.enabled = true;
            aligner.best_approx(params, pe, f, rf, scheme, limits, d_genome_words, genome_n_words, genome_len, reads, timed);
            hip::synchronize();
            const char* names[10] = { "map", "select_init", "select", "locate", "anchor_score", "opposite_score", "reduce", "mapq", "traceback", "finish" };
            for (int k = 0; k < 10; ++k) out_stage_ms[k] = timed.clock.ms.count(names[k]) ? timed.clock.ms[names[k]] : 0.0;
        }

        const uint32_t B = aligner.BATCH_SIZE;
        for (int slot = 0; slot < 2 && h_best; ++slot) {
            const std::vector<io::Alignment> best = (slot ? aligner.best_data_dvec_o : aligner.best_data_dvec).to_host();
            for (uint32_t i = 0; i < n; ++i) { memcpy(&h_best[slot][i], &best[i], 8); memcpy(&h_best[slot][n + i], &best[B + i], 8); }
            const std::vector<uint8> mq = (slot ? aligner.mapq_dvec_o : aligner.mapq_dvec).to_host();              memcpy(h_mapq[slot], mq.data(), n);
            const std::vector<io::Cigar> cg = (slot ? aligner.cigar_o : aligner.cigar).to_host();                   memcpy(h_cigar[slot], cg.data(), size_t(n) * aligner.cigar_stride * 2u);
            const std::vector<uint32> cl = (slot ? aligner.cigar_len_o : aligner.cigar_len).to_host();              memcpy(h_cigar_len[slot], cl.data(), size_t(n) * 4u);
            const std::vector<uint32> so = (slot ? aligner.cigar_source_o : aligner.cigar_source).to_host();        memcpy(h_source[slot], so.data(), size_t(n) * 8u);
            const std::vector<uint32> si = (slot ? aligner.cigar_sink_o : aligner.cigar_sink).to_host();            memcpy(h_sink[slot], si.data(), size_t(n) * 8u);
            const std::vector<uint8> md = (slot ? aligner.mds_o : aligner.mds).to_host();                           memcpy(h_mds[slot], md.data(), size_t(n) * aligner.mds_stride);
            const std::vector<uint32> ml = (slot ? aligner.mds_len_o : aligner.mds_len).to_host();                  memcpy(h_mds_len[slot], ml.data(), size_t(n) * 4u);
        }
        h_stats[0] = stats.extensions; h_stats[1] = stats.rounds; h_stats[2] = stats.seeding_passes; h_stats[3] = stats.queue.size();
        for (size_t k = 0; k < stats.queue.size() && k < 8; ++k) h_stats[4 + k] = stats.queue[k];
        return 0;
    } catch (const nvbio::hip_error& e) { fprintf(stderr, "aligner_shim: %s\n", e.what()); return 1; }
}

extern "C" __attribute__((visibility("default")))
int nvbio_aligner_best_approx_paired(const nvbio_hip_fmindex* fmi, const nvbio_hip_fmindex* rfmi, uint32_t n, uint32_t L,
                                     const uint32_t* const* d_rev_words, const uint64_t* rev_n_words, const uint64_t* const* d_rev_begin,
                                     const uint32_t* const* d_fwrc_words, const uint64_t* fwrc_n_words, const uint8_t* d_quals, uint64_t n_quals,
                                     const char* d_names, const uint32_t* d_names_idx,
                                     const uint32_t* d_both_words, uint64_t both_n_words, uint64_t mate_offset, const uint8_t* d_both_quals, uint64_t both_n_quals,
                                     const uint32_t* d_genome_words, uint64_t genome_n_words, uint32_t genome_len, const shim_params* sp, const shim_pe_params* pp,
                                     uint64_t* const* h_best, uint8_t* const* h_mapq, uint16_t* const* h_cigar, uint32_t* const* h_cigar_len, uint32_t* const* h_source,
                                     uint32_t* const* h_sink, uint8_t* const* h_mds, uint32_t* const* h_mds_len, uint64_t* h_stats)
{
    return paired_impl(fmi, rfmi, n, L, d_rev_words, rev_n_words, d_rev_begin, d_fwrc_words, fwrc_n_words, d_quals, n_quals, d_names, d_names_idx, d_both_words, both_n_words,
                       mate_offset, d_both_quals, both_n_quals, d_genome_words, genome_n_words, genome_len, sp, pp, h_best, h_mapq, h_cigar, h_cigar_len, h_source, h_sink,
                       h_mds, h_mds_len, h_stats, 0u, nullptr, nullptr);
}
// ... with each mate's own quality stream (fw + rc order, n_quals bytes each); d_both_quals holds mate m's at m * mate_offset
extern "C" __attribute__((visibility("default")))
int nvbio_aligner_best_approx_paired_quals(const nvbio_hip_fmindex* fmi, const nvbio_hip_fmindex* rfmi, uint32_t n, uint32_t L,
                                     const uint32_t* const* d_rev_words, const uint64_t* rev_n_words, const uint64_t* const* d_rev_begin,
                                     const uint32_t* const* d_fwrc_words, const uint64_t* fwrc_n_words, const uint8_t* const* d_mate_quals, uint64_t n_quals,
                                     const char* d_names, const uint32_t* d_names_idx,
                                     const uint32_t* d_both_words, uint64_t both_n_words, uint64_t mate_offset, const uint8_t* d_both_quals, uint64_t both_n_quals,
                                     const uint32_t* d_genome_words, uint64_t genome_n_words, uint32_t genome_len, const shim_params* sp, const shim_pe_params* pp,
                                     uint64_t* const* h_best, uint8_t* const* h_mapq, uint16_t* const* h_cigar, uint32_t* const* h_cigar_len, uint32_t* const* h_source,
                                     uint32_t* const* h_sink, uint8_t* const* h_mds, uint32_t* const* h_mds_len, uint64_t* h_stats, double* out_ms)
{
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = paired_impl(fmi, rfmi, n, L, d_rev_words, rev_n_words, d_rev_begin, d_fwrc_words, fwrc_n_words, d_mate_quals[0], n_quals, d_names, d_names_idx, d_both_words, both_n_words,
                       mate_offset, d_both_quals, both_n_quals, d_genome_words, genome_n_words, genome_len, sp, pp, h_best, h_mapq, h_cigar, h_cigar_len, h_source, h_sink,
                       h_mds, h_mds_len, h_stats, 0u, nullptr, nullptr, d_mate_quals);
    if (out_ms) out_ms[0] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return rc;
}
// the same driver on the wall clock (bench.py: e2e_leg.config5_shape_paired_end.cxx): mean of `reps` batches with one Aligner object,
// stage times from one more; only the best alignments of both slot sets come back (host, 2n words each)
extern "C" __attribute__((visibility("default")))
int nvbio_aligner_best_approx_paired_timed(const nvbio_hip_fmindex* fmi, const nvbio_hip_fmindex* rfmi, uint32_t n, uint32_t L,
                                     const uint32_t* const* d_rev_words, const uint64_t* rev_n_words, const uint64_t* const* d_rev_begin,
                                     const uint32_t* const* d_fwrc_words, const uint64_t* fwrc_n_words, const uint8_t* d_quals, uint64_t n_quals,
                                     const char* d_names, const uint32_t* d_names_idx,
                                     const uint32_t* d_both_words, uint64_t both_n_words, uint64_t mate_offset, const uint8_t* d_both_quals, uint64_t both_n_quals,
                                     const uint32_t* d_genome_words, uint64_t genome_n_words, uint32_t genome_len, const shim_params* sp, const shim_pe_params* pp,
                                     uint32_t reps, double* out_ms, double* out_stage_ms, uint64_t* h_stats)
{
    return paired_impl(fmi, rfmi, n, L, d_rev_words, rev_n_words, d_rev_begin, d_fwrc_words, fwrc_n_words, d_quals, n_quals, d_names, d_names_idx, d_both_words, both_n_words,
                       mate_offset, d_both_quals, both_n_quals, d_genome_words, genome_n_words, genome_len, sp, pp, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                       nullptr, nullptr, h_stats, reps ? reps : 1u, out_ms, out_stage_ms);
}

// The paired-end driver over many batches (bench.py's config-5 leg at its per-GPU share): `n_batches` batches of n pairs each with one
// Aligner per worker thread / stream, as nvbio_aligner_best_approx_pipelined.  A batch's inputs: shim_pair_batch (device pointers).
// d_records[b]: device, 32 bytes per pair = io::BestPairedAlignments {anchor best, anchor second, opposite best, opposite second}
// (nvbio/io/alignments.h:222-300) -- the record a multi-GPU run gathers.
struct shim_pair_batch { const uint32_t* rev_words[2]; const uint64_t* rev_begin[2]; const uint32_t* fwrc_words[2]; const uint32_t* both_words; };
extern "C" __attribute__((visibility("default")))
int nvbio_aligner_best_approx_paired_pipelined(const nvbio_hip_fmindex* fmi, const nvbio_hip_fmindex* rfmi, uint32_t n, uint32_t L, uint32_t n_batches,
                                               const shim_pair_batch* batches, const uint64_t* rev_n_words, const uint64_t* fwrc_n_words,
                                               const uint8_t* d_quals, uint64_t n_quals, const char* d_names, const uint32_t* d_names_idx,
                                               uint64_t both_n_words, uint64_t mate_offset, const uint8_t* d_both_quals, uint64_t both_n_quals,
                                               const uint32_t* d_genome_words, uint64_t genome_n_words, uint32_t genome_len, const shim_params* sp, const shim_pe_params* pp,
                                               uint32_t n_workers, double* out_wall_ms, uint64_t* const* d_records, uint64_t* h_stats /* extensions, rounds */)
{
    try {
        Params params;
        params.seed_len = sp->seed_len; params.seed_freq = SimpleFunc(SimpleFunc::Type(sp->seed_freq_type), sp->seed_freq_k, sp->seed_freq_m);
        params.min_read_len = sp->min_read_len; params.max_hits = sp->max_hits; params.max_reseed = sp->max_reseed; params.rep_seeds = sp->rep_seeds;
        params.allow_sub = sp->allow_sub; params.subseed_len = sp->subseed_len;
        params.select.randomized = sp->randomized != 0; params.select.top_seed = sp->top_seed; params.select.max_effort_init = sp->max_effort_init;
        params.select.max_effort = sp->max_effort; params.select.min_ext = sp->min_ext; params.select.max_ext = sp->max_ext;
        params.max_dist = sp->max_dist; params.alignment_type = sp->local ? LocalAlignment : EndToEndAlignment;
        params.no_multi_hits = sp->no_multi_hits != 0; params.hits_stride = sp->hits_stride; params.finish_alignments = sp->finish != 0; params.scoring_mode = sp->edit_distance ? EditDistanceMode : SmithWatermanMode;
        PairedParams pe;
        pe.pe_policy = pp->pe_policy; pe.pe_overlap = pp->pe_overlap != 0; pe.pe_unpaired = pp->pe_unpaired != 0; pe.pe_discordant = pp->pe_discordant != 0;
        pe.min_frag_len = pp->min_frag_len; pe.max_frag_len = pp->max_frag_len;
        aln::SmithWatermanScoringScheme scheme = sp->local ? aln::SmithWatermanScoringScheme::local() : aln::SmithWatermanScoringScheme();
        scheme.m_match = sp->match;
        const ScoreLimits limits(sp->match, SimpleFunc(SimpleFunc::Type(sp->score_min_type), sp->score_min_k, sp->score_min_m));
        fm_index_device f, rf; f.m = *fmi; rf.m = rfmi ? *rfmi : *fmi;
        if (n_workers == 0) n_workers = 1;
        std::vector<Aligner*> aligners(n_workers, nullptr);
        std::vector<void*>    streams(n_workers, nullptr);
        for (uint32_t w = 0; w < n_workers; ++w)
        {
            aligners[w] = new Aligner();
            aligners[w]->init(std::max(sp->batch_size, n), sp->batch_size);
            aligners[w]->init_paired();
            if (n_workers > 1) hip_check(nvbio_hip_stream_create(&streams[w], 1u), "nvbio_hip_stream_create");
        }
        std::vector<int> failed(n_workers, 0);
        std::vector<uint64_t> ext(n_workers, 0), rounds(n_workers, 0);
        auto run_batch = [&](const uint32_t w, const uint32_t b)
        {
            PairedReadBatch reads;
            for (int m = 0; m < 2; ++m) {
                ReadBatch& rb = reads.mate[m];
                rb.n = n; rb.len = L;
                rb.reversed = PackedStringSetView<4, true>(n, batches[b].rev_words[m], rev_n_words[m], batches[b].rev_begin[m], nullptr, L);
                rb.fw_rc_words = batches[b].fwrc_words[m]; rb.fw_rc_n_words = fwrc_n_words[m]; rb.rc_offset = uint64_t(n) * L;
                rb.quals = d_quals; rb.n_quals = n_quals; rb.names = d_names; rb.names_idx = d_names_idx;
            }
            reads.both_words = batches[b].both_words; reads.both_n_words = both_n_words; reads.mate_offset = mate_offset;
            reads.both_quals = d_both_quals; reads.both_n_quals = both_n_quals;
            Stats stats;
            Aligner& al = *aligners[w];
            al.best_approx(params, pe, f, rf, scheme, limits, d_genome_words, genome_n_words, genome_len, reads, stats, streams[w]);
            ext[w] += stats.extensions; rounds[w] += stats.rounds;
            // the 32-byte pair record: [a1 | a2 | o1 | o2] as four planes of n words
            hip_check(nvbio_hip_memcpy(d_records[b],          al.best_data_dvec.data(),                   uint64_t(n) * 8u, 3, streams[w]), "d2d");
            hip_check(nvbio_hip_memcpy(d_records[b] + n,      al.best_data_dvec.data() + al.BATCH_SIZE,   uint64_t(n) * 8u, 3, streams[w]), "d2d");
            hip_check(nvbio_hip_memcpy(d_records[b] + 2u * n, al.best_data_dvec_o.data(),                 uint64_t(n) * 8u, 3, streams[w]), "d2d");
            hip_check(nvbio_hip_memcpy(d_records[b] + 3u * n, al.best_data_dvec_o.data() + al.BATCH_SIZE, uint64_t(n) * 8u, 3, streams[w]), "d2d");
            hip::synchronize(streams[w]);
        };
        auto worker = [&](const uint32_t w, const uint32_t first, const uint32_t count)
        {
            try { for (uint32_t b = first + w; b < first + count; b += n_workers) run_batch(w, b); }
            catch (const std::exception& e) { fprintf(stderr, "aligner_shim worker %u: %s\n", w, e.what()); failed[w] = 1; }
        };
        auto pass = [&](const uint32_t first, const uint32_t count)
        {
            if (n_workers == 1) { worker(0, first, count); return; }
            std::vector<std::thread> th;
            for (uint32_t w = 0; w < n_workers; ++w) th.emplace_back(worker, w, first, count);
            for (auto& t : th) t.join();
        };
        pass(0, std::min(n_batches, 2u * n_workers));          // warm-up on the first batches: every worker's workspace reaches its size
        hip::synchronize();
        for (uint32_t w = 0; w < n_workers; ++w) { ext[w] = 0; rounds[w] = 0; }
        const auto t0 = std::chrono::steady_clock::now();
        pass(0, n_batches);
        hip::synchronize();
        out_wall_ms[0] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        h_stats[0] = h_stats[1] = 0;
        for (uint32_t w = 0; w < n_workers; ++w) { h_stats[0] += ext[w]; h_stats[1] += rounds[w]; delete aligners[w]; if (streams[w]) nvbio_hip_stream_destroy(streams[w]); }
        for (uint32_t w = 0; w < n_workers; ++w) if (failed[w]) return 1;
        return 0;
    } catch (const std::exception& e) { fprintf(stderr, "aligner_shim: %s\n", e.what()); return 1; }
}

// the all-mapping driver (Aligner::all): outputs for up to out_cap alignments; *h_n = how many there are
extern "C" __attribute__((visibility("default")))
int nvbio_aligner_all(const nvbio_hip_fmindex* fmi, const nvbio_hip_fmindex* rfmi, uint32_t n, uint32_t L,
                      const uint32_t* d_rev_words, uint64_t rev_n_words, const uint64_t* d_rev_begin,
                      const uint32_t* d_fwrc_words, uint64_t fwrc_n_words, const uint8_t* d_quals, uint64_t n_quals,
                      const uint32_t* d_genome_words, uint64_t genome_n_words, uint32_t genome_len, const uint32_t* h_seq_index, uint32_t n_seq_index,
                      const shim_params* sp, uint64_t out_cap, uint64_t* h_n, uint32_t* h_read_id, uint64_t* h_alignments, uint64_t* h_scored,
                      uint16_t* h_cigar /* cap*64 */, uint32_t* h_cigar_len, uint32_t* h_source, uint32_t* h_sink, uint8_t* h_mds /* cap*256 */, uint32_t* h_mds_len,
                      uint64_t* h_stats /* hits, ranges, unique */)
{
    try {
        Params params;
        params.seed_len = sp->seed_len; params.seed_freq = SimpleFunc(SimpleFunc::Type(sp->seed_freq_type), sp->seed_freq_k, sp->seed_freq_m);
        params.min_read_len = sp->min_read_len; params.max_hits = sp->max_hits; params.max_reseed = sp->max_reseed; params.rep_seeds = sp->rep_seeds;
        params.allow_sub = sp->allow_sub; params.subseed_len = sp->subseed_len;
        params.max_dist = sp->max_dist; params.alignment_type = sp->local ? LocalAlignment : EndToEndAlignment; params.hits_stride = sp->hits_stride;
        params.scoring_mode = sp->edit_distance ? EditDistanceMode : SmithWatermanMode;
        aln::SmithWatermanScoringScheme scheme = sp->local ? aln::SmithWatermanScoringScheme::local() : aln::SmithWatermanScoringScheme();
        scheme.m_match = sp->match;
        const ScoreLimits limits(sp->match, SimpleFunc(SimpleFunc::Type(sp->score_min_type), sp->score_min_k, sp->score_min_m));
        fm_index_device f, rf; f.m = *fmi; rf.m = rfmi ? *rfmi : *fmi;
        ReadBatch reads;
        reads.n = n; reads.len = L;
        reads.reversed = PackedStringSetView<4, true>(n, d_rev_words, rev_n_words, d_rev_begin, nullptr, L);
        reads.fw_rc_words = d_fwrc_words; reads.fw_rc_n_words = fwrc_n_words; reads.rc_offset = uint64_t(n) * L;
        reads.quals = d_quals; reads.n_quals = n_quals; reads.names = nullptr; reads.names_idx = nullptr;
        apply_ragged(reads, 0, d_rev_words, rev_n_words, d_rev_begin);

        Aligner aligner;
        aligner.init(std::max(sp->batch_size, n), sp->batch_size);
        Stats stats;
        aligner.all(params, f, rf, scheme, limits, d_genome_words, genome_n_words, genome_len, std::vector<uint32>(h_seq_index, h_seq_index + n_seq_index), reads, stats);
        const uint64_t m = aligner.n_alignments;
        *h_n = m; h_stats[0] = stats.hits; h_stats[1] = stats.ranges; h_stats[2] = stats.unique;
        if (m == 0 || m > out_cap) return 0;
        const uint32_t cs = aligner.cigar_stride, ms = aligner.mds_stride;
        hip_check(nvbio_hip_memcpy(h_read_id, aligner.output_read_info_dvec.data(), m * 4u, 2, nullptr), "d2h");
        hip_check(nvbio_hip_memcpy(h_alignments, aligner.output_alignments_dvec.data(), m * 8u, 2, nullptr), "d2h");
        hip_check(nvbio_hip_memcpy(h_scored, aligner.scored_alignments_dvec.data(), m * 8u, 2, nullptr), "d2h");
        hip_check(nvbio_hip_memcpy(h_cigar, aligner.cigar.data(), m * cs * 2u, 2, nullptr), "d2h");
        hip_check(nvbio_hip_memcpy(h_cigar_len, aligner.cigar_len.data(), m * 4u, 2, nullptr), "d2h");
        hip_check(nvbio_hip_memcpy(h_source, aligner.cigar_source.data(), m * 8u, 2, nullptr), "d2h");
        hip_check(nvbio_hip_memcpy(h_sink, aligner.cigar_sink.data(), m * 8u, 2, nullptr), "d2h");
        hip_check(nvbio_hip_memcpy(h_mds, aligner.mds.data(), m * ms, 2, nullptr), "d2h");
        hip_check(nvbio_hip_memcpy(h_mds_len, aligner.mds_len.data(), m * 4u, 2, nullptr), "d2h");
        return 0;
    } catch (const std::exception& e) { fprintf(stderr, "aligner_shim: %s\n", e.what()); return 1; }
}

// include/nvbio_hip/multi_device.h: n_devices host threads, one per device (the reference's multi-GPU shape, nvBowtie.cpp:809-864), each
// owning a contiguous block of n_total records of record_words int32 words, gathered to rank 0 over RCCL.  Record i holds {i, rank, i ^ 0x5A5A, ...}.
// Returns 0 when the root's table is complete and in order; 10 + k on a mismatch at rank k's block.
#include <nvbio_hip/multi_device.h>
extern "C" __attribute__((visibility("default")))
int nvbio_multi_device_selftest(uint32_t n_devices, uint64_t n_total, uint32_t record_words)
{
    try {
        hip::DeviceGroup group;
        hip::DeviceGroup::local(group, n_devices);
        const uint32_t world = uint32_t(group.size());
        const std::vector<uint64> counts = hip::shard_sizes(n_total, world);
        std::vector<uint32_t> table(size_t(n_total) * record_words, 0u);
        group.run([&](const hip::DeviceGroup::Rank& r)
        {
            const std::pair<uint64, uint64> mine = hip::shard_range(n_total, r.rank, r.world);
            std::vector<uint32_t> h(size_t(mine.second - mine.first) * record_words);
            for (uint64 i = mine.first; i < mine.second; ++i)
                for (uint32_t w = 0; w < record_words; ++w) h[size_t(i - mine.first) * record_words + w] = w == 0 ? uint32_t(i) : w == 1 ? r.rank : uint32_t(i) ^ (0x5A5Au * w);
            void* stream = nullptr;
            hip_check(nvbio_hip_stream_create(&stream, 1u), "nvbio_hip_stream_create");
            hip::device_vector<uint32_t> send(h.size() ? h.size() : 1), recv(r.rank == 0 ? table.size() : 1);
            if (!h.empty()) hip_check(nvbio_hip_memcpy(send.data(), h.data(), h.size() * 4u, 1, stream), "h2d");
            r.gather_records(send.data(), counts, record_words * 4u, r.rank == 0 ? recv.data() : nullptr, 0u, stream);
            hip::synchronize(stream);
            if (r.rank == 0 && !table.empty()) hip_check(nvbio_hip_memcpy(table.data(), recv.data(), table.size() * 4u, 2, stream), "d2h");
            hip::synchronize(stream);
            nvbio_hip_stream_destroy(stream);
        });
        for (uint32_t k = 0; k < world; ++k)
        {
            const std::pair<uint64, uint64> blk = hip::shard_range(n_total, k, world);
            for (uint64 i = blk.first; i < blk.second; ++i)
                for (uint32_t w = 0; w < record_words; ++w)
                    if (table[size_t(i) * record_words + w] != (w == 0 ? uint32_t(i) : w == 1 ? k : uint32_t(i) ^ (0x5A5Au * w))) return 10 + int(k);
        }
        return 0;
    } catch (const std::exception& e) { fprintf(stderr, "nvbio_multi_device_selftest: %s\n", e.what()); return 1; }
}

// ---- the host layer's SAM writer (include/nvbio_hip/sam.h) for the Python tools: tools/align_fastq.py, tools/nvbowtie_3gbp.py ----
#include <nvbio_hip/sam.h>
extern "C" __attribute__((visibility("default")))
int nvbio_write_sam_se(const char* path, int append, int with_header, uint32_t extra_flags, uint32_t n, const char* names, const uint32_t* names_index,
                       const uint8_t* symbols, const uint64_t* read_index, const uint8_t* quals, const uint64_t* best, const uint8_t* mapq,
                       const uint16_t* cigar, uint32_t cigar_stride, const uint32_t* cigar_len, const uint32_t* source, const uint8_t* mds, uint32_t mds_stride,
                       uint32_t n_seqs, const char* const* seq_names, const uint64_t* seq_index)
{
    io::SamReference ref;
    for (uint32_t k = 0; k < n_seqs; ++k) ref.names.push_back(seq_names[k]);
    ref.index.assign(seq_index, seq_index + n_seqs + 1u);
    FILE* f = fopen(path, append ? "ab" : "wb");
    if (!f) return 1;
    bool ok = true;
    if (with_header) { const std::string h = ref.header(); ok = fwrite(h.data(), 1, h.size(), f) == h.size(); }
    const io::SamBatchSE b = { n, names, names_index, symbols, read_index, quals, best, mapq, cigar, cigar_stride, cigar_len, source, mds, mds_stride };
    ok = ok && io::write_sam_se(f, b, ref, extra_flags);
    return (fclose(f) == 0 && ok) ? 0 : 2;
}

// the paired-end twin: two slot sets of host arrays (anchor, opposite) -> two SAM records per pair
extern "C" __attribute__((visibility("default")))
int nvbio_write_sam_pe(const char* path, int append, int with_header, uint32_t n, uint32_t len, const char* names, const uint32_t* names_index,
                       const uint8_t* const* symbols /* 2 */, const uint8_t* const* quals /* 2 */,
                       const uint64_t* const* best /* 2 */, const uint8_t* const* mapq, const uint16_t* const* cigar, uint32_t cigar_stride, const uint32_t* const* cigar_len,
                       const uint32_t* const* source, const uint8_t* const* mds, uint32_t mds_stride,
                       uint32_t n_seqs, const char* const* seq_names, const uint64_t* seq_index)
{
    io::SamReference ref;
    for (uint32_t k = 0; k < n_seqs; ++k) ref.names.push_back(seq_names[k]);
    ref.index.assign(seq_index, seq_index + n_seqs + 1u);
    FILE* f = fopen(path, append ? "ab" : "wb");
    if (!f) return 1;
    bool ok = true;
    if (with_header) { const std::string h = ref.header(); ok = fwrite(h.data(), 1, h.size(), f) == h.size(); }
    io::SamBatchPE b;
    b.n = n; b.len = len; b.names = names; b.names_index = names_index; b.cigar_stride = cigar_stride; b.mds_stride = mds_stride;
    for (int k = 0; k < 2; ++k)
    {
        b.symbols[k] = symbols[k]; b.quals[k] = quals[k];
        b.slot[k].best = best[k]; b.slot[k].mapq = mapq[k]; b.slot[k].cigar = cigar[k]; b.slot[k].cigar_len = cigar_len[k]; b.slot[k].source = source[k]; b.slot[k].mds = mds[k];
    }
    ok = ok && io::write_sam_pe(f, b, ref);
    return (fclose(f) == 0 && ok) ? 0 : 2;
}
