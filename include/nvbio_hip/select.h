// nvbio_hip/select.h -- host side of nvBowtie's hit-selection stage and of the per-round stages of its best-approx
// extension loop over libnvbio_hip.so.  Mirrors nvBowtie/bowtie2/cuda/select.h (select_init, select,
// SelectBestApproxContext), scoring_queues.h (ScoringQueues: active reads + hits), defs.h (packed_read, packed_seed),
// locate.h (locate), score.h (score_best's stream set-up) and reduce.h (ReduceBestApproxContext, score_reduce).
#pragma once
#include "mapping.h"
#include "reduce.h"
#include <utility>

namespace nvbio {
namespace bowtie2 {
namespace cuda {

/// packed_read (defs.h:152-162) and packed_seed (defs.h:171-181): the kernels write exactly these bit layouts
struct packed_read
{
    packed_read() {}
    packed_read(const uint32 _read_id, const uint32 _top_flag = 1u) : read_id(_read_id), top_flag(_top_flag) {}
    uint32 read_id:31, top_flag:1;
};
struct packed_seed
{
    packed_seed() {}
    packed_seed(const uint32 _pos_in_read, const uint32 _index_dir, const uint32 _rc, const uint32 _top_flag) :
        pos_in_read(_pos_in_read), index_dir(_index_dir), rc(_rc), top_flag(_top_flag) {}
    uint32 pos_in_read:12, index_dir:1, rc:1, top_flag:1;
};
static_assert(sizeof(packed_read) == 4 && sizeof(packed_seed) == 4, "packed_read / packed_seed must be one word");

/// the fields of ParamsPOD the selection / reduction stages read (params.h:98-114), with nvBowtie's defaults
struct SelectParamsPOD
{
    SelectParamsPOD() : randomized(true), top_seed(0), max_effort_init(15), max_effort(15), min_ext(30), max_ext(400) {}
    bool randomized; uint32 top_seed, max_effort_init, max_effort, min_ext, max_ext;
};

/// The device-side state the selection stage adds to the hit deques: the pipeline's trys / rseeds and the deques'
/// probability trees (SeedHitDequeArray::m_probs).  Read r's tree is probs[r * probs_stride ..): its LEAVES only -- the selection kernels
/// rebuild the sums above them on chip (nvbio_amd/csrc/select.hip), so a 16-slot deque's row is 64 bytes instead of two 128-byte lines.
struct SelectState
{
    SelectState(const uint32 n_reads, const uint32 hits_stride) :
        probs_stride(row_pitch(hits_stride)), probs(size_t(n_reads) * row_pitch(hits_stride)),
        trys(n_reads), rseeds(n_reads) {}
    /// floats per row: the hit slots rounded to 16-byte pieces (vector loads); rows wider than 32 slots keep room for the sums too, as scratch
    static uint32 row_pitch(const uint32 hits_stride) { return hits_stride <= 32u ? ((hits_stride + 3u) & ~3u) : ((nvbio_hip_sum_tree_node_count(hits_stride) + 31u) & ~31u); }
    uint32 probs_stride;
    hip::device_vector<float>  probs;
    hip::device_vector<uint32> trys, rseeds;
};

/// HitQueues + the active-read ping-pong queue of ScoringQueues (scoring_queues.h:133-260), as plain arrays: a round's
/// selected hits grouped per active read by hit_begin (what ReadHitsIndex's links express).
struct ScoringQueues
{
    ScoringQueues(const uint32 max_reads, const uint32 max_hits) :
        active_in(max_reads), active_out(max_reads), hit_begin(size_t(max_reads) + 1u), hit_read_id(max_hits), hit_loc(max_hits), hit_seed(max_hits),
        sizes(2), host_sizes(2), in_size(0), hits_size(0) {}
    hip::device_vector<packed_read> active_in, active_out;
    hip::device_vector<uint64>      hit_begin;
    hip::device_vector<uint32>      hit_read_id, hit_loc;
    hip::device_vector<packed_seed> hit_seed;
    hip::device_vector<uint32>      sizes;
    hip::pinned_words               host_sizes;       ///< where select() has the device leave a round's (in_size, hits_size) for the waiting host thread
    uint32 in_size, hits_size;
    void swap() { std::swap(active_in.m_ptr, active_out.m_ptr); std::swap(active_in.m_size, active_out.m_size); }
};

/// select_init( pipeline, params ) (select.h:82-100): read names as SequenceData keeps them (name_stream / name_index)
inline void select_init(const uint32 count, const char* d_read_names, const uint32* d_read_names_idx, const SeedHitDequeArrayDeviceView hits,
                        SelectState& state, const SelectParamsPOD params, void* hip_stream = nullptr)
{
    hip_check(nvbio_hip_select_init(count, d_read_names, d_read_names_idx, reinterpret_cast<const uint64*>(hits.hits), hits.stride, hits.counts,
                                    state.probs.data(), state.probs_stride, state.trys.data(), state.rseeds.data(), params.max_effort_init,
                                    params.randomized ? 1 : 0, int32(params.top_seed), hip_stream), "nvbio_hip_select_init");
}
/// the same for the reads of a seeding pass's queue only (a re-seeding pass: a few percent of the batch)
inline void select_init(const uint32 n_queue, const uint32* d_queue, const char* d_read_names, const uint32* d_read_names_idx, const SeedHitDequeArrayDeviceView hits,
                        SelectState& state, const SelectParamsPOD params, void* hip_stream = nullptr)
{
    hip_check(nvbio_hip_select_init_queued(n_queue, d_queue, d_read_names, d_read_names_idx, reinterpret_cast<const uint64*>(hits.hits), hits.stride, hits.counts,
                                           state.probs.data(), state.probs_stride, state.trys.data(), state.rseeds.data(), params.max_effort_init,
                                           params.randomized ? 1 : 0, int32(params.top_seed), hip_stream), "nvbio_hip_select_init_queued");
}

/// select( context, pipeline, params ) (select.h:139-152): one round over queues.active_in[0..in_size); on return the output
/// queue has been swapped in (in_size = surviving reads, hits_size = selected hits) -- the driver's active_read_queues.swap()
inline void select(SeedHitDequeArrayDeviceView hits, SelectState& state, ScoringQueues& queues, const uint32 n_hits_per_read,
                   const SelectParamsPOD params, void* hip_stream = nullptr)
{
    const uint64 temp_bytes = nvbio_hip_select_temp_bytes(queues.in_size, n_hits_per_read);
    hip::device_vector<uint8> temp(temp_bytes);
    // The two sizes decide what the host queues next.  The kernels leave them in pinned host memory and the host thread reads them as they
    // land (hip::pinned_words): a round then costs no stream synchronisation and no copy, only the wait for the round's last kernel.  What
    // the host queues next is ordered behind that kernel on the same stream, so nothing is read early.  NVBIO_HIP_POLL_SIZES=0 (read once)
    // keeps the synchronise-and-copy form.
    static const bool poll = [] { const char* e = getenv("NVBIO_HIP_POLL_SIZES"); return !(e && atoi(e) == 0); }();
    const bool polled = poll && queues.host_sizes.ptr != nullptr;
    if (polled) queues.host_sizes.arm();
    hip_check(nvbio_hip_select(params.randomized ? 1 : 0, n_hits_per_read, reinterpret_cast<const uint32*>(queues.active_in.data()), queues.in_size,
                               reinterpret_cast<uint64*>(hits.hits), hits.stride, hits.counts, state.probs.data(), state.probs_stride,
                               state.rseeds.data(), state.trys.data(), reinterpret_cast<uint32*>(queues.active_out.data()), queues.hit_begin.data(),
                               queues.hit_read_id.data(), queues.hit_loc.data(), reinterpret_cast<uint32*>(queues.hit_seed.data()),
                               polled ? const_cast<uint32*>(queues.host_sizes.ptr) : queues.sizes.data(), temp.data(), temp_bytes, hip_stream), "nvbio_hip_select");
    uint32 s[2];
    if (polled) { queues.host_sizes.wait_words(2u, hip_stream); s[0] = queues.host_sizes.ptr[0]; s[1] = queues.host_sizes.ptr[1]; }
    else        { const std::vector<uint32> h = queues.sizes.to_host(hip_stream); s[0] = h[0]; s[1] = h[1]; }
    queues.swap();
    queues.in_size = s[0]; queues.hits_size = s[1];
}

/// locate( pipeline, params ) (locate.h): hit.loc from SA coordinates to the read's start in the genome
inline void locate(const fm_index_device& fmi, const fm_index_device& rfmi, ScoringQueues& queues, void* hip_stream = nullptr)
{
    hip_check(nvbio_hip_locate_hits(&fmi.m, &rfmi.m, queues.hits_size, queues.hit_loc.data(), reinterpret_cast<const uint32*>(queues.hit_seed.data()), hip_stream),
              "nvbio_hip_locate_hits");
}

/// BestScoreStream's per-hit set-up (score_best_inl.h:95-126): pattern offsets, genome windows and score thresholds of a
/// round's hits, ready for BatchedBandedAlignmentScore over (patterns at pattern_begin, texts at text_begin / text_len)
inline void score_best_setup(const ScoringQueues& queues, const uint64* d_read_begin, const uint32* d_read_len, const uint32 fixed_read_len,
                             const uint64 rc_offset, const uint32 band_len, const uint32 genome_length,
                             const io::Alignment* best_data, const uint32 best_stride, const int32 score_limit,
                             uint64* pattern_begin, uint32* pattern_len, uint64* text_begin, uint32* text_len, int32* min_score, int32* known_score = nullptr,
                             uint32* job_count = nullptr, uint32* job_hit = nullptr, void* hip_stream = nullptr)
{
    hip_check(nvbio_hip_score_best_setup(queues.hits_size, queues.hit_read_id.data(), queues.hit_loc.data(), reinterpret_cast<const uint32*>(queues.hit_seed.data()),
                                         d_read_begin, d_read_len, fixed_read_len, rc_offset, band_len, genome_length,
                                         reinterpret_cast<const uint64*>(best_data), best_stride, score_limit, pattern_begin, pattern_len,
                                         text_begin, text_len, min_score, known_score, job_count, job_hit, hip_stream), "nvbio_hip_score_best_setup");
}

/// ReduceBestApproxContext (reduce.h:63-105)
struct ReduceBestApproxContext
{
    ReduceBestApproxContext(uint32* trys, const uint32 n_ext) : m_trys(trys), m_ext(n_ext) {}
    uint32* m_trys; uint32 m_ext;
};

/// score_reduce( context, pipeline, params ) (reduce.h:136-147) over the round's hits; hit_score = the raw DP scores
inline void score_reduce(const ReduceBestApproxContext context, SeedHitDequeArrayDeviceView hits, const ScoringQueues& queues, const int32* d_hit_score,
                         const uint32* d_read_len, const uint32 fixed_read_len, io::Alignment* best_data, const uint32 best_stride,
                         const int32 worst_score, const SelectParamsPOD params, const int32* d_known_score = nullptr, void* hip_stream = nullptr,
                         const uint32* d_hit_sink = nullptr, uint32* d_best_sink = nullptr)
{
    hip_check(nvbio_hip_score_reduce_best_approx(queues.in_size, reinterpret_cast<const uint32*>(queues.active_in.data()), queues.hit_begin.data(), d_hit_score,
                                                 queues.hit_loc.data(), reinterpret_cast<const uint32*>(queues.hit_seed.data()), d_read_len, fixed_read_len,
                                                 reinterpret_cast<uint64*>(best_data), best_stride, worst_score, context.m_trys, hits.counts,
                                                 context.m_ext, params.min_ext, params.max_ext, params.max_effort, d_known_score, d_hit_sink, d_best_sink, hip_stream),
              "nvbio_hip_score_reduce_best_approx");
}

} // namespace cuda
} // namespace bowtie2
} // namespace nvbio
