// nvbio_hip/mapping.h -- host side of nvBowtie's seed mapping stage over libnvbio_hip.so.
// Mirrors nvBowtie/bowtie2/cuda/mapping.h (map / map_exact / map_approx / map_case_pruning over an input
// queue of reads), seed_hit.h (SeedHit) and the fields of params.h the stage reads.
#pragma once
#include <cmath>
#include <vector>
#include "fmindex.h"
#include "strings.h"

namespace nvbio {
namespace bowtie2 {
namespace cuda {

enum ReadType { STANDARD = 0u, COMPLEMENT = 1u };          // seed_hit.h:43-44
enum DirType  { FORWARD  = 0u, REVERSE    = 1u };

/// SeedHit (seed_hit.h:54-223): an SA range plus where the seed came from; the kernels write exactly
/// this bit layout (exclusive range: [begin, begin + delta))
struct SeedHit
{
    uint32 m_range_begin;
    uint32 m_range_delta:20, m_pos:10, m_rc:1, m_indexdir:1;
    uint2    get_range()      const { return make_uint2(m_range_begin, m_range_begin + m_range_delta); }
    uint32   get_range_size() const { return m_range_delta; }
    uint32   get_posinread()  const { return m_pos; }
    ReadType get_readtype()   const { return ReadType(m_rc); }
    DirType  get_indexdir()   const { return DirType(m_indexdir); }
};
static_assert(sizeof(SeedHit) == 8, "SeedHit must be two words");

/// The per-read hit sets.  The reference bump-allocates each read's deque in an arena with an atomic
/// (seed_hit_deque_array.h); here read r owns the fixed slot hits[r*stride .. r*stride + counts[r]).
struct SeedHitDequeArrayDeviceView
{
    SeedHit* hits; uint32 stride; uint32* counts;
};

/// input half of nvbio::cuda::PingPongQueuesView<uint32>: the reads to map (NULL queue = all reads 0..in_size)
struct PingPongQueuesView { uint32 in_size; const uint32* in_queue; };

/// SimpleFunc (func.h:39-70): single-precision k + m*f(x), truncated
struct SimpleFunc
{
    enum Type { LinearFunc = 0, LogFunc = 1, SqrtFunc = 2 };
    SimpleFunc(const Type _type = LinearFunc, const float _k = 0.0f, const float _m = 1.0f) : type(_type), k(_k), m(_m) {}
    int32 operator() (const int32 x) const
    { return int32(k + m * (type == LogFunc ? logf(float(x)) : type == SqrtFunc ? sqrtf(float(x)) : float(x))); }
    Type type; float k, m;
};

/// the fields of ParamsPOD the mapping stage reads (params.h:100-120), with nvBowtie's end-to-end defaults
struct ParamsPOD
{
    ParamsPOD() : seed_len(22), seed_freq(SimpleFunc::SqrtFunc, 1.0f, 1.15f), min_read_len(12), max_hits(100),
                  max_reseed(2), rep_seeds(300), allow_sub(0), subseed_len(0) {}
    uint32 seed_len; SimpleFunc seed_freq; uint32 min_read_len, max_hits, max_reseed, rep_seeds, allow_sub, subseed_len;

    /// params.seed_freq(L) for every read length up to max_read_len: uploaded once, read by the kernels
    std::vector<uint32> seed_freq_table(const uint32 max_read_len) const
    {
        std::vector<uint32> t(max_read_len + 1u, 0u);
        for (uint32 L = 1; L <= max_read_len; ++L) { const int32 f = seed_freq(int32(L)); t[L] = f > 0 ? uint32(f) : 0u; }
        return t;
    }
};

/// map(): one run of seed mapping for the reads in the input queue, with the algorithm choice of map_t
/// (mapping_inl.h:809-843).  d_seed_freq_table: device copy of params.seed_freq_table(max_read_len).
template <typename read_batch_type>
inline void map_with(
    const int32                     algorithm,
    const read_batch_type&          read_batch,
    const fm_index_device&          fmi,
    const fm_index_device&          rfmi,
    const uint32                    retry,
    const PingPongQueuesView        queues,
    uint8*                          reseed,
    SeedHitDequeArrayDeviceView     hits,
    const ParamsPOD                 params,
    const uint32*                   d_seed_freq_table,
    const bool                      fw,
    const bool                      rc,
    void*                           hip_stream = nullptr)
{
    const nvbio_hip_map_params p = { params.seed_len, params.min_read_len, params.max_hits, params.max_reseed, retry, params.rep_seeds, fw ? 1u : 0u, rc ? 1u : 0u };
    const nvbio_hip_string_set r = read_batch.abi();
    hip_check(nvbio_hip_map(algorithm, params.subseed_len, &fmi.m, &rfmi.m, &r, queues.in_queue, queues.in_size, &p, d_seed_freq_table,
                            reinterpret_cast<uint64*>(hits.hits), hits.stride, hits.counts, reseed, hip_stream), "nvbio_hip_map");
}

template <typename read_batch_type>
inline void map(
    const read_batch_type&          read_batch,
    const fm_index_device&          fmi,
    const fm_index_device&          rfmi,
    const uint32                    retry,
    const PingPongQueuesView        queues,
    uint8*                          reseed,
    SeedHitDequeArrayDeviceView     hits,
    const ParamsPOD                 params,
    const uint32*                   d_seed_freq_table,
    const bool                      fw,
    const bool                      rc,
    void*                           hip_stream = nullptr)
{
    const int32 algorithm = !params.allow_sub ? NVBIO_HIP_EXACT_MAPPING : (params.subseed_len == 0 ? NVBIO_HIP_CASE_PRUNING_MAPPING : NVBIO_HIP_APPROX_MAPPING);
    map_with(algorithm, read_batch, fmi, rfmi, retry, queues, reseed, hits, params, d_seed_freq_table, fw, rc, hip_stream);
}

/// the all-mapping driver's choice (aligner_all.h:177-212): map_exact, or -- with params.allow_sub -- map_approx with params.subseed_len
/// exact symbols (none by default) and one mismatch in the rest of the seed; never the case-pruning mapper of the best modes
template <typename read_batch_type>
inline void map_all(
    const read_batch_type&          read_batch,
    const fm_index_device&          fmi,
    const fm_index_device&          rfmi,
    const PingPongQueuesView        queues,
    uint8*                          reseed,
    SeedHitDequeArrayDeviceView     hits,
    const ParamsPOD                 params,
    const uint32*                   d_seed_freq_table,
    const bool                      fw,
    const bool                      rc,
    void*                           hip_stream = nullptr)
{
    map_with(params.allow_sub ? NVBIO_HIP_APPROX_MAPPING : NVBIO_HIP_EXACT_MAPPING, read_batch, fmi, rfmi, 0u, queues, reseed, hits, params, d_seed_freq_table, fw, rc, hip_stream);
}

} // namespace cuda
} // namespace bowtie2
} // namespace nvbio
