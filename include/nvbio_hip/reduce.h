// nvbio_hip/reduce.h -- host side of nvBowtie's score reduction and mapping quality stages over libnvbio_hip.so:
// io::Alignment / io::BestAlignments (nvbio/io/alignments.h:80-176), init_alignments (nvBowtie/bowtie2/cuda/
// aligner.h:323-366), score_reduce (reduce_inl.h:71-160), BowtieMapq2 / BowtieMapq3 (mapq.h:42-330).
#pragma once
#include <vector>
#include "types.h"
#include "mapping.h"

namespace nvbio {
namespace io {

/// io::Alignment: the reference's bit layout (alignments.h:130-131)
struct Alignment
{
    static uint32 max_ed()    { return 255u; }
    static int32  max_score() { return (1 << 17) - 1; }
    static int32  min_score() { return -((1 << 17) - 1); }
    Alignment() {}
    Alignment(const uint32 pos, const uint32 ed, const int32 score, const uint32 rc, const uint32 mate = 0, const bool paired = false, const bool discordant = false)
    {
        m_align = pos; m_ed = ed; m_score = score < 0 ? uint32(-score) : uint32(score); m_score_sgn = score < 0 ? 1u : 0u;
        m_rc = rc; m_mate = mate; m_paired = paired ? 1u : 0u; m_discordant = discordant ? 1u : 0u;
    }
    int32  score()      const { return m_score_sgn ? -int32(m_score) : int32(m_score); }
    bool   is_aligned() const { return m_align != uint32(-1); }
    uint32 alignment()  const { return m_align; }
    bool   is_rc()      const { return m_rc; }
    uint32 ed()         const { return m_ed; }
    static Alignment invalid() { return Alignment(uint32(-1), max_ed(), max_score(), 0u, 0u, false); }
    uint32 m_score_sgn : 1, m_score : 17, m_ed : 10, m_rc : 1, m_mate : 1, m_paired : 1, m_discordant : 1;
    uint32 m_align;
};
static_assert(sizeof(Alignment) == 8, "io::Alignment must be two words");

} // namespace io

namespace bowtie2 {
namespace cuda {

/// the fields of SmithWatermanScoringScheme the reduction / MAPQ stages read (scoring.h:272-281,347)
struct ScoreLimits
{
    ScoreLimits(const int32 _match = 0, const SimpleFunc _score_min = SimpleFunc(SimpleFunc::LinearFunc, -0.6f, -0.6f))
        : match(_match), score_min(_score_min), monotone(_match == 0) {}
    int32 perfect_score(const uint32 read_len) const { return int32(read_len) * match; }
    int32 min_score(const uint32 read_len) const { return score_min(int32(read_len)); }
    /// min_score(L) for L in [0, max_read_len]: uploaded once, read by the kernels
    std::vector<int32> min_score_table(const uint32 max_read_len) const
    { std::vector<int32> t(max_read_len + 1u, 0); for (uint32 L = 1; L <= max_read_len; ++L) t[L] = min_score(L); return t; }
    int32 match; SimpleFunc score_min; bool monotone;
};

/// init_alignments( reads, threshold_score, best_data, best_stride, mate )
inline void init_alignments(const uint32 n_reads, const uint32* d_read_len, const uint32 fixed_read_len, const int32* d_min_score_table,
                            io::Alignment* best_data, const uint32 best_stride, const uint32 mate = 0, void* hip_stream = nullptr)
{ hip_check(nvbio_hip_init_alignments(n_reads, d_read_len, fixed_read_len, d_min_score_table, mate, reinterpret_cast<uint64*>(best_data), best_stride, hip_stream), "nvbio_hip_init_alignments"); }

/// score_reduce: fold the extension results of the active reads into best_data (CSR over the hit arrays)
inline void score_reduce(const uint32 n_active, const uint32* d_read_ids, const uint64* d_hit_begin,
                         const int32* d_hit_score, const uint32* d_hit_loc, const uint8* d_hit_rc,
                         const uint32* d_read_len, const uint32 fixed_read_len,
                         io::Alignment* best_data, const uint32 best_stride, void* hip_stream = nullptr)
{ hip_check(nvbio_hip_score_reduce(n_active, d_read_ids, d_hit_begin, d_hit_score, d_hit_loc, d_hit_rc, d_read_len, fixed_read_len,
                                   reinterpret_cast<uint64*>(best_data), best_stride, hip_stream), "nvbio_hip_score_reduce"); }

/// BowtieMapq2 (version 2) / BowtieMapq3 (version 3) over all reads
inline void mapq(const uint32 version, const ScoreLimits& sc, const int32* d_min_score_table, const uint32 n_reads,
                 const io::Alignment* best_data, const uint32 best_stride, const uint32* d_read_len, const uint32 fixed_read_len,
                 uint8* d_mapq, void* hip_stream = nullptr)
{ hip_check(nvbio_hip_mapq(int32(version), sc.match, sc.monotone ? 1 : 0, d_min_score_table, n_reads, reinterpret_cast<const uint64*>(best_data), best_stride,
                           d_read_len, fixed_read_len, d_mapq, hip_stream), "nvbio_hip_mapq"); }

} // namespace cuda
} // namespace bowtie2
} // namespace nvbio
