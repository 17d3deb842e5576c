#!/usr/bin/env python3
"""Repeatable "binds unchanged" check of the drop-in template layer (VERDICT r2, item 1c / missing item 4).

Runs ONLY in the build container (needs /root/reference; the GPU box has neither the reference nor a use for this).  For each case it
reads a line range of a reference source file IN PLACE, wraps the verbatim text in a temporary translation unit (under a temp
directory that is deleted afterwards -- the reference text is never copied into the repository and never travels), compiles the TU
with `hipcc --offload-arch=gfx950 -I include/nvbio_hip/compat` and lets static_asserts in the wrapper state which execution the
drop-in layer picks for the reference's own class.  Around the verbatim text the wrapper provides only what the APPLICATION side of
that file would (the file's own enums / typedefs outside the range, nvBowtie's pipeline / hit-queue types) -- never a library type.

Writes a log (reference file:line, sha256 of the extracted text, compiler, verdict, first error lines) to profiles/r03/ref_bind_check.log
and exits non-zero when a case does not compile.

    python tools/ref_bind_check.py [--keep] [--log PATH]
"""
import argparse
import hashlib
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
COMPAT = os.path.join(ROOT, "include", "nvbio_hip", "compat")


def ref_lines(rel, first, last):
    with open(os.path.join(REF, rel), "r", errors="replace") as f:
        lines = f.readlines()
    return "".join(lines[first - 1:last])


# ------------------------------------------------------------------------------------------------------------------------------
# the cases: (name, [(file, first, last), ...], wrapper with {0}, {1}, ... standing for the verbatim ranges)
# ------------------------------------------------------------------------------------------------------------------------------
CASES = []

CASES.append(("sw-benchmark AlignmentStream -> tuned score + tuned traceback-free enact", [("sw-benchmark/sw-benchmark.cu", 66, 218)], r"""
#include <nvbio/basic/cuda/ldg.h>
#include <nvbio/basic/packedstream.h>
#include <nvbio/basic/packedstream_loader.h>
#include <nvbio/basic/vector_view.h>
#include <nvbio/io/sequence/sequence.h>
#include <nvbio/basic/dna.h>
#include <nvbio/alignment/alignment.h>
#include <nvbio/alignment/batched.h>
#include <nvbio/alignment/sink.h>
enum { MAX_READ_LENGTH = 1024 };
{0}
// sw-benchmark.cu:604-657 instantiates these aligners over the stream
typedef aln::GotohAligner<aln::LOCAL, aln::SimpleGotohScheme, aln::TextBlockingTag>        gotoh_local;
typedef aln::GotohAligner<aln::SEMI_GLOBAL, aln::SimpleGotohScheme, aln::TextBlockingTag>  gotoh_semi;
typedef aln::EditDistanceAligner<aln::SEMI_GLOBAL, aln::TextBlockingTag>                   ed_semi;
static_assert(aln::priv::recognised< AlignmentStream<gotoh_local> >::zero_copy, "sw-benchmark's stream must run on the tuned kernels in place");
static_assert(aln::priv::recognised< AlignmentStream<gotoh_semi> >::zero_copy, "");
static_assert(aln::priv::recognised< AlignmentStream<ed_semi> >::zero_copy, "");
void instantiate(const uint32* p, const uint32* t, int16* s)
{
    aln::SimpleGotohScheme scoring; scoring.m_match = 2; scoring.m_mismatch = -1; scoring.m_gap_open = -2; scoring.m_gap_ext = -1;
    { typedef AlignmentStream<gotoh_local> stream_type; aln::BatchedAlignmentScore<stream_type, aln::DeviceThreadScheduler> b;
      b.enact(stream_type(gotoh_local(scoring), 0u, p, p, 100u, 0u, t, 150u, s)); }
    { typedef AlignmentStream<gotoh_semi> stream_type; aln::BatchedBandedAlignmentScore<15u, stream_type, aln::DeviceThreadScheduler> b;
      b.enact(stream_type(gotoh_semi(scoring), 0u, p, p, 100u, 0u, t, 150u, s)); }
    { typedef AlignmentStream<ed_semi> stream_type; aln::BatchedAlignmentScore<stream_type, aln::DeviceThreadScheduler> b;
      b.enact(stream_type(ed_semi(), 0u, p, p, 100u, 0u, t, 150u, s)); }
}
"""))

CASES.append(("nvbio-test fmindex_test locate_kernel over 32- and 64-bit fm_index", [("nvbio-test/fmindex_test.cu", 59, 92)], r"""
#include <nvbio/basic/dna.h>
#include <nvbio/basic/cached_iterator.h>
#include <nvbio/basic/packedstream.h>
#include <nvbio/basic/deinterleaved_iterator.h>
#include <nvbio/fmindex/bwt.h>
#include <nvbio/fmindex/ssa.h>
#include <nvbio/fmindex/fmindex.h>
using namespace nvbio;
{0}
}} // anonymous namespace
template <typename index_type> struct layout
{{
    typedef PackedStream<const index_type*, uint8, 2, true, index_type>           bwt_type;
    typedef rank_dictionary<2, 64, bwt_type, const index_type*, const uint32*>    rank_dict_type;
    typedef fm_index<rank_dict_type, SSA_index_multiple_context<16, const index_type*> > fm_index_type;
}};
void instantiate()
{{
    hipLaunchKernelGGL((locate_kernel<64, layout<uint32>::fm_index_type, uint32>), dim3(1), dim3(1), 0, 0, 0u, 0u, 0u, (const uint32*)0, layout<uint32>::fm_index_type(), (const uint32*)0, (uint32*)0);
    hipLaunchKernelGGL((locate_kernel<64, layout<uint64>::fm_index_type, uint64>), dim3(1), dim3(1), 0, 0, 0u, 0u, 0u, (const uint64*)0, layout<uint64>::fm_index_type(), (const uint32*)0, (uint32*)0);
}}
"""))

CASES.append(("nvbio-test alignment_test AlignmentStream (4-bit / 2-bit little-endian, M x N) -> tuned", [("nvbio-test/alignment_test.cu", 52, 173)], r"""
#include <nvbio/basic/cuda/ldg.h>
#include <nvbio/basic/packedstream.h>
#include <nvbio/basic/packedstream_loader.h>
#include <nvbio/basic/vector_view.h>
#include <nvbio/basic/dna.h>
#include <nvbio/alignment/alignment.h>
#include <nvbio/alignment/batched.h>
#include <nvbio/alignment/sink.h>
using namespace nvbio;
{0}
typedef GotohAligner<LOCAL, SimpleGotohScheme>  local_gotoh;
typedef EditDistanceAligner<SEMI_GLOBAL>        semi_ed;
typedef SmithWatermanAligner<GLOBAL, SimpleSmithWatermanScheme> global_sw;
static_assert(priv::recognised< AlignmentStream<local_gotoh, 150, 500> >::zero_copy, "alignment_test's stream must run on the tuned kernels in place");
static_assert(priv::recognised< AlignmentStream<semi_ed, 150, 500, uncached_tag_type> >::zero_copy, "");
void instantiate(const uint32* p, const uint32* t, int16* s)
{{
    {{ typedef AlignmentStream<local_gotoh, 150, 500> st; BatchedAlignmentScore<st, DeviceThreadScheduler> b; b.enact(st(local_gotoh(SimpleGotohScheme(2, -1, -5, -3)), 0, p, t, s)); }}
    {{ typedef AlignmentStream<semi_ed, 150, 500> st; BatchedAlignmentScore<st, DeviceStagedThreadScheduler> b; b.enact(st(semi_ed(), 0, p, t, s)); }}
    {{ typedef AlignmentStream<global_sw, 150, 181> st; BatchedBandedAlignmentScore<31u, st, DeviceThreadScheduler> b; b.enact(st(global_sw(SimpleSmithWatermanScheme(2, -1, -1, -1)), 0, p, t, s)); }}
    {{ typedef AlignmentStream<local_gotoh, 150, 181> st; BatchedBandedAlignmentScore<15u, st, HostThreadScheduler> b; b.enact(st(local_gotoh(SimpleGotohScheme(2, -1, -5, -3)), 0, p, t, s)); }}
}}
}} // namespace aln
}} // namespace nvbio
"""))

# nvBowtie's own stream machinery: the strings container + stream base (alignment_utils.h), the single-end score stream (score_best_inl.h) and the
# scheme (scoring.h) -- verbatim; the wrapper supplies nvBowtie's pipeline / hit-queue / params types (application types, not library types)
CASES.append(("nvBowtie AlignmentStrings + AlignmentStreamBase + BestScoreStream + SmithWatermanScoringScheme -> tuned, on the read views in place",
              [("nvBowtie/bowtie2/cuda/scoring.h", 53, 125), ("nvBowtie/bowtie2/cuda/scoring.h", 196, 356),
               ("nvBowtie/bowtie2/cuda/alignment_utils.h", 114, 345), ("nvBowtie/bowtie2/cuda/score_best_inl.h", 48, 148),
               ("nvBowtie/bowtie2/cuda/func.h", 39, 70)], r"""
#include <nvbio/basic/types.h>
#include <nvbio/basic/cuda/ldg.h>
#include <nvbio/basic/packedstream.h>
#include <nvbio/basic/packedstream_loader.h>
#include <nvbio/basic/vector_view.h>
#include <nvbio/io/utils.h>
#include <nvbio/io/alignments.h>
#include <nvbio/io/sequence/sequence.h>
#include <nvbio/alignment/alignment.h>
#include <nvbio/alignment/batched.h>
#include <map>
#include <string>
#define NVBIO_CUDA_DEBUG_PRINT_IF(...)
#define DP_REPORT_MULTIPLE 0
namespace nvbio {{ namespace bowtie2 {{ namespace cuda {{
// ---- nvBowtie/bowtie2/cuda/func.h, SimpleFunc (verbatim)
{4}
// ---- application-side types of nvBowtie the verbatim ranges refer to (params.h, pipeline_states.h, scoring_queues.h)
struct ParamsPOD {{ struct Debug {{ NVBIO_HOST_DEVICE bool show_score_info(uint32) const {{ return false; }} NVBIO_HOST_DEVICE bool show_score(uint32, bool) const {{ return false; }} }} debug; }};
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE float phred_to_maq(const int q) {{ return float(q < 40 ? q : 40) / 10.0f; }}
struct SeedHitLike {{ uint32 rc; }};
struct HitQueuesDeviceView;
template <typename Q> struct HitReference {{ uint32 read_id; SeedHitLike seed; uint32 loc; int32 score; uint32 sink; }};
struct HitArray {{ NVBIO_HOST_DEVICE HitReference<HitQueuesDeviceView>& operator[](const uint32 i) const {{ return data[i]; }} HitReference<HitQueuesDeviceView>* data; }};
struct ScoringQueuesView {{ HitArray hits; }};
struct ReadBatch
{{
    static const uint32 SEQUENCE_BITS = 4; static const bool SEQUENCE_BIG_ENDIAN = true;
    typedef nvbio::cuda::ldg_pointer<uint32> sequence_storage_iterator; typedef nvbio::cuda::ldg_pointer<uint8> qual_storage_iterator;
    typedef PackedStream<sequence_storage_iterator, uint8, 4, true> sequence_stream_type;
    NVBIO_HOST_DEVICE sequence_stream_type sequence_stream() const {{ return sequence_stream_type(sequence_storage_iterator(words)); }}
    NVBIO_HOST_DEVICE qual_storage_iterator qual_stream() const {{ return qual_storage_iterator(quals); }}
    NVBIO_HOST_DEVICE uint2 get_range(const uint32 i) const {{ return make_uint2(index[i], index[i + 1]); }}
    NVBIO_HOST_DEVICE uint32 max_read_len() const {{ return 100u; }}
    const uint32* words; const uint8* quals; const uint32* index;
}};
template <typename scheme_t> struct PipelineLike
{{
    typedef scheme_t scheme_type; typedef ReadBatch read_batch_type;
    typedef PackedStream<nvbio::cuda::ldg_pointer<uint32>, uint8, 2, true> genome_iterator;
    read_batch_type reads, reads_o; genome_iterator genome; uint32 genome_length;
    const uint32* idx_queue; ScoringQueuesView scoring_queues; uint32 hits_queue_size;
    const io::Alignment* best_alignments; uint32 best_stride; int32 score_limit; uint8* dp_buffer; uint64 dp_buffer_size;
}};
// ---- nvBowtie/bowtie2/cuda/scoring.h, cost functions and SmithWatermanScoringScheme (verbatim)
{0}
{1}
// ---- nvBowtie/bowtie2/cuda/alignment_utils.h (verbatim; opens namespace detail)
namespace detail {{
{2}
// ---- nvBowtie/bowtie2/cuda/score_best_inl.h, BestScoreStream (verbatim)
{3}
}} // namespace detail
typedef SmithWatermanScoringScheme<>                         scheme_type;
typedef PipelineLike<scheme_type>                            pipeline_type;
typedef scheme_type::local_aligner_type                      local_aligner;
typedef detail::BestScoreStream<local_aligner, pipeline_type> stream_type;
static_assert(aln::priv::quality_scheme<scheme_type>::value, "nvBowtie's scheme is recognised as a quality scheme");
static_assert(aln::priv::recognised<stream_type>::staged && aln::priv::recognised<stream_type>::stage_quals, "nvBowtie's BestScoreStream must run on the tuned kernels");
static_assert(aln::priv::recognised<stream_type>::view, "... and in place: its patterns are io::ReadStream views of packed reads with byte-pointer qualities");
void instantiate(const pipeline_type& pipeline, const scheme_type& scheme, const ParamsPOD params)
{{
    aln::BatchedBandedAlignmentScore<15u, stream_type, aln::DeviceThreadScheduler> batch;
    batch.enact(stream_type(15u, pipeline, scheme.local_aligner(), params), pipeline.dp_buffer_size, pipeline.dp_buffer);
}}
}} }} }} // namespaces
"""))

CASES.append(("nvBowtie BestTracebackStream + Backtracker (CIGAR-forming, 1024-entry context) -> tuned (staged) banded and full-matrix tracebacks",
              [("nvBowtie/bowtie2/cuda/scoring.h", 53, 125), ("nvBowtie/bowtie2/cuda/scoring.h", 196, 356),
               ("nvBowtie/bowtie2/cuda/alignment_utils.h", 114, 345), ("nvBowtie/bowtie2/cuda/traceback_inl.h", 46, 199),
               ("nvBowtie/bowtie2/cuda/func.h", 39, 70)], r"""
#include <nvbio/basic/types.h>
#include <nvbio/basic/cuda/ldg.h>
#include <nvbio/basic/packedstream.h>
#include <nvbio/basic/packedstream_loader.h>
#include <nvbio/basic/vector_view.h>
#include <nvbio/io/utils.h>
#include <nvbio/io/alignments.h>
#include <nvbio/io/sequence/sequence.h>
#include <nvbio/alignment/alignment.h>
#include <nvbio/alignment/batched.h>
#include <map>
#include <string>
#define NVBIO_CUDA_DEBUG_PRINT_IF(...)
#define NVBIO_CUDA_DEBUG_CHECK_IF(...)
#define NVBIO_CUDA_ASSERT_IF(...)
#define DP_REPORT_MULTIPLE 0
#define MAXIMUM_READ_LENGTH 512
#define MAXIMUM_INSERT_LENGTH 1024
namespace nvbio {{ namespace bowtie2 {{ namespace cuda {{
{4}
// ---- application-side types of nvBowtie the verbatim ranges refer to (params.h, defs.h, pipeline_states.h, the CIGAR arena)
enum MateType {{ AnchorMate = 0, OppositeMate = 1 }};
struct ParamsPOD {{ struct Debug {{ bool asserts; NVBIO_HOST_DEVICE bool show_traceback(uint32) const {{ return false; }} }} debug; }};
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE float phred_to_maq(const int q) {{ return float(q < 40 ? q : 40) / 10.0f; }}
struct ReadBatch
{{
    static const uint32 SEQUENCE_BITS = 4; static const bool SEQUENCE_BIG_ENDIAN = true;
    typedef nvbio::cuda::ldg_pointer<uint32> sequence_storage_iterator; typedef nvbio::cuda::ldg_pointer<uint8> qual_storage_iterator;
    typedef PackedStream<sequence_storage_iterator, uint8, 4, true> sequence_stream_type;
    NVBIO_HOST_DEVICE sequence_stream_type sequence_stream() const {{ return sequence_stream_type(sequence_storage_iterator(words)); }}
    NVBIO_HOST_DEVICE qual_storage_iterator qual_stream() const {{ return qual_storage_iterator(quals); }}
    NVBIO_HOST_DEVICE uint2 get_range(const uint32 i) const {{ return make_uint2(index[i], index[i + 1]); }}
    const uint32* words; const uint8* quals; const uint32* index;
}};
struct CigarArena {{ NVBIO_HOST_DEVICE io::Cigar* alloc(const uint32 read_id, const uint32) const {{ return data + read_id * 64u; }} io::Cigar* data; }};
template <typename scheme_t> struct PipelineLike
{{
    typedef scheme_t scheme_type; typedef ReadBatch read_batch_type;
    typedef PackedStream<nvbio::cuda::ldg_pointer<uint32>, uint8, 2, true> genome_iterator;
    NVBIO_HOST_DEVICE read_batch_type get_reads(const uint32 mate) const {{ return mate ? reads_o : reads; }}
    read_batch_type reads, reads_o; genome_iterator genome; uint32 genome_length;
    CigarArena cigar; uint2* cigar_coords; uint8* dp_buffer; uint64 dp_buffer_size;
}};
{0}
{1}
namespace detail {{
{2}
{3}
}} // namespace detail
typedef SmithWatermanScoringScheme<>                         scheme_type;
typedef PipelineLike<scheme_type>                            pipeline_type;
typedef scheme_type::local_aligner_type                      local_aligner;
typedef detail::BestTracebackStream<0u, local_aligner, pipeline_type> stream_type;
static_assert(aln::priv::recognised_tb<stream_type>::staged && aln::priv::recognised_tb<stream_type>::stage_quals, "nvBowtie's BestTracebackStream must run on the tuned kernels");
void instantiate(const pipeline_type& pipeline, const scheme_type& scheme, const ParamsPOD params, io::Alignment* best)
{{
    const stream_type stream(AnchorMate, 0u, NULL, best, 0u, 15u, pipeline, scheme.local_aligner(), params);
    {{ aln::BatchedBandedAlignmentTraceback<15u, 64u, stream_type> batch; batch.enact(stream, pipeline.dp_buffer_size, pipeline.dp_buffer); }}      // traceback_inl.h:239-251
    {{ aln::BatchedAlignmentTraceback<1024u, stream_type> batch; batch.enact(stream, pipeline.dp_buffer_size, pipeline.dp_buffer); }}              // traceback_inl.h:880-905
}}
}} }} }} // namespaces
"""))

# ------------------------------------------------------------------------------------------------------------------------------
# round 4: every caller SURVEY 8(b) names.  nvBowtie's seed mappers and locate wrappers, the remaining score streams, fmmap's
# pipeline, and whole TUs of nvbio-test.
# ------------------------------------------------------------------------------------------------------------------------------
NVBOWTIE_APP_PRELUDE = r"""
#include <nvbio/basic/types.h>
#include <nvbio/basic/numbers.h>
#include <nvbio/basic/cuda/ldg.h>
#include <nvbio/basic/cuda/pingpong_queues.h>
#include <nvbio/basic/packedstream.h>
#include <nvbio/basic/packedstream_loader.h>
#include <nvbio/basic/priority_deque.h>
#include <nvbio/basic/transform_iterator.h>
#include <nvbio/basic/index_transform_iterator.h>
#include <nvbio/basic/deinterleaved_iterator.h>
#include <nvbio/basic/vector_view.h>
#include <nvbio/io/utils.h>
#include <nvbio/fmindex/fmindex.h>
#define USE_REVERSE_INDEX 0
"""

CASES.append(("nvBowtie mapping_inl.h: check_N, match_range, store_deque, 1-mismatch map<> (range rank4 + comp), seed_mapper<EXACT|APPROX|CASE_PRUNING>, map_whole_read_kernel -- over the production uint4 index",
              [("nvBowtie/bowtie2/cuda/seed_hit.h", 47, 245), ("nvBowtie/bowtie2/cuda/utils.h", 51, 53), ("nvBowtie/bowtie2/cuda/mapping_inl.h", 60, 505)],
              NVBOWTIE_APP_PRELUDE + r"""
namespace nvbio {{ namespace bowtie2 {{ namespace cuda {{
enum {{ BLOCKDIM = 96 }};                                                                       // nvBowtie defs.h:89
// ---- nvBowtie/bowtie2/cuda/seed_hit.h, SeedHit + hit_compare (verbatim)
{0}
// ---- nvBowtie/bowtie2/cuda/utils.h, inclusive_to_exclusive (verbatim)
{1}
// ---- application-side types of nvBowtie the verbatim range refers to (params.h, seed_hit_deque_array.h's device view)
struct ParamsPOD {{ uint32 max_hits, subseed_len, min_read_len; }};
struct SeedHitDequeArrayDeviceView
{{
    typedef vector_view<SeedHit*>                                  hit_vector_type;
    typedef priority_deque<SeedHit, hit_vector_type, hit_compare>  hit_deque_type;
    NVBIO_DEVICE SeedHit* alloc_deque(const uint32 read_id, const uint32 size) {{ return size ? hits + atomicAdd(pool, size) : NULL; }}
    NVBIO_DEVICE void     resize_deque(const uint32 read_id, const uint32 size) {{ counts[read_id] = size; }}
    SeedHit* hits; uint32* counts; uint32* pool;
}};
struct ReadBatch
{{
    static const uint32 SEQUENCE_BITS = 4; static const bool SEQUENCE_BIG_ENDIAN = true;
    typedef nvbio::cuda::ldg_pointer<uint32> sequence_storage_iterator;
    typedef PackedStream<sequence_storage_iterator, uint8, 4, true> sequence_stream_type;
    NVBIO_HOST_DEVICE sequence_stream_type sequence_stream() const {{ return sequence_stream_type(sequence_storage_iterator(words)); }}
    NVBIO_HOST_DEVICE uint2 get_range(const uint32 i) const {{ return make_uint2(index[i], index[i + 1]); }}
    NVBIO_HOST_DEVICE uint32 size() const {{ return n; }}
    const uint32* words; const uint32* index; uint32 n;
}};
namespace detail {{
// ---- nvBowtie/bowtie2/cuda/mapping_inl.h:60-505 (verbatim; closes its doxygen groups only)
{2}
}} // namespace detail

// nvBowtie's FM-index type over the interleaved uint4 layout (nvbio/io/fmindex/fmindex.h:159-174)
typedef nvbio::cuda::ldg_pointer<uint4>                                     bwt_occ_iterator;
typedef deinterleaved_iterator<2, 0, bwt_occ_iterator>                      bwt_iterator;
typedef deinterleaved_iterator<2, 1, bwt_occ_iterator>                      occ_iterator;
typedef PackedStream<bwt_iterator, uint8, 2, true>                          bwt_type;
typedef rank_dictionary<2, 64, bwt_type, occ_iterator, nvbio::cuda::ldg_pointer<uint32> > rank_dict_type;
typedef fm_index<rank_dict_type, SSA_index_multiple_context<16, const uint32*>, nvbio::cuda::ldg_pointer<uint32> > fm_index_type;
static_assert(nvbio::priv::block64<2u, 64u, bwt_type, occ_iterator>::value, "the production layout takes the one-record-per-end rank path");
void instantiate(const ReadBatch reads, const fm_index_type fmi, const nvbio::cuda::PingPongQueuesView<uint32> queues, SeedHitDequeArrayDeviceView hits, const ParamsPOD params)
{{
    hipLaunchKernelGGL((detail::map_whole_read_kernel<ReadBatch, fm_index_type, fm_index_type>), dim3(1), dim3(BLOCKDIM), 0, 0, reads, fmi, fmi, queues, (uint8*)0, hits, params, true, true);
}}
// the three seed mappers, each through a kernel of the shape of map_queues_kernel (mapping_inl.h:507-600)
template <detail::MappingAlgorithm ALG>
__global__ void mapper_kernel(const ReadBatch reads, const fm_index_type fmi, SeedHitDequeArrayDeviceView hits, const ParamsPOD params)
{{
    typedef PackedStringLoader<ReadBatch::sequence_storage_iterator, 4, true, uncached_tag> loader_type;
    loader_type loader;
    SeedHit local_hits[512];
    SeedHitDequeArrayDeviceView::hit_deque_type hitheap(SeedHitDequeArrayDeviceView::hit_vector_type(0, local_hits), true);
    const uint2 range = reads.get_range(threadIdx.x);
    const loader_type::iterator seed = loader.load(reads.sequence_stream() + range.x, 22u);
    uint32 range_sum = 0, range_count = 0;
    detail::seed_mapper<ALG>::enact(fmi, fmi, seed, range, range.x, 22u, hitheap, range_sum, range_count, params, true, true);
    detail::store_deque(hits, threadIdx.x, hitheap.size(), local_hits);
}}
void instantiate_mappers(const ReadBatch reads, const fm_index_type fmi, SeedHitDequeArrayDeviceView hits, const ParamsPOD params)
{{
    hipLaunchKernelGGL((mapper_kernel<detail::EXACT_MAPPING>), dim3(1), dim3(64), 0, 0, reads, fmi, hits, params);
    hipLaunchKernelGGL((mapper_kernel<detail::APPROX_MAPPING>), dim3(1), dim3(64), 0, 0, reads, fmi, hits, params);
    hipLaunchKernelGGL((mapper_kernel<detail::CASE_PRUNING_MAPPING>), dim3(1), dim3(64), 0, 0, reads, fmi, hits, params);
}}
}} }} }} // namespaces
"""))

CASES.append(("nvBowtie locate_inl.h: locate / locate_init / locate_lookup (same_type) + locate_kernel, locate_init_kernel, locate_lookup_kernel",
              [("nvBowtie/bowtie2/cuda/locate_inl.h", 39, 208)],
              NVBOWTIE_APP_PRELUDE + r"""
#define NVBIO_CUDA_ASSERT_IF(...)
namespace nvbio {{ namespace bowtie2 {{ namespace cuda {{
enum {{ BLOCKDIM = 96 }};
// ---- application-side types (scoring_queues.h: the hit queues a locate kernel rewrites in place; params.h)
struct ParamsPOD {{ uint32 dummy; }};
struct packed_seed {{ uint32 pos_in_read : 10, index_dir : 1, rc : 1, top_flag : 1; }};
struct HitQueuesDeviceView {{ packed_seed* seed; uint32* loc; uint32* ssa; }};
template <typename Q> struct HitReference
{{
    NVBIO_HOST_DEVICE HitReference(Q& q, const uint32 i) : seed(q.seed[i]), loc(q.loc[i]), ssa(q.ssa[i]) {{}}
    packed_seed& seed; uint32& loc; uint32& ssa;
}};
struct ReadBatch {{ uint32 n; }};
// ---- nvBowtie/bowtie2/cuda/locate_inl.h:39-208 (verbatim, opens namespace detail)
{0}
}} // namespace detail
typedef nvbio::cuda::ldg_pointer<uint4>                                     bwt_occ_iterator;
typedef PackedStream<deinterleaved_iterator<2, 0, bwt_occ_iterator>, uint8, 2, true> bwt_type;
typedef rank_dictionary<2, 64, bwt_type, deinterleaved_iterator<2, 1, bwt_occ_iterator>, nvbio::cuda::ldg_pointer<uint32> > rank_dict_type;
typedef fm_index<rank_dict_type, SSA_index_multiple_context<16, const uint32*>, nvbio::cuda::ldg_pointer<uint32> > fm_index_type;
void instantiate(const ReadBatch reads, const fm_index_type fmi, const uint32* idx, HitQueuesDeviceView hits, const ParamsPOD params)
{{
    hipLaunchKernelGGL((detail::locate_kernel<ReadBatch, fm_index_type, fm_index_type>), dim3(1), dim3(BLOCKDIM), 0, 0, reads, fmi, fmi, 1u, idx, hits, params);
    hipLaunchKernelGGL((detail::locate_init_kernel<ReadBatch, fm_index_type, fm_index_type>), dim3(1), dim3(BLOCKDIM), 0, 0, reads, fmi, fmi, 1u, idx, hits, params);
    hipLaunchKernelGGL((detail::locate_lookup_kernel<ReadBatch, fm_index_type, fm_index_type>), dim3(1), dim3(BLOCKDIM), 0, 0, reads, fmi, fmi, 1u, idx, hits, params);
}}
}} }} }} // namespaces
"""))

NVBOWTIE_SCORE_PRELUDE = r"""
#include <nvbio/basic/types.h>
#include <nvbio/basic/numbers.h>
#include <nvbio/basic/cuda/ldg.h>
#include <nvbio/basic/packedstream.h>
#include <nvbio/basic/packedstream_loader.h>
#include <nvbio/basic/vector_view.h>
#include <nvbio/io/utils.h>
#include <nvbio/io/alignments.h>
#include <nvbio/io/sequence/sequence.h>
#include <nvbio/alignment/alignment.h>
#include <nvbio/alignment/batched.h>
#include <map>
#include <string>
#define NVBIO_CUDA_DEBUG_PRINT_IF(...)
#define NVBIO_CUDA_DEBUG_CHECK_IF(...)
#define NVBIO_CUDA_ASSERT_IF(...)
#define DP_REPORT_MULTIPLE 0
namespace nvbio {{ namespace bowtie2 {{ namespace cuda {{
// ---- nvBowtie/bowtie2/cuda/func.h, SimpleFunc (verbatim)
{4}
// ---- application-side types of nvBowtie the verbatim ranges refer to (params.h, pipeline_states.h, scoring_queues.h)
struct ParamsPOD
{{
    struct Debug {{ NVBIO_HOST_DEVICE bool show_score_info(uint32) const {{ return false; }} NVBIO_HOST_DEVICE bool show_score(uint32, bool) const {{ return false; }} }} debug;
    uint32 pe_policy, min_frag_len, max_frag_len; bool pe_overlap, pe_dovetail, pe_unpaired;
}};
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE float phred_to_maq(const int q) {{ return float(q < 40 ? q : 40) / 10.0f; }}
struct SeedHitLike {{ uint32 rc; }};
struct HitQueuesDeviceView;
template <typename Q> struct HitReference {{ uint32 read_id; SeedHitLike seed; uint32 loc; int32 score; uint32 sink; int32 opposite_score, opposite_score2; uint32 opposite_loc, opposite_sink, opposite_loc2, opposite_sink2; }};
struct HitArray {{ NVBIO_HOST_DEVICE HitReference<HitQueuesDeviceView>& operator[](const uint32 i) const {{ return data[i]; }} HitReference<HitQueuesDeviceView>* data; }};
struct ScoringQueuesView {{ HitArray hits; }};
struct ReadBatch
{{
    static const uint32 SEQUENCE_BITS = 4; static const bool SEQUENCE_BIG_ENDIAN = true;
    typedef nvbio::cuda::ldg_pointer<uint32> sequence_storage_iterator; typedef nvbio::cuda::ldg_pointer<uint8> qual_storage_iterator;
    typedef PackedStream<sequence_storage_iterator, uint8, 4, true> sequence_stream_type;
    NVBIO_HOST_DEVICE sequence_stream_type sequence_stream() const {{ return sequence_stream_type(sequence_storage_iterator(words)); }}
    NVBIO_HOST_DEVICE qual_storage_iterator qual_stream() const {{ return qual_storage_iterator(quals); }}
    NVBIO_HOST_DEVICE uint2 get_range(const uint32 i) const {{ return make_uint2(index[i], index[i + 1]); }}
    NVBIO_HOST_DEVICE uint32 max_read_len() const {{ return 150u; }}
    NVBIO_HOST_DEVICE uint32 max_sequence_len() const {{ return 150u; }}
    const uint32* words; const uint8* quals; const uint32* index;
}};
template <typename scheme_t> struct PipelineLike
{{
    typedef scheme_t scheme_type; typedef ReadBatch read_batch_type;
    typedef PackedStream<nvbio::cuda::ldg_pointer<uint32>, uint8, 2, true> genome_iterator;
    read_batch_type reads, reads_o; genome_iterator genome; uint32 genome_length; uint32 anchor;
    const uint32* idx_queue; const uint32* opposite_queue; uint32 opposite_queue_size; ScoringQueuesView scoring_queues; uint32 hits_queue_size;
    const io::Alignment* best_alignments; const io::Alignment* best_alignments_o; uint32 best_stride; int32 score_limit;
    scheme_type scoring_scheme; uint8* dp_buffer; uint64 dp_buffer_size; uint32* buffer_read_info; io::Alignment* buffer_alignments;
}};
// ---- nvBowtie/bowtie2/cuda/scoring.h, cost functions and SmithWatermanScoringScheme (verbatim)
{0}
{1}
namespace detail {{
// ---- nvBowtie/bowtie2/cuda/alignment_utils.h (verbatim)
{2}
// ---- the score stream and its driver (verbatim)
{3}
}} // namespace detail
typedef SmithWatermanScoringScheme<>                         scheme_type;
typedef PipelineLike<scheme_type>                            pipeline_type;
typedef scheme_type::local_aligner_type                      local_aligner;
"""
NVBOWTIE_SCORE_RANGES = [("nvBowtie/bowtie2/cuda/scoring.h", 53, 125), ("nvBowtie/bowtie2/cuda/scoring.h", 196, 356),
                         ("nvBowtie/bowtie2/cuda/alignment_utils.h", 42, 345)]
FUNC_H = ("nvBowtie/bowtie2/cuda/func.h", 39, 70)

CASES.append(("nvBowtie score_paired_inl.h: BestAnchorScoreStream + banded_anchor_score_best (the enact calls of :212-237) -> tuned, on the read views in place",
              NVBOWTIE_SCORE_RANGES + [("nvBowtie/bowtie2/cuda/score_paired_inl.h", 48, 243), FUNC_H],
              NVBOWTIE_SCORE_PRELUDE + r"""
typedef detail::BestAnchorScoreStream<local_aligner, pipeline_type> stream_type;
static_assert(aln::priv::recognised<stream_type>::staged && aln::priv::recognised<stream_type>::stage_quals, "nvBowtie's BestAnchorScoreStream must run on the tuned kernels");
static_assert(aln::priv::recognised<stream_type>::view, "... on the read views in place");
void instantiate(const pipeline_type& pipeline, const scheme_type& scheme, const ParamsPOD params)
{{
    detail::banded_anchor_score_best(15u, pipeline, scheme.local_aligner(), params);
    detail::banded_anchor_score_best(31u, pipeline, scheme.end_to_end_aligner(), params);
}}
}} }} }} // namespaces
"""))

CASES.append(("nvBowtie score_all_inl.h: AllScoreStream + banded_score_all (the enact calls of :194-219) -> tuned, on the read views in place",
              NVBOWTIE_SCORE_RANGES + [("nvBowtie/bowtie2/cuda/score_all_inl.h", 48, 225), FUNC_H],
              NVBOWTIE_SCORE_PRELUDE + r"""
typedef detail::AllScoreStream<local_aligner, pipeline_type> stream_type;
static_assert(aln::priv::recognised<stream_type>::staged && aln::priv::recognised<stream_type>::stage_quals, "nvBowtie's AllScoreStream must run on the tuned kernels");
static_assert(aln::priv::recognised<stream_type>::view, "... on the read views in place");
void instantiate(const pipeline_type& pipeline, const scheme_type& scheme, const ParamsPOD params)
{{
    detail::banded_score_all(15u, pipeline, scheme.local_aligner(), params, 0u, 64u, (uint32*)0);
}}
}} }} }} // namespaces
"""))

CASES.append(("nvBowtie score_opposite_inl.h: BestOppositeScoreStream + opposite_score_best (full-matrix enact of :266-269, DeviceThreadBlockScheduler<128,9>) -> tuned (staged)",
              NVBOWTIE_SCORE_RANGES + [("nvBowtie/bowtie2/cuda/score_opposite_inl.h", 48, 274), FUNC_H],
              NVBOWTIE_SCORE_PRELUDE + r"""
typedef detail::BestOppositeScoreStream<local_aligner, pipeline_type> stream_type;
static_assert(aln::priv::recognised<stream_type>::staged && aln::priv::recognised<stream_type>::stage_quals, "nvBowtie's BestOppositeScoreStream must run on the tuned kernels");
void instantiate(const pipeline_type& pipeline, const scheme_type& scheme, const ParamsPOD params)
{{
    detail::opposite_score_best(pipeline, scheme.local_aligner(), params);
}}
}} }} }} // namespaces
"""))

CASES.append(("examples/fmmap/fmmap.cu: Pipeline (FMIndexFilterDevice over io::FMIndexDataDevice::fm_index_type), hit_to_diagonal, extract_seeds -> the filter's tuned route, seeds in place",
              [("examples/fmmap/fmmap.cu", 87, 149)], r"""
#include <nvbio/basic/vector.h>
#include <nvbio/basic/shared_pointer.h>
#include <nvbio/basic/dna.h>
#include <nvbio/strings/string_set.h>
#include <nvbio/strings/infix.h>
#include <nvbio/strings/seeds.h>
#include <nvbio/fmindex/filter.h>
#include <nvbio/io/sequence/sequence.h>
#include <nvbio/io/fmindex/fmindex.h>
using namespace nvbio;
struct Params {{ uint32 seed_len; uint32 seed_intv; uint32 merge_intv; }};
{0}
typedef io::SequenceDataAccess<DNA_N>::sequence_string_set_type read_string_set_type;
typedef InfixSet<read_string_set_type, const string_set_infix_coord_type*> seed_string_set_type;
static_assert(fmindex::production_layout<Pipeline::fm_index_type>::ok, "io::FMIndexDataDevice's index is the production layout: FMIndexFilterDevice runs the gfx950 kernels on it");
static_assert(nvbio::priv::packed_view<seed_string_set_type::string_type>::ok && nvbio::priv::packed_view<seed_string_set_type::string_type>::BITS == 4u,
              "a seed cut out of the packed read set is a window of packed words: ranked in place");
seed_string_set_type instantiate(const io::SequenceDataDevice& reads, nvbio::vector<device_tag, string_set_infix_coord_type>& coords)
{{
    const io::SequenceDataAccess<DNA_N> access(reads);
    return extract_seeds(access.sequence_string_set(), 22u, 10u, coords);
}}
"""))

# ------------------------------------------------------------------------------------------------------------------------------
# whole translation units of the reference's own test suite: compiled AS THEY LIE (hipcc -x hip <path under /root/reference>),
# linked against libnvbio_hip.so with a two-line main, and -- where the test is host code -- RUN here.  The only thing added on
# the command line is tools/port_cuda_calls.h (-include): the renames a maintainer's port applies to the APPLICATION's own CUDA
# runtime calls (cudaDeviceSynchronize -> hipDeviceSynchronize, ...); the library itself supplies no CUDA shim.
# `oracle/Makefile ref_tests` builds the same binaries into oracle/_ref/ so that the GPU suite can run them on the MI355X box.
# ------------------------------------------------------------------------------------------------------------------------------
WHOLE = []
WHOLE.append(dict(name="nvbio-test/rank_test.cu, whole TU, compiled as it lies and RUN (300 k symbols; uint32 / uint4 / uint64 dictionaries): the reference's own rank test passes on the drop-in rank_dictionary",
                  tu="nvbio-test/rank_test.cu", install="ref_rank_test", main="namespace nvbio { int rank_test(int argc, char* argv[]); }\nint main(int argc, char** argv) { return nvbio::rank_test(argc - 1, argv + 1); }\n",
                  run=["-length", "300"], expect="rank test... done"))
WHOLE.append(dict(name="nvbio-test/alignment_test.cu, whole TU (includes nvbio-test/alignment_test_utils.h in place), compiled as it lies, linked; host-only part RUN here (the six banded edit-distance literals), the device part runs on the GPU box from oracle/_ref/",
                  tu="nvbio-test/alignment_test.cu", install="ref_alignment_test", main="namespace nvbio { namespace aln { void test(int argc, char* argv[]); } }\nint main(int argc, char** argv) { nvbio::aln::test(argc - 1, argv + 1); return 0; }\n",
                  run=[], expect="synthetic Edit Distance test 6... passed!", may_abort=True))
WHOLE.append(dict(name="nvbio-test/fmindex_test.cu:56-717 (everything but the file-based backtracking test): host + device FM-index synthetic tests, 32- and 64-bit; compiled, linked; host part RUN here (SSA from FM-index, match, locate), the device part on the GPU box",
                  tu=None, install="ref_fmindex_test", ranges=[("nvbio-test/fmindex_test.cu", 56, 717)],
                  wrapper="#include <nvbio/basic/omp.h>\n#include <stdio.h>\n#include <stdlib.h>\n#include <string.h>\n#include <vector>\n#include <algorithm>\n#include <nvbio/basic/timer.h>\n#include <nvbio/basic/console.h>\n#include <nvbio/basic/dna.h>\n#include <nvbio/basic/cached_iterator.h>\n#include <nvbio/basic/packedstream.h>\n#include <nvbio/basic/deinterleaved_iterator.h>\n#include <nvbio/basic/cuda/ldg.h>\n#include <nvbio/fmindex/bwt.h>\n#include <nvbio/fmindex/ssa.h>\n#include <nvbio/fmindex/fmindex.h>\n{0}\nint main(int argc, char** argv)\n{\n    const uint32 len = argc > 1 ? atoi(argv[1]) : 100000, q = argc > 2 ? atoi(argv[2]) : 10000;\n    synthetic_test<uint32>(len, q);\n    synthetic_test<uint64>(len, q);\n    fprintf(stderr, \"fmindex synthetic test... done\\n\");\n    return 0;\n}\n",
                  main=None, run=["20000", "2000"], expect="cpu alignment... done", may_abort=True))
WHOLE.append(dict(name="sw-benchmark/sw-benchmark.cu, whole TU -- the program BASELINE's headline numbers come from -- compiled as it lies and linked (FASTQ reads + FASTA reference through the drop-in io::open_sequence_file / FASTA_inc_reader, AlignmentStream on the tuned kernels, its own per-thread kernel on the generic lane code); RUN here up to the first device allocation, on the GPU box from oracle/_ref/ to its own GCUPS printout",
                  tu="sw-benchmark/sw-benchmark.cu", install="ref_sw_benchmark", main=None, files="sw", run=["{reads}", "{ref}"], expect="reading reference file \"{ref}\"... done (3000 bps)", may_abort=True))

WHOLE.append(dict(name="examples/fmmap/fmmap.cu, whole TU (+ its util.h in place) -- FMIndexFilterDevice::rank / locate over an InfixSet of seeds cut from the read string-set (strings/infix.h, strings/seeds.h), "
                       "SparseStringSet windows into batch_banded_alignment_score<31> with the bit-vector edit-distance aligner and BestSink<int16>, cuda::reduce_by_key / cuda::reduce -- compiled as it lies and linked; "
                       "RUN here through the index and reference loaders, on the GPU box from oracle/_ref/ to its own 'aligned % reads' line",
                  tu="examples/fmmap/fmmap.cu", install="ref_fmmap", main=None, files="fmmap", run=["{index}", "{reads}"], expect="FMIndexData: loading... done", may_abort=True))

def sw_benchmark_files(tmp):
    """a 3 000-bp FASTA reference and 64 FASTQ reads cut from it, for the host part of the sw-benchmark run"""
    import random
    rnd = random.Random(7)
    ref = "".join(rnd.choice("ACGT") for _ in range(3000))
    ref_name, reads_name = os.path.join(tmp, "ref.fa"), os.path.join(tmp, "reads.fq")
    with open(ref_name, "w") as f:
        f.write(">chr1 synthetic\n" + "\n".join(ref[i:i + 70] for i in range(0, len(ref), 70)) + "\n")
    with open(reads_name, "w") as f:
        for i in range(64):
            p = rnd.randrange(0, 2900)
            f.write("@read%d\n%s\n+\n%s\n" % (i, ref[p:p + 100], "I" * 100))
    return dict(reads=reads_name, ref=ref_name)



def fmmap_files(tmp):
    """a 50-kbp index (.bwt / .sa / .wpac / .ann / .amb through nvbio_amd.io's writers) and 32 reads cut from it, for the host part of the fmmap run"""
    import numpy as np
    sys.path.insert(0, ROOT)
    from nvbio_amd import io as nio
    from oracle import pyoracle as O
    rng = np.random.default_rng(5)
    text = rng.integers(0, 4, 50_000, dtype=np.uint8)
    prefix = os.path.join(tmp, "fmmap_genome")
    nio.save_fmindex(prefix, O.FMIndex(text))
    nio.write_wpac(prefix + ".wpac", text.size, O.pack(text, 2, True)); nio.write_bns(prefix, ["chr1"], [text.size])
    reads_name = os.path.join(tmp, "fmmap_reads.fq")
    with open(reads_name, "w") as f:
        for i in range(32):
            q = int(rng.integers(200, 49_000))
            f.write("@read%d\n%s\n+\n%s\n" % (i, "".join("ACGT"[c] for c in text[q:q + 100]), "I" * 100))
    return dict(index=prefix, reads=reads_name)

def whole_cases(tmp, out, only, install=False):
    """compile + link (+ run) the whole-TU cases; returns the number of failures"""
    failed = 0
    port = os.path.join(ROOT, "tools", "port_cuda_calls.h")
    inc = os.path.join(tmp, "inc")
    os.makedirs(inc, exist_ok=True)
    if not os.path.exists(os.path.join(inc, "nvbio-test")):
        os.symlink(os.path.join(REF, "nvbio-test"), os.path.join(inc, "nvbio-test"))       # <nvbio-test/alignment_test_utils.h>, read in place
    for k, c in enumerate(WHOLE):
        idx = len(CASES) + k
        if only and idx not in only:
            continue
        exe = os.path.join(tmp, "whole%d" % k)
        srcs = []
        shas = []
        if c["tu"]:
            path = os.path.join(REF, c["tu"])
            srcs += ["-x", "hip", path]
            shas.append("%s  (whole file, %d lines, sha256 %s)" % (c["tu"], sum(1 for _ in open(path, errors="replace")), hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]))
        else:
            texts = [ref_lines(*r) for r in c["ranges"]]
            body = c["wrapper"]
            for i, t in enumerate(texts):
                body = body.replace("{%d}" % i, t)
            src = os.path.join(tmp, "whole%d.hip" % k)
            open(src, "w").write(body)
            srcs += [src]
            for (rel, a, b), t in zip(c["ranges"], texts):
                shas.append("%s:%d-%d  (%d lines verbatim, sha256 %s)" % (rel, a, b, b - a + 1, hashlib.sha256(t.encode()).hexdigest()[:16]))
        if c["main"]:
            m = os.path.join(tmp, "whole%d_main.hip" % k)
            open(m, "w").write(c["main"])
            srcs += ["-x", "hip", m]
        t0 = time.time()
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-std=c++17", "-O2", "-fopenmp", "-include", port] + srcs +
                           ["-I" + COMPAT, "-I" + inc, "-L" + os.path.join(ROOT, "nvbio_amd", "lib"), "-lnvbio_hip", "-lz",
                            "-Wl,-rpath,$ORIGIN/../../nvbio_amd/lib", "-o", exe], capture_output=True, text=True)
        ok = r.returncode == 0
        ran = ""
        if ok and c.get("run") is not None:
            subst = sw_benchmark_files(tmp) if c.get("files") == "sw" else fmmap_files(tmp) if c.get("files") == "fmmap" else {}
            run_args = [a.format(**subst) for a in c["run"]]
            expect = c["expect"].format(**subst) if subst else c["expect"]
            rr = subprocess.run([exe] + run_args, capture_output=True, text=True, timeout=600,
                                env=dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "nvbio_amd", "lib") + ":" + os.environ.get("LD_LIBRARY_PATH", "")))
            text = (rr.stdout + rr.stderr).replace("\r", "\n")
            good = expect in text and (rr.returncode == 0 or c.get("may_abort"))
            shown = expect.replace(tmp, "<tmp>")
            ran = "ran `%s`: %s (exit %d%s)" % (" ".join([os.path.basename(exe)] + [a.replace(tmp, "<tmp>") for a in run_args]), "reached \"%s\"" % shown if good else "DID NOT reach \"%s\"" % shown,
                                               rr.returncode, "; stops where the first device allocation needs a GPU" if c.get("may_abort") and rr.returncode != 0 else "")
            ok = ok and good
        failed += 0 if ok else 1
        if r.returncode == 0 and install:
            dst = os.path.join(ROOT, "oracle", "_ref", c["install"])
            os.makedirs(os.path.dirname(dst), exist_ok=True)
            shutil.copy2(exe, dst)
        out.append("[%s] %s" % ("PASS" if ok else "FAIL", c["name"]))
        out += ["       " + x for x in shas]
        out.append("       compiled and linked in %.1f s: %s" % (time.time() - t0, "yes" if r.returncode == 0 else "NO"))
        if ran:
            out.append("       " + ran)
        if r.returncode != 0:
            out += ["       " + e.replace(tmp, "<tmp>") for e in [l for l in r.stderr.splitlines() if "error" in l][:12]]
        out.append("")
    return failed


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--keep", action="store_true", help="keep the temporary TUs (for debugging; they hold reference text, do not commit them)")
    ap.add_argument("--log", default=os.path.join(ROOT, "profiles", "r05", "ref_bind_check.log"))
    ap.add_argument("--only", type=int, nargs="*", help="run only these case numbers")
    ap.add_argument("--install-ref-tests", action="store_true",
                    help="also copy the whole-TU binaries to oracle/_ref/ (git-ignored; they travel to the GPU box, where tests/test_ref_tests_gpu.py runs them)")
    args = ap.parse_args()
    if not os.path.isdir(REF):
        print("ref_bind_check: %s is not here (this check runs in the build container only)" % REF)
        return 0
    os.makedirs(os.path.dirname(args.log), exist_ok=True)
    ver = subprocess.run([HIPCC, "--version"], capture_output=True, text=True).stdout.splitlines()[0:1]
    out = ["ref_bind_check  %s" % time.strftime("%Y-%m-%d %H:%M:%S"),
           "compiler: %s" % (ver[0] if ver else HIPCC),
           "flags: --offload-arch=gfx950 -std=c++17 -O1 -c -I include/nvbio_hip/compat   (the reference text is read in place, wrapped in a temp TU, never stored)", ""]
    tmp = tempfile.mkdtemp(prefix="refbind_")
    failed = 0
    try:
        for k, (name, ranges, wrapper) in enumerate(CASES):
            if args.only and k not in args.only:
                continue
            texts = [ref_lines(*r) for r in ranges]
            src = os.path.join(tmp, "case%d.hip" % k)
            body = wrapper
            # wrappers written with doubled braces are format strings; the first case (single braces) substitutes by hand
            if "{{" in wrapper:
                body = wrapper.format(*texts)
            else:
                for i, t in enumerate(texts):
                    body = body.replace("{%d}" % i, t)
            with open(src, "w") as f:
                f.write(body)
            t0 = time.time()
            r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-std=c++17", "-O1", "-fopenmp", "-c", "-I" + COMPAT, src, "-o", os.path.join(tmp, "case%d.o" % k)],
                               capture_output=True, text=True)
            ok = r.returncode == 0
            failed += 0 if ok else 1
            out.append("[%s] %s" % ("PASS" if ok else "FAIL", name))
            for (rel, a, b), t in zip(ranges, texts):
                out.append("       %s:%d-%d  (%d lines verbatim, sha256 %s)" % (rel, a, b, b - a + 1, hashlib.sha256(t.encode()).hexdigest()[:16]))
            out.append("       compiled in %.1f s, static_asserts of the wrapper hold: %s" % (time.time() - t0, "yes" if ok else "NO"))
            if not ok:
                errs = [l for l in r.stderr.splitlines() if "error" in l][:12]
                out += ["       " + e.replace(tmp, "<tmp>") for e in errs]
            out.append("")
        failed += whole_cases(tmp, out, args.only, args.install_ref_tests)
    finally:
        if args.keep:
            print("kept", tmp)
        else:
            shutil.rmtree(tmp, ignore_errors=True)
    total = len(CASES) + len(WHOLE) if not args.only else len(args.only)
    out.append("%d / %d cases bind unchanged" % (total - failed, total))
    text = "\n".join(out) + "\n"
    with open(args.log, "w") as f:
        f.write(text)
    sys.stdout.write(text)
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
