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
CASES.append(("nvBowtie AlignmentStrings + AlignmentStreamBase + BestScoreStream + SmithWatermanScoringScheme -> tuned (staged)",
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
using namespace nvbio::io;
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
using namespace nvbio::io;
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--keep", action="store_true", help="keep the temporary TUs (for debugging; they hold reference text, do not commit them)")
    ap.add_argument("--log", default=os.path.join(ROOT, "profiles", "r03", "ref_bind_check.log"))
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
    finally:
        if args.keep:
            print("kept", tmp)
        else:
            shutil.rmtree(tmp, ignore_errors=True)
    out.append("%d / %d cases bind unchanged" % (len(CASES) - failed, len(CASES)))
    text = "\n".join(out) + "\n"
    with open(args.log, "w") as f:
        f.write(text)
    sys.stdout.write(text)
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
