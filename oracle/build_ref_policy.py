#!/usr/bin/env python3
"""oracle/_ref/libref_policy.so -- nvBowtie's own MAPQ and score-reduction code, compiled (test infrastructure; never linked into
the product).

Runs ONLY in the build container (needs /root/reference).  nvBowtie's headers do not compile as they lie in this image (defs.h ->
nvbio/basic/numbers.h -> <cuda_fp16.h> -> <nv/target>, which the image lacks; nothing is written in that header's place).  What this
recipe does instead is what tools/ref_bind_check.py does for the drop-in layer: it reads the LINE RANGES that hold the computation,
in place, wraps the verbatim text in a temporary translation unit (deleted afterwards; the reference text is never stored in the
repository), and compiles it with g++ against the drop-in layer's headers (<nvbio/basic/types.h>: uint32 / int32 / uint2,
NVBIO_HOST_DEVICE ...; <nvbio/alignment/alignment.h>: the aligner types scoring.h names in typedefs).  Every line of policy logic in
the library is the reference's:

    nvbio/io/alignments.h:37-474            io::Alignment bit-field, BestAlignments, PairedAlignments, BestPairedAlignments
    nvbio/io/alignments_inl.h:30-139        distinct_alignments
    nvBowtie/bowtie2/cuda/func.h:32-76      SimpleFunc (the --score-min function)
    nvBowtie/bowtie2/cuda/defs.h:84-183     BLOCKDIM ..., DebugState, packed_read, packed_seed
    nvBowtie/bowtie2/cuda/scoring.h         43-62 (enums), 82-440 (cost functions, EditDistance / SmithWaterman / Uber scoring schemes)
    nvBowtie/bowtie2/cuda/scoring_inl.h     102-122 (SmithWatermanScoringScheme's default constructor)
    nvBowtie/bowtie2/cuda/params.h:95-138   ParamsPOD
    nvBowtie/bowtie2/cuda/mapq.h:36-335     BowtieMapq3, BowtieMapq2
    nvBowtie/bowtie2/cuda/aligner_best_approx.h:48-82          MapqFunctorSE  (the single-end call site)
    nvBowtie/bowtie2/cuda/aligner_best_approx_paired.h:49-96   MapqFunctorPE  (the paired-end call site)
    nvBowtie/bowtie2/cuda/alignment_utils.h:40-98              detail::frame_opposite_mate
    nvBowtie/bowtie2/cuda/reduce.h:50-131                      ReduceBestApproxContext, ReduceBestExactContext
    nvBowtie/bowtie2/cuda/reduce_inl.h:47-498                  score_reduce_kernel, score_reduce_paired_kernel, try_update / update_* / replace_*
    nvBowtie/bowtie2/cuda/aligner.h:323-346                    init_alignments_kernel

The two kernels are templates over the pipeline type and name nvBowtie's hit-queue views (ScoringQueuesDeviceView,
ReadHitsReference<>) -- application types of nvBowtie, which the harness supplies in array-of-hits form (the same stance as
tools/ref_bind_check.py: around verbatim text the wrapper provides application-side types, never the logic under test).  A kernel
"launch" is a host loop over thread ids (`__global__` is defined away, threadIdx / blockIdx are harness variables).

    python oracle/build_ref_policy.py [--keep]      ->  oracle/_ref/libref_policy.so  (+ a build log beside it with the sha256 of every range)
"""
import argparse
import hashlib
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
COMPAT = os.path.join(ROOT, "include", "nvbio_hip", "compat")

RANGES = [
    ("nvbio/io/alignments.h", 37, 474),                                 # 0
    ("nvbio/io/alignments_inl.h", 30, 139),                             # 1
    ("nvBowtie/bowtie2/cuda/func.h", 32, 76),                           # 2
    ("nvBowtie/bowtie2/cuda/defs.h", 84, 183),                          # 3
    ("nvBowtie/bowtie2/cuda/scoring.h", 43, 62),                        # 4
    ("nvBowtie/bowtie2/cuda/scoring.h", 82, 440),                       # 5
    ("nvBowtie/bowtie2/cuda/scoring_inl.h", 102, 122),                  # 6
    ("nvBowtie/bowtie2/cuda/params.h", 95, 138),                        # 7
    ("nvBowtie/bowtie2/cuda/mapq.h", 36, 335),                          # 8
    ("nvBowtie/bowtie2/cuda/aligner_best_approx.h", 48, 82),            # 9
    ("nvBowtie/bowtie2/cuda/aligner_best_approx_paired.h", 49, 96),     # 10
    ("nvBowtie/bowtie2/cuda/alignment_utils.h", 40, 98),                # 11
    ("nvBowtie/bowtie2/cuda/reduce.h", 50, 131),                        # 12
    ("nvBowtie/bowtie2/cuda/reduce_inl.h", 47, 498),                    # 13
    ("nvBowtie/bowtie2/cuda/aligner.h", 323, 346),                      # 14
]

WRAPPER = r"""
#include <algorithm>
#include <cmath>
#include <cstring>
#include <cstdio>
#include <map>
#include <string>
#include <vector>
#include <cstdint>
#include <nvbio/basic/types.h>
#include <nvbio/alignment/alignment.h>              // the drop-in layer's aligner types (scoring.h names them in typedefs)
#include <nvbio/io/sequence/sequence_traits.h>      // io::PairedEndPolicy
#define __global__                                  /* a "launch" is a host loop, below */
#ifndef NVBIO_CUDA_DEBUG_PRINT_IF
#define NVBIO_CUDA_DEBUG_PRINT_IF(...)
#endif
static struct { unsigned x; } threadIdx, blockIdx;
{0}
}   // namespace nvbio (alignments.h closes io only inside the range)
{1}
{2}
namespace nvbio { namespace bowtie2 { namespace cuda {
{3}
} } }
{4}
{5}
namespace nvbio { namespace bowtie2 { namespace cuda {
{6}
{7}
} } }
{8}
namespace nvbio { namespace bowtie2 { namespace cuda {
{9}
{10}
{11}
}   // namespace detail (alignment_utils.h)

// ---- nvBowtie's hit-queue views, in array-of-hits form (application types; scoring_queues.h holds the production ones) ----
struct HitRecord
{
    uint32 read_id; packed_seed seed; int32 score; uint32 loc, sink;
    uint32 opposite_loc; int32 opposite_score; uint32 opposite_sink; int32 opposite_score2; uint32 opposite_sink2;
};
struct ScoringQueuesDeviceView
{
    uint32            n_active;
    const uint32*     read_ids;     // active slot -> read id
    const uint64_t*   hit_begin;    // active slot -> [begin, end) in hits
    const HitRecord*  hits;
    uint32 active_read_count() const { return n_active; }
};
template <typename ScoringQueuesType>
struct ReadHitsReference
{
    typedef HitRecord reference;
    ReadHitsReference(ScoringQueuesType& q, const uint32 slot) : m_q(q), m_slot(slot) {}
    packed_read read_info() const { return packed_read(m_q.read_ids[m_slot]); }
    uint32 size() const { return uint32(m_q.hit_begin[m_slot + 1] - m_q.hit_begin[m_slot]); }
    reference operator[] (const uint32 i) const { return m_q.hits[m_q.hit_begin[m_slot] + i]; }
    uint32 slot(const uint32 i) const { return uint32(m_q.hit_begin[m_slot] + i); }
    ScoringQueuesType& m_q; uint32 m_slot;
};
{12}
{13}
{14}
} } }

using namespace nvbio;
using namespace nvbio::bowtie2::cuda;

struct ReadLengths              // what MapqFunctorSE / PE ask of a read batch: sequence_index()[i + 1] - sequence_index()[i]
{
    const uint32* index;
    const uint32* sequence_index() const { return index; }
    uint2 get_range(const uint32 i) const { return make_uint2(index[i], index[i + 1]); }       // the kernels' pipeline.reads
    uint32 n; uint32 size() const { return n; }
};
struct HitsEraser { uint8_t* erased; void erase(const uint32 read_id) { erased[read_id] = 1; } };   // pipeline.hits.erase(): stop the read's traversal
struct Pipeline
{
    ScoringQueuesDeviceView scoring_queues;
    io::Alignment*          best_alignments;
    io::Alignment*          best_alignments_o;
    uint32                  best_stride;
    ReadLengths             reads;
    HitsEraser              hits;
    uint32                  anchor;
    int32                   score_limit;
};
typedef SmithWatermanScoringScheme<> scheme_type;
static scheme_type make_scheme(int match, int min_type, float min_k, float min_m, int monotone)
{
    scheme_type sc;
    sc.m_score_min = SimpleFunc(SimpleFunc::Type(min_type), min_k, min_m);
    sc.m_match     = scheme_type::MatchCost(match, match);
    sc.m_monotone  = monotone != 0;
    return sc;
}
static_assert(sizeof(io::Alignment) == 8, "io::Alignment is two words");

#define API extern "C" __attribute__((visibility("default")))

/// best: 2 x stride io::Alignment records as 64-bit words (bit-field word low, position high); seq_index: n + 1 offsets
API void ref_mapq_se(int version, int match, int min_type, float min_k, float min_m, int monotone,
                     uint32_t n, const uint64_t* best, uint32_t stride, const uint32_t* seq_index, uint8_t* out)
{
    const scheme_type sc = make_scheme(match, min_type, min_k, min_m, monotone);
    const ReadLengths reads = { seq_index, n };
    const io::Alignment* b = reinterpret_cast<const io::Alignment*>(best);
    if (version == 3) { MapqFunctorSE<BowtieMapq3<scheme_type>, ReadLengths> f(BowtieMapq3<scheme_type>(sc), b, stride, reads); for (uint32_t r = 0; r < n; ++r) out[r] = f(r); }
    else              { MapqFunctorSE<BowtieMapq2<scheme_type>, ReadLengths> f(BowtieMapq2<scheme_type>(sc), b, stride, reads); for (uint32_t r = 0; r < n; ++r) out[r] = f(r); }
}
API void ref_mapq_pe(int version, int mate, int match, int min_type, float min_k, float min_m, int monotone,
                     uint32_t n, const uint64_t* best, const uint64_t* best_o, uint32_t stride,
                     const uint32_t* seq_index1, const uint32_t* seq_index2, uint8_t* out)
{
    const scheme_type sc = make_scheme(match, min_type, min_k, min_m, monotone);
    const ReadLengths reads1 = { seq_index1, n }, reads2 = { seq_index2, n };
    const io::Alignment* b  = reinterpret_cast<const io::Alignment*>(best);
    const io::Alignment* bo = reinterpret_cast<const io::Alignment*>(best_o);
    if (version == 3) { MapqFunctorPE<BowtieMapq3<scheme_type>, ReadLengths> f(uint32(mate), BowtieMapq3<scheme_type>(sc), b, bo, stride, reads1, reads2); for (uint32_t r = 0; r < n; ++r) out[r] = f(r); }
    else              { MapqFunctorPE<BowtieMapq2<scheme_type>, ReadLengths> f(uint32(mate), BowtieMapq2<scheme_type>(sc), b, bo, stride, reads1, reads2); for (uint32_t r = 0; r < n; ++r) out[r] = f(r); }
}
/// io::Alignment's own constructor and accessors, for pinning the bit-field packing
API uint64_t ref_alignment_pack(uint32_t pos, uint32_t ed, int32_t score, uint32_t rc, uint32_t mate, int paired, int discordant)
{
    const io::Alignment a(pos, ed, score, rc, mate, paired != 0, discordant != 0);
    uint64_t w; std::memcpy(&w, &a, 8); return w;
}
API uint64_t ref_alignment_invalid() { const io::Alignment a = io::Alignment::invalid(); uint64_t w; std::memcpy(&w, &a, 8); return w; }
API void ref_alignment_unpack(uint64_t w, int32_t* out /* score, aligned, pos, rc, ed, mate, paired, unpaired, concordant, discordant */)
{
    io::Alignment a; std::memcpy(&a, &w, 8);
    out[0] = a.score(); out[1] = a.is_aligned(); out[2] = int32_t(a.alignment()); out[3] = a.is_rc(); out[4] = int32_t(a.ed());
    out[5] = int32_t(a.mate()); out[6] = a.is_paired(); out[7] = a.is_unpaired(); out[8] = a.is_concordant(); out[9] = a.is_discordant();
}
/// the distinct_alignments tests the reduction stages use (alignments_inl.h)
API int ref_distinct(uint32_t pos1, int rc1, uint32_t pos2, int rc2, uint32_t dist) { return io::distinct_alignments(pos1, rc1 != 0, pos2, rc2 != 0, dist); }
API int ref_distinct_paired(uint32_t apos1, uint32_t opos1, int arc1, int orc1, uint32_t apos2, uint32_t opos2, int arc2, int orc2, uint32_t dist)
{ return io::distinct_alignments(apos1, opos1, arc1 != 0, orc1 != 0, apos2, opos2, arc2 != 0, orc2 != 0, dist); }
/// BestPairedAlignments' derived quantities as the MAPQ and reporting stages read them
API void ref_best_paired(const uint64_t* w /* a1 a2 o1 o2 */, int32_t* out /* is_aligned is_paired has_second has_second_paired best_score second_score */)
{
    io::Alignment a[4]; std::memcpy(a, w, 32);
    const io::BestPairedAlignments b(io::BestAlignments(a[0], a[1]), io::BestAlignments(a[2], a[3]));
    out[0] = b.is_aligned(); out[1] = b.is_paired(); out[2] = b.has_second(); out[3] = b.has_second_paired(); out[4] = b.best_score(); out[5] = b.second_score();
}
API int ref_simple_func(int type, float k, float m, int x) { return SimpleFunc(SimpleFunc::Type(type), k, m)(x); }

// ---- score_reduce_kernel / score_reduce_paired_kernel: one "thread" per active read, launched as a host loop ----
struct ReduceHits       // the harness's flat hit arrays (one entry per extension result, grouped per active read by hit_begin)
{
    const uint32_t* loc; const uint32_t* sink; const int32_t* score; const uint8_t* rc; const uint8_t* top_flag;
    const uint32_t* o_loc; const uint32_t* o_sink; const uint32_t* o_sink2; const int32_t* o_score; const int32_t* o_score2;
};
static std::vector<HitRecord> gather_hits(const uint32_t n_active, const uint32_t* read_ids, const uint64_t* hit_begin, const ReduceHits& h, const bool paired)
{
    std::vector<HitRecord> hits(hit_begin[n_active]);
    for (uint32_t t = 0; t < n_active; ++t)
        for (uint64_t i = hit_begin[t]; i < hit_begin[t + 1]; ++i)
        {
            HitRecord r; std::memset(&r, 0, sizeof(r));
            r.read_id = read_ids[t]; r.seed = packed_seed(0u, 0u, h.rc[i], h.top_flag ? h.top_flag[i] : 0u);
            r.score = h.score[i]; r.loc = h.loc[i]; r.sink = h.sink ? h.sink[i] : 0u;
            if (paired) { r.opposite_loc = h.o_loc[i]; r.opposite_score = h.o_score[i]; r.opposite_sink = h.o_sink[i]; r.opposite_score2 = h.o_score2[i]; r.opposite_sink2 = h.o_sink2[i]; }
            hits[i] = r;
        }
    return hits;
}
template <typename Kernel>
static void launch(const uint32_t n_active, Kernel k)
{
    for (uint32_t t = 0; t < n_active; ++t) { blockIdx.x = t / BLOCKDIM; threadIdx.x = t % BLOCKDIM; k(); }
}
/// init_alignments_kernel with the scheme's threshold_score() as the worst-score function (aligner_best_approx.h: init_alignments(reads, threshold_score, best, stride, mate))
API void ref_init_alignments(int min_type, float min_k, float min_m, uint32_t n, const uint32_t* seq_index, uint32_t mate, uint64_t* best, uint32_t stride)
{
    const ReadLengths reads = { seq_index, n };
    const SimpleFunc f(SimpleFunc::Type(min_type), min_k, min_m);
    launch(n, [&] { init_alignments_kernel(reads, f, reinterpret_cast<io::Alignment*>(best), stride, mate); });
}
/// context: 0 = ReduceBestExactContext, 1 = ReduceBestApproxContext(trys, n_ext) with params.{max_effort, min_ext, max_ext}
API void ref_score_reduce(int context, uint32_t n_active, const uint32_t* read_ids, const uint64_t* hit_begin,
                          const uint32_t* loc, const int32_t* score, const uint8_t* rc, const uint8_t* top_flag,
                          const uint32_t* seq_index, uint32_t* trys, uint32_t n_ext, uint32_t max_effort, uint32_t min_ext, uint32_t max_ext,
                          uint64_t* best, uint32_t stride, uint8_t* erased)
{
    const ReduceHits h = { loc, nullptr, score, rc, top_flag, nullptr, nullptr, nullptr, nullptr, nullptr };
    const std::vector<HitRecord> hits = gather_hits(n_active, read_ids, hit_begin, h, false);
    Pipeline p; std::memset(&p, 0, sizeof(p));
    p.scoring_queues.n_active = n_active; p.scoring_queues.read_ids = read_ids; p.scoring_queues.hit_begin = hit_begin; p.scoring_queues.hits = hits.data();
    p.best_alignments = reinterpret_cast<io::Alignment*>(best); p.best_stride = stride; p.reads.index = seq_index; p.hits.erased = erased;
    ParamsPOD params; std::memset(&params, 0, sizeof(params));
    params.max_effort = max_effort; params.min_ext = min_ext; params.max_ext = max_ext;
    if (context == 0) launch(n_active, [&] { detail::score_reduce_kernel<scheme_type>(ReduceBestExactContext(), p, params); });
    else              launch(n_active, [&] { detail::score_reduce_kernel<scheme_type>(ReduceBestApproxContext(trys, n_ext), p, params); });
}
API void ref_score_reduce_paired(int context, uint32_t n_active, const uint32_t* read_ids, const uint64_t* hit_begin,
                                 const uint32_t* loc, const uint32_t* sink, const int32_t* score, const uint8_t* rc, const uint8_t* top_flag,
                                 const uint32_t* o_loc, const uint32_t* o_sink, const uint32_t* o_sink2, const int32_t* o_score, const int32_t* o_score2,
                                 const uint32_t* seq_index, uint32_t anchor, int pe_policy, int pe_unpaired, int32_t score_limit,
                                 uint32_t* trys, uint32_t n_ext, uint32_t max_effort, uint32_t min_ext, uint32_t max_ext,
                                 uint64_t* best, uint64_t* best_o, uint32_t stride, uint8_t* erased)
{
    const ReduceHits h = { loc, sink, score, rc, top_flag, o_loc, o_sink, o_sink2, o_score, o_score2 };
    const std::vector<HitRecord> hits = gather_hits(n_active, read_ids, hit_begin, h, true);
    Pipeline p; std::memset(&p, 0, sizeof(p));
    p.scoring_queues.n_active = n_active; p.scoring_queues.read_ids = read_ids; p.scoring_queues.hit_begin = hit_begin; p.scoring_queues.hits = hits.data();
    p.best_alignments = reinterpret_cast<io::Alignment*>(best); p.best_alignments_o = reinterpret_cast<io::Alignment*>(best_o); p.best_stride = stride;
    p.reads.index = seq_index; p.hits.erased = erased; p.anchor = anchor; p.score_limit = score_limit;
    ParamsPOD params; std::memset(&params, 0, sizeof(params));
    params.max_effort = max_effort; params.min_ext = min_ext; params.max_ext = max_ext; params.pe_policy = uint32(pe_policy); params.pe_unpaired = pe_unpaired != 0;
    if (context == 0) launch(n_active, [&] { detail::score_reduce_paired_kernel<scheme_type>(ReduceBestExactContext(), p, params); });
    else              launch(n_active, [&] { detail::score_reduce_paired_kernel<scheme_type>(ReduceBestApproxContext(trys, n_ext), p, params); });
}
"""


def ref_lines(rel, first, last):
    with open(os.path.join(REF, rel), "r", errors="replace") as f:
        lines = f.readlines()
    return "".join(lines[first - 1:last])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--keep", action="store_true")
    args = ap.parse_args()
    if not os.path.isdir(REF):
        print("build_ref_policy: %s is not here (the prebuilt oracle/_ref/libref_policy.so is what travels)" % REF)
        return 0
    out_dir = os.path.join(HERE, "_ref")
    os.makedirs(out_dir, exist_ok=True)
    tmp = tempfile.mkdtemp(prefix="refpolicy_")
    try:
        body = WRAPPER
        log = ["libref_policy.so: verbatim reference ranges compiled by g++ against include/nvbio_hip/compat/nvbio/basic/types.h", ""]
        for i, (rel, a, b) in enumerate(RANGES):
            t = ref_lines(rel, a, b)
            body = body.replace("{%d}" % i, t)
            log.append("%s:%d-%d  (%d lines verbatim, sha256 %s)" % (rel, a, b, b - a + 1, hashlib.sha256(t.encode()).hexdigest()[:16]))
        src = os.path.join(tmp, "ref_policy.cpp")
        with open(src, "w") as f:
            f.write(body)
        cmd = ["g++", "-O2", "-w", "-fPIC", "-shared", "-std=c++17", "-fvisibility=hidden", "-ffp-contract=off", "-I" + COMPAT, src, "-o", os.path.join(out_dir, "libref_policy.so")]
        r = subprocess.run(cmd, capture_output=True, text=True)
        log += ["", " ".join(c.replace(tmp, "<tmp>").replace(ROOT + "/", "") for c in cmd), "exit %d" % r.returncode]
        log += [l.replace(tmp, "<tmp>") for l in r.stderr.splitlines() if "error" in l][:20]
        with open(os.path.join(out_dir, "ref_policy_build.log"), "w") as f:
            f.write("\n".join(log) + "\n")
        if r.returncode != 0:
            sys.stderr.write(r.stderr[-4000:])
        return r.returncode
    finally:
        if args.keep:
            print("kept", tmp)
        else:
            shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    sys.exit(main())
