/*
 * nvbio_hip.h -- C-ABI of the MI355X (gfx950) seed-and-extend hot path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++ or torch types.
 * Every pointer is a DEVICE pointer into caller-owned HBM unless stated
 * otherwise; nothing is allocated, retained or freed by the library except
 * where a `temp` buffer is passed in explicitly.  All entry points enqueue work
 * on `stream` (a hipStream_t passed as void*; NULL = the null stream) and
 * return immediately with 0 or a hipError_t value; they never throw and never
 * fall back to the CPU.
 *
 * The reference (NVlabs/nvbio) has no ABI for this path -- it is C++ templates
 * (SURVEY.md 8b).  Each entry point below names the reference interface it
 * replaces; nvbio_amd/include/nvbio_hip/ *.h holds the C++ host layer that
 * mirrors those template names on top of this ABI.
 */
#ifndef NVBIO_HIP_H
#define NVBIO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NVBIO_HIP_ABI_VERSION 2

/* nvbio::aln::AlignmentType (nvbio/alignment/alignment_base.h:54) */
enum { NVBIO_HIP_GLOBAL = 0, NVBIO_HIP_LOCAL = 1, NVBIO_HIP_SEMI_GLOBAL = 2 };

/* A set of strings stored in one packed word stream, i.e. what the reference
 * expresses as a string-set over nvbio::PackedStream<const uint32*,uint8,BITS,
 * BIG_ENDIAN> (nvbio/basic/packedstream.h, packedstream_inl.h:336-400):
 * string i = symbols [begin[i], begin[i]+length[i]) of the stream.
 *   words      the stream's uint32 words (device)
 *   n_words    number of words allocated; the kernels read whole 16-symbol
 *              groups and clamp every word load to [0,n_words), so a string may
 *              end anywhere in the stream without padding
 *   bits       2 or 4 (the production read / genome formats; 8-bit strings
 *              are packed by the host layer first)
 *   big_endian PackedStream's BIG_ENDIAN_T
 *   begin      n symbol offsets (uint64, device)
 *   length     n lengths (device), or NULL -> every string has `fixed_length` symbols */
typedef struct nvbio_hip_string_set {
    const uint32_t* words;
    uint64_t        n_words;
    uint32_t        bits;
    uint32_t        big_endian;
    const uint64_t* begin;
    const uint32_t* length;
    uint32_t        fixed_length;
    uint32_t        _pad;
} nvbio_hip_string_set;

/* nvbio::aln::SimpleGotohScheme (nvbio/alignment/utils.h:114-134) */
typedef struct nvbio_hip_gotoh_scheme {
    int32_t match, mismatch, gap_open, gap_ext;
} nvbio_hip_gotoh_scheme;

/* nvbio::aln::BestSink<int32> (nvbio/alignment/sink.h:68-89) is returned as
 * two arrays: score[n] and sink[2n] = {sink.x, sink.y}. */

/*
 * Replaces  BatchedBandedAlignmentScore<BAND_LEN, stream_type,
 *           DeviceThreadBlockScheduler<128,1>>::enact(stream)
 *           (nvbio/alignment/batched.h:333-353, batched_banded_inl.h:135-162) and
 *           batch_banded_alignment_score<BAND_LEN>(aligner, patterns, texts, sinks, ...)
 *           (nvbio/alignment/batched_inl.h:1074-1101)
 * for aligner = GotohAligner<TYPE, SimpleGotohScheme>, trivial qualities:
 * job i scores patterns[i] against texts[i] exactly as
 * priv::banded::gotoh_alignment_score_dispatch<BAND_LEN,TYPE>::run
 * (nvbio/alignment/gotoh/gotoh_banded_inl.h:415-658) into a fresh BestSink.
 * band_len in {3,5,7,15,31}.  A job whose text is shorter than its pattern
 * leaves the sink invalid (score = -(1<<30), sink = (0xFFFFFFFF,0xFFFFFFFF)).
 */
int nvbio_hip_banded_gotoh_score(
    const nvbio_hip_gotoh_scheme* scheme /* host */, int32_t type, uint32_t band_len,
    const nvbio_hip_string_set* patterns /* host struct, device arrays */,
    const nvbio_hip_string_set* texts    /* host struct, device arrays */,
    uint32_t max_pattern_len, uint32_t max_text_len /* the reference's max_pattern_length /
        max_text_length arguments (batched.h:217-231); 0 = unknown.  max_pattern_len only sizes
        the per-lane LDS staging of the strings (longer jobs read HBM directly): results never
        depend on either. */,
    uint32_t n, int32_t* out_score, uint32_t* out_sink, void* stream);

/*
 * Full-matrix (unbanded) Gotoh score.  Replaces
 *   BatchedAlignmentScore<stream, DeviceThreadScheduler>::enact (nvbio/alignment/batched.h:310-329,
 *   batched_inl.h:383-460) and batch_alignment_score(...) (batched.h:160-190)
 * for aligner = GotohAligner<TYPE, SimpleGotohScheme, TextBlockingTag> -- the instantiation of
 * sw-benchmark (sw-benchmark/sw-benchmark.cu:604-631): job i scores patterns[i] against the whole of
 * texts[i] exactly as priv::gotoh_alignment_score_dispatch<8,TYPE,TextBlockingTag,.>::run
 * (nvbio/alignment/gotoh/gotoh_inl.h:969-1489) does into a fresh BestSink, including its int16
 * boundary column, its tie order and its early exit against min_score[i] (NULL = never).
 * out_ok (nullable) receives that function's bool (0 = early-exited).  Patterns up to 1024 symbols
 * (max_pattern_len / max_text_len are required for ragged sets: they select the register layout). */
int nvbio_hip_gotoh_score(
    const nvbio_hip_gotoh_scheme* scheme /* host */, int32_t type,
    const nvbio_hip_string_set* patterns, const nvbio_hip_string_set* texts,
    uint32_t max_pattern_len, uint32_t max_text_len, const int32_t* min_score /* device, nullable */,
    uint32_t n, int32_t* out_score, uint32_t* out_sink, uint8_t* out_ok, void* stream);

/* nvbio::aln::SimpleSmithWatermanScheme (nvbio/alignment/utils.h:92-109): linear gap costs.
 * EditDistanceAligner is this scheme with {0,-1,-1,-1} (ed/ed_utils.h:44-51). */
typedef struct nvbio_hip_sw_scheme { int32_t match, mismatch, deletion, insertion; } nvbio_hip_sw_scheme;

/* As nvbio_hip_banded_gotoh_score / nvbio_hip_gotoh_score, for
 *   aligner = SmithWatermanAligner<TYPE, SimpleSmithWatermanScheme>  (sw/sw_banded_inl.h:340-520; sw/sw_inl.h:881-1222,
 *             TextBlockingTag, 16-column blocks, int16 boundary column, no early exit)
 *   aligner = EditDistanceAligner<TYPE>                              (ed/ed_banded_inl.h, ed/ed_inl.h:85-99: the same
 *             code with EditDistanceSWScheme) -- the second leg of sw-benchmark (sw-benchmark.cu:641-657).
 * Schemes with deletion != insertion are refused (801); the full-matrix form also refuses schemes/lengths whose
 * values could leave int16 ((M [+ N for GLOBAL] + 4) * max|cost| >= 30000). */
int nvbio_hip_banded_sw_score(
    const nvbio_hip_sw_scheme* scheme /* host */, int32_t type, uint32_t band_len,
    const nvbio_hip_string_set* patterns, const nvbio_hip_string_set* texts,
    uint32_t max_pattern_len, uint32_t max_text_len,
    uint32_t n, int32_t* out_score, uint32_t* out_sink, void* stream);
int nvbio_hip_sw_score(
    const nvbio_hip_sw_scheme* scheme /* host */, int32_t type,
    const nvbio_hip_string_set* patterns, const nvbio_hip_string_set* texts,
    uint32_t max_pattern_len, uint32_t max_text_len,
    uint32_t n, int32_t* out_score, uint32_t* out_sink, void* stream);

/* The general full-matrix score: aligner kind x algorithm tag (alignment_base.h:72-79), i.e.
 *   BatchedAlignmentScore<stream,...>::enact for GotohAligner / SmithWatermanAligner / EditDistanceAligner with
 *   PatternBlockingTag (the default of make_*_aligner<TYPE>(...); gotoh_inl.h:459-900, sw_inl.h:417-760) or TextBlockingTag.
 * scheme4 = {match, mismatch, gap_open, gap_ext} (Gotoh) or {match, mismatch, deletion, insertion} (SW; ED = {0,-1,-1,-1}).
 * The two tags compute the same matrix; they differ in the visiting order (which of several equal LOCAL maxima the sink
 * keeps), and in the early exit against min_score[i]: per block of 8 (16) pattern symbols for PatternBlockingTag
 * (max_i H[i][block end] + (M - block end) * match < min_score), per block of text columns for TextBlockingTag.
 * PatternBlockingTag runs on the 16-bit sweep only (values within int16, else 801) and needs patterns of >= 1 symbol. */
enum { NVBIO_HIP_GOTOH_ALIGNER = 0, NVBIO_HIP_SW_ALIGNER = 1 };
enum { NVBIO_HIP_PATTERN_BLOCKING = 0, NVBIO_HIP_TEXT_BLOCKING = 1 };
int nvbio_hip_alignment_score(
    int32_t aligner, int32_t algorithm, const int32_t* scheme4 /* host */, int32_t type,
    const nvbio_hip_string_set* patterns, const nvbio_hip_string_set* texts,
    uint32_t max_pattern_len, uint32_t max_text_len, const int32_t* min_score /* device, nullable */,
    uint32_t n, int32_t* out_score, uint32_t* out_sink, uint8_t* out_ok /* nullable */, void* stream);

/* nvBowtie's SmithWatermanScoringScheme<QualCost,ConstantCost> as the Gotoh aligner sees it
 * (nvBowtie/bowtie2/cuda/scoring.h:283-293): substitution(r,q,qq) = (r == q) ? match : mismatch[qq],
 * with mismatch[qq] = -m_mmp(qq) tabulated by the host for every quality byte (the float->int
 * truncation of QualCost, scoring.h:86-104, stays on the host), and the four gap accessors. */
typedef struct nvbio_hip_gotoh_qual_scheme {
    int32_t match;
    int32_t pattern_gap_open, pattern_gap_ext, text_gap_open, text_gap_ext;   /* as returned by the accessors (<= 0) */
    int32_t mismatch[256];
} nvbio_hip_gotoh_qual_scheme;

/* As nvbio_hip_banded_gotoh_score, for GotohAligner<TYPE, SmithWatermanScoringScheme<...>> with a
 * quality string per pattern: replaces the instantiation in nvBowtie's extension stage
 * (nvBowtie/bowtie2/cuda/score_best_inl.h:153-201).  quals[b + i] is the quality of symbol i of the
 * pattern that begins at stream index b (nvBowtie stores read qualities at the reads' own offsets);
 * n_quals = bytes allocated. */
int nvbio_hip_banded_gotoh_score_qual(
    const nvbio_hip_gotoh_qual_scheme* scheme /* host */, int32_t type, uint32_t band_len,
    const nvbio_hip_string_set* patterns, const uint8_t* quals, uint64_t n_quals,
    const nvbio_hip_string_set* texts,
    uint32_t max_pattern_len, uint32_t max_text_len,
    uint32_t n, int32_t* out_score, uint32_t* out_sink, void* stream);

/* The same with a per-job VIEW of each pattern, so that nvBowtie's unedited streams run in place.  nvBowtie never materialises
 * the string it aligns: AlignmentStrings::load (nvBowtie/bowtie2/cuda/alignment_utils.h:170-218) hands the DP an
 * nvbio::io::ReadStream (nvbio/io/utils.h:100-330) over the stored read -- reads are stored reversed, (REVERSE, STANDARD) shows
 * the forward strand and (FORWARD, COMPLEMENT) the reverse complement -- and the reference's per-thread DP reads it symbol by
 * symbol through operator[].  pattern_flags[i] (device, one byte per job, NULL = no views) describes job i's view of the stored
 * symbols [begin[i], begin[i] + length[i]):  bit 0 = walk them backwards (pattern symbol k is stored symbol begin + length - 1 - k,
 * and its quality is quals[begin + length - 1 - k]),  bit 1 = complement the bases (c < 4 -> 3 - c; N stays).  The kernels turn
 * each 16-symbol group round in registers as they fetch it; results are those of the materialised string, bit for bit. */
int nvbio_hip_banded_gotoh_score_qual_views(
    const nvbio_hip_gotoh_qual_scheme* scheme /* host */, int32_t type, uint32_t band_len,
    const nvbio_hip_string_set* patterns, const uint8_t* quals, uint64_t n_quals, const uint8_t* pattern_flags /* device, nullable */,
    const nvbio_hip_string_set* texts,
    uint32_t max_pattern_len, uint32_t max_text_len,
    uint32_t n, int32_t* out_score, uint32_t* out_sink, void* stream);

/* The same scorer with ONE WAVE PER JOB (nvbio_amd/csrc/banded_gotoh_wave.hip): lane j of a wave owns band cell j and the wave sweeps the
 * anti-diagonals 2 i + j = w -- 2 M + BAND steps instead of M * BAND cells in sequence.  For batches too small to fill the chip (a few
 * thousand jobs: nvBowtie's extension rounds after the first), where the lane-per-job kernel takes as long as one job's cells in sequence
 * whatever the batch size.  Same results bit for bit.  Job k (k < n, or < *n_on_device when given) is job_index[k] of the string sets
 * (job_index == NULL: k itself) and writes out_score / out_sink at that index -- a list of live jobs run in place -- or, with out_index, at
 * out_index[that index].  gate (device, optional): the launch does nothing unless *gate <= gate_limit -- the lane-per-job entry below takes the same
 * pair and runs when *gate > gate_limit, so a driver queues both and the device picks the form that suits the batch, with no host round trip.
 * Patterns up to 512 symbols (hipErrorNotSupported beyond; max_pattern_len announces the longest of a ragged set); no pattern views. */
int nvbio_hip_banded_gotoh_score_qual_wave(
    const nvbio_hip_gotoh_qual_scheme* scheme, int32_t type, uint32_t band_len,
    const nvbio_hip_string_set* patterns, const uint8_t* quals /* nullable: mismatch[0] throughout */, uint64_t n_quals,
    const nvbio_hip_string_set* texts, uint32_t max_pattern_len,
    uint32_t n, const uint32_t* n_on_device /* nullable */, const uint32_t* job_index /* nullable */, const uint32_t* out_index /* nullable */,
    const uint32_t* gate /* nullable */, uint32_t gate_limit,
    int32_t* out_score, uint32_t* out_sink, void* stream);

/* The same scorer with a threshold per job -- the reference's `min_score` argument of banded_alignment_score (alignment_inl.h; its windowed
 * form gives up on a job once max(H) < min_score + remaining_rows * match, gotoh_banded_inl.h:622-634), which nvBowtie's scoring stage sets to
 * the read's second-best score (score_best_inl.h:113-116) and whose reduction only asks whether a score is ABOVE that (reduce_inl.h:111-135).
 *   job i ends above min_score[i]:   out_score / out_sink are what nvbio_hip_banded_gotoh_score_qual_views reports, bit for bit;
 *   job i ends at or below it:       out_score[i] is SOME value <= min_score[i] (an upper bound of the true score), out_sink[i] = (-1, -1).
 * min_score == NULL or min_score[i] == INT32_MIN: exact.  Persistent waves pull jobs from *work_counter (device, 4 bytes, zeroed by the call):
 * a lane whose job is given up or done takes the next (nvbio_amd/csrc/banded_gotoh_bounded.h).  n_on_device (optional, device): the number
 * of jobs, read by the kernel -- `n` is then only an upper bound (array capacity), and the host need not know the count.  out_index (optional,
 * device): job i's results are written at out_score[out_index[i]] / out_sink[out_index[i]] -- a compacted batch of jobs (nvbio_hip_score_best_setup's
 * job_hit) writing straight back at its hits.  gate / gate_limit (min_score == NULL only): the launch does nothing unless *gate > gate_limit (see
 * nvbio_hip_banded_gotoh_score_qual_wave). */
int nvbio_hip_banded_gotoh_score_qual_bounded(
    const nvbio_hip_gotoh_qual_scheme* scheme, int32_t type, uint32_t band_len,
    const nvbio_hip_string_set* patterns, const uint8_t* quals, uint64_t n_quals, const uint8_t* pattern_flags,
    const nvbio_hip_string_set* texts,
    uint32_t max_pattern_len, uint32_t max_text_len,
    uint32_t n, const uint32_t* n_on_device, const int32_t* min_score, uint32_t* work_counter, const uint32_t* out_index,
    const uint32_t* gate /* nullable; with min_score == NULL only */, uint32_t gate_limit,
    int32_t* out_score, uint32_t* out_sink, void* stream);

/* Batched banded Gotoh traceback.  Replaces
 *   BatchedBandedAlignmentTraceback<BAND_LEN, CHECKPOINTS, stream, DeviceThreadScheduler>::enact
 *   (nvbio/alignment/batched.h:460-476, batched_banded_inl.h:250-420)
 * for aligner = GotohAligner<TYPE, scheme> and the CIGAR-forming backtracer nvBowtie passes
 * (Backtracker<io::Cigar*>, nvBowtie/bowtie2/cuda/alignment_utils.h:125-168), the instantiation of
 * nvBowtie's traceback stage (traceback_inl.h:205-262).  Job i re-scores patterns[i] in the band of
 * texts[i] (banded_alignment_traceback, nvbio/alignment/banded_inl.h:352-423) and walks the flow flags
 * back (gotoh_banded_inl.h:878-960):
 *   out_score[i], out_sink[2i..]     Alignment::score / ::sink   (== nvbio_hip_banded_gotoh_score's)
 *   out_source[2i..]                 Alignment::source           ((-1,-1) when there is no valid sink)
 *   out_cigar[i*cigar_stride + k]    io::Cigar as a uint16: type | len << 2 (nvbio/io/alignments.h:57-75;
 *                                    0 = M, 1 = I, 2 = D, 3 = soft clip), in the backtracer's order, i.e.
 *                                    the END of the alignment first, exactly what context->cigar holds
 *   out_cigar_len[i]                 backtracer.size; entries beyond cigar_stride are counted, not written
 * CHECKPOINTS has no equivalent: the flow flags of the whole band are kept in `temp`
 * (nvbio_hip_banded_gotoh_traceback_temp_bytes), which reproduces the reference for every interval
 * as long as the DP values fit its int16 checkpoints; schemes/lengths with
 * (max_pattern_len + band_len + 2) * max|cost| >= 32000 are refused with 801.
 * max_pattern_len is required for ragged pattern sets. */
uint64_t nvbio_hip_banded_gotoh_traceback_temp_bytes(uint32_t band_len, uint32_t max_pattern_len, uint32_t n);
int nvbio_hip_banded_gotoh_traceback(
    const nvbio_hip_gotoh_scheme* scheme /* host */, int32_t type, uint32_t band_len,
    const nvbio_hip_string_set* patterns, const nvbio_hip_string_set* texts,
    uint32_t max_pattern_len, uint32_t max_text_len, uint32_t n,
    int32_t* out_score, uint32_t* out_sink, uint32_t* out_source,
    uint16_t* out_cigar, uint32_t cigar_stride, uint32_t* out_cigar_len,
    void* temp, uint64_t temp_bytes, void* stream);
int nvbio_hip_banded_gotoh_traceback_qual(
    const nvbio_hip_gotoh_qual_scheme* scheme /* host */, int32_t type, uint32_t band_len,
    const nvbio_hip_string_set* patterns, const uint8_t* quals, uint64_t n_quals,
    const nvbio_hip_string_set* texts,
    uint32_t max_pattern_len, uint32_t max_text_len, uint32_t n,
    int32_t* out_score, uint32_t* out_sink, uint32_t* out_source,
    uint16_t* out_cigar, uint32_t cigar_stride, uint32_t* out_cigar_len,
    void* temp, uint64_t temp_bytes, void* stream);
/* the same, with score[] / sink[] filled by the caller with what nvbio_hip_banded_gotoh_score_qual reports for these jobs (skips the score pass) */
int nvbio_hip_banded_gotoh_traceback_qual_known(
    const nvbio_hip_gotoh_qual_scheme* scheme /* host */, int32_t type, uint32_t band_len,
    const nvbio_hip_string_set* patterns, const uint8_t* quals, uint64_t n_quals,
    const nvbio_hip_string_set* texts,
    uint32_t max_pattern_len, uint32_t max_text_len, uint32_t n,
    int32_t* score /* in/out */, uint32_t* sink /* in/out */, uint32_t* out_source,
    uint16_t* out_cigar, uint32_t cigar_stride, uint32_t* out_cigar_len,
    void* temp, uint64_t temp_bytes, void* stream);

/* Batched full-matrix Gotoh traceback.  Replaces
 *   BatchedAlignmentTraceback<CHECKPOINTS, stream, DeviceThreadScheduler>::enact (nvbio/alignment/batched.h:432-452)
 * for aligner = GotohAligner<TYPE, SimpleGotohScheme> with nvBowtie's CIGAR-forming backtracer: alignment_traceback
 * (alignment_inl.h:365-480) -- the sink of the pattern-blocking score pass, the flow flags of gotoh_inl.h:512-560 /
 * :407-425, the walk of gotoh_inl.h:1806-1870 and the driver's completion along the first row / column (:443-466).
 * Outputs as nvbio_hip_banded_gotoh_traceback (source/sink: x = text position, y = pattern position).
 * The flow flags of the whole matrix live in `temp` (nvbio_hip_gotoh_traceback_temp_bytes: 4 bytes per 8 cells + 4 bytes
 * per text symbol, per job); CHECKPOINTS has no equivalent.  Requires the int16-exact range of the full-matrix scorer. */
uint64_t nvbio_hip_gotoh_traceback_temp_bytes(uint32_t max_pattern_len, uint32_t max_text_len, uint32_t n);
int nvbio_hip_gotoh_traceback(
    const nvbio_hip_gotoh_scheme* scheme /* host */, int32_t type,
    const nvbio_hip_string_set* patterns, const nvbio_hip_string_set* texts,
    uint32_t max_pattern_len, uint32_t max_text_len, uint32_t n,
    int32_t* out_score, uint32_t* out_sink, uint32_t* out_source,
    uint16_t* out_cigar, uint32_t cigar_stride, uint32_t* out_cigar_len,
    void* temp, uint64_t temp_bytes, void* stream);

/* nvbio_hip_gotoh_traceback with nvBowtie's quality-aware scheme (its opposite-mate traceback, traceback_inl.h): mismatch
 * penalties from the quality bytes, pattern gap costs in the recurrences, text gap costs on the column before the pattern. */
int nvbio_hip_gotoh_traceback_qual(
    const nvbio_hip_gotoh_qual_scheme* scheme /* host */, int32_t type,
    const nvbio_hip_string_set* patterns, const uint8_t* quals, uint64_t n_quals, const nvbio_hip_string_set* texts,
    uint32_t max_pattern_len, uint32_t max_text_len, uint32_t n,
    int32_t* out_score, uint32_t* out_sink, uint32_t* out_source,
    uint16_t* out_cigar, uint32_t cigar_stride, uint32_t* out_cigar_len,
    void* temp, uint64_t temp_bytes, void* stream);

/* The two full-matrix tracebacks for callers that know, per job, the score of the best alignment over the job's text AND that it ends
 * at the text's last symbol -- nvBowtie's opposite-mate tracebacks, whose windows are [alignment, alignment + sink) of a scoring pass
 * (traceback_inl.h:833-905).  The text rows no alignment of that score can reach (all but the last M + gaps-the-score-allows) are
 * dropped before any DP runs; score, sink, source and CIGAR are those of the plain forms (argument: full_traceback.hip,
 * crop_windows_kernel).  known_score: device int32[n].  The premise is checked per job on what the cropped DP found (score ==
 * known_score and the sink on the window's last row); a job that fails the check -- a known_score that is too high cuts the window too
 * short -- is traced again over its whole window, so its outputs are those of the plain form (one 4-byte read-back per call finds the
 * count; nvbio_hip_known_score_redone() totals it).  Under a TRUE premise the result is the plain form's in every case, ties included: a
 * tied cell the cropped DP visits first lies off the last row, fails the check, and the job is traced again.  Not detectable under a
 * FALSE premise: a window whose plain-order best alignment (of score known_score, or better in the dropped rows) is not the one the kept
 * rows hold with exactly known_score ending at the last row -- the caller then gets that other alignment.  GLOBAL: same as the plain forms.
 * Unlike the plain forms, these two BLOCK the host: the per-job check is read back (one device-to-host copy and a synchronisation of
 * `stream`), and a batch with failed checks allocates and frees a redo buffer before returning. */
int nvbio_hip_gotoh_traceback_known_score(
    const nvbio_hip_gotoh_scheme* scheme /* host */, int32_t type,
    const nvbio_hip_string_set* patterns, const nvbio_hip_string_set* texts, const int32_t* known_score,
    uint32_t max_pattern_len, uint32_t max_text_len, uint32_t n,
    int32_t* out_score, uint32_t* out_sink, uint32_t* out_source,
    uint16_t* out_cigar, uint32_t cigar_stride, uint32_t* out_cigar_len,
    void* temp, uint64_t temp_bytes, void* stream);
int nvbio_hip_gotoh_traceback_qual_known_score(
    const nvbio_hip_gotoh_qual_scheme* scheme /* host */, int32_t type,
    const nvbio_hip_string_set* patterns, const uint8_t* quals, uint64_t n_quals, const nvbio_hip_string_set* texts, const int32_t* known_score,
    uint32_t max_pattern_len, uint32_t max_text_len, uint32_t n,
    int32_t* out_score, uint32_t* out_sink, uint32_t* out_source,
    uint16_t* out_cigar, uint32_t cigar_stride, uint32_t* out_cigar_len,
    void* temp, uint64_t temp_bytes, void* stream);

uint64_t nvbio_hip_known_score_redone(void);     /* jobs the two forms above traced again since the library was loaded */

/* The tracebacks for SmithWatermanAligner / EditDistanceAligner (deletion == insertion), banded (sw_banded_inl.h:405-470,
 * 748-800) and full matrix (sw_inl.h:389-396, 475-500, 1660-1700): arguments and temp sizes as the Gotoh forms.  The banded
 * reference context does not mark zero cells, so its LOCAL walk always reaches the first pattern row; reproduced. */
int nvbio_hip_banded_sw_traceback(
    const nvbio_hip_sw_scheme* scheme /* host */, int32_t type, uint32_t band_len,
    const nvbio_hip_string_set* patterns, const nvbio_hip_string_set* texts,
    uint32_t max_pattern_len, uint32_t max_text_len, uint32_t n,
    int32_t* out_score, uint32_t* out_sink, uint32_t* out_source,
    uint16_t* out_cigar, uint32_t cigar_stride, uint32_t* out_cigar_len,
    void* temp, uint64_t temp_bytes, void* stream);
int nvbio_hip_sw_traceback(
    const nvbio_hip_sw_scheme* scheme /* host */, int32_t type,
    const nvbio_hip_string_set* patterns, const nvbio_hip_string_set* texts,
    uint32_t max_pattern_len, uint32_t max_text_len, uint32_t n,
    int32_t* out_score, uint32_t* out_sink, uint32_t* out_source,
    uint16_t* out_cigar, uint32_t cigar_stride, uint32_t* out_cigar_len,
    void* temp, uint64_t temp_bytes, void* stream);

/* nvbio_hip_alignment_score for GotohAligner<TYPE, SmithWatermanScoringScheme<...>, algorithm_tag>: nvBowtie's full-matrix
 * scoring of the opposite mate (score_opposite_inl.h:266-269).  Mismatch penalties come from the quality bytes (as in
 * nvbio_hip_banded_gotoh_score_qual); the recurrences use the pattern gap costs, and one boundary line of the matrix is
 * initialised with the text gap costs -- which one depends on the tag, exactly as gotoh_inl.h:82-88 / :693-697 / :1171-1175
 * have it.  16-bit sweep only (801 otherwise). */
int nvbio_hip_alignment_score_qual(
    const nvbio_hip_gotoh_qual_scheme* scheme /* host */, int32_t algorithm, int32_t type,
    const nvbio_hip_string_set* patterns, const uint8_t* quals, uint64_t n_quals, const nvbio_hip_string_set* texts,
    uint32_t max_pattern_len, uint32_t max_text_len, const int32_t* min_score /* device, nullable */,
    uint32_t n, int32_t* out_score, uint32_t* out_sink, uint8_t* out_ok /* nullable */, void* stream);

/* ... with job control, for batches whose size only the device knows.  wave_form = 0: the same dispatch, run only when *gate > gate_limit
 * (gate == NULL: always).  wave_form = 1: ONE job per wave (64 lanes x ceil(M / 64) rows: the shortest sweep a job can have) over a list --
 * entry k < *n_on_device is job job_index[k] of the arrays and writes its outputs there -- run only when *gate <= gate_limit.  A driver queues
 * both with the same gate (the count of live jobs): many jobs take the throughput kernels, a few take the low-latency form, and the host never
 * waits for the count.  16-bit sweep only (801 otherwise). */
int nvbio_hip_alignment_score_qual_jobs(
    const nvbio_hip_gotoh_qual_scheme* scheme /* host */, int32_t algorithm, int32_t type,
    const nvbio_hip_string_set* patterns, const uint8_t* quals, uint64_t n_quals, const nvbio_hip_string_set* texts,
    uint32_t max_pattern_len, uint32_t max_text_len, const int32_t* min_score /* device, nullable */,
    uint32_t n, const uint32_t* n_on_device, const uint32_t* job_index, const uint32_t* gate, uint32_t gate_limit, int32_t wave_form,
    int32_t* out_score, uint32_t* out_sink, uint8_t* out_ok /* nullable */, void* stream);

/* nvbio::fm_index<rank_dictionary<2,64,PackedStream<.,uint8,2,true>,.,.>, SSA_index_multiple_context<SA_INT>, const uint32*>
 * (nvbio/fmindex/fmindex.h:341-387) in the production interleaved layout
 * (nvbio/io/fmindex/fmindex.h:159-174, fmindex_impl.cu:305-327):
 *   bwt_occ  8 x uint32 per 64 BWT symbols: 4 words of big-endian 2-bit BWT,
 *            then #A,#C,#G,#T before the block
 *   ssa      ssa[k] = SA[k*sa_int], ssa[0] = 0xFFFFFFFF (nvbio/fmindex/ssa_inl.h:263-309)
 * This struct lives on the host; bwt_occ / ssa point to device memory. */
typedef struct nvbio_hip_fmindex {
    uint32_t        length;
    uint32_t        primary;
    uint32_t        L2[5];
    uint32_t        sa_int;     /* any power of two; 16 in the reference's files.  With 288 GB of HBM
                                   a denser SSA (down to 1 = the full SA) is affordable and shortens
                                   locate's LF walk from ~sa_int steps to none; results are identical */
    const uint32_t* bwt_occ;
    const uint32_t* ssa;
    const uint32_t* ktab;       /* optional accelerator, NULL = none: the match range (uint2) of every
                                   k-mer, built by nvbio_hip_fm_build_ktab; match() then resolves a
                                   seed's last ktab_k symbols with one lookup instead of ktab_k
                                   backward-search steps -- same ranges, bit for bit */
    uint32_t        ktab_k;
    uint32_t        _pad;
    const uint32_t* dimer;      /* optional, NULL = none: the MI355X line-native index built by
                                   nvbio_hip_fm_build_dimer_index (128-byte records holding a two-symbol BWT).
                                   When attached, match / the seed mappers / locate consume two symbols
                                   (two text positions) per 128-byte HBM line instead of one -- same SA ranges,
                                   iterators and positions, bit for bit.  Set by nvbio_hip_fm_attach_dimer_index,
                                   which also fills the constants below from the buffer's header */
    uint32_t        dimer_p1, dimer_fill1;
    uint32_t        dimer_S[4], dimer_T[4];
    const uint32_t* trimer;     /* optional, NULL = none; needs `dimer`: the three-symbol rank arrays built by
                                   nvbio_hip_fm_build_trimer_index (10.7 B per SA row).  Backward search then consumes three
                                   symbols per step -- same ranges, bit for bit */
} nvbio_hip_fmindex;

/* Replaces nvbio::rank(fmi, k, c) (nvbio/fmindex/fmindex_inl.h:36-57) over n
 * independent point queries: out[i] = rank(fmi, k[i], c[i]). */
int nvbio_hip_fm_rank(const nvbio_hip_fmindex* fmi, const uint32_t* k, const uint8_t* c,
                      uint32_t n, uint32_t* out, void* stream);

/* Replaces nvbio::rank4(fmi, k) (fmindex_inl.h:111-135): out[4i..4i+3]. */
int nvbio_hip_fm_rank4(const nvbio_hip_fmindex* fmi, const uint32_t* k,
                       uint32_t n, uint32_t* out, void* stream);

/* Replaces nvbio::rank(fmi, range, c) (fmindex_inl.h:66-99): range/out are uint2 arrays. */
int nvbio_hip_fm_rank_range(const nvbio_hip_fmindex* fmi, const uint32_t* range, const uint8_t* c,
                            uint32_t n, uint32_t* out, void* stream);

/* Replaces nvbio::match(fmi, pattern, len) (fmindex_inl.h:280-341) and
 * nvBowtie's match_range (nvBowtie/bowtie2/cuda/mapping_inl.h:83-97) over a
 * string-set: out_range[2i..2i+1] = inclusive SA range of seeds[i], (1,0) if
 * the seed holds a symbol > 3. */
int nvbio_hip_fm_match(const nvbio_hip_fmindex* fmi, const nvbio_hip_string_set* seeds,
                       uint32_t n, uint32_t* out_range, void* stream);

/* The line-native two-symbol index (layout: nvbio_amd/csrc/fmindex_dimer.h).  The reference's index spends one
 * 32-byte record per symbol and range end (nvbio/io/fmindex/fmindex_impl.cu:305-322), which on MI355X is one
 * 128-byte fabric request each; this one answers a backward-search step for TWO pattern symbols, or two LF steps
 * of locate, from one 128-byte record.  Built on the device from `fmi` (its bwt_occ; dimer/ktab fields ignored):
 *   out_dimer  nvbio_hip_fm_dimer_index_bytes(length) bytes, 128-byte aligned  (128 B header + 1 B per SA row)
 *   temp       nvbio_hip_fm_build_dimer_index_temp_bytes(length) bytes
 * nvbio_hip_fm_attach_dimer_index copies the header's constants into *fmi and sets fmi->dimer (synchronises
 * `stream` once; pass dimer = NULL to detach); it fails with hipErrorInvalidValue if the buffer was built
 * from a different index. */
uint64_t nvbio_hip_fm_dimer_index_bytes(uint32_t length);
uint64_t nvbio_hip_fm_build_dimer_index_temp_bytes(uint32_t length);
int nvbio_hip_fm_build_dimer_index(const nvbio_hip_fmindex* fmi, uint32_t* out_dimer, void* temp, uint64_t temp_bytes, void* stream);
int nvbio_hip_fm_attach_dimer_index(nvbio_hip_fmindex* fmi, const uint32_t* dimer, void* stream);

/* The three-symbol rank arrays on top of the two-symbol index: for each of the 64 trimers "abc" an array of 16-byte records
 * {C3[abc] + #rows before the record whose three preceding text symbols are abc, 96-bit mask of the record's rows holding abc},
 * so that one backward-search step consumes three pattern symbols for one 16-byte load per range end.  Built on the device
 * from `fmi`'s bwt_occ; out_trimer: nvbio_hip_fm_trimer_index_bytes(length) bytes, 128-byte aligned (a 128-byte header, then the
 * arrays); temp: nvbio_hip_fm_build_trimer_index_temp_bytes(length).  Attach with nvbio_hip_fm_attach_trimer_index, which checks the
 * header against `fmi` (magic, length, primary, stride, L2 check word) and that fmi->dimer is attached; NULL detaches. */
uint64_t nvbio_hip_fm_trimer_index_bytes(uint32_t length);
uint64_t nvbio_hip_fm_build_trimer_index_temp_bytes(uint32_t length);
int nvbio_hip_fm_build_trimer_index(const nvbio_hip_fmindex* fmi, uint32_t* out_trimer, void* temp, uint64_t temp_bytes, void* stream);
int nvbio_hip_fm_attach_trimer_index(nvbio_hip_fmindex* fmi, const uint32_t* trimer, void* stream);

/* ---- `_host` twins: the reference's HostThreadScheduler / host paths (batched_banded_inl.h:97-128,
 * batched_inl.h:236-300, the host fm_index functions) behind the same argument lists with HOST pointers everywhere and a
 * thread count (0 = OpenMP's default) instead of a stream.  Strings may be 2-, 4- or 8-bit packed.  Explicit entry points
 * for CPU-side callers; the device entry points never fall back to them. */
int nvbio_hip_banded_gotoh_score_host(const nvbio_hip_gotoh_scheme* scheme, int32_t type, uint32_t band_len,
                                      const nvbio_hip_string_set* patterns, const nvbio_hip_string_set* texts,
                                      uint32_t n, int32_t* out_score, uint32_t* out_sink, int32_t n_threads);
int nvbio_hip_banded_sw_score_host(const nvbio_hip_sw_scheme* scheme, int32_t type, uint32_t band_len,
                                   const nvbio_hip_string_set* patterns, const nvbio_hip_string_set* texts,
                                   uint32_t n, int32_t* out_score, uint32_t* out_sink, int32_t n_threads);
int nvbio_hip_alignment_score_host(int32_t aligner, int32_t algorithm, const int32_t* scheme4, int32_t type,
                                   const nvbio_hip_string_set* patterns, const nvbio_hip_string_set* texts, const int32_t* min_score /* nullable */,
                                   uint32_t n, int32_t* out_score, uint32_t* out_sink, uint8_t* out_ok /* nullable */, int32_t n_threads);
int nvbio_hip_fm_rank_host(const nvbio_hip_fmindex* fmi, const uint32_t* k, const uint8_t* c, uint32_t n, uint32_t* out, int32_t n_threads);
int nvbio_hip_fm_match_host(const nvbio_hip_fmindex* fmi, const nvbio_hip_string_set* seeds, uint32_t n, uint32_t* out_range, int32_t n_threads);
int nvbio_hip_fm_locate_host(const nvbio_hip_fmindex* fmi, const uint32_t* sa_rows, uint32_t n, uint32_t* out_pos, int32_t n_threads);

/* Builds the optional k-mer table for `fmi` (fmi->ktab is ignored): out_ktab[2c..2c+1] =
 * match(fmi, kmer c), where kmer c has symbol t (0 = first) at bits [2t,2t+2) of c, for all
 * 4^k codes; 1 <= k <= 16; out_ktab holds 2*4^k words (k=12: 128 MiB, k=14: 2 GiB, k=15: 8 GiB, k=16: 32 GiB). */
int nvbio_hip_fm_build_ktab(const nvbio_hip_fmindex* fmi, uint32_t k, uint32_t* out_ktab, void* stream);
/* The suffix array sampled every sa_int_out rows (a power of two <= fmi->sa_int; 1 = the whole array), derived on the device from `fmi`'s own
 * sampled array: out_ssa[k] = locate(k * sa_int_out), nvbio_hip_fm_dense_ssa_entries(length, sa_int_out) = (length + sa_int_out) / sa_int_out
 * words (out_ssa[0] = 0xFFFFFFFF like every SSA, ssa_inl.h:263-309).  An index carrying it (ssa = out_ssa, sa_int = sa_int_out) locates with
 * at most sa_int_out - 1 LF steps -- none at 1 -- and returns the same positions: what 288 GB of HBM are for (12 GB at 3 Gbp). */
uint64_t nvbio_hip_fm_dense_ssa_entries(uint32_t length, uint32_t sa_int);
int nvbio_hip_fm_build_dense_ssa(const nvbio_hip_fmindex* fmi, uint32_t sa_int_out, uint32_t* out_ssa, void* stream);

/* nvBowtie's seeding parameters as map_queues_kernel reads them (nvBowtie/bowtie2/cuda/params.h:100-120). */
typedef struct nvbio_hip_map_params {
    uint32_t seed_len;       /* params.seed_len (22 end-to-end / 20 local)                       */
    uint32_t min_read_len;   /* shorter reads get no hits                                        */
    uint32_t max_hits;       /* per-read cap: beyond it the largest SA ranges are dropped (100)  */
    uint32_t max_reseed;     /* retry_stride = seed_freq / (max_reseed + 1)                      */
    uint32_t retry;          /* re-seeding round: seeds start at retry * retry_stride            */
    uint32_t rep_seeds;      /* reseed[] = no hit, or mean range size >= rep_seeds               */
    uint32_t fw, rc;         /* strands to map                                                   */
} nvbio_hip_map_params;

/* Replaces nvBowtie's exact seed mapping stage, map_queues_kernel<EXACT_MAPPING>
 * (nvBowtie/bowtie2/cuda/mapping_inl.h:511-592 with seed_mapper<EXACT_MAPPING>, :229-312, and
 * match_range, :83-97): for queue entry id (read in_queue[id], or id if in_queue is NULL) every seed
 * window of the read is searched on both strands and each non-empty SA range is stored as a SeedHit
 * word pair {range_begin; range_delta:20, pos_in_read:10, rc:1, index_dir:1} (seed_hit.h:54-223,
 * exclusive range) at out_hits[read * hits_stride + k], k < out_counts[read].  seed_freq_by_len[L] is
 * the host-tabulated params.seed_freq(L) (SimpleFunc, func.h:39-70) for every read length L that
 * occurs.  out_reseed (nullable) receives the reference's re-seeding predicate per queue entry. */
int nvbio_hip_map_exact(const nvbio_hip_fmindex* fmi, const nvbio_hip_string_set* reads,
                        const uint32_t* in_queue, uint32_t n, const nvbio_hip_map_params* params /* host */,
                        const uint32_t* seed_freq_by_len /* device */,
                        uint64_t* out_hits, uint32_t hits_stride, uint32_t* out_counts, uint8_t* out_reseed, void* stream);

/* nvBowtie's seed mapping algorithms (mapping_inl.h:118-123; chosen by map_t, :809-843, from
 * params.allow_sub / params.subseed_len). */
enum { NVBIO_HIP_EXACT_MAPPING = 0, NVBIO_HIP_APPROX_MAPPING = 1, NVBIO_HIP_CASE_PRUNING_MAPPING = 2 };

/* Replaces map_queues_kernel<ALGO> for all three algorithms (mapping_inl.h:511-592):
 *   EXACT         as nvbio_hip_map_exact (rfmi, subseed_len ignored)
 *   APPROX        seed_mapper<APPROX_MAPPING> (:318-365): per seed, map<CHECK_EXACT> of the stored seed and
 *                 map<IGNORE_EXACT> of its reversed complement on the forward index -- exact in the first
 *                 subseed_len scan symbols, one substitution (enumerated with rank4) in the rest
 *   CASE_PRUNING  seed_mapper<CASE_PRUNING_MAPPING> (:372-428): the four half-exact / half-one-mismatch
 *                 searches over the forward index fmi and the index of the reversed genome rfmi
 * with map<> as in mapping_inl.h:124-223 (N handling included).  Hits are SeedHit word pairs as in
 * nvbio_hip_map_exact, index_dir (bit 31) = 1 for hits found on rfmi.  Seeds up to 32 symbols.
 * A read's hits are stored in the array order of the reference's hit deque (see the selection stage below). */
int nvbio_hip_map(int32_t algorithm, uint32_t subseed_len,
                  const nvbio_hip_fmindex* fmi, const nvbio_hip_fmindex* rfmi /* CASE_PRUNING only */,
                  const nvbio_hip_string_set* reads, const uint32_t* in_queue, uint32_t n,
                  const nvbio_hip_map_params* params /* host */, const uint32_t* seed_freq_by_len /* device */,
                  uint64_t* out_hits, uint32_t hits_stride, uint32_t* out_counts, uint8_t* out_reseed, void* stream);

/* nvBowtie's reduction of extension results to the best two alignments per read: score_reduce_kernel
 * (nvBowtie/bowtie2/cuda/reduce_inl.h:71-160).  An io::Alignment (nvbio/io/alignments.h:80-131) is a uint64:
 * low word {score_sgn:1, score:17, ed:10, rc:1, mate:1, paired:1, discordant:1}, high word m_align;
 * best_alignments[read] is the best, best_alignments[read + best_stride] the second best (initialised by
 * nvbio_hip_init_alignments).  Active read t (read id read_ids[t], or t) owns the extension results
 * [hit_begin[t], hit_begin[t+1]) of hit_score / hit_loc / hit_rc, walked in that order: results at an already
 * recorded location are skipped, a higher score becomes the best (the old best the second), and a score above
 * the second best replaces it only if distinct_alignments(best, hit, read_len/2) (alignments_inl.h:35-47).
 * The context hooks of the reference (hit statistics / extension give-up counters) are not part of it. */
uint64_t nvbio_hip_alignment_invalid(void);
/* init_alignments (nvBowtie/bowtie2/cuda/aligner.h:323-366): best and second best of every read start unaligned
 * (pos -1, ed 255) with score worst_score_by_len[read length] = the scheme's threshold score (device table). */
int nvbio_hip_init_alignments(uint32_t n_reads, const uint32_t* read_len /* nullable */, uint32_t fixed_read_len,
                              const int32_t* worst_score_by_len /* device */, uint32_t mate,
                              uint64_t* best_alignments, uint32_t best_stride, void* stream);
int nvbio_hip_score_reduce(uint32_t n_active, const uint32_t* read_ids /* nullable */, const uint64_t* hit_begin,
                           const int32_t* hit_score, const uint32_t* hit_loc, const uint8_t* hit_rc,
                           const uint32_t* read_len /* by read id, nullable */, uint32_t fixed_read_len,
                           uint64_t* best_alignments, uint32_t best_stride, void* stream);

/* ---- nvBowtie hit selection and the per-round stages of its best-approx extension loop --------------------------
 * (driver: Aligner::best_approx_score, nvBowtie/bowtie2/cuda/aligner_best_approx.h:522-840.)
 *
 * The hit arena written by nvbio_hip_map* IS the reference's SeedHitDequeArray content: a read's SeedHits sit at
 * hits[read_id * hits_stride] in the array order of its priority_deque (interval heap, nvbio/basic/interval_heap.h;
 * slot 0 = a largest range, top() = slot 1, or slot 0 when alone), hit_counts[read_id] is the deque size.  The
 * probability tree of the randomized selection (SumTree<float*>, nvbio/basic/sum_tree.h) of read r lives at
 * probs[r * probs_stride] as its LEAVES: leaf i (i < hit_counts[r]) at probs[r * probs_stride + i].  Every other node of a SumTree is
 * the float sum of its two children, so the kernels rebuild them on chip -- bit for bit the values the reference keeps in memory -- and
 * a pick that exhausts a hit writes one zero instead of a leaf and four ancestors (the stage is bound by the lines it moves).
 * probs_stride >= hits_stride (a multiple of 4 lets the kernels use 16-byte accesses); rows wider than 32 hit slots are handled one
 * lane per read with the sums in the row itself, as scratch: probs_stride >= nvbio_hip_sum_tree_node_count(hits_stride) there.
 * Active reads are packed_read words (defs.h:152-162: read_id:31, top_flag:1); a selected hit is {read_id, loc, seed}
 * with seed a packed_seed word (defs.h:171-181: pos_in_read:12, index_dir:1, rc:1, top_flag:1).
 * Floating point: single precision, every operation rounded separately (the host-compiled arithmetic). */
uint32_t nvbio_hip_sum_tree_node_count(uint32_t size);
/* select_init_kernel (select.cu:36-103): trys[r] = max_effort_init (if trys); when randomized, rseeds[r] = hash of the
 * read's name (names NUL-terminated in read_names at read_names_idx[r]; with read_names NULL rseeds is left as the
 * caller set it), probs = the leaves: 1 / range_size^2 per hit (the first 0 if top_seed). */
int nvbio_hip_select_init(uint32_t n_reads, const char* read_names, const uint32_t* read_names_idx,
                          const uint64_t* hits, uint32_t hits_stride, const uint32_t* hit_counts,
                          float* probs, uint32_t probs_stride, uint32_t* trys /* nullable */, uint32_t* rseeds,
                          uint32_t max_effort_init, int32_t randomized, int32_t top_seed, void* stream);
/* The same for the reads of a queue only (queue: device uint32[n_queue] read ids): a re-seeding pass gives hits to its queue's reads
 * alone, and no read outside it is selected from before the next select_init reaches it, so the state the reference's whole-batch
 * kernel writes for the others is never read. */
int nvbio_hip_select_init_queued(uint32_t n_queue, const uint32_t* queue, const char* read_names, const uint32_t* read_names_idx,
                                 const uint64_t* hits, uint32_t hits_stride, const uint32_t* hit_counts,
                                 float* probs, uint32_t probs_stride, uint32_t* trys /* nullable */, uint32_t* rseeds,
                                 uint32_t max_effort_init, int32_t randomized, int32_t top_seed, void* stream);
/* select (select_inl.h:74-607; dispatch :744-795): one round of hit selection for the reads of active_in.  Reads whose
 * try counter is 0 or whose deque is empty leave the queue; the others hand out up to n_multi SA rows each (n_multi == 1:
 * select_kernel / rand_select_kernel, > 1: the *_multi_kernel forms), from the deque's top range downwards, or -- randomized
 * -- from hits sampled in proportion to probs with the read's LCG.  Outputs, in the order of active_in (the reference's
 * slot order is atomic-dependent): active_out[n_out], the selected hits grouped per read with hit_begin[n_out + 1]
 * offsets, out_sizes = {n_out, n_hits} (device).  active_out / hit_begin need n_active (+1) entries, the hit arrays
 * n_active * n_multi.  hits / hit_counts / probs / rseeds are updated in place.  n_multi <= 4096. */
uint64_t nvbio_hip_select_temp_bytes(uint32_t n_active, uint32_t n_multi);
int nvbio_hip_select(int32_t randomized, uint32_t n_multi, const uint32_t* active_in, uint32_t n_active,
                     uint64_t* hits, uint32_t hits_stride, uint32_t* hit_counts,
                     float* probs, uint32_t probs_stride, uint32_t* rseeds, const uint32_t* trys,
                     uint32_t* active_out, uint64_t* hit_begin, uint32_t* hit_read_id, uint32_t* hit_loc, uint32_t* hit_seed,
                     uint32_t* out_sizes, void* temp, uint64_t temp_bytes, void* stream);
/* Diagnostic: runs programs of deque operations (ops[i]: 0 = push vals[i], dropping the bottom first when the deque
 * holds caps[i] hits; 1 = pop_top; 2 = pop_bottom; program c = [case_start[c], case_start[c+1])) through the device's
 * hit deque and writes the deque's array after every operation from out_states[state_start[c]] on.  scratch: n_cases *
 * scratch_stride words.  Used to replay vectors recorded from the reference's interval heap. */
int nvbio_hip_hit_deque_replay(uint32_t n_cases, const uint32_t* case_start, const uint8_t* ops, const uint64_t* vals, const uint32_t* caps,
                               const uint64_t* state_start, uint64_t* scratch, uint32_t scratch_stride, uint64_t* out_states, void* stream);
/* locate_kernel (locate_inl.h:53-143): hit_loc[i] (an SA row of fmi, or of rfmi when the seed's index_dir is REVERSE)
 * becomes the read's start in genome coordinates: locate - pos_in_read, mirrored (length-1-locate) for rfmi; uint32
 * arithmetic, so a read hanging over the genome start wraps.  rfmi may be NULL when no hit uses it. */
int nvbio_hip_locate_hits(const nvbio_hip_fmindex* fmi, const nvbio_hip_fmindex* rfmi, uint32_t n,
                          uint32_t* hit_loc, const uint32_t* hit_seed, void* stream);
/* BestScoreStream::init_context (score_best_inl.h:95-126): per hit, the genome window [loc - band/2 (clamped at 0),
 * + band + read_len (clamped at genome_length)), the threshold max(second best score of the read, score_limit) and
 * the pattern (read r's forward copy at read_begin[r] -- or r * fixed_read_len -- or its reverse complement rc_offset
 * symbols further, by the seed's rc bit).  A window beginning at or beyond its end gets length 0 (the reference
 * would read out of bounds); the alignment then fails like any text shorter than its pattern. */
int nvbio_hip_score_best_setup(uint32_t n_hits, const uint32_t* hit_read_id, const uint32_t* hit_loc, const uint32_t* hit_seed,
                               const uint64_t* read_begin /* nullable */, const uint32_t* read_len /* nullable */, uint32_t fixed_read_len,
                               uint64_t rc_offset, uint32_t band_len, uint32_t genome_length,
                               const uint64_t* best_alignments, uint32_t best_stride, int32_t score_limit,
                               uint64_t* pattern_begin, uint32_t* pattern_len /* nullable iff fixed */,
                               uint64_t* text_begin, uint32_t* text_len, int32_t* min_score,
                               int32_t* known_score /* nullable, see below */,
                               uint32_t* job_count /* nullable, device */, uint32_t* job_hit /* nullable, n_hits */, void* stream);
/* known_score (optional, a pure saving): a hit whose (strand, read start) equals one of the read's two recorded alignments would be
 * scored over the same window again and get the recorded score.  With known_score != NULL such hits get known_score[i] = that score
 * and an empty window (INT32_MIN marks the others); handing the array to nvbio_hip_score_reduce_best_approx makes the reduction use
 * it in place of the (absent) DP result, so every output is what the reference computes by re-running the DP.
 * job_count / job_hit (optional, both or neither, need known_score): compacted jobs.  Only the hits that still need a DP are written,
 * at slots 0..*job_count-1 of pattern_begin / pattern_len / text_begin / text_len / min_score, with job_hit[slot] = the hit; run the DP
 * over *job_count jobs, scatter its scores into known_score (nvbio_hip_scatter_rows(n_jobs, job_hit, scores, known_score, 4)) and
 * hand known_score to the reduction as hit_score.  The slot order is not deterministic; everything indexed by hit is. */
/* score_reduce_kernel with ReduceBestApproxContext (reduce_inl.h:71-160, reduce.h:63-105): nvbio_hip_score_reduce over
 * packed active reads and packed seeds, plus the give-up counters: see reduce.hip.  hit_score is the raw DP score
 * (clamped to worst_score = scheme_type::worst_score here); n_ext = extensions done before this round. */
int nvbio_hip_score_reduce_best_approx(uint32_t n_active, const uint32_t* active_reads, const uint64_t* hit_begin,
                                       const int32_t* hit_score, const uint32_t* hit_loc, const uint32_t* hit_seed,
                                       const uint32_t* read_len /* nullable */, uint32_t fixed_read_len,
                                       uint64_t* best_alignments, uint32_t best_stride, int32_t worst_score,
                                       uint32_t* trys, uint32_t* hit_counts,
                                       uint32_t n_ext, uint32_t min_ext, uint32_t max_ext, uint32_t max_effort,
                                       const int32_t* known_score /* nullable: from nvbio_hip_score_best_setup */,
                                       const uint32_t* hit_sink /* nullable uint2[n_hits]: the DP sinks */, uint32_t* best_sink /* nullable uint2[n_reads] */,
                                       void* stream);
/* hit_sink / best_sink (optional, both or neither): whenever a hit becomes a read's best alignment its DP sink is kept in best_sink[read].
 * nvbio_hip_traceback_best_known then lays out, per traceback job, the score and sink the banded scorer would report over the job's
 * window (the same DP the extension ran; unaligned entries: a failed alignment, score -2^30, sink (-1,-1)), and
 * nvbio_hip_banded_gotoh_traceback_qual_known starts from them instead of scoring every job again.  Results are the traceback's. */
int nvbio_hip_traceback_best_known(uint32_t n, const uint32_t* idx /* nullable */, const uint64_t* best_alignments, const uint32_t* best_sink /* nullable */,
                                   int32_t* out_score, uint32_t* out_sink /* nullable with best_sink: scores only */, void* stream);

/* The paired-end form: score_reduce_paired_kernel (reduce_inl.h:355-500).  Per extension result the anchor mate's
 * {loc, sink (genome end), score, rc} and the opposite mate's {loc, sink, sink2, score, score2} (the stream's hit.* fields,
 * score_opposite_inl.h:203-235); the pair is `paired` when the opposite score exceeds score_limit.  best_alignments holds
 * the anchor (or, while no pair was found and pe_unpaired, mate 1) entries, best_alignments_o the opposite (mate 2) ones.
 * anchor: 0 / 1; pe_policy: io::PairedEndPolicy (FF 0, FR 1, RF 2, RR 3). */
int nvbio_hip_score_reduce_paired(uint32_t n_active, const uint32_t* read_ids /* nullable */, const uint64_t* hit_begin,
    const uint32_t* hit_loc, const uint32_t* hit_sink, const int32_t* hit_score, const uint8_t* hit_rc,
    const uint32_t* opposite_loc, const uint32_t* opposite_sink, const uint32_t* opposite_sink2,
    const int32_t* opposite_score, const int32_t* opposite_score2,
    const uint32_t* read_len /* nullable */, uint32_t fixed_read_len, uint32_t anchor, int32_t pe_policy, int32_t pe_unpaired, int32_t score_limit,
    uint64_t* best_alignments, uint64_t* best_alignments_o, uint32_t best_stride, void* stream);

/* The paired-end parameters the opposite-mate stage reads (nvBowtie/bowtie2/cuda/params.h: pe_policy, min_frag_len,
 * max_frag_len, pe_overlap) plus the pipeline's score_limit, anchor (0 / 1) and genome_length. */
typedef struct nvbio_hip_pe_params {
    int32_t  pe_policy, min_frag_len, max_frag_len, pe_overlap, score_limit;
    uint32_t anchor, genome_length;
} nvbio_hip_pe_params;

/* BestOppositeScoreStream::init_context (nvBowtie/bowtie2/cuda/score_opposite_inl.h:92-200) for every scored anchor hit:
 * the opposite mate's score threshold (compute_target_score, alignment_utils.h:100-111, clamped by the mate's own worst score
 * and score_limit), its strand (frame_opposite_mate), and the genome window [begin, end) allowed by the fragment-length limits
 * and max_text_gaps (utils_inl.h:181-204).  out_valid = the function's bool (threshold reachable, window inside the genome and
 * non-empty, location not already recorded in the best pairs).  The windows + thresholds feed nvbio_hip_alignment_score_qual. */
int nvbio_hip_opposite_mate_windows(uint32_t n_hits, const uint32_t* hit_read_id, const uint8_t* hit_rc, const uint32_t* hit_loc, const int32_t* hit_score,
    const uint32_t* a_read_len /* nullable */, const uint32_t* o_read_len /* nullable */, uint32_t a_fixed_len, uint32_t o_fixed_len,
    const uint64_t* best_alignments, const uint64_t* best_alignments_o, uint32_t best_stride,
    int32_t match, const int32_t* min_score_by_len /* device */, int32_t text_gap_open, int32_t text_gap_ext, const nvbio_hip_pe_params* params /* host */,
    uint8_t* out_valid, int32_t* out_min_score, uint8_t* out_read_rc, uint32_t* out_genome_begin, uint32_t* out_genome_end, void* stream);

/* ---- the per-round stages of nvBowtie's PAIRED best-approx loop (Aligner::best_approx_score, aligner_best_approx_paired.h:455-700):
 * select / locate as in the single-end loop, then
 *   anchor_score_best    BestAnchorScoreStream (score_paired_inl.h:54-150): nvbio_hip_anchor_score_setup (window, pattern, threshold
 *                        from the read's best pairs; hits at recorded locations are skipped), the banded scorer, _finish (hit.score =
 *                        score >= threshold ? score : worst_score; hit.sink = window begin + sink.x)
 *   opposite_score_best  BestOppositeScoreStream (score_opposite_inl.h:54-235): nvbio_hip_opposite_score_setup = nvbio_hip_opposite_mate_windows
 *                        over packed seeds, valid only for hits whose anchor score is not worst_score (the opposite queue), the full-matrix
 *                        scorer over the valid hits, _finish (scatter of hit.opposite_* for those; the caller pre-fills worst_score)
 *   score_reduce_paired  with ReduceBestApproxContext: nvbio_hip_score_reduce_paired_best_approx
 * and after the loops nvbio_hip_mark_discordant (aligner_init.cu:457-480).  The reference's skip test of the anchor stream reads
 * context->min_score before setting it (score_paired_inl.h:128); that term is taken as false here. */
int nvbio_hip_anchor_score_setup(uint32_t n_hits, const uint32_t* hit_read_id, const uint32_t* hit_loc, const uint32_t* hit_seed,
    const uint64_t* a_read_begin /* nullable */, const uint32_t* a_read_len /* nullable */, const uint32_t* o_read_len /* nullable */,
    uint32_t a_fixed_len, uint32_t o_fixed_len, uint64_t rc_offset,
    uint32_t band_len, uint32_t genome_length, const uint64_t* best_alignments, const uint64_t* best_alignments_o, uint32_t best_stride,
    int32_t match, const int32_t* min_score_by_len /* device */, int32_t score_limit, uint32_t anchor,
    uint64_t* pattern_begin, uint32_t* pattern_len /* nullable iff fixed */, uint64_t* text_begin, uint32_t* text_len, int32_t* min_score, void* stream);
int nvbio_hip_anchor_score_finish(uint32_t n_hits, const int32_t* raw_score, const uint32_t* raw_sink /* uint2[n] */, const uint64_t* text_begin,
    const int32_t* min_score, int32_t worst_score, int32_t* hit_score, uint32_t* hit_sink, void* stream);
int nvbio_hip_opposite_score_setup(uint32_t n_hits, const uint32_t* hit_read_id, const uint32_t* hit_seed, const uint32_t* hit_loc, const int32_t* hit_score,
    int32_t worst_score, const uint32_t* a_read_len, const uint32_t* o_read_len, uint32_t a_fixed_len, uint32_t o_fixed_len,
    const uint64_t* best_alignments, const uint64_t* best_alignments_o, uint32_t best_stride,
    int32_t match, const int32_t* min_score_by_len /* device */, int32_t text_gap_open, int32_t text_gap_ext, const nvbio_hip_pe_params* params /* host */,
    uint8_t* out_valid, int32_t* out_min_score, uint8_t* out_read_rc, uint32_t* out_genome_begin, uint32_t* out_genome_end,
    /* optional (all NULL / 0 to skip): the full-matrix job of every hit -- pattern = the opposite mate's forward copy at o_read_begin[r]
     * (or r * o_fixed_len), its reverse complement o_rc_offset symbols further; text = the window, empty for invalid hits */
    const uint64_t* o_read_begin, uint64_t o_rc_offset, uint64_t* out_pattern_begin, uint64_t* out_text_begin, uint32_t* out_text_len, void* stream);
/* valid_idx != NULL: raw results of the n_valid scored hits valid_idx[k]; valid_idx == NULL: one raw result per hit (n_valid = n_hits) and
 * valid_flags[h] says whether hit h was scored (the others get worst_score) */
int nvbio_hip_opposite_score_finish(uint32_t n_valid, const uint32_t* valid_idx, const uint8_t* valid_flags, const int32_t* raw_score /* [n_valid] */, const uint32_t* raw_sink /* uint2[n_valid] */,
    const int32_t* min_score /* by hit */, const uint32_t* genome_begin /* by hit */, int32_t worst_score,
    int32_t* opposite_score, int32_t* opposite_score2, uint32_t* opposite_loc, uint32_t* opposite_sink, uint32_t* opposite_sink2, void* stream);
int nvbio_hip_score_reduce_paired_best_approx(uint32_t n_active, const uint32_t* active_reads, const uint64_t* hit_begin,
    const uint32_t* hit_loc, const uint32_t* hit_sink, const int32_t* hit_score, const uint32_t* hit_seed,
    const uint32_t* opposite_loc, const uint32_t* opposite_sink, const uint32_t* opposite_sink2, const int32_t* opposite_score, const int32_t* opposite_score2,
    const uint32_t* read_len /* nullable */, uint32_t fixed_read_len, uint32_t anchor, int32_t pe_policy, int32_t pe_unpaired, int32_t score_limit,
    uint64_t* best_alignments, uint64_t* best_alignments_o, uint32_t best_stride,
    uint32_t* trys, uint32_t* hit_counts, uint32_t n_ext, uint32_t min_ext, uint32_t max_ext, uint32_t max_effort, void* stream);
/* The anchor memo (optional, a pure saving).  The anchor's banded DP is a pure function of (pair, which mate, strand, window); the reference's skip test
 * (aligner_best_approx_paired.h: BestAnchorScoreStream::init_context compares a hit's position with the recorded alignments' window begins) almost
 * never fires, so every seed of a read that points at the placement the read already tried pays the DP again and absorbs the identical score.
 * memo: 6 words per pair, zero-initialised by the caller, holding the last job {window begin, strand, mate} and its raw score and sink.
 *   anchor_memo_mark (after anchor_score_setup, before the DP): text_len_out[i] = text_len[i], or 0 for a hit whose job equals the pair's entry
 *     (from_memo[i] = 1) or the job of the hit before it in this round (from_memo[i] = 2): the scorer's lane returns at once on an empty text.
 *     text_len_out must not alias text_len (the update reads the set-up lengths).  live_count / live_idx (optional): the hits left with a window,
 *     counted and listed on the device -- the job_index / gate of nvbio_hip_banded_gotoh_score_qual_wave for rounds that hold only a few.
 *   anchor_score_finish_memo: nvbio_hip_anchor_score_finish, a marked hit taking its raw score and sink from the entry / the hit it repeats.
 *   anchor_memo_update (after the finish): per active read, the last hit of the round that had a window becomes the pair's entry.
 * Every output equals what re-running the DP gives. */
int nvbio_hip_anchor_memo_mark(uint32_t n_hits, const uint32_t* hit_read_id, const uint32_t* hit_seed, const uint64_t* text_begin, const uint32_t* text_len,
    uint32_t anchor, const uint32_t* memo, uint8_t* from_memo, uint32_t* text_len_out,
    uint32_t* live_count /* nullable; device, zeroed by the call */, uint32_t* live_idx /* nullable with it: the hits that still need their DP */, void* stream);
int nvbio_hip_anchor_score_finish_memo(uint32_t n_hits, const int32_t* raw_score, const uint32_t* raw_sink, const uint64_t* text_begin, const int32_t* min_score,
    int32_t worst_score, const uint8_t* from_memo, const uint32_t* hit_read_id, const uint32_t* memo, int32_t* hit_score, uint32_t* hit_sink, void* stream);
int nvbio_hip_anchor_memo_update(uint32_t n_active, const uint32_t* active_reads, const uint64_t* hit_begin, const uint32_t* hit_read_id, const uint32_t* hit_seed,
    const uint64_t* text_begin, const uint32_t* text_len_setup, const uint8_t* from_memo, const int32_t* raw_score, const uint32_t* raw_sink, uint32_t anchor, uint32_t* memo, void* stream);
/* The opposite-mate memo (optional, a pure saving).  The opposite mate's DP is a pure function of (pair, opposite strand, window,
 * threshold); the reference re-runs it for every anchor hit landing on a placement it already tried and absorbs the identical result
 * (in the second anchor pass that is every seed of every read).  memo: 6 words per pair, zero-initialised by the caller, holding the
 * last scored job and its outputs.
 *   opposite_memo_lookup (after opposite_score_setup, before the DP): a valid hit whose job equals the pair's entry gets its
 *     opposite_* outputs from it, valid[i] = 2 and (if text_len != NULL) an empty text; nvbio_hip_opposite_score_finish leaves such
 *     hits alone (with the valid_idx form, pass only the hits with valid == 1).
 *   opposite_memo_update (after opposite_score_finish): per active read, the last hit with valid == 1 becomes the pair's entry.
 * Every output equals what re-running the DP gives; hits of one pair in the SAME round are not matched against each other. */
/* idx[0 .. *count) = the indices i < n with flags[i] == value (in no particular order); count is zeroed by the call.  The list of jobs a round
 * still has to score, for the job_index forms of the scorers. */
int nvbio_hip_list_flagged(uint32_t n, const uint8_t* flags, uint32_t value, uint32_t* count, uint32_t* idx, void* stream);
int nvbio_hip_opposite_memo_lookup(uint32_t n_hits, const uint32_t* hit_read_id, uint8_t* valid, const uint8_t* read_rc, const uint32_t* genome_begin,
    const uint32_t* genome_end, const int32_t* min_score, uint32_t anchor, const uint32_t* memo, int32_t worst_score,
    int32_t* opposite_score, int32_t* opposite_score2, uint32_t* opposite_loc, uint32_t* opposite_sink, uint32_t* opposite_sink2,
    uint32_t* text_len /* nullable */, void* stream);
int nvbio_hip_opposite_memo_update(uint32_t n_active, const uint32_t* active_reads, const uint64_t* hit_begin, const uint8_t* valid, const uint8_t* read_rc,
    const uint32_t* genome_begin, const uint32_t* genome_end, const int32_t* min_score, const int32_t* opposite_score, const uint32_t* opposite_sink,
    uint32_t anchor, uint32_t* memo, void* stream);
int nvbio_hip_mark_discordant(uint32_t n_reads, uint64_t* best_alignments, uint64_t* best_alignments_o, uint32_t best_stride, void* stream);

/* pack_read(top_seed) over a read queue (nvBowtie/bowtie2/cuda/defs.h:185-205): out[i] = {read_id:31 = queue[i], top_flag:1},
 * the form the active-read queues of the extension rounds hold.  queue == NULL stands for the identity queue 0 .. n-1 (with
 * top_flag 0: the initial seed queue of a batch, written on the device). */
int nvbio_hip_pack_read_queue(uint32_t n, const uint32_t* queue, uint32_t top_flag, uint32_t* out, void* stream);

/* ---- what nvBowtie's host drivers do between the stages (they use thrust / nvbio primitives for it) ----
 * mark_unaligned (aligner_init.cu:421-444): reseed[t] = 1 for every queued read whose best alignment is still unaligned.
 * copy_flagged (nvbio/basic/primitives.h, used at aligner_best_approx.h:273-280): out = the in[i] with flags[i] != 0, in
 * order; *out_count (device) = how many.
 * traceback_best_setup = BestTracebackStream::init_context (traceback_inl.h:104-136) over best_alignments[idx[i]] (or [i]):
 * want 0: every aligned entry, banded window [alignment - band/2, + band + read_len); want 1: concordant entries, the
 * opposite mate's window [alignment, alignment + sink) (opposite_traceback_best); want 2: aligned, not concordant, banded
 * window.  Patterns: the read's copy at read_begin[r] (or r * fixed_read_len), + rc_offset for reverse-complement
 * alignments, + mate_offset when the alignment's mate bit is set.  Entries that do not qualify get out_valid 0 and an
 * empty window. */
int nvbio_hip_mark_unaligned(uint32_t n_active, const uint32_t* active_reads, const uint64_t* best_alignments, uint8_t* reseed, void* stream);
uint64_t nvbio_hip_copy_flagged_temp_bytes(uint32_t n);
int nvbio_hip_copy_flagged(uint32_t n, const uint32_t* in, const uint8_t* flags, uint32_t* out, uint32_t* out_count /* device */,
                           void* temp, uint64_t temp_bytes, void* stream);
int nvbio_hip_scatter_rows(uint32_t n, const uint32_t* idx, const void* src, void* dst, uint32_t row_bytes /* multiple of 4 */, void* stream);   /* dst row idx[i] = src row i */
int nvbio_hip_gather_rows(uint32_t n, const uint32_t* idx, const void* src, void* dst, uint32_t row_bytes /* multiple of 4 */, void* stream);    /* dst row i = src row idx[i] */
int nvbio_hip_traceback_best_setup(uint32_t n, const uint32_t* idx /* nullable */, const uint64_t* best_alignments, uint32_t band_len, uint32_t genome_length,
                                   const uint64_t* read_begin /* nullable */, const uint32_t* read_len /* nullable */, uint32_t fixed_read_len,
                                   uint64_t rc_offset, uint64_t mate_offset, int32_t want,
                                   uint8_t* out_valid, uint64_t* pattern_begin, uint32_t* pattern_len /* nullable iff fixed */,
                                   uint64_t* text_begin, uint32_t* text_len, void* stream);
/* ... for pairs whose mates have their own lengths (the reference's paired driver takes them as they come, aligner_best_approx_paired.h): mate m's
 * reads sit at read_begin[m][r] (or r * fixed_read_len[m]) of ITS half of the pattern stream, reverse complements rc_offset[m] further; mate 1's
 * half starts mate_offset symbols into the stream.  The entry's mate bit picks the half.  pattern_len is always written. */
int nvbio_hip_traceback_best_setup_mates(uint32_t n, const uint32_t* idx /* nullable */, const uint64_t* best_alignments, uint32_t band_len, uint32_t genome_length,
                                         const uint64_t* const read_begin[2] /* host array of device pointers, each nullable */, const uint32_t* const read_len[2],
                                         const uint32_t fixed_read_len[2], const uint64_t rc_offset[2], uint64_t mate_offset, int32_t want,
                                         uint8_t* out_valid, uint64_t* pattern_begin, uint32_t* pattern_len, uint64_t* text_begin, uint32_t* text_len, void* stream);

/* finish_alignment_kernel + BestTracebackStream::finish (nvBowtie/bowtie2/cuda/traceback_inl.h:523-722, :177-189): for job i with
 * valid[i] != 0, replay its CIGAR (cigar[i * cigar_stride ..], cigar_len[i] words stored end first, first text column
 * cigar_source[2i] & 0xFFFF) over pattern i and text i (the window of nvbio_hip_traceback_best_setup) and write
 *   - out_mds[i * mds_stride ..]: the MD string in nvbio's byte code (io::MDS_OP, nvbio/io/alignments.h:46-52): bytes 0-1 its length,
 *     then {MDS_MATCH 0, run <= 255} / {MDS_MISMATCH 1, read symbol} / {MDS_INSERTION 2 | MDS_DELETION 3, length byte, symbols};
 *     out_mds_len[i] its length (bytes beyond mds_stride are counted, not stored);
 *   - the alignment best_alignments[idx ? idx[i] : i] rewritten: m_align = the text's begin, m_ed = mismatches + inserted / deleted
 *     symbols (soft clips excluded), m_score = sum over the SUBSTITUTION columns of scoring_scheme.score (scoring.h:301-311):
 *     -n_penalty when the read symbol is N, else match or mismatch_by_quality[q] (host table of -m_mmp(q)); minus, per INSERTION run of
 *     l read symbols, cumulative_deletion(l) = -(text_gap_open + (l - 1) text_gap_ext), and per DELETION run cumulative_insertion(l) =
 *     -(pattern_gap_open + (l - 1) pattern_gap_ext) (traceback_inl.h:664-665).  gap_costs (host, 4 ints): pattern_gap_open, pattern_gap_ext,
 *     text_gap_open, text_gap_ext as nvbio_hip_gotoh_qual_scheme holds them (<= 0).
 * Jobs with valid 0, no CIGAR, or a CIGAR longer than cigar_stride are skipped (out_mds_len 0, alignment untouched). */
int nvbio_hip_finish_alignment(uint32_t n, const uint8_t* valid, const nvbio_hip_string_set* patterns, const uint8_t* quals /* nullable */, uint64_t n_quals,
                               const nvbio_hip_string_set* texts, const uint16_t* cigar, uint32_t cigar_stride, const uint32_t* cigar_len,
                               const uint32_t* cigar_source /* uint2[n] */, int32_t match, const int32_t* mismatch_by_quality /* host, 256 */, int32_t n_penalty,
                               const int32_t* gap_costs /* host, 4 */, const uint32_t* idx /* nullable */, uint64_t* best_alignments, uint8_t* out_mds, uint32_t mds_stride, uint32_t* out_mds_len, void* stream);

/* ---- all-mapping mode (Aligner::all / score_all, nvBowtie/bowtie2/cuda/aligner_all.h:47-694) ----
 * Every row of every SA range in every read's hit deque is located, de-duplicated, extended, and reported when its score reaches
 * scoring_scheme.min_score(read_len).  The host driver (nvbio_amd.aligner.all_mapping) does the scans, sorts and compactions the
 * reference does with thrust; the device stages are:
 *   gather_ranges (mapping.cu:39-67): out_ranges[t] = the size of SA range t, ranges numbered read by read in deque array order
 *     (hit_count_scan = inclusive scan of the deque sizes).
 *   select_all (select.cu:175-219): hit begin + t, numbered through hit_range_scan (inclusive scan of out_ranges) ->
 *     out_loc = its SA row (range begin + row), out_seed = packed_seed(pos_in_read, index_dir, rc, 0), out_read_id.
 *   mark_straddling (locate_inl.h:213-244): flags[t] = 0 when hit idx_queue[t]'s seed [loc, loc + seed_len] crosses a boundary of
 *     sequence_index (n_sequences + 1 offsets) -- the indexing (hit idx_queue[t], flag t) is the reference's.
 *   score_all_setup (AllScoreStream::init_context, score_all_inl.h:99-127): job i = hit idx[i] (or i): window
 *     [loc - band/2, + band + read_len) clamped to the genome, pattern = the read's copy (+ rc_offset on the reverse strand).
 *   score_all_output (AllScoreStream::output, :131-153): out_flags[i] = score[i] >= min_score_by_len[read_len]; out_alignments[i] =
 *     io::Alignment(loc, 0, score, rc); out_read_id[i] (the reference appends the accepted ones to a ring buffer in atomic order;
 *     here the caller compacts by the flags, in job order).
 *   traceback_all_setup (AllTracebackStream::init_context, traceback_inl.h:361-385): the same windows from alignment words. */
int nvbio_hip_gather_ranges(uint32_t n_ranges, uint32_t n_reads, const uint64_t* hits, uint32_t hits_stride, const uint32_t* hit_count_scan,
                            uint64_t* out_ranges, void* stream);
int nvbio_hip_select_all(uint64_t begin, uint32_t count, uint32_t n_reads, uint32_t n_ranges, const uint64_t* hits, uint32_t hits_stride,
                         const uint32_t* hit_count_scan, const uint64_t* hit_range_scan, uint32_t* out_loc, uint32_t* out_seed, uint32_t* out_read_id,
                         void* stream);
int nvbio_hip_mark_straddling(uint32_t n, const uint32_t* idx_queue, uint32_t n_sequences, const uint32_t* sequence_index, const uint32_t* hit_loc,
                              uint32_t seed_len, uint8_t* flags, void* stream);
int nvbio_hip_score_all_setup(uint32_t n, const uint32_t* idx /* nullable */, const uint32_t* hit_read_id, const uint32_t* hit_loc, const uint32_t* hit_seed,
                              const uint64_t* read_begin /* nullable */, const uint32_t* read_len /* nullable */, uint32_t fixed_read_len, uint64_t rc_offset,
                              uint32_t band_len, uint32_t genome_length,
                              uint64_t* pattern_begin, uint32_t* pattern_len /* nullable iff fixed */, uint64_t* text_begin, uint32_t* text_len, void* stream);
int nvbio_hip_score_all_output(uint32_t n, const uint32_t* idx /* nullable */, const uint32_t* hit_read_id, const uint32_t* hit_loc, const uint32_t* hit_seed,
                               const int32_t* score, const int32_t* min_score_by_len /* device */, const uint32_t* read_len /* nullable */, uint32_t fixed_read_len,
                               uint8_t* out_flags, uint64_t* out_alignments, uint32_t* out_read_id, void* stream);
int nvbio_hip_traceback_all_setup(uint32_t n, const uint64_t* alignments, const uint32_t* read_id,
                                  const uint64_t* read_begin /* nullable */, const uint32_t* read_len /* nullable */, uint32_t fixed_read_len, uint64_t rc_offset,
                                  uint32_t band_len, uint32_t genome_length,
                                  uint64_t* pattern_begin, uint32_t* pattern_len /* nullable iff fixed */, uint64_t* text_begin, uint32_t* text_len, void* stream);

/* The library primitives score_all calls between those kernels (thrust::inclusive_scan, Aligner::sort_hi_bits / sort_64_bits of
 * aligner_sort.cu:39-92, the dedup transform of aligner_all.h:498-509), on hipCUB; temp: nvbio_hip_all_mapping_temp_bytes(n) bytes.
 *   sort_hi_bits: out_idx = the permutation that stably sorts keys[i] >> 16.
 *   sort_hits: out_idx = the permutation that stably sorts SortingKeys = loc + (read_id << 33) + (rc << 32);
 *     out_first[i] = 1 when sorted position i holds the first of a run of equal keys (the dedup flags before mark_straddling). */
uint64_t nvbio_hip_all_mapping_temp_bytes(uint32_t n);
int nvbio_hip_inclusive_scan_u32(uint32_t n, const uint32_t* in, uint32_t* out, void* temp, uint64_t temp_bytes, void* stream);
int nvbio_hip_inclusive_scan_u64(uint32_t n, const uint64_t* in, uint64_t* out, void* temp, uint64_t temp_bytes, void* stream);
int nvbio_hip_sort_hi_bits(uint32_t n, const uint32_t* keys, uint32_t* out_idx, void* temp, uint64_t temp_bytes, void* stream);
int nvbio_hip_sort_hits(uint32_t n, const uint32_t* hit_read_id, const uint32_t* hit_loc, const uint32_t* hit_seed, uint32_t* out_idx, uint8_t* out_first,
                        void* temp, uint64_t temp_bytes, void* stream);
/*   sort_hits_pingpong: sort_hits, and the index Aligner::all's mark_straddling actually reads (aligner_all.h:520): `pipeline.idx_queue` is the
 *     pointer sort_hi_bits returned (:465) -- one half of the ping-pong index buffer, which sort_64_bits (:500) has refilled and sorted through
 *     since.  Both sorts are replayed on one pair of halves as aligner_sort.cu:37-86 makes them; out_stale[n] = what the first sort's half holds
 *     after the second (the final index when both end in the same half, the last pass but one's otherwise). */
int nvbio_hip_sort_hits_pingpong(uint32_t n, const uint32_t* hit_read_id, const uint32_t* hit_loc, const uint32_t* hit_seed, uint32_t* out_idx, uint8_t* out_first,
                                 uint32_t* out_stale, void* temp, uint64_t temp_bytes, void* stream);

/* BowtieMapq2 / BowtieMapq3 (nvBowtie/bowtie2/cuda/mapq.h:42-330) for single-end reads:
 * out_mapq[r] from the best / second-best alignment of read r; perfect_score(len) = len * match,
 * min_score(len) = min_score_by_len[len] (the scheme's SimpleFunc tabulated by the host, scoring.h:272-281),
 * monotone = scheme.m_monotone (match bonus == 0).  Every read is evaluated, aligned or not, as the reference's MapqFunctorSE / PE
 * do: an unaligned read (Alignment::invalid(), score 2^17 - 1) gets the calculator's top no-second value and the reference's writers
 * zero it (output_sam.cpp:462).  version: 2 or 3. */
int nvbio_hip_mapq(int32_t version, int32_t match, int32_t monotone, const int32_t* min_score_by_len /* device */,
                   uint32_t n_reads, const uint64_t* best_alignments, uint32_t best_stride,
                   const uint32_t* read_len /* nullable */, uint32_t fixed_read_len, uint8_t* out_mapq, void* stream);

/* The same calculators over BestPairedAlignments(anchor pair, opposite pair) (mapq.h:56-58, 156-166): when the best alignment
 * is paired, scores and the score range are the sums over both mates (version 2) / the quality is 44 (version 3).  As for single-end
 * reads, an UNALIGNED mate is evaluated too and does not get 0 here (the reference's functors do the same; its writers print 0 for a read
 * they flag unmapped, output_sam.cpp:462): a consumer that wants "unaligned = 0" zeroes it itself, as this repository's SAM writers do
 * (include/nvbio_hip/sam.h, tools/align_fastq.py; tests/test_io_formats.py checks the records of unaligned reads). */
int nvbio_hip_mapq_paired(int32_t version, int32_t match, int32_t monotone, const int32_t* min_score_by_len /* device */,
                          uint32_t n_reads, const uint64_t* best_alignments, const uint64_t* best_alignments_o, uint32_t best_stride,
                          const uint32_t* read_len /* nullable */, const uint32_t* o_read_len /* nullable */,
                          uint32_t fixed_read_len, uint32_t o_fixed_read_len, uint8_t* out_mapq, void* stream);

/* Replaces nvbio::locate(fmi, i) (fmindex_inl.h:466-501) and nvBowtie's
 * locate_kernel (nvBowtie/bowtie2/cuda/locate_inl.h:122-148). */
int nvbio_hip_fm_locate(const nvbio_hip_fmindex* fmi, const uint32_t* sa_rows,
                        uint32_t n, uint32_t* out_pos, void* stream);

/* Two-pass form: locate_ssa_iterator / lookup_ssa_iterator (fmindex_inl.h:511-569;
 * nvBowtie locate_init_kernel / locate_lookup_kernel, locate_inl.h:153-208).
 * out_it / it are uint2 arrays {sampled row, steps}. */
int nvbio_hip_fm_locate_ssa_iterator(const nvbio_hip_fmindex* fmi, const uint32_t* sa_rows,
                                     uint32_t n, uint32_t* out_it, void* stream);
int nvbio_hip_fm_lookup_ssa_iterator(const nvbio_hip_fmindex* fmi, const uint32_t* it,
                                     uint32_t n, uint32_t* out_pos, void* stream);

/* Replaces FMIndexFilter<device_tag,fm_index>::rank(index, string_set)
 * (nvbio/fmindex/filter.h:139-170, filter_inl.h:268-300): out_range = match of
 * every seed, out_slots = inclusive scan of the range sizes (uint64).
 * temp: device scratch of nvbio_hip_fm_filter_temp_bytes(n) bytes.
 * The total number of hits is out_slots[n-1] (left on the device). */
uint64_t nvbio_hip_fm_filter_temp_bytes(uint32_t n);
int nvbio_hip_fm_filter_rank(const nvbio_hip_fmindex* fmi, const nvbio_hip_string_set* seeds,
                             uint32_t n, uint32_t* out_range, uint64_t* out_slots,
                             void* temp, uint64_t temp_bytes, void* stream);

/* Replaces FMIndexFilter<device_tag,fm_index>::locate(begin, end, hits)
 * (filter_inl.h:306-402): hits[2(h-begin)..] = {text position, seed id} for the
 * global hit indices h in [begin,end). */
int nvbio_hip_fm_filter_locate(const nvbio_hip_fmindex* fmi, const uint32_t* range, const uint64_t* slots,
                               uint32_t n_queries, uint64_t begin, uint64_t end,
                               uint32_t* out_hits, void* stream);

/* Replaces build_occurrence_table<2,64> (nvbio/fmindex/rank_dictionary_inl.h:42-77)
 * fused with the bwt|occ interleave of the loader (fmindex_impl.cu:305-327):
 * bwt_words = big-endian 2-bit BWT, 4*ceil(n/64) words (zero padded);
 * out_bwt_occ = 8*ceil(n/64) words; out_L2 = 5 device words.
 * temp: device scratch of nvbio_hip_build_bwt_occ_temp_bytes(n) bytes. */
uint64_t nvbio_hip_build_bwt_occ_temp_bytes(uint32_t n);
int nvbio_hip_build_bwt_occ(uint32_t n, const uint32_t* bwt_words, uint32_t* out_bwt_occ, uint32_t* out_L2,
                            void* temp, uint64_t temp_bytes, void* stream);

/* Device-memory helpers for host code that has no HIP headers (the C++ host layer in
 * include/nvbio_hip/ uses them where the reference uses thrust::device_vector).
 * kind: 1 = host->device, 2 = device->host, 3 = device->device. */
int nvbio_hip_device_malloc(void** ptr, uint64_t bytes);
int nvbio_hip_device_free(void* ptr);             /* hipFree's contract: waits for the device */
int nvbio_hip_device_free_ordered(void* ptr);     /* opt-in, does not stop the host: see "Memory helpers" below */
int nvbio_hip_device_free_after(void* ptr, void* stream);   /* the same, for a block whose work was queued on `stream` (any stream of the caller's) */
int nvbio_hip_device_trim(void);                  /* hipFree every idle block of this device's cache */
int nvbio_hip_device_mem_info(uint64_t* free_bytes, uint64_t* total_bytes, uint64_t* idle_cached_bytes);
int nvbio_hip_memcpy(void* dst, const void* src, uint64_t bytes, int kind, void* stream);
int nvbio_hip_memset(void* dst, int value, uint64_t bytes, void* stream);
int nvbio_hip_stream_synchronize(void* stream);
int nvbio_hip_stream_query(void* stream);         /* hipStreamQuery: 0 = everything queued so far has finished, 600 (hipErrorNotReady) = not yet */
/* Pinned host memory the device writes to through the same pointer (hipHostMalloc): where a kernel leaves the few words its host thread is
 * waiting for -- the sizes of a selection round's queues (include/nvbio_hip/select.h) -- so that the thread reads them as they land instead of
 * going through a stream synchronisation and a copy per round. */
int nvbio_hip_host_malloc(void** ptr, uint64_t bytes);
int nvbio_hip_host_free(void* ptr);

/* Streams.  The reference runs one host thread per device on its default stream (nvBowtie.cpp:809-864, compute_thread.cu:74-117).
 * Every nvbio_hip_* entry takes the stream its work is queued on, so a driver object per host thread, each on its own non-blocking
 * stream, keeps several batches in flight on one device: one batch's fabric-bound seeding (map / locate: random 128-byte lines, idle
 * VALUs) overlaps another's VALU-bound extension / selection / traceback (+7-10 % measured, profiles/r03/cosched_two_streams.txt).
 * (Round 3 also exported a CU-masked stream, a grid cap for the seeding kernels and a seeding token; none of them beat two plain
 * streams in any measured configuration -- a saturated fabric taxes co-resident VALU kernels whichever CUs they sit on,
 * profiles/r03/cosched_cu_mask.txt, cosched_token_grid_limit.txt -- and they were removed.)  Results never depend on any of this. */
int nvbio_hip_stream_create(void** stream, uint32_t non_blocking);
int nvbio_hip_stream_destroy(void* stream);

/* Multi-GPU.  The reference runs one host thread per device over a replicated index, all writing into one shared
 * output (nvBowtie/nvBowtie.cpp:809-864, bowtie2/cuda/compute_thread.cu:74-117): no collective.  Here every device aligns a contiguous
 * block of the reads and the fixed-size result records (16 B per read, 32 B per pair; 4-12 B for the extension stage alone) are gathered
 * to one rank over RCCL / xGMI with grouped ncclSend / ncclRecv -- the ONLY collective of the path.  RCCL is bound at run time (dlopen);
 * nvbio_hip_comm_available() = 0 means it could not be, and the comm entries return hipErrorNotSupported (801).
 *   one process, one host thread per device:  nvbio_hip_comm_init_all(comms, n, devices)   (ncclCommInitAll)
 *   one process per device (a launcher):      rank 0 makes nvbio_hip_comm_unique_id(id), the launcher ships the 128 bytes to every rank,
 *                                             each rank calls nvbio_hip_comm_init_rank(&comm, world, rank, id) on its own device
 * nvbio_hip_gather_records: counts[r] records of record_bytes bytes from rank r (all ranks pass the same counts); the root's recv holds
 * them in rank order.  Device buffers, queued on `stream`.  RCCL failures are returned as 2000 + ncclResult_t. */
int nvbio_hip_comm_available(void);
int nvbio_hip_device_count(void);
int nvbio_hip_set_device(int device);
int nvbio_hip_get_device(void);
int nvbio_hip_comm_unique_id(uint8_t* id128);
int nvbio_hip_comm_init_rank(void** comm, int world, int rank, const uint8_t* id128);
int nvbio_hip_comm_init_all(void** comms, int n_devices, const int* devices /* nullable: 0..n-1 */);
int nvbio_hip_comm_destroy(void* comm);
int nvbio_hip_comm_rank(void* comm, int* rank, int* world);
int nvbio_hip_gather_records(void* comm, const void* send, const uint64_t* counts, uint32_t record_bytes, void* recv /* root only */, int root, void* stream);
/* after a local failure: unblock the peers waiting on this communicator (ncclCommAbort); it is unusable afterwards */
int nvbio_hip_comm_abort(void* comm);
/* The transport seam under nvbio_hip_gather_records.  The gather is a plan of sends / receives / one copy (nvbio_hip/gather_plan.h)
 * executed through these five operations; the default table is RCCL.  Installing another table (NULL restores RCCL) lets the CPU suite
 * run the plan over host memory at worlds of 2..8, and lets a deployment without RCCL keep the entry point.  `comm` is whatever the
 * table's owner made it; every function returns 0 or an error code. */
typedef struct nvbio_hip_comm_transport {
    int (*rank)(void* comm, int* rank, int* world);
    int (*group_start)(void* comm);
    int (*group_end)(void* comm);
    int (*send)(void* comm, const void* buf, uint64_t bytes, int peer, void* stream);
    int (*recv)(void* comm, void* buf, uint64_t bytes, int peer, void* stream);
    int (*copy)(void* comm, void* dst, const void* src, uint64_t bytes, void* stream);
    int (*abort)(void* comm);
} nvbio_hip_comm_transport;
void nvbio_hip_comm_set_transport(const nvbio_hip_comm_transport* transport);

/* Test switches.  A few named integers select alternative executions of the same results, for the parity suite to cover both
 * (NVBIO_HIP_FORCE_32BIT, NVBIO_HIP_NO_STAGING, NVBIO_HIP_FULL_GENERIC, NVBIO_HIP_FULL_SINGLE_JOB, NVBIO_HIP_FULL_ROWS, NVBIO_HIP_ED_SWEEP,
 * NVBIO_HIP_SELECT_LANES, NVBIO_HIP_TRACEBACK_LANES).  Each is seeded ONCE per process from the environment variable of the same name and
 * changed afterwards only through nvbio_hip_set_test_switch (atomic; safe while other threads are inside library calls -- a call in flight
 * uses the value it read when it started).  0 = the default execution.  Production code leaves them alone. */
int nvbio_hip_set_test_switch(const char* name, int value);     /* hipErrorInvalidValue for an unknown name */
int nvbio_hip_get_test_switch(const char* name);                /* -1 for an unknown name */
const char* nvbio_hip_test_switch_name(int index);              /* the index-th switch's name, NULL past the last: how a binding enumerates them */
/* Memory helpers: nvbio_hip_device_malloc / nvbio_hip_device_free hand out hipMalloc'ed blocks kept in a per-device cache of the library's own
 * (a freed block is listed and reused for a request of at least half its size; up to NVBIO_HIP_POOL_KEEP_MB = 2048 MB stay listed;
 * nvbio_hip_device_trim hands the idle ones back, nvbio_hip_device_mem_info says how much is idle).  Not a hipMemPool: under two host threads
 * on one device -- the reference's nvBowtie --device 0 --device 0 -- ROCm 7.0's pool loses the contents of live blocks
 * (profiles/r05/two_threads_pool.txt).
 *   nvbio_hip_device_free          hipFree's contract (the reference's cudaFree): the device is idle before the block can change hands, whatever
 *                                  stream used it -- the caller's own non-blocking streams and torch's pool streams included.  What
 *                                  nvbio::vector<device_tag> / device_buffer of the drop-in layer call.
 *   nvbio_hip_device_free_ordered  opt-in, does not stop the host.  The caller vouches that all work on the block was queued, before the call, on the
 *                                  default stream, a blocking stream or a stream made by nvbio_hip_stream_create: the default stream is put
 *                                  behind what each of the latter holds now (hipStreamWaitEvent), and nvbio_hip_device_malloc returns after
 *                                  the default stream has drained, so the next owner never overlaps the previous one's work, whichever thread
 *                                  it is.  What this repository's C++ host layer uses (include/nvbio_hip/types.h).
 *   nvbio_hip_device_free_after    the ordered form for a block all of whose work was queued on ONE known stream of the caller's (registered or not):
 *                                  the default stream is put behind that stream too.  What a batch object's scratch (device_buffer, compat
 *                                  alignment/batched.h) uses: it dies right after enact(stream) returns, with its kernels still queued there.
 * Call them from a thread bound to the device the block lives on; per-batch storage lives in a hip::device_arena and comes through here once. */

/* Library / device introspection (host). */
int         nvbio_hip_abi_version(void);
const char* nvbio_hip_arch(void);           /* "gfx950" */
const char* nvbio_hip_last_kernel(void);    /* name of the last kernel variant launched by this thread */

#ifdef __cplusplus
}
#endif
#endif /* NVBIO_HIP_H */
