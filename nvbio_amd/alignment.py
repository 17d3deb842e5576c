"""Host layer mirroring nvbio::aln for the banded Gotoh path.

Names follow the reference: AlignmentType (alignment_base.h:54), SimpleGotohScheme
(utils.h:114-134), GotohAligner / make_gotoh_aligner (alignment_base.h:255-298),
BatchedBandedAlignmentScore<BAND_LEN,stream,scheduler>::enact (batched.h:333-353) and
batch_banded_alignment_score<BAND_LEN> (batched_inl.h:1074-1101).  All compute happens in
libnvbio_hip.so; there is no CPU path here.
"""
import ctypes as C

import torch

import numpy as np

from ._lib import lib, check, GotohSchemeStruct, GotohQualSchemeStruct, current_stream_ptr

GLOBAL, LOCAL, SEMI_GLOBAL = 0, 1, 2


class SimpleGotohScheme:
    def __init__(self, match, mismatch, gap_open, gap_ext):
        self.m_match, self.m_mismatch, self.m_gap_open, self.m_gap_ext = int(match), int(mismatch), int(gap_open), int(gap_ext)

    def struct(self):
        return GotohSchemeStruct(self.m_match, self.m_mismatch, self.m_gap_open, self.m_gap_ext)


class SmithWatermanScoringScheme:
    """nvBowtie's scoring scheme as its Gotoh aligner sees it (nvBowtie/bowtie2/cuda/scoring.h:206-356):
    constant match bonus, quality-dependent mismatch penalty QualCost(min,max) (scoring.h:86-104) or a
    constant one, affine read / reference gap costs.  Defaults are bowtie2's end-to-end values."""

    def __init__(self, match=0, mmp_min=2, mmp_max=6, read_gap_const=5, read_gap_coeff=3,
                 ref_gap_const=5, ref_gap_coeff=3, mm_cost="qual", score_min=(0, -0.6, -0.6)):
        self.m_match = int(match)
        self.m_score_min = score_min                               # SimpleFunc (type, k, m): LinearFunc -0.6 -0.6 (scoring_inl.h:108)
        self.m_monotone = int(match) == 0                          # scoring_inl.h:143
        self.m_mmp_min, self.m_mmp_max, self.mm_cost = int(mmp_min), int(mmp_max), mm_cost
        self.m_read_gap_const, self.m_read_gap_coeff = int(read_gap_const), int(read_gap_coeff)
        self.m_ref_gap_const, self.m_ref_gap_coeff = int(ref_gap_const), int(ref_gap_coeff)

    @staticmethod
    def local():
        """SmithWatermanScoringScheme::local() (scoring_inl.h:81-101): match 2, score-min = log, 0 + 10 ln(len)"""
        return SmithWatermanScoringScheme(match=2, score_min=(1, 0.0, 10.0))

    @staticmethod
    def edit_distance(max_dist=15):
        """EditDistanceAligner's costs (ed_utils.h:44-51: match 0, everything else -1) in this scheme's terms, with nvBowtie's edit-distance
        threshold score-min = -max_dist (params.cpp:203-204): with them the Gotoh recurrences are the linear-gap ones cell for cell (gap
        open == gap extension, so H >= E, F), which is how the drivers' scoring stages run --scoring ed on the quality-scheme kernels"""
        return SmithWatermanScoringScheme(match=0, mmp_min=1, mmp_max=1, read_gap_const=0, read_gap_coeff=1, ref_gap_const=0, ref_gap_coeff=1,
                                          mm_cost="constant", score_min=(0, -float(max_dist), 0.0))

    def perfect_score(self, read_len):                             # scoring.h:281
        return int(read_len) * self.m_match

    def min_score(self, read_len):                                 # scoring.h:272
        from .mapping import simple_func
        return simple_func(*self.m_score_min, read_len)

    def mmp(self, q):
        """m_mmp(q).  QualCost: min + int(frac * (max - min)), frac = float(min(q,40) / 40.0f), in
        single precision with truncation, exactly as the reference writes it."""
        if self.mm_cost == "constant":
            return self.m_mmp_max                                  # ConstantCost(min,max) keeps max
        frac = np.float32(np.float32(min(int(q), 40)) / np.float32(40.0))
        return self.m_mmp_min + int(np.float32(frac * np.float32(self.m_mmp_max - self.m_mmp_min)))

    # the aln::GotohAligner interface (scoring.h:283-293)
    def match(self, q=0): return self.m_match
    def mismatch(self, q=0): return -self.mmp(q)
    def pattern_gap_open(self): return -self.m_read_gap_const - self.m_read_gap_coeff
    def pattern_gap_extension(self): return -self.m_read_gap_coeff
    def text_gap_open(self): return -self.m_ref_gap_const - self.m_ref_gap_coeff
    def text_gap_extension(self): return -self.m_ref_gap_coeff

    def struct(self):
        s = GotohQualSchemeStruct()
        s.match = self.match()
        s.pattern_gap_open, s.pattern_gap_ext = self.pattern_gap_open(), self.pattern_gap_extension()
        s.text_gap_open, s.text_gap_ext = self.text_gap_open(), self.text_gap_extension()
        for q in range(256):
            s.mismatch[q] = self.mismatch(q)
        return s


PATTERN_BLOCKING, TEXT_BLOCKING = 0, 1      # algorithm tags (alignment_base.h:72-79); only the full-matrix batch looks at them


class GotohAligner:
    def __init__(self, aln_type, scheme, algorithm=TEXT_BLOCKING):
        assert aln_type in (GLOBAL, LOCAL, SEMI_GLOBAL) and algorithm in (PATTERN_BLOCKING, TEXT_BLOCKING)
        self.type, self.scheme, self.algorithm = aln_type, scheme, algorithm


def make_gotoh_aligner(aln_type, scheme, algorithm=TEXT_BLOCKING):
    """make_gotoh_aligner<TYPE, algorithm_tag>(scheme).  NOTE: this layer defaults to the text-blocking form (what
    sw-benchmark instantiates); the reference's C++ default tag is PatternBlockingTag -- pass PATTERN_BLOCKING for it."""
    return GotohAligner(aln_type, scheme, algorithm)


class SimpleSmithWatermanScheme:
    """nvbio::aln::SimpleSmithWatermanScheme (utils.h:92-109): linear gap costs."""

    def __init__(self, match, mismatch, deletion, insertion):
        self.m_match, self.m_mismatch, self.m_deletion, self.m_insertion = int(match), int(mismatch), int(deletion), int(insertion)

    def struct(self):      # nvbio_hip_sw_scheme has the layout of four int32
        return GotohSchemeStruct(self.m_match, self.m_mismatch, self.m_deletion, self.m_insertion)


class SmithWatermanAligner:
    """SmithWatermanAligner<TYPE, scheme> (alignment_base.h): linear-gap DP."""

    def __init__(self, aln_type, scheme, algorithm=TEXT_BLOCKING):
        assert aln_type in (GLOBAL, LOCAL, SEMI_GLOBAL) and algorithm in (PATTERN_BLOCKING, TEXT_BLOCKING)
        self.type, self.scheme, self.algorithm = aln_type, scheme, algorithm


class EditDistanceAligner(SmithWatermanAligner):
    """EditDistanceAligner<TYPE>: the SW code with EditDistanceSWScheme (ed_utils.h:44-51)."""

    def __init__(self, aln_type, algorithm=TEXT_BLOCKING):
        super().__init__(aln_type, SimpleSmithWatermanScheme(0, -1, -1, -1), algorithm)


def make_smith_waterman_aligner(aln_type, scheme, algorithm=TEXT_BLOCKING):
    return SmithWatermanAligner(aln_type, scheme, algorithm)


def make_edit_distance_aligner(aln_type, algorithm=TEXT_BLOCKING):
    return EditDistanceAligner(aln_type, algorithm)


class BatchedBandedAlignmentScore:
    """BatchedBandedAlignmentScore<BAND_LEN, stream, DeviceThreadBlockScheduler>.

    `enact` takes the pieces of the reference's stream concept that the C-ABI needs:
    the aligner, the pattern / text string sets and the output sink arrays."""

    def __init__(self, band_len):
        if band_len not in (3, 5, 7, 15, 31):
            raise ValueError("unsupported BAND_LEN %d" % band_len)
        self.band_len = band_len

    @staticmethod
    def min_temp_storage(max_pattern_len, max_text_len, stream_size):
        return 0    # batched_banded_inl.h:143-147

    max_temp_storage = min_temp_storage

    def enact(self, aligner, patterns, texts, out_score, out_sink, max_pattern_length=0, max_text_length=0, quals=None, pattern_flags=None,
              min_score=None, n_on_device=None, out_index=None):
        """quals: uint8 device tensor indexed like the pattern stream's symbols -- required by (and only
        used with) a SmithWatermanScoringScheme aligner, as nvBowtie's read qualities are.
        pattern_flags: optional uint8 device tensor, one byte per job -- the io::ReadStream view nvBowtie's streams take of a stored
        read (bit 0 = walk it backwards, bit 1 = complement), applied by the kernel as it fetches (quality scheme only).
        min_score: optional int32 device tensor, one threshold per job (the reference's min_score argument, quality scheme only): a job that
        cannot end above its threshold is given up and reports some score <= the threshold and the sink (-1, -1); jobs that end above it are
        exact (nvbio_hip_banded_gotoh_score_qual_bounded).  n_on_device: optional int32[1] device tensor holding the job count.  out_index: optional
        int32 device tensor: job i's results go to out_score[out_index[i]] / out_sink[out_index[i]]."""
        n = len(patterns)
        if patterns.length is None:
            max_pattern_length = max_pattern_length or patterns.fixed_length
        if texts.length is None:
            max_text_length = max_text_length or texts.fixed_length
        assert len(texts) == n
        assert out_score.dtype == torch.int32 and out_score.numel() >= n and out_score.is_cuda
        assert out_sink.dtype == torch.int32 and out_sink.numel() >= 2 * n and out_sink.is_cuda
        sc = aligner.scheme.struct()
        ps, ts = patterns.struct(), texts.struct()
        if isinstance(aligner, SmithWatermanAligner):
            err = lib().nvbio_hip_banded_sw_score(
                C.byref(sc), aligner.type, self.band_len, C.byref(ps), C.byref(ts),
                int(max_pattern_length), int(max_text_length), n,
                C.c_void_p(out_score.data_ptr()), C.c_void_p(out_sink.data_ptr()), current_stream_ptr())
            check(err, "nvbio_hip_banded_sw_score")
            return
        if isinstance(aligner.scheme, SmithWatermanScoringScheme):
            assert quals is not None and quals.dtype == torch.uint8 and quals.is_cuda and quals.is_contiguous()
            if pattern_flags is not None:
                assert pattern_flags.dtype == torch.uint8 and pattern_flags.is_cuda and pattern_flags.numel() >= n
            if min_score is not None or n_on_device is not None or out_index is not None:
                assert min_score is None or (min_score.dtype == torch.int32 and min_score.is_cuda and min_score.numel() >= n)
                counter = torch.empty(1, dtype=torch.int32, device=out_score.device)
                err = lib().nvbio_hip_banded_gotoh_score_qual_bounded(
                    C.byref(sc), aligner.type, self.band_len, C.byref(ps), C.c_void_p(quals.data_ptr()), quals.numel(),
                    C.c_void_p(pattern_flags.data_ptr()) if pattern_flags is not None else None, C.byref(ts),
                    int(max_pattern_length), int(max_text_length), n,
                    C.c_void_p(n_on_device.data_ptr()) if n_on_device is not None else None,
                    C.c_void_p(min_score.data_ptr()) if min_score is not None else None, C.c_void_p(counter.data_ptr()),
                    C.c_void_p(out_index.data_ptr()) if out_index is not None else None, None, 0,
                    C.c_void_p(out_score.data_ptr()), C.c_void_p(out_sink.data_ptr()), current_stream_ptr())
                check(err, "nvbio_hip_banded_gotoh_score_qual_bounded")
                return
            err = lib().nvbio_hip_banded_gotoh_score_qual_views(
                C.byref(sc), aligner.type, self.band_len, C.byref(ps), C.c_void_p(quals.data_ptr()), quals.numel(),
                C.c_void_p(pattern_flags.data_ptr()) if pattern_flags is not None else None, C.byref(ts),
                int(max_pattern_length), int(max_text_length), n,
                C.c_void_p(out_score.data_ptr()), C.c_void_p(out_sink.data_ptr()), current_stream_ptr())
            check(err, "nvbio_hip_banded_gotoh_score_qual_views")
            return
        assert pattern_flags is None, "pattern views are a feature of the quality-scheme entry point"
        err = lib().nvbio_hip_banded_gotoh_score(
            C.byref(sc), aligner.type, self.band_len, C.byref(ps), C.byref(ts),
            int(max_pattern_length), int(max_text_length), n,
            C.c_void_p(out_score.data_ptr()), C.c_void_p(out_sink.data_ptr()), current_stream_ptr())
        check(err, "nvbio_hip_banded_gotoh_score")


def batch_banded_alignment_score_wave(band_len, aligner, patterns, texts, quals, out_score=None, out_sink=None, max_pattern_length=0, job_index=None, n_on_device=None):
    """The quality-scheme banded scorer with one wave per job (nvbio_hip_banded_gotoh_score_qual_wave): the anti-diagonal sweep, for small batches.
    job_index: optional int32 device tensor -- job k is job_index[k] of the string sets and writes its outputs there (len(job_index) jobs run)."""
    n_sets = len(patterns)
    dev = patterns.words.device
    if out_score is None:
        out_score = torch.empty(n_sets, dtype=torch.int32, device=dev)
    if out_sink is None:
        out_sink = torch.empty((n_sets, 2), dtype=torch.int32, device=dev)
    n = int(job_index.numel()) if job_index is not None else n_sets
    sc = aligner.scheme.struct()
    ps, ts = patterns.struct(), texts.struct()
    if patterns.length is None:
        max_pattern_length = max_pattern_length or patterns.fixed_length
    check(lib().nvbio_hip_banded_gotoh_score_qual_wave(
        C.byref(sc), aligner.type, band_len, C.byref(ps), C.c_void_p(quals.data_ptr()) if quals is not None else None, quals.numel() if quals is not None else 0,
        C.byref(ts), int(max_pattern_length), n, C.c_void_p(n_on_device.data_ptr()) if n_on_device is not None else None,
        C.c_void_p(job_index.data_ptr()) if job_index is not None else None, None, None, 0,
        C.c_void_p(out_score.data_ptr()), C.c_void_p(out_sink.data_ptr()), current_stream_ptr()), "nvbio_hip_banded_gotoh_score_qual_wave")
    return out_score, out_sink


def batch_banded_alignment_score(band_len, aligner, patterns, texts, out_score=None, out_sink=None,
                                 max_pattern_length=0, max_text_length=0, quals=None, pattern_flags=None, min_score=None, n_on_device=None):
    """batch_banded_alignment_score<BAND_LEN>(aligner, patterns, texts, sinks, DeviceThreadScheduler()).
    Returns (score[n] int32, sink[n,2] int32 holding the uint32 bit patterns)."""
    n = len(patterns)
    dev = patterns.words.device
    if out_score is None:
        out_score = torch.empty(n, dtype=torch.int32, device=dev)
    if out_sink is None:
        out_sink = torch.empty((n, 2), dtype=torch.int32, device=dev)
    BatchedBandedAlignmentScore(band_len).enact(aligner, patterns, texts, out_score, out_sink,
                                                max_pattern_length, max_text_length, quals, pattern_flags, min_score, n_on_device)
    return out_score, out_sink


class BatchedBandedAlignmentTraceback:
    """BatchedBandedAlignmentTraceback<BAND_LEN, CHECKPOINTS, stream, DeviceThreadScheduler> (batched.h:460-476)
    with nvBowtie's CIGAR-forming backtracer (alignment_utils.h:125-168).  CHECKPOINTS is accepted for
    signature parity and ignored: the whole band's flow flags live in the temp storage."""

    def __init__(self, band_len, checkpoints=32):
        if band_len not in (3, 5, 7, 15, 31):
            raise ValueError("unsupported BAND_LEN %d" % band_len)
        self.band_len = band_len
        self.checkpoints = checkpoints

    def min_temp_storage(self, max_pattern_len, max_text_len, stream_size):
        return int(lib().nvbio_hip_banded_gotoh_traceback_temp_bytes(self.band_len, int(max_pattern_len), int(stream_size)))

    max_temp_storage = min_temp_storage

    def enact(self, aligner, patterns, texts, out_score, out_sink, out_source, out_cigar, out_cigar_len,
              max_pattern_length=0, max_text_length=0, quals=None, temp=None, known=False):
        """out_cigar: int16 [n, cigar_stride] device tensor holding the io::Cigar uint16 bit patterns.  known: out_score / out_sink hold on
        entry what the banded scorer reports for these jobs (quality-aware scheme only): the score pass is skipped."""
        n = len(patterns)
        assert len(texts) == n
        if patterns.length is None:
            max_pattern_length = max_pattern_length or patterns.fixed_length
        if texts.length is None:
            max_text_length = max_text_length or texts.fixed_length
        for t, k in ((out_score, 1), (out_sink, 2), (out_source, 2), (out_cigar_len, 1)):
            assert t.dtype == torch.int32 and t.is_cuda and t.is_contiguous() and t.numel() >= k * n
        assert out_cigar.dtype == torch.int16 and out_cigar.is_cuda and out_cigar.is_contiguous() and out_cigar.dim() == 2 and out_cigar.shape[0] >= n
        need = self.min_temp_storage(max_pattern_length, max_text_length, n)
        if temp is None:
            temp = torch.empty(max(need, 8), dtype=torch.uint8, device=patterns.words.device)
        assert temp.is_cuda and temp.numel() * temp.element_size() >= need
        sc = aligner.scheme.struct()
        ps, ts = patterns.struct(), texts.struct()
        tail = (int(max_pattern_length), int(max_text_length), n,
                C.c_void_p(out_score.data_ptr()), C.c_void_p(out_sink.data_ptr()), C.c_void_p(out_source.data_ptr()),
                C.c_void_p(out_cigar.data_ptr()), int(out_cigar.shape[1]), C.c_void_p(out_cigar_len.data_ptr()),
                C.c_void_p(temp.data_ptr()), temp.numel() * temp.element_size(), current_stream_ptr())
        if isinstance(aligner, SmithWatermanAligner):
            err = lib().nvbio_hip_banded_sw_traceback(C.byref(sc), aligner.type, self.band_len, C.byref(ps), C.byref(ts), *tail)
            check(err, "nvbio_hip_banded_sw_traceback")
        elif isinstance(aligner.scheme, SmithWatermanScoringScheme):
            assert quals is not None and quals.dtype == torch.uint8 and quals.is_cuda and quals.is_contiguous()
            fn = lib().nvbio_hip_banded_gotoh_traceback_qual_known if known else lib().nvbio_hip_banded_gotoh_traceback_qual
            err = fn(C.byref(sc), aligner.type, self.band_len, C.byref(ps), C.c_void_p(quals.data_ptr()), quals.numel(), C.byref(ts), *tail)
            check(err, "nvbio_hip_banded_gotoh_traceback_qual")
        else:
            err = lib().nvbio_hip_banded_gotoh_traceback(C.byref(sc), aligner.type, self.band_len, C.byref(ps), C.byref(ts), *tail)
            check(err, "nvbio_hip_banded_gotoh_traceback")
        # the kernel reads `temp` asynchronously: keep it alive until the stream has passed it
        temp.record_stream(torch.cuda.current_stream())


def batch_banded_alignment_traceback(band_len, aligner, patterns, texts, max_pattern_length=0, max_text_length=0,
                                     quals=None, cigar_stride=None, known=None):
    """One call form: returns dict(score[n], sink[n,2], source[n,2], cigar[n,stride] int16, cigar_len[n]).  known = (score int32[n],
    sink int32[n,2]) of every job as the banded scorer reports them, when the caller has them already (taken over as outputs)."""
    n = len(patterns)
    dev = patterns.words.device
    maxM = max_pattern_length or patterns.fixed_length
    if cigar_stride is None:
        cigar_stride = min(int(maxM) + band_len + 2, 64)
    out = dict(score=known[0] if known else torch.empty(n, dtype=torch.int32, device=dev),
               sink=known[1] if known else torch.empty((n, 2), dtype=torch.int32, device=dev),
               source=torch.empty((n, 2), dtype=torch.int32, device=dev),
               cigar=torch.zeros((max(n, 1), cigar_stride), dtype=torch.int16, device=dev),
               cigar_len=torch.empty(n, dtype=torch.int32, device=dev))
    BatchedBandedAlignmentTraceback(band_len).enact(aligner, patterns, texts, out["score"], out["sink"], out["source"],
                                                    out["cigar"], out["cigar_len"], max_pattern_length, max_text_length, quals, known=bool(known))
    return out


def batch_alignment_traceback(aligner, patterns, texts, max_pattern_length=0, max_text_length=0, cigar_stride=64, quals=None, known_score=None):
    """BatchedAlignmentTraceback<CHECKPOINTS, stream>::enact (batched.h:432-452) for the full-matrix Gotoh aligner with
    nvBowtie's backtracer: returns dict(score, sink, source, cigar int16[n,stride], cigar_len) as the banded form does.
    known_score (int32[n], Gotoh aligners): the caller knows every job's best score and that its alignment ends at the last text symbol
    (opposite-mate tracebacks); same results, the unreachable text rows are dropped first (nvbio_hip_gotoh_traceback*_known_score)."""
    n = len(patterns)
    assert len(texts) == n and isinstance(aligner.scheme, (SimpleGotohScheme, SimpleSmithWatermanScheme, SmithWatermanScoringScheme))
    dev = patterns.words.device
    maxM = max_pattern_length or patterns.fixed_length
    maxN = max_text_length or texts.fixed_length
    need = int(lib().nvbio_hip_gotoh_traceback_temp_bytes(int(maxM), int(maxN), n))
    temp = torch.empty(max(need, 8), dtype=torch.uint8, device=dev)
    out = dict(score=torch.empty(n, dtype=torch.int32, device=dev), sink=torch.empty((n, 2), dtype=torch.int32, device=dev),
               source=torch.empty((n, 2), dtype=torch.int32, device=dev), cigar=torch.zeros((max(n, 1), cigar_stride), dtype=torch.int16, device=dev),
               cigar_len=torch.empty(n, dtype=torch.int32, device=dev))
    sc = aligner.scheme.struct()
    ps, ts = patterns.struct(), texts.struct()
    tail = (int(maxM), int(maxN), n, C.c_void_p(out["score"].data_ptr()), C.c_void_p(out["sink"].data_ptr()), C.c_void_p(out["source"].data_ptr()),
            C.c_void_p(out["cigar"].data_ptr()), cigar_stride, C.c_void_p(out["cigar_len"].data_ptr()),
            C.c_void_p(temp.data_ptr()), temp.numel(), current_stream_ptr())
    if known_score is not None:
        assert known_score.dtype == torch.int32 and known_score.numel() == n and known_score.is_contiguous() and not isinstance(aligner, SmithWatermanAligner)
    if isinstance(aligner.scheme, SmithWatermanScoringScheme):
        assert quals is not None and quals.dtype == torch.uint8 and quals.is_cuda and quals.is_contiguous()
        if known_score is not None:
            check(lib().nvbio_hip_gotoh_traceback_qual_known_score(C.byref(sc), aligner.type, C.byref(ps), C.c_void_p(quals.data_ptr()), quals.numel(), C.byref(ts),
                                                                   C.c_void_p(known_score.data_ptr()), *tail), "nvbio_hip_gotoh_traceback_qual_known_score")
        else:
            check(lib().nvbio_hip_gotoh_traceback_qual(C.byref(sc), aligner.type, C.byref(ps), C.c_void_p(quals.data_ptr()), quals.numel(), C.byref(ts), *tail),
                  "nvbio_hip_gotoh_traceback_qual")
        temp.record_stream(torch.cuda.current_stream())
        return out
    if known_score is not None:
        check(lib().nvbio_hip_gotoh_traceback_known_score(C.byref(sc), aligner.type, C.byref(ps), C.byref(ts), C.c_void_p(known_score.data_ptr()), *tail),
              "nvbio_hip_gotoh_traceback_known_score")
        temp.record_stream(torch.cuda.current_stream())
        return out
    fn = lib().nvbio_hip_sw_traceback if isinstance(aligner, SmithWatermanAligner) else lib().nvbio_hip_gotoh_traceback
    err = fn(C.byref(sc), aligner.type, C.byref(ps), C.byref(ts), int(maxM), int(maxN), n,
                                          C.c_void_p(out["score"].data_ptr()), C.c_void_p(out["sink"].data_ptr()), C.c_void_p(out["source"].data_ptr()),
                                          C.c_void_p(out["cigar"].data_ptr()), cigar_stride, C.c_void_p(out["cigar_len"].data_ptr()),
                                          C.c_void_p(temp.data_ptr()), temp.numel(), current_stream_ptr())
    check(err, "nvbio_hip_{gotoh,sw}_traceback")
    temp.record_stream(torch.cuda.current_stream())
    return out


class BatchedAlignmentScore:
    """BatchedAlignmentScore<stream, DeviceThreadScheduler> (batched.h:310-329) for the full-matrix Gotoh
    score with the text-blocking aligner sw-benchmark instantiates (sw-benchmark.cu:604-631)."""

    def enact(self, aligner, patterns, texts, out_score, out_sink, max_pattern_length=0, max_text_length=0,
              min_score=None, out_ok=None, quals=None):
        n = len(patterns)
        assert len(texts) == n and isinstance(aligner.scheme, (SimpleGotohScheme, SimpleSmithWatermanScheme, SmithWatermanScoringScheme))
        if patterns.length is None:
            max_pattern_length = max_pattern_length or patterns.fixed_length
        if texts.length is None:
            max_text_length = max_text_length or texts.fixed_length
        sc = aligner.scheme.struct()
        ps, ts = patterns.struct(), texts.struct()
        if isinstance(aligner.scheme, SmithWatermanScoringScheme):
            assert quals is not None and quals.dtype == torch.uint8 and quals.is_cuda and quals.is_contiguous()
            err = lib().nvbio_hip_alignment_score_qual(
                C.byref(sc), getattr(aligner, "algorithm", TEXT_BLOCKING), aligner.type, C.byref(ps), C.c_void_p(quals.data_ptr()), quals.numel(), C.byref(ts),
                int(max_pattern_length), int(max_text_length), C.c_void_p(min_score.data_ptr()) if min_score is not None else None, n,
                C.c_void_p(out_score.data_ptr()), C.c_void_p(out_sink.data_ptr()),
                C.c_void_p(out_ok.data_ptr()) if out_ok is not None else None, current_stream_ptr())
            check(err, "nvbio_hip_alignment_score_qual")
            return
        if getattr(aligner, "algorithm", TEXT_BLOCKING) == PATTERN_BLOCKING:
            s4 = (C.c_int32 * 4)(*[getattr(sc, f) for f, _ in sc._fields_])
            err = lib().nvbio_hip_alignment_score(
                1 if isinstance(aligner, SmithWatermanAligner) else 0, PATTERN_BLOCKING, s4, aligner.type, C.byref(ps), C.byref(ts),
                int(max_pattern_length), int(max_text_length), C.c_void_p(min_score.data_ptr()) if min_score is not None else None, n,
                C.c_void_p(out_score.data_ptr()), C.c_void_p(out_sink.data_ptr()),
                C.c_void_p(out_ok.data_ptr()) if out_ok is not None else None, current_stream_ptr())
            check(err, "nvbio_hip_alignment_score")
            return
        if isinstance(aligner, SmithWatermanAligner):
            # the text-blocking SW / ED form never exits early (sw_inl.h:1075-1222): min_score is not consulted
            err = lib().nvbio_hip_sw_score(
                C.byref(sc), aligner.type, C.byref(ps), C.byref(ts), int(max_pattern_length), int(max_text_length), n,
                C.c_void_p(out_score.data_ptr()), C.c_void_p(out_sink.data_ptr()), current_stream_ptr())
            check(err, "nvbio_hip_sw_score")
            if out_ok is not None:
                out_ok.fill_(1)
            return
        err = lib().nvbio_hip_gotoh_score(
            C.byref(sc), aligner.type, C.byref(ps), C.byref(ts), int(max_pattern_length), int(max_text_length),
            C.c_void_p(min_score.data_ptr()) if min_score is not None else None, n,
            C.c_void_p(out_score.data_ptr()), C.c_void_p(out_sink.data_ptr()),
            C.c_void_p(out_ok.data_ptr()) if out_ok is not None else None, current_stream_ptr())
        check(err, "nvbio_hip_gotoh_score")


def batch_alignment_score(aligner, patterns, texts, max_pattern_length=0, max_text_length=0, min_score=None, quals=None):
    """batch_alignment_score(aligner, patterns, texts, sinks, DeviceThreadScheduler(), maxP, maxT)
    (batched.h:160-190).  Returns (score[n], sink[n,2], ok[n] uint8)."""
    n = len(patterns)
    dev = patterns.words.device
    score = torch.empty(n, dtype=torch.int32, device=dev)
    sink = torch.empty((n, 2), dtype=torch.int32, device=dev)
    ok = torch.empty(n, dtype=torch.uint8, device=dev)
    BatchedAlignmentScore().enact(aligner, patterns, texts, score, sink, max_pattern_length, max_text_length, min_score, ok, quals)
    return score, sink, ok
