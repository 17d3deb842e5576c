"""Host layer mirroring nvbio::aln for the banded Gotoh path.

Names follow the reference: AlignmentType (alignment_base.h:54), SimpleGotohScheme
(utils.h:114-134), GotohAligner / make_gotoh_aligner (alignment_base.h:255-298),
BatchedBandedAlignmentScore<BAND_LEN,stream,scheduler>::enact (batched.h:333-353) and
batch_banded_alignment_score<BAND_LEN> (batched_inl.h:1074-1101).  All compute happens in
libnvbio_hip.so; there is no CPU path here.
"""
import ctypes as C

import torch

from ._lib import lib, check, GotohSchemeStruct, current_stream_ptr

GLOBAL, LOCAL, SEMI_GLOBAL = 0, 1, 2


class SimpleGotohScheme:
    def __init__(self, match, mismatch, gap_open, gap_ext):
        self.m_match, self.m_mismatch, self.m_gap_open, self.m_gap_ext = int(match), int(mismatch), int(gap_open), int(gap_ext)

    def struct(self):
        return GotohSchemeStruct(self.m_match, self.m_mismatch, self.m_gap_open, self.m_gap_ext)


class GotohAligner:
    def __init__(self, aln_type, scheme):
        assert aln_type in (GLOBAL, LOCAL, SEMI_GLOBAL)
        self.type, self.scheme = aln_type, scheme


def make_gotoh_aligner(aln_type, scheme):
    return GotohAligner(aln_type, scheme)


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

    def enact(self, aligner, patterns, texts, out_score, out_sink, max_pattern_length=0, max_text_length=0):
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
        err = lib().nvbio_hip_banded_gotoh_score(
            C.byref(sc), aligner.type, self.band_len, C.byref(ps), C.byref(ts),
            int(max_pattern_length), int(max_text_length), n,
            C.c_void_p(out_score.data_ptr()), C.c_void_p(out_sink.data_ptr()), current_stream_ptr())
        check(err, "nvbio_hip_banded_gotoh_score")


def batch_banded_alignment_score(band_len, aligner, patterns, texts, out_score=None, out_sink=None,
                                 max_pattern_length=0, max_text_length=0):
    """batch_banded_alignment_score<BAND_LEN>(aligner, patterns, texts, sinks, DeviceThreadScheduler()).
    Returns (score[n] int32, sink[n,2] int32 holding the uint32 bit patterns)."""
    n = len(patterns)
    dev = patterns.words.device
    if out_score is None:
        out_score = torch.empty(n, dtype=torch.int32, device=dev)
    if out_sink is None:
        out_sink = torch.empty((n, 2), dtype=torch.int32, device=dev)
    BatchedBandedAlignmentScore(band_len).enact(aligner, patterns, texts, out_score, out_sink,
                                                max_pattern_length, max_text_length)
    return out_score, out_sink
