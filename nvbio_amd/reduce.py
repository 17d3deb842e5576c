"""Host layer of nvBowtie's score reduction and mapping quality stages (nvBowtie/bowtie2/cuda/reduce.h:
score_reduce; mapq.h: BowtieMapq2 / BowtieMapq3), computed by libnvbio_hip.so."""
import ctypes as C

import numpy as np
import torch

from ._lib import lib, check, current_stream_ptr, PeParamsStruct


def _vp(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class BestAlignments:
    """pipeline.best_alignments: io::Alignment[2][best_stride] as int64 (low word = the bit-field word,
    high word = m_align), initialised as init_alignments does (aligner.h:323-366): unaligned, with the
    scheme's threshold score for the read's length."""

    def __init__(self, n_reads, scheme, read_len=None, fixed_read_len=0, max_read_len=None, device="cuda", mate=0):
        self.n, self.stride = n_reads, n_reads
        self.data = torch.empty((2, n_reads), dtype=torch.int64, device=device)
        max_len = int(max_read_len or fixed_read_len or int(read_len.max()))
        table = torch.tensor([scheme.min_score(L) if L > 0 else 0 for L in range(max_len + 1)], dtype=torch.int32, device=device)
        check(lib().nvbio_hip_init_alignments(n_reads, _vp(read_len), int(fixed_read_len), _vp(table), int(mate), _vp(self.data), self.stride,
                                              current_stream_ptr()), "nvbio_hip_init_alignments")
        table.record_stream(torch.cuda.current_stream())

    def _field(self, k, shift, mask):
        return (self.data[k] >> shift) & mask

    def score(self, k=0):
        mag = self._field(k, 1, 0x1FFFF)
        return torch.where((self.data[k] & 1) != 0, -mag, mag)

    def is_aligned(self, k=0):
        return ((self.data[k] >> 32) & 0xFFFFFFFF) != 0xFFFFFFFF

    def alignment(self, k=0):
        return (self.data[k] >> 32) & 0xFFFFFFFF

    def is_rc(self, k=0):
        return self._field(k, 28, 1)


def score_reduce(best, hit_begin, hit_score, hit_loc, hit_rc, read_len=None, fixed_read_len=0, read_ids=None):
    """score_reduce (reduce_inl.h:71-160): fold the extension results of every active read, in order, into
    `best`.  hit_begin int64[n_active+1] (CSR), hit_score int32, hit_loc int32 (uint32 bits), hit_rc uint8."""
    n_active = hit_begin.numel() - 1
    assert hit_begin.dtype == torch.int64 and hit_score.dtype == torch.int32 and hit_loc.dtype == torch.int32 and hit_rc.dtype == torch.uint8
    check(lib().nvbio_hip_score_reduce(n_active, _vp(read_ids), _vp(hit_begin), _vp(hit_score), _vp(hit_loc), _vp(hit_rc),
                                       _vp(read_len), int(fixed_read_len), _vp(best.data), best.stride, current_stream_ptr()), "nvbio_hip_score_reduce")
    return best


def mapq(best, scheme, read_len=None, fixed_read_len=0, version=2, max_read_len=None):
    """BowtieMapq2 / BowtieMapq3 over all reads -> uint8[n]."""
    dev = best.data.device
    max_len = int(max_read_len or fixed_read_len or int(read_len.max()))
    table = torch.tensor([scheme.min_score(L) if L > 0 else 0 for L in range(max_len + 1)], dtype=torch.int32, device=dev)
    out = torch.empty(best.n, dtype=torch.uint8, device=dev)
    check(lib().nvbio_hip_mapq(int(version), int(scheme.m_match), int(bool(scheme.m_monotone)), _vp(table), best.n, _vp(best.data), best.stride,
                               _vp(read_len), int(fixed_read_len), _vp(out), current_stream_ptr()), "nvbio_hip_mapq")
    table.record_stream(torch.cuda.current_stream())      # read asynchronously by the kernel
    return out


PE_POLICY_FF, PE_POLICY_FR, PE_POLICY_RF, PE_POLICY_RR = 0, 1, 2, 3      # io::PairedEndPolicy (sequence.h:190-196)


def score_reduce_paired(best, best_o, hit_begin, hit_loc, hit_sink, hit_score, hit_rc, o_loc, o_sink, o_sink2, o_score, o_score2,
                        anchor, pe_policy=PE_POLICY_FR, pe_unpaired=True, score_limit=-(1 << 17) + 1, read_len=None, fixed_read_len=0, read_ids=None):
    """score_reduce_paired (reduce_inl.h:355-500): fold the paired extension results of every active read into
    best (anchor / mate-1 entries) and best_o (opposite / mate-2 entries)."""
    n_active = hit_begin.numel() - 1
    check(lib().nvbio_hip_score_reduce_paired(n_active, _vp(read_ids), _vp(hit_begin), _vp(hit_loc), _vp(hit_sink), _vp(hit_score), _vp(hit_rc),
                                              _vp(o_loc), _vp(o_sink), _vp(o_sink2), _vp(o_score), _vp(o_score2), _vp(read_len), int(fixed_read_len),
                                              int(anchor), int(pe_policy), int(bool(pe_unpaired)), int(score_limit),
                                              _vp(best.data), _vp(best_o.data), best.stride, current_stream_ptr()), "nvbio_hip_score_reduce_paired")
    return best, best_o


def mapq_paired(best, best_o, scheme, read_len=None, o_read_len=None, fixed_read_len=0, o_fixed_read_len=0, version=2, max_read_len=None):
    """BowtieMapq2 / BowtieMapq3 over BestPairedAlignments(best, best_o) -> uint8[n]."""
    dev = best.data.device
    max_len = int(max_read_len or max(fixed_read_len, o_fixed_read_len) or max(int(read_len.max()), int(o_read_len.max())))
    table = torch.tensor([scheme.min_score(L) if L > 0 else 0 for L in range(max_len + 1)], dtype=torch.int32, device=dev)
    out = torch.empty(best.n, dtype=torch.uint8, device=dev)
    check(lib().nvbio_hip_mapq_paired(int(version), int(scheme.m_match), int(bool(scheme.m_monotone)), _vp(table), best.n, _vp(best.data), _vp(best_o.data), best.stride,
                                      _vp(read_len), _vp(o_read_len), int(fixed_read_len), int(o_fixed_read_len), _vp(out), current_stream_ptr()), "nvbio_hip_mapq_paired")
    table.record_stream(torch.cuda.current_stream())
    return out


def opposite_mate_windows(hit_read_id, hit_rc, hit_loc, hit_score, best, best_o, scheme, anchor, genome_length,
                          a_read_len=None, o_read_len=None, a_fixed_len=0, o_fixed_len=0, pe_policy=PE_POLICY_FR,
                          min_frag_len=0, max_frag_len=500, pe_overlap=True, score_limit=-(1 << 17) + 1, max_read_len=None):
    """BestOppositeScoreStream::init_context (score_opposite_inl.h:92-200) over all scored anchor hits ->
    dict(valid uint8, min_score int32, read_rc uint8, genome_begin int32, genome_end int32)."""
    dev = hit_loc.device
    n = hit_loc.numel()
    max_len = int(max_read_len or max(a_fixed_len, o_fixed_len) or max(int(a_read_len.max()), int(o_read_len.max())))
    table = torch.tensor([scheme.min_score(L) if L > 0 else 0 for L in range(max_len + 1)], dtype=torch.int32, device=dev)
    out = dict(valid=torch.empty(n, dtype=torch.uint8, device=dev), min_score=torch.empty(n, dtype=torch.int32, device=dev),
               read_rc=torch.empty(n, dtype=torch.uint8, device=dev), genome_begin=torch.empty(n, dtype=torch.int32, device=dev),
               genome_end=torch.empty(n, dtype=torch.int32, device=dev))
    pp = PeParamsStruct(int(pe_policy), int(min_frag_len), int(max_frag_len), int(bool(pe_overlap)), int(score_limit), int(anchor), int(genome_length))
    check(lib().nvbio_hip_opposite_mate_windows(n, _vp(hit_read_id), _vp(hit_rc), _vp(hit_loc), _vp(hit_score), _vp(a_read_len), _vp(o_read_len),
                                                int(a_fixed_len), int(o_fixed_len), _vp(best.data), _vp(best_o.data), best.stride,
                                                int(scheme.m_match), _vp(table), int(scheme.text_gap_open()), int(scheme.text_gap_extension()), C.byref(pp),
                                                _vp(out["valid"]), _vp(out["min_score"]), _vp(out["read_rc"]), _vp(out["genome_begin"]), _vp(out["genome_end"]),
                                                current_stream_ptr()), "nvbio_hip_opposite_mate_windows")
    table.record_stream(torch.cuda.current_stream())
    return out
