"""Host layer of nvBowtie's exact seed mapping stage (nvBowtie/bowtie2/cuda/mapping.h: map(),
mapping_inl.h:511-592): reads -> per-read SeedHit sets, computed by libnvbio_hip.so."""
import ctypes as C

import numpy as np
import torch

from ._lib import lib, check, MapParamsStruct, current_stream_ptr

LINEAR_FUNC, LOG_FUNC, SQRT_FUNC = 0, 1, 2      # SimpleFunc::Type (func.h:41)


def simple_func(ftype, k, m, x):
    """nvBowtie's SimpleFunc(type,k,m)(x) = int32(k + m * f(float(x))) in single precision (func.h:47-52)."""
    xf = np.float32(x)
    fx = np.log(xf, dtype=np.float32) if ftype == LOG_FUNC else np.sqrt(xf, dtype=np.float32) if ftype == SQRT_FUNC else xf
    return int(np.float32(np.float32(k) + np.float32(np.float32(m) * fx)))


class MappingParams:
    """The fields of nvBowtie's Params that the mapping stage reads (params.h:100-120); defaults are
    its end-to-end defaults (params.cpp:120-163): seed_len 22, seed_freq = 1 + 1.15*sqrt(len),
    max_hits 100, max_reseed 2, rep_seeds 300."""

    def __init__(self, seed_len=22, seed_freq=(SQRT_FUNC, 1.0, 1.15), min_read_len=12, max_hits=100,
                 max_reseed=2, rep_seeds=300):
        self.seed_len, self.seed_freq, self.min_read_len = seed_len, seed_freq, min_read_len
        self.max_hits, self.max_reseed, self.rep_seeds = max_hits, max_reseed, rep_seeds

    def seed_freq_table(self, max_read_len, device):
        t = np.array([max(simple_func(*self.seed_freq, x), 0) if x > 0 else 0 for x in range(max_read_len + 1)], dtype=np.uint32)
        return torch.from_numpy(t.view(np.int32)).to(device)

    def struct(self, retry=0, fw=True, rc=True):
        return MapParamsStruct(self.seed_len, self.min_read_len, self.max_hits, self.max_reseed, retry, self.rep_seeds, int(fw), int(rc))


def map_exact(fmi, reads, params, max_read_len, retry=0, fw=True, rc=True, in_queue=None, hits_stride=None):
    """Returns (hits int64[n_reads, stride] holding the SeedHit word pairs, counts int32[n_reads],
    reseed uint8[n_queue])."""
    dev = reads.words.device
    n_reads = len(reads)
    n = in_queue.numel() if in_queue is not None else n_reads
    if hits_stride is None:
        min_freq = max(1, min(simple_func(*params.seed_freq, x) for x in range(max(params.min_read_len, 1), max_read_len + 1)))
        hits_stride = min(params.max_hits, 2 * (max_read_len // min_freq + 1))
    sf = params.seed_freq_table(max_read_len, dev)
    hits = torch.zeros((n_reads, hits_stride), dtype=torch.int64, device=dev)
    counts = torch.zeros(n_reads, dtype=torch.int32, device=dev)
    reseed = torch.zeros(n, dtype=torch.uint8, device=dev)
    s, rs, mp = fmi.struct(), reads.struct(), params.struct(retry, fw, rc)
    q = C.c_void_p(in_queue.data_ptr()) if in_queue is not None else None
    check(lib().nvbio_hip_map_exact(C.byref(s), C.byref(rs), q, n, C.byref(mp), C.c_void_p(sf.data_ptr()),
                                    C.c_void_p(hits.data_ptr()), hits_stride, C.c_void_p(counts.data_ptr()),
                                    C.c_void_p(reseed.data_ptr()), current_stream_ptr()), "nvbio_hip_map_exact")
    return hits, counts, reseed


EXACT_MAPPING, APPROX_MAPPING, CASE_PRUNING_MAPPING = 0, 1, 2      # MappingAlgorithm (mapping_inl.h:118-123)


def map_seeds(fmi, rfmi, reads, params, max_read_len, allow_sub=0, subseed_len=0, retry=0, fw=True, rc=True, in_queue=None, hits_stride=64, algorithm=None):
    """nvBowtie's map() with the algorithm choice of map_t (mapping_inl.h:809-843): allow_sub == 0 -> exact;
    allow_sub and subseed_len == 0 -> case pruning (uses rfmi, the index of the reversed genome);
    allow_sub and subseed_len > 0 -> exact subseed + one mismatch in the rest.  Returns (hits, counts, reseed)
    as map_exact does; index_dir = 1 marks hits found on rfmi."""
    if algorithm is None:             # (the all-mapping driver names its mapper itself: map_exact or map_approx, aligner_all.h:177-212)
        algorithm = EXACT_MAPPING if not allow_sub else (CASE_PRUNING_MAPPING if subseed_len == 0 else APPROX_MAPPING)
    dev = reads.words.device
    n_reads = len(reads)
    n = in_queue.numel() if in_queue is not None else n_reads
    sf = params.seed_freq_table(max_read_len, dev)
    hits = torch.zeros((n_reads, hits_stride), dtype=torch.int64, device=dev)
    counts = torch.zeros(n_reads, dtype=torch.int32, device=dev)
    reseed = torch.zeros(n, dtype=torch.uint8, device=dev)
    s, rs, mp = fmi.struct(), reads.struct(), params.struct(retry, fw, rc)
    r = rfmi.struct() if rfmi is not None else None
    q = C.c_void_p(in_queue.data_ptr()) if in_queue is not None else None
    check(lib().nvbio_hip_map(algorithm, int(subseed_len), C.byref(s), C.byref(r) if r is not None else None, C.byref(rs), q, n, C.byref(mp),
                              C.c_void_p(sf.data_ptr()), C.c_void_p(hits.data_ptr()), hits_stride, C.c_void_p(counts.data_ptr()),
                              C.c_void_p(reseed.data_ptr()), current_stream_ptr()), "nvbio_hip_map")
    return hits, counts, reseed


def unpack_seed_hits(words):
    """SeedHit word pairs (int64, little-endian: low word = range_begin) -> dict of int64 arrays."""
    w = words.to(torch.int64)
    lo, hi = w & 0xFFFFFFFF, (w >> 32) & 0xFFFFFFFF
    return {"range_begin": lo, "range_delta": hi & 0xFFFFF, "pos_in_read": (hi >> 20) & 0x3FF, "rc": (hi >> 30) & 1, "index_dir": (hi >> 31) & 1}
