"""Host layer mirroring nvbio::fm_index free functions and FMIndexFilter.

Reference names: rank / rank4 / match / locate / locate_ssa_iterator / lookup_ssa_iterator
(nvbio/fmindex/fmindex_inl.h), FMIndexFilter<device_tag,...>::rank / locate
(nvbio/fmindex/filter.h:139-203), FMIndexDataDevice (nvbio/io/fmindex/fmindex.h:294-362).
All compute happens in libnvbio_hip.so.
"""
import ctypes as C

import torch

from ._lib import lib, check, FMIndexStruct, current_stream_ptr


def _vp(t):
    return C.c_void_p(t.data_ptr())


class FMIndexDevice:
    """Device-resident FM-index in the reference's interleaved bwt|occ layout."""

    def __init__(self, length, primary, L2, bwt_occ, ssa=None, sa_int=16, ktab=None, ktab_k=0, dimer=None, dimer_consts=None, trimer=None):
        assert bwt_occ.dtype == torch.int32 and bwt_occ.is_contiguous()
        assert bwt_occ.data_ptr() % 32 == 0
        self.length, self.primary, self.L2 = int(length), int(primary), [int(x) for x in L2]
        self.bwt_occ, self.ssa, self.sa_int = bwt_occ, ssa, int(sa_int)
        self.ktab, self.ktab_k = ktab, int(ktab_k)
        self.dimer, self.dimer_consts = dimer, dimer_consts     # line-native two-symbol index + its header constants
        self.trimer = trimer                                    # three-symbol rank arrays on top of it

    def _copy(self, **kw):
        d = dict(length=self.length, primary=self.primary, L2=self.L2, bwt_occ=self.bwt_occ, ssa=self.ssa, sa_int=self.sa_int,
                 ktab=self.ktab, ktab_k=self.ktab_k, dimer=self.dimer, dimer_consts=self.dimer_consts, trimer=self.trimer)
        d.update(kw)
        return FMIndexDevice(**d)

    def with_dimer(self):
        """A copy of this index carrying the MI355X line-native two-symbol index (128-byte records, 1 byte per SA
        row; nvbio_amd/csrc/fmindex_dimer.h), built on the device: match / the seed mappers / locate then consume
        two symbols per HBM line.  Results are bit-identical."""
        L = lib()
        dev = self.bwt_occ.device
        nbytes = int(L.nvbio_hip_fm_dimer_index_bytes(self.length))
        buf = torch.empty(nbytes // 4 + 32, dtype=torch.int32, device=dev)
        off = (-buf.data_ptr() % 128) // 4
        dimer = buf[off:off + nbytes // 4]
        tb = int(L.nvbio_hip_fm_build_dimer_index_temp_bytes(self.length))
        temp = torch.empty(tb, dtype=torch.uint8, device=dev)
        s = self._copy(dimer=None, dimer_consts=None, ktab=None, ktab_k=0).struct()
        check(L.nvbio_hip_fm_build_dimer_index(C.byref(s), _vp(dimer), _vp(temp), tb, current_stream_ptr()), "nvbio_hip_fm_build_dimer_index")
        check(L.nvbio_hip_fm_attach_dimer_index(C.byref(s), _vp(dimer), current_stream_ptr()), "nvbio_hip_fm_attach_dimer_index")
        consts = (int(s.dimer_p1), int(s.dimer_fill1), [int(x) for x in s.dimer_S], [int(x) for x in s.dimer_T])
        del temp
        return self._copy(dimer=dimer, dimer_consts=consts)

    def without_dimer(self):
        return self._copy(dimer=None, dimer_consts=None, trimer=None)

    def with_trimer(self):
        """A copy carrying, on top of the two-symbol index (built if absent), the three-symbol rank arrays (10.7 bytes per SA
        row): backward search then consumes three symbols per step.  Results are bit-identical."""
        base = self if self.dimer is not None else self.with_dimer()
        L = lib()
        dev = self.bwt_occ.device
        nbytes = int(L.nvbio_hip_fm_trimer_index_bytes(self.length))
        buf = torch.empty(nbytes // 4 + 32, dtype=torch.int32, device=dev)
        off = (-buf.data_ptr() % 128) // 4
        tri = buf[off:off + nbytes // 4]
        tb = int(L.nvbio_hip_fm_build_trimer_index_temp_bytes(self.length))
        temp = torch.empty(tb, dtype=torch.uint8, device=dev)
        s = base._copy(dimer=None, dimer_consts=None, ktab=None, ktab_k=0, trimer=None).struct()
        check(L.nvbio_hip_fm_build_trimer_index(C.byref(s), _vp(tri), _vp(temp), tb, current_stream_ptr()), "nvbio_hip_fm_build_trimer_index")
        torch.cuda.current_stream().synchronize()
        del temp
        chk = base.struct()         # the header must describe THIS index (stale / foreign arrays are refused, not silently used)
        check(L.nvbio_hip_fm_attach_trimer_index(C.byref(chk), _vp(tri), current_stream_ptr()), "nvbio_hip_fm_attach_trimer_index")
        return base._copy(trimer=tri)

    def with_ktab(self, k=12):
        """A copy of this index carrying the k-mer table accelerator (4^k uint2 entries in HBM)."""
        tab = torch.empty((4 ** k, 2), dtype=torch.int32, device=self.bwt_occ.device)
        s = self.struct()
        check(lib().nvbio_hip_fm_build_ktab(C.byref(s), k, _vp(tab), current_stream_ptr()), "nvbio_hip_fm_build_ktab")
        return self._copy(ktab=tab, ktab_k=k)

    def with_dense_ssa(self, sa_int):
        """A copy of this index with the suffix array sampled every `sa_int` rows (power of two,
        1 = full SA), derived on the device by locating the sampled rows with the current SSA."""
        assert sa_int >= 1 and (sa_int & (sa_int - 1)) == 0
        rows = torch.arange(0, self.length + 1, sa_int, dtype=torch.int64, device=self.bwt_occ.device).to(torch.int32)
        ssa = locate(self, rows)
        return self._copy(ssa=ssa, sa_int=sa_int)

    def with_dense_ssa_native(self, sa_int):
        """with_dense_ssa through the library's own builder (nvbio_hip_fm_build_dense_ssa): no row array, 4 bytes per kept row"""
        assert sa_int >= 1 and (sa_int & (sa_int - 1)) == 0 and sa_int <= self.sa_int and self.ssa is not None
        n_out = int(lib().nvbio_hip_fm_dense_ssa_entries(self.length, sa_int))
        ssa = torch.empty(n_out, dtype=torch.int32, device=self.bwt_occ.device)
        s = self.struct()
        check(lib().nvbio_hip_fm_build_dense_ssa(C.byref(s), sa_int, _vp(ssa), current_stream_ptr()), "nvbio_hip_fm_build_dense_ssa")
        return self._copy(ssa=ssa, sa_int=sa_int)

    def hbm_default(self, budget_bytes=0, policy=None):
        """The index this device's HBM is there for -- nvbio::fm_index_hbm::build of the C++ host layer (include/nvbio_hip/fmindex.h), same
        policy: the line-native two-symbol records, the densest suffix array and the largest k-mer table that fit `budget_bytes` (0 = 35 % of
        the memory that is free now).  NVBIO_HIP_INDEX=lean|line_native|rich overrides.  Results stay bit-identical.
        -> (index, description dict)"""
        import os
        from ._lib import device_mem_info
        policy = policy or os.environ.get("NVBIO_HIP_INDEX", "auto")
        idx = self
        if policy != "lean":
            free_b, _, idle_b = device_mem_info()
            free_b += torch.cuda.memory_reserved(self.bwt_occ.device) - torch.cuda.memory_allocated(self.bwt_occ.device)   # what torch's allocator would reuse
            budget = budget_bytes or int((free_b + idle_b) * 0.35)
            if policy == "rich":
                budget = 1 << 62
            L = lib()
            need = int(L.nvbio_hip_fm_dimer_index_bytes(self.length)) + int(L.nvbio_hip_fm_build_dimer_index_temp_bytes(self.length))
            if idx.dimer is None and need <= budget:
                idx = idx.with_dimer(); budget -= int(L.nvbio_hip_fm_dimer_index_bytes(self.length))
            if policy != "line_native":
                if idx.ssa is not None:                  # the densest suffix array that fits, sized first
                    s = 1
                    while s < idx.sa_int:
                        need = int(L.nvbio_hip_fm_dense_ssa_entries(self.length, s)) * 4
                        if need <= budget:
                            idx = idx.with_dense_ssa_native(s); budget -= need
                            break
                        if policy == "rich":
                            raise RuntimeError("hbm_default: the whole suffix array does not fit")
                        s *= 2
                if idx.ktab is None:                     # every k-mer's range, the largest k that fits and that the text can fill
                    kmax = 8
                    while kmax < 16 and (1 << (2 * kmax)) < self.length:
                        kmax += 1
                    for k in range(kmax, 7, -1):
                        if (8 << (2 * k)) <= budget:
                            idx = idx.with_ktab(k); budget -= 8 << (2 * k)
                            break
        return idx, {"line_native": idx.dimer is not None, "ktab_k": idx.ktab_k if idx.ktab is not None else 0, "sa_int": idx.sa_int, "policy": policy}

    def struct(self):
        s = FMIndexStruct()
        s.length, s.primary, s.sa_int = self.length, self.primary, self.sa_int
        for i in range(5):
            s.L2[i] = self.L2[i]
        s.bwt_occ = self.bwt_occ.data_ptr()
        s.ssa = self.ssa.data_ptr() if self.ssa is not None else None
        s.ktab = self.ktab.data_ptr() if self.ktab is not None else None
        s.ktab_k = self.ktab_k if self.ktab is not None else 0
        if self.dimer is not None:
            s.dimer = self.dimer.data_ptr()
            s.dimer_p1, s.dimer_fill1 = self.dimer_consts[0], self.dimer_consts[1]
            for i in range(4):
                s.dimer_S[i], s.dimer_T[i] = self.dimer_consts[2][i], self.dimer_consts[3][i]
        else:
            s.dimer = None
        s.trimer = self.trimer.data_ptr() if (self.trimer is not None and self.dimer is not None) else None
        return s

    @staticmethod
    def from_host(host_index, device="cuda"):
        """host_index: any object with length, primary, L2, bwt_occ (uint32 np), ssa (uint32 np), sa_int."""
        import numpy as np
        bo = torch.from_numpy(np.ascontiguousarray(host_index.bwt_occ).view(np.int32)).to(device)
        ssa = torch.from_numpy(np.ascontiguousarray(host_index.ssa).view(np.int32)).to(device)
        return FMIndexDevice(host_index.length, host_index.primary, host_index.L2, bo, ssa, host_index.sa_int)


def rank(fmi, k, c):
    n = k.numel()
    out = torch.empty(n, dtype=torch.int32, device=k.device)
    s = fmi.struct()
    check(lib().nvbio_hip_fm_rank(C.byref(s), _vp(k), _vp(c), n, _vp(out), current_stream_ptr()), "nvbio_hip_fm_rank")
    return out


def rank4(fmi, k):
    n = k.numel()
    out = torch.empty((n, 4), dtype=torch.int32, device=k.device)
    s = fmi.struct()
    check(lib().nvbio_hip_fm_rank4(C.byref(s), _vp(k), n, _vp(out), current_stream_ptr()), "nvbio_hip_fm_rank4")
    return out


def rank_range(fmi, ranges, c):
    n = ranges.numel() // 2
    out = torch.empty((n, 2), dtype=torch.int32, device=ranges.device)
    s = fmi.struct()
    check(lib().nvbio_hip_fm_rank_range(C.byref(s), _vp(ranges), _vp(c), n, _vp(out), current_stream_ptr()), "nvbio_hip_fm_rank_range")
    return out


def match(fmi, seeds, out=None):
    n = len(seeds)
    if out is None:
        out = torch.empty((n, 2), dtype=torch.int32, device=seeds.words.device)
    s, ss = fmi.struct(), seeds.struct()
    check(lib().nvbio_hip_fm_match(C.byref(s), C.byref(ss), n, _vp(out), current_stream_ptr()), "nvbio_hip_fm_match")
    return out


def locate(fmi, rows, out=None):
    n = rows.numel()
    if out is None:
        out = torch.empty(n, dtype=torch.int32, device=rows.device)
    s = fmi.struct()
    check(lib().nvbio_hip_fm_locate(C.byref(s), _vp(rows), n, _vp(out), current_stream_ptr()), "nvbio_hip_fm_locate")
    return out


def locate_ssa_iterator(fmi, rows):
    n = rows.numel()
    out = torch.empty((n, 2), dtype=torch.int32, device=rows.device)
    s = fmi.struct()
    check(lib().nvbio_hip_fm_locate_ssa_iterator(C.byref(s), _vp(rows), n, _vp(out), current_stream_ptr()), "nvbio_hip_fm_locate_ssa_iterator")
    return out


def lookup_ssa_iterator(fmi, its):
    n = its.numel() // 2
    out = torch.empty(n, dtype=torch.int32, device=its.device)
    s = fmi.struct()
    check(lib().nvbio_hip_fm_lookup_ssa_iterator(C.byref(s), _vp(its), n, _vp(out), current_stream_ptr()), "nvbio_hip_fm_lookup_ssa_iterator")
    return out


class FMIndexFilter:
    """FMIndexFilter<device_tag, fm_index> (nvbio/fmindex/filter.h:139-203)."""

    def __init__(self):
        self.n_queries = 0
        self.n_occurrences = 0
        self.ranges = None
        self.slots = None
        self.index = None

    def rank(self, index, string_set):
        n = len(string_set)
        dev = string_set.words.device
        self.index, self.n_queries = index, n
        self.ranges = torch.empty((n, 2), dtype=torch.int32, device=dev)
        self.slots = torch.empty(n, dtype=torch.int64, device=dev)
        tb = int(lib().nvbio_hip_fm_filter_temp_bytes(n))
        temp = torch.empty(tb, dtype=torch.uint8, device=dev)
        s, ss = index.struct(), string_set.struct()
        check(lib().nvbio_hip_fm_filter_rank(C.byref(s), C.byref(ss), n, _vp(self.ranges), _vp(self.slots),
                                            _vp(temp), tb, current_stream_ptr()), "nvbio_hip_fm_filter_rank")
        self.n_occurrences = int(self.slots[n - 1].item()) if n else 0
        return self.n_occurrences

    def locate(self, begin, end, hits=None):
        if hits is None:
            hits = torch.empty((end - begin, 2), dtype=torch.int32, device=self.ranges.device)
        s = self.index.struct()
        check(lib().nvbio_hip_fm_filter_locate(C.byref(s), _vp(self.ranges), _vp(self.slots), self.n_queries,
                                              begin, end, _vp(hits), current_stream_ptr()), "nvbio_hip_fm_filter_locate")
        return hits


def build_bwt_occ(n, bwt_words):
    """Device build_occurrence_table<2,64> + interleave.  bwt_words: int32 tensor, 4*ceil(n/64) words,
    big-endian 2-bit BWT.  Returns (bwt_occ int32[8*ceil(n/64)], L2 list of 5 ints)."""
    n_blocks = (n + 63) // 64
    assert bwt_words.numel() >= 4 * n_blocks and bwt_words.data_ptr() % 16 == 0
    dev = bwt_words.device
    out = torch.empty(8 * n_blocks, dtype=torch.int32, device=dev)
    L2 = torch.empty(5, dtype=torch.int32, device=dev)
    tb = int(lib().nvbio_hip_build_bwt_occ_temp_bytes(n))
    temp = torch.empty(tb, dtype=torch.uint8, device=dev)
    check(lib().nvbio_hip_build_bwt_occ(n, _vp(bwt_words), _vp(out), _vp(L2), _vp(temp), tb, current_stream_ptr()), "nvbio_hip_build_bwt_occ")
    L2h = [int(x) & 0xFFFFFFFF for x in L2.cpu().tolist()]
    return out, L2h
