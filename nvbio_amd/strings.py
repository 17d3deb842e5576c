"""Packed string sets in device memory (the reference's PackedStream / string-set inputs,
nvbio/basic/packedstream.h, nvbio/strings/string_set.h)."""
import ctypes as C

import torch

from ._lib import StringSetStruct


def pack_symbols(sym, bits, big_endian, pad_words=4):
    """Pack a uint8 symbol tensor into PackedStream words (int32 tensor holding the raw bits).
    Layout per nvbio/basic/packedstream_inl.h:336-400."""
    assert bits in (2, 4, 8)
    per = 32 // bits
    n = sym.numel()
    nw = (n + per - 1) // per + pad_words
    buf = torch.zeros(nw * per, dtype=torch.int64, device=sym.device)
    buf[:n] = sym.to(torch.int64) & ((1 << bits) - 1)
    buf = buf.view(nw, per)
    k = torch.arange(per, dtype=torch.int64, device=sym.device)
    sh = (32 - bits - k * bits) if big_endian else (k * bits)
    words = (buf << sh).sum(dim=1)
    return words.to(torch.int32)   # wraps to the raw 32-bit pattern


class PackedStringSet:
    """n strings inside one packed stream: string i = symbols [begin[i], begin[i]+length[i])."""

    def __init__(self, words, bits, big_endian, begin, length=None, fixed_length=0):
        assert words.dtype == torch.int32 and words.is_contiguous()
        assert begin.dtype == torch.int64 and begin.is_contiguous()
        if length is not None:
            assert length.dtype == torch.int32 and length.is_contiguous() and length.numel() == begin.numel()
        self.words, self.bits, self.big_endian = words, int(bits), int(bool(big_endian))
        self.begin, self.length, self.fixed_length = begin, length, int(fixed_length)

    def __len__(self):
        return self.begin.numel()

    def struct(self):
        s = StringSetStruct()
        s.words = self.words.data_ptr()
        s.n_words = self.words.numel()
        s.bits = self.bits
        s.big_endian = self.big_endian
        s.begin = self.begin.data_ptr()
        s.length = self.length.data_ptr() if self.length is not None else None
        s.fixed_length = self.fixed_length
        return s

    @staticmethod
    def from_host(words_u32, bits, big_endian, begin_u64, length_u32=None, fixed_length=0, device="cuda"):
        """Build from numpy arrays (uint32 words, uint64 begin, uint32 length)."""
        import numpy as np
        w = torch.from_numpy(np.ascontiguousarray(words_u32).view(np.int32)).to(device)
        b = torch.from_numpy(np.ascontiguousarray(begin_u64).view(np.int64)).to(device)
        ln = None
        if length_u32 is not None:
            ln = torch.from_numpy(np.ascontiguousarray(length_u32).view(np.int32)).to(device)
        return PackedStringSet(w, bits, big_endian, b, ln, fixed_length)
