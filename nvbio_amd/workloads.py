"""Synthetic workloads of BASELINE.json / SURVEY.md 8(d), generated with torch on any device.

Only data generation lives here (no alignment / index arithmetic)."""
import torch

from .strings import PackedStringSet, pack_symbols


def _pack_chunked(sym_flat, bits, big_endian, chunk_syms=1 << 26):
    """pack_symbols over a long symbol tensor in word-aligned chunks (bounds temp memory)."""
    per = 32 // bits
    n = sym_flat.numel()
    chunk_syms -= chunk_syms % per
    parts = []
    for s in range(0, n, chunk_syms):
        e = min(n, s + chunk_syms)
        parts.append(pack_symbols(sym_flat[s:e], bits, big_endian, pad_words=0))
    parts.append(torch.zeros(4, dtype=torch.int32, device=sym_flat.device))
    return torch.cat(parts)


def make_sw_symbols(n, read_len=100, ref_len=150, seed=0x5EED0001, device="cpu",
                    sub_rate=0.04, indel_frac=0.3, n_frac=0.01, offset=7):
    """Reads/refs of the banded-SW configs: each read is a copy of
    ref[offset : offset+read_len) with `sub_rate` substitutions, an optional 1-3 bp indel
    (`indel_frac` of the reads) and, for `n_frac` of the reads, one N (code 4).
    Returns (reads uint8 [n,read_len] in 0..4, refs uint8 [n,ref_len] in 0..3)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    assert offset + read_len + 3 <= ref_len
    ref = torch.randint(0, 4, (n, ref_len), dtype=torch.uint8, generator=g, device=device)
    k = torch.arange(read_len, device=device, dtype=torch.int64).unsqueeze(0)
    has = torch.rand(n, generator=g, device=device) < indel_frac
    is_ins = torch.rand(n, generator=g, device=device) < 0.5
    ilen = torch.randint(1, 4, (n,), generator=g, device=device)
    pos = torch.randint(10, read_len - 10, (n,), generator=g, device=device)
    has, is_ins, ilen, pos = has.unsqueeze(1), is_ins.unsqueeze(1), ilen.unsqueeze(1), pos.unsqueeze(1)
    shift_del = (k >= pos).to(torch.int64) * ilen
    shift_ins = -((k >= pos + ilen).to(torch.int64) * ilen)
    shift = torch.where(has & ~is_ins, shift_del, torch.where(has & is_ins, shift_ins, torch.zeros_like(shift_del)))
    src = offset + k + shift
    read = torch.gather(ref, 1, src)
    rnd = torch.randint(0, 4, (n, read_len), dtype=torch.uint8, generator=g, device=device)
    ins_mask = has & is_ins & (k >= pos) & (k < pos + ilen)
    read = torch.where(ins_mask, rnd, read)
    sub = torch.rand((n, read_len), generator=g, device=device) < sub_rate
    delta = torch.randint(1, 4, (n, read_len), dtype=torch.uint8, generator=g, device=device)
    read = torch.where(sub, (read + delta) & 3, read)
    has_n = torch.rand(n, generator=g, device=device) < n_frac
    npos = torch.randint(0, read_len, (n,), generator=g, device=device)
    nmask = has_n.unsqueeze(1) & (k == npos.unsqueeze(1))
    read = torch.where(nmask, torch.full_like(read, 4), read)
    return read, ref


def make_sw_batch(n, read_len=100, ref_len=150, seed=0x5EED0001, device="cpu", **kw):
    """Packed string sets of the banded-SW configs: reads 4-bit big-endian (the nvBowtie /
    sw-benchmark read format), reference windows 2-bit little-endian (sw-benchmark.cu:73-74)."""
    read, ref = make_sw_symbols(n, read_len, ref_len, seed, device, **kw)
    pw = _pack_chunked(read.reshape(-1), 4, True)
    tw = _pack_chunked(ref.reshape(-1), 2, False)
    idx = torch.arange(n, dtype=torch.int64, device=device)
    patterns = PackedStringSet(pw, 4, True, idx * read_len, None, read_len)
    texts = PackedStringSet(tw, 2, False, idx * ref_len, None, ref_len)
    return patterns, texts


def make_random_bwt(n, seed=0x5EED0003, device="cpu", chunk=1 << 28):
    """A uniform i.i.d. 2-bit symbol string of length n packed big-endian, padded to whole
    64-symbol blocks: statistically what the BWT of an i.i.d. genome looks like.  Used for the
    rank() bandwidth configuration."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    n_words = ((n + 63) // 64) * 4
    parts = []
    for s in range(0, n_words, chunk):
        e = min(n_words, s + chunk)
        parts.append(torch.randint(-(1 << 31), (1 << 31), (e - s,), dtype=torch.int64, generator=g, device=device).to(torch.int32))
    words = torch.cat(parts)
    # zero the padding past n
    rem = n % 64
    if rem:
        last = (n // 64) * 4
        for w in range(4):
            lo = w * 16
            keep = max(0, min(16, rem - lo))
            mask = 0 if keep == 0 else ((0xFFFFFFFF << (32 - 2 * keep)) & 0xFFFFFFFF)
            v = int(words[last + w].item()) & 0xFFFFFFFF & mask
            words[last + w] = v - (1 << 32) if v >= (1 << 31) else v
    return words
