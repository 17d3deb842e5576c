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


# ----------------------------------------------------------------------------------------------
# Synthetic FM-index construction on the device (workload tooling for the FM-index configs; index
# construction itself is out of the hot path's scope -- the reference does it offline with nvBWT).
# ----------------------------------------------------------------------------------------------
_SORT_LIMIT = (1 << 31) - 1          # torch.sort / torch.nonzero refuse more elements than INT_MAX


def _sort_big(key, limit=None):
    """torch.sort(key) -> (sorted, order) for tensors of any length: above `limit` elements the
    keys are split into equal-width value buckets (the keys here are near-uniform), each bucket's
    positions are collected chunk-wise and sorted on its own; buckets concatenate in order."""
    limit = limit or _SORT_LIMIT
    n = key.numel()
    if n <= limit:
        return torch.sort(key)
    dev = key.device
    lo, hi = int(key.min().item()), int(key.max().item()) + 1
    nb = 4 * ((n + limit - 1) // limit)
    out_k = torch.empty(n, dtype=key.dtype, device=dev)
    out_i = torch.empty(n, dtype=torch.int64, device=dev)
    w = (hi - lo + nb - 1) // nb
    at = 0
    for b in range(nb):
        blo, bhi = lo + b * w, min(hi, lo + (b + 1) * w)
        if blo >= bhi:
            continue
        parts = []
        for c0 in range(0, n, limit):
            kc = key[c0:c0 + limit]
            parts.append(torch.nonzero((kc >= blo) & (kc < bhi)).squeeze(1) + c0)
        idx = torch.cat(parts) if len(parts) > 1 else parts[0]
        del parts
        if idx.numel() > limit:
            raise RuntimeError("_sort_big: bucket too large; keys are not near-uniform")
        sk, so = torch.sort(key[idx])
        m = idx.numel()
        out_k[at:at + m] = sk
        out_i[at:at + m] = idx[so]
        at += m
        del idx, sk, so
    assert at == n
    return out_k, out_i


def suffix_array(text, h0=16):
    """Suffix array of a device symbol tensor (values 0..3) by prefix doubling with torch sorts.
    Returns int64 [n+1] with the reference's padding convention (bwt.h:36-45): row 0 is the empty
    '$' suffix (SA[0] = n).  For i.i.d. text two or three rounds suffice (h = 16, 32, 64)."""
    n = text.numel()
    dev = text.device
    d = torch.zeros(n + 1 + h0, dtype=torch.int64, device=dev)
    d[:n] = text.to(torch.int64) + 1                      # 0 = past the end (sorts first)
    key = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    for k in range(h0):
        key = key * 5 + d[k:k + n + 1]
    del d
    h = h0
    while True:
        skey, order = _sort_big(key)
        del key
        newr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        newr[1:] = torch.cumsum((skey[1:] != skey[:-1]).to(torch.int64), 0)
        del skey
        done = int(newr[-1].item()) == n
        if done:
            return order
        rank = torch.empty(n + 1, dtype=torch.int64, device=dev)
        rank[order] = newr
        del newr, order
        nxt = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        if h <= n:
            nxt[: n + 1 - h] = rank[h:] + 1
        key = rank * (n + 2) + nxt
        del rank, nxt
        h *= 2


def build_fm_index(text, sa_int=16, keep_sa=False):
    """FM-index of a device symbol tensor in the reference's production layout: BWT from the SA as
    gen_bwt_from_sa does (bwt.h:47-60), big-endian 2-bit packing, occurrence table + interleave by
    the device build_occurrence_table kernel, SSA sampled every sa_int rows with ssa[0] = -1
    (ssa_inl.h:263-309).  Returns nvbio_amd.FMIndexDevice (and the SA if keep_sa)."""
    from .fmindex import FMIndexDevice, build_bwt_occ
    n = text.numel()
    dev = text.device
    sa = suffix_array(text)
    primary = int(torch.argmin(sa).item())                    # the row whose suffix is the whole text
    prev = torch.cat([sa[:primary], sa[primary + 1:]]) - 1     # n rows, the '$' row dropped
    bwt = text[prev]
    del prev
    n_blocks = (n + 63) // 64
    pad = torch.zeros(n_blocks * 64, dtype=torch.uint8, device=dev)
    pad[:n] = bwt
    del bwt
    words = _pack_chunked(pad, 2, True)[: n_blocks * 4].contiguous()
    del pad
    bwt_occ, L2 = build_bwt_occ(n, words)
    ssa = sa[::sa_int].to(torch.int32).contiguous()
    ssa[0] = -1
    fmi = FMIndexDevice(n, primary, L2, bwt_occ, ssa, sa_int)
    return (fmi, sa) if keep_sa else fmi


def make_seeds(text, n_seeds, seed_len=22, seed=0x5EED0004, random_frac=0.1, bits=2, big_endian=True):
    """Seeds of the FM-index configs: 90 % sampled from the genome,
    10 % random; packed as one fixed-length string set."""
    dev = text.device
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    n = text.numel()
    pos = torch.randint(0, n - seed_len, (n_seeds,), generator=g, device=dev)
    idx = pos.unsqueeze(1) + torch.arange(seed_len, device=dev).unsqueeze(0)
    sym = text[idx]
    rnd = torch.rand(n_seeds, generator=g, device=dev) < random_frac
    rsym = torch.randint(0, 4, (n_seeds, seed_len), dtype=torch.uint8, generator=g, device=dev)
    sym = torch.where(rnd.unsqueeze(1), rsym, sym)
    words = _pack_chunked(sym.reshape(-1), bits, big_endian)
    begin = torch.arange(n_seeds, dtype=torch.int64, device=dev) * seed_len
    return PackedStringSet(words, bits, big_endian, begin, None, seed_len)
