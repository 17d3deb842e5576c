"""On-disk formats either side of the hot path, as the reference reads / writes them:

  <prefix>.bwt / .rbwt   [uint32 primary][uint32 cumFreq x4][uint32 BWT words: 2-bit big-endian, '$' removed]
                         load_bwt (nvbio/io/fmindex/fmindex_impl.cu:119-178), save_bwt (nvBWT/nvBWT.cu:312-331)
  <prefix>.sa / .rsa     [primary][cumFreq x4][sa_intv][seq_length][uint32 ssa[1..]]   (ssa[0] = -1 is implied)
                         load_sa (fmindex_impl.cu:180-262), save_ssa (nvBWT.cu:336-353)
  <prefix>.wpac          [uint64 seq_length][uint32 words: 2-bit big-endian genome]      (nvBWT.cu:222-248, sequence_pac.cpp:94-145)
  <prefix>.pac           BWA's byte-packed genome, 4 symbols per byte big-endian, trailing length byte
                         (nvBWT.cu:253-298, sequence_pac.cpp:147-190)

File parsing is host work (numpy); the occurrence table is built on the device by libnvbio_hip.so
(nvbio_hip_build_bwt_occ), as FMIndexDataHost::load does on the host (fmindex_impl.cu:264-328)."""
import os

import numpy as np
import torch

from .fmindex import FMIndexDevice, build_bwt_occ

FORWARD, REVERSE, SA = 0x02, 0x04, 0x10          # FMIndexData flags (nvbio/io/fmindex/fmindex.h:86-88)
SA_INT, OCC_INT = 16, 64


class FileMismatch(RuntimeError):
    pass


def read_bwt(path):
    """-> (primary, cum_freq[4], seq_length, bwt_words uint32[4*ceil(n/64)] zero padded)"""
    raw = np.fromfile(path, dtype=np.uint32)
    if raw.size < 5:
        raise IOError("failed reading bwt \"%s\"" % path)
    primary, cum = int(raw[0]), raw[1:5].copy()
    n = int(cum[3])                                   # the sum of the frequencies gives the total length
    seq_words = (n + 15) // 16
    padded = (seq_words + 3) // 4 * 4
    body = raw[5:]
    if (body.size + 3) // 4 * 4 != padded:
        raise IOError("failed reading bwt \"%s\"" % path)
    words = np.zeros(max(padded, 4 * ((n + 63) // 64)), dtype=np.uint32)
    words[:body.size] = body
    return primary, cum, n, words


def write_bwt(path, primary, cum_freq, bwt_words, seq_length):
    with open(path, "wb") as f:
        np.array([primary], dtype=np.uint32).tofile(f)
        np.asarray(cum_freq, dtype=np.uint32)[:4].tofile(f)
        np.asarray(bwt_words, dtype=np.uint32)[:(seq_length + 15) // 16].tofile(f)


def read_sa(path, seq_length, primary, sa_int=SA_INT):
    """-> ssa uint32[(n + sa_int) / sa_int] with ssa[0] = 0xFFFFFFFF"""
    raw = np.fromfile(path, dtype=np.uint32)
    if raw.size < 7:
        raise IOError("failed reading SSA \"%s\"" % path)
    if int(raw[0]) != primary:
        raise FileMismatch("SA file mismatch \"%s\": expected primary %u, got %u" % (path, primary, int(raw[0])))
    if int(raw[5]) != sa_int:
        raise FileMismatch("unsupported SA interval (found %u, expected %u)" % (int(raw[5]), sa_int))
    if int(raw[6]) != seq_length:
        raise FileMismatch("SA file mismatch \"%s\": expected length %u, got %u" % (path, seq_length, int(raw[6])))
    size = (seq_length + sa_int) // sa_int
    if raw.size - 7 < size - 1:
        raise IOError("failed reading SSA \"%s\"" % path)
    ssa = np.empty(size, dtype=np.uint32)
    ssa[0] = 0xFFFFFFFF
    ssa[1:] = raw[7:7 + size - 1]
    return ssa


def write_sa(path, primary, cum_freq, ssa, seq_length, sa_int=SA_INT):
    with open(path, "wb") as f:
        np.array([primary], dtype=np.uint32).tofile(f)
        np.asarray(cum_freq, dtype=np.uint32)[:4].tofile(f)
        np.array([sa_int, seq_length], dtype=np.uint32).tofile(f)
        np.asarray(ssa, dtype=np.uint32)[1:].tofile(f)


def read_wpac(path):
    """-> (seq_length, uint32 words, 2-bit big-endian)"""
    with open(path, "rb") as f:
        n = int(np.fromfile(f, dtype=np.uint64, count=1)[0])
        words = np.fromfile(f, dtype=np.uint32, count=(n + 15) // 16)
    if words.size != (n + 15) // 16:
        raise IOError("failed reading %s" % path)
    return n, words


def write_wpac(path, seq_length, words):
    with open(path, "wb") as f:
        np.array([seq_length], dtype=np.uint64).tofile(f)
        np.asarray(words, dtype=np.uint32)[:(seq_length + 15) // 16].tofile(f)


def read_pac(path):
    """BWA .pac -> (seq_length, uint32 words, 2-bit big-endian: the layout of the .wpac words)"""
    raw = np.fromfile(path, dtype=np.uint8)
    if raw.size < 2:
        raise IOError("failed reading %s" % path)
    n = (raw.size - 2) * 4 + int(raw[-1])
    nbytes = (n + 3) // 4
    body = np.zeros((nbytes + 3) // 4 * 4, dtype=np.uint8)
    body[:nbytes] = raw[:nbytes]
    # byte b holds symbols 4b..4b+3 from its top bits down; a big-endian word is the same order over 4 bytes
    words = body.view(">u4").astype(np.uint32)
    return n, words


def write_pac(path, seq_length, words):
    w = np.asarray(words, dtype=np.uint32)[:(seq_length + 15) // 16]
    body = w.astype(">u4").view(np.uint8)[:(seq_length + 3) // 4]
    with open(path, "wb") as f:
        body.tofile(f)
        if seq_length % 4 == 0:                       # the file size is always l_pac/4 + 1 + 1 (nvBWT.cu:285-294)
            np.array([0], dtype=np.uint8).tofile(f)
        np.array([seq_length % 4], dtype=np.uint8).tofile(f)


def load_genome(prefix):
    """sequence_pac.cpp:68-80: <prefix>.wpac if present, else <prefix>.pac"""
    if os.path.exists(prefix + ".wpac"):
        return read_wpac(prefix + ".wpac")
    return read_pac(prefix + ".pac")


class BNTSeq:
    """The .ann / .amb pair of a BWA-style reference (nvbio/basic/bnt.h, bnt.cpp:83-163; format of BWA 0.6.1): sequence names,
    comments, offsets and lengths in the concatenated genome, and the runs of ambiguous symbols that were replaced when packing."""

    def __init__(self, l_pac, seed, names, annos, gis, offsets, lengths, n_ambs, holes):
        self.l_pac, self.seed, self.names, self.annos, self.gis = int(l_pac), int(seed), list(names), list(annos), list(gis)
        self.offsets, self.lengths, self.n_ambs, self.holes = list(offsets), list(lengths), list(n_ambs), list(holes)

    @property
    def n_seqs(self):
        return len(self.names)

    def sequence_index(self):
        """SequenceData's sequence index as BNTLoader builds it (sequence_pac.cpp:214-226): [0, offset_i + len_i ...]"""
        return [0] + [o + l for o, l in zip(self.offsets, self.lengths)]

    def locate(self, pos):
        """genome coordinate -> (sequence number, coordinate inside it)"""
        idx = self.sequence_index()
        k = int(np.searchsorted(np.asarray(idx, np.int64), pos, side="right")) - 1
        k = min(max(k, 0), self.n_seqs - 1)
        return k, int(pos) - idx[k]


def read_bns(prefix):
    """load_bns (bnt.cpp:83-163)"""
    with open(prefix + ".ann") as f:
        l_pac, n_seqs, seed = f.readline().split()[:3]
        names, annos, gis, offsets, lengths, n_ambs = [], [], [], [], [], []
        for _ in range(int(n_seqs)):
            head = f.readline().rstrip("\n").split(" ", 2)
            gis.append(int(head[0])); names.append(head[1]); annos.append(head[2].lstrip(" ") if len(head) > 2 else "")
            o, l, a = f.readline().split()[:3]
            offsets.append(int(o)); lengths.append(int(l)); n_ambs.append(int(a))
    holes = []
    with open(prefix + ".amb") as f:
        lp, ns, nh = f.readline().split()[:3]
        if int(lp) != int(l_pac) or int(ns) != int(n_seqs):
            raise FileMismatch("%s.ann and %s.amb describe different references" % (prefix, prefix))
        for _ in range(int(nh)):
            o, l, c = f.readline().split()[:3]
            holes.append((int(o), int(l), c))
    return BNTSeq(l_pac, seed, names, annos, gis, offsets, lengths, n_ambs, holes)


def write_bns(prefix, names, lengths, annos=None, holes=(), seed=11):
    """The .ann / .amb files of sequences laid end to end (what `bwa index` / nvBWT's front end write)."""
    offsets = np.concatenate([[0], np.cumsum(lengths)[:-1]]).astype(np.int64) if len(lengths) else np.zeros(0, np.int64)
    l_pac = int(np.sum(lengths))
    annos = annos or [""] * len(names)
    per_seq = [sum(1 for (o, l, _) in holes if offsets[i] <= o < offsets[i] + lengths[i]) for i in range(len(names))]
    with open(prefix + ".ann", "w") as f:
        f.write("%d %d %u\n" % (l_pac, len(names), seed))
        for i, nm in enumerate(names):
            f.write("0 %s%s\n" % (nm, (" " + annos[i]) if annos[i] else ""))
            f.write("%d %d %d\n" % (int(offsets[i]), int(lengths[i]), per_seq[i]))
    with open(prefix + ".amb", "w") as f:
        f.write("%d %d %u\n" % (l_pac, len(names), len(holes)))
        for o, l, c in holes:
            f.write("%d %d %s\n" % (o, l, c))


class FMIndexDataDevice:
    """io::FMIndexDataHost::load + io::FMIndexDataDevice (nvbio/io/fmindex/fmindex.h:200-362): reads
    <prefix>.bwt/.sa (FORWARD) and .rbwt/.rsa (REVERSE), builds the interleaved bwt|occ records on the device
    and exposes `index()` / `rindex()` as FMIndexDevice objects."""

    def __init__(self, prefix, flags=FORWARD | REVERSE | SA, device="cuda", hbm_rich=True):
        """hbm_rich (the default on this hardware): index() / rindex() come back in the form the device's free memory allows --
        line-native records, 12-mer table, a denser suffix array (FMIndexDevice.hbm_default; NVBIO_HIP_INDEX=lean turns it off).
        lean_index() / lean_rindex() are the arrays exactly as loaded."""
        self.flags = flags
        self._fwd = self._load(prefix + ".bwt", prefix + ".sa", flags, device) if flags & FORWARD else None
        self._rev = self._load(prefix + ".rbwt", prefix + ".rsa", flags, device) if flags & REVERSE else None
        one = self._fwd or self._rev
        self.seq_length = one.length if one else 0
        self._lean = (self._fwd, self._rev)
        self.description = {"line_native": False, "ktab_k": 0, "sa_int": SA_INT, "policy": "lean"}
        if hbm_rich:
            if self._fwd is not None:
                self._fwd, self.description = self._fwd.hbm_default()
            if self._rev is not None:
                self._rev, _ = self._rev.hbm_default()

    def lean_index(self):
        return self._lean[0]

    def lean_rindex(self):
        return self._lean[1]

    @staticmethod
    def _load(bwt_path, sa_path, flags, device):
        primary, _, n, words = read_bwt(bwt_path)
        d_words = torch.from_numpy(words.view(np.int32)).to(device)
        bwt_occ, L2 = build_bwt_occ(n, d_words)
        ssa = None
        if (flags & SA) and os.path.exists(sa_path):
            try:
                ssa = torch.from_numpy(read_sa(sa_path, n, primary).view(np.int32)).to(device)
            except FileMismatch:
                ssa = None                             # "just skip the ssa file" (fmindex_impl.cu:253-256)
        return FMIndexDevice(n, primary, L2, bwt_occ, ssa, SA_INT)

    def index(self):
        return self._fwd

    def rindex(self):
        return self._rev

    def genome_length(self):
        return self.seq_length


def save_fmindex(prefix, host_index, reverse=False):
    """Writes a host index (an object with length, primary, bwt (symbols), ssa, sa_int) the way nvBWT does."""
    from .strings import pack_symbols
    n = host_index.length
    words = pack_symbols(torch.from_numpy(np.ascontiguousarray(host_index.bwt, dtype=np.uint8)), 2, True, pad_words=0).numpy().view(np.uint32)
    cum = np.cumsum(np.bincount(np.asarray(host_index.bwt, dtype=np.uint8), minlength=4)[:4]).astype(np.uint32)
    write_bwt(prefix + (".rbwt" if reverse else ".bwt"), host_index.primary, cum, words, n)
    write_sa(prefix + (".rsa" if reverse else ".sa"), host_index.primary, cum, host_index.ssa, n, host_index.sa_int)


# ----------------------------------------------------------------------------------------------------------------------
# reads: FASTQ -> the packed SequenceData layout (nvbio/io/sequence/sequence_fastq.cpp, sequence_encoder.cpp:118-230)
# ----------------------------------------------------------------------------------------------------------------------
SEQ_FORWARD, SEQ_REVERSE, SEQ_FORWARD_COMPLEMENT, SEQ_REVERSE_COMPLEMENT = 0x1, 0x2, 0x4, 0x8     # SequenceEncoding (sequence.h:172-178)
PHRED, PHRED33, PHRED64, SOLEXA = 0, 1, 2, 3                                                       # QualityEncoding (sequence.h:163-169)

_NT4 = np.full(256, 4, dtype=np.uint8)                     # nst_nt4_encode: ACGT (either case) -> 0..3, anything else -> 4
for _c, _v in zip("ACGTacgt", (0, 1, 2, 3, 0, 1, 2, 3)):
    _NT4[ord(_c)] = _v
_SOLEXA_TO_PHRED = np.array([0, 1, 1, 1, 1, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 7, 8, 9, 10] + list(range(10, 246)), dtype=np.uint8)


class SequenceDataHost:
    """io::SequenceDataHost for DNA_N reads: symbols 4-bit big-endian packed (sequence_traits.h), `sequence_index`
    (n+1 symbol offsets), one phred byte per symbol, names."""

    def __init__(self, symbols, index, quals, names):
        self.symbols, self.sequence_index, self.quals, self.names = symbols, index, quals, names

    def size(self):
        return self.sequence_index.size - 1

    def bps(self):
        return int(self.sequence_index[-1])

    def max_sequence_len(self):
        return int(np.diff(self.sequence_index).max()) if self.size() else 0

    def to_device(self, device="cuda"):
        """-> (PackedStringSet over the 4-bit big-endian stream, quals uint8 tensor)"""
        from .strings import PackedStringSet, pack_symbols
        words = pack_symbols(torch.from_numpy(self.symbols), 4, True).to(device)
        begin = torch.from_numpy(self.sequence_index[:-1].astype(np.int64)).to(device)
        length = torch.from_numpy(np.diff(self.sequence_index).astype(np.int32)).to(device)
        return PackedStringSet(words, 4, True, begin, length, 0), torch.from_numpy(self.quals).to(device)


def _convert_quality(q, encoding):
    if encoding == PHRED33:
        return (q - 33).astype(np.uint8)
    if encoding == PHRED64:
        return (q - 64).astype(np.uint8)
    if encoding == SOLEXA:
        return _SOLEXA_TO_PHRED[q]
    return q.astype(np.uint8)


def read_fastq(path, max_reads=None, flags=SEQ_FORWARD, quality_encoding=PHRED33):
    """Loads a (plain-text, 4-line-record) FASTQ file the way the reference's loader encodes it: each strand
    requested in `flags` is appended per read in the order FORWARD, REVERSE, FORWARD_COMPLEMENT, REVERSE_COMPLEMENT
    (sequence_encoder.cpp:265-300); bases through nst_nt4_encode; qualities converted to phred and reversed with the
    read.  nvBowtie loads its reads with io::REVERSE (nvBowtie.cpp:579)."""
    syms, quals, names, lens = [], [], [], []
    with open(path, "rb") as f:
        while max_reads is None or len(names) < max_reads:
            h = f.readline()
            if not h:
                break
            if not h.startswith(b"@"):
                raise IOError("FASTQ: record does not start with '@' at read %d" % len(names))
            seq = f.readline().rstrip(b"\r\n")
            plus = f.readline()
            ql = f.readline().rstrip(b"\r\n")
            if not plus.startswith(b"+") or len(ql) != len(seq):
                raise IOError("FASTQ: malformed record %d" % len(names))
            bp = _NT4[np.frombuffer(seq, dtype=np.uint8)]
            q = _convert_quality(np.frombuffer(ql, dtype=np.uint8), quality_encoding)
            comp = np.where(bp < 4, 3 - bp, 4).astype(np.uint8)
            for flag, s_, q_ in ((SEQ_FORWARD, bp, q), (SEQ_REVERSE, bp[::-1], q[::-1]),
                                 (SEQ_FORWARD_COMPLEMENT, comp, q), (SEQ_REVERSE_COMPLEMENT, comp[::-1], q[::-1])):
                if flags & flag:
                    syms.append(s_); quals.append(q_); lens.append(len(seq))
            names.append(h[1:].split()[0].decode() if len(h) > 1 else "")
    index = np.zeros(len(lens) + 1, dtype=np.uint32)
    index[1:] = np.cumsum(lens)
    cat = lambda xs: np.concatenate(xs).astype(np.uint8) if xs else np.zeros(0, np.uint8)
    return SequenceDataHost(cat(syms), index, cat(quals), names)


def sam_md_string(mds):
    """The MD:Z value and the XM / XO / XG counters SamOutput::generate_md_string derives from an alignment's byte-coded MDS
    (nvbio/io/output/output_sam.cpp:233-314): consecutive MATCH tokens are summed (as a byte, like the reference's uint8 counter), a
    MISMATCH prints the symbol the MDS holds (finish_alignment stores the READ symbol there, traceback_inl.h:646), a DELETION prints
    '^' + the reference symbols + '0', an INSERTION prints nothing.  -> (md, mm, gapo, gape)."""
    n = int(mds[0]) | (int(mds[1]) << 8)
    out, mm, gapo, gape, i = [], 0, 0, 0, 2
    while i < n:
        op = int(mds[i]); i += 1
        if op == 0:
            run = int(mds[i]); i += 1
            while i < n and int(mds[i]) == 0:                      # (the reference adds the next token's OP byte, 0, and then reads that token's
                run = (run + int(mds[i])) & 0xFF; i += 1           #  count as an op code, which matches no case: runs past 255 are garbled -- kept)
            out.append(str(run))
        elif op == 1:
            out.append("ACGTN"[min(int(mds[i]), 4)]); i += 1; mm += 1
        elif op == 2:
            l = int(mds[i]); i += 1 + l; gapo += 1; gape += l - 1
        elif op == 3:
            l = int(mds[i]); i += 1
            out.append("^" + "".join("ACGTN"[min(int(c), 4)] for c in mds[i:i + l]) + "0"); i += l; gapo += 1; gape += l - 1
    return "".join(out), mm, gapo, gape


# ---------------------------------------------------------------------------------------------------------------------
# BAM output (nvbio/io/output/output_bam.cpp): the records the SAM writer prints, in BAM's binary layout inside BGZF blocks.
# Tag types follow BamOutput::output_alignment (:470-510): NM / XM / XO / XG as 'C' (uint8), AS as 'I' (uint32 of the score's bits),
# MD as 'Z'; bin is always 0 (:374).
# ---------------------------------------------------------------------------------------------------------------------
_BAM_SEQ = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}
_BAM_CIGAR = {c: i for i, c in enumerate("MIDNSHP=X")}
_BGZF_EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def _bgzf_block(payload):
    import struct
    import zlib
    comp = zlib.compressobj(6, zlib.DEFLATED, -15)
    data = comp.compress(payload) + comp.flush()
    bsize = len(data) + 25                      # total block size - 1
    return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", bsize) + data +
            struct.pack("<II", zlib.crc32(payload) & 0xFFFFFFFF, len(payload) & 0xFFFFFFFF))


def _bam_bin(beg, end):
    """The UCSC binning index of the 0-based half-open interval [beg, end) (SAM spec 5.3, reg2bin)."""
    end -= 1
    for shift, offset in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        if beg >> shift == end >> shift:
            return offset + (beg >> shift)
    return 0


def sam_to_bam(sam_text, path, spec_bins=False):
    """Write the SAM text produced by tools/align_fastq.py (header + records over one reference) as a BAM file."""
    import re
    import struct
    header, records = [], []
    for ln in sam_text.splitlines():
        (header if ln.startswith("@") else records).append(ln)
    refs = [(m.group(1), int(m.group(2))) for m in (re.match(r"@SQ\tSN:(\S+)\tLN:(\d+)", h) for h in header) if m]
    ref_id = {name: i for i, (name, _) in enumerate(refs)}
    text = ("\n".join(header) + "\n").encode()
    out = bytearray(b"BAM\x01" + struct.pack("<i", len(text)) + text + struct.pack("<i", len(refs)))
    for name, ln_ in refs:
        out += struct.pack("<i", len(name) + 1) + name.encode() + b"\0" + struct.pack("<i", ln_)
    for ln in records:
        f = ln.split("\t")
        qname, flag, rname, pos, mapq, cigar, rnext, pnext, tlen, seq, qual = f[:11]
        ops = [(int(l), _BAM_CIGAR[o]) for l, o in re.findall(r"(\d+)([MIDNSHP=X])", cigar)] if cigar != "*" else []
        rid = ref_id.get(rname, -1)
        nid = rid if rnext == "=" else ref_id.get(rnext, -1)
        # SEQ '*' = no sequence stored (l_seq 0, no SEQ / QUAL bytes); QUAL '*' = l_seq bytes of 0xFF (SAM spec 4.2.3)
        l_seq = 0 if seq == "*" else len(seq)
        ref_len = sum(l for l, o in ops if o in (0, 2, 3, 7, 8))                  # M D N = X consume the reference
        beg = int(pos) - 1
        # bin: BamOutput leaves it 0 ("BAM alignment bin is always 0", output_bam.cpp:374); bam_bin() below gives the
        # spec's value for callers that want an indexable file (sam_to_bam(..., spec_bins=True))
        bin_ = _bam_bin(max(beg, 0), max(beg, 0) + max(ref_len, 1)) if (spec_bins and beg >= 0) else (4680 if spec_bins else 0)
        body = struct.pack("<iiBBHHHIiii", rid, beg, len(qname) + 1, int(mapq), bin_, len(ops), int(flag),
                           l_seq, nid, int(pnext) - 1, int(tlen))
        body += qname.encode() + b"\0" + b"".join(struct.pack("<I", (l << 4) | o) for l, o in ops)
        if l_seq:
            nib = [_BAM_SEQ[c] for c in seq] + [0]
            body += bytes((nib[2 * k] << 4) | nib[2 * k + 1] for k in range((l_seq + 1) // 2))
            body += b"\xff" * l_seq if qual == "*" else bytes(ord(c) - 33 for c in qual)
        for tag in f[11:]:
            key, ty, val = tag.split(":", 2)
            if ty == "Z":
                body += key.encode() + b"Z" + val.encode() + b"\0"
            elif key == "AS":
                body += key.encode() + b"I" + struct.pack("<I", int(val) & 0xFFFFFFFF)
            else:
                body += key.encode() + b"C" + struct.pack("<B", int(val) & 0xFF)
        out += struct.pack("<i", len(body)) + body
    with open(path, "wb") as fh:
        for s in range(0, len(out), 0xFF00):
            fh.write(_bgzf_block(bytes(out[s:s + 0xFF00])))
        fh.write(_BGZF_EOF)


def read_bam(path):
    """Parse a BAM file back into (header text, [(name, length)], list of record dicts) -- for the tests."""
    import gzip
    import struct
    raw = gzip.open(path, "rb").read()              # BGZF = concatenated gzip members
    assert raw[:4] == b"BAM\x01"
    l_text = struct.unpack_from("<i", raw, 4)[0]
    text = raw[8:8 + l_text].decode()
    o = 8 + l_text
    n_ref = struct.unpack_from("<i", raw, o)[0]; o += 4
    refs = []
    for _ in range(n_ref):
        l = struct.unpack_from("<i", raw, o)[0]; o += 4
        name = raw[o:o + l - 1].decode(); o += l
        refs.append((name, struct.unpack_from("<i", raw, o)[0])); o += 4
    recs = []
    while o < len(raw):
        bs = struct.unpack_from("<i", raw, o)[0]; o += 4
        rid, pos, l_name, mapq, _bin, n_cig, flag, l_seq, nid, npos, tlen = struct.unpack_from("<iiBBHHHIiii", raw, o)
        p = o + 32
        name = raw[p:p + l_name - 1].decode(); p += l_name
        cig = "".join("%d%s" % (w >> 4, "MIDNSHP=X"[w & 15]) for w in struct.unpack_from("<%dI" % n_cig, raw, p)); p += 4 * n_cig
        sq = raw[p:p + (l_seq + 1) // 2]; p += (l_seq + 1) // 2
        seq = "".join("=ACMGRSVTWYHKDBN"[(sq[k // 2] >> (4 if k % 2 == 0 else 0)) & 15] for k in range(l_seq))
        qual = "".join(chr(q + 33) for q in raw[p:p + l_seq]); p += l_seq
        tags = {}
        while p < o + bs:
            key, ty = raw[p:p + 2].decode(), chr(raw[p + 2]); p += 3
            if ty == "Z":
                e = raw.index(b"\0", p); tags[key] = raw[p:e].decode(); p = e + 1
            elif ty in "Ii":
                tags[key] = struct.unpack_from("<I" if ty == "I" else "<i", raw, p)[0]; p += 4
            elif ty in "Ss":
                tags[key] = struct.unpack_from("<H" if ty == "S" else "<h", raw, p)[0]; p += 2
            else:                                           # C / c
                tags[key] = raw[p] if ty == "C" else struct.unpack_from("<b", raw, p)[0]; p += 1
        recs.append(dict(name=name, flag=flag, ref=rid, pos=pos + 1, mapq=mapq, cigar=cig or "*", next_ref=nid, pnext=npos + 1, tlen=tlen, seq=seq, qual=qual, tags=tags))
        o += bs
    return text, refs, recs
