"""A numpy / pure-Python model of the line-native two-symbol FM-index (nvbio_amd/csrc/fmindex_dimer.h):
the record layout built from a suffix array, and the queries exactly as the HIP code states them.
Test tooling: it pins the layout and the formulas on the CPU suite (model vs the oracle's reference-layout
match / locate), and on the GPU the device-built buffer must equal the model's word for word."""
import numpy as np

MAGIC = 0x44694D32
M32 = 0xFFFFFFFF


def n_records(n):
    return ((n + 1) >> 7) + 1


def pd_records(n):
    return (n + 1) // 96 + 1


def build(text, sa, L2):
    """-> uint32 array: 32-dword header, 32 dwords per plane record, then 16 per-dimer arrays of 4-dword records."""
    t = np.asarray(text, dtype=np.int64)
    sa = np.asarray(sa, dtype=np.int64)
    n = t.size
    primary = int(np.nonzero(sa == 0)[0][0])
    p1 = int(np.nonzero(sa == 1)[0][0]) if n >= 1 else M32
    b = np.where(sa >= 1, t[np.maximum(sa - 1, 0)], 0)
    a = np.where(sa >= 2, t[np.maximum(sa - 2, 0)], 0)
    nib = (a * 4 + b).astype(np.int64)                  # fillers: primary -> 0, p1 -> (0, T[0])
    nr = n_records(n)
    rows = np.zeros(nr * 128, dtype=np.int64)
    rows[:n + 1] = nib
    valid = np.zeros(nr * 128, dtype=bool)
    valid[:n + 1] = True
    npd = pd_records(n)
    out = np.zeros(32 + 32 * nr + 16 * 4 * npd, dtype=np.uint32)
    # C2[slot b*4+a] = (first row whose suffix starts with "ab") - 1 = L2[a] + #{a in BWT rows <= L2[b]}
    # (rows <= k, the '$' row excluded)
    bw = np.where(sa >= 1, t[np.maximum(sa - 1, 0)], -1)
    C2 = np.zeros(16, dtype=np.int64)
    for bb in range(4):
        for aa in range(4):
            k = int(L2[bb])
            C2[bb * 4 + aa] = int(L2[aa]) + int(np.count_nonzero(bw[:k + 1] == aa))
    # per-block counts of stored nibbles (fillers included), exclusive prefix
    blk_rows = rows.reshape(nr, 128)
    blk_valid = valid.reshape(nr, 128)
    cnt = np.zeros((nr, 16), dtype=np.int64)
    for v in range(16):
        cnt[:, (v & 3) * 4 + (v >> 2)] = np.count_nonzero((blk_rows == v) & blk_valid, axis=1)
    excl = np.cumsum(cnt, axis=0) - cnt
    rec = out[32:32 + 32 * nr].reshape(nr, 32)
    rec[:, :16] = ((excl + C2[None, :]) & M32).astype(np.uint32)
    for p in range(4):
        bits = ((blk_rows >> p) & 1).astype(np.uint64)          # (nr, 128)
        for w in range(4):
            word = np.zeros(nr, dtype=np.uint64)
            for r in range(32):
                word |= bits[:, 32 * w + r] << np.uint64(r)
            rec[:, 16 + 4 * p + w] = word.astype(np.uint32)
    hdr = out[:32]
    hdr[0], hdr[1], hdr[2], hdr[3] = MAGIC, n, primary, p1
    hdr[4] = int(t[0]) if n >= 1 else 0
    l2 = int(L2[2])
    hdr[5], hdr[6], hdr[7] = nr, npd, int(L2[1]) ^ (((l2 << 11) | (l2 >> 21)) & M32)
    # per-dimer arrays: pd[v][r] = {C2 + #{true dimers v in rows < 96r}, 96-bit mask of rows 96r..96r+95 holding v}
    true = np.zeros(npd * 96, dtype=np.int64) - 1
    true[:n + 1] = nib
    true[primary] = -1
    if n >= 1:
        true[p1] = -1
    tb = true.reshape(npd, 96)
    pd = out[32 + 32 * nr:].reshape(16, npd, 4)
    for v in range(16):
        m = (tb == v)
        c = np.count_nonzero(m, axis=1)
        pd[v, :, 0] = ((np.cumsum(c) - c + C2[(v & 3) * 4 + (v >> 2)]) & M32).astype(np.uint32)
        for w in range(3):
            word = np.zeros(npd, dtype=np.uint64)
            for r in range(32):
                word |= m[:, 32 * w + r].astype(np.uint64) << np.uint64(r)
            pd[v, :, 1 + w] = word.astype(np.uint32)
    for c in range(4):
        k = int(C2[c * 4:c * 4 + 4].sum())
        hdr[8 + c] = (int(L2[c]) - k) & M32
        hdr[12 + c] = (-k) & M32
    hdr[16:32] = (C2 & M32).astype(np.uint32)
    return out


TRIMER_MAGIC = 0x54724D33


def build_trimer(text, sa, L2):
    """-> uint32 array: 128-dword header (C3 at dword 64), then pk[code][record] of 4 dwords (fmindex_trimer.hip)."""
    t = np.asarray(text, dtype=np.int64)
    sa = np.asarray(sa, dtype=np.int64)
    n = t.size
    R = pd_records(n)
    code = np.where(sa >= 3, t[np.maximum(sa - 3, 0)] * 16 + t[np.maximum(sa - 2, 0)] * 4 + t[np.maximum(sa - 1, 0)], -1)
    rows = np.zeros(R * 96, dtype=np.int64) - 1
    rows[:n + 1] = code
    tb = rows.reshape(R, 96)
    out = np.zeros(128 + 64 * 4 * R, dtype=np.uint32)
    out[0], out[1], out[2], out[3] = TRIMER_MAGIC, n, int(np.nonzero(sa == 0)[0][0]), R
    # C3[abc] = (first row whose suffix starts with "abc") - 1: rows are sorted by suffix, so count the suffixes below "abc"
    # (shorter suffixes that are proper prefixes of "abc" sort before it)
    sfx = [tuple(t[p:p + 3]) for p in range(n)] + [()]
    pk = out[128:].reshape(64, R, 4)
    for c in range(64):
        key = (c >> 4, (c >> 2) & 3, c & 3)
        out[64 + c] = (sum(1 for s_ in sfx if s_ < key) - 1) & M32
        m = (tb == c)
        cnt = np.count_nonzero(m, axis=1)
        pk[c, :, 0] = ((np.cumsum(cnt) - cnt + int(out[64 + c])) & M32).astype(np.uint32)
        for w in range(3):
            word = np.zeros(R, dtype=np.uint64)
            for r in range(32):
                word |= m[:, 32 * w + r].astype(np.uint64) << np.uint64(r)
            pk[c, :, 1 + w] = word.astype(np.uint32)
    return out


class Model:
    def __init__(self, buf, trimer=None):
        self.buf = buf
        self.tri = trimer
        h = buf[:32]
        assert int(h[0]) == MAGIC
        self.n, self.primary, self.p1, self.fill1 = int(h[1]), int(h[2]), int(h[3]), int(h[4])
        self.nr, self.npd = int(h[5]), int(h[6])
        self.S = [int(x) for x in h[8:12]]
        self.T = [int(x) for x in h[12:16]]
        self.lines = 0          # records touched (distinct per step), for the traffic model

    def rec(self, k):
        return self.buf[32 + 32 * k: 64 + 32 * k]

    @staticmethod
    def _plane(r, p):
        return int(r[16 + 4 * p]) | int(r[17 + 4 * p]) << 32 | int(r[18 + 4 * p]) << 64 | int(r[19 + 4 * p]) << 96

    def _match_bits(self, r, v, nplanes):
        m = (1 << 128) - 1
        for p in range(nplanes):
            pl = self._plane(r, p)
            m &= pl if (v >> p) & 1 else ~pl
        return m & ((1 << 128) - 1)

    @staticmethod
    def _prefix(m, w):
        return bin(m & ((1 << w) - 1)).count("1")

    def filler(self, e, v):
        return int(v == 0 and e > self.primary) + int(v == self.fill1 and e > self.p1)

    def D(self, e, a, b):
        q = e // 96
        o = 32 + 32 * self.nr + 4 * ((a * 4 + b) * self.npd + q)
        r = self.buf[o:o + 4]
        m = int(r[1]) | int(r[2]) << 32 | int(r[3]) << 64
        return (int(r[0]) + self._prefix(m, e - 96 * q)) & M32

    def R(self, e, c):
        r = self.rec(e >> 7)
        k = int(r[c * 4]) + int(r[c * 4 + 1]) + int(r[c * 4 + 2]) + int(r[c * 4 + 3])
        return (self.S[c] + k + self._prefix(self._match_bits(r, c, 2), e & 127) - int(c == 0 and e > self.primary)) & M32

    def step2(self, x, y, a, b):
        self.lines += 1 if x // 768 == (y + 1) // 768 else 2
        return (self.D(x, a, b) + 1) & M32, self.D((y + 1) & M32, a, b)

    def D3(self, e, code):
        q = e // 96
        R = int(self.tri[3])
        o = 128 + 4 * (code * R + q)
        r = self.tri[o:o + 4]
        m = int(r[1]) | int(r[2]) << 32 | int(r[3]) << 64
        return (int(r[0]) + self._prefix(m, e - 96 * q)) & M32

    def step3(self, x, y, a, b, c):
        self.lines += 1 if x // 768 == (y + 1) // 768 else 2
        code = a * 16 + b * 4 + c
        return (self.D3(x, code) + 1) & M32, self.D3((y + 1) & M32, code)

    def step1(self, x, y, c):
        self.lines += 1 if (x >> 7) == ((y + 1) >> 7) else 2
        return (self.R(x, c) + 1) & M32, self.R((y + 1) & M32, c)

    def match(self, seed):
        """fm_match_from: seed symbols consumed from the last one, in 16-symbol groups counted from the seed start."""
        x, y = 0, self.n
        i = len(seed) - 1
        pairs = True
        triples = self.tri is not None
        while i >= 0 and x <= y:
            g0 = i & ~15
            while i >= g0 and x <= y:
                c = int(seed[i])
                if c > 3:
                    return 1, 0
                if pairs and i > g0:
                    b = int(seed[i - 1])
                    if b <= 3:
                        if triples and i - 2 >= g0:
                            a = int(seed[i - 2])
                            if a <= 3:
                                nx, ny = self.step3(x, y, a, b, c)
                                if nx <= ny:
                                    x, y, i = nx, ny, i - 3
                                    continue
                                triples = False
                        nx, ny = self.step2(x, y, b, c)
                        if nx <= ny:
                            x, y, i = nx, ny, i - 2
                            continue
                        pairs = triples = False
                x, y = self.step1(x, y, c)
                i -= 1
        return x, y

    def locate_it(self, j, sa_int):
        mask = sa_int - 1
        t = 0
        while j & mask:
            if j == self.primary:
                j, t = 0, t + 1
                break
            self.lines += 1
            r = self.rec(j >> 7)
            rr = j & 127
            nib = sum(((self._plane(r, p) >> rr) & 1) << p for p in range(4))
            b, a = nib & 3, nib >> 2
            w = rr + 1
            kb = int(r[b * 4]) + int(r[b * 4 + 1]) + int(r[b * 4 + 2]) + int(r[b * 4 + 3])
            j1 = (self.S[b] + kb + self._prefix(self._match_bits(r, b, 2), w) - int(b == 0 and j > self.primary)) & M32
            if (j1 & mask) == 0:
                j, t = j1, t + 1
                break
            if j1 == self.primary:
                j, t = 0, t + 2
                break
            j = (int(r[b * 4 + a]) + self._prefix(self._match_bits(r, nib, 4), w) - self.filler(j + 1, nib)) & M32
            t += 2
        return j, t
