"""The composed seed-and-extend driver (nvbio_amd.pipeline) on the HIP kernels vs the identical glue
over the CPU oracle: best score and position per read must agree exactly."""
import numpy as np
import pytest
import torch

import nvbio_amd as nvb
from nvbio_amd import pipeline as P, workloads as W
from oracle import pyoracle as O

pytestmark = pytest.mark.gpu


class OracleBackend:
    """Same three calls, computed by the oracle on the host (tests only)."""

    def __init__(self, host_fmi, map_params, max_read_len):
        self.host, self.mp, self.max_read_len = host_fmi, map_params, max_read_len

    def map_exact(self, reads_rev, hits_stride):
        hr = O.StringSet.from_device(reads_rev)
        sf = self.mp.seed_freq_table(self.max_read_len, "cpu").numpy().view(np.uint32)
        pd = dict(seed_len=self.mp.seed_len, min_read_len=self.mp.min_read_len, max_hits=self.mp.max_hits,
                  max_reseed=self.mp.max_reseed, retry=0, rep_seeds=self.mp.rep_seeds, fw=1, rc=1)
        h, c, _ = O.map_exact(self.host, hr, pd, sf, hits_stride)
        # generation order is the same in both implementations, but sort anyway: the glue is order-free
        return torch.from_numpy(np.sort(np.where(np.arange(hits_stride)[None, :] < c[:, None], h, np.uint64(2**64 - 1)), axis=1).view(np.int64)), \
            torch.from_numpy(c.view(np.int32))

    def locate(self, rows):
        return torch.from_numpy(self.host.locate(rows.numpy().view(np.uint32)).view(np.int32))

    def score(self, band, aligner, patterns, texts):
        sc = aligner.scheme
        s, k = O.batch_banded_gotoh_score(band, aligner.type, (sc.m_match, sc.m_mismatch, sc.m_gap_open, sc.m_gap_ext),
                                          O.StringSet.from_device(patterns), O.StringSet.from_device(texts))
        return torch.from_numpy(s), torch.from_numpy(k.view(np.int32))


class OracleBackendFull(OracleBackend):
    """... plus the stages after scoring, computed by the oracle."""

    def init_best(self, n, scheme, read_len):
        return torch.from_numpy(O.init_alignments(np.full(n, read_len, np.uint32), scheme.m_score_min).view(np.int64))

    def reduce(self, best, hit_begin, score, loc, rc, read_len):
        b = best.numpy().view(np.uint64)
        O.score_reduce(b, hit_begin.numpy().view(np.uint64), score.numpy(), loc.numpy().view(np.uint32), rc.numpy(), np.full(b.shape[1], read_len, np.uint32))
        return best

    def mapq(self, best, scheme, read_len, version=2):
        b = best.numpy().view(np.uint64)
        return torch.from_numpy(O.mapq(version, scheme.m_match, scheme.m_score_min, scheme.m_monotone, b, np.full(b.shape[1], read_len, np.uint32)))

    def traceback(self, band, aligner, patterns, texts, cigar_stride):
        sc = aligner.scheme
        r = O.batch_banded_gotoh_traceback(band, aligner.type, (sc.m_match, sc.m_mismatch, sc.m_gap_open, sc.m_gap_ext),
                                           O.StringSet.from_device(patterns), O.StringSet.from_device(texts), cigar_stride)
        return dict(score=torch.from_numpy(r["score"]), sink=torch.from_numpy(r["sink"].view(np.int32)), source=torch.from_numpy(r["source"].view(np.int32)),
                    cigar=torch.from_numpy(r["cigar"].view(np.int16)), cigar_len=torch.from_numpy(r["cigar_len"].view(np.int32)))


class OracleBackendPaired(OracleBackendFull):
    """... plus the paired-end stages."""

    def init_best_mate(self, n, scheme, read_len, mate):
        return torch.from_numpy(O.init_alignments(np.full(n, read_len, np.uint32), scheme.m_score_min, mate).view(np.int64))

    @staticmethod
    def _q(scheme):
        st = scheme.struct()
        return st, np.array([st.mismatch[q] for q in range(256)], dtype=np.int32)

    def score_qual(self, band, aligner, patterns, texts, quals):
        st, lut = self._q(aligner.scheme)
        s, k = O.batch_banded_gotoh_score_qual(band, aligner.type, (st.match, st.pattern_gap_open, st.pattern_gap_ext, st.text_gap_open, st.text_gap_ext, 0), lut,
                                               quals.numpy(), O.StringSet.from_device(patterns), O.StringSet.from_device(texts))
        return torch.from_numpy(s), torch.from_numpy(k.view(np.int32))

    def opposite_windows(self, read_id, rc, loc, score, best, best_o, scheme, anchor, genome_len, read_len, min_frag_len=0, max_frag_len=500, score_limit=0):
        n = best.shape[1]
        r = O.opposite_windows(read_id.numpy(), rc.numpy(), loc.numpy().view(np.uint32), score.numpy(), np.full(n, read_len, np.uint32), np.full(n, read_len, np.uint32),
                               best.numpy().view(np.uint64), best_o.numpy().view(np.uint64), scheme.m_match, scheme.m_score_min, scheme.text_gap_open(), scheme.text_gap_extension(),
                               1, min_frag_len, max_frag_len, True, score_limit, anchor, genome_len)
        return dict(valid=torch.from_numpy(r["valid"]), min_score=torch.from_numpy(r["min_score"]), read_rc=torch.from_numpy(r["read_rc"]),
                    genome_begin=torch.from_numpy(r["genome_begin"].view(np.int32)), genome_end=torch.from_numpy(r["genome_end"].view(np.int32)))

    def full_score_qual(self, aligner, patterns, texts, max_m, max_n, min_score, quals):
        st, lut = self._q(aligner.scheme)
        s, k, ok = O.batch_gotoh_score_qual(0, aligner.type, (st.match, st.pattern_gap_open, st.pattern_gap_ext, st.text_gap_open, st.text_gap_ext), lut, quals.numpy(),
                                            O.StringSet.from_device(patterns), O.StringSet.from_device(texts), min_score=min_score.numpy())
        return torch.from_numpy(s), torch.from_numpy(k.view(np.int32)), torch.from_numpy(ok)

    def reduce_paired(self, best, best_o, hit_begin, loc, sink, score, rc, o_loc, o_sink, o_sink2, o_score, o_score2, anchor, read_len, score_limit=0):
        n = best.shape[1]
        u = lambda t: t.numpy().view(np.uint32)
        O.score_reduce_paired(best.numpy().view(np.uint64), best_o.numpy().view(np.uint64), hit_begin.numpy().view(np.uint64), u(loc), u(sink), score.numpy(), rc.numpy(),
                              u(o_loc), u(o_sink), u(o_sink2), o_score.numpy(), o_score2.numpy(), np.full(n, read_len, np.uint32), anchor, 1, True, score_limit)

    def mapq_paired(self, best, best_o, scheme, read_len):
        n = best.shape[1]
        return torch.from_numpy(O.mapq_paired(2, scheme.m_match, scheme.m_score_min, scheme.m_monotone, best.numpy().view(np.uint64), best_o.numpy().view(np.uint64),
                                              np.full(n, read_len, np.uint32), np.full(n, read_len, np.uint32)))


def test_align_paired_end_matches_oracle(cuda):
    """BASELINE config 5's shape at test size: 2 x 150 bp FR pairs, LOCAL band 31 in nvBowtie's local scheme; every stage's
    output equals the same glue over the oracle, and the pairs come back paired, at their fragment's two ends."""
    g = torch.Generator().manual_seed(7)
    n_genome, n_pairs, L = 1 << 20, 600, 150
    text = torch.randint(0, 4, (n_genome,), dtype=torch.uint8, generator=g)
    host = O.FMIndex(text.numpy())
    sym1, sym2, pos, flen = P.make_read_pairs(text, n_pairs, L, seed=11)
    mp = nvb.MappingParams(seed_len=20)
    gw_host = W._pack_chunked(text, 2, True)
    e = P.align_paired_end(OracleBackendPaired(host, mp, L), sym1, sym2, gw_host, n_genome)
    fmi = nvb.FMIndexDevice.from_host(host, cuda)
    r = P.align_paired_end(P.HipBackend(fmi, None, mp, L), sym1.to(cuda), sym2.to(cuda), gw_host.to(cuda), n_genome)
    torch.cuda.synchronize()
    assert (r["n_jobs"], r["n_opposite"]) == (e["n_jobs"], e["n_opposite"]) and r["n_opposite"] > n_pairs
    for key in ("best", "best_o", "mapq"):
        assert torch.equal(r[key].cpu(), e[key]), key
    b, bo = r["best"].cpu(), r["best_o"].cpu()
    paired = ((b[0] >> 30) & 1) != 0
    assert paired.float().mean() > 0.9
    # the anchor entry holds the anchor mate's start; the opposite entry holds its window begin + the offset of its alignment's end
    a_pos, o_pos, o_end = (b[0] >> 32) & 0xFFFFFFFF, (bo[0] >> 32) & 0xFFFFFFFF, ((bo[0] >> 32) & 0xFFFFFFFF) + ((bo[0] >> 18) & 0x3FF)
    left, right = pos, pos + flen - L
    anchor_ok = ((a_pos - left).abs() <= 3) | ((a_pos - right).abs() <= 3)
    opp_ok = ((o_end - (left + L)).abs() <= 3) | ((o_end - (right + L)).abs() <= 3)
    assert ((anchor_ok & opp_ok) | ~paired).float().mean() > 0.97
    assert int(r["mapq"].cpu()[paired].to(torch.int32).min()) >= 0


def test_align_single_end_matches_oracle(cuda):
    """seed -> locate -> extend -> score_reduce -> MAPQ -> traceback: every stage's output identical to the same
    glue over the oracle; reads come back at their true positions with sensible qualities and CIGARs."""
    g = torch.Generator().manual_seed(6)
    n_genome, n_reads, L = 1 << 20, 2000, 100
    text = torch.randint(0, 4, (n_genome,), dtype=torch.uint8, generator=g)
    text[5000:5200] = text[9000:9200]                       # a repeat: reads from it have a second best alignment
    host = O.FMIndex(text.numpy())
    sym, pos, is_rc = P.make_reads(text, n_reads, L, seed=10)
    sym[:40] = text[5030:5130].unsqueeze(0).expand(40, L)   # reads inside the repeat
    pos[:40] = 5030; is_rc[:40] = False
    mp = nvb.MappingParams()
    gw_host = W._pack_chunked(text, 2, True)
    e = P.align_single_end(OracleBackendFull(host, mp, L), sym, gw_host, n_genome)
    fmi = nvb.FMIndexDevice.from_host(host, cuda)
    r = P.align_single_end(P.HipBackend(fmi, None, mp, L), sym.to(cuda), gw_host.to(cuda), n_genome)
    torch.cuda.synchronize()
    assert r["n_jobs"] == e["n_jobs"]
    for key in ("best", "mapq", "cigar", "cigar_len", "source", "sink", "tb_score"):
        assert torch.equal(r[key].cpu(), e[key]), key
    best = r["best"].cpu()
    aligned = ((best[0] >> 32) & 0xFFFFFFFF) != 0xFFFFFFFF
    assert aligned.float().mean() > 0.85
    # the traceback re-scores to the reduced best score, and its CIGAR consumes the read
    mag = (best[0] >> 1) & 0x1FFFF
    bscore = torch.where((best[0] & 1) != 0, -mag, mag)
    assert torch.equal(r["tb_score"].cpu().to(torch.int64), bscore[aligned])
    cig = r["cigar"].cpu().to(torch.int32) & 0xFFFF
    kk = torch.arange(cig.shape[1])[None, :] < r["cigar_len"].cpu()[:, None]
    consumed = (((cig >> 2) * ((cig & 3) != 2)) * kk).sum(1)
    assert bool((consumed[aligned] == L).all())
    # unique reads: high quality; reads from the repeat: a second best alignment and quality <= 1
    mq = r["mapq"].cpu()
    second = ((best[1] >> 32) & 0xFFFFFFFF) != 0xFFFFFFFF
    assert bool(second[:40].all()) and int(mq[:40].max()) <= 1
    assert float((mq[40:][aligned[40:]] >= 8).float().mean()) > 0.9       # 4 % substitutions at -6: mostly 23..42


def test_seed_and_extend_matches_oracle(cuda):
    g = torch.Generator().manual_seed(5)
    n_genome, n_reads, L = 1 << 20, 3000, 100
    text = torch.randint(0, 4, (n_genome,), dtype=torch.uint8, generator=g)
    host = O.FMIndex(text.numpy())
    sym, pos, is_rc = P.make_reads(text, n_reads, L, seed=9)
    mp = nvb.MappingParams()
    gw_host = W._pack_chunked(text, 2, True)
    # CPU: identical glue over the oracle
    es, ep, ej = P.seed_and_extend(OracleBackend(host, mp, L), sym, gw_host, n_genome)
    # GPU: the product path
    fmi = nvb.FMIndexDevice.from_host(host, cuda)
    be = P.HipBackend(fmi, None, mp, L)
    gs, gp, gj = P.seed_and_extend(be, sym.to(cuda), gw_host.to(cuda), n_genome)
    torch.cuda.synchronize()
    assert gj == ej and gj > 0.8 * n_reads
    assert torch.equal(gs.cpu(), es) and torch.equal(gp.cpu(), ep)
    # and the driver finds the reads: the best window starts band/2 before the true position
    found = (gp.cpu() >= 0)
    assert found.float().mean() > 0.85          # reads without an error-free 22-mer seed window are not found
    exp = torch.clamp(pos - 7, min=0)
    assert ((gp.cpu() == exp) | ~found).float().mean() > 0.98
