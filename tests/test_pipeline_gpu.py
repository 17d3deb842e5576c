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
    assert gj == ej and gj > n_reads
    assert torch.equal(gs.cpu(), es) and torch.equal(gp.cpu(), ep)
    # and the driver finds the reads: the best window starts band/2 before the true position
    found = (gp.cpu() >= 0)
    assert found.float().mean() > 0.85          # reads without an error-free 22-mer seed window are not found
    exp = torch.clamp(pos - 7, min=0)
    assert ((gp.cpu() == exp) | ~found).float().mean() > 0.98
