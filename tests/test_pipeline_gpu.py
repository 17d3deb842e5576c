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
