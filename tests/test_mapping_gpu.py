"""nvBowtie exact seed mapping (seed hit sets) through the C-ABI vs the oracle's restatement of
map_queues_kernel<EXACT_MAPPING>.  A read's hits are compared as the array its hit deque holds (the
reference's own checksum is order independent, nvBowtie/bowtie2/cuda/checksums.h; the selection stage is not)."""
import numpy as np
import pytest
import torch

import nvbio_amd as nvb
from nvbio_amd import mapping as M
from oracle import pyoracle as O

pytestmark = pytest.mark.gpu


def make_reads(rng, text, n, ragged):
    reads = []
    for i in range(n):
        L = int(rng.integers(8, 160)) if ragged else 100
        p = int(rng.integers(0, text.size - L))
        r = text[p:p + L].copy()
        if i % 2 == 0:                                   # reverse-complement strand reads
            r = (3 - r)[::-1].copy()
        mut = rng.random(L) < 0.03
        r[mut] = rng.integers(0, 4, int(mut.sum()), dtype=np.uint8)
        if i % 17 == 0:
            r[int(rng.integers(0, L))] = 4               # an N
        reads.append(r[::-1].copy())                     # nvBowtie stores reads reversed (io::REVERSE)
    return reads


@pytest.mark.parametrize("ragged", [False, True])
def test_seed_hit_sets(cuda, ragged):
    rng = np.random.default_rng(11 + ragged)
    text = rng.integers(0, 4, 1 << 18, dtype=np.uint8)
    text[5000:5600] = np.tile(np.array([0, 1], dtype=np.uint8), 300)      # a repeat: wide SA ranges
    host = O.FMIndex(text)
    fmi = nvb.FMIndexDevice.from_host(host, cuda)
    fdim = fmi.with_dimer()
    ftri = fdim.with_trimer()
    reads = make_reads(rng, text, 4000, ragged)
    hr = O.StringSet.from_lists(reads, 4, True)
    dr = nvb.PackedStringSet.from_host(hr.words, 4, True, hr.begin, hr.length, device=cuda)
    params = nvb.MappingParams()
    max_len = 160
    sf = O.simple_func_table(2, 1.0, 1.15, max_len + 1)
    assert (params.seed_freq_table(max_len, "cpu").numpy().view(np.uint32)[1:] == sf[1:]).all()   # host table == C float arithmetic
    for retry, fw, rc, max_hits, queue in ((0, 1, 1, 100, None), (1, 1, 1, 100, None), (2, 1, 0, 100, None), (0, 0, 1, 100, None),
                                           (0, 1, 1, 3, None), (0, 1, 1, 100, rng.permutation(4000)[:1500].astype(np.uint32))):
        params.max_hits = max_hits
        stride = 40
        pd = dict(seed_len=params.seed_len, min_read_len=params.min_read_len, max_hits=max_hits, max_reseed=params.max_reseed,
                  retry=retry, rep_seeds=params.rep_seeds, fw=fw, rc=rc)
        eh, ec, er = O.map_exact(host, hr, pd, sf, stride, in_queue=queue)
        dq = torch.from_numpy(queue.view(np.int32)).to(cuda) if queue is not None else None
        gh, gc, gr = nvb.map_exact(fmi, dr, params, max_len, retry=retry, fw=bool(fw), rc=bool(rc), in_queue=dq, hits_stride=stride)
        # the k-mer table accelerator must not change anything
        for kk in (9, 12):
            kh, kc, kr = nvb.map_exact(fmi.with_ktab(kk), dr, params, max_len, retry=retry, fw=bool(fw), rc=bool(rc), in_queue=dq, hits_stride=stride)
            assert torch.equal(kh, gh) and torch.equal(kc, gc) and torch.equal(kr, gr), kk
        # nor may the line-native two-symbol index (alone, and under the k-mer table)
        for fv in (fdim, fdim.with_ktab(9), ftri, ftri.with_ktab(9)):
            kh, kc, kr = nvb.map_exact(fv, dr, params, max_len, retry=retry, fw=bool(fw), rc=bool(rc), in_queue=dq, hits_stride=stride)
            assert torch.equal(kc, gc) and torch.equal(kr, gr), fv.ktab_k
            assert torch.equal(kh, gh), fv.ktab_k
        torch.cuda.synchronize()
        gh, gc, gr = gh.cpu().numpy().view(np.uint64), gc.cpu().numpy().view(np.uint32), gr.cpu().numpy()
        ids = queue if queue is not None else np.arange(4000)
        assert (gc[ids] == ec[ids]).all()
        assert (gr == er).all()
        assert gc.max() > 4 or max_hits == 3
        for r in ids:
            # the deque's array order (interval heap layout), which of several equal-sized hits is dropped at the cap included
            assert (gh[r, :gc[r]] == eh[r, :ec[r]]).all(), (r, gh[r, :gc[r]], eh[r, :ec[r]])
    # the hits decode to seeds that really occur: fw hit rows locate to text positions holding the seed
    u = nvb.unpack_seed_hits(torch.from_numpy(gh.view(np.int64)))
    assert int(u["index_dir"].max()) == 0


@pytest.mark.parametrize("allow_sub,subseed", [(1, 10), (1, 16), (1, 0)])
def test_one_mismatch_seed_hit_sets(cuda, allow_sub, subseed):
    """map_queues_kernel<APPROX_MAPPING / CASE_PRUNING_MAPPING> (params.allow_sub, params.subseed_len) vs the
    oracle's restatement; per-read deque arrays compared verbatim."""
    rng = np.random.default_rng(300 + subseed)
    text = rng.integers(0, 4, 1 << 17, dtype=np.uint8)
    text[5000:5600] = np.tile(np.array([0, 1], dtype=np.uint8), 300)
    host, rhost = O.FMIndex(text), O.FMIndex(text[::-1].copy())
    fmi, rfmi = nvb.FMIndexDevice.from_host(host, cuda), nvb.FMIndexDevice.from_host(rhost, cuda)
    n = 3000
    reads = make_reads(rng, text, n, ragged=True)
    for i in range(0, n, 29):                            # a second N in some reads
        reads[i][int(rng.integers(0, reads[i].size))] = 4
    hr = O.StringSet.from_lists(reads, 4, True)
    dr = nvb.PackedStringSet.from_host(hr.words, 4, True, hr.begin, hr.length, device=cuda)
    max_len, stride, most = 160, 200, 0
    for seed_len, retry, fw, rc, max_hits in ((22, 0, 1, 1, 200), (20, 1, 1, 1, 200), (32, 0, 1, 0, 200), (13, 2, 0, 1, 200), (22, 0, 1, 1, 5)):
        if subseed > seed_len:
            continue
        params = nvb.MappingParams(seed_len=seed_len, max_hits=max_hits)
        sf = O.simple_func_table(2, 1.0, 1.15, max_len + 1)
        pd = dict(seed_len=seed_len, min_read_len=params.min_read_len, max_hits=max_hits, max_reseed=params.max_reseed,
                  retry=retry, rep_seeds=params.rep_seeds, fw=fw, rc=rc)
        algo = 2 if subseed == 0 else 1
        eh, ec, er = O.map_seeds(algo, subseed, host, rhost, hr, pd, sf, stride)
        # neither the k-mer table nor the line-native two-symbol index may change anything
        for f_dev, rf_dev in ((fmi, rfmi), (fmi.with_ktab(8), rfmi.with_ktab(8)), (fmi.with_dimer(), rfmi.with_dimer()),
                              (fmi.with_dimer().with_ktab(8), rfmi.with_dimer()), (fmi.with_trimer(), rfmi.with_trimer())):
            gh, gc, gr = nvb.map_seeds(f_dev, rf_dev, dr, params, max_len, allow_sub=allow_sub, subseed_len=subseed, retry=retry,
                                       fw=bool(fw), rc=bool(rc), hits_stride=stride)
            torch.cuda.synchronize()
            gh, gc, gr = gh.cpu().numpy().view(np.uint64), gc.cpu().numpy().view(np.uint32), gr.cpu().numpy()
            assert (gc == ec).all(), (seed_len, np.nonzero(gc != ec)[0][:5])
            assert (gr == er).all()
            for r in range(n):
                assert (gh[r, :gc[r]] == eh[r, :ec[r]]).all(), (seed_len, r, gh[r, :gc[r]], eh[r, :ec[r]])       # deque array order
        most = max(most, int(ec.max()))
    assert most >= 6
    if subseed == 0:
        assert int(nvb.unpack_seed_hits(torch.from_numpy(gh.view(np.int64)))["index_dir"].max()) == 1
