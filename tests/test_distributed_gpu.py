"""Two ranks over RCCL on two MI355X of one node: the result gather and the alignment-record gather of the sharded drivers,
with the device-side fit check of the compact formats.  Self-skips on a single-GPU box (the driver's GPU tier has one)."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n, q):
    import torch.distributed as dist
    import nvbio_amd as nvb
    from nvbio_amd import workloads as W
    from nvbio_amd.distributed import RecordGather, ResultGather, alignment_records, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        patterns, texts = W.make_sw_batch(n, seed=5, device=dev)              # the same batch on every rank; each scores its block
        lo, hi = shard_range(n, rank, world)
        sub_p = nvb.PackedStringSet(patterns.words, 4, True, patterns.begin[lo:hi].contiguous(), None, 100)
        sub_t = nvb.PackedStringSet(texts.words, 2, False, texts.begin[lo:hi].contiguous(), None, 150)
        al = nvb.make_gotoh_aligner(nvb.LOCAL, nvb.SimpleGotohScheme(2, -1, -2, -1))
        score, sink = nvb.batch_banded_alignment_score(15, al, sub_p, sub_t)
        ok = True
        for rb in (12, 8, 4):
            g = ResultGather(n, dst=0, device=dev, record_bytes=rb)
            out = g.gather(score, sink)
            if rank == 0:
                full_s, full_k = nvb.batch_banded_alignment_score(15, al, patterns, texts)
                ok = ok and bool(torch.equal(out[0], full_s) and torch.equal(out[1], full_k))
        best = (torch.arange(lo, hi, device=dev, dtype=torch.int64) * 977 << 32) | 5
        mapq = (torch.arange(lo, hi, device=dev) % 43).to(torch.uint8)
        table = RecordGather(n, 4, dst=0, device=dev).gather(alignment_records(best, mapq, lo))
        if rank == 0:
            ids = torch.arange(n, device=dev, dtype=torch.int64)
            ok = ok and bool((table[:, 3].to(torch.int64) == ids).all() and (table[:, 2].to(torch.int64) == ids % 43).all())
            q.put(ok)
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()


def test_two_rank_rccl_gather():
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs on one node (RCCL over xGMI); this box has %d" % torch.cuda.device_count())
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 200001, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True
