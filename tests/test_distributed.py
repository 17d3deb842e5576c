"""world_size-2 test (gloo, CPU) of the multi-GPU layout: contiguous read sharding and the
gather of result records to rank 0.  The per-rank 'compute' here is the CPU oracle (this is a
test of the sharding / gather logic, not of a kernel)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from nvbio_amd import workloads as W
from nvbio_amd.distributed import RecordGather, ResultGather, alignment_records, shard_range, shard_sizes
from oracle import pyoracle as O


def test_shard_ranges_cover_everything():
    for n in (0, 1, 7, 100, 1001):
        for world in (1, 2, 3, 8):
            r = [shard_range(n, g, world) for g in range(world)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[i][1] == r[i + 1][0] for i in range(world - 1))
            assert sum(shard_sizes(n, world)) == n


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n, q, record_bytes=8):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        patterns, texts = W.make_sw_batch(n, seed=77, device="cpu")        # same batch on every rank
        hp, ht = O.StringSet.from_device(patterns), O.StringSet.from_device(texts)
        ht.length[5::7] = 60                                               # some texts shorter than their pattern: untouched sinks
        lo, hi = shard_range(n, rank, world)
        sub_p = O.StringSet(hp.words, hp.bits, hp.big_endian, hp.begin[lo:hi], hp.length[lo:hi])
        sub_t = O.StringSet(ht.words, ht.bits, ht.big_endian, ht.begin[lo:hi], ht.length[lo:hi])
        s, k = O.batch_banded_gotoh_score(15, O.LOCAL, (2, -1, -2, -1), sub_p, sub_t, n_threads=1)
        g = ResultGather(n, dst=0, device="cpu", record_bytes=record_bytes)
        for _ in range(2):                                                   # buffers are reusable
            out = g.gather(torch.from_numpy(s), torch.from_numpy(k.view(np.int32)))
        # sinks that do not fit a compact format must be reported, not truncated (a legitimate sink of (0xFF, 0xFF) or
        # (0xFFFF, 0xFFFF) would otherwise come back as "untouched")
        overflow_seen = True
        if record_bytes != 12:
            big = k.copy()
            if rank == 1:
                big[0] = (0xFF, 0xFF) if record_bytes == 4 else (0xFFFF, 0xFFFF)
            try:
                g.gather(torch.from_numpy(s), torch.from_numpy(big.view(np.int32)))
                overflow_seen = rank != 0
            except OverflowError:
                overflow_seen = rank == 0
        # the 16-byte alignment records of the end-to-end drivers
        best = torch.from_numpy(((np.arange(lo, hi, dtype=np.int64) * 977) << 32) | (np.arange(lo, hi, dtype=np.int64) & 0xFFFF))
        mapq = torch.from_numpy((np.arange(lo, hi) % 43).astype(np.uint8))
        rg = RecordGather(n, 4, dst=0, device="cpu")
        table = rg.gather(alignment_records(best, mapq, lo))
        if rank == 0:
            es, ek = O.batch_banded_gotoh_score(15, O.LOCAL, (2, -1, -2, -1), hp, ht, n_threads=1)
            ok = bool((out[0].numpy() == es).all() and (out[1].numpy().view(np.uint32) == ek).all()) and overflow_seen
            ids = np.arange(n, dtype=np.int64)
            t = table.numpy()
            ok = ok and bool((t[:, 3] == ids).all() and (t[:, 2] == ids % 43).all() and (t[:, 1].view(np.uint32) == ((ids * 977) & 0xFFFFFFFF)).all()
                             and (t[:, 0] == (ids & 0xFFFF)).all())
            q.put(ok)
        else:
            assert out is None and table is None and overflow_seen
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n,record_bytes", [(1001, 8), (64, 12), (1001, 4)])
def test_two_rank_shard_and_gather(n, record_bytes):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, q, record_bytes)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_cxx_shard_range_is_the_same_split(tmp_path):
    """include/nvbio_hip/multi_device.h (hip::shard_range / shard_sizes, what DeviceGroup's ranks use) against nvbio_amd.distributed:
    a host-only program built with g++ (the header's device-facing members are inline and not called)."""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("needs g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "shard.cpp"
    src.write_text(
        "#include <cstdio>\n#include <cstdlib>\n#include <nvbio_hip/multi_device.h>\n"
        "int main(int argc, char** argv) {\n"
        "  for (int i = 1; i + 1 < argc; i += 2) {\n"
        "    const unsigned long long n = strtoull(argv[i], 0, 10); const unsigned w = unsigned(atoi(argv[i + 1]));\n"
        "    const std::vector<nvbio::uint64> sz = nvbio::hip::shard_sizes(n, w);\n"
        "    for (unsigned r = 0; r < w; ++r) { const auto p = nvbio::hip::shard_range(n, r, w); printf(\"%llu %llu %llu \", (unsigned long long)p.first, (unsigned long long)p.second, (unsigned long long)sz[r]); }\n"
        "    printf(\"\\n\");\n  }\n  return 0;\n}\n")
    exe = tmp_path / "shard"
    lib = os.path.join(root, "nvbio_amd", "lib")
    cmd = ["g++", "-std=c++17", "-O1", "-I", os.path.join(root, "include"), str(src), "-o", str(exe), "-L", lib, "-lnvbio_hip", "-Wl,-rpath," + lib]
    if not os.path.exists(os.path.join(lib, "libnvbio_hip.so")):
        pytest.skip("libnvbio_hip.so is not built")
    subprocess.check_call(cmd)
    cases = [(0, 1), (1, 1), (1, 8), (7, 8), (8, 8), (9, 8), (10_000_000, 8), (200_000_000, 8), (25_000_001, 3), ((1 << 33) + 5, 7)]
    out = subprocess.check_output([str(exe)] + [str(x) for c in cases for x in c]).decode().strip().split("\n")
    for (n, w), line in zip(cases, out):
        v = [int(x) for x in line.split()]
        for r in range(w):
            lo, hi = shard_range(n, r, w)
            assert v[3 * r:3 * r + 3] == [lo, hi, hi - lo], (n, w, r)


def test_cxx_record_gather_and_device_group_over_a_host_transport():
    """The C++ side of the path's only collective without a multi-GPU node: nvbio_hip_gather_records (the plan of
    include/nvbio_hip/gather_plan.h) and hip::DeviceGroup -- one host thread per rank -- over a host-memory transport installed through
    nvbio_hip_comm_set_transport (tests/cxx/comm_plan_test.cpp): worlds of 2, 3, 5, 8; even, ragged, empty and one-rank-only shards;
    1 / 4 / 8-word records; several roots; shard_sizes() of ragged totals.  The root's buffer must hold every rank's records in rank
    order and global read order (a swapped offset or rank fails it), grouped receives must not deadlock against the sends, and a rank
    that throws before the collective must surface as an exception instead of a hang (DeviceGroup aborts the communicators)."""
    import ctypes
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "tests", "cxx", "libcomm_plan_test.so")
    assert os.path.exists(path), "build with python -c 'import __graft_entry__ as g; g.build()'"
    assert ctypes.CDLL(path).comm_plan_selftest() == 0


def _transport_worker(rank, world, port, n, width, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nvbio_amd.distributed import HostTransportComm, CxxRecordGather, shard_range
        comm = HostTransportComm()
        ok = True
        for dst in range(world):
            g = CxxRecordGather(comm, n, width, dst=dst, device="cpu")
            lo, hi = shard_range(n, rank, world)
            rec = (torch.arange(lo, hi, dtype=torch.int32).unsqueeze(1) * 7 + torch.arange(width, dtype=torch.int32).unsqueeze(0)).contiguous()
            for _ in range(2):                                    # twice: the table is reused, as bench.py's double buffer does
                t = g.gather(rec)
            if rank == dst:
                want = torch.arange(n, dtype=torch.int32).unsqueeze(1) * 7 + torch.arange(width, dtype=torch.int32).unsqueeze(0)
                ok = ok and bool(torch.equal(t, want)) and all(bool(torch.equal(g.shard(r), want[shard_range(n, r, world)[0]:shard_range(n, r, world)[1]])) for r in range(world))
            else:
                ok = ok and t is None
        comm.close()
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n,width", [(2, 1001, 4), (3, 17, 1), (2, 1, 2)])
def test_cxx_gather_over_a_gloo_backed_transport(world, n, width):
    """The C++ gather (nvbio_hip_gather_records: the plan of include/nvbio_hip/gather_plan.h executed through the transport seam) between REAL
    processes: nvbio_amd.distributed.HostTransportComm fills the seam with gloo sends / receives, so everything bench.py --gpus N runs above
    ncclSend / ncclRecv -- CxxRecordGather's tables, counts, offsets, every root, ragged and empty shards -- runs here with world_size > 1."""
    if not os.path.exists(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nvbio_amd", "lib", "libnvbio_hip.so")):
        pytest.skip("libnvbio_hip.so is not built")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_transport_worker, args=(r, world, port, n, width, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    got = sorted(q.get(timeout=5) for _ in range(world))
    assert got == [(r, True) for r in range(world)]
