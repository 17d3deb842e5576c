"""Multi-GPU layout of the hot path: one process per GPU, the read batch sharded in contiguous
blocks, the FM-index replicated, no collective inside the pipeline, and one gather of the
fixed-size result records to rank 0 at the end.  The reference has no
collective at all (one host thread per device writing to a shared output,
nvBowtie/nvBowtie.cpp:809-864); the gather below is what replaces its shared output object.

Backend: "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests."""
import torch
import torch.distributed as dist


def shard_range(n_total, rank, world):
    """Contiguous block sharding: rank g owns reads [g*ceil(R/G), min(R, (g+1)*ceil(R/G)))."""
    per = (n_total + world - 1) // world
    lo = min(n_total, rank * per)
    hi = min(n_total, lo + per)
    return lo, hi


def shard_sizes(n_total, world):
    return [shard_range(n_total, r, world)[1] - shard_range(n_total, r, world)[0] for r in range(world)]


class RecordGather:
    """Gathers fixed-width int32 records [n_r, width] from every rank to `dst`, in rank order -- the collective of the
    path (16 B per read single-end, 32 B per pair).  xGMI is point-to-point, so this is a gather (the root
    receives (G-1)/G of the bytes once over its 7 links, nobody else receives anything), not a ring all-gather.
    Buffers are allocated once; `gather()` can be enqueued on a side stream to overlap the next batch's kernels.
    One extra row per rank carries a status word (see ResultGather)."""

    def __init__(self, n_total, width, dst=0, device=None, group=None):
        self.group, self.dst, self.width = group, dst, int(width)
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.sizes = shard_sizes(n_total, self.world)
        self.n_total = n_total
        self.pad = max(self.sizes) if self.sizes else 0
        self.bufs = None
        if self.world > 1 and self.rank == dst:
            self.bufs = [torch.empty((self.pad + 1, self.width), dtype=torch.int32, device=device) for _ in range(self.world)]
        self.send = torch.zeros((self.pad + 1, self.width), dtype=torch.int32, device=device) if self.world > 1 else None

    def _exchange(self):
        if self.send.is_cuda and dist.get_backend(self.group) != "nccl":
            # gloo cannot gather device tensors: stage through the host (debug / CPU-test configurations only)
            host = [torch.empty(b.shape, dtype=b.dtype) for b in self.bufs] if self.rank == self.dst else None
            dist.gather(self.send.cpu(), host, dst=self.dst, group=self.group)
            if host is not None:
                for b, h in zip(self.bufs, host):
                    b.copy_(h)
        else:
            dist.gather(self.send, self.bufs if self.rank == self.dst else None, dst=self.dst, group=self.group)

    def status(self):
        """On dst: the per-rank status words of the last gather (ONE stacked device-to-host copy for all ranks)."""
        if self.bufs is None:
            return []
        return [int(v) for v in torch.stack([b[self.pad, 0] for b in self.bufs]).cpu().tolist()]

    def gather(self, records, concat=True, status=None):
        """records: int32 [n_r, width] of this rank.  Returns the [n_total, width] table on dst (None elsewhere);
        concat=False leaves the records in `self.bufs[r][:self.sizes[r]]` and returns True on dst."""
        n = records.shape[0]
        assert n == self.sizes[self.rank] and records.shape[1] == self.width
        if self.world == 1:
            return records
        self.send[:n] = records
        self.send[self.pad, 0] = 0 if status is None else status        # always written: no stale word from an earlier call
        self._exchange()
        if self.rank != self.dst:
            return None
        if not concat:
            return True
        return torch.cat([self.bufs[r][: self.sizes[r]] for r in range(self.world)], dim=0)


def pack_result_records(score, sink, record_bytes):
    """(score[n], sink[n,2]) -- BestSink<int32> -- as int32 records of 12 / 8 / 4 bytes (see ResultGather) plus a device flag that is
    non-zero when some record does not fit the compact format."""
    sink = sink.view(-1, 2)
    n = score.numel()
    untouched = (sink[:, 0] == -1) & (sink[:, 1] == -1)
    status = None
    rec = torch.empty((n, record_bytes // 4), dtype=torch.int32, device=score.device)
    if record_bytes == 4:
        fits = untouched | ((sink[:, 0] >= 0) & (sink[:, 0] < 0xFF) & (sink[:, 1] >= 0) & (sink[:, 1] < 0xFF) & (score > -32768) & (score < 32768))
        status = (~fits).any().to(torch.int32)
        s16 = torch.clamp(score, min=-32768)          # only the untouched-sink score lies below
        rec[:, 0] = (s16 << 16) | ((sink[:, 0] & 0xFF) << 8) | (sink[:, 1] & 0xFF)
    elif record_bytes == 8:
        fits = untouched | ((sink[:, 0] >= 0) & (sink[:, 0] < 0xFFFF) & (sink[:, 1] >= 0) & (sink[:, 1] < 0xFFFF))
        status = (~fits).any().to(torch.int32)
        rec[:, 0] = score
        rec[:, 1] = (sink[:, 0] << 16) | (sink[:, 1] & 0xFFFF)
    else:
        rec[:, 0] = score
        rec[:, 1:] = sink
    return rec, status


class CxxComm:
    """The C++ / RCCL side of a launcher-driven run (include/nvbio_hip/multi_device.h: DeviceGroup::from_unique_id): rank 0 makes the
    RCCL unique id through the C-ABI, torch.distributed ships its 128 bytes, every rank opens its communicator on its own device.
    After this torch.distributed takes no part in the data path."""

    def __init__(self, group=None):
        import ctypes as C
        from ._lib import lib, check
        self.L, self.C = lib(), C
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # Every rank reaches every collective below whatever happens locally: a rank that cannot bind RCCL (or rank 0 failing to make the
        # id) says so in the exchange and ALL ranks raise together -- nobody is left waiting inside ncclCommInitRank for a peer that gave up.
        ident = (C.c_uint8 * 128)()
        problem = None
        if not self.L.nvbio_hip_comm_available():
            problem = "RCCL could not be bound (librccl.so.1)"
        elif self.rank == 0:
            err = self.L.nvbio_hip_comm_unique_id(ident)
            if err != 0:
                problem = "nvbio_hip_comm_unique_id failed with %d" % err
        if self.world > 1:
            reports = [None] * self.world
            dist.all_gather_object(reports, (problem, bytes(ident) if self.rank == 0 else None), group=group)
        else:
            reports = [(problem, bytes(ident))]
        bad = [(r, p[0]) for r, p in enumerate(reports) if p[0] is not None]
        if bad:
            raise RuntimeError("C++ / RCCL communicator not opened: " + "; ".join("rank %d: %s" % b for b in bad))
        ident = (C.c_uint8 * 128).from_buffer_copy(reports[0][1])
        self.comm = C.c_void_p()
        check(self.L.nvbio_hip_comm_init_rank(C.byref(self.comm), self.world, self.rank, ident), "nvbio_hip_comm_init_rank")

    def close(self):
        if self.comm:
            self.L.nvbio_hip_comm_destroy(self.comm)
            self.comm = None


class HostTransportComm:
    """CxxComm's interface where RCCL cannot serve -- two ranks sharing one device (tests/test_bench_multirank_gpu.py), or no device at all
    (the CPU suite): the transport seam of the C-ABI (nvbio_hip_comm_set_transport) is filled with functions that carry the bytes through
    torch.distributed's gloo group, staged through host memory.  Everything above the seam -- nvbio_hip_gather_records, the gather plan, the
    record tables of CxxRecordGather, bench.py's double-buffered gather -- is the code a multi-GPU node runs; only ncclSend / ncclRecv
    themselves are replaced.  One per process (the transport table is process-wide)."""

    def __init__(self, group=None):
        import ctypes as C
        from ._lib import lib
        self.L, self.C, self.group = lib(), C, group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.on_device = torch.cuda.is_available()
        self._queued = None            # receives posted inside a group, executed at its end (NCCL's group semantics)
        vp, u64, ci = C.c_void_p, C.c_uint64, C.c_int
        F = C.CFUNCTYPE

        def guard(fn):
            def wrapped(*a):
                try:
                    return fn(*a) or 0
                except Exception as e:      # noqa: BLE001 -- an exception must not unwind through the C caller
                    import sys
                    sys.stderr.write("HostTransportComm: %r\n" % (e,))
                    return 999
            return wrapped

        def f_rank(comm, r, w):
            r[0], w[0] = self.rank, self.world

        def f_start(comm):
            self._queued = []

        def f_end(comm):
            q, self._queued = self._queued or [], None
            for buf, nbytes, peer, stream in q:
                self._recv_now(buf, nbytes, peer, stream)

        def f_send(comm, buf, nbytes, peer, stream):
            host = torch.empty(int(nbytes), dtype=torch.uint8)
            self._copy(host.data_ptr(), buf, nbytes, 2, stream)
            dist.send(host, dst=self._global(peer), group=self.group)

        def f_recv(comm, buf, nbytes, peer, stream):
            if self._queued is not None:
                self._queued.append((buf, nbytes, peer, stream))
            else:
                self._recv_now(buf, nbytes, peer, stream)

        def f_copy(comm, dst, src, nbytes, stream):
            self._copy(dst, src, nbytes, 3, stream)

        def f_abort(comm):
            pass

        self._types = (F(ci, vp, C.POINTER(ci), C.POINTER(ci)), F(ci, vp), F(ci, vp), F(ci, vp, vp, u64, ci, vp), F(ci, vp, vp, u64, ci, vp),
                       F(ci, vp, vp, vp, u64, vp), F(ci, vp))
        fns = (f_rank, f_start, f_end, f_send, f_recv, f_copy, f_abort)
        self._callbacks = [t(guard(f)) for t, f in zip(self._types, fns)]           # kept alive for as long as the table is installed

        class Table(C.Structure):
            _fields_ = [(n, t) for n, t in zip(("rank", "group_start", "group_end", "send", "recv", "copy", "abort"), self._types)]
        self._table = Table(*self._callbacks)
        self.L.nvbio_hip_comm_set_transport.argtypes = [C.c_void_p]
        self.L.nvbio_hip_comm_set_transport(C.byref(self._table))
        self._token = C.c_int(self.rank)
        self.comm = C.cast(C.pointer(self._token), C.c_void_p)                      # any non-null handle: the table's functions ignore it

    def _global(self, peer):
        return dist.get_global_rank(self.group, peer) if self.group is not None else peer

    def _copy(self, dst, src, nbytes, kind, stream):
        """kind: 1 host -> device, 2 device -> host, 3 device -> device (nvbio_hip_memcpy's); plain memmove without a device"""
        if nbytes == 0:
            return
        if not self.on_device:
            self.C.memmove(dst, src, int(nbytes)); return
        from ._lib import check
        check(self.L.nvbio_hip_memcpy(self.C.c_void_p(dst), self.C.c_void_p(src), self.C.c_uint64(nbytes), kind, self.C.c_void_p(stream)), "nvbio_hip_memcpy")
        check(self.L.nvbio_hip_stream_synchronize(self.C.c_void_p(stream)), "nvbio_hip_stream_synchronize")

    def _recv_now(self, buf, nbytes, peer, stream):
        host = torch.empty(int(nbytes), dtype=torch.uint8)
        dist.recv(host, src=self._global(peer), group=self.group)
        self._copy(buf, host.data_ptr(), nbytes, 1, stream)

    def close(self):
        if self.comm:
            self.L.nvbio_hip_comm_set_transport(None)
            self.comm = None


class CxxRecordGather:
    """RecordGather's interface over nvbio_hip_gather_records (grouped ncclSend / ncclRecv issued from C++ on the caller's current HIP
    stream): int32 records [n_r, width] from every rank to `dst`, in rank order, into ONE contiguous [n_total, width] table on the root."""

    def __init__(self, comm, n_total, width, dst=0, device=None):
        import ctypes as C
        self.comm, self.dst, self.width = comm, dst, int(width)
        self.rank, self.world = comm.rank, comm.world
        self.sizes = shard_sizes(n_total, self.world)
        self.n_total = n_total
        self.counts = (C.c_uint64 * self.world)(*self.sizes)
        self.table = torch.empty((n_total, self.width), dtype=torch.int32, device=device) if self.rank == dst else None
        self.offsets = [sum(self.sizes[:r]) for r in range(self.world)]

    def gather(self, records, concat=True, status=None):
        from ._lib import check, current_stream_ptr
        import ctypes as C
        assert records.shape[0] == self.sizes[self.rank] and records.shape[1] == self.width and records.dtype == torch.int32 and records.is_contiguous()
        recv = C.c_void_p(self.table.data_ptr()) if self.table is not None else None
        check(self.comm.L.nvbio_hip_gather_records(self.comm.comm, C.c_void_p(records.data_ptr()), self.counts, self.width * 4, recv, self.dst, current_stream_ptr()),
              "nvbio_hip_gather_records")
        if self.rank != self.dst:
            return None
        return self.table if concat else True

    def shard(self, r):
        """the root's view of rank r's records"""
        return self.table[self.offsets[r]: self.offsets[r] + self.sizes[r]]


class ResultGather(RecordGather):
    """Gathers per-rank (score[n_r], sink[n_r,2]) records -- BestSink<int32> -- to `dst` in rank order.

    Record sizes (`record_bytes`):
      12  {score, sink.x, sink.y}                          lossless, the default
       8  {score, sink.x << 16 | sink.y}                   sinks below 65535 (any short-read batch)
       4  {score:int16 << 16 | sink.x:8 << 8 | sink.y:8}   sinks below 255 and |score| < 32767, e.g. 100 bp
          reads in band-15 windows: at 3.3 G reads/s per GPU the root of an 8-GPU node takes in ~93 GB/s
    An untouched sink (score -2^30, sink (-1,-1): text shorter than pattern) travels as all-ones fields in the compact
    formats.  Every rank checks on the device that its records fit the chosen format (no host sync) and ships the verdict
    in the status row; the root raises OverflowError instead of returning truncated values (`concat=False` callers ask
    `overflowed()`)."""

    def __init__(self, n_total, dst=0, device=None, group=None, compact=False, record_bytes=None):
        if record_bytes is None:
            record_bytes = 8 if compact else 12
        assert record_bytes in (4, 8, 12)
        self.record_bytes = record_bytes
        super().__init__(n_total, record_bytes // 4, dst=dst, device=device, group=group)

    def overflowed(self):
        return any(self.status())

    def gather(self, score, sink, concat=True):
        """Returns (score[n_total], sink[n_total,2]) on dst, None elsewhere.  concat=False leaves the
        records in the per-rank receive buffers (`self.bufs[r][:self.sizes[r]]`) and returns True on dst:
        no extra copy, for callers that consume them in place."""
        n = score.numel()
        assert n == self.sizes[self.rank]
        if self.world == 1:
            return score, sink
        rec, status = pack_result_records(score, sink, self.record_bytes)
        self.send[:n] = rec
        self.send[self.pad, 0] = 0 if status is None else status          # always written: no stale word from an earlier call
        self._exchange()
        if self.rank != self.dst:
            return None
        if not concat:
            return True
        # 12-byte records are lossless: their status is always 0 and reading it back would be a host sync for nothing
        if self.record_bytes != 12 and self.overflowed():
            raise OverflowError("ResultGather: a rank holds sinks / scores that do not fit %d-byte records; use record_bytes=12" % self.record_bytes)
        rec = torch.cat([self.bufs[r][: self.sizes[r]] for r in range(self.world)], dim=0)
        if self.record_bytes == 12:
            return rec[:, 0].contiguous(), rec[:, 1:].contiguous()
        if self.record_bytes == 8:
            sc = rec[:, 0].contiguous()
            sx, sy, none = (rec[:, 1] >> 16) & 0xFFFF, rec[:, 1] & 0xFFFF, 0xFFFF
        else:
            sc = rec[:, 0] >> 16                            # arithmetic shift: sign-extends the int16 score
            sx, sy, none = (rec[:, 0] >> 8) & 0xFF, rec[:, 0] & 0xFF, 0xFF
        # an untouched sink (0xFFFFFFFF, 0xFFFFFFFF) travels as all-ones fields (no fitting sink has them): restore it
        bad = (sx == none) & (sy == none)
        sx = torch.where(bad, torch.full_like(sx, -1), sx)
        sy = torch.where(bad, torch.full_like(sy, -1), sy)
        if self.record_bytes == 4:
            sc = torch.where(bad & (sc == -32768), torch.full_like(sc, -(1 << 30)), sc)
        return sc.contiguous(), torch.stack([sx, sy], dim=1).contiguous()


def alignment_records(best, mapq, first_read_id):
    """nvBowtie's per-read result as the 16-byte record the ranks gather: {io::Alignment low word (score, edit
    distance, strand / mate / pairing flags), alignment position, MAPQ, global read id}.  best: int64 [n] io::Alignment words
    of the best alignments, mapq: uint8 [n]."""
    n = best.numel()
    rec = torch.empty((n, 4), dtype=torch.int32, device=best.device)
    rec[:, 0] = (best & 0xFFFFFFFF).to(torch.int32)
    rec[:, 1] = ((best >> 32) & 0xFFFFFFFF).to(torch.int32)
    rec[:, 2] = mapq.to(torch.int32)
    rec[:, 3] = torch.arange(first_read_id, first_read_id + n, device=best.device, dtype=torch.int64).to(torch.int32)
    return rec
