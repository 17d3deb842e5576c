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


class ResultGather:
    """Gathers per-rank (score[n_r], sink[n_r,2]) records to `dst` in rank order.

    xGMI is point-to-point, so this is a gather (send/recv to the root over its 7 links), not a
    ring all-gather: the root receives (G-1)/G of the bytes once, nobody else receives anything.
    compact=True (sinks below 65536, i.e. any short-read batch) packs a record into 8 bytes
    {score, sink.x << 16 | sink.y} -- at 3 G reads/s per GPU the root of an 8-GPU node then takes in
    ~190 GB/s instead of ~280 GB/s.  Buffers are allocated once; `gather()` can be enqueued on a side
    stream to overlap the next batch's kernel."""

    def __init__(self, n_total, dst=0, device=None, group=None, compact=True):
        self.group, self.dst, self.compact = group, dst, compact
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.sizes = shard_sizes(n_total, self.world)
        self.n_total = n_total
        self.pad = max(self.sizes) if self.sizes else 0
        self.width = 2 if compact else 3
        self.bufs = None
        if self.world > 1 and self.rank == dst:
            self.bufs = [torch.empty((self.pad, self.width), dtype=torch.int32, device=device) for _ in range(self.world)]
        self.send = torch.empty((self.pad, self.width), dtype=torch.int32, device=device) if self.world > 1 else None

    def gather(self, score, sink, concat=True):
        """Returns (score[n_total], sink[n_total,2]) on dst, None elsewhere.  concat=False leaves the
        records in the per-rank receive buffers (`self.bufs[r][:self.sizes[r]]`) and returns True on dst:
        no extra copy, for callers that consume them in place."""
        n = score.numel()
        assert n == self.sizes[self.rank]
        if self.world == 1:
            return score, sink
        sink = sink.view(-1, 2)
        self.send[:n, 0] = score
        if self.compact:
            self.send[:n, 1] = (sink[:, 0] << 16) | (sink[:, 1] & 0xFFFF)
        else:
            self.send[:n, 1:] = sink
        dist.gather(self.send, self.bufs if self.rank == self.dst else None, dst=self.dst, group=self.group)
        if self.rank != self.dst:
            return None
        if not concat:
            return True
        rec = torch.cat([self.bufs[r][: self.sizes[r]] for r in range(self.world)], dim=0)
        if self.compact:
            sx = (rec[:, 1] >> 16) & 0xFFFF
            sy = rec[:, 1] & 0xFFFF
            # an invalid sink (0xFFFFFFFF, 0xFFFFFFFF) round-trips as (0xFFFF, 0xFFFF): restore it
            bad = (sx == 0xFFFF) & (sy == 0xFFFF)
            sx = torch.where(bad, torch.full_like(sx, -1), sx)
            sy = torch.where(bad, torch.full_like(sy, -1), sy)
            return rec[:, 0].contiguous(), torch.stack([sx, sy], dim=1).contiguous()
        return rec[:, 0].contiguous(), rec[:, 1:].contiguous()
