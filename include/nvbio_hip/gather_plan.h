/* nvbio_hip/gather_plan.h -- the arithmetic of the path's only collective, as a pure function (C, header-only).
 *
 * Every rank aligns a contiguous block of the reads; rank r holds counts[r] fixed-size records and the root's receive buffer must hold
 * all of them in RANK ORDER (record k of rank r at byte (counts[0] + ... + counts[r-1] + k) * record_bytes) -- the order in which the
 * reference's single shared output sees the reads (nvBowtie/nvBowtie.cpp:809-864).  What a rank has to do for that:
 *     a non-root rank    one SEND of its records to the root (none when it has no records)
 *     the root           one RECV per other rank that has records, at that rank's offset, and one COPY of its own records to its offset
 * nvbio_hip_gather_records (nvbio_amd/csrc/comm.hip) executes this plan over RCCL -- the RECVs inside one ncclGroupStart / ncclGroupEnd;
 * the CPU suite executes it over a host-memory transport at worlds of 1..8 with ragged and empty shards (tests/cxx/comm_plan_test.cpp),
 * so that a swapped offset or a mis-ordered rank shows up without a multi-GPU node. */
#ifndef NVBIO_HIP_GATHER_PLAN_H
#define NVBIO_HIP_GATHER_PLAN_H
#include <stdint.h>

enum { NVBIO_HIP_GATHER_RECV = 0, NVBIO_HIP_GATHER_SEND = 1, NVBIO_HIP_GATHER_COPY = 2 };

typedef struct nvbio_hip_gather_op {
    int32_t  kind;      /* NVBIO_HIP_GATHER_* */
    int32_t  peer;      /* RECV: the sending rank; SEND: the root; COPY: this rank */
    uint64_t offset;    /* RECV / COPY: byte offset into the root's receive buffer; SEND: 0 (the whole send buffer) */
    uint64_t bytes;
} nvbio_hip_gather_op;

/* Fills ops[] (room for `world` entries) with what `rank` does; returns the number of operations, or -1 on invalid arguments. */
static inline int nvbio_hip_gather_plan(const uint64_t* counts, int world, int rank, int root, uint32_t record_bytes, nvbio_hip_gather_op* ops)
{
    int n = 0, k;
    uint64_t offset = 0;
    if (!counts || !ops || world < 1 || rank < 0 || rank >= world || root < 0 || root >= world || record_bytes == 0) return -1;
    if (rank != root)
    {
        if (counts[rank] == 0) return 0;
        ops[0].kind = NVBIO_HIP_GATHER_SEND; ops[0].peer = root; ops[0].offset = 0; ops[0].bytes = counts[rank] * record_bytes;
        return 1;
    }
    for (k = 0; k < world; ++k)
    {
        const uint64_t bytes = counts[k] * record_bytes;
        if (bytes)
        {
            ops[n].kind = (k == root) ? NVBIO_HIP_GATHER_COPY : NVBIO_HIP_GATHER_RECV;
            ops[n].peer = k; ops[n].offset = offset; ops[n].bytes = bytes;
            ++n;
        }
        offset += bytes;
    }
    return n;
}

#endif
