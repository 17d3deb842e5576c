// compat/nvbio/basic/cuda/work_queue.h -- WorkQueue<PolicyTag, WorkUnit, BLOCKDIM>::consume(stream): run a stream of work units to
// completion on the device, with optional utilization counters (nvbio/basic/cuda/work_queue.h:44-338, work_queue_inl.h).  A stream offers
// size() and get(i, &unit, slot); a unit offers run(stream) -> "call me again".  This layer implements the in-place policy (every lane
// keeps its unit until it is done; the grid strides over the stream) for every policy tag: the persistent / ordered / multi-pass
// policies of the reference reschedule the same units for occupancy and compute the same results.
#pragma once
#include "../types.h"
#include "../numbers.h"
#include "arch.h"
#if defined(__HIPCC__)
#include <thrust/device_vector.h>
#include <thrust/fill.h>

namespace nvbio {
namespace cuda {

struct InplaceQueueTag {};
struct PersistentWarpsQueueTag {};
struct PersistentThreadsQueueTag {};
struct OrderedQueueTag {};
template <typename T> struct MultiPassQueueTag {};

enum WorkQueueStatsEvent { STREAM_EVENT = 0, RUN_EVENT = 1 };

/// device-side handle on seven uint64 counters: active lanes x2, issued warps x2, iterations {sum, max, events}
struct WorkQueueStatsView
{
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE WorkQueueStatsView() : active_lanes(NULL), issued_warps(NULL), iterations(NULL) {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE WorkQueueStatsView(uint64* _active_lanes, uint64* _issued_warps, uint64* _iterations)
        : active_lanes(_active_lanes), issued_warps(_issued_warps), iterations(_iterations) {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool valid() const { return active_lanes != NULL; }

    /// count the lanes of this (virtual, 32-lane) warp that are active at an event; one lane per warp reports
    NVBIO_FORCEINLINE __device__ void sample(const WorkQueueStatsEvent type)
    {
        if (!valid()) return;
        const uint32 active = priv::half_ballot(1);
        if (__popc(active >> warp_tid()) == 1)
        {
            atomicAdd(reinterpret_cast<unsigned long long*>(active_lanes + type), (unsigned long long)__popc(active));
            atomicAdd(reinterpret_cast<unsigned long long*>(issued_warps + type), 1ull);
        }
    }
    NVBIO_FORCEINLINE __device__ void sample_iterations(const uint32 i)
    {
        if (!valid()) return;
        atomicAdd(reinterpret_cast<unsigned long long*>(iterations), (unsigned long long)i);
        atomicMax(reinterpret_cast<uint32*>(iterations + 1u), i);
        atomicAdd(reinterpret_cast<unsigned long long*>(iterations + 2u), 1ull);
    }
    uint64* active_lanes;
    uint64* issued_warps;
    uint64* iterations;
};

struct WorkQueueStats
{
    typedef WorkQueueStatsView View;
    WorkQueueStats() : counters(7, uint64(0)) {}
    void clear() { thrust::fill(counters.begin(), counters.end(), uint64(0)); }
    View view() { uint64* c = thrust::raw_pointer_cast(counters.data()); return View(c, c + 2u, c + 4u); }
    float utilization(const WorkQueueStatsEvent type) const { const uint64 w = counters[2 + type]; return w ? float(uint64(counters[0 + type])) / float(w * Arch::WARP_SIZE) : 1.0f; }
    float avg_iterations() const { const uint64 n = counters[6]; return n ? float(uint64(counters[4])) / float(n) : 0.0f; }
    float max_iterations() const { return float(uint64(counters[5])); }
private:
    thrust::device_vector<uint64> counters;
};
inline WorkQueueStatsView view(WorkQueueStats* stats) { return stats ? stats->view() : WorkQueueStatsView(); }

/// how a unit is relocated between queue slots (policies that compact their queues call it; the in-place policy never moves a unit)
struct DefaultMover
{
    template <typename WorkStreamT, typename WorkUnitT>
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE void move(const WorkStreamT&, const uint2, WorkUnitT* src_unit, const uint2, WorkUnitT* dst_unit) const { *dst_unit = *src_unit; }
};

namespace wq {
template <uint32 BLOCKDIM, typename WorkUnitT, typename WorkStreamT>
__global__ void inplace_work_queue_kernel(const WorkStreamT stream, WorkQueueStatsView stats)
{
    const uint32 stride = gridDim.x * BLOCKDIM, lane = threadIdx.x + blockIdx.x * BLOCKDIM, n = stream.size();
    WorkUnitT unit;
    for (uint32 base = 0; base < n; base += stride)
    {
        const uint32 work_id = base + lane;
        if (work_id >= n) continue;
        stream.get(work_id, &unit, make_uint2(lane, 0u));
        stats.sample(STREAM_EVENT);
        uint32 iterations = 0;
        do { stats.sample(RUN_EVENT); ++iterations; } while (unit.run(stream));
        stats.sample_iterations(iterations);
    }
}
} // namespace wq

template <typename PolicyTag, typename WorkUnitT, uint32 BLOCKDIM>
struct WorkQueue
{
    typedef WorkUnitT WorkUnit;
    WorkQueue() {}
    void set_capacity(const uint32) {}
    template <typename WorkStream>
    void consume(const WorkStream stream, WorkQueueStats* stats = NULL) { consume(stream, DefaultMover(), stats); }
    template <typename WorkStream, typename WorkMover>
    void consume(const WorkStream stream, const WorkMover, WorkQueueStats* stats = NULL)
    {
        const uint32 n_blocks = uint32(max_active_blocks(wq::inplace_work_queue_kernel<BLOCKDIM, WorkUnit, WorkStream>, BLOCKDIM, 0u));
        hipLaunchKernelGGL((wq::inplace_work_queue_kernel<BLOCKDIM, WorkUnit, WorkStream>), dim3(n_blocks), dim3(BLOCKDIM), 0, 0, stream, view(stats));
    }
};

} // namespace cuda
} // namespace nvbio
#endif
