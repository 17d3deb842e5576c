// compat/nvbio/basic/cuda/pingpong_queues.h -- the input / output work queues nvBowtie's kernels pass along
// (nvbio/basic/cuda/pingpong_queues.h): a kernel reads in_queue[0, in_size) and appends to out_queue through *out_size.
#pragma once
#include "../types.h"
#include "arch.h"
#include "../thrust_view.h"
#if defined(__HIPCC__)
#include <thrust/device_vector.h>
#endif

namespace nvbio {
namespace cuda {

template <typename T = uint32>
struct PingPongQueuesView
{
    uint32   in_size;
    const T* in_queue;
    uint32*  out_size;
    T*       out_queue;
};

#if defined(__HIPCC__)
template <typename T = uint32>
struct PingPongQueues
{
    typedef PingPongQueuesView<T> device_view_type;
    typedef PingPongQueuesView<T> plain_view_type;

    PingPongQueues() : in_size(0) {}
    /// bytes the queues take for `size` entries; allocates them unless do_alloc is false
    uint64 resize_arena(const uint32 size, const bool do_alloc = true)
    {
        if (do_alloc) { in_size = 0; in_queue.resize(size); out_queue.resize(size); out_size.resize(1); }
        return 2ull * size * sizeof(T) + sizeof(uint32);
    }
    void   resize(const uint32 size) { in_size = size; }
    void   clear_output() { out_size[0] = 0; }
    void   swap() { in_size = out_size[0]; in_queue.swap(out_queue); }
    uint32 output_size() const { return out_size[0]; }
    const T* raw_input_queue()  const { return thrust::raw_pointer_cast(in_queue.data()); }
    const T* raw_output_queue() const { return thrust::raw_pointer_cast(out_queue.data()); }
    T*       raw_output_queue()       { return thrust::raw_pointer_cast(out_queue.data()); }
    device_view_type device_view()
    {
        device_view_type q;
        q.in_size = in_size;
        q.in_queue = thrust::raw_pointer_cast(in_queue.data());
        q.out_size = thrust::raw_pointer_cast(out_size.data());
        q.out_queue = thrust::raw_pointer_cast(out_queue.data());
        return q;
    }
    uint32                        in_size;
    thrust::device_vector<T>      in_queue;
    thrust::device_vector<uint32> out_size;
    thrust::device_vector<T>      out_queue;
};
#endif

} // namespace cuda

#if defined(__HIPCC__)
template <typename T> inline cuda::PingPongQueuesView<T> device_view(cuda::PingPongQueues<T>& queues) { return queues.device_view(); }
template <typename T> inline cuda::PingPongQueuesView<T> plain_view(cuda::PingPongQueues<T>& queues)  { return queues.device_view(); }
#endif

} // namespace nvbio
