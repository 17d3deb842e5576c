// compat/nvbio/basic/cuda/sort.h -- SortBuffers / SortEnactor: radix sort of keys (and values) held in ping-pong device buffers
// (nvbio/basic/cuda/sort.h:38-185).  sort() leaves the sorted data in buffers.keys[selector] / values[selector].  Runs hipCUB's device
// radix sort on the double buffers; the enactor keeps its scratch between calls.
#pragma once
#include "../types.h"
#if defined(__HIPCC__)
#include <hipcub/hipcub.hpp>
#include <thrust/device_vector.h>
#include <stdexcept>

namespace nvbio {
namespace cuda {

template <typename Keys, typename Values = null_type>
struct SortBuffers
{
    SortBuffers() : selector(0) {}
    Keys   current_keys()   const { return keys[selector]; }
    Values current_values() const { return values[selector]; }
    uint32 selector;
    Keys   keys[2];
    Values values[2];
};

struct SortEnactor
{
    SortEnactor() {}
    ~SortEnactor() {}

    void sort(const uint32 count, SortBuffers<uint8*,  uint32*>& buffers, const uint32 begin_bit = 0, const uint32 end_bit = 8)  { pairs(count, buffers, begin_bit, end_bit); }
    void sort(const uint32 count, SortBuffers<uint16*, uint32*>& buffers, const uint32 begin_bit = 0, const uint32 end_bit = 16) { pairs(count, buffers, begin_bit, end_bit); }
    void sort(const uint32 count, SortBuffers<uint32*, uint32*>& buffers, const uint32 begin_bit = 0, const uint32 end_bit = 32) { pairs(count, buffers, begin_bit, end_bit); }
    void sort(const uint32 count, SortBuffers<uint32*, uint64*>& buffers, const uint32 begin_bit = 0, const uint32 end_bit = 32) { pairs(count, buffers, begin_bit, end_bit); }
    void sort(const uint32 count, SortBuffers<uint64*, uint32*>& buffers, const uint32 begin_bit = 0, const uint32 end_bit = 64) { pairs(count, buffers, begin_bit, end_bit); }
    void sort(const uint32 count, SortBuffers<uint8*>&  buffers, const uint32 begin_bit = 0, const uint32 end_bit = 8)  { keys(count, buffers, begin_bit, end_bit); }
    void sort(const uint32 count, SortBuffers<uint16*>& buffers, const uint32 begin_bit = 0, const uint32 end_bit = 16) { keys(count, buffers, begin_bit, end_bit); }
    void sort(const uint32 count, SortBuffers<uint32*>& buffers, const uint32 begin_bit = 0, const uint32 end_bit = 32) { keys(count, buffers, begin_bit, end_bit); }
    void sort(const uint32 count, SortBuffers<uint64*>& buffers, const uint32 begin_bit = 0, const uint32 end_bit = 64) { keys(count, buffers, begin_bit, end_bit); }

private:
    static void check(const hipError_t e) { if (e != hipSuccess) throw std::runtime_error(std::string("SortEnactor: ") + hipGetErrorString(e)); }
    void* scratch(const size_t bytes) { if (m_temp.size() < bytes) m_temp.resize(bytes); return thrust::raw_pointer_cast(m_temp.data()); }

    template <typename K, typename V>
    void pairs(const uint32 count, SortBuffers<K*, V*>& b, const uint32 begin_bit, const uint32 end_bit)
    {
        if (count == 0) return;
        hipcub::DoubleBuffer<K> k(b.keys[b.selector], b.keys[1u - b.selector]);
        hipcub::DoubleBuffer<V> v(b.values[b.selector], b.values[1u - b.selector]);
        size_t bytes = 0;
        check(hipcub::DeviceRadixSort::SortPairs(NULL, bytes, k, v, int(count), int(begin_bit), int(end_bit)));
        check(hipcub::DeviceRadixSort::SortPairs(scratch(bytes), bytes, k, v, int(count), int(begin_bit), int(end_bit)));
        if (k.selector) b.selector = 1u - b.selector;
    }
    template <typename K>
    void keys(const uint32 count, SortBuffers<K*>& b, const uint32 begin_bit, const uint32 end_bit)
    {
        if (count == 0) return;
        hipcub::DoubleBuffer<K> k(b.keys[b.selector], b.keys[1u - b.selector]);
        size_t bytes = 0;
        check(hipcub::DeviceRadixSort::SortKeys(NULL, bytes, k, int(count), int(begin_bit), int(end_bit)));
        check(hipcub::DeviceRadixSort::SortKeys(scratch(bytes), bytes, k, int(count), int(begin_bit), int(end_bit)));
        if (k.selector) b.selector = 1u - b.selector;
    }
    thrust::device_vector<uint8> m_temp;
};

} // namespace cuda
} // namespace nvbio
#endif
