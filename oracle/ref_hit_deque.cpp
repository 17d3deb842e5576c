// ref_hit_deque.cpp -- test harness around the REFERENCE's own interval heap (test infrastructure only).
//
// nvbio/basic/interval_heap.h is the one file on the path that compiles from its own source: it includes
// only <iterator>.  The two attribute macros it expects (NVBIO_FORCEINLINE, NVBIO_HOST_DEVICE, defined by
// nvbio/basic/types.h next to the CUDA includes that make that header unbuildable here) are given empty on
// the compiler command line (oracle/Makefile); no reference header is replaced or restated.  The header is
// read where it lies under /root/reference and only the built object goes to oracle/_ref/.
//
// Exposes the heap operations priority_deque<SeedHit, ..., hit_compare> performs
// (priority_deque.h:320-325, :354-357, :397-402, :412-417) on SeedHit words, with hit_compare
// (nvBowtie/bowtie2/cuda/seed_hit.h:235-244) on the 20-bit range size, to pin oracle/nvbio_oracle.c's
// restatement (tests/test_oracle_kat.py).
#include <cstdint>
#include <nvbio/basic/interval_heap.h>

namespace {
struct hit_compare {
    bool operator()(const uint64_t f, const uint64_t s) const { return ((f >> 32) & 0xFFFFFu) > ((s >> 32) & 0xFFFFFu); }
};
}

extern "C" __attribute__((visibility("default"))) void ref_hit_deque_push(uint64_t* a, uint32_t n)
{
    nvbio::heap::push_interval_heap(a, a + n, hit_compare());
}
extern "C" __attribute__((visibility("default"))) void ref_hit_deque_pop_bottom(uint64_t* a, uint32_t n)
{
    nvbio::heap::pop_interval_heap_min(a, a + n, hit_compare());
}
extern "C" __attribute__((visibility("default"))) void ref_hit_deque_pop_top(uint64_t* a, uint32_t n)
{
    nvbio::heap::pop_interval_heap_max(a, a + n, hit_compare());
}
extern "C" __attribute__((visibility("default"))) void ref_hit_deque_make(uint64_t* a, uint32_t n)
{
    nvbio::heap::make_interval_heap(a, a + n, hit_compare());      // priority_deque(seq, constructed = false), priority_deque.h:320-325
}
extern "C" __attribute__((visibility("default"))) int ref_hit_deque_is_heap(uint64_t* a, uint32_t n)
{
    return nvbio::heap::is_interval_heap(a, a + n, hit_compare()) ? 1 : 0;
}
