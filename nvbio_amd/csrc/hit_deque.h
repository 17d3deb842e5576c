// hit_deque.h -- the per-read seed-hit deque of nvBowtie, as device functions over a read's slot of the
// hit arena (SeedHit word pairs: .x = range_begin, .y = {range_delta:20, pos:10, rc:1, indexdir:1}).
//
// Replaces priority_deque<SeedHit, vector_view<SeedHit*>, hit_compare> (nvbio/basic/priority_deque.h:329-421,
// nvBowtie/bowtie2/cuda/seed_hit_deque_array.h:175-354) over the interval heap of nvbio/basic/interval_heap.h
// (:195-260, :356-533) with hit_compare = "larger range first" (seed_hit.h:235-244): even slots are interval
// lower bounds, odd slots upper bounds; slot 0 holds a hit of largest range (what pop_bottom drops when the
// deque is full), slot 1 (slot 0 when alone) a hit of smallest range (top(): what the selection stage
// extends first).  The array order decides which of several equal-sized hits is met first -- by top() and by
// the randomized selection, which samples array slots -- so every exchange is the reference's; the mappers
// build the deque directly in the read's arena slot (no 512-entry local-memory copy, no arena atomics).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nvb {

// where the heap's elements live: SeedHit words in a read's arena row, or -- for rebuilding a heap on chip (select.hip) -- one key per hit,
// (range size << 8 | slot), in LDS, lane-interleaved
struct RowHits
{
    typedef uint2 value_type;
    uint2* p;
    __device__ __forceinline__ uint2& operator[](const int i) const { return p[i]; }
    __device__ __forceinline__ static bool before(const uint2 f, const uint2 s) { return (f.y & 0xFFFFFu) > (s.y & 0xFFFFFu); }
};
struct LdsKeys
{
    typedef uint32_t value_type;
    uint32_t* p;
    __device__ __forceinline__ uint32_t& operator[](const int i) const { return p[i * 256]; }
    __device__ __forceinline__ static bool before(const uint32_t f, const uint32_t s) { return (f >> 8) > (s >> 8); }
};

template <typename Cells>
struct HitDequeT
{
    Cells a;

    __device__ __forceinline__ static bool before(const typename Cells::value_type f, const typename Cells::value_type s) { return Cells::before(f, s); }
    __device__ __forceinline__ void swap(const int i, const int j) const { const typename Cells::value_type t = a[i]; a[i] = a[j]; a[j] = t; }

    __device__ void sift_up(int i, const bool lower, const int stop = 2) const
    {
        while (i >= stop) {
            const int parent = ((i / 2 - 1) | 1) ^ (lower ? 1 : 0);
            if (!(lower ? before(a[i], a[parent]) : before(a[parent], a[i]))) break;
            swap(i, parent);
            i = parent;
        }
    }
    __device__ void leaf_upper(const int n, const int i, const int stop = 2) const
    {
        const int co = (i * 2 < n) ? i * 2 : (i ^ 1);
        if (before(a[i], a[co])) { swap(i, co); sift_up(co, true, stop); }
        else sift_up(i, false, stop);
    }
    __device__ void leaf_lower(const int n, const int i, const int stop = 2) const
    {
        int co = i | 1;
        if (co >= n) { if (co == 1) return; co = (co / 2 - 1) | 1; }
        if (before(a[co], a[i])) { swap(i, co); sift_up(co, false, stop); }
        else sift_up(i, true, stop);
    }
    __device__ void sift_down(const int n, int i, const bool lower, const int stop = 2) const
    {
        const int end_parent = n / 2 - ((lower && (n & 3) == 0) ? 2 : 1);
        while (i < end_parent) {
            int child = i * 2 + (lower ? 2 : 1);
            if (lower ? before(a[child + 2], a[child]) : before(a[child], a[child + 2])) child += 2;
            swap(i, child);
            i = child;
        }
        if (i <= end_parent + (lower ? 0 : 1)) {
            int child = i * 2 + (lower ? 2 : 1);
            if (child < n) {
                if (!lower && child + 1 < n && before(a[child], a[child + 1])) {
                    ++child;
                    swap(i, child);
                    leaf_lower(n, child, stop);
                    return;
                }
                swap(i, child);
                i = child;
            }
        }
        if (lower) leaf_lower(n, i, stop); else leaf_upper(n, i, stop);
    }
    // The bottom-up construction (make_interval_heap, interval_heap.h:356-384).  The reference's selection kernels run it every time they take
    // hits[read_id]: SeedHitDequeArrayDeviceView::get_deque(read_id, build_heap = false) hands that false to priority_deque(seq, constructed)
    // (seed_hit_deque_array_inl.h:104-109, priority_deque.h:320-325).  The ranges shrank since the last round, equal sizes are common, and the
    // probability-tree leaves do not move with the hits -- so the arrangement this leaves decides later picks, and every exchange is kept.
    __device__ void make(const int n) const
    {
        if (n <= 1) return;
        const int end_parent = n / 2 - 1;
        int i = n ^ (n & 1);
        do {
            i -= 2;
            const int stop = (i <= end_parent) ? (i * 2 + 2) : n;
            if (before(a[i + 1], a[i])) swap(i + 1, i);
            sift_down(n, i + 1, false, stop);
            sift_down(n, i, true, stop);
        } while (i >= 2);
    }
    // a[n-1] holds the new hit (priority_deque::push)
    __device__ void push(const int n) const { if ((n - 1) & 1) leaf_upper(n, n - 1); else leaf_lower(n, n - 1); }
    // the deque becomes a[0..n-1)
    __device__ void pop_bottom(const int n) const { swap(0, n - 1); sift_down(n - 1, 0, true); }
    __device__ void pop_top(const int n) const { if (n <= 2) return; swap(1, n - 1); sift_down(n - 1, 1, false); }
    __device__ __forceinline__ int top(const int n) const { return n == 1 ? 0 : 1; }
};
typedef HitDequeT<RowHits> HitDeque;

} // namespace nvb
