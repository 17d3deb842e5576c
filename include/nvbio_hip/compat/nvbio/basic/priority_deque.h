// compat/nvbio/basic/priority_deque.h -- priority_deque<Type,Sequence,Compare> (nvbio/basic/priority_deque.h:120-421): a
// double-ended priority queue over a caller-supplied random-access container, the per-read seed-hit container of nvBowtie
// (priority_deque<SeedHit, vector_view<SeedHit*>, hit_compare>, nvBowtie/bowtie2/cuda/seed_hit_deque_array.h:163-164).
//
// An interval heap in array form: slots 2k / 2k+1 hold the lower / upper bound of node k; slot 0 is a minimum under Compare
// (bottom()), slot 1 -- slot 0 when alone -- a maximum (top()).  Which of several equivalent elements surfaces first depends on
// the exchanges made, and nvBowtie's selection stage samples the array slots directly, so the exchange sequence is the
// reference's (interval_heap.h:195-260, 356-533), element for element -- the same restatement the device kernels carry for
// SeedHit words (nvbio_amd/csrc/hit_deque.h), pinned by replaying operation programs recorded from the reference's compiled
// heap (tests/golden/hit_deque_vectors.npz).
#pragma once
#include "types.h"
#include <functional>
#include <vector>

namespace nvbio {

template <typename Type, typename Sequence = std::vector<Type>, typename Compare = std::less<Type> >
struct priority_deque
{
    typedef Sequence                         container_type;
    typedef Type                             value_type;
    typedef Compare                          value_compare;
    typedef uint32                           size_type;
    typedef const Type&                      const_reference;

    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE explicit priority_deque(const Compare& comp = Compare(), const Sequence& seq = Sequence()) : m_seq(seq), m_comp(comp) { heapify(); }
    /// over a container that already is an interval heap (constructed == true), or any content otherwise
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE priority_deque(const Sequence& seq, const bool constructed = false) : m_seq(seq) { if (!constructed) heapify(); }
    /// the tag form of the same: priority_deque(seq, priority_deque::CONSTRUCTED) never touches the container (priority_deque.h:173,194-196) --
    /// which is what lets it wrap a read-only view of a stored heap
    enum Constructed { CONSTRUCTED };
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE priority_deque(const Sequence& seq, const Constructed) : m_seq(seq) {}

    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool      empty() const { return m_seq.empty(); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE size_type size()  const { return size_type(m_seq.size()); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE void      clear() { m_seq.clear(); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE const Sequence& sequence() const { return m_seq; }

    /// a maximum / a minimum under Compare
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE const_reference top()    const { return m_seq[size() == 1u ? 0u : 1u]; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE const_reference bottom() const { return m_seq[0]; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE const_reference maximum() const { return top(); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE const_reference minimum() const { return bottom(); }

    NVBIO_HOST_DEVICE void push(const Type& v)
    {
        m_seq.push_back(v);
        const int n = int(size());
        if ((n - 1) & 1) leaf_upper(n, n - 1); else leaf_lower(n, n - 1);
    }
    NVBIO_HOST_DEVICE void pop() { pop_top(); }          // std::priority_queue's name for it (priority_deque.h:225)
    NVBIO_HOST_DEVICE void pop_top()
    {
        const int n = int(size());
        if (n > 2) { swap_slots(1, n - 1); sift_down(n - 1, 1, false); }
        m_seq.pop_back();
    }
    NVBIO_HOST_DEVICE void pop_bottom()
    {
        const int n = int(size());
        swap_slots(0, n - 1);
        sift_down(n - 1, 0, true);
        m_seq.pop_back();
    }

private:
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool before(const int i, const int j) { return m_comp(m_seq[i], m_seq[j]); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE void swap_slots(const int i, const int j) { const Type t = m_seq[i]; m_seq[i] = m_seq[j]; m_seq[j] = t; }
    /// the bottom-up construction (make_interval_heap, interval_heap.h:356-384): node pairs from the last one to the first, each put in order and
    /// sifted down inside its own subtree (`stop` = the slot below which the way back up ends).  nvBowtie runs it on every hits[read_id] its
    /// selection kernels take -- SeedHitDequeArrayDeviceView::get_deque() passes constructed = false (seed_hit_deque_array_inl.h:104-109) -- over hits
    /// whose ranges shrank since the last round, so the arrangement it leaves is part of the results.
    NVBIO_HOST_DEVICE void heapify()
    {
        const int n = int(size());
        if (n <= 1) return;
        const int end_parent = n / 2 - 1;
        int i = n ^ (n & 1);
        do
        {
            i -= 2;
            const int stop = (i <= end_parent) ? (i * 2 + 2) : n;
            if (before(i + 1, i)) swap_slots(i + 1, i);
            sift_down(n, i + 1, false, stop);
            sift_down(n, i, true, stop);
        }
        while (i >= 2);
    }
    NVBIO_HOST_DEVICE void sift_up(int i, const bool lower, const int stop = 2)
    {
        while (i >= stop)
        {
            const int parent = ((i / 2 - 1) | 1) ^ (lower ? 1 : 0);
            if (!(lower ? before(i, parent) : before(parent, i))) break;
            swap_slots(i, parent);
            i = parent;
        }
    }
    /// slot i, on the upper side of a leaf of an n-element heap, may be out of place
    NVBIO_HOST_DEVICE void leaf_upper(const int n, const int i, const int stop = 2)
    {
        const int co = (i * 2 < n) ? i * 2 : (i ^ 1);
        if (before(i, co)) { swap_slots(i, co); sift_up(co, true, stop); }
        else sift_up(i, false, stop);
    }
    NVBIO_HOST_DEVICE void leaf_lower(const int n, const int i, const int stop = 2)
    {
        int co = i | 1;
        if (co >= n) { if (co == 1) return; co = (co / 2 - 1) | 1; }
        if (before(co, i)) { swap_slots(i, co); sift_up(co, false, stop); }
        else sift_up(i, true, stop);
    }
    NVBIO_HOST_DEVICE void sift_down(const int n, int i, const bool lower, const int stop = 2)
    {
        const int end_parent = n / 2 - ((lower && (n & 3) == 0) ? 2 : 1);
        while (i < end_parent)
        {
            int child = i * 2 + (lower ? 2 : 1);
            if (lower ? before(child + 2, child) : before(child, child + 2)) child += 2;
            swap_slots(i, child);
            i = child;
        }
        if (i <= end_parent + (lower ? 0 : 1))
        {
            int child = i * 2 + (lower ? 2 : 1);
            if (child < n)
            {
                if (!lower && child + 1 < n && before(child, child + 1))
                {
                    ++child;
                    swap_slots(i, child);
                    leaf_lower(n, child, stop);
                    return;
                }
                swap_slots(i, child);
                i = child;
            }
        }
        if (lower) leaf_lower(n, i, stop); else leaf_upper(n, i, stop);
    }

    Sequence m_seq;
    Compare  m_comp;
};

} // namespace nvbio
