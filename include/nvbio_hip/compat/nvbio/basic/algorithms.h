// compat/nvbio/basic/algorithms.h -- searches and merges over sorted ranges, callable from kernels (nvbio/basic/algorithms.h:54-322):
// find_pivot, lower_bound / upper_bound (+ _index forms), merge, merge_by_key.  Argument order as in the reference: (value, begin, n).
#pragma once
#include "types.h"

namespace nvbio {

/// first element of [begin, begin + n) for which predicate holds, in a range where the predicate is false then true
template <typename Iterator, typename Predicate>
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE Iterator find_pivot(Iterator begin, const uint32 n, const Predicate predicate)
{
    uint32 lo = 0, hi = n;
    while (lo < hi) { const uint32 mid = lo + (hi - lo) / 2u; if (predicate(begin[mid])) hi = mid; else lo = mid + 1u; }
    return begin + lo;
}
/// first element not less than x
template <typename Iterator, typename Value, typename index_type>
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE Iterator lower_bound(const Value x, Iterator begin, const index_type n)
{
    index_type lo = 0, hi = n;
    while (lo < hi) { const index_type mid = lo + (hi - lo) / 2; if (begin[mid] < x) lo = mid + 1; else hi = mid; }
    return begin + lo;
}
/// first element greater than x
template <typename Iterator, typename Value, typename index_type>
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE Iterator upper_bound(const Value x, Iterator begin, const index_type n)
{
    index_type lo = 0, hi = n;
    while (lo < hi) { const index_type mid = lo + (hi - lo) / 2; if (x < begin[mid]) hi = mid; else lo = mid + 1; }
    return begin + lo;
}
template <typename Iterator, typename Value, typename index_type>
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE index_type lower_bound_index(const Value x, Iterator begin, const index_type n) { return index_type(lower_bound(x, begin, n) - begin); }
template <typename Iterator, typename Value, typename index_type>
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE index_type upper_bound_index(const Value x, Iterator begin, const index_type n) { return index_type(upper_bound(x, begin, n) - begin); }

/// stable two-way merge of sorted ranges (ties take the first range)
template <typename In1, typename In2, typename Out>
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE void merge(In1 first1, In1 end1, In2 first2, In2 end2, Out output)
{
    for (;;)
    {
        const bool left = first1 != end1, right = first2 != end2;
        if (!left && !right) return;
        if (right && (!left || *first2 < *first1)) { *output = *first2; ++first2; }
        else                                       { *output = *first1; ++first1; }
        ++output;
    }
}
template <typename Key1, typename Key2, typename Val1, typename Val2, typename KeyOut, typename ValOut>
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE void merge_by_key(Key1 first1, Key1 end1, Key2 first2, Key2 end2, Val1 values1, Val2 values2, KeyOut output_keys, ValOut output_values)
{
    for (;;)
    {
        const bool left = first1 != end1, right = first2 != end2;
        if (!left && !right) return;
        if (right && (!left || *first2 < *first1)) { *output_keys = *first2; *output_values = *values2; ++first2; ++values2; }
        else                                       { *output_keys = *first1; *output_values = *values1; ++first1; ++values1; }
        ++output_keys; ++output_values;
    }
}

} // namespace nvbio
