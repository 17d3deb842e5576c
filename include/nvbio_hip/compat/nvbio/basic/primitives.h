// compat/nvbio/basic/primitives.h -- the parallel primitives the reference's applications chain between their kernels
// (nvbio/basic/primitives.h:60-448), selected by a system tag: any / all / is_sorted / is_segment_sorted, for_each, transform, reduce,
// scans, copy_flagged / copy_if, runlength_encode, reduce_by_key, vectorized lower / upper bound, radix_sort, merge_by_key.  Executed by
// rocThrust on the device (thrust::device policy) and on the host (thrust::host); the temp_storage vectors of the reference's signatures
// are accepted and left to thrust's own allocator.
#pragma once
#include "types.h"
#include "numbers.h"
#include "console.h"
#include "vector.h"
#include "algorithms.h"
#include "cuda/arch.h"
#include "cuda/timer.h"
#include "cuda/sort.h"
#if defined(__HIPCC__)
#include <thrust/execution_policy.h>
#include <thrust/reduce.h>
#include <thrust/scan.h>
#include <thrust/copy.h>
#include <thrust/sort.h>
#include <thrust/binary_search.h>
#include <thrust/merge.h>
#include <thrust/for_each.h>
#include <thrust/transform.h>
#include <thrust/logical.h>
#include <thrust/functional.h>
#include <thrust/iterator/constant_iterator.h>
#include <thrust/iterator/counting_iterator.h>
#include <thrust/iterator/zip_iterator.h>
#include <thrust/iterator/transform_iterator.h>
#include <thrust/tuple.h>

namespace nvbio {

namespace priv {
template <typename system_tag> struct thrust_policy {};
template <> struct thrust_policy<device_tag> { static decltype(thrust::device) get() { return thrust::device; } };
template <> struct thrust_policy<host_tag>   { static decltype(thrust::host)   get() { return thrust::host; } };
struct identity_bool { template <typename T> NVBIO_HOST_DEVICE bool operator()(const T v) const { return v ? true : false; } };
/// (values[i-1], values[i], flags[i-1], flags[i]) -> out of order inside a segment?
struct segment_disorder
{
    template <typename Tuple> NVBIO_HOST_DEVICE bool operator()(const Tuple t) const
    { return thrust::get<2>(t) == thrust::get<3>(t) && thrust::get<1>(t) < thrust::get<0>(t); }
};
struct pair_disorder { template <typename Tuple> NVBIO_HOST_DEVICE bool operator()(const Tuple t) const { return thrust::get<1>(t) < thrust::get<0>(t); } };
} // namespace priv

template <typename system_tag, typename PredicateIterator>
inline bool any(const uint32 n, const PredicateIterator pred) { return thrust::any_of(priv::thrust_policy<system_tag>::get(), pred, pred + n, priv::identity_bool()); }
template <typename system_tag, typename PredicateIterator>
inline bool all(const uint32 n, const PredicateIterator pred) { return thrust::all_of(priv::thrust_policy<system_tag>::get(), pred, pred + n, priv::identity_bool()); }

template <typename system_tag, typename Iterator>
inline bool is_sorted(const uint32 n, const Iterator values)
{
    if (n < 2u) return true;
    return !thrust::any_of(priv::thrust_policy<system_tag>::get(), thrust::make_zip_iterator(thrust::make_tuple(values, values + 1)),
                           thrust::make_zip_iterator(thrust::make_tuple(values + (n - 1u), values + n)), priv::pair_disorder());
}
/// sorted inside every run of equal head flags
template <typename system_tag, typename Iterator, typename Headflags>
inline bool is_segment_sorted(const uint32 n, const Iterator values, const Headflags flags)
{
    if (n < 2u) return true;
    return !thrust::any_of(priv::thrust_policy<system_tag>::get(), thrust::make_zip_iterator(thrust::make_tuple(values, values + 1, flags, flags + 1)),
                           thrust::make_zip_iterator(thrust::make_tuple(values + (n - 1u), values + n, flags + (n - 1u), flags + n)), priv::segment_disorder());
}

template <typename system_tag, typename Iterator, typename Functor>
inline void for_each(const uint64 n, const Iterator in, Functor functor) { thrust::for_each(priv::thrust_policy<system_tag>::get(), in, in + n, functor); }

template <typename system_tag, typename Iterator, typename Output, typename Functor>
inline void transform(const uint32 n, const Iterator in, const Output out, const Functor functor)
{ thrust::transform(priv::thrust_policy<system_tag>::get(), in, in + n, out, functor); }
template <typename system_tag, typename Iterator1, typename Iterator2, typename Output, typename Functor>
inline void transform(const uint32 n, const Iterator1 in1, const Iterator2 in2, const Output out, const Functor functor)
{ thrust::transform(priv::thrust_policy<system_tag>::get(), in1, in1 + n, in2, out, functor); }

/// fold of a non-empty range (the first element seeds it)
template <typename system_tag, typename InputIterator, typename BinaryOp>
inline typename std::iterator_traits<InputIterator>::value_type reduce(const uint32 n, InputIterator in, BinaryOp op, nvbio::vector<system_tag, uint8>&)
{
    typedef typename std::iterator_traits<InputIterator>::value_type T;
    if (n == 0u) return T();
    const T first = *in;
    return thrust::reduce(priv::thrust_policy<system_tag>::get(), in + 1, in + n, first, op);
}
template <typename system_tag, typename InputIterator, typename OutputIterator, typename BinaryOp>
inline void inclusive_scan(const uint32 n, InputIterator in, OutputIterator out, BinaryOp op, nvbio::vector<system_tag, uint8>&)
{ thrust::inclusive_scan(priv::thrust_policy<system_tag>::get(), in, in + n, out, op); }
template <typename system_tag, typename InputIterator, typename OutputIterator, typename BinaryOp, typename Identity>
inline void exclusive_scan(const uint32 n, InputIterator in, OutputIterator out, BinaryOp op, Identity identity, nvbio::vector<system_tag, uint8>&)
{ thrust::exclusive_scan(priv::thrust_policy<system_tag>::get(), in, in + n, out, identity, op); }

/// stable compaction by a stencil of flags / by a predicate; returns the number kept
template <typename system_tag, typename InputIterator, typename FlagsIterator, typename OutputIterator>
inline uint32 copy_flagged(const uint32 n, InputIterator in, FlagsIterator flags, OutputIterator out, nvbio::vector<system_tag, uint8>&)
{ return uint32(thrust::copy_if(priv::thrust_policy<system_tag>::get(), in, in + n, flags, out, priv::identity_bool()) - out); }
template <typename system_tag, typename InputIterator, typename OutputIterator, typename Predicate>
inline uint32 copy_if(const uint32 n, InputIterator in, OutputIterator out, const Predicate pred, nvbio::vector<system_tag, uint8>&)
{ return uint32(thrust::copy_if(priv::thrust_policy<system_tag>::get(), in, in + n, out, pred) - out); }

template <typename system_tag, typename InputIterator, typename OutputIterator, typename CountIterator>
inline uint32 runlength_encode(const uint32 n, InputIterator in, OutputIterator out, CountIterator counts, nvbio::vector<system_tag, uint8>&)
{
    typedef typename std::iterator_traits<CountIterator>::value_type count_type;
    return uint32(thrust::reduce_by_key(priv::thrust_policy<system_tag>::get(), in, in + n, thrust::make_constant_iterator(count_type(1)), out, counts).first - out);
}
template <typename system_tag, typename KeyIterator, typename ValueIterator, typename OutputKeyIterator, typename OutputValueIterator, typename ReductionOp>
inline uint32 reduce_by_key(const uint32 n, KeyIterator keys_in, ValueIterator values_in, OutputKeyIterator keys_out, OutputValueIterator values_out,
                            ReductionOp reduction_op, nvbio::vector<system_tag, uint8>&)
{
    typedef typename std::iterator_traits<KeyIterator>::value_type key_type;
    return uint32(thrust::reduce_by_key(priv::thrust_policy<system_tag>::get(), keys_in, keys_in + n, values_in, keys_out, values_out, thrust::equal_to<key_type>(), reduction_op).first - keys_out);
}

/// for each of n values, its lower / upper bound among n_keys sorted keys
template <typename system_tag, typename KeyIterator, typename ValueIterator, typename OutputIterator>
inline void lower_bound(const uint32 n, ValueIterator values, const uint32 n_keys, KeyIterator keys, OutputIterator indices)
{ thrust::lower_bound(priv::thrust_policy<system_tag>::get(), keys, keys + n_keys, values, values + n, indices); }
template <typename system_tag, typename KeyIterator, typename ValueIterator, typename OutputIterator>
inline void upper_bound(const uint32 n, ValueIterator values, const uint32 n_keys, KeyIterator keys, OutputIterator indices)
{ thrust::upper_bound(priv::thrust_policy<system_tag>::get(), keys, keys + n_keys, values, values + n, indices); }

template <typename system_tag, typename KeyIterator>
inline void radix_sort(const uint32 n, KeyIterator keys, nvbio::vector<system_tag, uint8>&) { thrust::sort(priv::thrust_policy<system_tag>::get(), keys, keys + n); }
template <typename system_tag, typename KeyIterator, typename ValueIterator>
inline void radix_sort(const uint32 n, KeyIterator keys, ValueIterator values, nvbio::vector<system_tag, uint8>&)
{ thrust::stable_sort_by_key(priv::thrust_policy<system_tag>::get(), keys, keys + n, values); }

template <typename system_tag, typename K1, typename K2, typename V1, typename V2, typename KO, typename VO>
inline void merge_by_key(const uint32 A_len, const uint32 B_len, const K1 A_keys, const K2 B_keys, const V1 A_values, const V2 B_values, KO C_keys, VO C_values,
                         nvbio::vector<system_tag, uint8>&)
{ thrust::merge_by_key(priv::thrust_policy<system_tag>::get(), A_keys, A_keys + A_len, B_keys, B_keys + B_len, A_values, B_values, C_keys, C_values); }

/// a callable for_each; the device flavour of the reference tunes its grid between calls, this one leaves the launch to thrust
template <typename system_tag>
struct for_each_enactor
{
    template <typename Iterator, typename Functor> void operator()(const uint64 n, const Iterator in, Functor functor) { for_each<system_tag>(n, in, functor); }
    template <typename Functor> void operator()(const uint64 n, Functor functor) { for_each<system_tag>(n, thrust::make_counting_iterator<uint64>(0), functor); }
};

} // namespace nvbio
#endif
