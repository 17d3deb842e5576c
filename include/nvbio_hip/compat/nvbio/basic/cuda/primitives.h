// compat/nvbio/basic/cuda/primitives.h -- the device-only spellings of the parallel primitives (nvbio/basic/cuda/primitives.h:74-330):
// nvbio::cuda::reduce / inclusive_scan / exclusive_scan / copy_flagged / copy_if / runlength_encode / reduce_by_key ... taking a
// thrust::device_vector<uint8> of temporary storage.  examples/fmmap/fmmap.cu:376-385,522-526 calls reduce_by_key and reduce this way.
// They are the device_tag instances of the system-tagged functions in ../primitives.h (rocThrust underneath; the temporary-storage
// vector is accepted and left alone).
#pragma once
#include "../primitives.h"
#if defined(__HIPCC__)

namespace nvbio {
namespace cuda {

namespace priv { inline nvbio::vector<device_tag, uint8>& no_temp() { static thread_local nvbio::vector<device_tag, uint8> v; return v; } }

template <typename PredicateIterator> inline bool any(const uint32 n, const PredicateIterator pred) { return nvbio::any<device_tag>(n, pred); }
template <typename PredicateIterator> inline bool all(const uint32 n, const PredicateIterator pred) { return nvbio::all<device_tag>(n, pred); }
template <typename Iterator> inline bool is_sorted(const uint32 n, const Iterator values) { return nvbio::is_sorted<device_tag>(n, values); }
template <typename Iterator, typename Headflags>
inline bool is_segment_sorted(const uint32 n, const Iterator values, const Headflags flags) { return nvbio::is_segment_sorted<device_tag>(n, values, flags); }

template <typename InputIterator, typename BinaryOp>
inline typename std::iterator_traits<InputIterator>::value_type reduce(const uint32 n, InputIterator in, BinaryOp op, thrust::device_vector<uint8>&)
{ return nvbio::reduce<device_tag>(n, in, op, priv::no_temp()); }
template <typename InputIterator, typename OutputIterator, typename BinaryOp>
inline void inclusive_scan(const uint32 n, InputIterator in, OutputIterator out, BinaryOp op, thrust::device_vector<uint8>&)
{ nvbio::inclusive_scan<device_tag>(n, in, out, op, priv::no_temp()); }
template <typename InputIterator, typename OutputIterator, typename BinaryOp, typename Identity>
inline void exclusive_scan(const uint32 n, InputIterator in, OutputIterator out, BinaryOp op, Identity identity, thrust::device_vector<uint8>&)
{ nvbio::exclusive_scan<device_tag>(n, in, out, op, identity, priv::no_temp()); }
template <typename InputIterator, typename FlagsIterator, typename OutputIterator>
inline uint32 copy_flagged(const uint32 n, InputIterator in, FlagsIterator flags, OutputIterator out, thrust::device_vector<uint8>&)
{ return nvbio::copy_flagged<device_tag>(n, in, flags, out, priv::no_temp()); }
template <typename InputIterator, typename OutputIterator, typename Predicate>
inline uint32 copy_if(const uint32 n, InputIterator in, OutputIterator out, const Predicate pred, thrust::device_vector<uint8>&)
{ return nvbio::copy_if<device_tag>(n, in, out, pred, priv::no_temp()); }
template <typename InputIterator, typename OutputIterator, typename CountIterator>
inline uint32 runlength_encode(const uint32 n, InputIterator in, OutputIterator out, CountIterator counts, thrust::device_vector<uint8>&)
{ return nvbio::runlength_encode<device_tag>(n, in, out, counts, priv::no_temp()); }
template <typename KeyIterator, typename ValueIterator, typename OutputKeyIterator, typename OutputValueIterator, typename ReductionOp>
inline uint32 reduce_by_key(const uint32 n, KeyIterator keys_in, ValueIterator values_in, OutputKeyIterator keys_out, OutputValueIterator values_out,
                            ReductionOp reduction_op, thrust::device_vector<uint8>&)
{ return nvbio::reduce_by_key<device_tag>(n, keys_in, values_in, keys_out, values_out, reduction_op, priv::no_temp()); }

} // namespace cuda
} // namespace nvbio
#endif
