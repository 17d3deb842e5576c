// compat/nvbio/basic/packedstream_loader.h -- PackedStringLoader<StorageIterator,BITS,BIG_ENDIAN,Tag>
// (nvbio/basic/packedstream_loader.h:60-140).  The reference's lmem_cache_tag copies a string's words into per-thread
// local memory before the DP reads them symbol by symbol; local memory is scratch (HBM) on gfx950, and the tuned kernels
// behind BatchedAlignmentScore stage strings in LDS themselves, so both tags here hand back a view of the stream in place.
// `iterator` therefore equals `input_iterator`, which is also what lets the batch dispatcher recognise such a string as
// "packed words in HBM" and route the job to the tuned kernels.
#pragma once
#include "packedstream.h"
#include "cached_iterator.h"

namespace nvbio {

template <uint32 CACHE_SIZE> struct lmem_cache_tag {};
struct uncached_tag {};

template <typename StorageIterator, uint32 SYMBOL_SIZE_T, bool BIG_ENDIAN_T, typename Tag = uncached_tag>
struct PackedStringLoader
{
    typedef typename std::iterator_traits<StorageIterator>::value_type       storage_type;
    typedef PackedStream<StorageIterator, uint8, SYMBOL_SIZE_T, BIG_ENDIAN_T>  input_stream;
    typedef input_stream                                                      input_iterator;
    typedef input_stream                                                      iterator;

    /// the whole string
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE iterator load(const input_stream stream, const uint32 length) { (void)length; return stream; }
    /// a substring window of it (the staged schedulers' form): the view is the same, only the range that will be read differs
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE iterator load(const input_stream stream, const uint32 length, const uint2 substring_range, const uint32 rev_flag)
    { (void)length; (void)substring_range; (void)rev_flag; return stream; }
};

} // namespace nvbio
