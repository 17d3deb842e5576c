// compat/nvbio/io/utils.h -- the read views nvBowtie's alignment streams are built from (nvbio/io/utils.h:45-330):
//   ReadType / DirType          STANDARD | COMPLEMENT, FORWARD | REVERSE
//   ReadStream<Stream,Qual>     a read, or its reverse / complement / reverse-complement, as a string: symbol k is symbol
//                               (rev ? last - k : first + k) of the stream, complemented (c < 4 -> 3 - c) when comp; quality(k)
//                               walks the quality stream in the same order; qualities() is the matching quality string
//   ReadLoader<Batch,Tag>       loads read `range` of a sequence batch as a ReadStream (PackedStringLoader for the symbols)
//   SequenceStreamLoader        loads a range of the batch's symbol stream as a plain string
// nvBowtie stores reads reversed and asks for (REVERSE, STANDARD) to see the forward strand and (FORWARD, COMPLEMENT) to see
// the reverse complement (nvBowtie/bowtie2/cuda/alignment_utils.h:194-211).  The batch dispatcher (alignment/batched.h)
// recognises a ReadStream over packed words and runs such streams on the tuned kernels.
#pragma once
#include "../basic/types.h"
#include "../basic/packedstream.h"
#include "../basic/packedstream_loader.h"
#include "../basic/vector_view.h"

namespace nvbio {

// (the reference declares everything in this header in namespace nvbio, not nvbio::io: utils.h:37-347)

enum ReadType { STANDARD = 0u, COMPLEMENT = 1u };
enum DirType  { FORWARD  = 0u, REVERSE    = 1u };

template <typename IndexType> struct ReverseXform
{
    typedef IndexType index_type; typedef index_type argument_type; typedef index_type result_type;
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE ReverseXform() : pos(0) {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE ReverseXform(const index_type n) : pos(n - 1) {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE index_type operator()(const index_type i) const { return pos - i; }
    const index_type pos;
};
template <typename IndexType> struct OffsetXform
{
    typedef IndexType index_type; typedef index_type argument_type; typedef index_type result_type;
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE OffsetXform() : pos(0) {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE OffsetXform(const index_type n) : pos(n) {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE index_type operator()(const index_type i) const { return pos + i; }
    const index_type pos;
};

struct quality_nop {};

/// the qualities of a ReadStream as a string (refers to the read it was made from, which must outlive it)
template <typename ReadStreamType>
struct ReadStreamQualities
{
    static const uint32 SYMBOL_SIZE = 8u;
    typedef uint8   value_type;
    typedef uint8   reference;
    typedef uint32  index_type;
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE ReadStreamQualities() : m_read(NULL) {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE ReadStreamQualities(const ReadStreamType& read) : m_read(&read) {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint8  operator[](const uint32 pos) const { return m_read->quality(pos); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 length() const { return m_read->length(); }
    const ReadStreamType* m_read;
};

template <typename StreamType, typename QualType = quality_nop>
struct ReadStream
{
    static const uint32 SYMBOL_SIZE = stream_traits<StreamType>::SYMBOL_SIZE;
    typedef typename stream_traits<StreamType>::symbol_type value_type;
    typedef value_type                                      reference;
    typedef uint32                                          index_type;
    typedef ReadStream<StreamType, QualType>                this_type;
    typedef ReadStreamQualities<this_type>                  qual_string_type;

    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE ReadStream() : rev(0), comp(0), first(0), last(uint32(-1)) {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE ReadStream(const StreamType& s, const uint2 range)
        : rev(0), comp(0), first(range.x), last(range.y - 1), stream(s) {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE ReadStream(const StreamType& s, const QualType q, const uint2 range)
        : rev(0), comp(0), first(range.x), last(range.y - 1), stream(s), qual(q) {}

    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE void   set_flags(const DirType d, const ReadType t) { rev = (d == REVERSE); comp = (t == COMPLEMENT); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 length() const { return 1u + last - first; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE value_type operator[](const uint32 pos) const
    {
        const value_type c = stream[rev ? last - pos : first + pos];
        return comp ? (c < 4 ? value_type(3 - c) : c) : c;
    }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint8 quality(const uint32 pos) const { return qual[rev ? last - pos : first + pos]; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE qual_string_type qualities() const { return qual_string_type(*this); }

    uint32      rev, comp;          ///< walk backwards / complement the bases
    uint32      first, last;        ///< first and last symbol of the read in `stream`
    StreamType  stream;
    QualType    qual;
};
template <typename S, typename Q> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 length(const ReadStream<S, Q>& read) { return read.length(); }

/// loads one read of a sequence batch (any type with sequence_stream(), qual_stream(), SEQUENCE_BITS, SEQUENCE_BIG_ENDIAN and the
/// two storage iterator typedefs, i.e. io::SequenceDataAccess)
template <typename SequenceDataT, typename Tag>
struct ReadLoader
{
    typedef typename SequenceDataT::sequence_storage_iterator  read_storage;
    typedef typename SequenceDataT::qual_storage_iterator      qual_iterator;
    typedef PackedStringLoader<read_storage, SequenceDataT::SEQUENCE_BITS, SequenceDataT::SEQUENCE_BIG_ENDIAN, Tag>  loader_type;
    typedef typename loader_type::iterator                     read_iterator;
    typedef ReadStream<read_iterator, qual_iterator>           string_type;

    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE string_type load(const SequenceDataT& batch, const uint2 range, const DirType dir, const ReadType op)
    {
        const qual_iterator quals(batch.qual_stream() + range.x);
        string_type read(loader.load(batch.sequence_stream() + range.x, range.y - range.x), quals, make_uint2(0u, range.y - range.x));
        read.set_flags(dir, op);
        return read;
    }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE string_type load(const SequenceDataT& batch, const uint2 range, const DirType dir, const ReadType op, const uint2 subrange)
    {
        const qual_iterator quals(batch.qual_stream() + range.x);
        string_type read(loader.load(batch.sequence_stream() + range.x, range.y - range.x, subrange, dir == REVERSE ? 1u : 0u), quals, make_uint2(0u, range.y - range.x));
        read.set_flags(dir, op);
        return read;
    }
    loader_type loader;
};

template <typename SequenceDataT, typename Tag>
struct SequenceStreamLoader
{
    typedef typename SequenceDataT::sequence_storage_iterator  stream_storage;
    typedef PackedStringLoader<stream_storage, SequenceDataT::SEQUENCE_BITS, SequenceDataT::SEQUENCE_BIG_ENDIAN, Tag>  loader_type;
    typedef typename loader_type::iterator                     stream_iterator;
    typedef vector_view<stream_iterator>                       string_type;
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE string_type load(const SequenceDataT& batch, const uint2 range)
    { return string_type(range.y - range.x, loader.load(batch.sequence_stream() + range.x, range.y - range.x)); }
    loader_type loader;
};

template <typename S, typename Q> struct string_traits< ReadStream<S, Q> > { typedef typename ReadStream<S, Q>::value_type value_type; typedef uint32 index_type; };
template <typename R> struct string_traits< ReadStreamQualities<R> > { typedef uint8 value_type; typedef uint32 index_type; };

} // namespace nvbio
