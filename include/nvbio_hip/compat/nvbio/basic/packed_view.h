// compat/nvbio/basic/packed_view.h -- compile-time recognition of "a string that is a window of packed 32-bit words in memory":
// the test the drop-in layer uses to hand a caller's own string types to the tuned gfx950 kernels behind the C-ABI
// (include/nvbio_hip.h) without copying anything.  A vector_view over a PackedStream (nvbio/basic/packedstream.h) whose storage
// iterator is a word pointer -- plain, cuda::ldg_pointer (nvbio/basic/cuda/ldg.h) or const_cached_iterator
// (nvbio/basic/cached_iterator.h) around one -- is described by (address of its first word, symbol offset into it, length).
#pragma once
#include "packedstream.h"
#include "cached_iterator.h"
#include "vector_view.h"
#include "cuda/ldg.h"

namespace nvbio {
namespace priv {

template <typename It> struct word_pointer { static const bool ok = false; };
template <> struct word_pointer<const uint32*> { static const bool ok = true; NVBIO_FORCEINLINE NVBIO_HOST_DEVICE static const uint32* get(const uint32* p) { return p; } };
template <> struct word_pointer<uint32*>       { static const bool ok = true; NVBIO_FORCEINLINE NVBIO_HOST_DEVICE static const uint32* get(uint32* p) { return p; } };
template <> struct word_pointer< cuda::ldg_pointer<uint32> > { static const bool ok = true; NVBIO_FORCEINLINE NVBIO_HOST_DEVICE static const uint32* get(cuda::ldg_pointer<uint32> p) { return p.base; } };
template <typename It> struct word_pointer< const_cached_iterator<It> > { static const bool ok = word_pointer<It>::ok;
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE static const uint32* get(const_cached_iterator<It> p) { return word_pointer<It>::get(p.base()); } };

/// a PackedStream positioned somewhere in its words
template <typename S> struct packed_stream_where { static const bool ok = false; static const uint32 BITS = 0; static const bool BE = false; };
template <typename I, uint32 B, bool E, typename X>
struct packed_stream_where< PackedStream<I, uint8, B, E, X> >
{
    static const bool   ok   = word_pointer<I>::ok && (B == 2u || B == 4u);
    static const uint32 BITS = B;
    static const bool   BE   = E;
    /// first word address (in words from address 0) and symbol offset of the stream position from it
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE static void where(const PackedStream<I, uint8, B, E, X>& ps, uint64& word0, uint32& first)
    {
        const uint32 per = 32u / B;
        word0 = uint64(reinterpret_cast<uintptr_t>(word_pointer<I>::get(ps.stream()))) / 4u + uint64(ps.index()) / per;
        first = uint32(uint64(ps.index()) % per);
    }
};

template <typename S> struct packed_view { static const bool ok = false; static const uint32 BITS = 0; static const bool BE = false; };
template <typename I, uint32 B, bool E, typename X, typename VI>
struct packed_view< vector_view< PackedStream<I, uint8, B, E, X>, VI > >
{
    typedef packed_stream_where< PackedStream<I, uint8, B, E, X> > stream_where;
    static const bool   ok   = stream_where::ok;
    static const uint32 BITS = B;
    static const bool   BE   = E;
    typedef vector_view< PackedStream<I, uint8, B, E, X>, VI > view_type;
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE static void where(const view_type& v, uint64& word0, uint32& first) { stream_where::where(v.begin(), word0, first); }
};

/// ... and a string of BYTES in memory (vector_view<const uint8*>, what the reference's own tests align: nvbio-test/alignment_test.cu:242-335):
/// an 8-bit little-endian "packed" stream whose first word is the one holding the first byte.  The tuned kernels take such PATTERNS
/// (nvbio_hip.h: bits == 8); texts stay 2-bit.
template <typename It> struct byte_string_pointer { static const bool ok = false; };
template <> struct byte_string_pointer<const uint8*> { static const bool ok = true; NVBIO_FORCEINLINE NVBIO_HOST_DEVICE static const uint8* get(const uint8* p) { return p; } };
template <> struct byte_string_pointer<uint8*>       { static const bool ok = true; NVBIO_FORCEINLINE NVBIO_HOST_DEVICE static const uint8* get(uint8* p) { return p; } };
template <> struct byte_string_pointer< cuda::ldg_pointer<uint8> > { static const bool ok = true; NVBIO_FORCEINLINE NVBIO_HOST_DEVICE static const uint8* get(cuda::ldg_pointer<uint8> p) { return p.base; } };
template <typename I, typename VI>
struct packed_view< vector_view<I, VI> >
{
    static const bool   ok   = byte_string_pointer<I>::ok;
    static const uint32 BITS = 8u;
    static const bool   BE   = false;
    typedef vector_view<I, VI> view_type;
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE static void where(const view_type& v, uint64& word0, uint32& first)
    {
        const uintptr_t a = reinterpret_cast<uintptr_t>(byte_string_pointer<I>::get(v.begin()));
        word0 = uint64(a) / 4u; first = uint32(a & 3u);
    }
};

} // namespace priv
} // namespace nvbio
