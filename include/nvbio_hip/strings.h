// nvbio_hip/strings.h -- packed string sets as the hot path consumes them.
//
// The reference passes string sets as template types (ConcatenatedStringSet / SparseStringSet
// over PackedStream<const uint32*,uint8,BITS,BIG_ENDIAN_T>, nvbio/strings/string_set.h,
// nvbio/basic/packedstream.h).  The kernels behind the C-ABI only need the layout facts those
// types encode, so this view carries them as template parameters + plain device pointers.
#pragma once
#include "types.h"

namespace nvbio {

/// pack symbols into PackedStream words on the host (packedstream_inl.h:336-400)
template <uint32 BITS, bool BIG_ENDIAN_T>
inline std::vector<uint32> pack_symbols(const uint8* sym, uint64 n, uint32 pad_words = 4)
{
    const uint32 per = 32u / BITS, mask = (1u << BITS) - 1u;
    std::vector<uint32> w((n + per - 1) / per + pad_words, 0u);
    for (uint64 i = 0; i < n; ++i) {
        const uint32 k = uint32(i % per);
        const uint32 off = BIG_ENDIAN_T ? (32u - BITS - k * BITS) : (k * BITS);
        w[i / per] |= (uint32(sym[i]) & mask) << off;
    }
    return w;
}

/// A device-resident set of strings inside one packed stream:
/// string i = symbols [begin[i], begin[i] + length[i])  (length == NULL: fixed_length).
/// Plays the role of ConcatenatedStringSet<PackedStream<...>::iterator, const uint64*> /
/// SparseStringSet in the reference's batch calls.
template <uint32 BITS, bool BIG_ENDIAN_T>
struct PackedStringSetView
{
    static const uint32 SYMBOL_SIZE = BITS;
    static const bool   IS_BIG_ENDIAN = BIG_ENDIAN_T;

    uint32        m_size;
    const uint32* m_words;       // device
    uint64        m_n_words;
    const uint64* m_begin;       // device
    const uint32* m_length;      // device or NULL
    uint32        m_fixed_length;

    PackedStringSetView() : m_size(0), m_words(nullptr), m_n_words(0), m_begin(nullptr), m_length(nullptr), m_fixed_length(0) {}
    PackedStringSetView(uint32 size, const uint32* words, uint64 n_words, const uint64* begin, const uint32* length, uint32 fixed_length = 0)
        : m_size(size), m_words(words), m_n_words(n_words), m_begin(begin), m_length(length), m_fixed_length(fixed_length) {}

    uint32 size() const { return m_size; }

    nvbio_hip_string_set abi() const {
        nvbio_hip_string_set s;
        s.words = m_words; s.n_words = m_n_words; s.bits = BITS; s.big_endian = BIG_ENDIAN_T ? 1u : 0u;
        s.begin = m_begin; s.length = m_length; s.fixed_length = m_fixed_length; s._pad = 0;
        return s;
    }
};

/// owning device storage for a PackedStringSetView, built from host symbol strings
template <uint32 BITS, bool BIG_ENDIAN_T>
struct PackedStringSetDevice
{
    hip::device_vector<uint32> words;
    hip::device_vector<uint64> begin;
    hip::device_vector<uint32> length;
    uint32                     n;

    PackedStringSetDevice() : n(0) {}
    explicit PackedStringSetDevice(const std::vector<std::vector<uint8> >& strings) { assign(strings); }

    void assign(const std::vector<std::vector<uint8> >& strings) {
        n = uint32(strings.size());
        std::vector<uint64> b(n); std::vector<uint32> l(n); std::vector<uint8> cat;
        for (uint32 i = 0; i < n; ++i) { b[i] = cat.size(); l[i] = uint32(strings[i].size()); cat.insert(cat.end(), strings[i].begin(), strings[i].end()); }
        const std::vector<uint32> w = pack_symbols<BITS, BIG_ENDIAN_T>(cat.data(), cat.size());
        words.assign(w.data(), w.size()); begin.assign(b.data(), b.size()); length.assign(l.data(), l.size());
    }
    PackedStringSetView<BITS, BIG_ENDIAN_T> view() const {
        return PackedStringSetView<BITS, BIG_ENDIAN_T>(n, words.data(), words.size(), begin.data(), length.data());
    }
};

} // namespace nvbio
