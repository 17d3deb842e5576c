// compat/nvbio/io/sequence/sequence.h -- the read / reference containers either side of the hot path (nvbio/io/sequence/sequence.h,
// sequence_access.h, sequence_encoder.h): SequenceDataInfo, the storage-less views, SequenceDataHost / Device,
// SequenceDataAccess<ALPHABET>, the batch encoder, and FASTQ / FASTA input streams behind open_sequence_file() -- what
// sw-benchmark.cu:513-575 and nvBowtie.cpp:573-600 drive.  Layout as the reference's: symbols packed big-endian at the alphabet's width
// into one uint32 stream with NO padding between sequences, `sequence_index` = n + 1 symbol offsets, one phred byte per symbol, names as
// NUL-terminated strings with their own n + 1 byte offsets.  The device flavour needs hipcc (rocThrust vectors); everything else is host C++.
#pragma once
#include "sequence_traits.h"
#include "../../basic/packedstream.h"
#include "../../basic/vector_view.h"
#include "../../basic/vector.h"
#include "../../basic/cuda/ldg.h"
#include "../../strings/string_set.h"
#include "../../fasta/fasta.h"
#include "../../basic/console.h"
#include "../../basic/omp.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <typeinfo>
#include <vector>

namespace nvbio {
namespace io {

enum QualityEncoding  { Phred = 0, Phred33 = 1, Phred64 = 2, Solexa = 3 };
enum SequenceEncoding { FORWARD = 0x0001, REVERSE = 0x0002, FORWARD_COMPLEMENT = 0x0004, REVERSE_COMPLEMENT = 0x0008 };
enum SequenceFlags    { SEQUENCE_DATA = 0x0001, SEQUENCE_QUALS = 0x0002, SEQUENCE_NAMES = 0x0004 };

/// the counts every flavour of sequence data carries (sequence.h:201-242)
struct SequenceDataInfo
{
    NVBIO_HOST_DEVICE SequenceDataInfo() : m_alphabet(PROTEIN), m_n_seqs(0), m_name_stream_len(0), m_sequence_stream_len(0), m_sequence_stream_words(0),
        m_has_qualities(0), m_min_sequence_len(uint32(-1)), m_max_sequence_len(0), m_avg_sequence_len(0) {}

    NVBIO_HOST_DEVICE Alphabet alphabet()         const { return m_alphabet; }
    NVBIO_HOST_DEVICE uint32   size()             const { return m_n_seqs; }
    NVBIO_HOST_DEVICE uint32   bps()              const { return m_sequence_stream_len; }
    NVBIO_HOST_DEVICE uint32   words()            const { return m_sequence_stream_words; }
    NVBIO_HOST_DEVICE uint32   qs()               const { return m_has_qualities ? m_sequence_stream_len : 0u; }
    NVBIO_HOST_DEVICE uint32   name_stream_len()  const { return m_name_stream_len; }
    NVBIO_HOST_DEVICE bool     has_qualities()    const { return m_has_qualities != 0; }
    NVBIO_HOST_DEVICE uint32   max_sequence_len() const { return m_max_sequence_len; }
    NVBIO_HOST_DEVICE uint32   min_sequence_len() const { return m_min_sequence_len; }
    NVBIO_HOST_DEVICE uint32   avg_sequence_len() const { return m_avg_sequence_len; }

    Alphabet m_alphabet;
    uint32   m_n_seqs, m_name_stream_len, m_sequence_stream_len, m_sequence_stream_words, m_has_qualities;
    uint32   m_min_sequence_len, m_max_sequence_len, m_avg_sequence_len;
};
NVBIO_HOST_DEVICE inline bool operator==(const SequenceDataInfo& a, const SequenceDataInfo& b)
{
    return a.m_alphabet == b.m_alphabet && a.m_n_seqs == b.m_n_seqs && a.m_name_stream_len == b.m_name_stream_len && a.m_sequence_stream_len == b.m_sequence_stream_len &&
           a.m_sequence_stream_words == b.m_sequence_stream_words && a.m_has_qualities == b.m_has_qualities && a.m_min_sequence_len == b.m_min_sequence_len &&
           a.m_max_sequence_len == b.m_max_sequence_len && a.m_avg_sequence_len == b.m_avg_sequence_len;
}
NVBIO_HOST_DEVICE inline bool operator!=(const SequenceDataInfo& a, const SequenceDataInfo& b) { return !(a == b); }

namespace priv {
template <typename It> struct const_iterator_of             { typedef It type; };
template <typename T>  struct const_iterator_of<T*>         { typedef const T* type; };
} // namespace priv

/// a storage-less view: the counts plus five iterators (sequence.h:296-400)
template <typename IndexIterator = uint32*, typename SequenceStorageIterator = uint32*, typename QualStorageIterator = char*, typename NameStorageIterator = char*>
struct SequenceDataViewCore : public SequenceDataInfo
{
    typedef IndexIterator            index_iterator;
    typedef SequenceStorageIterator  sequence_storage_iterator;
    typedef QualStorageIterator      qual_storage_iterator;
    typedef NameStorageIterator      name_storage_iterator;
    typedef typename priv::const_iterator_of<IndexIterator>::type            const_index_iterator;
    typedef typename priv::const_iterator_of<SequenceStorageIterator>::type  const_sequence_storage_iterator;
    typedef typename priv::const_iterator_of<QualStorageIterator>::type      const_qual_storage_iterator;
    typedef typename priv::const_iterator_of<NameStorageIterator>::type      const_name_storage_iterator;

    NVBIO_HOST_DEVICE SequenceDataViewCore() : m_name_stream(), m_name_index(), m_sequence_stream(), m_sequence_index(), m_qual_stream() {}
    NVBIO_HOST_DEVICE SequenceDataViewCore(const SequenceDataInfo& info, const SequenceStorageIterator sequence_stream, const IndexIterator sequence_index,
                                           const QualStorageIterator qual_stream, const NameStorageIterator name_stream, const IndexIterator name_index)
        : SequenceDataInfo(info), m_name_stream(name_stream), m_name_index(name_index), m_sequence_stream(sequence_stream), m_sequence_index(sequence_index), m_qual_stream(qual_stream) {}
    template <typename I, typename S, typename Q, typename N>
    NVBIO_HOST_DEVICE SequenceDataViewCore(const SequenceDataViewCore<I, S, Q, N>& in)
        : SequenceDataInfo(in), m_name_stream(NameStorageIterator(in.m_name_stream)), m_name_index(IndexIterator(in.m_name_index)),
          m_sequence_stream(SequenceStorageIterator(in.m_sequence_stream)), m_sequence_index(IndexIterator(in.m_sequence_index)), m_qual_stream(QualStorageIterator(in.m_qual_stream)) {}

    NVBIO_HOST_DEVICE index_iterator                  name_index()             { return m_name_index; }
    NVBIO_HOST_DEVICE index_iterator                  sequence_index()         { return m_sequence_index; }
    NVBIO_HOST_DEVICE name_storage_iterator           name_stream()            { return m_name_stream; }
    NVBIO_HOST_DEVICE sequence_storage_iterator       sequence_storage()       { return m_sequence_stream; }
    NVBIO_HOST_DEVICE qual_storage_iterator           qual_stream()            { return m_qual_stream; }
    NVBIO_HOST_DEVICE const_index_iterator            name_index()       const { return m_name_index; }
    NVBIO_HOST_DEVICE const_index_iterator            sequence_index()   const { return m_sequence_index; }
    NVBIO_HOST_DEVICE const_name_storage_iterator     name_stream()      const { return m_name_stream; }
    NVBIO_HOST_DEVICE const_sequence_storage_iterator sequence_storage() const { return m_sequence_stream; }
    NVBIO_HOST_DEVICE const_qual_storage_iterator     qual_stream()      const { return m_qual_stream; }
    NVBIO_HOST_DEVICE uint2 get_range(const uint32 i) const { return make_uint2(sequence_index()[i], sequence_index()[i + 1]); }

    name_storage_iterator     m_name_stream;
    index_iterator            m_name_index;
    sequence_storage_iterator m_sequence_stream;
    index_iterator            m_sequence_index;
    qual_storage_iterator     m_qual_stream;
};
typedef SequenceDataViewCore<uint32*, uint32*, char*, char*>                         SequenceDataView;
typedef SequenceDataViewCore<const uint32*, const uint32*, const char*, const char*> ConstSequenceDataView;
typedef SequenceDataViewCore<cuda::ldg_pointer<uint32>, cuda::ldg_pointer<uint32>, const char*, const char*> LdgSequenceDataView;

/// the polymorphic base of the containers: whatever holds the storage can be looked at through a plain view (sequence.h:414-430)
struct SequenceData : public SequenceDataInfo
{
    typedef SequenceDataView       plain_view_type;
    typedef ConstSequenceDataView  const_plain_view_type;
    virtual ~SequenceData() {}
    virtual operator plain_view_type()             { return plain_view_type(); }
    virtual operator const_plain_view_type() const { return const_plain_view_type(); }
};

namespace priv {
#if defined(__HIPCC__)
template <typename In, typename Out> inline void seq_copy(In in, const uint32 n, Out out) { thrust::copy(in, in + n, out); }
template <typename system_tag, typename T> struct seq_vector { typedef nvbio::vector<system_tag, T> type; };
template <typename V> inline typename V::value_type*       seq_ptr(V& v)       { return nvbio::raw_pointer(static_cast<typename V::base_type&>(v)); }
template <typename V> inline const typename V::value_type* seq_ptr(const V& v) { return nvbio::raw_pointer(static_cast<const typename V::base_type&>(v)); }
/// A 2-bit DNA stream in device memory is a reference genome, and its users address it with 32-bit coordinates they do not always
/// keep inside it: nvBowtie's locate stage stores `SA position - offset of the seed in the read` (locate_inl.h:142,204), which wraps
/// below zero for a read with an insertion lying over the first bases of the genome, and the scoring streams then load a window at
/// that coordinate (score_best_inl.h:111-115, alignment_utils.h:216-218) -- a read ~1 GiB past the stream.  On the hardware the
/// reference was written for such a read lands in some other mapped allocation and the garbage scores below the threshold; here it
/// is a memory fault that ends the process.  With 288 GB of HBM the robust answer is cheap: reserve the stream's allocation so that
/// EVERY 32-bit coordinate (2^32 symbols = 2^28 words = 1 GiB, plus a window of slack) is addressable, and zero the tail, so a wrapped
/// window reads as poly-A and fails the score test deterministically.  NVBIO_HIP_COMPAT_NO_COORD_COVER=1 switches it off.
template <typename system_tag> struct coordinate_cover { template <typename V> static void apply(V&, const uint32) {} };
template <> struct coordinate_cover<device_tag>
{
    template <typename V> static void apply(V& v, const uint32 alphabet)
    {
        if (alphabet != uint32(DNA) || v.empty()) return;
        const char* off = getenv("NVBIO_HIP_COMPAT_NO_COORD_COVER");
        if (off && off[0] == '1') return;
        const size_t all = (size_t(1) << 28) + 1024u;
        if (v.capacity() >= all) return;
        const size_t n = v.size();
        // (a device too full for the extra GiB keeps the stream as it is: the application then runs as it would on the reference's hardware)
        try { v.reserve(all); } catch (...) { (void)hipGetLastError(); log_warning(stderr, "reference stream: no room to cover the 32-bit coordinate space (%zu MB)\n", all * sizeof(uint32) >> 20); return; }
        (void)hipMemset(nvbio::raw_pointer(static_cast<typename V::base_type&>(v)) + n, 0, (all - n) * sizeof(uint32));
    }
};
#else
template <typename system_tag> struct coordinate_cover { template <typename V> static void apply(V&, const uint32) {} };
template <typename In, typename Out> inline void seq_copy(In in, const uint32 n, Out out) { std::copy(in, in + n, out); }
template <typename system_tag, typename T> struct seq_vector {};                               // the device flavour needs hipcc
template <typename T> struct seq_vector<host_tag, T> { typedef std::vector<T> type; };
template <typename T> inline T*       seq_ptr(std::vector<T>& v)       { return v.empty() ? (T*)0 : v.data(); }
template <typename T> inline const T* seq_ptr(const std::vector<T>& v) { return v.empty() ? (const T*)0 : v.data(); }
#endif
} // namespace priv

/// sequence data owning its storage in host or device memory (sequence.h:436-560)
template <typename system_tag>
struct SequenceDataStorage : public SequenceData
{
    typedef SequenceDataView       plain_view_type;
    typedef ConstSequenceDataView  const_plain_view_type;
    typedef typename priv::seq_vector<system_tag, uint32>::type  word_vector;
    typedef typename priv::seq_vector<system_tag, char>::type    char_vector;
    typedef typename word_vector::iterator        index_iterator;
    typedef typename word_vector::iterator        sequence_storage_iterator;
    typedef typename char_vector::iterator        qual_storage_iterator;
    typedef typename char_vector::iterator        name_storage_iterator;
    typedef typename word_vector::const_iterator  const_index_iterator;
    typedef typename word_vector::const_iterator  const_sequence_storage_iterator;
    typedef typename char_vector::const_iterator  const_qual_storage_iterator;
    typedef typename char_vector::const_iterator  const_name_storage_iterator;

    SequenceDataStorage() {}
    template <typename other_tag> SequenceDataStorage(const SequenceDataStorage<other_tag>& other) { *this = other; }
    SequenceDataStorage(const SequenceDataStorage& other) : SequenceData() { *this = other; }
    /// from any container behind the base class: host-side data (loaded or mapped) is seen through its plain view
    SequenceDataStorage(const SequenceData& other)
    {
        if (const SequenceDataStorage* same = dynamic_cast<const SequenceDataStorage*>(&other)) *this = *same;
        else *this = static_cast<const_plain_view_type>(other);
    }
    template <typename I, typename S, typename Q, typename N> SequenceDataStorage(const SequenceDataViewCore<I, S, Q, N>& other) { *this = other; }

    SequenceDataStorage& operator=(const SequenceDataStorage& other) { return this->template assign_storage<system_tag>(other); }
    template <typename other_tag> SequenceDataStorage& operator=(const SequenceDataStorage<other_tag>& other) { return this->template assign_storage<other_tag>(other); }

    /// copy out of a view over HOST memory (raw pointers are host pointers to thrust, as in the reference's thrust::copy calls, sequence.h:510-520)
    template <typename I, typename S, typename Q, typename N>
    SequenceDataStorage& operator=(const SequenceDataViewCore<I, S, Q, N>& other)
    {
        this->SequenceDataInfo::operator=(other);
        m_sequence_vec.resize(m_sequence_stream_words); m_sequence_index_vec.resize(m_n_seqs + 1u);
        m_name_vec.resize(m_name_stream_len);           m_name_index_vec.resize(m_n_seqs + 1u);
        m_qual_vec.resize(m_has_qualities ? m_sequence_stream_len : 0u);
        priv::seq_copy(other.sequence_storage(), m_sequence_stream_words, m_sequence_vec.begin());
        priv::seq_copy(other.sequence_index(),   m_n_seqs + 1u,           m_sequence_index_vec.begin());
        priv::seq_copy(other.name_stream(),      m_name_stream_len,       m_name_vec.begin());
        priv::seq_copy(other.name_index(),       m_n_seqs + 1u,           m_name_index_vec.begin());
        if (m_has_qualities) priv::seq_copy(other.qual_stream(), m_sequence_stream_len, m_qual_vec.begin());
        priv::coordinate_cover<system_tag>::apply(m_sequence_vec, uint32(m_alphabet));
        return *this;
    }

    virtual operator plain_view_type()
    { return plain_view_type(*this, priv::seq_ptr(m_sequence_vec), priv::seq_ptr(m_sequence_index_vec), priv::seq_ptr(m_qual_vec), priv::seq_ptr(m_name_vec), priv::seq_ptr(m_name_index_vec)); }
    virtual operator const_plain_view_type() const
    { return const_plain_view_type(*this, priv::seq_ptr(m_sequence_vec), priv::seq_ptr(m_sequence_index_vec), priv::seq_ptr(m_qual_vec), priv::seq_ptr(m_name_vec), priv::seq_ptr(m_name_index_vec)); }

    /// room for n_seqs sequences of n_bps symbols in total
    void reserve(const uint32 n_seqs, const uint32 n_bps)
    {
        const uint32 per_word = 32u / bits_per_symbol(m_alphabet);
        m_sequence_index_vec.reserve(n_seqs + 1u); m_sequence_vec.reserve((n_bps + per_word - 1u) / per_word);
        m_qual_vec.reserve(n_bps); m_name_index_vec.reserve(n_seqs + 1u);
    }

    index_iterator                  name_index()             { return m_name_index_vec.begin(); }
    index_iterator                  sequence_index()         { return m_sequence_index_vec.begin(); }
    name_storage_iterator           name_stream()            { return m_name_vec.begin(); }
    sequence_storage_iterator       sequence_storage()       { return m_sequence_vec.begin(); }
    qual_storage_iterator           qual_stream()            { return m_qual_vec.begin(); }
    const_index_iterator            name_index()       const { return m_name_index_vec.begin(); }
    const_index_iterator            sequence_index()   const { return m_sequence_index_vec.begin(); }
    const_name_storage_iterator     name_stream()      const { return m_name_vec.begin(); }
    const_sequence_storage_iterator sequence_storage() const { return m_sequence_vec.begin(); }
    const_qual_storage_iterator     qual_stream()      const { return m_qual_vec.begin(); }

    word_vector m_sequence_vec;
    word_vector m_sequence_index_vec;
    char_vector m_qual_vec;
    char_vector m_name_vec;
    word_vector m_name_index_vec;

private:
    template <typename other_tag>
    SequenceDataStorage& assign_storage(const SequenceDataStorage<other_tag>& other)
    {
        this->SequenceDataInfo::operator=(other);
        m_sequence_vec = other.m_sequence_vec; m_sequence_index_vec = other.m_sequence_index_vec;
        m_qual_vec = other.m_qual_vec; m_name_vec = other.m_name_vec; m_name_index_vec = other.m_name_index_vec;
        priv::coordinate_cover<system_tag>::apply(m_sequence_vec, uint32(m_alphabet));
        return *this;
    }
};
typedef SequenceDataStorage<host_tag>   SequenceDataHost;
typedef SequenceDataStorage<device_tag> SequenceDataDevice;

} // namespace io

template <typename system_tag> inline io::SequenceDataView      plain_view(io::SequenceDataStorage<system_tag>& data)       { return io::SequenceDataView(data); }
template <typename system_tag> inline io::ConstSequenceDataView plain_view(const io::SequenceDataStorage<system_tag>& data) { return io::ConstSequenceDataView(data); }
inline io::SequenceDataView      plain_view(io::SequenceData& data)                { return io::SequenceDataView(data); }
inline io::ConstSequenceDataView plain_view(const io::SequenceData& data)          { return io::ConstSequenceDataView(data); }
inline io::SequenceDataView      plain_view(io::SequenceDataView& view)            { return view; }
inline io::ConstSequenceDataView plain_view(const io::ConstSequenceDataView& view) { return view; }

namespace io {

/// alphabet-aware accessors over any view or container (sequence_access.h:52-180)
template <Alphabet SEQUENCE_ALPHABET_T, typename SequenceDataT = ConstSequenceDataView>
struct SequenceDataAccess
{
    static const Alphabet SEQUENCE_ALPHABET         = SEQUENCE_ALPHABET_T;
    static const uint32   SEQUENCE_BITS             = SequenceDataTraits<SEQUENCE_ALPHABET_T>::SEQUENCE_BITS;
    static const bool     SEQUENCE_BIG_ENDIAN       = SequenceDataTraits<SEQUENCE_ALPHABET_T>::SEQUENCE_BIG_ENDIAN;
    static const uint32   SEQUENCE_SYMBOLS_PER_WORD = SequenceDataTraits<SEQUENCE_ALPHABET_T>::SEQUENCE_SYMBOLS_PER_WORD;

    typedef typename SequenceDataT::const_index_iterator             index_iterator;
    typedef typename SequenceDataT::const_sequence_storage_iterator  sequence_storage_iterator;
    typedef typename SequenceDataT::const_qual_storage_iterator      qual_storage_iterator;
    typedef typename SequenceDataT::const_name_storage_iterator      name_storage_iterator;
    typedef SequenceDataViewCore<index_iterator, sequence_storage_iterator, qual_storage_iterator, name_storage_iterator>  sequence_reference;

    typedef PackedStream<sequence_storage_iterator, uint8, SEQUENCE_BITS, SEQUENCE_BIG_ENDIAN>  sequence_stream_type;
    typedef vector_view<sequence_stream_type>   sequence_string;
    typedef vector_view<qual_storage_iterator>  qual_string;
    typedef vector_view<name_storage_iterator>  name_string;
    typedef ConcatenatedStringSet<sequence_stream_type, index_iterator>   sequence_string_set_type;
    typedef ConcatenatedStringSet<qual_storage_iterator, index_iterator>  qual_string_set_type;
    typedef ConcatenatedStringSet<name_storage_iterator, index_iterator>  name_string_set_type;

    /// (a container converts to its plain view on the way in: SequenceDataAccess<DNA_N> access( host_or_device_data ))
    NVBIO_HOST_DEVICE SequenceDataAccess(const SequenceDataT& data) : m_data(data) {}

    NVBIO_HOST_DEVICE uint32 size()             const { return m_data.size(); }
    NVBIO_HOST_DEVICE uint32 bps()              const { return m_data.bps(); }
    NVBIO_HOST_DEVICE uint32 words()            const { return m_data.words(); }
    NVBIO_HOST_DEVICE uint32 name_stream_len()  const { return m_data.name_stream_len(); }
    NVBIO_HOST_DEVICE uint32 max_sequence_len() const { return m_data.max_sequence_len(); }
    NVBIO_HOST_DEVICE uint32 min_sequence_len() const { return m_data.min_sequence_len(); }
    NVBIO_HOST_DEVICE uint32 avg_sequence_len() const { return m_data.avg_sequence_len(); }

    NVBIO_HOST_DEVICE index_iterator            name_index()       const { return m_data.name_index(); }
    NVBIO_HOST_DEVICE index_iterator            sequence_index()   const { return m_data.sequence_index(); }
    NVBIO_HOST_DEVICE name_storage_iterator     name_stream()      const { return m_data.name_stream(); }
    NVBIO_HOST_DEVICE sequence_storage_iterator sequence_storage() const { return m_data.sequence_storage(); }
    NVBIO_HOST_DEVICE qual_storage_iterator     qual_stream()      const { return m_data.qual_stream(); }

    NVBIO_HOST_DEVICE uint2 get_range(const uint32 i) const { return make_uint2(sequence_index()[i], sequence_index()[i + 1]); }
    NVBIO_HOST_DEVICE sequence_stream_type sequence_stream() const { return sequence_stream_type(sequence_storage()); }
    NVBIO_HOST_DEVICE sequence_string_set_type sequence_string_set() const { return sequence_string_set_type(size(), sequence_stream(), sequence_index()); }
    NVBIO_HOST_DEVICE qual_string_set_type     qual_string_set()     const { return qual_string_set_type(size(), qual_stream(), sequence_index()); }
    NVBIO_HOST_DEVICE name_string_set_type     name_string_set()     const { return name_string_set_type(size(), name_stream(), name_index()); }
    NVBIO_HOST_DEVICE sequence_string get_read(const uint32 i) const { const uint2 r = get_range(i); return sequence_string(r.y - r.x, sequence_stream() + r.x); }
    NVBIO_HOST_DEVICE qual_string     get_quals(const uint32 i) const { const uint2 r = get_range(i); return qual_string(r.y - r.x, qual_stream() + r.x); }
    NVBIO_HOST_DEVICE name_string     get_name(const uint32 i) const { return name_string(name_index()[i + 1] - name_index()[i] - 1u, name_stream() + name_index()[i]); }

    const sequence_reference m_data;
};
template <Alphabet ALPHABET, typename SequenceDataT>
inline SequenceDataAccess<ALPHABET, SequenceDataT> make_access(const SequenceDataT& data) { return SequenceDataAccess<ALPHABET, SequenceDataT>(data); }

namespace priv {
/// ASCII base -> {A, C, G, T, N} = {0..4}; '-' -> 5 (sequence_encoder.cpp:38-60)
inline uint8 nt4_code(const uint8 c)
{
    switch (c) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': return 3; case '-': return 5; default: return 4; }
}
/// file quality byte -> phred (sequence_encoder.cpp:64-118)
inline uint8 phred_quality(const QualityEncoding e, const uint8 q)
{
    if (e == Phred33) return uint8(q - 33u);
    if (e == Phred64) return uint8(q - 64u);
    if (e == Solexa)  { static const uint8 low[20] = { 0, 1, 1, 1, 1, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 7, 8, 9, 10 }; return q < 20u ? low[q] : uint8(q - 10u); }
    return q;
}
} // namespace priv

/// appends encoded sequences to a SequenceDataHost (sequence_encoder.h:47-140, sequence_encoder.cpp:265-420)
struct SequenceDataEncoder
{
    enum StrandOp { NO_OP = 0x0000, REVERSE_OP = 0x0001, COMPLEMENT_OP = 0x0002, REVERSE_COMPLEMENT_OP = 0x0003 };

    SequenceDataEncoder(const Alphabet alphabet, SequenceDataHost* data, const bool append = false) : m_alphabet(alphabet), m_data(data), m_append(append) {}
    virtual ~SequenceDataEncoder() {}

    virtual void reserve(const uint32 n_seqs, const uint32 n_bps) { m_data->reserve(n_seqs, n_bps); }

    /// start a batch: empty the container unless appending
    virtual void begin_batch()
    {
        if (m_append && m_data->m_sequence_index_vec.size() > 0) return;
        static_cast<SequenceDataInfo&>(*m_data) = SequenceDataInfo();
        m_data->m_alphabet = m_alphabet; m_data->m_has_qualities = 1u;
        m_data->m_sequence_vec.clear(); m_data->m_qual_vec.clear(); m_data->m_name_vec.clear();
        m_data->m_sequence_index_vec.assign(1, 0u); m_data->m_name_index_vec.assign(1, 0u);
    }

    /// append one sequence: trimmed by trim5 / trim3, truncated to max_sequence_len, optionally reversed and / or complemented
    virtual void push_back(const uint32 in_len, const char* name, const uint8* base_pairs, const uint8* quality, const QualityEncoding quality_encoding,
                           const uint32 max_sequence_len, const uint32 trim3, const uint32 trim5, const StrandOp op)
    {
        const uint32 trimmed = in_len > trim3 + trim5 ? in_len - trim3 - trim5 : 0u;
        const uint32 len = std::min(trimmed, max_sequence_len);
        base_pairs += trim5; if (quality) quality += trim5;
        const uint32 bits = bits_per_symbol(m_alphabet), per_word = 32u / bits, begin = m_data->m_sequence_stream_len, end = begin + len;
        m_data->m_sequence_vec.resize((end + per_word - 1u) / per_word, 0u);
        m_data->m_qual_vec.resize(end);
        uint32* words = priv::seq_ptr(m_data->m_sequence_vec);
        for (uint32 i = 0; i < len; ++i)
        {
            const uint32 src = (op & REVERSE_OP) ? len - 1u - i : i;
            uint32 s;
            if (m_alphabet == DNA || m_alphabet == DNA_N)
            {
                s = priv::nt4_code(base_pairs[src]);
                if (op & COMPLEMENT_OP) s = s < 4u ? 3u - s : 4u;
            }
            else s = base_pairs[src];
            const uint32 pos = begin + i, shift = 32u - bits - (pos % per_word) * bits;          // big-endian inside the word
            words[pos / per_word] = (words[pos / per_word] & ~(((1u << bits) - 1u) << shift)) | ((s & ((1u << bits) - 1u)) << shift);
            m_data->m_qual_vec[pos] = quality ? char(priv::phred_quality(quality_encoding, quality[src])) : char(0);
        }
        m_data->m_n_seqs++;
        m_data->m_sequence_stream_len = end;
        m_data->m_sequence_stream_words = (end + per_word - 1u) / per_word;
        m_data->m_sequence_index_vec.push_back(end);
        m_data->m_min_sequence_len = std::min(m_data->m_min_sequence_len, len);
        m_data->m_max_sequence_len = std::max(m_data->m_max_sequence_len, len);
        const size_t name_len = strlen(name);
        m_data->m_name_vec.insert(m_data->m_name_vec.end(), name, name + name_len + 1u);
        m_data->m_name_stream_len = uint32(m_data->m_name_vec.size());
        m_data->m_name_index_vec.push_back(m_data->m_name_stream_len);
    }

    virtual void end_batch()
    { m_data->m_avg_sequence_len = m_data->m_n_seqs ? uint32(ceilf(float(m_data->m_sequence_stream_len) / float(m_data->m_n_seqs))) : 0u; }

    /// one parsed record of a text file
    struct TextRecord { const char* name; const uint8* bases; const uint8* quals; uint32 name_len, len; };
    /// append a whole batch of records, each once per strand operation in `ops`, exactly as the push_back calls would -- but the
    /// packing of the symbols and the quality conversion are spread over the OpenMP threads (the text scanner feeding this runs at
    /// ~10 M records/s; one thread packing symbol by symbol ran at 1.2 M).  Sequences of different threads may share a word: words
    /// are OR-ed atomically into zero-initialised storage.
    void push_back_batch(const std::vector<TextRecord>& records, const QualityEncoding quality_encoding,
                         const uint32 max_sequence_len, const uint32 trim3, const uint32 trim5, const StrandOp* ops, const uint32 n_ops)
    {
        const size_t n = records.size() * n_ops;
        if (n == 0) return;
        const uint32 bits = bits_per_symbol(m_alphabet), per_word = 32u / bits, mask = (1u << bits) - 1u;
        const bool dna = (m_alphabet == DNA || m_alphabet == DNA_N);
        const uint32 first = m_data->m_n_seqs;
        m_data->m_sequence_index_vec.resize(first + n + 1u);
        m_data->m_name_index_vec.resize(first + n + 1u);
        uint32* index = priv::seq_ptr(m_data->m_sequence_index_vec);
        uint32* nindex = priv::seq_ptr(m_data->m_name_index_vec);
        uint32 end = m_data->m_sequence_stream_len, min_len = m_data->m_min_sequence_len, max_len = m_data->m_max_sequence_len;
        size_t name_bytes = m_data->m_name_vec.size();
        for (size_t r = 0; r < records.size(); ++r)
        {
            const uint32 in_len = records[r].len, trimmed = in_len > trim3 + trim5 ? in_len - trim3 - trim5 : 0u, len = std::min(trimmed, max_sequence_len);
            const size_t name_len = size_t(records[r].name_len) + 1u;
            for (uint32 o = 0; o < n_ops; ++o)
            {
                end += len; name_bytes += name_len;
                index[first + r * n_ops + o + 1u] = end; nindex[first + r * n_ops + o + 1u] = uint32(name_bytes);
            }
            min_len = std::min(min_len, len); max_len = std::max(max_len, len);
        }
        m_data->m_sequence_vec.resize((end + per_word - 1u) / per_word, 0u);
        m_data->m_qual_vec.resize(end);
        m_data->m_name_vec.resize(name_bytes);
        uint32* words = priv::seq_ptr(m_data->m_sequence_vec);
        char*   qout  = priv::seq_ptr(m_data->m_qual_vec);
        char*   nout  = priv::seq_ptr(m_data->m_name_vec);
        uint8 phred[256], code[2][256];                 // file byte -> phred; file byte -> symbol, plain and complemented
        for (uint32 q = 0; q < 256u; ++q)
        {
            phred[q] = priv::phred_quality(quality_encoding, uint8(q));
            const uint32 c = dna ? priv::nt4_code(uint8(q)) : q;
            code[0][q] = uint8(c & mask); code[1][q] = uint8((dna ? (c < 4u ? 3u - c : 4u) : c) & mask);
        }
        const uint32 word_shift = per_word == 16u ? 4u : per_word == 8u ? 3u : per_word == 4u ? 2u : 0u;
        const int64 n_records = int64(records.size());
        #pragma omp parallel for schedule(static) num_threads(usable_omp_threads())
        for (int64 r = 0; r < n_records; ++r)
        {
            const uint8* bp = records[r].bases + trim5;
            const uint8* qp = records[r].quals ? records[r].quals + trim5 : NULL;
            for (uint32 o = 0; o < n_ops; ++o)
            {
                const size_t  e = first + size_t(r) * n_ops + o;
                const uint32  begin = index[e], len = index[e + 1u] - begin;
                const StrandOp op = ops[o];
                const uint8* lut = code[(op & COMPLEMENT_OP) ? 1 : 0];
                const bool   rev = (op & REVERSE_OP) != 0;
                const ptrdiff_t step = rev ? -1 : 1;
                // qualities
                if (qp) { const uint8* q = rev ? qp + len - 1u : qp; char* o_ = qout + begin; for (uint32 i = 0; i < len; ++i, q += step) o_[i] = char(phred[*q]); }
                else if (len) memset(qout + begin, 0, len);
                // symbols: big-endian inside the word.  The words a sequence shares with its neighbours (its first and last, unless it
                // starts / ends on a word boundary) are OR-ed atomically; the ones it owns are stored
                const uint8* b_ = rev ? bp + len - 1u : bp;
                uint32 i = 0, pos = begin;
                if (word_shift)
                {
                    if (pos & (per_word - 1u))
                    {
                        uint32 acc = 0;
                        for (; i < len && (pos & (per_word - 1u)); ++i, ++pos, b_ += step) acc |= uint32(lut[*b_]) << (32u - bits - (pos & (per_word - 1u)) * bits);
                        __atomic_fetch_or(&words[(pos - 1u) >> word_shift], acc, __ATOMIC_RELAXED);
                    }
                    for (; i + per_word <= len; i += per_word, pos += per_word)
                    {
                        uint32 acc = 0;
                        for (uint32 k = 0; k < per_word; ++k, b_ += step) acc |= uint32(lut[*b_]) << (32u - bits - k * bits);
                        words[pos >> word_shift] = acc;
                    }
                    if (i < len)
                    {
                        uint32 acc = 0;
                        for (; i < len; ++i, ++pos, b_ += step) acc |= uint32(lut[*b_]) << (32u - bits - (pos & (per_word - 1u)) * bits);
                        __atomic_fetch_or(&words[(pos - 1u) >> word_shift], acc, __ATOMIC_RELAXED);
                    }
                }
                else
                {
                    uint32 acc = 0;
                    for (; i < len; ++i, ++pos, b_ += step)
                    {
                        const uint32 slot = pos % per_word;
                        acc |= uint32(lut[*b_]) << (32u - bits - slot * bits);
                        if (slot == per_word - 1u || i == len - 1u) { __atomic_fetch_or(&words[pos / per_word], acc, __ATOMIC_RELAXED); acc = 0; }
                    }
                }
                memcpy(nout + nindex[e], records[r].name, records[r].name_len); nout[nindex[e] + records[r].name_len] = '\0';
            }
        }
        m_data->m_n_seqs += uint32(n);
        m_data->m_sequence_stream_len = end;
        m_data->m_sequence_stream_words = (end + per_word - 1u) / per_word;
        m_data->m_min_sequence_len = min_len; m_data->m_max_sequence_len = max_len;
        m_data->m_name_stream_len = uint32(name_bytes);
    }
    /// true when push_back_batch() may stand in for push_back(): the encoder is this class itself, not a caller's subclass of it
    bool is_plain() const { return typeid(*this) == typeid(SequenceDataEncoder); }

    const SequenceDataInfo* info() const { return m_data; }
    Alphabet alphabet() const { return m_alphabet; }

private:
    Alphabet          m_alphabet;
    SequenceDataHost* m_data;
    bool              m_append;
};
inline SequenceDataEncoder* create_encoder(const Alphabet alphabet, SequenceDataHost* data) { return new SequenceDataEncoder(alphabet, data); }

/// a source of sequence batches
struct SequenceDataInputStream
{
    virtual ~SequenceDataInputStream() {}
    /// load up to batch_size sequences / batch_bps symbols through the encoder; returns the number loaded
    virtual int  next(SequenceDataEncoder* encoder, const uint32 batch_size, const uint32 batch_bps = uint32(-1)) = 0;
    virtual bool is_ok() = 0;
    virtual bool rewind() { return false; }
};
typedef SequenceDataInputStream SequenceDataStream;

/// the next batch of a stream into `data`, encoded over `alphabet` (sequence_encoder.cpp:455-500)
inline int next(const Alphabet alphabet, SequenceDataHost* data, SequenceDataInputStream* stream, const uint32 batch_size, const uint32 batch_bps = uint32(-1))
{
    SequenceDataEncoder encoder(alphabet, data);
    return stream->next(&encoder, batch_size, batch_bps);
}

namespace priv {
/// FASTQ ('@') and FASTA ('>') text, plain or gzip (sequence_fastq.cpp:60-300, sequence_fasta.cpp).  The grammar, as the reference's
/// character loop has it: bytes <= 31 are skipped before a record marker; the rest of the marker's line is the name ('\r' dropped);
/// FASTQ bases are the characters '!'..'~' up to a '+', the rest of that line is ignored, and as many '!'..'~' characters as there were
/// bases are the qualities, over however many lines; FASTA bases run to the next line that starts with '>', every base with the
/// quality byte 50.  A record without bases contributes nothing.
///
/// The file is read in blocks and cut into lines with memchr; a record whose bases and qualities each sit on one clean line -- every
/// record of an ordinary file -- is handed on as pointers into the block, anything else is normalised into a side buffer first.  With
/// the plain encoder a whole batch of records is then packed by all OpenMP threads at once (SequenceDataEncoder::push_back_batch);
/// a caller's own encoder subclass gets the same records one push_back at a time.
struct TextSequenceFile : public SequenceDataInputStream
{
    TextSequenceFile(const char* name, const QualityEncoding qualities, const uint32 max_seqs, const uint32 max_sequence_len, const SequenceEncoding flags,
                     const uint32 trim3, const uint32 trim5)
        : m_src(name, 1u << 16), m_qualities(qualities), m_max_seqs(max_seqs), m_max_len(max_sequence_len), m_flags(flags), m_trim3(trim3), m_trim5(trim5), m_loaded(0),
          m_ok(m_src.valid()), m_eof(false), m_cur(0), m_raw_eof(false) {}

    bool is_ok() { return m_ok; }
    bool rewind() { m_src.rewind(); m_loaded = 0; m_eof = false; m_raw.clear(); m_cur = 0; m_raw_eof = false; return true; }

    int next(SequenceDataEncoder* encoder, const uint32 batch_size, const uint32 batch_bps = uint32(-1))
    {
        const uint32 want = std::min(m_max_seqs - m_loaded, batch_size);
        if (!m_ok || want == 0u) return 0;
        encoder->begin_batch();
        { const uint32 n = std::min(want, 1u << 20); encoder->reserve(n, batch_bps == uint32(-1) ? n * 100u : std::min(batch_bps, 1u << 28)); }
        SequenceDataEncoder::StrandOp ops[4]; uint32 n_ops = 0;              // one sequence per strand requested, in this order
        if (m_flags & FORWARD)            ops[n_ops++] = SequenceDataEncoder::NO_OP;
        if (m_flags & REVERSE)            ops[n_ops++] = SequenceDataEncoder::REVERSE_OP;
        if (m_flags & FORWARD_COMPLEMENT) ops[n_ops++] = SequenceDataEncoder::COMPLEMENT_OP;
        if (m_flags & REVERSE_COMPLEMENT) ops[n_ops++] = SequenceDataEncoder::REVERSE_COMPLEMENT_OP;
        const SequenceDataInfo* info = encoder->info();
        // what the last batch consumed of the block goes; what it did not is the start of this one
        if (m_cur) { m_raw.erase(m_raw.begin(), m_raw.begin() + std::min(m_cur, m_raw.size())); m_cur = 0; }
        m_scan.clear(); m_extra.clear();
        // the stop rule of a record-at-a-time loop: enough sequences, or enough symbols, counted as they are appended
        uint64 n_seqs = info->size(), n_bps = info->bps();
        while (n_ops && n_seqs < want && n_bps < batch_bps && !m_eof)
        {
            Scan rec;
            if (!scan_record(rec)) break;
            if (rec.len == 0u) continue;
            m_scan.push_back(rec);
            const uint32 trimmed = rec.len > m_trim3 + m_trim5 ? rec.len - m_trim3 - m_trim5 : 0u;
            n_seqs += n_ops; n_bps += uint64(std::min(trimmed, m_max_len)) * n_ops;
        }
        // (the block and the side buffer stop growing here: offsets become pointers)
        m_records.resize(m_scan.size());
        for (size_t i = 0; i < m_scan.size(); ++i)
        {
            const Scan& r = m_scan[i];
            SequenceDataEncoder::TextRecord& t = m_records[i];
            t.name = reinterpret_cast<const char*>(at(r.name)); t.bases = at(r.bases); t.quals = at(r.quals); t.name_len = r.name_len; t.len = r.len;
        }
        if (encoder->is_plain()) encoder->push_back_batch(m_records, m_qualities, m_max_len, m_trim3, m_trim5, ops, n_ops);
        else
        {
            std::string name;
            for (size_t i = 0; i < m_records.size(); ++i)
            {
                const SequenceDataEncoder::TextRecord& t = m_records[i];
                name.assign(t.name, t.name_len);
                for (uint32 o = 0; o < n_ops; ++o) encoder->push_back(t.len, name.c_str(), t.bases, t.quals, m_qualities, m_max_len, m_trim3, m_trim5, ops[o]);
            }
        }
        m_loaded += info->size();
        encoder->end_batch();
        return int(info->size());
    }

private:
    static const uint64 IN_EXTRA = uint64(1) << 63;
    struct Scan { uint64 name, bases, quals; uint32 name_len, len; };       // offsets into the block, or (IN_EXTRA) the side buffer

    const uint8* at(const uint64 off) const { return (off & IN_EXTRA) ? m_extra.data() + (off & ~IN_EXTRA) : m_raw.data() + off; }
    static bool clean(const uint8* p, const size_t n) { uint32 bad = 0; for (size_t i = 0; i < n; ++i) bad |= uint32(uint8(p[i] - 0x21u) > 0x5Du); return bad == 0u; }
    void append_printable(const size_t b, const size_t e) { for (size_t i = b; i < e; ++i) if (m_raw[i] >= 0x21u && m_raw[i] <= 0x7Eu) m_extra.push_back(m_raw[i]); }

    /// more of the file into the block
    bool fill()
    {
        if (m_raw_eof) return false;
        const size_t chunk = size_t(4) << 20, old = m_raw.size();
        m_raw.resize(old + chunk);
        const size_t got = m_src.read(m_raw.data() + old, chunk);
        m_raw.resize(old + got);
        if (got == 0) m_raw_eof = true;
        return got > 0;
    }
    /// the next line [b, e) of the block, without its line end ('\n' or "\r\n"); false when the file is exhausted
    bool line(size_t& b, size_t& e)
    {
        size_t from = m_cur;
        for (;;)
        {
            const void* nl = from < m_raw.size() ? memchr(m_raw.data() + from, '\n', m_raw.size() - from) : NULL;
            if (nl) { b = m_cur; e = size_t(static_cast<const uint8*>(nl) - m_raw.data()); m_cur = e + 1u; break; }
            from = m_raw.size();
            if (!fill())
            {
                if (m_cur >= m_raw.size()) return false;
                b = m_cur; e = m_raw.size(); m_cur = e; break;                // a last line without a line end
            }
        }
        if (e > b && m_raw[e - 1u] == '\r') --e;
        return true;
    }
    bool fail(const char* what, const Scan& rec)
    {
        m_ok = false;
        fprintf(stderr, "sequence file: %s in record \"%.*s\"\n", what, int(rec.name_len), reinterpret_cast<const char*>(at(rec.name)));
        return false;
    }

    bool scan_record(Scan& rec)
    {
        size_t b = 0, e = 0, s = 0;
        for (;;)                                                               // blank lines between records
        {
            if (!line(b, e)) { m_eof = true; return false; }
            for (s = b; s < e && m_raw[s] <= 31u; ++s) {}
            if (s < e) break;
        }
        const uint8 marker = m_raw[s];
        if (marker != '@' && marker != '>') { m_ok = false; fprintf(stderr, "sequence file: parsing error (record starts with '%c')\n", marker); return false; }
        rec.name = s + 1u; rec.name_len = uint32(e - s - 1u); rec.len = 0; rec.bases = rec.quals = 0;
        if (rec.name_len && memchr(m_raw.data() + s + 1u, '\r', rec.name_len))
        {
            rec.name = IN_EXTRA | m_extra.size();
            for (size_t i = s + 1u; i < e; ++i) if (m_raw[i] != '\r') m_extra.push_back(m_raw[i]);
            rec.name_len = uint32(m_extra.size() - (rec.name & ~IN_EXTRA));
        }
        if (marker == '>')
        {
            rec.bases = IN_EXTRA | m_extra.size();
            for (;;)
            {
                const size_t before = m_cur;
                if (!line(b, e)) break;
                if (e > b && m_raw[b] == '>') { m_cur = before; break; }       // the next record's marker: not ours
                append_printable(b, e);
            }
            rec.len = uint32(m_extra.size() - (rec.bases & ~IN_EXTRA));
            rec.quals = IN_EXTRA | m_extra.size();
            m_extra.insert(m_extra.end(), rec.len, uint8(50u));               // FASTA carries no qualities: the byte 50 goes through the quality encoding (sequence_fasta.cpp:60-61)
            return true;
        }
        // FASTQ bases: up to the first '+'
        bool direct = true; size_t first_b = 0, first_e = 0; uint32 lines = 0;
        for (;;)
        {
            if (!line(b, e)) return fail("incomplete read", rec);
            const void* plus = e > b ? memchr(m_raw.data() + b, '+', e - b) : NULL;
            const size_t stop = plus ? size_t(static_cast<const uint8*>(plus) - m_raw.data()) : e;
            if (stop > b)
            {
                if (lines == 0 && clean(m_raw.data() + b, stop - b)) { first_b = b; first_e = stop; }
                else
                {
                    if (direct) { direct = false; rec.bases = IN_EXTRA | m_extra.size(); if (lines) append_printable(first_b, first_e); }
                    append_printable(b, stop);
                }
                ++lines;
            }
            if (plus) break;
        }
        if (direct) { rec.bases = first_b; rec.len = uint32(first_e - first_b); }
        else rec.len = uint32(m_extra.size() - (rec.bases & ~IN_EXTRA));
        // FASTQ qualities: as many printable characters as bases
        uint32 have = 0; direct = true; lines = 0;
        while (have < rec.len)
        {
            if (!line(b, e)) return fail("incomplete read", rec);
            if (e == b) continue;
            if (lines == 0 && e - b == rec.len && clean(m_raw.data() + b, e - b)) { rec.quals = b; have = rec.len; break; }
            if (direct) { direct = false; rec.quals = IN_EXTRA | m_extra.size(); }
            append_printable(b, e);
            have = uint32(m_extra.size() - (rec.quals & ~IN_EXTRA));
            ++lines;
        }
        // (the character loop takes exactly rec.len quality characters and trips over the rest of the line at the next record)
        if (have > rec.len) return fail("more qualities than bases", rec);
        return true;
    }

    nvbio::priv::byte_source m_src;
    QualityEncoding    m_qualities;
    uint32             m_max_seqs, m_max_len;
    SequenceEncoding   m_flags;
    uint32             m_trim3, m_trim5, m_loaded;
    bool               m_ok, m_eof;
    std::vector<uint8> m_raw, m_extra;          // the block being cut into records; the normalised irregular ones
    size_t             m_cur;                   // first unconsumed byte of the block
    bool               m_raw_eof;
    std::vector<Scan>  m_scan;
    std::vector<SequenceDataEncoder::TextRecord> m_records;
};
} // namespace priv

namespace priv {
/// reads out of alignment files (sequence_sam.cpp:360-495, sequence_bam.cpp:200-388): every record that is not a secondary alignment
/// (flag 0x100) contributes its name, SEQ and QUAL; a record flagged reverse-complemented (0x10) holds the read on the other strand, so the
/// strand operation is turned round (FORWARD: reverse-complement it back; REVERSE: complement only; ...).  SAM qualities are phred + 33, BAM
/// qualities plain phred; BAM bases are 4-bit codes of "=ACMGRSVTWYHKDBN".  BGZF is a series of gzip members: zlib's gzread walks it.
struct AlignmentSequenceFile : public SequenceDataInputStream
{
    AlignmentSequenceFile(const char* name, const bool bam, const uint32 max_seqs, const uint32 max_sequence_len, const SequenceEncoding flags, const uint32 trim3, const uint32 trim5)
        : m_src(name, 1u << 16), m_bam(bam), m_max_seqs(max_seqs), m_max_len(max_sequence_len), m_flags(flags), m_trim3(trim3), m_trim5(trim5), m_loaded(0), m_ok(m_src.valid()), m_eof(false)
    { if (m_ok) m_ok = skip_header(); }

    bool is_ok() { return m_ok; }
    bool rewind() { m_src.rewind(); m_loaded = 0; m_eof = false; m_ok = m_src.valid() && skip_header(); return m_ok; }

    int next(SequenceDataEncoder* encoder, const uint32 batch_size, const uint32 batch_bps = uint32(-1))
    {
        const uint32 want = std::min(m_max_seqs - m_loaded, batch_size);
        if (!m_ok || want == 0u) return 0;
        encoder->begin_batch();
        const SequenceDataInfo* info = encoder->info();
        while (info->size() < want && info->bps() < batch_bps && !m_eof && m_ok)
        {
            uint32 record_flags = 0;
            if (!(m_bam ? read_bam_record(record_flags) : read_sam_record(record_flags))) break;
            if (record_flags & 0x100u) continue;                                      // secondary alignment
            const bool rc = (record_flags & 0x10u) != 0u;
            typedef SequenceDataEncoder E;
            const QualityEncoding q = m_bam ? Phred : Phred33;
            const uint32 len = uint32(m_bp.size());
            if (m_flags & FORWARD)            encoder->push_back(len, m_name.c_str(), m_bp.data(), m_q.data(), q, m_max_len, m_trim3, m_trim5, rc ? E::REVERSE_COMPLEMENT_OP : E::NO_OP);
            if (m_flags & REVERSE)            encoder->push_back(len, m_name.c_str(), m_bp.data(), m_q.data(), q, m_max_len, m_trim3, m_trim5, rc ? E::COMPLEMENT_OP : E::REVERSE_OP);
            if (m_flags & FORWARD_COMPLEMENT) encoder->push_back(len, m_name.c_str(), m_bp.data(), m_q.data(), q, m_max_len, m_trim3, m_trim5, rc ? E::REVERSE_OP : E::COMPLEMENT_OP);
            if (m_flags & REVERSE_COMPLEMENT) encoder->push_back(len, m_name.c_str(), m_bp.data(), m_q.data(), q, m_max_len, m_trim3, m_trim5, rc ? E::NO_OP : E::REVERSE_COMPLEMENT_OP);
        }
        m_loaded += info->size();
        encoder->end_batch();
        return int(info->size());
    }

private:
    bool get(void* out, const size_t n) { return m_src.read(static_cast<uint8*>(out), n) == n; }
    bool skip(size_t n) { uint8 buf[256]; while (n) { const size_t k = std::min(n, sizeof(buf)); if (!get(buf, k)) return false; n -= k; } return true; }
    bool skip_header()
    {
        if (!m_bam) return true;                                                        // SAM: '@' lines are dropped as records are read
        char magic[4]; int32 l_text = 0, n_ref = 0;
        if (!get(magic, 4) || memcmp(magic, "BAM\1", 4) != 0 || !get(&l_text, 4) || !skip(size_t(l_text)) || !get(&n_ref, 4)) return false;
        for (int32 i = 0; i < n_ref; ++i) { int32 l_name = 0; if (!get(&l_name, 4) || !skip(size_t(l_name) + 4u)) return false; }
        return true;
    }
    bool read_sam_record(uint32& record_flags)
    {
        std::vector<uint8> line;
        for (;;)
        {
            line.clear();
            if (!m_src.get_line(line)) { m_eof = true; return false; }
            while (!line.empty() && (line.back() == '\r' || line.back() == '\n')) line.pop_back();
            if (!line.empty() && line[0] != '@') break;
        }
        // QNAME FLAG RNAME POS MAPQ CIGAR RNEXT PNEXT TLEN SEQ QUAL ...
        size_t field[12]; uint32 n = 0; field[n++] = 0;
        for (size_t i = 0; i < line.size() && n < 12u; ++i) if (line[i] == '\t') field[n++] = i + 1u;
        if (n < 11u) { m_ok = false; fprintf(stderr, "SAM file: a record with %u fields\n", n); return false; }
        if (n < 12u) field[n] = line.size() + 1u;
        auto text = [&](const uint32 k) { return std::string(reinterpret_cast<const char*>(line.data()) + field[k], field[k + 1] - 1u - field[k]); };
        m_name = text(0); record_flags = uint32(strtol(text(1).c_str(), NULL, 0));
        const std::string seq = text(9), qual = text(10);
        m_bp.assign(seq.begin(), seq.end());
        if (qual == "*") m_q.assign(m_bp.size(), uint8(33u)); else m_q.assign(qual.begin(), qual.end());
        if (m_q.size() != m_bp.size()) { m_ok = false; fprintf(stderr, "SAM file: SEQ and QUAL of \"%s\" differ in length\n", m_name.c_str()); return false; }
        return true;
    }
    bool read_bam_record(uint32& record_flags)
    {
        int32 block_size = 0;
        if (m_src.read(reinterpret_cast<uint8*>(&block_size), 4u) != 4u) { m_eof = true; return false; }
        struct { int32 refID, pos; uint32 bin_mq_nl, flag_nc; int32 l_seq, next_refID, next_pos, tlen; } h;
        if (block_size < 32 || !get(&h, 32u)) { m_ok = false; fprintf(stderr, "error processing BAM file (truncated record)\n"); return false; }
        record_flags = h.flag_nc >> 16;
        const uint32 l_name = h.bin_mq_nl & 0xFFu, n_cigar = h.flag_nc & 0xFFFFu, l_seq = uint32(h.l_seq);
        const size_t body = size_t(l_name) + 4u * n_cigar + (l_seq + 1u) / 2u + l_seq;
        if (body + 32u > size_t(block_size)) { m_ok = false; fprintf(stderr, "error processing BAM file (record sizes)\n"); return false; }
        std::vector<char> name(l_name + 1u, '\0'); std::vector<uint8> packed((l_seq + 1u) / 2u);
        m_q.resize(l_seq);
        if (!get(name.data(), l_name) || !skip(4u * n_cigar) || !get(packed.data(), packed.size()) || !get(m_q.data(), l_seq) || !skip(size_t(block_size) - 32u - body))
        { m_ok = false; fprintf(stderr, "error processing BAM file (could not fetch a record)\n"); return false; }
        m_name = name.data();
        m_bp.resize(l_seq);
        for (uint32 c = 0; c < l_seq; ++c) m_bp[c] = uint8("=ACMGRSVTWYHKDBN"[(packed[c / 2u] >> ((c & 1u) ? 0u : 4u)) & 15u]);
        for (uint32 c = 0; c < l_seq; ++c) if (m_q[c] == 0xFFu) m_q[c] = 0u;              // "no quality stored"
        return true;
    }

    nvbio::priv::byte_source m_src;
    bool               m_bam;
    uint32             m_max_seqs, m_max_len;
    SequenceEncoding   m_flags;
    uint32             m_trim3, m_trim5, m_loaded;
    bool               m_ok, m_eof;
    std::string        m_name;
    std::vector<uint8> m_bp, m_q;
};
/// one read per line, no names, every base with the best quality character '~' (sequence_txt.cpp:49-190); empty lines contribute nothing
struct TxtSequenceFile : public SequenceDataInputStream
{
    TxtSequenceFile(const char* name, const QualityEncoding qualities, const uint32 max_seqs, const uint32 max_sequence_len, const SequenceEncoding flags, const uint32 trim3, const uint32 trim5)
        : m_src(name, 1u << 16), m_qualities(qualities), m_max_seqs(max_seqs), m_max_len(max_sequence_len), m_flags(flags), m_trim3(trim3), m_trim5(trim5), m_loaded(0), m_eof(false) {}
    bool is_ok() { return m_src.valid(); }
    bool rewind() { m_src.rewind(); m_loaded = 0; m_eof = false; return true; }
    int next(SequenceDataEncoder* encoder, const uint32 batch_size, const uint32 batch_bps = uint32(-1))
    {
        const uint32 want = std::min(m_max_seqs - m_loaded, batch_size);
        if (!is_ok() || want == 0u) return 0;
        encoder->begin_batch();
        const SequenceDataInfo* info = encoder->info();
        typedef SequenceDataEncoder E;
        while (info->size() < want && info->bps() < batch_bps && !m_eof)
        {
            m_bp.clear();
            if (!m_src.get_line(m_bp)) { m_eof = true; break; }
            while (!m_bp.empty() && m_bp.back() == '\r') m_bp.pop_back();
            if (m_bp.empty()) continue;
            if (m_q.size() < m_bp.size()) m_q.resize(m_bp.size(), uint8('~'));
            const uint32 len = uint32(m_bp.size());
            if (m_flags & FORWARD)            encoder->push_back(len, "", m_bp.data(), m_q.data(), m_qualities, m_max_len, m_trim3, m_trim5, E::NO_OP);
            if (m_flags & REVERSE)            encoder->push_back(len, "", m_bp.data(), m_q.data(), m_qualities, m_max_len, m_trim3, m_trim5, E::REVERSE_OP);
            if (m_flags & FORWARD_COMPLEMENT) encoder->push_back(len, "", m_bp.data(), m_q.data(), m_qualities, m_max_len, m_trim3, m_trim5, E::COMPLEMENT_OP);
            if (m_flags & REVERSE_COMPLEMENT) encoder->push_back(len, "", m_bp.data(), m_q.data(), m_qualities, m_max_len, m_trim3, m_trim5, E::REVERSE_COMPLEMENT_OP);
        }
        m_loaded += info->size();
        encoder->end_batch();
        return int(info->size());
    }
private:
    nvbio::priv::byte_source m_src;
    QualityEncoding    m_qualities;
    uint32             m_max_seqs, m_max_len;
    SequenceEncoding   m_flags;
    uint32             m_trim3, m_trim5, m_loaded;
    bool               m_eof;
    std::vector<uint8> m_bp, m_q;
};
inline bool has_suffix(const char* name, const char* suffix)
{ const size_t n = strlen(name), k = strlen(suffix); return n >= k && strcmp(name + n - k, suffix) == 0; }
} // namespace priv

/// open a file of reads: FASTQ / FASTA text (plain or .gz; the record marker decides which), or the reads of a .sam / .bam file
/// (sequence_priv.cpp:84-220 picks by extension as well), or one read per line of a .txt file.  NULL when the file cannot be opened.
inline SequenceDataInputStream* open_sequence_file(const char* sequence_file_name, const QualityEncoding qualities = Phred33, const uint32 max_seqs = uint32(-1),
                                                   const uint32 max_sequence_len = uint32(-1), const SequenceEncoding flags = FORWARD, const uint32 trim3 = 0, const uint32 trim5 = 0)
{
    if (priv::has_suffix(sequence_file_name, ".sam") || priv::has_suffix(sequence_file_name, ".bam"))
    {
        priv::AlignmentSequenceFile* a = new priv::AlignmentSequenceFile(sequence_file_name, priv::has_suffix(sequence_file_name, ".bam"), max_seqs, max_sequence_len, flags, trim3, trim5);
        if (!a->is_ok()) { delete a; return NULL; }
        return a;
    }
    if (priv::has_suffix(sequence_file_name, ".txt") || priv::has_suffix(sequence_file_name, ".txt.gz"))
    {
        priv::TxtSequenceFile* t = new priv::TxtSequenceFile(sequence_file_name, qualities, max_seqs, max_sequence_len, flags, trim3, trim5);
        if (!t->is_ok()) { delete t; return NULL; }
        return t;
    }
    priv::TextSequenceFile* f = new priv::TextSequenceFile(sequence_file_name, qualities, max_seqs, max_sequence_len, flags, trim3, trim5);
    if (!f->is_ok()) { delete f; return NULL; }
    return f;
}

namespace priv {
inline bool file_exists(const std::string& name) { FILE* f = fopen(name.c_str(), "rb"); if (f) fclose(f); return f != NULL; }
/// a BWA-style packed reference: <prefix>.ann beside <prefix>.pac or <prefix>.wpac (sequence_pac.cpp:225-245)
inline bool is_pac_archive(const char* prefix)
{
    const std::string p(prefix);
    return file_exists(p + ".ann") && (file_exists(p + ".pac") || file_exists(p + ".wpac"));
}
/// load it: names and sequence offsets from .ann (line 1: l_pac n_seqs seed; per sequence "gi name comment" and "offset len n_ambs"), symbols
/// from .wpac ([uint64 length][2-bit big-endian words]) or .pac (BWA's byte-packed genome, last byte = symbols in the last data byte),
/// re-packed at the alphabet's width; no qualities (sequence_pac.cpp:60-330, basic/bnt.cpp:83-163)
inline bool load_pac_archive(const Alphabet alphabet, SequenceDataHost* data, const char* prefix)
{
    const std::string p(prefix);
    FILE* ann = fopen((p + ".ann").c_str(), "r");
    if (!ann) return false;
    unsigned long long l_pac = 0; int n_seqs = 0; unsigned seed = 0;
    if (fscanf(ann, "%llu %d %u", &l_pac, &n_seqs, &seed) != 3 || n_seqs <= 0) { fclose(ann); log_error(stderr, "loading BNS files failed\n"); return false; }
    static_cast<SequenceDataInfo&>(*data) = SequenceDataInfo();
    data->m_alphabet = alphabet; data->m_n_seqs = uint32(n_seqs); data->m_sequence_stream_len = uint32(l_pac); data->m_avg_sequence_len = uint32(l_pac / n_seqs);
    data->m_sequence_index_vec.assign(1, 0u); data->m_name_index_vec.assign(1, 0u); data->m_name_vec.clear(); data->m_qual_vec.clear();
    char line[8192];
    (void)fgets(line, sizeof(line), ann);                                                    // rest of the first line
    for (int i = 0; i < n_seqs; ++i)
    {
        if (!fgets(line, sizeof(line), ann)) { fclose(ann); log_error(stderr, "loading BNS files failed\n"); return false; }
        char* name = strchr(line, ' ');                                                      // "gi name [comment]"
        name = name ? name + 1 : line;
        name[strcspn(name, " \n\r")] = 0;
        data->m_name_vec.insert(data->m_name_vec.end(), name, name + strlen(name) + 1u);
        data->m_name_index_vec.push_back(uint32(data->m_name_vec.size()));
        unsigned long long offset = 0; unsigned len = 0, n_ambs = 0;
        if (!fgets(line, sizeof(line), ann) || sscanf(line, "%llu %u %u", &offset, &len, &n_ambs) != 3) { fclose(ann); log_error(stderr, "loading BNS files failed\n"); return false; }
        data->m_sequence_index_vec.push_back(uint32(offset + len));
        data->m_min_sequence_len = std::min(data->m_min_sequence_len, uint32(len));
        data->m_max_sequence_len = std::max(data->m_max_sequence_len, uint32(len));
    }
    fclose(ann);
    data->m_name_stream_len = uint32(data->m_name_vec.size());
    data->m_has_qualities = 0u;
    const uint32 n = uint32(l_pac), bits = bits_per_symbol(alphabet), per_word = 32u / bits;
    const uint32 seq_words = (n + per_word - 1u) / per_word, aligned_words = (seq_words + 3u) & ~3u;
    data->m_sequence_stream_words = aligned_words;
    data->m_sequence_vec.assign(aligned_words, 0u);
    // the 2-bit symbols, from either file
    std::vector<uint32> packed((size_t(n) + 15u) / 16u, 0u);
    FILE* f = fopen((p + ".wpac").c_str(), "rb");
    if (f)
    {
        uint64 len = 0;
        const bool ok = fread(&len, sizeof(len), 1, f) == 1 && uint32(len) == n && fread(packed.data(), sizeof(uint32), packed.size(), f) == packed.size();
        fclose(f);
        if (!ok) { log_error(stderr, "failed reading %s.wpac\n", prefix); return false; }
    }
    else
    {
        f = fopen((p + ".pac").c_str(), "rb");
        if (!f) { log_warning(stderr, "unable to open %s.[w]pac\n", prefix); return false; }
        fseek(f, 0, SEEK_END);
        const long size = ftell(f);
        std::vector<uint8> raw(size > 0 ? size_t(size) : 0u);
        fseek(f, 0, SEEK_SET);
        const bool ok = size >= 2 && fread(raw.data(), 1, raw.size(), f) == raw.size();
        fclose(f);
        // BWA writes ceil(n / 4) data bytes, one zero byte when n is a multiple of 4, then the count byte (n % 4)
        if (!ok || uint32(size - 1 - (raw[size - 1] == 0 ? 1 : 0)) != (n + 3u) / 4u && uint32(size - 1) != (n + 3u) / 4u)
        { log_error(stderr, "mismatching sequence lengths in %s.pac\n", prefix); return false; }
        for (uint32 b = 0; b < (n + 3u) / 4u; ++b) packed[b >> 2] |= uint32(raw[b]) << (24u - 8u * (b & 3u));
    }
    uint32* words = priv::seq_ptr(data->m_sequence_vec);
    if (bits == 2u) std::copy(packed.begin(), packed.end(), words);
    else
        for (uint32 i = 0; i < n; ++i)
        {
            const uint32 sym = (packed[i >> 4] >> (30u - 2u * (i & 15u))) & 3u;
            words[i / per_word] |= sym << (32u - bits - (i % per_word) * bits);
        }
    return true;
}
} // namespace priv

/// load a whole file into `sequence_data`: a packed reference archive (<name>.ann + .pac / .wpac) or a FASTQ / FASTA file
inline bool load_sequence_file(const Alphabet alphabet, SequenceDataHost* sequence_data, const char* sequence_file_name,
                               const SequenceFlags load_flags = SequenceFlags(SEQUENCE_DATA | SEQUENCE_QUALS | SEQUENCE_NAMES), const QualityEncoding qualities = Phred33)
{
    (void)load_flags;
    if (priv::is_pac_archive(sequence_file_name)) return priv::load_pac_archive(alphabet, sequence_data, sequence_file_name);
    SequenceDataInputStream* f = open_sequence_file(sequence_file_name, qualities);
    if (f == NULL) return false;
    const int n = next(alphabet, sequence_data, f, uint32(-1), uint32(-1));
    const bool ok = f->is_ok() && n >= 0;
    delete f;
    return ok;
}
inline SequenceDataHost* load_sequence_file(const Alphabet alphabet, const char* sequence_file_name,
                                            const SequenceFlags load_flags = SequenceFlags(SEQUENCE_DATA | SEQUENCE_QUALS | SEQUENCE_NAMES), const QualityEncoding qualities = Phred33)
{
    SequenceDataHost* data = new SequenceDataHost();
    if (!load_sequence_file(alphabet, data, sequence_file_name, load_flags, qualities)) { delete data; return NULL; }
    return data;
}

} // namespace io
} // namespace nvbio
