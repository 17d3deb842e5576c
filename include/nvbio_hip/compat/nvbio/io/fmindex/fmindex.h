// compat/nvbio/io/fmindex/fmindex.h -- the FM-index as an application loads it (nvbio/io/fmindex/fmindex.h:52-375, fmindex_impl.cu):
// io::FMIndexDataHost::load(prefix, flags) reads <prefix>.bwt / .rbwt (2-bit big-endian BWT words behind a primary + cumulative-count
// header) and <prefix>.sa / .rsa (sampled suffix arrays, interval 16), interleaves BWT words and running symbol counts into the
// production layout (per 64 symbols: one uint4 of BWT, one uint4 of counts), and io::FMIndexDataDevice mirrors it in device memory.  The
// fm_index_type / rank_dict_type / ssa_type typedefs are the reference's, over this layer's templates -- index() / rindex() /
// partial_index() hand out what nvBowtie's mappers and locate kernels take.
#pragma once
#include "../../basic/types.h"
#include "../../basic/vector.h"
#include "../../basic/console.h"
#include "../../basic/deinterleaved_iterator.h"
#include "../../basic/cuda/ldg.h"
#include "../../basic/packedstream.h"
#include "../../fmindex/bwt.h"
#include "../../fmindex/fmindex.h"
#include "../../fmindex/ssa.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <string>
#include <vector>
#if defined(__HIPCC__) && !defined(NVBIO_HIP_COMPAT_NO_TUNED)
#include "../../../../../nvbio_hip.h"
#define NVBIO_HIP_COMPAT_IO_TUNED 1
#endif

namespace nvbio {
namespace io {

/// counts, flags and raw pointers shared by every flavour (fmindex.h:52-130)
struct FMIndexDataCore
{
    static const uint32 FORWARD = 0x02;
    static const uint32 REVERSE = 0x04;
    static const uint32 SA      = 0x10;

    static const uint32 BWT_BITS             = 2u;
    static const bool   BWT_BIG_ENDIAN       = true;
    static const uint32 BWT_SYMBOLS_PER_WORD = 32u / BWT_BITS;
    static const uint32 OCC_INT = 64;
    static const uint32 SA_INT  = 16;

    typedef const uint32*                                       bwt_occ_type;
    typedef const uint32*                                       count_table_type;
    typedef SSA_index_multiple_context<SA_INT, const uint32*>   ssa_type;

    FMIndexDataCore() : m_flags(0), m_seq_length(0), m_bwt_occ_words(0), m_sa_words(0), m_primary(0), m_rprimary(0),
        m_L2(NULL), m_bwt_occ(NULL), m_rbwt_occ(NULL), m_count_table(NULL), m_ssa((const uint32*)NULL), m_rssa((const uint32*)NULL) {}

    uint32        flags()         const { return m_flags; }
    uint32        length()        const { return m_seq_length; }
    uint32        primary()       const { return m_primary; }
    uint32        rprimary()      const { return m_rprimary; }
    bool          has_ssa()       const { return m_ssa.m_ssa  != NULL; }
    bool          has_rssa()      const { return m_rssa.m_ssa != NULL; }
    const uint32* bwt_occ()       const { return m_bwt_occ; }
    const uint32* rbwt_occ()      const { return m_rbwt_occ; }
    const uint32* count_table()   const { return m_count_table; }
    uint32        bwt_occ_words() const { return m_bwt_occ_words; }
    uint32        sa_words()      const { return m_sa_words; }
    ssa_type      ssa()           const { return m_ssa; }
    ssa_type      rssa()          const { return m_rssa; }
    const uint32* L2()            const { return m_L2; }

    uint32   m_flags, m_seq_length, m_bwt_occ_words, m_sa_words, m_primary, m_rprimary;
    uint32*  m_L2;
    uint32*  m_bwt_occ;
    uint32*  m_rbwt_occ;
    uint32*  m_count_table;
    ssa_type m_ssa, m_rssa;
};

/// the host-side view with the reference's iterator / index typedefs (fmindex.h:135-190)
struct FMIndexData : public FMIndexDataCore
{
    typedef const uint4*                                          bwt_occ_type;
    typedef deinterleaved_iterator<2, 0, bwt_occ_type>            bwt_type;
    typedef deinterleaved_iterator<2, 1, bwt_occ_type>            occ_type;
    typedef const uint32*                                         count_table_type;
    typedef SSA_index_multiple<SA_INT>                            ssa_storage_type;
    typedef PackedStream<bwt_type, uint8, BWT_BITS, BWT_BIG_ENDIAN> bwt_stream_type;
    typedef rank_dictionary<BWT_BITS, FMIndexDataCore::OCC_INT, bwt_stream_type, occ_type, count_table_type>  rank_dict_type;
    typedef fm_index<rank_dict_type, ssa_type>                    fm_index_type;
    typedef fm_index<rank_dict_type, null_type>                   partial_fm_index_type;

    FMIndexData() {}
    virtual ~FMIndexData() {}

    occ_type  occ_iterator()  const { return occ_type(bwt_occ_type(bwt_occ())); }
    occ_type  rocc_iterator() const { return occ_type(bwt_occ_type(rbwt_occ())); }
    bwt_type  bwt_iterator()  const { return bwt_type(bwt_occ_type(bwt_occ())); }
    bwt_type  rbwt_iterator() const { return bwt_type(bwt_occ_type(rbwt_occ())); }
    ssa_type  ssa_iterator()  const { return ssa(); }
    ssa_type  rssa_iterator() const { return rssa(); }
    count_table_type count_table_iterator() const { return count_table_type(count_table()); }

    rank_dict_type rank_dict()  const { return rank_dict_type(bwt_stream_type(bwt_iterator()),  occ_iterator(),  count_table_iterator()); }
    rank_dict_type rrank_dict() const { return rank_dict_type(bwt_stream_type(rbwt_iterator()), rocc_iterator(), count_table_iterator()); }
    fm_index_type  index()  const { return fm_index_type(length(), primary(),  L2(), rank_dict(),  ssa_iterator()); }
    fm_index_type  rindex() const { return fm_index_type(length(), rprimary(), L2(), rrank_dict(), rssa_iterator()); }
    partial_fm_index_type partial_index()  const { return partial_fm_index_type(length(), primary(),  L2(), rank_dict(),  null_type()); }
    partial_fm_index_type rpartial_index() const { return partial_fm_index_type(length(), rprimary(), L2(), rrank_dict(), null_type()); }

};

namespace priv {
inline bool read_index_words(FILE* f, uint32* dst, const size_t n) { return n == 0 || fread(dst, sizeof(uint32), n, f) == n; }

/// <name>: [primary][cumulative counts x 4][BWT words]; out = per 64 symbols { 4 BWT words, 4 running counts }; L2[c + 1] = symbols <= c
inline bool load_bwt_occ(const char* name, std::vector<uint32>& bwt_occ, uint32& seq_length, uint32& primary, uint32* L2)
{
    FILE* f = fopen(name, "rb");
    if (!f) { log_warning(stderr, "unable to open bwt \"%s\"\n", name); return false; }
    uint32 header[5];
    if (!read_index_words(f, header, 5)) { log_error(stderr, "failed reading bwt \"%s\"\n", name); fclose(f); return false; }
    primary = header[0]; seq_length = header[4];
    const uint32 n_words = (seq_length + 15u) / 16u, n_blocks = (seq_length + 63u) / 64u;
    std::vector<uint32> bwt(size_t(n_blocks) * 4u, 0u);
    const bool ok = read_index_words(f, bwt.data(), n_words);
    fclose(f);
    if (!ok) { log_error(stderr, "failed reading bwt \"%s\"\n", name); return false; }
    bwt_occ.assign(size_t(n_blocks) * 8u, 0u);
    uint32 running[4] = { 0, 0, 0, 0 };
    for (uint32 b = 0; b < n_blocks; ++b)
    {
        uint32* record = bwt_occ.data() + size_t(b) * 8u;
        for (uint32 w = 0; w < 4; ++w) { record[w] = bwt[size_t(b) * 4u + w]; record[4 + w] = running[w]; }
        const uint32 last = std::min(seq_length, (b + 1u) * 64u);
        for (uint32 i = b * 64u; i < last; ++i) running[(bwt[i >> 4] >> (30u - 2u * (i & 15u))) & 3u]++;
    }
    L2[0] = 0;
    for (uint32 c = 0; c < 4; ++c) L2[c + 1] = L2[c] + running[c];
    return true;
}
/// <name>: [primary][cumulative counts x 4][interval][length][ssa 1 ..]; a file that does not belong to this BWT is skipped
inline bool load_ssa(const char* name, std::vector<uint32>& ssa, const uint32 seq_length, const uint32 primary, const uint32 sa_int)
{
    FILE* f = fopen(name, "rb");
    if (!f) return false;
    uint32 header[7];
    bool ok = read_index_words(f, header, 7);
    if (ok && header[0] != primary)    { log_error(stderr, "SA file mismatch \"%s\"\n  expected primary %u, got %u\n", name, primary, header[0]); ok = false; }
    if (ok && header[5] != sa_int)     { log_error(stderr, "unsupported SA interval (found %u, expected %u)\n", header[5], sa_int); ok = false; }
    if (ok && header[6] != seq_length) { log_error(stderr, "SA file mismatch \"%s\"\n  expected length %u, got %u\n", name, seq_length, header[6]); ok = false; }
    if (ok)
    {
        ssa.assign((size_t(seq_length) + sa_int) / sa_int, 0u);
        ssa[0] = uint32(-1);
        ok = read_index_words(f, ssa.data() + 1, ssa.size() - 1u);
        if (!ok) { log_error(stderr, "failed reading SSA \"%s\"\n", name); ssa.clear(); }
    }
    fclose(f);
    return ok;
}
} // namespace priv

/// an index loaded into host memory (fmindex.h:196-215)
struct FMIndexDataHost : public FMIndexData
{
    /// returns 1 on success, 0 on failure
    int load(const char* genome_prefix, const uint32 flags = FORWARD | REVERSE | SA)
    {
        const std::string prefix(genome_prefix);
        log_visible(stderr, "FMIndexData: loading... started\n");
        log_visible(stderr, "  genome : %s\n", genome_prefix);
        m_flags = flags;
        uint32 L2r[5];
        if (flags & FORWARD) { if (!priv::load_bwt_occ((prefix + ".bwt").c_str(),  m_bwt_occ_vec,  m_seq_length, m_primary,  m_L2_vec)) return 0; }
        if (flags & REVERSE) { if (!priv::load_bwt_occ((prefix + ".rbwt").c_str(), m_rbwt_occ_vec, m_seq_length, m_rprimary, (flags & FORWARD) ? L2r : m_L2_vec)) return 0; }
        if (flags & SA)
        {
            if (flags & FORWARD) priv::load_ssa((prefix + ".sa").c_str(),  m_ssa_vec,  m_seq_length, m_primary,  SA_INT);
            if (flags & REVERSE) priv::load_ssa((prefix + ".rsa").c_str(), m_rssa_vec, m_seq_length, m_rprimary, SA_INT);
        }
        gen_bwt_count_table(m_count_table_vec);
        m_bwt_occ_words = uint32(std::max(m_bwt_occ_vec.size(), m_rbwt_occ_vec.size()));
        m_sa_words      = uint32(std::max(m_ssa_vec.size(), m_rssa_vec.size()));
        m_L2          = m_L2_vec;
        m_count_table = m_count_table_vec;
        m_bwt_occ     = m_bwt_occ_vec.empty()  ? NULL : m_bwt_occ_vec.data();
        m_rbwt_occ    = m_rbwt_occ_vec.empty() ? NULL : m_rbwt_occ_vec.data();
        m_ssa         = ssa_type(m_ssa_vec.empty()  ? (const uint32*)NULL : m_ssa_vec.data());
        m_rssa        = ssa_type(m_rssa_vec.empty() ? (const uint32*)NULL : m_rssa_vec.data());
        log_visible(stderr, "FMIndexData: loading... done\n");
        log_verbose(stderr, "  length   : %u\n  primary  : %u (reverse %u)\n  storage  : %.1f MB\n", m_seq_length, m_primary, m_rprimary,
                    float(sizeof(uint32)) * float(m_bwt_occ_vec.size() + m_rbwt_occ_vec.size() + m_ssa_vec.size() + m_rssa_vec.size()) / float(1024 * 1024));
        return 1;
    }

    std::vector<uint32> m_bwt_occ_vec, m_rbwt_occ_vec, m_ssa_vec, m_rssa_vec;
    uint32              m_count_table_vec[256];
    uint32              m_L2_vec[5];
};

/// an index published by a server process through shared memory (fmindex.h:250-290).  No such server exists for this layer: load() fails
/// and callers fall back to FMIndexDataHost (see io/sequence/sequence_mmap.h)
struct FMIndexDataMMAP : public FMIndexData
{
    int load(const char*) { return 0; }
};

/// build sampled suffix arrays from the index itself, for indices loaded without them
inline void init_ssa(const FMIndexData& driver_data, FMIndexData::ssa_storage_type& ssa, FMIndexData::ssa_storage_type& rssa)
{
    typedef FMIndexData::partial_fm_index_type partial_type;
    if (driver_data.bwt_occ())  ssa  = FMIndexData::ssa_storage_type(driver_data.partial_index());
    if (driver_data.rbwt_occ()) rssa = FMIndexData::ssa_storage_type(driver_data.rpartial_index());
    (void)sizeof(partial_type);
}

#if defined(__HIPCC__)
/// the device mirror (fmindex.h:294-362): same layout, iterators that read through cuda::ldg_pointer
struct FMIndexDataDevice : public FMIndexData
{
    static const uint32 FORWARD = 0x02;
    static const uint32 REVERSE = 0x04;
    static const uint32 SA      = 0x10;

    typedef cuda::ldg_pointer<uint4>                              bwt_occ_type;
    typedef deinterleaved_iterator<2, 0, bwt_occ_type>            bwt_type;
    typedef deinterleaved_iterator<2, 1, bwt_occ_type>            occ_type;
    typedef cuda::ldg_pointer<uint32>                             count_table_type;
    typedef cuda::ldg_pointer<uint32>                             ssa_ldg_type;
    typedef SSA_index_multiple_device<SA_INT>                     ssa_storage_type;
    typedef PackedStream<bwt_type, uint8, BWT_BITS, BWT_BIG_ENDIAN> bwt_stream_type;
    typedef SSA_index_multiple_context<FMIndexDataCore::SA_INT, ssa_ldg_type>  ssa_type;
    typedef rank_dictionary<BWT_BITS, FMIndexDataCore::OCC_INT, bwt_stream_type, occ_type, count_table_type>  rank_dict_type;
    typedef fm_index<rank_dict_type, ssa_type>                    fm_index_type;
    typedef fm_index<rank_dict_type, null_type>                   partial_fm_index_type;

    /// copy the parts of host_data named by flags (FORWARD / REVERSE / SA) into device memory
    FMIndexDataDevice(const FMIndexData& host_data, const uint32 flags = FORWARD | REVERSE) : m_allocated(0)
    {
        m_flags = flags; m_seq_length = host_data.m_seq_length; m_primary = host_data.m_primary; m_rprimary = host_data.m_rprimary;
        m_bwt_occ_words = host_data.m_bwt_occ_words; m_sa_words = host_data.m_sa_words;
        const size_t n_occ = size_t((m_seq_length + 63u) / 64u) * 8u, n_sa = (size_t(m_seq_length) + SA_INT) / SA_INT;
        upload(m_L2_vec, host_data.m_L2, 5u);                    m_L2 = nvbio::raw_pointer(m_L2_vec);
        upload(m_count_table_vec, host_data.m_count_table, 256u); m_count_table = nvbio::raw_pointer(m_count_table_vec);
        if ((flags & FORWARD) && host_data.m_bwt_occ)  { upload(m_bwt_occ_vec,  host_data.m_bwt_occ,  n_occ); m_bwt_occ  = nvbio::raw_pointer(m_bwt_occ_vec); }
        if ((flags & REVERSE) && host_data.m_rbwt_occ) { upload(m_rbwt_occ_vec, host_data.m_rbwt_occ, n_occ); m_rbwt_occ = nvbio::raw_pointer(m_rbwt_occ_vec); }
        if (flags & SA)
        {
            if ((flags & FORWARD) && host_data.has_ssa())  { upload(m_ssa_vec,  host_data.m_ssa.m_ssa,  n_sa); m_ssa  = FMIndexDataCore::ssa_type(nvbio::raw_pointer(m_ssa_vec)); }
            if ((flags & REVERSE) && host_data.has_rssa()) { upload(m_rssa_vec, host_data.m_rssa.m_ssa, n_sa); m_rssa = FMIndexDataCore::ssa_type(nvbio::raw_pointer(m_rssa_vec)); }
        }
        // the line-native records of each index loaded (fmindex/line_native.h): 3.7 bytes per SA row of the 288 GB, built on the device from the
        // arrays just uploaded; index() / rindex() carry their address.  NVBIO_HIP_COMPAT_LINE_NATIVE=0 keeps the reference layout alone.
        m_native = m_rnative = NULL;
        m_full_sa = m_rfull_sa = NULL;
#if defined(NVBIO_HIP_COMPAT_IO_TUNED)
        const char* off = getenv("NVBIO_HIP_COMPAT_LINE_NATIVE");
        if (!(off && off[0] == '0'))
        {
            if (m_bwt_occ)  m_native  = build_line_native(m_native_vec,  m_bwt_occ,  m_primary,  m_ssa.m_ssa,  host_data.m_L2);
            if (m_rbwt_occ) m_rnative = build_line_native(m_rnative_vec, m_rbwt_occ, m_rprimary, m_rssa.m_ssa, host_data.m_L2);
        }
        // ... and the whole suffix array of each index that came with a sampled one (4 bytes per row: 12 GB at 3 Gbp), when a third of the free memory
        // holds it: locate_ssa_iterator is then one load instead of an LF walk of ~15 steps.  NVBIO_HIP_COMPAT_FULL_SA=0 keeps the sampled array alone.
        const char* fsa = getenv("NVBIO_HIP_COMPAT_FULL_SA");
        if (!(fsa && fsa[0] == '0'))
        {
            if (m_bwt_occ  && m_ssa.m_ssa)  m_full_sa  = build_full_sa(m_full_sa_vec,  m_bwt_occ,  m_primary,  m_ssa.m_ssa,  host_data.m_L2, m_native);
            if (m_rbwt_occ && m_rssa.m_ssa) m_rfull_sa = build_full_sa(m_rfull_sa_vec, m_rbwt_occ, m_rprimary, m_rssa.m_ssa, host_data.m_L2, m_rnative);
        }
#endif
    }
    uint64 allocated() const { return m_allocated; }
    const uint32* line_native()  const { return m_native; }       ///< device memory, NULL = not built
    const uint32* rline_native() const { return m_rnative; }

    occ_type  occ_iterator()  const { return occ_type(bwt_occ_type(reinterpret_cast<const uint4*>(bwt_occ()))); }
    occ_type  rocc_iterator() const { return occ_type(bwt_occ_type(reinterpret_cast<const uint4*>(rbwt_occ()))); }
    bwt_type  bwt_iterator()  const { return bwt_type(bwt_occ_type(reinterpret_cast<const uint4*>(bwt_occ()))); }
    bwt_type  rbwt_iterator() const { return bwt_type(bwt_occ_type(reinterpret_cast<const uint4*>(rbwt_occ()))); }
    ssa_type  ssa_iterator()  const { return ssa_type(ssa_ldg_type(m_ssa.m_ssa)); }
    ssa_type  rssa_iterator() const { return ssa_type(ssa_ldg_type(m_rssa.m_ssa)); }
    count_table_type count_table_iterator() const { return count_table_type(count_table()); }

    rank_dict_type rank_dict()  const { return rank_dict_type(bwt_stream_type(bwt_iterator()),  occ_iterator(),  count_table_iterator()); }
    rank_dict_type rrank_dict() const { return rank_dict_type(bwt_stream_type(rbwt_iterator()), rocc_iterator(), count_table_iterator()); }
    fm_index_type  index()  const { fm_index_type f(length(), primary(),  L2(), rank_dict(),  ssa_iterator());  f.set_line_native(m_native);  f.set_full_sa(m_full_sa);  return f; }
    fm_index_type  rindex() const { fm_index_type f(length(), rprimary(), L2(), rrank_dict(), rssa_iterator()); f.set_line_native(m_rnative); f.set_full_sa(m_rfull_sa); return f; }
    const uint32* full_sa()  const { return m_full_sa; }          ///< device memory, NULL = not built
    const uint32* rfull_sa() const { return m_rfull_sa; }
    partial_fm_index_type partial_index()  const { return partial_fm_index_type(length(), primary(),  L2(), rank_dict(),  null_type()); }
    partial_fm_index_type rpartial_index() const { return partial_fm_index_type(length(), rprimary(), L2(), rrank_dict(), null_type()); }

private:
#if defined(NVBIO_HIP_COMPAT_IO_TUNED)
    /// nvbio_hip_fm_build_dimer_index over one uploaded index; NULL (and a warning) when the device cannot spare the room or the build fails
    const uint32* build_line_native(nvbio::vector<device_tag, uint32>& store, const uint32* bwt_occ, const uint32 primary, const uint32* ssa, const uint32* host_L2)
    {
        nvbio_hip_fmindex m;
        memset(&m, 0, sizeof(m));
        m.length = m_seq_length; m.primary = primary; m.sa_int = SA_INT;
        for (int i = 0; i < 5; ++i) m.L2[i] = host_L2[i];
        m.bwt_occ = bwt_occ; m.ssa = ssa;
        const uint64 bytes = nvbio_hip_fm_dimer_index_bytes(m_seq_length), tb = nvbio_hip_fm_build_dimer_index_temp_bytes(m_seq_length);
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || uint64(free_b) < 2u * (bytes + tb))
        { (void)hipGetLastError(); log_warning(stderr, "FMIndexDataDevice: no room for the line-native index (%.1f GB), staying on the reference layout\n", float(bytes) * 1.0e-9f); return NULL; }
        try
        {
            store.resize(size_t(bytes / 4u) + 64u);
            nvbio::vector<device_tag, uint8> temp(size_t(tb) + 16u);
            uint32* base = nvbio::raw_pointer(store);
            base += ((128u - uint32(uintptr_t(base) & 127u)) & 127u) / 4u;
            if (nvbio_hip_fm_build_dimer_index(&m, base, nvbio::raw_pointer(temp), tb, 0) != 0 || hipStreamSynchronize(0) != hipSuccess)
            { (void)hipGetLastError(); store.clear(); log_warning(stderr, "FMIndexDataDevice: building the line-native index failed, staying on the reference layout\n"); return NULL; }
            m_allocated += bytes;
            return base;
        }
        catch (...) { (void)hipGetLastError(); store.clear(); return NULL; }
    }
    /// nvbio_hip_fm_build_dense_ssa(sa_int = 1) over one uploaded index; NULL when the device cannot spare the room or the build fails
    const uint32* build_full_sa(nvbio::vector<device_tag, uint32>& store, const uint32* bwt_occ, const uint32 primary, const uint32* ssa, const uint32* host_L2, const uint32* native)
    {
        nvbio_hip_fmindex m;
        memset(&m, 0, sizeof(m));
        m.length = m_seq_length; m.primary = primary; m.sa_int = SA_INT;
        for (int i = 0; i < 5; ++i) m.L2[i] = host_L2[i];
        m.bwt_occ = bwt_occ; m.ssa = ssa;
        if (native && nvbio_hip_fm_attach_dimer_index(&m, native, 0) != 0) { (void)hipGetLastError(); m.dimer = NULL; }      // (the walk that fills the array goes two positions per line with it)
        const uint64 bytes = nvbio_hip_fm_dense_ssa_entries(m_seq_length, 1u) * 4u;
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || uint64(free_b) < 3u * bytes) { (void)hipGetLastError(); return NULL; }
        try
        {
            store.resize(size_t(bytes / 4u));
            uint32* out = nvbio::raw_pointer(store);
            if (nvbio_hip_fm_build_dense_ssa(&m, 1u, out, 0) != 0 || hipStreamSynchronize(0) != hipSuccess)
            { (void)hipGetLastError(); store.clear(); log_warning(stderr, "FMIndexDataDevice: building the whole suffix array failed, staying on the sampled one\n"); return NULL; }
            m_allocated += bytes;
            return out;
        }
        catch (...) { (void)hipGetLastError(); store.clear(); return NULL; }
    }
#endif
    void upload(nvbio::vector<device_tag, uint32>& dst, const uint32* src, const size_t n)
    {
        dst.resize(n);
        if (n) thrust::copy(src, src + n, dst.begin());
        m_allocated += uint64(n) * sizeof(uint32);
    }
    uint64                             m_allocated;
    nvbio::vector<device_tag, uint32>  m_bwt_occ_vec, m_rbwt_occ_vec, m_ssa_vec, m_rssa_vec, m_count_table_vec, m_L2_vec, m_native_vec, m_rnative_vec, m_full_sa_vec, m_rfull_sa_vec;
    const uint32*                      m_native;
    const uint32*                      m_rnative;
    const uint32*                      m_full_sa;
    const uint32*                      m_rfull_sa;
};

inline void init_ssa(const FMIndexDataDevice& driver_data, FMIndexDataDevice::ssa_storage_type& ssa, FMIndexDataDevice::ssa_storage_type& rssa)
{
    if (driver_data.bwt_occ())  ssa.init(driver_data.partial_index());
    if (driver_data.rbwt_occ()) rssa.init(driver_data.rpartial_index());
}
#endif // __HIPCC__

} // namespace io
} // namespace nvbio
