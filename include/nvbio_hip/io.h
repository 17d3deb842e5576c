// nvbio_hip/io.h -- on-disk index / genome formats either side of the hot path, with the reference's
// class names: io::FMIndexDataHost::load (nvbio/io/fmindex/fmindex_impl.cu:340-480), io::FMIndexDataDevice
// (nvbio/io/fmindex/fmindex.h:294-362), the .pac / .wpac genome readers (io/sequence/sequence_pac.cpp:94-190).
//   <prefix>.bwt/.rbwt  [uint32 primary][uint32 cumFreq x4][uint32 BWT words, 2-bit big-endian, '$' removed]
//   <prefix>.sa/.rsa    [primary][cumFreq x4][sa_intv][seq_length][uint32 ssa[1..]]
//   <prefix>.wpac       [uint64 seq_length][uint32 words, 2-bit big-endian]
//   <prefix>.pac        BWA byte-packed genome + trailing length byte
// File parsing is host work; the occurrence table is built on the device (nvbio_hip_build_bwt_occ).
#pragma once
#include <cstdio>
#include <string>
#include <vector>
#include "fmindex.h"

namespace nvbio {
namespace io {

struct FMIndexDataCore
{
    static const uint32 FORWARD = 0x02;
    static const uint32 REVERSE = 0x04;
    static const uint32 SA      = 0x10;
    static const uint32 BWT_BITS = 2u, BWT_SYMBOLS_PER_WORD = 16u, OCC_INT = 64u, SA_INT = 16u;
};

namespace priv {
inline bool read_words(FILE* f, uint32* dst, size_t n) { return n == 0 || fread(dst, sizeof(uint32), n, f) == n; }

/// load_bwt (fmindex_impl.cu:119-178): words padded to a multiple of 4 (and to whole 64-symbol blocks), slack zeroed
inline bool load_bwt(const char* name, std::vector<uint32>& words, uint32& seq_length, uint32& primary)
{
    FILE* f = fopen(name, "rb");
    if (!f) { fprintf(stderr, "warning: unable to open bwt \"%s\"\n", name); return false; }
    uint32 hdr[5];
    if (!read_words(f, hdr, 5)) { fprintf(stderr, "error: failed reading bwt \"%s\"\n", name); fclose(f); return false; }
    primary = hdr[0]; seq_length = hdr[4];                    // the last cumulative frequency is the total length
    const uint32 seq_words = (seq_length + 15u) / 16u;
    words.assign(size_t(4u) * ((seq_length + 63u) / 64u), 0u);
    if (!read_words(f, words.data(), seq_words)) { fprintf(stderr, "error: failed reading bwt \"%s\"\n", name); fclose(f); return false; }
    fclose(f);
    return true;
}
/// load_sa (fmindex_impl.cu:180-262): a mismatching file is skipped, not an error
inline bool load_sa(const char* name, std::vector<uint32>& ssa, const uint32 seq_length, const uint32 primary, const uint32 sa_int)
{
    FILE* f = fopen(name, "rb");
    if (!f) return false;
    uint32 hdr[7];
    bool ok = read_words(f, hdr, 7);
    if (ok && hdr[0] != primary)    { fprintf(stderr, "SA file mismatch \"%s\"\n  expected primary %u, got %u\n", name, primary, hdr[0]); ok = false; }
    if (ok && hdr[5] != sa_int)     { fprintf(stderr, "unsupported SA interval (found %u, expected %u)\n", hdr[5], sa_int); ok = false; }
    if (ok && hdr[6] != seq_length) { fprintf(stderr, "SA file mismatch \"%s\"\n  expected length %u, got %u\n", name, seq_length, hdr[6]); ok = false; }
    if (ok) {
        const uint32 sa_size = (seq_length + sa_int) / sa_int;
        ssa.assign(sa_size, 0u);
        ssa[0] = uint32(-1);
        ok = read_words(f, ssa.data() + 1, sa_size - 1u);
        if (!ok) { fprintf(stderr, "error: failed reading SSA \"%s\"\n", name); ssa.clear(); }
    }
    fclose(f);
    return ok;
}
} // namespace priv

/// The host side: the raw BWT words and sampled SA of the forward and/or reverse index.  (The reference's
/// host object also holds the interleaved bwt|occ table; here that table only ever exists on the device.)
struct FMIndexDataHost : public FMIndexDataCore
{
    FMIndexDataHost() : m_flags(0), m_seq_length(0), m_primary(0), m_rprimary(0) {}

    /// load <prefix>.bwt/.sa and/or <prefix>.rbwt/.rsa; returns 1 on success, 0 on failure (fmindex_impl.cu:340-480)
    int load(const char* genome_prefix, const uint32 flags = FORWARD | REVERSE | SA)
    {
        m_flags = flags;
        const std::string p(genome_prefix);
        if (flags & FORWARD) { if (!priv::load_bwt((p + ".bwt").c_str(),  m_bwt,  m_seq_length, m_primary))  return 0; }
        if (flags & REVERSE) { if (!priv::load_bwt((p + ".rbwt").c_str(), m_rbwt, m_seq_length, m_rprimary)) return 0; }
        if (flags & SA) {
            if (flags & FORWARD) priv::load_sa((p + ".sa").c_str(),  m_ssa,  m_seq_length, m_primary,  SA_INT);
            if (flags & REVERSE) priv::load_sa((p + ".rsa").c_str(), m_rssa, m_seq_length, m_rprimary, SA_INT);
        }
        return 1;
    }
    uint32 genome_length() const { return m_seq_length; }
    bool   has_ssa()  const { return !m_ssa.empty(); }
    bool   has_rssa() const { return !m_rssa.empty(); }

    uint32 m_flags, m_seq_length, m_primary, m_rprimary;
    std::vector<uint32> m_bwt, m_rbwt, m_ssa, m_rssa;
};

/// The device side: interleaved bwt|occ records (built on the device) + SSA, as fm_index_device views
struct FMIndexDataDevice : public FMIndexDataCore
{
    /// hbm_rich (the default on this hardware): index() / rindex() come back in the form the device's free memory allows (enrich())
    FMIndexDataDevice(const FMIndexDataHost& host, const uint32 flags = FORWARD | REVERSE, const bool hbm_rich = true) : m_seq_length(host.m_seq_length)
    {
        for (int i = 0; i < 5; ++i) m_L2[i] = m_rL2[i] = 0;
        if ((flags & FORWARD) && !host.m_bwt.empty())  upload(host.m_bwt,  host.m_ssa,  host.m_primary,  m_bwt_occ,  m_ssa,  m_L2,  m_index);
        if ((flags & REVERSE) && !host.m_rbwt.empty()) upload(host.m_rbwt, host.m_rssa, host.m_rprimary, m_rbwt_occ, m_rssa, m_rL2, m_rindex);
        if (hbm_rich) enrich();
    }
    FMIndexDataDevice(const FMIndexDataDevice&) = delete;              // index() points into this object's arrays
    FMIndexDataDevice& operator=(const FMIndexDataDevice&) = delete;
    /// the forward / reverse index as the drivers should use it: the HBM-rich form once enrich() has run, else as loaded
    const fm_index_device& index()  const { return m_rich ? m_hbm.index  : m_index; }
    const fm_index_device& rindex() const { return m_rich ? m_rhbm.index : m_rindex; }
    /// the indices exactly as loaded (the reference's layout: bwt|occ records + SA sampled every 16 rows)
    const fm_index_device& lean_index()  const { return m_index; }
    const fm_index_device& lean_rindex() const { return m_rindex; }
    uint32 genome_length() const { return m_seq_length; }
    /// build what this device's free memory allows on top of the loaded arrays (fm_index_hbm: line-native records, 12-mer table, a
    /// denser suffix array; NVBIO_HIP_INDEX=lean turns it off).  Results of every stage stay bit-identical.
    void enrich(void* stream = nullptr)
    {
        if (m_index.m.bwt_occ)  m_hbm.build(m_index, 0, stream);
        if (m_rindex.m.bwt_occ) m_rhbm.build(m_rindex, 0, stream);
        m_rich = true;
    }
    std::string description() const { return m_rich ? m_hbm.description() : std::string("reference_layout sa_int=16"); }

private:
    void upload(const std::vector<uint32>& bwt, const std::vector<uint32>& ssa, const uint32 primary,
                hip::device_vector<uint32>& d_bwt_occ, hip::device_vector<uint32>& d_ssa, uint32* L2, fm_index_device& out)
    {
        hip::device_vector<uint32> d_bwt(bwt);
        d_bwt_occ.resize(bwt.size() * 2u);
        build_bwt_occ(m_seq_length, d_bwt.data(), d_bwt_occ.data(), L2);
        if (!ssa.empty()) d_ssa.assign(ssa.data(), ssa.size());
        out = fm_index_device(m_seq_length, primary, L2, d_bwt_occ.data(), ssa.empty() ? nullptr : d_ssa.data(), SA_INT);
    }
    uint32 m_seq_length, m_L2[5], m_rL2[5];
    hip::device_vector<uint32> m_bwt_occ, m_rbwt_occ, m_ssa, m_rssa;
    fm_index_device m_index, m_rindex;
    fm_index_hbm    m_hbm, m_rhbm;
    bool            m_rich = false;
};

/// genome loaders: (seq_length, 2-bit big-endian words); <prefix>.wpac if present, else <prefix>.pac
inline bool load_wpac(const char* name, uint32& seq_length, std::vector<uint32>& words)
{
    FILE* f = fopen(name, "rb");
    if (!f) return false;
    uint64 len = 0;
    bool ok = fread(&len, sizeof(len), 1, f) == 1;
    seq_length = uint32(len);
    words.assign((size_t(seq_length) + 15u) / 16u, 0u);
    ok = ok && priv::read_words(f, words.data(), words.size());
    fclose(f);
    return ok;
}
inline bool load_pac(const char* name, uint32& seq_length, std::vector<uint32>& words)
{
    FILE* f = fopen(name, "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    const long size = ftell(f);
    if (size < 2) { fclose(f); return false; }
    std::vector<uint8> raw(size_t(size), 0);
    fseek(f, 0, SEEK_SET);
    const bool ok = fread(raw.data(), 1, raw.size(), f) == raw.size();
    fclose(f);
    if (!ok) return false;
    seq_length = uint32(size - 2) * 4u + raw[size - 1];               // sequence_pac.cpp:150-158
    words.assign((size_t(seq_length) + 15u) / 16u, 0u);
    for (uint32 b = 0; b < (seq_length + 3u) / 4u; ++b)               // byte b = symbols 4b..4b+3, top bits first
        words[b >> 2] |= uint32(raw[b]) << (24u - 8u * (b & 3u));
    return true;
}
inline bool load_genome(const char* prefix, uint32& seq_length, std::vector<uint32>& words)
{
    const std::string p(prefix);
    return load_wpac((p + ".wpac").c_str(), seq_length, words) || load_pac((p + ".pac").c_str(), seq_length, words);
}

} // namespace io
} // namespace nvbio
