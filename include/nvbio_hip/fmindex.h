// nvbio_hip/fmindex.h -- nvbio::fm_index batch interface on MI355X.
//
// Mirrors  fm_index (nvbio/fmindex/fmindex.h:341-387) as a POD over the production interleaved
// bwt|occ layout, the free functions rank / rank4 / match / locate / locate_ssa_iterator /
// lookup_ssa_iterator (nvbio/fmindex/fmindex_inl.h) in their batched form, and
// FMIndexFilterDevice (nvbio/fmindex/filter.h:139-203) with the reference's member names.
#pragma once
#include <stdlib.h>
#include <string>
#include "strings.h"

namespace nvbio {

/// fm_index< rank_dictionary<2,64,...uint4 interleaved>, SSA_index_multiple_context<SA_INT>, const uint32* >
struct fm_index_device
{
    typedef uint32 index_type;
    typedef uint2  range_type;
    nvbio_hip_fmindex m;

    fm_index_device() { m.length = m.primary = 0; m.sa_int = 16; m.bwt_occ = nullptr; m.ssa = nullptr; m.ktab = nullptr; m.ktab_k = 0; m._pad = 0; m.dimer = nullptr; m.trimer = nullptr; m.dimer_p1 = m.dimer_fill1 = 0; for (int i = 0; i < 5; ++i) m.L2[i] = 0; for (int i = 0; i < 4; ++i) m.dimer_S[i] = m.dimer_T[i] = 0; }
    fm_index_device(uint32 length, uint32 primary, const uint32* L2, const uint32* bwt_occ, const uint32* ssa, uint32 sa_int = 16)
    { m.length = length; m.primary = primary; for (int i = 0; i < 5; ++i) m.L2[i] = L2[i]; m.bwt_occ = bwt_occ; m.ssa = ssa; m.sa_int = sa_int; m.ktab = nullptr; m.ktab_k = 0; m._pad = 0; m.dimer = nullptr; m.trimer = nullptr; m.dimer_p1 = m.dimer_fill1 = 0; for (int i = 0; i < 4; ++i) m.dimer_S[i] = m.dimer_T[i] = 0; }

    /// attach the optional k-mer table built by build_ktab() (caller keeps the storage alive)
    void set_ktab(const uint32* ktab, uint32 k) { m.ktab = ktab; m.ktab_k = k; }
    // attach the line-native two-symbol index built by nvbio_hip_fm_build_dimer_index (NULL detaches)
    int  set_dimer(const uint32* dimer, void* stream = nullptr) { return nvbio_hip_fm_attach_dimer_index(&m, dimer, stream); }
    /// attach the three-symbol arrays built by nvbio_hip_fm_build_trimer_index (validated against this index; NULL detaches)
    int  set_trimer(const uint32* trimer, void* stream = nullptr) { return nvbio_hip_fm_attach_trimer_index(&m, trimer, stream); }

    index_type length() const { return m.length; }
    index_type primary() const { return m.primary; }
    index_type count(const uint32 c) const { return m.L2[c + 1] - m.L2[c]; }
    index_type L2(const uint32 c) const { return m.L2[c]; }
    uint32     symbol_count() const { return 4u; }
    uint32     symbol_size() const { return 2u; }
};

/// out[i] = rank(fmi, k[i], c[i])                                  fmindex_inl.h:36-57
inline void rank(const fm_index_device& fmi, uint32 n, const uint32* k, const uint8* c, uint32* out, void* stream = nullptr)
{ hip_check(nvbio_hip_fm_rank(&fmi.m, k, c, n, out, stream), "nvbio_hip_fm_rank"); }
/// out[i] = rank4(fmi, k[i])                                       fmindex_inl.h:111-135
inline void rank4(const fm_index_device& fmi, uint32 n, const uint32* k, uint4* out, void* stream = nullptr)
{ hip_check(nvbio_hip_fm_rank4(&fmi.m, k, n, reinterpret_cast<uint32*>(out), stream), "nvbio_hip_fm_rank4"); }
/// out[i] = rank(fmi, range[i], c[i])                              fmindex_inl.h:66-99
inline void rank(const fm_index_device& fmi, uint32 n, const uint2* range, const uint8* c, uint2* out, void* stream = nullptr)
{ hip_check(nvbio_hip_fm_rank_range(&fmi.m, reinterpret_cast<const uint32*>(range), c, n, reinterpret_cast<uint32*>(out), stream), "nvbio_hip_fm_rank_range"); }
/// ranges[i] = match(fmi, string_set[i], length(string_set[i]))    fmindex_inl.h:280-341
template <typename string_set_type>
inline void match(const fm_index_device& fmi, const string_set_type& string_set, uint2* ranges, void* stream = nullptr)
{ const nvbio_hip_string_set s = string_set.abi(); hip_check(nvbio_hip_fm_match(&fmi.m, &s, string_set.size(), reinterpret_cast<uint32*>(ranges), stream), "nvbio_hip_fm_match"); }
/// builds the k-mer table accelerator of match(): 2*4^k words into d_ktab
inline void build_ktab(const fm_index_device& fmi, uint32 k, uint32* d_ktab, void* stream = nullptr)
{ hip_check(nvbio_hip_fm_build_ktab(&fmi.m, k, d_ktab, stream), "nvbio_hip_fm_build_ktab"); }
/// out[i] = locate(fmi, rows[i])                                   fmindex_inl.h:466-501
inline void locate(const fm_index_device& fmi, uint32 n, const uint32* rows, uint32* out, void* stream = nullptr)
{ hip_check(nvbio_hip_fm_locate(&fmi.m, rows, n, out, stream), "nvbio_hip_fm_locate"); }
inline void locate_ssa_iterator(const fm_index_device& fmi, uint32 n, const uint32* rows, uint2* out, void* stream = nullptr)
{ hip_check(nvbio_hip_fm_locate_ssa_iterator(&fmi.m, rows, n, reinterpret_cast<uint32*>(out), stream), "nvbio_hip_fm_locate_ssa_iterator"); }
inline void lookup_ssa_iterator(const fm_index_device& fmi, uint32 n, const uint2* it, uint32* out, void* stream = nullptr)
{ hip_check(nvbio_hip_fm_lookup_ssa_iterator(&fmi.m, reinterpret_cast<const uint32*>(it), n, out, stream), "nvbio_hip_fm_lookup_ssa_iterator"); }

/// The index an MI355X's HBM is there for, built on the device from a loaded index and owning the extra arrays: the line-native two-symbol
/// records (one 128-byte line per backward-search step pair), a denser -- when it fits, the whole -- suffix array (12 GB at 3 Gbp: locate
/// without an LF walk) and the match range of every k-mer for the largest k that fits (k = 16: 34 GB, 6 search steps left of a 22-mer's 22).  Every result is bit-identical to the lean index's; only the time
/// changes (seeding + locate about 2x).  What is built follows the memory that is free when build() runs:
/// `budget_bytes` (0 = 35 % of the device's free memory; the drivers' per-batch workspace and a reverse index want their share too).
/// NVBIO_HIP_INDEX=lean|line_native|rich overrides (lean: nothing built; line_native: the two-symbol records only; rich: everything, the
/// whole suffix array, fail if it does not fit).
struct fm_index_hbm
{
    hip::device_vector<uint32> dimer, ktab, ssa;
    fm_index_device            index;          ///< the view to hand to map / locate / the drivers (points into `base`'s and this object's arrays)
    uint32                     ktab_k, sa_int;
    bool                       line_native;
    fm_index_hbm() : ktab_k(0), sa_int(0), line_native(false) {}

    void build(const fm_index_device& base, uint64 budget_bytes = 0, void* stream = nullptr)
    {
        index = base; ktab_k = 0; sa_int = base.m.sa_int; line_native = false;
        const char* env = getenv("NVBIO_HIP_INDEX");
        const std::string policy = env ? env : "auto";
        if (policy == "lean" || base.m.bwt_occ == nullptr) return;
        uint64 free_b = 0, total_b = 0, idle_b = 0;
        hip_check(nvbio_hip_device_mem_info(&free_b, &total_b, &idle_b), "nvbio_hip_device_mem_info");
        uint64 budget = budget_bytes ? budget_bytes : uint64(double(free_b + idle_b) * 0.35);
        if (policy == "rich") budget = ~uint64(0);
        const uint32 n = base.m.length;
        // the two-symbol records
        {
            const uint64 bytes = nvbio_hip_fm_dimer_index_bytes(n), temp_bytes = nvbio_hip_fm_build_dimer_index_temp_bytes(n);
            if (bytes + temp_bytes <= budget)
            {
                dimer.resize((bytes + 3u) / 4u);
                hip::device_vector<uint8> temp(temp_bytes);
                nvbio_hip_fmindex plain = base.m; plain.dimer = nullptr; plain.trimer = nullptr; plain.ktab = nullptr; plain.ktab_k = 0;
                hip_check(nvbio_hip_fm_build_dimer_index(&plain, dimer.data(), temp.data(), temp_bytes, stream), "nvbio_hip_fm_build_dimer_index");
                hip_check(index.set_dimer(dimer.data(), stream), "nvbio_hip_fm_attach_dimer_index");
                hip::synchronize(stream);
                line_native = true; budget -= bytes;
            }
        }
        if (policy == "line_native") return;
        // the densest suffix array that fits (locate without an LF walk at 1): the most valuable of the extras, so it is sized first
        if (base.m.ssa)
            for (uint32 s = 1; s < base.m.sa_int; s *= 2)
            {
                const uint64 entries = nvbio_hip_fm_dense_ssa_entries(n, s);
                if (entries * 4u > budget) { if (policy == "rich") throw hip_error("fm_index_hbm: the whole suffix array does not fit", 2); continue; }
                ssa.resize(entries);
                hip_check(nvbio_hip_fm_build_dense_ssa(&index.m, s, ssa.data(), stream), "nvbio_hip_fm_build_dense_ssa");
                hip::synchronize(stream);
                index.m.ssa = ssa.data(); index.m.sa_int = s; sa_int = s; budget -= entries * 4u;
                break;
            }
        // the match range of every k-mer, the largest k that fits and that the text can fill (4^k <= ~4 x length): 16 -> 34 GB, 15 -> 8.6 GB, 12 -> 128 MB
        {
            uint32 kmax = 8u;
            while (kmax < 16u && (uint64(1) << (2u * kmax)) < uint64(n)) ++kmax;
            for (uint32 k = kmax; k >= 8u; --k)
            {
                const uint64 bytes = (uint64(2) << (2u * k)) * 4u;
                if (bytes > budget) continue;
                ktab.resize(size_t(2) << (2u * k));
                build_ktab(index, k, ktab.data(), stream);
                index.set_ktab(ktab.data(), k); ktab_k = k; budget -= bytes;
                break;
            }
        }
    }
    /// "line_native ktab12 sa_int=1": what the index a run used was
    std::string description() const
    {
        return std::string(line_native ? "line_native" : "reference_layout") + (ktab_k ? " ktab" + std::to_string(ktab_k) : std::string()) + " sa_int=" + std::to_string(sa_int);
    }
};

/// FMIndexFilter<device_tag, fm_index>  (filter.h:139-203)
template <typename system_tag, typename fm_index_type> struct FMIndexFilter {};

template <typename fm_index_type>
struct FMIndexFilter<device_tag, fm_index_type>
{
    typedef device_tag    system_tag;
    typedef fm_index_type index_type;
    typedef uint32        coord_type;
    typedef uint2         range_type;
    typedef uint2         hit_type;

    FMIndexFilter() : m_n_queries(0), m_n_occurrences(0) {}

    /// rank all strings of the set; returns the total number of hits
    template <typename string_set_type>
    uint64 rank(const fm_index_type& index, const string_set_type& string_set)
    {
        m_n_queries = string_set.size();
        m_index = index;
        m_ranges.resize(m_n_queries); m_slots.resize(m_n_queries);
        if (m_n_queries == 0) { m_n_occurrences = 0; return 0; }
        const uint64 tb = nvbio_hip_fm_filter_temp_bytes(m_n_queries);
        if (d_temp_storage.size() < tb) d_temp_storage.resize(tb);
        const nvbio_hip_string_set s = string_set.abi();
        hip_check(nvbio_hip_fm_filter_rank(&m_index.m, &s, m_n_queries, reinterpret_cast<uint32*>(m_ranges.data()), m_slots.data(),
                                           d_temp_storage.data(), d_temp_storage.size(), nullptr), "nvbio_hip_fm_filter_rank");
        hip_check(nvbio_hip_memcpy(&m_n_occurrences, m_slots.data() + (m_n_queries - 1), 8, 2, nullptr), "nvbio_hip_memcpy");
        return m_n_occurrences;
    }
    /// enumerate the hits [begin,end) as (text position, string id) pairs into device memory
    void locate(const uint64 begin, const uint64 end, hit_type* hits)
    {
        hip_check(nvbio_hip_fm_filter_locate(&m_index.m, reinterpret_cast<const uint32*>(m_ranges.data()), m_slots.data(), m_n_queries,
                                             begin, end, reinterpret_cast<uint32*>(hits), nullptr), "nvbio_hip_fm_filter_locate");
    }
    uint64            n_hits() const { return m_n_occurrences; }
    const range_type* ranges() const { return m_ranges.data(); }
    const uint64*     ranks()  const { return m_slots.data(); }

    uint32                          m_n_queries;
    index_type                      m_index;
    uint64                          m_n_occurrences;
    hip::device_vector<range_type>  m_ranges;
    hip::device_vector<uint64>      m_slots;
    hip::device_vector<uint8>       d_temp_storage;
};
typedef FMIndexFilter<device_tag, fm_index_device> FMIndexFilterDevice;

/// device build_occurrence_table<2,64> + interleave (rank_dictionary_inl.h:42-77, fmindex_impl.cu:305-327)
inline void build_bwt_occ(uint32 n, const uint32* d_bwt_words, uint32* d_bwt_occ, uint32 L2_host[5])
{
    const uint64 tb = nvbio_hip_build_bwt_occ_temp_bytes(n);
    hip::device_vector<uint8> temp(tb);
    hip::device_vector<uint32> dL2(5);
    hip_check(nvbio_hip_build_bwt_occ(n, d_bwt_words, d_bwt_occ, dL2.data(), temp.data(), tb, nullptr), "nvbio_hip_build_bwt_occ");
    const std::vector<uint32> h = dL2.to_host();
    for (int i = 0; i < 5; ++i) L2_host[i] = h[i];
}

} // namespace nvbio
