// nvbio_hip/sam.h -- SAM records of the single-end best-mapping driver's results (host side, after the batch came back):
// the mandatory fields and the tags SamOutput writes for an aligned read (nvbio/io/output/output_sam.cpp:316-366: NM, AS, XM, XO,
// XG, MD), its conventions kept -- an unaligned read is flag 4 with '*' fields and the read as loaded; an alignment that runs over the
// end of its reference sequence is printed in full with the UNMAPPED flag and mapping quality 0 (:454-462); MD / XM / XO / XG come
// from the byte-coded MDS finish_alignment leaves (generate_md_string :233-314, its uint8 run counter included).
// Records are formatted by all OpenMP threads into per-thread buffers and written in read order.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include <algorithm>
#if defined(_OPENMP)
#include <omp.h>
#endif

namespace nvbio {
namespace io {

/// the reference's sequences: names and n + 1 offsets into the concatenated genome (<prefix>.ann, or one sequence)
struct SamReference
{
    std::vector<std::string> names;
    std::vector<uint64_t>    index;
    uint32_t sequence_of(const uint64_t pos) const
    {
        const uint32_t k = uint32_t(std::upper_bound(index.begin(), index.end(), pos) - index.begin());
        return k == 0u ? 0u : std::min<uint32_t>(k - 1u, uint32_t(names.size()) - 1u);
    }
    std::string header(const char* program = "nvbio_amd") const
    {
        std::string h = "@HD\tVN:1.0\tSO:unsorted\n";
        for (size_t k = 0; k < names.size(); ++k) h += "@SQ\tSN:" + names[k] + "\tLN:" + std::to_string(index[k + 1] - index[k]) + "\n";
        h += std::string("@PG\tID:") + program + "\tPN:" + program + "\n";
        return h;
    }
};

/// what Aligner::best_approx (finish = true) returns for a batch, on the host
struct SamBatchSE
{
    uint32_t        n;
    const char*     names;        const uint32_t* names_index;      ///< n + 1 offsets, each name 0-terminated inside its slot
    const uint8_t*  symbols;      const uint64_t* read_index;       ///< forward reads, one symbol (0..4) per byte; n + 1 offsets
    const uint8_t*  quals;                                          ///< phred, same offsets
    const uint64_t* best;                                           ///< io::Alignment words of the best alignment: low = flags / score / ed, high = position
    const uint8_t*  mapq;
    const uint16_t* cigar;        uint32_t cigar_stride;            ///< io::Cigar words, end first
    const uint32_t* cigar_len;
    const uint32_t* source;                                         ///< (x, y) per read: where in the traceback window the alignment starts
    const uint8_t*  mds;          uint32_t mds_stride;
};

namespace priv {
inline void put_uint(std::string& s, uint64_t v) { char b[24]; int k = 24; do { b[--k] = char('0' + v % 10u); v /= 10u; } while (v); s.append(b + k, size_t(24 - k)); }
inline void put_int(std::string& s, int64_t v) { if (v < 0) { s.push_back('-'); put_uint(s, uint64_t(-v)); } else put_uint(s, uint64_t(v)); }

/// MD:Z text and the mismatch / gap-open / gap-extension counters of one MDS
/// `stride`: the bytes the row owns.  The 16-bit length in mds[0..1] is what finish_alignment meant to write; a string longer than the row was cut
/// there, so the walk stops at min(length, stride) and no operand is read outside the row (a token cut by the row's end ends the string).  Returns false
/// when the row was too short for its string: the caller prints MD:Z:* for that record rather than half a string.
inline bool md_string(const uint8_t* mds, const uint32_t stride, std::string& md, int32_t& mm, int32_t& gapo, int32_t& gape)      // (int32 counters, output_sam.h:83-85)
{
    static const char dna[] = "ACGTN";
    const uint32_t len = stride >= 2u ? (uint32_t(mds[0]) | (uint32_t(mds[1]) << 8)) : 0u;
    const uint32_t n = std::min(len, stride);
    mm = gapo = gape = 0;
    uint32_t i = 2u;
    while (i < n)
    {
        const uint32_t op = mds[i++];
        if (i >= stride) break;                                                       // an operand would lie outside the row: cut (inside the row the walk is the reference's, the byte past the length included)
        if (op == 0u)
        {
            uint32_t run = mds[i++];
            while (i < n && mds[i] == 0u) { run = (run + mds[i]) & 0xFFu; ++i; }      // the reference's byte counter walks over the next MATCH token's op byte
            put_uint(md, run);
        }
        else if (op == 1u) { md.push_back(dna[std::min<uint32_t>(mds[i], 4u)]); ++i; ++mm; }
        else if (op == 2u) { const uint32_t l = mds[i]; i += 1u + l; ++gapo; gape += int32_t(l) - 1; }
        else if (op == 3u)
        {
            const uint32_t l = mds[i++];
            md.push_back('^');
            for (uint32_t k = 0; k < l && i + k < stride; ++k) md.push_back(dna[std::min<uint32_t>(mds[i + k], 4u)]);
            md.push_back('0');
            i += l; ++gapo; gape += int32_t(l) - 1;
        }
    }
    return len <= stride;
}

inline void se_record(const SamBatchSE& b, const SamReference& ref, const uint32_t i, const uint32_t extra_flags, std::string& out, std::string& md)
{
    static const char dna[] = "ACGTN";
    const uint32_t w   = uint32_t(b.best[i] & 0xFFFFFFFFull);
    const uint32_t pos = uint32_t(b.best[i] >> 32);
    const uint8_t* seq = b.symbols + b.read_index[i];
    const uint8_t* ql  = b.quals + b.read_index[i];
    const uint32_t L   = uint32_t(b.read_index[i + 1] - b.read_index[i]);
    out.append(b.names + b.names_index[i]);
    if (pos == 0xFFFFFFFFu)
    {
        out.append("\t4\t*\t0\t0\t*\t*\t0\t0\t");                 // (UNMAPPED alone, output_sam.cpp:427)
        for (uint32_t k = 0; k < L; ++k) out.push_back(dna[std::min<uint32_t>(seq[k], 4u)]);
        out.push_back('\t');
        for (uint32_t k = 0; k < L; ++k) out.push_back(char(ql[k] + 33));
        out.push_back('\n');
        return;
    }
    const bool     rc    = (w >> 28) & 1u;
    const int32_t  score = int32_t((w >> 1) & 0x1FFFFu) * ((w & 1u) ? -1 : 1);
    const uint32_t ed    = (w >> 18) & 0x3FFu;
    const uint16_t* cg   = b.cigar + uint64_t(i) * b.cigar_stride;
    const uint32_t  cl   = std::min(b.cigar_len[i], b.cigar_stride);          // (a CIGAR longer than its row was cut at the row's end)
    uint64_t ref_len = 0;
    for (uint32_t k = 0; k < cl; ++k) { const uint32_t t = cg[k] & 3u; if (t == 0u || t == 2u) ref_len += cg[k] >> 2; }
    const uint64_t at   = uint64_t(pos) + b.source[2u * i];
    const uint32_t sq   = ref.sequence_of(at);
    const bool     over = at + ref_len > ref.index[sq + 1];
    out.push_back('\t'); put_uint(out, (rc ? 16u : 0u) | (over ? 4u : 0u) | extra_flags);
    out.push_back('\t'); out.append(ref.names[sq]);
    out.push_back('\t'); put_uint(out, at - ref.index[sq] + 1u);
    out.push_back('\t'); put_uint(out, over ? 0u : b.mapq[i]);
    out.push_back('\t');
    if (cl == 0u) out.push_back('*');
    for (uint32_t k = cl; k-- > 0u;) { put_uint(out, cg[k] >> 2); out.push_back("MIDS"[cg[k] & 3u]); }
    out.append("\t*\t0\t0\t");
    if (rc) { for (uint32_t k = L; k-- > 0u;) out.push_back(seq[k] < 4u ? dna[3u - seq[k]] : 'N'); }
    else    { for (uint32_t k = 0; k < L; ++k) out.push_back(dna[std::min<uint32_t>(seq[k], 4u)]); }
    out.push_back('\t');
    if (rc) { for (uint32_t k = L; k-- > 0u;) out.push_back(char(ql[k] + 33)); }
    else    { for (uint32_t k = 0; k < L; ++k) out.push_back(char(ql[k] + 33)); }
    int32_t mm, gapo, gape;
    md.clear();
    const bool whole = md_string(b.mds + uint64_t(i) * b.mds_stride, b.mds_stride, md, mm, gapo, gape);
    out.append("\tNM:i:"); put_uint(out, ed);
    out.append("\tAS:i:"); put_int(out, score);
    out.append("\tXM:i:"); put_int(out, mm);
    out.append("\tXO:i:"); put_int(out, gapo);
    out.append("\tXG:i:"); put_int(out, gape);
    out.append("\tMD:Z:"); if (md.empty() || !whole) out.push_back('*'); else out.append(md);
    out.push_back('\n');
}

/// one slot set of the paired-end driver (Aligner::best_approx over a PairedReadBatch, finish = true): the anchor slots or the opposite slots
struct SamSlotsPE
{
    const uint64_t* best;  const uint8_t* mapq;
    const uint16_t* cigar; const uint32_t* cigar_len; const uint32_t* source;
    const uint8_t*  mds;
};
} // namespace priv

/// what the paired-end driver returns for a batch of equal-length mates, on the host: two slot sets (anchor, opposite), each naming its mate in the
/// alignment word; reads of mate m: symbols[m] / quals[m], one row of `len` bytes per pair; one name per pair
struct SamBatchPE
{
    uint32_t        n, len;
    const char*     names;        const uint32_t* names_index;
    const uint8_t*  symbols[2];   const uint8_t*  quals[2];
    priv::SamSlotsPE slot[2];
    uint32_t        cigar_stride, mds_stride;
};

namespace priv {
/// the two records of pair i, slot 0's first (SamOutput's paired fields, output_sam.cpp:372-520): flags READ_1 / READ_2 by the alignment's mate,
/// REVERSE, PAIRED, PROPER_PAIR when the mate's alignment is concordant, MATE_UNMAPPED, MATE_REVERSE; RNEXT '=', PNEXT, TLEN = span of the two
/// alignments, negative for the rightmost one; an unaligned read carries the UNMAPPED flag alone, printed on the strand its word names
inline void pe_records(const SamBatchPE& b, const SamReference& ref, const uint32_t i, std::string& out, std::string& md)
{
    static const char dna[] = "ACGTN";
    struct F { bool ok; uint32_t w; uint64_t pos, ref_len; bool rc, concordant; uint32_t mate, cl; };
    F f[2];
    for (int k = 0; k < 2; ++k)
    {
        const uint32_t w = uint32_t(b.slot[k].best[i] & 0xFFFFFFFFull), p = uint32_t(b.slot[k].best[i] >> 32);
        f[k].w = w; f[k].ok = p != 0xFFFFFFFFu; f[k].rc = (w >> 28) & 1u; f[k].mate = (w >> 29) & 1u;
        f[k].concordant = ((w >> 30) & 1u) && !((w >> 31) & 1u);
        f[k].cl = std::min(b.slot[k].cigar_len[i], b.cigar_stride); f[k].ref_len = 0; f[k].pos = 0;
        if (!f[k].ok) continue;
        const uint16_t* cg = b.slot[k].cigar + uint64_t(i) * b.cigar_stride;
        for (uint32_t c = 0; c < f[k].cl; ++c) { const uint32_t t = cg[c] & 3u; if (t == 0u || t == 2u) f[k].ref_len += cg[c] >> 2; }
        f[k].pos = uint64_t(p) + b.slot[k].source[2u * i];
    }
    const uint32_t L = b.len;
    for (int k = 0; k < 2; ++k)
    {
        const F& a = f[k]; const F& m = f[1 - k];
        const uint8_t* seq = b.symbols[a.mate] + uint64_t(i) * L;
        const uint8_t* ql  = b.quals[a.mate] + uint64_t(i) * L;
        out.append(b.names + b.names_index[i]);
        auto put_read = [&](const bool rc) {
            if (rc) { for (uint32_t x = L; x-- > 0u;) out.push_back(seq[x] < 4u ? dna[3u - seq[x]] : 'N'); }
            else    { for (uint32_t x = 0; x < L; ++x) out.push_back(dna[std::min<uint32_t>(seq[x], 4u)]); }
            out.push_back('\t');
            if (rc) { for (uint32_t x = L; x-- > 0u;) out.push_back(char(ql[x] + 33)); }
            else    { for (uint32_t x = 0; x < L; ++x) out.push_back(char(ql[x] + 33)); }
        };
        if (!a.ok)
        {
            out.append("\t4\t*\t0\t0\t*\t*\t0\t0\t");
            put_read(a.rc);
            out.push_back('\n');
            continue;
        }
        uint32_t flags = (a.mate ? 0x80u : 0x40u) | (a.rc ? 0x10u : 0u) | 0x1u;
        if (m.ok && m.concordant) flags |= 0x2u;
        if (!m.ok) flags |= 0x8u;
        if (m.rc) flags |= 0x20u;
        const uint32_t sq = ref.sequence_of(a.pos);
        const bool over = a.pos + a.ref_len > ref.index[sq + 1];
        if (over) flags |= 0x4u;
        out.push_back('\t'); put_uint(out, flags);
        out.push_back('\t'); out.append(ref.names[sq]);
        out.push_back('\t'); put_uint(out, a.pos - ref.index[sq] + 1u);
        out.push_back('\t'); put_uint(out, over ? 0u : b.slot[k].mapq[i]);
        out.push_back('\t');
        const uint16_t* cg = b.slot[k].cigar + uint64_t(i) * b.cigar_stride;
        if (a.cl == 0u) out.push_back('*');
        for (uint32_t c = a.cl; c-- > 0u;) { put_uint(out, cg[c] >> 2); out.push_back("MIDS"[cg[c] & 3u]); }
        out.push_back('\t');
        if (m.ok)
        {
            const uint32_t msq = ref.sequence_of(m.pos);
            const bool same = msq == sq;
            if (same) out.push_back('='); else out.append(ref.names[msq]);
            out.push_back('\t'); put_uint(out, m.pos - ref.index[msq] + 1u);
            int64_t tlen = int64_t(std::max(m.pos + m.ref_len, a.pos + a.ref_len)) - int64_t(std::min(m.pos, a.pos));
            if (m.pos < a.pos) tlen = -tlen;
            if (!same) tlen = 0;
            out.push_back('\t'); put_int(out, tlen);
        }
        else
        {
            out.append("=\t"); put_uint(out, a.pos - ref.index[sq] + 1u); out.append("\t0");
        }
        out.push_back('\t');
        put_read(a.rc);
        const int32_t  score = int32_t((a.w >> 1) & 0x1FFFFu) * ((a.w & 1u) ? -1 : 1);
        const uint32_t ed    = (a.w >> 18) & 0x3FFu;
        int32_t mm, gapo, gape;
        md.clear();
        const bool whole = md_string(b.slot[k].mds + uint64_t(i) * b.mds_stride, b.mds_stride, md, mm, gapo, gape);
        out.append("\tNM:i:"); put_uint(out, ed);
        out.append("\tAS:i:"); put_int(out, score);
        out.append("\tXM:i:"); put_int(out, mm);
        out.append("\tXO:i:"); put_int(out, gapo);
        out.append("\tXG:i:"); put_int(out, gape);
        out.append("\tMD:Z:"); if (md.empty() || !whole) out.push_back('*'); else out.append(md);
        out.push_back('\n');
    }
}

template <typename Batch, typename Record>
inline bool write_records(FILE* f, const Batch& batch, Record record)
{
#if defined(_OPENMP)
    const int n_threads = std::max(1, omp_get_max_threads());
#else
    const int n_threads = 1;
#endif
    const uint32_t chunk = 16384u;
    bool ok = true;
    // the threads' buffers live for the whole call: a fresh 5 MB string per thread and step is a fresh mapping whose pages fault in again
    std::vector<std::string> bufs; bufs.resize(size_t(n_threads));
    for (uint64_t base = 0; base < batch.n && ok; base += uint64_t(chunk) * uint64_t(n_threads))
    {
        #pragma omp parallel for schedule(static, 1) num_threads(n_threads)
        for (int t = 0; t < n_threads; ++t)
        {
            const uint64_t lo = base + uint64_t(t) * chunk, hi = std::min<uint64_t>(lo + chunk, batch.n);
            std::string md;
            bufs[size_t(t)].clear();
            bufs[size_t(t)].reserve(size_t(chunk) * 320u);
            for (uint64_t i = lo; i < hi; ++i) record(uint32_t(i), bufs[size_t(t)], md);
        }
        for (int t = 0; t < n_threads && ok; ++t)
            ok = bufs[size_t(t)].empty() || fwrite(bufs[size_t(t)].data(), 1, bufs[size_t(t)].size(), f) == bufs[size_t(t)].size();
    }
    return ok;
}
} // namespace priv

/// append the batch's records to `f`; extra_flags are OR-ed into the FLAG of every aligned read (SamOutput marks single-end alignments READ_1 = 64 too)
inline bool write_sam_se(FILE* f, const SamBatchSE& batch, const SamReference& ref, const uint32_t extra_flags = 0u)
{
    return priv::write_records(f, batch, [&](const uint32_t i, std::string& out, std::string& md) { priv::se_record(batch, ref, i, extra_flags, out, md); });
}
/// append the two records of every pair of the batch to `f`, in pair order
inline bool write_sam_pe(FILE* f, const SamBatchPE& batch, const SamReference& ref)
{
    return priv::write_records(f, batch, [&](const uint32_t i, std::string& out, std::string& md) { priv::pe_records(batch, ref, i, out, md); });
}

} // namespace io
} // namespace nvbio
