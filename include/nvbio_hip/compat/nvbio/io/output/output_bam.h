// compat/nvbio/io/output/output_bam.h -- the BAM writer (nvbio/io/output/output_bam.h, output_bam.cpp:39-700, bam_format.h): the binary
// twin of SamOutput.  Header: "BAM\1", the text @HD / @RG / @PG (BamOutput prints no @SQ lines and no CL: field, output_bam.cpp:607-650),
// then the reference sequences' names and lengths.  One record per read (two per pair, anchor first): refID, pos, bin_mq_nl with bin 0
// ("BAM alignment bin is always 0", output_bam.cpp:374), flag_nc, l_seq, next_refID, next_pos, tlen, the name, the CIGAR words (length << 4
// | op), the read as 4-bit codes (=ACMGRSVTWYHKDBN), the phred qualities, and for mapped records the tags NM:c AS:i XM:c XO:c XG:c MD:Z
// (output_bam.cpp:448-519).  An unmapped read -- or one whose alignment runs over the end of its reference sequence -- carries refID = pos =
// next_refID = next_pos = -1 and the UNMAPPED flag alone or, for the second kind, beside the strand / pair flags.
//
// One deliberate difference: BamOutput stores `next_refID = mate's sequence - this read's sequence` (output_bam.cpp:392), which names the
// wrong sequence for every pair not on sequence 0; the mate's sequence id is written here, as the format asks.
//
// Records are formatted by all OpenMP threads into per-thread buffers (as SamOutput does), the stream is cut into 0xFF00-byte BGZF
// blocks and those are deflated in parallel too; BGZF is a series of gzip members with a "BC" extra field holding the block size, closed
// by the 28-byte empty block.
#pragma once
#include "output_sam.h"
#include <zlib.h>

namespace nvbio {
namespace io {

struct BamOutput : public OutputFile
{
    enum BamAlignmentFlags { BAM_FLAGS_PAIRED = 1, BAM_FLAGS_PROPER_PAIR = 2, BAM_FLAGS_UNMAPPED = 4, BAM_FLAGS_MATE_UNMAPPED = 8, BAM_FLAGS_REVERSE = 16,
                             BAM_FLAGS_MATE_REVERSE = 32, BAM_FLAGS_READ_1 = 64, BAM_FLAGS_READ_2 = 128, BAM_FLAGS_SECONDARY = 256, BAM_FLAGS_FAILED_QC = 512, BAM_FLAGS_DUPLICATE = 1024 };

    BamOutput(const char* _file_name, AlignmentType _alignment_type, BNT _bnt) : OutputFile(_file_name, _alignment_type, _bnt), fp(NULL)
    {
        fp = fopen(_file_name, "wb");
        if (fp == NULL) { log_error(stderr, "BamOutput: could not open %s for writing\n", _file_name); return; }
        setvbuf(fp, NULL, _IOFBF, 1u << 20);
    }
    ~BamOutput() { close(); }

    void header()
    {
        std::string text = "@HD\tVN:1.3\n";
        if (!rg_id.empty()) text += "@RG\tID:" + rg_id + rg_string + "\n";
        text += "@PG\tID:" + pg_id + "\tPN:" + pg_name + "\tVN:" + pg_version + "\n";
        std::string h("BAM\1", 4);
        put32(h, uint32(text.size())); h += text;
        put32(h, bnt.n_seqs);
        for (uint32 i = 0; i < bnt.n_seqs; ++i)
        {
            const char* name = bnt.names + bnt.names_index[i];
            put32(h, uint32(strlen(name) + 1u)); h.append(name, strlen(name) + 1u);
            put32(h, bnt.sequence_index[i + 1] - bnt.sequence_index[i]);
        }
        ScopedLock hold(&mutex);
        pending += h;
        flush(true);                                    // the header ends its own block, as BamOutput's does (output_bam.cpp:672)
    }
    void process(struct HostOutputBatchSE& batch)
    {
        float seconds = 0.0f;
        {
            ScopedTimer<float> timer(&seconds);
            ScopedLock hold(&mutex);
            const size_t n_threads = size_t(usable_omp_threads());
            std::vector<std::string> part(n_threads);
            #pragma omp parallel num_threads(int(part.size()))
            {
                const uint32 t = uint32(omp_get_thread_num()), nt = uint32(omp_get_num_threads());
                const uint32 lo = uint32(uint64(batch.count) * t / nt), hi = uint32(uint64(batch.count) * (t + 1u) / nt);
                part[t].reserve(size_t(hi - lo) * 260u);
                for (uint32 c = lo; c < hi; ++c) record(get(batch, c), AlignmentData::invalid(), part[t]);
            }
            for (size_t t = 0; t < part.size(); ++t) pending += part[t];
            flush(false);
        }
        iostats.n_reads += batch.count;
        iostats.output_process_timings.add(batch.count, seconds);
    }
    void process(struct HostOutputBatchPE& batch)
    {
        float seconds = 0.0f;
        {
            ScopedTimer<float> timer(&seconds);
            ScopedLock hold(&mutex);
            const size_t n_threads = size_t(usable_omp_threads());
            std::vector<std::string> part(n_threads);
            #pragma omp parallel num_threads(int(part.size()))
            {
                const uint32 t = uint32(omp_get_thread_num()), nt = uint32(omp_get_num_threads());
                const uint32 lo = uint32(uint64(batch.count) * t / nt), hi = uint32(uint64(batch.count) * (t + 1u) / nt);
                part[t].reserve(size_t(hi - lo) * 600u);
                for (uint32 c = lo; c < hi; ++c)
                {
                    const AlignmentData anchor = get_anchor_mate(batch, c), opposite = get_opposite_mate(batch, c);
                    record(anchor, opposite, part[t]);
                    record(opposite, anchor, part[t]);
                }
            }
            for (size_t t = 0; t < part.size(); ++t) pending += part[t];
            flush(false);
        }
        iostats.n_reads += batch.count;
        iostats.output_process_timings.add(batch.count, seconds);
    }
    void close(void)
    {
        if (fp == NULL) return;
        {
            ScopedLock hold(&mutex);
            flush(true);
            static const unsigned char eof[28] = { 0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
            fwrite(eof, 1, sizeof(eof), fp);
        }
        fclose(fp); fp = NULL;
    }

private:
    static const uint32 BLOCK = 0xFF00u;                 // uncompressed bytes per BGZF block

    static void put32(std::string& o, const uint32 v) { const char b[4] = { char(v & 255u), char((v >> 8) & 255u), char((v >> 16) & 255u), char(v >> 24) }; o.append(b, 4); }
    static void put16(std::string& o, const uint32 v) { const char b[2] = { char(v & 255u), char((v >> 8) & 255u) }; o.append(b, 2); }
    static void tag_c(std::string& o, const char* tag, const uint32 v) { o.append(tag, 2); o.push_back('c'); o.push_back(char(uint8(v))); }
    static void tag_i(std::string& o, const char* tag, const uint32 v) { o.append(tag, 2); o.push_back('i'); put32(o, v); }

    uint32 sequence_of(const uint32 pos) const { return uint32(std::upper_bound(bnt.sequence_index, bnt.sequence_index + bnt.n_seqs, pos) - bnt.sequence_index) - 1u; }

    /// one gzip member holding `n` bytes (BGZF: extra field "BC", 2, total block size - 1)
    static void bgzf_block(const unsigned char* data, const uint32 n, std::string& out)
    {
        unsigned char buf[0x10000];
        z_stream zs; memset(&zs, 0, sizeof(zs));
        deflateInit2(&zs, 6, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);                 // raw deflate
        zs.next_in = const_cast<unsigned char*>(data); zs.avail_in = n;
        zs.next_out = buf + 18; zs.avail_out = sizeof(buf) - 18 - 8;
        const int rc = deflate(&zs, Z_FINISH);
        uint32 clen = uint32(zs.total_out);
        deflateEnd(&zs);
        if (rc != Z_STREAM_END)                                                         // incompressible: store
        {
            memset(&zs, 0, sizeof(zs));
            deflateInit2(&zs, 0, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
            zs.next_in = const_cast<unsigned char*>(data); zs.avail_in = n;
            zs.next_out = buf + 18; zs.avail_out = sizeof(buf) - 18 - 8;
            deflate(&zs, Z_FINISH); clen = uint32(zs.total_out); deflateEnd(&zs);
        }
        const uint32 bsize = 18u + clen + 8u - 1u, crc = uint32(crc32(crc32(0L, Z_NULL, 0), data, n));
        const unsigned char head[18] = { 0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, (unsigned char)(bsize & 255u), (unsigned char)(bsize >> 8) };
        memcpy(buf, head, 18);
        unsigned char* tail = buf + 18 + clen;
        for (uint32 k = 0; k < 4; ++k) { tail[k] = (unsigned char)((crc >> (8u * k)) & 255u); tail[4 + k] = (unsigned char)((n >> (8u * k)) & 255u); }
        out.append(reinterpret_cast<const char*>(buf), 18u + clen + 8u);
    }
    /// compress and write the whole blocks of `pending` (everything when `all`)
    void flush(const bool all)
    {
        const size_t n_blocks = all ? (pending.size() + BLOCK - 1u) / BLOCK : pending.size() / BLOCK;
        if (n_blocks == 0) return;
        std::vector<std::string> z(n_blocks);
        const unsigned char* data = reinterpret_cast<const unsigned char*>(pending.data());
        const size_t total = pending.size();
        #pragma omp parallel for schedule(dynamic, 4) num_threads(usable_omp_threads())
        for (int64 b = 0; b < int64(n_blocks); ++b)
        {
            const size_t lo = size_t(b) * BLOCK, hi = std::min(total, lo + BLOCK);
            bgzf_block(data + lo, uint32(hi - lo), z[size_t(b)]);
        }
        for (size_t b = 0; b < n_blocks; ++b) fwrite(z[b].data(), 1, z[b].size(), fp);
        pending.erase(0, std::min(total, n_blocks * size_t(BLOCK)));
    }

    /// append the BAM record of `a` (whose mate, for pairs, is `mate`)
    void record(const AlignmentData& a, const AlignmentData& mate, std::string& out)
    {
        static const uint8 code[5] = { 1, 2, 4, 8, 15 };                               // A C G T N in =ACMGRSVTWYHKDBN
        const bool   rc = a.aln->is_rc();
        const size_t name_len = strlen(a.read_name) + 1u;
        uint32 mapq = a.mapq;
        int32  ref_id = -1, pos = -1, next_ref = -1, next_pos = -1, tlen = 0;
        uint32 flags = 0, n_cigar = 0;
        bool   mapped = false;
        const bool aligned = !(a.aln->is_aligned() == false || int(mapq) < mapq_filter);
        uint32 span = 0, seq_id = 0;
        if (!aligned) flags = BAM_FLAGS_UNMAPPED;
        else
        {
            flags = (a.aln->mate() ? BAM_FLAGS_READ_2 : BAM_FLAGS_READ_1) | (rc ? BAM_FLAGS_REVERSE : 0u);
            if (alignment_type == PAIRED_END)
            {
                flags |= BAM_FLAGS_PAIRED;
                if (mate.aln->is_concordant()) flags |= BAM_FLAGS_PROPER_PAIR;
                if (!mate.aln->is_aligned())   flags |= BAM_FLAGS_MATE_UNMAPPED;
                if (mate.aln->is_rc())         flags |= BAM_FLAGS_MATE_REVERSE;
            }
            span = reference_cigar_length(a.cigar, a.cigar_len); seq_id = sequence_of(a.cigar_pos);
            if (a.cigar_pos + span > bnt.sequence_index[seq_id + 1]) flags |= BAM_FLAGS_UNMAPPED;       // bridges two reference sequences
            else
            {
                uint32 consumed = 0;
                for (uint32 i = 0; i < a.cigar_len; ++i) if (a.cigar[i].m_type != Cigar::DELETION) consumed += a.cigar[i].m_len;
                if (consumed != a.read_len) { log_error(stderr, "BAM output : cigar length doesn't match read %u (%u != %u)\n", a.read_id, consumed, a.read_len); return; }
                mapped = true; n_cigar = a.cigar_len;
                ref_id = int32(seq_id); pos = int32(a.cigar_pos - bnt.sequence_index[seq_id]);
                if (alignment_type == PAIRED_END)
                {
                    if (mate.aln->is_aligned())
                    {
                        const uint32 o_span = reference_cigar_length(mate.cigar, mate.cigar_len), o_seq = sequence_of(mate.cigar_pos);
                        next_ref = int32(o_seq); next_pos = int32(mate.cigar_pos - bnt.sequence_index[o_seq]);
                        if (o_seq == seq_id)
                        {
                            tlen = int32(std::max(mate.cigar_pos + o_span, a.cigar_pos + span) - std::min(mate.cigar_pos, a.cigar_pos));
                            if (mate.cigar_pos < a.cigar_pos) tlen = -tlen;
                        }
                    }
                    else { next_ref = ref_id; next_pos = pos; }
                }
            }
        }
        const size_t at = out.size();
        put32(out, 0u);                                                                 // block_size, patched below
        put32(out, uint32(ref_id)); put32(out, uint32(pos));
        put32(out, uint32(name_len) | ((mapped ? mapq : 0u) << 8));                     // bin 0
        put32(out, (flags << 16) | n_cigar);
        put32(out, a.read_len);
        put32(out, uint32(next_ref)); put32(out, uint32(next_pos)); put32(out, uint32(tlen));
        out.append(a.read_name, name_len);
        // the stored (reversed) operation list, first operation first; BAM's op codes are M I D N S: Cigar's SUBSTITUTION, INSERTION,
        // DELETION, SOFT_CLIPPING = 0 1 2 3 -> 0 1 2 4
        for (uint32 i = n_cigar; i-- > 0;) { static const uint32 op[4] = { 0u, 1u, 2u, 4u }; put32(out, (uint32(a.cigar[i].m_len) << 4) | op[a.cigar[i].m_type & 3u]); }
        {
            const size_t s0 = out.size();
            out.resize(s0 + (a.read_len + 1u) / 2u + a.read_len);
            char* seq = &out[s0]; char* qual = seq + (a.read_len + 1u) / 2u;
            for (uint32 i = 0; i < a.read_len; ++i)
            {
                const uint32 src = rc ? i : a.read_len - 1u - i;
                uint8 s = a.read_data[src];
                if (rc) s = s < 4 ? uint8(3u - s) : uint8(4);
                const uint8 c = code[s < 5 ? s : 4];
                if (i & 1u) seq[i / 2u] = char(uint8(seq[i / 2u]) | c); else seq[i / 2u] = char(c << 4);
                qual[i] = a.qual[src];
            }
        }
        if (mapped)
        {
            std::string md; uint32 mm = 0, gapo = 0, gape = 0;
            md_text(a, md, mm, gapo, gape);
            tag_c(out, "NM", a.aln->ed()); tag_i(out, "AS", uint32(a.aln->score()));
            tag_c(out, "XM", mm); tag_c(out, "XO", gapo); tag_c(out, "XG", gape);
            if (!md.empty()) { out.append("MDZ", 3); out += md; out.push_back('\0'); }
        }
        const uint32 block_size = uint32(out.size() - at - 4u);
        for (uint32 k = 0; k < 4; ++k) out[at + k] = char((block_size >> (8u * k)) & 255u);
    }
    /// the MD:Z value and the mismatch / gap counters (the SAM writer's, output_bam.cpp:141-232 is the same program)
    static void md_text(const AlignmentData& a, std::string& md, uint32& mm, uint32& gapo, uint32& gape)
    {
        mm = gapo = gape = 0;
        if (a.mds_vec == NULL) { log_warning(stderr, "  BAM: alignment %u from read %u has an empty MD string\n", a.aln_id, a.read_id); return; }
        const uint8* p = a.mds_vec;
        const uint32 end = uint32(p[0]) | (uint32(p[1]) << 8);
        uint32 i = 2;
        do
        {
            const uint8 op = p[i++];
            if (op == MDS_MATCH)
            {
                uint8 run = p[i++]; while (i < end && p[i] == MDS_MATCH) run = uint8(run + p[i++]);
                char b[4]; int n = 0; uint32 v = run; do { b[n++] = char('0' + v % 10u); v /= 10u; } while (v); while (n) md.push_back(b[--n]);
            }
            else if (op == MDS_MISMATCH) { md += dna_to_char(p[i++]); ++mm; }
            else if (op == MDS_INSERTION){ const uint8 l = p[i++]; i += l; ++gapo; gape += l - 1u; }
            else if (op == MDS_DELETION) { const uint8 l = p[i++]; md += '^'; for (uint8 k = 0; k < l; ++k) md += dna_to_char(p[i++]); md += '0'; ++gapo; gape += l - 1u; }
        } while (i < end);
    }

    FILE*       fp;
    Mutex       mutex;
    std::string pending;       // formatted records not yet in a block
};

} // namespace io
} // namespace nvbio
