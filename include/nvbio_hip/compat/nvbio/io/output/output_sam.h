// compat/nvbio/io/output/output_sam.h -- the SAM text writer (nvbio/io/output/output_sam.h, output_sam.cpp:120-580, output_priv.h):
// header (@HD / @RG / @PG / @SQ), one record per read (two per pair, anchor first), the tag set NM AS XM XO XG MD.  Reads are stored
// reversed by the aligner (io::REVERSE): a forward alignment prints them back to front, a reverse-complemented one complements them in
// stored order.  CIGARs are stored last operation first.  An alignment that runs over the end of its reference sequence is flagged
// unmapped with mapping quality 0 but still printed in full, as the reference does.
#pragma once
#include "output_file.h"
#include "output_batch.h"
#include "../../basic/threads.h"
#include "../../basic/timer.h"
#include "../../basic/dna.h"
#include "../../basic/omp.h"
#include <algorithm>
#include <string>
#include <vector>

namespace nvbio {
namespace io {

/// one alignment of a batch with everything a writer needs resolved (output_priv.h:49-135)
struct AlignmentData
{
    typedef io::SequenceDataAccess<DNA_N>           read_access_type;
    typedef read_access_type::sequence_stream_type  read_type;

    bool             valid;
    const Alignment* aln;
    uint32           aln_id, read_id, mapq;
    uint32           read_offset, read_len;
    const char*      read_name;
    read_type        read_data;
    const char*      qual;
    const Cigar*     cigar;
    uint32           cigar_pos, cigar_len;
    const uint8*     mds_vec;

    AlignmentData() : valid(false), aln(NULL), aln_id(uint32(-1)), read_id(uint32(-1)), mapq(0), read_offset(uint32(-1)), read_len(uint32(-1)), read_name(NULL),
        qual(NULL), cigar(NULL), cigar_pos(uint32(-1)), cigar_len(uint32(-1)), mds_vec(NULL) {}

    AlignmentData(const Alignment* _aln, const uint32 _mapq, const uint32 _aln_id, const uint32 _read_id, const io::SequenceDataHost* reads,
                  const HostCigarArray* cigars, const HostMdsArray* mds)
        : valid(true), aln(_aln), aln_id(_aln_id), read_id(_read_id), mapq(_mapq)
    {
        const read_access_type access(*reads);
        read_offset = access.sequence_index()[read_id];
        read_len    = access.sequence_index()[read_id + 1] - read_offset;
        read_name   = &*access.name_stream() + access.name_index()[read_id];
        read_data   = access.sequence_stream() + read_offset;
        qual        = &*access.qual_stream() + read_offset;
        const uint2 coord = cigars->coords[aln_id];
        cigar     = cigars->array[aln_id];
        cigar_pos = compute_cigar_pos(coord.x, aln->alignment());
        cigar_len = coord.y;
        mds_vec   = (*mds)[aln_id];
    }
    static AlignmentData invalid() { return AlignmentData(); }
};

namespace priv {
inline uint32 batch_read_id(const thrust::host_vector<uint32>& ids, const uint32 aln_id) { return ids.size() ? uint32(ids[aln_id]) : aln_id; }
} // namespace priv

inline AlignmentData get(HostOutputBatchSE& batch, const uint32 aln_id)
{ return AlignmentData(&batch.alignments[aln_id], batch.mapq[aln_id], aln_id, priv::batch_read_id(batch.read_ids, aln_id), batch.read_data, &batch.cigar, &batch.mds); }
/// slot set 0 holds the anchors, slot set 1 their opposite mates; which read file a slot belongs to is its alignment's mate bit
inline AlignmentData get_anchor_mate(HostOutputBatchPE& batch, const uint32 aln_id)
{ const uint32 m = batch.alignments[0][aln_id].mate(); return AlignmentData(&batch.alignments[0][aln_id], batch.mapq[0][aln_id], aln_id, priv::batch_read_id(batch.read_ids, aln_id), batch.read_data[m], &batch.cigar[0], &batch.mds[0]); }
inline AlignmentData get_opposite_mate(HostOutputBatchPE& batch, const uint32 aln_id)
{ const uint32 m = batch.alignments[1][aln_id].mate(); return AlignmentData(&batch.alignments[1][aln_id], batch.mapq[1][aln_id], aln_id, priv::batch_read_id(batch.read_ids, aln_id), batch.read_data[m], &batch.cigar[1], &batch.mds[1]); }
inline AlignmentData get_mate(HostOutputBatchPE& batch, const uint32 aln_id, const AlignmentMate mate)
{ return batch.alignments[0][aln_id].mate() == uint32(mate) ? get_anchor_mate(batch, aln_id) : get_opposite_mate(batch, aln_id); }

struct SamOutput : public OutputFile
{
    enum SamAlignmentFlags { SAM_FLAGS_PAIRED = 1, SAM_FLAGS_PROPER_PAIR = 2, SAM_FLAGS_UNMAPPED = 4, SAM_FLAGS_MATE_UNMAPPED = 8, SAM_FLAGS_REVERSE = 16,
                             SAM_FLAGS_MATE_REVERSE = 32, SAM_FLAGS_READ_1 = 64, SAM_FLAGS_READ_2 = 128, SAM_FLAGS_SECONDARY = 256, SAM_FLAGS_FAILED_QC = 512, SAM_FLAGS_DUPLICATE = 1024 };

    SamOutput(const char* _file_name, AlignmentType _alignment_type, BNT _bnt) : OutputFile(_file_name, _alignment_type, _bnt), fp(NULL)
    {
        fp = _file_name ? fopen(_file_name, "wt") : stdout;
        if (fp == NULL) { log_error(stderr, "SamOutput: could not open %s for writing\n", _file_name); return; }
        if (_file_name) setvbuf(fp, NULL, _IOFBF, 1u << 20);
    }
    ~SamOutput() { if (fp && fp != stdout) fclose(fp); }

    void header()
    {
        std::string h = "@HD\tVN:1.3\n";
        if (!rg_id.empty()) h += "@RG\tID:" + rg_id + rg_string + "\n";
        h += "@PG\tID:" + pg_id + "\tPN:" + pg_name + "\tVN:" + pg_version + "\tCL:\"" + pg_args + "\"\n";
        for (uint32 i = 0; i < bnt.n_seqs; ++i)
            h += std::string("@SQ\tSN:") + (bnt.names + bnt.names_index[i]) + "\tLN:" + std::to_string(bnt.sequence_index[i + 1] - bnt.sequence_index[i]) + "\n";
        fwrite(h.data(), 1, h.size(), fp);
    }
    void process(struct HostOutputBatchSE& batch)
    {
        float seconds = 0.0f;
        {
            ScopedTimer<float> timer(&seconds);
            ScopedLock hold(&mutex);
            // the records of a batch are independent: every OpenMP thread formats a contiguous share, the shares are written in order
            const size_t n_threads = size_t(usable_omp_threads());
            std::vector<std::string> text(n_threads);
            #pragma omp parallel num_threads(int(text.size()))
            {
                const uint32 t = uint32(omp_get_thread_num()), nt = uint32(omp_get_num_threads());
                const uint32 lo = uint32(uint64(batch.count) * t / nt), hi = uint32(uint64(batch.count) * (t + 1u) / nt);
                text[t].reserve(size_t(hi - lo) * 320u);
                for (uint32 c = lo; c < hi; ++c) record(get(batch, c), AlignmentData::invalid(), text[t]);
            }
            for (size_t t = 0; t < text.size(); ++t) fwrite(text[t].data(), 1, text[t].size(), fp);
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
            std::vector<std::string> text(n_threads);
            #pragma omp parallel num_threads(int(text.size()))
            {
                const uint32 t = uint32(omp_get_thread_num()), nt = uint32(omp_get_num_threads());
                const uint32 lo = uint32(uint64(batch.count) * t / nt), hi = uint32(uint64(batch.count) * (t + 1u) / nt);
                text[t].reserve(size_t(hi - lo) * 800u);
                for (uint32 c = lo; c < hi; ++c)
                {
                    const AlignmentData anchor = get_anchor_mate(batch, c), opposite = get_opposite_mate(batch, c);
                    record(anchor, opposite, text[t]);
                    record(opposite, anchor, text[t]);
                }
            }
            for (size_t t = 0; t < text.size(); ++t) fwrite(text[t].data(), 1, text[t].size(), fp);
        }
        iostats.n_reads += batch.count;
        iostats.output_process_timings.add(batch.count, seconds);
    }
    void close(void) { if (fp && fp != stdout) fclose(fp); fp = NULL; }

private:
    uint32 sequence_of(const uint32 pos) const { return uint32(std::upper_bound(bnt.sequence_index, bnt.sequence_index + bnt.n_seqs, pos) - bnt.sequence_index) - 1u; }

    /// decimal text without a temporary string
    static void put(std::string& o, uint32 v) { char b[12]; int n = 0; do { b[n++] = char('0' + v % 10u); v /= 10u; } while (v); while (n) o.push_back(b[--n]); }
    static void put(std::string& o, const int32 v) { if (v < 0) { o.push_back('-'); put(o, uint32(-int64(v))); } else put(o, uint32(v)); }

    /// "3M1D7M" from the stored (reversed) operation list; returns the read bases it consumes
    static uint32 cigar_text(const AlignmentData& a, std::string& out)
    {
        uint32 consumed = 0;
        for (uint32 i = a.cigar_len; i-- > 0;)
        {
            const Cigar& op = a.cigar[i];
            put(out, uint32(op.m_len)); out += "MIDS"[op.m_type];
            if (op.m_type != Cigar::DELETION) consumed += op.m_len;
        }
        return consumed;
    }
    /// the MD:Z value and the mismatch / gap-open / gap-extension counts of an MD program (match runs are summed as bytes, like the reference's counter)
    static void md_text(const AlignmentData& a, std::string& md, uint32& mm, uint32& gapo, uint32& gape)
    {
        mm = gapo = gape = 0;
        if (a.mds_vec == NULL) { log_warning(stderr, "  SAM: alignment %u from read %u has an empty MD string\n", a.aln_id, a.read_id); return; }
        const uint8* p = a.mds_vec;
        const uint32 end = uint32(p[0]) | (uint32(p[1]) << 8);
        uint32 i = 2;
        do
        {
            const uint8 op = p[i++];
            if (op == MDS_MATCH)         { uint8 run = p[i++]; while (i < end && p[i] == MDS_MATCH) run = uint8(run + p[i++]); put(md, uint32(run)); }
            else if (op == MDS_MISMATCH) { md += dna_to_char(p[i++]); ++mm; }
            else if (op == MDS_INSERTION){ const uint8 l = p[i++]; i += l; ++gapo; gape += l - 1u; }
            else if (op == MDS_DELETION) { const uint8 l = p[i++]; md += '^'; for (uint8 k = 0; k < l; ++k) md += dna_to_char(p[i++]); md += '0'; ++gapo; gape += l - 1u; }
        } while (i < end);
    }
    /// append the SAM line of `a` (whose mate, for pairs, is `mate`)
    void record(const AlignmentData& a, const AlignmentData& mate, std::string& out)
    {
        // the read as the file had it (or its reverse complement), and its qualities, as text
        const bool rc = a.aln->is_rc();
        auto put_read = [&](std::string& o)
        {
            const size_t at = o.size();
            o.resize(at + 2u * size_t(a.read_len) + 1u);
            char* seq = &o[at]; char* qual = seq + a.read_len + 1u;
            for (uint32 i = 0; i < a.read_len; ++i)
            {
                const uint32 src = rc ? i : a.read_len - 1u - i;
                const uint8 s = a.read_data[src];
                seq[i]  = dna_to_char(rc ? (s < 4 ? uint8(3u - s) : uint8(4)) : s);
                qual[i] = char(a.qual[src] + 33);
            }
            seq[a.read_len] = '\t';
        };
        uint32 mapq = a.mapq;
        out += a.read_name;
        if (!(a.aln->is_aligned() || int(mapq) < mapq_filter))
        {
            out += "\t4\t*\t0\t0\t*\t*\t0\t0\t"; put_read(out); out += '\n';
            return;
        }
        uint32 flags = (a.aln->mate() ? SAM_FLAGS_READ_2 : SAM_FLAGS_READ_1) | (rc ? SAM_FLAGS_REVERSE : 0u);
        if (alignment_type == PAIRED_END)
        {
            flags |= SAM_FLAGS_PAIRED;
            if (mate.aln->is_concordant()) flags |= SAM_FLAGS_PROPER_PAIR;
            if (!mate.aln->is_aligned())   flags |= SAM_FLAGS_MATE_UNMAPPED;
            if (mate.aln->is_rc())         flags |= SAM_FLAGS_MATE_REVERSE;
        }
        const uint32 span = reference_cigar_length(a.cigar, a.cigar_len), seq_id = sequence_of(a.cigar_pos);
        if (a.cigar_pos + span > bnt.sequence_index[seq_id + 1]) { flags |= SAM_FLAGS_UNMAPPED; mapq = 0; }       // bridges two reference sequences
        std::string cigar;
        if (cigar_text(a, cigar) != a.read_len)
        {
            log_error(stderr, "SAM output : cigar length doesn't match read %u\n", a.read_id);
            out.resize(out.size() - strlen(a.read_name));
            return;
        }
        const char* rnext = "*"; uint32 pnext = 0; int32 tlen = 0;
        if (alignment_type == PAIRED_END)
        {
            rnext = "="; pnext = a.cigar_pos - bnt.sequence_index[seq_id] + 1u;
            if (mate.aln->is_aligned())
            {
                const uint32 o_span = reference_cigar_length(mate.cigar, mate.cigar_len), o_seq = sequence_of(mate.cigar_pos);
                pnext = mate.cigar_pos - bnt.sequence_index[o_seq] + 1u;
                if (o_seq != seq_id) rnext = bnt.names + bnt.names_index[o_seq];
                else
                {
                    tlen = int32(std::max(mate.cigar_pos + o_span, a.cigar_pos + span) - std::min(mate.cigar_pos, a.cigar_pos));
                    if (mate.cigar_pos < a.cigar_pos) tlen = -tlen;
                }
            }
        }
        std::string md; uint32 mm, gapo, gape;
        md_text(a, md, mm, gapo, gape);
        out += '\t'; put(out, flags);
        out += '\t'; out += bnt.names + bnt.names_index[seq_id];
        out += '\t'; put(out, uint32(a.cigar_pos - bnt.sequence_index[seq_id] + 1u));
        out += '\t'; put(out, mapq);
        out += '\t'; out += cigar;
        out += '\t'; out += rnext;
        out += '\t'; put(out, pnext);
        out += '\t'; put(out, tlen);
        out += '\t'; put_read(out);
        out += "\tNM:i:"; put(out, uint32(a.aln->ed()));
        out += "\tAS:i:"; put(out, int32(a.aln->score()));
        out += "\tXM:i:"; put(out, mm);
        out += "\tXO:i:"; put(out, gapo);
        out += "\tXG:i:"; put(out, gape);
        out += "\tMD:Z:"; out += md.empty() ? std::string("*") : md;
        out += '\n';
    }

    FILE* fp;
    Mutex mutex;
};

} // namespace io
} // namespace nvbio
