// compat/nvbio/io/output/output_sam.h -- the SAM text writer (nvbio/io/output/output_sam.h, output_sam.cpp:120-580, output_priv.h):
// header (@HD / @RG / @PG / @SQ), one record per read (two per pair, anchor first), the tag set NM AS XM XO XG MD.  Reads are stored
// reversed by the aligner (io::REVERSE): a forward alignment prints them back to front, a reverse-complemented one complements them in
// stored order.  CIGARs are stored last operation first.  An alignment that runs over the end of its reference sequence is flagged
// unmapped with mapping quality 0 but still printed in full, as the reference does.
#pragma once
#include "output_file.h"
#include "output_batch.h"
#include "output_writer.h"
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
        writer.open(fp);
    }
    ~SamOutput() { close(); }

    void header()
    {
        std::string h = "@HD\tVN:1.3\n";
        if (!rg_id.empty()) h += "@RG\tID:" + rg_id + rg_string + "\n";
        h += "@PG\tID:" + pg_id + "\tPN:" + pg_name + "\tVN:" + pg_version + "\tCL:\"" + pg_args + "\"\n";
        for (uint32 i = 0; i < bnt.n_seqs; ++i)
            h += std::string("@SQ\tSN:") + (bnt.names + bnt.names_index[i]) + "\tLN:" + std::to_string(bnt.sequence_index[i + 1] - bnt.sequence_index[i]) + "\n";
        ScopedLock hold(&mutex);
        writer.drain();
        if (fp) fwrite(h.data(), 1, h.size(), fp);
    }
    void process(struct HostOutputBatchSE& batch)
    {
        Timer timer; timer.start();
        {
            ScopedLock hold(&mutex);
            // the records of a batch are independent: every OpenMP thread formats a contiguous share; the shares go to the file in order, on
            // the writer's thread (output_writer.h), while the caller is back at its device
            priv::OrderedFileWriter::chunk_list text = writer.take(size_t(usable_omp_threads()));
            #pragma omp parallel num_threads(int(text.size()))
            {
                const uint32 t = uint32(omp_get_thread_num()), nt = uint32(omp_get_num_threads());
                const uint32 lo = uint32(uint64(batch.count) * t / nt), hi = uint32(uint64(batch.count) * (t + 1u) / nt);
                text[t].reserve(size_t(hi - lo) * 320u);
                for (uint32 c = lo; c < hi; ++c) record(get(batch, c), AlignmentData::invalid(), text[t]);
            }
            timer.stop();
            const float format_seconds = timer.seconds();
            timer.start();
            writer.push(std::move(text), format_seconds);
            timer.stop();
            iostats.n_reads += batch.count;
            iostats.output_process_timings.add(batch.count, format_seconds + timer.seconds());
        }
    }
    void process(struct HostOutputBatchPE& batch)
    {
        Timer timer; timer.start();
        {
            ScopedLock hold(&mutex);
            priv::OrderedFileWriter::chunk_list text = writer.take(size_t(usable_omp_threads()));
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
            timer.stop();
            const float format_seconds = timer.seconds();
            timer.start();
            writer.push(std::move(text), format_seconds);
            timer.stop();
            iostats.n_reads += batch.count;
            iostats.output_process_timings.add(batch.count, format_seconds + timer.seconds());
        }
    }
    void close(void)
    {
        ScopedLock hold(&mutex);
        writer.finish();                                  // everything handed over is in the file before it is closed
        if (fp && fp != stdout) fclose(fp); else if (fp) fflush(fp);
        fp = NULL; writer.open(NULL);
    }

private:
    uint32 sequence_of(const uint32 pos) const { return uint32(std::upper_bound(bnt.sequence_index, bnt.sequence_index + bnt.n_seqs, pos) - bnt.sequence_index) - 1u; }

    // text through a pointer into room the caller asked the buffer for: decimal numbers, strings, single characters
    static char* put(char* p, uint32 v) { char b[12]; int n = 0; do { b[n++] = char('0' + v % 10u); v /= 10u; } while (v); while (n) *p++ = b[--n]; return p; }
    static char* put(char* p, const int32 v) { if (v < 0) { *p++ = '-'; return put(p, uint32(-int64(v))); } return put(p, uint32(v)); }
    static char* put(char* p, const char* s) { const size_t n = strlen(s); memcpy(p, s, n); return p + n; }
    static char* tab(char* p) { *p++ = '\t'; return p; }

    /// read bases an alignment's CIGAR consumes
    static uint32 cigar_read_length(const AlignmentData& a)
    {
        uint32 consumed = 0;
        for (uint32 i = 0; i < a.cigar_len; ++i) if (a.cigar[i].m_type != Cigar::DELETION) consumed += a.cigar[i].m_len;
        return consumed;
    }
    /// "3M1D7M" from the stored (reversed) operation list: at most 6 characters per operation
    static char* cigar_text(const AlignmentData& a, char* p)
    {
        for (uint32 i = a.cigar_len; i-- > 0;) { const Cigar& op = a.cigar[i]; p = put(p, uint32(op.m_len)); *p++ = "MIDS"[op.m_type]; }
        return p;
    }
    /// bytes of an MD program (0 when there is none)
    static uint32 md_program_bytes(const AlignmentData& a) { return a.mds_vec ? (uint32(a.mds_vec[0]) | (uint32(a.mds_vec[1]) << 8)) : 0u; }
    /// the MD:Z value and the mismatch / gap-open / gap-extension counts of an MD program (match runs are summed as bytes, like the reference's
    /// counter); the text takes at most 2 characters per program byte
    static char* md_text(const AlignmentData& a, char* p, uint32& mm, uint32& gapo, uint32& gape)
    {
        mm = gapo = gape = 0;
        if (a.mds_vec == NULL) { log_warning(stderr, "  SAM: alignment %u from read %u has an empty MD string\n", a.aln_id, a.read_id); return p; }
        const uint8* m = a.mds_vec;
        const uint32 end = uint32(m[0]) | (uint32(m[1]) << 8);
        uint32 i = 2;
        do
        {
            const uint8 op = m[i++];
            if (op == MDS_MATCH)         { uint8 run = m[i++]; while (i < end && m[i] == MDS_MATCH) run = uint8(run + m[i++]); p = put(p, uint32(run)); }
            else if (op == MDS_MISMATCH) { *p++ = dna_to_char(m[i++]); ++mm; }
            else if (op == MDS_INSERTION){ const uint8 l = m[i++]; i += l; ++gapo; gape += l - 1u; }
            else if (op == MDS_DELETION) { const uint8 l = m[i++]; *p++ = '^'; for (uint8 k = 0; k < l; ++k) *p++ = dna_to_char(m[i++]); *p++ = '0'; ++gapo; gape += l - 1u; }
        } while (i < end);
        return p;
    }
    /// the read as the file had it (or its reverse complement), a tab, its qualities: 2 * read_len + 1 characters
    static char* read_text(const AlignmentData& a, const bool rc, char* seq)
    {
        // (4-bit codes: 0..3 = ACGT, anything else prints as N; the complement of N is N)
        static const char fw_text[17] = "ACGTNNNNNNNNNNNN", rc_text[17] = "TGCANNNNNNNNNNNN";
        const uint32 n = a.read_len;
        char* qual = seq + n + 1u;
        if (rc) for (uint32 i = 0; i < n; ++i) { seq[i] = rc_text[a.read_data[i] & 15u];          qual[i] = char(a.qual[i] + 33); }
        else    for (uint32 i = 0; i < n; ++i) { seq[i] = fw_text[a.read_data[n - 1u - i] & 15u]; qual[i] = char(a.qual[n - 1u - i] + 33); }
        seq[n] = '\t';
        return qual + n;
    }
    /// append the SAM line of `a` (whose mate, for pairs, is `mate`)
    void record(const AlignmentData& a, const AlignmentData& mate, priv::TextBuffer& out)
    {
        const bool   rc = a.aln->is_rc();
        const size_t name_len = strlen(a.read_name);
        uint32 mapq = a.mapq;
        if (!(a.aln->is_aligned() || int(mapq) < mapq_filter))
        {
            char* p = out.room(name_len + 2u * size_t(a.read_len) + 32u);
            memcpy(p, a.read_name, name_len); p += name_len;
            p = put(p, "\t4\t*\t0\t0\t*\t*\t0\t0\t"); p = read_text(a, rc, p); *p++ = '\n';
            out.commit(p);
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
        if (cigar_read_length(a) != a.read_len)
        {
            log_error(stderr, "SAM output : cigar length doesn't match read %u\n", a.read_id);
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
        const char*  ref_name = bnt.names + bnt.names_index[seq_id];
        // the longest this line can get: the names, 6 characters per CIGAR operation, the read twice, 2 per MD program byte, the fixed fields
        char* p = out.room(name_len + strlen(ref_name) + strlen(rnext) + 6u * size_t(a.cigar_len) + 2u * size_t(a.read_len) + 2u * size_t(md_program_bytes(a)) + 160u);
        memcpy(p, a.read_name, name_len); p += name_len;
        p = tab(p); p = put(p, flags);
        p = tab(p); p = put(p, ref_name);
        p = tab(p); p = put(p, uint32(a.cigar_pos - bnt.sequence_index[seq_id] + 1u));
        p = tab(p); p = put(p, mapq);
        p = tab(p); p = cigar_text(a, p);
        p = tab(p); p = put(p, rnext);
        p = tab(p); p = put(p, pnext);
        p = tab(p); p = put(p, tlen);
        p = tab(p); p = read_text(a, rc, p);
        p = put(p, "\tNM:i:"); p = put(p, uint32(a.aln->ed()));
        p = put(p, "\tAS:i:"); p = put(p, int32(a.aln->score()));
        // the MD text is written where it goes; the counters it yields are printed in front of it
        char  md_buf[64];
        const uint32 md_room = 2u * md_program_bytes(a) + 2u;
        std::vector<char> md_big;
        char* md = md_buf;
        if (md_room > sizeof(md_buf)) { md_big.resize(md_room); md = md_big.data(); }
        uint32 mm, gapo, gape;
        const char* md_end = md_text(a, md, mm, gapo, gape);
        p = put(p, "\tXM:i:"); p = put(p, mm);
        p = put(p, "\tXO:i:"); p = put(p, gapo);
        p = put(p, "\tXG:i:"); p = put(p, gape);
        p = put(p, "\tMD:Z:");
        if (md_end == md) *p++ = '*'; else { memcpy(p, md, size_t(md_end - md)); p += md_end - md; }
        *p++ = '\n';
        out.commit(p);
    }

    FILE* fp;
    Mutex mutex;
    priv::OrderedFileWriter writer;
};

} // namespace io
} // namespace nvbio
