// tests/compat/host_basic.cpp -- host-compiled (g++, no HIP) callers of the small building blocks of the drop-in layer:
//   replay        priority_deque<uint64, vector_view<uint64*>, cmp> driven by operation programs recorded from the REFERENCE's
//                 compiled interval heap (tests/golden/hit_deque_vectors.npz), array state compared after every operation
//   rank_ranges   rank_dictionary / fm_index range forms (rank4, rank_all, comp) over separate uint32 arrays and over uint4 arrays
//   gaps          max_text_gaps / max_pattern_gaps
#include <nvbio/basic/priority_deque.h>
#include <nvbio/basic/vector_view.h>
#include <stdio.h>
#include <stdint.h>
using namespace nvbio;
struct cmp { bool operator()(const uint64 f, const uint64 s) const { return ((f >> 32) & 0xFFFFFu) > ((s >> 32) & 0xFFFFFu); } };
extern "C" int replay(const uint8_t* ops, const uint64_t* vals, const uint32_t* caps, const uint32_t* sizes, const uint64_t* states, const uint32_t* case_start, int n_cases)
{
    uint64_t pos = 0;
    for (int c = 0; c < n_cases; ++c)
    {
        uint64 store[64];
        typedef vector_view<uint64*> vec;
        priority_deque<uint64, vec, cmp> dq(vec(0u, store), true);
        for (uint32 k = case_start[c]; k < case_start[c + 1]; ++k)
        {
            if (ops[k] == 0) { if (dq.size() == caps[k]) dq.pop_bottom(); dq.push(vals[k]); }
            else if (ops[k] == 1) dq.pop_top(); else dq.pop_bottom();
            if (dq.size() != sizes[k]) return -(int)k - 1;
            for (uint32 i = 0; i < sizes[k]; ++i) if (store[i] != states[pos + i]) return -(int)k - 1;
            pos += sizes[k];
        }
    }
    return 0;
}

#include <nvbio/basic/numbers.h>
#include <nvbio/fmindex/fmindex.h>
#include <nvbio/alignment/alignment_base.h>

// rank4 / rank_all of both ends of (lo[q], hi[q]) over the plain-array fm_index; returns the number of disagreements between
// the range forms, the point forms and rank_all
extern "C" int rank_ranges(uint32 n, uint32 primary, const uint32* L2, const uint32* bwt, const uint32* occ, uint32 n_queries,
                           const uint32* lo, const uint32* hi, uint32* out_lo, uint32* out_hi)
{
    typedef PackedStream<const uint32*, uint8, 2, true>                         bwt_type;
    typedef rank_dictionary<2, 64, bwt_type, const uint32*, const uint32*>      dict_type;
    typedef fm_index<dict_type, null_type>                                      fmi_type;
    const fmi_type fmi(n, primary, L2, dict_type(bwt_type(bwt), occ, (const uint32*)0), null_type());
    int bad = 0;
    for (uint32 q = 0; q < n_queries; ++q)
    {
        uint4 l, h;
        rank4(fmi, make_uint2(lo[q], hi[q]), &l, &h);
        fmi_type::vector_type al, ah;
        rank_all(fmi, make_uint2(lo[q], hi[q]), &al, &ah);
        for (uint32 c = 0; c < 4; ++c)
        {
            out_lo[4 * q + c] = comp(l, c); out_hi[4 * q + c] = comp(h, c);
            const uint2 r = rank(fmi, make_uint2(lo[q], hi[q]), uint8(c));
            if (al[c] != comp(l, c) || ah[c] != comp(h, c) || r.x != comp(l, c) || r.y != comp(h, c) || rank(fmi, lo[q], uint8(c)) != comp(l, c)) ++bad;
        }
    }
    return bad;
}
extern "C" void gaps(int match, int open, int ext, int min_score, int len, uint32* out)
{
    const aln::SimpleGotohScheme g(match, -1, open, ext);
    out[0] = aln::max_text_gaps(aln::make_gotoh_aligner<aln::LOCAL>(g), min_score, len);
    out[1] = aln::max_pattern_gaps(aln::make_gotoh_aligner<aln::LOCAL>(g), min_score, len);
    out[2] = aln::max_text_gaps(aln::make_edit_distance_aligner<aln::SEMI_GLOBAL>(), min_score, len);
}

// ---- the text sequence loader (io/sequence/sequence.h): batch form (the plain encoder) and record-at-a-time form (any subclass) ----
#include <nvbio/io/sequence/sequence.h>
#include <memory>
namespace {
struct RecordwiseEncoder : nvbio::io::SequenceDataEncoder
{
    RecordwiseEncoder(const nvbio::Alphabet a, nvbio::io::SequenceDataHost* d) : nvbio::io::SequenceDataEncoder(a, d) {}
};
}
/// load `path` in batches of `batch_size`; symbols / qualities / names ('\0'-separated) / index of all batches are concatenated.
/// returns the number of sequences, or -1 when a capacity is exceeded, -2 when the file does not open
extern "C" int load_reads(const char* path, uint32 flags, uint32 qenc, uint32 max_len, uint32 trim3, uint32 trim5, uint32 batch_size, int recordwise,
                          uint32* index, uint8* symbols, uint8* quals, char* names, uint32 cap_seqs, uint32 cap_syms, uint32 cap_names, uint32* info)
{
    std::unique_ptr<io::SequenceDataInputStream> f(io::open_sequence_file(path, io::QualityEncoding(qenc), uint32(-1), max_len, io::SequenceEncoding(flags), trim3, trim5));
    if (!f) return -2;
    uint32 n = 0, syms = 0, name_bytes = 0;
    index[0] = 0;
    for (;;)
    {
        io::SequenceDataHost data;
        int got;
        if (recordwise) { RecordwiseEncoder enc(DNA_N, &data); got = f->next(&enc, batch_size); }
        else got = io::next(DNA_N, &data, f.get(), batch_size);
        if (got <= 0) break;
        const io::SequenceDataAccess<DNA_N> access(data);
        if (n + data.size() > cap_seqs || syms + data.bps() > cap_syms || name_bytes + data.m_name_stream_len > cap_names) return -1;
        for (uint32 i = 0; i < data.bps(); ++i) { symbols[syms + i] = access.sequence_stream()[i]; quals[syms + i] = uint8(access.qual_stream()[i]); }
        for (uint32 i = 0; i < data.size(); ++i) index[n + i + 1] = syms + access.sequence_index()[i + 1];
        for (uint32 i = 0; i < data.m_name_stream_len; ++i) names[name_bytes + i] = access.name_stream()[i];
        info[0] = data.min_sequence_len(); info[1] = data.max_sequence_len(); info[2] = data.avg_sequence_len();
        n += data.size(); syms += data.bps(); name_bytes += data.m_name_stream_len;
    }
    info[3] = name_bytes; info[4] = f->is_ok() ? 1u : 0u;
    return int(n);
}
// (load_reads above opens .sam / .bam names through the same factory: AlignmentSequenceFile)
