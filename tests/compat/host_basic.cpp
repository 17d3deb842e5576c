// tests/compat/host_basic.cpp -- host-compiled (g++, no HIP) callers of the small building blocks of the drop-in layer:
//   heapify_arrays priority_deque(seq, false) over arbitrary arrays
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
            else if (ops[k] == 1) dq.pop_top();
            else if (ops[k] == 2) dq.pop_bottom();
            else if (ops[k] == 3) { uint64& w = store[uint32(vals[k])]; w = (w & ~(uint64(0xFFFFFu) << 32)) | ((vals[k] >> 32) << 32); }     // a range shrinks in place
            else dq = priority_deque<uint64, vec, cmp>(vec(dq.size(), store), false);                                                      // taken again, as hits[read_id] is: rebuilt
            if (dq.size() != sizes[k]) return -(int)k - 1;
            for (uint32 i = 0; i < sizes[k]; ++i) if (store[i] != states[pos + i]) return -(int)k - 1;
            pos += sizes[k];
        }
    }
    return 0;
}

// priority_deque(seq, constructed = false) over arbitrary arrays: what it leaves (the reference's make_interval_heap)
extern "C" void heapify_arrays(uint64_t* data, const uint32_t* sizes, int n_cases, uint32_t stride)
{
    typedef vector_view<uint64*> vec;
    for (int c = 0; c < n_cases; ++c)
        priority_deque<uint64, vec, cmp> dq(vec(sizes[c], reinterpret_cast<uint64*>(data) + uint64(c) * stride), false);
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

// ---- the bit-vector banded edit distance (alignment.h: EditDistanceAligner<TYPE, MyersTag<A>>) and infix sets (strings/infix.h) ----
#include <nvbio/alignment/alignment.h>
#include <nvbio/strings/infix.h>
#include <nvbio/basic/packedstream.h>
/// n jobs: pattern i = symbols [pb[i], pb[i] + pl[i]) of the byte string `pat`, text i likewise; band 31 / 15, alphabet 5 / 4.  A job is run
/// through an InfixSet over each string (the form examples/fmmap/fmmap.cu:337-345 hands its strings over in)
template <uint32 BAND, aln::AlignmentType TYPE, uint32 A, typename sink_t>
static void myers_jobs(const uint8* pat, const uint8* txt, const uint2* pc, const uint2* tc, uint32 n, int32 min_score, int32* score, uint32* sink)
{
    const InfixSet<const uint8*, const uint2*> patterns(n, pat, pc), texts(n, txt, tc);
    for (uint32 i = 0; i < n; ++i)
    {
        sink_t s;
        aln::banded_alignment_score<BAND>(aln::make_edit_distance_aligner<TYPE, aln::MyersTag<A> >(), patterns[i], texts[i], min_score, s);
        score[i] = s.score; sink[2 * i] = s.sink.x; sink[2 * i + 1] = s.sink.y;
    }
}
extern "C" int banded_myers(uint32 band, int type, uint32 alphabet, int sink_bits, const uint8* pat, const uint8* txt, const uint32* pc, const uint32* tc, uint32 n,
                            int32 min_score, int32* score, uint32* sink)
{
    const uint2* p2 = reinterpret_cast<const uint2*>(pc); const uint2* t2 = reinterpret_cast<const uint2*>(tc);
#define MYERS_CASE(B, T, AL, S) if (band == B && type == int(T) && alphabet == AL && sink_bits == int(8 * sizeof(S::score_type))) { myers_jobs<B, T, AL, S>(pat, txt, p2, t2, n, min_score, score, sink); return 0; }
    MYERS_CASE(31u, aln::SEMI_GLOBAL, 5u, aln::BestSink<int16>)
    MYERS_CASE(31u, aln::SEMI_GLOBAL, 5u, aln::BestSink<int32>)
    MYERS_CASE(31u, aln::GLOBAL, 5u, aln::BestSink<int32>)
    MYERS_CASE(15u, aln::SEMI_GLOBAL, 4u, aln::BestSink<int32>)
    MYERS_CASE(15u, aln::GLOBAL, 4u, aln::BestSink<int32>)
    MYERS_CASE(7u, aln::SEMI_GLOBAL, 2u, aln::BestSink<int32>)
#undef MYERS_CASE
    return -1;
}
/// infixes of a string set: out[k] = symbols of infix k, concatenated; returns the total
extern "C" uint32 read_set_infixes(const uint32* words, const uint32* offsets, uint32 n_strings, const uint32* coords4, uint32 n_infixes, uint8* out, uint32* ids)
{
    typedef PackedStream<const uint32*, uint8, 4, true> stream_type;
    typedef ConcatenatedStringSet<stream_type, const uint32*> set_type;
    const set_type set(n_strings, stream_type(words), offsets);
    const InfixSet<set_type, const string_set_infix_coord_type*> infixes(n_infixes, set, reinterpret_cast<const string_set_infix_coord_type*>(coords4));
    uint32 total = 0;
    for (uint32 k = 0; k < infixes.size(); ++k)
    {
        const InfixSet<set_type, const string_set_infix_coord_type*>::string_type s = infixes[k];
        ids[k] = string_id(s);
        if (length(s) != infix_end(s) - infix_begin(s) || s.begin()[0] != s[0]) return uint32(-1);
        for (uint32 i = 0; i < length(s); ++i) out[total++] = s[i];
    }
    return total;
}

// ---- the randomized hit selection the way nvBowtie's kernels run it over the drop-in SumTree (select.cu:86-102, select_inl.h:146-175, 200-252):
//      leaves 1 / delta^2, LCG draws, sample(), pop_front, zero an emptied hit's leaf -- the picks of `rounds` rounds for one read
#include <nvbio/basic/sum_tree.h>
extern "C" uint32 sum_tree_picks(uint32 n_hits, const uint32* begins, uint32* deltas /* 20-bit, modified */, uint32 rseed, uint32 rounds, float* cells /* node_count(n_hits) */,
                                 uint32* out_rows, uint32* out_hit)
{
    typedef SumTree<float*> ProbTree;
    std::vector<uint32> begin(begins, begins + n_hits);
    for (uint32 i = 0; i < n_hits; ++i) cells[i] = 1.0f / (float(deltas[i]) * float(deltas[i]));
    ProbTree tree(n_hits ? n_hits : 1u, cells);
    tree.setup();
    uint32 made = 0;
    for (uint32 r = 0; r < rounds; ++r)
    {
        if (tree.sum() <= 0.0f) break;
        uint32 hit_id = 0; bool found = false;
        for (uint32 i = 0; i < 10 && !found; ++i)
        {
            rseed = 1664525u * rseed + 1013904223u;
            const float rf = float(rseed) / float(0xFFFFFFFFu);
            const uint32 id = sample(tree, rf);
            if (deltas[id] != 0u) { hit_id = id; found = true; }
        }
        if (deltas[hit_id] == 0u) break;
        out_rows[made] = begin[hit_id]++; out_hit[made] = hit_id; ++made;
        deltas[hit_id] = (deltas[hit_id] - 1u) & 0xFFFFFu;
        if (deltas[hit_id] == 0u) tree.set(hit_id, 0.0f);
    }
    return made;
}
