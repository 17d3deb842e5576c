// ref_basic.cpp -- test harness around REFERENCE code that compiles from its own sources with nothing but headers this
// image ships (test infrastructure only; built into oracle/_ref/libref_basic.so, never linked into the product).
//
// nvbio/basic/types.h includes CUDA's <vector_types.h> / <vector_functions.h>.  The image has no CUDA toolkit, but it does
// carry NVIDIA's own CUDA runtime headers inside the Triton wheel (triton/backends/nvidia/include) and rocThrust's tag
// headers under /opt/rocm/include; the Makefile puts those on the include path.  No header is written, replaced or
// restated for this build.  What that reaches (probed header by header): every reference header that does not pull
// nvbio/basic/numbers.h (which needs <cuda_fp16.h> -> <nv/target>, absent here).  On or next to the hot path those are
//   nvbio/basic/popcount.h        popc_2bit / hibits_2bit / popc_2bit_all: the counting inside rank()  
//   nvbio/fmindex/bwt.h           gen_sa (contrib/sais.h), gen_bwt_from_sa, gen_bwt_count_table: SA / BWT / primary conventions
//   nvbio/basic/priority_deque.h  the hit deque itself (push / top / bottom / pop_top / pop_bottom over the interval heap)
//   nvbio/basic/algorithms.h      upper_bound as FMIndexFilter::locate uses it
//   nvbio/io/bam_format.h         the BAM record header BamOutput writes
//   nvbio/basic/bnt.{h,cpp}       load_bns / save_bns of the .ann / .amb pair
// The alignment DP, fm_index, sum_tree and nvBowtie headers include numbers.h and stay unbuilt.
#include <cstdint>
#include <cstring>
#include <vector>
#include <nvbio/basic/types.h>
#include <nvbio/basic/popcount.h>
#include <nvbio/basic/priority_deque.h>
#include <nvbio/basic/vector_view.h>
#include <nvbio/basic/algorithms.h>
#include <nvbio/fmindex/bwt.h>
#include <nvbio/io/bam_format.h>
#include <nvbio/basic/bnt.h>

#define API extern "C" __attribute__((visibility("default")))

using namespace nvbio;

// ------------------------------------------------------------------ popcount.h
API void ref_popc_2bit(const uint32_t* x, const uint8_t* c, uint32_t n, uint32_t* out)
{ for (uint32_t i = 0; i < n; ++i) out[i] = popc_2bit(x[i], int(c[i])); }
API void ref_popc_2bit_prefix(const uint32_t* mask, const uint8_t* c, const uint32_t* i_mod, uint32_t n, uint32_t* out)
{ for (uint32_t i = 0; i < n; ++i) out[i] = popc_2bit(mask[i], int(c[i]), i_mod[i]); }
API void ref_hibits_2bit(const uint32_t* mask, const uint32_t* i_mod, uint32_t n, uint32_t* out)
{ for (uint32_t i = 0; i < n; ++i) out[i] = hibits_2bit(mask[i], i_mod[i]); }
API void ref_popc_2bit_all(const uint32_t* b, uint32_t n, uint32_t* out /* packed 4 x 8 bit */)
{
    uint32 table[256];
    gen_bwt_count_table(table);
    for (uint32_t i = 0; i < n; ++i) out[i] = popc_2bit_all(b[i], table);
}
API void ref_popc_2bit_all_prefix(const uint32_t* mask, const uint32_t* i_mod, uint32_t n, uint32_t* out)
{
    uint32 table[256];
    gen_bwt_count_table(table);
    for (uint32_t i = 0; i < n; ++i) out[i] = popc_2bit_all(mask[i], table, i_mod[i]);
}
/// rank on the interleaved production layout, composed from the reference's popcount functions exactly as
/// dispatch_rank<2,64,...,uint4,uint4>::run does (rank_dictionary_inl.h:502-513: counter + whole words + masked word)
API void ref_dict_rank(const uint32_t* bwt_occ, const uint32_t* idx, const uint8_t* c, uint32_t n, uint32_t* out)
{
    for (uint32_t q = 0; q < n; ++q)
    {
        const uint32 i = idx[q];
        if (i == 0xFFFFFFFFu) { out[q] = 0; continue; }
        const uint32 k = i >> 6, m = (i & 63u) >> 4;
        const uint32* rec = bwt_occ + 8u * k;
        uint32 r = rec[4 + c[q]];
        for (uint32 w = 0; w < m; ++w) r += popc_2bit(rec[w], int(c[q]));
        r += popc_2bit(rec[m], int(c[q]), ~i & 15u);
        out[q] = r;
    }
}

// ------------------------------------------------------------------ bwt.h
API void ref_gen_bwt_count_table(uint32_t* table) { gen_bwt_count_table(table); }
/// SA (n+1 rows, SA[0] = n) and BWT with the '$' row dropped; returns primary
API uint32_t ref_gen_sa_bwt(uint32_t n, const uint8_t* T, int32_t* SA, uint8_t* bwt /* n+1 scratch */)
{
    gen_sa(n, T, SA);
    std::vector<uint8_t> text(T, T + n);
    std::vector<uint8_t> b(n + 1, 0);
    const uint32 primary = gen_bwt_from_sa(n, text.data(), SA, b.data());
    std::memcpy(bwt, b.data(), n + 1);
    return primary;
}

// ------------------------------------------------------------------ priority_deque.h  (nvBowtie's hit deque)
namespace {
struct hit_compare {      // nvBowtie/bowtie2/cuda/seed_hit.h:235-244: ordered by the 20-bit range size
    bool operator()(const uint64_t f, const uint64_t s) const { return ((f >> 32) & 0xFFFFFu) > ((s >> 32) & 0xFFFFFu); }
};
typedef vector_view<uint64_t*> storage_type;
typedef priority_deque<uint64_t, storage_type, hit_compare> deque_type;
}
/// replay a sequence of operations on a deque living in `storage` (op 0 = push(value), 1 = pop_top, 2 = pop_bottom);
/// after every operation records size, top() and minimum() (0 when empty).  Returns the final size.
/// (bottom() itself is unusable in the reference: priority_deque.h declares it as `{ return bottom(); }`, an endless
/// self-call; nvBowtie never calls it.  minimum() is what it was meant to forward to.)
API uint32_t ref_priority_deque_replay(uint64_t* storage, uint32_t n_ops, const uint8_t* ops, const uint64_t* values,
                                       uint32_t* sizes, uint64_t* tops, uint64_t* bottoms)
{
    deque_type deque(storage_type(0u, storage), false);
    for (uint32_t i = 0; i < n_ops; ++i)
    {
        if (ops[i] == 0) deque.push(values[i]);
        else if (ops[i] == 1) { if (!deque.empty()) deque.pop_top(); }
        else { if (!deque.empty()) deque.pop_bottom(); }
        sizes[i] = uint32_t(deque.size());
        tops[i] = deque.empty() ? 0ull : deque.top();
        bottoms[i] = deque.empty() ? 0ull : deque.minimum();
    }
    return uint32_t(deque.size());
}

// ------------------------------------------------------------------ algorithms.h
API void ref_upper_bound_u64(const uint64_t* slots, uint32_t n, const uint64_t* keys, uint32_t n_keys, uint32_t* out)
{ for (uint32_t i = 0; i < n_keys; ++i) out[i] = uint32_t(upper_bound(keys[i], slots, n) - slots); }

// ------------------------------------------------------------------ bam_format.h
/// the fixed part of a BAM alignment record as io::BAM_alignment lays it out (bam_format.h:60-71): fields[0..9] =
/// block_size, refID, pos, bin, mapq, l_read_name, flag, n_cigar_op, l_seq, next_refID, next_pos, tlen
API int ref_bam_alignment_fields(const uint8_t* record, int32_t* fields /* 12 */)
{
    io::BAM_alignment a;
    std::memcpy(&a, record, sizeof(a));
    fields[0] = a.block_size; fields[1] = a.refID; fields[2] = a.pos;
    fields[3] = int32_t(a.bin_mq_nl >> 16); fields[4] = int32_t((a.bin_mq_nl >> 8) & 0xFF); fields[5] = int32_t(a.bin_mq_nl & 0xFF);
    fields[6] = int32_t(a.flag_nc >> 16); fields[7] = int32_t(a.flag_nc & 0xFFFF);
    fields[8] = a.l_seq; fields[9] = a.next_refID; fields[10] = a.next_pos; fields[11] = a.tlen;
    return int(sizeof(a));
}

// ------------------------------------------------------------------ bnt.cpp
/// load_bns(prefix): the counts, and per sequence offset / length / n_ambs / gi + names packed with '\n'
API int ref_load_bns(const char* prefix, int64_t* l_pac, int32_t* n_seqs, uint32_t* seed, int32_t* n_holes,
                     int64_t* offsets, int32_t* lengths, int32_t* n_ambs, uint32_t* gis, char* names, uint32_t names_cap,
                     int64_t* hole_offsets, int32_t* hole_lengths, char* hole_chars, uint32_t max_seqs, uint32_t max_holes)
{
    BNTSeq bns;
    try { load_bns(bns, prefix); } catch (...) { return 1; }
    *l_pac = bns.l_pac; *n_seqs = bns.n_seqs; *seed = bns.seed; *n_holes = bns.n_holes;
    if (uint32_t(bns.n_seqs) > max_seqs || uint32_t(bns.n_holes) > max_holes) return 2;
    std::string all;
    for (int32_t i = 0; i < bns.n_seqs; ++i)
    {
        offsets[i] = bns.anns_data[i].offset; lengths[i] = bns.anns_data[i].len; n_ambs[i] = bns.anns_data[i].n_ambs; gis[i] = bns.anns_data[i].gi;
        all += bns.anns_info[i].name + "\t" + bns.anns_info[i].anno + "\n";
    }
    if (all.size() + 1 > names_cap) return 3;
    std::memcpy(names, all.c_str(), all.size() + 1);
    for (int32_t i = 0; i < bns.n_holes; ++i) { hole_offsets[i] = bns.ambs[i].offset; hole_lengths[i] = bns.ambs[i].len; hole_chars[i] = bns.ambs[i].amb; }
    return 0;
}
/// save_bns of a BNTSeq described by arrays (so that the product's reader can be run on files the reference wrote)
API int ref_save_bns(const char* prefix, int64_t l_pac, uint32_t seed, int32_t n_seqs, const int64_t* offsets, const int32_t* lengths,
                     const int32_t* n_ambs, const char* const* names, const char* const* annos,
                     int32_t n_holes, const int64_t* hole_offsets, const int32_t* hole_lengths, const char* hole_chars)
{
    BNTSeq bns;
    bns.l_pac = l_pac; bns.seed = seed; bns.n_seqs = n_seqs; bns.n_holes = n_holes;
    bns.anns_info.resize(n_seqs); bns.anns_data.resize(n_seqs); bns.ambs.resize(n_holes);
    for (int32_t i = 0; i < n_seqs; ++i)
    {
        bns.anns_info[i].name = names[i]; bns.anns_info[i].anno = annos[i];
        bns.anns_data[i].offset = offsets[i]; bns.anns_data[i].len = lengths[i]; bns.anns_data[i].n_ambs = n_ambs[i]; bns.anns_data[i].gi = 0;
    }
    for (int32_t i = 0; i < n_holes; ++i) { bns.ambs[i].offset = hole_offsets[i]; bns.ambs[i].len = hole_lengths[i]; bns.ambs[i].amb = hole_chars[i]; }
    try { save_bns(bns, prefix); } catch (...) { return 1; }
    return 0;
}
