// tests/compat/fm_callers.hip -- caller kernels of the FM-index written against the reference's template interface
// (the way nvbio-test/fmindex_test.cu:63-92 and rank_test.cu:55-86 use it): a kernel that backward-searches a slice of
// the packed genome with match() and locates the first row of its range, and one that issues rank / rank4 point queries;
// over 32- and 64-bit indices, with separate bwt / occ arrays and with the interleaved uint4 production layout seen
// through deinterleaved_iterator (nvbio/io/fmindex/fmindex.h:159-174).  Compiled with `hipcc -I include/nvbio_hip/compat`.
#include <nvbio/basic/types.h>
#include <nvbio/basic/numbers.h>
#include <nvbio/basic/cached_iterator.h>
#include <nvbio/basic/packedstream.h>
#include <nvbio/basic/deinterleaved_iterator.h>
#include <nvbio/basic/cuda/ldg.h>
#include <nvbio/fmindex/bwt.h>
#include <nvbio/fmindex/ssa.h>
#include <nvbio/fmindex/fmindex.h>

using namespace nvbio;

template <typename FMIndexType, typename word_type>
__global__ void search_and_locate_kernel(const uint32 n_queries, const uint32 query_len, const word_type* genome_words, const FMIndexType fmi,
                                         const uint32* starts, typename FMIndexType::index_type* ranges, typename FMIndexType::index_type* positions,
                                         typename FMIndexType::index_type* reverse_ranges)
{
    typedef typename FMIndexType::index_type index_type;
    typedef typename FMIndexType::range_type range_type;
    const uint32 q = threadIdx.x + blockIdx.x * blockDim.x;
    if (q >= n_queries) return;

    typedef const_cached_iterator<const word_type*>                    cached_words;
    typedef PackedStream<cached_words, uint8, 2, true, index_type>     genome_string;
    const genome_string genome((cached_words(genome_words)));

    const range_type range = match(fmi, genome + starts[q], query_len);
    ranges[2 * q] = range.x; ranges[2 * q + 1] = range.y;
    positions[q] = (range.x <= range.y) ? locate(fmi, range.x) : index_type(-1);
    if (range.x <= range.y)
    {
        // the two-pass form must agree with locate()
        const range_type it = locate_ssa_iterator(fmi, range.y);
        if (lookup_ssa_iterator(fmi, it) != locate(fmi, range.y)) positions[q] = index_type(-2);
    }
    const range_type rrange = match_reverse(fmi, genome + starts[q], query_len);
    reverse_ranges[2 * q] = rrange.x; reverse_ranges[2 * q + 1] = rrange.y;
}

template <typename FMIndexType>
__global__ void rank_kernel(const uint32 n, const FMIndexType fmi, const typename FMIndexType::index_type* rows, const uint8* symbols,
                            typename FMIndexType::index_type* out, typename FMIndexType::index_type* out4)
{
    typedef typename FMIndexType::index_type index_type;
    const uint32 q = threadIdx.x + blockIdx.x * blockDim.x;
    if (q >= n) return;
    out[q] = rank(fmi, rows[q], symbols[q]);
    const typename FMIndexType::rank_dictionary_type::vec4_type r4 = rank4(fmi, rows[q]);
    out4[4 * q] = r4.x; out4[4 * q + 1] = r4.y; out4[4 * q + 2] = r4.z; out4[4 * q + 3] = r4.w;
    // the range form resolves both ends like two point queries
    const typename FMIndexType::range_type rr = rank(fmi, make_vector(index_type(rows[q] - 1), rows[q]), symbols[q]);
    if (rr.y != out[q] || rr.x != rank(fmi, index_type(rows[q] - 1), symbols[q])) out[q] = index_type(-7);
}

// the range forms a caller of nvBowtie's shape uses (mapping_inl.h:83-97, 128-220): rank4 / rank_all of both ends of an SA range,
// read back with comp(); and a one-mismatch seed search built on them -- exact over query[0, len1), one substitution allowed in
// query[len1, len2), walking the query front to back (each symbol is PREPENDED to the matched string).  Slot 0 of a query's
// output holds the exact range, slot 1 + 3 * (i - len1) + k the range with position i replaced by the k-th other symbol; (1, 0) = none.
template <typename FMIndexType>
__global__ void rank4_range_kernel(const uint32 n, const FMIndexType fmi, const typename FMIndexType::index_type* lo_rows, const typename FMIndexType::index_type* hi_rows,
                                   typename FMIndexType::index_type* out_lo, typename FMIndexType::index_type* out_hi, uint32* agree)
{
    typedef typename FMIndexType::index_type                        index_type;
    typedef typename FMIndexType::rank_dictionary_type::vec4_type   vec4_type;
    const uint32 q = threadIdx.x + blockIdx.x * blockDim.x;
    if (q >= n) return;
    vec4_type lo, hi;
    rank4(fmi, make_vector(lo_rows[q], hi_rows[q]), &lo, &hi);
    for (uint32 c = 0; c < 4u; ++c) { out_lo[4 * q + c] = comp(lo, c); out_hi[4 * q + c] = comp(hi, c); }
    // rank_all must say the same, and each end must equal its point query
    typename FMIndexType::vector_type al, ah;
    rank_all(fmi, make_vector(lo_rows[q], hi_rows[q]), &al, &ah);
    const vec4_type pl = rank4(fmi, lo_rows[q]), ph = rank4(fmi, hi_rows[q]);
    bool ok = true;
    for (uint32 c = 0; c < 4u; ++c)
        ok = ok && al[c] == comp(lo, c) && ah[c] == comp(hi, c) && comp(pl, c) == comp(lo, c) && comp(ph, c) == comp(hi, c)
                && rank(fmi, make_vector(lo_rows[q], hi_rows[q]), uint8(c)).x == comp(lo, c) && rank(fmi, make_vector(lo_rows[q], hi_rows[q]), uint8(c)).y == comp(hi, c);
    agree[q] = ok ? 1u : 0u;
}

template <typename FMIndexType, typename Query>
NVBIO_FORCEINLINE NVBIO_DEVICE uint2 extend_exact(const FMIndexType& fmi, uint2 range, const Query query, const uint32 begin, const uint32 end)
{
    for (uint32 i = begin; i < end && range.x <= range.y; ++i)
    {
        const uint8 c = query[i];
        if (c > 3) return make_uint2(1u, 0u);
        const uint2 r = rank(fmi, make_uint2(range.x - 1u, range.y), c);
        range = make_uint2(fmi.L2(c) + r.x + 1u, fmi.L2(c) + r.y);
    }
    return range;
}
template <typename FMIndexType>
__global__ void one_mismatch_kernel(const uint32 n_queries, const uint32 len1, const uint32 len2, const uint32* genome_words, const FMIndexType fmi,
                                    const uint32* starts, uint32* out_ranges)
{
    const uint32 q = threadIdx.x + blockIdx.x * blockDim.x;
    if (q >= n_queries) return;
    typedef PackedStream<const uint32*, uint8, 2, true> genome_string;
    const genome_string query = genome_string(genome_words) + starts[q];
    const uint32 slots = 1u + 3u * (len2 - len1);
    uint32* out = out_ranges + 2ull * slots * q;
    for (uint32 s = 0; s < slots; ++s) { out[2 * s] = 1u; out[2 * s + 1] = 0u; }

    uint2 base = extend_exact(fmi, make_uint2(0u, fmi.length()), query, 0u, len1);
    for (uint32 i = len1; i < len2 && base.x <= base.y; ++i)
    {
        const uint8 c = query[i];
        uint4 lo, hi;
        rank4(fmi, make_uint2(base.x - 1u, base.y), &lo, &hi);
        uint32 k = 0;
        for (uint8 sub = 0; sub < 4; ++sub)
        {
            if (sub == c) continue;
            if (comp(hi, sub) > comp(lo, sub))
            {
                const uint2 r = extend_exact(fmi, make_uint2(fmi.L2(sub) + comp(lo, sub) + 1u, fmi.L2(sub) + comp(hi, sub)), query, i + 1u, len2);
                if (r.x <= r.y) { out[2 * (1u + 3u * (i - len1) + k)] = r.x; out[2 * (1u + 3u * (i - len1) + k) + 1] = r.y; }
            }
            ++k;
        }
        base = make_uint2(fmi.L2(c) + comp(lo, c) + 1u, fmi.L2(c) + comp(hi, c));
    }
    if (base.x <= base.y) { out[0] = base.x; out[1] = base.y; }
}

// separate arrays, as the reference's synthetic tests build them (fmindex_test.cu:418-600)
template <typename index_type>
struct SeparateLayout
{
    typedef PackedStream<const index_type*, uint8, 2, true, index_type>                 bwt_type;
    typedef rank_dictionary<2, 64, bwt_type, const index_type*, const uint32*>          rank_dict_type;
    typedef SSA_index_multiple_context<16, const index_type*>                           ssa_type;
    typedef fm_index<rank_dict_type, ssa_type>                                          fm_index_type;
    static fm_index_type make(index_type n, index_type primary, const index_type* L2, const void* bwt, const void* occ, const uint32* count_table, const index_type* ssa)
    {
        return fm_index_type(n, primary, L2, rank_dict_type(bwt_type((const index_type*)bwt), (const index_type*)occ, count_table), ssa_type(ssa));
    }
};
// the production layout: one uint4 of bwt words and one uint4 of counters per 64 symbols, interleaved
struct InterleavedLayout
{
    typedef cuda::ldg_pointer<uint4>                                bwt_occ_type;
    typedef deinterleaved_iterator<2, 0, bwt_occ_type>              bwt_words;
    typedef deinterleaved_iterator<2, 1, bwt_occ_type>              occ_type;
    typedef PackedStream<bwt_words, uint8, 2, true>                 bwt_type;
    typedef rank_dictionary<2, 64, bwt_type, occ_type, cuda::ldg_pointer<uint32> >     rank_dict_type;
    typedef SSA_index_multiple_context<16, cuda::ldg_pointer<uint32> >                  ssa_type;
    typedef fm_index<rank_dict_type, ssa_type>                      fm_index_type;
    static fm_index_type make(uint32 n, uint32 primary, const uint32* L2, const void* bwt_occ, const void*, const uint32* count_table, const uint32* ssa)
    {
        const bwt_occ_type base((const uint4*)bwt_occ);
        return fm_index_type(n, primary, L2, rank_dict_type(bwt_type(bwt_words(base)), occ_type(base), cuda::ldg_pointer<uint32>(count_table)),
                             ssa_type(cuda::ldg_pointer<uint32>(ssa)));
    }
};

#define API extern "C" __attribute__((visibility("default")))

template <typename Layout, typename index_type>
static int run_search(index_type n, index_type primary, const index_type* L2, const void* bwt, const void* occ, const uint32* count_table, const index_type* ssa,
                      uint32 n_queries, uint32 query_len, const void* genome_words, const uint32* starts, index_type* ranges, index_type* positions, index_type* rranges)
{
    const typename Layout::fm_index_type fmi = Layout::make(n, primary, L2, bwt, occ, count_table, ssa);
    hipLaunchKernelGGL((search_and_locate_kernel<typename Layout::fm_index_type, index_type>), dim3((n_queries + 127u) / 128u), dim3(128), 0, 0,
                       n_queries, query_len, (const index_type*)genome_words, fmi, starts, ranges, positions, rranges);
    return int(hipDeviceSynchronize());
}
template <typename Layout, typename index_type>
static int run_rank(index_type n, index_type primary, const index_type* L2, const void* bwt, const void* occ, const uint32* count_table,
                    uint32 n_queries, const index_type* rows, const uint8* symbols, index_type* out, index_type* out4)
{
    const typename Layout::fm_index_type fmi = Layout::make(n, primary, L2, bwt, occ, count_table, (const index_type*)NULL);
    hipLaunchKernelGGL((rank_kernel<typename Layout::fm_index_type>), dim3((n_queries + 127u) / 128u), dim3(128), 0, 0, n_queries, fmi, rows, symbols, out, out4);
    return int(hipDeviceSynchronize());
}

// layout: 0 = separate arrays with 32-bit words / indices, 1 = separate arrays with 64-bit words / indices, 2 = interleaved uint4 (32-bit)
API int compat_fm_search(int layout, unsigned long long n, unsigned long long primary, const void* L2, const void* bwt, const void* occ, const unsigned* count_table,
                         const void* ssa, unsigned n_queries, unsigned query_len, const void* genome_words, const unsigned* starts, void* ranges, void* positions, void* rranges)
{
    if (layout == 0) return run_search< SeparateLayout<uint32>, uint32 >(uint32(n), uint32(primary), (const uint32*)L2, bwt, occ, count_table, (const uint32*)ssa, n_queries, query_len, genome_words, starts, (uint32*)ranges, (uint32*)positions, (uint32*)rranges);
    if (layout == 1) return run_search< SeparateLayout<uint64>, uint64 >(uint64(n), uint64(primary), (const uint64*)L2, bwt, occ, count_table, (const uint64*)ssa, n_queries, query_len, genome_words, starts, (uint64*)ranges, (uint64*)positions, (uint64*)rranges);
    if (layout == 2) return run_search< InterleavedLayout, uint32 >(uint32(n), uint32(primary), (const uint32*)L2, bwt, occ, count_table, (const uint32*)ssa, n_queries, query_len, genome_words, starts, (uint32*)ranges, (uint32*)positions, (uint32*)rranges);
    return -1;
}
API int compat_fm_rank(int layout, unsigned long long n, unsigned long long primary, const void* L2, const void* bwt, const void* occ, const unsigned* count_table,
                       unsigned n_queries, const void* rows, const unsigned char* symbols, void* out, void* out4)
{
    if (layout == 0) return run_rank< SeparateLayout<uint32>, uint32 >(uint32(n), uint32(primary), (const uint32*)L2, bwt, occ, count_table, n_queries, (const uint32*)rows, symbols, (uint32*)out, (uint32*)out4);
    if (layout == 1) return run_rank< SeparateLayout<uint64>, uint64 >(uint64(n), uint64(primary), (const uint64*)L2, bwt, occ, count_table, n_queries, (const uint64*)rows, symbols, (uint64*)out, (uint64*)out4);
    if (layout == 2) return run_rank< InterleavedLayout, uint32 >(uint32(n), uint32(primary), (const uint32*)L2, bwt, occ, count_table, n_queries, (const uint32*)rows, symbols, (uint32*)out, (uint32*)out4);
    return -1;
}

template <typename Layout, typename index_type>
static int run_rank4_range(index_type n, index_type primary, const index_type* L2, const void* bwt, const void* occ, const uint32* count_table,
                           uint32 n_queries, const index_type* lo_rows, const index_type* hi_rows, index_type* out_lo, index_type* out_hi, uint32* agree)
{
    const typename Layout::fm_index_type fmi = Layout::make(n, primary, L2, bwt, occ, count_table, (const index_type*)NULL);
    hipLaunchKernelGGL((rank4_range_kernel<typename Layout::fm_index_type>), dim3((n_queries + 127u) / 128u), dim3(128), 0, 0, n_queries, fmi, lo_rows, hi_rows, out_lo, out_hi, agree);
    return int(hipDeviceSynchronize());
}
API int compat_fm_rank4_range(int layout, unsigned long long n, unsigned long long primary, const void* L2, const void* bwt, const void* occ, const unsigned* count_table,
                              unsigned n_queries, const void* lo_rows, const void* hi_rows, void* out_lo, void* out_hi, unsigned* agree)
{
    if (layout == 0) return run_rank4_range< SeparateLayout<uint32>, uint32 >(uint32(n), uint32(primary), (const uint32*)L2, bwt, occ, count_table, n_queries, (const uint32*)lo_rows, (const uint32*)hi_rows, (uint32*)out_lo, (uint32*)out_hi, agree);
    if (layout == 1) return run_rank4_range< SeparateLayout<uint64>, uint64 >(uint64(n), uint64(primary), (const uint64*)L2, bwt, occ, count_table, n_queries, (const uint64*)lo_rows, (const uint64*)hi_rows, (uint64*)out_lo, (uint64*)out_hi, agree);
    if (layout == 2) return run_rank4_range< InterleavedLayout, uint32 >(uint32(n), uint32(primary), (const uint32*)L2, bwt, occ, count_table, n_queries, (const uint32*)lo_rows, (const uint32*)hi_rows, (uint32*)out_lo, (uint32*)out_hi, agree);
    return -1;
}
API int compat_fm_one_mismatch(int layout, unsigned n, unsigned primary, const unsigned* L2, const void* bwt, const void* occ, const unsigned* count_table,
                               unsigned n_queries, unsigned len1, unsigned len2, const unsigned* genome_words, const unsigned* starts, unsigned* out_ranges)
{
    if (layout == 0)
    {
        const SeparateLayout<uint32>::fm_index_type fmi = SeparateLayout<uint32>::make(n, primary, L2, bwt, occ, count_table, (const uint32*)NULL);
        hipLaunchKernelGGL((one_mismatch_kernel<SeparateLayout<uint32>::fm_index_type>), dim3((n_queries + 127u) / 128u), dim3(128), 0, 0, n_queries, len1, len2, genome_words, fmi, starts, out_ranges);
    }
    else if (layout == 2)
    {
        const InterleavedLayout::fm_index_type fmi = InterleavedLayout::make(n, primary, L2, bwt, occ, count_table, (const uint32*)NULL);
        hipLaunchKernelGGL((one_mismatch_kernel<InterleavedLayout::fm_index_type>), dim3((n_queries + 127u) / 128u), dim3(128), 0, 0, n_queries, len1, len2, genome_words, fmi, starts, out_ranges);
    }
    else return -1;
    return int(hipDeviceSynchronize());
}

// the same templates on the host (fmindex_test.cu:376-413 runs its cpu alignment loop this way)
API int compat_fm_search_host(unsigned n, unsigned primary, const unsigned* L2, const unsigned* bwt, const unsigned* occ, const unsigned* ssa,
                              unsigned n_queries, unsigned query_len, const unsigned* genome_words, const unsigned* starts, unsigned* ranges, unsigned* positions)
{
    uint32 count_table[256];
    gen_bwt_count_table(count_table);
    const SeparateLayout<uint32>::fm_index_type fmi = SeparateLayout<uint32>::make(n, primary, L2, bwt, occ, count_table, ssa);
    const PackedStream<const uint32*, uint8, 2, true> genome(genome_words);
    for (uint32 q = 0; q < n_queries; ++q)
    {
        const uint2 range = match(fmi, genome + starts[q], query_len);
        ranges[2 * q] = range.x; ranges[2 * q + 1] = range.y;
        positions[q] = range.x <= range.y ? locate(fmi, range.x) : 0xFFFFFFFFu;
    }
    return 0;
}

// ---- the line-native records under the per-thread functions (fmindex/line_native.h): the same walks on an index with and without them ----
// A walk in the shape of nvBowtie's match_range (mapping_inl.h:83-97): the index is held BY VALUE for the length of the loop, one
// rank(index, (x - 1, y), c) per symbol, front to back; every raw pair of counts it gets is written out.
template <typename FMType, typename Query>
__device__ uint2 traced_walk(uint2 range, const FMType index, const Query query, const uint32 begin, const uint32 end, uint32* trace)
{
    for (uint32 i = begin; i < end && range.x <= range.y; ++i)
    {
        const uint8 c = query[i];
        const uint2 cnts = rank(index, make_uint2(range.x - 1u, range.y), c);
        trace[2u * i] = cnts.x; trace[2u * i + 1u] = cnts.y;
        range.x = index.L2(c) + cnts.x + 1u;
        range.y = index.L2(c) + cnts.y;
    }
    return range;
}
/// mode 0: one walk per lane; mode 1: two walks INTERLEAVED on one index object (each call finds the other walk's kept step); mode 2: after every
/// step the same query is asked again (a kept step must not answer a repeated range wrongly); mode 3: ONE index object in __shared__ memory walked
/// by every lane of the block at once; mode 4: one object in global memory behind a pointer, walked by every lane of the grid (the reference's
/// fm_index is a read-only view: a shared object must answer as a private copy does, whatever the other lanes ask of it meanwhile);
/// rows: locate_ssa_iterator + lookup
template <typename FMIndexType, typename Query>
__device__ uint2 traced_walk_shared(uint2 range, const FMIndexType* index, const Query query, const uint32 end, uint32* trace)
{
    for (uint32 i = 0; i < end && range.x <= range.y; ++i)
    {
        const uint8 c = query[i];
        const uint2 cnts = rank(*index, make_uint2(range.x - 1u, range.y), c);          // *index: the one object every lane uses
        trace[2u * i] = cnts.x; trace[2u * i + 1u] = cnts.y;
        range.x = index->L2(c) + cnts.x + 1u;
        range.y = index->L2(c) + cnts.y;
    }
    return range;
}
template <typename FMIndexType>
__global__ void native_walk_kernel(const uint32 n_queries, const uint32 query_len, const uint32* genome_words, const FMIndexType fmi, const uint32 mode,
                                   const uint32* starts, uint32* ranges, uint32* trace, const uint32 n_rows, const uint32* rows, uint32* iterators, uint32* positions,
                                   FMIndexType* global_index)
{
    const uint32 q = threadIdx.x + blockIdx.x * blockDim.x;
    typedef PackedStream<const uint32*, uint8, 2, true> genome_string;
    const genome_string genome(genome_words);
    __shared__ unsigned char s_index_bytes[sizeof(FMIndexType)] __attribute__((aligned(16)));
    FMIndexType* s_index = reinterpret_cast<FMIndexType*>(s_index_bytes);
    if (mode == 3u)
    {
        if (threadIdx.x == 0u) *s_index = fmi;
        __syncthreads();
    }
    if (q < n_queries && (mode == 3u || mode == 4u))
    {
        uint32* tr = trace + uint64(q) * 2u * query_len;
        for (uint32 i = 0; i < 2u * query_len; ++i) tr[i] = 0xFFFFFFFFu;
        const uint2 range = traced_walk_shared(make_uint2(0u, fmi.length()), mode == 3u ? s_index : global_index, genome + starts[q], query_len, tr);
        ranges[2u * q] = range.x; ranges[2u * q + 1u] = range.y;
    }
    else if (q < n_queries)
    {
        uint32* tr = trace + uint64(q) * 2u * query_len;
        for (uint32 i = 0; i < 2u * query_len; ++i) tr[i] = 0xFFFFFFFFu;
        uint2 range = make_uint2(0u, fmi.length());
        if (mode == 0u) range = traced_walk(range, fmi, genome + starts[q], 0u, query_len, tr);
        else if (mode == 1u)
        {
            const FMIndexType index = fmi;
            const genome_string a = genome + starts[q], b = genome + starts[(q + 1u) % n_queries];
            uint2 other = range;
            for (uint32 i = 0; i < query_len && range.x <= range.y; ++i)
            {
                const uint8 c = a[i];
                const uint2 cnts = rank(index, make_uint2(range.x - 1u, range.y), c);
                tr[2u * i] = cnts.x; tr[2u * i + 1u] = cnts.y;
                range.x = index.L2(c) + cnts.x + 1u; range.y = index.L2(c) + cnts.y;
                if (other.x <= other.y)
                {
                    const uint8 d = b[i];
                    const uint2 o = rank(index, make_uint2(other.x - 1u, other.y), d);
                    other.x = index.L2(d) + o.x + 1u; other.y = index.L2(d) + o.y;
                }
            }
        }
        else
        {
            const FMIndexType index = fmi;
            const genome_string a = genome + starts[q];
            for (uint32 i = 0; i < query_len && range.x <= range.y; ++i)
            {
                const uint8 c = a[i];
                const uint2 cnts = rank(index, make_uint2(range.x - 1u, range.y), c);
                const uint2 again = rank(index, make_uint2(range.x - 1u, range.y), uint8((c + 1u) & 3u));     // same range, another symbol
                const uint2 once_more = rank(index, make_uint2(range.x - 1u, range.y), c);
                tr[2u * i] = cnts.x; tr[2u * i + 1u] = (once_more.x == cnts.x && once_more.y == cnts.y) ? cnts.y : 0xDEADBEEFu;
                (void)again;
                range.x = index.L2(c) + cnts.x + 1u; range.y = index.L2(c) + cnts.y;
            }
        }
        ranges[2u * q] = range.x; ranges[2u * q + 1u] = range.y;
    }
    if (q < n_rows)
    {
        const uint2 it = locate_ssa_iterator(fmi, rows[q]);
        iterators[2u * q] = it.x; iterators[2u * q + 1u] = it.y;
        positions[q] = lookup_ssa_iterator(fmi, it);
    }
}
API int compat_fm_native_walk(unsigned n, unsigned primary, const unsigned* L2, const void* bwt_occ, const unsigned* count_table, const unsigned* ssa,
                              const unsigned* line_native /* NULL = reference layout */, unsigned mode, unsigned n_queries, unsigned query_len,
                              const unsigned* genome_words, const unsigned* starts, unsigned* ranges, unsigned* trace, unsigned n_rows, const unsigned* rows,
                              unsigned* iterators, unsigned* positions)
{
    InterleavedLayout::fm_index_type fmi = InterleavedLayout::make(n, primary, L2, bwt_occ, NULL, count_table, ssa);
    fmi.set_line_native(line_native);
    const uint32 m = n_queries > n_rows ? n_queries : n_rows;
    InterleavedLayout::fm_index_type* d_index = NULL;              // mode 4: the object itself in device memory
    if (hipMalloc(reinterpret_cast<void**>(&d_index), sizeof(fmi)) != hipSuccess) return 2;
    if (hipMemcpy(d_index, &fmi, sizeof(fmi), hipMemcpyHostToDevice) != hipSuccess) return 2;
    hipLaunchKernelGGL((native_walk_kernel<InterleavedLayout::fm_index_type>), dim3((m + 127u) / 128u), dim3(128), 0, 0, n_queries, query_len, genome_words, fmi, mode,
                       starts, ranges, trace, n_rows, rows, iterators, positions, d_index);
    const int e = int(hipDeviceSynchronize());
    (void)hipFree(d_index);
    return e;
}
