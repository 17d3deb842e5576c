// tests/compat/filter_callers.hip -- a caller of FMIndexFilter written to the shape of examples/fmmap/fmmap.cu:92-97 (the
// Pipeline's FMIndexFilterDevice<fm_index_type> member) and :293-367 (rank the seed string-set, loop over batches of hits: locate,
// turn (index-pos, seed-id) into diagonals (text-pos - seed-pos, read-id), build the read / genome infix sets and score them with
// batch_banded_alignment_score<31> over an edit-distance aligner into BestSink<int16>).  It includes the reference's header names
// and is compiled with `hipcc -I include/nvbio_hip/compat`; the fm_index is the production layout exactly as
// nvbio/io/fmindex/fmindex.h:159-174 composes it.  The extern "C" entry points exist so that the Python tests can drive it.
#include <nvbio/basic/types.h>
#include <nvbio/basic/vector.h>
#include <nvbio/basic/packedstream.h>
#include <nvbio/basic/deinterleaved_iterator.h>
#include <nvbio/basic/cuda/ldg.h>
#include <nvbio/strings/string_set.h>
#include <nvbio/fmindex/bwt.h>
#include <nvbio/fmindex/ssa.h>
#include <nvbio/fmindex/fmindex.h>
#include <nvbio/fmindex/filter.h>
#include <nvbio/alignment/alignment.h>
#include <nvbio/alignment/batched.h>
#include <thrust/transform.h>
#include <string.h>

using namespace nvbio;

// the device fm-index type of nvbio::io::FMIndexDataDevice (io/fmindex/fmindex.h:159-174)
typedef cuda::ldg_pointer<uint4>                                                bwt_occ_type;
typedef deinterleaved_iterator<2, 0, bwt_occ_type>                              bwt_words_type;
typedef deinterleaved_iterator<2, 1, bwt_occ_type>                              occ_type;
typedef PackedStream<bwt_words_type, uint8, 2, true>                            bwt_type;
typedef rank_dictionary<2, 64, bwt_type, occ_type, cuda::ldg_pointer<uint32> >  rank_dict_type;
typedef SSA_index_multiple_context<16, cuda::ldg_pointer<uint32> >              ssa_type;
typedef fm_index<rank_dict_type, ssa_type>                                      fm_index_type;

// the pipeline state (fmmap.cu:88-98)
struct Pipeline
{
    typedef FMIndexFilterDevice<fm_index_type> fm_filter_type;
    fm_filter_type fm_filter;
};

// seed coordinates: (read id, begin, end) inside the read (the role of string_set_infix_coord_type)
struct seed_coord { uint32 read_id, begin, end; };

// transform an (index-pos, seed-id) hit into a diagonal (text-pos = index-pos - seed-pos, read-id)   (fmmap.cu:100-126)
struct hit_to_diagonal
{
    typedef uint2 argument_type;
    typedef uint2 result_type;
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE hit_to_diagonal(const seed_coord* _seed_coords) : seed_coords(_seed_coords) {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint2 operator()(const uint2 hit) const
    {
        const seed_coord seed = seed_coords[hit.y];
        return make_uint2(hit.x - seed.begin, seed.read_id);
    }
    const seed_coord* seed_coords;
};
// the read infixes of the hit diagonals (fmmap.cu:152-174)
struct read_infixes
{
    typedef uint2 argument_type;
    typedef uint2 result_type;
    NVBIO_HOST_DEVICE read_infixes(const uint32* _read_index) : read_index(_read_index) {}
    NVBIO_HOST_DEVICE uint2 operator()(const uint2 diagonal) const { return make_uint2(read_index[diagonal.y], read_index[diagonal.y + 1]); }
    const uint32* read_index;
};
// the genome infixes of the hit diagonals (fmmap.cu:176-212)
template <uint32 BAND_LEN>
struct genome_infixes
{
    typedef uint2 argument_type;
    typedef uint2 result_type;
    NVBIO_HOST_DEVICE genome_infixes(const uint32 _genome_len, const uint32* _read_index) : genome_len(_genome_len), read_index(_read_index) {}
    NVBIO_HOST_DEVICE uint2 operator()(const uint2 diagonal) const
    {
        const uint32 read_len = read_index[diagonal.y + 1] - read_index[diagonal.y];
        const uint32 text_pos = diagonal.x;
        const uint32 genome_begin = text_pos > BAND_LEN / 2 ? text_pos - BAND_LEN / 2 : 0u;
        const uint32 genome_end   = nvbio::min(genome_begin + read_len + BAND_LEN, genome_len);
        return make_uint2(genome_begin, genome_end);
    }
    const uint32 genome_len; const uint32* read_index;
};
// the seeds as infixes of the reads
struct seed_ranges
{
    typedef seed_coord argument_type;
    typedef uint2      result_type;
    NVBIO_HOST_DEVICE seed_ranges(const uint32* _read_index) : read_index(_read_index) {}
    NVBIO_HOST_DEVICE uint2 operator()(const seed_coord s) const { return make_uint2(read_index[s.read_id] + s.begin, read_index[s.read_id] + s.end); }
    const uint32* read_index;
};

#define API extern "C" __attribute__((visibility("default")))

struct fmmap_args
{
    uint32 n, primary; const uint32* L2 /* device */; const uint32* bwt_occ; const uint32* ssa; const uint32* count_table;
    const uint32* genome_words; uint32 genome_len;                           // 2-bit big-endian
    const uint32* read_words; const uint32* read_index; uint32 n_reads, max_read_len;     // 2-bit big-endian reads
    const seed_coord* seeds; uint32 n_seeds;
    uint32 batch_size; uint32 line_native;
    unsigned long long* out_n_hits;          // host
    uint2* out_diagonals; int16* out_scores; uint2* out_sinks; uint32 out_capacity;
    uint2* out_ranges; unsigned long long* out_ranks;
};

static Pipeline* g_pipeline = nullptr;

API int compat_fmmap(const fmmap_args* a, char* rank_path, char* locate_path, char* score_path)
{
    try {
        if (!g_pipeline) g_pipeline = new Pipeline();            // the filter is a long-lived member, as in fmmap (it keeps its buffers)
        Pipeline::fm_filter_type& fm_filter = g_pipeline->fm_filter;
        fm_filter.set_line_native(a->line_native != 0);
        const bwt_occ_type base((const uint4*)a->bwt_occ);
        const fm_index_type fm_index(a->n, a->primary, a->L2, rank_dict_type(bwt_type(bwt_words_type(base)), occ_type(base), cuda::ldg_pointer<uint32>(a->count_table)),
                                     ssa_type(cuda::ldg_pointer<uint32>(a->ssa)));

        typedef PackedStream<cuda::ldg_pointer<uint32>, uint8, 2, true> read_stream;
        typedef PackedStream<cuda::ldg_pointer<uint32>, uint8, 2, true> genome_string;
        const read_stream   reads((cuda::ldg_pointer<uint32>(a->read_words)));
        const genome_string genome((cuda::ldg_pointer<uint32>(a->genome_words)));

        // the seed string-set
        nvbio::vector<device_tag, uint2> seed_infix_coords(a->n_seeds);
        thrust::transform(thrust::device_ptr<const seed_coord>(a->seeds), thrust::device_ptr<const seed_coord>(a->seeds) + a->n_seeds, seed_infix_coords.begin(), seed_ranges(a->read_index));
        typedef nvbio::vector<device_tag, uint2>::const_iterator infix_iterator;
        const SparseStringSet<read_stream, const uint2*> seed_string_set(a->n_seeds, reads, nvbio::plain_view(seed_infix_coords));

        const uint32 batch_size = a->batch_size;
        typedef uint2 hit_type;
        nvbio::vector<device_tag, hit_type> hits(batch_size);

        // first step: rank the query seeds
        const uint64 n_hits = fm_filter.rank(fm_index, seed_string_set);
        strncpy(rank_path, fm_filter.last_path(), 15);
        *a->out_n_hits = n_hits;
        (void)hipMemcpy(a->out_ranges, fm_filter.ranges(), sizeof(uint2) * a->n_seeds, hipMemcpyDeviceToDevice);
        (void)hipMemcpy(a->out_ranks, fm_filter.ranks(), sizeof(uint64) * a->n_seeds, hipMemcpyDeviceToDevice);

        nvbio::vector<device_tag, aln::BestSink<int16> > sinks(batch_size);
        nvbio::vector<device_tag, uint2> genome_infix_coords(batch_size);
        nvbio::vector<device_tag, uint2> read_infix_coords(batch_size);
        static const uint32 BAND_LEN = 31;

        // loop through large batches of hits and locate & merge them
        for (uint64 hits_begin = 0; hits_begin < n_hits && hits_begin < a->out_capacity; hits_begin += batch_size)
        {
            const uint64 hits_end = nvbio::min(nvbio::min(hits_begin + batch_size, n_hits), uint64(a->out_capacity));
            fm_filter.locate(hits_begin, hits_end, hits.begin());
            strncpy(locate_path, fm_filter.last_path(), 15);
            (void)hipDeviceSynchronize();

            // (index-pos, seed-id) -> diagonals (text-pos = index-pos - seed-pos, read-id)
            thrust::transform(hits.begin(), hits.begin() + (hits_end - hits_begin), hits.begin(), hit_to_diagonal(a->seeds));
            thrust::transform(hits.begin(), hits.begin() + (hits_end - hits_begin), read_infix_coords.begin(), read_infixes(a->read_index));
            thrust::transform(hits.begin(), hits.begin() + (hits_end - hits_begin), genome_infix_coords.begin(), genome_infixes<BAND_LEN>(a->genome_len, a->read_index));

            const SparseStringSet<read_stream, const uint2*>   read_infix_set(uint32(hits_end - hits_begin), reads, nvbio::plain_view(read_infix_coords));
            const SparseStringSet<genome_string, const uint2*> genome_infix_set(uint32(hits_end - hits_begin), genome, nvbio::plain_view(genome_infix_coords));

            typedef aln::MyersTag<5u> myers_dna5_tag;
            typedef aln::EditDistanceAligner<aln::SEMI_GLOBAL, myers_dna5_tag> aligner_type;
            typedef aln::priv::StringSetAlignmentStream<aligner_type, SparseStringSet<read_stream, const uint2*>, SparseStringSet<genome_string, const uint2*>,
                                                        nvbio::vector<device_tag, aln::BestSink<int16> >::iterator> stream_type;
            strncpy(score_path, aln::priv::recognised<stream_type>::value ? "tuned" : "generic", 15);
            aln::batch_banded_alignment_score<BAND_LEN>(
                aln::make_edit_distance_aligner<aln::SEMI_GLOBAL, myers_dna5_tag>(),
                read_infix_set, genome_infix_set, sinks.begin(), aln::DeviceThreadScheduler(),
                a->max_read_len, a->max_read_len + BAND_LEN);
            (void)hipDeviceSynchronize();

            (void)hipMemcpy(a->out_diagonals + hits_begin, nvbio::plain_view(hits), sizeof(uint2) * (hits_end - hits_begin), hipMemcpyDeviceToDevice);
            // BestSink<int16> = { int16 score; uint2 sink } : unpack on the host side of this shim
            nvbio::vector<host_tag, aln::BestSink<int16> > h_sinks(sinks);
            std::vector<int16> sc(hits_end - hits_begin); std::vector<uint2> sk(hits_end - hits_begin);
            for (uint64 k = 0; k < hits_end - hits_begin; ++k) { sc[k] = h_sinks[k].score; sk[k] = h_sinks[k].sink; }
            (void)hipMemcpy(a->out_scores + hits_begin, sc.data(), sizeof(int16) * sc.size(), hipMemcpyHostToDevice);
            (void)hipMemcpy(a->out_sinks + hits_begin, sk.data(), sizeof(uint2) * sk.size(), hipMemcpyHostToDevice);
        }
        return int(hipDeviceSynchronize());
    } catch (const std::exception& e) { fprintf(stderr, "compat_fmmap: %s\n", e.what()); return -1; }
}

// the generic execution: the same filter over separate bwt / occ arrays with 64-bit coordinates (fmindex_test.cu:418-717 builds such indices)
typedef PackedStream<const uint64*, uint8, 2, true, uint64>                         bwt64_type;
typedef rank_dictionary<2, 64, bwt64_type, const uint64*, const uint32*>            rank_dict64_type;
typedef SSA_index_multiple_context<16, const uint64*>                               ssa64_type;
typedef fm_index<rank_dict64_type, ssa64_type>                                      fm_index64_type;

API int compat_filter64(unsigned long long n, unsigned long long primary, const uint64* L2, const uint64* bwt, const uint64* occ, const uint32* count_table, const uint64* ssa,
                        const uint32* read_words, const uint2* seed_ranges_dev, uint32 n_seeds, unsigned long long* out_n_hits, ulonglong2* out_ranges,
                        unsigned long long* out_ranks, ulonglong2* out_hits, uint32 out_capacity, char* path)
{
    try {
        const fm_index64_type fm_index(n, primary, L2, rank_dict64_type(bwt64_type(bwt), occ, count_table), ssa64_type(ssa));
        FMIndexFilterDevice<fm_index64_type> filter;
        typedef PackedStream<const uint32*, uint8, 2, true> read_stream;
        const SparseStringSet<read_stream, const uint2*> seeds(n_seeds, read_stream(read_words), seed_ranges_dev);
        const uint64 n_hits = filter.rank(fm_index, seeds);
        *out_n_hits = n_hits;
        (void)hipMemcpy(out_ranges, filter.ranges(), sizeof(ulonglong2) * n_seeds, hipMemcpyDeviceToDevice);
        (void)hipMemcpy(out_ranks, filter.ranks(), sizeof(uint64) * n_seeds, hipMemcpyDeviceToDevice);
        filter.locate(0, nvbio::min(n_hits, uint64(out_capacity)), out_hits);
        strncpy(path, filter.last_path(), 15);
        return int(hipDeviceSynchronize());
    } catch (const std::exception& e) { fprintf(stderr, "compat_filter64: %s\n", e.what()); return -1; }
}

// the host filter over host arrays (FMIndexFilterHost)
typedef PackedStream<const uint32*, uint8, 2, true>                                 bwt32_type;
typedef rank_dictionary<2, 64, bwt32_type, const uint32*, const uint32*>            rank_dict32_type;
typedef fm_index<rank_dict32_type, SSA_index_multiple_context<16, const uint32*> >  fm_index32_type;

API int compat_filter_host(uint32 n, uint32 primary, const uint32* L2, const uint32* bwt, const uint32* occ, const uint32* ssa,
                           const uint32* read_words, const uint2* seed_ranges_host, uint32 n_seeds, unsigned long long* out_n_hits, uint2* out_ranges,
                           unsigned long long* out_ranks, uint2* out_hits, uint32 out_capacity)
{
    uint32 count_table[256];
    gen_bwt_count_table(count_table);
    const fm_index32_type fm_index(n, primary, L2, rank_dict32_type(bwt32_type(bwt), occ, count_table), SSA_index_multiple_context<16, const uint32*>(ssa));
    FMIndexFilterHost<fm_index32_type> filter;
    typedef PackedStream<const uint32*, uint8, 2, true> read_stream;
    const SparseStringSet<read_stream, const uint2*> seeds(n_seeds, read_stream(read_words), seed_ranges_host);
    const uint64 n_hits = filter.rank(fm_index, seeds);
    *out_n_hits = n_hits;
    memcpy(out_ranges, filter.ranges(), sizeof(uint2) * n_seeds);
    memcpy(out_ranks, filter.ranks(), sizeof(uint64) * n_seeds);
    filter.locate(0, nvbio::min(n_hits, uint64(out_capacity)), out_hits);
    return 0;
}
