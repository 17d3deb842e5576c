// compat/nvbio/strings/seeds.h -- seed enumeration (nvbio/strings/seeds.h:45-126, seeds_inl.h:171-229): given a seeding functor
// (how many seeds a string of a given length yields, and where the i-th lies), fill a vector with the coordinates of every seed of
// a string or of every string of a set.  examples/fmmap/fmmap.cu:139-142 calls enumerate_string_set_seeds with
// uniform_seeds_functor<> and a device vector of string_set_infix_coord_type.
//
// Own execution: the per-string seed counts are scanned (exclusive), then ONE work item per string writes that string's seeds at its
// scanned offset -- a few consecutive 16-byte stores per item, no search over the scan.  The coordinate type of the output vector decides
// what is kept of (string id, begin, end): a scalar keeps the first number, 2 / 4 components the first two / all (priv::seed_coord<>).
#pragma once
#include "../basic/types.h"
#include "../basic/vector.h"
#include "../fmindex/rank_dictionary.h"     // vector_traits
#include "string_set.h"
#include "infix.h"
#if defined(__HIPCC__)
#include <thrust/scan.h>
#include <thrust/for_each.h>
#include <thrust/transform.h>
#include <thrust/iterator/counting_iterator.h>
#include <thrust/execution_policy.h>
#endif

namespace nvbio {

/// seeds of one length taken every `interval` symbols (seeds.h:94-126)
template <typename index_type = uint32>
struct uniform_seeds_functor
{
    typedef index_type                                  argument_type;
    typedef index_type                                  result_type;
    typedef typename vector_type<index_type, 2u>::type  range_type;

    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uniform_seeds_functor(const uint32 _len, const uint32 _interval) : len(_len), interval(_interval) {}

    /// seeds of a string of `length` symbols: positions 0, interval, 2 interval, ... while the whole seed fits
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE index_type operator()(const index_type length) const
    { return length < len ? index_type(0) : index_type((length - len) / interval + 1u); }

    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE range_type seed(const uint32, const index_type i) const
    { return make_vector(index_type(i * interval), index_type(i * interval + len)); }

    const uint32 len;
    const uint32 interval;
};

namespace priv {
/// what an output coordinate type keeps of (a, b, c): seeds of one string pass (begin, end, 0), seeds of a set (string id, begin, end)
template <typename C, uint32 DIM = vector_traits<C>::DIM> struct seed_coord {};
template <typename C> struct seed_coord<C, 1u> { template <typename I> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE static C make(const I a, const I, const I)     { return C(a); } };
template <typename C> struct seed_coord<C, 2u> { template <typename I> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE static C make(const I a, const I b, const I)   { C r; r.x = a; r.y = b; return r; } };
template <typename C> struct seed_coord<C, 4u> { template <typename I> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE static C make(const I a, const I b, const I c) { C r; r.x = a; r.y = b; r.z = c; r.w = 0; return r; } };

template <typename seed_functor, typename coord_type, typename index_type>
struct write_string_seed
{
    write_string_seed(const index_type l, const seed_functor s, coord_type* o) : string_len(l), seeder(s), out(o) {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE void operator()(const index_type i) const
    {
        const typename seed_functor::range_type r = seeder.seed(string_len, i);
        out[i] = seed_coord<coord_type>::make(index_type(r.x), index_type(r.y), index_type(0));
    }
    const index_type string_len; const seed_functor seeder; coord_type* out;
};
template <typename string_set_type, typename seed_functor>
struct count_set_seeds
{
    typedef uint32 argument_type; typedef uint64 result_type;
    count_set_seeds(const string_set_type s, const seed_functor f) : string_set(s), seeder(f) {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint64 operator()(const uint32 id) const { return uint64(seeder(uint32(length(string_set[id])))); }
    const string_set_type string_set; const seed_functor seeder;
};
template <typename string_set_type, typename seed_functor, typename coord_type>
struct write_set_seeds
{
    write_set_seeds(const string_set_type s, const seed_functor f, const uint64* o, coord_type* c) : string_set(s), seeder(f), offsets(o), out(c) {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE void operator()(const uint32 id) const
    {
        const uint32 len = uint32(length(string_set[id]));
        const uint32 n   = uint32(seeder(len));
        coord_type*  dst = out + offsets[id];
        for (uint32 i = 0; i < n; ++i)
        {
            const typename seed_functor::range_type r = seeder.seed(len, i);
            dst[i] = seed_coord<coord_type>::make(id, uint32(r.x), uint32(r.y));
        }
    }
    const string_set_type string_set; const seed_functor seeder; const uint64* offsets; coord_type* out;
};
#if defined(__HIPCC__)
inline thrust::detail::device_t exec_policy(device_tag) { return thrust::device; }
inline thrust::detail::host_t   exec_policy(host_tag)   { return thrust::host; }
#endif
} // namespace priv

#if defined(__HIPCC__)
/// the seeds of one string of `string_len` symbols (seeds.h:70-74)
template <typename index_type, typename seed_functor, typename index_vector_type>
index_type enumerate_string_seeds(const index_type string_len, const seed_functor seeder, index_vector_type& indices)
{
    typedef typename index_vector_type::system_tag system_tag;
    typedef typename index_vector_type::value_type coord_type;
    const index_type n_seeds = seeder(string_len);
    indices.resize(n_seeds);
    thrust::for_each(priv::exec_policy(system_tag()), thrust::make_counting_iterator<index_type>(0), thrust::make_counting_iterator<index_type>(0) + n_seeds,
                     priv::write_string_seed<seed_functor, coord_type, index_type>(string_len, seeder, raw_pointer(indices)));
    return n_seeds;
}

/// the seeds of every string of a set, in string order (seeds.h:88-92); returns their number
template <typename string_set_type, typename seed_functor, typename index_vector_type>
uint64 enumerate_string_set_seeds(const string_set_type string_set, const seed_functor seeder, index_vector_type& indices)
{
    typedef typename index_vector_type::system_tag system_tag;
    typedef typename index_vector_type::value_type coord_type;
    const uint32 n_strings = string_set.size();
    if (n_strings == 0u) { indices.resize(0); return 0u; }
    nvbio::vector<system_tag, uint64> offsets(n_strings + 1u);
    thrust::transform(priv::exec_policy(system_tag()), thrust::make_counting_iterator<uint32>(0u), thrust::make_counting_iterator<uint32>(0u) + n_strings,
                      offsets.begin(), priv::count_set_seeds<string_set_type, seed_functor>(string_set, seeder));
    thrust::exclusive_scan(priv::exec_policy(system_tag()), offsets.begin(), offsets.begin() + n_strings + 1u, offsets.begin(), uint64(0));
    const uint64 n_seeds = offsets[n_strings];
    indices.resize(n_seeds);
    thrust::for_each(priv::exec_policy(system_tag()), thrust::make_counting_iterator<uint32>(0u), thrust::make_counting_iterator<uint32>(0u) + n_strings,
                     priv::write_set_seeds<string_set_type, seed_functor, coord_type>(string_set, seeder, raw_pointer(offsets), raw_pointer(indices)));
    return n_seeds;
}
#endif

} // namespace nvbio
