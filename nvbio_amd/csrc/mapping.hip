// mapping.hip -- nvBowtie exact seed mapping on gfx950: reads -> per-read sets of SeedHits.
//
// Computes what the reference's
//   map_queues_kernel<EXACT_MAPPING>   nvBowtie/bowtie2/cuda/mapping_inl.h:511-592
//   seed_mapper<EXACT_MAPPING>::enact  mapping_inl.h:229-312   (USE_REVERSE_INDEX 0, defs.h:62)
//   match_range                        mapping_inl.h:83-97
// push into each read's hit deque, as SeedHit words (seed_hit.h:54-223): for every seed window of
// the read (pos += seed_freq(read_len), starting at retry * seed_freq/(max_reseed+1)) without an N,
// the forward scan of the stored seed on the forward index (flags STANDARD/FORWARD,
// pos_in_read = read_end - pos - seed_len) and the reverse scan of its complement (flags
// COMPLEMENT/FORWARD, pos_in_read = pos - read_begin); non-empty SA ranges only, stored exclusive.
//
// Difference in mechanism, not in the result: the reference keeps a 512-entry deque in per-thread local
// memory and copies it to an arena slot taken with an atomic bump, so the slot order is run-dependent.
// Here the deque (hit_deque.h: the reference's interval heap, exchange for exchange) is built in place in
// the fixed slot read_id * hits_stride -- no local-memory copy, no atomics -- and a read's hits come out
// in exactly the array order the reference's store_deque writes (mapping_inl.h:101-115), which is what the
// selection stage's top() and randomized sampling act on.  When a read produces more than max_hits hits,
// the deque's bottom (a hit of largest range size) is dropped per extra hit (:268-270).
#include "fmindex_device.h"
#include "hit_deque.h"

namespace nvb {

struct MapParams {
    uint32_t seed_len, min_read_len, max_hits, max_reseed, retry, rep_seeds, fw, rc;
};

// match_range over the transformed seed: scan symbol t = comp(seed[reverse ? len-1-t : t])
__device__ __forceinline__ uint2 match_range_x(const Fmi& f, const Stream& s, uint64_t begin, uint32_t len, bool reverse, bool complement, bool& has_n)
{
    uint32_t x = 0, y = f.length;
    uint32_t t_start = 0;
    bool pairs = f.dm.base != nullptr;
    bool triples = pairs && f.tm.pk != nullptr;
    const uint32_t k = f.ktab_k;
    if (k != 0u && len >= k)
    {
        // the first k scan symbols through the k-mer table (entry = match of that k-mer; code has the
        // symbol consumed LAST at bits [0,2) -- see nvbio_hip_fm_build_ktab)
        uint64_t grp = (s.bits == 2) ? expand_2to4(fetch16_2bit(s, reverse ? begin + len - k : begin))
                                     : fetch16_4bit(s, reverse ? begin + len - k : begin);
        grp &= (k == 16u) ? ~0ull : ((1ull << (4u * k)) - 1ull);
        if ((grp & 0xCCCCCCCCCCCCCCCCull) != 0ull) { has_n = true; return make_uint2(1u, 0u); }
        const uint32_t packed = nibbles_to_2bit(grp);                         // group symbol j at bits 2j
        const uint32_t mask = (k == 16u) ? 0xFFFFFFFFu : ((1u << (2u * k)) - 1u);
        // forward scan: scan symbol i = group symbol i, consumed first -> highest field: reverse the fields.
        // reverse scan: scan symbol i = group symbol k-1-i -> already in table order; complement = ~.
        uint32_t code = reverse ? packed : (rev2(packed) >> (32u - 2u * k));
        if (complement) code = ~code & mask;
        const uint2 r = f.ktab[code];
        if (r.x > r.y) return r;
        x = r.x; y = r.y; t_start = k;
    }
    for (uint32_t t0 = t_start; t0 < len && x <= y; t0 += 16u)
    {
        // the 16-symbol group holding scan symbols [t0, t0+16)
        const uint32_t cnt = (len - t0) < 16u ? (len - t0) : 16u;
        const uint64_t g0  = reverse ? begin + len - t0 - cnt : begin + t0;
        const uint64_t grp = (s.bits == 2) ? expand_2to4(fetch16_2bit(s, g0)) : fetch16_4bit(s, g0);
        uint32_t u = 0;
        while (u < cnt && x <= y)
        {
            uint32_t c = uint32_t(grp >> (4u * (reverse ? cnt - 1u - u : u))) & 15u;
            if (c > 3u) { has_n = true; return make_uint2(1u, 0u); }
            if (complement) c = 3u - c;
            if (pairs && u + 1u < cnt)
            {
                // two (three) scan symbols per line fetch on the two- (three-) symbol index (see fm_match_from)
                uint32_t b = uint32_t(grp >> (4u * (reverse ? cnt - 2u - u : u + 1u))) & 15u;
                if (b <= 3u)
                {
                    if (complement) b = 3u - b;
                    if (triples && u + 2u < cnt)
                    {
                        uint32_t a = uint32_t(grp >> (4u * (reverse ? cnt - 3u - u : u + 2u))) & 15u;
                        if (a <= 3u)
                        {
                            if (complement) a = 3u - a;
                            const uint2 r = tm_step3(f.tm, x, y, a, b, c);
                            if (r.x <= r.y) { x = r.x; y = r.y; u += 3u; continue; }
                            triples = false;
                        }
                    }
                    const uint2 r = dm_step2(f.dm, x, y, b, c);
                    if (r.x <= r.y) { x = r.x; y = r.y; u += 2u; continue; }
                    pairs = false; triples = false;
                }
            }
            const uint2 r = fm_step(f, x, y, c);
            x = r.x; y = r.y; ++u;
        }
    }
    return make_uint2(x, y);
}

__device__ __forceinline__ uint2 seed_hit_pack(uint32_t begin, uint32_t delta, uint32_t pos, uint32_t rc)
{
    return make_uint2(begin, (delta & 0xFFFFFu) | ((pos & 0x3FFu) << 20) | ((rc & 1u) << 30));    // indexdir = FORWARD = 0
}

__device__ __forceinline__ void
map_exact_one(const uint32_t id, const Fmi& f, const StringSet& reads, const uint32_t* __restrict__ in_queue,
              const MapParams& p, const uint32_t* __restrict__ seed_freq_by_len,
              uint2* __restrict__ out_hits, uint32_t hits_stride, uint32_t* __restrict__ out_counts, uint8_t* __restrict__ out_reseed)
{
    const uint32_t read_id = in_queue ? in_queue[id] : id;
    const uint64_t rb   = reads.begin[read_id];
    const uint32_t rlen = reads.length ? reads.length[read_id] : reads.fixed_length;
    uint2* hits = out_hits + uint64_t(read_id) * hits_stride;
    if (rlen < p.min_read_len) { out_counts[read_id] = 0; return; }

    const uint32_t seed_len     = p.seed_len < rlen ? p.seed_len : rlen;
    const uint32_t seed_freq    = seed_freq_by_len[rlen];
    const uint32_t retry_stride = seed_freq / (p.max_reseed + 1u);
    const uint32_t cap = p.max_hits < hits_stride ? p.max_hits : hits_stride;
    uint32_t nh = 0, range_sum = 0, range_count = 0;

    for (uint64_t pos = rb + uint64_t(p.retry) * retry_stride; pos + seed_len <= rb + rlen; pos += seed_freq)
    {
        // a seed holding an N (or any symbol > 3) yields nothing on either strand (mapping_inl.h:258, :90)
        bool has_n = false;
        #pragma unroll 1
        for (uint32_t strand = 0; strand < 2u && !has_n; ++strand)
        {
            if (strand == 0 ? !p.fw : !p.rc)
            {
                if (strand == 0) {      // the N test precedes the strand switches: do it without the search
                    for (uint32_t t0 = 0; t0 < seed_len; t0 += 16u) {
                        const uint32_t cnt = (seed_len - t0) < 16u ? (seed_len - t0) : 16u;
                        uint64_t grp = (reads.s.bits == 2) ? 0ull : fetch16_4bit(reads.s, pos + t0);
                        grp &= cnt == 16u ? ~0ull : ((1ull << (4u * cnt)) - 1ull);
                        has_n |= (grp & 0xCCCCCCCCCCCCCCCCull) != 0ull;
                    }
                }
                continue;
            }
            const uint2 r = match_range_x(f, reads.s, pos, seed_len, strand != 0, strand != 0, has_n);
            if (has_n) break;
            if (r.x > r.y) continue;
            const uint32_t pir = strand == 0 ? uint32_t(rb + rlen - pos - seed_len) : uint32_t(pos - rb);
            const HitDeque deque = { hits };
            if (nh == cap) { deque.pop_bottom(int(nh)); --nh; }                  // :268-270
            hits[nh++] = seed_hit_pack(r.x, r.y + 1u - r.x, pir, strand);
            deque.push(int(nh));
            range_sum += r.y - r.x + 1u; range_count++;
        }
    }
    out_counts[read_id] = nh;
    if (out_reseed) out_reseed[id] = (range_count == 0u || range_sum >= p.rep_seeds * range_count) ? 1 : 0;
}
// grid-stride (a launch may hold fewer blocks than reads / 256), so that this
// fabric-bound kernel leaves wave slots and registers of every CU to a VALU-bound kernel of another stream
__global__ void __launch_bounds__(256)
map_exact_kernel(const Fmi f, const StringSet reads, const uint32_t* __restrict__ in_queue, uint32_t n,
                 const MapParams p, const uint32_t* __restrict__ seed_freq_by_len,
                 uint2* __restrict__ out_hits, uint32_t hits_stride, uint32_t* __restrict__ out_counts, uint8_t* __restrict__ out_reseed)
{
    for (uint64_t id = uint64_t(blockIdx.x) * 256u + threadIdx.x; id < n; id += uint64_t(gridDim.x) * 256u)
        map_exact_one(uint32_t(id), f, reads, in_queue, p, seed_freq_by_len, out_hits, hits_stride, out_counts, out_reseed);
}

// ------------------------------------------------------------------ one-mismatch mappers
// map<find_exact>                    mapping_inl.h:124-223
// seed_mapper<APPROX_MAPPING>        mapping_inl.h:318-365
// seed_mapper<CASE_PRUNING_MAPPING>  mapping_inl.h:372-428
// A seed (<= 32 symbols) is held in scan order as 32 nibbles in two registers, so the four readers of
// the reference (forward / reverse x plain / complement) are one fetch plus a nibble permutation, and
// the mismatch enumeration never goes back to memory for read symbols.
struct SeedVec { uint64_t lo, hi; };

__device__ __forceinline__ uint32_t sv_sym(const SeedVec& q, const uint32_t t)
{
    return uint32_t(t < 16u ? q.lo >> (4u * t) : q.hi >> (4u * (t - 16u))) & 15u;
}
// stored symbols [pos, pos+len) as nibbles, symbol r at nibble r; nibbles >= len are zero
__device__ __forceinline__ SeedVec sv_load(const Stream& s, const uint64_t pos, const uint32_t len)
{
    SeedVec q;
    q.lo = (s.bits == 2) ? expand_2to4(fetch16_2bit(s, pos)) : fetch16_4bit(s, pos);
    q.hi = 0;
    if (len > 16u) q.hi = (s.bits == 2) ? expand_2to4(fetch16_2bit(s, pos + 16u)) : fetch16_4bit(s, pos + 16u);
    if (len < 16u) q.lo &= (1ull << (4u * len)) - 1ull;
    else if (len < 32u) q.hi &= (1ull << (4u * (len - 16u))) - 1ull;
    return q;
}
__device__ __forceinline__ uint64_t rev_nibbles(uint64_t x)
{
    x = (uint64_t(__builtin_bswap32(uint32_t(x))) << 32) | __builtin_bswap32(uint32_t(x >> 32));      // byte reverse
    return ((x & 0xF0F0F0F0F0F0F0F0ull) >> 4) | ((x & 0x0F0F0F0F0F0F0F0Full) << 4);
}
// ReverseXform: scan symbol t = stored symbol len-1-t
__device__ __forceinline__ SeedVec sv_reverse(const SeedVec& q, const uint32_t len)
{
    // reverse all 32 nibbles, then shift the len valid ones down
    const uint64_t rl = rev_nibbles(q.hi), rh = rev_nibbles(q.lo);       // 128-bit value {rh:rl}
    const uint32_t sh = 4u * (32u - len);                                // 0..124
    SeedVec o;
    if (sh == 0u)       { o.lo = rl; o.hi = rh; }
    else if (sh < 64u)  { o.lo = (rl >> sh) | (rh << (64u - sh)); o.hi = rh >> sh; }
    else if (sh == 64u) { o.lo = rh; o.hi = 0; }
    else                { o.lo = rh >> (sh - 64u); o.hi = 0; }
    return o;
}
// complement_functor<4>: c < 4 -> 3 - c, N unchanged
__device__ __forceinline__ SeedVec sv_complement(const SeedVec& q, const uint32_t len)
{
    auto comp64 = [](const uint64_t x, const uint32_t n) {
        const uint64_t valid = n >= 16u ? ~0ull : ((1ull << (4u * n)) - 1ull);
        const uint64_t isn   = ((x >> 2) | (x >> 3)) & 0x1111111111111111ull;        // nibble > 3
        return x ^ (0x3333333333333333ull & valid & ~(isn * 3ull));
    };
    SeedVec o;
    o.lo = comp64(q.lo, len < 16u ? len : 16u);
    o.hi = comp64(q.hi, len > 16u ? len - 16u : 0u);
    return o;
}

// match_range (mapping_inl.h:80-96) over scan symbols [a,b) of q
__device__ __forceinline__ uint2 match_span(const Fmi& f, const SeedVec& q, const uint32_t a, const uint32_t b, uint2 r)
{
    bool pairs = f.dm.base != nullptr;
    bool triples = pairs && f.tm.pk != nullptr;
    uint32_t i = a;
    while (i < b && r.x <= r.y)
    {
        const uint32_t c = sv_sym(q, i);
        if (c > 3u) return make_uint2(1u, 0u);
        if (pairs && i + 1u < b)
        {
            const uint32_t c2 = sv_sym(q, i + 1u);
            if (c2 <= 3u)
            {
                if (triples && i + 2u < b)
                {
                    const uint32_t c3 = sv_sym(q, i + 2u);
                    if (c3 <= 3u)
                    {
                        const uint2 k = tm_step3(f.tm, r.x, r.y, c3, c2, c);
                        if (k.x <= k.y) { r = k; i += 3u; continue; }
                        triples = false;
                    }
                }
                const uint2 k = dm_step2(f.dm, r.x, r.y, c2, c);
                if (k.x <= k.y) { r = k; i += 2u; continue; }
                pairs = false; triples = false;
            }
        }
        r = fm_step(f, r.x, r.y, c);
        ++i;
    }
    return r;
}

struct HitHeap {
    uint2*   hits;
    uint32_t nh, cap, range_sum, range_count;
    __device__ __forceinline__ void push(const uint2 r /* inclusive */, const uint32_t flags)
    {
        const HitDeque deque = { hits };
        if (nh == cap) { deque.pop_bottom(int(nh)); --nh; }
        hits[nh++] = make_uint2(r.x, ((r.y + 1u - r.x) & 0xFFFFFu) | flags);
        deque.push(int(nh));
        range_sum += r.y - r.x + 1u; range_count++;
    }
};
__device__ __forceinline__ uint32_t hit_flags(const uint32_t pos, const uint32_t rc, const uint32_t indexdir)
{
    return ((pos & 0x3FFu) << 20) | ((rc & 1u) << 30) | ((indexdir & 1u) << 31);
}

template <bool FIND_EXACT>
__device__ __forceinline__ void map_one_mismatch(const SeedVec& q, uint32_t len1, const uint32_t len2, const Fmi& f, const uint32_t flags, HitHeap& heap)
{
    // (a read shorter than subseed_len: the reference indexes outside its seed; here the whole seed is exact)
    len1 = len1 < len2 ? len1 : len2;
    // Ns: none in the exact part, at most one in the rest; exact matching runs up to it (:141-155)
    const uint64_t nlo = ((q.lo >> 2) | (q.lo >> 3)) & 0x1111111111111111ull, nhi = ((q.hi >> 2) | (q.hi >> 3)) & 0x1111111111111111ull;
    const uint32_t n_cnt = __popcll(nlo) + __popcll(nhi);
    if (n_cnt > 1u) return;
    if (n_cnt == 1u) {
        const uint32_t n_pos = nlo ? uint32_t(__ffsll((long long)nlo) - 1) >> 2 : 16u + (uint32_t(__ffsll((long long)nhi) - 1) >> 2);
        if (n_pos < len1) return;
        len1 = n_pos;
    }
    uint2 base = make_uint2(0u, f.length);
    uint32_t t0 = 0;
    const uint32_t k = f.ktab_k;
    if (k != 0u && k <= 16u && len1 >= k)
    {
        const uint32_t packed = nibbles_to_2bit(q.lo & (k == 16u ? ~0ull : ((1ull << (4u * k)) - 1ull)));   // scan symbol i at bits 2i
        base = f.ktab[rev2(packed) >> (32u - 2u * k)];                                                     // first consumed -> highest field
        t0 = k;
    }
    base = match_span(f, q, t0, len1, base);
    for (uint32_t i = len1; i < len2 && base.x <= base.y; ++i)
    {
        const uint32_t c = sv_sym(q, i);
        uint4 lo, hi;
        fm_step4(f, base.x, base.y, lo, hi);
        #pragma unroll 1
        for (uint32_t sub = 0; sub < 4u; ++sub)
        {
            const uint32_t l = comp(lo, sub), h = comp(hi, sub);
            if (sub != c && h > l)
            {
                const uint2 r = match_span(f, q, i + 1u, len2, make_uint2(f.L2[sub] + l + 1u, f.L2[sub] + h));
                if (r.x <= r.y) heap.push(r, flags);
            }
        }
        if (c < 4u) { base.x = f.L2[c] + comp(lo, c) + 1u; base.y = f.L2[c] + comp(hi, c); }
        else { base = make_uint2(1u, 0u); break; }
    }
    if (FIND_EXACT && base.x <= base.y) heap.push(base, flags);
}

template <int ALGO>     // 1 = APPROX_MAPPING, 2 = CASE_PRUNING_MAPPING
__device__ __forceinline__ void
map_mismatch_one(const uint32_t id, const Fmi& f, const Fmi& rf, const StringSet& reads, const uint32_t* __restrict__ in_queue,
                 const MapParams& p, const uint32_t subseed_len, const uint32_t* __restrict__ seed_freq_by_len,
                 uint2* __restrict__ out_hits, uint32_t hits_stride, uint32_t* __restrict__ out_counts, uint8_t* __restrict__ out_reseed)
{
    const uint32_t read_id = in_queue ? in_queue[id] : id;
    const uint64_t rb   = reads.begin[read_id];
    const uint32_t rlen = reads.length ? reads.length[read_id] : reads.fixed_length;
    if (rlen < p.min_read_len) { out_counts[read_id] = 0; return; }

    const uint32_t seed_len     = p.seed_len < rlen ? p.seed_len : rlen;
    const uint32_t seed_freq    = seed_freq_by_len[rlen];
    const uint32_t retry_stride = seed_freq / (p.max_reseed + 1u);
    HitHeap heap = { out_hits + uint64_t(read_id) * hits_stride, 0u, p.max_hits < hits_stride ? p.max_hits : hits_stride, 0u, 0u };

    for (uint64_t pos = rb + uint64_t(p.retry) * retry_stride; pos + seed_len <= rb + rlen; pos += seed_freq)
    {
        const SeedVec fr = sv_load(reads.s, pos, seed_len);          // f_reader
        const uint32_t rel = uint32_t(pos - rb);
        if (ALGO == 1)
        {
            // two symbols equal to 4 -> nothing (:346); other symbols > 3 do not occur in reads
            const uint64_t e4lo = (fr.lo >> 2) & ~(fr.lo >> 3) & ~(fr.lo >> 1) & ~fr.lo & 0x1111111111111111ull;
            const uint64_t e4hi = (fr.hi >> 2) & ~(fr.hi >> 3) & ~(fr.hi >> 1) & ~fr.hi & 0x1111111111111111ull;
            if (__popcll(e4lo) + __popcll(e4hi) >= 2u) continue;
            if (p.fw) map_one_mismatch<true >(fr, subseed_len, seed_len, f, hit_flags(rlen - rel - seed_len, 0u, 0u), heap);
            if (p.rc) map_one_mismatch<false>(sv_complement(sv_reverse(fr, seed_len), seed_len), subseed_len, seed_len, f, hit_flags(rel, 1u, 0u), heap);
        }
        else
        {
            const SeedVec rr = sv_reverse(fr, seed_len);
            if (p.fw) map_one_mismatch<true >(fr, seed_len / 2u,        seed_len, f,  hit_flags(rlen - rel - seed_len, 0u, 0u), heap);
            if (p.fw) map_one_mismatch<false>(rr, (seed_len + 1u) / 2u, seed_len, rf, hit_flags(rlen - rel - 1u,       0u, 1u), heap);
            if (p.rc) map_one_mismatch<true >(sv_complement(fr, seed_len), seed_len / 2u,        seed_len, rf, hit_flags(rel + seed_len - 1u, 1u, 1u), heap);
            if (p.rc) map_one_mismatch<false>(sv_complement(rr, seed_len), (seed_len + 1u) / 2u, seed_len, f,  hit_flags(rel,                 1u, 0u), heap);
        }
    }
    out_counts[read_id] = heap.nh;
    if (out_reseed) out_reseed[id] = (heap.range_count == 0u || heap.range_sum >= p.rep_seeds * heap.range_count) ? 1 : 0;
}
template <int ALGO>
__global__ void __launch_bounds__(256)
map_mismatch_kernel(const Fmi f, const Fmi rf, const StringSet reads, const uint32_t* __restrict__ in_queue, uint32_t n,
                    const MapParams p, const uint32_t subseed_len, const uint32_t* __restrict__ seed_freq_by_len,
                    uint2* __restrict__ out_hits, uint32_t hits_stride, uint32_t* __restrict__ out_counts, uint8_t* __restrict__ out_reseed)
{
    for (uint64_t id = uint64_t(blockIdx.x) * 256u + threadIdx.x; id < n; id += uint64_t(gridDim.x) * 256u)
        map_mismatch_one<ALGO>(uint32_t(id), f, rf, reads, in_queue, p, subseed_len, seed_freq_by_len, out_hits, hits_stride, out_counts, out_reseed);
}

} // namespace nvb

using namespace nvb;

NVB_API int nvbio_hip_map(int32_t algorithm, uint32_t subseed_len, const nvbio_hip_fmindex* fmi, const nvbio_hip_fmindex* rfmi,
                          const nvbio_hip_string_set* reads, const uint32_t* in_queue, uint32_t n, const nvbio_hip_map_params* params,
                          const uint32_t* seed_freq_by_len,
                          uint64_t* out_hits, uint32_t hits_stride, uint32_t* out_counts, uint8_t* out_reseed, void* stream)
{
    if (algorithm == NVBIO_HIP_EXACT_MAPPING)
        return nvbio_hip_map_exact(fmi, reads, in_queue, n, params, seed_freq_by_len, out_hits, hits_stride, out_counts, out_reseed, stream);
    if (algorithm != NVBIO_HIP_APPROX_MAPPING && algorithm != NVBIO_HIP_CASE_PRUNING_MAPPING) return hipErrorInvalidValue;
    if (!fmi || !fmi->bwt_occ || !reads || !params) return hipErrorInvalidValue;
    if (algorithm == NVBIO_HIP_CASE_PRUNING_MAPPING && (!rfmi || !rfmi->bwt_occ)) return hipErrorInvalidValue;
    if (!(reads->bits == 2 || reads->bits == 4)) return hipErrorNotSupported;
    if (params->seed_len > 32u) return hipErrorNotSupported;            // the mismatch mappers keep a seed in two registers
    if (algorithm == NVBIO_HIP_APPROX_MAPPING && subseed_len > params->seed_len) return hipErrorInvalidValue;
    if (n == 0) return hipSuccess;
    if (!reads->words || !reads->begin || reads->n_words == 0 || !seed_freq_by_len || !out_hits || !out_counts || hits_stride == 0)
        return hipErrorInvalidValue;
    MapParams p;
    p.seed_len = params->seed_len; p.min_read_len = params->min_read_len; p.max_hits = params->max_hits;
    p.max_reseed = params->max_reseed; p.retry = params->retry; p.rep_seeds = params->rep_seeds;
    p.fw = params->fw; p.rc = params->rc;
    const Fmi f = make_fmi(fmi), rf = make_fmi(algorithm == NVBIO_HIP_CASE_PRUNING_MAPPING ? rfmi : fmi);
    const dim3 grid(seeding_grid(n)), block(256);
    uint2* hits = reinterpret_cast<uint2*>(out_hits);
    if (algorithm == NVBIO_HIP_APPROX_MAPPING) {
        g_last_kernel = "map_mismatch_kernel<APPROX>";
        hipLaunchKernelGGL(map_mismatch_kernel<1>, grid, block, 0, to_stream(stream), f, rf, make_string_set(reads), in_queue, n, p, subseed_len,
                           seed_freq_by_len, hits, hits_stride, out_counts, out_reseed);
    } else {
        g_last_kernel = "map_mismatch_kernel<CASE_PRUNING>";
        hipLaunchKernelGGL(map_mismatch_kernel<2>, grid, block, 0, to_stream(stream), f, rf, make_string_set(reads), in_queue, n, p, subseed_len,
                           seed_freq_by_len, hits, hits_stride, out_counts, out_reseed);
    }
    return hipGetLastError();
}


NVB_API int nvbio_hip_map_exact(const nvbio_hip_fmindex* fmi, const nvbio_hip_string_set* reads,
                                const uint32_t* in_queue, uint32_t n, const nvbio_hip_map_params* params,
                                const uint32_t* seed_freq_by_len,
                                uint64_t* out_hits, uint32_t hits_stride, uint32_t* out_counts, uint8_t* out_reseed, void* stream)
{
    if (!fmi || !fmi->bwt_occ || !reads || !params) return hipErrorInvalidValue;
    if (!(reads->bits == 2 || reads->bits == 4)) return hipErrorNotSupported;
    if (n == 0) return hipSuccess;
    if (!reads->words || !reads->begin || reads->n_words == 0 || !seed_freq_by_len || !out_hits || !out_counts || hits_stride == 0)
        return hipErrorInvalidValue;
    MapParams p;
    p.seed_len = params->seed_len; p.min_read_len = params->min_read_len; p.max_hits = params->max_hits;
    p.max_reseed = params->max_reseed; p.retry = params->retry; p.rep_seeds = params->rep_seeds;
    p.fw = params->fw; p.rc = params->rc;
    Fmi f = make_fmi(fmi);
    g_last_kernel = "map_exact_kernel";
    hipLaunchKernelGGL(map_exact_kernel, dim3(seeding_grid(n)), dim3(256), 0, to_stream(stream), f, make_string_set(reads), in_queue, n, p,
                       seed_freq_by_len, reinterpret_cast<uint2*>(out_hits), hits_stride, out_counts, out_reseed);
    return hipGetLastError();
}
