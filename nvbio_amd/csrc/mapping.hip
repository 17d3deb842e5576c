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
// Differences in mechanism, not in the hit sets: the reference keeps a 512-entry interval heap in
// per-thread local memory and copies it to an arena slot taken with an atomic bump, so the order of
// a read's hits (heap layout) and the slot order (atomics) are run-dependent, and its own checksum
// compares the sorted sets (checksums.h).  Here a read's hits are written in generation order to a
// fixed slot read_id * hits_stride -- no local-memory heap, no atomics, deterministic.  When a read
// produces more than max_hits hits, a hit of largest range size is dropped per extra hit exactly as
// priority_deque::pop_bottom does (which of several equal-sized largest hits goes is unspecified in
// both).
#include "fmindex_device.h"

namespace nvb {

struct MapParams {
    uint32_t seed_len, min_read_len, max_hits, max_reseed, retry, rep_seeds, fw, rc;
};

// match_range over the transformed seed: scan symbol t = comp(seed[reverse ? len-1-t : t])
__device__ __forceinline__ uint2 match_range_x(const Fmi& f, const Stream& s, uint64_t begin, uint32_t len, bool reverse, bool complement, bool& has_n)
{
    uint32_t x = 0, y = f.length;
    uint32_t t_start = 0;
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
        for (uint32_t u = 0; u < cnt && x <= y; ++u)
        {
            uint32_t c = uint32_t(grp >> (4u * (reverse ? cnt - 1u - u : u))) & 15u;
            if (c > 3u) { has_n = true; return make_uint2(1u, 0u); }
            if (complement) c = 3u - c;
            const uint2 r = fm_rank2(f, x - 1u, y, c);
            x = f.L2[c] + r.x + 1u;
            y = f.L2[c] + r.y;
        }
    }
    return make_uint2(x, y);
}

__device__ __forceinline__ uint2 seed_hit_pack(uint32_t begin, uint32_t delta, uint32_t pos, uint32_t rc)
{
    return make_uint2(begin, (delta & 0xFFFFFu) | ((pos & 0x3FFu) << 20) | ((rc & 1u) << 30));    // indexdir = FORWARD = 0
}

__global__ void __launch_bounds__(256)
map_exact_kernel(const Fmi f, const StringSet reads, const uint32_t* __restrict__ in_queue, uint32_t n,
                 const MapParams p, const uint32_t* __restrict__ seed_freq_by_len,
                 uint2* __restrict__ out_hits, uint32_t hits_stride, uint32_t* __restrict__ out_counts, uint8_t* __restrict__ out_reseed)
{
    const uint32_t id = blockIdx.x * 256u + threadIdx.x;
    if (id >= n) return;
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
            if (nh == cap)
            {
                uint32_t worst = 0, wsize = hits[0].y & 0xFFFFFu;
                for (uint32_t h = 1; h < nh; ++h) { const uint32_t sz = hits[h].y & 0xFFFFFu; if (sz > wsize) { wsize = sz; worst = h; } }
                hits[worst] = hits[--nh];
            }
            hits[nh++] = seed_hit_pack(r.x, r.y + 1u - r.x, pir, strand);
            range_sum += r.y - r.x + 1u; range_count++;
        }
    }
    out_counts[read_id] = nh;
    if (out_reseed) out_reseed[id] = (range_count == 0u || range_sum >= p.rep_seeds * range_count) ? 1 : 0;
}

} // namespace nvb

using namespace nvb;

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
    hipLaunchKernelGGL(map_exact_kernel, dim3((n + 255u) / 256u), dim3(256), 0, to_stream(stream), f, make_string_set(reads), in_queue, n, p,
                       seed_freq_by_len, reinterpret_cast<uint2*>(out_hits), hits_stride, out_counts, out_reseed);
    return hipGetLastError();
}
