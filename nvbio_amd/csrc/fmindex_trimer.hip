// fmindex_trimer.hip -- builds the three-symbol rank arrays (fmindex_dimer.h, "three symbols per step") on the device from an
// index in the reference's layout.
//   rows    one lane per SA row i: c = BWT[i], j = LF(i), b = BWT[j], j' = LF(j), a = BWT[j']  (T[SA-1], T[SA-2], T[SA-3]); the rows
//           with SA < 3 hold no trimer.  Three waves cover two 96-row records; their ballots of the six code bits are combined
//           per trimer into the records' match masks and counts.
//   scan    64 exclusive prefix sums over the records, one per trimer (hipCUB), in place
//   finish  counters = C3[abc] + prefix,  C3[abc] = (first row whose suffix starts with "abc") - 1 by three single steps
// Buffer: 512-byte header {magic, length, primary, stride, ..., C3[64] at dword 64}, then pk[code][record] uint4.
#include "fmindex_device.h"
#include <hipcub/hipcub.hpp>

namespace nvb {

enum { TRIMER_MAGIC = 0x54724D33, TRIMER_HEADER_DWORDS = 128 };      // "TrM3"

// BWT symbol of SA row i (i != primary) and LF(i), from one record of the source layout
__device__ __forceinline__ uint32_t lf_step(const Fmi& f, const uint32_t i, uint32_t& sym)
{
    const uint32_t k = (i < f.primary) ? i : i - 1u;
    const Record r = load_record(f, k >> 6);
    const uint32_t word = comp(r.bwt, (k & 63u) >> 4);
    sym = (word >> (30u - ((k & 15u) << 1))) & 3u;
    return f.L2[sym] + comp(r.occ, sym) + block_count(r.bwt, (k & 63u) + 1u, sym);
}

__global__ void __launch_bounds__(192)
trimer_rows_kernel(const Fmi f, const uint32_t stride, uint4* __restrict__ pk, uint32_t* __restrict__ counts)
{
    __shared__ uint64_t planes[3][7];
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint64_t row = uint64_t(blockIdx.x) * 192u + threadIdx.x;
    uint32_t code = 0; bool valid = false;
    if (row <= uint64_t(f.length) && uint32_t(row) != f.primary)
    {
        uint32_t c, b, a;
        const uint32_t j = lf_step(f, uint32_t(row), c);
        if (j != f.primary)
        {
            const uint32_t j2 = lf_step(f, j, b);
            if (j2 != f.primary)
            {
                const uint32_t k = (j2 < f.primary) ? j2 : j2 - 1u;
                const uint32_t word = comp(f.rec[2ull * (k >> 6)], (k & 63u) >> 4);
                a = (word >> (30u - ((k & 15u) << 1))) & 3u;
                code = a * 16u + b * 4u + c; valid = true;
            }
        }
    }
    #pragma unroll
    for (int p = 0; p < 6; ++p) { const uint64_t bal = __builtin_amdgcn_ballot_w64(((code >> p) & 1u) != 0u); if (lane == 0u) planes[wv][p] = bal; }
    { const uint64_t bal = __builtin_amdgcn_ballot_w64(valid); if (lane == 0u) planes[wv][6] = bal; }
    __syncthreads();
    if (threadIdx.x < 128u)
    {
        const uint32_t t = threadIdx.x & 63u, rec = threadIdx.x >> 6;
        uint64_t m[3];
        #pragma unroll
        for (int w = 0; w < 3; ++w)
        {
            uint64_t v = planes[w][6];
            #pragma unroll
            for (int p = 0; p < 6; ++p) v &= ((t >> p) & 1u) ? planes[w][p] : ~planes[w][p];
            m[w] = v;
        }
        // record 0 = block rows 0..95, record 1 = block rows 96..191
        const uint64_t lo = rec == 0u ? m[0] : ((m[1] >> 32) | (m[2] << 32));
        const uint32_t hi = rec == 0u ? uint32_t(m[1]) : uint32_t(m[2] >> 32);
        const uint64_t r = 2ull * blockIdx.x + rec;
        if (r < stride)
        {
            pk[uint64_t(t) * stride + r] = make_uint4(0u, uint32_t(lo), uint32_t(lo >> 32), hi);
            counts[uint64_t(t) * stride + r] = __popcll(lo) + __popc(hi);
        }
    }
}

__global__ void trimer_header_kernel(const Fmi f, const uint32_t stride, uint32_t* __restrict__ out)
{
    const uint32_t t = threadIdx.x;      // 0..63: code a*16 + b*4 + c
    const uint32_t a = t >> 4, b = (t >> 2) & 3u, c = t & 3u;
    // the range of "abc" by three single steps from the whole index: only its first row matters
    const uint32_t x1 = f.L2[c] + 1u;
    const uint32_t x2 = f.L2[b] + fm_rank(f, x1 - 1u, b) + 1u;
    const uint32_t x3 = f.L2[a] + fm_rank(f, x2 - 1u, a) + 1u;
    out[64u + t] = x3 - 1u;
    if (t == 0u) { out[0] = TRIMER_MAGIC; out[1] = f.length; out[2] = f.primary; out[3] = stride; out[4] = f.L2[1] ^ ((f.L2[2] << 11) | (f.L2[2] >> 21)); }
}

__global__ void __launch_bounds__(256)
trimer_counters_kernel(const uint32_t stride, const uint32_t* __restrict__ excl, const uint32_t* __restrict__ header, uint4* __restrict__ pk)
{
    const uint64_t id = uint64_t(blockIdx.x) * 256u + threadIdx.x;
    if (id >= 64ull * stride) return;
    const uint32_t t = uint32_t(id / stride);
    reinterpret_cast<uint32_t*>(pk + id)[0] = header[64u + t] + excl[id];
}

static inline uint64_t align256t(uint64_t x) { return (x + 255ull) & ~255ull; }
static inline uint32_t trimer_stride(uint32_t length) { return uint32_t((uint64_t(length) + 1u) / 96u) + 1u; }

} // namespace nvb

using namespace nvb;

NVB_API uint64_t nvbio_hip_fm_trimer_index_bytes(uint32_t length)
{
    return 4ull * TRIMER_HEADER_DWORDS + 64ull * 16ull * trimer_stride(length);
}

NVB_API uint64_t nvbio_hip_fm_build_trimer_index_temp_bytes(uint32_t length)
{
    const uint32_t R = trimer_stride(length);
    size_t scan = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, scan, (uint32_t*)nullptr, (uint32_t*)nullptr, int(R));
    return align256t(64ull * R * 4ull) + align256t(scan) + 256u;
}

NVB_API int nvbio_hip_fm_build_trimer_index(const nvbio_hip_fmindex* fmi, uint32_t* out_trimer, void* temp, uint64_t temp_bytes, void* stream)
{
    if (!fmi || !fmi->bwt_occ || !out_trimer || !temp) return hipErrorInvalidValue;
    if (fmi->length == 0 || fmi->length >= 0xFFFFFF00u) return hipErrorInvalidValue;
    if (temp_bytes < nvbio_hip_fm_build_trimer_index_temp_bytes(fmi->length)) return hipErrorInvalidValue;
    if ((reinterpret_cast<uintptr_t>(out_trimer) & 127u) != 0) return hipErrorInvalidValue;
    Fmi f = make_fmi(fmi);
    f.ktab = nullptr; f.ktab_k = 0; f.dm.base = nullptr; f.tm.pk = nullptr;
    const uint32_t R = trimer_stride(fmi->length);
    uint32_t* counts = reinterpret_cast<uint32_t*>(temp);
    void*  scan_tmp = reinterpret_cast<uint8_t*>(temp) + align256t(64ull * R * 4ull);
    size_t scan_bytes = size_t(temp_bytes - align256t(64ull * R * 4ull));
    uint4* pk = reinterpret_cast<uint4*>(out_trimer + TRIMER_HEADER_DWORDS);
    hipStream_t s = to_stream(stream);
    g_last_kernel = "trimer_rows_kernel";
    if (hipError_t e = hipMemsetAsync(out_trimer, 0, 4u * TRIMER_HEADER_DWORDS, s)) return e;
    hipLaunchKernelGGL(trimer_rows_kernel, dim3((R + 1u) / 2u), dim3(192), 0, s, f, R, pk, counts);
    if (hipError_t e = hipGetLastError()) return e;
    for (uint32_t t = 0; t < 64u; ++t)
        if (hipError_t e = hipcub::DeviceScan::ExclusiveSum(scan_tmp, scan_bytes, counts + uint64_t(t) * R, counts + uint64_t(t) * R, int(R), s)) return e;
    hipLaunchKernelGGL(trimer_header_kernel, dim3(1), dim3(64), 0, s, f, R, out_trimer);
    hipLaunchKernelGGL(trimer_counters_kernel, dim3(uint32_t((64ull * R + 255u) / 256u)), dim3(256), 0, s, R, counts, out_trimer, pk);
    return hipGetLastError();
}

// Attach with validation, like nvbio_hip_fm_attach_dimer_index: the header written by trimer_header_kernel must describe THIS index
// (magic, length, primary, the stride the kernels derive from the length, the L2 check word) and the two-symbol index must be attached
// already (leftover symbols and empty-range replays use its steps).  trimer == NULL detaches.
NVB_API int nvbio_hip_fm_attach_trimer_index(nvbio_hip_fmindex* fmi, const uint32_t* trimer, void* stream)
{
    using namespace nvb;
    if (!fmi) return hipErrorInvalidValue;
    if (!trimer) { fmi->trimer = nullptr; return hipSuccess; }
    if (!fmi->dimer) return hipErrorInvalidValue;
    uint32_t h[5];
    if (hipError_t e = hipMemcpyAsync(h, trimer, sizeof(h), hipMemcpyDeviceToHost, to_stream(stream))) return e;
    if (hipError_t e = hipStreamSynchronize(to_stream(stream))) return e;
    if (h[0] != uint32_t(TRIMER_MAGIC) || h[1] != fmi->length || h[2] != fmi->primary || h[3] != trimer_stride(fmi->length)) return hipErrorInvalidValue;
    if (h[4] != (fmi->L2[1] ^ ((fmi->L2[2] << 11) | (fmi->L2[2] >> 21)))) return hipErrorInvalidValue;
    fmi->trimer = trimer;
    return hipSuccess;
}
