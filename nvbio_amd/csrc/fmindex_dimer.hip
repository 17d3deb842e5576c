// fmindex_dimer.hip -- builds the line-native two-symbol index (layout and queries: fmindex_dimer.h) on
// the device from an index in the reference's layout (bwt|occ records, nvbio/io/fmindex/fmindex_impl.cu:305-327).
//
//   pass A  one lane per SA row i: b = BWT[i], j = LF(i) on the source index, a = BWT[j]  (T[SA-1], T[SA-2]);
//           wave ballots of the four nibble bits ARE the bit-planes; 16 lanes popcount the nibble matches
//           -> per-block counts.  The lane whose j is `primary` is row p1 (SA = 1).
//   scan    exclusive prefix of the 16 counts over blocks (hipCUB)
//   pass C  C2[ab] = L2[a] + rank(L2[b], a): the row before the first suffix starting with "ab"; header constants
//   pass D  counters = C2 + prefix
//   pass E  per-dimer arrays: one lane per 96-row record reads the planes (dword-aligned: 96 = 3 x 32), forms the 16
//           match masks without the filler rows, writes them and their popcounts; scan; counters = C2 + prefix
// 3 Gbp: 23.4 M plane records (3.0 GB) + 16 x 31.3 M per-dimer records (8.0 GB), built in ~0.1 s.
#include "fmindex_device.h"
#include <hipcub/hipcub.hpp>

namespace nvb {

struct Count16 { uint32_t v[16]; };
struct Count16Sum {
    __host__ __device__ __forceinline__ Count16 operator()(const Count16& x, const Count16& y) const
    { Count16 r; for (int i = 0; i < 16; ++i) r.v[i] = x.v[i] + y.v[i]; return r; }
};

// BWT symbol of SA row i (i != primary) on the source layout
__device__ __forceinline__ uint32_t src_bwt(const Fmi& f, const uint32_t i)
{
    const uint32_t k = (i < f.primary) ? i : i - 1u;
    const uint4 w = f.rec[2ull * (k >> 6)];
    const uint32_t word = comp(w, (k & 63u) >> 4);
    return (word >> (30u - ((k & 15u) << 1))) & 3u;
}

// one workgroup of 128 lanes = one record
__global__ void __launch_bounds__(128)
dimer_rows_kernel(const Fmi f, uint32_t* __restrict__ out /* header + records */, Count16* __restrict__ counts)
{
    __shared__ uint32_t part[2][16];
    const uint32_t blk  = blockIdx.x;
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint64_t row  = uint64_t(blk) * 128u + threadIdx.x;
    const bool in = row <= uint64_t(f.length);
    uint32_t nib = 0;
    if (in && uint32_t(row) != f.primary)
    {
        const uint32_t i = uint32_t(row);
        const uint32_t k = (i < f.primary) ? i : i - 1u;
        const Record r = load_record(f, k >> 6);
        const uint32_t word = comp(r.bwt, (k & 63u) >> 4);
        const uint32_t b = (word >> (30u - ((k & 15u) << 1))) & 3u;
        const uint32_t j = f.L2[b] + comp(r.occ, b) + block_count(r.bwt, (k & 63u) + 1u, b);        // LF(i)
        uint32_t a = 0;
        if (j == f.primary) {
            DimerHeader* h = reinterpret_cast<DimerHeader*>(out);
            h->p1 = i; h->fill1 = b;
        }
        else a = src_bwt(f, j);
        nib = a * 4u + b;
    }
    uint64_t pl[4];
    #pragma unroll
    for (int p = 0; p < 4; ++p) pl[p] = __builtin_amdgcn_ballot_w64(((nib >> p) & 1u) != 0u);
    const uint64_t valid = __builtin_amdgcn_ballot_w64(in);
    uint32_t* rec = out + 32u + 32ull * blk;
    if (lane < 4u)
    {
        // plane `lane`: this wave's 64 rows are dwords 2*wv, 2*wv+1 of the plane's uint4
        const uint64_t v = lane == 0u ? pl[0] : lane == 1u ? pl[1] : lane == 2u ? pl[2] : pl[3];
        rec[16u + 4u * lane + 2u * wv]      = uint32_t(v);
        rec[16u + 4u * lane + 2u * wv + 1u] = uint32_t(v >> 32);
    }
    if (lane < 16u)
    {
        const uint64_t m = ((lane & 1u) ? pl[0] : ~pl[0]) & ((lane & 2u) ? pl[1] : ~pl[1]) &
                           ((lane & 4u) ? pl[2] : ~pl[2]) & ((lane & 8u) ? pl[3] : ~pl[3]) & valid;
        part[wv][lane] = __popcll(m);
    }
    __syncthreads();
    if (threadIdx.x < 16u)
    {
        // counter slot b*4+a of nibble value a*4+b
        const uint32_t v = threadIdx.x, a = v >> 2, b = v & 3u;
        counts[blk].v[b * 4u + a] = part[0][v] + part[1][v];
    }
}

__global__ void dimer_header_kernel(const Fmi f, uint32_t n_records, uint32_t pd_stride, uint32_t* __restrict__ out)
{
    __shared__ uint32_t c2s[16];
    DimerHeader* h = reinterpret_cast<DimerHeader*>(out);
    const uint32_t t = threadIdx.x;
    if (t < 16u)
    {
        const uint32_t b = t >> 2, a = t & 3u;                    // slot b*4+a
        c2s[t] = f.L2[a] + fm_rank(f, f.L2[b], a);
        h->C2[t] = c2s[t];
    }
    __syncthreads();
    if (t < 4u)
    {
        const uint32_t k = c2s[t * 4u] + c2s[t * 4u + 1u] + c2s[t * 4u + 2u] + c2s[t * 4u + 3u];
        h->S[t]  = f.L2[t] - k;
        h->T[t]  = 0u - k;
    }
    if (t == 0u) { h->magic = DIMER_MAGIC; h->length = f.length; h->primary = f.primary; h->n_records = n_records; h->pd_stride = pd_stride;
                   h->L2_chk = f.L2[1] ^ ((f.L2[2] << 11) | (f.L2[2] >> 21)); }
}

__global__ void __launch_bounds__(256)
dimer_counters_kernel(uint32_t n_records, const Count16* __restrict__ excl, uint32_t* __restrict__ out)
{
    const uint64_t id = uint64_t(blockIdx.x) * 256u + threadIdx.x;
    if (id >= uint64_t(n_records) * 16u) return;
    const uint32_t blk = uint32_t(id >> 4), slot = uint32_t(id) & 15u;
    const uint32_t c2 = out[16u + slot];
    out[32u + 32ull * blk + slot] = c2 + excl[blk].v[slot];
}

// one lane per per-dimer record r (rows 96r .. 96r+95)
__global__ void __launch_bounds__(256)
pd_masks_kernel(uint32_t length, uint32_t primary, uint32_t n_records, uint32_t pd_stride, uint32_t* __restrict__ out, Count16* __restrict__ counts)
{
    const uint32_t r = blockIdx.x * 256u + threadIdx.x;
    if (r >= pd_stride) return;
    const uint32_t p1 = reinterpret_cast<const DimerHeader*>(out)->p1;
    uint4* pd = reinterpret_cast<uint4*>(out + 32u + 32ull * n_records);
    uint32_t pl[4][3];
    uint32_t valid[3];
    #pragma unroll
    for (uint32_t t = 0; t < 3u; ++t)
    {
        const uint64_t g = 3ull * r + t;                       // dword of the global plane stream = rows 32g .. 32g+31
        const uint64_t row0 = 32ull * g;
        const bool have = (g >> 2) < n_records;
        #pragma unroll
        for (int p = 0; p < 4; ++p)
            pl[p][t] = have ? out[32u + 32ull * (g >> 2) + 16u + 4u * p + uint32_t(g & 3u)] : 0u;
        uint32_t v = 0u;
        if (row0 <= uint64_t(length))
        {
            const uint64_t cnt = uint64_t(length) + 1u - row0;   // rows row0 .. length
            v = cnt >= 32u ? 0xFFFFFFFFu : ((1u << uint32_t(cnt)) - 1u);
        }
        if (uint64_t(primary) >= row0 && uint64_t(primary) < row0 + 32u) v &= ~(1u << (primary & 31u));
        if (p1 != 0xFFFFFFFFu && uint64_t(p1) >= row0 && uint64_t(p1) < row0 + 32u) v &= ~(1u << (p1 & 31u));
        valid[t] = v;
    }
    Count16 c;
    #pragma unroll
    for (uint32_t v = 0; v < 16u; ++v)
    {
        uint32_t m[3];
        #pragma unroll
        for (uint32_t t = 0; t < 3u; ++t)
            m[t] = ((v & 1u) ? pl[0][t] : ~pl[0][t]) & ((v & 2u) ? pl[1][t] : ~pl[1][t]) &
                   ((v & 4u) ? pl[2][t] : ~pl[2][t]) & ((v & 8u) ? pl[3][t] : ~pl[3][t]) & valid[t];
        c.v[v] = __popc(m[0]) + __popc(m[1]) + __popc(m[2]);
        pd[uint64_t(v) * pd_stride + r] = make_uint4(0u, m[0], m[1], m[2]);
    }
    counts[r] = c;
}

__global__ void __launch_bounds__(256)
pd_counters_kernel(uint32_t n_records, uint32_t pd_stride, const Count16* __restrict__ excl, uint32_t* __restrict__ out)
{
    const uint32_t r = blockIdx.x * 256u + threadIdx.x;
    if (r >= pd_stride) return;
    uint4* pd = reinterpret_cast<uint4*>(out + 32u + 32ull * n_records);
    const Count16 e = excl[r];
    #pragma unroll
    for (uint32_t v = 0; v < 16u; ++v)
    {
        const uint32_t c2 = out[16u + (v & 3u) * 4u + (v >> 2)];                 // header C2, slot b*4+a
        reinterpret_cast<uint32_t*>(pd + uint64_t(v) * pd_stride + r)[0] = c2 + e.v[v];
    }
}

__global__ void dimer_init_header_kernel(uint32_t* __restrict__ out)
{
    if (threadIdx.x < 32u) out[threadIdx.x] = 0u;
    if (threadIdx.x == 0u) { reinterpret_cast<DimerHeader*>(out)->p1 = 0xFFFFFFFFu; }
}

static inline uint64_t align256(uint64_t x) { return (x + 255ull) & ~255ull; }
static inline uint32_t dimer_records(uint32_t length) { return uint32_t((uint64_t(length) + 1u) >> 7) + 1u; }
static inline uint32_t pd_records(uint32_t length) { return uint32_t((uint64_t(length) + 1u) / 96u) + 1u; }

} // namespace nvb

using namespace nvb;

NVB_API uint64_t nvbio_hip_fm_dimer_index_bytes(uint32_t length)
{
    return 128ull + 128ull * dimer_records(length) + 16ull * 16ull * pd_records(length);
}

NVB_API uint64_t nvbio_hip_fm_build_dimer_index_temp_bytes(uint32_t length)
{
    const uint32_t nr = pd_records(length);          // >= dimer_records(length): the same temp serves both scans
    size_t scan = 0;
    (void)hipcub::DeviceScan::ExclusiveScan(nullptr, scan, (Count16*)nullptr, (Count16*)nullptr, Count16Sum(), Count16{}, int(nr));
    return align256(uint64_t(nr) * sizeof(Count16)) + align256(scan) + 256u;
}

NVB_API int nvbio_hip_fm_build_dimer_index(const nvbio_hip_fmindex* fmi, uint32_t* out_dimer, void* temp, uint64_t temp_bytes, void* stream)
{
    if (!fmi || !fmi->bwt_occ || !out_dimer || !temp) return hipErrorInvalidValue;
    if (fmi->length == 0 || fmi->length >= 0xFFFFFF00u) return hipErrorInvalidValue;
    if (temp_bytes < nvbio_hip_fm_build_dimer_index_temp_bytes(fmi->length)) return hipErrorInvalidValue;
    if ((reinterpret_cast<uintptr_t>(out_dimer) & 127u) != 0) return hipErrorInvalidValue;       // records must be line-aligned
    Fmi f = make_fmi(fmi);
    f.ktab = nullptr; f.ktab_k = 0; f.dm.base = nullptr;
    const uint32_t nr = dimer_records(fmi->length), npd = pd_records(fmi->length);
    Count16* counts = reinterpret_cast<Count16*>(temp);
    void*  scan_tmp = reinterpret_cast<uint8_t*>(temp) + align256(uint64_t(npd) * sizeof(Count16));
    size_t scan_bytes = size_t(temp_bytes - align256(uint64_t(npd) * sizeof(Count16)));
    hipStream_t s = to_stream(stream);
    g_last_kernel = "dimer_rows_kernel";
    hipLaunchKernelGGL(dimer_init_header_kernel, dim3(1), dim3(64), 0, s, out_dimer);
    hipLaunchKernelGGL(dimer_rows_kernel, dim3(nr), dim3(128), 0, s, f, out_dimer, counts);
    if (hipError_t e = hipGetLastError()) return e;
    if (hipError_t e = hipcub::DeviceScan::ExclusiveScan(scan_tmp, scan_bytes, counts, counts, Count16Sum(), Count16{}, int(nr), s)) return e;
    hipLaunchKernelGGL(dimer_header_kernel, dim3(1), dim3(64), 0, s, f, nr, npd, out_dimer);
    hipLaunchKernelGGL(dimer_counters_kernel, dim3(uint32_t((uint64_t(nr) * 16u + 255u) / 256u)), dim3(256), 0, s, nr, counts, out_dimer);
    if (hipError_t e = hipGetLastError()) return e;
    hipLaunchKernelGGL(pd_masks_kernel, dim3((npd + 255u) / 256u), dim3(256), 0, s, fmi->length, fmi->primary, nr, npd, out_dimer, counts);
    if (hipError_t e = hipGetLastError()) return e;
    if (hipError_t e = hipcub::DeviceScan::ExclusiveScan(scan_tmp, scan_bytes, counts, counts, Count16Sum(), Count16{}, int(npd), s)) return e;
    hipLaunchKernelGGL(pd_counters_kernel, dim3((npd + 255u) / 256u), dim3(256), 0, s, nr, npd, counts, out_dimer);
    return hipGetLastError();
}

NVB_API int nvbio_hip_fm_attach_dimer_index(nvbio_hip_fmindex* fmi, const uint32_t* dimer, void* stream)
{
    if (!fmi) return hipErrorInvalidValue;
    if (!dimer) { fmi->dimer = nullptr; return hipSuccess; }
    DimerHeader h;
    if (hipError_t e = hipMemcpyAsync(&h, dimer, sizeof(h), hipMemcpyDeviceToHost, to_stream(stream))) return e;
    if (hipError_t e = hipStreamSynchronize(to_stream(stream))) return e;
    if (h.magic != uint32_t(DIMER_MAGIC) || h.length != fmi->length || h.primary != fmi->primary) return hipErrorInvalidValue;
    if (h.L2_chk != (fmi->L2[1] ^ ((fmi->L2[2] << 11) | (fmi->L2[2] >> 21)))) return hipErrorInvalidValue;
    fmi->dimer = dimer;
    fmi->dimer_p1 = h.p1; fmi->dimer_fill1 = h.fill1;
    for (int c = 0; c < 4; ++c) { fmi->dimer_S[c] = h.S[c]; fmi->dimer_T[c] = h.T[c]; }
    return hipSuccess;
}
