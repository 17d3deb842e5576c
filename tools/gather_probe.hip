// gather_probe.hip -- micro-probe: what does a random 32-byte record gather cost on MI355X, and
// which request sizes does the L2 send to the fabric for it?  (run under rocprofv3 --pmc
// TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t hash32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x;
}
__device__ __forceinline__ uint64_t rec_index(uint32_t tid, uint32_t salt, uint64_t nrec)
{
    const uint64_t h = (uint64_t(hash32(tid ^ salt)) << 32) | hash32(tid * 2654435761u + salt);
    return h % nrec;
}

template <int MODE>
__device__ __forceinline__ uint4 ld16(const uint4* p)
{
    uint4 v;
    if (MODE == 0) v = *p;
    else if (MODE == 1) {
        typedef unsigned int u4 __attribute__((ext_vector_type(4)));
        const u4 t = __builtin_nontemporal_load(reinterpret_cast<const u4*>(p));
        v = make_uint4(t.x, t.y, t.z, t.w);
    }
    else if (MODE == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (MODE == 3) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (MODE == 4) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (MODE == 5) asm volatile("global_load_dwordx4 %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// BYTES per record read: 16, 32, 64 or 128; records are 32-byte aligned slots of `stride` bytes
template <int MODE, int BYTES>
__global__ void __launch_bounds__(256) gather(const uint4* __restrict__ tab, uint64_t nrec, uint32_t stride16, uint32_t salt, uint32_t* __restrict__ out)
{
    const uint32_t tid = blockIdx.x * 256u + threadIdx.x;
    const uint4* p = tab + rec_index(tid, salt, nrec) * stride16;
    uint32_t acc = 0;
    #pragma unroll
    for (int i = 0; i < BYTES / 16; ++i) { const uint4 v = ld16<MODE>(p + i); acc += v.x ^ v.y ^ v.z ^ v.w; }
    out[tid] = acc;
}

template <int MODE, int BYTES>
static void run(const char* name, const uint4* tab, uint64_t nrec, uint32_t stride16, uint32_t q, uint32_t* out)
{
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    gather<MODE, BYTES><<<q / 256, 256>>>(tab, nrec, stride16, 1u, out);
    CHECK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int r = 0; r < 3; ++r) {
        CHECK(hipEventRecord(e0));
        gather<MODE, BYTES><<<q / 256, 256>>>(tab, nrec, stride16, 77u + r, out);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    printf("%-34s %8.3f ms  %7.2f Gq/s  useful %7.1f GB/s\n", name, best, q / best / 1e6, double(q) * BYTES / best / 1e6);
}

int main(int argc, char** argv)
{
    const uint64_t table_bytes = argc > 1 ? strtoull(argv[1], 0, 10) : 1500000000ull;
    const uint32_t q = 1u << 28;
    uint32_t* out; CHECK(hipMalloc(&out, size_t(q) * 4));
    for (int mem = 0; mem < 3; ++mem) {
        void* tab = nullptr;
        const char* mname = mem == 0 ? "hipMalloc" : mem == 1 ? "uncached" : "finegrained";
        hipError_t e = mem == 0 ? hipMalloc(&tab, table_bytes)
                     : mem == 1 ? hipExtMallocWithFlags(&tab, table_bytes, hipDeviceMallocUncached)
                                : hipExtMallocWithFlags(&tab, table_bytes, hipDeviceMallocFinegrained);
        if (e != hipSuccess) { printf("%s alloc failed: %s\n", mname, hipGetErrorString(e)); (void)hipGetLastError(); continue; }
        CHECK(hipMemset(tab, 0x5a, table_bytes));
        printf("== memory: %s, table %.0f MB, %u random gathers\n", mname, table_bytes / 1e6, q);
        const uint4* t = (const uint4*)tab;
        run<0, 32>("plain 32B (stride 32)", t, table_bytes / 32, 2, q, out);
        run<1, 32>("nontemporal 32B", t, table_bytes / 32, 2, q, out);
        run<2, 32>("sc1 32B", t, table_bytes / 32, 2, q, out);
        run<3, 32>("sc0 sc1 32B", t, table_bytes / 32, 2, q, out);
        run<4, 32>("sc0 sc1 nt 32B", t, table_bytes / 32, 2, q, out);
        run<5, 32>("sc0 32B", t, table_bytes / 32, 2, q, out);
        run<0, 16>("plain 16B (stride 32)", t, table_bytes / 32, 2, q, out);
        run<0, 64>("plain 64B (stride 64)", t, table_bytes / 64, 4, q, out);
        run<0, 128>("plain 128B (stride 128)", t, table_bytes / 128, 8, q, out);
        CHECK(hipFree(tab));
    }
    // small table: L2 / MALL resident
    for (uint64_t mb : {16ull, 128ull}) {
        void* tab; CHECK(hipMalloc(&tab, mb << 20)); CHECK(hipMemset(tab, 1, mb << 20));
        char nm[64]; snprintf(nm, sizeof nm, "plain 32B, %llu MB table", (unsigned long long)mb);
        run<0, 32>(nm, (const uint4*)tab, (mb << 20) / 32, 2, q, out);
        CHECK(hipFree(tab));
    }
    return 0;
}
