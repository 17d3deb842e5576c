// pool_probe.hip -- does the default HIP memory pool keep freed blocks (hipMemPoolAttrReleaseThreshold) so that a driver's
// per-batch hipMallocAsync / hipFreeAsync pairs cost nothing after the first batch?
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
int main()
{
    hipMemPool_t pool; hipDeviceGetDefaultMemPool(&pool, 0);
    uint64_t keep = ~0ull; printf("set threshold: %d\n", (int)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep));
    const size_t sizes[4] = { size_t(16) << 30, size_t(1) << 30, size_t(80) << 20, size_t(4) << 20 };
    for (int round = 0; round < 4; ++round)
    {
        const auto t0 = std::chrono::steady_clock::now();
        void* p[4];
        for (int i = 0; i < 4; ++i) if (hipMallocAsync(&p[i], sizes[i], nullptr) != hipSuccess) { printf("alloc failed\n"); return 1; }
        hipMemsetAsync(p[0], 0, sizes[0], nullptr);
        for (int i = 0; i < 4; ++i) hipFreeAsync(p[i], nullptr);
        hipStreamSynchronize(nullptr);
        printf("round %d pooled: %.2f ms\n", round, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
    for (int round = 0; round < 3; ++round)
    {
        const auto t0 = std::chrono::steady_clock::now();
        void* p[4];
        for (int i = 0; i < 4; ++i) hipMalloc(&p[i], sizes[i]);
        hipMemsetAsync(p[0], 0, sizes[0], nullptr);
        hipStreamSynchronize(nullptr);
        for (int i = 0; i < 4; ++i) hipFree(p[i]);
        printf("round %d hipMalloc/hipFree: %.2f ms\n", round, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
    return 0;
}
