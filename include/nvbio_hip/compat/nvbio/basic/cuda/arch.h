// compat/nvbio/basic/cuda/arch.h -- device queries and launch helpers (nvbio/basic/cuda/arch.h:60-120, arch_inl.h): what nvBowtie asks of
// the device when it sizes a launch (max_active_blocks), checks a kernel (check_error) or reports the device.
//
// WARPS.  The reference's applications are written for 32-lane warps: they size shared broadcast slots as BLOCKDIM >> 5, index them with
// warp_id() = threadIdx.x >> 5 and aggregate atomics over __ballot masks held in 32-bit words (nvBowtie/bowtie2/cuda/utils.h:58-71).  A
// gfx950 wavefront has 64 lanes.  This layer therefore presents each wavefront as TWO virtual warps of 32 lanes: Arch::WARP_SIZE = 32,
// warp_tid() / warp_id() as in the reference, and -- for translation units that include this header -- __ballot / __any / __all over the
// calling lane's 32-lane half.  Code written to the reference's warp contract (leader election by popc of the mask below my lane, a
// broadcast through a per-warp shared slot) then runs unchanged: each half elects its own leader, both halves execute in lockstep.
#pragma once
#include "../types.h"
#include "../numbers.h"
#include "../console.h"
#include "../exceptions.h"
#if defined(__HIPCC__)
#include <thrust/version.h>
#include <thrust/device_vector.h>
#include <thrust/host_vector.h>
// rocPRIM / hipCUB call the 64-lane ::__ballot / ::__any / ::__all themselves: they are parsed here, BEFORE the 32-lane names below are
// defined, so that the primitives this layer builds on keep the hardware's meaning (a later first include of them would fail to compile
// rather than change meaning: the macros expand to namespace-qualified calls)
#include <rocprim/rocprim.hpp>
#include <hipcub/hipcub.hpp>
#endif

namespace nvbio {
namespace cuda {

struct Arch
{
    static const uint32 LOG_WARP_SIZE = 5;
    static const uint32 WARP_SIZE     = 1u << LOG_WARP_SIZE;      // virtual warps, see above
};

#if defined(__HIPCC__)
namespace priv {
inline hipDeviceProp_t current_device_properties()
{
    int device = 0;
    hipDeviceProp_t p;
    (void)hipGetDevice(&device);
    (void)hipGetDeviceProperties(&p, device);
    return p;
}
/// the calling lane's half of the wavefront's ballot, as a 32-bit mask with the half's lanes at bits 0..31
NVBIO_FORCEINLINE __device__ uint32 half_ballot(const int predicate)
{
    const unsigned long long m = __ballot(predicate);
    return uint32(m >> (__lane_id() & 32u));
}
NVBIO_FORCEINLINE __device__ uint32 half_active() { return half_ballot(1); }
} // namespace priv

/// compute-capability style (major, minor) of the current device: the gfx number split as gfx<major><minor> (950 -> 9, 50)
inline void device_arch(uint32& major, uint32& minor)
{
    const hipDeviceProp_t p = priv::current_device_properties();
    major = uint32(p.major); minor = uint32(p.minor);
}
inline uint32 max_grid_size() { return uint32(priv::current_device_properties().maxGridSize[0]); }
inline size_t multiprocessor_count() { return size_t(priv::current_device_properties().multiProcessorCount); }

template <typename KernelFunction>
inline hipFuncAttributes function_attributes(KernelFunction kernel)
{
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(kernel));
    return a;
}
template <typename KernelFunction>
inline size_t num_registers(KernelFunction kernel) { return size_t(function_attributes(kernel).numRegs); }

/// resident blocks of `kernel` per compute unit / on the whole device at this block size
template <typename KernelFunction>
inline size_t max_active_blocks_per_multiprocessor(KernelFunction kernel, const size_t CTA_SIZE, const size_t dynamic_smem_bytes)
{
    int n = 0;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, int(CTA_SIZE), dynamic_smem_bytes);
    return size_t(n > 0 ? n : 1);
}
template <typename KernelFunction>
inline size_t max_active_blocks(KernelFunction kernel, const size_t CTA_SIZE, const size_t dynamic_smem_bytes)
{
    return max_active_blocks_per_multiprocessor(kernel, CTA_SIZE, dynamic_smem_bytes) * multiprocessor_count();
}
template <typename KernelFunction>
inline size_t max_blocksize_with_highest_occupancy(KernelFunction kernel, size_t dynamic_smem_bytes_per_thread)
{
    int grid = 0, block = 0;
    (void)dynamic_smem_bytes_per_thread;
    (void)hipOccupancyMaxPotentialBlockSize(&grid, &block, kernel, 0, 0);
    return size_t(block > 0 ? block : 256);
}

inline bool is_tcc_enabled() { return false; }

/// throw a cuda_error if the last runtime call or launch failed
inline void check_error(const char* message)
{
    const hipError_t error = hipGetLastError();
    if (error != hipSuccess)
    {
        const char* text = hipGetErrorString(error);
        log_error(stderr, "%s: %s\n", message, text);
        throw cuda_error(text);
    }
}
#endif // __HIPCC__

/// barrier over groups of N consecutive threads: whole-block barrier on the device, nothing on the host
template <uint32 N>
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE void syncthreads()
{
#if defined(NVBIO_DEVICE_COMPILATION)
    __syncthreads();
#endif
}

} // namespace cuda
} // namespace nvbio

#if defined(__HIPCC__)
// the 32-lane warp contract for code compiled after this point (the reference's arch.h redefines the same names for CUDA >= 9, arch.h:46-59)
#undef  __ballot
#undef  __any
#undef  __all
#define __ballot(p) nvbio::cuda::priv::half_ballot(p)
#define __any(p)    (nvbio::cuda::priv::half_ballot(p) != 0u)
#define __all(p)    (nvbio::cuda::priv::half_ballot(p) == nvbio::cuda::priv::half_active())
#endif
