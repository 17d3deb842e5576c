// tools/port_cuda_calls.h -- NOT part of the library.  The renames a maintainer's port applies to the APPLICATION's own CUDA
// runtime calls when the reference's test programs are compiled as they lie (tools/ref_bind_check.py, `make -C oracle ref_tests`):
// the drop-in layer under include/nvbio_hip/compat replaces nvbio's headers, not the CUDA runtime, and ships no CUDA shim.
//
// Three things live here, all on the APPLICATION's side of the boundary:
//   1. renames of the CUDA runtime entry points the application calls itself (below);
//   2. tools/port/cuda_runtime.h, a one-line header that includes this file, because nvBowtie's own sources say #include <cuda_runtime.h>;
//   3. __CUDACC__: nvBowtie guards its kernels with `#if defined(__CUDACC__)` meaning "a device compiler is compiling this" -- hipcc is
//      one.  The macro is defined only AFTER the HIP and rocThrust configuration headers have been parsed, so that they keep seeing an AMD build.
#pragma once
#include <hip/hip_runtime.h>
#include <thrust/detail/config.h>
#include <thrust/device_vector.h>
#include <thrust/host_vector.h>
#include <thrust/copy.h>
#include <thrust/fill.h>
#include <thrust/sort.h>
#include <thrust/scan.h>
#include <thrust/reduce.h>
#include <thrust/transform.h>
#include <thrust/for_each.h>
#include <thrust/binary_search.h>
#include <thrust/merge.h>
#include <thrust/logical.h>
#include <thrust/iterator/counting_iterator.h>
#include <thrust/iterator/constant_iterator.h>
#include <thrust/iterator/transform_iterator.h>
#include <thrust/iterator/zip_iterator.h>
#include <rocprim/rocprim.hpp>
#include <hipcub/hipcub.hpp>
#if !defined(__CUDACC__)
#define __CUDACC__ 1
#define __CUDACC_VER_MAJOR__ 12
#define __CUDACC_VER_MINOR__ 0
#endif
// ... and `#if defined(__CUDA_ARCH__) && __CUDA_ARCH__ > 0` meaning "this is the device pass" (seed_hit_deque_array_inl.h:80: without it
// alloc_deque() compiles to `return NULL` and no seed hit is ever stored).  Same rule: defined after the toolchain's own headers.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__CUDA_ARCH__)
#define __CUDA_ARCH__ 900
#endif
#define cudaSetDevice               hipSetDevice
#define cudaGetDeviceCount          hipGetDeviceCount
#define cudaSetDeviceFlags          hipSetDeviceFlags
#define cudaDeviceMapHost           hipDeviceMapHost
#define cudaDeviceLmemResizeToMax   hipDeviceLmemResizeToMax
#define cudaDeviceProp              hipDeviceProp_t
#define cudaGetDeviceProperties     hipGetDeviceProperties
#define cudaMemGetInfo              hipMemGetInfo
#define cudaMemcpy                  hipMemcpy
#define cudaMemcpyDeviceToHost      hipMemcpyDeviceToHost
#define cudaMemcpyHostToDevice      hipMemcpyHostToDevice
#define cudaMemcpyDeviceToDevice    hipMemcpyDeviceToDevice
#define cudaDeviceGetLimit          hipDeviceGetLimit
#define cudaLimitStackSize          hipLimitStackSize
#define cudaError_t                 hipError_t
#define cudaSuccess                 hipSuccess
#define cudaGetLastError            hipGetLastError
#define cudaGetErrorString          hipGetErrorString
#define cudaEvent_t            hipEvent_t
#define cudaEventCreate        hipEventCreate
#define cudaEventRecord        hipEventRecord
#define cudaEventSynchronize   hipEventSynchronize
#define cudaEventElapsedTime   hipEventElapsedTime
#define cudaEventDestroy       hipEventDestroy
#define cudaThreadSynchronize  hipDeviceSynchronize
#define cudaDeviceSynchronize  hipDeviceSynchronize
