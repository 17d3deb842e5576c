// tools/port_cuda_calls.h -- NOT part of the library.  The renames a maintainer's port applies to the APPLICATION's own CUDA
// runtime calls when the reference's test programs are compiled as they lie (tools/ref_bind_check.py, `make -C oracle ref_tests`):
// the drop-in layer under include/nvbio_hip/compat replaces nvbio's headers, not the CUDA runtime, and ships no CUDA shim.
#pragma once
#include <hip/hip_runtime.h>
#define cudaEvent_t            hipEvent_t
#define cudaEventCreate        hipEventCreate
#define cudaEventRecord        hipEventRecord
#define cudaEventSynchronize   hipEventSynchronize
#define cudaEventElapsedTime   hipEventElapsedTime
#define cudaEventDestroy       hipEventDestroy
#define cudaThreadSynchronize  hipDeviceSynchronize
#define cudaDeviceSynchronize  hipDeviceSynchronize
