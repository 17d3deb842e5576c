// tools/port/cuda_runtime.h -- NOT part of the library: nvBowtie's own sources include <cuda_runtime.h>; a port replaces that include by the HIP
// runtime and the renames in tools/port_cuda_calls.h.  Only tools/nvbowtie_tu_check.py puts this directory on the include path.
#pragma once
#include "../port_cuda_calls.h"
