// compat/nvbio/basic/omp.h -- OpenMP entry points, or single-thread stand-ins when the TU is built without -fopenmp
// (nvbio/basic/omp.h)
#pragma once
#if defined(_OPENMP)
#include <omp.h>
#else
inline void omp_set_nested(const int) {}
inline void omp_set_num_threads(const int) {}
inline int  omp_get_max_threads() { return 1; }
inline int  omp_get_num_threads() { return 1; }
inline int  omp_get_thread_num()  { return 0; }
inline int  omp_get_num_procs()   { return 1; }
#endif
