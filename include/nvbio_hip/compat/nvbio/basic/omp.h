// compat/nvbio/basic/omp.h -- OpenMP entry points, or single-thread stand-ins when the TU is built without -fopenmp
// (nvbio/basic/omp.h)
#pragma once
#include <stdlib.h>
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

#include <stdio.h>
#include <algorithm>
/// The threads a parallel loop of this layer asks for.  Applications size OpenMP by the machine (nvBowtie.cpp:209:
/// omp_set_num_threads(omp_get_num_procs())), which inside a container with a CPU quota is far more than may run at once -- 256
/// threads on a 16-CPU quota turned the parallel SAM writer slower than one thread.  The cgroup quota, where there is one, caps it.
inline int usable_omp_threads()
{
    static const int cached = [] {
        int n = std::max(omp_get_max_threads(), 1);
        long long quota = -1, period = -1;
        if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r"))                                          // cgroup v2: "<quota|max> <period>"
        { char q[32] = { 0 }; if (fscanf(f, "%31s %lld", q, &period) == 2 && q[0] != 'm') quota = atoll(q); fclose(f); }
        else
        {
            if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r"))  { if (fscanf(g, "%lld", &quota) != 1) quota = -1; fclose(g); }
            if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%lld", &period) != 1) period = -1; fclose(g); }
        }
        if (quota > 0 && period > 0) n = std::min<long long>(n, std::max<long long>(1, (quota + period - 1) / period));
        return n;
    }();
    return cached;
}
