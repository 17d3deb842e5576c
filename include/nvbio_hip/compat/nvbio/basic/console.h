// compat/nvbio/basic/console.h -- the levelled printf front the reference's tests and tools report through
// (nvbio/basic/console.h:33-80).  Header-only: one vfprintf per call behind a process-wide verbosity; colours are
// not reproduced.  Plumbing, not hot path -- present so that whole reference translation units compile.
#pragma once
#include <stdio.h>
#include <stdarg.h>

enum Verbosity { V_ERROR = 0, V_WARNING = 1, V_VISIBLE = 2, V_INFO = 3, V_STATS = 4, V_VERBOSE = 5, V_DEBUG = 6 };

namespace nvbio_hip_console {
inline Verbosity& level() { static Verbosity v = V_VERBOSE; return v; }
inline void emit(const Verbosity v, FILE* file, const char* tag, const char* fmt, va_list args)
{
    if (v > level()) return;
    if (tag) fputs(tag, file);
    vfprintf(file, fmt, args);
}
} // namespace nvbio_hip_console

inline void set_verbosity(Verbosity v) { nvbio_hip_console::level() = v; }
inline Verbosity get_verbosity() { return nvbio_hip_console::level(); }
inline void textcolor(unsigned int) {}

#define NVBIO_HIP_LOG_FN(name, verbosity, tag)                                                                          \
    inline void name(FILE* file, const char* string, ...)                                                               \
    { va_list a; va_start(a, string); nvbio_hip_console::emit(verbosity, file, tag, string, a); va_end(a); }            \
    inline void name##_cont(FILE* file, const char* string, ...)                                                        \
    { va_list a; va_start(a, string); nvbio_hip_console::emit(verbosity, file, NULL, string, a); va_end(a); }           \
    inline void name##_nl(FILE* file) { if (verbosity <= nvbio_hip_console::level()) fputc('\n', file); }
NVBIO_HIP_LOG_FN(log_visible, V_VISIBLE, "visible : ")
NVBIO_HIP_LOG_FN(log_info,    V_INFO,    "info    : ")
NVBIO_HIP_LOG_FN(log_stats,   V_STATS,   "stats   : ")
NVBIO_HIP_LOG_FN(log_verbose, V_VERBOSE, "verbose : ")
NVBIO_HIP_LOG_FN(log_debug,   V_DEBUG,   "debug   : ")
NVBIO_HIP_LOG_FN(log_warning, V_WARNING, "warning : ")
NVBIO_HIP_LOG_FN(log_error,   V_ERROR,   "error   : ")
#undef NVBIO_HIP_LOG_FN
