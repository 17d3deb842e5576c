// compat/nvbio/basic/exceptions.h -- the four exception types the reference throws and its applications catch
// (nvbio/basic/exceptions.h:38-80): printf-style constructors, what().  Header-only here (the reference defines the constructors
// in exceptions.cpp); each object carries its own message.
#pragma once
#include "types.h"
#include <stdarg.h>
#include <stdio.h>
#include <string>

namespace nvbio {
namespace priv {
inline std::string format_message(const char* format, va_list args)
{
    char text[4096];
    vsnprintf(text, sizeof(text), format, args);
    return std::string(text);
}
} // namespace priv

#define NVBIO_HIP_EXCEPTION_TYPE(name)                                                                                   \
    struct name                                                                                                          \
    {                                                                                                                    \
        name(const char* format, ...) { va_list a; va_start(a, format); m_what = priv::format_message(format, a); va_end(a); } \
        const char* what() const { return m_what.c_str(); }                                                              \
    private:                                                                                                             \
        std::string m_what;                                                                                              \
    };
NVBIO_HIP_EXCEPTION_TYPE(cuda_error)
NVBIO_HIP_EXCEPTION_TYPE(bad_alloc)
NVBIO_HIP_EXCEPTION_TYPE(runtime_error)
NVBIO_HIP_EXCEPTION_TYPE(logic_error)
#undef NVBIO_HIP_EXCEPTION_TYPE

} // namespace nvbio
