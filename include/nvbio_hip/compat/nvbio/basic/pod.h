// compat/nvbio/basic/pod.h -- pod_type<T>: the plain vector type a small struct is moved through memory as (nvbio/basic/pod.h:36-78;
// io::Alignment travels as one uint2, BestAlignments as one uint4), with read / write helpers and the two proxy types built on them.
#pragma once
#include "types.h"
#include <string.h>

namespace nvbio {

template <typename T> struct pod_type { typedef T type; };

template <typename T>
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE void write(T* ptr, const T& e)
{
    typedef typename pod_type<T>::type P;
    static_assert(sizeof(P) == sizeof(T), "pod_type must have the size of the type it carries");
    *reinterpret_cast<P*>(ptr) = reinterpret_cast<const P&>(e);
}
template <typename T>
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE T read(const T* ptr)
{
    typedef typename pod_type<T>::type P;
    const P p = *reinterpret_cast<const P*>(ptr);
    T out; memcpy(&out, &p, sizeof(T));
    return out;
}
template <typename T> struct pod_writer
{
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE pod_writer() : ptr(NULL) {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE pod_writer(T& e) : ptr(&e) {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE void operator=(const T& e) { write(ptr, e); }
    T* ptr;
};
template <typename T> struct pod_reader
{
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE pod_reader() : ptr(NULL) {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE pod_reader(const T& e) : ptr(&e) {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE operator T() { return read(ptr); }
    const T* ptr;
};

} // namespace nvbio
