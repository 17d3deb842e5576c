// compat/nvbio/basic/cuda/ldg.h -- nvbio::cuda::ldg_pointer<T> (nvbio/basic/cuda/ldg.h): a read-only pointer the
// reference routes through __ldg.  gfx950 has no separate read-only path; plain global loads are used.
#pragma once
#include "../types.h"

namespace nvbio {
namespace cuda {

template <typename T>
struct ldg_pointer
{
    typedef T                                value_type;
    typedef T                                reference;
    typedef const T*                         pointer;
    typedef ptrdiff_t                        difference_type;
    typedef std::random_access_iterator_tag  iterator_category;

    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE ldg_pointer() : base(nullptr) {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE ldg_pointer(const T* p) : base(p) {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE T operator[](const uint64 i) const { return base[i]; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE T operator*() const { return *base; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE ldg_pointer operator+(const difference_type d) const { return ldg_pointer(base + d); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE ldg_pointer operator-(const difference_type d) const { return ldg_pointer(base - d); }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE difference_type operator-(const ldg_pointer o) const { return base - o.base; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE ldg_pointer& operator+=(const difference_type d) { base += d; return *this; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE ldg_pointer& operator++() { ++base; return *this; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool operator==(const ldg_pointer o) const { return base == o.base; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool operator!=(const ldg_pointer o) const { return base != o.base; }
    const T* base;
};
template <typename T> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE ldg_pointer<T> make_ldg_pointer(const T* p) { return ldg_pointer<T>(p); }

} // namespace cuda
} // namespace nvbio
