// compat/nvbio/basic/static_vector.h -- StaticVector<T,DIM> (nvbio/basic/static_vector.h:36-260): a fixed-size vector with
// component-wise arithmetic, the type rank_all() returns and the reference's rank test accumulates its expected counts in
// (nvbio-test/rank_test.cu:58-80).  DIM 2 / 4 of a 32- or 64-bit unsigned convert to and from the matching HIP vector
// (uint2 / uint4 / ulonglong2 / ulonglong4).
#pragma once
#include "types.h"

namespace nvbio {

template <typename T, uint32 DIM>
struct StaticVectorBase
{
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE const T& operator[](const uint32 i) const { return data[i]; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE       T& operator[](const uint32 i)       { return data[i]; }
    T data[DIM];
};

namespace priv {
template <typename T, uint32 DIM> struct has_vector { static const bool value = false; typedef null_type type; };
template <> struct has_vector<uint32, 2> { static const bool value = true; typedef uint2 type; };
template <> struct has_vector<uint32, 4> { static const bool value = true; typedef uint4 type; };
template <> struct has_vector<uint64, 2> { static const bool value = true; typedef ulonglong2 type; };
template <> struct has_vector<uint64, 4> { static const bool value = true; typedef ulonglong4 type; };
} // namespace priv

template <typename T, uint32 DIM>
struct StaticVector : public StaticVectorBase<T, DIM>
{
    typedef StaticVectorBase<T, DIM>                    base;
    typedef typename priv::has_vector<T, DIM>::type     base_type;      ///< the HIP vector of the same shape, if there is one

    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE StaticVector() {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE explicit StaticVector(const T v) { for (uint32 d = 0; d < DIM; ++d) base::data[d] = v; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE StaticVector(const base_type& v)
    {
        static_assert(priv::has_vector<T, DIM>::value, "no HIP vector type of this shape");
        for (uint32 d = 0; d < DIM; ++d) base::data[d] = priv::vec_comp<base_type>::get(v, d);
    }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE StaticVector& operator=(const StaticVectorBase<T, DIM>& o) { for (uint32 d = 0; d < DIM; ++d) base::data[d] = o.data[d]; return *this; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE operator base_type() const
    {
        static_assert(priv::has_vector<T, DIM>::value, "no HIP vector type of this shape");
        base_type v;
        for (uint32 d = 0; d < DIM; ++d) priv::vec_comp<base_type>::put(v, d, base::data[d]);
        return v;
    }
};

template <typename T> struct vector_traits;
template <typename T, uint32 DIM_T> struct vector_traits< StaticVectorBase<T, DIM_T> > { typedef T value_type; static const uint32 DIM = DIM_T; };
template <typename T, uint32 DIM_T> struct vector_traits< StaticVector<T, DIM_T> >     { typedef T value_type; static const uint32 DIM = DIM_T; };

#define NVBIO_HIP_SV_OP(OP)                                                                                                           \
    template <typename T, uint32 DIM> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE                                                             \
    StaticVector<T, DIM>& operator OP##=(StaticVector<T, DIM>& a, const StaticVector<T, DIM>& b) { for (uint32 d = 0; d < DIM; ++d) a[d] OP##= b[d]; return a; } \
    template <typename T, uint32 DIM> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE                                                             \
    StaticVector<T, DIM> operator OP(const StaticVector<T, DIM>& a, const StaticVector<T, DIM>& b) { StaticVector<T, DIM> r(a); r OP##= b; return r; }
NVBIO_HIP_SV_OP(+) NVBIO_HIP_SV_OP(-) NVBIO_HIP_SV_OP(*) NVBIO_HIP_SV_OP(/)
#undef NVBIO_HIP_SV_OP

template <typename T, uint32 DIM> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE
StaticVector<T, DIM> min(const StaticVector<T, DIM>& a, const StaticVector<T, DIM>& b) { StaticVector<T, DIM> r; for (uint32 d = 0; d < DIM; ++d) r[d] = a[d] < b[d] ? a[d] : b[d]; return r; }
template <typename T, uint32 DIM> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE
StaticVector<T, DIM> max(const StaticVector<T, DIM>& a, const StaticVector<T, DIM>& b) { StaticVector<T, DIM> r; for (uint32 d = 0; d < DIM; ++d) r[d] = a[d] < b[d] ? b[d] : a[d]; return r; }
template <typename T, uint32 DIM> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool any(const StaticVector<T, DIM>& a) { for (uint32 d = 0; d < DIM; ++d) if (a[d]) return true; return false; }
template <typename T, uint32 DIM> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool all(const StaticVector<T, DIM>& a) { for (uint32 d = 0; d < DIM; ++d) if (!a[d]) return false; return true; }
template <typename T, uint32 DIM> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE
bool operator==(const StaticVector<T, DIM>& a, const StaticVector<T, DIM>& b) { for (uint32 d = 0; d < DIM; ++d) if (!(a[d] == b[d])) return false; return true; }
template <typename T, uint32 DIM> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE
bool operator!=(const StaticVector<T, DIM>& a, const StaticVector<T, DIM>& b) { return !(a == b); }

} // namespace nvbio
