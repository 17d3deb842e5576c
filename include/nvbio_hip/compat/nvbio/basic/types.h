// compat/nvbio/basic/types.h -- the vocabulary types and qualifiers a caller written against nvbio expects
// (nvbio/basic/types.h, numbers.h), for translation units compiled by hipcc (device + host) or by a host
// compiler (host only).  Part of the drop-in template layer: `-I include/nvbio_hip/compat` makes the reference's
// own include lines (<nvbio/alignment/alignment.h>, <nvbio/fmindex/fmindex.h>, ...) resolve here.
#pragma once
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define NVBIO_HOST_DEVICE __host__ __device__
#define NVBIO_HOST        __host__
#define NVBIO_DEVICE      __device__
#define NVBIO_FORCEINLINE __forceinline__
#else
#define NVBIO_HOST_DEVICE
#define NVBIO_HOST
#define NVBIO_DEVICE
#define NVBIO_FORCEINLINE inline __attribute__((always_inline))
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define NVBIO_DEVICE_COMPILATION 1
#endif
#define NVBIO_CUDA_DEBUG_ASSERT(...)
#define NVBIO_CUDA_ASSERT(...)
#define NVBIO_CUDA_ASSERT_IF(...)
#define NVBIO_CUDA_DEBUG_STATEMENT(x)
#define NVBIO_CUDA_DEBUG_PRINT(...)
#if defined(NVBIO_HIP_COMPAT_DEBUG_TEXT)
/* diagnosis builds: print the FORMAT TEXT of the application's debug statements when their condition holds (the arguments are not evaluated:
 * several of nvBowtie's own debug lines do not compile, reduce_inl.h:186) */
#define NVBIO_HIP_FIRST_ARG(fmt, ...) fmt
#define NVBIO_CUDA_DEBUG_PRINT_IF(cond, ...) if (cond) printf("%s", NVBIO_HIP_FIRST_ARG(__VA_ARGS__, ""))
#elif defined(NVBIO_HIP_COMPAT_DEBUG_FULL)
/* ... or, for translation units whose debug statements do compile, the statements themselves */
#define NVBIO_CUDA_DEBUG_PRINT_IF(cond, ...) if (cond) printf(__VA_ARGS__)
#else
#define NVBIO_CUDA_DEBUG_PRINT_IF(...)
#endif
#define NVBIO_CUDA_DEBUG_CHECK_IF(...)
#define NVBIO_CUDA_DEBUG_SELECT(debug_val, normal_val) (normal_val)
#define NVBIO_VAR_UNUSED __attribute__((unused))

#include "version.h"
#include <stdint.h>
#include <stddef.h>
#include <iterator>
#include <limits>
#include <new>

namespace nvbio {

typedef uint8_t  uint8;   typedef int8_t  int8;
typedef uint16_t uint16;  typedef int16_t int16;
typedef uint32_t uint32;  typedef int32_t int32;
typedef uint64_t uint64;  typedef int64_t int64;

#if !defined(__HIPCC__)
struct uint2 { uint32 x, y; };
struct uint4 { uint32 x, y, z, w; };
struct ulonglong2 { unsigned long long x, y; };
struct ulonglong4 { unsigned long long x, y, z, w; };
inline uint2 make_uint2(uint32 x, uint32 y) { uint2 r = { x, y }; return r; }
inline uint4 make_uint4(uint32 x, uint32 y, uint32 z, uint32 w) { uint4 r = { x, y, z, w }; return r; }
inline ulonglong2 make_ulonglong2(unsigned long long x, unsigned long long y) { ulonglong2 r = { x, y }; return r; }
inline ulonglong4 make_ulonglong4(unsigned long long x, unsigned long long y, unsigned long long z, unsigned long long w) { ulonglong4 r = { x, y, z, w }; return r; }
#else
using ::uint2; using ::uint4; using ::ulonglong2; using ::ulonglong4;
using ::make_uint2; using ::make_uint4; using ::make_ulonglong2; using ::make_ulonglong4;
#endif
typedef ulonglong2 uint64_2;
typedef ulonglong4 uint64_4;

struct host_tag {};
struct device_tag {};
struct null_type {};

/// vector_type<T,N>::type and make_vector (nvbio/basic/types.h): uint32 -> uint2/uint4, uint64 -> ulonglong2/4
template <typename T, uint32 N> struct vector_type {};
template <> struct vector_type<uint32, 2> { typedef uint2 type; };
template <> struct vector_type<uint32, 4> { typedef uint4 type; };
template <> struct vector_type<uint64, 2> { typedef ulonglong2 type; };
template <> struct vector_type<uint64, 4> { typedef ulonglong4 type; };
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint2 make_vector(const uint32 x, const uint32 y) { return make_uint2(x, y); }
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint4 make_vector(const uint32 x, const uint32 y, const uint32 z, const uint32 w) { return make_uint4(x, y, z, w); }
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE ulonglong2 make_vector(const uint64 x, const uint64 y) { return make_ulonglong2(x, y); }
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE ulonglong4 make_vector(const uint64 x, const uint64 y, const uint64 z, const uint64 w) { return make_ulonglong4(x, y, z, w); }

template <typename T> struct signed_type {};
template <> struct signed_type<uint32> { typedef int32 type; };
template <> struct signed_type<int32>  { typedef int32 type; };
template <> struct signed_type<uint64> { typedef int64 type; };
template <> struct signed_type<int64>  { typedef int64 type; };
template <typename T> struct unsigned_type {};
template <> struct unsigned_type<uint32> { typedef uint32 type; };
template <> struct unsigned_type<int32>  { typedef uint32 type; };
template <> struct unsigned_type<uint64> { typedef uint64 type; };
template <> struct unsigned_type<int64>  { typedef uint64 type; };

/// to_const / reference_subtype / device_view_subtype / plain_view_subtype (types.h:168-204): the view a container hands to kernels
template <typename T> struct to_const           { typedef T type; };
template <typename T> struct to_const<T&>       { typedef const T& type; };
template <typename T> struct to_const<T*>       { typedef const T* type; };
template <typename T> struct to_const<const T&> { typedef const T& type; };
template <typename T> struct to_const<const T*> { typedef const T* type; };
template <typename T> struct reference_subtype            { typedef typename T::reference type; };
template <typename T> struct reference_subtype<T*>        { typedef T&                    type; };
template <typename T> struct reference_subtype<const T*>  { typedef const T&              type; };
template <>           struct reference_subtype<null_type> { typedef null_type             type; };
template <typename T> struct device_view_subtype            { typedef typename T::device_view_type type; };
template <>           struct device_view_subtype<null_type> { typedef null_type type; };
template <typename T> struct device_view_subtype<const T*>  { typedef const T*  type; };
template <typename T> struct device_view_subtype<T*>        { typedef T*        type; };
template <typename T> struct plain_view_subtype            { typedef typename T::plain_view_type       type; };
template <typename T> struct plain_view_subtype<const T>   { typedef typename T::const_plain_view_type type; };
template <>           struct plain_view_subtype<null_type> { typedef null_type type; };
template <typename T> struct plain_view_subtype<const T*>  { typedef const T*  type; };
template <typename T> struct plain_view_subtype<T*>        { typedef T*        type; };
/// reinterpret the bits of a value as another type of the same size (types.h: binary_cast)
template <typename Out, typename In>
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE Out binary_cast(const In in)
{
    static_assert(sizeof(Out) == sizeof(In), "binary_cast between types of different sizes");
    Out out; __builtin_memcpy(&out, &in, sizeof(Out)); return out;
}

/// same_type<A,B>::pred and equal<A,B>()   (nvbio/basic/types.h:222-230)
template <typename A, typename B> struct same_type { static const bool pred = false; };
template <typename A>             struct same_type<A, A> { static const bool pred = true; };
template <typename A, typename B> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool equal() { return same_type<A, B>::pred; }
/// a rounded up to a multiple of the power of two N   (types.h:283)
template <uint32 N, typename I> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE I align(const I a) { return N > 1u ? I((a + I(N - 1u)) & ~I(N - 1u)) : a; }
/// binary_switch / if_true selectors used in template signatures
template <typename A, typename B, uint32 N> struct binary_switch { typedef B type; };
template <typename A, typename B> struct binary_switch<A, B, 0> { typedef A type; };
template <bool B, typename T, typename F> struct if_true { typedef T type; };
template <typename T, typename F> struct if_true<false, T, F> { typedef F type; };

/// Field_traits<T>::min() / max() with the reference's values (numbers.h:795-850): the 32- and 64-bit signed extremes are
/// +-2^30 and +-2^62, not the type limits -- BestSink starts at -2^30 and streams use it as "no threshold"
template <typename T> struct Field_traits {};
template <> struct Field_traits<int8>   { NVBIO_FORCEINLINE NVBIO_HOST_DEVICE static int8   min() { return int8(-128); }        NVBIO_FORCEINLINE NVBIO_HOST_DEVICE static int8   max() { return int8(127); } };
template <> struct Field_traits<int16>  { NVBIO_FORCEINLINE NVBIO_HOST_DEVICE static int16  min() { return int16(-32768); }     NVBIO_FORCEINLINE NVBIO_HOST_DEVICE static int16  max() { return int16(32767); } };
template <> struct Field_traits<int32>  { NVBIO_FORCEINLINE NVBIO_HOST_DEVICE static int32  min() { return -(1 << 30); }        NVBIO_FORCEINLINE NVBIO_HOST_DEVICE static int32  max() { return (1 << 30); } };
template <> struct Field_traits<int64>  { NVBIO_FORCEINLINE NVBIO_HOST_DEVICE static int64  min() { return -(int64(1) << 62); } NVBIO_FORCEINLINE NVBIO_HOST_DEVICE static int64  max() { return int64(1) << 62; } };
template <> struct Field_traits<uint8>  { NVBIO_FORCEINLINE NVBIO_HOST_DEVICE static uint8  min() { return 0; } NVBIO_FORCEINLINE NVBIO_HOST_DEVICE static uint8  max() { return uint8(255); } };
template <> struct Field_traits<uint16> { NVBIO_FORCEINLINE NVBIO_HOST_DEVICE static uint16 min() { return 0; } NVBIO_FORCEINLINE NVBIO_HOST_DEVICE static uint16 max() { return uint16(0xFFFF); } };
template <> struct Field_traits<uint32> { NVBIO_FORCEINLINE NVBIO_HOST_DEVICE static uint32 min() { return 0; } NVBIO_FORCEINLINE NVBIO_HOST_DEVICE static uint32 max() { return 0xFFFFFFFFu; } };
template <> struct Field_traits<uint64> { NVBIO_FORCEINLINE NVBIO_HOST_DEVICE static uint64 min() { return 0; } NVBIO_FORCEINLINE NVBIO_HOST_DEVICE static uint64 max() { return ~uint64(0); } };

template <typename T> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE T min(const T a, const T b) { return a < b ? a : b; }
template <typename T> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE T max(const T a, const T b) { return a < b ? b : a; }
template <typename T> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE T max3(const T a, const T b, const T c) { return nvbio::max(nvbio::max(a, b), c); }
template <typename T> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE T min3(const T a, const T b, const T c) { return nvbio::min(nvbio::min(a, b), c); }
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 divide_ri(const uint32 a, const uint32 b) { return (a + b - 1u) / b; }
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 popc(const uint32 x) { return uint32(__builtin_popcount(x)); }
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 popc(const uint64 x) { return uint32(__builtin_popcountll(x)); }

namespace priv {
/// the scalar words of a (possibly vector-valued) storage element: uint4 -> 4 x uint32, ulonglong2 -> 2 x uint64, T -> T.
/// PackedStream, deinterleaved_iterator and the rank dictionary see every word iterator through this.
template <typename V> struct vec_comp { typedef V type; static const uint32 N = 1;
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE static V get(const V& v, uint32) { return v; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE static void put(V& v, uint32, const V x) { v = x; } };
#define NVBIO_HIP_VEC_COMP2(V, T) template <> struct vec_comp<V> { typedef T type; static const uint32 N = 2; \
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE static T get(const V& v, const uint32 k) { return k == 0u ? v.x : v.y; } \
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE static void put(V& v, const uint32 k, const T x) { if (k == 0u) v.x = x; else v.y = x; } };
#define NVBIO_HIP_VEC_COMP4(V, T) template <> struct vec_comp<V> { typedef T type; static const uint32 N = 4; \
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE static T get(const V& v, const uint32 k) { return k <= 1u ? (k == 0u ? v.x : v.y) : (k == 2u ? v.z : v.w); } \
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE static void put(V& v, const uint32 k, const T x) { if (k == 0u) v.x = x; else if (k == 1u) v.y = x; else if (k == 2u) v.z = x; else v.w = x; } };
NVBIO_HIP_VEC_COMP2(uint2, uint32) NVBIO_HIP_VEC_COMP4(uint4, uint32) NVBIO_HIP_VEC_COMP2(ulonglong2, uint64) NVBIO_HIP_VEC_COMP4(ulonglong4, uint64)
#undef NVBIO_HIP_VEC_COMP2
#undef NVBIO_HIP_VEC_COMP4
} // namespace priv

/// string_traits<Iterator>::value_type
template <typename T> struct string_traits { typedef typename T::value_type value_type; typedef uint32 index_type; };
template <typename T> struct string_traits<T*> { typedef T value_type; typedef uint32 index_type; };
template <typename T> struct string_traits<const T*> { typedef T value_type; typedef uint32 index_type; };

} // namespace nvbio
