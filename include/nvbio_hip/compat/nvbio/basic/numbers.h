// compat/nvbio/basic/numbers.h -- the small integer / vector-component helpers the hot-path callers use
// (nvbio/basic/numbers.h:60-300, 1300-1370): component access on the HIP vector types (comp / set / select, what
// nvBowtie's 1-mismatch map<> reads the rank4 counters with, mapping_inl.h:178-205), util::count_occurrences (the
// N filter of the seed mappers, mapping_inl.h:258,346), rounding divisions, bit masks, and the complement / reverse /
// cast functors the seed readers are built from.  Field_traits, min / max, divide_ri on uint32 live in types.h.
#pragma once
#include "types.h"

namespace nvbio {

#if defined(__HIPCC__)
/// lane and warp of a thread in the reference's 32-lane terms (numbers.h:64-65; see cuda/arch.h on virtual warps)
/// warp_tid() converts to uint32 everywhere, with one extra: `n - warp_tid()` keeps its type, and a 32-bit mask shifted by such a difference
/// follows CUDA's rule -- a shift by 32 or more gives 0 -- where C++ leaves it undefined and gfx950 would shift by (n & 31).  nvBowtie's
/// warp-aggregated queue allocation elects its leader with `__popc( mask << (32u - warp_tid()) ) == 0` (nvBowtie/bowtie2/cuda/utils.h:62):
/// lane 0 shifts by 32 and must see 0.
struct warp_lane  { uint32 v; NVBIO_FORCEINLINE NVBIO_HOST_DEVICE operator uint32() const { return v; } };
struct warp_shift { uint32 v; NVBIO_FORCEINLINE NVBIO_HOST_DEVICE operator uint32() const { return v; } };
template <typename T> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE warp_shift operator-(const T a, const warp_lane b) { const warp_shift s = { uint32(a) - b.v }; return s; }
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 operator<<(const uint32 m, const warp_shift s) { return s.v >= 32u ? 0u : (m << s.v); }
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 operator>>(const uint32 m, const warp_shift s) { return s.v >= 32u ? 0u : (m >> s.v); }
NVBIO_FORCEINLINE __device__ warp_lane warp_tid() { const warp_lane l = { threadIdx.x & 31u }; return l; }
NVBIO_FORCEINLINE __device__ uint32    warp_id()  { return threadIdx.x >> 5; }
#endif

namespace util {

template <uint32 N> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 lo_bits() { return N >= 32u ? 0xFFFFFFFFu : (1u << (N & 31u)) - 1u; }
template <uint32 N> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 hi_bits() { return ~lo_bits<N>(); }

/// how many of begin[0, size) equal val, giving up once max_occ have been seen
template <typename Iterator, typename T> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE
uint32 count_occurrences(const Iterator begin, uint32 size, const T val, const uint32 max_occ = uint32(-1))
{
    uint32 n = 0;
    for (uint32 i = 0; i < size && n < max_occ; ++i) n += (begin[i] == val) ? 1u : 0u;
    return n;
}

/// x / y rounded towards +inf, towards zero, and to the nearest integer (ties away from the floor)
template <typename L, typename R> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE L divide_ri(const L x, const R y) { return L((x + (y - 1)) / y); }
template <typename L, typename R> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE L divide_rz(const L x, const R y) { return L(x / y); }
template <typename L, typename R> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE L round_i(const L x, const R y) { return L(y * divide_ri(x, y)); }
template <typename L, typename R> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE L round_z(const L x, const R y) { return L(y * divide_rz(x, y)); }
template <typename L, typename R> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE L round(const L x, const R y)
{
    const L lo = round_z(x, y);
    return R((x - lo) * 2) > y ? L(lo + L(1)) : lo;
}

} // namespace util

#if defined(__HIPCC__)
/// the c-th component of a HIP vector by value, and writers -- one definition per (arity, element) pair
#define NVBIO_HIP_COMP2(V, T, C) \
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE T comp(const V a, const C c) { return c == 0 ? T(a.x) : T(a.y); }
#define NVBIO_HIP_COMP4(V, T, C) \
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE T comp(const V a, const C c) { return c < 2 ? (c == 0 ? T(a.x) : T(a.y)) : (c == 2 ? T(a.z) : T(a.w)); }
#define NVBIO_HIP_SET2(V, T) \
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE void set(V& a, const uint32 c, const T v) { if (c == 0) a.x = v; else a.y = v; }
#define NVBIO_HIP_SET4(V, T) \
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE void set(V& a, const uint32 c, const T v) { if (c == 0) a.x = v; else if (c == 1) a.y = v; else if (c == 2) a.z = v; else a.w = v; }
NVBIO_HIP_COMP2(uchar2, uint8, char)      NVBIO_HIP_COMP2(char2, char, char)
NVBIO_HIP_COMP4(uchar4, uint8, char)      NVBIO_HIP_COMP4(char4, char, char)
NVBIO_HIP_COMP2(uint2, uint32, uint32)    NVBIO_HIP_COMP2(int2, int32, uint32)    NVBIO_HIP_COMP2(ulonglong2, uint64, uint32)
NVBIO_HIP_COMP4(uint4, uint32, uint32)    NVBIO_HIP_COMP4(int4, int32, uint32)    NVBIO_HIP_COMP4(ulonglong4, uint64, uint32)
NVBIO_HIP_COMP4(ushort4, uint16, uint32)
NVBIO_HIP_SET2(uint2, uint32)             NVBIO_HIP_SET2(ulonglong2, uint64)
NVBIO_HIP_SET4(uint4, uint32)             NVBIO_HIP_SET4(ulonglong4, uint64)
#undef NVBIO_HIP_COMP2
#undef NVBIO_HIP_COMP4
#undef NVBIO_HIP_SET2
#undef NVBIO_HIP_SET4
/// a reference to the c-th component
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32& select(uint4& a, const uint32 c) { return c < 2 ? (c == 0 ? a.x : a.y) : (c == 2 ? a.z : a.w); }
#else
NVBIO_FORCEINLINE uint32 comp(const uint2 a, const uint32 c) { return c == 0 ? a.x : a.y; }
NVBIO_FORCEINLINE uint32 comp(const uint4 a, const uint32 c) { return c < 2 ? (c == 0 ? a.x : a.y) : (c == 2 ? a.z : a.w); }
NVBIO_FORCEINLINE uint64 comp(const ulonglong2 a, const uint32 c) { return c == 0 ? a.x : a.y; }
NVBIO_FORCEINLINE uint64 comp(const ulonglong4 a, const uint32 c) { return c < 2 ? (c == 0 ? a.x : a.y) : (c == 2 ? a.z : a.w); }
NVBIO_FORCEINLINE void set(uint2& a, const uint32 c, const uint32 v) { (c == 0 ? a.x : a.y) = v; }
NVBIO_FORCEINLINE void set(uint4& a, const uint32 c, const uint32 v) { (c < 2 ? (c == 0 ? a.x : a.y) : (c == 2 ? a.z : a.w)) = v; }
NVBIO_FORCEINLINE uint32& select(uint4& a, const uint32 c) { return c < 2 ? (c == 0 ? a.x : a.y) : (c == 2 ? a.z : a.w); }
#endif

/// floor(log2(n)) for n > 0, and power-of-two tests
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 log2(const uint32 n) { return n ? 31u - uint32(__builtin_clz(n)) : 0u; }
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bool   is_pow2(const uint32 n) { return (n & (n - 1u)) == 0u; }
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 next_power_of_two(const uint32 n) { return n <= 1u ? 1u : 1u << (32u - uint32(__builtin_clz(n - 1u))); }

/// the linear congruential generator the reference's tests draw their synthetic strings from (numbers.h:610-625; the
/// Numerical Recipes constants)
struct LCG_random
{
    static const uint32 MAX = 0xFFFFFFFFu;
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE LCG_random(const uint32 s = 0) : m_s(s) {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 next() { m_s = m_s * 1664525u + 1013904223u; return m_s; }
    uint32 m_s;
};

/// index -> len - 1 - index
template <typename IndexType = uint32> struct reverse_functor
{
    typedef IndexType index_type; typedef IndexType argument_type; typedef IndexType result_type;
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE reverse_functor() : m_len(0) {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE reverse_functor(const index_type len) : m_len(len) {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE index_type operator()(const index_type i) const { return m_len - i - 1; }
    index_type m_len;
};
/// symbol -> its complement inside an alphabet of ALPHABET_SIZE symbols; anything else (N) is left alone
template <uint32 ALPHABET_SIZE> struct complement_functor
{
    typedef uint8 argument_type; typedef uint8 result_type;
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE complement_functor() {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint8 operator()(const uint8 c) const { return c < ALPHABET_SIZE ? uint8(ALPHABET_SIZE - 1u - c) : c; }
};
template <typename T, typename R> struct cast_functor
{
    typedef T argument_type; typedef R result_type;
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE R operator()(const T v) const { return R(v); }
};

/// the top bits of a 32-bit word as a narrower key (numbers.h:1064-1103): what nvBowtie sorts its hits by
template <typename T, typename U> struct hi_bits_functor {};
template <> struct hi_bits_functor<uint8, uint32>  { typedef uint32 argument_type; typedef uint8  result_type; NVBIO_FORCEINLINE NVBIO_HOST_DEVICE result_type operator()(const argument_type op) const { return result_type(op >> 24u); } };
template <> struct hi_bits_functor<uint16, uint32> { typedef uint32 argument_type; typedef uint16 result_type; NVBIO_FORCEINLINE NVBIO_HOST_DEVICE result_type operator()(const argument_type op) const { return result_type(op >> 16u); } };
template <> struct hi_bits_functor<uint32, uint32> { typedef uint32 argument_type; typedef uint32 result_type; NVBIO_FORCEINLINE NVBIO_HOST_DEVICE result_type operator()(const argument_type op) const { return op; } };

/// the small functor vocabulary handed to transform / reduce / copy_if (numbers.h:920-1380)
struct add_functor { template <typename T> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE T operator()(const T a, const T b) const { return a + b; } };
struct min_functor { template <typename T> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE T operator()(const T a, const T b) const { return a < b ? a : b; } };
struct max_functor { template <typename T> NVBIO_FORCEINLINE NVBIO_HOST_DEVICE T operator()(const T a, const T b) const { return a > b ? a : b; } };
#define NVBIO_HIP_UNARY_TEST(name, expr)                                                                                     \
    template <typename T> struct name { typedef T argument_type; typedef bool result_type;                                    \
        NVBIO_FORCEINLINE NVBIO_HOST_DEVICE result_type operator()(const T op) const { return expr; } };
NVBIO_HIP_UNARY_TEST(is_true_functor,  op ? true : false)
NVBIO_HIP_UNARY_TEST(is_false_functor, op ? false : true)
#undef NVBIO_HIP_UNARY_TEST
#define NVBIO_HIP_BOUND_TEST(name, cmp)                                                                                      \
    template <typename T> struct name { typedef T argument_type; typedef bool result_type;                                    \
        NVBIO_FORCEINLINE NVBIO_HOST_DEVICE name(const T k) : m_k(k) {}                                                       \
        NVBIO_FORCEINLINE NVBIO_HOST_DEVICE result_type operator()(const T op) const { return op cmp m_k; }                   \
        const T m_k; };
NVBIO_HIP_BOUND_TEST(equal_to_functor, ==)
NVBIO_HIP_BOUND_TEST(not_equal_to_functor, !=)
#undef NVBIO_HIP_BOUND_TEST
#define NVBIO_HIP_BINARY_TEST(name, cmp)                                                                                     \
    template <typename T> struct name { typedef T first_argument_type; typedef T second_argument_type; typedef bool result_type; \
        NVBIO_FORCEINLINE NVBIO_HOST_DEVICE result_type operator()(const T a, const T b) const { return a cmp b; } };
NVBIO_HIP_BINARY_TEST(equal_functor, ==)
NVBIO_HIP_BINARY_TEST(not_equal_functor, !=)
#undef NVBIO_HIP_BINARY_TEST
/// op -> perm[op]
template <typename Iterator, typename index_type = uint32>
struct gather_functor
{
    typedef index_type argument_type; typedef typename std::iterator_traits<Iterator>::value_type result_type;
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE gather_functor(const Iterator perm) : m_perm(perm) {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE result_type operator()(const argument_type op) const { return m_perm[op]; }
    Iterator m_perm;
};
template <typename Iterator> inline gather_functor<Iterator> make_gather_functor(const Iterator perm) { return gather_functor<Iterator>(perm); }
/// op -> fun2(fun1(op))
template <typename Functor2, typename Functor1>
struct composition_functor
{
    typedef typename Functor1::argument_type argument_type; typedef typename Functor2::result_type result_type;
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE composition_functor(const Functor2 fun2, const Functor1 fun1) : m_fun1(fun1), m_fun2(fun2) {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE result_type operator()(const argument_type op) const { return m_fun2(m_fun1(op)); }
    Functor1 m_fun1; Functor2 m_fun2;
};
template <typename Functor2, typename Functor1>
inline composition_functor<Functor2, Functor1> make_composition_functor(const Functor2 fun2, const Functor1 fun1) { return composition_functor<Functor2, Functor1>(fun2, fun1); }
/// v -> component c of the vector v (x, y, z, w = 0 .. 3)
template <typename T>
struct component_functor
{
    typedef T argument_type; typedef typename priv::vec_comp<T>::type result_type;
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE component_functor(const uint32 c) : m_c(c) {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE result_type operator()(const argument_type op) const { return priv::vec_comp<T>::get(op, m_c); }
    uint32 m_c;
};
/// a binary functor with its first / second argument fixed
template <typename Functor>
struct bind_first_functor
{
    typedef typename Functor::second_argument_type argument_type; typedef typename Functor::first_argument_type const_type; typedef typename Functor::result_type result_type;
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bind_first_functor(const const_type c) : m_fun(), m_c(c) {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bind_first_functor(const Functor fun, const const_type c) : m_fun(fun), m_c(c) {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE result_type operator()(const argument_type op) const { return m_fun(m_c, op); }
    Functor m_fun; const_type m_c;
};
template <typename Functor>
struct bind_second_functor
{
    typedef typename Functor::first_argument_type argument_type; typedef typename Functor::second_argument_type const_type; typedef typename Functor::result_type result_type;
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bind_second_functor(const const_type c) : m_fun(), m_c(c) {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE bind_second_functor(const Functor fun, const const_type c) : m_fun(fun), m_c(c) {}
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE result_type operator()(const argument_type op) const { return m_fun(op, m_c); }
    Functor m_fun; const_type m_c;
};
template <typename T> struct negate_functor { typedef T argument_type; typedef T result_type; NVBIO_FORCEINLINE NVBIO_HOST_DEVICE T operator()(const T op) const { return -op; } };

} // namespace nvbio
