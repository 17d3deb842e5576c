// compat/nvbio/basic/atomics.h -- host-side atomics and fences, and the host/device atomic_add family (nvbio/basic/atomics.h:58-462).
// On the host: GCC/Clang __atomic builtins (sequentially consistent); in kernels: the HIP atomics.
#pragma once
#include "types.h"

namespace nvbio {

inline void host_release_fence() { __atomic_thread_fence(__ATOMIC_RELEASE); }
inline void host_acquire_fence() { __atomic_thread_fence(__ATOMIC_ACQUIRE); }

#define NVBIO_HIP_HOST_ATOMIC(T)                                                                                        \
    inline T host_atomic_add(T* value, const T op) { return __atomic_fetch_add(value, op, __ATOMIC_SEQ_CST); }          \
    inline T host_atomic_sub(T* value, const T op) { return __atomic_fetch_sub(value, op, __ATOMIC_SEQ_CST); }
NVBIO_HIP_HOST_ATOMIC(int32) NVBIO_HIP_HOST_ATOMIC(uint32) NVBIO_HIP_HOST_ATOMIC(int64) NVBIO_HIP_HOST_ATOMIC(uint64)
#undef NVBIO_HIP_HOST_ATOMIC
inline uint32 host_atomic_or(uint32* value, const uint32 op) { return __atomic_fetch_or(value, op, __ATOMIC_SEQ_CST); }
inline uint64 host_atomic_or(uint64* value, const uint64 op) { return __atomic_fetch_or(value, op, __ATOMIC_SEQ_CST); }

// the same operation from either side: device atomics in a kernel, host atomics elsewhere; all return the value found
#if defined(NVBIO_DEVICE_COMPILATION)
#define NVBIO_HIP_ATOMIC_BODY(dev, host) return dev;
#else
#define NVBIO_HIP_ATOMIC_BODY(dev, host) return host;
#endif
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE int32  atomic_add(int32* value, const int32 op)   { NVBIO_HIP_ATOMIC_BODY(atomicAdd(value, op), host_atomic_add(value, op)) }
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 atomic_add(uint32* value, const uint32 op) { NVBIO_HIP_ATOMIC_BODY(atomicAdd(value, op), host_atomic_add(value, op)) }
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint64 atomic_add(uint64* value, const uint64 op) { NVBIO_HIP_ATOMIC_BODY(uint64(atomicAdd(reinterpret_cast<unsigned long long*>(value), (unsigned long long)op)), host_atomic_add(value, op)) }
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE int32  atomic_sub(int32* value, const int32 op)   { NVBIO_HIP_ATOMIC_BODY(atomicSub(value, op), host_atomic_sub(value, op)) }
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 atomic_sub(uint32* value, const uint32 op) { NVBIO_HIP_ATOMIC_BODY(atomicSub(value, op), host_atomic_sub(value, op)) }
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 atomic_or(uint32* value, const uint32 op)  { NVBIO_HIP_ATOMIC_BODY(atomicOr(value, op), host_atomic_or(value, op)) }
NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint64 atomic_or(uint64* value, const uint64 op)  { NVBIO_HIP_ATOMIC_BODY(uint64(atomicOr(reinterpret_cast<unsigned long long*>(value), (unsigned long long)op)), host_atomic_or(value, op)) }
#undef NVBIO_HIP_ATOMIC_BODY

inline int32 atomic_increment(int32 volatile* value) { return __atomic_add_fetch(value, 1, __ATOMIC_SEQ_CST); }
inline int64 atomic_increment(int64 volatile* value) { return __atomic_add_fetch(value, int64(1), __ATOMIC_SEQ_CST); }
inline int32 atomic_decrement(int32 volatile* value) { return __atomic_sub_fetch(value, 1, __ATOMIC_SEQ_CST); }
inline int64 atomic_decrement(int64 volatile* value) { return __atomic_sub_fetch(value, int64(1), __ATOMIC_SEQ_CST); }

/// an integer with atomic ++ / -- / += / -= (the counter SharedPointer takes)
template <typename intT>
struct AtomicInt
{
    AtomicInt() : m_value(0) {}
    AtomicInt(const intT value) : m_value(value) {}
    intT operator++(int) { return __atomic_fetch_add(&m_value, intT(1), __ATOMIC_SEQ_CST); }
    intT operator--(int) { return __atomic_fetch_sub(&m_value, intT(1), __ATOMIC_SEQ_CST); }
    intT operator++()    { return __atomic_add_fetch(&m_value, intT(1), __ATOMIC_SEQ_CST); }
    intT operator--()    { return __atomic_sub_fetch(&m_value, intT(1), __ATOMIC_SEQ_CST); }
    intT operator+=(const intT v) { return __atomic_add_fetch(&m_value, v, __ATOMIC_SEQ_CST); }
    intT operator-=(const intT v) { return __atomic_sub_fetch(&m_value, v, __ATOMIC_SEQ_CST); }
    operator intT() const { return __atomic_load_n(&m_value, __ATOMIC_SEQ_CST); }
    bool operator==(const intT v) const { return intT(*this) == v; }
    bool operator!=(const intT v) const { return intT(*this) != v; }
    volatile intT m_value;
};
typedef AtomicInt<int32> AtomicInt32;
typedef AtomicInt<int64> AtomicInt64;

} // namespace nvbio
