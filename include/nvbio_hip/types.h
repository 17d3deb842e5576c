// nvbio_hip/types.h -- basic types of the C++ host layer (plain C++14, no HIP headers needed).
//
// The layer mirrors the reference's host-side interface for the hot path
// (nvbio::aln::batch_banded_alignment_score, nvbio::aln::BatchedBandedAlignmentScore,
// nvbio::fm_index, nvbio::FMIndexFilterDevice ...) on top of the C-ABI in nvbio_hip.h, so
// that a caller written against nvbio keeps its names and argument order.
#pragma once
#include <stdint.h>
#include <stdexcept>
#include <string>
#include <vector>
#include "../nvbio_hip.h"

namespace nvbio {

typedef uint8_t  uint8;
typedef uint16_t uint16;
typedef int32_t  int32;
typedef uint32_t uint32;
typedef int64_t  int64;
typedef uint64_t uint64;

#if defined(__HIP__) || defined(HIP_INCLUDE_HIP_HIP_RUNTIME_H)
using ::uint2;
using ::uint4;
#else
struct uint2 { uint32 x, y; };
struct uint4 { uint32 x, y, z, w; };
inline uint2 make_uint2(uint32 x, uint32 y) { uint2 r = { x, y }; return r; }
inline uint4 make_uint4(uint32 x, uint32 y, uint32 z, uint32 w) { uint4 r = { x, y, z, w }; return r; }
#endif

struct device_tag {};
struct host_tag {};

/// the reference reports device failures as nvbio::cuda_error exceptions thrown from setup /
/// synchronisation points (nvbio/basic/exceptions.h); this is its counterpart
struct hip_error : public std::runtime_error {
    int code;
    hip_error(const char* what, int c) : std::runtime_error(std::string(what) + " failed, hipError " + std::to_string(c)), code(c) {}
};
inline void hip_check(int err, const char* what) { if (err != 0) throw hip_error(what, err); }

namespace hip {

/// a minimal device vector (the reference's callers use thrust::device_vector here)
template <typename T>
struct device_vector {
    T*     m_ptr;
    size_t m_size;
    device_vector() : m_ptr(nullptr), m_size(0) {}
    explicit device_vector(size_t n) : m_ptr(nullptr), m_size(0) { resize(n); }
    device_vector(const std::vector<T>& h) : m_ptr(nullptr), m_size(0) { assign(h.data(), h.size()); }
    device_vector(const device_vector&) = delete;
    device_vector& operator=(const device_vector&) = delete;
    ~device_vector() { if (m_ptr) nvbio_hip_device_free(m_ptr); }
    void resize(size_t n) {
        if (n == m_size) return;
        if (m_ptr) { nvbio_hip_device_free(m_ptr); m_ptr = nullptr; }
        void* p = nullptr;
        hip_check(nvbio_hip_device_malloc(&p, uint64(n) * sizeof(T)), "nvbio_hip_device_malloc");
        m_ptr = static_cast<T*>(p); m_size = n;
    }
    void assign(const T* h, size_t n) {
        resize(n);
        hip_check(nvbio_hip_memcpy(m_ptr, h, uint64(n) * sizeof(T), 1, nullptr), "nvbio_hip_memcpy(h2d)");
    }
    std::vector<T> to_host() const {
        std::vector<T> h(m_size);
        hip_check(nvbio_hip_memcpy(h.data(), m_ptr, uint64(m_size) * sizeof(T), 2, nullptr), "nvbio_hip_memcpy(d2h)");
        return h;
    }
    T*       data()       { return m_ptr; }
    const T* data() const { return m_ptr; }
    size_t   size() const { return m_size; }
};

inline void synchronize(void* stream = nullptr) { hip_check(nvbio_hip_stream_synchronize(stream), "nvbio_hip_stream_synchronize"); }

} // namespace hip
} // namespace nvbio
