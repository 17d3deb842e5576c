// nvbio_hip/types.h -- basic types of the C++ host layer (plain C++14, no HIP headers needed).
//
// The layer mirrors the reference's host-side interface for the hot path
// (nvbio::aln::batch_banded_alignment_score, nvbio::aln::BatchedBandedAlignmentScore,
// nvbio::fm_index, nvbio::FMIndexFilterDevice ...) on top of the C-ABI in nvbio_hip.h, so
// that a caller written against nvbio keeps its names and argument order.
#pragma once
#include <stdint.h>
#include <algorithm>
#include <stdexcept>
#include <string>
#include <utility>
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
/// the library this translation unit was compiled against: argument lists changed between ABI versions, so a wrapper built against one
/// header must not call a library of another (checked once, by the first driver object / device group / arena that is made)
inline void check_abi()
{
    static const int have = nvbio_hip_abi_version();
    if (have != NVBIO_HIP_ABI_VERSION) throw hip_error(("libnvbio_hip.so has ABI version " + std::to_string(have) + ", these headers are version " + std::to_string(NVBIO_HIP_ABI_VERSION) + ": nvbio_hip_abi_version()").c_str(), have);
}
/// a failed allocation says how much was asked for (an out-of-memory report without the size is useless to whoever sizes the batches)
inline void hip_check_alloc(int err, unsigned long long bytes) { if (err != 0) throw hip_error(("nvbio_hip_device_malloc(" + std::to_string(bytes) + " bytes)").c_str(), err); }

namespace hip {

/// A grow-only device arena for a driver's per-batch working set.  The reference's Aligner allocates its queues once in
/// init(); a driver written with scoped vectors would instead allocate and free gigabytes per batch (a 10 M-read batch needs
/// a 16 GB traceback buffer), and neither hipMalloc (~0.1 s per GB-sized block) nor a cache of freed blocks (which hands
/// the big block to the first smaller request of the next batch, or keeps gigabytes idle) makes that cheap.  Vectors built while an arena_scope is active take
/// their storage from the arena and never free it; reset() rewinds, and after the first batch the arena is one block.
struct device_arena
{
    device_arena() : used(0), taken(0) {}
    ~device_arena() { for (size_t i = 0; i < blocks.size(); ++i) nvbio_hip_device_free_ordered(blocks[i].first); }
    device_arena(const device_arena&) = delete;
    device_arena& operator=(const device_arena&) = delete;
    void* take(uint64 bytes)
    {
        bytes = (bytes + 255u) & ~uint64(255);
        taken += bytes;
        if (blocks.empty() || used + bytes > blocks.back().second)
        {
            // a new block: as large as everything so far (few blocks for a large first batch), but never more than 4 GiB beyond the request
            uint64 total = 0; for (size_t i = 0; i < blocks.size(); ++i) total += blocks[i].second;
            const uint64 step = std::min<uint64>(std::max<uint64>(total, uint64(64) << 20), uint64(4) << 30);
            const uint64 want = std::max<uint64>(bytes, step);
            void* p = nullptr;
            hip_check_alloc(nvbio_hip_device_malloc(&p, want), want);
            blocks.push_back(std::make_pair(p, want)); used = 0;
        }
        void* r = static_cast<uint8*>(blocks.back().first) + used;
        used += bytes;
        return r;
    }
    /// rewind; several blocks (the arena grew during the last batch) are replaced by ONE sized by what that batch took plus an eighth --
    /// not by the blocks' capacities, which would double the arena every time a batch needs a little more than the one before
    void reset()
    {
        if (blocks.size() > 1)
        {
            for (size_t i = 0; i < blocks.size(); ++i) nvbio_hip_device_free_ordered(blocks[i].first);
            blocks.clear();
            const uint64 size = ((taken + taken / 8u) + (uint64(64) << 20) - 1u) & ~((uint64(64) << 20) - 1u);
            void* p = nullptr;
            hip_check_alloc(nvbio_hip_device_malloc(&p, size), size);
            blocks.push_back(std::make_pair(p, size));
        }
        used = 0; taken = 0;
    }
    std::vector<std::pair<void*, uint64> > blocks;
    uint64 used, taken;             // bytes handed out of the last block / of all blocks since the last reset
};
inline device_arena*& current_arena() { static thread_local device_arena* a = nullptr; return a; }
struct arena_scope
{
    arena_scope(device_arena& a) : prev(current_arena()) { a.reset(); current_arena() = &a; }
    ~arena_scope() { current_arena() = prev; }
    device_arena* prev;
};

/// a minimal device vector (the reference's callers use thrust::device_vector here)
template <typename T>
struct device_vector {
    T*     m_ptr;
    size_t m_size;
    bool   m_in_arena;
    device_vector() : m_ptr(nullptr), m_size(0), m_in_arena(false) {}
    explicit device_vector(size_t n) : m_ptr(nullptr), m_size(0), m_in_arena(false) { resize(n); }
    device_vector(const std::vector<T>& h) : m_ptr(nullptr), m_size(0), m_in_arena(false) { assign(h.data(), h.size()); }
    device_vector(const device_vector&) = delete;
    device_vector& operator=(const device_vector&) = delete;
    ~device_vector() { if (m_ptr && !m_in_arena) nvbio_hip_device_free_ordered(m_ptr); }
    void resize(size_t n) {
        if (n == m_size) return;
        if (m_ptr && !m_in_arena) nvbio_hip_device_free_ordered(m_ptr);
        m_ptr = nullptr;
        if (current_arena() && (m_in_arena || m_size == 0)) { m_ptr = static_cast<T*>(current_arena()->take(uint64(n ? n : 1) * sizeof(T))); m_in_arena = true; m_size = n; return; }
        void* p = nullptr;
        hip_check_alloc(nvbio_hip_device_malloc(&p, uint64(n) * sizeof(T)), uint64(n) * sizeof(T));
        m_ptr = static_cast<T*>(p); m_size = n; m_in_arena = false;
    }
    /// (copies go through `stream`: a driver that shares its device with other host threads must not touch the NULL stream, which
    /// synchronises with every blocking stream of the device)
    void assign(const T* h, size_t n, void* stream = nullptr) {
        resize(n);
        hip_check(nvbio_hip_memcpy(m_ptr, h, uint64(n) * sizeof(T), 1, stream), "nvbio_hip_memcpy(h2d)");
    }
    std::vector<T> to_host(void* stream = nullptr) const {
        std::vector<T> h(m_size);
        hip_check(nvbio_hip_memcpy(h.data(), m_ptr, uint64(m_size) * sizeof(T), 2, stream), "nvbio_hip_memcpy(d2h)");
        return h;
    }
    T*       data()       { return m_ptr; }
    const T* data() const { return m_ptr; }
    size_t   size() const { return m_size; }
};

inline void synchronize(void* stream = nullptr) { hip_check(nvbio_hip_stream_synchronize(stream), "nvbio_hip_stream_synchronize"); }

/// A few words of pinned host memory a kernel writes through the same pointer: where a host thread waits for the handful of numbers that
/// decide what it queues next (the sizes of a selection round's queues) without a stream synchronisation and a copy per round.
/// wait_words(n): arm() set the words to a value no result takes; spin until the kernel has overwritten the first n -- looking at the
/// stream now and then, so that a failed launch ends in an exception rather than a spin.  If the memory cannot be had (ptr == nullptr) the
/// caller falls back to synchronise-and-copy.
struct pinned_words
{
    static const uint32 unset = 0xFFFFFFFFu;
    /// (a driver makes one of these per batch: the last one a thread gave up is kept for its next -- pinning and unpinning host memory are
    /// not cheap calls, and the second may wait for the device; the one block a thread keeps is never handed back)
    explicit pinned_words(const uint32 n) : ptr(nullptr), count(n)
    {
        spare_slot& sp = spare();
        if (sp.ptr && sp.count >= n) { ptr = sp.ptr; sp.ptr = nullptr; return; }
        void* p = nullptr;
        if (nvbio_hip_host_malloc(&p, uint64(n) * 4u) == 0) ptr = static_cast<volatile uint32*>(p);
    }
    ~pinned_words()
    {
        if (!ptr) return;
        spare_slot& sp = spare();
        if (sp.ptr == nullptr) { sp.ptr = ptr; sp.count = count; }
        else nvbio_hip_host_free(const_cast<uint32*>(ptr));
    }
    pinned_words(const pinned_words&) = delete;
    pinned_words& operator=(const pinned_words&) = delete;
    void arm() { for (uint32 i = 0; i < count; ++i) ptr[i] = unset; }
    void wait_words(const uint32 n, void* stream) const
    {
        for (uint64 spins = 1;; ++spins)
        {
            bool all = true;
            for (uint32 i = 0; i < n; ++i) all = all && (ptr[i] != unset);
            if (all) return;
            if ((spins & 0xFFFFu) == 0u)
            {
                const int q = nvbio_hip_stream_query(stream);
                if (q == 0)
                {
                    // the stream has drained: the words are there now, or the kernel never wrote them
                    for (uint32 i = 0; i < n; ++i) if (ptr[i] == unset) throw std::runtime_error("pinned_words: the stream drained without the awaited words being written");
                    return;
                }
                if (q != 600) hip_check(q, "nvbio_hip_stream_query");
            }
#if defined(__x86_64__) || defined(__i386__)
            __builtin_ia32_pause();
#endif
        }
    }
    volatile uint32* ptr;
    uint32           count;
private:
    struct spare_slot { volatile uint32* ptr; uint32 count; };
    static spare_slot& spare() { static thread_local spare_slot s = { nullptr, 0u }; return s; }
};

} // namespace hip
} // namespace nvbio
