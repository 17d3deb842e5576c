// compat/nvbio/basic/vector_array.h -- an array of variable-length vectors carved out of one arena by an atomic bump pointer
// (nvbio/basic/vector_array.h:44-420): VectorArrayView (device-callable alloc / lookup), DeviceVectorArray / HostVectorArray (storage).
// nvBowtie keeps the CIGARs and MD strings of a batch in these.
#pragma once
#include "types.h"
#include "thrust_view.h"
#include "atomics.h"
#include "vector.h"
#if defined(__HIPCC__)
#include <thrust/fill.h>
#endif

namespace nvbio {

template <typename T>
struct VectorArrayView
{
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE
    VectorArrayView(T* arena = NULL, uint32* index = NULL, uint32* sizes = NULL, uint32* pool = NULL, uint32 size = 0u)
        : m_arena(arena), m_index(index), m_sizes(sizes), m_pool(pool), m_size(size) {}

    /// reserve `size` entries for vector `index`; NULL (and slot = arena size) when the arena is exhausted
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE T* alloc(const uint32 index, const uint32 size)
    {
        const uint32 slot = atomic_add(m_pool, size);
        const bool fits = slot + size < m_size;
        m_index[index] = fits ? slot : m_size;
        m_sizes[index] = fits ? size : 0u;
        return fits ? m_arena + slot : (T*)NULL;
    }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE T*     operator[](const uint32 index) const { return m_index[index] < m_size ? m_arena + m_index[index] : (T*)NULL; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 slot(const uint32 index) const { return m_index[index]; }
    NVBIO_FORCEINLINE NVBIO_HOST_DEVICE uint32 size(const uint32 index) const { return m_sizes[index]; }

    T*      m_arena;
    uint32* m_index;
    uint32* m_sizes;
    uint32* m_pool;
    uint32  m_size;
};

#if defined(__HIPCC__)
namespace priv {
/// the storage both flavours share, over thrust vectors of one memory space
template <typename T, template <typename, typename...> class Vec>
struct vector_array_storage
{
    vector_array_storage() : m_pool(1, 0u) {}

    /// n vectors over an arena of `arena` entries; returns the bytes needed (allocates unless do_alloc is false)
    uint64 resize(const uint32 size, const uint32 arena, const bool do_alloc = true)
    {
        if (do_alloc)
        {
            m_arena.resize(arena); m_index.resize(size); m_sizes.resize(size);
            thrust::fill(m_index.begin(), m_index.begin() + size, arena);          // "not allocated"
            thrust::fill(m_sizes.begin(), m_sizes.begin() + size, uint32(0));
        }
        return uint64(sizeof(T)) * arena + uint64(sizeof(uint32)) * size * 2u;
    }
    bool   has_overflown()        { return uint32(m_pool[0]) > m_arena.size(); }
    void   clear()                { m_pool[0] = 0u; }
    uint32 size() const           { return uint32(m_index.size()); }
    uint32 allocated_size() const { return uint32(m_pool[0]); }
    uint32 arena_size() const     { return uint32(m_arena.size()); }

    template <typename Other> void copy_from(const Other& o) { m_arena = o.m_arena; m_index = o.m_index; m_sizes = o.m_sizes; m_pool = o.m_pool; }
    template <typename Other> void swap_with(Other& o) { m_arena.swap(o.m_arena); m_index.swap(o.m_index); m_sizes.swap(o.m_sizes); m_pool.swap(o.m_pool); }
    VectorArrayView<T> view() { return VectorArrayView<T>(nvbio::raw_pointer(m_arena), nvbio::raw_pointer(m_index), nvbio::raw_pointer(m_sizes), nvbio::raw_pointer(m_pool), uint32(m_arena.size())); }

    Vec<T>      m_arena;
    Vec<uint32> m_index;
    Vec<uint32> m_sizes;
    Vec<uint32> m_pool;
};
} // namespace priv

template <typename T>
struct DeviceVectorArray : public priv::vector_array_storage<T, thrust::device_vector>
{
    typedef device_tag          system_tag;
    typedef VectorArrayView<T>  device_view_type;
    typedef VectorArrayView<T>  plain_view_type;
    DeviceVectorArray& operator=(const DeviceVectorArray<T>& vec) { this->copy_from(vec); return *this; }
    DeviceVectorArray& swap(DeviceVectorArray<T>& vec) { this->swap_with(vec); return *this; }
    device_view_type device_view() { return this->view(); }
    plain_view_type  plain_view()  { return this->view(); }
};
template <typename T>
struct HostVectorArray : public priv::vector_array_storage<T, thrust::host_vector>
{
    typedef host_tag            system_tag;
    typedef VectorArrayView<T>  plain_view_type;
    HostVectorArray& operator=(const DeviceVectorArray<T>& vec) { this->copy_from(vec); return *this; }
    HostVectorArray& operator=(const HostVectorArray<T>& vec)   { this->copy_from(vec); return *this; }
    HostVectorArray& swap(HostVectorArray<T>& vec) { this->swap_with(vec); return *this; }
    const T* operator[](const uint32 index) const { return this->m_index[index] < this->m_arena.size() ? &this->m_arena[0] + this->m_index[index] : (const T*)NULL; }
    uint32   slot(const uint32 index) const { return this->m_index[index]; }
    plain_view_type plain_view() { return this->view(); }
};
template <typename T> inline VectorArrayView<T> device_view(DeviceVectorArray<T>& vec) { return vec.device_view(); }
template <typename T> inline VectorArrayView<T> plain_view(DeviceVectorArray<T>& vec)  { return vec.device_view(); }
template <typename T> inline VectorArrayView<T> plain_view(HostVectorArray<T>& vec)    { return vec.plain_view(); }
#endif

} // namespace nvbio
