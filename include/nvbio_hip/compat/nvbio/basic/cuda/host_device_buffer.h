// compat/nvbio/basic/cuda/host_device_buffer.h -- buffers visible from both sides (nvbio/basic/cuda/host_device_buffer.h:42-200):
// host_device_buffer_sync (a host and a device vector, to_host() copies), host_device_buffer_zero_copy (pinned host memory mapped
// into the device's address space), and cuda::copy / device_pointer over them and over thrust vectors.
#pragma once
#include "../types.h"
#include "../vector.h"
#include "../console.h"
#if defined(__HIPCC__)
#include <thrust/fill.h>
#include <thrust/copy.h>

namespace nvbio {
namespace cuda {

template <typename T>
struct host_device_buffer
{
    virtual ~host_device_buffer() {}
    virtual uint32   size() const { return 0u; }
    virtual void     resize(const uint32) {}
    virtual void     fill(const T) {}
    virtual void     to_host() {}
    virtual const T* host_ptr()   const { return NULL; }
    virtual const T* device_ptr() const { return NULL; }
    virtual T*       host_ptr()   { return NULL; }
    virtual T*       device_ptr() { return NULL; }
};

/// one allocation of pinned, device-mapped host memory: both pointers address the same bytes
template <typename T>
struct host_device_buffer_zero_copy : host_device_buffer<T>
{
    host_device_buffer_zero_copy() : m_host(NULL), m_size(0) {}
    host_device_buffer_zero_copy(const uint32 size) : m_host(NULL), m_size(0) { resize(size); }
    ~host_device_buffer_zero_copy() { release(); }
    uint32 size() const { return m_size; }
    void resize(const uint32 size)
    {
        T* fresh = NULL;
        if (size)
        {
            const hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&fresh), sizeof(T) * size_t(size), hipHostMallocMapped);
            if (e != hipSuccess) { log_error(stderr, "host_device_buffer_zero_copy::resize(): failed locking %llu bytes\n  %s\n", (unsigned long long)(sizeof(T) * size_t(size)), hipGetErrorString(e)); throw e; }
            for (uint32 i = 0; i < size && i < m_size; ++i) fresh[i] = m_host[i];
        }
        release();
        m_host = fresh; m_size = size;
    }
    void fill(const T val) { for (uint32 i = 0; i < m_size; ++i) m_host[i] = val; }
    void to_host() {}
    const T* host_ptr()   const { return m_host; }
    T*       host_ptr()         { return m_host; }
    const T* device_ptr() const { return mapped(); }
    T*       device_ptr()       { return mapped(); }
private:
    T* mapped() const
    {
        T* d = NULL;
        if (m_host) { const hipError_t e = hipHostGetDevicePointer(reinterpret_cast<void**>(&d), m_host, 0u);
                      if (e != hipSuccess) { log_error(stderr, "host_device_buffer_zero_copy::device_ptr(): failed mapping\n  %s\n", hipGetErrorString(e)); throw e; } }
        return d;
    }
    void release() { if (m_host) (void)hipHostFree(m_host); m_host = NULL; m_size = 0; }
    T*     m_host;
    uint32 m_size;
};

/// a host and a device copy; the device side is the one written, to_host() brings it back
template <typename T>
struct host_device_buffer_sync : host_device_buffer<T>
{
    host_device_buffer_sync() {}
    host_device_buffer_sync(const uint32 size) { resize(size); }
    uint32 size() const { return uint32(m_hvec.size()); }
    void resize(const uint32 size) { m_hvec.resize(size); m_dvec.resize(size); }
    void fill(const T val) { thrust::fill(m_dvec.begin(), m_dvec.end(), val); }
    void to_host() { m_hvec = m_dvec; }
    const T* host_ptr()   const { return nvbio::raw_pointer(m_hvec); }
    const T* device_ptr() const { return nvbio::raw_pointer(m_dvec); }
    T*       host_ptr()         { return nvbio::raw_pointer(m_hvec); }
    T*       device_ptr()       { return nvbio::raw_pointer(m_dvec); }
private:
    thrust::host_vector<T>   m_hvec;
    thrust::device_vector<T> m_dvec;
};

template <typename T> inline void copy(const thrust::device_vector<T>& dvec, thrust::host_vector<T>& hvec) { hvec = dvec; }
template <typename T> inline void copy(host_device_buffer<T>& dvec, thrust::host_vector<T>& hvec)
{
    dvec.to_host();
    hvec.resize(dvec.size());
    thrust::copy(dvec.host_ptr(), dvec.host_ptr() + dvec.size(), hvec.begin());
}
template <typename T> inline const T* device_pointer(const thrust::device_vector<T>& dvec) { return nvbio::raw_pointer(dvec); }
template <typename T> inline T*       device_pointer(thrust::device_vector<T>& dvec)       { return nvbio::raw_pointer(dvec); }
template <typename T> inline const T* device_pointer(const host_device_buffer<T>& dvec)    { return dvec.device_ptr(); }
template <typename T> inline T*       device_pointer(host_device_buffer<T>& dvec)          { return dvec.device_ptr(); }

} // namespace cuda
} // namespace nvbio
#endif
