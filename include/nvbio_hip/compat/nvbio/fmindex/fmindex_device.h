// compat/nvbio/fmindex/fmindex_device.h -- fm_index_device_data<OCC_INT>: device copies of separately stored BWT words, occurrence
// counters and L2 (nvbio/fmindex/fmindex_device.h:37-72); the raw pointers are public, the object frees them.
#pragma once
#include "fmindex.h"
#if defined(__HIPCC__)

namespace nvbio {

template <uint32 OCC_INT>
struct fm_index_device_data
{
    fm_index_device_data(const uint32 len, const uint32* bwt, const uint32* occ, const uint32* L2) : m_bwt(NULL), m_occ(NULL), m_L2(NULL)
    {
        const size_t bwt_bytes = sizeof(uint32) * size_t((len + 16u) / 16u);
        const size_t occ_bytes = size_t(uint64(sizeof(uint32)) * 4u * uint64(len + OCC_INT - 1u) / OCC_INT);
        upload(&m_L2, L2, sizeof(uint32) * 5u); upload(&m_bwt, bwt, bwt_bytes); upload(&m_occ, occ, occ_bytes);
    }
    ~fm_index_device_data() { (void)hipFree(m_L2); (void)hipFree(m_bwt); (void)hipFree(m_occ); }
    fm_index_device_data(const fm_index_device_data&) = delete;
    fm_index_device_data& operator=(const fm_index_device_data&) = delete;

    uint32* m_bwt;
    uint32* m_occ;
    uint32* m_L2;
private:
    static void upload(uint32** dst, const uint32* src, const size_t bytes)
    { if (hipMalloc(reinterpret_cast<void**>(dst), bytes) == hipSuccess) (void)hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice); }
};

} // namespace nvbio
#endif
