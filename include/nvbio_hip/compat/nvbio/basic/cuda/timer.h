// compat/nvbio/basic/cuda/timer.h -- cuda::Timer: seconds between two points of the default stream, by device events
// (nvbio/basic/cuda/timer.h:38-119), and ScopedTimer<T> which adds the elapsed time to *time on destruction.
#pragma once
#include "../types.h"
#if defined(__HIPCC__)

namespace nvbio {
namespace cuda {

struct Timer
{
    Timer()  { (void)hipEventCreate(&m_start); (void)hipEventCreate(&m_stop); }
    ~Timer() { (void)hipEventDestroy(m_start); (void)hipEventDestroy(m_stop); }
    Timer(const Timer&) = delete;
    Timer& operator=(const Timer&) = delete;
    void  start() { (void)hipEventRecord(m_start, 0); }
    void  stop()  { (void)hipEventRecord(m_stop, 0); (void)hipEventSynchronize(m_stop); }
    float seconds() const { float ms = 0.0f; (void)hipEventElapsedTime(&ms, m_start, m_stop); return ms * 1.0e-3f; }
    hipEvent_t m_start, m_stop;
};

template <typename T>
struct ScopedTimer
{
     ScopedTimer(T* time) : m_time(time), m_timer() { m_timer.start(); }
    ~ScopedTimer() { m_timer.stop(); *m_time += m_timer.seconds(); }
    T*    m_time;
    Timer m_timer;
};

} // namespace cuda
} // namespace nvbio
#endif
