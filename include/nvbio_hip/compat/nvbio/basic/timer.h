// compat/nvbio/basic/timer.h -- the wall-clock Timer the reference's tests and tools bracket their loops with
// (nvbio/basic/timer.h:52-120).  Header-only over std::chrono::steady_clock; not part of the hot path, here so that
// whole translation units of the reference (nvbio-test/rank_test.cu, ...) compile against `-I include/nvbio_hip/compat`.
#pragma once
#include "types.h"
#include <chrono>

namespace nvbio {

struct Timer
{
    Timer() : m_elapsed(0.0) {}
    void  start() { m_t0 = std::chrono::steady_clock::now(); }
    void  stop()  { m_elapsed = std::chrono::duration<double>(std::chrono::steady_clock::now() - m_t0).count(); }
    float seconds() const { return float(m_elapsed); }
private:
    std::chrono::steady_clock::time_point m_t0;
    double                                m_elapsed;
};

/// adds the lifetime of the object to *accumulator
template <typename T>
struct ScopedTimer
{
    ScopedTimer(T* accumulator) : m_acc(accumulator) { m_timer.start(); }
    ~ScopedTimer() { m_timer.stop(); *m_acc += T(m_timer.seconds()); }
private:
    T*    m_acc;
    Timer m_timer;
};

struct FakeTimer
{
    void  start() {}
    void  stop() {}
    float seconds() const { return 0.0f; }
};

} // namespace nvbio
