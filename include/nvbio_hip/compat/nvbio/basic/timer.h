// compat/nvbio/basic/timer.h -- the wall-clock Timer the reference's tests and tools bracket their loops with
// (nvbio/basic/timer.h:52-120).  Header-only over std::chrono::steady_clock; not part of the hot path, here so that
// whole translation units of the reference (nvbio-test/rank_test.cu, ...) compile against `-I include/nvbio_hip/compat`.
#pragma once
#include "types.h"
#include "numbers.h"
#include <algorithm>
#include <chrono>
#include <deque>
#include <string>
#include <utility>

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

/// a running record of timed calls of one pipeline stage (timer.h:125-190): totals, a history of the last 10 000 (items, seconds)
/// pairs, per-log2(items) bins, and 32 user counters -- the fields nvBowtie's report generator reads directly
struct TimeSeries
{
    TimeSeries() : num(0), calls(0), time(0.0f), device_time(0.0f), max_speed(0.0f)
    {
        for (uint32 i = 0; i < 32; ++i)
        { bin_calls[i] = 0; bin_items[i] = 0; bin_time[i] = 0.0f; bin_speed[i] = 0.0f; user[i] = 0.0f; user_names[i] = NULL; user_units[i] = ""; user_avg[i] = false; }
    }
    /// one call that processed c items in t seconds (dt of them measured on the device)
    void add(const uint32 c, const float t, const float dt = 0.0f)
    {
        const float speed = float(c) / t;
        ++num; calls += c; time += t; device_time += dt;
        max_speed = std::max(max_speed, speed);
        if (info.size() == 10000) info.pop_front();
        info.push_back(std::make_pair(c, t));
        const uint32 bin = c ? nvbio::log2(c) : 0u;
        bin_calls[bin] += 1; bin_items[bin] += c; bin_time[bin] += t; bin_speed[bin] += speed;
    }
    float avg_speed() const { return float(calls) / time; }

    std::string name, units;
    uint32      num;
    uint64      calls;
    float       time, device_time, max_speed;
    uint32      bin_calls[32];
    uint64      bin_items[32];
    float       bin_time[32], bin_speed[32];
    std::deque< std::pair<uint32, float> > info;
    float       user[32];
    const char* user_names[32];
    const char* user_units[32];
    bool        user_avg[32];
};

} // namespace nvbio
