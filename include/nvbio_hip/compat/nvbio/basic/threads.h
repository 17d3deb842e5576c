// compat/nvbio/basic/threads.h -- the small threading vocabulary the reference's applications are written on
// (nvbio/basic/threads.h:42-253): Thread<Derived> (create() runs Derived::run() on a new thread, join()), Mutex / ScopedLock,
// a locked WorkQueue, core counts, yield().  Header-only over <thread> / <mutex>.
#pragma once
#include "types.h"
#include "numbers.h"
#include "atomics.h"
#include "shared_pointer.h"
#include <memory>
#include <mutex>
#include <queue>
#include <thread>

namespace nvbio {

inline uint32 num_logical_cores()  { const unsigned n = std::thread::hardware_concurrency(); return n ? uint32(n) : 1u; }
inline uint32 num_physical_cores() { return num_logical_cores(); }      // (SMT siblings are not told apart here)

class ThreadBase
{
public:
    ThreadBase() : m_id(0) {}
    void   set_id(const uint32 id) { m_id = id; }
    uint32 get_id() const { return m_id; }
    void create(void* (*func)(void*), void* arg) { m_thread = std::make_shared<std::thread>([func, arg] { (void)func(arg); }); }
    void join() { if (m_thread && m_thread->joinable()) m_thread->join(); }
private:
    uint32                       m_id;
    std::shared_ptr<std::thread> m_thread;
};

/// CRTP thread: struct Worker : Thread<Worker> { void run(); };  worker.create(); ... worker.join();
template <typename DerivedThreadType>
class Thread : public ThreadBase
{
public:
    void create() { ThreadBase::create(&Thread::execute, static_cast<DerivedThreadType*>(this)); }
    void join()   { ThreadBase::join(); }
private:
    static void* execute(void* arg) { static_cast<DerivedThreadType*>(arg)->run(); return NULL; }
};

class Mutex
{
public:
    Mutex() : m_impl(std::make_shared<std::mutex>()) {}
    void lock()   { m_impl->lock(); }
    void unlock() { m_impl->unlock(); }
private:
    std::shared_ptr<std::mutex> m_impl;        // copies of a Mutex share the lock, as the reference's handle does
};
class ScopedLock
{
public:
     ScopedLock(Mutex* mutex) : m_mutex(mutex) { m_mutex->lock(); }
    ~ScopedLock() { m_mutex->unlock(); }
private:
    Mutex* m_mutex;
};

/// a queue of work items popped under a lock; the callback is told (items handed out so far - 1, items pushed)
template <typename WorkItemT, typename ProgressCallbackT>
class WorkQueue
{
public:
    typedef WorkItemT          WorkItem;
    typedef ProgressCallbackT  ProgressCallback;
    WorkQueue() : m_callback(), m_size(0u) {}
    void push(const WorkItem work)        { m_queue.push(work); ++m_size; }
    void locked_push(const WorkItem work) { ScopedLock hold(&m_lock); m_queue.push(work); ++m_size; }
    bool pop(WorkItem& work)
    {
        ScopedLock hold(&m_lock);
        if (m_queue.empty()) return false;
        work = m_queue.front(); m_queue.pop();
        m_callback(m_size - uint32(m_queue.size()) - 1u, m_size);
        return true;
    }
    void set_callback(const ProgressCallback callback) { m_callback = callback; }
private:
    ProgressCallback     m_callback;
    std::queue<WorkItem> m_queue;
    Mutex                m_lock;
    uint32               m_size;
};

/// the batch size that spreads total_count items over whole rounds of thread_count batches of about batch_size
inline uint32 balance_batch_size(const uint32 batch_size, const uint32 total_count, const uint32 thread_count)
{
    const uint32 batches = (total_count + batch_size - 1u) / batch_size;
    const uint32 rounds  = (batches + thread_count - 1u) / thread_count;
    const uint32 even    = rounds * thread_count;
    return (total_count + even - 1u) / even;
}
inline void yield() { std::this_thread::yield(); }

} // namespace nvbio
