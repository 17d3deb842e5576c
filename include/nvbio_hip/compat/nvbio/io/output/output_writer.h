// compat/nvbio/io/output/output_writer.h -- the file writes of an output file on a thread of their own (no counterpart in the reference,
// whose writers format and write inside OutputFile::process(), output_sam.cpp:441-470).
//
// process() is called by an aligner's compute thread between two batches of device work; what it needs from that thread is the
// formatting (it reads the batch and the reads, both of which the caller overwrites right after).  The write() calls that follow only need
// the text.  OrderedFileWriter takes the text of a batch and writes it to the file in the order the batches were handed over while the
// caller is back at its device; at most `depth` batches wait (one: a batch's write is far shorter than its alignment, and every further
// set of buffers is 300 MB of first-touch page faults), so the memory held is bounded and a slow disk still throttles the aligner.
// The text lives in TextBuffers -- plain growable runs of bytes a formatter asks for room in once per record and then writes through a
// pointer -- which the writer hands back when their bytes are in the file, so a batch is formatted into memory the previous batches
// already touched (a fresh 200 MB of std::string per batch spent more time in page faults than in formatting).
// NVBIO_HIP_SYNC_OUTPUT=1 turns it off (every write happens before process() returns); NVBIO_HIP_OUTPUT_TRACE=1 prints, at close, how long
// the callers formatted, waited for a free slot, and how long the writer thread spent in fwrite.
#pragma once
#include "../../basic/console.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace nvbio {
namespace io {
namespace priv {

/// a growable run of bytes: room(n) returns where the next n bytes go (valid until the next room()), commit(p) says how far they went
struct TextBuffer
{
    TextBuffer() : data(NULL), size(0), capacity(0) {}
    ~TextBuffer() { free(data); }
    TextBuffer(TextBuffer&& o) noexcept : data(o.data), size(o.size), capacity(o.capacity) { o.data = NULL; o.size = o.capacity = 0; }
    TextBuffer& operator=(TextBuffer&& o) noexcept { if (this != &o) { free(data); data = o.data; size = o.size; capacity = o.capacity; o.data = NULL; o.size = o.capacity = 0; } return *this; }
    TextBuffer(const TextBuffer&) = delete;
    TextBuffer& operator=(const TextBuffer&) = delete;

    void  clear() { size = 0; }
    void  reserve(const size_t n) { if (n > capacity) grow(n); }
    char* room(const size_t n) { if (size + n > capacity) grow(size + n); return data + size; }
    void  commit(const char* end) { size = size_t(end - data); }
    void  append(const char* s, const size_t n) { char* p = room(n); memcpy(p, s, n); size += n; }

    char*  data;
    size_t size, capacity;
private:
    void grow(const size_t need)
    {
        size_t cap = capacity ? capacity : size_t(1u << 16);
        while (cap < need) cap += cap / 2u;
        char* p = static_cast<char*>(realloc(data, cap));
        if (p == NULL) { log_error(stderr, "output: out of memory growing a text buffer to %llu bytes\n", (unsigned long long)cap); abort(); }
        data = p; capacity = cap;
    }
};

class OrderedFileWriter
{
public:
    typedef std::vector<TextBuffer> chunk_list;      ///< the text of one batch, in file order

    OrderedFileWriter() : m_fp(NULL), m_depth(1u), m_in_flight(0u), m_stop(false), m_failed(false), m_started(false),
                          m_batches(0u), m_bytes(0u), m_format_s(0.0), m_wait_s(0.0), m_write_s(0.0)
    {
        const char* s = getenv("NVBIO_HIP_SYNC_OUTPUT");  m_async = !(s && atoi(s) == 1);
        const char* t = getenv("NVBIO_HIP_OUTPUT_TRACE"); m_trace = (t && atoi(t) == 1);
    }
    ~OrderedFileWriter() { finish(); }

    /// the file every later push() goes to (the caller keeps ownership and closes it after finish())
    void open(FILE* fp) { m_fp = fp; }
    bool asynchronous() const { return m_async; }

    /// `n` empty buffers for the next batch: the ones an earlier batch was written from, where there are any
    chunk_list take(const size_t n)
    {
        chunk_list list;
        { std::unique_lock<std::mutex> hold(m_lock); if (!m_pool.empty()) { list.swap(m_pool.back()); m_pool.pop_back(); } }
        list.resize(n);
        for (size_t i = 0; i < n; ++i) list[i].clear();
        return list;
    }
    /// hand over the text of one batch; `format_seconds` is what the caller spent producing it (for the trace only)
    void push(chunk_list&& text, const double format_seconds = 0.0)
    {
        if (m_fp == NULL) return;
        if (!m_async) { m_format_s += format_seconds; ++m_batches; write_now(text); recycle(text); return; }
        const auto t0 = std::chrono::steady_clock::now();
        std::unique_lock<std::mutex> hold(m_lock);
        if (!m_started) { m_thread = std::thread(&OrderedFileWriter::run, this); m_started = true; }
        m_room.wait(hold, [this] { return m_in_flight < m_depth; });
        m_wait_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        m_format_s += format_seconds; ++m_batches;
        m_queue.emplace_back(std::move(text));
        ++m_in_flight;
        m_work.notify_one();
    }
    /// returns once everything handed over so far is in the FILE's buffer (callers that write to the file themselves -- a header, a
    /// trailer -- call this first)
    void drain()
    {
        if (!m_started) return;
        std::unique_lock<std::mutex> hold(m_lock);
        m_room.wait(hold, [this] { return m_in_flight == 0u; });
    }
    /// drain, stop the thread, report
    void finish()
    {
        if (m_started)
        {
            drain();
            { std::unique_lock<std::mutex> hold(m_lock); m_stop = true; m_work.notify_one(); }
            m_thread.join();
            m_started = false; m_stop = false;
        }
        if (m_trace && m_batches)
        {
            fprintf(stderr, "output  : %llu batches, %.1f MB: formatted in %.3f s (callers), %.3f s in fwrite (%s), callers waited %.3f s for a free slot\n",
                    (unsigned long long)m_batches, double(m_bytes) / 1.0e6, m_format_s, m_write_s, m_async ? "writer thread" : "callers", m_wait_s);
            m_batches = 0u;
        }
    }
    bool failed() const { return m_failed; }

private:
    void write_now(const chunk_list& text)
    {
        const auto t0 = std::chrono::steady_clock::now();
        for (size_t i = 0; i < text.size(); ++i)
        {
            if (text[i].size == 0u) continue;
            if (fwrite(text[i].data, 1, text[i].size, m_fp) != text[i].size && !m_failed)
            { m_failed = true; log_error(stderr, "output: a write to the alignment file failed (disk full?)\n"); }
            m_bytes += text[i].size;
        }
        m_write_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    // written buffers wait for the next batch (depth + 1 lists at most: one being filled while `depth` wait)
    void recycle_locked(chunk_list& text) { if (m_pool.size() <= m_depth) { m_pool.emplace_back(); m_pool.back().swap(text); } chunk_list().swap(text); }
    void recycle(chunk_list& text) { std::unique_lock<std::mutex> hold(m_lock); recycle_locked(text); }
    void run()
    {
        for (;;)
        {
            chunk_list text;
            {
                std::unique_lock<std::mutex> hold(m_lock);
                m_work.wait(hold, [this] { return m_stop || !m_queue.empty(); });
                if (m_queue.empty()) return;                       // m_stop, nothing left
                text.swap(m_queue.front()); m_queue.pop_front();
            }
            write_now(text);
            { std::unique_lock<std::mutex> hold(m_lock); recycle_locked(text); --m_in_flight; }
            m_room.notify_all();
        }
    }

    FILE*                   m_fp;
    bool                    m_async, m_trace;
    uint32_t                m_depth, m_in_flight;
    bool                    m_stop, m_failed, m_started;
    std::mutex              m_lock;
    std::condition_variable m_work, m_room;
    std::deque<chunk_list>  m_queue;
    std::vector<chunk_list> m_pool;
    std::thread             m_thread;
    unsigned long long      m_batches, m_bytes;
    double                  m_format_s, m_wait_s, m_write_s;
};

} // namespace priv
} // namespace io
} // namespace nvbio
