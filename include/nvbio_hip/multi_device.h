// nvbio_hip/multi_device.h -- the multi-GPU host layer in C++ (north star: "host code stays C++").
//
// The reference's multi-GPU mode is one host thread per device over a replicated index, each thread pulling batches of reads and
// writing into a shared output (nvBowtie/nvBowtie.cpp:809-864, bowtie2/cuda/compute_thread.cu:74-117).  Here:
//   * shard_range()            contiguous block sharding of a read batch: rank g owns [g * ceil(R / G), min(R, (g + 1) * ceil(R / G)))
//   * DeviceGroup              the ranks of a run and their RCCL communicators, in either of two shapes:
//                                - DeviceGroup::local(n): ONE process, one host thread per device (run() spawns them, each bound to its
//                                  device with nvbio_hip_set_device) -- the reference's shape;
//                                - DeviceGroup::from_unique_id(rank, world, id): one process per device under a launcher (torchrun, mpirun),
//                                  which ships rank 0's nvbio_hip_comm_unique_id() bytes to every rank;
//   * DeviceGroup::Rank::gather_records()   the path's only collective: every rank's fixed-size records (16 B per read: io::Alignment word,
//                                  position, MAPQ, read id; 32 B per pair: io::BestPairedAlignments) to the root, in rank order, over
//                                  RCCL ncclSend / ncclRecv (nvbio_hip_gather_records).
// Everything inside a rank -- index, Aligner, streams -- is what a single-GPU run uses.
#pragma once
#include <atomic>
#include <functional>
#include <string>
#include <stdexcept>
#include <thread>
#include <vector>
#include "types.h"

namespace nvbio {
namespace hip {

/// rank `rank` of `world` owns items [first, second) of n_total (nvbio_amd/distributed.py: shard_range -- the same split)
inline std::pair<uint64, uint64> shard_range(const uint64 n_total, const uint32 rank, const uint32 world)
{
    if (world == 0u) return std::make_pair(uint64(0), uint64(0));
    const uint64 per = (n_total + world - 1u) / world;
    const uint64 lo = std::min<uint64>(n_total, uint64(rank) * per);
    return std::make_pair(lo, std::min<uint64>(n_total, lo + per));
}
inline std::vector<uint64> shard_sizes(const uint64 n_total, const uint32 world)
{
    std::vector<uint64> s(world);
    for (uint32 r = 0; r < world; ++r) { const std::pair<uint64, uint64> p = shard_range(n_total, r, world); s[r] = p.second - p.first; }
    return s;
}

struct DeviceGroup
{
    /// what a rank's body sees
    struct Rank
    {
        uint32 rank, world;
        int    device;
        void*  comm;           // RCCL communicator (NULL in a world of one)

        /// counts[r] records of record_bytes bytes from every rank r to `root` (recv: device memory on the root, NULL elsewhere), queued on `stream`
        void gather_records(const void* send, const std::vector<uint64>& counts, const uint32 record_bytes, void* recv, const uint32 root = 0, void* stream = nullptr) const
        {
            if (world == 1u)
            {
                if (counts[0] && recv != send) hip_check(nvbio_hip_memcpy(recv, send, counts[0] * record_bytes, 3, stream), "nvbio_hip_memcpy(d2d)");
                return;
            }
            hip_check(nvbio_hip_gather_records(comm, send, counts.data(), record_bytes, recv, int(root), stream), "nvbio_hip_gather_records");
        }
    };

    DeviceGroup() : m_owns_comms(true), m_aborted(false) { check_abi(); }
    /// (an aborted communicator is already released -- ncclCommAbort frees it --, so the failure path of run() must not destroy it again)
    ~DeviceGroup() { if (m_owns_comms && !m_aborted.load()) for (size_t i = 0; i < m_ranks.size(); ++i) if (m_ranks[i].comm) (void)nvbio_hip_comm_destroy(m_ranks[i].comm); }
    DeviceGroup(const DeviceGroup&) = delete;
    DeviceGroup& operator=(const DeviceGroup&) = delete;

    /// one process, one host thread per device: devices 0 .. n-1 (n = 0: all the process sees)
    static void local(DeviceGroup& g, uint32 n_devices = 0)
    {
        const int have = nvbio_hip_device_count();
        if (n_devices == 0) n_devices = uint32(have);
        if (n_devices == 0 || int(n_devices) > have) throw hip_error("DeviceGroup::local (device count)", 101);
        g.m_ranks.resize(n_devices);
        std::vector<void*> comms(n_devices, nullptr);
        if (n_devices > 1u)
        {
            std::vector<int> devs(n_devices);
            for (uint32 d = 0; d < n_devices; ++d) devs[d] = int(d);
            hip_check(nvbio_hip_comm_init_all(comms.data(), int(n_devices), devs.data()), "nvbio_hip_comm_init_all");
        }
        for (uint32 d = 0; d < n_devices; ++d) { Rank r = { d, n_devices, int(d), comms[d] }; g.m_ranks[d] = r; }
    }
    /// one process per device: this process is rank `rank` of `world` on the CURRENT device; id128 = rank 0's nvbio_hip_comm_unique_id()
    static void from_unique_id(DeviceGroup& g, const uint32 rank, const uint32 world, const uint8* id128)
    {
        void* comm = nullptr;
        if (world > 1u) hip_check(nvbio_hip_comm_init_rank(&comm, int(world), int(rank), id128), "nvbio_hip_comm_init_rank");
        Rank r = { rank, world, nvbio_hip_get_device(), comm };
        g.m_ranks.assign(1, r);
    }

    /// ranks over communicators made elsewhere (one per local rank, all of one world); devices[i] < 0 = do not bind a device (hosts
    /// without one: the CPU suite runs the group over a host-memory transport, nvbio_hip_comm_set_transport).  The group does not own
    /// these communicators.
    static void from_comms(DeviceGroup& g, const std::vector<void*>& comms, const std::vector<int>& devices)
    {
        g.m_ranks.resize(comms.size());
        for (size_t i = 0; i < comms.size(); ++i) { Rank r = { uint32(i), uint32(comms.size()), devices[i], comms[i] }; g.m_ranks[i] = r; }
        g.m_owns_comms = false;
    }

    size_t size() const { return m_ranks.size(); }
    const Rank& operator[](const size_t i) const { return m_ranks[i]; }

    /// run `body` once per local rank: on its own host thread bound to its device when this group holds several (compute_thread.cu:95:
    /// cudaSetDevice per thread), inline otherwise.  The first exception of any thread is rethrown after all have joined.  A rank that
    /// fails never reaches the collective its peers are waiting in, so the failing thread ABORTS every communicator of the group
    /// (ncclCommAbort): the peers' pending receives return an error, their threads end, and the caller gets the original exception
    /// instead of a hang.
    void run(const std::function<void(const Rank&)>& body) const
    {
        if (m_ranks.size() == 1u) { body(m_ranks[0]); return; }
        std::vector<std::thread> threads;
        std::vector<std::string> errors(m_ranks.size());
        std::atomic<int> first_failed(-1);
        auto fail = [&](const size_t i, const char* what) {
            errors[i] = (what && what[0]) ? what : "error";
            int expected = -1;
            if (first_failed.compare_exchange_strong(expected, int(i)))
            {
                m_aborted.store(true);
                for (size_t k = 0; k < m_ranks.size(); ++k) if (m_ranks[k].comm) (void)nvbio_hip_comm_abort(m_ranks[k].comm);
            }
        };
        for (size_t i = 0; i < m_ranks.size(); ++i)
            threads.emplace_back([&, i] {
                try { if (m_ranks[i].device >= 0) hip_check(nvbio_hip_set_device(m_ranks[i].device), "nvbio_hip_set_device"); body(m_ranks[i]); }
                catch (const std::exception& e) { fail(i, e.what()); }
                catch (...) { fail(i, "unknown exception"); }
            });
        for (size_t i = 0; i < threads.size(); ++i) threads[i].join();
        const int f = first_failed.load();
        if (f >= 0) throw std::runtime_error("rank " + std::to_string(f) + ": " + errors[size_t(f)]);
    }

private:
    std::vector<Rank> m_ranks;
    bool              m_owns_comms;
    mutable std::atomic<bool> m_aborted;      ///< run() aborted (= released) the communicators: the group is dead, nothing left to destroy
};

} // namespace hip
} // namespace nvbio
