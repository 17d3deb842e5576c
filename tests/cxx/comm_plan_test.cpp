// tests/cxx/comm_plan_test.cpp -- the path's only collective on the CPU suite: nvbio_hip_gather_records (the plan of
// include/nvbio_hip/gather_plan.h) and hip::DeviceGroup (one host thread per rank) driven over a HOST-MEMORY transport installed through
// nvbio_hip_comm_set_transport, at worlds of 2, 3, 5 and 8 with ragged and empty shards and every root -- so that a swapped offset, a
// mis-ordered rank, a deadlock between the grouped receives and the sends, or a hang after one rank fails is caught without a multi-GPU
// node.  Mirrors the shape of the reference's multi-GPU mode: one host thread per device writing into one output in read order
// (nvBowtie/nvBowtie.cpp:809-864).
//
// The transport: a "communicator" is {rank, world, hub}; send deposits the bytes in hub->box[src][dst] and wakes the receiver; recv inside a
// group is only REGISTERED, group_end waits for each box and copies it out -- the semantics of grouped ncclSend / ncclRecv that matter here
// (nothing completes before the group closes; a receive blocks until its sender arrives); abort wakes everybody with an error.
#include <nvbio_hip.h>
#include <nvbio_hip/types.h>
#include <nvbio_hip/multi_device.h>
#include <nvbio_hip/gather_plan.h>
#include <condition_variable>
#include <mutex>
#include <chrono>
#include <vector>
#include <string.h>
#include <stdio.h>

using namespace nvbio;

namespace {

struct Hub
{
    explicit Hub(int w) : world(w), box(size_t(w) * w), full(size_t(w) * w, 0), aborted(false) {}
    int world;
    std::mutex m; std::condition_variable cv;
    std::vector< std::vector<uint8> > box; std::vector<int> full; bool aborted;
};
struct FakeComm
{
    int rank, world; Hub* hub;
    struct Pending { void* dst; uint64 bytes; int peer; };
    std::vector<Pending> pending; bool in_group = false;
};
int f_rank(void* c, int* r, int* w) { FakeComm* f = static_cast<FakeComm*>(c); *r = f->rank; *w = f->world; return 0; }
int f_group_start(void* c) { static_cast<FakeComm*>(c)->in_group = true; return 0; }
int take(FakeComm* f, const FakeComm::Pending& p)
{
    Hub& h = *f->hub;
    std::unique_lock<std::mutex> lock(h.m);
    const size_t slot = size_t(p.peer) * h.world + f->rank;
    if (!h.cv.wait_for(lock, std::chrono::seconds(20), [&] { return h.full[slot] || h.aborted; })) return 9001;      // a deadlock would show up here
    if (!h.full[slot]) return 9002;                                                                               // aborted
    if (h.box[slot].size() != p.bytes) return 9003;
    memcpy(p.dst, h.box[slot].data(), p.bytes);
    h.full[slot] = 0;
    return 0;
}
int f_group_end(void* c)
{
    FakeComm* f = static_cast<FakeComm*>(c);
    f->in_group = false;
    int err = 0;
    for (size_t k = 0; k < f->pending.size() && !err; ++k) err = take(f, f->pending[k]);
    f->pending.clear();
    return err;
}
int f_send(void* c, const void* buf, uint64 bytes, int peer, void*)
{
    FakeComm* f = static_cast<FakeComm*>(c);
    Hub& h = *f->hub;
    std::lock_guard<std::mutex> lock(h.m);
    if (h.aborted) return 9002;
    const size_t slot = size_t(f->rank) * h.world + peer;
    h.box[slot].assign(static_cast<const uint8*>(buf), static_cast<const uint8*>(buf) + bytes);
    h.full[slot] = 1;
    h.cv.notify_all();
    return 0;
}
int f_recv(void* c, void* buf, uint64 bytes, int peer, void*)
{
    FakeComm* f = static_cast<FakeComm*>(c);
    const FakeComm::Pending p = { buf, bytes, peer };
    if (f->in_group) { f->pending.push_back(p); return 0; }
    return take(f, p);
}
int f_copy(void*, void* dst, const void* src, uint64 bytes, void*) { if (dst != src) memcpy(dst, src, bytes); return 0; }
int f_abort(void* c)
{
    Hub& h = *static_cast<FakeComm*>(c)->hub;
    std::lock_guard<std::mutex> lock(h.m);
    h.aborted = true;
    h.cv.notify_all();
    return 0;
}
const nvbio_hip_comm_transport k_fake = { f_rank, f_group_start, f_group_end, f_send, f_recv, f_copy, f_abort };

/// record k of rank r: its rank, its index within the rank, its GLOBAL index (what a read id is), a check word
void fill(std::vector<uint32>& rec, const uint32 words, const uint32 rank, const uint64 count, const uint64 first_global)
{
    rec.assign(size_t(count) * words, 0u);
    for (uint64 k = 0; k < count; ++k)
        for (uint32 w = 0; w < words; ++w)
            rec[k * words + w] = w == 0 ? rank : w == 1 ? uint32(k) : w == 2 ? uint32(first_global + k) : (rank * 2654435761u) ^ uint32(k * 40503u + w);
}

int run_case(const std::vector<uint64>& counts, const uint32 words, const uint32 root, const int fail_rank)
{
    const uint32 world = uint32(counts.size());
    Hub hub{ int(world) };
    std::vector<FakeComm> comms(world);
    std::vector<void*> handles(world);
    std::vector<int> devices(world, -1);
    for (uint32 r = 0; r < world; ++r) { comms[r].rank = int(r); comms[r].world = int(world); comms[r].hub = &hub; handles[r] = &comms[r]; }
    hip::DeviceGroup group;
    hip::DeviceGroup::from_comms(group, handles, devices);
    uint64 total = 0;
    std::vector<uint64> first(world);
    for (uint32 r = 0; r < world; ++r) { first[r] = total; total += counts[r]; }
    std::vector<uint32> out(size_t(total) * words + 1u, 0xDEADBEEFu);
    bool threw = false;
    try
    {
        group.run([&](const hip::DeviceGroup::Rank& me) {
            if (int(me.rank) == fail_rank) throw std::runtime_error("planted failure before the collective");
            std::vector<uint32> rec;
            fill(rec, words, me.rank, counts[me.rank], first[me.rank]);
            me.gather_records(rec.empty() ? NULL : rec.data(), counts, words * 4u, me.rank == root ? out.data() : NULL, root);
        });
    }
    catch (const std::exception& e) { threw = true; if (fail_rank < 0) { fprintf(stderr, "unexpected: %s\n", e.what()); return 1; } }
    if (fail_rank >= 0) return threw ? 0 : 2;                           // the planted failure must surface (and must not hang: take() times out)
    // the root's buffer: rank order, each rank's records in their own order, nothing past the end
    std::vector<uint32> expect, rec;
    for (uint32 r = 0; r < world; ++r) { fill(rec, words, r, counts[r], first[r]); expect.insert(expect.end(), rec.begin(), rec.end()); }
    if (out[size_t(total) * words] != 0xDEADBEEFu) return 3;
    if (total && memcmp(out.data(), expect.data(), expect.size() * 4u) != 0) return 4;
    if (words > 2) for (uint64 g = 0; g < total; ++g) if (out[g * words + 2] != uint32(g)) return 5;           // global read order
    return 0;
}

} // namespace

extern "C" int comm_plan_selftest()
{
    nvbio_hip_comm_set_transport(&k_fake);
    int bad = 0, n_cases = 0;
    const uint64 shapes[][8] = { { 5, 5, 5, 5, 5, 5, 5, 5 }, { 7, 0, 3, 11, 0, 0, 1, 6 }, { 0, 0, 0, 0, 0, 0, 0, 9 }, { 1000, 999, 998, 0, 1, 2, 3, 4 }, { 0, 0, 0, 0, 0, 0, 0, 0 } };
    for (uint32 world : { 2u, 3u, 5u, 8u })          // (a world of one copies device to device without a communicator: the GPU suite has it)
        for (const auto& shape : shapes)
            for (uint32 words : { 1u, 4u, 8u })
                for (uint32 root = 0; root < world; root += (world > 3u ? 3u : 1u))
                {
                    const std::vector<uint64> counts(shape, shape + world);
                    const int e = run_case(counts, words, root, -1);
                    ++n_cases;
                    if (e) { fprintf(stderr, "comm_plan: world %u words %u root %u failed with %d\n", world, words, root, e); ++bad; }
                }
    // sharding as the drivers do it: shard_sizes of a ragged total
    for (uint32 world : { 2u, 3u, 8u })
        for (uint64 total : { uint64(100003), uint64(7), uint64(8), uint64(0) })
        {
            const int e = run_case(hip::shard_sizes(total, world), 4u, 0u, -1);
            ++n_cases;
            if (e) { fprintf(stderr, "comm_plan: shard_sizes(%llu, %u) failed with %d\n", (unsigned long long)total, world, e); ++bad; }
        }
    // a rank that throws before the collective: the others must not wait for it
    for (uint32 world : { 2u, 3u, 8u })
        for (int fail_rank : { 0, int(world) - 1 })
        {
            const int e = run_case(std::vector<uint64>(world, 50), 4u, 0u, fail_rank);
            ++n_cases;
            if (e) { fprintf(stderr, "comm_plan: failure of rank %d in a world of %u was not reported (%d)\n", fail_rank, world, e); ++bad; }
        }
    // the plan itself: two ranks' receive offsets swapped must be detected by the checks above -- prove the checker bites
    {
        const uint64 counts[3] = { 4, 2, 3 };
        nvbio_hip_gather_op ops[3];
        const int n = nvbio_hip_gather_plan(counts, 3, 0, 0, 8u, ops);
        if (n != 3 || ops[0].kind != NVBIO_HIP_GATHER_COPY || ops[1].offset != 32u || ops[2].offset != 48u || ops[1].peer != 1 || ops[2].bytes != 24u) { fprintf(stderr, "comm_plan: plan of the root is wrong\n"); ++bad; }
        if (nvbio_hip_gather_plan(counts, 3, 2, 0, 8u, ops) != 1 || ops[0].kind != NVBIO_HIP_GATHER_SEND || ops[0].bytes != 24u) { fprintf(stderr, "comm_plan: plan of a sender is wrong\n"); ++bad; }
        if (nvbio_hip_gather_plan(counts, 3, 3, 0, 8u, ops) != -1 || nvbio_hip_gather_plan(counts, 0, 0, 0, 8u, ops) != -1) ++bad;
        ++n_cases;
    }
    nvbio_hip_comm_set_transport(NULL);
    fprintf(stderr, "comm_plan: %d cases, %d failed\n", n_cases, bad);
    return bad;
}
