// comm.hip -- the one collective of the path: fixed-size result records gathered to one rank over RCCL / xGMI, from C++.
//
// The reference has no collective: nvBowtie runs one host thread per device over a replicated index and every thread writes into a
// shared output object (nvBowtie/nvBowtie.cpp:809-864, bowtie2/cuda/compute_thread.cu:74-117).  Here every device (one host thread
// each in a single process, or one process each under a launcher) aligns a contiguous block of the reads; the gather of the 16-byte
// alignment records / 32-byte pair records / 4-12-byte score records replaces the shared output.  xGMI is point to point, so the gather
// is grouped ncclSend / ncclRecv towards the root (the root takes in (G-1)/G of the bytes once over its 7 links; nobody else receives
// anything) rather than a ring all-gather.
//
// RCCL is bound at run time (dlopen): a single-GPU user of libnvbio_hip.so needs no librccl, and a process that already holds one
// (PyTorch ships its own librccl.so with the same SONAME) keeps using that instance.
#include "common.h"
#include "../../include/nvbio_hip/gather_plan.h"
#include <dlfcn.h>
#include <atomic>
#include <vector>
#include <string.h>
#include <mutex>
#include <rccl/rccl.h>

namespace nvb {

struct Rccl
{
    void* handle;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*);
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int);
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*);
    ncclResult_t (*CommDestroy)(ncclComm_t);
    ncclResult_t (*CommAbort)(ncclComm_t);
    ncclResult_t (*CommCount)(const ncclComm_t, int*);
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*);
    ncclResult_t (*GroupStart)();
    ncclResult_t (*GroupEnd)();
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    const char*  (*GetErrorString)(ncclResult_t);
};

static const Rccl* rccl()
{
    static std::once_flag once;
    static Rccl api = {};
    std::call_once(once, [] {
        const char* names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
        void* h = nullptr;
        for (const char* n : names) if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD)) != nullptr) break;          // an instance this process already holds
        for (const char* n : names) { if (h) break; h = dlopen(n, RTLD_NOW | RTLD_LOCAL); }
        if (!h) return;
        bool ok = true;
        auto sym = [&](const char* n) { void* p = dlsym(h, n); if (!p) ok = false; return p; };
        api.GetUniqueId   = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
        api.CommInitRank  = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
        api.CommInitAll   = reinterpret_cast<decltype(api.CommInitAll)>(sym("ncclCommInitAll"));
        api.CommDestroy   = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
        api.CommAbort     = reinterpret_cast<decltype(api.CommAbort)>(dlsym(h, "ncclCommAbort"));          // optional
        api.CommCount     = reinterpret_cast<decltype(api.CommCount)>(sym("ncclCommCount"));
        api.CommUserRank  = reinterpret_cast<decltype(api.CommUserRank)>(sym("ncclCommUserRank"));
        api.GroupStart    = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
        api.GroupEnd      = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
        api.Send          = reinterpret_cast<decltype(api.Send)>(sym("ncclSend"));
        api.Recv          = reinterpret_cast<decltype(api.Recv)>(sym("ncclRecv"));
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
        if (ok) api.handle = h;
    });
    return api.handle ? &api : nullptr;
}

// RCCL results are reported in their own range so that they cannot be mistaken for hipError_t values
static inline int rc(const ncclResult_t r) { return r == ncclSuccess ? 0 : 2000 + int(r); }

} // namespace nvb

using namespace nvb;

NVB_API int nvbio_hip_comm_available(void) { return rccl() ? 1 : 0; }

NVB_API int nvbio_hip_device_count(void) { int n = 0; return hipGetDeviceCount(&n) == hipSuccess ? n : 0; }
NVB_API int nvbio_hip_set_device(int device) { return hipSetDevice(device); }
NVB_API int nvbio_hip_get_device(void) { int d = -1; return hipGetDevice(&d) == hipSuccess ? d : -1; }

NVB_API int nvbio_hip_comm_unique_id(uint8_t* id128)
{
    const Rccl* r = rccl();
    if (!r) return hipErrorNotSupported;
    if (!id128) return hipErrorInvalidValue;
    ncclUniqueId id;
    if (int e = rc(r->GetUniqueId(&id))) return e;
    memcpy(id128, id.internal, NCCL_UNIQUE_ID_BYTES);
    return 0;
}

NVB_API int nvbio_hip_comm_init_rank(void** comm, int world, int rank, const uint8_t* id128)
{
    const Rccl* r = rccl();
    if (!r) return hipErrorNotSupported;
    if (!comm || !id128 || world < 1 || rank < 0 || rank >= world) return hipErrorInvalidValue;
    ncclUniqueId id;
    memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
    ncclComm_t c = nullptr;
    const int e = rc(r->CommInitRank(&c, world, id, rank));
    *comm = c;
    return e;
}

NVB_API int nvbio_hip_comm_init_all(void** comms, int n_devices, const int* devices)
{
    const Rccl* r = rccl();
    if (!r) return hipErrorNotSupported;
    if (!comms || n_devices < 1) return hipErrorInvalidValue;
    return rc(r->CommInitAll(reinterpret_cast<ncclComm_t*>(comms), n_devices, devices));
}

NVB_API int nvbio_hip_comm_destroy(void* comm)
{
    const Rccl* r = rccl();
    if (!r) return hipErrorNotSupported;
    return comm ? rc(r->CommDestroy(static_cast<ncclComm_t>(comm))) : 0;
}

// ---- the transport seam.  nvbio_hip_gather_records executes the plan of include/nvbio_hip/gather_plan.h through a table of five
// operations; the default table is RCCL (grouped ncclSend / ncclRecv + a device-to-device copy for the root's own records).
// nvbio_hip_comm_set_transport installs another one (NULL: back to RCCL): the CPU suite drives the plan over host memory at worlds a
// build container cannot have, and a deployment without RCCL could put MPI or shared memory behind the same entry point.
namespace nvb {
static std::atomic<const nvbio_hip_comm_transport*> g_transport{nullptr};

static int rccl_rank(void* comm, int* rank, int* world)
{
    const Rccl* r = rccl();
    if (!r) return hipErrorNotSupported;
    if (int e = rc(r->CommUserRank(static_cast<ncclComm_t>(comm), rank))) return e;
    return rc(r->CommCount(static_cast<ncclComm_t>(comm), world));
}
static int rccl_group_start(void*) { const Rccl* r = rccl(); return r ? rc(r->GroupStart()) : int(hipErrorNotSupported); }
static int rccl_group_end(void*)   { const Rccl* r = rccl(); return r ? rc(r->GroupEnd()) : int(hipErrorNotSupported); }
static int rccl_send(void* comm, const void* buf, uint64_t bytes, int peer, void* stream)
{ const Rccl* r = rccl(); return r ? rc(r->Send(buf, bytes, ncclUint8, peer, static_cast<ncclComm_t>(comm), to_stream(stream))) : int(hipErrorNotSupported); }
static int rccl_recv(void* comm, void* buf, uint64_t bytes, int peer, void* stream)
{ const Rccl* r = rccl(); return r ? rc(r->Recv(buf, bytes, ncclUint8, peer, static_cast<ncclComm_t>(comm), to_stream(stream))) : int(hipErrorNotSupported); }
static int rccl_copy(void*, void* dst, const void* src, uint64_t bytes, void* stream)
{ return dst == src ? 0 : int(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, to_stream(stream))); }
static int rccl_abort(void* comm)
{ const Rccl* r = rccl(); return (r && r->CommAbort && comm) ? rc(r->CommAbort(static_cast<ncclComm_t>(comm))) : 0; }
static const nvbio_hip_comm_transport k_rccl_transport = { rccl_rank, rccl_group_start, rccl_group_end, rccl_send, rccl_recv, rccl_copy, rccl_abort };

static const nvbio_hip_comm_transport* transport()
{
    const nvbio_hip_comm_transport* t = g_transport.load(std::memory_order_acquire);
    return t ? t : &k_rccl_transport;
}
} // namespace nvb

NVB_API int nvbio_hip_comm_rank(void* comm, int* rank, int* world)
{
    if (!comm || !rank || !world) return hipErrorInvalidValue;
    return transport()->rank(comm, rank, world);
}

NVB_API void nvbio_hip_comm_set_transport(const nvbio_hip_comm_transport* t) { nvb::g_transport.store(t, std::memory_order_release); }

/* unblock peers waiting on this communicator after a local failure (ncclCommAbort); the communicator is unusable afterwards */
NVB_API int nvbio_hip_comm_abort(void* comm) { return comm ? transport()->abort(comm) : 0; }

// counts[r] records of record_bytes bytes from rank r (every rank passes the same counts); the root's recv buffer holds them in rank
// order (sum(counts) records), other ranks pass recv = NULL.  Queued on `stream`; send / recv buffers are device memory.
NVB_API int nvbio_hip_gather_records(void* comm, const void* send, const uint64_t* counts, uint32_t record_bytes, void* recv, int root, void* stream)
{
    const nvbio_hip_comm_transport* t = transport();
    if (t == &k_rccl_transport && !rccl()) return hipErrorNotSupported;
    if (!comm || !counts || record_bytes == 0) return hipErrorInvalidValue;
    int rank = 0, world = 0;
    if (int e = t->rank(comm, &rank, &world)) return e;
    if (root < 0 || root >= world) return hipErrorInvalidValue;
    if (counts[rank] != 0 && !send) return hipErrorInvalidValue;
    if (rank == root && !recv) return hipErrorInvalidValue;
    std::vector<nvbio_hip_gather_op> ops(size_t(world) > 0 ? size_t(world) : 1u);
    const int n_ops = nvbio_hip_gather_plan(counts, world, rank, root, record_bytes, ops.data());
    if (n_ops < 0) return hipErrorInvalidValue;
    // the receives of the root form one group (all posted before any is waited for); sends and the local copy stand alone
    bool grouped = false;
    for (int k = 0; k < n_ops; ++k) grouped = grouped || ops[k].kind == NVBIO_HIP_GATHER_RECV;
    if (grouped) { if (int e = t->group_start(comm)) return e; }
    for (int k = 0; k < n_ops; ++k)
        if (ops[k].kind == NVBIO_HIP_GATHER_RECV)
        { if (int e = t->recv(comm, static_cast<uint8_t*>(recv) + ops[k].offset, ops[k].bytes, ops[k].peer, stream)) { (void)t->group_end(comm); return e; } }
    if (grouped) { if (int e = t->group_end(comm)) return e; }
    for (int k = 0; k < n_ops; ++k)
    {
        if (ops[k].kind == NVBIO_HIP_GATHER_SEND) { if (int e = t->send(comm, send, ops[k].bytes, ops[k].peer, stream)) return e; }
        else if (ops[k].kind == NVBIO_HIP_GATHER_COPY) { if (int e = t->copy(comm, static_cast<uint8_t*>(recv) + ops[k].offset, send, ops[k].bytes, stream)) return e; }
    }
    return 0;
}
