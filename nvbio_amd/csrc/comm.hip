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
#include <dlfcn.h>
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

NVB_API int nvbio_hip_comm_rank(void* comm, int* rank, int* world)
{
    const Rccl* r = rccl();
    if (!r) return hipErrorNotSupported;
    if (!comm || !rank || !world) return hipErrorInvalidValue;
    if (int e = rc(r->CommUserRank(static_cast<ncclComm_t>(comm), rank))) return e;
    return rc(r->CommCount(static_cast<ncclComm_t>(comm), world));
}

// counts[r] records of record_bytes bytes from rank r (every rank passes the same counts); the root's recv buffer holds them in rank
// order (sum(counts) records), other ranks pass recv = NULL.  Queued on `stream`; send / recv buffers are device memory.
NVB_API int nvbio_hip_gather_records(void* comm, const void* send, const uint64_t* counts, uint32_t record_bytes, void* recv, int root, void* stream)
{
    const Rccl* r = rccl();
    if (!r) return hipErrorNotSupported;
    if (!comm || !counts || record_bytes == 0) return hipErrorInvalidValue;
    int rank = 0, world = 0;
    if (int e = nvbio_hip_comm_rank(comm, &rank, &world)) return e;
    if (root < 0 || root >= world) return hipErrorInvalidValue;
    if (counts[rank] != 0 && !send) return hipErrorInvalidValue;
    if (rank == root && !recv) return hipErrorInvalidValue;
    hipStream_t s = to_stream(stream);
    ncclComm_t c = static_cast<ncclComm_t>(comm);
    if (rank == root)
    {
        uint64_t off = 0;
        if (int e = rc(r->GroupStart())) return e;
        for (int k = 0; k < world; ++k)
        {
            const uint64_t bytes = counts[k] * record_bytes;
            if (k != root && bytes) { if (int e = rc(r->Recv(static_cast<uint8_t*>(recv) + off, bytes, ncclUint8, k, c, s))) { (void)r->GroupEnd(); return e; } }
            off += bytes;
        }
        if (int e = rc(r->GroupEnd())) return e;
        uint64_t own = 0;
        for (int k = 0; k < root; ++k) own += counts[k] * record_bytes;
        if (counts[root]) { if (hipError_t e = hipMemcpyAsync(static_cast<uint8_t*>(recv) + own, send, counts[root] * record_bytes, hipMemcpyDeviceToDevice, s)) return e; }
        return 0;
    }
    if (counts[rank] == 0) return 0;
    return rc(r->Send(send, counts[rank] * record_bytes, ncclUint8, root, c, s));
}
