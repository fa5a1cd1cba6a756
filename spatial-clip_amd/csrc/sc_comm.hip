// RCCL behind the C ABI (sc_comm_init / destroy, sc_allgather_feats_async, sc_reduce_scatter_grads_async):
// the three collectives of the data-parallel step -- the packed feature all-gather (ClipLoss / SpatialLoss
// `gather_features`, src/open_clip/loss.py:21-65), its backward reduce-scatter, and the bucketed gradient all-reduce
// that stands in for DDP's (configs/trainer/ddp.yaml) -- enqueued on a caller-supplied HIP stream, with no torch type
// in the signatures.  One process per GPU; the communicator is an opaque handle owned by the caller.
//
// RCCL is resolved at run time (dlopen / dlsym): the library that PyTorch already loaded is reused when there is one
// (RTLD_NOLOAD first), so a process never ends up with two RCCL instances, and the kernel library itself carries no
// link-time dependency on RCCL -- single-GPU use needs none.
#include "sc_common.h"
#include "sc_kernels.h"
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <string.h>

namespace {

struct Rccl {
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*ReduceScatter)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

// Resolved once; C++11 guarantees that the initialisation of a function-local static runs exactly once even when
// several threads call in at the same time (the header promises re-entrancy).
Rccl load_rccl() {
    Rccl r;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void* h = nullptr;
    for (const char* n : names)
        if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD)) != nullptr) break;          // the instance the process already has
    if (!h)
        for (const char* n : names)
            if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL)) != nullptr) break;
    if (!h) {
        sc_set_error("RCCL not found (tried librccl.so.1, librccl.so, /opt/rocm/lib): %s", dlerror());
        return r;
    }
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(h, "ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
    r.AllGather = (decltype(r.AllGather))dlsym(h, "ncclAllGather");
    r.ReduceScatter = (decltype(r.ReduceScatter))dlsym(h, "ncclReduceScatter");
    r.AllReduce = (decltype(r.AllReduce))dlsym(h, "ncclAllReduce");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllGather && r.ReduceScatter && r.AllReduce;
    if (!r.ok) sc_set_error("RCCL library lacks a collective entry point");
    return r;
}

Rccl& rccl() {
    static Rccl r = load_rccl();
    return r;
}

int fail(const char* what, ncclResult_t rc) {
    Rccl& r = rccl();
    sc_set_error("%s: %s", what, r.GetErrorString ? r.GetErrorString(rc) : "RCCL error");
    return -3;
}

}  // namespace

extern "C" int sc_comm_unique_id(void* id_out_128) {
    Rccl& r = rccl();
    if (!r.ok) return -1;
    SC_CHECK(id_out_128 != nullptr, "sc_comm_unique_id: null buffer");
    ncclUniqueId id;
    const ncclResult_t rc = r.GetUniqueId(&id);
    if (rc != ncclSuccess) return fail("ncclGetUniqueId", rc);
    memcpy(id_out_128, id.internal, NCCL_UNIQUE_ID_BYTES);
    return 0;
}

extern "C" long long sc_comm_init(const void* id_128, int rank, int world) {
    Rccl& r = rccl();
    if (!r.ok) return 0;
    if (id_128 == nullptr || world < 1 || rank < 0 || rank >= world) {
        sc_set_error("sc_comm_init: bad arguments rank=%d world=%d", rank, world);
        return 0;
    }
    ncclUniqueId id;
    memcpy(id.internal, id_128, NCCL_UNIQUE_ID_BYTES);
    ncclComm_t comm = nullptr;
    const ncclResult_t rc = r.CommInitRank(&comm, world, id, rank);       // binds to the calling thread's current HIP device
    if (rc != ncclSuccess) {
        fail("ncclCommInitRank", rc);
        return 0;
    }
    return (long long)(uintptr_t)comm;
}

extern "C" int sc_comm_destroy(void* comm) {
    Rccl& r = rccl();
    if (!r.ok) return -1;
    if (comm == nullptr) return 0;
    const ncclResult_t rc = r.CommDestroy((ncclComm_t)comm);
    return rc == ncclSuccess ? 0 : fail("ncclCommDestroy", rc);
}

extern "C" int sc_allgather_feats_async(void* comm, const void* send, void* recv, long long bytes_per_rank, void* stream) {
    Rccl& r = rccl();
    if (!r.ok) return -1;
    SC_CHECK(comm && send && recv && bytes_per_rank > 0, "sc_allgather_feats_async: bad arguments (bytes=%lld)", bytes_per_rank);
    const ncclResult_t rc = r.AllGather(send, recv, (size_t)bytes_per_rank, ncclInt8, (ncclComm_t)comm, (hipStream_t)stream);
    return rc == ncclSuccess ? 0 : fail("ncclAllGather", rc);
}

extern "C" int sc_reduce_scatter_grads_async(void* comm, const float* send, float* recv, long long floats_per_rank,
                                             void* stream) {
    Rccl& r = rccl();
    if (!r.ok) return -1;
    SC_CHECK(comm && send && recv && floats_per_rank > 0, "sc_reduce_scatter_grads_async: bad arguments");
    const ncclResult_t rc = r.ReduceScatter(send, recv, (size_t)floats_per_rank, ncclFloat32, ncclSum, (ncclComm_t)comm,
                                            (hipStream_t)stream);
    return rc == ncclSuccess ? 0 : fail("ncclReduceScatter", rc);
}

extern "C" int sc_allreduce_sum_async(void* comm, float* buf, long long n, void* stream) {
    Rccl& r = rccl();
    if (!r.ok) return -1;
    SC_CHECK(comm && buf && n > 0, "sc_allreduce_sum_async: bad arguments");
    const ncclResult_t rc = r.AllReduce(buf, buf, (size_t)n, ncclFloat32, ncclSum, (ncclComm_t)comm, (hipStream_t)stream);
    return rc == ncclSuccess ? 0 : fail("ncclAllReduce", rc);
}
