// Gradient exchange behind the C ABI: RCCL collectives over xGMI as plan-able entry points (SURVEY.md section 8b
// xl_comm_{init,rs,ag,bcast,reduce,destroy}).  The reference wraps its model in DistributedDataParallel and lets torch's
// reducer issue NCCL all-reduces from backward hooks (ref x-lxmert/src/pretrain/lxmert_pretrain.py:102-106, 694-700); here the
// trainer owns the flat gradient buffer, knows when a contiguous slice of it is final, and issues the collective for that slice
// itself -- through these calls the issue points are ordinary entries of the step's launch plan (csrc/plan.hip), so a
// data-parallel step is ONE xl_plan_run like the single-GPU one.
//
// RCCL is bound at run time (dlopen of librccl.so.1: the copy torch already loaded, if any, else the ROCm one), so the library
// itself loads -- and the single-GPU path runs -- on a machine without it.  One communicator per process (one process per
// GPU); every collective runs on the communicator's own HIP stream, ordered after the compute stream by an event recorded when
// the call is issued, and the compute stream waits for all of them with xl_comm_wait -- neither compute stream ever blocks on a
// collective it does not need.
#include <dlfcn.h>
#include <stdlib.h>
#include <mutex>
#include <vector>
#include "common.h"

namespace xl {

struct NcclId { char internal[128]; };
typedef void* NcclComm;
// (ncclDataType_t / ncclRedOp_t values of rccl.h; checked against the header by tests/test_cabi.py)
enum { NCCL_UINT8 = 1, NCCL_INT64 = 4, NCCL_FLOAT32 = 7, NCCL_BFLOAT16 = 9, NCCL_SUM = 0, NCCL_MAX = 2 };

struct Rccl {
    void* handle = nullptr;
    int (*GetUniqueId)(NcclId*) = nullptr;
    int (*CommInitRank)(NcclComm*, int, NcclId, int) = nullptr;
    int (*CommDestroy)(NcclComm) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
    int (*ReduceScatter)(const void*, void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, NcclComm, hipStream_t) = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
    int (*Reduce)(const void*, void*, size_t, int, int, int, NcclComm, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    int (*CommCount)(NcclComm, int*) = nullptr;          // optional (ncclCommCount)
};
static std::mutex g_rccl_mu;
static Rccl g_rccl;                // function table, filled once (immutable afterwards)

static const Rccl* rccl() {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.handle != nullptr) return &g_rccl;
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);          // the copy this process already has (torch's)
    if (h == nullptr) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (h == nullptr) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (h == nullptr) { set_error("xl_comm: cannot load librccl.so.1: %s", dlerror()); return nullptr; }
    Rccl r;
    r.handle = h;
#define XL_SYM(field, name) *(void**)(&r.field) = dlsym(h, name); if (r.field == nullptr) { set_error("xl_comm: librccl lacks %s", name); return nullptr; }
    XL_SYM(GetUniqueId, "ncclGetUniqueId") XL_SYM(CommInitRank, "ncclCommInitRank") XL_SYM(CommDestroy, "ncclCommDestroy")
    XL_SYM(AllReduce, "ncclAllReduce") XL_SYM(ReduceScatter, "ncclReduceScatter") XL_SYM(AllGather, "ncclAllGather")
    XL_SYM(Broadcast, "ncclBroadcast") XL_SYM(Reduce, "ncclReduce") XL_SYM(GetErrorString, "ncclGetErrorString")
#undef XL_SYM
    *(void**)(&r.CommCount) = dlsym(h, "ncclCommCount");
    g_rccl = r;
    return &g_rccl;
}

struct Comm {
    NcclComm comm = nullptr;
    int rank = 0, nranks = 1;
    hipStream_t stream = nullptr;      // the collectives' own stream (the caller's, or created here)
    bool own_stream = false;
    hipEvent_t ev[64];                 // ring of "slice ready" events (a wait refers to the record that precedes it at enqueue time)
    unsigned next = 0;
    hipEvent_t done = nullptr;
};
static std::mutex g_comm_mu;
static std::vector<Comm*> g_comms;     // handle = index + 1

static Comm* comm_of(int64_t h) {
    std::lock_guard<std::mutex> lk(g_comm_mu);
    return (h >= 1 && h <= (int64_t)g_comms.size()) ? g_comms[h - 1] : nullptr;
}

static int nccl_type(int dtype) { return dtype == XL_F32 ? NCCL_FLOAT32 : dtype == XL_BF16 ? NCCL_BFLOAT16 : -1; }

#define XL_NCCL(call, what)                                                                             \
    do {                                                                                                \
        const int rc__ = (call);                                                                        \
        if (rc__ != 0) { set_error("%s: RCCL error %d: %s", what, rc__, r->GetErrorString(rc__)); return XL_ERR_RCCL; } \
    } while (0)

// the collective stream continues after everything queued on `after` so far
static int order_after(Comm* c, hipStream_t after, const char* what) {
    hipEvent_t e = c->ev[c->next++ & 63u];
    hipError_t he = hipEventRecord(e, after);
    if (he == hipSuccess) he = hipStreamWaitEvent(c->stream, e, 0);
    XL_CHECK_ARG(he == hipSuccess, XL_ERR_HIP, "%s: %s", what, hipGetErrorString(he));
    return XL_OK;
}

}  // namespace xl

using namespace xl;

extern "C" int xl_comm_unique_id(void* id128) {
    XL_CHECK_ARG(id128 != nullptr, XL_ERR_BAD_ARG, "xl_comm_unique_id: null buffer");
    const Rccl* r = rccl();
    if (r == nullptr) return XL_ERR_RCCL;
    XL_NCCL(r->GetUniqueId(reinterpret_cast<NcclId*>(id128)), "xl_comm_unique_id");
    return XL_OK;
}

extern "C" int64_t xl_comm_init(const void* id128, int rank, int nranks, void* comm_stream) {
    if (id128 == nullptr || nranks < 1 || rank < 0 || rank >= nranks) { set_error("xl_comm_init: bad arguments"); return 0; }
    const Rccl* r = rccl();
    if (r == nullptr) return 0;
    Comm* c = new Comm();
    c->rank = rank; c->nranks = nranks;
    NcclId id;
    memcpy(&id, id128, sizeof id);
    const int rc = r->CommInitRank(&c->comm, nranks, id, rank);
    if (rc != 0) { set_error("xl_comm_init: RCCL error %d: %s", rc, r->GetErrorString(rc)); delete c; return 0; }
    bool ok = true;
    for (int i = 0; i < 64; ++i) c->ev[i] = nullptr;
    if (comm_stream != nullptr) c->stream = (hipStream_t)comm_stream;
    else { ok = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess; c->own_stream = ok; }
    for (int i = 0; i < 64 && ok; ++i) ok = hipEventCreateWithFlags(&c->ev[i], hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&c->done, hipEventDisableTiming) == hipSuccess;
    if (!ok) {                     // give back everything made so far: the communicator, the events, an own stream, the object
        set_error("xl_comm_init: stream / event creation failed");
        (void)r->CommDestroy(c->comm);
        for (int i = 0; i < 64; ++i) if (c->ev[i] != nullptr) (void)hipEventDestroy(c->ev[i]);
        if (c->done != nullptr) (void)hipEventDestroy(c->done);
        if (c->own_stream) (void)hipStreamDestroy(c->stream);
        delete c;
        return 0;
    }
    std::lock_guard<std::mutex> lk(g_comm_mu);
    g_comms.push_back(c);
    return (int64_t)g_comms.size();
}

// number of ranks of the communicator AS RCCL REPORTS IT (ncclCommCount): what a scaling line should print next to its own idea of the
// world size; -1 if the handle is stale, the value given to xl_comm_init if the library lacks the query
extern "C" int xl_comm_nranks(int64_t comm) {
    Comm* c = comm_of(comm);
    if (c == nullptr) return -1;
    const Rccl* r = rccl();
    int n = c->nranks;
    if (r != nullptr && r->CommCount != nullptr && c->comm != nullptr) {
        int q = 0;
        if (r->CommCount(c->comm, &q) == 0) n = q;
    }
    return n;
}

extern "C" int xl_comm_destroy(int64_t comm) {
    Comm* c = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_comm_mu);
        if (comm >= 1 && comm <= (int64_t)g_comms.size()) { c = g_comms[comm - 1]; g_comms[comm - 1] = nullptr; }
    }
    if (c == nullptr) return XL_OK;
    const Rccl* r = rccl();
    (void)hipStreamSynchronize(c->stream);
    if (r != nullptr && c->comm != nullptr) (void)r->CommDestroy(c->comm);
    for (int i = 0; i < 64; ++i) (void)hipEventDestroy(c->ev[i]);
    (void)hipEventDestroy(c->done);
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return XL_OK;
}

extern "C" int xl_comm_allreduce(int64_t comm, void* buf, int64_t count, int dtype, void* after_stream) {
    Comm* c = comm_of(comm);
    const Rccl* r = rccl();
    XL_CHECK_ARG(c != nullptr && r != nullptr, XL_ERR_BAD_ARG, "xl_comm_allreduce: unknown communicator");
    XL_CHECK_ARG(buf != nullptr && count > 0 && nccl_type(dtype) >= 0, XL_ERR_BAD_ARG, "xl_comm_allreduce: bad arguments");
    int rc = order_after(c, (hipStream_t)after_stream, "xl_comm_allreduce");
    if (rc) return rc;
    static const bool skip1 = getenv("XL_COMM_SKIP_SINGLE") != nullptr;       // diagnostic: stream plumbing without the RCCL launch
    if (skip1 && c->nranks == 1) return XL_OK;
    XL_NCCL(r->AllReduce(buf, buf, (size_t)count, nccl_type(dtype), NCCL_SUM, c->comm, c->stream), "xl_comm_allreduce");
    return XL_OK;
}

extern "C" int xl_comm_reduce_scatter(int64_t comm, const void* send, void* recv, int64_t recv_count, int dtype, void* after_stream) {
    Comm* c = comm_of(comm);
    const Rccl* r = rccl();
    XL_CHECK_ARG(c != nullptr && r != nullptr, XL_ERR_BAD_ARG, "xl_comm_reduce_scatter: unknown communicator");
    XL_CHECK_ARG(send && recv && recv_count > 0 && nccl_type(dtype) >= 0, XL_ERR_BAD_ARG, "xl_comm_reduce_scatter: bad arguments");
    int rc = order_after(c, (hipStream_t)after_stream, "xl_comm_reduce_scatter");
    if (rc) return rc;
    XL_NCCL(r->ReduceScatter(send, recv, (size_t)recv_count, nccl_type(dtype), NCCL_SUM, c->comm, c->stream), "xl_comm_reduce_scatter");
    return XL_OK;
}

extern "C" int xl_comm_allgather(int64_t comm, const void* send, void* recv, int64_t send_count, int dtype, void* after_stream) {
    Comm* c = comm_of(comm);
    const Rccl* r = rccl();
    XL_CHECK_ARG(c != nullptr && r != nullptr, XL_ERR_BAD_ARG, "xl_comm_allgather: unknown communicator");
    XL_CHECK_ARG(send && recv && send_count > 0 && nccl_type(dtype) >= 0, XL_ERR_BAD_ARG, "xl_comm_allgather: bad arguments");
    int rc = order_after(c, (hipStream_t)after_stream, "xl_comm_allgather");
    if (rc) return rc;
    XL_NCCL(r->AllGather(send, recv, (size_t)send_count, nccl_type(dtype), c->comm, c->stream), "xl_comm_allgather");
    return XL_OK;
}

extern "C" int xl_comm_bcast(int64_t comm, void* buf, int64_t bytes, int root, void* after_stream) {
    Comm* c = comm_of(comm);
    const Rccl* r = rccl();
    XL_CHECK_ARG(c != nullptr && r != nullptr, XL_ERR_BAD_ARG, "xl_comm_bcast: unknown communicator");
    XL_CHECK_ARG(buf != nullptr && bytes > 0 && root >= 0 && root < c->nranks, XL_ERR_BAD_ARG, "xl_comm_bcast: bad arguments");
    int rc = order_after(c, (hipStream_t)after_stream, "xl_comm_bcast");
    if (rc) return rc;
    XL_NCCL(r->Broadcast(buf, buf, (size_t)bytes, NCCL_UINT8, root, c->comm, c->stream), "xl_comm_bcast");
    return XL_OK;
}

extern "C" int xl_comm_reduce(int64_t comm, void* buf, int64_t count, int dtype, int root, void* after_stream) {
    Comm* c = comm_of(comm);
    const Rccl* r = rccl();
    XL_CHECK_ARG(c != nullptr && r != nullptr, XL_ERR_BAD_ARG, "xl_comm_reduce: unknown communicator");
    XL_CHECK_ARG(buf != nullptr && count > 0 && nccl_type(dtype) >= 0 && root >= 0 && root < c->nranks, XL_ERR_BAD_ARG,
                 "xl_comm_reduce: bad arguments");
    int rc = order_after(c, (hipStream_t)after_stream, "xl_comm_reduce");
    if (rc) return rc;
    XL_NCCL(r->Reduce(buf, buf, (size_t)count, nccl_type(dtype), NCCL_SUM, root, c->comm, c->stream), "xl_comm_reduce");
    return XL_OK;
}

extern "C" int xl_comm_wait(int64_t comm, void* stream) {
    Comm* c = comm_of(comm);
    XL_CHECK_ARG(c != nullptr, XL_ERR_BAD_ARG, "xl_comm_wait: unknown communicator");
    hipError_t he = hipEventRecord(c->done, c->stream);
    if (he == hipSuccess) he = hipStreamWaitEvent((hipStream_t)stream, c->done, 0);
    XL_CHECK_ARG(he == hipSuccess, XL_ERR_HIP, "xl_comm_wait: %s", hipGetErrorString(he));
    return XL_OK;
}
