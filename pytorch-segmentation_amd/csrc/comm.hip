// Thin RCCL wrappers of the C ABI (comm_{init, allreduce_async, wait}) — the exchange steps of one-process-per-GPU
// data parallelism over xGMI, for hosts that do not go through torch.distributed:
//   gradient all-reduce (sum / average)            replaces nn.DataParallel's gather-to-GPU-0 + reduce_add (base/base_trainer.py:33-38)
//   all-gather of the SyncBN Welford partials      replaces torch.cuda.comm.reduce_add / broadcast_coalesced in
//                                                  utils/sync_batchnorm/batchnorm.py:117-126 and the master/slave pipes of comm.py:102-133
// A communicator owns a side HIP stream: *_async makes the side stream wait for the caller's stream (the buffer is complete),
// enqueues the collective there, and records an event; segmi_comm_wait makes any stream wait for that event — so a bucket's
// all-reduce overlaps whatever the compute stream does next (the rest of backward) without a host synchronisation.
// RCCL is bound at run time (dlopen of the librccl the process already carries — torch ships one — else the system's): libsegmi
// has no link-time dependency on it, and a single-GPU process never touches it.
#include "segmi_common.h"
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <cstring>
#include <mutex>

namespace {

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    bool ok = false;
};
Rccl g_rccl;
std::once_flag g_once;

void load_rccl() {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {                       // a copy that is already mapped (torch's) wins: one RCCL per process
        g_rccl.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        if (g_rccl.lib) break;
    }
    for (int i = 0; i < 3 && !g_rccl.lib; ++i) g_rccl.lib = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
    if (!g_rccl.lib) return;
#define SEGMI_SYM(field, name) g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(g_rccl.lib, name))
    SEGMI_SYM(GetUniqueId, "ncclGetUniqueId");
    SEGMI_SYM(CommInitRank, "ncclCommInitRank");
    SEGMI_SYM(CommDestroy, "ncclCommDestroy");
    SEGMI_SYM(AllReduce, "ncclAllReduce");
    SEGMI_SYM(AllGather, "ncclAllGather");
#undef SEGMI_SYM
    g_rccl.ok = g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.CommDestroy && g_rccl.AllReduce && g_rccl.AllGather;
}
bool rccl() {
    std::call_once(g_once, load_rccl);
    return g_rccl.ok;
}

}  // namespace

struct segmi_comm {
    ncclComm_t comm;
    hipStream_t side;
    hipEvent_t ready, done;
    int world, rank;
};

extern "C" {

int segmi_comm_available(void) { return rccl() ? 1 : 0; }

int segmi_comm_unique_id_bytes(void) { return (int)sizeof(ncclUniqueId); }

int segmi_comm_get_unique_id(void* id_out, size_t bytes) {
    if (!id_out || bytes < sizeof(ncclUniqueId)) return SEGMI_ERR_BADARG;
    if (!rccl()) return SEGMI_ERR_LAUNCH;
    ncclUniqueId id;
    if (g_rccl.GetUniqueId(&id) != ncclSuccess) return SEGMI_ERR_LAUNCH;
    memcpy(id_out, &id, sizeof(id));
    return SEGMI_OK;
}

int segmi_comm_init(segmi_comm** out, int world, int rank, const void* unique_id, size_t bytes) {
    if (!out || world < 1 || rank < 0 || rank >= world || !unique_id || bytes < sizeof(ncclUniqueId)) return SEGMI_ERR_BADARG;
    if (!rccl()) return SEGMI_ERR_LAUNCH;
    segmi_comm* c = new segmi_comm();
    c->world = world; c->rank = rank;
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    if (g_rccl.CommInitRank(&c->comm, world, id, rank) != ncclSuccess) { delete c; return SEGMI_ERR_LAUNCH; }
    if (hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->ready, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->done, hipEventDisableTiming) != hipSuccess) {
        g_rccl.CommDestroy(c->comm);
        delete c;
        return SEGMI_ERR_LAUNCH;
    }
    *out = c;
    return SEGMI_OK;
}

int segmi_comm_world(const segmi_comm* c) { return c ? c->world : 0; }

static int begin(segmi_comm* c, hipStream_t producer) {
    if (hipEventRecord(c->ready, producer) != hipSuccess) return SEGMI_ERR_LAUNCH;          // the buffer is complete on the caller's stream
    return hipStreamWaitEvent(c->side, c->ready, 0) == hipSuccess ? SEGMI_OK : SEGMI_ERR_LAUNCH;
}

int segmi_comm_allreduce_async(segmi_comm* c, const float* send, float* recv, size_t count, int average, segmi_stream_t stream) {
    if (!c || !send || !recv || count == 0) return SEGMI_ERR_BADARG;
    int rc = begin(c, (hipStream_t)stream);
    if (rc != SEGMI_OK) return rc;
    if (g_rccl.AllReduce(send, recv, count, ncclFloat, average ? ncclAvg : ncclSum, c->comm, c->side) != ncclSuccess) return SEGMI_ERR_LAUNCH;
    return hipEventRecord(c->done, c->side) == hipSuccess ? SEGMI_OK : SEGMI_ERR_LAUNCH;
}

int segmi_comm_allgather_async(segmi_comm* c, const float* send, float* recv, size_t count_per_rank, segmi_stream_t stream) {
    if (!c || !send || !recv || count_per_rank == 0) return SEGMI_ERR_BADARG;
    int rc = begin(c, (hipStream_t)stream);
    if (rc != SEGMI_OK) return rc;
    if (g_rccl.AllGather(send, recv, count_per_rank, ncclFloat, c->comm, c->side) != ncclSuccess) return SEGMI_ERR_LAUNCH;
    return hipEventRecord(c->done, c->side) == hipSuccess ? SEGMI_OK : SEGMI_ERR_LAUNCH;
}

int segmi_comm_wait(segmi_comm* c, segmi_stream_t stream) {
    if (!c) return SEGMI_ERR_BADARG;
    return hipStreamWaitEvent((hipStream_t)stream, c->done, 0) == hipSuccess ? SEGMI_OK : SEGMI_ERR_LAUNCH;
}

int segmi_comm_destroy(segmi_comm* c) {
    if (!c) return SEGMI_ERR_BADARG;
    hipStreamSynchronize(c->side);
    g_rccl.CommDestroy(c->comm);
    hipEventDestroy(c->ready);
    hipEventDestroy(c->done);
    hipStreamDestroy(c->side);
    delete c;
    return SEGMI_OK;
}

}  // extern "C"
