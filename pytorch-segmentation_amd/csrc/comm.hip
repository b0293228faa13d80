// Thin RCCL wrappers of the C ABI (comm_{init, allreduce_async, wait}) — the exchange steps of one-process-per-GPU
// data parallelism over xGMI, for hosts that do not go through torch.distributed:
//   gradient all-reduce (sum / average)            replaces nn.DataParallel's gather-to-GPU-0 + reduce_add (base/base_trainer.py:33-38)
//   all-gather of the SyncBN Welford partials      replaces torch.cuda.comm.reduce_add / broadcast_coalesced in
//                                                  utils/sync_batchnorm/batchnorm.py:117-126 and the master/slave pipes of comm.py:102-133
// A communicator owns a side HIP stream: *_async makes the side stream wait for the caller's stream (the buffer is complete),
// enqueues the collective there, records an event OF ITS OWN (a ring of SEGMI_COMM_EVENTS events) and hands back its ticket;
// segmi_comm_wait_ticket makes any stream wait for exactly that collective — bucket i's optimizer step starts while buckets
// i+1... are still on the wire — and segmi_comm_wait for the latest one.  No host synchronisation anywhere.
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

constexpr int SEGMI_COMM_EVENTS = 64;     // a ticket older than the ring waits for the collective that re-used its event: a LATER one
                                           // on the same in-order side stream, so the wait is still sufficient
struct segmi_comm {
    ncclComm_t comm;
    hipStream_t side;
    hipEvent_t ready, done[SEGMI_COMM_EVENTS];
    long next;                             // ticket of the next collective (tickets start at 1; 0 = nothing enqueued yet)
    int world, rank, device;
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
    c->next = 1;
    bool ok = hipGetDevice(&c->device) == hipSuccess && hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) == hipSuccess &&
              hipEventCreateWithFlags(&c->ready, hipEventDisableTiming) == hipSuccess;
    for (int i = 0; ok && i < SEGMI_COMM_EVENTS; ++i) ok = hipEventCreateWithFlags(&c->done[i], hipEventDisableTiming) == hipSuccess;
    if (!ok) {
        g_rccl.CommDestroy(c->comm);
        delete c;
        return SEGMI_ERR_LAUNCH;
    }
    *out = c;
    return SEGMI_OK;
}

int segmi_comm_world(const segmi_comm* c) { return c ? c->world : 0; }

// the communicator lives on the device it was created on: a call from a thread whose current device is another one would enqueue
// on the wrong device's streams
static bool on_device(const segmi_comm* c) {
    int dev = -1;
    return hipGetDevice(&dev) == hipSuccess && dev == c->device;
}
static int begin(segmi_comm* c, hipStream_t producer) {
    if (!on_device(c)) return SEGMI_ERR_BADARG;
    if (hipEventRecord(c->ready, producer) != hipSuccess) return SEGMI_ERR_LAUNCH;          // the buffer is complete on the caller's stream
    return hipStreamWaitEvent(c->side, c->ready, 0) == hipSuccess ? SEGMI_OK : SEGMI_ERR_LAUNCH;
}
static int finish(segmi_comm* c, long* ticket_out) {
    const long ticket = c->next++;
    if (hipEventRecord(c->done[ticket % SEGMI_COMM_EVENTS], c->side) != hipSuccess) return SEGMI_ERR_LAUNCH;
    if (ticket_out) *ticket_out = ticket;
    return SEGMI_OK;
}

int segmi_comm_allreduce_async(segmi_comm* c, const float* send, float* recv, size_t count, int average, long* ticket_out,
                               segmi_stream_t stream) {
    if (!c || !send || !recv || count == 0) return SEGMI_ERR_BADARG;
    int rc = begin(c, (hipStream_t)stream);
    if (rc != SEGMI_OK) return rc;
    if (g_rccl.AllReduce(send, recv, count, ncclFloat, average ? ncclAvg : ncclSum, c->comm, c->side) != ncclSuccess) return SEGMI_ERR_LAUNCH;
    return finish(c, ticket_out);
}

int segmi_comm_allgather_async(segmi_comm* c, const float* send, float* recv, size_t count_per_rank, long* ticket_out,
                               segmi_stream_t stream) {
    if (!c || !send || !recv || count_per_rank == 0) return SEGMI_ERR_BADARG;
    int rc = begin(c, (hipStream_t)stream);
    if (rc != SEGMI_OK) return rc;
    if (g_rccl.AllGather(send, recv, count_per_rank, ncclFloat, c->comm, c->side) != ncclSuccess) return SEGMI_ERR_LAUNCH;
    return finish(c, ticket_out);
}

int segmi_comm_wait_ticket(segmi_comm* c, long ticket, segmi_stream_t stream) {
    if (!c || ticket < 1 || ticket >= c->next || !on_device(c)) return SEGMI_ERR_BADARG;
    return hipStreamWaitEvent((hipStream_t)stream, c->done[ticket % SEGMI_COMM_EVENTS], 0) == hipSuccess ? SEGMI_OK : SEGMI_ERR_LAUNCH;
}

int segmi_comm_wait(segmi_comm* c, segmi_stream_t stream) {
    if (!c) return SEGMI_ERR_BADARG;
    if (c->next == 1) return SEGMI_OK;                 // nothing was ever enqueued
    return segmi_comm_wait_ticket(c, c->next - 1, stream);
}

int segmi_comm_destroy(segmi_comm* c) {
    if (!c) return SEGMI_ERR_BADARG;
    hipStreamSynchronize(c->side);
    g_rccl.CommDestroy(c->comm);
    hipEventDestroy(c->ready);
    for (int i = 0; i < SEGMI_COMM_EVENTS; ++i) hipEventDestroy(c->done[i]);
    hipStreamDestroy(c->side);
    delete c;
    return SEGMI_OK;
}

}  // extern "C"
