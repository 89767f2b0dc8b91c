// RCCL gather of per-GPU feature blocks to one rank (north_star: "RCCL gather over xGMI of the
// resulting feature matrices").  A gather is 7 independent peer->root transfers, each on its own
// xGMI link; it is issued as one grouped ncclSend/ncclRecv batch on the library stream.
// librccl.so is opened lazily so that hosts without a GPU can still load libpaa_hip.so.
#pragma once
#include <rccl/rccl.h>

namespace {
struct RcclApi {
    void *h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
RcclApi g_rccl;
ncclComm_t g_comm = nullptr;
int g_world = 1, g_rank = 0;
int *g_bar = nullptr;
// the gather runs on its own stream so that it overlaps the next batch's kernels: g_ev_ready orders it
// after the producing kernels, g_gather_done[send buffer] orders the next writer of that buffer after it
hipStream_t g_comm_stream = nullptr;
hipEvent_t g_ev_ready = nullptr;
std::map<const void *, hipEvent_t> g_gather_done;

int rccl_load() {
    if (g_rccl.h) return PAA_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names) { h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (h) break; }
    if (!h) return fail(PAA_ERR_COMM, "cannot dlopen librccl: %s", dlerror());
#define PAA_SYM(field, sym)                                                      \
    *(void **)(&g_rccl.field) = dlsym(h, sym);                                   \
    if (!g_rccl.field) return fail(PAA_ERR_COMM, "librccl lacks %s", sym);
    PAA_SYM(GetUniqueId, "ncclGetUniqueId")
    PAA_SYM(CommInitRank, "ncclCommInitRank")
    PAA_SYM(CommDestroy, "ncclCommDestroy")
    PAA_SYM(Send, "ncclSend")
    PAA_SYM(Recv, "ncclRecv")
    PAA_SYM(GroupStart, "ncclGroupStart")
    PAA_SYM(GroupEnd, "ncclGroupEnd")
    PAA_SYM(AllReduce, "ncclAllReduce")
    PAA_SYM(GetErrorString, "ncclGetErrorString")
#undef PAA_SYM
    g_rccl.h = h;
    return PAA_OK;
}
}  // namespace

#define NCCL_TRY(expr)                                                                       \
    do {                                                                                     \
        ncclResult_t r_ = (expr);                                                            \
        if (r_ != ncclSuccess) return fail(PAA_ERR_COMM, "%s: %s", #expr, g_rccl.GetErrorString(r_)); \
    } while (0)

static_assert(sizeof(ncclUniqueId) <= PAA_COMM_ID_BYTES, "unique id does not fit");

extern "C" int paa_comm_unique_id(void *id_out) {
    int rc = rccl_load();
    if (rc) return rc;
    if (!id_out) return fail(PAA_ERR_ARG, "null id");
    ncclUniqueId id;
    NCCL_TRY(g_rccl.GetUniqueId(&id));
    memset(id_out, 0, PAA_COMM_ID_BYTES);
    memcpy(id_out, &id, sizeof(id));
    return PAA_OK;
}

extern "C" int paa_comm_init(int world_size, int rank, const void *id_bytes) {
    int rc = ensure_init();
    if (rc) return rc;
    if ((rc = rccl_load())) return rc;
    if (!id_bytes || world_size < 1 || rank < 0 || rank >= world_size) return fail(PAA_ERR_ARG, "bad comm arguments");
    {
        // One process per GPU.  Two ranks on ONE device make ncclCommInitRank wait forever on this stack (measured: no
        // "duplicate GPU" error, a hang), so the case is refused before RCCL is called.  Ranks that are each restricted
        // to their own single device (HIP_VISIBLE_DEVICES per rank) opt in with PAA_COMM_SHARED_DEVICES=1.
        int n_dev = 0;
        const char *opt = getenv("PAA_COMM_SHARED_DEVICES");
        if (hipGetDeviceCount(&n_dev) == hipSuccess && world_size > n_dev && !(opt && opt[0] == '1'))
            return fail(PAA_ERR_COMM, "paa_comm_init: %d ranks but only %d visible device(s): one process per GPU "
                        "(set PAA_COMM_SHARED_DEVICES=1 if every rank sees only its own device)", world_size, n_dev);
    }
    if (g_comm) paa_comm_destroy();
    ncclUniqueId id;
    memcpy(&id, id_bytes, sizeof(id));
    NCCL_TRY(g_rccl.CommInitRank(&g_comm, world_size, id, rank));
    g_world = world_size;
    g_rank = rank;
    HIP_TRY(hipMalloc((void **)&g_bar, sizeof(int)));
    HIP_TRY(hipMemset(g_bar, 0, sizeof(int)));
    HIP_TRY(hipStreamCreateWithFlags(&g_comm_stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&g_ev_ready, hipEventDisableTiming));
    return PAA_OK;
}

extern "C" int paa_comm_destroy(void) {
    if (g_comm) {
        if (g_main_stream) (void)hipStreamSynchronize(g_main_stream);
        if (g_comm_stream) (void)hipStreamSynchronize(g_comm_stream);
        g_rccl.CommDestroy(g_comm);
        g_comm = nullptr;
    }
    for (auto &kv : g_gather_done) (void)hipEventDestroy(kv.second);
    g_gather_done.clear();
    if (g_ev_ready) { (void)hipEventDestroy(g_ev_ready); g_ev_ready = nullptr; }
    if (g_comm_stream) { (void)hipStreamDestroy(g_comm_stream); g_comm_stream = nullptr; }
    if (g_bar) { (void)hipFree(g_bar); g_bar = nullptr; }
    g_world = 1;
    g_rank = 0;
    return PAA_OK;
}

extern "C" int paa_comm_gather_f64(const double *d_send, const int64_t *counts, int root, double *d_recv) {
    if (!counts) return fail(PAA_ERR_ARG, "null counts");
    std::lock_guard<std::mutex> lk(g_mu);      // g_gather_done is shared with paa_plan_execute
    if (!g_comm) {
        if (g_world != 1) return fail(PAA_ERR_COMM, "communicator not initialised");
        if (d_recv && d_send && d_recv != d_send)
            HIP_TRY(hipMemcpyAsync(d_recv, d_send, (size_t)counts[0] * 8, hipMemcpyDeviceToDevice, g_main_stream));
        return PAA_OK;
    }
    // order the gather after everything queued so far on the compute stream
    HIP_TRY(hipEventRecord(g_ev_ready, g_main_stream));
    HIP_TRY(hipStreamWaitEvent(g_comm_stream, g_ev_ready, 0));
    NCCL_TRY(g_rccl.GroupStart());
    if (g_rank == root) {
        long long off = 0;
        for (int r = 0; r < g_world; ++r) {
            if (r != root && counts[r] > 0)
                NCCL_TRY(g_rccl.Recv(d_recv + off, (size_t)counts[r], ncclDouble, r, g_comm, g_comm_stream));
            off += counts[r];
        }
    } else if (counts[g_rank] > 0) {
        NCCL_TRY(g_rccl.Send(d_send, (size_t)counts[g_rank], ncclDouble, root, g_comm, g_comm_stream));
    }
    NCCL_TRY(g_rccl.GroupEnd());
    if (g_rank == root && d_send && counts[root] > 0) {
        long long off = 0;
        for (int r = 0; r < root; ++r) off += counts[r];
        if (d_recv + off != d_send)
            HIP_TRY(hipMemcpyAsync(d_recv + off, d_send, (size_t)counts[root] * 8, hipMemcpyDeviceToDevice, g_comm_stream));
    }
    if (d_send) {       // the next kernel that writes d_send must wait for this gather (see paa_plan_execute)
        hipEvent_t &ev = g_gather_done[d_send];
        if (!ev) HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(ev, g_comm_stream));
    }
    return PAA_OK;
}

// called by paa_plan_execute before it overwrites d_out
static int comm_wait_buffer_free(const void *d_out) {
    auto it = g_gather_done.find(d_out);
    if (it != g_gather_done.end() && it->second) HIP_TRY(hipStreamWaitEvent(g_main_stream, it->second, 0));
    return PAA_OK;
}
static int comm_sync() {
    if (g_comm_stream) HIP_TRY(hipStreamSynchronize(g_comm_stream));
    return PAA_OK;
}

extern "C" int paa_comm_barrier(void) {
    if (!g_comm) return PAA_OK;
    HIP_TRY(hipStreamSynchronize(g_main_stream));
    NCCL_TRY(g_rccl.AllReduce(g_bar, g_bar, 1, ncclInt, ncclSum, g_comm, g_comm_stream));
    HIP_TRY(hipStreamSynchronize(g_comm_stream));
    return PAA_OK;
}
