// RCCL gather of per-GPU feature blocks to one rank (north_star: "RCCL gather over xGMI of the
// resulting feature matrices").  A gather is 7 independent peer->root transfers, each on its own
// xGMI link; it is issued as one grouped ncclSend/ncclRecv batch on the library stream.
// librccl.so is opened lazily so that hosts without a GPU can still load libpaa_hip.so.
#pragma once
#include <dirent.h>
#include <fcntl.h>
#include <signal.h>
#include <sys/stat.h>
#include <unistd.h>

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

// ---- "one process per GPU" guard ---------------------------------------------------------------------------
// Two ranks of one communicator on the SAME physical device make ncclCommInitRank wait forever on this stack (measured:
// no "duplicate GPU" error, a hang).  world_size says nothing about that (multi-node jobs, launchers that pin one visible
// device per rank), so the check is on the device itself: before RCCL is called every rank drops a marker file
//   <tmp>/paa_comm_<hash of the unique id>_<PCI bus id>.<rank>     (content: its pid)
// and refuses to go on when a marker of ANOTHER live rank of the same job sits on the same bus id.  Node-local by
// construction (that is where two processes can share a device); markers are removed by paa_comm_destroy and ignored when
// their pid is gone.  Python callers with an all-gather on their control plane check the bus ids of all ranks up front
// (distributed.RcclGather), which fails on every rank at once; this is the backstop for plain C callers.
std::string g_comm_marker;
unsigned long long comm_id_hash(const void *id_bytes, size_t n) {
    unsigned long long h = 1469598103934665603ULL;                       // FNV-1a
    for (size_t i = 0; i < n; ++i) { h ^= ((const unsigned char *)id_bytes)[i]; h *= 1099511628211ULL; }
    return h;
}
const char *comm_tmp_dir() {
    const char *d = getenv("PAA_COMM_MARKER_DIR");
    if (d && d[0]) return d;
    d = getenv("TMPDIR");
    return (d && d[0]) ? d : "/tmp";
}
// marker stem of (job, device) and the directory scan: 0 when no other live rank of this job uses the device, the other
// rank's number + 1 otherwise
std::string comm_marker_stem(const void *id_bytes, const char *bus_id) {
    char stem[256], host[64] = "";
    std::string bus(bus_id);
    for (char &ch : bus) if (ch == ':' || ch == '.' || ch == '/') ch = '-';
    // the host (and the pid namespace: two containers that share /tmp do not see each other's pids) is part of the name:
    // on a shared TMPDIR the markers of other nodes are other files, never probed with kill() and never unlinked
    if (gethostname(host, sizeof(host) - 1) != 0) host[0] = 0;
    host[sizeof(host) - 1] = 0;
    for (char *c = host; *c; ++c) if (*c == '/' || *c == '.') *c = '-';
    char pidns[64] = "";
    const ssize_t nl = readlink("/proc/self/ns/pid", pidns, sizeof(pidns) - 1);
    pidns[nl > 0 ? nl : 0] = 0;
    snprintf(stem, sizeof(stem), "paa_comm_%016llx_%s_%08llx_%s.", comm_id_hash(id_bytes, sizeof(ncclUniqueId)), host,
             comm_id_hash(pidns, strlen(pidns)) & 0xffffffffULL, bus.c_str());
    return stem;
}
int comm_scan_device(const std::string &stem, int rank) {
    const std::string dir = comm_tmp_dir();
    int clash = 0;
    if (DIR *dp = opendir(dir.c_str())) {
        while (struct dirent *de = readdir(dp)) {
            if (strncmp(de->d_name, stem.c_str(), stem.size()) != 0) continue;
            if (strstr(de->d_name + stem.size(), ".tmp.")) continue;       // another rank's marker in the making (see comm_claim_device)
            const int other = atoi(de->d_name + stem.size());
            if (other == rank) continue;
            long pid = 0;
            const std::string path = dir + "/" + de->d_name;
            if (FILE *f = fopen(path.c_str(), "r")) { if (fscanf(f, "%ld", &pid) != 1) pid = 0; fclose(f); }
            // markers appear complete (link() of a finished temp file), so an empty or unparsable one is not ours to judge:
            // it is neither a clash nor unlinked (unlinking a live rank's marker would hide the clash the guard exists for)
            if (pid <= 0) continue;
            if (kill((pid_t)pid, 0) == 0 || errno == EPERM) { clash = other + 1; break; }
            unlink(path.c_str());                                           // left behind by a process that is gone
        }
        closedir(dp);
    }
    return clash;
}
void comm_release_device() {
    if (!g_comm_marker.empty()) unlink(g_comm_marker.c_str());
    g_comm_marker.clear();
}

// drops this rank's marker, looks for others, and looks again after a short pause: ranks that a launcher starts together
// then ALL see the clash (a rank that arrives much later still fails alone, while the early one waits inside RCCL)
int comm_claim_device(const void *id_bytes, int rank, const char *bus_id) {
    const std::string stem = comm_marker_stem(id_bytes, bus_id);
    g_comm_marker = std::string(comm_tmp_dir()) + "/" + stem + std::to_string(rank);
    // (a world-writable directory: never follow a link someone planted, never truncate a file that is not ours)
    (void)unlink(g_comm_marker.c_str());                                  // a marker of an earlier run of this very rank
    // the pid is written to a private temp name first and the finished file is link()ed to the marker name: a concurrent
    // scan on another rank never reads a half-written marker (it would take pid 0 for "stale" -- advisor, round 4)
    const std::string tmp = g_comm_marker + ".tmp." + std::to_string((long)getpid());
    (void)unlink(tmp.c_str());
    const int fd = open(tmp.c_str(), O_CREAT | O_EXCL | O_NOFOLLOW | O_WRONLY, 0600);
    if (fd < 0) { g_comm_marker.clear(); return 0; }                     // no writable tmp directory: no check possible
    char line[32];
    const int len = snprintf(line, sizeof(line), "%ld\n", (long)getpid());
    const bool written = write(fd, line, (size_t)len) == len;
    close(fd);
    const bool linked = written && link(tmp.c_str(), g_comm_marker.c_str()) == 0;
    (void)unlink(tmp.c_str());
    if (!linked) { g_comm_marker.clear(); return 0; }
    static bool at_exit = false;
    if (!at_exit) { atexit(comm_release_device); at_exit = true; }
    int clash = comm_scan_device(stem, rank);
    if (!clash) {
        usleep(150 * 1000);
        clash = comm_scan_device(stem, rank);
    }
    return clash;
}
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

// file name (inside PAA_COMM_MARKER_DIR / TMPDIR) of the one-process-per-GPU marker of `rank` of the job `id_bytes` on the
// selected device: tests plant a foreign rank's marker with it instead of re-deriving the naming scheme
extern "C" int paa_debug_comm_marker_name(const void *id_bytes, int rank, char *out, int capacity) {
    if (!id_bytes || !out || capacity < 32 || rank < 0) return fail(PAA_ERR_ARG, "bad argument");
    int rc = ensure_init();
    if (rc) return rc;
    char bus[64] = "";
    if ((rc = paa_device_bus_id(bus, (int)sizeof(bus)))) return rc;
    const std::string name = comm_marker_stem(id_bytes, bus) + std::to_string(rank);
    if ((int)name.size() + 1 > capacity) return fail(PAA_ERR_ARG, "capacity %d < %zu", capacity, name.size() + 1);
    memcpy(out, name.c_str(), name.size() + 1);
    return PAA_OK;
}

extern "C" int paa_comm_init(int world_size, int rank, const void *id_bytes) {
    int rc = ensure_init();
    if (rc) return rc;
    if ((rc = rccl_load())) return rc;
    if (!id_bytes || world_size < 1 || rank < 0 || rank >= world_size) return fail(PAA_ERR_ARG, "bad comm arguments");
    if (g_comm) paa_comm_destroy();
    if (world_size > 1) {
        char bus[64] = "";
        if (paa_device_bus_id(bus, (int)sizeof(bus)) == PAA_OK) {
            const int clash = comm_claim_device(id_bytes, rank, bus);
            if (clash) {
                // (the marker stays until paa_comm_destroy / process exit: the other rank's second look must still find it)
                return fail(PAA_ERR_COMM, "paa_comm_init: ranks %d and %d of this job both use the device at PCI %s: one "
                            "process per GPU (ncclCommInitRank would never return)", rank, clash - 1, bus);
            }
        }
    }
    ncclUniqueId id;
    memcpy(&id, id_bytes, sizeof(id));
    {
        const ncclResult_t r_ = g_rccl.CommInitRank(&g_comm, world_size, id, rank);
        if (r_ != ncclSuccess) {
            comm_release_device();
            return fail(PAA_ERR_COMM, "ncclCommInitRank: %s", g_rccl.GetErrorString(r_));
        }
    }
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
    comm_release_device();
    g_world = 1;
    g_rank = 0;
    return PAA_OK;
}

// rank r's block lands at d_recv + displs[r] (doubles) on the root: a chunked gather can fill the final rank-major
// layout piece by piece while later chunks are still being computed
extern "C" int paa_comm_gatherv_f64(const double *d_send, const int64_t *counts, const int64_t *displs, int root,
                                    double *d_recv) {
    if (!counts || !displs) return fail(PAA_ERR_ARG, "null counts / displs");
    std::lock_guard<std::mutex> lk(g_mu);      // g_gather_done is shared with paa_plan_execute
    if (!g_comm) {
        if (g_world != 1) return fail(PAA_ERR_COMM, "communicator not initialised");
        if (d_recv && d_send && d_recv + displs[0] != d_send && counts[0] > 0)
            HIP_TRY(hipMemcpyAsync(d_recv + displs[0], d_send, (size_t)counts[0] * 8, hipMemcpyDeviceToDevice, g_main_stream));
        return PAA_OK;
    }
    if (root < 0 || root >= g_world) return fail(PAA_ERR_ARG, "bad root");
    if (g_rank == root && !d_recv) return fail(PAA_ERR_ARG, "root needs a receive buffer");
    // order the gather after everything queued so far on the compute stream
    HIP_TRY(hipEventRecord(g_ev_ready, g_main_stream));
    HIP_TRY(hipStreamWaitEvent(g_comm_stream, g_ev_ready, 0));
    NCCL_TRY(g_rccl.GroupStart());
    // a call that fails inside the group must not leave the group open (every later RCCL call of this thread would
    // be swallowed by it): remember the first error, close the group, then report
    ncclResult_t in_group = ncclSuccess;
    if (g_rank == root) {
        for (int r = 0; r < g_world && in_group == ncclSuccess; ++r)
            if (r != root && counts[r] > 0)
                in_group = g_rccl.Recv(d_recv + displs[r], (size_t)counts[r], ncclDouble, r, g_comm, g_comm_stream);
    } else if (counts[g_rank] > 0) {
        if (!d_send) in_group = ncclInvalidArgument;
        else in_group = g_rccl.Send(d_send, (size_t)counts[g_rank], ncclDouble, root, g_comm, g_comm_stream);
    }
    const ncclResult_t at_end = g_rccl.GroupEnd();
    if (in_group != ncclSuccess) return fail(PAA_ERR_COMM, "ncclSend / ncclRecv of the gather: %s", g_rccl.GetErrorString(in_group));
    if (at_end != ncclSuccess) return fail(PAA_ERR_COMM, "ncclGroupEnd of the gather: %s", g_rccl.GetErrorString(at_end));
    if (g_rank == root && d_send && counts[root] > 0 && d_recv + displs[root] != d_send)
        HIP_TRY(hipMemcpyAsync(d_recv + displs[root], d_send, (size_t)counts[root] * 8, hipMemcpyDeviceToDevice, g_comm_stream));
    if (d_send) {       // the next kernel that writes d_send must wait for this gather (see paa_plan_execute)
        hipEvent_t &ev = g_gather_done[d_send];
        if (!ev) HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(ev, g_comm_stream));
    }
    return PAA_OK;
}

extern "C" int paa_comm_gather_f64(const double *d_send, const int64_t *counts, int root, double *d_recv) {
    if (!counts) return fail(PAA_ERR_ARG, "null counts");
    std::vector<int64_t> displs((size_t)std::max(g_world, 1));
    long long off = 0;
    for (int r = 0; r < std::max(g_world, 1); ++r) { displs[r] = off; off += counts[r]; }
    return paa_comm_gatherv_f64(d_send, counts, displs.data(), root, d_recv);
}

// a freed buffer's events go with it (every chunk piece of extract_sharded is a fresh allocation; views into an allocation
// -- restart-file pieces at an offset -- are keyed by their own address, so every key inside the allocation goes).  The
// events are taken out of the map under g_mu and waited for WITHOUT it: a gather that waits for a slow or dead peer must not
// stall every other thread's plan_build / plan_execute / gather behind the lock (advisor, round 4).
static void comm_forget_buffer(const void *ptr) {
    std::vector<hipEvent_t> gone;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (g_gather_done.empty()) return;
        const char *lo = static_cast<const char *>(ptr), *hi = lo + 1;
        void *base = nullptr;
        size_t size = 0;
        if (hipMemGetAddressRange(&base, &size, const_cast<void *>(ptr)) == hipSuccess && base == ptr && size > 0) hi = lo + size;
        else (void)hipGetLastError();
        for (auto it = g_gather_done.begin(); it != g_gather_done.end();) {
            const char *k = static_cast<const char *>(it->first);
            if (k >= lo && k < hi) {
                if (it->second) gone.push_back(it->second);
                it = g_gather_done.erase(it);
            } else {
                ++it;
            }
        }
    }
    for (hipEvent_t ev : gone) { (void)hipEventSynchronize(ev); (void)hipEventDestroy(ev); }
}
static hipStream_t comm_stream_or_null() { return g_comm ? g_comm_stream : nullptr; }
// work queued on `s` (the communication stream) starts after everything queued so far on the compute stream (g_mu held)
static int comm_order_after_compute(hipStream_t s) {
    HIP_TRY(hipEventRecord(g_ev_ready, g_main_stream));
    HIP_TRY(hipStreamWaitEvent(s, g_ev_ready, 0));
    return PAA_OK;
}
// called by the kernels' entry points before they overwrite a caller's buffer (g_mu held)
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
