// libpaa_hip.so -- C ABI (include/paa_hip.h) over the gfx950 kernels.
// Host side: table cache, plans (tile lists, clip descriptors), scratch buffers, stream.
// There is deliberately NO CPU compute path in this library.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <condition_variable>
#include <mutex>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

#include "../../include/paa_hip.h"
// the kernels of the feature families are instantiated in their own translation units (family_*.hip); this unit sees their
// host-side layout / selection code and the launch entry points of family_launch.hpp
#define PAA_NO_HOST_LAUNCHERS
#include "family_launch.hpp"
#include "kernels_aux.hpp"
#include "kernels_big.hpp"
#include "kernels_wg.hpp"
#include "kernels_wgr.hpp"
#include "kernels_wgs.hpp"
#include "kernels_sim.hpp"
#include "kernels_svm.hpp"
#include "tables.hpp"

using namespace paa;

// ------------------------------------------------------------------------------------------
// error handling
// ------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(e_ == hipErrorOutOfMemory ? PAA_ERR_OOM : PAA_ERR_HIP, "%s: %s (%s:%d)", \
                        #expr, hipGetErrorString(e_), __FILE__, __LINE__);                    \
    } while (0)

// ------------------------------------------------------------------------------------------
// global state (one process drives one GPU)
// ------------------------------------------------------------------------------------------
struct TableSet {
    double fs;
    int window;
    FftPlan fft;
    MelTable mel;
    ChromaTable chroma;
    // device copies
    double2 *d_tw = nullptr, *d_post = nullptr;
    int *d_mel_lo = nullptr, *d_mel_cnt = nullptr, *d_mel_off = nullptr;
    double *d_mel_w = nullptr, *d_dct = nullptr;
    int *d_ch_start = nullptr, *d_ch_src = nullptr;
    double *d_ch_w = nullptr;
    FastTables fast;        // extra tables of the specialised kernels (may be empty)
    // kernel choice per (mode, rows, fast-kernel step): the family that took the shape, its layout and its table blob on the
    // device -- a host-buffer call builds a plan per call and must not rebuild / re-upload 20-40 KB of tables each time
    std::map<std::tuple<int, int, int>, std::shared_ptr<struct FamilyChoice>> choices;
};

struct Scratch {
    void *p = nullptr;
    size_t cap = 0;
};

static std::mutex g_mu;
static std::mutex g_api_mu;     // the self-similarity entry points share their scratch buffers: one call at a time
static std::atomic<int> g_device{-1};
// paa_init / paa_shutdown / the implicit first-call initialisation are serialised by this mutex (recursive: a device switch
// inside paa_init shuts the old device down).  A live lane stream is never overwritten: streams are created only while
// g_device < 0 and destroyed only by paa_shutdown.
static std::recursive_mutex g_init_mu;
static hipStream_t g_main_stream = nullptr;     // plan API, RCCL ordering, everything not running in a lane
static hipEvent_t g_ev0 = nullptr, g_ev1 = nullptr;
static std::map<std::pair<double, int>, std::unique_ptr<TableSet>> g_tables;
// Host-buffer entry points (NumPy in -> NumPy out) run in LANES: each lane has its own stream and scratch buffers, so
// calls from several host threads overlap on the device and on both PCIe directions instead of queueing behind one
// mutex.  A thread holds a lane for the duration of one call; every launch / copy of that call goes to cs().
constexpr int kCopyRanges = 4;        // frame ranges of one long clip: range k is copied back while range k + 1 computes
struct Lane {
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr;        // D2H of finished frame ranges (run_host_st)
    hipEvent_t range_done[kCopyRanges] = {nullptr, nullptr, nullptr, nullptr};
    Scratch in, out, mid;
    bool busy = false;
};
constexpr int kLanes = 4;
static Lane g_lanes[kLanes];
static std::mutex g_lane_mu;
static std::condition_variable g_lane_cv;
static int g_lanes_active = 0, g_lanes_peak = 0;
static thread_local Lane *tl_lane = nullptr;
static inline hipStream_t cs() { return tl_lane ? tl_lane->stream : g_main_stream; }
struct LaneGuard {
    Lane *l = nullptr;
    bool nested = false;
    LaneGuard() {
        if (tl_lane) { l = tl_lane; nested = true; return; }
        std::unique_lock<std::mutex> lk(g_lane_mu);
        for (;;) {
            for (int i = 0; i < kLanes; ++i)
                if (!g_lanes[i].busy && g_lanes[i].stream) { l = &g_lanes[i]; break; }
            if (l) break;
            g_lane_cv.wait(lk);
        }
        l->busy = true;
        g_lanes_peak = std::max(g_lanes_peak, ++g_lanes_active);
        tl_lane = l;
    }
    ~LaneGuard() {
        if (nested || !l) return;
        tl_lane = nullptr;
        {
            std::lock_guard<std::mutex> lk(g_lane_mu);
            l->busy = false;
            --g_lanes_active;
        }
        g_lane_cv.notify_one();
    }
};
static Scratch g_sim_z, g_sim_small, g_sim_cand, g_sim_in, g_sim_out, g_sim_filt;     // self-similarity row
static int g_force_generic = 0;
static int g_f800_waves = 8;          // PAA_F800_WAVES: waves per workgroup of the 800/400 kernel (4 or 8)
static int g_num_cu = 256;       // multiProcessorCount of the selected device (MI355X: 256)
// optional per-launch timing of the feature kernel (bench.py's roofline leg)
static int g_prof = 0;              // 0 = off, n = every n-th feature-kernel launch is bracketed by an event pair
static long long g_prof_seen = 0;
static std::vector<std::pair<hipEvent_t, hipEvent_t>> g_prof_ev;
static size_t g_prof_used = 0;
static double g_prof_ms = 0.0;
static long long g_prof_n = 0;

static int comm_wait_buffer_free(const void *d_out);
static void comm_forget_buffer(const void *ptr);
static int comm_sync();

// HIP's current device is a per-thread setting (device 0 until hipSetDevice): a host thread other than the one that
// called paa_init(d) would otherwise allocate on device 0 while the library's streams and tables live on device d.
static thread_local int tl_device = -1;
static int ensure_init() {
    if (g_device.load(std::memory_order_acquire) < 0) {
        // first call of the process: two threads may arrive here together; the second one finds the device selected
        std::lock_guard<std::recursive_mutex> lk(g_init_mu);
        if (g_device.load(std::memory_order_acquire) < 0) {
            const int rc = paa_init(0);
            if (rc) return rc;
        }
    }
    const int dev = g_device.load(std::memory_order_acquire);
    if (tl_device != dev) {
        HIP_TRY(hipSetDevice(dev));
        tl_device = dev;
    }
    return PAA_OK;
}

template <typename T>
static int upload(T **dst, const void *src, size_t count) {
    if (*dst) (void)hipFree(*dst);          // a retry after a failed build must not leak the earlier copy
    *dst = nullptr;
    if (count == 0) count = 1;
    HIP_TRY(hipMalloc((void **)dst, count * sizeof(T)));
    if (src) HIP_TRY(hipMemcpy(*dst, src, count * sizeof(T), hipMemcpyHostToDevice));
    return PAA_OK;
}

static int scratch_reserve(Scratch &s, size_t bytes) {
    if (bytes <= s.cap) return PAA_OK;
    if (s.p) { (void)hipFree(s.p); s.p = nullptr; s.cap = 0; }
    size_t want = bytes + bytes / 8 + 4096;
    HIP_TRY(hipMalloc(&s.p, want));
    s.cap = want;
    return PAA_OK;
}

// Plan-owned device arrays (clip descriptors, tiles, statistics partials ...) come from a small cache instead of
// hipMalloc / hipFree: the host-buffer entry points build and drop a plan per call, and every hipFree is a device-wide
// synchronisation that would serialise the lanes.  Blocks are rounded up to a power of two (>= 4 KB), kept up to 256 MB.
struct DevPool {
    std::mutex mu;
    std::multimap<size_t, void *> idle;
    std::map<void *, size_t> size_of;
    size_t idle_bytes = 0;
};
static DevPool g_pool;
static int pool_alloc(void **out, size_t bytes) {
    size_t cls = 4096;
    while (cls < bytes) cls <<= 1;
    {
        std::lock_guard<std::mutex> lk(g_pool.mu);
        auto it = g_pool.idle.find(cls);
        if (it != g_pool.idle.end()) {
            *out = it->second;
            g_pool.idle.erase(it);
            g_pool.idle_bytes -= cls;
            return PAA_OK;
        }
    }
    HIP_TRY(hipMalloc(out, cls));
    std::lock_guard<std::mutex> lk(g_pool.mu);
    g_pool.size_of[*out] = cls;
    return PAA_OK;
}
static void pool_free(void *p) {
    if (!p) return;
    std::unique_lock<std::mutex> lk(g_pool.mu);
    auto it = g_pool.size_of.find(p);
    if (it == g_pool.size_of.end()) { lk.unlock(); (void)hipFree(p); return; }
    if (g_pool.idle_bytes + it->second <= ((size_t)256 << 20)) {
        g_pool.idle.emplace(it->second, p);
        g_pool.idle_bytes += it->second;
        return;
    }
    g_pool.size_of.erase(it);
    lk.unlock();
    (void)hipFree(p);
}
static void pool_release_all() {
    std::lock_guard<std::mutex> lk(g_pool.mu);
    for (auto &kv : g_pool.idle) { g_pool.size_of.erase(kv.second); (void)hipFree(kv.second); }
    g_pool.idle.clear();
    g_pool.idle_bytes = 0;
}
// upload into a pooled block (plans); asynchronous on the calling thread's stream, ordered before the plan's kernels.
// The host source must stay valid until the copy has been issued from pageable memory (hipMemcpyAsync stages it).
template <typename T>
static int upload_pooled(T **dst, const void *src, size_t count) {
    pool_free(*dst);
    *dst = nullptr;
    if (count == 0) count = 1;
    int rc = pool_alloc((void **)dst, count * sizeof(T));
    if (rc) return rc;
    if (src) HIP_TRY(hipMemcpy(*dst, src, count * sizeof(T), hipMemcpyHostToDevice));
    return PAA_OK;
}

static void free_family_choices(TableSet &t);      // (lib_dispatch.hpp)
static void free_tables(TableSet &t) {
    free_family_choices(t);
    (void)hipFree(t.d_tw); (void)hipFree(t.d_post); (void)hipFree(t.d_mel_lo); (void)hipFree(t.d_mel_cnt);
    (void)hipFree(t.d_mel_off); (void)hipFree(t.d_mel_w); (void)hipFree(t.d_dct); (void)hipFree(t.d_ch_start);
    (void)hipFree(t.d_ch_src); (void)hipFree(t.d_ch_w);
    fast_tables_free(t.fast);
}

// builds (or fetches) the tables of one (fs, window); need_feat = 0 for the spectrogram
static int get_tables(double fs, int window, bool need_mel, bool need_chroma, TableSet **out) {
    auto key = std::make_pair(fs, window);
    auto it = g_tables.find(key);
    if (it == g_tables.end()) {
        std::unique_ptr<TableSet> t(new TableSet());
        t->fs = fs;
        t->window = window;
        build_fft_plan(window, t->fft);
        int rc;
        double dct[kNumMfcc * kNumMel];
        build_dct(dct);
        if ((rc = upload(&t->d_tw, t->fft.tw.data(), t->fft.tw.size() / 2)) ||
            (rc = upload(&t->d_post, t->fft.post.data(), t->fft.post.size() / 2)) ||
            (rc = upload(&t->d_dct, dct, (size_t)kNumMfcc * kNumMel))) {
            free_tables(*t);
            return rc;
        }
        it = g_tables.emplace(key, std::move(t)).first;
    }
    TableSet *t = it->second.get();
    const int nfft = window / 2;
    if (need_mel && !t->d_mel_w) {
        int rc = build_mel(fs, nfft, t->mel);
        if (rc == PAA_ERR_MEL_INDEX)
            return fail(rc, "mel filter bank indexes bin >= num_fft=%d at fs=%g (IndexError in the reference, "
                            "ShortTermFeatures.py:230-231)", nfft, fs);
        if ((rc = upload(&t->d_mel_lo, t->mel.lo.data(), t->mel.lo.size()))) return rc;
        if ((rc = upload(&t->d_mel_cnt, t->mel.cnt.data(), t->mel.cnt.size()))) return rc;
        if ((rc = upload(&t->d_mel_off, t->mel.off.data(), t->mel.off.size()))) return rc;
        if ((rc = upload(&t->d_mel_w, t->mel.w.data(), t->mel.w.size()))) return rc;
    }
    if (need_chroma && !t->d_ch_w) {
        int rc = build_chroma(fs, nfft, t->chroma);
        if (rc == PAA_ERR_CHROMA_VALUE)
            return fail(rc, "shape mismatch: chroma slot exceeds num_fft=%d (ValueError in the reference, "
                            "ShortTermFeatures.py:293)", nfft);
        if (rc == PAA_ERR_CHROMA_INDEX)
            return fail(rc, "chroma slot out of bounds for num_fft=%d (IndexError in the reference, "
                            "ShortTermFeatures.py:291)", nfft);
        if ((rc = upload(&t->d_ch_start, t->chroma.class_start, 13))) return rc;
        if ((rc = upload(&t->d_ch_src, t->chroma.src.data(), t->chroma.src.size()))) return rc;
        if ((rc = upload(&t->d_ch_w, t->chroma.w.data(), t->chroma.w.size()))) return rc;
    }
    *out = t;
    return PAA_OK;
}

#include "lib_plan.hpp"
#include "lib_similarity.hpp"

// ------------------------------------------------------------------------------------------
// library / device management
// ------------------------------------------------------------------------------------------
extern "C" const char *paa_version(void) { return "paa_hip 0.1 (gfx950)"; }
extern "C" const char *paa_last_error(void) { return g_err; }

extern "C" int paa_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return fail(PAA_ERR_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e));
    return n;
}

// PCI bus id ("0000:75:00.0") of the selected device: identifies the PHYSICAL device whatever HIP_VISIBLE_DEVICES says
extern "C" int paa_device_bus_id(char *out, int capacity) {
    if (!out || capacity < 16) return fail(PAA_ERR_ARG, "bus id buffer too small");
    int rc = ensure_init();
    if (rc) return rc;
    HIP_TRY(hipDeviceGetPCIBusId(out, capacity, g_device.load()));
    return PAA_OK;
}

extern "C" int paa_init(int device_id) {
    std::lock_guard<std::recursive_mutex> init_lock(g_init_mu);
    if (g_device == device_id && g_main_stream) {
        if (tl_device != device_id) { HIP_TRY(hipSetDevice(device_id)); tl_device = device_id; }
        return PAA_OK;
    }
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n < 1)
        return fail(PAA_ERR_HIP, "no HIP device (%s); this library has no CPU path", hipGetErrorString(e));
    if (device_id < 0 || device_id >= n) return fail(PAA_ERR_ARG, "device %d out of range (%d devices)", device_id, n);
    if (g_device >= 0 && g_device != device_id) {
        // live plans and caller-owned device buffers point into the current device: switching would leave them dangling
        if (g_live_plans.load() > 0)
            return fail(PAA_ERR_ARG, "paa_init(%d): %d plan(s) of device %d are still alive; destroy them first",
                        device_id, g_live_plans.load(), g_device.load());
        paa_shutdown();
    }
    HIP_TRY(hipSetDevice(device_id));
    tl_device = device_id;
    // (a failed earlier attempt may have left some of these behind: create only what is missing)
    if (!g_main_stream) HIP_TRY(hipStreamCreateWithFlags(&g_main_stream, hipStreamNonBlocking));
    for (int i = 0; i < kLanes; ++i) {
        if (!g_lanes[i].stream) HIP_TRY(hipStreamCreateWithFlags(&g_lanes[i].stream, hipStreamNonBlocking));
        if (!g_lanes[i].copy_stream) HIP_TRY(hipStreamCreateWithFlags(&g_lanes[i].copy_stream, hipStreamNonBlocking));
        for (int k = 0; k < kCopyRanges; ++k)
            if (!g_lanes[i].range_done[k]) HIP_TRY(hipEventCreateWithFlags(&g_lanes[i].range_done[k], hipEventDisableTiming));
    }
    if (!g_ev0) HIP_TRY(hipEventCreate(&g_ev0));
    if (!g_ev1) HIP_TRY(hipEventCreate(&g_ev1));
    {
        int ncu = 0;
        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device_id) == hipSuccess && ncu > 0) g_num_cu = ncu;
    }
    const char *fg = experiment_env("PAA_HIP_FORCE_GENERIC");
    g_force_generic = (fg && fg[0] == '1') ? 1 : 0;
    const char *fw = experiment_env("PAA_F800_WAVES");          // 4: one wave per SIMD, 8: two (A/B switch, default 8)
    g_f800_waves = (fw && fw[0] == '4') ? 4 : 8;
    g_device.store(device_id, std::memory_order_release);      // published last: ensure_init's fast path sees a complete state
    return PAA_OK;
}

extern "C" void paa_shutdown(void) {
    std::lock_guard<std::recursive_mutex> init_lock(g_init_mu);
    if (g_device < 0) return;
    (void)paa_comm_destroy();          // communicator, its stream and events
    if (g_main_stream) (void)hipStreamSynchronize(g_main_stream);
    for (auto &kv : g_tables) free_tables(*kv.second);
    g_tables.clear();
    pool_release_all();
    for (Scratch *s : {&g_sim_z, &g_sim_small, &g_sim_cand, &g_sim_in, &g_sim_out, &g_sim_filt}) { if (s->p) (void)hipFree(s->p); s->p = nullptr; s->cap = 0; }
    for (Lane &ln : g_lanes) {
        if (ln.stream) { (void)hipStreamSynchronize(ln.stream); (void)hipStreamDestroy(ln.stream); ln.stream = nullptr; }
        if (ln.copy_stream) { (void)hipStreamSynchronize(ln.copy_stream); (void)hipStreamDestroy(ln.copy_stream); ln.copy_stream = nullptr; }
        for (hipEvent_t &ev : ln.range_done) { if (ev) (void)hipEventDestroy(ev); ev = nullptr; }
        for (Scratch *s : {&ln.in, &ln.out, &ln.mid}) { if (s->p) (void)hipFree(s->p); s->p = nullptr; s->cap = 0; }
    }
    if (g_ev0) (void)hipEventDestroy(g_ev0);
    if (g_ev1) (void)hipEventDestroy(g_ev1);
    if (g_main_stream) (void)hipStreamDestroy(g_main_stream);
    g_ev0 = g_ev1 = nullptr;
    g_main_stream = nullptr;
    ++lds_attr_generation();                 // the next device has not seen any hipFuncSetAttribute
    g_device.store(-1, std::memory_order_release);
}

extern "C" int paa_dev_alloc(size_t bytes, void **out_ptr) {
    int rc = ensure_init();
    if (rc) return rc;
    if (!out_ptr) return fail(PAA_ERR_ARG, "null out_ptr");
    HIP_TRY(hipMalloc(out_ptr, bytes ? bytes : 1));
    return PAA_OK;
}
extern "C" int paa_dev_free(void *ptr) {
    if (ptr) {
        comm_forget_buffer(ptr);
        HIP_TRY(hipFree(ptr));
    }
    return PAA_OK;
}
extern "C" int paa_memcpy_d2d(void *dst, const void *src, size_t bytes) {
    int rc = ensure_init();
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, cs()));
    return PAA_OK;
}
extern "C" int paa_memcpy_h2d(void *dst, const void *src, size_t bytes) {
    int rc = ensure_init();
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, cs()));
    HIP_TRY(hipStreamSynchronize(cs()));
    return PAA_OK;
}
extern "C" int paa_memcpy_d2h(void *dst, const void *src, size_t bytes) {
    int rc = ensure_init();
    if (rc) return rc;
    if ((rc = comm_sync())) return rc;       // a gather into src may still run on the communication stream
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, cs()));
    HIP_TRY(hipStreamSynchronize(cs()));
    return PAA_OK;
}
// the same for a buffer that is only a SOURCE of queued gathers (or not gathered at all): ordered behind the kernels of the
// compute stream, not behind the communication stream -- a rank whose peer died in the exchange can still save its block
extern "C" int paa_memcpy_d2h_compute(void *dst, const void *src, size_t bytes) {
    int rc = ensure_init();
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, cs()));
    HIP_TRY(hipStreamSynchronize(cs()));
    return PAA_OK;
}
extern "C" int paa_dev_sync(void) {
    int rc = ensure_init();
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(cs()));
    return comm_sync();
}
extern "C" int paa_timer_start(void) {
    int rc = ensure_init();
    if (rc) return rc;
    HIP_TRY(hipEventRecord(g_ev0, cs()));
    return PAA_OK;
}
extern "C" int paa_timer_stop(float *ms) {
    int rc = ensure_init();
    if (rc) return rc;
    HIP_TRY(hipEventRecord(g_ev1, cs()));
    HIP_TRY(hipEventSynchronize(g_ev1));
    HIP_TRY(hipEventElapsedTime(ms, g_ev0, g_ev1));
    return PAA_OK;
}

extern "C" int paa_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_prof = on > 0 ? on : 0;
    g_prof_seen = 0;
    return PAA_OK;
}
// folds every recorded launch into (total ms, launches) and resets the recorder
extern "C" int paa_prof_read(double *total_ms, int64_t *launches) {
    int rc = ensure_init();
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(g_mu);
    HIP_TRY(hipStreamSynchronize(cs()));
    for (size_t i = 0; i < g_prof_used; ++i) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, g_prof_ev[i].first, g_prof_ev[i].second));
        g_prof_ms += ms;
        ++g_prof_n;
    }
    g_prof_used = 0;
    if (total_ms) *total_ms = g_prof_ms;
    if (launches) *launches = g_prof_n;
    g_prof_ms = 0.0;
    g_prof_n = 0;
    return PAA_OK;
}

// ------------------------------------------------------------------------------------------
// shape helpers
// ------------------------------------------------------------------------------------------
extern "C" int64_t paa_num_frames(int64_t n, int window, int step) {
    if (window < 1 || step < 1 || n < window) return 0;
    return (n - window) / step + 1;
}
extern "C" int64_t paa_num_mid_windows(int64_t T, int64_t mid_step_ratio) {
    if (T < 1 || mid_step_ratio < 1) return 0;
    return (T + mid_step_ratio - 1) / mid_step_ratio;
}
extern "C" int64_t paa_spectrogram_rows(int64_t n, int window, int step, int64_t *filled) {
    if (filled) *filled = 0;
    if (window < 1 || step < 1) return 0;
    // int((n - window) / step) + 1 with Python's true division then truncation toward zero (:413)
    const int64_t num = n - window;
    const int64_t rows = (num >= 0 ? num / step : -((-num) / step)) + 1;
    int64_t f = 0;
    for (int64_t p = window; p < n - window + 1; p += step) ++f;      // range(window, n - window + 1, step) :415
    if (filled) *filled = f;
    return rows;
}
extern "C" int64_t paa_chromagram_rows(int64_t n, int window, int step, int64_t *filled) {
    if (filled) *filled = 0;
    if (window < 1 || step < 1) return 0;
    const int64_t num = n - step - window;
    const int64_t rows = (num >= 0 ? num / step : -((-num) / step)) + 1;   // :347
    int64_t f = 0;
    for (int64_t p = window; p < n - step; p += step) ++f;                // range(window, n - step, step) :349
    if (filled) *filled = f;
    return rows;
}

#include "lib_host_api.hpp"

// ------------------------------------------------------------------------------------------
// RCCL gather (one process per GPU; librccl is loaded lazily so CPU-only hosts can load us)
// ------------------------------------------------------------------------------------------
#include "comm_rccl.hpp"

#include "lib_debug.hpp"
