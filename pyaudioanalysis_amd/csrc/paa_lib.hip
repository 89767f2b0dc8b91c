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
#include <utility>
#include <vector>

#include "../../include/paa_hip.h"
#include "kernels_aux.hpp"
#include "kernels_big.hpp"
#include "kernels_ct.hpp"
#include "kernels_fast.hpp"
#include "kernels_generic.hpp"
#include "kernels_mix.hpp"
#include "kernels_reg.hpp"
#include "kernels_sim.hpp"
#include "kernels_svm.hpp"
#include "kernels_tri.hpp"
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
struct Lane {
    hipStream_t stream = nullptr;
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

static void free_tables(TableSet &t) {
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

// ------------------------------------------------------------------------------------------
// plans
// ------------------------------------------------------------------------------------------
constexpr int kStatChunk = 65536;          // samples per statistics workgroup (upper bound, see stat_chunk_for)
// The statistics pass is an HBM-bound stream with every workgroup resident at once (8 per CU): 879 chunks of a one-hour
// clip put 4 workgroups on some CUs and 3 on others, and the pass lasts as long as the CUs with 4.  A batch of at least one
// chunk per CU is therefore cut into a whole multiple of num_cu chunks (1024 x 56 256 samples for the hour).
static int stat_chunk_for(long long total_samples, int num_cu) {
    const long long blocks = (total_samples + kStatChunk - 1) / kStatChunk;
    if (blocks < num_cu) return kStatChunk;
    const long long want = (blocks + num_cu - 1) / num_cu * num_cu;
    const long long len = ((total_samples + want - 1) / want + 63) / 64 * 64;      // multiples of 64 samples keep the 16-byte body aligned
    return (int)std::min<long long>(kStatChunk, std::max<long long>(len, 4096));
}

// Run length for the one-wave-per-run kernels.  A clip of T frames is cut into k = ceil(T / cap) runs of
// len = ceil(T / k) frames rounded up to the kernel's quantum (so no clip ends in a short leftover run); a workgroup takes
// wg_runs consecutive runs and the chip holds num_cu workgroups at a time, so a launch lasts about
// ceil(workgroups / num_cu) rounds of (longest run + halo) frames.  The cap that minimises that estimate is returned:
// one 1-hour clip -> 2000 runs of 72 frames (one round); 12 500 clips of 399 frames -> two runs of 200 per clip instead
// of 244 + 155 (the short run's wave idled for a third of its workgroup's life); 1000 clips of 1199 frames -> 6 x 200.
static int choose_run_cap(const std::vector<ClipDev> &clips, int quantum, int min_run, int max_run, int halo, int wg_runs,
                          int num_cu, int shrink = 0) {
    // shrink: frames by which every run but a clip's first is shorter (kernels whose halo rides inside the first iteration:
    // the tile list gives those runs len - shrink frames, so a clip has more runs than T / len)
    std::map<long long, long long> hist;                       // frames per clip -> number of such clips
    for (const ClipDev &c : clips)
        if (c.T > 0) ++hist[c.T];
    if (hist.empty()) return max_run;
    long long best_cost = -1;
    int best = max_run;
    for (int cap = max_run / quantum * quantum; cap >= min_run; cap -= quantum) {
        long long runs = 0, longest = 0;
        for (const auto &kv : hist) {
            const long long k = (kv.first + cap - 1) / cap;
            const long long len = ((kv.first + k - 1) / k + quantum - 1) / quantum * quantum;
            const long long later = std::max<long long>(len - shrink, 1);
            runs += kv.second * ((kv.first <= len) ? 1 : 1 + (kv.first - len + later - 1) / later);
            longest = std::max(longest, len);
        }
        const long long wgs = (runs + wg_runs - 1) / wg_runs;
        const long long rounds = (wgs + num_cu - 1) / num_cu;
        const long long cost = rounds * (longest + halo);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = cap; }     // ties: the longer run (fewer halos)
    }
    return best;
}
// the runs of one clip of T frames under a cap: k = ceil(T / cap) runs of ceil(T / k) frames, rounded up to the quantum
static inline int clip_run_length(long long T, int cap, int quantum) {
    const long long k = (T + cap - 1) / cap;
    return (int)(((T + k - 1) / k + quantum - 1) / quantum * quantum);
}

struct paa_plan {
    long long n_clips = 0;
    int sample_kind = 0;
    int mode = 0;                   // 0 features, 1 spectrogram, 2 chromagram
    int row_width = 0;              // doubles per frame of the slab in modes 1/2
    std::vector<ClipDev> clips;
    std::vector<long long> alloc_rows;   // modes 1/2: rows the reference allocates per clip
    long long total_frames = 0, out_doubles = 0;
    TableSet *tab = nullptr;
    int stat_chunk = kStatChunk;     // samples per statistics chunk of this plan
    PlanDev P;
    ClipDev *d_clips = nullptr;
    ClipNorm *d_norms = nullptr;
    Tile *d_tiles = nullptr;
    StatChunk *d_chunks = nullptr;
    void *d_psum = nullptr, *d_pmin = nullptr, *d_pmax = nullptr;
    long long *d_mid_off = nullptr;
    GenLayout gl;                    // generic kernel: LDS layout + table blob
    reg::RegLayout rl;               // register-FFT kernel (windows 2 R1 R2): LDS layout, blob in d_gen_blob
    int reg = 0;
    unsigned char *d_gen_blob = nullptr;
    int big = 0;                     // window beyond the LDS envelope: Stockham passes through HBM scratch
    void *d_big = nullptr;
    size_t big_bytes = 0;
    long long mid_off_step = -1;
    long long n_tiles = 0, n_chunks = 0;
    size_t lds = 0;
    int fast = 0;                    // 1: specialised kernel
    FastLaunch fl;
    int mixk = 0;                    // 1: in-place mixed-radix kernel (kernels_mix.hpp); table blob in d_gen_blob
    mix::MixLayout ml;
    int ct = 0;                      // 1: register-FFT family for windows 2 RA RB (kernels_ct.hpp); table blob in d_gen_blob
    ct::CtLaunch cl;
    int tri = 0;                     // 1: three-pass register FFT for the large default windows (kernels_tri.hpp); blob in d_gen_blob
    tri::TriLaunch trl;
    std::string kernel_name;
};

static std::atomic<int> g_live_plans{0};          // plans hold raw pointers into the device's table sets (freed outside g_mu too)
static void plan_free(paa_plan *p) {
    if (!p) return;
    --g_live_plans;
    // (the caller has synchronised the stream the plan ran on: pooled blocks may be handed to the next plan at once)
    pool_free(p->d_clips); pool_free(p->d_norms); pool_free(p->d_tiles); pool_free(p->d_chunks);
    pool_free(p->d_psum); pool_free(p->d_pmin); pool_free(p->d_pmax); pool_free(p->d_mid_off); pool_free(p->d_gen_blob);
    if (p->d_big) (void)hipFree(p->d_big);
    delete p;
}

// deleter of the per-call plans of the host-buffer entry points: an early error return may leave kernels of this call in
// flight on the lane's stream, and the plan's pooled blocks go straight to the next plan
static void plan_free_synced(paa_plan *p) {
    if (!p) return;
    if (cs()) (void)hipStreamSynchronize(cs());
    plan_free(p);
}

static int plan_build(const int64_t *offsets, int64_t n_clips, int sample_kind, double fs, int window, int step,
                      int deltas, int mode, paa_plan **out) {
    int rc = ensure_init();
    if (rc) return rc;
    if (!offsets || n_clips < 1 || !out) return fail(PAA_ERR_ARG, "null offsets / no clips");
    if (window < 2 || step < 1) return fail(PAA_ERR_ARG, "window=%d step=%d: need window >= 2, step >= 1", window, step);
    if (sample_kind < 0 || sample_kind > 2)
        return fail(PAA_ERR_ARG, "sample_kind must be 0 (int16), 1 (float64) or 2 (interleaved stereo int16)");
    if (!(fs > 0)) return fail(PAA_ERR_ARG, "sampling rate must be positive");
    std::unique_ptr<paa_plan, void (*)(paa_plan *)> p(new paa_plan(), plan_free);
    ++g_live_plans;
    p->n_clips = n_clips;
    p->sample_kind = sample_kind;
    p->mode = mode;
    TableSet *tab = nullptr;
    if ((rc = get_tables(fs, window, mode == 0, mode != 1, &tab))) return rc;
    p->tab = tab;
    const int Nf = window / 2;
    const int F = (mode == 0) ? kBase * (deltas ? 2 : 1) : 0;
    p->row_width = (mode == 1) ? Nf : (mode == 2 ? 12 : 0);

    // ---- clips
    p->clips.resize(n_clips);
    p->alloc_rows.assign(n_clips, 0);
    long long out_off = 0, total_frames = 0, n_chunks = 0;
    p->stat_chunk = stat_chunk_for(offsets[n_clips] - offsets[0], g_num_cu);
    const int kChunk = p->stat_chunk;
    for (int64_t c = 0; c < n_clips; ++c) {
        const long long n = offsets[c + 1] - offsets[c];
        if (n < 0) return fail(PAA_ERR_ARG, "offsets must be non-decreasing (clip %lld)", (long long)c);
        ClipDev &cd = p->clips[c];
        cd.sample_off = offsets[c];
        cd.n = n;
        cd.out_off = out_off;
        long long T = 0, rows = 0;
        if (mode == 0) {
            T = paa_num_frames(n, window, step);
            if (T < 1)
                return fail(PAA_ERR_TOO_SHORT, "need at least one array to concatenate (clip %lld has %lld samples, "
                            "window %d)", (long long)c, n, window);
            rows = T;
            out_off += (long long)F * T;
        } else {
            int64_t filled = 0;
            rows = (mode == 1) ? paa_spectrogram_rows(n, window, step, &filled)
                               : paa_chromagram_rows(n, window, step, &filled);
            if (rows < 1)
                return fail(PAA_ERR_TOO_SHORT, "signal too short for window %d / step %d (clip %lld, %lld samples)",
                            window, step, (long long)c, n);
            // full-length frames only; a truncated chromagram tail frame is added by the caller
            long long full = 0;
            for (long long pos = window; pos + window <= n && full < filled; pos += step) ++full;
            T = full;
            out_off += rows * p->row_width;
        }
        if (T > 0x7fffffffLL) return fail(PAA_ERR_ARG, "clip %lld has too many frames", (long long)c);
        p->alloc_rows[c] = rows;
        cd.T = (int)T;
        cd.stat_first = (int)n_chunks;
        cd.stat_count = (int)((n + kChunk - 1) / kChunk);
        cd.pad = 0;
        n_chunks += cd.stat_count;
        total_frames += T;
    }
    p->total_frames = total_frames;
    p->out_doubles = out_off;
    p->n_chunks = n_chunks;

    // ---- device plan
    PlanDev &P = p->P;
    memset(&P, 0, sizeof(P));
    P.W = window; P.S = step; P.Nf = Nf; P.Nc = tab->fft.len; P.even = tab->fft.even;
    P.n_pass = (int)tab->fft.radix.size();
    if (P.n_pass > 24) return fail(PAA_ERR_UNSUPPORTED, "window %d needs more than 24 FFT passes", window);
    for (int i = 0; i < P.n_pass; ++i) P.radix[i] = tab->fft.radix[i];
    P.tw = tab->d_tw; P.post = tab->d_post;
    P.mel_lo = tab->d_mel_lo; P.mel_cnt = tab->d_mel_cnt; P.mel_off = tab->d_mel_off; P.mel_w = tab->d_mel_w;
    P.dct = tab->d_dct; P.ch_start = tab->d_ch_start; P.ch_src = tab->d_ch_src; P.ch_w = tab->d_ch_w;
    P.fs = fs; P.deltas = deltas ? 1 : 0; P.F = F;
    P.blk_t = window / 10; P.blk_f = Nf / 10;
    P.mode = mode;
    P.frame_origin = (mode == 0) ? 0 : window;
    { const char *dbg = experiment_env("PAA_KERNEL_DEBUG"); P.debug = dbg ? atoi(dbg) : 0; }

    // ---- kernel choice + tiles
    p->fast = 0;
    if (mode == 0 && !g_force_generic) {
        rc = fast_select(window, step, sample_kind, fs, tab->fast, tab->fft, tab->mel, tab->chroma, p->fl, g_f800_waves);
        if (rc < 0) return fail(rc, "building the tables of the specialised kernel failed");
        p->fast = rc;
    }
    int run, run_quantum = 4;
    int run_halo = 0;                // frames a run with t0 > 0 starts early INSIDE its first iteration (ct kernels)
    if (!p->fast && !g_force_generic) {
        std::vector<unsigned char> blob;
        if (ct::ct_select(window, mode, fs, tab->fft, mode == 0 ? &tab->mel : nullptr, mode != 1 ? &tab->chroma : nullptr,
                          p->cl, blob)) {
            if ((rc = upload_pooled(&p->d_gen_blob, blob.data(), blob.size()))) return rc;
            p->ct = 1;
        }
    }
    // window 1102 (config 5): spectrogram / chromagram rows stay with the prime-factor kernel (2.8e8 frames/s against 2.3e8);
    // the FEATURE matrix goes to the three-pass real-input kernel below since round 4 -- same rate (1.28e8 / 1.30e8), but
    // 64-byte chunked row stores instead of 6-frame row segments (-DPAA_EXPERIMENTS builds: PAA_REG_1102=1 for A/B runs)
    const bool reg_wanted = reg::reg_supported(window) && (mode != 0 || experiment_env("PAA_REG_1102"));
    if (!p->fast && !p->ct && !g_force_generic && tab->fft.even && reg_wanted) {
        // windows 2 R1 R2 with coprime primes (config 5: 1102): several frames per wave, prime-factor FFT in registers
        using SH = reg::Shape1102;
        std::vector<unsigned char> blob;
        reg::reg_layout(tab->fft, mode == 0 ? &tab->mel : nullptr, mode != 1 ? &tab->chroma : nullptr, F, SH::NFP, SH::Q,
                        p->rl, &blob);
        if ((size_t)p->rl.table_bytes + (size_t)p->rl.wave_bytes <= 160 * 1024) {
            if ((rc = upload_pooled(&p->d_gen_blob, blob.data(), blob.size()))) return rc;
            p->reg = 1;
        }
    }
    if (!p->fast && !p->ct && !p->reg && !g_force_generic) {
        // the reference's default 50 ms windows at 48 / 44.1 kHz (2400, 2205): three-pass FFT in registers, 7 waves per CU
        std::vector<unsigned char> blob;
        if ((window != 1102 || mode == 0) && tri::tri_select(window, mode, fs, mode == 0 ? &tab->mel : nullptr, mode != 1 ? &tab->chroma : nullptr, p->trl, blob)) {
            if ((rc = upload_pooled(&p->d_gen_blob, blob.data(), blob.size()))) return rc;
            p->tri = 1;
        }
    }
    if (!p->fast && !p->ct && !p->reg && !p->tri && !g_force_generic && !experiment_env("PAA_NO_MIX")) {
        // FFT lengths made of 2, 3, 5, 7, 11, 13 (50 ms at 44.1 / 48 kHz, 1024, ...): in-place transform, 4 waves per CU
        std::vector<unsigned char> blob;
        if (mix::mix_layout(tab->fft, mode == 0 ? &tab->mel : nullptr, mode != 1 ? &tab->chroma : nullptr, F, p->ml, &blob)) {
            if ((rc = upload_pooled(&p->d_gen_blob, blob.data(), blob.size()))) return rc;
            p->mixk = 1;
        }
    }
    if (p->tri) {
        // one wave per run, one frame per iteration; a run with t0 > 0 recomputes 1 frame (2 with deltas) first
        run_quantum = 1;
        run = choose_run_cap(p->clips, 1, 8, 96, (mode == 0) ? (deltas ? 2 : 1) : 0, p->trl.waves, g_num_cu);
        p->lds = p->trl.lds;
        p->kernel_name = p->trl.name;
    } else if (p->mixk) {
        p->lds = mix::mix_lds_bytes(p->ml);
        // one wave per run, one frame at a time (halo: 1 frame, 2 with deltas): about two chip-wide rounds, 8..64 frames per run
        const long long slots = (long long)g_num_cu * p->ml.waves * 2;
        const long long per = (total_frames + slots - 1) / slots;
        run = (int)std::min<long long>(64, std::max<long long>(8, (per + 3) / 4 * 4));
        p->kernel_name = (mode == 0) ? "st_mix" : (mode == 1 ? "spectrogram_mix" : "chromagram_mix");
    } else if (p->ct) {
        // one wave per run, 4 frames per iteration; a run with t0 > 0 starts 1 frame early (2 with deltas) inside its first
        // iteration, so the first run of a clip gets `run` frames and the others run - halo: every run is whole iterations
        run_quantum = 4;
        run_halo = (mode == 0) ? (deltas ? 2 : 1) : 0;
        run = choose_run_cap(p->clips, 4, 16, 256, 0, p->cl.waves, g_num_cu, run_halo);
        p->lds = p->cl.lds;
        p->kernel_name = p->cl.name;
    } else if (p->reg) {
        p->lds = (size_t)p->rl.table_bytes + (size_t)p->rl.waves * p->rl.wave_bytes;
        // runs are multiples of Q frames (halo = one iteration); see choose_run_cap
        const int q = reg::Shape1102::Q;
        run_quantum = q;
        run = choose_run_cap(p->clips, q, 4 * q, 32 * q, q, p->rl.waves, g_num_cu);
        p->kernel_name = (mode == 0) ? "st_reg_29x19" : (mode == 1 ? "spectrogram_reg_29x19" : "chromagram_reg_29x19");
    } else if (p->fast) {
        // one wave per run, in multiples of the 4-frame quad, at most fl.run frames (halo = one quad); see choose_run_cap
        run_quantum = 4;
        run = choose_run_cap(p->clips, 4, 16, p->fl.run, 4, p->fl.waves_per_cu, g_num_cu);
        if (const char *rc_env = experiment_env("PAA_RUN_CAP")) run = std::max(16, atoi(rc_env) / 4 * 4);      // A/B experiments only
        p->lds = p->fl.lds;
        p->kernel_name = p->fl.name;
    } else {
        std::vector<unsigned char> blob;
        generic_layout(tab->fft, mode == 0 ? &tab->mel : nullptr, mode != 1 ? &tab->chroma : nullptr, F, p->gl, &blob);
        p->lds = generic_lds_bytes(p->gl);
        if (p->lds > 160 * 1024) {
            p->big = 1;                 // no CPU fallback: the same passes run through HBM scratch instead
            p->lds = 0;
        } else if ((rc = upload_pooled(&p->d_gen_blob, blob.data(), blob.size()))) {
            return rc;
        }
        // one wave per run: about two chip-wide rounds of (256 CUs x waves per workgroup), 8..64 frames per run
        {
            const long long slots = (long long)g_num_cu * p->gl.waves * 2;
            const long long per = (total_frames + slots - 1) / slots;
            run = (int)std::min<long long>(64, std::max<long long>(8, (per + 3) / 4 * 4));
        }
        p->kernel_name = p->big ? "big_window_hbm_passes"
                                : (mode == 0) ? "st_generic" : (mode == 1 ? "spectrogram_generic" : "chromagram_generic");
    }
    std::vector<Tile> tiles;
    tiles.reserve((size_t)(total_frames / run + n_clips));
    for (int64_t c = 0; c < n_clips; ++c) {
        const long long T = p->clips[c].T;
        if (T <= 0) continue;
        const int len = clip_run_length(T, run, run_quantum);          // equal runs per clip
        for (long long t0 = 0; t0 < T;) {
            const long long want = (t0 > 0) ? len - run_halo : len;
            Tile tl; tl.clip = (int)c; tl.t0 = (int)t0; tl.cnt = (int)std::min<long long>(want, T - t0); tl.pad = 0;
            tiles.push_back(tl);
            t0 += tl.cnt;
        }
    }
    p->n_tiles = (long long)tiles.size();
    if (p->n_tiles > 0x7fffffffLL || n_chunks > 0x7fffffffLL || n_clips > 0x7fffffffLL)
        return fail(PAA_ERR_UNSUPPORTED, "batch too large for one launch (%lld runs, %lld statistics chunks, %lld clips)",
                    p->n_tiles, n_chunks, (long long)n_clips);
    std::vector<StatChunk> chunks;
    chunks.reserve((size_t)n_chunks);
    for (int64_t c = 0; c < n_clips; ++c)
        for (int i = 0; i < p->clips[c].stat_count; ++i) {
            StatChunk ch; ch.start = p->clips[c].sample_off + (long long)i * kChunk;
            ch.len = (int)std::min<long long>(kChunk, p->clips[c].n - (long long)i * kChunk);
            ch.clip = (int)c;
            chunks.push_back(ch);
        }
    if ((rc = upload_pooled(&p->d_clips, p->clips.data(), p->clips.size()))) return rc;
    if ((rc = upload_pooled(&p->d_tiles, tiles.data(), tiles.size()))) return rc;
    if ((rc = upload_pooled(&p->d_chunks, chunks.data(), chunks.size()))) return rc;
    if ((rc = upload_pooled(&p->d_norms, (const void *)nullptr, (size_t)n_clips))) return rc;
    const size_t nch = (size_t)std::max<long long>(n_chunks, 1);
    if ((rc = pool_alloc(&p->d_psum, nch * 8)) || (rc = pool_alloc(&p->d_pmin, nch * 8)) ||
        (rc = pool_alloc(&p->d_pmax, nch * 8))) return rc;
    // every one-launch feature kernel folds the statistics partials into the clip constants itself (its waves' prologue);
    // chromagram plans keep clip_params_kernel (the truncated-tail kernel of the host entry point reads its output), and so
    // does the big-window path (a chain of small kernels)
    P.st_sum = p->d_psum; P.st_min = p->d_pmin; P.st_max = p->d_pmax;
    P.st_scale = sample_kind == 1 ? sample_scale<double>() : (sample_kind == 2 ? sample_scale<stereo16>() : sample_scale<int16_t>());
    P.norms_inline = (!p->big && mode != 2) ? 1 : 0;
    *out = p.release();
    return PAA_OK;
}

static int launch_stats(paa_plan *p, const void *d_packed) {
    if (p->n_chunks > 0) {
        if (p->sample_kind == 0)
            hipLaunchKernelGGL(clip_stats_i16_kernel, dim3((unsigned)p->n_chunks), dim3(256), 0, cs(),
                               (const int16_t *)d_packed, p->d_chunks, (long long *)p->d_psum, (int *)p->d_pmin,
                               (int *)p->d_pmax);
        else if (p->sample_kind == 2)
            hipLaunchKernelGGL(clip_stats_stereo_kernel, dim3((unsigned)p->n_chunks), dim3(256), 0, cs(),
                               (const stereo16 *)d_packed, p->d_chunks, (long long *)p->d_psum, (int *)p->d_pmin,
                               (int *)p->d_pmax);
        else
            hipLaunchKernelGGL(clip_stats_f64_kernel, dim3((unsigned)p->n_chunks), dim3(256), 0, cs(),
                               (const double *)d_packed, p->d_chunks, (double *)p->d_psum, (double *)p->d_pmin,
                               (double *)p->d_pmax);
    }
    const unsigned gb = (unsigned)p->n_clips;
    if (p->P.norms_inline) {
        HIP_TRY(hipGetLastError());
        return PAA_OK;
    }
    if (p->sample_kind == 1)
        hipLaunchKernelGGL((clip_params_kernel<double, double>), dim3(gb), dim3(64), 0, cs(), p->d_clips,
                           p->n_clips, (const double *)p->d_psum, (const double *)p->d_pmin,
                           (const double *)p->d_pmax, sample_scale<double>(), p->P.W, p->d_norms);
    else
        hipLaunchKernelGGL((clip_params_kernel<long long, int>), dim3(gb), dim3(64), 0, cs(), p->d_clips,
                           p->n_clips, (const long long *)p->d_psum, (const int *)p->d_pmin, (const int *)p->d_pmax,
                           p->sample_kind == 2 ? sample_scale<stereo16>() : sample_scale<int16_t>(), p->P.W, p->d_norms);
    HIP_TRY(hipGetLastError());
    return PAA_OK;
}

template <typename T>
static int launch_generic(paa_plan *p, const void *d_packed, double *d_out) {
    static LdsAttrCache attr;
    if (!attr.covers(p->lds)) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&st_generic_kernel<T>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)std::max<size_t>(p->lds, 64 * 1024)));
        attr.set(std::max<size_t>(p->lds, 64 * 1024));
    }
    const unsigned grid = (unsigned)((p->n_tiles + p->gl.waves - 1) / p->gl.waves);
    hipLaunchKernelGGL(st_generic_kernel<T>, dim3(grid), dim3(64 * p->gl.waves), p->lds, cs(), p->P, p->gl,
                       p->d_gen_blob, (const T *)d_packed, p->d_clips, p->d_norms, p->d_tiles, (int)p->n_tiles, d_out);
    HIP_TRY(hipGetLastError());
    return PAA_OK;
}

template <typename T, int TWG, int LEAN>
static int launch_mix(paa_plan *p, const void *d_packed, double *d_out) {
    static LdsAttrCache attr;
    if (!attr.covers(p->lds)) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&mix::st_mix_kernel<T, TWG, LEAN>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)std::max<size_t>(p->lds, 64 * 1024)));
        attr.set(std::max<size_t>(p->lds, 64 * 1024));
    }
    const unsigned grid = (unsigned)((p->n_tiles + p->ml.waves - 1) / p->ml.waves);
    hipLaunchKernelGGL((mix::st_mix_kernel<T, TWG, LEAN>), dim3(grid), dim3(64 * p->ml.waves), p->lds, cs(), p->P, p->ml,
                       p->d_gen_blob, (const T *)d_packed, p->d_clips, p->d_norms, p->d_tiles, (int)p->n_tiles, d_out);
    HIP_TRY(hipGetLastError());
    return PAA_OK;
}
template <typename T>
static int launch_mix_any(paa_plan *p, const void *d_packed, double *d_out) {
    if (p->ml.lean && p->ml.pad_shift == 5)
        return p->ml.tw_global ? launch_mix<T, 1, 2>(p, d_packed, d_out) : launch_mix<T, 0, 2>(p, d_packed, d_out);
    if (p->ml.lean)
        return p->ml.tw_global ? launch_mix<T, 1, 1>(p, d_packed, d_out) : launch_mix<T, 0, 1>(p, d_packed, d_out);
    return p->ml.tw_global ? launch_mix<T, 1, 0>(p, d_packed, d_out) : launch_mix<T, 0, 0>(p, d_packed, d_out);
}

template <typename T>
static int launch_reg(paa_plan *p, const void *d_packed, double *d_out) {
    using SH = reg::Shape1102;
    static LdsAttrCache attr;
    if (!attr.covers(p->lds)) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&reg::st_reg_kernel<SH, T>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)std::max<size_t>(p->lds, 64 * 1024)));
        attr.set(std::max<size_t>(p->lds, 64 * 1024));
    }
    const unsigned grid = (unsigned)((p->n_tiles + p->rl.waves - 1) / p->rl.waves);
    hipLaunchKernelGGL((reg::st_reg_kernel<SH, T>), dim3(grid), dim3(64 * p->rl.waves), p->lds, cs(), p->P, p->rl,
                       p->d_gen_blob, (const T *)d_packed, p->d_clips, p->d_norms, p->d_tiles, (int)p->n_tiles, d_out);
    HIP_TRY(hipGetLastError());
    return PAA_OK;
}

// windows beyond the LDS envelope: chunked Stockham passes through HBM scratch (kernels_big.hpp)
template <typename T>
static int run_big(paa_plan *p, const void *d_packed, double *d_out) {
    const PlanDev &P = p->P;
    const long long Nc = P.Nc, Nf = P.Nf;
    const size_t per_frame = (size_t)Nc * 32 + (size_t)Nf * 8 + 24;
    long long maxT = 0;
    for (auto &cd : p->clips) maxT = std::max<long long>(maxT, cd.T);
    long long C = (long long)std::max<size_t>(1, ((size_t)1 << 30) / per_frame);
    C = std::min<long long>(std::min<long long>(C, 65535), std::max<long long>(maxT, 1));
    const size_t need = (size_t)C * Nc * 32 + (size_t)(C + 1) * Nf * 8 + (size_t)C * 24 + 256;
    if (need > p->big_bytes) {
        if (p->d_big) { HIP_TRY(hipStreamSynchronize(cs())); (void)hipFree(p->d_big); p->d_big = nullptr; }
        HIP_TRY(hipMalloc(&p->d_big, need));
        p->big_bytes = need;
    }
    double2 *bufA = reinterpret_cast<double2 *>(p->d_big);
    double2 *bufB = bufA + C * Nc;
    double *spec = reinterpret_cast<double *>(bufB + C * Nc);
    double *tfeat = spec + (C + 1) * Nf;
    const unsigned gx = (unsigned)std::min<long long>(64, (std::max<long long>(Nc, P.W) + 255) / 256);
    for (long long c = 0; c < p->n_clips; ++c) {
        const ClipDev &cd = p->clips[c];
        const T *x0 = (const T *)d_packed + cd.sample_off + P.frame_origin;
        double *oc = d_out + cd.out_off;
        long long prev_n = 0;
        for (long long t0 = 0; t0 < cd.T; t0 += C) {
            const long long n = std::min<long long>(C, cd.T - t0);
            if (t0 > 0 && P.mode != 1)       // carry the last spectrum of the previous chunk into row 0
                HIP_TRY(hipMemcpyAsync(spec, spec + prev_n * Nf, (size_t)Nf * 8, hipMemcpyDeviceToDevice, cs()));
            hipLaunchKernelGGL(big_load_kernel<T>, dim3(gx, (unsigned)n), dim3(256), 0, cs(), P, x0, t0, ClipNorm(),
                               p->d_norms, (int)c, bufA);
            if (P.mode == 0)
                hipLaunchKernelGGL(big_time_kernel, dim3((unsigned)n), dim3(64), 0, cs(), P, bufA, tfeat);
            double2 *src = bufA, *dst = bufB;
            int Ns = 1;
            for (int q = 0; q < P.n_pass; ++q) {
                hipLaunchKernelGGL(big_pass_kernel, dim3(gx, (unsigned)n), dim3(256), 0, cs(), (int)Nc, P.radix[q], Ns,
                                   P.tw, src, dst);
                Ns *= P.radix[q];
                std::swap(src, dst);
            }
            if (P.mode == 1) {
                hipLaunchKernelGGL(big_post_kernel, dim3(gx, (unsigned)n), dim3(256), 0, cs(), P, src, oc, t0);
            } else {
                hipLaunchKernelGGL(big_post_kernel, dim3(gx, (unsigned)n), dim3(256), 0, cs(), P, src, spec, 1LL);
                hipLaunchKernelGGL(big_feat_kernel, dim3((unsigned)n), dim3(64), 0, cs(), P, spec, tfeat, t0,
                                   (long long)cd.T, oc);
            }
            HIP_TRY(hipGetLastError());
            prev_n = n;
        }
        if (P.mode == 0 && P.deltas) {
            const long long items = (long long)kBase * cd.T;
            hipLaunchKernelGGL(big_delta_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, cs(),
                               (long long)cd.T, oc);
            HIP_TRY(hipGetLastError());
        }
    }
    return PAA_OK;
}

extern "C" int paa_plan_execute(paa_plan_t *plan, const void *d_packed, double *d_out) {
    if (!plan || !d_packed || !d_out) return fail(PAA_ERR_ARG, "null plan / buffer");
    { const int rc_init = ensure_init(); if (rc_init) return rc_init; }
    std::lock_guard<std::mutex> lk(g_mu);
    int rc = comm_wait_buffer_free(d_out);      // a gather of this buffer may still be in flight
    if (rc) return rc;
    rc = launch_stats(plan, d_packed);
    if (rc) return rc;
    if (plan->big)
        return plan->sample_kind == 0 ? run_big<int16_t>(plan, d_packed, d_out)
             : plan->sample_kind == 2 ? run_big<stereo16>(plan, d_packed, d_out) : run_big<double>(plan, d_packed, d_out);
    if (plan->n_tiles == 0) return PAA_OK;
    hipEvent_t pe0 = nullptr, pe1 = nullptr;
    if (g_prof && (g_prof_seen++ % g_prof) == 0) {
        if (g_prof_used == g_prof_ev.size()) {
            hipEvent_t a, b;
            HIP_TRY(hipEventCreate(&a));
            HIP_TRY(hipEventCreate(&b));
            g_prof_ev.emplace_back(a, b);
        }
        pe0 = g_prof_ev[g_prof_used].first;
        pe1 = g_prof_ev[g_prof_used].second;
        ++g_prof_used;
        HIP_TRY(hipEventRecord(pe0, cs()));
    }
    struct StopEv { hipEvent_t e; ~StopEv() { if (e) (void)hipEventRecord(e, cs()); } } stop_ev{pe1};
    if (plan->fast) {
        rc = fast_launch(plan->fl, plan->P, plan->tab->fast, d_packed, plan->d_clips, plan->d_norms, plan->d_tiles,
                         plan->n_tiles, d_out, cs());
        if (rc) return fail(PAA_ERR_HIP, "launch of %s failed: %s", plan->kernel_name.c_str(),
                            hipGetErrorString(hipGetLastError()));
        return PAA_OK;
    }
    if (plan->ct) {
        rc = ct::ct_launch(plan->cl, plan->sample_kind, plan->P, plan->d_gen_blob, d_packed, plan->d_clips, plan->d_norms,
                           plan->d_tiles, plan->n_tiles, d_out, cs());
        if (rc) return fail(PAA_ERR_HIP, "launch of %s failed: %s", plan->kernel_name.c_str(),
                            hipGetErrorString(hipGetLastError()));
        return PAA_OK;
    }
    if (plan->tri) {
        rc = tri::tri_launch(plan->trl, plan->sample_kind, plan->P, plan->d_gen_blob, d_packed, plan->d_clips, plan->d_norms,
                             plan->d_tiles, plan->n_tiles, d_out, cs());
        if (rc) return fail(PAA_ERR_HIP, "launch of %s failed: %s", plan->kernel_name.c_str(),
                            hipGetErrorString(hipGetLastError()));
        return PAA_OK;
    }
    if (plan->reg)
        return plan->sample_kind == 0 ? launch_reg<int16_t>(plan, d_packed, d_out)
             : plan->sample_kind == 2 ? launch_reg<stereo16>(plan, d_packed, d_out) : launch_reg<double>(plan, d_packed, d_out);
    if (plan->mixk)
        return plan->sample_kind == 0 ? launch_mix_any<int16_t>(plan, d_packed, d_out)
             : plan->sample_kind == 2 ? launch_mix_any<stereo16>(plan, d_packed, d_out)
                                      : launch_mix_any<double>(plan, d_packed, d_out);
    return plan->sample_kind == 0 ? launch_generic<int16_t>(plan, d_packed, d_out)
         : plan->sample_kind == 2 ? launch_generic<stereo16>(plan, d_packed, d_out)
                                  : launch_generic<double>(plan, d_packed, d_out);
}

extern "C" int paa_plan_create(const int64_t *offsets, int64_t n_clips, int sample_kind, double fs, int window,
                               int step, int deltas, paa_plan_t **out_plan) {
    std::lock_guard<std::mutex> lk(g_mu);
    return plan_build(offsets, n_clips, sample_kind, fs, window, step, deltas, 0, out_plan);
}

// device-resident plan of the spectrogram (mode 1, :389-452) / chromagram (mode 2, :324-386) rows of one or more clips:
// full-length frames only, row t of a clip at out + out_offset(clip) + t * row_width (Nf or 12 doubles); rows the reference
// allocates but never fills, and the truncated chromagram tail frame, are the host entry points' business
extern "C" int paa_plan_create_mode(const int64_t *offsets, int64_t n_clips, int sample_kind, double fs, int window,
                                    int step, int mode, paa_plan_t **out_plan) {
    if (mode < 0 || mode > 2) return fail(PAA_ERR_ARG, "mode must be 0 (features), 1 (spectrogram) or 2 (chromagram)");
    std::lock_guard<std::mutex> lk(g_mu);
    return plan_build(offsets, n_clips, sample_kind, fs, window, step, 0, mode, out_plan);
}

extern "C" int paa_plan_destroy(paa_plan_t *plan) {
    if (g_device.load() >= 0) (void)ensure_init();       // (binds the calling thread to the library's device)
    std::lock_guard<std::mutex> lk(g_mu);
    if (cs()) (void)hipStreamSynchronize(cs());
    plan_free(plan);
    return PAA_OK;
}

extern "C" int64_t paa_plan_total_frames(const paa_plan_t *plan) { return plan ? plan->total_frames : 0; }
extern "C" int64_t paa_plan_out_doubles(const paa_plan_t *plan) { return plan ? plan->out_doubles : 0; }
extern "C" const char *paa_plan_kernel_name(const paa_plan_t *plan) { return plan ? plan->kernel_name.c_str() : ""; }

extern "C" int paa_plan_out_offsets(const paa_plan_t *plan, int64_t *out_offsets) {
    if (!plan || !out_offsets) return fail(PAA_ERR_ARG, "null plan / buffer");
    for (long long c = 0; c < plan->n_clips; ++c) out_offsets[c] = plan->clips[c].out_off;
    return PAA_OK;
}

extern "C" int64_t paa_plan_mid_doubles(const paa_plan_t *plan, int64_t mid_step_ratio) {
    if (!plan || mid_step_ratio < 1) return 0;
    long long tot = 0;
    for (long long c = 0; c < plan->n_clips; ++c)
        tot += 2LL * plan->P.F * paa_num_mid_windows(plan->clips[c].T, mid_step_ratio);
    return tot;
}

extern "C" int paa_plan_mid_execute(paa_plan_t *plan, const double *d_st, int64_t mid_ratio, int64_t mid_step_ratio,
                                    double *d_mid) {
    if (!plan || !d_st || !d_mid) return fail(PAA_ERR_ARG, "null plan / buffer");
    { const int rc_init = ensure_init(); if (rc_init) return rc_init; }
    if (plan->mode != 0) return fail(PAA_ERR_ARG, "mid-term statistics need a feature plan");
    if (mid_step_ratio < 1)
        return fail(PAA_ERR_ARG, "mid_step / short_step rounds to %lld: the reference loops forever "
                    "(MidTermFeatures.py:102,124)", (long long)mid_step_ratio);
    std::lock_guard<std::mutex> lk(g_mu);
    { const int rc_w = comm_wait_buffer_free(d_mid); if (rc_w) return rc_w; }      // a gather of this buffer may still read it
    long long maxM = 0;
    if (plan->mid_off_step != mid_step_ratio) {
        std::vector<long long> off(plan->n_clips);
        long long o = 0;
        for (long long c = 0; c < plan->n_clips; ++c) {
            off[c] = o;
            o += 2LL * plan->P.F * paa_num_mid_windows(plan->clips[c].T, mid_step_ratio);
        }
        if (cs()) HIP_TRY(hipStreamSynchronize(cs()));
        int rc = upload_pooled(&plan->d_mid_off, off.data(), off.size());
        if (rc) return rc;
        plan->mid_off_step = mid_step_ratio;
    }
    for (long long c = 0; c < plan->n_clips; ++c)
        maxM = std::max<long long>(maxM, paa_num_mid_windows(plan->clips[c].T, mid_step_ratio));
    const long long items = (long long)plan->P.F * maxM;
    const int bpc = (int)((items + 15) / 16);          // 16 (row, window) items per 256-thread block
    const long long grid = plan->n_clips * bpc;
    if (grid > 0x7fffffffLL) return fail(PAA_ERR_UNSUPPORTED, "mid-term grid too large");
    hipLaunchKernelGGL(mid_stats_kernel, dim3((unsigned)grid), dim3(256), 0, cs(), plan->d_clips, plan->d_mid_off,
                       d_st, plan->P.F, (long long)mid_ratio, (long long)mid_step_ratio, bpc, d_mid);
    HIP_TRY(hipGetLastError());
    return PAA_OK;
}

// beat rate of every clip of an executed plan (deltas on or off: rows 0..18 are used)
extern "C" int paa_plan_beat_execute(paa_plan_t *plan, const double *d_st, double window_size, double *d_beat) {
    if (!plan || !d_st || !d_beat) return fail(PAA_ERR_ARG, "null plan / buffer");
    if (plan->mode != 0) return fail(PAA_ERR_ARG, "beat extraction needs a feature plan");
    if (!(window_size > 0)) return fail(PAA_ERR_ARG, "window_size must be positive");
    { const int rc_init = ensure_init(); if (rc_init) return rc_init; }
    const int max_beat = (int)nearbyint(2.0 / window_size);          // int(round(2.0 / window_size)), :33
    if (max_beat < 1 || max_beat > 4096) return fail(PAA_ERR_UNSUPPORTED, "beat histogram of %d bins", max_beat);
    std::lock_guard<std::mutex> lk(g_mu);
    { const int rc_w = comm_wait_buffer_free(d_beat); if (rc_w) return rc_w; }
    const size_t lds = (size_t)kBeatRows * (kBeatTile + 1) * 8 + (size_t)kBeatRows * max_beat * 4;
    if (lds > 160 * 1024)
        return fail(PAA_ERR_UNSUPPORTED, "beat histogram of %d bins needs %zu bytes of LDS (160 KB per workgroup)", max_beat, lds);
    if (lds > 64 * 1024)
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&beat_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(beat_kernel, dim3((unsigned)plan->n_clips), dim3(64), lds, cs(), plan->d_clips, d_st,
                       window_size, max_beat, d_beat);
    HIP_TRY(hipGetLastError());
    return PAA_OK;
}

// ------------------------------------------------------------------------------------------
// self-similarity matrix / thumbnail filter (audioSegmentation.py:40-55, 1141-1165)
// ------------------------------------------------------------------------------------------
static std::mutex g_sim_mu;      // the scratch buffers below are shared: one enqueue sequence at a time

extern "C" int64_t paa_thumbnail_rows(int64_t n_vec, int m_filter) {
    if (m_filter < 1 || n_vec < m_filter) return 0;
    return n_vec - m_filter + 1;
}

extern "C" int paa_dev_self_similarity(const double *d_feats, int n_dims, int64_t n_vec, int64_t ld, double *d_sim) {
    if (!d_feats || !d_sim) return fail(PAA_ERR_ARG, "null buffer");
    if (n_dims < 1 || n_vec < 1 || ld < n_vec) return fail(PAA_ERR_ARG, "bad feature matrix shape %d x %lld (ld %lld)", n_dims, (long long)n_vec, (long long)ld);
    if (n_vec > 46340LL * 4) return fail(PAA_ERR_UNSUPPORTED, "%lld vectors: similarity matrix too large", (long long)n_vec);
    int rc = ensure_init();
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(g_sim_mu);
    const int dims_pad = (n_dims + 3) / 4 * 4;
    const long long ldz = (n_vec + kSimTile - 1) / kSimTile * kSimTile;
    {
        std::lock_guard<std::mutex> lk2(g_mu);
        if ((rc = scratch_reserve(g_sim_z, (size_t)dims_pad * ldz * 8))) return rc;
        if ((rc = scratch_reserve(g_sim_small, (size_t)(2 * n_dims + ldz) * 8))) return rc;
    }
    double *d_mean = (double *)g_sim_small.p, *d_scale = d_mean + n_dims, *d_norm = d_scale + n_dims;
    hipLaunchKernelGGL(sim_row_stats_kernel, dim3((unsigned)n_dims), dim3(256), 0, cs(), d_feats, (long long)n_vec,
                       (long long)ld, d_mean, d_scale);
    hipLaunchKernelGGL(sim_normalize_kernel, dim3((unsigned)((ldz + 255) / 256)), dim3(256), 0, cs(), d_feats,
                       n_dims, dims_pad, (long long)n_vec, (long long)ld, ldz, d_mean, d_scale, (double *)g_sim_z.p,
                       d_norm);
    const unsigned tiles = (unsigned)(ldz / kSimTile);
    const size_t lds = (size_t)2 * kSimChunk * kSimPitch * 8 + 256 * 8;
    static LdsAttrCache attr;
    if (!attr.covers(lds)) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&sim_gram_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr.set(lds);
    }
    const long long n_tri = (long long)tiles * (tiles + 1) / 2;        // tiles on and above the diagonal; the rest are mirrored
    const dim3 gram_grid((unsigned)std::min<long long>(n_tri, 2LL * g_num_cu));
    hipLaunchKernelGGL(sim_gram_kernel, gram_grid, dim3(512), lds, cs(), (const double *)g_sim_z.p, dims_pad,
                       (long long)n_vec, ldz, d_norm, d_sim);
    HIP_TRY(hipGetLastError());
    return PAA_OK;
}

extern "C" int paa_dev_thumbnail_filter(const double *d_sim, int64_t n_vec, int m_filter, double band, double limit_1,
                                        double limit_2, double *d_filt, int64_t *pos2) {
    if (!d_sim || !d_filt || !pos2) return fail(PAA_ERR_ARG, "null buffer");
    const long long R = paa_thumbnail_rows(n_vec, m_filter);
    if (R < 1)
        return fail(PAA_ERR_ARG, "fewer feature vectors (%lld) than the thumbnail filter length (%d)",
                    (long long)n_vec, m_filter);
    if (!(limit_1 >= 0.0) || !(limit_2 >= 0.0)) return fail(PAA_ERR_ARG, "limit_1 / limit_2 must be >= 0");
    int rc = ensure_init();
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(g_sim_mu);
    const long long lim_lo = (long long)(limit_1 * (double)R), lim_hi = (long long)(limit_2 * (double)R);   // int(), :1157-1160
    // thumb_diag: block (bx, by) = diagonal offsets 256 bx .. of rows kDiagRun by ..; offsets past R - i0 exit at once
    const unsigned gx = (unsigned)((R + 255) / 256), gy = (unsigned)((R + kDiagRun - 1) / kDiagRun);
    const unsigned mx = (unsigned)((R + 1023) / 1024), my = (unsigned)((R + kMaskRows - 1) / kMaskRows);
    if (my > 65535u || gy > 65535u) return fail(PAA_ERR_UNSUPPORTED, "%lld rows: thumbnail matrix too large", R);
    const long long n_blk = (long long)gx * gy;
    {
        std::lock_guard<std::mutex> lk2(g_mu);
        if ((rc = scratch_reserve(g_sim_cand, (size_t)(3 * n_blk + 4) * 8))) return rc;
    }
    double *d_min = (double *)g_sim_cand.p, *d_cval = d_min + n_blk + 1;
    long long *d_cidx = (long long *)(d_cval + n_blk), *d_best = d_cidx + n_blk;
    hipLaunchKernelGGL(thumb_diag_kernel, dim3(gx, gy), dim3(256), 0, cs(), d_sim, (long long)n_vec, m_filter, R, band,
                       lim_lo, lim_hi, d_filt, d_min, d_cval, d_cidx);
    hipLaunchKernelGGL(thumb_min_kernel, dim3(1), dim3(1024), 0, cs(), (const double *)d_min, n_blk, d_min + n_blk);
    hipLaunchKernelGGL(thumb_fill_kernel, dim3(mx, my), dim3(256), 0, cs(), d_filt, R, band, lim_lo, lim_hi,
                       (const double *)(d_min + n_blk));
    hipLaunchKernelGGL(thumb_argmax_kernel, dim3(1), dim3(1024), 0, cs(), (const double *)d_cval,
                       (const long long *)d_cidx, n_blk, (const double *)(d_min + n_blk), R, band, lim_lo, lim_hi, d_best);
    HIP_TRY(hipGetLastError());
    long long best = 0;
    HIP_TRY(hipMemcpyAsync(&best, d_best, 8, hipMemcpyDeviceToHost, cs()));
    HIP_TRY(hipStreamSynchronize(cs()));
    pos2[0] = best / R;
    pos2[1] = best % R;
    return PAA_OK;
}


extern "C" int paa_self_similarity_f64(const double *feats, int n_dims, int64_t n_vec, double *sim) {
    if (!feats || !sim) return fail(PAA_ERR_ARG, "null buffer");
    if (n_dims < 1 || n_vec < 1) return fail(PAA_ERR_ARG, "empty feature matrix");
    int rc = ensure_init();
    if (rc) return rc;
    std::lock_guard<std::mutex> api_lock(g_api_mu);
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if ((rc = scratch_reserve(g_sim_in, (size_t)n_dims * n_vec * 8))) return rc;
        if ((rc = scratch_reserve(g_sim_out, (size_t)n_vec * n_vec * 8))) return rc;
    }
    HIP_TRY(hipMemcpyAsync(g_sim_in.p, feats, (size_t)n_dims * n_vec * 8, hipMemcpyHostToDevice, cs()));
    if ((rc = paa_dev_self_similarity((const double *)g_sim_in.p, n_dims, n_vec, n_vec, (double *)g_sim_out.p))) return rc;
    HIP_TRY(hipMemcpyAsync(sim, g_sim_out.p, (size_t)n_vec * n_vec * 8, hipMemcpyDeviceToHost, cs()));
    HIP_TRY(hipStreamSynchronize(cs()));
    return PAA_OK;
}

extern "C" int paa_thumbnail_f64(const double *feats, int n_dims, int64_t n_vec, int m_filter, double band,
                                 double limit_1, double limit_2, double *filt, int64_t *pos2) {
    if (!feats || !filt || !pos2) return fail(PAA_ERR_ARG, "null buffer");
    if (n_dims < 1 || n_vec < 1) return fail(PAA_ERR_ARG, "empty feature matrix");
    const long long R = paa_thumbnail_rows(n_vec, m_filter);
    if (R < 1)
        return fail(PAA_ERR_ARG, "fewer feature vectors (%lld) than the thumbnail filter length (%d)",
                    (long long)n_vec, m_filter);
    int rc = ensure_init();
    if (rc) return rc;
    std::lock_guard<std::mutex> api_lock(g_api_mu);
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if ((rc = scratch_reserve(g_sim_in, (size_t)n_dims * n_vec * 8))) return rc;
        if ((rc = scratch_reserve(g_sim_out, (size_t)n_vec * n_vec * 8))) return rc;
        if ((rc = scratch_reserve(g_sim_filt, (size_t)R * R * 8))) return rc;
    }
    HIP_TRY(hipMemcpyAsync(g_sim_in.p, feats, (size_t)n_dims * n_vec * 8, hipMemcpyHostToDevice, cs()));
    if ((rc = paa_dev_self_similarity((const double *)g_sim_in.p, n_dims, n_vec, n_vec, (double *)g_sim_out.p))) return rc;
    if ((rc = paa_dev_thumbnail_filter((const double *)g_sim_out.p, n_vec, m_filter, band, limit_1, limit_2,
                                       (double *)g_sim_filt.p, pos2))) return rc;
    HIP_TRY(hipMemcpyAsync(filt, g_sim_filt.p, (size_t)R * R * 8, hipMemcpyDeviceToHost, cs()));
    HIP_TRY(hipStreamSynchronize(cs()));
    return PAA_OK;
}

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
    for (int i = 0; i < kLanes; ++i)
        if (!g_lanes[i].stream) HIP_TRY(hipStreamCreateWithFlags(&g_lanes[i].stream, hipStreamNonBlocking));
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

// ------------------------------------------------------------------------------------------
// host-buffer entry points
// ------------------------------------------------------------------------------------------
static int run_host_st(const void *packed, const int64_t *offsets, int64_t n_clips, int sample_kind, double fs,
                       int window, int step, int deltas, double *out, const int64_t *out_offsets,
                       int64_t mid_ratio, int64_t mid_step, double *mid_out, const int64_t *mid_out_offsets) {
    if (!packed || !offsets) return fail(PAA_ERR_ARG, "null signal");
    { const int rc0 = ensure_init(); if (rc0) return rc0; }      // the lanes exist once a device is selected
    LaneGuard lane;       // own stream + scratch for this call (see Lane)
    const bool want_mid = mid_out != nullptr;
    if (want_mid && !deltas) return fail(PAA_ERR_ARG, "mid-term features are defined over the 68 delta rows");
    if (want_mid && mid_step < 1)
        return fail(PAA_ERR_ARG, "mid_step / short_step rounds to %lld: the reference loops forever "
                    "(MidTermFeatures.py:102,124)", (long long)mid_step);
    paa_plan *plan = nullptr;
    int rc;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        rc = plan_build(offsets, n_clips, sample_kind, fs, window, step, deltas, 0, &plan);
    }
    if (rc) return rc;
    std::unique_ptr<paa_plan, void (*)(paa_plan *)> guard(plan, plan_free_synced);
    // sample_kind 2: the host buffer holds interleaved stereo int16 (4 bytes per frame); the kernels sum L + R in their
    // loads (fused stereo_to_mono: the mono signal is never materialised)
    const size_t esz = sample_kind == 0 ? 2 : (sample_kind == 2 ? 4 : 8);
    const long long base = offsets[0], n_total = offsets[n_clips] - base;
    // samples are uploaded from offsets[0]; rebase the clip offsets accordingly
    if (base != 0) {
        for (auto &cd : plan->clips) cd.sample_off -= base;
        HIP_TRY(hipMemcpy(plan->d_clips, plan->clips.data(), plan->clips.size() * sizeof(ClipDev), hipMemcpyHostToDevice));
        std::vector<StatChunk> chunks;
        for (int64_t c = 0; c < n_clips; ++c)
            for (int i = 0; i < plan->clips[c].stat_count; ++i) {
                StatChunk ch; ch.start = plan->clips[c].sample_off + (long long)i * plan->stat_chunk;
                ch.len = (int)std::min<long long>(plan->stat_chunk, plan->clips[c].n - (long long)i * plan->stat_chunk);
                ch.clip = (int)c;
                chunks.push_back(ch);
            }
        if (!chunks.empty())
            HIP_TRY(hipMemcpy(plan->d_chunks, chunks.data(), chunks.size() * sizeof(StatChunk), hipMemcpyHostToDevice));
    }
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if ((rc = scratch_reserve(lane.l->in, (size_t)n_total * esz + 64))) return rc;
        if ((rc = scratch_reserve(lane.l->out, (size_t)plan->out_doubles * 8))) return rc;
    }
    HIP_TRY(hipMemcpyAsync(lane.l->in.p, (const char *)packed + (size_t)base * esz, (size_t)n_total * esz,
                           hipMemcpyHostToDevice, cs()));
    const void *d_samples = lane.l->in.p;
    if ((rc = paa_plan_execute(plan, d_samples, (double *)lane.l->out.p))) return rc;
    if (want_mid) {
        const long long md = paa_plan_mid_doubles(plan, mid_step);
        {
            std::lock_guard<std::mutex> lk(g_mu);
            if ((rc = scratch_reserve(lane.l->mid, (size_t)md * 8))) return rc;
        }
        if ((rc = paa_plan_mid_execute(plan, (const double *)lane.l->out.p, mid_ratio, mid_step, (double *)lane.l->mid.p))) return rc;
        // slabs are back to back in clip order on the device
        long long o = 0;
        for (int64_t c = 0; c < n_clips; ++c) {
            const long long cnt = 2LL * plan->P.F * paa_num_mid_windows(plan->clips[c].T, mid_step);
            double *dst = mid_out + (mid_out_offsets ? mid_out_offsets[c] : o);
            HIP_TRY(hipMemcpyAsync(dst, (double *)lane.l->mid.p + o, (size_t)cnt * 8, hipMemcpyDeviceToHost, cs()));
            o += cnt;
        }
    }
    if (out) {
        if (!out_offsets) {
            HIP_TRY(hipMemcpyAsync(out, lane.l->out.p, (size_t)plan->out_doubles * 8, hipMemcpyDeviceToHost, cs()));
        } else {
            // coalesce runs of clips whose destination slabs are contiguous too
            int64_t c = 0;
            while (c < n_clips) {
                int64_t e = c;
                long long cnt = 0;
                while (e < n_clips && out_offsets[e] - out_offsets[c] == plan->clips[e].out_off - plan->clips[c].out_off) {
                    cnt = plan->clips[e].out_off - plan->clips[c].out_off + (long long)plan->P.F * plan->clips[e].T;
                    ++e;
                }
                HIP_TRY(hipMemcpyAsync(out + out_offsets[c], (double *)lane.l->out.p + plan->clips[c].out_off,
                                       (size_t)cnt * 8, hipMemcpyDeviceToHost, cs()));
                c = e;
            }
        }
    }
    HIP_TRY(hipStreamSynchronize(cs()));
    return PAA_OK;
}

extern "C" int paa_st_features_i16(const int16_t *signal, int64_t n, double fs, int window, int step, int deltas,
                                   double *out) {
    if (!out) return fail(PAA_ERR_ARG, "null out");
    const int64_t off[2] = {0, n};
    return run_host_st(signal, off, 1, 0, fs, window, step, deltas, out, nullptr, 0, 0, nullptr, nullptr);
}
extern "C" int paa_st_features_f64(const double *signal, int64_t n, double fs, int window, int step, int deltas,
                                   double *out) {
    if (!out) return fail(PAA_ERR_ARG, "null out");
    const int64_t off[2] = {0, n};
    return run_host_st(signal, off, 1, 1, fs, window, step, deltas, out, nullptr, 0, 0, nullptr, nullptr);
}
extern "C" int paa_st_features_stereo_i16(const int16_t *interleaved, int64_t n, double fs, int window, int step,
                                          int deltas, double *out) {
    if (!out) return fail(PAA_ERR_ARG, "null out");
    const int64_t off[2] = {0, n};
    return run_host_st(interleaved, off, 1, 2, fs, window, step, deltas, out, nullptr, 0, 0, nullptr, nullptr);
}
extern "C" int paa_mid_features_stereo_i16(const int16_t *interleaved, int64_t n, double fs, int window, int step,
                                           int64_t mid_ratio, int64_t mid_step_ratio, double *mid_out, double *st_out) {
    if (!mid_out) return fail(PAA_ERR_ARG, "null mid_out");
    const int64_t off[2] = {0, n};
    return run_host_st(interleaved, off, 1, 2, fs, window, step, 1, st_out, nullptr, mid_ratio, mid_step_ratio, mid_out, nullptr);
}
extern "C" int paa_mid_features_i16(const int16_t *signal, int64_t n, double fs, int window, int step,
                                    int64_t mid_ratio, int64_t mid_step_ratio, double *mid_out, double *st_out) {
    if (!mid_out) return fail(PAA_ERR_ARG, "null mid_out");
    const int64_t off[2] = {0, n};
    return run_host_st(signal, off, 1, 0, fs, window, step, 1, st_out, nullptr, mid_ratio, mid_step_ratio, mid_out, nullptr);
}
extern "C" int paa_mid_features_f64(const double *signal, int64_t n, double fs, int window, int step,
                                    int64_t mid_ratio, int64_t mid_step_ratio, double *mid_out, double *st_out) {
    if (!mid_out) return fail(PAA_ERR_ARG, "null mid_out");
    const int64_t off[2] = {0, n};
    return run_host_st(signal, off, 1, 1, fs, window, step, 1, st_out, nullptr, mid_ratio, mid_step_ratio, mid_out, nullptr);
}
extern "C" int paa_st_features_batch_i16(const int16_t *packed, const int64_t *offsets, int64_t n_clips, double fs,
                                         int window, int step, int deltas, double *out, const int64_t *out_offsets) {
    if (!out) return fail(PAA_ERR_ARG, "null out");
    return run_host_st(packed, offsets, n_clips, 0, fs, window, step, deltas, out, out_offsets, 0, 0, nullptr, nullptr);
}
extern "C" int paa_st_features_batch_f64(const double *packed, const int64_t *offsets, int64_t n_clips, double fs,
                                         int window, int step, int deltas, double *out, const int64_t *out_offsets) {
    if (!out) return fail(PAA_ERR_ARG, "null out");
    return run_host_st(packed, offsets, n_clips, 1, fs, window, step, deltas, out, out_offsets, 0, 0, nullptr, nullptr);
}
extern "C" int paa_mid_features_batch_f64(const double *packed, const int64_t *offsets, int64_t n_clips, double fs,
                                          int window, int step, int64_t mid_ratio, int64_t mid_step_ratio,
                                          double *mid_out, const int64_t *mid_out_offsets, double *st_out,
                                          const int64_t *st_out_offsets) {
    if (!mid_out) return fail(PAA_ERR_ARG, "null mid_out");
    return run_host_st(packed, offsets, n_clips, 1, fs, window, step, 1, st_out, st_out_offsets, mid_ratio,
                       mid_step_ratio, mid_out, mid_out_offsets);
}
extern "C" int paa_mid_features_batch_i16(const int16_t *packed, const int64_t *offsets, int64_t n_clips, double fs,
                                          int window, int step, int64_t mid_ratio, int64_t mid_step_ratio,
                                          double *mid_out, const int64_t *mid_out_offsets, double *st_out,
                                          const int64_t *st_out_offsets) {
    if (!mid_out) return fail(PAA_ERR_ARG, "null mid_out");
    return run_host_st(packed, offsets, n_clips, 0, fs, window, step, 1, st_out, st_out_offsets, mid_ratio,
                       mid_step_ratio, mid_out, mid_out_offsets);
}

// ---- spectrogram / chromagram ---------------------------------------------------------------
#include "kernels_tail.hpp"

static int run_host_spec(const void *signal, int64_t n, int sample_kind, double fs, int window, int step, int mode,
                         double *out) {
    if (!signal || !out) return fail(PAA_ERR_ARG, "null signal / out");
    { const int rc0 = ensure_init(); if (rc0) return rc0; }
    LaneGuard lane;       // own stream + scratch for this call (see Lane)
    const int64_t off[2] = {0, n};
    paa_plan *plan = nullptr;
    int rc;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        rc = plan_build(off, 1, sample_kind, fs, window, step, 0, mode, &plan);
    }
    if (rc) return rc;
    std::unique_ptr<paa_plan, void (*)(paa_plan *)> guard(plan, plan_free_synced);
    // sample_kind 2: the host buffer holds interleaved stereo int16 (4 bytes per frame); the kernels sum L + R in their
    // loads (fused stereo_to_mono, audioBasicIO.py:156-168)
    const size_t esz = sample_kind == 0 ? 2 : (sample_kind == 2 ? 4 : 8);
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if ((rc = scratch_reserve(lane.l->in, (size_t)n * esz + 64))) return rc;
        if ((rc = scratch_reserve(lane.l->out, (size_t)plan->out_doubles * 8))) return rc;
    }
    HIP_TRY(hipMemcpyAsync(lane.l->in.p, signal, (size_t)n * esz, hipMemcpyHostToDevice, cs()));
    const void *d_samples = lane.l->in.p;
    HIP_TRY(hipMemsetAsync(lane.l->out.p, 0, (size_t)plan->out_doubles * 8, cs()));   // trailing rows stay 0 (:413-422)
    if ((rc = paa_plan_execute(plan, d_samples, (double *)lane.l->out.p))) return rc;
    if (mode == 2) {
        // the reference FFTs a truncated last frame when fewer than `window` samples remain (:349-355)
        int64_t filled = 0;
        paa_chromagram_rows(n, window, step, &filled);
        if (filled > plan->clips[0].T) {
            const long long pos = (long long)window + (long long)plan->clips[0].T * step;
            // the shortest (last) truncated frame decides whether the reference can index X[0:num_fft]
            const long long last_len = n - ((long long)window + (filled - 1) * step);
            if (last_len < window / 2)
                return fail(PAA_ERR_CHROMA_VALUE, "truncated last chromagram frame shorter than num_fft "
                            "(ValueError in the reference, ShortTermFeatures.py:288)");
            rc = launch_chroma_tail(plan->P, sample_kind, d_samples, pos, n, (int)(filled - plan->clips[0].T), plan->d_norms,
                                    (double *)lane.l->out.p + (long long)plan->clips[0].T * 12, cs());
            if (rc == -2) return fail(PAA_ERR_UNSUPPORTED, "truncated chromagram tail frame with window %d does not fit LDS", window);
            if (rc) return fail(PAA_ERR_HIP, "chromagram tail launch failed");
        }
    }
    HIP_TRY(hipMemcpyAsync(out, lane.l->out.p, (size_t)plan->out_doubles * 8, hipMemcpyDeviceToHost, cs()));
    HIP_TRY(hipStreamSynchronize(cs()));
    return PAA_OK;
}

extern "C" int paa_spectrogram_i16(const int16_t *s, int64_t n, double fs, int w, int st, double *out) {
    return run_host_spec(s, n, 0, fs, w, st, 1, out);
}
extern "C" int paa_spectrogram_f64(const double *s, int64_t n, double fs, int w, int st, double *out) {
    return run_host_spec(s, n, 1, fs, w, st, 1, out);
}
extern "C" int paa_chromagram_i16(const int16_t *s, int64_t n, double fs, int w, int st, double *out) {
    return run_host_spec(s, n, 0, fs, w, st, 2, out);
}
extern "C" int paa_chromagram_f64(const double *s, int64_t n, double fs, int w, int st, double *out) {
    return run_host_spec(s, n, 1, fs, w, st, 2, out);
}
extern "C" int paa_spectrogram_stereo_i16(const int16_t *lr, int64_t n, double fs, int w, int st, double *out) {
    return run_host_spec(lr, n, 2, fs, w, st, 1, out);
}
extern "C" int paa_chromagram_stereo_i16(const int16_t *lr, int64_t n, double fs, int w, int st, double *out) {
    return run_host_spec(lr, n, 2, fs, w, st, 2, out);
}

// ------------------------------------------------------------------------------------------
// onset probability of silence_removal: binary probabilistic SVC over all frames (audioSegmentation.py:744-748)
// ------------------------------------------------------------------------------------------
extern "C" int paa_svm_binary_proba_f64(const double *feats, int n_dims, int64_t n_frames, const double *mean,
                                        const double *scale, const double *support_vectors, const double *dual_coef,
                                        int n_sv, double intercept, double gamma, double prob_a, double prob_b,
                                        double *prob1) {
    int rc = ensure_init();
    if (rc) return rc;
    if (!feats || !mean || !scale || !support_vectors || !dual_coef || !prob1) return fail(PAA_ERR_ARG, "null argument");
    if (n_dims < 1 || n_dims > kSvmMaxDims) return fail(PAA_ERR_ARG, "n_dims must be 1..%d", kSvmMaxDims);
    if (n_frames < 1 || n_sv < 1) return fail(PAA_ERR_ARG, "need at least one frame and one support vector");
    LaneGuard lane;       // own stream + scratch for this call (see Lane)
    const size_t fb = (size_t)n_dims * n_frames * 8, sb = (size_t)n_sv * n_dims * 8;
    const size_t small = (size_t)(2 * n_dims + n_sv) * 8 + sb;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if ((rc = scratch_reserve(lane.l->in, fb + 64))) return rc;
        if ((rc = scratch_reserve(lane.l->mid, small + 64))) return rc;
        if ((rc = scratch_reserve(lane.l->out, (size_t)n_frames * 8))) return rc;
    }
    double *d_small = (double *)lane.l->mid.p;
    double *d_mean = d_small, *d_scale = d_small + n_dims, *d_coef = d_small + 2 * n_dims, *d_sv = d_coef + n_sv;
    HIP_TRY(hipMemcpyAsync(lane.l->in.p, feats, fb, hipMemcpyHostToDevice, cs()));
    HIP_TRY(hipMemcpyAsync(d_mean, mean, (size_t)n_dims * 8, hipMemcpyHostToDevice, cs()));
    HIP_TRY(hipMemcpyAsync(d_scale, scale, (size_t)n_dims * 8, hipMemcpyHostToDevice, cs()));
    HIP_TRY(hipMemcpyAsync(d_coef, dual_coef, (size_t)n_sv * 8, hipMemcpyHostToDevice, cs()));
    HIP_TRY(hipMemcpyAsync(d_sv, support_vectors, sb, hipMemcpyHostToDevice, cs()));
    hipLaunchKernelGGL(svm_binary_proba_kernel, dim3((unsigned)((n_frames + 255) / 256)), dim3(256), 0, cs(),
                       (const double *)lane.l->in.p, n_dims, (long long)n_frames, (long long)n_frames, d_mean, d_scale, d_sv,
                       d_coef, n_sv, intercept, gamma, prob_a, prob_b, (double *)lane.l->out.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(prob1, lane.l->out.p, (size_t)n_frames * 8, hipMemcpyDeviceToHost, cs()));
    HIP_TRY(hipStreamSynchronize(cs()));
    return PAA_OK;
}

// ------------------------------------------------------------------------------------------
// RCCL gather (one process per GPU; librccl is loaded lazily so CPU-only hosts can load us)
// ------------------------------------------------------------------------------------------
#include "comm_rccl.hpp"

// ------------------------------------------------------------------------------------------
// introspection for tests
// ------------------------------------------------------------------------------------------
// per-phase cycle totals of st_fast_800 (only in builds with -DPAA_F800_TIMING; zeros otherwise); resets them
extern "C" int paa_debug_phase_cycles(uint64_t *out16) {
    if (!out16) return fail(PAA_ERR_ARG, "null");
    for (int i = 0; i < 16; ++i) out16[i] = 0;
#if defined(PAA_F800_TIMING) || defined(PAA_F800_TRACE)
    int rc = ensure_init();
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(cs()));
    unsigned long long host[16];
    HIP_TRY(hipMemcpyFromSymbol(host, HIP_SYMBOL(f800::g_phase_cycles), sizeof(host)));
    for (int i = 0; i < 16; ++i) out16[i] = host[i];
    unsigned long long zero[16] = {0};
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(f800::g_phase_cycles), zero, sizeof(zero)));
#endif
    return PAA_OK;
}

// highest number of host-buffer calls that were in flight at the same time since the last query (lanes, see Lane);
// resets the mark.  Lets a test show that calls from several threads really overlap.
extern "C" int paa_debug_lane_peak(void) {
    std::lock_guard<std::mutex> lk(g_lane_mu);
    const int p = g_lanes_peak;
    g_lanes_peak = g_lanes_active;
    return p;
}

// per-wave trace of the last st_fast_800 launch (PAA_F800_TIMING builds): 4 words per run, up to 4096 runs
extern "C" int paa_debug_wave_trace(uint64_t *out, int max_waves) {
    if (!out || max_waves < 1) return fail(PAA_ERR_ARG, "null");
#if defined(PAA_F800_TIMING) || defined(PAA_F800_TRACE)
    int rc = ensure_init();
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(cs()));
    const size_t n = (size_t)std::min(max_waves, 4096) * 4 * sizeof(unsigned long long);
    HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(f800::g_wave_trace), n));
    return std::min(max_waves, 4096);
#else
    return 0;
#endif
}

extern "C" int paa_debug_mel_bank(double fs, int num_fft, double *out_dense) {
    if (!out_dense || num_fft < 1) return fail(PAA_ERR_ARG, "bad argument");
    MelTable t;
    int rc = build_mel(fs, num_fft, t);
    if (rc) return fail(rc, "mel filter bank indexes bin >= num_fft");
    std::fill(out_dense, out_dense + (size_t)kNumMel * num_fft, 0.0);
    for (int m = 0; m < kNumMel; ++m)
        for (int i = 0; i < t.cnt[m]; ++i) out_dense[(size_t)m * num_fft + t.lo[m] + i] = t.w[t.off[m] + i];
    return PAA_OK;
}
extern "C" int paa_debug_dct(double *out_13x40) {
    if (!out_13x40) return fail(PAA_ERR_ARG, "null");
    build_dct(out_13x40);
    return PAA_OK;
}
extern "C" int paa_debug_chroma(double fs, int num_fft, int capacity, int32_t *src, double *weight, int32_t *slot) {
    ChromaTable t;
    int rc = build_chroma(fs, num_fft, t);
    if (rc) return fail(rc, "chroma table error");
    const int n = (int)t.flat_src.size();
    if (n > capacity) return fail(PAA_ERR_ARG, "capacity %d < %d entries", capacity, n);
    for (int i = 0; i < n; ++i) { src[i] = t.flat_src[i]; weight[i] = t.flat_w[i]; slot[i] = t.flat_slot[i]; }
    return n;
}
extern "C" int paa_debug_run_plan(const int64_t *frames, int64_t n_clips, int quantum, int min_run, int max_run, int halo,
                                  int wg_runs, int num_cu, int32_t *run_cap, int64_t *n_runs, int32_t *longest) {
    if (!frames || n_clips < 0 || quantum < 1 || min_run < quantum || max_run < min_run || halo < 0 || wg_runs < 1 ||
        num_cu < 1 || !run_cap || !n_runs || !longest)
        return fail(PAA_ERR_ARG, "bad argument");
    std::vector<ClipDev> clips((size_t)n_clips);
    for (int64_t c = 0; c < n_clips; ++c) { memset(&clips[(size_t)c], 0, sizeof(ClipDev)); clips[(size_t)c].T = (int)frames[c]; }
    const int cap = choose_run_cap(clips, quantum, min_run, max_run, halo, wg_runs, num_cu);
    long long runs = 0;
    int lmax = 0;
    for (const ClipDev &c : clips) {
        if (c.T <= 0) continue;
        const int len = clip_run_length(c.T, cap, quantum);
        runs += (c.T + len - 1) / len;
        lmax = std::max(lmax, len);
    }
    *run_cap = cap; *n_runs = runs; *longest = lmax;
    return PAA_OK;
}
// the same for the kernels whose halo rides inside a run's first iteration (2 RA RB family): every run of a clip but the
// first is `shrink` frames shorter, which is what the tile list does -- the run count follows that rule
extern "C" int paa_debug_run_plan_shrink(const int64_t *frames, int64_t n_clips, int quantum, int min_run, int max_run,
                                         int shrink, int wg_runs, int num_cu, int32_t *run_cap, int64_t *n_runs,
                                         int32_t *longest) {
    if (!frames || n_clips < 0 || quantum < 1 || min_run < quantum || max_run < min_run || shrink < 0 || wg_runs < 1 ||
        num_cu < 1 || !run_cap || !n_runs || !longest)
        return fail(PAA_ERR_ARG, "bad argument");
    std::vector<ClipDev> clips((size_t)n_clips);
    for (int64_t c = 0; c < n_clips; ++c) { memset(&clips[(size_t)c], 0, sizeof(ClipDev)); clips[(size_t)c].T = (int)frames[c]; }
    const int cap = choose_run_cap(clips, quantum, min_run, max_run, 0, wg_runs, num_cu, shrink);
    long long runs = 0;
    int lmax = 0;
    for (const ClipDev &c : clips) {
        if (c.T <= 0) continue;
        const int len = clip_run_length(c.T, cap, quantum);
        for (long long t0 = 0; t0 < c.T; ++runs) t0 += (t0 > 0) ? std::max(len - shrink, 1) : len;      // the tile rule
        lmax = std::max(lmax, len);
    }
    *run_cap = cap; *n_runs = runs; *longest = lmax;
    return PAA_OK;
}
// host side of the mixed-radix kernel for a window (no device needed): radix schedule of the in-place DIF transform and the
// position that holds Z[k] afterwards.  Returns the number of passes, 0 when the window is not for that kernel.
extern "C" int paa_debug_mix_plan(int window, int32_t *radices, int32_t *fft_len, uint16_t *perm, int perm_capacity,
                                  int32_t *waves, int32_t *tw_global) {
    if (window < 2 || !radices || !fft_len) return fail(PAA_ERR_ARG, "bad argument");
    FftPlan p;
    build_fft_plan(window, p);
    *fft_len = p.len;
    mix::MixLayout L;
    if (!mix::mix_layout(p, nullptr, nullptr, 34, L, nullptr)) return 0;
    std::vector<int> radix(L.radix, L.radix + L.n_pass);
    for (int i = 0; i < L.n_pass; ++i) radices[i] = L.radix[i];
    if (perm && perm_capacity >= p.len) {
        std::vector<unsigned short> pm;
        mix::mix_permutation(p.len, radix, pm);
        memcpy(perm, pm.data(), (size_t)p.len * 2);
    }
    if (waves) *waves = L.waves;
    if (tw_global) *tw_global = L.tw_global;
    return L.n_pass;
}
extern "C" int paa_debug_fft_plan(int window, int32_t *radices, int32_t *fft_len) {
    if (window < 2 || !radices || !fft_len) return fail(PAA_ERR_ARG, "bad argument");
    FftPlan p;
    build_fft_plan(window, p);
    *fft_len = p.len;
    const int n = (int)p.radix.size();
    for (int i = 0; i < n && i < 32; ++i) radices[i] = p.radix[i];
    return n;
}
