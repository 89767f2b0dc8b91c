// Plans: clip descriptors, statistics chunks, run lengths and tile lists, the kernel choice (lib_dispatch.hpp) and the
// device-resident plan API (paa_plan_create / _execute / _mid_execute / _beat_execute ...).  One of the units paa_lib.hip is
// made of (included there, after the library state; not a translation unit of its own).
#pragma once
// ------------------------------------------------------------------------------------------
// plans
// ------------------------------------------------------------------------------------------
#ifndef PAA_STAT_CHUNK
#define PAA_STAT_CHUNK 131072         // (A/B of scripts/rounds/r05/gpu_r05aj.sh against 65536: two workgroups per CU for the one-hour clip instead of
                                      // four leave the pass at 20.4 us and halve the partials every wave of the feature kernel folds
                                      // in its prologue: 259.9 -> 258.5 us.  int32 partial sums: 512 samples of |v| < 2^16 per thread)
#endif
constexpr int kStatChunk = PAA_STAT_CHUNK;          // samples per statistics workgroup (upper bound, see stat_chunk_for)
// The statistics pass is an HBM-bound stream with every workgroup resident at once (8 per CU): 879 chunks of 64 K samples of a
// one-hour clip put 4 workgroups on some CUs and 3 on others, and the pass lasts as long as the CUs with 4.  A batch of at least
// one chunk per CU is therefore cut into a whole multiple of num_cu chunks (512 x 112 512 samples for the hour).
static int stat_chunk_for(long long total_samples, int num_cu) {
    const long long blocks = (total_samples + kStatChunk - 1) / kStatChunk;
    if (blocks < num_cu) {
        // a short batch (one 60 s stereo clip = 41 chunks of 64 K) left 215 CUs idle and each workgroup with sixteen dependent
        // rounds of loads: 16.7 us for 10.6 MB.  Four chunks per CU of at least 4096 samples (one round of loads) instead.
        const long long len = ((total_samples + 4LL * num_cu - 1) / (4LL * num_cu) + 63) / 64 * 64;
        return (int)std::min<long long>(kStatChunk, std::max<long long>(len, 4096));
    }
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

// Run lengths of a plan that fills LESS than one round of a one-workgroup-per-CU kernel (the one-hour clip: 2000 equal runs of
// 72 frames = 250 workgroups of eight on 256 CUs, scripts/experiments/run_geometry.py: 256 workgroups take the same time).  Every
// clip is re-cut into its share of num_cu x wg_runs runs whose ITERATION counts (quanta) differ by at most one; runs after a
// clip's first hold `shrink` halo frames inside their first iteration, i.e. store that many frames less.  Along the order
// (workgroup, SIMD, wave of the SIMD) long and short runs alternate, so the waves that share a SIMD (w, w + 4, ...) get long and
// short ones in the plan's overall ratio.  Returns false (lens untouched) when the equal runs stay.
static bool balanced_runs(const std::vector<ClipDev> &clips, int run, int quantum, int shrink, int wg_runs, int num_cu, int min_run,
                          std::vector<std::vector<int>> &lens) {
    if (wg_runs < 2 || quantum < 1 || shrink < 0 || (shrink > 0 && shrink >= quantum)) return false;
    long long total = 0, runs = 0;
    std::vector<long long> need(clips.size(), 0);
    for (size_t c = 0; c < clips.size(); ++c) {
        const long long T = clips[c].T;
        if (T <= 0) continue;
        total += T;
        const long long len = clip_run_length(T, run, quantum), later = std::max<long long>(len - shrink, 1);
        need[c] = (T <= len) ? 1 : 1 + (T - len + later - 1) / later;          // the tile rule of the equal runs
        runs += need[c];
    }
    const long long slots = (long long)num_cu * wg_runs;
    if (runs >= slots || (runs + wg_runs - 1) / wg_runs > num_cu || total < slots * min_run) return false;
    // runs per clip: proportional to its frames (at least what the cap asks for), never shorter than min_run frames
    std::vector<long long> k(clips.size(), 0);
    long long given = 0;
    for (size_t c = 0; c < clips.size(); ++c) {
        const long long T = clips[c].T;
        if (T <= 0) continue;
        k[c] = std::max<long long>(need[c], std::min<long long>(slots * T / total, std::max<long long>(T / min_run, 1)));
        given += k[c];
    }
    if (given > slots) return false;
    std::vector<std::vector<int>> out(clips.size());
    // waves w, w + 4, w + 8 ... of a workgroup share a SIMD (profiles/r05_fast800_wave_trace.txt): per = waves per SIMD
    const int per = (wg_runs % 4 == 0) ? wg_runs / 4 : 0;
    long long first_tile = 0;          // tile index of the clip's first run (runs of all clips are laid out consecutively)
    for (size_t c = 0; c < clips.size(); ++c) {
        const long long T = clips[c].T, kc = k[c];
        if (T <= 0) continue;
        const long long iters = (T + (kc - 1) * shrink + quantum - 1) / quantum, base = iters / kc, longs = iters % kc;
        if (base < 1 || (base + (longs ? 1 : 0)) * quantum > run) return false;
        std::vector<int> &l = out[c];
        l.assign((size_t)kc, (int)(base * quantum));
        // position u of the (workgroup, SIMD, member) order -> run; `longs` of the kc positions get one more iteration, spread evenly
        std::vector<std::pair<long long, long long>> order;
        order.reserve((size_t)kc);
        for (long long i = 0; i < kc; ++i) {
            const long long tile = first_tile + i, w = tile % wg_runs, j = tile / wg_runs;
            order.emplace_back(per ? (j * 4 + w % 4) * per + w / 4 : tile, i);
        }
        std::sort(order.begin(), order.end());
        for (long long u = 0; u < kc; ++u)
            if ((u + 1) * longs / kc > u * longs / kc) l[(size_t)order[(size_t)u].second] += quantum;
        long long sum = 0;
        for (size_t i = 0; i < l.size(); ++i) { if (i > 0) l[i] -= shrink; sum += l[i]; }
        l.back() -= (int)(sum - T);          // the iterations cover whole quanta: the last run gives back what the clip does not have
        if (l.back() <= 0) return false;
        first_tile += kc;
    }
    lens.swap(out);
    return true;
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
    unsigned char *d_gen_blob = nullptr;
    bool blob_cached = false;        // d_gen_blob belongs to the table set's FamilyChoice (not freed with the plan)
    void *d_block = nullptr;         // the plan's one device block: d_clips, d_tiles, d_chunks, d_norms, d_psum / pmin / pmax point into it
    int big = 0;                     // window beyond the LDS envelope of the one-wave-per-frame kernels
    void *d_big = nullptr;
    size_t big_bytes = 0;
    int wg = 0;                      // ... whose transform still fits ONE WORKGROUP's LDS (kernels_wg.hpp); else HBM passes (kernels_big.hpp)
    wg::WgLayout wl;
    std::vector<wg::FrameRef> wg_frames;              // every frame of the plan, chunk after chunk (a chunk's rows fit the scratch)
    std::vector<std::pair<long long, long long>> wg_chunks;     // [first, last) into wg_frames
    wg::FrameRef *d_wg_frames = nullptr;
    std::vector<wg::FrameRef> wg_tasks;               // split transforms (wl.r0 > 0): (frame, sub-transform pair) records, chunk after chunk
    std::vector<std::pair<long long, long long>> wg_task_chunks;
    wg::FrameRef *d_wg_tasks = nullptr;
    unsigned short *d_wg_perm = nullptr;
    long long wg_rows = 0;           // spectrum rows of the largest chunk
    int wgs_r0 = 0, wgs_q = 0;       // r0 > 0: the split runs on kernels_wgs.hpp (r0 x q samples: 44 100, 22 050, 48 000, 32 000, 24 000): wg_tasks holds (frame, task type) records
    int wgr = 0;                     // > 0: shape id of the fused three-pass kernel (kernels_wgr.hpp): 16 000- / 8 000-sample windows
    std::vector<Tile> wgr_runs;      // runs of consecutive frames, about one per CU
    Tile *d_wgr_runs = nullptr;
    wgr::WgrTab *d_wgr_tab = nullptr;      // mel constants + chroma lists of the plan's (fs, window)
    long long mid_off_step = -1;
    long long n_tiles = 0, n_chunks = 0;
    size_t lds = 0;
    int fast = 0;                    // 1: specialised kernel
    FastLaunch fl;
    int mixk = 0;                    // 1: in-place mixed-radix kernel (kernels_mix.hpp); table blob in d_gen_blob
    mix::MixLayout ml;
    int bluk = 0;                    // 1: Bluestein kernel (kernels_blu.hpp); table blob (LDS + global tables) in d_gen_blob
    blu::BluLayout bl;
    int ct = 0;                      // 1: register-FFT family for windows 2 RA RB (kernels_ct.hpp); table blob in d_gen_blob
    ct::CtLaunch cl;
    int tri = 0;                     // 1: three-pass register FFT for the large default windows (kernels_tri.hpp); blob in d_gen_blob
    tri::TriLaunch trl;
    int family = -1;                 // index into kFamilies (lib_dispatch.hpp); -1: the big-window path
    std::vector<Tile> tiles_host;    // host copy of the tile list (plans built for a ranged launch only)
    std::string kernel_name;
};

static std::atomic<int> g_live_plans{0};          // plans hold raw pointers into the device's table sets (freed outside g_mu too)
static void plan_free(paa_plan *p) {
    if (!p) return;
    --g_live_plans;
    // (the caller has synchronised the stream the plan ran on: pooled blocks may be handed to the next plan at once)
    pool_free(p->d_block);          // clips, tiles, statistics chunks / partials, clip constants: one pooled block
    pool_free(p->d_mid_off);
    pool_free(p->d_wg_frames);
    pool_free(p->d_wg_tasks);
    pool_free(p->d_wg_perm);
    pool_free(p->d_wgr_runs);
    pool_free(p->d_wgr_tab);
    if (!p->blob_cached) pool_free(p->d_gen_blob);
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

#include "lib_dispatch.hpp"

// ranges > 1: the caller will launch the plan's tiles in that many consecutive groups (run_host_st's copy-back pipeline)
static int plan_build(const int64_t *offsets, int64_t n_clips, int sample_kind, double fs, int window, int step,
                      int deltas, int mode, paa_plan **out, int ranges = 1) {
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

    // ---- kernel choice + tiles: the first family of kFamilies (lib_dispatch.hpp) that takes the shape; its run rule
    FamilyCtx fc{p.get(), tab, fs, window, step, deltas, mode, sample_kind, F, total_frames, ranges};
    RunRule rr;
    rc = choose_family(fc, rr);
    if (rc) return rc;
    const int run = rr.run, run_quantum = rr.quantum, run_halo = rr.halo_inside;
    std::vector<Tile> tiles;
    tiles.reserve((size_t)(total_frames / run + n_clips));
    std::vector<std::vector<int>> run_lens;
#ifndef PAA_BALANCED_RUNS
#define PAA_BALANCED_RUNS 1           // (0: A/B build of scripts/rounds/r05/gpu_r05w.sh -- equal runs, 250 workgroups for the one-hour clip)
#endif
    const bool balanced = PAA_BALANCED_RUNS && rr.fill_wg_runs > 0 && ranges <= 1 &&
                          balanced_runs(p->clips, run, run_quantum, run_halo, rr.fill_wg_runs, g_num_cu, rr.fill_min_run, run_lens);
    for (int64_t c = 0; c < n_clips; ++c) {
        const long long T = p->clips[c].T;
        if (T <= 0) continue;
        if (balanced) {
            long long t0 = 0;
            for (int cnt : run_lens[(size_t)c]) {
                Tile tl; tl.clip = (int)c; tl.t0 = (int)t0; tl.cnt = cnt; tl.pad = 0;
                tiles.push_back(tl);
                t0 += cnt;
            }
            continue;
        }
        const int len = clip_run_length(T, run, run_quantum);          // equal runs per clip
        for (long long t0 = 0; t0 < T;) {
            const long long want = (t0 > 0) ? len - run_halo : len;
            Tile tl; tl.clip = (int)c; tl.t0 = (int)t0; tl.cnt = (int)std::min<long long>(want, T - t0); tl.pad = 0;
            tiles.push_back(tl);
            t0 += tl.cnt;
        }
    }
    p->n_tiles = (long long)tiles.size();
    if (ranges > 1) p->tiles_host = tiles;          // (the host pipeline cuts the list at frame boundaries)
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
    // ONE pooled device block and ONE upload per plan (the host-buffer entry points build a plan per call):
    //   [clip descriptors | tiles | statistics chunks] (uploaded) [clip constants | partial sums | minima | maxima]
    {
        const size_t nch = (size_t)std::max<long long>(n_chunks, 1);
        auto up = [](size_t b) { return (b + 255) / 256 * 256; };
        const size_t o_clips = 0, o_tiles = o_clips + up(p->clips.size() * sizeof(ClipDev));
        const size_t o_chunks = o_tiles + up(std::max<size_t>(tiles.size(), 1) * sizeof(Tile));
        const size_t o_norms = o_chunks + up(std::max<size_t>(chunks.size(), 1) * sizeof(StatChunk));
        const size_t o_sum = o_norms + up((size_t)n_clips * sizeof(ClipNorm));
        const size_t o_min = o_sum + up(nch * 8), o_max = o_min + up(nch * 8), total = o_max + up(nch * 8);
        if ((rc = pool_alloc(&p->d_block, total))) return rc;
        std::vector<unsigned char> stage(o_norms, 0);
        memcpy(stage.data() + o_clips, p->clips.data(), p->clips.size() * sizeof(ClipDev));
        if (!tiles.empty()) memcpy(stage.data() + o_tiles, tiles.data(), tiles.size() * sizeof(Tile));
        if (!chunks.empty()) memcpy(stage.data() + o_chunks, chunks.data(), chunks.size() * sizeof(StatChunk));
        HIP_TRY(hipMemcpy(p->d_block, stage.data(), o_norms, hipMemcpyHostToDevice));
        unsigned char *b = reinterpret_cast<unsigned char *>(p->d_block);
        p->d_clips = reinterpret_cast<ClipDev *>(b + o_clips);
        p->d_tiles = reinterpret_cast<Tile *>(b + o_tiles);
        p->d_chunks = reinterpret_cast<StatChunk *>(b + o_chunks);
        p->d_norms = reinterpret_cast<ClipNorm *>(b + o_norms);
        p->d_psum = b + o_sum; p->d_pmin = b + o_min; p->d_pmax = b + o_max;
    }
    // windows beyond the one-wave kernels whose transform fits one workgroup's LDS: the frame list of kernels_wg.hpp
    std::unique_ptr<wgr::WgrTab> wgr_tab;
    if (p->big && wgr::wgr_shape_id(window)) {
        wgr_tab.reset(new wgr::WgrTab());
        if (!wgr::wgr_build_tab(wgr::wgr_threads(wgr::wgr_shape_id(window)), mode == 0 ? &tab->mel : nullptr,
                                mode != 1 ? &tab->chroma : nullptr, *wgr_tab))
            wgr_tab.reset();          // (a mel bank this kernel's lane jobs cannot hold: kernels_wg.hpp takes the window)
    }
    if (wgr_tab) {
        // the 1 s windows of music_thumbnailing at 16 / 8 kHz: one fused launch, the transform in registers (kernels_wgr.hpp)
        p->wgr = wgr::wgr_shape_id(window);
        if ((rc = upload_pooled(&p->d_wgr_tab, wgr_tab.get(), 1))) return rc;
        wgr::wgr_build_runs(p->clips, g_num_cu, p->wgr_runs);
        if (p->wgr_runs.size() > 0x7fffffffULL) return fail(PAA_ERR_UNSUPPORTED, "too many runs for one launch");
        if ((rc = upload_pooled(&p->d_wgr_runs, p->wgr_runs.data(), std::max<size_t>(p->wgr_runs.size(), 1)))) return rc;
        p->kernel_name = std::string(mode == 0 ? "st" : (mode == 1 ? "spectrogram" : "chromagram")) + "_wgr_" + wgr::wgr_shape_name(p->wgr);
    } else if (p->big) {
        std::vector<unsigned short> perm;
        if (wg::wg_layout(tab->fft, p->wl, perm)) {
            // spectrum scratch: one row of Nf doubles per frame of a chunk, at most 1 GiB; a chunk that starts inside a clip
            // begins with that clip's previous frame once more (halo: only its spectrum row is wanted)
            const long long cap = std::max<long long>(2, ((long long)1 << 30) / ((long long)Nf * 8));
            long long first = 0;
            for (int64_t c = 0; c < n_clips; ++c)
                for (long long t = 0; t < p->clips[c].T; ++t) {
                    long long in_chunk = (long long)p->wg_frames.size() - first;
                    if (in_chunk >= cap) {
                        p->wg_chunks.emplace_back(first, (long long)p->wg_frames.size());
                        first = (long long)p->wg_frames.size();
                        in_chunk = 0;
                        if (t > 0 && mode != 1) p->wg_frames.push_back(wg::FrameRef{(int)c, (int)(t - 1), 0, 1});
                    }
                    p->wg_frames.push_back(wg::FrameRef{(int)c, (int)t, (int)((long long)p->wg_frames.size() - first), 0});
                }
            if ((long long)p->wg_frames.size() > first) p->wg_chunks.emplace_back(first, (long long)p->wg_frames.size());
            for (auto &ch : p->wg_chunks) {
                p->wg_rows = std::max(p->wg_rows, ch.second - ch.first);
                for (long long i = ch.first; i < ch.second; ++i) p->wg_frames[(size_t)i].row = (int)(i - ch.first);
            }
            if ((rc = upload_pooled(&p->d_wg_frames, p->wg_frames.data(), std::max<size_t>(p->wg_frames.size(), 1)))) return rc;
            if (p->wl.r0 && wgs::wgs_select(window).r0) {
                // the real-input split on register passes (kernels_wgs.hpp): the tasks of a frame side by side (FrameRef::halo = halo | type << 8)
                p->wgs_r0 = wgs::wgs_select(window).r0;
                p->wgs_q = wgs::wgs_select(window).q;
                const int n_types = wgs::wgs_task_types(p->wgs_r0);
                for (auto &ch : p->wg_chunks) {
                    const long long t0 = (long long)p->wg_tasks.size();
                    auto push = [&](long long i, int ty) {
                        wg::FrameRef f = p->wg_frames[(size_t)i];
                        f.halo |= ty << 8;
                        p->wg_tasks.push_back(f);
                    };
                    if (p->wgs_r0 == 6) {
                        // three sub-transforms per frame: {1, 2} and the packed one -- the packed units of two CONSECUTIVE frames of a clip (consecutive
                        // rows) share a task (type 1, on the first frame's record); a frame without such a partner runs its packed unit alone (type 2)
                        for (long long i = ch.first; i < ch.second;) {
                            const wg::FrameRef &a = p->wg_frames[(size_t)i];
                            const bool pair = i + 1 < ch.second && p->wg_frames[(size_t)i + 1].clip == a.clip && p->wg_frames[(size_t)i + 1].t == a.t + 1 &&
                                              p->wg_frames[(size_t)i + 1].row == a.row + 1;
                            push(i, 0);
                            if (pair) { push(i + 1, 0); push(i, 1); i += 2; }
                            else { push(i, 2); i += 1; }
                        }
                    } else {
                        for (long long i = ch.first; i < ch.second; ++i)
                            for (int ty = 0; ty < n_types; ++ty) push(i, ty);
                    }
                    p->wg_task_chunks.emplace_back(t0, (long long)p->wg_tasks.size());
                }
                if (p->wg_tasks.size() > 0x7fffffffULL) return fail(PAA_ERR_UNSUPPORTED, "too many frames for the split transform");
                if ((rc = upload_pooled(&p->d_wg_tasks, p->wg_tasks.data(), std::max<size_t>(p->wg_tasks.size(), 1)))) return rc;
            } else if (p->wl.r0) {
                // tasks of a frame: sub-transform 0 alone, the pairs {q, r0 - q}, r0 / 2 alone (FrameRef::halo = halo | q << 8)
                const int r0 = p->wl.r0;
                for (auto &ch : p->wg_chunks) {
                    const long long t0 = (long long)p->wg_tasks.size();
                    // (the pairs first: they cost twice what the single sub-transforms do, and tasks are handed out in list order)
                    for (int pairs = 1; pairs >= 0; --pairs)
                        for (long long i = ch.first; i < ch.second; ++i)
                            for (int q = 0; 2 * q <= r0; ++q) {
                                if ((q != 0 && 2 * q != r0) != (pairs != 0)) continue;
                                wg::FrameRef f = p->wg_frames[(size_t)i];
                                f.halo |= q << 8;
                                p->wg_tasks.push_back(f);
                            }
                    p->wg_task_chunks.emplace_back(t0, (long long)p->wg_tasks.size());
                }
                if (p->wg_tasks.size() > 0x7fffffffULL) return fail(PAA_ERR_UNSUPPORTED, "too many frames for the split transform");
                if ((rc = upload_pooled(&p->d_wg_tasks, p->wg_tasks.data(), std::max<size_t>(p->wg_tasks.size(), 1)))) return rc;
            }
            if ((rc = upload_pooled(&p->d_wg_perm, perm.data(), perm.size()))) return rc;
            p->wg = 1;
            p->kernel_name = p->wgs_r0 ? std::string(mode == 0 ? "st" : (mode == 1 ? "spectrogram" : "chromagram")) + "_wgs_" + std::to_string(p->wgs_r0) + "x" + std::to_string(p->wgs_q)
                           : p->wl.r0 ? ((mode == 0) ? "st_wg_split_fft" : (mode == 1 ? "spectrogram_wg_split_fft" : "chromagram_wg_split_fft"))
                                      : ((mode == 0) ? "st_wg_lds_fft" : (mode == 1 ? "spectrogram_wg_lds_fft" : "chromagram_wg_lds_fft"));
        }
    }
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

// paa_prof_enable(n): every n-th feature-kernel launch is bracketed by an event pair on the calling thread's stream (g_mu held
// by the caller); the closing event is recorded when the scope ends, i.e. right behind the launch
struct ProfScope {
    hipEvent_t stop = nullptr;
    int begin() {
        if (!(g_prof && (g_prof_seen++ % g_prof) == 0)) return PAA_OK;
        if (g_prof_used == g_prof_ev.size()) {
            hipEvent_t a, b;
            HIP_TRY(hipEventCreate(&a));
            HIP_TRY(hipEventCreate(&b));
            g_prof_ev.emplace_back(a, b);
        }
        const hipEvent_t start = g_prof_ev[g_prof_used].first;
        stop = g_prof_ev[g_prof_used].second;
        ++g_prof_used;
        HIP_TRY(hipEventRecord(start, cs()));
        return PAA_OK;
    }
    ~ProfScope() { if (stop) (void)hipEventRecord(stop, cs()); }
};

// windows beyond the one-wave kernels whose transform fits one workgroup's LDS (kernels_wg.hpp): per chunk of frames one
// launch for the spectra of ALL its frames and one for their features, then one for the delta rows of all clips
template <typename T>
static int run_wg(paa_plan *p, const void *d_packed, double *d_out) {
    const PlanDev &P = p->P;
    const size_t psum_row = p->wgs_r0 ? (size_t)(p->wgs_r0 / 2) * 4 : 0;                            // kernels_wgs.hpp: the units' partial sums of a row
    const size_t need = (size_t)p->wg_rows * ((size_t)P.Nf * 8 + 24 + psum_row * 8) + 256;       // (+ the task counter of the split transform)
    bool fresh = false;
    if (need > p->big_bytes) {
        if (p->d_big) { HIP_TRY(hipStreamSynchronize(cs())); (void)hipFree(p->d_big); p->d_big = nullptr; p->big_bytes = 0; }
        HIP_TRY(hipMalloc(&p->d_big, need));
        p->big_bytes = need;
        fresh = true;
    }
    double *spec = reinterpret_cast<double *>(p->d_big);
    double *tfeat = spec + (size_t)p->wg_rows * P.Nf;
    double *psum = tfeat + 3 * (size_t)p->wg_rows;
    int *task_counter = reinterpret_cast<int *>(psum + psum_row * (size_t)p->wg_rows);
    // (kernels_wgs.hpp: one counter per XCD segment + the workgroups that are done; its last workgroup leaves them at zero)
    if (fresh && p->wgs_r0) HIP_TRY(hipMemsetAsync(task_counter, 0, 16 * sizeof(int), cs()));
    static LdsAttrCache attr;
    if (!attr.covers((size_t)p->wl.lds_bytes)) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&wg::wg_spectrum_kernel<T, 512>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, p->wl.lds_bytes));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&wg::wg_spectrum_kernel<T, 768>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, p->wl.lds_bytes));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&wg::wg_split_kernel<T, 512>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, p->wl.lds_bytes));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&wg::wg_split_kernel<T, 768>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, p->wl.lds_bytes));
        attr.set((size_t)p->wl.lds_bytes);
    }
    static LdsAttrCache attr_feat;
    if (!attr_feat.covers((size_t)p->wl.feat_lds_bytes)) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&wg::wg_feat_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    p->wl.feat_lds_bytes));
        attr_feat.set((size_t)p->wl.feat_lds_bytes);
    }
    for (size_t ci = 0; ci < p->wg_chunks.size(); ++ci) {
        const auto &ch = p->wg_chunks[ci];
        const unsigned n = (unsigned)(ch.second - ch.first);
        const wg::FrameRef *fr = p->d_wg_frames + ch.first;
        ProfScope prof_scope;          // (bench.py's event pairs bracket the spectrum kernel: the dominant one of this path)
        { const int rc_p = prof_scope.begin(); if (rc_p) return rc_p; }
        if (p->wl.r0) {
            // split transforms: persistent workgroups over (frame, sub-transform pair) tasks, one per CU
            const auto &tc = p->wg_task_chunks[ci];
            const unsigned nt = (unsigned)(tc.second - tc.first);
            const wg::FrameRef *tk = p->d_wg_tasks + tc.first;
            const unsigned grid = std::min<unsigned>(nt, (unsigned)g_num_cu);
            if (!p->wgs_r0) HIP_TRY(hipMemsetAsync(task_counter, 0, sizeof(int), cs()));
            if (p->wgs_r0) {
                if (launch::wgs(p->wgs_r0, p->wgs_q, p->sample_kind, P, d_packed, p->d_clips, p->d_norms, tk, (int)nt, task_counter, g_num_cu, spec, tfeat, psum, d_out, cs()))
                    return fail(PAA_ERR_HIP, "launch of %s failed: %s", p->kernel_name.c_str(), hipGetErrorString(hipGetLastError()));
            } else if (p->wl.threads == 768)
                hipLaunchKernelGGL((wg::wg_split_kernel<T, 768>), dim3(grid), dim3(768), (size_t)p->wl.lds_bytes, cs(), P, p->wl,
                                   p->d_wg_perm, (const T *)d_packed, p->d_clips, p->d_norms, tk, (int)nt, task_counter, spec, d_out);
            else
                hipLaunchKernelGGL((wg::wg_split_kernel<T, 512>), dim3(grid), dim3(512), (size_t)p->wl.lds_bytes, cs(), P, p->wl,
                                   p->d_wg_perm, (const T *)d_packed, p->d_clips, p->d_norms, tk, (int)nt, task_counter, spec, d_out);
            if (prof_scope.stop) { (void)hipEventRecord(prof_scope.stop, cs()); prof_scope.stop = nullptr; }
            if (P.mode == 0 && !p->wgs_r0)          // (kernels_wgs.hpp forms the time-domain features in its {1, 2} tasks)
                hipLaunchKernelGGL((wg::wg_time_kernel<T>), dim3(n), dim3(64 * wg::kTimeWaves), 0, cs(), P, (const T *)d_packed, p->d_clips, p->d_norms,
                                   fr, tfeat);
        } else {
        // persistent workgroups: as many as the LDS footprint lets the chip hold at once, each walks frames b, b + grid, ...
        const unsigned per_cu = (unsigned)std::max<size_t>(1, ((size_t)160 * 1024) / (size_t)p->wl.lds_bytes);
        const unsigned grid = std::min<unsigned>(n, (unsigned)g_num_cu * std::min<unsigned>(per_cu, p->wl.threads == 768 ? 2u : 4u));
        if (p->wl.threads == 768)
            hipLaunchKernelGGL((wg::wg_spectrum_kernel<T, 768>), dim3(grid), dim3(768), (size_t)p->wl.lds_bytes, cs(), P, p->wl,
                               p->d_wg_perm, (const T *)d_packed, p->d_clips, p->d_norms, fr, (int)n, spec, tfeat, d_out);
        else
            hipLaunchKernelGGL((wg::wg_spectrum_kernel<T, 512>), dim3(grid), dim3(512), (size_t)p->wl.lds_bytes, cs(), P, p->wl,
                               p->d_wg_perm, (const T *)d_packed, p->d_clips, p->d_norms, fr, (int)n, spec, tfeat, d_out);
        if (prof_scope.stop) { (void)hipEventRecord(prof_scope.stop, cs()); prof_scope.stop = nullptr; }
        }
        if (P.mode != 1 && p->wgs_r0) {
            if (launch::wgs_feat(p->wgs_r0, p->wgs_q, P, fr, (int)n, p->d_clips, spec, tfeat, psum, d_out, cs()))
                return fail(PAA_ERR_HIP, "launch of the feature kernel of %s failed: %s", p->kernel_name.c_str(), hipGetErrorString(hipGetLastError()));
        } else if (P.mode != 1) {
            if (p->wl.feat_staged)
                hipLaunchKernelGGL(wg::wg_feat_kernel<true>, dim3(n), dim3(wg::kFeatThreads), (size_t)p->wl.feat_lds_bytes, cs(), P, fr,
                                   p->d_clips, spec, tfeat, d_out);
            else
                hipLaunchKernelGGL(wg::wg_feat_kernel<false>, dim3(n), dim3(wg::kFeatThreads), (size_t)p->wl.feat_lds_bytes, cs(), P, fr,
                                   p->d_clips, spec, tfeat, d_out);
        }
        HIP_TRY(hipGetLastError());
    }
    if (P.mode == 0 && P.deltas) {
        long long maxT = 0;
        for (auto &cd : p->clips) maxT = std::max<long long>(maxT, cd.T);
        const long long gx = ((long long)kBase * maxT + 255) / 256;
        if (gx > 0x7fffffffLL)
            return fail(PAA_ERR_UNSUPPORTED, "delta grid too large (%lld frames in one clip)", maxT);
        // gridDim.y holds at most 65 535 clips: larger batches go in blocks (advisor, round 5: they used to be refused here
        // although run_big, which took them before round 5, loops the same way)
        for (long long c0 = 0; c0 < p->n_clips; c0 += 65535) {
            const unsigned ny = (unsigned)std::min<long long>(65535, p->n_clips - c0);
            hipLaunchKernelGGL(wg::wg_delta_kernel, dim3((unsigned)gx, ny), dim3(256), 0, cs(), p->d_clips + c0, d_out);
            HIP_TRY(hipGetLastError());
        }
    }
    return PAA_OK;
}

// the fused three-pass kernel (kernels_wgr.hpp): ONE launch for all frames of all clips, then the delta rows
static int run_wgr(paa_plan *p, const void *d_packed, double *d_out) {
    const PlanDev &P = p->P;
    if (!p->wgr_runs.empty()) {
        ProfScope prof_scope;
        { const int rc_p = prof_scope.begin(); if (rc_p) return rc_p; }
        if (launch::wgr(p->wgr, p->sample_kind, P.mode, P, d_packed, p->d_clips, p->d_norms, p->d_wgr_runs, (long long)p->wgr_runs.size(),
                        g_num_cu, p->d_wgr_tab, d_out, cs()))
            return fail(PAA_ERR_HIP, "launch of %s failed: %s", p->kernel_name.c_str(), hipGetErrorString(hipGetLastError()));
    }
    if (P.mode == 0 && P.deltas) {
        long long maxT = 0;
        for (auto &cd : p->clips) maxT = std::max<long long>(maxT, cd.T);
        const long long gx = ((long long)kBase * maxT + 255) / 256;
        if (gx > 0x7fffffffLL) return fail(PAA_ERR_UNSUPPORTED, "delta grid too large (%lld frames in one clip)", maxT);
        for (long long c0 = 0; c0 < p->n_clips; c0 += 65535) {
            const unsigned ny = (unsigned)std::min<long long>(65535, p->n_clips - c0);
            hipLaunchKernelGGL(wg::wg_delta_kernel, dim3((unsigned)gx, ny), dim3(256), 0, cs(), p->d_clips + c0, d_out);
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
    if (plan->wgr) return run_wgr(plan, d_packed, d_out);
    if (plan->wg)
        return plan->sample_kind == 0 ? run_wg<int16_t>(plan, d_packed, d_out)
             : plan->sample_kind == 2 ? run_wg<stereo16>(plan, d_packed, d_out) : run_wg<double>(plan, d_packed, d_out);
    if (plan->big)
        return plan->sample_kind == 0 ? run_big<int16_t>(plan, d_packed, d_out)
             : plan->sample_kind == 2 ? run_big<stereo16>(plan, d_packed, d_out) : run_big<double>(plan, d_packed, d_out);
    if (plan->n_tiles == 0) return PAA_OK;
    ProfScope prof_scope;
    { const int rc_p = prof_scope.begin(); if (rc_p) return rc_p; }
    if (plan->family < 0) return fail(PAA_ERR_UNSUPPORTED, "plan without a kernel family");
    rc = kFamilies[plan->family].launch(plan, d_packed, d_out, plan->d_tiles, plan->n_tiles, cs());
    if (rc) return fail(PAA_ERR_HIP, "launch of %s failed: %s", plan->kernel_name.c_str(), hipGetErrorString(hipGetLastError()));
    return PAA_OK;
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

// [34][T_c] base slabs of n_clips clips, back to back -> [68][T_c] slabs back to back (delta rows re-formed on the device;
// bit-identical to what a 68-row plan stores).  Queued behind everything on the communication stream when a communicator
// exists (the slabs usually arrive through paa_comm_gatherv_f64), else on the library stream.  The tile list of the last
// frames[] is kept (a sharded job expands the same batch shape step after step).
struct DeltaPlanCache {
    std::vector<int64_t> frames;
    DeltaTile *d_tiles = nullptr;
    long long n_tiles = 0;
};
static DeltaPlanCache g_delta_cache;
static hipStream_t comm_stream_or_null();
static int comm_order_after_compute(hipStream_t s);
extern "C" int paa_dev_expand_deltas(const double *d_base, const int64_t *frames, int64_t n_clips, double *d_out) {
    if (!d_base || !frames || !d_out || n_clips < 1) return fail(PAA_ERR_ARG, "null buffer / no clips");
    if (d_base == d_out) return fail(PAA_ERR_ARG, "in-place expansion is not possible (rows move)");
    { const int rc_init = ensure_init(); if (rc_init) return rc_init; }
    std::lock_guard<std::mutex> lk(g_mu);
    DeltaPlanCache &dc = g_delta_cache;
    if ((int64_t)dc.frames.size() != n_clips || memcmp(dc.frames.data(), frames, (size_t)n_clips * 8) != 0) {
        constexpr int kTile = 2048;
        std::vector<DeltaTile> tiles;
        long long tot = 0;
        for (int64_t c = 0; c < n_clips; ++c) {
            const long long T = frames[c];
            if (T < 0 || T > 0x7fffffffLL) return fail(PAA_ERR_ARG, "clip %lld: %lld frames", (long long)c, T);
            for (long long t0 = 0; t0 < T; t0 += kTile) {
                DeltaTile tl;
                tl.base_off = (long long)kBase * tot; tl.out_off = 2LL * kBase * tot;
                tl.T = (int)T; tl.t0 = (int)t0; tl.cnt = (int)std::min<long long>(kTile, T - t0); tl.pad = 0;
                tiles.push_back(tl);
            }
            tot += T;
        }
        if (tiles.size() > 0x7fffffffULL) return fail(PAA_ERR_UNSUPPORTED, "too many tiles");
        // a launch that still reads the old list may be in flight on ANY lane's stream or on the communication stream (another
        // host thread may have queued it): the list is replaced once per batch shape, so a device-wide wait is affordable
        HIP_TRY(hipDeviceSynchronize());
        const int rc = upload_pooled(&dc.d_tiles, tiles.data(), tiles.size());
        if (rc) return rc;
        dc.n_tiles = (long long)tiles.size();
        dc.frames.assign(frames, frames + n_clips);
    }
    if (dc.n_tiles == 0) return PAA_OK;
    hipStream_t s = comm_stream_or_null();
    if (s) { const int rc_o = comm_order_after_compute(s); if (rc_o) return rc_o; }
    else s = cs();
    hipLaunchKernelGGL(expand_deltas_kernel, dim3((unsigned)dc.n_tiles, (unsigned)kBase), dim3(256), 0, s, dc.d_tiles, d_base, d_out);
    HIP_TRY(hipGetLastError());
    return PAA_OK;
}
