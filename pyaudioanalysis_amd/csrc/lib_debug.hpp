// Introspection for the tests (paa_debug_*): tables, FFT plans, run-length choices, lane overlap, phase cycles of the timing
// builds.  One of the units paa_lib.hip is made of.
#pragma once
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
    unsigned long long acc[16] = {0};
    // every translation unit with kernels keeps its own counters (family_*.hip)
    if (launch::phase_fast(acc, nullptr, 0) < 0 || launch::phase_ct(acc, nullptr, 0) < 0 || launch::phase_tri_a(acc, nullptr, 0) < 0 ||
        launch::phase_tri_b(acc, nullptr, 0) < 0 || launch::phase_tri_c(acc, nullptr, 0) < 0 || launch::phase_rmg(acc, nullptr, 0) < 0 || launch::phase_blu(acc, nullptr, 0) < 0 ||
        launch::phase_wgr(acc, nullptr, 0) < 0)
        return fail(PAA_ERR_HIP, "reading the phase counters failed");
    for (int i = 0; i < 16; ++i) out16[i] = acc[i];
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
    unsigned long long acc[16] = {0};
    const int n = launch::phase_fast(acc, reinterpret_cast<unsigned long long *>(out), max_waves);      // (also clears that unit's counters)
    if (n < 0) return fail(PAA_ERR_HIP, "reading the wave trace failed");
    return n;
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
// the lane jobs the three-pass kernels cut the mel sums (K = 40 filters) and the chroma gather (K = 12 classes) into
// (kernels_tri.hpp: lane_jobs): jobs256 = 64 x {start, n, woff, ctl}
extern "C" int paa_debug_lane_jobs(const int32_t *first, const int32_t *wfirst, const int32_t *cnt, int n_owners, int32_t *jobs256) {
    if (!first || !wfirst || !cnt || !jobs256 || n_owners < 1 || n_owners > 64) return fail(PAA_ERR_ARG, "bad argument");
    for (int k = 0; k < n_owners; ++k)
        if (cnt[k] < 0) return fail(PAA_ERR_ARG, "negative count");
    tri::lane_jobs(n_owners, first, wfirst, cnt, reinterpret_cast<tri::LaneJob *>(jobs256));
    return PAA_OK;
}
// the run lengths paa_plan_create gives a plan that fills less than one round of a one-workgroup-per-CU kernel (lib_plan.hpp:
// balanced_runs).  Returns the number of runs written to `lens` (clip after clip), 0 when the equal runs stay, < 0 on error.
extern "C" int64_t paa_debug_balanced_runs(const int64_t *frames, int64_t n_clips, int run_cap, int quantum, int shrink, int wg_runs,
                                           int num_cu, int min_run, int32_t *lens, int64_t capacity) {
    if (!frames || n_clips < 0 || !lens || quantum < 1 || run_cap < quantum || wg_runs < 1 || num_cu < 1 || min_run < 1)
        return fail(PAA_ERR_ARG, "bad argument");
    std::vector<ClipDev> clips((size_t)n_clips);
    for (int64_t c = 0; c < n_clips; ++c) { memset(&clips[(size_t)c], 0, sizeof(ClipDev)); clips[(size_t)c].T = (int)frames[c]; }
    std::vector<std::vector<int>> out;
    if (!balanced_runs(clips, run_cap, quantum, shrink, wg_runs, num_cu, min_run, out)) return 0;
    int64_t n = 0;
    for (const auto &l : out)
        for (int v : l) {
            if (n >= capacity) return fail(PAA_ERR_ARG, "capacity %lld too small", (long long)capacity);
            lens[n++] = v;
        }
    return n;
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
// host tables of the fused three-pass kernel (kernels_wgr.hpp), no device needed.  Returns the shape id of the window (0: not its
// window; -1: a table it cannot hold -- the plan then keeps kernels_wg.hpp).  mel_job[512][4]: per thread {first bin, index of its
// weight in the mel table, stride, bins}; mel_fil[40][2]: per filter {first thread, threads}; ch_n[12], ch_src[12][64], ch_w[12][64].
extern "C" int paa_debug_wgr_tables(double fs, int window, int32_t *mel_job, int32_t *mel_fil, int32_t *ch_n, int32_t *ch_src, double *ch_w) {
    const int id = wgr::wgr_shape_id(window);
    if (!id) return 0;
    MelTable mel;
    ChromaTable chroma;
    if (build_mel(fs, window / 2, mel) != PAA_OK || build_chroma(fs, window / 2, chroma) != PAA_OK) return -1;
    std::unique_ptr<wgr::WgrTab> t(new wgr::WgrTab());
    if (!wgr::wgr_build_tab(wgr::wgr_threads(id), &mel, &chroma, *t)) return -1;
    if (mel_job) memcpy(mel_job, t->mel_job, sizeof(t->mel_job));
    if (mel_fil) memcpy(mel_fil, t->mel_fil, sizeof(t->mel_fil));
    if (ch_n) memcpy(ch_n, t->ch_n, sizeof(t->ch_n));
    if (ch_src) memcpy(ch_src, t->ch_src, sizeof(t->ch_src));
    if (ch_w) memcpy(ch_w, t->ch_w, sizeof(t->ch_w));
    return id;
}
extern "C" int paa_debug_wgr_runs(const int64_t *frames, int64_t n_clips, int num_cu, int32_t *runs3, int64_t capacity, int64_t *n_runs) {
    if (!frames || n_clips < 0 || num_cu < 1 || !n_runs) return fail(PAA_ERR_ARG, "bad arguments");
    std::vector<ClipDev> clips((size_t)n_clips);
    for (int64_t c = 0; c < n_clips; ++c) { memset(&clips[(size_t)c], 0, sizeof(ClipDev)); clips[(size_t)c].T = (int)frames[c]; }
    std::vector<Tile> runs;
    wgr::wgr_build_runs(clips, num_cu, runs);
    *n_runs = (int64_t)runs.size();
    for (size_t i = 0; i < runs.size() && (int64_t)i < capacity && runs3; ++i) {
        runs3[3 * i] = runs[i].clip; runs3[3 * i + 1] = runs[i].t0; runs3[3 * i + 2] = runs[i].cnt;
    }
    return PAA_OK;
}

// host side of the workgroup-per-frame kernels (kernels_wg.hpp) for a window, no device needed.  info32: {0: complex points Nc,
// 1: bins Nf, 2: passes, 3: r0 (0: the whole transform in LDS; > 0: split into r0 sub-transforms), 4: elements per (sub-)transform,
// 5: top (elements between pad slots), 6: threads of the spectrum kernel, 7: its LDS bytes, 8: permutation in LDS, 9: feature
// kernel stages the row, 10: its LDS bytes, 11 ...: per pass radix, span, twiddle stride (three ints each)}; perm: the padded LDS
// position of output k of the (sub-)transform.  Returns 1, or 0 when the window is not for these kernels.
extern "C" int paa_debug_wg_plan(int window, int32_t *info32, uint16_t *perm, int perm_capacity) {
    if (window < 2 || !info32) return fail(PAA_ERR_ARG, "bad argument");
    FftPlan p;
    build_fft_plan(window, p);
    wg::WgLayout L;
    std::vector<unsigned short> pm;
    if (!wg::wg_layout(p, L, pm)) return 0;
    memset(info32, 0, 48 * sizeof(int32_t));
    info32[0] = p.len; info32[1] = window / 2; info32[2] = L.n_pass; info32[3] = L.r0; info32[4] = L.r0 ? L.sub : p.len;
    info32[5] = L.top; info32[6] = L.threads; info32[7] = L.lds_bytes; info32[8] = L.perm_lds; info32[9] = L.feat_staged;
    info32[10] = L.feat_lds_bytes;
    for (int i = 0; i < L.n_pass && i < 12; ++i) { info32[11 + 3 * i] = L.radix[i]; info32[12 + 3 * i] = L.span[i]; info32[13 + 3 * i] = L.tws[i]; }
    if (perm) {
        if (perm_capacity < info32[4]) return fail(PAA_ERR_ARG, "perm capacity %d < %d", perm_capacity, info32[4]);
        memcpy(perm, pm.data(), (size_t)info32[4] * 2);
    }
    return 1;
}
// kernels_wgs.hpp: the constants of the real-input split and the bin each element of a unit-major spectrum row holds (host restatement of
// wgs::bin_of)
extern "C" int paa_debug_wgs_plan(int window, int32_t *info16, int32_t *bin_of, int capacity) {
    if (window < 2 || !info16) return fail(PAA_ERR_ARG, "bad argument");
    const wgs::Sel sel = wgs::wgs_select(window);
    const int r0 = sel.r0, q = sel.q;
    if (!r0) return 0;
    memset(info16, 0, 16 * sizeof(int32_t));
    const int nf = window / 2;
    const bool a = q == wgs::S3675::Q;
    info16[0] = r0; info16[1] = q;
    info16[2] = a ? wgs::S3675::R1 : wgs::S4000::R1; info16[3] = a ? wgs::S3675::R2 : wgs::S4000::R2; info16[4] = a ? wgs::S3675::R3 : wgs::S4000::R3;
    info16[5] = a ? wgs::S3675::A : wgs::S4000::A; info16[6] = a ? wgs::S3675::NT : wgs::S4000::NT;
    info16[7] = a ? wgs::S3675::LDS_BYTES : wgs::S4000::LDS_BYTES; info16[8] = wgs::wgs_task_types(r0);
    info16[9] = wgs::LCAP;
    info16[10] = a ? (r0 == 12 ? wgs::feat_lds<12, wgs::S3675::Q>() : wgs::feat_lds<6, wgs::S3675::Q>())
                   : (r0 == 12 ? wgs::feat_lds<12, wgs::S4000::Q>() : (r0 == 8 ? wgs::feat_lds<8, wgs::S4000::Q>() : wgs::feat_lds<6, wgs::S4000::Q>()));
    info16[11] = (nf + 64 * r0 - 1) / (64 * r0); info16[12] = wgs::kFeatT;
    info16[13] = a ? (wgs::S3675::P2K0 ? 1 : 0) : (wgs::S4000::P2K0 ? 1 : 0);
    if (bin_of) {
        if (capacity < nf) return fail(PAA_ERR_ARG, "capacity %d < %d", capacity, nf);
        const int h0 = r0 / 2;
        for (int idx = 0; idx < nf; ++idx) {
            const int u = idx / q, kap = idx % q, mm = (u + 1) + r0 * kap;
            bin_of[idx] = (u == h0 - 1) ? h0 * kap : (mm < nf ? mm : window - mm);
        }
    }
    return 1;
}
// host side of the Bluestein kernel for a window (no device needed): info8 = {log2 M, R0, R1, R2, waves, LDS bytes, table_bytes,
// total_bytes}, offsets3 = {chirp, FFT(b) / M in pass order, pass twiddles} into the blob (LDS tables first, global tables behind them).
// Returns the blob size (0: the window goes to another kernel; blob may be null to query the size).
extern "C" int paa_debug_blu_plan(int window, double fs, int32_t *info8, int32_t *offsets3, unsigned char *blob, int capacity) {
    if (window < 2 || !info8 || !offsets3) return fail(PAA_ERR_ARG, "bad argument");
    (void)fs;
    FftPlan p;
    build_fft_plan(window, p);
    blu::BluLayout L;
    std::vector<unsigned char> b;
    if (!blu::blu_layout(p, nullptr, nullptr, 0, L, &b)) return 0;
    int r[4];
    blu::blu_radices(L.log2m, r);
    // (four passes: r[1], r[2] are the middle passes, r[3] the last; info8[2] = first middle pass | second << 8, info8[3] = last pass)
    info8[0] = L.log2m; info8[1] = r[0]; info8[2] = r[3] ? (r[1] | (r[2] << 8)) : r[1]; info8[3] = r[3] ? r[3] : r[2]; info8[4] = L.waves;
    info8[5] = (int32_t)blu::blu_lds_bytes(L); info8[6] = L.table_bytes; info8[7] = L.total_bytes;
    offsets3[0] = L.off_g_chirp; offsets3[1] = L.off_g_bp; offsets3[2] = L.off_g_tw;
    if (L.packed) info8[0] |= 0x100;          // packed form: W / 2 complex points; the post-twiddles exp(-2 pi i k / W) sit right behind the chirp

    if (blob) {
        if (capacity < (int)b.size()) return fail(PAA_ERR_ARG, "blob capacity %d < %d", capacity, (int)b.size());
        memcpy(blob, b.data(), b.size());
    }
    return (int)b.size();
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

// host side of the three-pass register-FFT kernels for a window (no device needed): the shape (R1, R2, R3, packed, plane row
// pitch P, waves per workgroup, pass-3 lane jobs, LDS bytes) into shape8, and the whole table blob (LDS part followed by the
// global part) with the offsets of its tables into offsets6 = {tw2, p3, g_tw1, g_post, table_bytes, total_bytes}.
// Returns the blob size (0: the window goes to another kernel; blob may be null to query the size).
extern "C" int paa_debug_tri_plan(int window, double fs, int32_t *shape8, int32_t *offsets6, unsigned char *blob, int capacity) {
    if (window < 2 || !shape8 || !offsets6) return fail(PAA_ERR_ARG, "bad argument");
    tri::TriLaunch tl;
    std::vector<unsigned char> b;
    if (!tri::tri_select(window, 1, fs, nullptr, nullptr, tl, b)) return 0;
    switch (tl.shape) {
#define PAA_TRI_DESCRIBE(ID, SH)                                                                               \
        case ID: shape8[0] = tri::SH::R1; shape8[1] = tri::SH::R2; shape8[2] = tri::SH::R3; shape8[3] = (tri::SH::PACKED ? 1 : 0) | (tri::SH::R3P << 8);    \
                 shape8[4] = tri::SH::P; shape8[5] = tri::SH::NW | (tri::SH::H1 << 8) | (tri::SH::H2 << 16); shape8[6] = tri::SH::NJOB3; shape8[7] = (int32_t)tl.lds; break;
        PAA_TRI_SHAPES(PAA_TRI_DESCRIBE)
#undef PAA_TRI_DESCRIBE
        default: return 0;
    }
    const tri::TriLayout &L = tl.layout;
    offsets6[0] = L.off_tw2; offsets6[1] = L.off_p3; offsets6[2] = L.off_g_tw1; offsets6[3] = L.off_g_post;
    offsets6[4] = L.table_bytes; offsets6[5] = L.total_bytes; offsets6[6] = L.off_split;
    if (blob) {
        if (capacity < (int)b.size()) return fail(PAA_ERR_ARG, "capacity %d < %zu", capacity, b.size());
        memcpy(blob, b.data(), b.size());
    }
    return (int)b.size();
}
