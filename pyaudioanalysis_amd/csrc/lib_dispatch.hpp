// Kernel choice of a plan: one table entry per feature-kernel family -- does it take this (window, step, sample type, mode)?
// how long are its runs? how is it launched? -- walked in order by plan_build; the first family that accepts owns the plan.
// Adding a family = one more entry (and its family_<name>.hip).  Included by paa_lib.hip after the plan structure.
#pragma once

struct FamilyCtx {
    paa_plan *p;
    TableSet *tab;
    double fs;
    int window, step, deltas, mode, sample_kind, F;
    long long total_frames;
    int ranges;          // >= 1: the plan will be launched in this many consecutive tile ranges (host pipeline): runs are cut as if
                         // the chip had `ranges` times its CUs, so that every range still fills it in one round
    int num_cu() const { return g_num_cu * (ranges > 1 ? ranges : 1); }
    const MelTable *mel() const { return mode == 0 ? &tab->mel : nullptr; }
    const ChromaTable *chroma() const { return mode != 1 ? &tab->chroma : nullptr; }
};
// how the tile list cuts a clip: runs of `run` frames in multiples of `quantum`; a run with t0 > 0 is halo_inside frames
// shorter (kernels whose look-back frames ride inside the run's first iteration)
struct RunRule {
    int run = 64, quantum = 4, halo_inside = 0;
    int fill_wg_runs = 0;        // > 0: runs per workgroup of a kernel with ONE workgroup per CU -- a plan whose equal runs fill less than
                                 // one round of the chip is re-cut into num_cu x fill_wg_runs runs of two lengths (lib_plan.hpp: balanced_runs)
    int fill_min_run = 16;       // ... none of them shorter than this
};
struct Family {
    const char *id;
    // 1: takes the shape (layout + tables are in the plan, kernel_name and lds set), 0: declines, < 0: error code
    int (*select)(FamilyCtx &c);
    void (*run_rule)(FamilyCtx &c, RunRule &r);
    int (*launch)(paa_plan *p, const void *d_packed, double *d_out, const Tile *tiles, long long n_tiles, hipStream_t stream);
};

static int upload_blob(paa_plan *p, const std::vector<unsigned char> &blob) {
    return upload_pooled(&p->d_gen_blob, blob.data(), blob.size());
}
// the usual rule of the one-frame-per-iteration kernels: about two chip-wide rounds, 8 .. 64 frames per run
static int two_round_run(long long total_frames, int waves, int num_cu) {
    const long long slots = (long long)num_cu * waves * 2;
    const long long per = (total_frames + slots - 1) / slots;
    return (int)std::min<long long>(64, std::max<long long>(8, (per + 3) / 4 * 4));
}

// ---- kernels_fast.hpp: int16, window 800, step 400 / 800, features only
static int fam_fast_select(FamilyCtx &c) {
    if (c.mode != 0 || g_force_generic) return 0;
    const int rc = fast_select(c.window, c.step, c.sample_kind, c.fs, c.tab->fast, c.tab->fft, c.tab->mel, c.tab->chroma, c.p->fl,
                               g_f800_waves);
    if (rc < 0) return fail(rc, "building the tables of the specialised kernel failed");
    if (rc) { c.p->fast = 1; c.p->lds = c.p->fl.lds; c.p->kernel_name = c.p->fl.name; }
    return rc;
}
static void fam_fast_rule(FamilyCtx &c, RunRule &r) {
    // one wave per run of whole 4-frame quads, at most fl.run frames; a run after a clip's first starts its first quad one frame
    // early (two with deltas: the flux of the last halo frame feeds a delta) -- the halo rides inside the first iteration, the run
    // stores that many frames less (kernels_fast.hpp: HALO); see choose_run_cap
    r.quantum = 4;
    r.halo_inside = c.deltas ? 2 : 1;
    r.run = choose_run_cap(c.p->clips, 4, 16, c.p->fl.run, 0, c.p->fl.waves_per_cu, c.num_cu(), r.halo_inside);
    r.fill_wg_runs = c.p->fl.waves_per_cu;
    if (const char *rc_env = experiment_env("PAA_RUN_CAP")) r.run = std::max(16, atoi(rc_env) / 4 * 4);      // A/B experiments only
}
static int fam_fast_launch(paa_plan *p, const void *d_packed, double *d_out, const Tile *tiles, long long n, hipStream_t s) {
    return launch::fast(p->fl, p->P, p->tab->fast, d_packed, p->d_clips, p->d_norms, tiles, n, d_out, s);
}

// ---- kernels_ct.hpp: windows 2 RA RB (800, 640, 400, 320), any step / sample type / mode
static int fam_ct_select(FamilyCtx &c) {
    if (g_force_generic) return 0;
    std::vector<unsigned char> blob;
    if (!ct::ct_select(c.window, c.mode, c.fs, c.tab->fft, c.mel(), c.chroma(), c.p->cl, blob)) return 0;
    const int rc = upload_blob(c.p, blob);
    if (rc) return rc;
    c.p->ct = 1; c.p->lds = c.p->cl.lds; c.p->kernel_name = c.p->cl.name;
    return 1;
}
static void fam_ct_rule(FamilyCtx &c, RunRule &r) {
    // one wave per run, 4 frames per iteration; a run with t0 > 0 starts 1 frame early (2 with deltas) inside its first
    // iteration, so the first run of a clip gets `run` frames and the others run - halo: every run is whole iterations
    r.quantum = 4;
    r.halo_inside = (c.mode == 0) ? (c.deltas ? 2 : 1) : 0;
    r.run = choose_run_cap(c.p->clips, 4, 16, 256, 0, c.p->cl.waves, c.num_cu(), r.halo_inside);
    r.fill_wg_runs = c.p->cl.waves;          // (one workgroup per CU; A/B scripts/rounds/r05/gpu_r05aj.sh: -0.4 ... -2.0 % on the feature shapes)
}
static int fam_ct_launch(paa_plan *p, const void *d_packed, double *d_out, const Tile *tiles, long long n, hipStream_t s) {
    return launch::ct(p->cl, p->sample_kind, p->P, p->d_gen_blob, d_packed, p->d_clips, p->d_norms, tiles, n, d_out, s);
}

// ---- kernels_tri.hpp: three-pass register FFT -- the reference's default 50 ms windows at 48 / 44.1 kHz (2400, 2205), the
// 40 ms ones (1920, 1764), 1600, 1200, config 5's feature matrix (1102) and the odd 551 (50 ms at 11.025 kHz)
static int fam_tri_select(FamilyCtx &c) {
    if (g_force_generic) return 0;
    std::vector<unsigned char> blob;
    if (!tri::tri_select(c.window, c.mode, c.fs, c.mel(), c.chroma(), c.p->trl, blob)) return 0;
    const int rc = upload_blob(c.p, blob);
    if (rc) return rc;
    c.p->tri = 1; c.p->lds = c.p->trl.lds; c.p->kernel_name = c.p->trl.name;
    return 1;
}
static void fam_tri_rule(FamilyCtx &c, RunRule &r) {
    // one wave per run, one frame per iteration; a run with t0 > 0 recomputes 1 frame (2 with deltas) first
    r.quantum = 1;
    r.run = choose_run_cap(c.p->clips, 1, 8, 96, (c.mode == 0) ? (c.deltas ? 2 : 1) : 0, c.p->trl.waves, c.num_cu());
    // (balanced runs -- RunRule::fill_wg_runs = trl.waves -- were A/B-ed here too, scripts/rounds/r05/gpu_r05ad.sh: 3072 runs of 19 / 20
    // frames instead of 3000 of 20 for config 5 changed nothing beyond the noise, 0.3202 / 0.3184 ms: the equal runs stay)
}
static int fam_tri_launch(paa_plan *p, const void *d_packed, double *d_out, const Tile *tiles, long long n, hipStream_t s) {
    return launch::tri(p->trl, p->sample_kind, p->P, p->d_gen_blob, d_packed, p->d_clips, p->d_norms, tiles, n, d_out, s);
}

// ---- kernels_mix.hpp: in-place mixed-radix transform for every other length made of 2, 3, 5, 7, 11, 13
static int fam_mix_select(FamilyCtx &c) {
    if (g_force_generic || experiment_env("PAA_NO_MIX")) return 0;
    std::vector<unsigned char> blob;
    if (!mix::mix_layout(c.tab->fft, c.mel(), c.chroma(), c.F, c.p->ml, &blob)) return 0;
    const int rc = upload_blob(c.p, blob);
    if (rc) return rc;
    c.p->mixk = 1;
    c.p->lds = mix::mix_lds_bytes(c.p->ml);
    c.p->kernel_name = (c.mode == 0) ? "st_mix" : (c.mode == 1 ? "spectrogram_mix" : "chromagram_mix");
    return 1;
}
static void fam_mix_rule(FamilyCtx &c, RunRule &r) {
    // one wave per run, one frame at a time (halo: 1 frame, 2 with deltas)
    r.quantum = 4;
    r.run = two_round_run(c.total_frames, c.p->ml.waves, c.num_cu());
}
static int fam_mix_launch(paa_plan *p, const void *d_packed, double *d_out, const Tile *tiles, long long n, hipStream_t s) {
    return launch::mix(p->ml, p->lds, p->sample_kind, p->P, p->d_gen_blob, d_packed, p->d_clips, p->d_norms, tiles, n, d_out, s);
}

// ---- kernels_blu.hpp: lengths with a prime factor above 13 (661, 1103, 736 ...): Bluestein's convolution on power-of-two transforms
static int fam_blu_select(FamilyCtx &c) {
    if (g_force_generic || experiment_env("PAA_NO_BLU")) return 0;
    std::vector<unsigned char> blob;
    if (!blu::blu_layout(c.tab->fft, c.mel(), c.chroma(), c.F, c.p->bl, &blob)) return 0;
    const int rc = upload_blob(c.p, blob);
    if (rc) return rc;
    c.p->bluk = 1;
    c.p->lds = blu::blu_lds_bytes(c.p->bl);
    static const char *names[3][6] = {{"st_blu_256", "st_blu_512", "st_blu_1024", "st_blu_2048", "st_blu_4096", "st_blu_8192"},
                                      {"spectrogram_blu_256", "spectrogram_blu_512", "spectrogram_blu_1024", "spectrogram_blu_2048", "spectrogram_blu_4096",
                                       "spectrogram_blu_8192"},
                                      {"chromagram_blu_256", "chromagram_blu_512", "chromagram_blu_1024", "chromagram_blu_2048", "chromagram_blu_4096",
                                       "chromagram_blu_8192"}};
    static const char *names_p[3][4] = {{"st_blu_512p", "st_blu_1024p", "st_blu_2048p", "st_blu_4096p"},
                                        {"spectrogram_blu_512p", "spectrogram_blu_1024p", "spectrogram_blu_2048p", "spectrogram_blu_4096p"},
                                        {"chromagram_blu_512p", "chromagram_blu_1024p", "chromagram_blu_2048p", "chromagram_blu_4096p"}};
    c.p->kernel_name = c.p->bl.packed ? names_p[c.mode][c.p->bl.log2m - 9] : names[c.mode][c.p->bl.log2m - 8];
    return 1;
}
static void fam_blu_rule(FamilyCtx &c, RunRule &r) {
    // one wave per run, one frame at a time (halo: 1 frame, 2 with deltas)
    r.quantum = 4;
    r.run = two_round_run(c.total_frames, c.p->bl.waves, c.num_cu());
}
static int fam_blu_launch(paa_plan *p, const void *d_packed, double *d_out, const Tile *tiles, long long n, hipStream_t s) {
    return launch::blu(p->bl, p->lds, p->sample_kind, p->P, p->d_gen_blob, d_packed, p->d_clips, p->d_norms, tiles, n, d_out, s);
}

// ---- kernels_generic.hpp: Stockham passes in LDS (what no other family takes: tiny windows, prime factors above 13 beyond 2730 samples); windows beyond the
// LDS envelope take the same passes through HBM scratch (kernels_big.hpp: plan->big, no CPU fallback)
static int fam_generic_select(FamilyCtx &c) {
    std::vector<unsigned char> blob;
    generic_layout(c.tab->fft, c.mel(), c.chroma(), c.F, c.p->gl, &blob);
    c.p->lds = generic_lds_bytes(c.p->gl);
    if (c.p->lds > 160 * 1024) {
        c.p->big = 1;
        c.p->lds = 0;
    } else {
        const int rc = upload_blob(c.p, blob);
        if (rc) return rc;
    }
    c.p->kernel_name = c.p->big ? "big_window_hbm_passes"
                                : (c.mode == 0) ? "st_generic" : (c.mode == 1 ? "spectrogram_generic" : "chromagram_generic");
    return 1;
}
static void fam_generic_rule(FamilyCtx &c, RunRule &r) {
    r.quantum = 4;
    r.run = two_round_run(c.total_frames, c.p->gl.waves, c.num_cu());
}
static int fam_generic_launch(paa_plan *p, const void *d_packed, double *d_out, const Tile *tiles, long long n, hipStream_t s) {
    return launch::generic(p->gl, p->lds, p->sample_kind, p->P, p->d_gen_blob, d_packed, p->d_clips, p->d_norms, tiles, n, d_out, s);
}

static const Family kFamilies[] = {
    {"fast", fam_fast_select, fam_fast_rule, fam_fast_launch},
    {"ct", fam_ct_select, fam_ct_rule, fam_ct_launch},
    {"tri", fam_tri_select, fam_tri_rule, fam_tri_launch},
    {"mix", fam_mix_select, fam_mix_rule, fam_mix_launch},
    {"blu", fam_blu_select, fam_blu_rule, fam_blu_launch},
    {"generic", fam_generic_select, fam_generic_rule, fam_generic_launch},
};
constexpr int kNumFamilies = (int)(sizeof(kFamilies) / sizeof(kFamilies[0]));

// what a select() left in the plan, kept with the table set of (fs, window): the next plan of the same (mode, rows) copies
// it back instead of rebuilding the layout and uploading the table blob again
struct FamilyChoice {
    int family = -1;
    int fast = 0, ct = 0, tri = 0, mixk = 0, bluk = 0, big = 0;
    FastLaunch fl;
    ct::CtLaunch cl;
    tri::TriLaunch trl;
    mix::MixLayout ml;
    blu::BluLayout bl;
    GenLayout gl;
    size_t lds = 0;
    std::string kernel_name;
    unsigned char *d_blob = nullptr;       // owned here (pool_free in free_family_choices)
};
static void free_family_choices(TableSet &t) {
    for (auto &kv : t.choices)
        if (kv.second && kv.second->d_blob) pool_free(kv.second->d_blob);
    t.choices.clear();
}

static int choose_family(FamilyCtx &c, RunRule &rr) {
    paa_plan *p = c.p;
    // the fast family looks at (step, sample type) too; nobody else does
    const int fast_key = (c.mode == 0 && c.window == 800 && c.sample_kind == 0 && (c.step == 400 || c.step == 800)) ? c.step : 0;
    const auto key = std::make_tuple(c.mode, c.F, fast_key);
#ifndef PAA_EXPERIMENTS          // (experiment switches change the choice from plan to plan: no cache in those builds)
    auto it = c.tab->choices.find(key);
    if (it != c.tab->choices.end() && !g_force_generic) {
        const FamilyChoice &fc = *it->second;
        p->family = fc.family;
        p->fast = fc.fast; p->ct = fc.ct; p->tri = fc.tri; p->mixk = fc.mixk; p->bluk = fc.bluk; p->big = fc.big;
        p->fl = fc.fl; p->cl = fc.cl; p->trl = fc.trl; p->ml = fc.ml; p->bl = fc.bl; p->gl = fc.gl;
        p->lds = fc.lds; p->kernel_name = fc.kernel_name;
        p->d_gen_blob = fc.d_blob; p->blob_cached = true;
        kFamilies[p->family].run_rule(c, rr);
        return PAA_OK;
    }
#endif
    for (int i = 0; i < kNumFamilies; ++i) {
        const int rc = kFamilies[i].select(c);
        if (rc < 0) return rc;
        if (rc == 0) continue;
        p->family = i;
        kFamilies[i].run_rule(c, rr);
#ifndef PAA_EXPERIMENTS
        if (!g_force_generic) {
            auto fc = std::make_shared<FamilyChoice>();
            fc->family = i;
            fc->fast = p->fast; fc->ct = p->ct; fc->tri = p->tri; fc->mixk = p->mixk; fc->bluk = p->bluk; fc->big = p->big;
            fc->fl = p->fl; fc->cl = p->cl; fc->trl = p->trl; fc->ml = p->ml; fc->bl = p->bl; fc->gl = p->gl;
            fc->lds = p->lds; fc->kernel_name = p->kernel_name;
            fc->d_blob = p->d_gen_blob;            // ownership moves to the table set
            p->blob_cached = true;
            c.tab->choices[key] = fc;
        }
#endif
        return PAA_OK;
    }
    return fail(PAA_ERR_UNSUPPORTED, "no kernel family takes window %d", c.window);
}
