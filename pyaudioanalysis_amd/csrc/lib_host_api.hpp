// Host-buffer entry points (NumPy in -> NumPy out): short-term / mid-term features, spectrogram / chromagram, the batched
// forms, and silence_removal's probabilistic SVM.  Every call runs in its own lane (stream + scratch).  One of the units
// paa_lib.hip is made of.
#pragma once
// ------------------------------------------------------------------------------------------
// host-buffer entry points
// ------------------------------------------------------------------------------------------
constexpr long long kRangedMinFrames = 65536;       // a range must still fill the chip: four ranges of >= 16 k frames

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
    // One long clip, whole matrix wanted: the frames are computed in kCopyRanges consecutive tile ranges and range k is copied
    // back (a 2-D copy of its columns) while range k + 1 computes -- the copy-back (0.71 ms for an hour at 34 rows) hides
    // the kernels instead of following them.  The clip-global normalisation still pins upload -> statistics -> first range.
    const long long frames_1 = (n_clips == 1) ? paa_num_frames(offsets[1] - offsets[0], window, step) : 0;
    const bool ranged = n_clips == 1 && out && !out_offsets && !want_mid && frames_1 >= kRangedMinFrames;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        rc = plan_build(offsets, n_clips, sample_kind, fs, window, step, deltas, 0, &plan, ranged ? kCopyRanges : 1);
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
    if (ranged && plan->family >= 0 && !plan->big && !plan->tiles_host.empty()) {
        double *d_out = (double *)lane.l->out.p;
        const long long T = plan->clips[0].T;
        const int F = plan->P.F;
        // every exit path -- an error return in the middle of the loops included -- waits for the copies already queued on the
        // lane's copy stream: they write into the CALLER's buffer (plan_free_synced only waits for the compute stream)
        struct CopyStreamGuard { hipStream_t s; ~CopyStreamGuard() { if (s) (void)hipStreamSynchronize(s); } } copy_guard{lane.l->copy_stream};
        {
            std::lock_guard<std::mutex> lk(g_mu);
            if ((rc = launch_stats(plan, d_samples))) return rc;
        }
        // tile ranges at (about) equal frame counts; tiles are in frame order
        long long first = 0;
        long long t_begin[kCopyRanges + 1];
        int used = 0;
        for (int r = 0; r < kCopyRanges && first < plan->n_tiles; ++r) {
            const long long t_target = (r == kCopyRanges - 1) ? T : T * (r + 1) / kCopyRanges;
            long long last = first;
            while (last < plan->n_tiles && (long long)plan->tiles_host[(size_t)last].t0 < t_target) ++last;
            if (last == first) continue;
            t_begin[used] = plan->tiles_host[(size_t)first].t0;
            {
                std::lock_guard<std::mutex> lk(g_mu);
                ProfScope prof_scope;          // (paa_prof_read counts the ranged launches like any other)
                if ((rc = prof_scope.begin())) return rc;
                rc = kFamilies[plan->family].launch(plan, d_samples, d_out, plan->d_tiles + first, last - first, cs());
            }
            if (rc) return fail(PAA_ERR_HIP, "launch of %s failed: %s", plan->kernel_name.c_str(), hipGetErrorString(hipGetLastError()));
            HIP_TRY(hipEventRecord(lane.l->range_done[used], cs()));
            first = last;
            ++used;
        }
        t_begin[used] = T;
        for (int r = 0; r < used; ++r) {
            const long long a = t_begin[r], b = t_begin[r + 1];
            HIP_TRY(hipStreamWaitEvent(lane.l->copy_stream, lane.l->range_done[r], 0));
            HIP_TRY(hipMemcpy2DAsync(out + a, (size_t)T * 8, d_out + a, (size_t)T * 8, (size_t)(b - a) * 8, (size_t)F,
                                     hipMemcpyDeviceToHost, lane.l->copy_stream));
        }
        HIP_TRY(hipStreamSynchronize(lane.l->copy_stream));
        HIP_TRY(hipStreamSynchronize(cs()));
        return PAA_OK;
    }
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
            const int n_tail = (int)(filled - plan->clips[0].T);
            const size_t spill = chroma_tail_spill_bytes(plan->P, n_tail);
            if (spill) {
                std::lock_guard<std::mutex> lk(g_mu);
                if ((rc = scratch_reserve(lane.l->mid, spill))) return rc;
            }
            rc = launch_chroma_tail(plan->P, sample_kind, d_samples, pos, n, n_tail, plan->d_norms,
                                    (double *)lane.l->out.p + (long long)plan->clips[0].T * 12,
                                    spill ? (double *)lane.l->mid.p : nullptr, cs());
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

// MidTermFeatures.beat_extraction (MidTermFeatures.py:18-84 + utilities.peakdet) of ONE short-term matrix in host memory:
// feats [n_rows][n_frames] (rows 0..18 are read, :30-31), window_size = short-term step in seconds; bpm_ratio receives
// (bpm, confidence).  The same beat_kernel the batched directory walkers run on matrices that never leave HBM.
extern "C" int paa_beat_extraction_f64(const double *feats, int n_rows, int64_t n_frames, double window_size, double *bpm_ratio) {
    if (!feats || !bpm_ratio) return fail(PAA_ERR_ARG, "null buffer");
    if (n_rows < 19) return fail(PAA_ERR_ARG, "beat extraction reads short-term rows 0..18: %d rows given", n_rows);
    if (n_frames < 1 || n_frames > 0x7fffffffLL) return fail(PAA_ERR_ARG, "n_frames=%lld", (long long)n_frames);
    if (!(window_size > 0)) return fail(PAA_ERR_ARG, "window_size must be positive");
    { const int rc0 = ensure_init(); if (rc0) return rc0; }
    const int max_beat = (int)nearbyint(2.0 / window_size);          // int(round(2.0 / window_size)), :33
    if (max_beat < 1 || max_beat > 4096) return fail(PAA_ERR_UNSUPPORTED, "beat histogram of %d bins", max_beat);
    const size_t lds = (size_t)kBeatRows * (kBeatTile + 1) * 8 + (size_t)kBeatRows * max_beat * 4;
    if (lds > 160 * 1024)
        return fail(PAA_ERR_UNSUPPORTED, "beat histogram of %d bins needs %zu bytes of LDS (160 KB per workgroup)", max_beat, lds);
    LaneGuard lane;
    const size_t rows_bytes = (size_t)19 * (size_t)n_frames * 8;      // rows 0..18 are contiguous at the head of the matrix
    int rc;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if ((rc = scratch_reserve(lane.l->in, rows_bytes))) return rc;
        if ((rc = scratch_reserve(lane.l->mid, 256))) return rc;
        if (lds > 64 * 1024)
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&beat_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    ClipDev cd;
    memset(&cd, 0, sizeof(cd));
    cd.T = (int)n_frames;
    unsigned char *small = reinterpret_cast<unsigned char *>(lane.l->mid.p);          // [ClipDev | pad | 2 doubles]
    HIP_TRY(hipMemcpyAsync(lane.l->in.p, feats, rows_bytes, hipMemcpyHostToDevice, cs()));
    HIP_TRY(hipMemcpyAsync(small, &cd, sizeof(cd), hipMemcpyHostToDevice, cs()));
    hipLaunchKernelGGL(beat_kernel, dim3(1), dim3(64), lds, cs(), reinterpret_cast<const ClipDev *>(small),
                       (const double *)lane.l->in.p, window_size, max_beat, reinterpret_cast<double *>(small + 128));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(bpm_ratio, small + 128, 16, hipMemcpyDeviceToHost, cs()));
    HIP_TRY(hipStreamSynchronize(cs()));
    return PAA_OK;
}
