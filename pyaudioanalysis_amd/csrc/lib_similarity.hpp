// Host entry points of the self-similarity matrix / thumbnail filter (audioSegmentation.py:40-55, 1141-1165) over
// kernels_sim.hpp.  One of the units paa_lib.hip is made of.
#pragma once
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
