// Self-similarity matrix and thumbnail filter.
// Replaces audioSegmentation.self_similarity_matrix (audioSegmentation.py:40-55) and the matrix part of
// audioSegmentation.music_thumbnailing (:1141-1165) on feature matrices that already live in HBM.
//
//   sim_row_stats   : per feature row: mean and scale exactly as scikit-learn's StandardScaler defines them
//                     (corrected two-pass variance, near-constant rows get scale 1)
//   sim_normalize   : z = (x - mean) / scale -> Z [Dp][ld] (rows padded to a multiple of 4, columns to 128, zeros),
//                     reciprocal column norms 1/|z_t|
//   sim_gram        : G = Z^T Z on the FP64 matrix cores (v_mfma_f64_16x16x4_f64; the one dense contraction of the
//                     whole code base), fused epilogue  sim = 1 - (1 - clip(G_ij / (|z_i| |z_j|)))  , diagonal = 1
//   thumb_diag      : moving sum along the diagonals (= convolve2d with eye(M), 'valid') + per-block minima
//   thumb_min       : folds the per-block minima
//   thumb_mask      : near-diagonal band, lower triangle and limit masks set to the global minimum; per-block
//                     arg-max (first maximum in row-major order, like numpy.argmax)
//   thumb_argmax    : folds the per-block candidates
#pragma once
#include "device_common.hpp"

namespace paa {

constexpr int kSimTile = 128;      // a workgroup (4 waves) produces a 128 x 128 tile, a wave 64 x 64
constexpr int kSimChunk = 32;      // feature rows staged in LDS per step
constexpr int kSimPitch = 132;     // LDS row pitch in doubles (128 + 4: the four k-rows of an operand start 8 banks apart)
constexpr int kDiagRun = 32;       // diagonal cells per thread in thumb_diag
constexpr int kMaskRows = 8;       // thumb_mask: rows per block (x 1024 columns)

// one block per feature row
__global__ __launch_bounds__(256) void sim_row_stats_kernel(const double *__restrict__ x, long long n, long long ld,
                                                             double *__restrict__ mean_out,
                                                             double *__restrict__ scale_out) {
    __shared__ double red[2][4];
    const double *row = x + (long long)blockIdx.x * ld;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double s = 0.0;
    for (long long t = threadIdx.x; t < n; t += 256) s += row[t];
    s = wsum(s);
    if (lane == 0) red[0][wave] = s;
    __syncthreads();
    const double mean = ((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) / (double)n;
    __syncthreads();
    double c = 0.0, q = 0.0;
    for (long long t = threadIdx.x; t < n; t += 256) {
        const double d = row[t] - mean;
        c += d;
        q = fma(d, d, q);
    }
    c = wsum(c);
    q = wsum(q);
    if (lane == 0) { red[0][wave] = c; red[1][wave] = q; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double corr = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        const double ss = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
        const double nn = (double)n;
        const double var = (ss - corr * corr / nn) / nn;
        const double ub = nn * kEps * var + (nn * mean * kEps) * (nn * mean * kEps);
        mean_out[blockIdx.x] = mean;
        scale_out[blockIdx.x] = (var <= ub) ? 1.0 : sqrt(var);
    }
}

// thread per column (also the padding columns n <= t < ld, which are written as zeros)
__global__ __launch_bounds__(256) void sim_normalize_kernel(const double *__restrict__ x, int n_dims, int dims_pad,
                                                             long long n, long long ld_in, long long ld,
                                                             const double *__restrict__ mean,
                                                             const double *__restrict__ scale, double *__restrict__ z,
                                                             double *__restrict__ norm /* 1/|z_t| */) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= ld) return;
    double acc = 0.0;
    for (int d = 0; d < dims_pad; ++d) {
        double v = 0.0;
        if (d < n_dims && t < n) {
            v = (x[(long long)d * ld_in + t] - mean[d]) / scale[d];
            acc = fma(v, v, acc);
        }
        z[(long long)d * ld + t] = v;
    }
    // reciprocal norm: 1/0 = inf keeps scipy's 0/0 = NaN for zero vectors (0 * inf) in the Gram epilogue
    norm[t] = 1.0 / sqrt(acc);
}

typedef double f64x4 __attribute__((ext_vector_type(4)));

// grid = min(tiles^2, 2 per CU) persistent workgroups of 512 threads, dynamic LDS = 2 * kSimChunk * kSimPitch * 8 bytes.
// The workgroup owns a 128 x 128 tile of G = Z^T Z, wave w the 32 x 64 block (rows 32 (w >> 1), columns 64 (w & 1))
// as 2 x 4 MFMA blocks (64 accumulator registers; 4 waves per SIMD so that one wave's store burst or panel wait is
// covered by the others' matrix work).  K (the feature axis) is consumed in chunks of kSimChunk rows: the two
// 128-column panels of Z are copied to LDS with 16-byte loads, then every k-step is 6 LDS reads for 8 MFMAs.
// HBM/L2 traffic: 64 KB per 1.05 Mflop of matrix work; the T^2 output write is the other stream.
__global__ __launch_bounds__(512, 4) void sim_gram_kernel(const double *__restrict__ z, int dims_pad, long long n,
                                                           long long ld, const double *__restrict__ rnorm,
                                                           double *__restrict__ sim) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sim_smem[];
    double *pa = reinterpret_cast<double *>(sim_smem);              // [kSimChunk][kSimPitch]: columns i0 .. i0+127
    double *pb = pa + kSimChunk * kSimPitch;                        // columns j0 .. j0+127
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wi = 32 * (wave >> 1), wj = 64 * (wave & 1);
    const int lm = lane & 15, lk = lane >> 4;
    // persistent workgroups: the stores of one tile drain while the next tile's panels load and multiply
    // (requesting the next tile's first chunk BEFORE the epilogue was measured: the 16 extra live registers spill at
    // the 128-register budget and the kernel gets slower)
    const long long tiles = ld / kSimTile;
    constexpr int PF = kSimChunk * 64 / 512;                         // double2 per thread and panel: 4
    double fax[PF], fay[PF], fbx[PF], fby[PF];                       // scalar arrays: these stay in registers
    const int prow = threadIdx.x >> 6, pc2 = threadIdx.x & 63;       // element e = threadIdx.x + 512 q -> row prow + 8 q
    // rows past the end of Z are replaced by its last row: they land in LDS rows no k-step reads
#define PAA_SIM_FETCH(k0_, i0_, j0_)                                                                         \
    _Pragma("unroll") for (int q = 0; q < PF; ++q) {                                                         \
        const long long grow = min((k0_) + prow + 8 * q, dims_pad - 1);                                      \
        const double2 ta_ = *reinterpret_cast<const double2 *>(z + grow * ld + (i0_) + 2 * pc2);             \
        const double2 tb_ = *reinterpret_cast<const double2 *>(z + grow * ld + (j0_) + 2 * pc2);             \
        fax[q] = ta_.x; fay[q] = ta_.y; fbx[q] = tb_.x; fby[q] = tb_.y;                                      \
    }
    if ((long long)blockIdx.x < tiles * tiles) {
        const long long t0 = blockIdx.x;
        PAA_SIM_FETCH(0, (t0 / tiles) * kSimTile, (t0 % tiles) * kSimTile)
    }
    for (long long tile = blockIdx.x; tile < tiles * tiles; tile += gridDim.x) {
    const long long i0 = (tile / tiles) * kSimTile, j0 = (tile % tiles) * kSimTile;
    f64x4 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f64x4){0.0, 0.0, 0.0, 0.0};
    for (int k0 = 0; k0 < dims_pad; k0 += kSimChunk) {
        const int kc = min(kSimChunk, dims_pad - k0);                // multiple of 4
        __syncthreads();                                             // previous chunk fully consumed
#pragma unroll
        for (int q = 0; q < PF; ++q) {
            *reinterpret_cast<double2 *>(pa + (prow + 8 * q) * kSimPitch + 2 * pc2) = make_double2(fax[q], fay[q]);
            *reinterpret_cast<double2 *>(pb + (prow + 8 * q) * kSimPitch + 2 * pc2) = make_double2(fbx[q], fby[q]);
        }
        __syncthreads();
        if (k0 + kSimChunk < dims_pad) PAA_SIM_FETCH(k0 + kSimChunk, i0, j0)
        // operand maps of v_mfma_f64_16x16x4_f64: A[m = lane & 15][k = lane >> 4], B[k = lane >> 4][n = lane & 15];
        // A = Z^T, so both operands are read with the same (k, column) pattern
        for (int kk = 0; kk < kc; kk += 4) {
            const double *ra = pa + (kk + lk) * kSimPitch + wi + lm;
            const double *rb = pb + (kk + lk) * kSimPitch + wj + lm;
            double av[2], bv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) bv[q] = rb[16 * q];
            av[0] = ra[0];
            av[1] = ra[16];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[a], bv[b], acc[a][b], 0, 0, 0);
        }
    }
    // C/D map of the f64 form: column = lane & 15, row = (lane >> 4) + 4 * reg.  Rows of one lane are 4 apart, so one
    // running pointer (step 4 n) serves the lane's 8 rows of a column block.
    const long long n4 = 4 * n;
    // the lane's 8 row norms and 4 column norms in one batch of loads (rnorm is padded to ld entries), so that the
    // stores below do not each wait for a load of their own
    double rr[8], cc[4];
#pragma unroll
    for (int u = 0; u < 8; ++u) rr[u] = rnorm[i0 + wi + lk + 4 * u];
#pragma unroll
    for (int b = 0; b < 4; ++b) cc[b] = rnorm[j0 + wj + 16 * b + lm];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const long long col = j0 + wj + 16 * b + lm;
        long long row = i0 + wi + lk;
        double *dst = sim + row * n + col;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (row < n && col < n) {
                    // scipy's cosine distance: 1 - clip(u.v / (|u| |v|)); similarity = 1 - distance (:53-54).
                    // The reciprocal norms are multiplied first so that sim[i][j] and sim[j][i] round identically;
                    // 0 * inf = NaN for zero vectors mirrors scipy's 0/0; squareform's zero diagonal -> exactly 1
                    double cosv = acc[a][b][r] * (rr[4 * a + r] * cc[b]);
                    if (fabs(cosv) > 1.0) cosv = copysign(1.0, cosv);
                    const double dist = 1.0 - cosv;
                    *dst = (row == col) ? 1.0 : 1.0 - dist;
                }
                row += 4;
                dst += n4;
            }
    }
    if (tile + gridDim.x < tiles * tiles) {       // first chunk of this workgroup's next tile
        const long long tn = tile + gridDim.x;
        PAA_SIM_FETCH(0, (tn / tiles) * kSimTile, (tn % tiles) * kSimTile)
    }
    }   // tile loop
#undef PAA_SIM_FETCH
}

// out[i][j] = sum_{k < M} S[i + k][j + k],  R = n - M + 1.  Thread (run, c): cells (i0 + s, c + s), s < kDiagRun,
// i0 = run * kDiagRun, start column c in [-(kDiagRun - 1), R): direct sum at its first cell, sliding update after.
__global__ __launch_bounds__(256) void thumb_diag_kernel(const double *__restrict__ sim, long long n, int M,
                                                          long long R, double *__restrict__ out,
                                                          double *__restrict__ block_min) {
    __shared__ double red[4];
    const long long c = (long long)blockIdx.x * 256 + threadIdx.x - (kDiagRun - 1);
    const long long i0 = (long long)blockIdx.y * kDiagRun;
    double vmin = __builtin_inf();
    const int s_lo = c < 0 ? (int)(-c) : 0;
    const long long lim = (R - i0 < R - c) ? R - i0 : R - c;
    const int s_hi = lim < kDiagRun ? (int)lim : kDiagRun;
    if (s_lo < s_hi) {
        const long long dstep = n + 1, ostep = R + 1;
        const double *p = sim + (i0 + s_lo) * n + (c + s_lo);       // S[i][j] of the first cell
        double run = 0.0;
        for (int k = 0; k < M; ++k) run += p[k * dstep];
        double *o = out + (i0 + s_lo) * R + (c + s_lo);
        o[0] = run;
        vmin = run;
        // sliding update: + S[i + M][j + M] - S[i][j]; the loads do not depend on the running sum
        const double *pn = p + (long long)M * dstep;
        const int cnt = s_hi - s_lo - 1;
#pragma unroll 8
        for (int s = 0; s < cnt; ++s) {
            run += pn[s * dstep] - p[s * dstep];
            o[(s + 1) * ostep] = run;
            vmin = fmin(vmin, run);
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    vmin = -wave_max(-vmin);
    if (lane == 0) red[wave] = vmin;
    __syncthreads();
    if (threadIdx.x == 0)
        block_min[(long long)blockIdx.y * gridDim.x + blockIdx.x] = fmin(fmin(red[0], red[1]), fmin(red[2], red[3]));
}

// folds the per-block minima of thumb_diag into min_out[0] (one block)
__global__ __launch_bounds__(1024) void thumb_min_kernel(const double *__restrict__ block_min, long long n_block_min,
                                                          double *__restrict__ min_out) {
    __shared__ double red[16];
    double m = __builtin_inf();
    for (long long b = threadIdx.x; b < n_block_min; b += 1024) m = fmin(m, block_min[b]);
    m = -wave_max(-m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w) m = fmin(m, red[w]);
        min_out[0] = m;
    }
}

// masks (:1149-1160) + per-block arg-max candidates.  grid (ceil(R / 1024), ceil(R / kMaskRows)); a block sweeps
// kMaskRows rows x 1024 columns.  Masked cells are only written, unmasked cells only read.
__global__ __launch_bounds__(256) void thumb_mask_kernel(double *__restrict__ f, long long R, double band,
                                                          long long lim_lo, long long lim_hi,
                                                          const double *__restrict__ min_ptr,
                                                          double *__restrict__ cand_val,
                                                          long long *__restrict__ cand_idx) {
    __shared__ double rv[4];
    __shared__ long long ri[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double min_sm = min_ptr[0];
    double v = -__builtin_inf();
    long long vi = 0x7fffffffffffffffLL;
    for (int rr = 0; rr < kMaskRows; ++rr) {
        const long long i = (long long)blockIdx.y * kMaskRows + rr;
        if (i >= R) break;
        const bool row_masked = i < lim_lo || i >= lim_hi;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const long long j = (long long)blockIdx.x * 1024 + 256 * q + threadIdx.x;
            if (j >= R) continue;
            const long long idx = i * R + j;
            const bool masked = row_masked || ((double)(j - i) < band) || j < lim_lo || j >= lim_hi;   // covers i > j
            double c;
            if (masked) { f[idx] = min_sm; c = min_sm; }
            else c = f[idx];
            // first maximum in row-major order: larger value wins, ties go to the smaller index (idx grows here)
            if (c > v) { v = c; vi = idx; }
        }
    }
    for (int off = 1; off < 64; off <<= 1) {
        const double ov = __shfl_xor(v, off, 64);
        const long long oi = __shfl_xor(vi, off, 64);
        if (ov > v || (ov == v && oi < vi)) { v = ov; vi = oi; }
    }
    if (lane == 0) { rv[wave] = v; ri[wave] = vi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w)
            if (rv[w] > v || (rv[w] == v && ri[w] < vi)) { v = rv[w]; vi = ri[w]; }
        cand_val[(long long)blockIdx.y * gridDim.x + blockIdx.x] = v;
        cand_idx[(long long)blockIdx.y * gridDim.x + blockIdx.x] = vi;
    }
}

__global__ __launch_bounds__(1024) void thumb_argmax_kernel(const double *__restrict__ cand_val,
                                                             const long long *__restrict__ cand_idx, long long n_cand,
                                                             long long *__restrict__ best_idx) {
    __shared__ double rv[16];
    __shared__ long long ri[16];
    double v = -__builtin_inf();
    long long vi = 0x7fffffffffffffffLL;
    for (long long b = threadIdx.x; b < n_cand; b += 1024) {
        const double ov = cand_val[b];
        const long long oi = cand_idx[b];
        if (ov > v || (ov == v && oi < vi)) { v = ov; vi = oi; }
    }
    for (int off = 1; off < 64; off <<= 1) {
        const double ov = __shfl_xor(v, off, 64);
        const long long oi = __shfl_xor(vi, off, 64);
        if (ov > v || (ov == v && oi < vi)) { v = ov; vi = oi; }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { rv[wave] = v; ri[wave] = vi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w)
            if (rv[w] > v || (rv[w] == v && ri[w] < vi)) { v = rv[w]; vi = ri[w]; }
        best_idx[0] = vi;
    }
}

}  // namespace paa
