// Self-similarity matrix and thumbnail filter.
// Replaces audioSegmentation.self_similarity_matrix (audioSegmentation.py:40-55) and the matrix part of
// audioSegmentation.music_thumbnailing (:1141-1165) on feature matrices that already live in HBM.
//
//   sim_row_stats   : per feature row: mean and scale exactly as scikit-learn's StandardScaler defines them
//                     (corrected two-pass variance, near-constant rows get scale 1)
//   sim_normalize   : z = (x - mean) / scale -> Z [Dp][ld] (rows padded to a multiple of 4, columns to 128, zeros),
//                     reciprocal column norms 1/|z_t|
//   sim_gram        : G = Z^T Z on the FP64 matrix cores (v_mfma_f64_16x16x4_f64; the one dense contraction of the
//                     whole code base), fused epilogue  sim = 1 - (1 - clip(G_ij / (|z_i| |z_j|)))  , diagonal = 1;
//                     only the tiles on and above the diagonal are multiplied, the others are their mirror images
//                     (transposed through LDS in the epilogue)
//   thumb_diag      : moving sum along the diagonals (= convolve2d with eye(M), 'valid') of the UPPER triangle only
//                     (the matrix is symmetric and everything below the diagonal is masked afterwards), fused with the
//                     masks (:1149-1160): unmasked cells are written and compete for the arg-max, masked cells only
//                     enter the minimum; per-block minima and arg-max candidates
//   thumb_min       : folds the per-block minima
//   thumb_fill      : masked cells (lower triangle, near-diagonal band, limits) = the global minimum
//   thumb_argmax    : folds the per-block candidates (first maximum in row-major order, NaN first, like
//                     numpy.argmax), including the degenerate case where a masked cell is the first maximum
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

// row-major enumeration of the tiles on and above the diagonal: row ti holds tiles - ti of them
__device__ __forceinline__ void sim_tri_tile(long long t, long long tiles, long long &ti, long long &tj) {
    const double b = 2.0 * (double)tiles + 1.0;
    long long r = (long long)((b - sqrt(b * b - 8.0 * (double)t)) * 0.5);
    r = r < 0 ? 0 : (r >= tiles ? tiles - 1 : r);
    while (r + 1 < tiles && (r + 1) * (2 * tiles - r) / 2 <= t) ++r;      // off(r) = r (2 tiles - r + 1) / 2
    while (r > 0 && r * (2 * tiles - r + 1) / 2 > t) --r;
    ti = r;
    tj = r + (t - r * (2 * tiles - r + 1) / 2);
}

// grid = min(tiles (tiles + 1) / 2, 2 per CU) persistent workgroups of 512 threads, dynamic LDS = 2 * kSimChunk *
// kSimPitch * 8 + 2048 bytes.  The workgroup owns a 128 x 128 tile of G = Z^T Z with tile row <= tile column, wave w the 32 x 64
// block (rows 32 (w >> 1), columns 64 (w & 1)) as 2 x 4 MFMA blocks (64 accumulator registers; 4 waves per SIMD so that
// one wave's store burst or panel wait is covered by the others' matrix work).  K (the feature axis) is consumed in
// chunks of kSimChunk rows: the two 128-column panels of Z are copied to LDS with 16-byte loads, then every k-step is
// 6 LDS reads for 8 MFMAs.  A tile above the diagonal is stored twice: as computed, and transposed (16 x 16 blocks
// through a per-wave LDS scratch, so that both stores write 128-byte row segments) -- sim[i][j] and sim[j][i] are the
// same bits, and the matrix work is halved.
// HBM/L2 traffic: 64 KB per 1.05 Mflop of matrix work; the T^2 output write is the other stream and the floor.
// (non-temporal stores were measured: 5 % faster at T = 8192, 40 % slower at T = 3001 where the rows are not line aligned)
__global__ __launch_bounds__(512, 4) void sim_gram_kernel(const double *__restrict__ z, int dims_pad, long long n,
                                                           long long ld, const double *__restrict__ rnorm,
                                                           double *__restrict__ sim) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sim_smem[];
    double *pa = reinterpret_cast<double *>(sim_smem);              // [kSimChunk][kSimPitch]: columns i0 .. i0+127
    double *pb = pa + kSimChunk * kSimPitch;                        // columns j0 .. j0+127
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // (uniform: scalar address math)
    const int wi = 32 * (wave >> 1), wj = 64 * (wave & 1);
    const int lm = lane & 15, lk = lane >> 4;
    double *nrm = pb + kSimChunk * kSimPitch;                       // [256]: 1/|z| of the tile's 128 rows, then its 128 columns
    double *tr = pa + wave * (16 * 17);                             // the wave's transposition scratch (panels are dead then)
    // persistent workgroups: the stores of one tile drain while the next tile's panels load and multiply
    // (requesting the next tile's first chunk BEFORE the epilogue was measured: the 16 extra live registers spill at
    // the 128-register budget and the kernel gets slower)
    const long long tiles = ld / kSimTile, n_tri = tiles * (tiles + 1) / 2;
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
    long long ti = 0, tj = 0;
    if ((long long)blockIdx.x < n_tri) {
        sim_tri_tile(blockIdx.x, tiles, ti, tj);
        PAA_SIM_FETCH(0, ti * kSimTile, tj * kSimTile)
    }
    for (long long tile = blockIdx.x; tile < n_tri; tile += gridDim.x) {
    const long long i0 = ti * kSimTile, j0 = tj * kSimTile;
    const bool mirror = ti != tj;
    f64x4 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f64x4){0.0, 0.0, 0.0, 0.0};
    for (int k0 = 0; k0 < dims_pad; k0 += kSimChunk) {
        const int kc = min(kSimChunk, dims_pad - k0);                // multiple of 4
        __syncthreads();                                             // previous chunk (or transposition scratch) consumed
#pragma unroll
        for (int q = 0; q < PF; ++q) {
            *reinterpret_cast<double2 *>(pa + (prow + 8 * q) * kSimPitch + 2 * pc2) = make_double2(fax[q], fay[q]);
            *reinterpret_cast<double2 *>(pb + (prow + 8 * q) * kSimPitch + 2 * pc2) = make_double2(fbx[q], fby[q]);
        }
        // (rnorm is padded to ld entries; kept in LDS so that the epilogue holds no norms in registers)
        if (k0 == 0) {
            int tn = threadIdx.x;                                    // (opaque: rnorm + 8 tid is not kept across the tile loop)
            asm volatile("" : "+v"(tn));
            if (tn < 256) nrm[tn] = rnorm[(tn < 128 ? i0 : j0 - 128) + tn];
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
    if (mirror) __syncthreads();                  // every wave is done with the panels: their LDS becomes the scratch
    // C/D map of the f64 form: column = lane & 15, row = (lane >> 4) + 4 * reg.
    // addresses = uniform tile / block base (scalar registers) + one 32-bit lane offset shared by every store:
    // direct   sim[i0 + wi + 16 a + 4 r + lk][j0 + wj + 16 b + lm],
    // mirrored sim[j0 + wj + 16 b + 4 r + lk][i0 + wi + 16 a + lm]
    // (lane and wave coordinates are made opaque per tile: everything below is invariant across the persistent tile loop,
    // and hoisted out of it -- 64 addresses, 32 diagonal predicates -- it is spilled around the matrix work)
    int elm = lm, elk = lk, ewi = wi, ewj = wj;
    asm volatile("" : "+v"(elm), "+v"(elk), "+s"(ewi), "+s"(ewj));
    const unsigned lane_off = (unsigned)(elk * n + elm);
    double *const tile_d = sim + i0 * n + j0, *const tile_m = sim + j0 * n + i0;
    const long long rem_i = n - i0 - ewi, rem_j = n - j0 - ewj;       // valid rows / columns from the wave's block origin
#pragma unroll
    for (int b = 0; b < 4; ++b) {
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            // (one block at a time: without the fence the scheduler hoists every block's norm loads and address math
            // above the first store and the 64 accumulator registers leave no room for them)
            __builtin_amdgcn_sched_barrier(0);
            const double ccb = nrm[128 + ewj + 16 * b + elm];
            const bool on_diag_tile = !mirror;
            double v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                // scipy's cosine distance: 1 - clip(u.v / (|u| |v|)); similarity = 1 - distance (:53-54).
                // The reciprocal norms are multiplied first so that sim[i][j] and sim[j][i] round identically
                // (diagonal tiles); 0 * inf = NaN for zero vectors mirrors scipy's 0/0; squareform's zero diagonal
                // -> exactly 1
                double cosv = acc[a][b][r] * (nrm[ewi + 16 * a + elk + 4 * r] * ccb);
                if (fabs(cosv) > 1.0) cosv = copysign(1.0, cosv);
                const double dist = 1.0 - cosv;
                v[r] = (on_diag_tile && ewi + 16 * a + 4 * r + elk == ewj + 16 * b + elm) ? 1.0 : 1.0 - dist;
                if (16 * a + 4 * r + elk < rem_i && 16 * b + elm < rem_j)
                    (tile_d + ((long long)(ewi + 16 * a + 4 * r) * n + ewj + 16 * b))[lane_off] = v[r];
            }
            if (mirror) {
                // block (rows R0 + elk + 4 r, column C0 + elm) -> its transpose: lane (elm, elk) takes element
                // (row elm, column elk + 4 r) and stores it at sim[C0 + elk + 4 r][R0 + elm]
#pragma unroll
                for (int r = 0; r < 4; ++r) tr[(elk + 4 * r) * 17 + elm] = v[r];
                wsync();
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double w = tr[elm * 17 + elk + 4 * r];
                    if (16 * b + 4 * r + elk < rem_j && 16 * a + elm < rem_i)
                        (tile_m + ((long long)(ewj + 16 * b + 4 * r) * n + ewi + 16 * a))[lane_off] = w;
                }
                wsync();
            }
        }
    }
    if (tile + gridDim.x < n_tri) {               // first chunk of this workgroup's next tile
        // (opaque per tile: hoisted out of the persistent loop, (double)(2 tiles + 1) lived in scratch at the 128-register budget)
        long long tiles_o = tiles;
        asm volatile("" : "+s"(tiles_o));
        sim_tri_tile(tile + gridDim.x, tiles_o, ti, tj);
        PAA_SIM_FETCH(0, ti * kSimTile, tj * kSimTile)
    }
    }   // tile loop
#undef PAA_SIM_FETCH
}

// numpy's order for min / argmax with NaN: NaN wins both; among equals the smaller flat index wins
__device__ __forceinline__ double nan_min(double a, double b) { return (a != a) ? a : ((b != b) ? b : fmin(a, b)); }
__device__ __forceinline__ bool argmax_better(double v_new, long long i_new, double v, long long i) {
    const bool nn = v_new != v_new, no = v != v;
    if (nn != no) return nn;
    if (nn) return i_new < i;
    return v_new > v || (v_new == v && i_new < i);
}
// (i, j) masked by :1149-1160: near-diagonal band, lower triangle, the limit_1 / limit_2 margins
__device__ __forceinline__ bool thumb_masked(long long i, long long j, double band, long long lim_lo, long long lim_hi) {
    return i > j || fabs((double)(j - i)) < band || i < lim_lo || i >= lim_hi || j < lim_lo || j >= lim_hi;
}

// out[i][j] = sum_{k < M} S[i + k][j + k],  R = n - M + 1, for j >= i only (S is symmetric bit for bit, so the lower
// triangle holds the same sums -- and is masked afterwards anyway).  Block (bx, by): rows i0 = by * kDiagRun .. + kDiagRun,
// thread = diagonal offset d = j - i = 256 bx + tid: cells (i0 + s, i0 + d + s), s < kDiagRun; direct sum at its first
// cell, sliding update after.  Every cell enters the minimum (the reference takes it BEFORE masking); unmasked cells are
// written and compete for the arg-max; masked cells are left to thumb_fill.
__global__ __launch_bounds__(256) void thumb_diag_kernel(const double *__restrict__ sim, long long n, int M,
                                                          long long R, double band, long long lim_lo, long long lim_hi,
                                                          double *__restrict__ out, double *__restrict__ block_min,
                                                          double *__restrict__ cand_val, long long *__restrict__ cand_idx) {
    __shared__ double red[4];
    __shared__ double rv[4];
    __shared__ long long ri[4];
    const long long i0 = (long long)blockIdx.y * kDiagRun;
    const long long d = (long long)blockIdx.x * 256 + threadIdx.x;
    double vmin = __builtin_inf();
    double v = -__builtin_inf();
    long long vi = 0x7fffffffffffffffLL;
    const long long lim = R - i0 - d;                               // cells of this diagonal run inside the matrix
    const int cnt = lim < kDiagRun ? (int)lim : kDiagRun;
    if (cnt > 0) {
        const long long dstep = n + 1, ostep = R + 1;
        const double *p = sim + i0 * n + (i0 + d);                  // S[i][j] of the first cell
        double run = 0.0;
        for (int k = 0; k < M; ++k) run += p[k * dstep];
        double *o = out + i0 * R + (i0 + d);
        const bool diag_masked = fabs((double)d) < band;
        const long long idx0 = i0 * R + (i0 + d);
        const long long lo_ = lim_lo - i0, hi_ = lim_hi - d - i0;       // s range of the unmasked cells
        const int s_lo = lo_ < 0 ? 0 : (lo_ > kDiagRun ? kDiagRun : (int)lo_), s_hi = hi_ < 0 ? 0 : (hi_ > kDiagRun ? kDiagRun : (int)hi_);
        const double *pn = p + (long long)M * dstep;
        // sliding update: + S[i + M][j + M] - S[i][j]; the loads do not depend on the running sum
#pragma unroll 8
        for (int s = 0; s < cnt; ++s) {
            if (s > 0) run += pn[(s - 1) * dstep] - p[(s - 1) * dstep];
            vmin = nan_min(vmin, run);
            // unmasked: lim_lo <= i and j = i + d < lim_hi (i <= j)
            if (!diag_masked && s >= s_lo && s < s_hi) {
                o[s * ostep] = run;
                if (argmax_better(run, idx0 + s * ostep, v, vi)) { v = run; vi = idx0 + s * ostep; }
            }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int off = 1; off < 64; off <<= 1) {
        vmin = nan_min(vmin, __shfl_xor(vmin, off, 64));
        const double ov = __shfl_xor(v, off, 64);
        const long long oi = __shfl_xor(vi, off, 64);
        if (argmax_better(ov, oi, v, vi)) { v = ov; vi = oi; }
    }
    if (lane == 0) { red[wave] = vmin; rv[wave] = v; ri[wave] = vi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) {
            vmin = nan_min(vmin, red[w]);
            if (argmax_better(rv[w], ri[w], v, vi)) { v = rv[w]; vi = ri[w]; }
        }
        const long long blk = (long long)blockIdx.y * gridDim.x + blockIdx.x;
        block_min[blk] = vmin;
        cand_val[blk] = v;
        cand_idx[blk] = vi;
    }
}

// folds the per-block minima of thumb_diag into min_out[0] (one block)
__global__ __launch_bounds__(1024) void thumb_min_kernel(const double *__restrict__ block_min, long long n_block_min,
                                                          double *__restrict__ min_out) {
    __shared__ double red[16];
    double m = __builtin_inf();
    for (long long b = threadIdx.x; b < n_block_min; b += 1024) m = nan_min(m, block_min[b]);
    for (int off = 1; off < 64; off <<= 1) m = nan_min(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w) m = nan_min(m, red[w]);
        min_out[0] = m;
    }
}

// masked cells = the global minimum (:1149-1160).  grid (ceil(R / 1024), ceil(R / kMaskRows)); a block sweeps kMaskRows
// rows x 1024 columns; write only.
__global__ __launch_bounds__(256) void thumb_fill_kernel(double *__restrict__ f, long long R, double band,
                                                          long long lim_lo, long long lim_hi,
                                                          const double *__restrict__ min_ptr) {
    const double min_sm = min_ptr[0];
    for (int rr = 0; rr < kMaskRows; ++rr) {
        const long long i = (long long)blockIdx.y * kMaskRows + rr;
        if (i >= R) break;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const long long j = (long long)blockIdx.x * 1024 + 256 * q + threadIdx.x;
            if (j < R && thumb_masked(i, j, band, lim_lo, lim_hi)) f[i * R + j] = min_sm;
        }
    }
}

// folds the arg-max candidates (one block).  The unmasked maximum wins unless it does not exceed the value every masked
// cell holds (the global minimum; NaN as soon as any cell is NaN): then the first cell in row-major order that holds the
// maximum may be a masked one, which thread 0 locates row by row.
__global__ __launch_bounds__(1024) void thumb_argmax_kernel(const double *__restrict__ cand_val,
                                                             const long long *__restrict__ cand_idx, long long n_cand,
                                                             const double *__restrict__ min_ptr, long long R, double band,
                                                             long long lim_lo, long long lim_hi,
                                                             long long *__restrict__ best_idx) {
    __shared__ double rv[16];
    __shared__ long long ri[16];
    double v = -__builtin_inf();
    long long vi = 0x7fffffffffffffffLL;
    for (long long b = threadIdx.x; b < n_cand; b += 1024) {
        const double ov = cand_val[b];
        const long long oi = cand_idx[b];
        if (argmax_better(ov, oi, v, vi)) { v = ov; vi = oi; }
    }
    for (int off = 1; off < 64; off <<= 1) {
        const double ov = __shfl_xor(v, off, 64);
        const long long oi = __shfl_xor(vi, off, 64);
        if (argmax_better(ov, oi, v, vi)) { v = ov; vi = oi; }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { rv[wave] = v; ri[wave] = vi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w)
            if (argmax_better(rv[w], ri[w], v, vi)) { v = rv[w]; vi = ri[w]; }
        const double m = min_ptr[0];
        const bool m_nan = m != m;
        const bool unmasked_wins = (vi != 0x7fffffffffffffffLL) && !m_nan && v > m;
        if (!unmasked_wins) {
            // the maximum of the masked matrix is the fill value itself: first masked cell in row-major order, against
            // the first unmasked cell that holds the same value (if any)
            long long first_masked = 0x7fffffffffffffffLL;
            for (long long i = 0; i < R && first_masked == 0x7fffffffffffffffLL; ++i) {
                if (thumb_masked(i, 0, band, lim_lo, lim_hi)) first_masked = i * R;
                else if (lim_hi < R) first_masked = i * R + (lim_hi > 0 ? lim_hi : 0);     // columns >= lim_hi are masked
                // (column 0 unmasked means i <= 0 + ... : no lower-triangle, band or lim_lo cell to the right of it)
            }
            const bool tie = (vi != 0x7fffffffffffffffLL) && (m_nan ? (v != v) : (v == m));
            vi = (tie && vi < first_masked) ? vi : first_masked;
        }
        best_idx[0] = vi;
    }
}

}  // namespace paa
