// Auxiliary kernels: clip statistics (clip-global normalisation constants) and mid-term statistics.
#pragma once
#include <float.h>

#include "device_common.hpp"

namespace paa {

// ---- clip statistics: replaces the two full passes of dc_normalize (ShortTermFeatures.py:14-19)
// and the /2^15 scaling (:567-568) with one streaming read.  Block b reduces chunk b to
// (sum, min, max); int16 sums are exact in int64, so the clip mean is bit-identical to NumPy's.
template <typename T> struct StatAcc;
template <> struct StatAcc<int16_t> { typedef long long sum_t; };
template <> struct StatAcc<double> { typedef double sum_t; };

__global__ __launch_bounds__(256) void clip_stats_i16_kernel(const int16_t *__restrict__ sig,
                                                              const StatChunk *__restrict__ chunks,
                                                              long long *__restrict__ psum,
                                                              int *__restrict__ pmin, int *__restrict__ pmax) {
    const StatChunk ch = chunks[blockIdx.x];
    const int tid = threadIdx.x;
    const long long a0 = ch.start, a1 = ch.start + ch.len;
    // 16-byte aligned body [b0, b1), scalar head/tail
    long long b0 = (a0 + 7) & ~7LL;
    if (b0 > a1) b0 = a1;
    const long long b1 = b0 + ((a1 - b0) & ~7LL);
    long long s = 0;
    int mn = 32767, mx = -32768;
    for (long long i = a0 + tid; i < b0; i += 256) { const int v = sig[i]; s += v; mn = min(mn, v); mx = max(mx, v); }
    for (long long i = b1 + tid; i < a1; i += 256) { const int v = sig[i]; s += v; mn = min(mn, v); mx = max(mx, v); }
    const int4 *body = reinterpret_cast<const int4 *>(sig + b0);
    const long long nvec = (b1 - b0) >> 3;
    // two samples per 32-bit operation (round 5): v_dot2_i32_i16 with (1, 1) adds a pair into the sum, packed 16-bit min / max keep
    // two running extremes -- 12 vector operations per 16 bytes instead of 32 (the pass was at 5.6 TB/s with the ALU work co-limiting)
    typedef short s16x2_t __attribute__((ext_vector_type(2)));
    const s16x2_t ones = {1, 1};
    s16x2_t mn2 = {32767, 32767}, mx2 = {-32768, -32768};
    int s32 = 0;
    auto fold = [&](const int4 &q) {
        const int w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const s16x2_t p = __builtin_bit_cast(s16x2_t, w[j]);
            s32 = __builtin_amdgcn_sdot2(p, ones, s32, false);
            mn2 = __builtin_elementwise_min(mn2, p);
            mx2 = __builtin_elementwise_max(mx2, p);
        }
    };
    // four independent 16-byte loads per thread in flight (a 64 K-sample chunk is 32 loads per thread): one load at a
    // time leaves the read stream latency-bound at ~4.4 TB/s
    long long i = tid;
    for (; i + 768 < nvec; i += 1024) {
        const int4 q0 = body[i], q1 = body[i + 256], q2 = body[i + 512], q3 = body[i + 768];
        fold(q0); fold(q1); fold(q2); fold(q3);
    }
    for (; i < nvec; i += 256) fold(body[i]);
    s += s32;
    mn = min(mn, min((int)mn2.x, (int)mn2.y));
    mx = max(mx, max((int)mx2.x, (int)mx2.y));
    __shared__ long long ss[4];
    __shared__ int smn[4], smx[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_xor(s, o, 64);
        mn = min(mn, __shfl_xor(mn, o, 64));
        mx = max(mx, __shfl_xor(mx, o, 64));
    }
    if ((tid & 63) == 0) { ss[tid >> 6] = s; smn[tid >> 6] = mn; smx[tid >> 6] = mx; }
    __syncthreads();
    if (tid == 0) {
        psum[blockIdx.x] = ss[0] + ss[1] + ss[2] + ss[3];
        pmin[blockIdx.x] = min(min(smn[0], smn[1]), min(smn[2], smn[3]));
        pmax[blockIdx.x] = max(max(smx[0], smx[1]), max(smx[2], smx[3]));
    }
}

// interleaved stereo int16 (L0 R0 L1 R1 ...; sample index = stereo frame): statistics of the sums L + R, i.e. of
// 2^16 x what stereo_to_mono + the /2^15 scaling produce (audioBasicIO.py:156-168, ShortTermFeatures.py:568); sums stay
// exact in int64.  Same shape as the int16 kernel: 16-byte loads (4 stereo frames), four in flight per thread.
__global__ __launch_bounds__(256) void clip_stats_stereo_kernel(const stereo16 *__restrict__ sig,
                                                                 const StatChunk *__restrict__ chunks,
                                                                 long long *__restrict__ psum,
                                                                 int *__restrict__ pmin, int *__restrict__ pmax) {
    const StatChunk ch = chunks[blockIdx.x];
    const int tid = threadIdx.x;
    const long long a0 = ch.start, a1 = ch.start + ch.len;
    long long b0 = (a0 + 3) & ~3LL;
    if (b0 > a1) b0 = a1;
    const long long b1 = b0 + ((a1 - b0) & ~3LL);
    const int *w32 = reinterpret_cast<const int *>(sig);
    long long s = 0;
    int mn = 0x7fffffff, mx = -0x7fffffff - 1;
    for (long long i = a0 + tid; i < b0; i += 256) { const int v = stereo_word_sum(w32[i]); s += v; mn = min(mn, v); mx = max(mx, v); }
    for (long long i = b1 + tid; i < a1; i += 256) { const int v = stereo_word_sum(w32[i]); s += v; mn = min(mn, v); mx = max(mx, v); }
    const int4 *body = reinterpret_cast<const int4 *>(w32 + b0);
    const long long nvec = (b1 - b0) >> 2;
    int s32 = 0;      // a 128 K-frame chunk: 512 values of |v| < 2^16 per thread -- fits
    auto fold = [&](const int4 &q) {
        const int v0 = stereo_word_sum(q.x), v1 = stereo_word_sum(q.y), v2 = stereo_word_sum(q.z), v3 = stereo_word_sum(q.w);
        s32 += (v0 + v1) + (v2 + v3);
        mn = min(mn, min(min(v0, v1), min(v2, v3)));
        mx = max(mx, max(max(v0, v1), max(v2, v3)));
    };
    long long i = tid;
    for (; i + 768 < nvec; i += 1024) {
        const int4 q0 = body[i], q1 = body[i + 256], q2 = body[i + 512], q3 = body[i + 768];
        fold(q0); fold(q1); fold(q2); fold(q3);
    }
    for (; i < nvec; i += 256) fold(body[i]);
    s += s32;
    __shared__ long long ss[4];
    __shared__ int smn[4], smx[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_xor(s, o, 64);
        mn = min(mn, __shfl_xor(mn, o, 64));
        mx = max(mx, __shfl_xor(mx, o, 64));
    }
    if ((tid & 63) == 0) { ss[tid >> 6] = s; smn[tid >> 6] = mn; smx[tid >> 6] = mx; }
    __syncthreads();
    if (tid == 0) {
        psum[blockIdx.x] = ss[0] + ss[1] + ss[2] + ss[3];
        pmin[blockIdx.x] = min(min(smn[0], smn[1]), min(smn[2], smn[3]));
        pmax[blockIdx.x] = max(max(smx[0], smx[1]), max(smx[2], smx[3]));
    }
}

// float64 samples: 16-byte loads, four in flight per thread (one 8-byte load at a time left this pass at ~2 TB/s)
__global__ __launch_bounds__(256) void clip_stats_f64_kernel(const double *__restrict__ sig,
                                                              const StatChunk *__restrict__ chunks,
                                                              double *__restrict__ psum,
                                                              double *__restrict__ pmin, double *__restrict__ pmax) {
    const StatChunk ch = chunks[blockIdx.x];
    const int tid = threadIdx.x;
    const long long a0 = ch.start, a1 = ch.start + ch.len;
    long long b0 = (a0 + 1) & ~1LL;
    if (b0 > a1) b0 = a1;
    const long long b1 = b0 + ((a1 - b0) & ~1LL);
    double s = 0.0, s2 = 0.0, mn = DBL_MAX, mx = -DBL_MAX;
    if (tid == 0 && a0 < b0) { const double v = sig[a0]; s += v; mn = fmin(mn, v); mx = fmax(mx, v); }
    if (tid == 1 && b1 < a1) { const double v = sig[b1]; s += v; mn = fmin(mn, v); mx = fmax(mx, v); }
    const double2 *body = reinterpret_cast<const double2 *>(sig + b0);
    const long long nvec = (b1 - b0) >> 1;
    auto fold = [&](const double2 &q) {
        s += q.x; s2 += q.y;
        mn = fmin(mn, fmin(q.x, q.y));
        mx = fmax(mx, fmax(q.x, q.y));
    };
    long long i = tid;
    for (; i + 768 < nvec; i += 1024) {
        const double2 q0 = body[i], q1 = body[i + 256], q2 = body[i + 512], q3 = body[i + 768];
        fold(q0); fold(q1); fold(q2); fold(q3);
    }
    for (; i < nvec; i += 256) fold(body[i]);
    s += s2;
    __shared__ double ss[4], smn[4], smx[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_xor(s, o, 64);
        mn = fmin(mn, __shfl_xor(mn, o, 64));
        mx = fmax(mx, __shfl_xor(mx, o, 64));
    }
    if ((tid & 63) == 0) { ss[tid >> 6] = s; smn[tid >> 6] = mn; smx[tid >> 6] = mx; }
    __syncthreads();
    if (tid == 0) {
        psum[blockIdx.x] = (ss[0] + ss[1]) + (ss[2] + ss[3]);
        pmin[blockIdx.x] = fmin(fmin(smn[0], smn[1]), fmin(smn[2], smn[3]));
        pmax[blockIdx.x] = fmax(fmax(smx[0], smx[1]), fmax(smx[2], smx[3]));
    }
}

// one wave per clip: fold the chunk partials (lane-strided, then a fixed xor tree -> deterministic),
// derive (mean, 1/(max|x-mean| + 1e-10))
template <typename SumT, typename MmT>
__global__ __launch_bounds__(64) void clip_params_kernel(const ClipDev *__restrict__ clips, long long n_clips,
                                                          const SumT *__restrict__ psum, const MmT *__restrict__ pmin,
                                                          const MmT *__restrict__ pmax, double sc, int window,
                                                          ClipNorm *__restrict__ norms) {
    const long long c = blockIdx.x;
    if (c >= n_clips) return;
    const ClipDev cd = clips[c];
    const ClipNorm nm = clip_norm_wave<SumT, MmT>(cd, psum, pmin, pmax, sc, window, (int)threadIdx.x);
    if (threadIdx.x != 0) return;
    norms[c] = nm;
}

// ---- mid-term statistics (MidTermFeatures.py:110-126): mean and population std of every
// short-term row over windows [m*step, min(m*step+ratio, T)), then nan_to_num.
__device__ __forceinline__ double nan_to_num(double v) {
    if (isnan(v)) return 0.0;
    if (isinf(v)) return v > 0 ? DBL_MAX : -DBL_MAX;
    return v;
}

// 16 lanes per (row, mid window): the window's values are read as 128-byte segments, summed with DPP
// reductions (4 windows per wave at a time); the second pass re-reads from L1/L2.
__global__ __launch_bounds__(256) void mid_stats_kernel(const ClipDev *__restrict__ clips,
                                                         const long long *__restrict__ mid_off,
                                                         const double *__restrict__ st, int nrows,
                                                         long long ratio, long long step, int blocks_per_clip,
                                                         double *__restrict__ mid) {
    const long long c = blockIdx.x / blocks_per_clip;
    const int chunk = blockIdx.x % blocks_per_clip;
    const ClipDev cd = clips[c];
    const long long T = cd.T;
    const long long M = (T + step - 1) / step;
    const int i = threadIdx.x & 15;
    const long long idx = (long long)chunk * 16 + (threadIdx.x >> 4);      // 16 windows per block
    const bool live = idx < (long long)nrows * M;
    const long long row = live ? idx / M : 0, m = live ? idx % M : 0;
    const double *x = st + cd.out_off + row * T;
    const long long b = m * step;
    long long e = b + ratio;
    if (e > T) e = T;
    if (!live) e = b;
    double s = 0.0, v = 0.0;
    const double n = (double)(e - b);
    double mean;
    if (ratio <= 64) {
        // the usual shapes (40 or 20 frames per mid-term window): the lane's <= 4 values are loaded once, all at once, and
        // serve both passes from registers (same operations in the same order as the general loop below)
        double xv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) xv[j] = (b + i + 16 * j < e) ? x[b + i + 16 * j] : 0.0;
#pragma unroll
        for (int j = 0; j < 4; ++j) if (b + i + 16 * j < e) s += xv[j];
        s = group_sum(s);
        mean = s / n;
#pragma unroll
        for (int j = 0; j < 4; ++j) if (b + i + 16 * j < e) { const double d = xv[j] - mean; v = fma(d, d, v); }
    } else {
        for (long long k = b + i; k < e; k += 16) s += x[k];
        s = group_sum(s);
        mean = s / n;
        for (long long k = b + i; k < e; k += 16) { const double d = x[k] - mean; v = fma(d, d, v); }
    }
    v = group_sum(v);
    if (live && i == 0) {
        double *o = mid + mid_off[c];
        o[row * M + m] = nan_to_num(mean);
        o[(row + nrows) * M + m] = nan_to_num(sqrt(v / n));
    }
}


// ---- delta rows from base rows (ShortTermFeatures.py:668-680): a clip's [34][T] slab of base features -> its [68][T] slab,
// rows 34..67 = differences of consecutive columns (column 0: zeros).  The feature kernels form their delta rows from the very
// values they store (v - vprev in FP64), so re-forming them from the stored base rows gives the same bits: a sharded job ships
// the 34 base rows over xGMI and the root completes the matrices (half the gathered bytes).  One block per (clip, <= 2048 frames, row).
struct DeltaTile {
    long long base_off, out_off;    // clip slabs (doubles): 34 T and 68 T prefix sums over the clips before
    int T, t0, cnt, pad;
};
__global__ __launch_bounds__(256) void expand_deltas_kernel(const DeltaTile *__restrict__ tiles,
                                                             const double *__restrict__ base, double *__restrict__ out) {
    const DeltaTile tl = tiles[blockIdx.x];
    const double *b = base + tl.base_off;
    double *o = out + tl.out_off;
    const long long T = tl.T;
    // blockIdx.y = base row: 34 workgroups per tile (one workgroup walking all 34 rows of its tile took 134 us for a 5 000-frame
    // batch -- a one-hour clip is only 71 tiles)
    const int r = blockIdx.y;
    const double *br = b + r * T;
    double *ob = o + r * T, *od = o + (long long)(kBase + r) * T;
#pragma unroll 4
    for (int k = threadIdx.x; k < tl.cnt; k += 256) {
        const long long t = tl.t0 + k;
        const double v = br[t];
        const double pv = (t > 0) ? br[t - 1] : v;
        ob[t] = v;
        od[t] = (t > 0) ? v - pv : 0.0;
    }
}

// ---- beat extraction (MidTermFeatures.py:18-84 + utilities.peakdet, utilities.py:33-102), one wave per clip.
// 18 short-term rows; per row: threshold = 2 mean|diff|, Billauer's peak detector (a sequential scan: lane r
// scans row r from an LDS tile that all 64 lanes fill with coalesced loads), histogram of the gaps between
// successive maxima; then the rows' histograms / T are added in row order, argmax -> (bpm, confidence).
constexpr int kBeatRows = 18;
constexpr int kBeatTile = 128;          // frames per LDS tile

__global__ __launch_bounds__(64) void beat_kernel(const ClipDev *__restrict__ clips, const double *__restrict__ st,
                                                   double window_size, int max_beat, double *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_beat[];
    double *tile = reinterpret_cast<double *>(smem_beat);                         // [18][kBeatTile + 1]
    int *hist = reinterpret_cast<int *>(tile + kBeatRows * (kBeatTile + 1));       // [18][max_beat]
    const int rows[kBeatRows] = {0, 1, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18};
    const ClipDev cd = clips[blockIdx.x];
    const long long T = cd.T;
    const double *base = st + cd.out_off;
    const int lane = threadIdx.x;
    for (int k = lane; k < kBeatRows * max_beat; k += 64) hist[k] = 0;

    // pass 1: threshold of every row (coalesced sweeps, wave reductions)
    double thr = 0.0;
    for (int r = 0; r < kBeatRows; ++r) {
        const double *x = base + (long long)rows[r] * T;
        double s = 0.0;
        for (long long t = lane; t + 1 < T; t += 64) s += fabs(x[t] - x[t + 1]);
        s = wsum(s);
        double d = 2.0 * (s / (double)(T - 1));                  // T == 1: 0/0 = NaN, like np.mean of an empty slice
        if (d <= 0.0) d = 0.0000000000000001;
        if (lane == r) thr = d;
    }
    // pass 2: sequential peak detection, lane r <-> row r
    double lo = INFINITY, hi = -INFINITY;
    long long hi_pos = 0, last_peak = -1;
    bool seek_max = true;
    wsync();
    for (long long t0 = 0; t0 < T; t0 += kBeatTile) {
        const int n = (int)min((long long)kBeatTile, T - t0);
        for (int r = 0; r < kBeatRows; ++r) {
            const double *x = base + (long long)rows[r] * T + t0;
            for (int k = lane; k < n; k += 64) tile[r * (kBeatTile + 1) + k] = x[k];
        }
        wsync();
        if (lane < kBeatRows) {
            const double *v = tile + lane * (kBeatTile + 1);
            for (int k = 0; k < n; ++k) {
                const double cur = v[k];
                if (cur > hi) { hi = cur; hi_pos = t0 + k; }
                if (cur < lo) lo = cur;
                if (seek_max) {
                    if (cur < hi - thr) {
                        if (last_peak >= 0) {
                            const long long gap = hi_pos - last_peak;
                            if (gap >= 1 && gap <= max_beat) hist[lane * max_beat + (int)gap - 1] += 1;
                        }
                        last_peak = hi_pos;
                        lo = cur;
                        seek_max = false;
                    }
                } else if (cur > lo + thr) {
                    hi = cur;
                    hi_pos = t0 + k;
                    seek_max = true;
                }
            }
        }
        wsync();
    }
    // aggregate: hist_all[b] = sum over rows (in order) of count / T
    double best = -1.0, total = 0.0;
    int best_b = 0;
    if (lane == 0) {
        for (int b = 0; b < max_beat; ++b) {
            double h = 0.0;
            for (int r = 0; r < kBeatRows; ++r) h += (double)hist[r * max_beat + b] / (double)T;
            total += h;
            if (h > best) { best = h; best_b = b; }
        }
        const double center = (double)best_b + 1.0;              // edges k + 0.5 -> centers 1, 2, ...
        out[2 * blockIdx.x] = 60.0 / (center * window_size);
        out[2 * blockIdx.x + 1] = best / (total + 0.00000001);
    }
}

}  // namespace paa
