// Register-FFT feature kernel for windows W = 2 R1 R2 with two coprime odd primes (BASELINE config 5: 1102 = 2 x 29 x 19
// at 44.1 kHz / 25 ms): several frames per wave-iteration, both FFT passes in registers, int16 / interleaved stereo int16 /
// float64 samples; also produces spectrogram / chromagram rows (modes 1 / 2).
//
// The complex FFT of length Nc = R1 R2 (real-input trick: z[n] = y[2n] + i y[2n+1]) is a PRIME-FACTOR transform: with
// n = (R2 n1 + R1 n2) mod Nc and k = (c1 k1 + c2 k2) mod Nc (c1 = R2 (R2^-1 mod R1), c2 = R1 (R1^-1 mod R2)) there are
// no twiddle factors between the passes.  Per iteration a wave handles Q frames:
//   time    : the time-domain features of the Q frames at once, 16 lanes per frame: an even split of the frame's samples
//             over the row (quad loads in batches, no bounds masks in the main loop), np.sign as two integer compares
//             against the clip mean in counts, one running energy + a snapshot at the entropy-block boundary, 4-step
//             DPP reductions inside the row
//   pass A  : lane (frame f, n2 < R2): radix-R1 DFT over n1 of samples fetched (normalised) from global memory; the
//             O(R^2) prime butterfly uses the cos/sin symmetry (x_j +- x_{R-j}): (R-1)^2 FMAs; outputs leave the lane as
//             they are produced: real parts into the frame's spectrum slot (used as a plane), imaginary parts are kept
//   exchange: real plane, then imaginary plane (the slot holds Nc doubles = exactly one plane)
//   pass B  : lane (frame f, p <= R1/2): the two radix-R2 DFTs of columns k1 = p and R1 - p, produced output pair by
//             output pair; Z[k] and Z[Nc-k] = (column R1-p, output R2-q) meet in the lane, so the real-FFT recombination
//             and |X| happen in registers and each magnitude is written once into the slot
//   features: the Q frames at once, 16 lanes per frame over an even split of the bins (same scheme); row segments staged
//             in LDS.  Spectrogram / chromagram rows need no previous spectrum: Q slots and eight waves per CU.
// Halo: a run that starts at t0 > 0 first processes frames t0-Q .. t0-1 without storing them.
//
// Replaces the while loop at ShortTermFeatures.py:608-682 and its helpers (:22-140, :236-321), and the loops of
// spectrogram (:415-422) / chromagram (:349-359), for these window sizes.
#pragma once
#include <type_traits>
#include <utility>
#include <vector>

#include "device_common.hpp"
#include "kernels_ct.hpp"          // PairLoad
#include "kernels_generic.hpp"
#include "tables.hpp"

namespace paa {
namespace reg {

// (the PrimeTab literals moved to csrc/prime_tables.hpp in round 6; this file is an archived experiment and is no longer built)
// output pair (k, R-k), 1 <= k <= (R-1)/2, of the length-R DFT from v0 and the sums / differences s_j = v_j + v_{R-j},
// d_j = v_j - v_{R-j}:  X[k] = A - iB, X[R-k] = A + iB with A = v0 + sum_j s_j cos(2 pi jk/R), B = sum_j d_j sin(2 pi jk/R)
template <int R, int K>
__device__ __forceinline__ void prime_pair(const double2 &v0, const double2 *s, const double2 *d, double2 &xk,
                                           double2 &xrk) {
    constexpr int H = (R - 1) / 2;
    double ax = v0.x, ay = v0.y, bx = 0.0, by = 0.0;
#pragma unroll
    for (int j = 1; j <= H; ++j) {
        constexpr int dummy = 0;
        (void)dummy;
        const int m = (j * K) % R;                         // compile-time after unrolling
        const int mm = (m <= H) ? m : R - m;
        const double c = PrimeTab<R>::c[mm - 1];
        const double sn = (m <= H) ? PrimeTab<R>::s[mm - 1] : -PrimeTab<R>::s[mm - 1];
        ax = fma(s[j - 1].x, c, ax);
        ay = fma(s[j - 1].y, c, ay);
        bx = fma(d[j - 1].x, sn, bx);
        by = fma(d[j - 1].y, sn, by);
    }
    xk = make_double2(ax + by, ay - bx);
    xrk = make_double2(ax - by, ay + bx);
}
template <int R>
__device__ __forceinline__ double2 prime_dc(const double2 &v0, const double2 *s) {
    constexpr int H = (R - 1) / 2;
    double x = v0.x, y = v0.y;
#pragma unroll
    for (int j = 0; j < H; ++j) { x += s[j].x; y += s[j].y; }
    return make_double2(x, y);
}
template <int R>
__device__ __forceinline__ void prime_fold(const double2 *v, double2 *s, double2 *d) {
    constexpr int H = (R - 1) / 2;
#pragma unroll
    for (int j = 1; j <= H; ++j) { s[j - 1] = cadd(v[j], v[R - j]); d[j - 1] = csub(v[j], v[R - j]); }
}

// compile-time loop over the output pairs
template <int R, int K, typename F>
__device__ __forceinline__ void for_pairs(F &&f) {
    if constexpr (K <= (R - 1) / 2) {
        f(std::integral_constant<int, K>{});
        for_pairs<R, K + 1>(f);
    }
}

constexpr int inv_mod(int a, int m) {
    int r = 1;
    for (int i = 1; i < m; ++i)
        if ((a * i) % m == 1) r = i;
    return r;
}

template <int R1_, int R2_, int Q_>
struct Shape {
    static constexpr int R1 = R1_, R2 = R2_, Q = Q_;
    static constexpr int NC = R1 * R2, W = 2 * NC, NF = NC;
    static constexpr int NP = (R1 + 1) / 2;                            // pass-B lanes per frame
    static constexpr int C1 = R2 * inv_mod(R2 % R1, R1);               // k = (C1 k1 + C2 k2) mod NC
    static constexpr int C2 = R1 * inv_mod(R1 % R2, R2);
    static constexpr int NFP = (NF + 1) & ~1;
    static_assert(Q * R2 <= 64 && Q * NP <= 64, "Q frames must fit the wave in both passes");
    static_assert((C1 % R1) == 0 || true, "");
};

// LDS layout: shared tables, then per wave: (Q + 1) spectrum slots, the output staging tile, fv / msp
struct RegLayout {
    int off_post, off_mello, off_melcnt, off_meloff, off_melw, off_dct, off_chstart, off_chsrc, off_chw;
    int off_bins;        // ushort2 [R2][16]: byte offsets of bin k and of its mirror NC - k (8 NF: nowhere, the slot's padding double)
                         // of pass-B lane pcol's output e (e = 0: q = 0, 2 j - 1: q = j, 2 j: q = R2 - j)
    int table_bytes, wave_bytes, waves;
    int ring;            // spectrum slots per wave: Q + 1 for the features (the flux needs the previous frame), Q for the rows
};

// zcr count, energy and energy entropy of one frame read from global memory (ShortTermFeatures.py:22-51)
template <typename T>
__device__ __forceinline__ TimeFeat time_features_global(const PlanDev &P, const T *__restrict__ x, const ClipNorm &nm,
                                                         int lane) {
    const int W = P.W, L = P.blk_t;
    const double sc = sample_scale<T>();
    auto y = [&](int n) { return fma(load_sample<T>(x + n), sc, -nm.mean) * nm.inv; };
    auto sgn = [](double v) { return (v > 0.0) - (v < 0.0); };
    double eblk[10];
    int zc = 0;
#pragma unroll
    for (int j = 0; j < 10; ++j) {
        double p = 0.0;
        for (int n = j * L + lane; n < (j + 1) * L; n += kWave) {
            const double v = y(n);
            p = fma(v, v, p);
            if (n > 0) zc += abs(sgn(v) - sgn(y(n - 1)));
        }
        eblk[j] = p;
    }
    // the ten wave reductions are independent chains: issued together they overlap instead of exposing ten latencies
#pragma unroll
    for (int j = 0; j < 10; ++j) eblk[j] = wsum(eblk[j]);
    double e_tail = 0.0;
    for (int n = 10 * L + lane; n < W; n += kWave) {
        const double v = y(n);
        e_tail = fma(v, v, e_tail);
        if (n > 0) zc += abs(sgn(v) - sgn(y(n - 1)));
    }
    TimeFeat tf;
    tf.e_tot = wsum(e_tail);
#pragma unroll
    for (int j = 0; j < 10; ++j) tf.e_tot += eblk[j];
    tf.zc = wsum_i(zc);
    double num = 0.0;
#pragma unroll
    for (int j = 0; j < 10; ++j)
        if (lane == j) num = eblk[j];
    const double s = fast_div(num, tf.e_tot + kEps);
    tf.ent_e = wsum((lane < 10) ? -(s * fast_log2(s + kEps)) : 0.0);
    return tf;
}

// ---- per-frame stages with the window known at compile time ----------------------------------------------
// The generic kernel's stages walk the frame in ten run-time loops (one per entropy block), each with its own
// dependent global / LDS loads and its own wave reduction: latency-bound (0.35 ms each on config 5).  With W and NF
// template constants every load of the frame is issued up front, the block a sample / bin belongs to is static, and the
// spectral block energies come from ONE wave scan (differences of the cumulative energy at the block boundaries).

// zcr count, energy and energy entropy of one frame read from global memory (ShortTermFeatures.py:22-51).
// Sample n = 64 r + lane sits in register row r of its lane (coalesced loads).
template <typename SH, typename T>
__device__ __forceinline__ TimeFeat time_features_fixed(const T *__restrict__ x, const ClipNorm &nm, int lane) {
    constexpr int W = SH::W, L = W / 10, NR = (W + kWave - 1) / kWave;
    static_assert(L >= kWave, "a 64-sample row may span at most two entropy blocks");
    const double sc = sample_scale<T>();
    double y[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int n = kWave * r + lane;
        // (rows are full except the last; clamp the address instead of branching, mask the value)
        const double v = load_sample<T>(x + min(n, W - 1));
        y[r] = (n < W) ? fma(v, sc, -nm.mean) * nm.inv : 0.0;
    }
    double eb[11];                 // ten blocks of L samples + the tail beyond 10 L
#pragma unroll
    for (int j = 0; j < 11; ++j) eb[j] = 0.0;
    int zc = 0, carry = 0;         // carry: sign of the last sample of the previous row
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const double v = y[r];
        const int sg = (v > 0.0) - (v < 0.0);
        // sign of sample n - 1: the lane below (wave_shr:1), lane 0 takes the previous row's lane 63
        const int sp = __builtin_amdgcn_update_dpp(carry, sg, 0x138, 0xF, 0xF, false);
        const int n = kWave * r + lane;
        zc += (n >= 1 && n < W) ? abs(sg - sp) : 0;
        carry = __builtin_amdgcn_readlane(sg, 63);
        constexpr int dummy = 0;
        (void)dummy;
        const int jlo = (kWave * r) / L;                               // static after unrolling
        const int bl = (jlo + 1) * L - kWave * r;                      // first lane of the row in block jlo + 1
        const double v2 = v * v;
        const int ja = jlo < 10 ? jlo : 10, jb = jlo + 1 < 10 ? jlo + 1 : 10;
        eb[ja] += (lane < bl) ? v2 : 0.0;
        if (bl < kWave) eb[jb] += (lane >= bl) ? v2 : 0.0;
    }
#pragma unroll
    for (int j = 0; j < 11; ++j) eb[j] = wsum(eb[j]);                  // independent chains
    TimeFeat tf;
    tf.e_tot = eb[10];
#pragma unroll
    for (int j = 0; j < 10; ++j) tf.e_tot += eb[j];
    tf.zc = wsum_i(zc);
    double num = 0.0;
#pragma unroll
    for (int j = 0; j < 10; ++j)
        if (lane == j) num = eb[j];
    const double s = fast_div(num, tf.e_tot + kEps);
    tf.ent_e = wsum((lane < 10) ? -(s * fast_log2(s + kEps)) : 0.0);
    return tf;
}

// the 34 base features of one frame into fv[0..33] (ShortTermFeatures.py:626-667); lane l owns the C consecutive bins
// [C l, C l + C) (C odd: conflict-free LDS reads), both spectra are pulled into registers once.  tmp: 12 doubles of LDS.
template <typename SH>
__device__ __forceinline__ void frame_features_fixed(const PlanDev &P, const Tabs &tb, const TimeFeat &tf,
                                                     const double *cur, const double *prv, double *fv, double *msp,
                                                     double *tmp, int lane) {
    constexpr int W = SH::W, NF = SH::NF, LB = NF / 10;
    constexpr int C0 = (NF + kWave - 1) / kWave, C = C0 | 1;
    static_assert(LB > C, "a lane's chunk may contain at most one block boundary");
    const double f0 = P.fs / (2.0 * (double)NF);
    const int kb = C * lane;
    double Xc[C], Xv[C];
#pragma unroll
    for (int m = 0; m < C; ++m) {
        const int k = min(kb + m, NF - 1);
        const double a = cur[k], b = prv[k];
        Xc[m] = (kb + m < NF) ? a : 0.0;
        Xv[m] = (kb + m < NF) ? b : 0.0;
    }
    // sweep A: sums, max, the lane's energy (:57-107)
    double sX = 0.0, sXp = 0.0, sM = 0.0, mx = 0.0, cs = 0.0;
#pragma unroll
    for (int m = 0; m < C; ++m) {
        sX += Xc[m];
        sXp += Xv[m];
        sM = fma((double)m, Xc[m], sM);
        mx = fmax(mx, Xc[m]);
        cs = fma(Xc[m], Xc[m], cs);
    }
    double sIX = f0 * fma((double)(kb + 1), sX, sM);                   // sum (k + 1) f0 X
    const double incl = wscan_incl(cs);
    const double excl = incl - cs;
    const double sP = readlane63(incl);
    sX = wsum(sX);
    sXp = wsum(sXp);
    sIX = wsum(sIX);
    mx = wmax_nonneg(mx);
    // cumulative energy at the block boundaries 0, LB, 2 LB, .. 10 LB: the lane whose chunk holds a boundary writes it
    {
        const int jb = (kb + LB - 1) / LB;                             // first boundary at or after the chunk start
        const int mb = jb * LB - kb;                                   // its offset inside the chunk
        double part = 0.0, cumb = excl;
#pragma unroll
        for (int m = 0; m < C; ++m) {
            cumb = (m == mb) ? excl + part : cumb;
            part = fma(Xc[m], Xc[m], part);
        }
        if (mb < C && jb <= 10 && kb < NF) tmp[jb] = cumb;
    }
    wsync();
    double ent_f;
    {
        const double num = (lane < 10) ? tmp[lane + 1] - tmp[lane] : 0.0;
        const double s = fast_div(num, sP + kEps);
        ent_f = wsum((lane < 10) ? -(s * fast_log2(s + kEps)) : 0.0);
    }
    const double sXe = sX + (double)NF * kEps;                         // np.sum(X + eps) (:118-119)
    sXp += (double)NF * kEps;
    // centroid, spread, flux (:57-82, :110-124)
    const double r = (mx == 0.0) ? 1.0 / kEps : fast_div(1.0, mx);
    const double den = sX * r + kEps;
    const double cen = fast_div(sIX * r, den);
    const double rX = fast_div(1.0, sXe), rXp = fast_div(1.0, sXp);
    double sSp = 0.0, sFl = 0.0;
#pragma unroll
    for (int m = 0; m < C; ++m) {
        const double dv = (double)(kb + m + 1) * f0 - cen;
        sSp = fma(dv * dv, Xc[m] * r, sSp);
        const double df = Xc[m] * rX - Xv[m] * rXp;
        sFl = fma(df, df, sFl);
    }
    sSp = wsum(sSp);
    sFl = wsum(sFl);
    const double spread = fast_sqrt(fast_div(sSp, den));
    // roll-off: first k with cumsum(X^2)[k] + eps > 0.9 * sum(X^2) (:127-140)
    int first = 0x7fffffff;
    {
        const double thr = 0.90 * sP;
        double run = excl;
#pragma unroll
        for (int m = 0; m < C; ++m) {
            run = fma(Xc[m], Xc[m], run);
            first = (first == 0x7fffffff && kb + m < NF && run + kEps > thr) ? kb + m : first;
        }
        first = wmin_i(first);
    }
    // MFCC: sparse mel dot, log10, 13 x 40 DCT (:236-254)
    if (lane < 40) {
        const int lo = tb.mel_lo[lane], cnt = tb.mel_cnt[lane];
        const double *w = tb.mel_w + tb.mel_off[lane];
        double a0 = 0.0, a1 = 0.0;
        int i = 0;
        for (; i + 2 <= cnt; i += 2) {
            a0 = fma(cur[lo + i], w[i], a0);
            a1 = fma(cur[lo + i + 1], w[i + 1], a1);
        }
        if (i < cnt) a0 = fma(cur[lo + i], w[i], a0);
        msp[lane] = fast_log10((a0 + a1) + kEps);
    }
    const double chroma = chroma_class(tb, cur, sP, lane);             // (:277-321)
    wsync();
    if (lane < 13) {
        const double *m = tb.dct + lane * tb.dct_stride;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
        for (int n = 0; n < 40; n += 4) {
            a0 = fma(m[n], msp[n], a0);
            a1 = fma(m[n + 1], msp[n + 1], a1);
            a2 = fma(m[n + 2], msp[n + 2], a2);
            a3 = fma(m[n + 3], msp[n + 3], a3);
        }
        fv[8 + lane] = (a0 + a1) + (a2 + a3);
    }
    if (lane < 12) fv[21 + lane] = chroma;
    // population std of the 12 chroma values (:667)
    const double cmean = wsum((lane < 12) ? chroma : 0.0) * (1.0 / 12.0);
    const double cd = (lane < 12) ? chroma - cmean : 0.0;
    const double cvar = wsum(cd * cd) * (1.0 / 12.0);
    if (lane == 0) {
        fv[0] = ((double)tf.zc / 2.0) / (double)(W - 1);
        fv[1] = tf.e_tot / (double)W;
        fv[2] = tf.ent_e;
        fv[3] = cen / (P.fs / 2.0);
        fv[4] = spread / (P.fs / 2.0);
        fv[5] = ent_f;
        fv[6] = (cur == prv) ? 0.0 : sFl;      // first frame: previous spectrum = itself (:624-625)
        fv[7] = (first == 0x7fffffff) ? 0.0 : (double)first / (double)NF;
        fv[33] = fast_sqrt(cvar);
    }
    wsync();
}

// ---- the same two stages for ALL frames of an iteration at once: 16 lanes per frame (frame f = lane / 16, Q <= 4) ------
// One frame at a time with the whole wave spends most of its instructions on 64-lane reductions and on stages that use 12,
// 13 or 40 lanes (profiles/r03d_reg_features_stereo: the time-domain and feature stages were 65 % of the kernel's VALU
// instructions).  Here lane (f, i) owns a contiguous chunk of its frame -- CT = 69 samples / CB = 35 bins, odd, so the
// 16 lanes of a frame read LDS conflict-free -- carries two partial block energies (a chunk meets at most two entropy
// blocks), and every reduction is a 4-step DPP reduction inside the 16-lane row, shared by the Q frames.
constexpr int kRegFlush = 6;        // frames staged per row segment (a multiple of Q = 3; the LDS budget of six waves)

// four consecutive samples in one load (element alignment only)
template <typename T> struct QuadLoad;
template <> struct QuadLoad<int16_t> {
    typedef short vec __attribute__((ext_vector_type(4), aligned(2)));
    static __device__ __forceinline__ void get(const int16_t *p, double (&o)[4]) {
        const vec s = *reinterpret_cast<const vec *>(p);
        o[0] = (double)s.x; o[1] = (double)s.y; o[2] = (double)s.z; o[3] = (double)s.w;
    }
};
template <> struct QuadLoad<stereo16> {      // four stereo frames = 16 bytes, each summed L + R in the load
    typedef int vec __attribute__((ext_vector_type(4), aligned(4)));
    static __device__ __forceinline__ void get(const stereo16 *p, double (&o)[4]) {
        const vec s = *reinterpret_cast<const vec *>(p);
        o[0] = (double)stereo_word_sum(s.x); o[1] = (double)stereo_word_sum(s.y);
        o[2] = (double)stereo_word_sum(s.z); o[3] = (double)stereo_word_sum(s.w);
    }
};
template <> struct QuadLoad<double> {
    typedef double vec __attribute__((ext_vector_type(2), aligned(8)));
    static __device__ __forceinline__ void get(const double *p, double (&o)[4]) {
        const vec a = *reinterpret_cast<const vec *>(p), b = *reinterpret_cast<const vec *>(p + 2);
        o[0] = a.x; o[1] = a.y; o[2] = b.x; o[3] = b.y;
    }
};

// np.sign(x * sc - mean) per sample.  Integer samples: two integer compares against the clip mean in counts -- for a whole
// count x, x * sc - mean > 0 <=> x > floor(mu) and < 0 <=> x < ceil(mu) (both tests fail only when x == mu exactly);
// float samples: from the bit pattern of the difference (0 for +-0, else +-1).
template <typename T> struct SampleSign {
    int hi, lo;
    __device__ __forceinline__ SampleSign(const ClipNorm &nm, double sc) {
        const double mu = nm.mean / sc;                             // exact: sc is a power of two
        const double fl = floor(mu);
        hi = (int)fl;
        lo = (fl == mu) ? hi : hi + 1;
    }
    __device__ __forceinline__ int of(double raw, double) const {
        const int x = (int)raw;                                     // (the samples are whole numbers)
        return (int)(x > hi) - (int)(x < lo);
    }
};
template <> struct SampleSign<double> {
    __device__ __forceinline__ SampleSign(const ClipNorm &, double) {}
    __device__ __forceinline__ int of(double, double d) const {
        const int h = __double2hiint(d), l = __double2loint(d);
        return (((h & 0x7fffffff) | l) != 0) ? ((h >> 31) | 1) : 0;
    }
};

// zcr count, energy, energy entropy (ShortTermFeatures.py:22-51) of frames tq .. tq+Q-1 -> tfs[4 f + {0, 1, 2}]
template <typename SH, typename T>
__device__ __forceinline__ void time_features_grouped(const T *__restrict__ x0, long long step, int tq, int tend,
                                                      const ClipNorm &nm, double *tfs, int lane) {
    constexpr int W = SH::W, L = W / 10, Q = SH::Q;
    // even split of the frame over the 16 lanes of its row: BASE samples each, the first REM lanes one more -- every lane's
    // first 4 NG samples are inside the frame (no masks in the main loop), the rest is a short masked epilogue
    constexpr int BASE = W / 16, REM = W % 16, NG = BASE / 4, EPI = BASE - 4 * NG + (REM ? 1 : 0);
    static_assert(BASE + 1 <= L, "a lane's samples may meet at most two entropy blocks");
    const int f = lane >> 4, i = lane & 15;
    const int t = (f < Q && tq + f < tend) ? tq + f : tq;           // idle groups shadow a valid frame
    const T *x = x0 + (long long)t * step;
    const double sc = sample_scale<T>();
    const SampleSign<T> sign(nm, sc);
    const int kb = BASE * i + min(i, REM), ke = kb + BASE + (i < REM ? 1 : 0);
    const int cat = min(kb / L, 10);
    const int bound = (cat >= 10) ? 0x7fffffff : (cat + 1) * L;    // first sample of the next entropy block
    // energy: one running sum of d^2 (d = x * sc - mean; the 1 / peak^2 factor is applied to the sums) and its value when
    // the chunk crosses into the next block
    double e_all = 0.0, e_snap = 0.0;
    int zc = 0;
    int sprev;
    {
        const double r0 = load_sample<T>(x + max(kb - 1, 0));
        sprev = sign.of(r0, fma(r0, sc, -nm.mean));                 // (lane 0: sample 0 against itself counts nothing)
    }
    auto one = [&](int n, double raw) {
        const double d = fma(raw, sc, -nm.mean);
        e_snap = (n == bound) ? e_all : e_snap;
        e_all = fma(d, d, e_all);
        const int sx = sign.of(raw, d);
        zc += abs(sx - sprev);
        sprev = sx;
    };
    auto batch = [&](auto gc, int g0) {                             // G groups of four samples, fetched together
        constexpr int G = decltype(gc)::value;
        double raw[G][4];
#pragma unroll
        for (int g = 0; g < G; ++g) QuadLoad<T>::get(x + kb + 4 * (g0 + g), raw[g]);
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int u = 0; u < 4; ++u) one(kb + 4 * (g0 + g) + u, raw[g][u]);
    };
    constexpr int G = 6;
    int g0 = 0;
#pragma unroll 1
    for (; g0 + G <= NG; g0 += G) batch(std::integral_constant<int, G>{}, g0);
    if constexpr (NG % G != 0) batch(std::integral_constant<int, NG % G>{}, NG - NG % G);
#pragma unroll
    for (int u = 0; u < EPI; ++u) {                                 // the lane's last one or two samples
        const int n = kb + 4 * NG + u;
        if (n < ke) one(n, load_sample<T>(x + n));
    }
    const double inv2 = nm.inv * nm.inv;
    const bool crossed = bound < ke;
    const double ea = (crossed ? e_snap : e_all) * inv2, eb = (crossed ? e_all - e_snap : 0.0) * inv2;
    double eblk[10];
#pragma unroll
    for (int j = 0; j < 10; ++j) eblk[j] = group_sum(((cat == j) ? ea : 0.0) + ((cat + 1 == j) ? eb : 0.0));
    double e_tot = group_sum(((cat == 10) ? ea : 0.0) + ((cat + 1 == 10) ? eb : 0.0));
#pragma unroll
    for (int j = 0; j < 10; ++j) e_tot += eblk[j];
    zc = group_sum_i(zc);
    double num = 0.0;
#pragma unroll
    for (int j = 0; j < 10; ++j)
        if (i == j) num = eblk[j];
    const double s = fast_div(num, e_tot + kEps);
    const double ent_e = group_sum((i < 10) ? -(s * fast_log2(s + kEps)) : 0.0);
    if (i == 0 && f < Q) { tfs[4 * f] = e_tot; tfs[4 * f + 1] = ent_e; tfs[4 * f + 2] = (double)zc; }
}

// the 34 base features (ShortTermFeatures.py:626-667) of the Q frames in slots (slot0 + f) % ring -> fvq[48 f + 0..33];
// mspq: Q x 40 doubles, tmpq: Q x 12 doubles, tfs: the time-domain results of the iteration
template <typename SH>
__device__ __forceinline__ void frame_features_grouped(const PlanDev &P, const Tabs &tb, const double *slots, int slot0,
                                                       int ring, int tq, const double *tfs, double *fvq, double *mspq,
                                                       double *tmpq, int lane) {
    constexpr int W = SH::W, NF = SH::NF, NFP = SH::NFP, Q = SH::Q, LB = NF / 10;
    const int f = lane >> 4, i = lane & 15;
    const int fe = (f < Q) ? f : 0;                                 // the idle group shadows frame 0 (nothing is written)
    const double *cur = slots + ((slot0 + fe) % ring) * NFP;
    const double *prv = (tq + fe == 0) ? cur : slots + ((slot0 + fe + Q) % ring) * NFP;
    double *fv = fvq + 48 * fe, *msp = mspq + 40 * fe, *tmp = tmpq + 12 * fe;
    const double f0 = P.fs / (2.0 * (double)NF);
    // even split of the bins over the 16 lanes of the row: BASE bins each, the first REM lanes one more: the main loops
    // run without masks, the extra bin is an epilogue
    constexpr int BASE = NF / 16, REM = NF % 16;
    static_assert(BASE + 1 < LB, "a lane's bins may contain at most one block boundary");
    const int kb = BASE * i + min(i, REM);
    const bool extra = i < REM;
    // sweep A: sums, max, the lane's energy (:57-107)
    double sX = 0.0, sXp = 0.0, sM = 0.0, mx = 0.0, cs = 0.0;
    auto sweep_a = [&](int m) {
        const double a = cur[kb + m], b = prv[kb + m];
        sX += a;
        sXp += b;
        sM = fma((double)m, a, sM);
        mx = fmax(mx, a);
        cs = fma(a, a, cs);
    };
#pragma unroll 2
    for (int m = 0; m < BASE; ++m) sweep_a(m);
    if (extra) sweep_a(BASE);
    double sIX = f0 * fma((double)(kb + 1), sX, sM);                   // sum (k + 1) f0 X
    const double incl = group_scan_incl(cs);
    const double excl = incl - cs;
    const double sP = group_max(incl);                                 // total = the (non-decreasing, >= 0) scan's last entry
    sX = group_sum(sX);
    sXp = group_sum(sXp);
    sIX = group_sum(sIX);
    mx = group_max(mx);
    // cumulative energy at the block boundaries 0, LB, .. 10 LB: the lane whose chunk holds a boundary writes it
    {
        const int jb = (kb + LB - 1) / LB;                             // first boundary at or after the chunk start
        const int mb = jb * LB - kb;                                   // its offset inside the chunk
        double part = 0.0, cumb = excl;
        auto bnd = [&](int m) {
            cumb = (m == mb) ? excl + part : cumb;
            const double a = cur[kb + m];
            part = fma(a, a, part);
        };
#pragma unroll 2
        for (int m = 0; m < BASE; ++m) bnd(m);
        if (extra) bnd(BASE);
        if (mb < BASE + (extra ? 1 : 0) && jb <= 10 && f < Q) tmp[jb] = cumb;
    }
    wsync();
    double ent_f;
    {
        const double num = (i < 10) ? tmp[i + 1] - tmp[i] : 0.0;
        const double s = fast_div(num, sP + kEps);
        ent_f = group_sum((i < 10) ? -(s * fast_log2(s + kEps)) : 0.0);
    }
    const double sXe = sX + (double)NF * kEps;                         // np.sum(X + eps) (:118-119)
    sXp += (double)NF * kEps;
    // centroid, spread, flux, roll-off (:57-82, :110-140)
    const double r = (mx == 0.0) ? 1.0 / kEps : fast_div(1.0, mx);
    const double den = sX * r + kEps;
    const double cen = fast_div(sIX * r, den);
    const double rX = fast_div(1.0, sXe), rXp = fast_div(1.0, sXp);
    const double thr = 0.90 * sP;
    double sSp = 0.0, sFl = 0.0, run = excl;
    int first = 0x7fffffff;
    auto sweep_b = [&](int m) {
        const double a = cur[kb + m], b = prv[kb + m];
        const double dv = (double)(kb + m + 1) * f0 - cen;
        sSp = fma(dv * dv, a * r, sSp);
        const double df = a * rX - b * rXp;
        sFl = fma(df, df, sFl);
        run = fma(a, a, run);
        first = (first == 0x7fffffff && run + kEps > thr) ? kb + m : first;
    };
#pragma unroll 2
    for (int m = 0; m < BASE; ++m) sweep_b(m);
    if (extra) sweep_b(BASE);
    sSp = group_sum(sSp);
    sFl = group_sum(sFl);
    first = group_min_i(first);
    const double spread = fast_sqrt(fast_div(sSp, den));
    // MFCC: sparse mel dot, log10 (:236-251): lane i takes filters i, 16 + i, 32 + i
#pragma unroll 1
    for (int fl = i; fl < 40; fl += 16) {
        const int lo = tb.mel_lo[fl], cnt = tb.mel_cnt[fl];
        const double *w = tb.mel_w + tb.mel_off[fl];
        double a0 = 0.0, a1 = 0.0;
        int n = 0;
        for (; n + 2 <= cnt; n += 2) {
            a0 = fma(cur[lo + n], w[n], a0);
            a1 = fma(cur[lo + n + 1], w[n + 1], a1);
        }
        if (n < cnt) a0 = fma(cur[lo + n], w[n], a0);
        const double lg = fast_log10((a0 + a1) + kEps);
        if (f < Q) msp[fl] = lg;
    }
    const double chroma = chroma_class(tb, cur, sP, i);                // (:277-321): lanes i < 12 of the row
    wsync();
    if (i < 13 && f < Q) {                                             // 13 x 40 DCT (:253)
        const double *mrow = tb.dct + i * tb.dct_stride;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
        for (int n = 0; n < 40; n += 4) {
            a0 = fma(mrow[n], msp[n], a0);
            a1 = fma(mrow[n + 1], msp[n + 1], a1);
            a2 = fma(mrow[n + 2], msp[n + 2], a2);
            a3 = fma(mrow[n + 3], msp[n + 3], a3);
        }
        fv[8 + i] = (a0 + a1) + (a2 + a3);
    }
    // population std of the 12 chroma values (:667)
    const double cv = (i < 12) ? chroma : 0.0;
    const double cmean = group_sum(cv) * (1.0 / 12.0);
    const double cd = (i < 12) ? cv - cmean : 0.0;
    const double cvar = group_sum(cd * cd) * (1.0 / 12.0);
    if (f < Q) {
        if (i < 12) fv[21 + i] = chroma;
        if (i == 0) {
            fv[0] = (tfs[4 * f + 2] / 2.0) / (double)(W - 1);
            fv[1] = tfs[4 * f] / (double)W;
            fv[2] = tfs[4 * f + 1];
            fv[3] = cen / (P.fs / 2.0);
            fv[4] = spread / (P.fs / 2.0);
            fv[5] = ent_f;
            fv[6] = (cur == prv) ? 0.0 : sFl;      // first frame: previous spectrum = itself (:624-625)
            fv[7] = (first == 0x7fffffff) ? 0.0 : (double)first / (double)NF;
            fv[33] = fast_sqrt(cvar);
        }
    }
    wsync();
}

template <typename SH, typename T>
__global__ __launch_bounds__(512) void st_reg_kernel(PlanDev P, RegLayout L, const unsigned char *__restrict__ blob,
                                                      const T *__restrict__ sig, const ClipDev *__restrict__ clips,
                                                      const ClipNorm *__restrict__ norms,
                                                      const Tile *__restrict__ tiles, int n_tiles,
                                                      double *__restrict__ out) {
    constexpr int R1 = SH::R1, R2 = SH::R2, Q = SH::Q, NC = SH::NC, NF = SH::NF, NP = SH::NP, NFP = SH::NFP;
    constexpr int H1 = (R1 - 1) / 2, H2 = (R2 - 1) / 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    {
        const int4 *src4 = reinterpret_cast<const int4 *>(blob);
        int4 *dst4 = reinterpret_cast<int4 *>(smem);
        for (int n = threadIdx.x; n < L.table_bytes / 16; n += blockDim.x) dst4[n] = src4[n];
    }
    __syncthreads();       // the only workgroup-wide barrier
    Tabs tb;
    tb.tw = nullptr;
    tb.post = reinterpret_cast<const double2 *>(smem + L.off_post);
    const unsigned char *tb_bins = smem + L.off_bins;
    tb.mel_lo = reinterpret_cast<const int *>(smem + L.off_mello);
    tb.mel_cnt = reinterpret_cast<const int *>(smem + L.off_melcnt);
    tb.mel_off = reinterpret_cast<const int *>(smem + L.off_meloff);
    tb.mel_w = reinterpret_cast<const double *>(smem + L.off_melw);
    tb.dct = reinterpret_cast<const double *>(smem + L.off_dct);
    tb.dct_stride = 41;
    tb.ch_start = reinterpret_cast<const int *>(smem + L.off_chstart);
    tb.ch_src = reinterpret_cast<const int *>(smem + L.off_chsrc);
    tb.ch_w = reinterpret_cast<const double *>(smem + L.off_chw);

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int tile_id = blockIdx.x * L.waves + wave;
    if (tile_id >= n_tiles) return;
    const int F = P.F > 0 ? P.F : 1;
    unsigned char *wb = smem + L.table_bytes + wave * L.wave_bytes;
    const int ring = L.ring;
    double *slots = reinterpret_cast<double *>(wb);                    // ring x NFP doubles
    double *otile = slots + ring * NFP;
    double *fv = otile + kRegFlush * F;                                // Q x 48: feature vectors of the iteration's frames
    double *msp = fv + 48 * Q;                                         // Q x 40: log mel energies
    double *tfs = msp + 40 * Q;                                        // Q x 4: time-domain results of the iteration
    double *tmpq = tfs + 4 * Q;                                        // Q x 12: cumulative energies at the block boundaries

    const Tile tl = tiles[tile_id];
    const ClipDev c = clips[tl.clip];
    const ClipNorm nm = wave_clip_norm<T>(P, c, norms, tl.clip, lane);
    const T *x0 = sig + c.sample_off + P.frame_origin;
    const long long Tc = c.T;
    double *oc = out + c.out_off;
    const double sc = sample_scale<T>();


    const int tend = tl.t0 + tl.cnt;
    const bool halo = (P.mode == 0) && tl.t0 >= Q;
    int slot0 = 1;                                          // slots of the iteration: slot0 .. slot0+Q-1 (mod Q+1); previous = slot0-1
    double vprev = 0.0;
    int nslot = 0, tbase = tl.t0;
    for (int tq = halo ? tl.t0 - Q : tl.t0; tq < tend; tq += Q, slot0 = (slot0 + Q) % ring) {
        const bool store_it = tq >= tl.t0;
        // lane roles, re-derived every iteration from an opaque copy of the lane index: otherwise the optimiser hoists
        // every lane-dependent address and bin index of the unrolled passes out of the loop and spills them
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));
        const int fa = lane_o / R2, n2 = lane_o - fa * R2;      // pass A: frame fa, input residue n2
        const bool act_a = fa < Q;
        const int fb = lane_o / NP, pcol = lane_o - fb * NP;    // pass B: frame fb, column pair (pcol, R1 - pcol)
        const bool act_b = fb < Q;
        const int pcolb = (pcol == 0) ? 0 : R1 - pcol;
        // ---------------- time-domain features of the iteration's Q frames, 16 lanes per frame (results pass through LDS)
        if (P.mode == 0 && (store_it || P.deltas) && !PAA_DEBUG_BIT(P.debug, 1)) time_features_grouped<SH, T>(x0, P.S, tq, tend, nm, tfs, lane_o);
        // ---------------- pass A: radix-R1 over n1, inputs z[(R2 n1 + R1 n2) mod NC] from global memory
        double yim[R1];
        {
            const int ta = tq + fa;
            const bool ok = act_a && ta < tend;
            double2 s[H1 > 0 ? H1 : 1], d[H1 > 0 ? H1 : 1];
            double2 v0;
            {
                double2 v[R1];
                const T *xf = x0 + (long long)(ok ? ta : tq) * P.S;
                // index R2 n1 + R1 n2 (< 2 NC: one conditional wrap) as ONE of two per-lane addresses + a compile-time offset: the
                // wrap happens from a compile-time threshold of n2 on
                const T *p_lo = xf + 2 * R1 * n2, *p_hi = p_lo - 2 * NC;
#pragma unroll
                for (int n1 = 0; n1 < R1; ++n1) {
                    const int wrap_from = (NC - R2 * n1 + R1 - 1) / R1;          // n2 >= wrap_from: R2 n1 + R1 n2 >= NC
                    const T *pp = (n2 >= wrap_from) ? p_hi : p_lo;
                    const double2 ab = ct::PairLoad<T>::get(pp + 2 * R2 * n1);   // one load per complex sample
                    v[n1] = make_double2(fma(ab.x, sc, -nm.mean) * nm.inv, fma(ab.y, sc, -nm.mean) * nm.inv);
                }
                v0 = v[0];
                prime_fold<R1>(v, s, d);
            }
            double *pl = slots + ((slot0 + (act_a ? fa : 0)) % ring) * NFP;
            const double2 x0c = prime_dc<R1>(v0, s);
            if (ok) pl[n2] = x0c.x;
            yim[0] = x0c.y;
            for_pairs<R1, 1>([&](auto kc) {
                constexpr int K = decltype(kc)::value;
                double2 xk, xr;
                prime_pair<R1, K>(v0, s, d, xk, xr);
                if (ok) { pl[K * R2 + n2] = xk.x; pl[(R1 - K) * R2 + n2] = xr.x; }
                yim[K] = xk.y;
                yim[R1 - K] = xr.y;
                // one output pair at a time: dependent FP64 operations issue back to back on gfx950, and the scheduler
                // would otherwise run all (R-1)/2 accumulations side by side and spill
                asm volatile("" : "+v"(yim[K]), "+v"(yim[R1 - K]));
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        wsync();
        // ---------------- exchange: real plane, then imaginary plane
        double2 a[R2], b[R2];
        {
            const double *pl = slots + ((slot0 + (act_b ? fb : 0)) % ring) * NFP;
#pragma unroll
            for (int r = 0; r < R2; ++r) { a[r].x = pl[pcol * R2 + r]; b[r].x = pl[pcolb * R2 + r]; }
            wsync();
            if (act_a && tq + fa < tend) {
                double *pw = slots + ((slot0 + fa) % ring) * NFP;
#pragma unroll
                for (int k1 = 0; k1 < R1; ++k1) pw[k1 * R2 + n2] = yim[k1];
            }
            wsync();
#pragma unroll
            for (int r = 0; r < R2; ++r) { a[r].y = pl[pcol * R2 + r]; b[r].y = pl[pcolb * R2 + r]; }
            wsync();
        }
        // ---------------- pass B: two radix-R2 DFTs, real-FFT recombination, magnitudes (ShortTermFeatures.py:617-621)
        if (act_b && tq + fb < tend) {
            double *sp = slots + ((slot0 + fb) % ring) * NFP;
            double2 sa[H2], da[H2], sb[H2], db[H2];
            const double2 a0 = a[0], b0 = b[0];
            prime_fold<R2>(a, sa, da);
            prime_fold<R2>(b, sb, db);
            const double mscale = 0.5 / (double)NF;            // E and O carry 1/2; X / len(X) (:621)
            // bin of (column p, output q): k = (C1 p + C2 q) mod NC; its partner NC - k is (column R1 - p, output R2 - q).  Where
            // the two magnitudes of an output go comes from the host table (byte offsets; the mirror of bin 0 -- the Nyquist bin the
            // reference drops -- goes to the slot's padding double): no index arithmetic, no predicates around the stores
            const unsigned *t_bins = reinterpret_cast<const unsigned *>(tb_bins) + pcol;
            unsigned char *spb = reinterpret_cast<unsigned char *>(sp);
            const unsigned char *postb = reinterpret_cast<const unsigned char *>(tb.post);
            auto bins = [&](unsigned ent, const double2 &zk, const double2 &zm) {
                // 2E = Z[k] + conj Z[NC-k], 2O = -i (Z[k] - conj Z[NC-k]); X[k] = E + w^k O, |X[NC-k]| = |E - w^k O|
                const unsigned ok_ = ent & 0xffffu, om_ = ent >> 16;
                const double2 e = make_double2(zk.x + zm.x, zk.y - zm.y);
                const double2 o = make_double2(zk.y + zm.y, zm.x - zk.x);
                const double2 t = cmul(*reinterpret_cast<const double2 *>(postb + 2 * ok_), o);
                const double xr = e.x + t.x, xi = e.y + t.y, yr = e.x - t.x, yi = e.y - t.y;
                *reinterpret_cast<double *>(spb + ok_) = mag_sqrt(fma(xr, xr, xi * xi)) * mscale;
                *reinterpret_cast<double *>(spb + om_) = mag_sqrt(fma(yr, yr, yi * yi)) * mscale;
            };
            bins(t_bins[0], prime_dc<R2>(a0, sa), prime_dc<R2>(b0, sb));
            for_pairs<R2, 1>([&](auto qc) {
                constexpr int QK = decltype(qc)::value;
                double2 xa, xar, xb, xbr;
                const unsigned e_up = t_bins[16 * (2 * QK - 1)], e_dn = t_bins[16 * (2 * QK)];
                prime_pair<R2, QK>(a0, sa, da, xa, xar);
                prime_pair<R2, QK>(b0, sb, db, xb, xbr);
                bins(e_up, xa, xbr);          // Z[k(p,q)] with Z[NC - k] = column R1-p, output R2-q
                bins(e_dn, xar, xb);          // Z[k(p,R2-q)] with column R1-p, output q
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        wsync();
        // ---------------- the 34 features of the Q frames at once (16 lanes per frame), then the staging tile
        if (P.mode == 0) {
            if ((store_it || P.deltas) && !PAA_DEBUG_BIT(P.debug, 2))
                frame_features_grouped<SH>(P, tb, slots, slot0, ring, tq, tfs, fv, msp, tmpq, lane_o);
#pragma nounroll
            for (int f = 0; f < Q; ++f) {
                const int t = tq + f;
                if (t >= tend) break;
                // with deltas the last halo frame supplies the previous column of the first stored frame
                if (store_it || (P.deltas && f == Q - 1)) {
                    const double v = (lane_o < kBase) ? fv[48 * f + lane_o] : 0.0;
                    if (store_it) {
                        if (lane_o < kBase) {
                            otile[nslot * F + lane_o] = v;
                            if (P.deltas) otile[nslot * F + kBase + lane_o] = (t == 0) ? 0.0 : v - vprev;
                        }
                        ++nslot;
                    }
                    vprev = v;
                }
                if (nslot == kRegFlush || (t == tend - 1 && nslot > 0)) {
                    wsync();
                    // row segments: nslot consecutive frames of feature row fr are contiguous in [F][T]
                    for (int idx = lane_o; idx < F * kRegFlush; idx += kWave) {
                        const int fr = idx / kRegFlush, i = idx % kRegFlush;
                        if (i < nslot) oc[(long long)fr * Tc + tbase + i] = otile[i * F + fr];
                    }
                    wsync();
                    tbase += nslot;
                    nslot = 0;
                }
            }
        } else {
            // ---------------- per frame: spectrogram row / chromagram row
#pragma nounroll
            for (int f = 0; f < Q; ++f) {
                const int t = tq + f;
                if (t >= tend) break;
                const double *cur = slots + ((slot0 + f) % ring) * NFP;
                if (P.mode == 1) {            // spectrogram row (ShortTermFeatures.py:422)
                    double *row = oc + (long long)t * NF;
                    // (write-once stream of 4.4 KB rows that are only 8-byte aligned: non-temporal stores keep the L2 from
                    // writing half-filled lines twice; measured +15 % on float64 input, neutral on int16)
                    for (int k = lane_o; k < NF; k += kWave) __builtin_nontemporal_store(cur[k], row + k);
                } else {                      // chromagram row (:356-359)
                    double p = 0.0;
                    for (int k = lane_o; k < NF; k += kWave) { const double X = cur[k]; p = fma(X, X, p); }
                    p = wsum(p);
                    const double ch = chroma_class(tb, cur, p, lane_o);
                    if (lane_o < 12) oc[(long long)t * 12 + lane_o] = ch;
                }
            }
        }
        wsync();
    }
}

// ---- host: which windows have an instance, LDS layout + table blob ----------------------------------------
typedef Shape<29, 19, 3> Shape1102;

inline bool reg_supported(int window) { return window == Shape1102::W; }

inline size_t reg_wave_bytes(int nfp, int ring, int q, int F) {
    size_t b = (size_t)ring * nfp * 8 + (size_t)kRegFlush * F * 8 + (size_t)q * (48 + 40 + 4 + 12) * 8;
    return (b + 15) / 16 * 16;
}

// fills the layout and the host image of the shared table region (no Stockham twiddles: prime-factor transform)
template <typename SH>
inline void reg_layout(const FftPlan &fft, const MelTable *mel, const ChromaTable *chroma, int F, RegLayout &L,
                       std::vector<unsigned char> *blob) {
    constexpr int nfp = SH::NFP, q = SH::Q;
    static_assert(SH::NP <= 16 && 8 * SH::NFP < 65536 && SH::NFP > SH::NF, "bin table: 16 lanes per frame, 16-bit byte offsets, a padding double");
    int off = 0;
    auto take = [&off](size_t bytes) { const int o = off; off += (int)((bytes + 15) / 16 * 16); return o; };
    const int Nc = fft.len;
    const size_t n_melw = mel ? mel->w.size() : 0, n_ch = chroma ? chroma->src.size() : 0;
    L.off_post = take((size_t)Nc * 16);
    L.off_mello = take(40 * 4);
    L.off_melcnt = take(40 * 4);
    L.off_meloff = take(40 * 4);
    L.off_melw = take(std::max<size_t>(n_melw, 1) * 8);
    L.off_dct = take(13 * 41 * 8);
    L.off_chstart = take(13 * 4);
    L.off_chsrc = take(std::max<size_t>(n_ch, 1) * 4);
    L.off_chw = take(std::max<size_t>(n_ch, 1) * 8);
    L.off_bins = take((size_t)SH::R2 * 16 * 4);
    L.table_bytes = off;
    // spectrogram / chromagram rows need no previous spectrum: Q slots, and eight waves (two per SIMD) fit the LDS
    L.ring = (F > 0) ? q + 1 : q;
    L.wave_bytes = (int)reg_wave_bytes(nfp, L.ring, q, F > 0 ? F : 1);
    L.waves = 8;
    while (L.waves > 1 && (size_t)L.table_bytes + (size_t)L.waves * L.wave_bytes > 160 * 1024) --L.waves;
    if (!blob) return;
    blob->assign((size_t)L.table_bytes, 0);
    unsigned char *b = blob->data();
    memcpy(b + L.off_post, fft.post.data(), (size_t)Nc * 16);
    {
        unsigned short *bt = reinterpret_cast<unsigned short *>(b + L.off_bins);
        for (int e = 0; e < SH::R2; ++e) {
            const int qq = (e == 0) ? 0 : (e % 2 ? (e + 1) / 2 : SH::R2 - e / 2);
            for (int pc = 0; pc < 16; ++pc) {
                const int k = (SH::C1 * std::min(pc, SH::NP - 1) + SH::C2 * qq) % SH::NC;
                bt[2 * (16 * e + pc)] = (unsigned short)(8 * k);
                bt[2 * (16 * e + pc) + 1] = (unsigned short)(8 * (k != 0 ? SH::NC - k : SH::NF));
            }
        }
    }
    if (mel && !mel->w.empty()) {
        memcpy(b + L.off_mello, mel->lo.data(), 40 * 4);
        memcpy(b + L.off_melcnt, mel->cnt.data(), 40 * 4);
        memcpy(b + L.off_meloff, mel->off.data(), 40 * 4);
        memcpy(b + L.off_melw, mel->w.data(), n_melw * 8);
        double dct[kNumMfcc * kNumMel];
        build_dct(dct);
        double *dd = reinterpret_cast<double *>(b + L.off_dct);
        for (int qq = 0; qq < 13; ++qq)
            for (int n = 0; n < 40; ++n) dd[qq * 41 + n] = dct[qq * 40 + n];
    }
    if (chroma && !chroma->src.empty()) {
        memcpy(b + L.off_chstart, chroma->class_start, 13 * 4);
        memcpy(b + L.off_chsrc, chroma->src.data(), n_ch * 4);
        memcpy(b + L.off_chw, chroma->w.data(), n_ch * 8);
    }
}

}  // namespace reg
}  // namespace paa
