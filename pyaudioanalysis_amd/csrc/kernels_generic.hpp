// Generic short-term feature kernel: any window the LDS envelope admits, any step, int16 or f64 samples;
// also produces spectrogram / chromagram rows (modes 1 / 2).
//
// Workgroup = up to 4 waves; after one barrier (the frame-invariant tables -- FFT twiddles, sparse mel bank,
// DCT, chroma gather list -- are copied from a host-built blob into LDS) every wave is autonomous and owns a
// RUN of consecutive frames of one clip.  Per frame:
//   1. the frame is read once from HBM/L2 (contiguous -> coalesced), normalised on load with the clip's
//      affine constants and written to LDS as the packed complex sequence of the real-input trick
//      (window/2 complex points when the window is even; odd windows use a full-length complex FFT);
//      the time-domain features read those same LDS values before the FFT overwrites them,
//   2. a Stockham autosort FFT runs in LDS (ping-pong buffers, host-chosen radix schedule, hard-coded radix
//      2/3/4/5 butterflies, O(R^2) passes unrolled by 4 for other primes such as 19 x 29 = 551),
//   3. the magnitude spectrum |X|/num_fft stays in LDS and the 34 features are reduced from it with DPP
//      wave reductions; the previous frame's spectrum is kept in LDS for the flux,
//   4. feature columns are staged [kFlush][F] in LDS and stored as row segments.
// Halo: a run that does not start at frame 0 first recomputes frame t0-1 (spectrum for the flux; all features
// when deltas are on) and, with deltas, the spectrum of t0-2.
//
// Replaces the while loop at ShortTermFeatures.py:608-682 and its helpers (:22-140, :236-321), and the loops
// of spectrogram (:415-422) / chromagram (:349-359).
#pragma once
#include <vector>

#include "device_common.hpp"
#include "tables.hpp"

namespace paa {

// frame-invariant tables as seen by device code (LDS copies in the feature kernel, global in the tail kernel)
struct Tabs {
    const double2 *tw, *post;
    const int *mel_lo, *mel_cnt, *mel_off;
    const double *mel_w;
    const double *dct;
    int dct_stride;
    const int *ch_start, *ch_src;
    const double *ch_w;
};

__device__ __forceinline__ Tabs tabs_global(const PlanDev &P) {
    Tabs t;
    t.tw = P.tw; t.post = P.post;
    t.mel_lo = P.mel_lo; t.mel_cnt = P.mel_cnt; t.mel_off = P.mel_off; t.mel_w = P.mel_w;
    t.dct = P.dct; t.dct_stride = 40;
    t.ch_start = P.ch_start; t.ch_src = P.ch_src; t.ch_w = P.ch_w;
    return t;
}

// LDS layout of the shared table blob + per-wave regions (host-built)
struct GenLayout {
    int off_tw, off_post, off_mello, off_melcnt, off_meloff, off_melw, off_dct, off_chstart, off_chsrc, off_chw;
    int table_bytes;     // multiple of 16
    int wave_bytes;      // per-wave region, multiple of 16
    int waves;           // waves per workgroup
};

// ---- Stockham passes --------------------------------------------------------------------
template <int R>
__device__ __forceinline__ void small_dft(double2 *v);
template <> __device__ __forceinline__ void small_dft<2>(double2 *v) { dft2(v); }
template <> __device__ __forceinline__ void small_dft<3>(double2 *v) { dft3(v); }
template <> __device__ __forceinline__ void small_dft<4>(double2 *v) { dft4(v); }
template <> __device__ __forceinline__ void small_dft<5>(double2 *v) { dft5(v); }

template <int R>
__device__ __forceinline__ void stockham_pass(const double2 *__restrict__ in, double2 *__restrict__ out,
                                              int Nc, int Ns, const double2 *__restrict__ tw, int lane) {
    const int nb = Nc / R;
    const int tstride = Nc / (Ns * R);
    for (int j = lane; j < nb; j += kWave) {
        const int k = j % Ns;
        double2 v[R];
#pragma unroll
        for (int r = 0; r < R; ++r) v[r] = in[j + r * nb];
        if (Ns > 1) {
#pragma unroll
            for (int r = 1; r < R; ++r) v[r] = cmul(v[r], tw[r * k * tstride]);
        }
        small_dft<R>(v);
        const int j0 = (j - k) * R + k;
#pragma unroll
        for (int r = 0; r < R; ++r) out[j0 + r * Ns] = v[r];
    }
}

// any radix: one output per lane-iteration; the R (input, twiddle) pairs are fetched four at a time
__device__ __forceinline__ void stockham_pass_any(const double2 *__restrict__ in, double2 *__restrict__ out,
                                                  int Nc, int R, int Ns, const double2 *__restrict__ tw, int lane) {
    const int nb = Nc / R;
    const int a = Nc / (Ns * R);
    for (int o = lane; o < Nc; o += kWave) {
        const int j = o % nb, q = o / nb;
        const int k = j % Ns;
        const int step = (int)(((long long)k * a + (long long)q * nb) % Nc);
        int idx = 0;
        double ar = 0.0, ai = 0.0, br = 0.0, bi = 0.0;
        int p = 0;
        for (; p + 4 <= R; p += 4) {
            double2 x[4], w[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                x[u] = in[j + (p + u) * nb];
                w[u] = tw[idx];
                idx += step;
                if (idx >= Nc) idx -= Nc;
            }
#pragma unroll
            for (int u = 0; u < 4; u += 2) {
                ar = fma(x[u].x, w[u].x, fma(-x[u].y, w[u].y, ar));
                ai = fma(x[u].x, w[u].y, fma(x[u].y, w[u].x, ai));
                br = fma(x[u + 1].x, w[u + 1].x, fma(-x[u + 1].y, w[u + 1].y, br));
                bi = fma(x[u + 1].x, w[u + 1].y, fma(x[u + 1].y, w[u + 1].x, bi));
            }
        }
        for (; p < R; ++p) {
            const double2 x = in[j + p * nb];
            const double2 w = tw[idx];
            ar = fma(x.x, w.x, fma(-x.y, w.y, ar));
            ai = fma(x.x, w.y, fma(x.y, w.x, ai));
            idx += step;
            if (idx >= Nc) idx -= Nc;
        }
        out[(j - k) * R + k + q * Ns] = make_double2(ar + br, ai + bi);
    }
}

// ---- FFT + magnitude of the frame already sitting in bufA as packed complex -> |X|/Nf in `spec` -------
__device__ __forceinline__ void frame_fft(const PlanDev &P, const Tabs &tb, double2 *bufA, double2 *bufB,
                                          double *spec, int lane) {
    const int Nc = P.Nc;
    double2 *src = bufA, *dst = bufB;
    int Ns = 1;
    for (int p = 0; p < P.n_pass; ++p) {
        const int R = P.radix[p];
        switch (R) {
            case 2: stockham_pass<2>(src, dst, Nc, Ns, tb.tw, lane); break;
            case 3: stockham_pass<3>(src, dst, Nc, Ns, tb.tw, lane); break;
            case 4: stockham_pass<4>(src, dst, Nc, Ns, tb.tw, lane); break;
            case 5: stockham_pass<5>(src, dst, Nc, Ns, tb.tw, lane); break;
            default: stockham_pass_any(src, dst, Nc, R, Ns, tb.tw, lane); break;
        }
        Ns *= R;
        double2 *t = src; src = dst; dst = t;
        wsync();
    }
    const double invNf = 1.0 / (double)P.Nf;     // X / len(X)  (ShortTermFeatures.py:621)
    if (P.even) {
        // X[k] = E[k] + w^k O[k],  E = (Z[k] + conj Z[H-k]) / 2,  O = -i (Z[k] - conj Z[H-k]) / 2
        for (int k = lane; k < P.Nf; k += kWave) {
            const double2 zk = src[k];
            const double2 zm = src[k == 0 ? 0 : Nc - k];
            const double2 e = make_double2(0.5 * (zk.x + zm.x), 0.5 * (zk.y - zm.y));
            const double2 o = make_double2(0.5 * (zk.y + zm.y), 0.5 * (zm.x - zk.x));
            const double2 wo = cmul(tb.post[k], o);
            const double xr = e.x + wo.x, xi = e.y + wo.y;
            spec[k] = mag_sqrt(fma(xr, xr, xi * xi)) * invNf;
        }
    } else {
        for (int k = lane; k < P.Nf; k += kWave) {
            const double2 z = src[k];
            spec[k] = mag_sqrt(fma(z.x, z.x, z.y * z.y)) * invNf;
        }
    }
    wsync();
}

// load + normalise one frame into bufA (packed complex); yv(n) below reads sample n back
template <typename T>
__device__ __forceinline__ void frame_load(const PlanDev &P, const T *__restrict__ x, ClipNorm nm, double2 *bufA,
                                           int lane) {
    const double sc = sample_scale<T>();
    if (P.even) {
        double *y = reinterpret_cast<double *>(bufA);
        for (int n = lane; n < P.W; n += kWave) y[n] = fma(load_sample<T>(x + n), sc, -nm.mean) * nm.inv;
    } else {
        for (int n = lane; n < P.W; n += kWave)
            bufA[n] = make_double2(fma(load_sample<T>(x + n), sc, -nm.mean) * nm.inv, 0.0);
    }
    wsync();
}

// chroma of one spectrum: lanes 0..11 return their pitch class (ShortTermFeatures.py:285-308)
__device__ __forceinline__ double chroma_class(const Tabs &tb, const double *spec, double sP, int lane) {
    double acc = 0.0;
    if (lane < 12) {
        const int b = tb.ch_start[lane], e = tb.ch_start[lane + 1];
        for (int i = b; i < e; ++i) {
            const double x = spec[tb.ch_src[i]];
            acc += (x * x) * tb.ch_w[i];
        }
        acc = (sP == 0.0) ? acc / kEps : fast_div(acc, sP);
    }
    return acc;
}

struct TimeFeat {
    double e_tot, ent_e;
    int zc;
};

// zcr count, energy and energy entropy of the normalised frame stored in LDS (ShortTermFeatures.py:22-51)
__device__ __forceinline__ TimeFeat time_features(const PlanDev &P, const double2 *bufA, int lane) {
    const int W = P.W, L = P.blk_t;
    const int st = P.even ? 1 : 2;                 // odd windows: y[n] = bufA[n].x
    const double *y = reinterpret_cast<const double *>(bufA);
    double eblk[10];
    int zc = 0;
#pragma unroll
    for (int j = 0; j < 10; ++j) {
        double p = 0.0;
        for (int n = j * L + lane; n < (j + 1) * L; n += kWave) {
            const double v = y[n * st];
            p = fma(v, v, p);
            if (n > 0) {
                const double u = y[(n - 1) * st];
                zc += abs(((v > 0.0) - (v < 0.0)) - ((u > 0.0) - (u < 0.0)));
            }
        }
        eblk[j] = p;
    }
    // the ten wave reductions are independent chains: issued together they overlap instead of exposing ten latencies
#pragma unroll
    for (int j = 0; j < 10; ++j) eblk[j] = wsum(eblk[j]);
    double e_tail = 0.0;
    for (int n = 10 * L + lane; n < W; n += kWave) {
        const double v = y[n * st];
        e_tail = fma(v, v, e_tail);
        if (n > 0) {
            const double u = y[(n - 1) * st];
            zc += abs(((v > 0.0) - (v < 0.0)) - ((u > 0.0) - (u < 0.0)));
        }
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

// ---- the 34 base features of one frame into fv[0..33] (ShortTermFeatures.py:626-667) -------
__device__ __forceinline__ void frame_features(const PlanDev &P, const Tabs &tb, const TimeFeat &tf,
                                               const double *cur, const double *prv, double *fv, double *msp,
                                               int lane) {
    const int W = P.W, Nf = P.Nf;
    // ---------- spectrum sweep A: sums, max, block energies (:57-107)
    const double f0 = P.fs / (2.0 * (double)Nf);
    double pblk[10];
    double sX = 0.0, sXp = 0.0, sIX = 0.0, mx = 0.0, p_tail = 0.0;
    {
        const int L = P.blk_f;
#pragma unroll
        for (int j = 0; j < 10; ++j) {
            double p = 0.0;
            for (int k = j * L + lane; k < (j + 1) * L; k += kWave) {
                const double X = cur[k];
                sX += X;
                sXp += prv[k];
                sIX = fma((double)(k + 1), X, sIX);
                mx = fmax(mx, X);
                p = fma(X, X, p);
            }
            pblk[j] = p;
        }
        for (int k = 10 * L + lane; k < Nf; k += kWave) {
            const double X = cur[k];
            sX += X;
            sXp += prv[k];
            sIX = fma((double)(k + 1), X, sIX);
            mx = fmax(mx, X);
            p_tail = fma(X, X, p_tail);
        }
        p_tail = wsum(p_tail);
    }
    // independent reduction chains, issued together
#pragma unroll
    for (int j = 0; j < 10; ++j) pblk[j] = wsum(pblk[j]);
    sX = wsum(sX);
    sXp = wsum(sXp);
    sIX = wsum(sIX) * f0;
    mx = wmax_nonneg(mx);
    // np.sum(X + eps) (:118-119) = sum X + Nf eps up to rounding
    const double sXe = sX + (double)Nf * kEps;
    sXp += (double)Nf * kEps;
    double sP = p_tail;
#pragma unroll
    for (int j = 0; j < 10; ++j) sP += pblk[j];

    // spectral entropy: lane j (<10) owns block j (:101-105)
    double ent_f;
    {
        double num = 0.0;
#pragma unroll
        for (int j = 0; j < 10; ++j)
            if (lane == j) num = pblk[j];
        const double s = fast_div(num, sP + kEps);
        ent_f = wsum((lane < 10) ? -(s * fast_log2(s + kEps)) : 0.0);
    }

    // ---------- centroid, then sweep B: spread + flux (:57-82, :110-124)
    const double r = (mx == 0.0) ? 1.0 / kEps : fast_div(1.0, mx);
    const double den = sX * r + kEps;
    const double cen = fast_div(sIX * r, den);
    const double rX = fast_div(1.0, sXe), rXp = fast_div(1.0, sXp);
    double sSp = 0.0, sFl = 0.0;
    for (int k = lane; k < Nf; k += kWave) {
        const double X = cur[k];
        const double dv = (double)(k + 1) * f0 - cen;
        sSp = fma(dv * dv, X * r, sSp);
        const double df = X * rX - prv[k] * rXp;
        sFl = fma(df, df, sFl);
    }
    sSp = wsum(sSp);
    sFl = wsum(sFl);
    const double spread = fast_sqrt(fast_div(sSp, den));

    // ---------- roll-off: first k with cumsum(X^2)[k] + eps > 0.9 * sum(X^2) (:127-140)
    int first = 0x7fffffff;
    {
        const double thr = 0.90 * sP;
        const int c = (Nf + kWave - 1) / kWave;
        const int kb = lane * c, ke = min(Nf, kb + c);
        double cs = 0.0;
        for (int k = kb; k < ke; ++k) { const double X = cur[k]; cs = fma(X, X, cs); }
        double run = wscan_incl(cs) - cs;
        for (int k = kb; k < ke; ++k) {
            const double X = cur[k];
            run = fma(X, X, run);
            if (run + kEps > thr) { first = k; break; }
        }
        first = wmin_i(first);
    }

    // ---------- MFCC: sparse mel dot, log10, 13 x 40 DCT (:236-254)
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
    // ---------- chroma (:277-321)
    const double chroma = chroma_class(tb, cur, sP, lane);
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
    if (lane == 0) {
        fv[0] = ((double)tf.zc / 2.0) / (double)(W - 1);
        fv[1] = tf.e_tot / (double)W;
        fv[2] = tf.ent_e;
        fv[3] = cen / (P.fs / 2.0);
        fv[4] = spread / (P.fs / 2.0);
        fv[5] = ent_f;
        fv[6] = (cur == prv) ? 0.0 : sFl;      // first frame: previous spectrum = itself (:624-625)
        fv[7] = (first == 0x7fffffff) ? 0.0 : (double)first / (double)Nf;
    }
    wsync();
    if (lane == 0) {        // population std of the 12 chroma values (:667)
        double m = 0.0;
        for (int i = 0; i < 12; ++i) m += fv[21 + i];
        m /= 12.0;
        double v = 0.0;
        for (int i = 0; i < 12; ++i) { const double d = fv[21 + i] - m; v = fma(d, d, v); }
        fv[33] = fast_sqrt(v / 12.0);
    }
    wsync();
}

// ---- host: LDS layout + table blob ------------------------------------------------------------------------
inline size_t generic_wave_bytes(int Nc, int Nf, int F) {
    const size_t nfp = (size_t)((Nf + 1) & ~1);
    size_t b = 2 * (size_t)Nc * 16 + 2 * nfp * 8 + (size_t)kFlush * F * 8 + 48 * 8 + 40 * 8;
    return (b + 15) / 16 * 16;
}

// fills the layout and (when blob != nullptr) the host image of the shared table region
inline void generic_layout(const FftPlan &fft, const MelTable *mel, const ChromaTable *chroma, int F, GenLayout &L,
                           std::vector<unsigned char> *blob) {
    int off = 0;
    auto take = [&off](size_t bytes) { const int o = off; off += (int)((bytes + 15) / 16 * 16); return o; };
    const int Nc = fft.len;
    const size_t n_melw = mel ? mel->w.size() : 0, n_ch = chroma ? chroma->src.size() : 0;
    L.off_tw = take((size_t)Nc * 16);
    L.off_post = take(fft.even ? (size_t)Nc * 16 : 16);
    L.off_mello = take(40 * 4);
    L.off_melcnt = take(40 * 4);
    L.off_meloff = take(40 * 4);
    L.off_melw = take(std::max<size_t>(n_melw, 1) * 8);
    L.off_dct = take(13 * 41 * 8);
    L.off_chstart = take(13 * 4);
    L.off_chsrc = take(std::max<size_t>(n_ch, 1) * 4);
    L.off_chw = take(std::max<size_t>(n_ch, 1) * 8);
    L.table_bytes = off;
    L.wave_bytes = (int)generic_wave_bytes(Nc, fft.window / 2, F > 0 ? F : 1);
    L.waves = 4;
    while (L.waves > 1 && (size_t)L.table_bytes + (size_t)L.waves * L.wave_bytes > 160 * 1024) --L.waves;
    if (!blob) return;
    blob->assign((size_t)L.table_bytes, 0);
    unsigned char *b = blob->data();
    memcpy(b + L.off_tw, fft.tw.data(), (size_t)Nc * 16);
    if (fft.even) memcpy(b + L.off_post, fft.post.data(), (size_t)Nc * 16);
    if (mel && !mel->w.empty()) {
        memcpy(b + L.off_mello, mel->lo.data(), 40 * 4);
        memcpy(b + L.off_melcnt, mel->cnt.data(), 40 * 4);
        memcpy(b + L.off_meloff, mel->off.data(), 40 * 4);
        memcpy(b + L.off_melw, mel->w.data(), n_melw * 8);
        double dct[kNumMfcc * kNumMel];
        build_dct(dct);
        double *d = reinterpret_cast<double *>(b + L.off_dct);
        for (int q = 0; q < 13; ++q)
            for (int n = 0; n < 40; ++n) d[q * 41 + n] = dct[q * 40 + n];
    }
    if (chroma && !chroma->src.empty()) {
        memcpy(b + L.off_chstart, chroma->class_start, 13 * 4);
        memcpy(b + L.off_chsrc, chroma->src.data(), n_ch * 4);
        memcpy(b + L.off_chw, chroma->w.data(), n_ch * 8);
    }
}

inline size_t generic_lds_bytes(const GenLayout &L) { return (size_t)L.table_bytes + (size_t)L.waves * L.wave_bytes; }

template <typename T>
__global__ __launch_bounds__(256, 2) void st_generic_kernel(PlanDev P, GenLayout L,
                                                             const unsigned char *__restrict__ blob,
                                                             const T *__restrict__ sig,
                                                             const ClipDev *__restrict__ clips,
                                                             const ClipNorm *__restrict__ norms,
                                                             const Tile *__restrict__ tiles, int n_tiles,
                                                             double *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    {
        const int4 *src4 = reinterpret_cast<const int4 *>(blob);
        int4 *dst4 = reinterpret_cast<int4 *>(smem);
        for (int n = threadIdx.x; n < L.table_bytes / 16; n += blockDim.x) dst4[n] = src4[n];
    }
    __syncthreads();       // the only workgroup-wide barrier
    Tabs tb;
    tb.tw = reinterpret_cast<const double2 *>(smem + L.off_tw);
    tb.post = reinterpret_cast<const double2 *>(smem + L.off_post);
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
    const int Nc = P.Nc, Nf = P.Nf, F = P.F > 0 ? P.F : 1;
    const int nfp = (Nf + 1) & ~1;
    unsigned char *wb = smem + L.table_bytes + wave * L.wave_bytes;
    double2 *bufA = reinterpret_cast<double2 *>(wb);
    double2 *bufB = bufA + Nc;
    double *spec0 = reinterpret_cast<double *>(bufB + Nc);
    double *spec1 = spec0 + nfp;
    double *otile = spec1 + nfp;
    double *fv = otile + kFlush * F;
    double *msp = fv + 48;

    const Tile tl = tiles[tile_id];
    const ClipDev c = clips[tl.clip];
    const ClipNorm nm = wave_clip_norm<T>(P, c, norms, tl.clip, lane);
    const T *x0 = sig + c.sample_off + P.frame_origin;
    const long long Tc = c.T;
    double *oc = out + c.out_off;

    const int hneed = (P.mode == 0) ? (P.deltas ? 2 : 1) : 0;
    const int h = min(hneed, tl.t0);
    double *prv = spec0, *cur = spec1;
    double vprev = 0.0;
    int nslot = 0, tbase = tl.t0;
    const int tend = tl.t0 + tl.cnt;
    for (int t = tl.t0 - h; t < tend; ++t) {
        const T *x = x0 + (long long)t * P.S;
        frame_load<T>(P, x, nm, bufA, lane);
        const bool want = (P.mode == 0) && ((t >= tl.t0) || (P.deltas && t == tl.t0 - 1));
        TimeFeat tf;
        tf.e_tot = 0.0; tf.ent_e = 0.0; tf.zc = 0;
        if (want) tf = time_features(P, bufA, lane);
        frame_fft(P, tb, bufA, bufB, cur, lane);
        if (P.mode == 1) {            // spectrogram row (ShortTermFeatures.py:422)
            double *row = oc + (long long)t * Nf;
            for (int k = lane; k < Nf; k += kWave) row[k] = cur[k];
        } else if (P.mode == 2) {     // chromagram row (:356-359)
            double p = 0.0;
            for (int k = lane; k < Nf; k += kWave) { const double X = cur[k]; p = fma(X, X, p); }
            p = wsum(p);
            const double ch = chroma_class(tb, cur, p, lane);
            if (lane < 12) oc[(long long)t * 12 + lane] = ch;
        } else {
            if (want) {
                frame_features(P, tb, tf, cur, (t == 0) ? cur : prv, fv, msp, lane);
                const double v = (lane < kBase) ? fv[lane] : 0.0;
                if (t >= tl.t0) {
                    if (lane < kBase) {
                        otile[nslot * F + lane] = v;
                        if (P.deltas) otile[nslot * F + kBase + lane] = (t == 0) ? 0.0 : v - vprev;
                    }
                    ++nslot;
                }
                vprev = v;
            }
            if (nslot == kFlush || (t == tend - 1 && nslot > 0)) {
                wsync();
                // row segments: nslot consecutive frames of feature row f are contiguous in [F][T]
                for (int idx = lane; idx < F * kFlush; idx += kWave) {
                    const int f = idx / kFlush, i = idx % kFlush;
                    if (i < nslot) oc[(long long)f * Tc + tbase + i] = otile[i * F + f];
                }
                wsync();
                tbase += nslot;
                nslot = 0;
            }
        }
        double *tmp = prv; prv = cur; cur = tmp;
    }
}

}  // namespace paa
