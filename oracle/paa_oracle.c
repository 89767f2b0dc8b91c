/*
 * paa_oracle.c -- plain-C restatement of the pyAudioAnalysis short-term path (CPU, single thread).
 *
 * TEST INFRASTRUCTURE ONLY -- never linked into, loaded by or called from the product package.  It is the
 * second, independent oracle (besides oracle/paa_oracle.py) and the scalar CPU baseline that bench.py's
 * `cpu_baseline` leg times ("kind": "port").  Pinned against the tests/golden fixtures (outputs of the unmodified
 * reference) by tests/test_oracle_vs_golden.py::test_c_oracle_*.
 *
 * Follows, operation by operation (paths relative to /root/reference/pyAudioAnalysis):
 *   normalisation ............ ShortTermFeatures.py:14-19, 567-570
 *   |FFT|[0:W/2] / (W/2) ..... ShortTermFeatures.py:617-621  (scipy.fftpack.fft = pocketfft, not in the tree;
 *                              restated here as a recursive mixed-radix DFT, O(N sum of prime factors))
 *   34 features .............. ShortTermFeatures.py:22-140, 191-321, 626-667
 *   deltas / layout .......... ShortTermFeatures.py:668-685
 *   spectrogram / chromagram . ShortTermFeatures.py:324-452 (incl. the truncated last chromagram frame)
 *   mid-term mean / std ...... MidTermFeatures.py:110-126
 *
 * Build: gcc -O2 -shared -fPIC oracle/paa_oracle.c -o oracle/_build/libpaa_oracle.so -lm   (oracle/Makefile)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define EPS 2.220446049250313e-16 /* sys.float_info.epsilon, ShortTermFeatures.py:11 */
#define NMEL 40
#define NMFCC 13
#define NBASE 34

typedef struct { double re, im; } cpx;

/* ---- recursive mixed-radix DFT (decimation in time); tw[j] = exp(-2 pi i j / n_top) ---------------------- */
static void dft_rec(const cpx *in, int stride, cpx *out, int n, int n_top, const cpx *tw, cpx *scratch) {
    if (n == 1) { out[0] = in[0]; return; }
    int p = 2;
    while (n % p) ++p;                    /* smallest prime factor */
    const int m = n / p;
    for (int r = 0; r < p; ++r)           /* p sub-transforms of length m on the decimated inputs */
        dft_rec(in + (size_t)r * stride, stride * p, out + (size_t)r * m, m, n_top, tw, scratch + (size_t)n);
    /* combine: X[k + q m] = sum_r W_p^{r q} (W_n^{r k} Y_r[k]) */
    const int tstep = n_top / n, pstep = n_top / p;
    for (int k = 0; k < m; ++k) {
        cpx *y = scratch;                 /* twiddled inputs of this butterfly */
        y[0] = out[k];
        for (int r = 1; r < p; ++r) {
            const cpx v = out[(size_t)r * m + k], w = tw[(size_t)r * k * tstep];
            y[r].re = v.re * w.re - v.im * w.im;
            y[r].im = v.re * w.im + v.im * w.re;
        }
        if (p == 2) {
            out[k].re = y[0].re + y[1].re; out[k].im = y[0].im + y[1].im;
            out[m + k].re = y[0].re - y[1].re; out[m + k].im = y[0].im - y[1].im;
            continue;
        }
        for (int q = 0; q < p; ++q) {     /* plain p-point DFT; in place is safe, the inputs live in y[] */
            double sr = y[0].re, si = y[0].im;
            int idx = 0;
            for (int r = 1; r < p; ++r) {
                idx += q; if (idx >= p) idx -= p;
                const cpx w = tw[(size_t)idx * pstep];
                sr += y[r].re * w.re - y[r].im * w.im;
                si += y[r].re * w.im + y[r].im * w.re;
            }
            out[(size_t)q * m + k].re = sr;
            out[(size_t)q * m + k].im = si;
        }
    }
}

typedef struct {
    double fs;
    int window, nfft;
    cpx *tw, *buf_in, *buf_out, *scratch;
    /* mel bank (dense rows would waste time: start/count/weights per filter) */
    int mel_lo[NMEL], mel_cnt[NMEL];
    double *mel_w[NMEL];
    double dct[NMFCC][NMEL];
    /* chroma gather list in ascending slot order */
    int n_ch, *ch_src, *ch_slot;
    double *ch_w;
} tables_t;

static void tables_free(tables_t *t) {
    free(t->tw); free(t->buf_in); free(t->buf_out); free(t->scratch);
    for (int m = 0; m < NMEL; ++m) free(t->mel_w[m]);
    free(t->ch_src); free(t->ch_slot); free(t->ch_w);
}

/* returns 0, or -6 / -7 / -8 for the reference's ValueError / IndexError / mel IndexError cases */
static int tables_build(tables_t *t, double fs, int window) {
    memset(t, 0, sizeof(*t));
    t->fs = fs; t->window = window; t->nfft = window / 2;
    const int n = window, nfft = t->nfft;
    t->tw = (cpx *)malloc(sizeof(cpx) * (size_t)n);
    t->buf_in = (cpx *)malloc(sizeof(cpx) * (size_t)n);
    t->buf_out = (cpx *)malloc(sizeof(cpx) * (size_t)n);
    t->scratch = (cpx *)malloc(sizeof(cpx) * (size_t)(4 * n + 64));
    for (int j = 0; j < n; ++j) {
        const long double a = -6.283185307179586476925286766559L * (long double)j / (long double)n;
        t->tw[j].re = (double)cosl(a); t->tw[j].im = (double)sinl(a);
    }
    /* mel bank, ShortTermFeatures.py:204-231 (bin axis k * fs / nfft: quirk kept) */
    double edges[NMEL + 2];
    for (int i = 0; i < 13; ++i) edges[i] = 133.33 + (double)i * (200.0 / 3.0);
    for (int i = 13; i < NMEL + 2; ++i) edges[i] = edges[12] * pow(1.0711703, (double)(i - 12));
    for (int m = 0; m < NMEL; ++m) {
        const double lo = edges[m], mid = edges[m + 1], hi = edges[m + 2], peak = 2.0 / (hi - lo);
        const long k_lo = (long)floor(lo * nfft / fs) + 1, k_mid = (long)floor(mid * nfft / fs) + 1,
                   k_hi = (long)floor(hi * nfft / fs) + 1;
        if (k_hi > k_lo && k_hi - 1 >= nfft) return -8;
        t->mel_lo[m] = (int)k_lo;
        t->mel_cnt[m] = (int)(k_hi > k_lo ? k_hi - k_lo : 0);
        t->mel_w[m] = (double *)calloc((size_t)(t->mel_cnt[m] + 1), sizeof(double));
        for (long k = k_lo; k < k_hi; ++k) {
            const double f = (double)k / (1.0 * nfft) * fs;
            t->mel_w[m][k - k_lo] = (k < k_mid) ? (peak / (mid - lo)) * (f - lo) : (peak / (hi - mid)) * (hi - f);
        }
    }
    for (int k = 0; k < NMFCC; ++k)
        for (int i = 0; i < NMEL; ++i) {
            double v = sqrt(2.0 / NMEL) * cos(M_PI * k * (2 * i + 1) / (2.0 * NMEL));
            if (k == 0) v *= 1.0 / sqrt(2.0);
            t->dct[k][i] = v;
        }
    /* chroma, ShortTermFeatures.py:257-302 */
    long *slot = (long *)malloc(sizeof(long) * (size_t)nfft);
    double *count = (double *)calloc((size_t)nfft, sizeof(double));
    int *owner = (int *)malloc(sizeof(int) * (size_t)nfft);
    long smax = -(1L << 40);
    int any_over = 0;
    for (int f = 0; f < nfft; ++f) {
        const double freq = ((double)(f + 1) * fs) / (double)(2 * nfft);
        slot[f] = (long)nearbyint(12.0 * log2(freq / 27.50));
        if (slot[f] > smax) smax = slot[f];
        if (slot[f] > nfft) any_over = 1;
    }
    if (smax >= nfft) { free(slot); free(count); free(owner); return any_over ? -6 : -7; }
    for (int f = 0; f < nfft; ++f) {
        int c = 0;
        for (int g = f; g >= 0 && slot[g] == slot[f]; --g) ++c;
        for (int g = f + 1; g < nfft && slot[g] == slot[f]; ++g) ++c;
        count[f] = (double)c;
        owner[f] = -1;
    }
    for (int f = 0; f < nfft; ++f) owner[slot[f] < 0 ? slot[f] + nfft : slot[f]] = f;
    t->ch_src = (int *)malloc(sizeof(int) * (size_t)nfft);
    t->ch_slot = (int *)malloc(sizeof(int) * (size_t)nfft);
    t->ch_w = (double *)malloc(sizeof(double) * (size_t)nfft);
    for (int p = 0; p < nfft; ++p) {
        if (owner[p] < 0) continue;
        const long sp = slot[p];
        t->ch_src[t->n_ch] = owner[p];
        t->ch_slot[t->n_ch] = p;
        t->ch_w[t->n_ch] = 1.0 / count[sp < 0 ? sp + nfft : sp];
        ++t->n_ch;
    }
    free(slot); free(count); free(owner);
    return 0;
}

static double block_entropy(const double *v2, int len, double total) {
    const int blk = len / 10;
    double ent = 0.0;
    for (int j = 0; j < 10; ++j) {
        double s = 0.0;
        for (int n = j * blk; n < (j + 1) * blk; ++n) s += v2[n];
        s /= (total + EPS);
        ent -= s * log2(s + EPS);
    }
    return ent;
}

static double sgn(double v) { return (v > 0.0) - (v < 0.0); }

/* one frame: x[0..W) normalised samples, X / Xp current / previous magnitude spectrum, out[34] */
static void frame_vector(const tables_t *t, const double *x, const double *X, const double *Xp, double *tmp,
                         double *out) {
    const int W = t->window, nf = t->nfft;
    const double fs = t->fs;
    double cross = 0.0, e_tot = 0.0;
    for (int n = 0; n < W; ++n) {
        if (n > 0) cross += fabs(sgn(x[n]) - sgn(x[n - 1]));
        tmp[n] = x[n] * x[n];
        e_tot += tmp[n];
    }
    out[0] = (cross / 2.0) / (double)(W - 1);
    out[1] = e_tot / (double)W;
    out[2] = block_entropy(tmp, W, e_tot);
    double peak = 0.0, sX = 0.0, sXe = 0.0, sXpe = 0.0, p_tot = 0.0;
    for (int k = 0; k < nf; ++k) {
        if (X[k] > peak) peak = X[k];
        sX += X[k]; sXe += X[k] + EPS; sXpe += Xp[k] + EPS;
        tmp[k] = X[k] * X[k];
        p_tot += tmp[k];
    }
    const double f0 = fs / (2.0 * nf), div = (peak == 0.0) ? EPS : peak;
    double num = 0.0, den = 0.0;
    for (int k = 0; k < nf; ++k) { num += (k + 1) * f0 * (X[k] / div); den += X[k] / div; }
    den += EPS;
    const double cen = num / den;
    double spr = 0.0, flux = 0.0;
    for (int k = 0; k < nf; ++k) {
        const double d = (k + 1) * f0 - cen;
        spr += d * d * (X[k] / div);
        const double df = X[k] / sXe - Xp[k] / sXpe;
        flux += df * df;
    }
    out[3] = cen / (fs / 2.0);
    out[4] = sqrt(spr / den) / (fs / 2.0);
    out[5] = block_entropy(tmp, nf, p_tot);
    out[6] = flux;
    out[7] = 0.0;
    {
        double run = 0.0;
        const double thr = 0.90 * p_tot;
        for (int k = 0; k < nf; ++k) {
            run += tmp[k];
            if (run + EPS > thr) { out[7] = (double)k / (double)nf; break; }
        }
    }
    double mspec[NMEL];
    for (int m = 0; m < NMEL; ++m) {
        double acc = 0.0;
        for (int i = 0; i < t->mel_cnt[m]; ++i) acc += X[t->mel_lo[m] + i] * t->mel_w[m][i];
        mspec[m] = log10(acc + EPS);
    }
    for (int k = 0; k < NMFCC; ++k) {
        double acc = 0.0;
        for (int i = 0; i < NMEL; ++i) acc += t->dct[k][i] * mspec[i];
        out[8 + k] = acc;
    }
    double chroma[12] = {0};
    for (int e = 0; e < t->n_ch; ++e) chroma[t->ch_slot[e] % 12] += tmp[t->ch_src[e]] * t->ch_w[e];
    double mean = 0.0;
    for (int c = 0; c < 12; ++c) { chroma[c] = (p_tot == 0.0) ? chroma[c] / EPS : chroma[c] / p_tot; mean += chroma[c]; }
    mean /= 12.0;
    double var = 0.0;
    for (int c = 0; c < 12; ++c) { out[21 + c] = chroma[c]; var += (chroma[c] - mean) * (chroma[c] - mean); }
    out[33] = sqrt(var / 12.0);
    (void)sX;
}

/* signal: float64 samples (int16 callers convert first, like np.double(), :567).  out: [F][T] row-major.
 * Returns T (>= 1), 0 when the clip is shorter than a window, or a negative table error. */
long long paa_c_feature_extraction(const double *signal, long long n, double fs, int window, int step, int deltas,
                                   double *out) {
    if (window < 2 || step < 1 || n < window) return 0;
    tables_t t;
    const int rc = tables_build(&t, fs, window);
    if (rc) { tables_free(&t); return rc; }
    const long long T = (n - window) / step + 1;
    const int F = deltas ? 2 * NBASE : NBASE, nf = t.nfft;
    double *x = (double *)malloc(sizeof(double) * (size_t)n);
    double mean = 0.0, peak = 0.0;
    for (long long i = 0; i < n; ++i) { x[i] = signal[i] / 32768.0; mean += x[i]; }
    mean /= (double)n;
    for (long long i = 0; i < n; ++i) { x[i] -= mean; if (fabs(x[i]) > peak) peak = fabs(x[i]); }
    for (long long i = 0; i < n; ++i) x[i] /= (peak + 1e-10);
    double *X = (double *)malloc(sizeof(double) * (size_t)nf), *Xp = (double *)malloc(sizeof(double) * (size_t)nf);
    double *tmp = (double *)malloc(sizeof(double) * (size_t)(window + nf + 8));
    double v[NBASE], vprev[NBASE];
    for (long long f = 0; f < T; ++f) {
        const double *fr = x + f * step;
        for (int i = 0; i < window; ++i) { t.buf_in[i].re = fr[i]; t.buf_in[i].im = 0.0; }
        dft_rec(t.buf_in, 1, t.buf_out, window, window, t.tw, t.scratch);
        for (int k = 0; k < nf; ++k) X[k] = hypot(t.buf_out[k].re, t.buf_out[k].im) / (double)nf;
        if (f == 0) memcpy(Xp, X, sizeof(double) * (size_t)nf);
        frame_vector(&t, fr, X, Xp, tmp, v);
        for (int r = 0; r < NBASE; ++r) {
            out[(long long)r * T + f] = v[r];
            if (deltas) out[(long long)(NBASE + r) * T + f] = (f == 0) ? 0.0 : v[r] - vprev[r];
        }
        memcpy(vprev, v, sizeof(v));
        memcpy(Xp, X, sizeof(double) * (size_t)nf);
    }
    (void)F;
    free(x); free(X); free(Xp); free(tmp);
    tables_free(&t);
    return T;
}


/* ---- spectrogram / chromagram / mid-term statistics (ShortTermFeatures.py:324-452, MidTermFeatures.py:110-126) ---- */
static double *normalized_copy(const double *signal, long long n) {
    double *x = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    double mean = 0.0, peak = 0.0;
    for (long long i = 0; i < n; ++i) { x[i] = signal[i] / 32768.0; mean += x[i]; }
    mean /= (double)n;
    for (long long i = 0; i < n; ++i) { x[i] -= mean; if (fabs(x[i]) > peak) peak = fabs(x[i]); }
    for (long long i = 0; i < n; ++i) x[i] /= (peak + 1e-10);
    return x;
}

/* |DFT_m(x)|[0:nf] / nf for a frame of m samples (m < window for the truncated chromagram tail) */
static void magnitude_any(const double *x, int m, int nf, double *X) {
    cpx *tw = (cpx *)malloc(sizeof(cpx) * (size_t)m), *in = (cpx *)malloc(sizeof(cpx) * (size_t)m);
    cpx *out = (cpx *)malloc(sizeof(cpx) * (size_t)m), *scr = (cpx *)malloc(sizeof(cpx) * (size_t)(4 * m + 64));
    for (int j = 0; j < m; ++j) {
        const long double a = -6.283185307179586476925286766559L * (long double)j / (long double)m;
        tw[j].re = (double)cosl(a); tw[j].im = (double)sinl(a);
        in[j].re = x[j]; in[j].im = 0.0;
    }
    dft_rec(in, 1, out, m, m, tw, scr);
    for (int k = 0; k < nf; ++k) X[k] = hypot(out[k].re, out[k].im) / (double)nf;
    free(tw); free(in); free(out); free(scr);
}

/* rows = int((n - W) / S) + 1 are allocated, frame i starts at W + i S (:413-422); out is [rows][W/2], zero-filled
 * rows stay zero.  Returns rows, 0 when too short. */
long long paa_c_spectrogram(const double *signal, long long n, int window, int step, double *out) {
    if (window < 2 || step < 1 || n < window) return 0;
    const int nf = window / 2;
    const long long rows = (n - window) / step + 1;
    memset(out, 0, sizeof(double) * (size_t)(rows * nf));
    double *x = normalized_copy(signal, n);
    long long i = 0;
    for (long long p = window; p < n - window + 1; p += step, ++i) magnitude_any(x + p, window, nf, out + i * nf);
    free(x);
    return rows;
}

/* rows = int((n - S - W) / S) + 1 (:347); loop range(W, n - S, S); the last frame may be shorter than W (:349-355).
 * out is [rows][12].  Returns rows (>= 1), 0 when too short, or a negative table error. */
long long paa_c_chromagram(const double *signal, long long n, double fs, int window, int step, double *out) {
    if (window < 2 || step < 1 || n - step - window < 0) return 0;
    tables_t t;
    const int rc = tables_build(&t, fs, window);
    if (rc) { tables_free(&t); return rc; }
    const int nf = t.nfft;
    const long long rows = (n - step - window) / step + 1;
    memset(out, 0, sizeof(double) * (size_t)(rows * 12));
    double *x = normalized_copy(signal, n);
    double *X = (double *)malloc(sizeof(double) * (size_t)nf);
    long long i = 0;
    for (long long p = window; p < n - step; p += step, ++i) {
        const long long len = (p + window <= n) ? window : n - p;
        if (len < nf) { free(x); free(X); tables_free(&t); return -6; }      /* X[0:nf] cannot be taken (:354) */
        magnitude_any(x + p, (int)len, nf, X);
        double tot = 0.0, c[12] = {0};
        for (int k = 0; k < nf; ++k) tot += X[k] * X[k];
        for (int e = 0; e < t.n_ch; ++e) c[t.ch_slot[e] % 12] += X[t.ch_src[e]] * X[t.ch_src[e]] * t.ch_w[e];
        for (int k = 0; k < 12; ++k) out[i * 12 + k] = (tot == 0.0) ? c[k] / EPS : c[k] / tot;
    }
    free(x); free(X);
    tables_free(&t);
    return rows;
}

/* st is [F][T]; out is [2F][M], M = ceil(T / step_ratio): mean and population std of st[i][c : min(c + ratio, T)],
 * non-finite values replaced like numpy.nan_to_num */
long long paa_c_mid_statistics(const double *st, int F, long long T, long long ratio, long long step_ratio, double *out) {
    if (F < 1 || T < 1 || step_ratio < 1) return 0;
    const long long M = (T + step_ratio - 1) / step_ratio;
    for (int i = 0; i < F; ++i)
        for (long long m = 0; m < M; ++m) {
            const long long c0 = m * step_ratio, c1 = (c0 + ratio < T) ? c0 + ratio : T;
            double mean = 0.0, var = 0.0;
            for (long long k = c0; k < c1; ++k) mean += st[(long long)i * T + k];
            mean /= (double)(c1 - c0);
            for (long long k = c0; k < c1; ++k) { const double d = st[(long long)i * T + k] - mean; var += d * d; }
            double sd = sqrt(var / (double)(c1 - c0));
            if (isnan(mean)) mean = 0.0;
            if (isnan(sd)) sd = 0.0;
            if (isinf(mean)) mean = mean > 0 ? 1.7976931348623157e308 : -1.7976931348623157e308;
            if (isinf(sd)) sd = 1.7976931348623157e308;
            out[(long long)i * M + m] = mean;
            out[(long long)(F + i) * M + m] = sd;
        }
    return M;
}
