// Generic short-term feature kernel: any window the LDS envelope admits, any step.
//
// One workgroup = one wave (64 lanes) = one RUN of consecutive frames of one clip.  Per frame:
//   1. the frame is read from HBM/L2 (raw int16 or f64 samples, contiguous -> coalesced),
//      normalised on load with the clip's affine constants and packed as a complex sequence
//      (real-input trick: window/2 complex points when the window is even),
//   2. a Stockham autosort FFT runs in LDS (ping-pong buffers, host-chosen radix schedule,
//      hard-coded radix 2/3/4/5 butterflies, O(R^2) passes for other primes),
//   3. the magnitude spectrum |X|/num_fft stays in LDS and the 34 features are reduced from it
//      with wave shuffles; the previous frame's spectrum is kept in LDS for the flux,
//   4. feature columns are staged [kFlush][F] in LDS and stored as row segments.
// Halo: a run that does not start at frame 0 first recomputes frame t0-1 (spectrum for the
// flux; all features when deltas are on) and, with deltas, the spectrum of t0-2.
//
// Replaces the while loop at ShortTermFeatures.py:608-682 and its helpers (:22-140, :236-321).
#pragma once
#include "device_common.hpp"

namespace paa {

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

// any radix: one output per lane-iteration, inputs re-read from LDS (R reads per output)
__device__ __forceinline__ void stockham_pass_any(const double2 *__restrict__ in, double2 *__restrict__ out,
                                                  int Nc, int R, int Ns, const double2 *__restrict__ tw, int lane) {
    const int nb = Nc / R;
    const int a = Nc / (Ns * R);
    for (int o = lane; o < Nc; o += kWave) {
        const int j = o % nb, q = o / nb;
        const int k = j % Ns;
        const int step = (int)(((long long)k * a + (long long)q * nb) % Nc);
        int idx = 0;
        double2 acc = make_double2(0.0, 0.0);
        for (int p = 0; p < R; ++p) {
            const double2 x = in[j + p * nb];
            const double2 w = tw[idx];
            acc.x += fma(x.x, w.x, -x.y * w.y);
            acc.y += fma(x.x, w.y, x.y * w.x);
            idx += step;
            if (idx >= Nc) idx -= Nc;
        }
        out[(j - k) * R + k + q * Ns] = acc;
    }
}

// ---- one frame: samples -> |X|/Nf in `spec` ------------------------------------------------
// T = int16_t or double.  x points at the frame's first sample.  `len` < W only for the
// truncated chromagram tail, which uses a different kernel; here len == W.
template <typename T>
__device__ __forceinline__ void frame_spectrum(const PlanDev &P, const T *__restrict__ x, ClipNorm nm,
                                               double2 *bufA, double2 *bufB, double *spec, int lane) {
    const double sc = 1.0 / 32768.0;
    const int Nc = P.Nc;
    if (P.even) {
        for (int n = lane; n < Nc; n += kWave) {
            const double re = fma(load_sample<T>(x + 2 * n), sc, -nm.mean) * nm.inv;
            const double im = fma(load_sample<T>(x + 2 * n + 1), sc, -nm.mean) * nm.inv;
            bufA[n] = make_double2(re, im);
        }
    } else {
        for (int n = lane; n < Nc; n += kWave)
            bufA[n] = make_double2(fma(load_sample<T>(x + n), sc, -nm.mean) * nm.inv, 0.0);
    }
    __syncthreads();
    double2 *src = bufA, *dst = bufB;
    int Ns = 1;
    for (int p = 0; p < P.n_pass; ++p) {
        const int R = P.radix[p];
        switch (R) {
            case 2: stockham_pass<2>(src, dst, Nc, Ns, P.tw, lane); break;
            case 3: stockham_pass<3>(src, dst, Nc, Ns, P.tw, lane); break;
            case 4: stockham_pass<4>(src, dst, Nc, Ns, P.tw, lane); break;
            case 5: stockham_pass<5>(src, dst, Nc, Ns, P.tw, lane); break;
            default: stockham_pass_any(src, dst, Nc, R, Ns, P.tw, lane); break;
        }
        Ns *= R;
        double2 *t = src; src = dst; dst = t;
        __syncthreads();
    }
    const double invNf = 1.0 / (double)P.Nf;     // X / len(X)  (ShortTermFeatures.py:621)
    if (P.even) {
        // X[k] = E[k] + w^k O[k],  E = (Z[k] + conj Z[H-k]) / 2,  O = -i (Z[k] - conj Z[H-k]) / 2
        for (int k = lane; k < P.Nf; k += kWave) {
            const double2 zk = src[k];
            const double2 zm = src[k == 0 ? 0 : Nc - k];
            const double2 e = make_double2(0.5 * (zk.x + zm.x), 0.5 * (zk.y - zm.y));
            const double2 d = make_double2(0.5 * (zk.x - zm.x), 0.5 * (zk.y + zm.y));
            const double2 o = make_double2(d.y, -d.x);
            const double2 wo = cmul(P.post[k], o);
            const double xr = e.x + wo.x, xi = e.y + wo.y;
            spec[k] = sqrt(fma(xr, xr, xi * xi)) * invNf;
        }
    } else {
        for (int k = lane; k < P.Nf; k += kWave) {
            const double2 z = src[k];
            spec[k] = sqrt(fma(z.x, z.x, z.y * z.y)) * invNf;
        }
    }
    __syncthreads();
}

// chroma of one spectrum: lanes 0..11 return their pitch class (ShortTermFeatures.py:285-308)
__device__ __forceinline__ double chroma_class(const PlanDev &P, const double *spec, double sP, int lane) {
    double acc = 0.0;
    if (lane < 12) {
        const int b = P.ch_start[lane], e = P.ch_start[lane + 1];
        for (int i = b; i < e; ++i) {
            const double x = spec[P.ch_src[i]];
            acc += (x * x) * P.ch_w[i];
        }
        acc = (sP == 0.0) ? acc / kEps : acc / sP;
    }
    return acc;
}

// ---- the 34 base features of one frame into fv[0..33] (ShortTermFeatures.py:626-667) -------
template <typename T>
__device__ __forceinline__ void frame_features(const PlanDev &P, const T *__restrict__ x, ClipNorm nm,
                                               const double *cur, const double *prv, double *fv,
                                               double *msp, int lane) {
    const double sc = 1.0 / 32768.0;
    const int W = P.W, Nf = P.Nf;
    // ---------- time domain: zcr, energy, energy entropy (:22-51)
    double eblk[10];
    double e_tail = 0.0;
    int zc = 0;
    {
        const int L = P.blk_t;
#pragma unroll
        for (int j = 0; j < 10; ++j) {
            double p = 0.0;
            for (int n = j * L + lane; n < (j + 1) * L; n += kWave) {
                const double d = fma(load_sample<T>(x + n), sc, -nm.mean);
                const double y = d * nm.inv;
                p = fma(y, y, p);
                if (n > 0) {
                    const double dp = fma(load_sample<T>(x + n - 1), sc, -nm.mean);
                    zc += abs(((d > 0.0) - (d < 0.0)) - ((dp > 0.0) - (dp < 0.0)));
                }
            }
            eblk[j] = wave_sum(p);
        }
        for (int n = 10 * L + lane; n < W; n += kWave) {
            const double d = fma(load_sample<T>(x + n), sc, -nm.mean);
            const double y = d * nm.inv;
            e_tail = fma(y, y, e_tail);
            if (n > 0) {
                const double dp = fma(load_sample<T>(x + n - 1), sc, -nm.mean);
                zc += abs(((d > 0.0) - (d < 0.0)) - ((dp > 0.0) - (dp < 0.0)));
            }
        }
        e_tail = wave_sum(e_tail);
        zc = wave_sum_i(zc);
    }
    double e_tot = e_tail;
#pragma unroll
    for (int j = 0; j < 10; ++j) e_tot += eblk[j];

    // ---------- spectrum sweep A: sums, max, block energies (:57-107)
    const double f0 = P.fs / (2.0 * (double)Nf);
    double pblk[10];
    double sX = 0.0, sXe = 0.0, sXp = 0.0, sIX = 0.0, mx = 0.0, p_tail = 0.0;
    {
        const int L = P.blk_f;
#pragma unroll
        for (int j = 0; j < 10; ++j) {
            double p = 0.0;
            for (int k = j * L + lane; k < (j + 1) * L; k += kWave) {
                const double X = cur[k];
                sX += X;
                sXe += X + kEps;
                sXp += prv[k] + kEps;
                sIX = fma((double)(k + 1) * f0, X, sIX);
                mx = fmax(mx, X);
                p = fma(X, X, p);
            }
            pblk[j] = wave_sum(p);
        }
        for (int k = 10 * L + lane; k < Nf; k += kWave) {
            const double X = cur[k];
            sX += X;
            sXe += X + kEps;
            sXp += prv[k] + kEps;
            sIX = fma((double)(k + 1) * f0, X, sIX);
            mx = fmax(mx, X);
            p_tail = fma(X, X, p_tail);
        }
        p_tail = wave_sum(p_tail);
    }
    sX = wave_sum(sX);
    sXe = wave_sum(sXe);
    sXp = wave_sum(sXp);
    sIX = wave_sum(sIX);
    mx = wave_max(mx);
    double sP = p_tail;
#pragma unroll
    for (int j = 0; j < 10; ++j) sP += pblk[j];

    // entropies: lane j (<10) owns time block j, lane 10+j spectral block j (:46-50, :101-105)
    double ent_e, ent_f;
    {
        double num = 0.0, den = 1.0;
#pragma unroll
        for (int j = 0; j < 10; ++j) {
            if (lane == j) { num = eblk[j]; den = e_tot + kEps; }
            if (lane == 10 + j) { num = pblk[j]; den = sP + kEps; }
        }
        const double s = num / den;
        const double term = (lane < 20) ? -(s * log2(s + kEps)) : 0.0;
        ent_e = wave_sum(lane < 10 ? term : 0.0);
        ent_f = wave_sum(lane >= 10 ? term : 0.0);
    }

    // ---------- centroid, then sweep B: spread + flux (:57-82, :110-124)
    const double r = (mx == 0.0) ? 1.0 / kEps : 1.0 / mx;
    const double den = sX * r + kEps;
    const double cen = (sIX * r) / den;
    const double rX = 1.0 / sXe, rXp = 1.0 / sXp;
    double sSp = 0.0, sFl = 0.0;
    for (int k = lane; k < Nf; k += kWave) {
        const double X = cur[k];
        const double dv = (double)(k + 1) * f0 - cen;
        sSp = fma(dv * dv, X * r, sSp);
        // separately rounded products: frame 0 (prv == cur) must give exactly 0 like the reference (:624-625)
        const double df = __dmul_rn(X, rX) - __dmul_rn(prv[k], rXp);
        sFl = fma(df, df, sFl);
    }
    sSp = wave_sum(sSp);
    sFl = wave_sum(sFl);
    const double spread = sqrt(sSp / den);

    // ---------- roll-off: first k with cumsum(X^2)[k] + eps > 0.9 * sum(X^2) (:127-140)
    int first = 0x7fffffff;
    {
        const double thr = 0.90 * sP;
        const int c = (Nf + kWave - 1) / kWave;
        const int kb = lane * c, ke = min(Nf, kb + c);
        double cs = 0.0;
        for (int k = kb; k < ke; ++k) { const double X = cur[k]; cs = fma(X, X, cs); }
        double run = wave_scan_incl(cs, lane) - cs;
        for (int k = kb; k < ke; ++k) {
            const double X = cur[k];
            run = fma(X, X, run);
            if (run + kEps > thr) { first = k; break; }
        }
        first = wave_min_i(first);
    }

    // ---------- MFCC: sparse mel dot, log10, 13 x 40 DCT (:236-254)
    if (lane < 40) {
        const int lo = P.mel_lo[lane], cnt = P.mel_cnt[lane];
        const double *w = P.mel_w + P.mel_off[lane];
        double acc = 0.0;
        for (int i = 0; i < cnt; ++i) acc = fma(cur[lo + i], w[i], acc);
        msp[lane] = log10(acc + kEps);
    }
    // ---------- chroma (:277-321)
    const double chroma = chroma_class(P, cur, sP, lane);
    __syncthreads();
    if (lane < 13) {
        const double *m = P.dct + lane * 40;
        double acc = 0.0;
        for (int n = 0; n < 40; ++n) acc = fma(m[n], msp[n], acc);
        fv[8 + lane] = acc;
    }
    if (lane < 12) fv[21 + lane] = chroma;
    if (lane == 0) {
        fv[0] = ((double)zc / 2.0) / (double)(W - 1);
        fv[1] = e_tot / (double)W;
        fv[2] = ent_e;
        fv[3] = cen / (P.fs / 2.0);
        fv[4] = spread / (P.fs / 2.0);
        fv[5] = ent_f;
        fv[6] = (cur == prv) ? 0.0 : sFl;      // first frame: previous spectrum = itself (:624-625)
        fv[7] = (first == 0x7fffffff) ? 0.0 : (double)first / (double)Nf;
    }
    __syncthreads();
    if (lane == 0) {        // population std of the 12 chroma values (:667)
        double m = 0.0;
        for (int i = 0; i < 12; ++i) m += fv[21 + i];
        m /= 12.0;
        double v = 0.0;
        for (int i = 0; i < 12; ++i) { const double d = fv[21 + i] - m; v = fma(d, d, v); }
        fv[33] = sqrt(v / 12.0);
    }
    __syncthreads();
}

// LDS bytes of the generic kernel
inline size_t generic_lds_bytes(int Nc, int Nf, int F) {
    const size_t nfp = (size_t)((Nf + 1) & ~1);
    return 2 * (size_t)Nc * 16 + 2 * nfp * 8 + (size_t)kFlush * F * 8 + 48 * 8 + 40 * 8;
}

template <typename T>
__global__ __launch_bounds__(64) void st_generic_kernel(PlanDev P, const T *__restrict__ sig,
                                                         const ClipDev *__restrict__ clips,
                                                         const ClipNorm *__restrict__ norms,
                                                         const Tile *__restrict__ tiles,
                                                         double *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    const int Nc = P.Nc, Nf = P.Nf, F = P.F;
    const int nfp = (Nf + 1) & ~1;
    double2 *bufA = reinterpret_cast<double2 *>(smem);
    double2 *bufB = bufA + Nc;
    double *spec0 = reinterpret_cast<double *>(bufB + Nc);
    double *spec1 = spec0 + nfp;
    double *otile = spec1 + nfp;
    double *fv = otile + kFlush * F;
    double *msp = fv + 48;

    const Tile tl = tiles[blockIdx.x];
    const ClipDev c = clips[tl.clip];
    const ClipNorm nm = norms[tl.clip];
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
        frame_spectrum<T>(P, x, nm, bufA, bufB, cur, lane);
        if (P.mode == 1) {            // spectrogram row (ShortTermFeatures.py:422)
            double *row = oc + (long long)t * Nf;
            for (int k = lane; k < Nf; k += kWave) row[k] = cur[k];
        } else if (P.mode == 2) {     // chromagram row (:356-359)
            double p = 0.0;
            for (int k = lane; k < Nf; k += kWave) { const double X = cur[k]; p = fma(X, X, p); }
            p = wave_sum(p);
            const double ch = chroma_class(P, cur, p, lane);
            if (lane < 12) oc[(long long)t * 12 + lane] = ch;
        } else {
            const bool want = (t >= tl.t0) || (P.deltas && t == tl.t0 - 1);
            if (want) {
                frame_features<T>(P, x, nm, cur, (t == 0) ? cur : prv, fv, msp, lane);
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
                __syncthreads();
                // row segments: nslot consecutive frames of feature row f are contiguous in [F][T]
                for (int idx = lane; idx < F * kFlush; idx += kWave) {
                    const int f = idx / kFlush, i = idx % kFlush;
                    if (i < nslot) oc[(long long)f * Tc + tbase + i] = otile[i * F + f];
                }
                __syncthreads();
                tbase += nslot;
                nslot = 0;
            }
        }
        double *tmp = prv; prv = cur; cur = tmp;
    }
}

}  // namespace paa
