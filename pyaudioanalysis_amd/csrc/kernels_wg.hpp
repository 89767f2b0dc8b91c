// Large windows -- beyond what one WAVE can hold in LDS (kernels_mix.hpp ends near 6 000 samples), e.g. the 1 s windows
// audioSegmentation.music_thumbnailing passes by default (audioSegmentation.py:1137: 16 000 samples at 16 kHz): ONE WORKGROUP
// per frame, the whole transform in LDS.
//
//   wg_spectrum_kernel  512 threads, one frame: samples -> normalised packed-complex sequence in LDS (Nc x 16 bytes: up to
//                       10 000 complex points, i.e. windows up to 20 000 samples even / 10 000 odd); zero-crossing count,
//                       energy and the ten entropy-block energies straight from that buffer (every wave a contiguous eighth of
//                       the frame); the in-place decimation-in-frequency passes of kernels_mix.hpp (radix 16 / 8 / 4 / 2 / 13 /
//                       11 / 7 / 5 / 3 butterflies in registers, two per lane in flight, one __syncthreads per pass, twiddles as the
//                       product of two LDS-resident factors W^(128 h) W^l -- no global load inside a pass); real-FFT recombination + |X| / num_fft read through the digit-reversal permutation and written
//                       ONCE to the frame's spectrum row in HBM (for spectrogram plans: straight into the output)
//   wg_feat_kernel      512 threads, one frame: the frame's and the previous frame's spectrum rows are staged in LDS (all loads in
//                       flight at once), then the 34 features with every sweep spread over the workgroup (block energies per block range, mel filters and
//                       chroma classes one wave at a time with all 64 lanes on the filter's bins)
//   wg_delta_kernel     rows 34..67 of every clip
//
// Three launches for ALL frames of ALL clips of a plan (kernels_big.hpp: about ten launches per clip and per chunk, every
// radix pass a round trip through HBM: 1.5 MB of traffic per 16 000-sample frame against the 16 KB + 64 KB this path moves).
// Windows whose transform does not fit the LDS (44 100 samples: 22 050 complex points = 353 KB) keep kernels_big.hpp.
// Replaces ShortTermFeatures.py:608-682 (+ helpers :22-140, :236-321), spectrogram (:415-422), chromagram (:349-359).
#pragma once
#include "kernels_mix.hpp"

namespace paa {
namespace wg {

constexpr int kThreads = 512;           // spectrum kernel: eight waves, two per SIMD (256 registers each: radix-16 / 13 butterflies fit)
constexpr int kWaves = kThreads / 64;
constexpr int kFeatThreads = 512;
constexpr int kFeatWaves = kFeatThreads / 64;
constexpr int kTwLo = 128;              // two-level twiddles: W^m = W^(128 (m >> 7)) W^(m & 127), both factors in LDS

struct WgLayout {
    int n_pass;
    int radix[mix::kMaxPass], span[mix::kMaxPass], tws[mix::kMaxPass];
    unsigned magic[mix::kMaxPass];
    int off_red, off_twlo, off_twhi, off_perm;      // byte offsets into the LDS behind the Nc x 16 transform buffer
    int n_twhi;
    int perm_lds;                   // 1: the digit-reversal permutation fits the LDS beside the buffer
    int lds_bytes;                  // spectrum kernel
    int feat_lds_bytes;             // feature kernel: the frame's and the previous frame's spectrum rows + 1 KB
};
// one frame of the launch: its clip, its index in the clip, the row of the spectrum scratch it writes, and whether it is only
// there to provide the previous spectrum of the next one (a chunk that starts inside a clip)
struct FrameRef {
    int clip, t, row, halo;
};

struct Tw2 {
    const double2 *lo, *hi;
    __device__ __forceinline__ double2 get(int m) const { return cmul(hi[m >> 7], lo[m & (kTwLo - 1)]); }
};

// U butterflies of one in-place DIF pass per lane (kernels_mix.hpp's dif_batch with the twiddles from the two LDS tables):
// butterfly b works on the R elements base + r * stride of its block; output q is multiplied by W_M^(q k) and goes back to
// base + q * stride.  Butterflies touch disjoint elements: no ordering inside a pass.
template <int R, int U>
__device__ __forceinline__ void wg_dif_batch(double2 *buf, int nb, int stride, int M, int tws, unsigned magic, const Tw2 &tw, int b0) {
    double2 v[U][R];
    int base[U], t1[U];
    bool act[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int b = b0 + kThreads * u;
        act[u] = b < nb;
        const int be = act[u] ? b : nb - 1;                     // (lanes past the end shadow a valid butterfly, stores masked)
        const int blk = (stride == 1) ? be : (int)__umulhi((unsigned)be, magic);
        const int k = be - __mul24(blk, stride);
        base[u] = __mul24(blk, M) + k;
        t1[u] = __mul24(k, tws);
#pragma unroll
        for (int r = 0; r < R; ++r) v[u][r] = buf[base[u] + r * stride];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        mix::Bfly<R>::run(v[u]);
        if (stride > 1) {
#pragma unroll
            for (int q0 = 1; q0 < R; q0 += 4) {
                double2 wl[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (q0 + j < R) wl[j] = tw.get((q0 + j) * t1[u]);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (q0 + j < R) v[u][mix::Bfly<R>::pos(q0 + j)] = cmul(v[u][mix::Bfly<R>::pos(q0 + j)], wl[j]);
            }
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if (act[u]) {
#pragma unroll
            for (int q = 0; q < R; ++q) buf[base[u] + q * stride] = v[u][mix::Bfly<R>::pos(q)];
        }
    }
}
template <int R>
__device__ __forceinline__ void wg_dif_pass(double2 *buf, int Nc, int M, int tws, unsigned magic, const Tw2 &tw, int tid) {
    constexpr int U = (R <= 8) ? 2 : 1;          // two butterflies in flight per lane where the registers allow it
    const int stride = M / R, nb = Nc / R;
    // (wave-uniform trip count: a wave whose first butterfly exists runs the batch)
    for (int b0 = tid; (b0 & ~63) < nb; b0 += U * kThreads) wg_dif_batch<R, U>(buf, nb, stride, M, tws, magic, tw, b0);
}

// P.mode decides where the row goes and whether the time-domain features are formed
template <typename T>
__global__ __launch_bounds__(kThreads) void wg_spectrum_kernel(PlanDev P, WgLayout L, const unsigned short *__restrict__ perm_g,
                                                               const T *__restrict__ sig, const ClipDev *__restrict__ clips,
                                                               const ClipNorm *__restrict__ norms,
                                                               const FrameRef *__restrict__ frames, double *__restrict__ spec,
                                                               double *__restrict__ tfeat, double *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double2 *buf = reinterpret_cast<double2 *>(smem);
    double *red = reinterpret_cast<double *>(smem + L.off_red);        // [kWaves][5]
    double2 *twlo = reinterpret_cast<double2 *>(smem + L.off_twlo), *twhi = reinterpret_cast<double2 *>(smem + L.off_twhi);
    unsigned short *perm_l = reinterpret_cast<unsigned short *>(smem + L.off_perm);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const FrameRef fr = frames[blockIdx.x];
    const ClipDev c = clips[fr.clip];
    const ClipNorm nm = norms[fr.clip];
    const T *x = sig + c.sample_off + P.frame_origin + (long long)fr.t * P.S;
    const int W = P.W, Nc = P.Nc, Nf = P.Nf;
    const double sc = sample_scale<T>();
    // ---- tables: the two twiddle factors (and the permutation when it fits)
    if (tid < kTwLo) twlo[tid] = P.tw[tid < Nc ? tid : 0];
    if (tid >= 256 && tid - 256 < L.n_twhi) twhi[tid - 256] = P.tw[(tid - 256) * kTwLo];
    if (L.perm_lds) {
        const unsigned *src = reinterpret_cast<const unsigned *>(perm_g);
        unsigned *dst = reinterpret_cast<unsigned *>(perm_l);
        for (int i = tid; i < (Nc + 1) / 2; i += kThreads) dst[i] = src[i];
    }
    // ---- load: y = (x / 2^15 - mean) / (max|.| + 1e-10) (ShortTermFeatures.py:567-570); even windows packed two samples per point
    if (P.even) {
#pragma unroll 4
        for (int p = tid; p < Nc; p += kThreads) {
            const double2 xx = ct::PairLoad<T>::get(x + 2 * p);
            buf[p] = make_double2(fma(xx.x, sc, -nm.mean) * nm.inv, fma(xx.y, sc, -nm.mean) * nm.inv);
        }
    } else {
#pragma unroll 4
        for (int n = tid; n < W; n += kThreads) buf[n] = make_double2(fma(load_sample<T>(x + n), sc, -nm.mean) * nm.inv, 0.0);
    }
    __syncthreads();
    // ---- time domain (:22-51): wave w owns samples [w per, (w + 1) per); per <= 1.25 blocks, so a wave meets at most three of
    // the ten entropy blocks (or the tail the reference leaves out of them, block "10")
    if (P.mode == 0 && !fr.halo) {
        const int st = P.even ? 1 : 2;
        const double *y = reinterpret_cast<const double *>(buf);
        const int LT = P.blk_t;
        const int per = (W + kWaves - 1) / kWaves;
        const int n0 = wave * per, n1 = min(W, n0 + per);
        const int b0 = min(n0 / LT, 10);
        const int bnd1 = (b0 < 10) ? (b0 + 1) * LT : 0x7fffffff, bnd2 = (b0 + 1 < 10) ? (b0 + 2) * LT : 0x7fffffff;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0;
        int zc = 0;
#pragma unroll 4
        for (int n = n0 + lane; n < n1; n += 64) {
            const double v = y[n * st];
            const double u = y[max(n - 1, 0) * st];          // (n = 0 meets itself: no sign change)
            const double e = v * v;
            a0 += (n < bnd1) ? e : 0.0;
            a1 += (n >= bnd1 && n < bnd2) ? e : 0.0;
            a2 += (n >= bnd2) ? e : 0.0;
            zc += abs(((v > 0.0) - (v < 0.0)) - ((u > 0.0) - (u < 0.0)));
        }
        a0 = wsum(a0); a1 = wsum(a1); a2 = wsum(a2);
        zc = wsum_i(zc);
        if (lane == 0) { red[5 * wave] = a0; red[5 * wave + 1] = a1; red[5 * wave + 2] = a2; red[5 * wave + 3] = (double)zc; red[5 * wave + 4] = (double)b0; }
    }
    __syncthreads();          // every wave has read its samples: the passes may overwrite the buffer
    if (P.mode == 0 && !fr.halo && wave == 0) {
        // block j (lane j < 11) = the parts of every wave that fall into it, added in wave order
        double E = 0.0, zct = 0.0;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) {
            const int bw = (int)red[5 * w + 4];
            E += (bw == lane) ? red[5 * w] : 0.0;
            E += (bw + 1 == lane) ? red[5 * w + 1] : 0.0;
            E += (bw + 2 == lane) ? red[5 * w + 2] : 0.0;
            zct += red[5 * w + 3];
        }
        const double e_tot = wsum((lane < 11) ? E : 0.0);
        const double s = fast_div(E, e_tot + kEps);
        const double ent = wsum((lane < 10) ? -(s * fast_log2(s + kEps)) : 0.0);
        if (lane == 0) {
            double *tfp = tfeat + 3 * (long long)fr.row;
            tfp[0] = e_tot; tfp[1] = ent; tfp[2] = zct;
        }
    }
    // ---- in-place DIF passes (kernels_mix.hpp's butterflies; two waves per SIMD and two butterflies per lane hide the latency)
    const Tw2 tw = {twlo, twhi};
    for (int p = 0; p < L.n_pass; ++p) {
        const int M = L.span[p], ts = L.tws[p];
        const unsigned mg = L.magic[p];
        switch (L.radix[p]) {
            case 2: wg_dif_pass<2>(buf, Nc, M, ts, mg, tw, tid); break;
            case 3: wg_dif_pass<3>(buf, Nc, M, ts, mg, tw, tid); break;
            case 4: wg_dif_pass<4>(buf, Nc, M, ts, mg, tw, tid); break;
            case 5: wg_dif_pass<5>(buf, Nc, M, ts, mg, tw, tid); break;
            case 7: wg_dif_pass<7>(buf, Nc, M, ts, mg, tw, tid); break;
            case 8: wg_dif_pass<8>(buf, Nc, M, ts, mg, tw, tid); break;
            case 11: wg_dif_pass<11>(buf, Nc, M, ts, mg, tw, tid); break;
            case 13: wg_dif_pass<13>(buf, Nc, M, ts, mg, tw, tid); break;
            default: wg_dif_pass<16>(buf, Nc, M, ts, mg, tw, tid); break;
        }
        __syncthreads();
    }
    // ---- |X| / num_fft (:617-621) through the digit-reversal permutation, written once to the frame's row
    double *row = (P.mode == 1) ? out + c.out_off + (long long)fr.t * Nf : spec + (long long)fr.row * Nf;
    const double invNf = 1.0 / (double)Nf;
    const unsigned short *perm = L.perm_lds ? perm_l : perm_g;
    if (P.even) {
        // bins k and Nc - k from one pair: X[k] = E + w^k O, X[Nc - k] = conj(E - w^k O), E = (Z[k] + conj Z[Nc - k]) / 2,
        // O = -i (Z[k] - conj Z[Nc - k]) / 2; four pairs in flight per lane
        const int npairs = Nc / 2 + 1;
        for (int k0 = tid; (k0 & ~63) < npairs; k0 += 4 * kThreads) {
            int kk[4], pa[4], pb[4];
            double2 pw[4], zk[4], zm[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                kk[u] = min(k0 + u * kThreads, npairs - 1);
                pw[u] = P.post[kk[u]];
                pa[u] = L.perm_lds ? perm_l[kk[u]] : perm_g[kk[u]];
                pb[u] = L.perm_lds ? perm_l[kk[u] == 0 ? 0 : Nc - kk[u]] : perm_g[kk[u] == 0 ? 0 : Nc - kk[u]];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) { zk[u] = buf[pa[u]]; zm[u] = buf[pb[u]]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = kk[u];
                const double2 e = make_double2(0.5 * (zk[u].x + zm[u].x), 0.5 * (zk[u].y - zm[u].y));
                const double2 o = make_double2(0.5 * (zk[u].y + zm[u].y), 0.5 * (zm[u].x - zk[u].x));
                const double2 wo = cmul(pw[u], o);
                const double ar = e.x + wo.x, ai = e.y + wo.y, br = e.x - wo.x, bi = e.y - wo.y;
                if (k0 + u * kThreads < npairs) {
                    row[k] = mag_sqrt(fma(ar, ar, ai * ai)) * invNf;
                    if (k > 0 && Nc - k != k) row[Nc - k] = mag_sqrt(fma(br, br, bi * bi)) * invNf;
                }
            }
        }
    } else {
#pragma unroll 4
        for (int k = tid; k < Nf; k += kThreads) {
            const double2 z = buf[perm[k]];
            row[k] = mag_sqrt(fma(z.x, z.x, z.y * z.y)) * invNf;
        }
    }
}

// sum over the workgroup: every wave's total through LDS (slot[kFeatWaves]); all threads return the same bits
__device__ __forceinline__ double bsum8(double v, double *slot, int lane, int wave) {
    v = wsum(v);
    if (lane == 0) slot[wave] = v;
    __syncthreads();
    const double r = ((slot[0] + slot[1]) + (slot[2] + slot[3])) + ((slot[4] + slot[5]) + (slot[6] + slot[7]));
    __syncthreads();
    return r;
}

// the 34 features (or the 12 chromagram values) of one frame from its spectrum row and the previous frame's: both rows are
// staged in LDS first (16 independent loads per lane and row), every sweep then runs from there
__global__ __launch_bounds__(kFeatThreads) void wg_feat_kernel(PlanDev P, const FrameRef *__restrict__ frames,
                                                              const ClipDev *__restrict__ clips, const double *__restrict__ spec,
                                                              const double *__restrict__ tfeat, double *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const FrameRef fr = frames[blockIdx.x];
    if (fr.halo) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const ClipDev c = clips[fr.clip];
    const int Nf = P.Nf, W = P.W;
    double *cur = reinterpret_cast<double *>(smem);
    double *prv = cur + Nf;
    double *fv = prv + Nf;               // [48]
    double *msp = fv + 48;               // [40]
    double *red = msp + 40;              // [kFeatWaves][16]
    double *slot = red + kFeatWaves * 16;      // [kFeatWaves]
    int *redi = reinterpret_cast<int *>(slot + kFeatWaves);
    {
        const double *gc = spec + (long long)fr.row * Nf;
        const double *gp = (fr.t == 0) ? gc : gc - Nf;      // frames are laid out in clip order: the previous frame is the previous row
#pragma unroll 8
        for (int k = tid; k < Nf; k += kFeatThreads) cur[k] = gc[k];
        if (P.mode == 0) {
#pragma unroll 8
            for (int k = tid; k < Nf; k += kFeatThreads) prv[k] = gp[k];
        }
    }
    __syncthreads();
    double *oc = out + c.out_off;
    const long long Tc = c.T;
    const Tabs tb = tabs_global(P);
    // ---- sweep A (:57-107): sums, maximum, the ten block energies + the tail, block range by block range
    const int LB = P.blk_f;
    double part[15];          // 0..9 blocks, 10 tail, 11 sum X, 12 sum X_prev, 13 sum (k + 1) X, 14 max
#pragma unroll
    for (int i = 0; i < 15; ++i) part[i] = 0.0;
#pragma unroll
    for (int j = 0; j < 11; ++j) {
        const int lo = j * LB, hi = (j < 10) ? lo + LB : Nf;
        double p = 0.0;
        for (int k = lo + tid; k < hi; k += kFeatThreads) {
            const double X = cur[k];
            part[11] += X;
            part[12] += prv[k];
            part[13] = fma((double)(k + 1), X, part[13]);
            part[14] = fmax(part[14], X);
            p = fma(X, X, p);
        }
        part[j] = p;
    }
#pragma unroll
    for (int i = 0; i < 14; ++i) part[i] = wsum(part[i]);
    part[14] = wmax_nonneg(part[14]);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 15; ++i) red[16 * wave + i] = part[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 14; ++i) {
        double a = 0.0;
#pragma unroll
        for (int w = 0; w < kFeatWaves; ++w) a += red[16 * w + i];
        part[i] = a;
    }
    {
        double m = 0.0;
#pragma unroll
        for (int w = 0; w < kFeatWaves; ++w) m = fmax(m, red[16 * w + 14]);
        part[14] = m;
    }
    __syncthreads();
    double sP = part[10];
#pragma unroll
    for (int j = 0; j < 10; ++j) sP += part[j];
    if (P.mode == 2) {
        // ---- chromagram row (:356-359): pitch class by pitch class, one wave per class, all lanes on the class's bins
        for (int cls = wave; cls < 12; cls += kFeatWaves) {
            const int b = tb.ch_start[cls], e = tb.ch_start[cls + 1];
            double acc = 0.0;
            for (int i = b + lane; i < e; i += 64) { const double xv = cur[tb.ch_src[i]]; acc = fma(xv * xv, tb.ch_w[i], acc); }
            acc = wsum(acc);
            if (lane == 0) oc[(long long)fr.t * 12 + cls] = (sP == 0.0) ? acc / kEps : fast_div(acc, sP);
        }
        return;
    }
    const double f0 = P.fs / (2.0 * (double)Nf);
    const double sX = part[11], mx = part[14];
    const double sIX = part[13] * f0;
    const double sXe = sX + (double)Nf * kEps;                // np.sum(X + eps) (:118-119)
    const double sXp = part[12] + (double)Nf * kEps;
    // spectral entropy (:85-107)
    double ent_f = 0.0;
#pragma unroll
    for (int j = 0; j < 10; ++j) {
        const double s = fast_div(part[j], sP + kEps);
        ent_f -= s * fast_log2(s + kEps);
    }
    // ---- sweep B: spread and flux (:57-82, :110-124)
    const double r = (mx == 0.0) ? 1.0 / kEps : fast_div(1.0, mx);
    const double den = sX * r + kEps;
    const double cen = fast_div(sIX * r, den);
    const double rX = fast_div(1.0, sXe), rXp = fast_div(1.0, sXp);
    double sSp = 0.0, sFl = 0.0;
#pragma unroll 4
    for (int k = tid; k < Nf; k += kFeatThreads) {
        const double X = cur[k];
        const double dv = (double)(k + 1) * f0 - cen;
        sSp = fma(dv * dv, X * r, sSp);
        const double df = X * rX - prv[k] * rXp;
        sFl = fma(df, df, sFl);
    }
    sSp = bsum8(sSp, slot, lane, wave);
    sFl = bsum8(sFl, slot, lane, wave);
    const double spread = fast_sqrt(fast_div(sSp, den));
    // ---- roll-off (:127-140): first k with cumsum(X^2)[k] + eps > 0.9 sum(X^2); contiguous chunks of ODD length (a lane's
    // chunk starts an odd number of doubles after its neighbour's: no LDS bank is hit twice), workgroup-wide scan
    int first = 0x7fffffff;
    {
        const double thr = 0.90 * sP;
        const int cch = ((Nf + kFeatThreads - 1) / kFeatThreads) | 1;
        const int kb = min(tid * cch, Nf), ke = min(Nf, kb + cch);
        double cs = 0.0;
        for (int k = kb; k < ke; ++k) { const double X = cur[k]; cs = fma(X, X, cs); }
        const double incl = wscan_incl(cs);
        if (lane == 63) slot[wave] = incl;
        __syncthreads();
        double run = incl - cs;
#pragma unroll
        for (int w = 0; w < kFeatWaves; ++w) run += (w < wave) ? slot[w] : 0.0;
        for (int k = kb; k < ke; ++k) {
            const double X = cur[k];
            run = fma(X, X, run);
            if (run + kEps > thr) { first = k; break; }
        }
        first = wmin_i(first);
        if (lane == 0) redi[wave] = first;
        __syncthreads();
        first = redi[0];
#pragma unroll
        for (int w = 1; w < kFeatWaves; ++w) first = min(first, redi[w]);
    }
    // ---- MFCC (:236-254): filter by filter, one wave per filter, all lanes on the filter's bins
    for (int m = wave; m < 40; m += kFeatWaves) {
        const int lo = tb.mel_lo[m], cnt = tb.mel_cnt[m];
        const double *wv = tb.mel_w + tb.mel_off[m];
        double a = 0.0;
#pragma unroll 4
        for (int i = lane; i < cnt; i += 64) a = fma(cur[lo + i], wv[i], a);
        a = wsum(a);
        if (lane == 0) msp[m] = fast_log10(a + kEps);
    }
    // ---- chroma (:277-321)
    for (int cls = wave; cls < 12; cls += kFeatWaves) {
        const int b = tb.ch_start[cls], e = tb.ch_start[cls + 1];
        double acc = 0.0;
#pragma unroll 4
        for (int i = b + lane; i < e; i += 64) { const double xv = cur[tb.ch_src[i]]; acc = fma(xv * xv, tb.ch_w[i], acc); }
        acc = wsum(acc);
        if (lane == 0) fv[21 + cls] = (sP == 0.0) ? acc / kEps : fast_div(acc, sP);
    }
    __syncthreads();
    if (tid < 13) {
        const double *m = tb.dct + tid * tb.dct_stride;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
        for (int n = 0; n < 40; n += 4) {
            a0 = fma(m[n], msp[n], a0);
            a1 = fma(m[n + 1], msp[n + 1], a1);
            a2 = fma(m[n + 2], msp[n + 2], a2);
            a3 = fma(m[n + 3], msp[n + 3], a3);
        }
        fv[8 + tid] = (a0 + a1) + (a2 + a3);
    }
    if (tid == 64) {
        const double *tfp = tfeat + 3 * (long long)fr.row;
        fv[0] = (tfp[2] / 2.0) / (double)(W - 1);
        fv[1] = tfp[0] / (double)W;
        fv[2] = tfp[1];
        fv[3] = cen / (P.fs / 2.0);
        fv[4] = spread / (P.fs / 2.0);
        fv[5] = ent_f;
        fv[6] = (fr.t == 0) ? 0.0 : sFl;       // first frame: previous spectrum = itself (:624-625)
        fv[7] = (first == 0x7fffffff) ? 0.0 : (double)first / (double)Nf;
        double mch = 0.0;                      // population std of the 12 chroma values (:667)
        for (int i = 0; i < 12; ++i) mch += fv[21 + i];
        mch /= 12.0;
        double v = 0.0;
        for (int i = 0; i < 12; ++i) { const double d = fv[21 + i] - mch; v = fma(d, d, v); }
        fv[33] = fast_sqrt(v / 12.0);
    }
    __syncthreads();
    if (tid < kBase) oc[(long long)tid * Tc + fr.t] = fv[tid];
}

// rows 34..67 of every clip (ShortTermFeatures.py:668-680): blockIdx.y = clip
__global__ __launch_bounds__(256) void wg_delta_kernel(const ClipDev *__restrict__ clips, double *__restrict__ out) {
    const ClipDev c = clips[blockIdx.y];
    const long long Tc = c.T, idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)kBase * Tc) return;
    double *oc = out + c.out_off;
    const long long rowi = idx / Tc, t = idx % Tc;
    const double *rp = oc + rowi * Tc;
    oc[(kBase + rowi) * Tc + t] = (t == 0) ? 0.0 : rp[t] - rp[t - 1];
}

// ---- host: does the window fit, radix schedule, permutation ----------------------------------------------------------------
// 1: the transform of this window runs in one workgroup's LDS (fills L and perm); 0: it does not (kernels_big.hpp keeps it)
inline int wg_layout(const FftPlan &fft, WgLayout &L, std::vector<unsigned short> &perm) {
    const int Nc = fft.len, Nf = fft.window / 2;
    if (Nc < 256 || Nc > 65535) return 0;
    std::vector<int> radix;
    if (!mix::mix_factor(Nc, radix)) return 0;
    memset(&L, 0, sizeof(L));
    auto up16 = [](size_t b) { return (b + 15) / 16 * 16; };
    size_t off = up16((size_t)Nc * 16);
    L.off_red = (int)off; off += up16((size_t)kWaves * 5 * 8);
    L.off_twlo = (int)off; off += (size_t)kTwLo * 16;
    L.n_twhi = (Nc + kTwLo - 1) / kTwLo;
    if (L.n_twhi > kThreads - 256) return 0;
    L.off_twhi = (int)off; off += (size_t)L.n_twhi * 16;
    L.off_perm = (int)off;
    L.perm_lds = (off + up16((size_t)Nc * 2 + 4) <= 160 * 1024) ? 1 : 0;
    if (L.perm_lds) off += up16((size_t)Nc * 2 + 4);
    if (off > 160 * 1024) return 0;
    L.lds_bytes = (int)off;
    const size_t feat = up16((size_t)Nf * 16 + (48 + 40 + kFeatWaves * 16 + kFeatWaves) * 8 + kFeatWaves * 4);
    if (feat > 160 * 1024) return 0;
    L.feat_lds_bytes = (int)feat;
    L.n_pass = (int)radix.size();
    int M = Nc;
    for (int p = 0; p < L.n_pass; ++p) {
        L.radix[p] = radix[p];
        L.span[p] = M;
        const unsigned stride = (unsigned)(M / radix[p]);
        L.magic[p] = stride > 1 ? (unsigned)((1ULL << 32) / stride) + 1u : 0u;
        L.tws[p] = Nc / M;
        M /= radix[p];
    }
    mix::mix_permutation(Nc, radix, perm);
    perm.push_back(0);          // (the LDS copy moves whole 32-bit words)
    return 1;
}

}  // namespace wg
}  // namespace paa
