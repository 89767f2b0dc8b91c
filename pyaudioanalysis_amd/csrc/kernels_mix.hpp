// Mixed-radix feature kernel for every window whose FFT length factors into {2, 3, 5, 7, 11, 13} and that no
// register-FFT kernel covers (50 ms at 44.1 / 48 kHz = 2205 / 2400 samples, 40 ms at 44.1 / 48 kHz = 1764 / 1920, 1024, ...):
// any step, int16 / interleaved stereo int16 / float64 samples, features / spectrogram / chromagram.
//
// One wave owns a run of consecutive frames, one frame at a time, like the generic kernel -- but the FFT is an IN-PLACE
// decimation-in-frequency transform (radix 16/8/4/2/13/11/7/5/3 butterflies held in registers, digit-reversed output read
// back through a host-built permutation), so a wave needs ONE complex buffer instead of the Stockham ping-pong pair, and
// the spectrum is written over the dead FFT buffer:
//
//     LDS per wave = Nc complex (FFT / current spectrum) + Nf doubles (previous spectrum) + the feature staging,
//
// 34 KB instead of 62 KB at window 2400: four waves per CU instead of one.  The two spectra never move: even frames of
// a run transform in [0, B) and leave their spectrum at the front, odd frames transform in [U, U + B) and leave it at
// the back, so the previous spectrum always sits in the third the current transform does not touch (B = 16 Nc bytes,
// U = 8 Nf bytes rounded to 16).
//
// A wave runs alone on its SIMD at the large windows (the LDS footprint, not the registers, sets the occupancy), so every
// stage hides latency through instruction-level parallelism: butterflies, magnitudes, sample loads, the mel / chroma
// gathers and both feature sweeps fetch in groups before they compute; the time-domain and spectral stages work on an
// even split of the frame over the lanes (contiguous chunks, two block partials per lane, no bounds masks in the main
// loops); the next frame's new samples are touched before the feature stage.  Windows whose LDS footprint allows more
// than four waves (up to ~1 700 samples) run a LEAN instance: <= 256 registers, radix <= 8, six to eight waves per CU.
// The butterflies are written so that R equal inputs give EXACT zeros in the non-DC outputs (dft5 in device_common.hpp,
// dft_prime below): the spectrum of a digitally silent frame is exact, as the reference's is for these lengths.
//
// Replaces the while loop at ShortTermFeatures.py:608-682 (+ helpers :22-140, :236-321) and the loops of spectrogram
// (:415-422) / chromagram (:349-359) for those windows; mel / DCT / chroma semantics are the generic kernel's
// (kernels_generic.hpp: Tabs, chroma_class).
#pragma once
#include <vector>

#include "kernels_ct.hpp"
#include "kernels_generic.hpp"

namespace paa {
namespace mix {
#if defined(PAA_F800_TIMING) || defined(PAA_F800_TRACE)
using f800::g_phase_cycles;        // per-phase cycle accounting of diagnostic builds (PAA_T0 / PAA_TICK / PAA_TEND)
using f800::g_wave_trace;
#endif

constexpr int kMaxPass = 12;
constexpr int kSlots = 24;          // magnitude pass: result pairs a lane keeps until every lane has read its inputs
constexpr int kSlotsLean = 12;      // ... in the lean variant (<= 256 registers: two waves per SIMD where the LDS allows it)

struct MixLayout {
    int off_tw, off_post, off_perm, off_mello, off_melcnt, off_meloff, off_melw, off_dct, off_chstart, off_chsrc, off_chw;
    int table_bytes;     // multiple of 16
    int wave_bytes;      // per-wave region, multiple of 16
    int waves;           // waves per workgroup
    int tw_global;       // 1: twiddles and post-twiddles stay in global memory (L1/L2 hits), 0: LDS copies
    int pad_shift;       // FFT buffer skew: element e lives at e + (e >> pad_shift) (5 for lengths that are multiples of 32:
                         // power-of-two strides would put a butterfly's operands on one bank; 31 = no skew)
    int lean;            // 1: the <= 256-register instance with up to eight waves per workgroup (windows whose LDS allows > 4)
    int unit_bytes;      // U: one spectrum, rounded to 16 bytes
    int buf_bytes;       // B: Nc complex
    int n_pass;
    int radix[kMaxPass];
    int span[kMaxPass];            // M of the pass: the block a butterfly lives in (M / radix = element stride)
    int tws[kMaxPass];             // Nc / M: twiddle W_M^j = tw[j * tws]
    unsigned magic[kMaxPass];      // floor(2^32 / stride) + 1: b / stride == umulhi(b, magic) for b < 2^16
};

__device__ __forceinline__ int skew(int e, int sh) { return sh < 0 ? e : e + (e >> sh); }      // (sh is a compile-time constant)

// ---- prime butterflies: X[q], X[R-q] from the sums / differences of the pairs (x_j, x_{R-j}) -------------------
template <int R> struct PrimeTab;
template <> struct PrimeTab<7> {
    static constexpr double c[7] = {1.00000000000000000000, 0.62348980185873359439, -0.22252093395631433737, -0.90096886790241903498, -0.90096886790241914600, -0.22252093395631458717, 0.62348980185873337234};
    static constexpr double s[7] = {0.00000000000000000000, 0.78183148246802980363, 0.97492791218182361934, 0.43388373911755823142, -0.43388373911755800938, -0.97492791218182361934, -0.78183148246802991466};
};
template <> struct PrimeTab<11> {
    static constexpr double c[11] = {1.00000000000000000000, 0.84125353283118120551, 0.41541501300188643508, -0.14231483827328500480, -0.65486073394528498959, -0.95949297361449736865, -0.95949297361449747967, -0.65486073394528521163, -0.14231483827328522684, 0.41541501300188604651, 0.84125353283118120551};
    static constexpr double s[11] = {0.00000000000000000000, 0.54064081745559755543, 0.90963199535451833011, 0.98982144188093279524, 0.75574957435425826890, 0.28173255684142967104, -0.28173255684142939348, -0.75574957435425815788, -0.98982144188093268422, -0.90963199535451855215, -0.54064081745559744441};
};
template <> struct PrimeTab<13> {
    static constexpr double c[13] = {1.00000000000000000000, 0.88545602565320991051, 0.56806474673115592289, 0.12053668025532300601, -0.35460488704253545489, -0.74851074817110119231, -0.97094181742605201180, -0.97094181742605212282, -0.74851074817110130333, -0.35460488704253589898, 0.12053668025532320029, 0.56806474673115481266, 0.88545602565321002153};
    static constexpr double s[13] = {0.00000000000000000000, 0.46472317204376850652, 0.82298386589365635224, 0.99270887409805397272, 0.93501624268541483342, 0.66312265824079519305, 0.23931566428755768339, -0.23931566428755743359, -0.66312265824079497101, -0.93501624268541472240, -0.99270887409805397272, -0.82298386589365701838, -0.46472317204376839550};
};

// Written so that R EQUAL inputs give exactly zero in every non-DC output (digital silence: see dft5 in
// device_common.hpp): sum_j cos(2 pi j q / R) over j = 1 .. (R-1)/2 is -1/2 for every q, so the coefficient of the pair
// the index j*(q) with j* q = H (mod R) -- any fixed pair p would do -- is replaced by -1/2 minus the others:
//     A_q = (x0 - s_p / 2) + sum_{j != p} cos(2 pi j q / R) (s_j - s_p),     s_j = x_j + x_{R-j}.
// The differences s_j - s_p are formed once; the B sums (sines of x_j - x_{R-j}) are zero for equal inputs anyway.
template <int R>
__device__ __forceinline__ void dft_prime(double2 *v) {
    constexpr int H = (R - 1) / 2;
    double2 sm[H], df[H];
#pragma unroll
    for (int j = 1; j <= H; ++j) {
        sm[j - 1] = cadd(v[j], v[R - j]);
        df[j - 1] = csub(v[j], v[R - j]);
    }
    const double2 x0 = v[0];
    double2 tot = x0;
#pragma unroll
    for (int j = 0; j < H; ++j) tot = cadd(tot, sm[j]);
    // pivot pair p = H: base = x0 - s_H / 2, rel[j] = s_j - s_H
    const double2 base = make_double2(fma(-0.5, sm[H - 1].x, x0.x), fma(-0.5, sm[H - 1].y, x0.y));
    double2 rel[H > 1 ? H - 1 : 1];
#pragma unroll
    for (int j = 0; j + 1 < H; ++j) rel[j] = csub(sm[j], sm[H - 1]);
#pragma unroll
    for (int q = 1; q <= H; ++q) {
        double ar = base.x, ai = base.y, br = 0.0, bi = 0.0;
#pragma unroll
        for (int j = 1; j <= H; ++j) {
            const double c = PrimeTab<R>::c[(j * q) % R], s = PrimeTab<R>::s[(j * q) % R];
            if (j < H) {
                ar = fma(c, rel[j - 1].x, ar);
                ai = fma(c, rel[j - 1].y, ai);
            }
            br = fma(s, df[j - 1].x, br);
            bi = fma(s, df[j - 1].y, bi);
        }
        // X[q] = A - i B, X[R-q] = A + i B  with A = x0 + sum cos (x_j + x_{R-j}), B = sum sin (x_j - x_{R-j})
        v[q] = make_double2(ar + bi, ai - br);
        v[R - q] = make_double2(ar - bi, ai + br);
    }
    v[0] = tot;
}

// radix 8 in natural order: two radix-4 halves, W8 twiddles on the odd half
__device__ __forceinline__ void dft8_natural(double2 *v) {
    const double h = 0.70710678118654752440;
    double2 a[4] = {v[0], v[2], v[4], v[6]}, b[4] = {v[1], v[3], v[5], v[7]};
    dft4(a);
    dft4(b);
    b[1] = make_double2(h * (b[1].x + b[1].y), h * (b[1].y - b[1].x));      // W8   = (h, -h)
    b[2] = make_double2(b[2].y, -b[2].x);                                   // W8^2 = -i
    b[3] = make_double2(h * (b[3].y - b[3].x), -h * (b[3].x + b[3].y));     // W8^3 = (-h, -h)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        v[k] = cadd(a[k], b[k]);
        v[k + 4] = csub(a[k], b[k]);
    }
}

// run() transforms v[] in place; X[q] ends at v[pos(q)]
template <int R> struct Bfly {
    static __device__ __forceinline__ void run(double2 *v) { dft_prime<R>(v); }
    static constexpr int pos(int q) { return q; }
};
template <> struct Bfly<2> { static __device__ __forceinline__ void run(double2 *v) { dft2(v); } static constexpr int pos(int q) { return q; } };
template <> struct Bfly<3> { static __device__ __forceinline__ void run(double2 *v) { dft3(v); } static constexpr int pos(int q) { return q; } };
template <> struct Bfly<4> { static __device__ __forceinline__ void run(double2 *v) { dft4(v); } static constexpr int pos(int q) { return q; } };
template <> struct Bfly<5> { static __device__ __forceinline__ void run(double2 *v) { dft5(v); } static constexpr int pos(int q) { return q; } };
template <> struct Bfly<8> { static __device__ __forceinline__ void run(double2 *v) { dft8_natural(v); } static constexpr int pos(int q) { return q; } };
template <> struct Bfly<16> {          // the 4 x 4 codelet of the 800-sample kernel: X[q] at v[4 (q % 4) + q / 4]
    static __device__ __forceinline__ void run(double2 *v) { f800::dft16<0>(v); }
    static constexpr int pos(int q) { return 4 * (q % 4) + q / 4; }
};

// ---- one in-place DIF pass: butterfly b works on the R elements base + r * stride of its block; output q is multiplied
// by W_M^(q k) and goes back to base + q * stride.  Butterflies touch disjoint elements: no ordering inside a pass.
// A wave runs alone on its SIMD (the LDS footprint decides the occupancy), so latency is hidden by instruction-level
// parallelism only: a lane fetches the elements and twiddles of U butterflies before it computes any of them.
template <int R, int U>
__device__ __forceinline__ void dif_batch(double2 *buf, int nb, int stride, int M, int tws, unsigned magic,
                                          const double2 *__restrict__ tw, int b0, int psh) {
    // radix <= 8: the twiddles are requested together with the operands; the radix-11 / 13 / 16 butterflies keep their
    // R operands and the codelet's temporaries live, so their twiddles come AFTER the codelet, eight at a time (all of them
    // up front spilled 19 registers in the instance that reads them from global memory)
    constexpr bool LATE = R > 8;
    double2 v[U][R], w[U][LATE ? 1 : R];
    int base[U], t1[U];
    bool act[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int b = b0 + kWave * u;
        act[u] = b < nb;
        const int be = act[u] ? b : nb - 1;                     // (inactive lanes shadow a valid butterfly, stores masked)
        const int blk = (stride == 1) ? be : (int)__umulhi((unsigned)be, magic);
        const int k = be - __mul24(blk, stride);                  // (all indices < 2^16: 24-bit multiplies are full rate)
        base[u] = __mul24(blk, M) + k;
        t1[u] = __mul24(k, tws);
#pragma unroll
        for (int r = 0; r < R; ++r) v[u][r] = buf[skew(base[u] + r * stride, psh)];
        if (!LATE && stride > 1) {
#pragma unroll
            for (int q = 1; q < R; ++q) w[u][LATE ? 0 : q] = tw[q * t1[u]];
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        Bfly<R>::run(v[u]);
        if (stride > 1) {
            if (LATE) {
#pragma unroll
                for (int q0 = 1; q0 < R; q0 += 8) {
                    double2 wl[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (q0 + j < R) wl[j] = tw[(q0 + j) * t1[u]];
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (q0 + j < R) v[u][Bfly<R>::pos(q0 + j)] = cmul(v[u][Bfly<R>::pos(q0 + j)], wl[j]);
                }
            } else {
#pragma unroll
                for (int q = 1; q < R; ++q) v[u][Bfly<R>::pos(q)] = cmul(v[u][Bfly<R>::pos(q)], w[u][LATE ? 0 : q]);
            }
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if (act[u]) {
#pragma unroll
            for (int q = 0; q < R; ++q) buf[skew(base[u] + q * stride, psh)] = v[u][Bfly<R>::pos(q)];
        }
    }
}
template <int R, int LEAN>
__device__ __forceinline__ void dif_pass(double2 *buf, int Nc, int M, int tws, unsigned magic,
                                         const double2 *__restrict__ tw, int lane, int psh) {
    // butterflies in flight per lane: as many as the register budget of the instance allows
    constexpr int U = LEAN ? ((R <= 5) ? 2 : 1) : ((R <= 5) ? 4 : (R <= 13 ? 2 : 1));
    const int stride = M / R, nb = Nc / R;
    const int iters = (nb + kWave - 1) / kWave;
    int i = 0;
    for (; i + U <= iters; i += U) dif_batch<R, U>(buf, nb, stride, M, tws, magic, tw, lane + kWave * i, psh);
    if (U == 4 && i + 2 <= iters) { dif_batch<R, 2>(buf, nb, stride, M, tws, magic, tw, lane + kWave * i, psh); i += 2; }
    if (i < iters) dif_batch<R, 1>(buf, nb, stride, M, tws, magic, tw, lane + kWave * i, psh);
}

// ---- the whole transform + |X| / num_fft (ShortTermFeatures.py:617-621) ------------------------------------------
// buf: the frame as packed complex (even windows: z[m] = y[2m] + i y[2m+1]; odd: z[n] = y[n]); cur: where the Nf
// magnitudes go -- it may overlap buf: every lane holds its results in registers until all inputs have been read.
template <int LEAN>
__device__ __forceinline__ void fft_passes_inplace(const PlanDev &P, const MixLayout &L, double2 *buf,
                                                   const double2 *__restrict__ tw, int lane) {
    constexpr int PSH = (LEAN == 2) ? 5 : -1;       // instance 2: lean + skewed buffer (lengths that are multiples of 32)
    const int Nc = P.Nc;
    for (int p = 0; p < L.n_pass; ++p) {
        const int M = L.span[p];
        const unsigned mg = L.magic[p];
        const int ts = L.tws[p];
        switch (L.radix[p]) {
            case 2: dif_pass<2, LEAN>(buf, Nc, M, ts, mg, tw, lane, PSH); break;
            case 3: dif_pass<3, LEAN>(buf, Nc, M, ts, mg, tw, lane, PSH); break;
            case 4: dif_pass<4, LEAN>(buf, Nc, M, ts, mg, tw, lane, PSH); break;
            case 5: dif_pass<5, LEAN>(buf, Nc, M, ts, mg, tw, lane, PSH); break;
            case 7: dif_pass<7, LEAN>(buf, Nc, M, ts, mg, tw, lane, PSH); break;
            case 8: dif_pass<8, LEAN>(buf, Nc, M, ts, mg, tw, lane, PSH); break;
            case 11: if constexpr (!LEAN) dif_pass<11, LEAN>(buf, Nc, M, ts, mg, tw, lane, PSH); break;     // (lean: radix <= 8)
            case 13: if constexpr (!LEAN) dif_pass<13, LEAN>(buf, Nc, M, ts, mg, tw, lane, PSH); break;
            default:                 // radix 16 (the lean instance's schedules stop at radix 8)
                if constexpr (!LEAN) dif_pass<16, LEAN>(buf, Nc, M, ts, mg, tw, lane, PSH);
                break;
        }
        wsync();
    }
}
template <int LEAN>
__device__ __forceinline__ void magnitudes_inplace(const PlanDev &P, const double2 *buf, const double2 *__restrict__ post,
                                                   const unsigned short *__restrict__ perm, double *cur, int lane) {
    constexpr int kSlots = LEAN ? kSlotsLean : mix::kSlots;
    const int Nc = P.Nc, Nf = P.Nf;
    const double invNf = 1.0 / (double)Nf;     // X / len(X)  (:621)
    double r0[kSlots], r1[kSlots];
    constexpr int G = 4;                       // slots fetched together (a lone wave hides latency through ILP only)
    if (P.even) {
        // bins k and Nc - k come from the same pair: X[k] = E + w^k O, X[Nc-k] = conj(E - w^k O) with
        // E = (Z[k] + conj Z[Nc-k]) / 2, O = -i (Z[k] - conj Z[Nc-k]) / 2
        const int npairs = Nc / 2 + 1;
#pragma unroll
        for (int j0 = 0; j0 < kSlots; j0 += G) {
            if (kWave * j0 < npairs) {
                double2 zk[G], zm[G], pw[G];
#pragma unroll
                for (int u = 0; u < G; ++u) {
                    const int k = min(lane + kWave * (j0 + u), npairs - 1);
                    zk[u] = buf[perm[k]];
                    zm[u] = buf[perm[k == 0 ? 0 : Nc - k]];
                    pw[u] = post[k];
                }
#pragma unroll
                for (int u = 0; u < G; ++u) {
                    const double2 e = make_double2(0.5 * (zk[u].x + zm[u].x), 0.5 * (zk[u].y - zm[u].y));
                    const double2 o = make_double2(0.5 * (zk[u].y + zm[u].y), 0.5 * (zm[u].x - zk[u].x));
                    const double2 wo = cmul(pw[u], o);
                    const double ar = e.x + wo.x, ai = e.y + wo.y, br = e.x - wo.x, bi = e.y - wo.y;
                    r0[j0 + u] = mag_sqrt(fma(ar, ar, ai * ai)) * invNf;
                    r1[j0 + u] = mag_sqrt(fma(br, br, bi * bi)) * invNf;
                }
            } else {
#pragma unroll
                for (int u = 0; u < G; ++u) { r0[j0 + u] = 0.0; r1[j0 + u] = 0.0; }
            }
        }
        wsync();
#pragma unroll
        for (int j = 0; j < kSlots; ++j) {
            if (kWave * j < npairs) {
                const int k = lane + kWave * j;
                if (k < npairs) {
                    cur[k] = r0[j];
                    if (k > 0 && Nc - k != k) cur[Nc - k] = r1[j];
                }
            }
        }
    } else {
#pragma unroll
        for (int j0 = 0; j0 < kSlots; j0 += G) {
            if (kWave * j0 < Nf) {
                double2 za[G], zb[G];
#pragma unroll
                for (int u = 0; u < G; ++u) {
                    const int k = lane + kWave * (j0 + u);
                    za[u] = buf[perm[min(k, Nf - 1)]];
                    zb[u] = buf[perm[min(k + kWave * kSlots, Nf - 1)]];
                }
#pragma unroll
                for (int u = 0; u < G; ++u) {
                    r0[j0 + u] = mag_sqrt(fma(za[u].x, za[u].x, za[u].y * za[u].y)) * invNf;
                    r1[j0 + u] = mag_sqrt(fma(zb[u].x, zb[u].x, zb[u].y * zb[u].y)) * invNf;
                }
            } else {
#pragma unroll
                for (int u = 0; u < G; ++u) { r0[j0 + u] = 0.0; r1[j0 + u] = 0.0; }
            }
        }
        wsync();
#pragma unroll
        for (int j = 0; j < kSlots; ++j) {
            if (kWave * j < Nf) {
                const int k = lane + kWave * j, k2 = k + kWave * kSlots;
                if (k < Nf) cur[k] = r0[j];
                if (k2 < Nf) cur[k2] = r1[j];
            }
        }
    }
    wsync();
}

// load + normalise one frame into buf; even windows fetch two consecutive samples per load (element alignment only);
// four loads in flight per lane
template <typename T>
__device__ __forceinline__ void frame_load_pairs(const PlanDev &P, const T *__restrict__ x, ClipNorm nm, double2 *buf, int lane,
                                                 int psh) {
    const double sc = sample_scale<T>();
    if (P.even) {
        const int Nc = P.Nc;
        int m = lane;
        for (; m + 3 * kWave < Nc; m += 4 * kWave) {
            double2 q[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) q[u] = ct::PairLoad<T>::get(x + 2 * (m + kWave * u));
#pragma unroll
            for (int u = 0; u < 4; ++u)
                buf[skew(m + kWave * u, psh)] = make_double2(fma(q[u].x, sc, -nm.mean) * nm.inv, fma(q[u].y, sc, -nm.mean) * nm.inv);
        }
        for (; m < Nc; m += kWave) {
            const double2 q = ct::PairLoad<T>::get(x + 2 * m);
            buf[skew(m, psh)] = make_double2(fma(q.x, sc, -nm.mean) * nm.inv, fma(q.y, sc, -nm.mean) * nm.inv);
        }
    } else {
        const int W = P.W;
        int n = lane;
        for (; n + 3 * kWave < W; n += 4 * kWave) {
            double q[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) q[u] = load_sample<T>(x + n + kWave * u);
#pragma unroll
            for (int u = 0; u < 4; ++u) buf[skew(n + kWave * u, psh)] = make_double2(fma(q[u], sc, -nm.mean) * nm.inv, 0.0);
        }
        for (; n < W; n += kWave) buf[skew(n, psh)] = make_double2(fma(load_sample<T>(x + n), sc, -nm.mean) * nm.inv, 0.0);
    }
    wsync();
}

// ---- time-domain and spectral stages on CONTIGUOUS per-lane chunks ---------------------------------------------
// Lane l owns a contiguous run of n / 64 (+ 1) elements of the frame (samples) or of the spectrum (bins).  A chunk meets at most two of the ten entropy blocks (or the last
// block and the tail the reference leaves out of the blocks, ShortTermFeatures.py:37-41, :93-98), so a lane carries two
// partial energies and the wave reduces per block; everything is read in groups of four (a lone wave hides LDS latency
// through instruction-level parallelism only).
struct Chunk {
    int base, kb, ke;    // elements every lane has (wave-uniform), first element, one past the last (base or base + 1 elements)
    int cat;             // entropy block of element kb (10 = tail)
    int bound;           // first element of the next category
};
// even split: n / 64 elements per lane, the first n % 64 lanes one more -- the main loops run over `base` elements without
// masks, the extra element is an epilogue
__device__ __forceinline__ Chunk make_chunk(int n, int block_len, int lane) {
    Chunk ch;
    const int rem = n % kWave;
    ch.base = n / kWave;
    ch.kb = ch.base * lane + min(lane, rem);
    ch.ke = ch.kb + ch.base + (lane < rem ? 1 : 0);
    ch.cat = min(ch.kb / block_len, 10);
    ch.bound = (ch.cat >= 10) ? 0x7fffffff : (ch.cat + 1) * block_len;
    return ch;
}
// E[j] = sum over lanes of the partial that belongs to block j (j = 10: tail); issued together (independent chains)
__device__ __forceinline__ void block_sums(const Chunk &ch, double ea, double eb, double (&blk)[10], double &tail) {
#pragma unroll
    for (int j = 0; j < 10; ++j) blk[j] = wsum(((ch.cat == j) ? ea : 0.0) + ((ch.cat + 1 == j) ? eb : 0.0));
    tail = wsum(((ch.cat == 10) ? ea : 0.0) + ((ch.cat + 1 == 10) ? eb : 0.0));
}

// wave minimum of non-negative ints on DPP (the generic xor tree goes through ds_bpermute: six dependent LDS round trips)
__device__ __forceinline__ int wmin_nonneg_i(int v) {
    v = group_min_i(v);
    v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, 0x142, 0xA, 0xF, false));      // row_bcast15 into rows 1, 3
    v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, 0x143, 0xC, 0xF, false));      // row_bcast31 into rows 2, 3
    return __builtin_amdgcn_readlane(v, 63);
}

// chroma of one spectrum: lanes 0..11 return their pitch class (ShortTermFeatures.py:285-308); the gather list is walked
// four entries at a time (index loads, then spectrum and weight loads, then the sums: a lone wave needs the batching)
__device__ __forceinline__ double chroma_class_batched(const Tabs &tb, const double *spec, double sP, int lane) {
    double acc = 0.0;
    if (lane < 12) {
        const int b = tb.ch_start[lane], e = tb.ch_start[lane + 1];
        int i = b;
        for (; i + 4 <= e; i += 4) {
            int src[4];
            double x[4], w[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) src[u] = tb.ch_src[i + u];
#pragma unroll
            for (int u = 0; u < 4; ++u) { x[u] = spec[src[u]]; w[u] = tb.ch_w[i + u]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) acc += (x[u] * x[u]) * w[u];          // same order as the one-by-one walk
        }
        for (; i < e; ++i) {
            const double x = spec[tb.ch_src[i]];
            acc += (x * x) * tb.ch_w[i];
        }
        acc = (sP == 0.0) ? acc / kEps : fast_div(acc, sP);
    }
    return acc;
}

// zcr count, energy and energy entropy of the normalised frame in LDS (ShortTermFeatures.py:22-51)
__device__ __forceinline__ TimeFeat time_features_chunked(const PlanDev &P, const double2 *buf, const Chunk &ch, int lane,
                                                          int psh) {
    const int W = P.W, even = P.even;
    // sample n: even windows pack two per complex slot (slot n / 2, half n & 1), odd windows one (real part); slots skewed
    auto at = [&](int n) { return even ? 2 * skew(n >> 1, psh) + (n & 1) : 2 * skew(n, psh); };
    (void)W;
    const double *y = reinterpret_cast<const double *>(buf);
    double ea = 0.0, eb = 0.0;
    int zc = 0;
    // np.sign from the bit pattern: 0 for +-0, else +-1 (six 32-bit operations instead of two FP64 compares and selects)
    auto sgn = [](double x) {
        const int hi = __double2hiint(x), lo = __double2loint(x);
        return (((hi & 0x7fffffff) | lo) != 0) ? ((hi >> 31) | 1) : 0;
    };
    int sprev = sgn(y[at(max(ch.kb - 1, 0))]);             // (lane 0: sample 0 against itself counts nothing)
    auto one = [&](int n, double x) {
        const double sq = x * x;
        const double sa = (n < ch.bound) ? sq : 0.0;
        ea += sa;
        eb += sq - sa;                                         // exactly 0 or sq
        const int sx = sgn(x);
        zc += abs(sx - sprev);
        sprev = sx;
    };
    int i = 0;
    for (; i + 4 <= ch.base; i += 4) {
        double v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = y[at(ch.kb + i + u)];
#pragma unroll
        for (int u = 0; u < 4; ++u) one(ch.kb + i + u, v[u]);
    }
    for (int n = ch.kb + i; n < ch.ke; ++n) one(n, y[at(n)]);
    double eblk[10], e_tail;
    block_sums(ch, ea, eb, eblk, e_tail);
    TimeFeat tf;
    tf.e_tot = e_tail;
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

// the 34 base features of one frame into fv[0..33] (ShortTermFeatures.py:626-667)
__device__ __forceinline__ void frame_features_chunked(const PlanDev &P, const Tabs &tb, const TimeFeat &tf,
                                                       const double *cur, const double *prv, double *fv, double *msp,
                                                       const Chunk &ch, int lane) {
    const int W = P.W, Nf = P.Nf;
    const double f0 = P.fs / (2.0 * (double)Nf);
    // ---------- sweep A over the lane's bins: sums, max, block energies (:57-107)
    double sX = 0.0, sXp = 0.0, sIX = 0.0, mx = 0.0, ea = 0.0, eb = 0.0;
    auto sweep_a = [&](int k, double X, double Xp) {
        sX += X;
        sXp += Xp;
        sIX = fma((double)(k + 1), X, sIX);
        mx = fmax(mx, X);
        const double sq = X * X;
        const double sa = (k < ch.bound) ? sq : 0.0;
        ea += sa;
        eb += sq - sa;                                         // exactly 0 or sq
    };
    {
        int i = 0;
        for (; i + 4 <= ch.base; i += 4) {
            double a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { a[u] = cur[ch.kb + i + u]; b[u] = prv[ch.kb + i + u]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) sweep_a(ch.kb + i + u, a[u], b[u]);
        }
        for (int k = ch.kb + i; k < ch.ke; ++k) sweep_a(k, cur[k], prv[k]);
    }
    double pblk[10], p_tail;
    block_sums(ch, ea, eb, pblk, p_tail);
    const double own = ea + eb;                     // energy of this lane's bins (roll-off scan)
    const double before = wscan_incl(own) - own;
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
    // spectral entropy: lane j (< 10) owns block j (:101-105)
    double ent_f;
    {
        double num = 0.0;
#pragma unroll
        for (int j = 0; j < 10; ++j)
            if (lane == j) num = pblk[j];
        const double s = fast_div(num, sP + kEps);
        ent_f = wsum((lane < 10) ? -(s * fast_log2(s + kEps)) : 0.0);
    }
    // ---------- centroid, then sweep B: spread + flux + roll-off (:57-82, :110-140)
    const double r = (mx == 0.0) ? 1.0 / kEps : fast_div(1.0, mx);
    const double den = sX * r + kEps;
    const double cen = fast_div(sIX * r, den);
    const double rX = fast_div(1.0, sXe), rXp = fast_div(1.0, sXp);
    const double thr = 0.90 * sP;
    double sSp = 0.0, sFl = 0.0, run = before;
    int first = 0x7fffffff;
    auto sweep_b = [&](int k, double X, double Xp) {
        const double dv = (double)(k + 1) * f0 - cen;
        sSp = fma(dv * dv, X * r, sSp);
        const double df = X * rX - Xp * rXp;
        sFl = fma(df, df, sFl);
        run = fma(X, X, run);                                   // cumsum(X^2)[k]
        if (run + kEps > thr) first = min(first, k);            // first k with cumsum + eps > 0.9 sum (:134-139)
    };
    {
        int i = 0;
        for (; i + 4 <= ch.base; i += 4) {
            double a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { a[u] = cur[ch.kb + i + u]; b[u] = prv[ch.kb + i + u]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) sweep_b(ch.kb + i + u, a[u], b[u]);
        }
        for (int k = ch.kb + i; k < ch.ke; ++k) sweep_b(k, cur[k], prv[k]);
    }
    sSp = wsum(sSp);
    sFl = wsum(sFl);
    first = wmin_nonneg_i(first);
    const double spread = fast_sqrt(fast_div(sSp, den));

    // ---------- MFCC: sparse mel dot, log10, 13 x 40 DCT (:236-254)
    if (lane < 40) {
        const int lo = tb.mel_lo[lane], cnt = tb.mel_cnt[lane];
        const double *w = tb.mel_w + tb.mel_off[lane];
        double a0 = 0.0, a1 = 0.0;
        int i = 0;
        for (; i + 8 <= cnt; i += 8) {                   // eight bins and weights in flight
            double xb[8], wb[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { xb[u] = cur[lo + i + u]; wb[u] = w[i + u]; }
#pragma unroll
            for (int u = 0; u < 8; u += 2) { a0 = fma(xb[u], wb[u], a0); a1 = fma(xb[u + 1], wb[u + 1], a1); }
        }
        for (; i + 2 <= cnt; i += 2) {
            a0 = fma(cur[lo + i], w[i], a0);
            a1 = fma(cur[lo + i + 1], w[i + 1], a1);
        }
        if (i < cnt) a0 = fma(cur[lo + i], w[i], a0);
        msp[lane] = fast_log10((a0 + a1) + kEps);
    }
    // ---------- chroma (:277-321)
    const double chroma = chroma_class_batched(tb, cur, sP, lane);
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
    // population std of the 12 chroma values (:667): lanes 0..11 of the first row
    {
        const double cv = (lane < 12) ? chroma : 0.0;
        const double mean = group_sum(cv) / 12.0;
        const double d = (lane < 12) ? cv - mean : 0.0;
        const double var = group_sum(d * d) / 12.0;
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
            fv[33] = fast_sqrt(var);
        }
    }
    wsync();
}

// ---- host: radix schedule, permutation, LDS layout + table blob ---------------------------------------------------
inline bool mix_factor(int n, std::vector<int> &radix, bool radix16 = true) {
    radix.clear();
    int twos = 0;
    while (n % 2 == 0) { ++twos; n /= 2; }
    std::vector<int> odd;
    for (int f : {13, 11, 7, 5, 3})
        while (n % f == 0) { odd.push_back(f); n /= f; }
    if (n != 1) return false;
    // big strides first for the power-of-two butterflies, the odd radices (conflict-free at small strides) last
    while (radix16 && twos >= 4 && twos != 5 && twos != 6) { radix.push_back(16); twos -= 4; }      // (5 = 8 4, 6 = 8 8)
    while (twos >= 3 && twos != 4) { radix.push_back(8); twos -= 3; }
    while (twos >= 2) { radix.push_back(4); twos -= 2; }
    if (twos) radix.push_back(2);
    for (int f : odd) radix.push_back(f);
    return (int)radix.size() <= kMaxPass && !radix.empty();
}

// position that holds Z[k] after the passes: digit q_p of k (least significant first) sits at weight Nc / (R_0 .. R_p)
inline void mix_permutation(int Nc, const std::vector<int> &radix, std::vector<unsigned short> &perm) {
    perm.assign((size_t)Nc, 0);
    for (int k = 0; k < Nc; ++k) {
        int rest = k, weight = Nc, pos = 0;
        for (int R : radix) {
            weight /= R;
            pos += (rest % R) * weight;
            rest /= R;
        }
        perm[k] = (unsigned short)pos;
    }
}

// 0: this window is not for the mixed-radix kernel.  Fills the layout and the host image of the shared table region.
inline int mix_layout(const FftPlan &fft, const MelTable *mel, const ChromaTable *chroma, int F, MixLayout &L,
                      std::vector<unsigned char> *blob) {
    const int Nc = fft.len, Nf = fft.window / 2;
    if (Nc < 2 || Nc > 60000 || Nf < 64) return 0;          // (chunks of the feature stages: at most two entropy blocks per lane)
    std::vector<int> radix;
    if (!mix_factor(Nc, radix)) return 0;
    if (fft.even ? (Nc / 2 + 1 > kWave * kSlots) : (Nf > 2 * kWave * kSlots)) return 0;
    memset(&L, 0, sizeof(L));
    auto set_passes = [&]() {
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
    };
    set_passes();
    L.unit_bytes = (Nf * 8 + 15) / 16 * 16;
    // (the skew is an instance of the kernel: it is decided with the lean instance below; room for it is reserved here)
    L.pad_shift = (Nc % 32 == 0 && !experiment_env("PAA_MIX_NO_SKEW")) ? 5 : 31;
    L.buf_bytes = (Nc + (Nc >> L.pad_shift)) * 16;
    const int FF = F > 0 ? F : 1;
    L.wave_bytes = (L.buf_bytes + L.unit_bytes + kFlush * FF * 8 + 48 * 8 + 40 * 8 + 15) / 16 * 16;
    const size_t n_melw = mel ? mel->w.size() : 0, n_ch = chroma ? chroma->src.size() : 0;
    const size_t n_post = fft.even ? (size_t)(Nc / 2 + 1) : 1;
    auto lay = [&](int tw_global, int max_waves) {
        int off = 0;
        auto take = [&off](size_t bytes) { const int o = off; off += (int)((bytes + 15) / 16 * 16); return o; };
        L.off_tw = take(tw_global ? 16 : (size_t)Nc * 16);
        L.off_post = take(tw_global ? 16 : n_post * 16);
        L.off_perm = take((size_t)Nc * 2);
        L.off_mello = take(40 * 4);
        L.off_melcnt = take(40 * 4);
        L.off_meloff = take(40 * 4);
        L.off_melw = take(std::max<size_t>(n_melw, 1) * 8);
        L.off_dct = take(13 * 41 * 8);
        L.off_chstart = take(13 * 4);
        L.off_chsrc = take(std::max<size_t>(n_ch, 1) * 4);
        L.off_chw = take(std::max<size_t>(n_ch, 1) * 8);
        L.table_bytes = off;
        int waves = max_waves;
        while (waves > 0 && (size_t)L.table_bytes + (size_t)waves * L.wave_bytes > 160 * 1024) --waves;
        return waves;
    };
    // the twiddle tables go to LDS unless that costs a wave
    int w_lds = lay(0, 4), w_glob = lay(1, 4);
    L.tw_global = (w_glob > w_lds) ? 1 : 0;
    if (const char *force = experiment_env("PAA_MIX_TW_GLOBAL")) L.tw_global = atoi(force) ? 1 : 0;      // A/B experiments
    L.waves = lay(L.tw_global, 4);
    if (L.waves < 1) return 0;
    // small windows: the LDS has room for more than four waves -- the lean instance (<= 256 registers, fewer butterflies
    // and magnitude slots in flight per lane) runs six to eight waves per CU, i.e. up to two per SIMD
    L.lean = 0;
    const bool lean_fits = fft.even ? (Nc / 2 + 1 <= kWave * kSlotsLean) : (Nf <= 2 * kWave * kSlotsLean);
    bool small_radices = true;         // (the radix-11 / 13 / 16 butterflies need more registers than the lean instance has)
    for (int r : radix) small_radices &= (r <= 8 || r == 16);
    if (lean_fits && small_radices && !experiment_env("PAA_MIX_NO_LEAN")) {
        w_lds = lay(0, 8); w_glob = lay(1, 8);
        const int g8 = (w_glob > w_lds) ? 1 : 0, w8 = std::max(w_lds, w_glob);
        if (w8 >= 6) { L.lean = 1; L.tw_global = g8; L.waves = lay(g8, 8); }
        else L.waves = lay(L.tw_global, 4);
    }
    if (L.lean) {                      // (the radix-16 codelet alone needs 128 registers of operands: radix 8 / 4 instead)
        if (!mix_factor(Nc, radix, false)) return 0;
        set_passes();
    }
    if (!blob) return 1;
    blob->assign((size_t)L.table_bytes, 0);
    unsigned char *b = blob->data();
    if (!L.tw_global) {
        memcpy(b + L.off_tw, fft.tw.data(), (size_t)Nc * 16);
        if (fft.even) memcpy(b + L.off_post, fft.post.data(), n_post * 16);
    }
    std::vector<unsigned short> perm;
    mix_permutation(Nc, radix, perm);
    if (L.lean && L.pad_shift == 5)
        for (auto &pp : perm) pp = (unsigned short)(pp + (pp >> 5));      // positions in the skewed buffer
    memcpy(b + L.off_perm, perm.data(), (size_t)Nc * 2);
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
    return 1;
}

inline size_t mix_lds_bytes(const MixLayout &L) { return (size_t)L.table_bytes + (size_t)L.waves * L.wave_bytes; }

// TWG = 1: twiddles / post-twiddles are read from global memory (the tables of a 2400-sample window would cost a wave);
// LEAN = 1: up to eight waves per workgroup, <= 256 registers (small windows)
template <typename T, int TWG, int LEAN>
__global__ __launch_bounds__(LEAN ? 512 : 256) void st_mix_kernel(PlanDev P, MixLayout L, const unsigned char *__restrict__ blob,
                                                      const T *__restrict__ sig, const ClipDev *__restrict__ clips,
                                                      const ClipNorm *__restrict__ norms, const Tile *__restrict__ tiles,
                                                      int n_tiles, double *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    {
        const int4 *src4 = reinterpret_cast<const int4 *>(blob);
        int4 *dst4 = reinterpret_cast<int4 *>(smem);
        for (int n = threadIdx.x; n < L.table_bytes / 16; n += blockDim.x) dst4[n] = src4[n];
    }
    __syncthreads();       // the only workgroup-wide barrier
    Tabs tb;
    tb.tw = TWG ? P.tw : reinterpret_cast<const double2 *>(smem + L.off_tw);
    tb.post = TWG ? P.post : reinterpret_cast<const double2 *>(smem + L.off_post);
    const unsigned short *perm = reinterpret_cast<const unsigned short *>(smem + L.off_perm);
    tb.mel_lo = reinterpret_cast<const int *>(smem + L.off_mello);
    tb.mel_cnt = reinterpret_cast<const int *>(smem + L.off_melcnt);
    tb.mel_off = reinterpret_cast<const int *>(smem + L.off_meloff);
    tb.mel_w = reinterpret_cast<const double *>(smem + L.off_melw);
    tb.dct = reinterpret_cast<const double *>(smem + L.off_dct);
    tb.dct_stride = 41;
    tb.ch_start = reinterpret_cast<const int *>(smem + L.off_chstart);
    tb.ch_src = reinterpret_cast<const int *>(smem + L.off_chsrc);
    tb.ch_w = reinterpret_cast<const double *>(smem + L.off_chw);

    constexpr int PSH = (LEAN == 2) ? 5 : -1;       // instance 2: lean + skewed FFT buffer
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int tile_id = blockIdx.x * L.waves + wave;
    if (tile_id >= n_tiles) return;
    const int Nf = P.Nf, F = P.F > 0 ? P.F : 1;
    unsigned char *wb = smem + L.table_bytes + wave * L.wave_bytes;
    double *otile = reinterpret_cast<double *>(wb + L.buf_bytes + L.unit_bytes);
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
    double vprev = 0.0;
    int nslot = 0, tbase = tl.t0, odd = 0;
    const int tend = tl.t0 + tl.cnt;
    const Chunk ch_t = make_chunk(P.W, P.blk_t, lane), ch_f = make_chunk(Nf, P.blk_f, lane);
    PAA_T0()
    for (int t = tl.t0 - h; t < tend; ++t, odd ^= 1) {
        // even frames of the run: transform at the front, spectrum at the very front, previous spectrum behind the buffer;
        // odd frames: transform one unit further, spectrum in its last unit, previous spectrum at the very front
        double2 *buf = reinterpret_cast<double2 *>(wb + (odd ? L.unit_bytes : 0));
        double *cur = reinterpret_cast<double *>(wb + (odd ? L.buf_bytes : 0));
        double *prv = reinterpret_cast<double *>(wb + (odd ? 0 : L.buf_bytes));
        const T *x = x0 + (long long)t * P.S;
        frame_load_pairs<T>(P, x, nm, buf, lane, PSH);
        PAA_TICK(0)
        const bool want = (P.mode == 0) && ((t >= tl.t0) || (P.deltas && t == tl.t0 - 1));
        TimeFeat tf;
        tf.e_tot = 0.0; tf.ent_e = 0.0; tf.zc = 0;
        if (want) tf = time_features_chunked(P, buf, ch_t, lane, PSH);
        int touched = 0;
        if (t + 1 < tend) {
            // pull the next frame's new samples (the S behind this frame's end) towards the L2 / L1 now: their latency then
            // runs under this frame's transform and feature stage (the values are only "used" at the end of the iteration)
            const char *nb = reinterpret_cast<const char *>(x + P.W);
            const int nbytes = P.S * (int)sizeof(T);
            for (int o = 64 * lane; o + 4 <= nbytes; o += 64 * kWave) touched ^= *reinterpret_cast<const int *>(nb + o);
        }
        PAA_TICK(1)
        fft_passes_inplace<LEAN>(P, L, buf, tb.tw, lane);
        PAA_TICK(2)
        magnitudes_inplace<LEAN>(P, buf, tb.post, perm, cur, lane);
        PAA_TICK(3)
        if (P.mode == 1) {            // spectrogram row (ShortTermFeatures.py:422)
            double *row = oc + (long long)t * Nf;
            for (int k = lane; k < Nf; k += kWave) __builtin_nontemporal_store(cur[k], row + k);
        } else if (P.mode == 2) {     // chromagram row (:356-359)
            double p = 0.0;
            for (int k = lane; k < Nf; k += kWave) { const double X = cur[k]; p = fma(X, X, p); }
            p = wsum(p);
            const double ch = chroma_class(tb, cur, p, lane);
            if (lane < 12) oc[(long long)t * 12 + lane] = ch;
        } else {
            if (want) {
                frame_features_chunked(P, tb, tf, cur, (t == 0) ? cur : prv, fv, msp, ch_f, lane);
                PAA_TICK(5)
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
        asm volatile("" ::"v"(touched));      // (the prefetch loads retire here at the latest)
        wsync();
        PAA_TICK(10)
    }
    PAA_TEND()
}

}  // namespace mix
}  // namespace paa
