// Windows beyond the LDS envelope of the in-LDS kernels (e.g. the 1 s = 16 000 ... 44 100-sample windows of
// audioSegmentation.music_thumbnailing, audioSegmentation.py:1137).  Same algorithm, but the Stockham passes
// ping-pong through HBM scratch, one launch per radix pass over a chunk of frames:
//   big_load  : samples -> normalised packed-complex sequence (bufA)
//   big_time  : zcr / energy / energy entropy per frame from bufA (one wave per frame)
//   big_pass  : one Stockham pass bufX -> bufY for every frame of the chunk (grid-stride over butterflies)
//   big_post  : real-FFT recombination + |X|/Nf -> spectrum rows (scratch, or the spectrogram itself)
//   big_feat  : the 34 features per frame from the spectrum rows in HBM (one wave per frame)
//   big_delta : rows 34..67 = row[t] - row[t-1]
// Traffic is O(passes * 32 B * N) per frame instead of O(1 kB): this path exists so that NO window size falls
// back to the CPU, not for speed.
#pragma once
#include "kernels_generic.hpp"

namespace paa {

template <typename T>
__global__ __launch_bounds__(256) void big_load_kernel(PlanDev P, const T *__restrict__ x0, long long t_first,
                                                        ClipNorm nm_unused, const ClipNorm *__restrict__ norms, int clip,
                                                        double2 *__restrict__ bufA) {
    const ClipNorm nm = norms[clip];
    const long long f = blockIdx.y;
    const T *x = x0 + (t_first + f) * (long long)P.S;
    double2 *dst = bufA + f * (long long)P.Nc;
    const double sc = sample_scale<T>();
    for (int n = blockIdx.x * 256 + threadIdx.x; n < P.W; n += gridDim.x * 256) {
        const double y = fma(load_sample<T>(x + n), sc, -nm.mean) * nm.inv;
        if (P.even) reinterpret_cast<double *>(dst)[n] = y;
        else dst[n] = make_double2(y, 0.0);
    }
    (void)nm_unused;
}

// one wave per frame: TimeFeat -> tfeat[f] = {e_tot, ent_e, zc}
__global__ __launch_bounds__(64) void big_time_kernel(PlanDev P, const double2 *__restrict__ bufA,
                                                       double *__restrict__ tfeat) {
    const long long f = blockIdx.x;
    const TimeFeat tf = time_features(P, bufA + f * (long long)P.Nc, threadIdx.x);
    if (threadIdx.x == 0) {
        tfeat[3 * f] = tf.e_tot;
        tfeat[3 * f + 1] = tf.ent_e;
        tfeat[3 * f + 2] = (double)tf.zc;
    }
}

// grid-stride versions of the Stockham passes
template <int R>
__device__ __forceinline__ void big_pass_r(const double2 *__restrict__ in, double2 *__restrict__ out, int Nc, int Ns,
                                           const double2 *__restrict__ tw, int first, int stride) {
    const int nb = Nc / R;
    const int tstride = Nc / (Ns * R);
    for (int j = first; j < nb; j += stride) {
        const int k = j % Ns;
        double2 v[R];
#pragma unroll
        for (int r = 0; r < R; ++r) v[r] = in[j + r * nb];
        if (Ns > 1) {
#pragma unroll
            for (int r = 1; r < R; ++r) v[r] = cmul(v[r], tw[(long long)r * k * tstride]);
        }
        small_dft<R>(v);
        const int j0 = (j - k) * R + k;
#pragma unroll
        for (int r = 0; r < R; ++r) out[j0 + r * Ns] = v[r];
    }
}

__global__ __launch_bounds__(256) void big_pass_kernel(int Nc, int R, int Ns, const double2 *__restrict__ tw,
                                                        const double2 *__restrict__ in, double2 *__restrict__ out) {
    const long long f = blockIdx.y;
    const double2 *src = in + f * (long long)Nc;
    double2 *dst = out + f * (long long)Nc;
    const int first = blockIdx.x * 256 + threadIdx.x, stride = gridDim.x * 256;
    switch (R) {
        case 2: big_pass_r<2>(src, dst, Nc, Ns, tw, first, stride); break;
        case 3: big_pass_r<3>(src, dst, Nc, Ns, tw, first, stride); break;
        case 4: big_pass_r<4>(src, dst, Nc, Ns, tw, first, stride); break;
        case 5: big_pass_r<5>(src, dst, Nc, Ns, tw, first, stride); break;
        default: {
            const int nb = Nc / R;
            const int a = Nc / (Ns * R);
            for (int o = first; o < Nc; o += stride) {
                const int j = o % nb, q = o / nb;
                const int k = j % Ns;
                const int step = (int)(((long long)k * a + (long long)q * nb) % Nc);
                int idx = 0;
                double ar = 0.0, ai = 0.0;
                for (int p = 0; p < R; ++p) {
                    const double2 x = src[j + (long long)p * nb];
                    const double2 w = tw[idx];
                    ar = fma(x.x, w.x, fma(-x.y, w.y, ar));
                    ai = fma(x.x, w.y, fma(x.y, w.x, ai));
                    idx += step;
                    if (idx >= Nc) idx -= Nc;
                }
                dst[(j - k) * R + k + q * Ns] = make_double2(ar, ai);
            }
        }
    }
}

// spectrum rows: spec[(f + row0) * Nf + k]
__global__ __launch_bounds__(256) void big_post_kernel(PlanDev P, const double2 *__restrict__ Z,
                                                        double *__restrict__ spec, long long row0) {
    const long long f = blockIdx.y;
    const double2 *src = Z + f * (long long)P.Nc;
    double *dst = spec + (f + row0) * (long long)P.Nf;
    const double invNf = 1.0 / (double)P.Nf;
    for (int k = blockIdx.x * 256 + threadIdx.x; k < P.Nf; k += gridDim.x * 256) {
        if (P.even) {
            const double2 zk = src[k];
            const double2 zm = src[k == 0 ? 0 : P.Nc - k];
            const double2 e = make_double2(0.5 * (zk.x + zm.x), 0.5 * (zk.y - zm.y));
            const double2 o = make_double2(0.5 * (zk.y + zm.y), 0.5 * (zm.x - zk.x));
            const double2 wo = cmul(P.post[k], o);
            const double xr = e.x + wo.x, xi = e.y + wo.y;
            dst[k] = sqrt(fma(xr, xr, xi * xi)) * invNf;
        } else {
            const double2 z = src[k];
            dst[k] = sqrt(fma(z.x, z.x, z.y * z.y)) * invNf;
        }
    }
}

// one wave per frame; spectrum row f+1 is the frame, row f the previous frame (row 0 of a chunk = carry-over)
__global__ __launch_bounds__(64) void big_feat_kernel(PlanDev P, const double *__restrict__ spec,
                                                       const double *__restrict__ tfeat, long long t_first,
                                                       long long Tc, double *__restrict__ oc) {
    __shared__ double fv[48];
    __shared__ double msp[40];
    const int lane = threadIdx.x;
    const long long f = blockIdx.x, t = t_first + f;
    const double *cur = spec + (f + 1) * (long long)P.Nf;
    const double *prv = (t == 0) ? cur : spec + f * (long long)P.Nf;
    const Tabs tb = tabs_global(P);
    if (P.mode == 2) {              // chromagram row
        double p = 0.0;
        for (int k = lane; k < P.Nf; k += kWave) { const double X = cur[k]; p = fma(X, X, p); }
        p = wsum(p);
        const double ch = chroma_class(tb, cur, p, lane);
        if (lane < 12) oc[t * 12 + lane] = ch;
        return;
    }
    TimeFeat tf;
    tf.e_tot = tfeat[3 * f];
    tf.ent_e = tfeat[3 * f + 1];
    tf.zc = (int)tfeat[3 * f + 2];
    frame_features(P, tb, tf, cur, prv, fv, msp, lane);
    if (lane < kBase) oc[(long long)lane * Tc + t] = fv[lane];
}

__global__ __launch_bounds__(256) void big_delta_kernel(long long Tc, double *__restrict__ oc) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)kBase * Tc) return;
    const long long row = idx / Tc, t = idx % Tc;
    const double *r = oc + row * Tc;
    oc[(kBase + row) * Tc + t] = (t == 0) ? 0.0 : r[t] - r[t - 1];
}

}  // namespace paa
