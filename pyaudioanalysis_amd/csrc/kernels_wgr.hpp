// Large even windows whose packed transform is N = R1 R2 R3 complex points with fat radices (10 / 20): the 1 s windows
// audioSegmentation.music_thumbnailing passes by default (audioSegmentation.py:1134-1138: 16 000 samples at 16 kHz = 20 x 20 x 20
// points, 8 000 samples at 8 kHz = 10 x 20 x 20) -- ONE launch for everything, ONE WORKGROUP per CU, the transform in REGISTERS.
//
// kernels_wg.hpp (round 5) ran these as five in-place radix passes over a 128 KB LDS buffer (a barrier and a full LDS round trip per
// pass, 216 vector instructions per point), wrote every spectrum row to HBM and read it twice in a second launch: 5.7e6 frames/s at
// 16 000 / 8 000, 15.5 x the algorithmic traffic.  Here:
//
//   a workgroup (NT = 448 threads for 16 000: 400 lane jobs per pass) owns a RUN of consecutive frames of one clip, one frame at a time:
//   load     : thread j < J1 = R2 R3 fetches z[j + J1 n0], n0 < R1 (sample pairs 2 (j + J1 n0): per load instruction the threads read one
//              contiguous span) and normalises them (ShortTermFeatures.py:567-570)
//   time     : the same registers give the energy, the ten entropy-block energies (block = n0 / (R1 / 10): static per register row) and
//              the sign changes (the sample before a pair is in the lane below: DPP wave_shr:1; lane 0 of a wave fetches it)
//   pass 1   : radix-R1 codelet (kernels_ct.hpp: 20 = 4 x 5 / 10 = 2 x 5 prime-factor forms, exact zeros for equal inputs), outputs
//              times W_N^(j k0) -- the powers of ONE table value, formed by squaring / multiplying
//   exchange : element (k0, n1, n2) at buf[k0 A1 + n1 R3 + n2]          (16-byte elements; A1, A2: scripts/dev/wgr_model.py -- every
//   pass 2   : thread (k0, n2): radix R2 over n1, outputs times W_(R2 R3)^(n2 k1)         ds_write_b128 / ds_read_b128 of the three
//   exchange : element (k0, k1, n2) at buf[k0 A2 + k1 B2 + n2]                               exchanges is bank-conflict free)
//   pass 3   : thread (k1, k0): radix R3 over n2 -> Z[k0 + R1 k1 + R1 R2 k2]
//   exchange : Z in natural order
//   recombine: thread t < J1 takes the pairs k = t + J1 jj, jj < R1 / 2: X[k] = E + w^k O and X[N - k] from Z[k], Z[N - k] (:617-621);
//              the R1 magnitudes stay in registers -- they are the PREVIOUS spectrum of the next frame's flux (a run that starts inside a
//              clip transforms the frame before it once more: halo) -- and go, in natural order, over the dead transform buffer
//   features : sums / maximum / spread / flux from the registers; ONE scan of the LDS row (contiguous chunks of R1 bins) gives the
//              roll-off bin and, as differences of the running energy at the block boundaries, the ten spectral-entropy blocks; mel
//              filters and chroma classes one wave at a time (kernels_wg.hpp's walk); the logarithms of the two entropies, the DCT and the
//              chroma deviation on four different waves at once
// Nothing but the samples (read once + the overlap of the windows from L2) and the feature columns touches HBM; spectrogram plans write
// each row once from the registers.  Three passes instead of five, three exchanges through LDS instead of ten round trips.
// Replaces ShortTermFeatures.py:608-682 (+ helpers :22-140, :236-321), spectrogram (:415-422), chromagram (:349-359) for these windows.
#pragma once
#include "kernels_tri.hpp"          // tri::Cd (register codelets)

namespace paa {
namespace wgr {

template <int R1_, int R2_, int R3_, int A1_, int A2_, int B2_>
struct Shape {
    static constexpr int R1 = R1_, R2 = R2_, R3 = R3_, A1 = A1_, A2 = A2_, B2 = B2_;
    static constexpr int N = R1 * R2 * R3, W = 2 * N, NF = N;
    static constexpr int J1 = R2 * R3, J2 = R1 * R3, J3 = R1 * R2;          // lane jobs of the three passes
    static constexpr int JMAX = J1 > J2 ? (J1 > J3 ? J1 : J3) : (J2 > J3 ? J2 : J3);
    static constexpr int NW = (JMAX + 63) / 64, NT = 64 * NW;
    static constexpr int NJR = R1 / 2;                    // pair jobs per recombination thread: k = t + J1 jj
    static constexpr int C = R1;                          // bins per thread of the scan (contiguous)
    static constexpr int CB = J1 / 10;                    // threads per spectral-entropy block
    static constexpr int RB = R1 / 10;                    // register rows per time-domain entropy block
    static constexpr int BUF0 = R1 * A1 > R1 * A2 ? R1 * A1 : R1 * A2;
    static constexpr int BUF = BUF0 > N ? BUF0 : N;       // double2 elements of the exchange buffer
    static constexpr int NFW = (40 + NW - 1) / NW;        // mel filters per wave
    // LDS behind the buffer (doubles): red [NW][16], red2 [NW][2], slot [NW], bnd [12], msp [40], fv [48], redi [NW] ints
    static constexpr int OFF_RED = BUF * 16;
    static constexpr int OFF_RED2 = OFF_RED + NW * 16 * 8;
    static constexpr int OFF_SLOT = OFF_RED2 + NW * 2 * 8;
    static constexpr int OFF_BND = OFF_SLOT + NW * 8;
    static constexpr int OFF_MSP = OFF_BND + 12 * 8;
    static constexpr int OFF_FV = OFF_MSP + 40 * 8;
    static constexpr int OFF_REDI = OFF_FV + 48 * 8;
    static constexpr int LDS_BYTES = (OFF_REDI + NW * 4 + 15) / 16 * 16;
    static_assert(R1 % 10 == 0, "time-domain entropy blocks: static per register row");
    static_assert(J1 % 10 == 0, "spectral entropy blocks: whole scan chunks");
    static_assert(A1 >= (R2 - 1) * R3 + R3 && B2 >= R3 && A2 >= (R2 - 1) * B2 + R3, "exchange rows");
    static_assert(LDS_BYTES <= 160 * 1024, "one workgroup's LDS");
    static_assert(NW >= 5, "the final stage spreads over five waves");
};
// pads from scripts/dev/wgr_model.py (reads and writes of both exchanges at the conflict-free cycle count)
typedef Shape<20, 20, 20, 404, 401, 20> S16000;
typedef Shape<10, 20, 20, 404, 439, 22> S8000;

__device__ __forceinline__ double2 csqr(double2 a) { return make_double2(fma(a.x, a.x, -a.y * a.y), 2.0 * (a.x * a.y)); }
// lane l receives the value of lane l - 1 (lane 0: `first`)
__device__ __forceinline__ int shr1(int v, int first) { return __builtin_amdgcn_update_dpp(first, v, 0x138, 0xF, 0xF, false); }

// W^q, q = 1 .. R - 1, as products of earlier powers (depth log2 R), times the codelet outputs
template <int R, typename CD>
__device__ __forceinline__ void twiddle_outputs(double2 *v, double2 w) {
    // (opaque: the powers of a loop-invariant value would be hoisted out of the frame loop -- 2 x 19 complex values per thread -- and spilled)
    asm volatile("" : "+v"(w.x), "+v"(w.y));
    double2 wq[R];
    wq[1] = w;
#pragma unroll
    for (int q = 2; q < R; ++q) wq[q] = (q % 2 == 0) ? csqr(wq[q / 2]) : cmul(wq[q / 2], wq[q - q / 2]);
#pragma unroll
    for (int q = 1; q < R; ++q) v[CD::pos(q)] = cmul(v[CD::pos(q)], wq[q]);
}

__device__ __forceinline__ int sign_code(double d) { return ((d > 0.0) ? 1 : 0) - ((d < 0.0) ? 1 : 0); }

// MODE 0: the 34 feature rows, 1: spectrogram rows, 2: chromagram rows
template <typename SH, typename T, int MODE>
__global__ __launch_bounds__(SH::NT) void wgr_kernel(PlanDev P, const T *__restrict__ sig, const ClipDev *__restrict__ clips,
                                                     const ClipNorm *__restrict__ norms, const Tile *__restrict__ runs, int n_runs,
                                                     double *__restrict__ out) {
    constexpr int R1 = SH::R1, R2 = SH::R2, R3 = SH::R3, A1 = SH::A1, A2 = SH::A2, B2 = SH::B2, N = SH::N, NF = SH::NF, W = SH::W;
    constexpr int J1 = SH::J1, J2 = SH::J2, J3 = SH::J3, NW = SH::NW, NJR = SH::NJR, C = SH::C;
    typedef tri::Cd<R1> CD1;
    typedef tri::Cd<R2> CD2;
    typedef tri::Cd<R3> CD3;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double2 *buf = reinterpret_cast<double2 *>(smem);
    double *mags = reinterpret_cast<double *>(smem);              // the frame's spectrum, natural order, over the dead buffer
    double *red = reinterpret_cast<double *>(smem + SH::OFF_RED);
    double *red2 = reinterpret_cast<double *>(smem + SH::OFF_RED2);
    double *slot = reinterpret_cast<double *>(smem + SH::OFF_SLOT);
    double *bnd = reinterpret_cast<double *>(smem + SH::OFF_BND);
    double *msp = reinterpret_cast<double *>(smem + SH::OFF_MSP);
    double *fv = reinterpret_cast<double *>(smem + SH::OFF_FV);
    int *redi = reinterpret_cast<int *>(smem + SH::OFF_REDI);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // ---- the thread's jobs (threads past a pass's job count shadow its last job; their stores are masked)
    const bool a1 = tid < J1, a2 = tid < J2, a3 = tid < J3;
    const int j1 = a1 ? tid : J1 - 1;
    const int t2 = a2 ? tid : J2 - 1, k0_2 = t2 / R3, n2_2 = t2 - k0_2 * R3;
    const int u3 = a3 ? tid : J3 - 1, k1_3 = u3 / R1, k0_3 = u3 - k1_3 * R1;
    const double2 w1 = P.tw[j1];                         // W_N^j
    const double2 w2 = P.tw[R1 * n2_2];                  // W_(R2 R3)^n2
    const int e1r = k0_2 * A1 + n2_2, e2w = k0_2 * A2 + n2_2, e2r = k0_3 * A2 + k1_3 * B2;
    const double sc = sample_scale<T>();
    const double invNf = 1.0 / (double)NF;
    const Tabs tb = tabs_global(P);
    const double f0 = P.fs / (2.0 * (double)NF);

    for (int run = blockIdx.x; run < n_runs; run += gridDim.x) {
        const Tile tl = runs[run];
        const ClipDev c = clips[tl.clip];
        const ClipNorm nm = norms[tl.clip];
        const T *xc = sig + c.sample_off + P.frame_origin;
        double *oc = out + c.out_off;
        const long long Tc = c.T;
        // inactive pass-1 threads re-read the last job's samples with scale and mean 0: exact zeros, no energy
        const double scl = a1 ? sc : 0.0, meanl = a1 ? nm.mean : 0.0, inv = nm.inv;
        double pm[2 * NJR];                              // the previous frame's magnitudes (this thread's bins)
        double sXp = 0.0;                                // ... and their sum
#pragma unroll
        for (int i = 0; i < 2 * NJR; ++i) pm[i] = 0.0;
        const int t_first = (MODE == 0 && tl.t0 > 0) ? tl.t0 - 1 : tl.t0;
        for (int t = t_first; t < tl.t0 + tl.cnt; ++t) {
            const bool halo = t < tl.t0;
            const T *x = xc + (long long)t * P.S;
            // ---------------- load + normalise (:567-570)
            double2 v[R1];
#pragma unroll
            for (int n0 = 0; n0 < R1; ++n0) {
                const double2 xx = ct::PairLoad<T>::get(x + 2 * (n0 * J1 + j1));
                v[n0] = make_double2(fma(xx.x, scl, -meanl) * inv, fma(xx.y, scl, -meanl) * inv);
            }
            // ---------------- time domain (:22-51) on the same registers
            if (MODE == 0 && !halo) {
                double eb[10];
#pragma unroll
                for (int b = 0; b < 10; ++b) eb[b] = 0.0;
                // the sample before each pair: the lane below holds it; lane 0 of a wave fetches it (the frame's first sample meets itself)
                int cl0[R1];
#pragma unroll
                for (int n0 = 0; n0 < R1; ++n0) cl0[n0] = 0;
                if (lane == 0) {
#pragma unroll
                    for (int n0 = 0; n0 < R1; ++n0) {
                        const int p = n0 * J1 + j1;
                        cl0[n0] = sign_code((p > 0) ? fma(load_sample<T>(x + 2 * p - 1), sc, -nm.mean) : v[n0].x);
                    }
                }
                int zc = 0;
#pragma unroll
                for (int n0 = 0; n0 < R1; ++n0) {
                    eb[n0 / SH::RB] += fma(v[n0].x, v[n0].x, v[n0].y * v[n0].y);
                    const int c0 = sign_code(v[n0].x), c1 = sign_code(v[n0].y);
                    const int left = shr1(c1, cl0[n0]);
                    zc += a1 ? abs(c0 - left) + abs(c1 - c0) : 0;
                }
#pragma unroll
                for (int b = 0; b < 10; ++b) eb[b] = wsum(eb[b]);
                zc = wsum_i(zc);
                if (lane == 0) {
#pragma unroll
                    for (int b = 0; b < 10; ++b) red[16 * wave + 4 + b] = eb[b];
                    red[16 * wave + 14] = (double)zc;
                }
            }
            // ---------------- pass 1: radix R1 over n0, outputs times W_N^(j k0)
            CD1::run(v);
            twiddle_outputs<R1, CD1>(v, w1);
            if (a1) {
#pragma unroll
                for (int q = 0; q < R1; ++q) buf[q * A1 + j1] = v[CD1::pos(q)];
            }
            __syncthreads();
            // ---------------- pass 2: radix R2 over n1 for (k0, n2), outputs times W_(R2 R3)^(n2 k1)
            double2 v2[R2];
#pragma unroll
            for (int r = 0; r < R2; ++r) v2[r] = buf[e1r + r * R3];
            CD2::run(v2);
            twiddle_outputs<R2, CD2>(v2, w2);
            __syncthreads();
            if (a2) {
#pragma unroll
                for (int q = 0; q < R2; ++q) buf[e2w + q * B2] = v2[CD2::pos(q)];
            }
            __syncthreads();
            // ---------------- pass 3: radix R3 over n2 for (k0, k1): Z[k0 + R1 k1 + R1 R2 k2], natural order into the buffer
            double2 v3[R3];
#pragma unroll
            for (int r = 0; r < R3; ++r) v3[r] = buf[e2r + r];
            CD3::run(v3);
            __syncthreads();
            if (a3) {
#pragma unroll
                for (int q = 0; q < R3; ++q) buf[u3 + q * J3] = v3[CD3::pos(q)];
            }
            __syncthreads();
            // ---------------- real-FFT recombination + |X| / num_fft (:617-621): pairs k = t + J1 jj and N - k (k = 0: bins 0 and N / 2)
            double mg[2 * NJR];
            // (opaque copy of the thread's first bin: everything derived from it below -- table addresses, (double)(k + 1) f0, ... -- is
            // formed again per frame instead of being hoisted out of the frame loop into 100 spilled registers)
            int jf = j1;
            asm volatile("" : "+v"(jf));
            const bool first0 = jf == 0;                   // this thread's pair jj = 0 is k = 0: bins 0 and N / 2
            {
                double2 zk[NJR], zm[NJR], pw[NJR];
#pragma unroll
                for (int jj = 0; jj < NJR; ++jj) {
                    const int k = jf + J1 * jj;
                    const bool k0 = (jj == 0) && first0;
                    pw[jj] = P.post[k];
                    zk[jj] = buf[k];
                    zm[jj] = buf[k0 ? N / 2 : N - k];
                }
#pragma unroll
                for (int jj = 0; jj < NJR; ++jj) {
                    const bool k0 = (jj == 0) && first0;
                    const double2 zh = k0 ? zk[jj] : zm[jj];          // (k = 0 pairs with itself)
                    const double2 e = make_double2(0.5 * (zk[jj].x + zh.x), 0.5 * (zk[jj].y - zh.y));
                    const double2 o = make_double2(0.5 * (zk[jj].y + zh.y), 0.5 * (zh.x - zk[jj].x));
                    const double2 wo = cmul(pw[jj], o);
                    const double ar = e.x + wo.x, ai = e.y + wo.y;
                    double br = e.x - wo.x, bi = e.y - wo.y;
                    if (k0) { br = zm[jj].x; bi = zm[jj].y; }               // bin N / 2: |Z[N / 2]| (w^(N/2) = -i turns O into the imaginary part)
                    mg[2 * jj] = a1 ? mag_sqrt(fma(ar, ar, ai * ai)) * invNf : 0.0;
                    mg[2 * jj + 1] = a1 ? mag_sqrt(fma(br, br, bi * bi)) * invNf : 0.0;
                }
            }
            if (MODE == 1) {
                double *row = oc + (long long)t * NF;
                if (a1) {
#pragma unroll
                    for (int jj = 0; jj < NJR; ++jj) {
                        const int k = jf + J1 * jj;
                        __builtin_nontemporal_store(mg[2 * jj], row + k);
                        __builtin_nontemporal_store(mg[2 * jj + 1], row + (((jj == 0) && first0) ? N / 2 : N - k));
                    }
                }
                __syncthreads();          // every pair has been read: the next frame may write the buffer
                continue;
            }
            // ---------------- sums over the thread's bins (:57-82, :110-124)
            double sXt = 0.0, sIXt = 0.0, mxt = 0.0;
#pragma unroll
            for (int jj = 0; jj < NJR; ++jj) {
                const int k = jf + J1 * jj, kh = ((jj == 0) && first0) ? N / 2 : N - k;
                sXt += mg[2 * jj] + mg[2 * jj + 1];
                sIXt = fma((double)(k + 1), mg[2 * jj], sIXt);
                sIXt = fma((double)(kh + 1), mg[2 * jj + 1], sIXt);
                mxt = fmax(mxt, fmax(mg[2 * jj], mg[2 * jj + 1]));
            }
            sXt = wsum(sXt);
            if (MODE == 0) { sIXt = wsum(sIXt); mxt = wmax_nonneg(mxt); }
            __syncthreads();              // every pair has been read: the magnitudes may overwrite the buffer
            if (lane == 0) { red[16 * wave] = sXt; red[16 * wave + 1] = sIXt; red[16 * wave + 2] = mxt; }
            if (!halo && a1) {
#pragma unroll
                for (int jj = 0; jj < NJR; ++jj) {
                    const int k = jf + J1 * jj;
                    mags[k] = mg[2 * jj];
                    mags[((jj == 0) && first0) ? N / 2 : N - k] = mg[2 * jj + 1];
                }
            }
            __syncthreads();
            double sX = 0.0, sIX = 0.0, mx = 0.0;
#pragma unroll
            for (int w = 0; w < NW; ++w) { sX += red[16 * w]; sIX += red[16 * w + 1]; mx = fmax(mx, red[16 * w + 2]); }
            if (halo) {
#pragma unroll
                for (int i = 0; i < 2 * NJR; ++i) pm[i] = mg[i];
                sXp = sX;
                __syncthreads();          // (red is rewritten by the next frame's time-domain stage)
                continue;
            }
            // ---------------- one scan of the row: chunk energies -> running energy at every chunk start
            double sq[C];
            double cs = 0.0;
            {
                const double2 *m2 = reinterpret_cast<const double2 *>(mags + jf * C);
#pragma unroll
                for (int i = 0; i < C / 2; ++i) {
                    const double2 mm = m2[i];
                    sq[2 * i] = a1 ? mm.x * mm.x : 0.0;
                    sq[2 * i + 1] = a1 ? mm.y * mm.y : 0.0;
                }
#pragma unroll
                for (int i = 0; i < C; ++i) cs += sq[i];
            }
            const double incl = wscan_incl(cs);
            if (lane == 63) slot[wave] = incl;
            int wo = wave;                                 // (opaque: the table records of the wave's filters / classes are fetched per frame,
            asm volatile("" : "+s"(wo));                   // not kept in -- spilled -- scalar registers across the frame loop)
            if (MODE == 0) {
                // ---------------- MFCC filter sums (:236-254): wave w owns the filters w, w + NW, ..., all 64 lanes on a filter's bins, the
                // filters of a wave walked together (kernels_wg.hpp)
                constexpr int NFW = SH::NFW;
                int lo[NFW], cnt[NFW];
                const double *wv[NFW];
                double a[NFW];
                int maxc = 0;
#pragma unroll
                for (int j = 0; j < NFW; ++j) {
                    const int m = wo + NW * j;
                    const int mm = (m < 40) ? m : 39;
                    lo[j] = tb.mel_lo[mm]; cnt[j] = (m < 40) ? tb.mel_cnt[mm] : 0; wv[j] = tb.mel_w + tb.mel_off[mm];
                    a[j] = 0.0;
                    maxc = max(maxc, cnt[j]);
                }
                for (int i = lane; i < maxc; i += 64) {
#pragma unroll
                    for (int j = 0; j < NFW; ++j)
                        if (i < cnt[j]) a[j] = fma(mags[lo[j] + i], wv[j][i], a[j]);
                }
#pragma unroll
                for (int j = 0; j < NFW; ++j) a[j] = wsum(a[j]);
                double mine = a[0];
#pragma unroll
                for (int j = 1; j < NFW; ++j) mine = (lane == j) ? a[j] : mine;
                if (lane < NFW && wo + NW * lane < 40) msp[wo + NW * lane] = fast_log10(mine + kEps);
            }
            __syncthreads();
            double run_e = incl - cs, sP = 0.0;
#pragma unroll
            for (int w = 0; w < NW; ++w) { run_e += (w < wave) ? slot[w] : 0.0; sP += slot[w]; }
            if (MODE == 2) {
                // ---------------- chromagram row (:356-359)
                for (int cls = wo; cls < 12; cls += NW) {
                    const int b = tb.ch_start[cls], e = tb.ch_start[cls + 1];
                    double acc = 0.0;
                    for (int i = b + lane; i < e; i += 64) { const double xv = mags[tb.ch_src[i]]; acc = fma(xv * xv, tb.ch_w[i], acc); }
                    acc = wsum(acc);
                    if (lane == 0) oc[(long long)t * 12 + cls] = (sP == 0.0) ? acc / kEps : fast_div(acc, sP);
                }
                __syncthreads();          // the row has been read: the next frame may write the buffer
                continue;
            }
            // the running energy at the ten block boundaries (spectral entropy, :85-107)
            if (a1 && (tid % SH::CB) == 0) bnd[tid / SH::CB] = run_e;
            if (tid == 0) bnd[10] = sP;
            // ---------------- roll-off (:127-140): first k with cumsum(X^2)[k] + eps > 0.9 sum(X^2)
            {
                const double thr = 0.90 * sP;
                int first = 0x7fffffff;
                double rr = run_e;
#pragma unroll
                for (int i = 0; i < C; ++i) {
                    rr += sq[i];
                    first = (a1 && first == 0x7fffffff && rr + kEps > thr) ? jf * C + i : first;
                }
                first = mix::wmin_nonneg_i(first);
                if (lane == 0) redi[wave] = first;
            }
            // ---------------- spread and flux (:57-82, :110-124) from the registers
            const double r = (mx == 0.0) ? 1.0 / kEps : fast_div(1.0, mx);
            const double den = sX * r + kEps;
            const double cen = fast_div(sIX * f0 * r, den);
            {
                const double sXe = sX + (double)NF * kEps;                // np.sum(X + eps) (:118-119)
                const double sXpe = sXp + (double)NF * kEps;
                const double rX = fast_div(1.0, sXe), rXp = fast_div(1.0, sXpe);
                double sSp = 0.0, sFl = 0.0;
#pragma unroll
                for (int jj = 0; jj < NJR; ++jj) {
                    const int k = jf + J1 * jj, kh = ((jj == 0) && first0) ? N / 2 : N - k;
                    const double dl = (double)(k + 1) * f0 - cen, dh = (double)(kh + 1) * f0 - cen;
                    sSp = fma(dl * dl, mg[2 * jj] * r, sSp);
                    sSp = fma(dh * dh, mg[2 * jj + 1] * r, sSp);
                    const double fl = mg[2 * jj] * rX - pm[2 * jj] * rXp, fh = mg[2 * jj + 1] * rX - pm[2 * jj + 1] * rXp;
                    sFl = fma(fl, fl, sFl);
                    sFl = fma(fh, fh, sFl);
                }
                sSp = wsum(sSp); sFl = wsum(sFl);
                if (lane == 0) { red2[2 * wave] = sSp; red2[2 * wave + 1] = sFl; }
            }
            // ---------------- chroma (:277-321): the pitch classes w and w + NW of a wave, walked together
            {
                const int c0 = wo, c1 = wo + NW;
                const bool two = c1 < 12;
                const int b0 = tb.ch_start[c0], e0 = tb.ch_start[c0 + 1];
                const int b1 = two ? tb.ch_start[c1] : 0, e1 = two ? tb.ch_start[c1 + 1] : 0;
                double acc0 = 0.0, acc1 = 0.0;
                const int n0 = e0 - b0, n1 = e1 - b1, nmax = max(n0, n1);
#pragma unroll 2
                for (int i = lane; i < nmax; i += 64) {
                    if (i < n0) { const double xv = mags[tb.ch_src[b0 + i]]; acc0 = fma(xv * xv, tb.ch_w[b0 + i], acc0); }
                    if (i < n1) { const double xv = mags[tb.ch_src[b1 + i]]; acc1 = fma(xv * xv, tb.ch_w[b1 + i], acc1); }
                }
                acc0 = wsum(acc0); acc1 = wsum(acc1);
                if (lane == 0) fv[21 + c0] = (sP == 0.0) ? acc0 / kEps : fast_div(acc0, sP);
                if (lane == 1 && two) fv[21 + c1] = (sP == 0.0) ? acc1 / kEps : fast_div(acc1, sP);
            }
            __syncthreads();
            // ---------------- the last mile, on five waves at once
            if (wave == 0) {
                if (lane < 13) {                                   // DCT (:250)
                    const double *m = tb.dct + lane * tb.dct_stride;
                    double a0 = 0.0, a1_ = 0.0, a2_ = 0.0, a3_ = 0.0;
#pragma unroll
                    for (int n = 0; n < 40; n += 4) {
                        a0 = fma(m[n], msp[n], a0);
                        a1_ = fma(m[n + 1], msp[n + 1], a1_);
                        a2_ = fma(m[n + 2], msp[n + 2], a2_);
                        a3_ = fma(m[n + 3], msp[n + 3], a3_);
                    }
                    fv[8 + lane] = (a0 + a1_) + (a2_ + a3_);
                }
            } else if (wave == 1) {                                // zero crossings, energy, energy entropy (:22-51)
                double E = 0.0, zct = 0.0;
#pragma unroll
                for (int w = 0; w < NW; ++w) {
                    E += (lane < 10) ? red[16 * w + 4 + (lane < 10 ? lane : 0)] : 0.0;
                    zct += red[16 * w + 14];
                }
                const double e_tot = wsum(E);
                const double s = fast_div(E, e_tot + kEps);
                const double ent = wsum((lane < 10) ? -(s * fast_log2(s + kEps)) : 0.0);
                if (lane == 0) {
                    fv[0] = (zct / 2.0) / (double)(W - 1);
                    fv[1] = e_tot / (double)W;
                    fv[2] = ent;
                }
            } else if (wave == 2) {                                // spectral entropy (:85-107)
                const int l = (lane < 10) ? lane : 0;
                const double Eb = bnd[l + 1] - bnd[l];
                const double s = fast_div(Eb, sP + kEps);
                const double ent = wsum((lane < 10) ? -(s * fast_log2(s + kEps)) : 0.0);
                if (lane == 0) fv[5] = ent;
            } else if (wave == 3) {                                // centroid, spread, flux, roll-off
                double sSp = 0.0, sFl = 0.0;
                int first = 0x7fffffff;
#pragma unroll
                for (int w = 0; w < NW; ++w) { sSp += red2[2 * w]; sFl += red2[2 * w + 1]; first = min(first, redi[w]); }
                if (lane == 0) {
                    fv[3] = cen / (P.fs / 2.0);
                    fv[4] = fast_sqrt(fast_div(sSp, den)) / (P.fs / 2.0);
                    fv[6] = (t == 0) ? 0.0 : sFl;                  // first frame: previous spectrum = itself (:624-625)
                    fv[7] = (first == 0x7fffffff) ? 0.0 : (double)first / (double)NF;
                }
            } else if (wave == 4) {                                // population std of the 12 chroma values (:667)
                const double cv = (lane < 12) ? fv[21 + (lane < 12 ? lane : 0)] : 0.0;
                const double mch = wsum(cv) / 12.0;
                const double d = (lane < 12) ? cv - mch : 0.0;
                const double var = wsum(d * d);
                if (lane == 0) fv[33] = fast_sqrt(var / 12.0);
            }
            __syncthreads();
            if (tid < kBase) oc[(long long)tid * Tc + t] = fv[tid];
#pragma unroll
            for (int i = 0; i < 2 * NJR; ++i) pm[i] = mg[i];
            sXp = sX;
        }       // frames of the run
    }       // runs of this workgroup
}

// ---- host -----------------------------------------------------------------------------------------------------------------------
// 0: kernels_wg.hpp / kernels_big.hpp keep the window; else the shape's id
inline int wgr_shape_id(int window) {
    if (window == S16000::W) return 1;
    if (window == S8000::W) return 2;
    return 0;
}
inline const char *wgr_shape_name(int id) { return id == 1 ? "20x20x20" : "10x20x20"; }

// Runs of consecutive frames, about one per CU (a run that starts inside a clip costs a halo transform in feature plans): every clip
// is cut into ceil(T / L) runs of nearly equal length, L = the per-CU share of all frames
inline void wgr_build_runs(const std::vector<ClipDev> &clips, int num_cu, std::vector<Tile> &runs) {
    long long total = 0;
    for (auto &c : clips) total += std::max(c.T, 0);
    const long long L = std::max<long long>(1, (total + num_cu - 1) / std::max(num_cu, 1));
    for (size_t ci = 0; ci < clips.size(); ++ci) {
        const long long Tc = clips[ci].T;
        if (Tc <= 0) continue;
        const long long n = (Tc + L - 1) / L;
        long long t0 = 0;
        for (long long i = 0; i < n; ++i) {
            const long long cnt = Tc / n + (i < Tc % n ? 1 : 0);
            Tile tl; tl.clip = (int)ci; tl.t0 = (int)t0; tl.cnt = (int)cnt; tl.pad = 0;
            runs.push_back(tl);
            t0 += cnt;
        }
    }
}

template <typename SH, typename T, int MODE>
inline int wgr_launch_one(const PlanDev &P, const void *d_packed, const ClipDev *clips, const ClipNorm *norms, const Tile *runs,
                          long long n_runs, int num_cu, double *d_out, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&wgr_kernel<SH, T, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                SH::LDS_BYTES) != hipSuccess) return -1;
        attr_set = true;
    }
    const unsigned grid = (unsigned)std::min<long long>(n_runs, num_cu);
    hipLaunchKernelGGL((wgr_kernel<SH, T, MODE>), dim3(grid), dim3(SH::NT), (size_t)SH::LDS_BYTES, stream, P, (const T *)d_packed, clips,
                       norms, runs, (int)n_runs, d_out);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace wgr
}  // namespace paa
