// Even windows of 12 x 3675 or 6 x 3675 samples -- the 1 s window audioSegmentation.music_thumbnailing passes by default
// (audioSegmentation.py:1134-1138) at 44.1 kHz (44 100 samples) and at 22.05 kHz (22 050): the packed transform (22 050 / 11 025 complex
// points) does not fit one CU's LDS.  kernels_wg.hpp (round 5) splits the PACKED sequence by r0 = 6 / 3 and has to keep the sub-transforms q
// and r0 - q side by side (the real-FFT recombination pairs bin k with Nc - k), runs five in-place radix passes over them and leaves the
// time-domain features to a kernel of their own.  Here (round 6) the REAL sequence is split instead:
//
//   W = r0 Q, Q = 3675 = 7 x 21 x 25.  For q = 1 .. r0/2 - 1 ("complex unit q")
//       a_q[k] = W_W^(q k) sum_r y[k + Q r] W_r0^(r q),   A_q = FFT_Q(a_q),   X[q + r0 kappa] = A_q[kappa]
//   -- bins beyond W / 2 are the mirrors of the bins r0 - q + r0 (Q - 1 - kappa), so unit q delivers |X| for every bin = +-q mod r0 -- and
//   the bins = 0 mod r0 / 2 come from ONE more transform of Q points ("packed unit"): u[n] = sum_(r < r0/2) y[n + 2 Q r] is real and 2 Q long,
//       v[k] = u[2 k] + i u[2 k + 1],   V = FFT_Q(v),   X[(r0/2) j] = E + W_(2Q)^j O   with E, O from V[j] and V[Q - j].
//   r0 / 2 units of equal cost per frame, NONE needs another one's outputs.  (scripts/dev/wgs_model.py restates this in NumPy.)
//
//   A TASK is one frame and two units -- {1, 2}, {3, 4}, {5, packed} for r0 = 12; {1, 2}, {packed} for r0 = 6 -- handled by a workgroup of
//   384 threads, three waves per unit; persistent workgroups take tasks from a counter.
//   stage 0 : all threads: k = tid + 384 i: the r0 samples y[k + Q r] (one coalesced 2-byte load per r), normalised (:567-570); the DFT over r
//             in difference form (equal samples give exact zeros: a digitally silent frame keeps its exact spectrum) for BOTH units of the
//             task, times W_W^(q k) (powers of ONE table value) -> a_q[k] into the unit's LDS buffer (natural order; rows of 525 padded to 535)
//   pass 1  : thread j, j + 175, j + 350 (525 jobs): radix 7 over n0 of a[j + 525 n0], times W_Q^(j k0), back IN PLACE
//   pass 2  : thread (n2, k0): radix 21 = 3 x 7 (kernels_tri.hpp) over n1 of (k0, n1, n2), times W_525^(n2 k1), in place
//   pass 3  : thread (k0, k1): radix 25 (kernels_fast.hpp) over n2 -> A[k0 + 7 k1 + 147 k2] in registers:
//             complex units: |A| / Nf straight to the frame's row (bin q + r0 kappa or its mirror); packed unit: one more exchange (natural
//             order), then the recombination of the pairs (j, Q - j)
//   Four workgroup barriers per task (six with a packed unit) against the seven of five in-place passes; 60 vector instructions per point.
//   The time-domain features (:22-51) of a frame are formed by the task that holds units {1, 2}: its stage 0 sees every sample of the frame.
// Features: kernels_wg.hpp's wg_feat_kernel on the rows, as before.
// Replaces ShortTermFeatures.py:608-682 (transform part), spectrogram (:415-422), chromagram (:349-359) for these windows.
#pragma once
#include "kernels_tri.hpp"          // tri::Cd<21>

namespace paa {
namespace wgs {

constexpr int R1 = 7, R2 = 21, R3 = 25;
constexpr int Q = R1 * R2 * R3;                 // 3675 points per unit
constexpr int J1 = R2 * R3, J2 = R1 * R3, J3 = R1 * R2;      // 525 / 175 / 147 lane jobs
constexpr int A = 535;                          // row pitch of the exchange buffer: element (k0, n1, n2) at k0 A + 25 n1 + n2 (scripts/dev/wgs_model.py
                                                // --lds: every ds_read_b128 / ds_write_b128 of the three passes at the conflict-free count but the
                                                // pass-3 reads, 450 LDS cycles against 250)
constexpr int UNIT_ELEMS = R1 * A;              // 3745 double2
constexpr int TU = 192, NT = 2 * TU;            // threads per unit / workgroup
constexpr int NWV = NT / 64;                     // six waves
constexpr int OFF_MISC = 2 * UNIT_ELEMS * 16;   // red [6][12] doubles (time domain: ten block energies, sign changes), next task
constexpr int LDS_BYTES = OFF_MISC + NWV * 12 * 8 + 16;
static_assert(LDS_BYTES <= 160 * 1024, "one workgroup's LDS");

__device__ __forceinline__ int pos_of(int k) { return k + (A - J1) * (int)__umulhi((unsigned)k, 8181136u); }      // k / 525 for k < 2^16 (ceil(2^32 / 525))
static_assert(((unsigned long long)8181136u * 525ull) >> 32 == 1 && ((unsigned long long)8181136u * 524ull) >> 32 == 0, "magic of 525");

__device__ __forceinline__ double2 csqr(double2 a) { return make_double2(fma(a.x, a.x, -a.y * a.y), 2.0 * (a.x * a.y)); }

// sum_r d_r W_6^r of three REAL values d0 + d1 W6 + d2 W6^2, from d0, d1 - d2, d1 + d2; sum_r g_r W_3^r from g0, g1 + g2, g1 - g2 -- equal g
// give an exact zero (2 g / 2 is exact)
constexpr double kS60 = 0.86602540378443864676;
__device__ __forceinline__ double2 dw6(double d0, double dm, double dp) { return make_double2(fma(0.5, dm, d0), -kS60 * dp); }
__device__ __forceinline__ double2 dw3(double g0, double gp, double gm) { return make_double2(fma(-0.5, gp, g0), -kS60 * gm); }

// the DFT over r of the R0 samples s[r] = x[k + Q r] at q = QQ (1 <= QQ < R0 / 2), in DIFFERENCE form: the sums and differences are formed
// in the samples' own type N -- int for the integer sample types (exact, a quarter of the FP64 issue cost), double for float64 --, so R0
// equal samples give exact zeros, and the clip mean never enters (sum_r W_r0^(r q) = 0): the result scales by sample_scale x inv afterwards
template <int R0, int QQ, typename N>
__device__ __forceinline__ double2 split_dft(const N *s) {
    auto D = [](N v) { return (double)v; };
    if constexpr (R0 == 6) {
        if constexpr (QQ == 1) { const N d0 = s[0] - s[3], d1 = s[1] - s[4], d2 = s[2] - s[5]; return dw6(D(d0), D(d1 - d2), D(d1 + d2)); }
        else { const N g0 = s[0] + s[3], g1 = s[1] + s[4], g2 = s[2] + s[5]; return dw3(D(g0), D(g1 + g2), D(g1 - g2)); }
    } else {
        static_assert(R0 == 12, "r0 = 6 or 12");
        if constexpr (QQ == 2) {
            const N d0 = (s[0] + s[6]) - (s[3] + s[9]), d1 = (s[1] + s[7]) - (s[4] + s[10]), d2 = (s[2] + s[8]) - (s[5] + s[11]);
            return dw6(D(d0), D(d1 - d2), D(d1 + d2));
        } else if constexpr (QQ == 4) {
            const N g0 = (s[0] + s[6]) + (s[3] + s[9]), g1 = (s[1] + s[7]) + (s[4] + s[10]), g2 = (s[2] + s[8]) + (s[5] + s[11]);
            return dw3(D(g0), D(g1 + g2), D(g1 - g2));
        } else {
            const N o0 = s[0] - s[6], o1 = s[1] - s[7], o2 = s[2] - s[8], o3 = s[3] - s[9], o4 = s[4] - s[10], o5 = s[5] - s[11];
            if constexpr (QQ == 3) return make_double2(D((o0 - o2) + o4), -D((o1 - o3) + o5));
            else if constexpr (QQ == 1)
                return make_double2(fma(kS60, D(o1 - o5), fma(0.5, D(o2 - o4), D(o0))), -fma(kS60, D(o2 + o4), fma(0.5, D(o1 + o5), D(o3))));
            else
                return make_double2(fma(-kS60, D(o1 - o5), fma(0.5, D(o2 - o4), D(o0))), -fma(-kS60, D(o2 + o4), fma(0.5, D(o1 + o5), D(o3))));
        }
    }
}
// one sample as an integer (int16 PCM; L + R of an interleaved stereo frame)
template <typename T> struct IntSample { static constexpr bool kInt = false; };
template <> struct IntSample<int16_t> {
    static constexpr bool kInt = true;
    static __device__ __forceinline__ int get(const int16_t *p) { return (int)(*p); }
};
template <> struct IntSample<stereo16> {
    static constexpr bool kInt = true;
    static __device__ __forceinline__ int get(const stereo16 *p) { return stereo_word_sum(*reinterpret_cast<const int *>(p)); }
};
// lane l receives the value of lane l - 1 (lane 0: `first`)
__device__ __forceinline__ int shr1(int v, int first) { return __builtin_amdgcn_update_dpp(first, v, 0x138, 0xF, 0xF, false); }
// w^QQ from w, at most three multiplications deep
template <int QQ>
__device__ __forceinline__ double2 cpow(double2 w) {
    if constexpr (QQ == 1) return w;
    else if constexpr (QQ == 2) return csqr(w);
    else if constexpr (QQ == 3) return cmul(csqr(w), w);
    else if constexpr (QQ == 4) return csqr(csqr(w));
    else return cmul(csqr(csqr(w)), w);
}

// task types (FrameRef::halo >> 8).  R0 = 12: 0 = units {1, 2} (+ the time-domain features), 1 = {3, 4}, 2 = {5, packed};
// R0 = 6: 0 = {1, 2} (+ time domain), 1 = {packed} (the first three waves idle)
template <int R0> __host__ __device__ constexpr int task_types() { return R0 == 12 ? 3 : 2; }

template <typename T, int R0>
__global__ __launch_bounds__(NT) void wgs_kernel(PlanDev P, const T *__restrict__ sig, const ClipDev *__restrict__ clips,
                                                 const ClipNorm *__restrict__ norms, const wg::FrameRef *__restrict__ tasks, int n_tasks,
                                                 int *next_task, double *__restrict__ spec, double *__restrict__ tfeat,
                                                 double *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double2 *buf_all = reinterpret_cast<double2 *>(smem);
    double *red = reinterpret_cast<double *>(smem + OFF_MISC);          // [6][12]
    int *s_next = reinterpret_cast<int *>(smem + OFF_MISC + NWV * 12 * 8);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int unit = __builtin_amdgcn_readfirstlane(tid >= TU ? 1 : 0);
    const int tu = tid - unit * TU;
    double2 *buf = buf_all + unit * UNIT_ELEMS;
    const int W = P.W, Nf = P.Nf;
    const double sc = sample_scale<T>();
    const double invNf = 1.0 / (double)Nf;
    constexpr int H0 = R0 / 2;                          // W_Q^j = tw[H0 j] (tw: W_Nc, Nc = H0 Q), W_(2Q)^j = post[H0 j] (post: W_W)
    struct Cur { long long x_off, out_off; double mean, inv; int t, row, type, halo; };
    auto fetch = [&](int ti) {
        const wg::FrameRef r = tasks[ti];
        const ClipDev cd = clips[r.clip];
        const ClipNorm n_ = norms[r.clip];
        Cur c;
        c.x_off = cd.sample_off + P.frame_origin + (long long)r.t * P.S;
        c.out_off = cd.out_off; c.mean = n_.mean; c.inv = n_.inv; c.t = r.t; c.row = r.row; c.type = r.halo >> 8; c.halo = r.halo & 1;
        return c;
    };
    if ((int)blockIdx.x >= n_tasks) return;
    Cur cu = fetch(blockIdx.x);
    for (int ti = blockIdx.x; ti < n_tasks;) {
        const T *x = sig + cu.x_off;
        const double mean = cu.mean, inv = cu.inv;
        const int type = __builtin_amdgcn_readfirstlane(cu.type);
        const bool packed_task = (type == task_types<R0>() - 1);             // unit b is the packed one
        const bool a_on = !(R0 == 6 && packed_task);                          // (R0 = 6: the packed unit is alone in its task)
        const bool time_task = (type == 0) && P.mode == 0 && !cu.halo;
        double *row = (P.mode == 1) ? out + cu.out_off + (long long)cu.t * Nf : spec + (long long)cu.row * Nf;
        __syncthreads();          // the previous task's reads of the buffers are done
        if (tid == 64) *s_next = (int)gridDim.x + atomicAdd(next_task, 1);
        // ---------------- stage 0: a_q[k] of both units from the samples (+ the time-domain features :22-51 in the {1, 2} task)
        {
            constexpr bool INT_T = IntSample<T>::kInt;
            typedef typename std::conditional<INT_T, int, double>::type N;
            // integer samples: y = (x - mu) sc inv with mu = mean / sc in counts; x' = x - m_int is a whole number, dmu = mu - m_int
            const double ys = INT_T ? sc * inv : 1.0;
            const double mu = mean * (1.0 / sc);
            const double mi_d = __builtin_rint(mu), dmu = mu - mi_d;
            const int m_int = (int)mi_d;
            SignRule sr = {0, 0, 0};
            if constexpr (INT_T) sr = sign_rule<T>(mean);
            auto sample = [&](const T *p) -> N {
                if constexpr (INT_T) return IntSample<T>::get(p);
                else return fma(*p, sc, -mean) * inv;
            };
            auto stage0 = [&](auto qa_c, auto qb_c, auto time_c) {
                constexpr int QA = decltype(qa_c)::value, QB = decltype(qb_c)::value;           // 0 = none / packed
                constexpr bool TIME = decltype(time_c)::value;
                double eb[10];
                int zc = 0;
#pragma unroll
                for (int b = 0; b < 10; ++b) eb[b] = 0.0;
#pragma unroll 2
                for (int i = 0; i < (Q + NT - 1) / NT; ++i) {          // (the same trip count for every thread; the last round's spare threads shadow k = Q - 1)
                    const int k0 = tid + NT * i;
                    const bool in = k0 < Q;
                    const int k = in ? k0 : Q - 1;
                    N s[R0];
                    N sl[R0];          // TIME: lane 0's left neighbours x[k - 1 + Q r] (the other lanes take them from the lane below)
                    if constexpr (QA > 0) {
#pragma unroll
                        for (int r = 0; r < R0; ++r) s[r] = sample(x + k + Q * r);
                        if constexpr (TIME) {
#pragma unroll
                            for (int r = 0; r < R0; ++r) sl[r] = s[r];
                            if (lane == 0) {
#pragma unroll
                                for (int r = 0; r < R0; ++r) sl[r] = sample(x + max(k - 1 + Q * r, 0));          // (the frame's first sample meets itself)
                            }
                        }
                    }
                    const double2 w = P.post[k];
                    const int pk = pos_of(k);
                    if constexpr (QA > 0) {
                        const double2 d = split_dft<R0, QA, N>(s);
                        const double2 va = cmul(make_double2(d.x * ys, d.y * ys), cpow<QA>(w));
                        if (in) buf_all[pk] = va;
                    }
                    if constexpr (QB > 0) {
                        const double2 d = split_dft<R0, QB, N>(s);
                        const double2 vb = cmul(make_double2(d.x * ys, d.y * ys), cpow<QB>(w));
                        if (in) buf_all[UNIT_ELEMS + pk] = vb;
                    } else {
                        // packed unit: v[k] = (u[2 k], u[2 k + 1]), u[n] = sum_(r < R0 / 2) y[n + 2 Q r]
                        double2 v;
                        if constexpr (INT_T) {
                            int ux = 0, uy = 0;
#pragma unroll
                            for (int r = 0; r < H0; ++r) {
                                const ct::PairRaw<T> xx = ct::PairRaw<T>::get(x + 2 * k + 2 * Q * r);
                                ux += xx.x0(); uy += xx.x1();
                            }
                            const double c0 = -(double)H0 * mean * inv;
                            v = make_double2(fma((double)ux, ys, c0), fma((double)uy, ys, c0));
                        } else {
                            v = make_double2(0.0, 0.0);
#pragma unroll
                            for (int r = 0; r < H0; ++r) {
                                const double2 xx = ct::PairLoad<T>::get(x + 2 * k + 2 * Q * r);
                                v.x += fma(xx.x, sc, -mean) * inv;
                                v.y += fma(xx.y, sc, -mean) * inv;
                            }
                        }
                        if (in) buf_all[UNIT_ELEMS + pk] = v;
                    }
                    if constexpr (TIME) {
                        // energies of the ten entropy blocks (LT = W / 10 samples): register row r covers the samples [Q r, Q r + Q), i.e. the
                        // blocks (Q r) / LT .. (Q r + Q - 1) / LT -- static per row --; sign changes against the sample before
                        constexpr int LT = R0 * Q / 10;
#pragma unroll
                        for (int r = 0; r < R0; ++r) {
                            double t;
                            int c, cl;
                            if constexpr (INT_T) {
                                t = (double)(s[r] - m_int) - dmu;
                                c = sgn1(s[r], sr); cl = sgn1(sl[r], sr);
                            } else {
                                t = s[r];
                                c = sgn1(s[r]); cl = sgn1(sl[r]);
                            }
                            const double sq = in ? t * t : 0.0;
                            const int b0 = (Q * r) / LT, b1 = (Q * r + Q - 1) / LT;          // (compile-time after unrolling)
                            if (b1 == b0) eb[b0] += sq;
                            else {
                                const int kb1 = (b0 + 1) * LT - Q * r;
                                const double lo = (k < kb1) ? sq : 0.0;
                                eb[b0] += lo;
                                if (b1 == b0 + 1) eb[b0 + 1] += sq - lo;
                                else {
                                    const int kb2 = (b0 + 2) * LT - Q * r;
                                    const double hi = (k >= kb2) ? sq : 0.0;
                                    eb[b0 + 2] += hi;
                                    eb[b0 + 1] += (sq - lo) - hi;
                                }
                            }
                            const int left = shr1(c, cl);
                            if (in) sad_acc(zc, c, left);
                        }
                    }
                }
                if constexpr (TIME) {
#pragma unroll
                    for (int b = 0; b < 10; ++b) { const double e = wsum(eb[b]); if (lane == 0) red[12 * wave + b] = e; }
                    const int z = wsum_i(zc);
                    if (lane == 0) red[12 * wave + 10] = (double)(z << (INT_T ? sr.sh : 0));
                }
            };
            typedef std::integral_constant<int, 0> I0;
            typedef std::integral_constant<int, 1> I1;
            typedef std::integral_constant<int, 2> I2;
            if (type == 0) {
                if (time_task) stage0(I1(), I2(), std::true_type());
                else stage0(I1(), I2(), std::false_type());
            } else if constexpr (R0 == 12) {
                if (type == 1) stage0(std::integral_constant<int, 3>(), std::integral_constant<int, 4>(), std::false_type());
                else stage0(std::integral_constant<int, 5>(), I0(), std::false_type());
            } else {
                stage0(I0(), I0(), std::false_type());
            }
            __syncthreads();
            if (time_task && wave == NWV - 1) {
                // block j (lane j < 10) = the waves' parts in wave order; energy, entropy of the block shares (:29-51), sign changes -> tfeat
                double E = 0.0, zct = 0.0;
                const int lj = lane < 10 ? lane : 0;
#pragma unroll
                for (int w = 0; w < NWV; ++w) { E += red[12 * w + lj]; zct += red[12 * w + 10]; }
                E *= ys * ys;
                const double e_tot = wsum((lane < 10) ? E : 0.0);
                const double sh = fast_div(E, e_tot + kEps);
                const double ent = wsum((lane < 10) ? -(sh * fast_log2(sh + kEps)) : 0.0);
                double *tfp = tfeat + 3 * (long long)cu.row;
                if (lane == 0) { tfp[0] = e_tot; tfp[1] = ent; tfp[2] = zct; }
            }
        }
        const bool u_on = unit == 1 || a_on;
        // ---------------- pass 1: radix 7 over n0 for the jobs j = tu, tu + 175, tu + 350; outputs times W_Q^(j k0); in place
        if (u_on && tu < J2) {
            double2 v[3][R1];
            double2 w[3];
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int j = tu + J2 * s;
                w[s] = P.tw[H0 * j];
#pragma unroll
                for (int n0 = 0; n0 < R1; ++n0) v[s][n0] = buf[n0 * A + j];
            }
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int j = tu + J2 * s;
                mix::dft_prime<R1>(v[s]);
                const double2 w2 = csqr(w[s]), w3 = cmul(w2, w[s]);
                const double2 w4 = csqr(w2), w5 = cmul(w3, w2), w6 = csqr(w3);
                buf[j] = v[s][0];
                buf[A + j] = cmul(v[s][1], w[s]);
                buf[2 * A + j] = cmul(v[s][2], w2);
                buf[3 * A + j] = cmul(v[s][3], w3);
                buf[4 * A + j] = cmul(v[s][4], w4);
                buf[5 * A + j] = cmul(v[s][5], w5);
                buf[6 * A + j] = cmul(v[s][6], w6);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
        // ---------------- pass 2: thread (n2, k0) = tu: radix 21 over n1, outputs times W_525^(n2 k1) = W_Q^(7 n2 k1); in place
        if (u_on && tu < J2) {
            const int n2 = tu / R1, k0 = tu - n2 * R1;
            double2 *e = buf + k0 * A + n2;
            const double2 w = P.tw[H0 * R1 * n2];
            double2 v[R2];
#pragma unroll
            for (int n1 = 0; n1 < R2; ++n1) v[n1] = e[n1 * R3];
            tri::Cd<R2>::run(v);
            // W^k1, k1 = 1 .. 20: squares and products, at most five multiplications deep
            double2 wq[R2];
            wq[1] = w;
#pragma unroll
            for (int q = 2; q < R2; ++q) wq[q] = (q % 2 == 0) ? csqr(wq[q / 2]) : cmul(wq[q / 2], wq[q - q / 2]);
            e[0] = v[tri::Cd<R2>::pos(0)];
#pragma unroll
            for (int k1 = 1; k1 < R2; ++k1) e[k1 * R3] = cmul(v[tri::Cd<R2>::pos(k1)], wq[k1]);
        }
        __syncthreads();
        // ---------------- the next task of this workgroup: its records now, a touch of its samples
        Cur cu_next = cu;
        const int t_next = *s_next;
        if (t_next < n_tasks) {
            cu_next = fetch(t_next);
            const char *xn = reinterpret_cast<const char *>(sig + cu_next.x_off);
            const int bytes = W * (int)sizeof(T);
            int touch = 0;
            for (int o = tid * 128; o < bytes; o += NT * 128) touch += *reinterpret_cast<const volatile char *>(xn + o);
            asm volatile("" ::"v"(touch));
        }
        // ---------------- pass 3: thread (k0, k1) = tu: radix 25 over n2 -> A[k0 + 7 k1 + 147 k2]
        {
            const bool a3 = u_on && tu < J3;
            const int u3 = a3 ? tu : J3 - 1;
            const int k0 = u3 / R2, k1 = u3 - k0 * R2;
            double2 v[R3];
            {
                const double2 *e = buf + k0 * A + k1 * R3;
#pragma unroll
                for (int n2 = 0; n2 < R3; ++n2) v[n2] = e[n2];
            }
            ct::Dft<R3>::template run<1>(v);
            const int kap0 = k0 + R1 * k1;
            const bool packed_unit = packed_task && unit == 1;
            if (!packed_unit) {
                // complex unit q: bin q + R0 kappa, or its mirror W - (q + R0 kappa)
                const int q = (R0 == 12) ? 2 * type + 1 + unit : 1 + unit;
                if (a3) {
#pragma unroll
                    for (int k2 = 0; k2 < R3; ++k2) {
                        const double2 z = v[ct::Dft<R3>::pos(k2)];
                        const int m = q + R0 * (kap0 + J3 * k2);
                        row[m < Nf ? m : W - m] = mag_sqrt(fma(z.x, z.x, z.y * z.y)) * invNf;
                    }
                }
            }
            if (packed_task) {          // (workgroup-uniform)
                __syncthreads();          // every thread of the packed unit has read its pass-3 inputs
                if (packed_unit && a3) {
#pragma unroll
                    for (int k2 = 0; k2 < R3; ++k2) buf[kap0 + J3 * k2] = v[ct::Dft<R3>::pos(k2)];
                }
                __syncthreads();
                if (packed_unit) {
                    // pairs (j, Q - j), j = 1 .. (Q - 1) / 2, and j = 0: X[H0 j] = E + w^j O, X[H0 (Q - j)] = conj(E - w^j O); X[0] = Re V[0] + Im V[0]
                    constexpr int NP = (Q - 1) / 2 + 1;
#pragma unroll 2
                    for (int i = 0; i < (NP + TU - 1) / TU; ++i) {
                        const int j0 = tu + TU * i;
                        const bool in = j0 < NP;
                        const int j = in ? j0 : NP - 1;
                        const double2 pw = P.post[H0 * j];
                        const double2 zk = buf[j], zm = buf[j == 0 ? 0 : Q - j];
                        const double2 e = make_double2(0.5 * (zk.x + zm.x), 0.5 * (zk.y - zm.y));
                        const double2 o = make_double2(0.5 * (zk.y + zm.y), 0.5 * (zm.x - zk.x));
                        const double2 wo = cmul(pw, o);
                        const double ar = e.x + wo.x, ai = e.y + wo.y, br = e.x - wo.x, bi = e.y - wo.y;
                        if (in) {
                            row[H0 * j] = mag_sqrt(fma(ar, ar, ai * ai)) * invNf;
                            if (j > 0) row[H0 * (Q - j)] = mag_sqrt(fma(br, br, bi * bi)) * invNf;
                        }
                    }
                }
            }
        }
        cu = cu_next;
        ti = t_next;
    }
}

// ---- host -------------------------------------------------------------------------------------------------------------------------
// r0 of a window this family takes (0: none)
inline int wgs_r0(int window) { return window == 12 * Q ? 12 : (window == 6 * Q ? 6 : 0); }

}  // namespace wgs
}  // namespace paa
