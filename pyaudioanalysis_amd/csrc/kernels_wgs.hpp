// Long even windows of r0 x Q samples, Q = 3675 or 4000 -- the 1 s window audioSegmentation.music_thumbnailing passes by default
// (audioSegmentation.py:1134-1138) at the rates music has: 44 100 samples at 44.1 kHz = 12 x 3675, 22 050 = 6 x 3675, 48 000 = 12 x 4000, 32 000 =
// 8 x 4000, 24 000 = 6 x 4000.  Their packed transform (22 050 complex points = 353 KB) does not fit one CU's LDS.  kernels_wg.hpp (round 5) splits the
// PACKED sequence by r0 = 6 / 3 and has to keep the sub-transforms q and r0 - q side by side (the real-FFT recombination pairs bin k with Nc - k), runs
// five in-place radix passes over them and leaves the time-domain features to a kernel of their own.  Here (round 6) the REAL
// sequence is split instead:
//
//   W = r0 Q, Q = 3675 = 7 x 21 x 25 (or 4000 = 8 x 20 x 25).  For q = 1 .. r0/2 - 1 ("complex unit q")
//       a_q[k] = W_W^(q k) sum_r y[k + Q r] W_r0^(r q),   A_q = FFT_Q(a_q),   X[q + r0 kappa] = A_q[kappa]
//   -- bins beyond W / 2 are the mirrors of the bins r0 - q + r0 (Q - 1 - kappa), so unit q delivers |X| for every bin = +-q mod r0 -- and
//   the bins = 0 mod r0 / 2 come from ONE more transform of Q points ("packed unit"): u[n] = sum_(r < r0/2) y[n + 2 Q r] is real and 2 Q long,
//       v[k] = u[2 k] + i u[2 k + 1],   V = FFT_Q(v),   X[(r0/2) j] = E + W_(2Q)^j O   with E, O from V[j] and V[Q - j].
//   r0 / 2 units of equal cost per frame, NONE needs another one's outputs.  (scripts/dev/wgs_model.py and tests/test_abi_cpu.py restate this in NumPy.)
//
//   wgs_kernel: a TASK is one frame and two units -- {1, 2}, {3, 4}, {5, packed} for r0 = 12; {1, 2}, {3, packed} for r0 = 8; {1, 2} and the packed
//   units of two consecutive frames for r0 = 6 -- handled by a workgroup of 512 threads, four waves per unit (passes 2 and 3 use three of them);
//   persistent workgroups take tasks from one counter per XCD (the tasks of a frame, and frames that overlap, read their samples through one L2).
//   stage 0 : all threads: k = tid + 512 i: the r0 samples x[k + Q r] (one coalesced 2-byte load per r, four rounds requested together); the DFT
//             over r in difference form ON INTEGERS (exact; equal samples give exact zeros: a digitally silent frame keeps its exact spectrum; the
//             clip mean cancels) for BOTH units of the task, scaled (:567-570), times W_W^(q k) (powers of ONE table value) -> a_q[k] into the
//             unit's LDS buffer (natural order; 3675: rows of 525 padded to 535).  The {1, 2} task also forms the frame's time-domain features
//             (:22-51) here: its stage 0 sees every sample of the frame
//   pass 1  : thread j, j + J1T, ... : radix 7 / 8 over n0 of a[j + R2 R3 n0], times W_Q^(j k0), back IN PLACE
//   pass 2  : thread (n2, k0): radix 21 = 3 x 7 / 20 = 4 x 5 (kernels_tri.hpp) over n1 of (k0, n1, n2), times W_(R2 R3)^(n2 k1), in place
//   pass 3  : thread (k1, k0): radix 25 (kernels_fast.hpp) over n2 -> A[k0 + R1 k1 + R1 R2 k2] in registers: complex units: |A| / Nf straight to the
//             frame's row, UNIT-MAJOR (unit q's Q magnitudes side by side: R1 R2 consecutive doubles per store instruction; in natural order the same
//             stores were 8 bytes every 96: 44 100 write transactions per task = 106 of 182 us) together with the unit's sum X, sum (k + 1) X, max X;
//             packed unit: one more exchange (natural order), then the recombination of the pairs (j, Q - j).  Spectrogram plans: natural order
//   Four workgroup barriers per task (six with a packed unit) against the seven of five in-place passes; 60 vector instructions per point.
//   wgs_feat_kernel: the features of a frame from its unit-major row and the previous frame's, read as residue streams (below).
// Replaces ShortTermFeatures.py:608-682 (+ helpers :22-140, :236-321), spectrogram (:415-422), chromagram (:349-359) for these windows.
#pragma once
#include "kernels_tri.hpp"          // tri::Cd<21>; kernels_generic.hpp: Tabs

namespace paa {
namespace wgs {

#ifndef PAA_WGS_ABLATE
#define PAA_WGS_ABLATE 0           // timing builds only (scripts/rounds/r06/gpu_r06y.sh): bit mask of phases that are skipped -- 1 stage 0, 2 pass 1,
                                   // 4 pass 2, 8 pass 3 + magnitudes, 16 the next task's touch; feature kernel: 32 sweep A, 64 sweep B, 128 roll-off, 256 mel, 512 chroma; 1024: no time-domain features in stage 0
#endif
constexpr int kAblate = PAA_WGS_ABLATE;
// first-pass codelets (run() transforms v[] in place, X[q] ends at v[pos(q)])
template <int R> struct C1;
template <> struct C1<7> {
    static __device__ __forceinline__ void run(double2 *v) { mix::dft_prime<7>(v); }
    static constexpr int pos(int q) { return q; }
};
template <> struct C1<8> {
    static __device__ __forceinline__ void run(double2 *v) { tri::Cd<8>::run(v); }
    static constexpr int pos(int q) { return tri::Cd<8>::pos(q); }
};
// A sub-transform of Q = R1 R2 R3 points.  A: row pitch of the exchange buffer, element (k0, n1, n2) at k0 A + R3 n1 + n2 (scripts/dev/wgs_model.py
// lds ...: every ds_read_b128 / ds_write_b128 of the three passes at or near the conflict-free count); TU: threads per unit; pass 1: thread tu < J1T takes
// the jobs tu, tu + J1T, ... (JPT of them); P2K0: pass-2 thread (k0, n2) = tu with n2 fastest (else k0 fastest)
template <int R1_, int R2_, int R3_, int A_, int TU_, int JPT_, int J1T_, bool P2K0_>
struct Shape {
    static constexpr int R1 = R1_, R2 = R2_, R3 = R3_, A = A_, TU = TU_, JPT = JPT_, J1T = J1T_;
    static constexpr bool P2K0 = P2K0_;
    static constexpr int Q = R1 * R2 * R3;
    static constexpr int J1 = R2 * R3, J2 = R1 * R3, J3 = R1 * R2;          // lane jobs of the three passes
    static constexpr int UNIT_ELEMS = R1 * A;              // double2
    static constexpr int NT = 2 * TU, NWV = NT / 64;
    static constexpr int OFF_MISC = 2 * UNIT_ELEMS * 16;   // red [NWV][12] doubles (time domain: ten block energies, sign changes), next task
    static constexpr int OFF_UP = OFF_MISC + NWV * 12 * 8 + 16;          // double [NWV][4]: the waves' parts of their unit's sum X, sum (k + 1) X, max X
    static constexpr int LDS_BYTES = OFF_UP + NWV * 4 * 8;
    static_assert(JPT * J1T == J1 && J1T <= TU && J2 <= TU && J3 <= TU && A >= J1 && TU % 64 == 0, "lane jobs");
    static_assert(LDS_BYTES <= 160 * 1024, "one workgroup's LDS");
    static_assert(R3 == 25, "pass 3 is the radix-25 codelet");
    // pass-3 output k2 < R3 / 2 is a bin below W / 2 for every thread and q, k2 > R3 / 2 a mirror (kappa = u3 + J3 k2 against Q / 2)
    static_assert(J3 - 1 + J3 * (R3 / 2 - 1) < Q / 2 - 1 && J3 * (R3 / 2 + 1) > Q / 2, "mirror split at k2 = R3 / 2");
    static __device__ __forceinline__ int pos_of(int k) { return k + (A - J1) * (int)((unsigned)k / (unsigned)J1); }
};
#ifndef PAA_WGS_TU3675
#define PAA_WGS_TU3675 256         // threads per unit of the 3675-point shape: four waves (192 -- three, what passes 2 and 3 need -- is 4-6 % slower:
                                   // stage 0 takes ten rounds of the workgroup instead of eight; A/B scripts/rounds/r06/gpu_r06ab.sh)
#endif
typedef Shape<7, 21, 25, 535, PAA_WGS_TU3675, 3, 175, false> S3675;       // 44 100 = 12 x 3675, 22 050 = 6 x 3675
typedef Shape<8, 20, 25, 500, 256, 2, 250, true> S4000;        // 48 000 = 12 x 4000, 32 000 = 8 x 4000, 24 000 = 6 x 4000

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
    } else if constexpr (R0 == 8) {
        constexpr double h = 0.70710678118654752440;
        if constexpr (QQ == 2) return make_double2(D((s[0] + s[4]) - (s[2] + s[6])), -D((s[1] + s[5]) - (s[3] + s[7])));
        else {
            const N o0 = s[0] - s[4], o1 = s[1] - s[5], o2 = s[2] - s[6], o3 = s[3] - s[7];
            if constexpr (QQ == 1) return make_double2(fma(h, D(o1 - o3), D(o0)), -fma(h, D(o1 + o3), D(o2)));
            else return make_double2(fma(-h, D(o1 - o3), D(o0)), fma(-h, D(o1 + o3), D(o2)));
        }
    } else {
        static_assert(R0 == 12, "r0 = 6, 8 or 12");
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

// task types (FrameRef::halo >> 8).  R0 = 12: 0 = units {1, 2} (+ the time-domain features), 1 = {3, 4}, 2 = {5, packed}; R0 = 8: 0 = {1, 2}, 1 = {3, packed};
// R0 = 6: 0 = {1, 2} (+ time domain), 1 = the packed units of THIS frame and of the NEXT one of the clip (the next row), 2 = {packed} alone (a
// frame without a partner: the first half of the workgroup idle)
template <int R0> __host__ __device__ constexpr int task_types() { return (R0 == 12 || R0 == 6) ? 3 : 2; }

template <typename T, int R0, typename SH>
__global__ __launch_bounds__(SH::NT) void wgs_kernel(PlanDev P, const T *__restrict__ sig, const ClipDev *__restrict__ clips,
                                                 const ClipNorm *__restrict__ norms, const wg::FrameRef *__restrict__ tasks, int n_tasks,
                                                 int *next_task, double *__restrict__ spec, double *__restrict__ tfeat,
                                                 double *__restrict__ psum, double *__restrict__ out) {
    constexpr int R1 = SH::R1, R2 = SH::R2, R3 = SH::R3, Q = SH::Q, J2 = SH::J2, J3 = SH::J3, A = SH::A, TU = SH::TU, NT = SH::NT;
    constexpr int NWV = SH::NWV, UNIT_ELEMS = SH::UNIT_ELEMS, OFF_MISC = SH::OFF_MISC, OFF_UP = SH::OFF_UP, JPT = SH::JPT, J1T = SH::J1T;
    auto pos_of = [](int k) { return SH::pos_of(k); };
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double2 *buf_all = reinterpret_cast<double2 *>(smem);
    double *red = reinterpret_cast<double *>(smem + OFF_MISC);          // [NWV][12]
    int *s_next = reinterpret_cast<int *>(smem + OFF_MISC + NWV * 12 * 8);
    double *upart = reinterpret_cast<double *>(smem + OFF_UP);
    const int tid_ = threadIdx.x, lane = tid_ & 63, wave = __builtin_amdgcn_readfirstlane(tid_ >> 6);
    const int unit = __builtin_amdgcn_readfirstlane(tid_ >= TU ? 1 : 0);
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
    // XCD-aware hand-out: workgroup b runs on XCD b % 8; the task list -- frame after frame, the tasks of a frame side by side -- is cut into eight
    // segments of whole frames, workgroups take tasks of THEIR segment only (one counter per segment, zero at launch): the tasks of a frame,
    // and frames that overlap, read their samples through one L2
    constexpr int NTY = task_types<R0>();
    const int nseg = (gridDim.x % 8 == 0) ? 8 : 1, seg = blockIdx.x % nseg, rank = blockIdx.x / nseg, per = gridDim.x / nseg;
    const int n_fr = (n_tasks + NTY - 1) / NTY;          // (R0 = 6: groups of three tasks = two frames; a frame without a partner is a group of two)
    const int seg_lo = min(n_tasks, (int)((long long)n_fr * seg / nseg) * NTY), seg_hi = min(n_tasks, (int)((long long)n_fr * (seg + 1) / nseg) * NTY);
    const bool has_task = seg_lo + rank < seg_hi;
    // a unit's sum X, sum (k + 1) X and max X -- what the feature kernel needs BEFORE its sweep over the row (centroid, the normalisations of
    // spread and flux) -- are formed here from the magnitudes as they leave: psum [row][R0 / 2][4].  The waves of a task park their parts in LDS;
    // the row's entries are written behind the next barrier of the workgroup (the next task's first one, or the one after the loop)
    int flush_row = -1, flush_row_b = -1, flush_ua = -1, flush_ub = -1;
    auto flush_partials = [&](int rwa, int rwb, int ua, int ub) {
        if (tid_ < 2) {
            const int un = tid_ == 0 ? ua : ub, rw = tid_ == 0 ? rwa : rwb;
            if (un >= 0) {
                const double *u = upart + 4 * (TU / 64) * tid_;
                double *d = psum + ((long long)rw * H0 + un) * 4;
                double a0 = u[0], a1 = u[1], a2 = u[2];
#pragma unroll
                for (int w = 1; w < TU / 64; ++w) { a0 += u[4 * w]; a1 += u[4 * w + 1]; a2 = fmax(a2, u[4 * w + 2]); }
                d[0] = a0; d[1] = a1; d[2] = a2;
            }
        }
    };
    Cur cu = fetch(has_task ? seg_lo + rank : 0);
    for (int ti = has_task ? seg_lo + rank : seg_hi; ti < seg_hi;) {
        const T *x = sig + cu.x_off;
        const double mean = cu.mean, inv = cu.inv;
        const int type = __builtin_amdgcn_readfirstlane(cu.type);
        // (the job indices are formed again in every task from an opaque copy of the thread index: as loop invariants they -- and the addresses
        // derived from them -- were hoisted out of the task loop into registers that were then spilled)
        int tid = tid_; asm volatile("" : "+v"(tid));
        __builtin_assume(tid >= 0 && tid < NT);
        const int tu = tid - unit * TU;
        double2 *buf = buf_all + unit * UNIT_ELEMS;
        const bool pair_task = R0 == 6 && type == 1;                          // both units packed: this frame's and the next one's
        const bool packed_task = R0 == 6 ? type >= 1 : (type == task_types<R0>() - 1);             // unit b is a packed one
        const bool a_on = !(R0 == 6 && type == 2);                            // (a packed unit without a partner is alone in its task)
        const T *xb = pair_task ? x + P.S : x;
        const bool time_task = (type == 0) && P.mode == 0 && !cu.halo && !(kAblate & 1024);
        double *row = (P.mode == 1) ? out + cu.out_off + (long long)cu.t * Nf : spec + (long long)cu.row * Nf;
        if (pair_task && unit == 1) row += Nf;          // (the next frame of the clip: the next row of the scratch / of the output)
        __syncthreads();          // the previous task's reads of the buffers are done (and its waves' partial sums are in LDS)
        if (tid == 64) *s_next = seg_lo + per + atomicAdd(next_task + seg, 1);
        if (flush_row >= 0) { flush_partials(flush_row, flush_row_b, flush_ua, flush_ub); flush_row = -1; }
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
            auto stage0 = [&](auto qa_c, auto qb_c, auto time_c, auto pa_c) {
                constexpr int QA = decltype(qa_c)::value, QB = decltype(qb_c)::value;           // 0 = none / packed
                constexpr bool TIME = decltype(time_c)::value;
                constexpr bool PA = decltype(pa_c)::value;                                      // unit a is a packed unit too (of frame x; unit b: of frame xb)
                double eb[10];
                int zc = 0;
#pragma unroll
                for (int b = 0; b < 10; ++b) eb[b] = 0.0;
                // k = tid + NT i, i < NI (the same trip count for every thread; the last round's spare threads shadow k = Q - 1).  The loads of BI
                // rounds are requested TOGETHER (one exposed memory latency per batch instead of one per round: 71 of the kernel's 182 us were
                // this stage with two rounds in flight), then the rounds are worked off in order
                constexpr int NI = (Q + NT - 1) / NT;
                constexpr int B5 = (NI % 5 == 0) ? 5 : 4;
                constexpr int BI = INT_T ? B5 : ((TIME || QB == 0) ? 2 : B5);
                static_assert(NI % BI == 0, "whole batches");
                auto batch = [&](const int ib) {
                    N sa[BI][R0];
                    N sla[BI][R0];          // TIME: lane 0's left neighbours x[k - 1 + Q r] (the other lanes take them from the lane below)
                    double2 wa[BI];
                    typedef typename std::conditional<INT_T, ct::PairRaw<T>, double2>::type Pr;
                    Pr pra[BI][H0];
                    Pr praa[PA ? BI : 1][H0];          // unit a's pairs (pair task)
#pragma unroll
                    for (int u = 0; u < BI; ++u) {
                        const int k = min(tid + NT * (ib + u), Q - 1);
                        if constexpr (QA > 0) {
#pragma unroll
                            for (int r = 0; r < R0; ++r) sa[u][r] = sample(x + k + Q * r);
                        }
                        if constexpr (QB == 0) {
#pragma unroll
                            for (int r = 0; r < H0; ++r) {
                                if constexpr (INT_T) pra[u][r] = ct::PairRaw<T>::get(xb + 2 * k + 2 * Q * r);
                                else pra[u][r] = ct::PairLoad<T>::get(xb + 2 * k + 2 * Q * r);
                                if constexpr (PA) {
                                    if constexpr (INT_T) praa[u][r] = ct::PairRaw<T>::get(x + 2 * k + 2 * Q * r);
                                    else praa[u][r] = ct::PairLoad<T>::get(x + 2 * k + 2 * Q * r);
                                }
                            }
                        }
                        wa[u] = P.post[k];
                    }
                    if constexpr (TIME) {
#pragma unroll
                        for (int u = 0; u < BI; ++u)
#pragma unroll
                            for (int r = 0; r < R0; ++r) sla[u][r] = sa[u][r];
                        if (lane == 0) {
#pragma unroll
                            for (int u = 0; u < BI; ++u) {
                                const int k = min(tid + NT * (ib + u), Q - 1);
#pragma unroll
                                for (int r = 0; r < R0; ++r) sla[u][r] = sample(x + max(k - 1 + Q * r, 0));          // (the frame's first sample meets itself)
                            }
                        }
                    }
#pragma unroll
                    for (int u = 0; u < BI; ++u) {
                    const int k0 = tid + NT * (ib + u);
                    const bool in = k0 < Q;
                    const int k = in ? k0 : Q - 1;
                    const N *s = sa[u];
                    const N *sl = sla[u];
                    const double2 w = wa[u];
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
                        auto packed_value = [&](const Pr *pr) {
                            if constexpr (INT_T) {
                                int ux = 0, uy = 0;
#pragma unroll
                                for (int r = 0; r < H0; ++r) { ux += pr[r].x0(); uy += pr[r].x1(); }
                                const double c0 = -(double)H0 * mean * inv;
                                return make_double2(fma((double)ux, ys, c0), fma((double)uy, ys, c0));
                            } else {
                                double2 v = make_double2(0.0, 0.0);
#pragma unroll
                                for (int r = 0; r < H0; ++r) {
                                    v.x += fma(pr[r].x, sc, -mean) * inv;
                                    v.y += fma(pr[r].y, sc, -mean) * inv;
                                }
                                return v;
                            }
                        };
                        const double2 v = packed_value(pra[u]);
                        if (in) buf_all[UNIT_ELEMS + pk] = v;
                        if constexpr (PA) {
                            const double2 va = packed_value(praa[u]);
                            if (in) buf_all[pk] = va;
                        }
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
                            // (k0 = tid + NT (ib + u) with 0 <= tid < NT known: the comparisons below are decided at compile time for every round
                            // that does not straddle a block boundary / the end of the unit, and the selects fold into one fma)
                            const double tm = (k0 < Q) ? t : 0.0;
                            const int b0 = (Q * r) / LT, b1 = (Q * r + Q - 1) / LT;
                            if (b1 == b0) eb[b0] = fma(tm, tm, eb[b0]);
                            else {
                                const int kb1 = (b0 + 1) * LT - Q * r;
                                if (b1 == b0 + 1) {
                                    const double tl = (k0 < kb1) ? tm : 0.0, th = (k0 < kb1) ? 0.0 : tm;
                                    eb[b0] = fma(tl, tl, eb[b0]);
                                    eb[b0 + 1] = fma(th, th, eb[b0 + 1]);
                                } else {
                                    const int kb2 = (b0 + 2) * LT - Q * r;
                                    const double tl = (k0 < kb1) ? tm : 0.0, th = (k0 >= kb2) ? tm : 0.0, tc = (k0 >= kb1 && k0 < kb2) ? tm : 0.0;
                                    eb[b0] = fma(tl, tl, eb[b0]);
                                    eb[b0 + 1] = fma(tc, tc, eb[b0 + 1]);
                                    eb[b0 + 2] = fma(th, th, eb[b0 + 2]);
                                }
                            }
                            const int left = shr1(c, cl);
                            if (k0 < Q) sad_acc(zc, c, left);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);          // (one round at a time)
                    }
                };
                if constexpr (TIME && INT_T) {
                    // (unrolled: the round numbers are compile-time, so the range checks of every round that straddles nothing fold)
#pragma unroll
                    for (int ib = 0; ib < NI; ib += BI) batch(ib);
                } else {
#pragma unroll 1
                    for (int ib = 0; ib < NI; ib += BI) batch(ib);
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
            typedef std::false_type F_;
            if (kAblate & 1) {
                for (int k = tid; k < Q; k += NT) buf_all[pos_of(k)] = buf_all[UNIT_ELEMS + pos_of(k)] = make_double2(1e-3 * (double)(k & 7), 1e-3);
            } else if (type == 0) {
                if (time_task) stage0(I1(), I2(), std::true_type(), F_());
                else stage0(I1(), I2(), std::false_type(), F_());
            } else if constexpr (R0 == 12) {
                if (type == 1) stage0(std::integral_constant<int, 3>(), std::integral_constant<int, 4>(), std::false_type(), F_());
                else stage0(std::integral_constant<int, 5>(), I0(), std::false_type(), F_());
            } else if constexpr (R0 == 8) {
                stage0(std::integral_constant<int, 3>(), I0(), std::false_type(), F_());
            } else {
                if (pair_task) stage0(I0(), I0(), std::false_type(), std::true_type());
                else stage0(I0(), I0(), std::false_type(), F_());
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
        // ---------------- pass 1: radix R1 over n0 for the jobs j = tu, tu + J1T, ...; outputs times W_Q^(j k0); in place
        if (u_on && tu < J1T && !(kAblate & 2)) {
            double2 v[JPT][R1];
            double2 w[JPT];
#pragma unroll
            for (int s = 0; s < JPT; ++s) {
                const int j = tu + J1T * s;
                w[s] = P.tw[H0 * j];
#pragma unroll
                for (int n0 = 0; n0 < R1; ++n0) v[s][n0] = buf[n0 * A + j];
            }
#pragma unroll
            for (int s = 0; s < JPT; ++s) {
                const int j = tu + J1T * s;
                C1<R1>::run(v[s]);
                double2 t = w[s];
                buf[j] = v[s][C1<R1>::pos(0)];
#pragma unroll
                for (int k0 = 1; k0 < R1; ++k0) {          // W^k0: W, W^2, W^3 = W^2 W, ... (at most R1 - 2 multiplications deep)
                    buf[k0 * A + j] = cmul(v[s][C1<R1>::pos(k0)], t);
                    if (k0 + 1 < R1) t = (k0 == 1) ? csqr(w[s]) : cmul(t, w[s]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
        // ---------------- pass 2: thread (n2, k0) = tu: radix R2 over n1, outputs times W_(R2 R3)^(n2 k1) = W_Q^(R1 n2 k1); in place
        if (u_on && tu < J2 && !(kAblate & 4)) {
            const int n2 = SH::P2K0 ? tu % R3 : tu / R1, k0 = SH::P2K0 ? tu / R3 : tu % R1;
            double2 *e = buf + k0 * A + n2;
            const double2 w = P.tw[H0 * R1 * n2];
            double2 v[R2];
#pragma unroll
            for (int n1 = 0; n1 < R2; ++n1) v[n1] = e[n1 * R3];
            tri::Cd<R2>::run(v);
            // W^k1, k1 = 1 .. 20, as W^(4 m) W^j: at most seven multiplications deep, five values live (all twenty at once were 80 registers)
            const double2 w2 = csqr(w), w3 = cmul(w2, w), w4 = csqr(w2);
            double2 b4 = w4;
            e[0] = v[tri::Cd<R2>::pos(0)];
#pragma unroll
            for (int k1 = 1; k1 < R2; ++k1) {
                const int j = k1 & 3;
                const double2 wj = (j == 1) ? w : (j == 2) ? w2 : w3;
                const double2 t = (k1 < 4) ? wj : ((j == 0) ? b4 : cmul(b4, wj));
                e[k1 * R3] = cmul(v[tri::Cd<R2>::pos(k1)], t);
                if (j == 3 && k1 > 3) b4 = cmul(b4, w4);
            }
        }
        __syncthreads();
        // ---------------- the next task of this workgroup: its records now, a touch of its samples
        Cur cu_next = cu;
        const int t_next = *s_next;
        if (t_next < seg_hi) {
            cu_next = fetch(t_next);
            if (!(kAblate & 16)) {
            const char *xn = reinterpret_cast<const char *>(sig + cu_next.x_off);
            const int bytes = W * (int)sizeof(T);
            int touch = 0;
            for (int o = tid * 128; o < bytes; o += NT * 128) touch += *reinterpret_cast<const volatile char *>(xn + o);
            asm volatile("" ::"v"(touch));
            }
        }
        // ---------------- pass 3: thread (k0, k1) = tu: radix 25 over n2 -> A[k0 + 7 k1 + 147 k2]
        if (!(kAblate & 8)) {
            const bool a3 = u_on && tu < J3;
            const int u3 = a3 ? tu : J3 - 1;
            const int k1 = u3 / R1, k0 = u3 - k1 * R1;          // (k0 fastest: kappa = u3 + 147 k2, consecutive over the lanes)
            double2 v[R3];
            {
                const double2 *e = buf + k0 * A + k1 * R3;
#pragma unroll
                for (int n2 = 0; n2 < R3; ++n2) v[n2] = e[n2];
            }
            ct::Dft<R3>::template run<1>(v);
            const bool packed_unit = (packed_task && unit == 1) || pair_task;
            double us = 0.0, uw = 0.0, um = 0.0;          // this thread's part of its unit's sum X, sum (k + 1) X, max X
            if (!packed_unit) {
                const int q = 2 * type + 1 + unit;
                if (P.mode == 1) {
                    // spectrogram rows go straight to the output, natural order: bin q + R0 kappa, or its mirror W - (q + R0 kappa)
                    if (a3) {
#pragma unroll
                        for (int k2 = 0; k2 < R3; ++k2) {
                            const double2 z = v[ct::Dft<R3>::pos(k2)];
                            const int m = q + R0 * (u3 + J3 * k2);
                            row[m < Nf ? m : W - m] = mag_sqrt(fma(z.x, z.x, z.y * z.y)) * invNf;
                        }
                    }
                } else if (a3) {
                    // scratch rows are UNIT-MAJOR: |A_q[kappa]| at (q - 1) Q + kappa -- 147 consecutive doubles per store instruction (natural
                    // order: 8-byte stores 96 bytes apart, 44 100 write transactions per task = 106 of the kernel's 182 us)
                    double *ru = row + (q - 1) * Q + u3;
                    double sd = 0.0, sm = 0.0;
#pragma unroll
                    for (int k2 = 0; k2 < R3; ++k2) {
                        const double2 z = v[ct::Dft<R3>::pos(k2)];
                        const double mg = mag_sqrt(fma(z.x, z.x, z.y * z.y)) * invNf;
                        ru[J3 * k2] = mg;
                        // bin + 1 = c0 + R0 J3 k2 (direct) or W + 2 - c0 - R0 J3 k2 (mirror), c0 = q + R0 u3 + 1: the k2 part with a constant
                        // per register, the rest once per thread from the two sums
                        // (direct for k2 < 12, mirror for k2 > 12 -- R0 J3 k2 against W / 2, any q and u3 --: decided at compile time)
                        const bool direct = k2 < R3 / 2 || (k2 == R3 / 2 && q + R0 * (u3 + J3 * k2) < Nf);
                        um = fmax(um, mg);
                        if (direct) { sd += mg; uw = fma((double)(R0 * J3 * k2), mg, uw); }
                        else { sm += mg; uw = fma(-(double)(R0 * J3 * k2), mg, uw); }
                    }
                    const double c0 = (double)(q + R0 * u3 + 1);
                    us = sd + sm;
                    uw = fma(c0, sd, fma((double)(W + 2) - c0, sm, uw));
                }
            }
            if (packed_task) {          // (workgroup-uniform)
                __syncthreads();          // every thread of the packed unit has read its pass-3 inputs
                if (packed_unit && a3) {
#pragma unroll
                    for (int k2 = 0; k2 < R3; ++k2) buf[u3 + J3 * k2] = v[ct::Dft<R3>::pos(k2)];
                }
                __syncthreads();
                if (packed_unit) {
                    // pairs (j, Q - j), j = 1 .. Q / 2, and j = 0: X[H0 j] = E + w^j O, X[H0 (Q - j)] = conj(E - w^j O); X[0] = Re V[0] + Im V[0]
                    constexpr int NP = Q / 2 + 1;          // (even Q: j = Q / 2 pairs with itself)
                    const bool nat = P.mode == 1;
                    double *rp = nat ? row : row + (H0 - 1) * Q;          // unit-major: |X[H0 j]| at (H0 - 1) Q + j
                    const int st = nat ? H0 : 1;
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
                            const double ma = mag_sqrt(fma(ar, ar, ai * ai)) * invNf;
                            const bool two = j > 0 && Q - j != j;
                            const double mb = two ? mag_sqrt(fma(br, br, bi * bi)) * invNf : 0.0;
                            rp[st * j] = ma;
                            if (two) rp[st * (Q - j)] = mb;
                            us += ma + mb;
                            uw = fma((double)(H0 * j + 1), ma, uw);
                            uw = fma(two ? (double)(H0 * (Q - j) + 1) : 0.0, mb, uw);
                            um = fmax(um, fmax(ma, mb));
                        }
                    }
                }
            }
            if (P.mode != 1) {
                us = wsum(us); uw = wsum(uw); um = wmax_nonneg(um);
                if (lane == 0) { upart[4 * wave] = us; upart[4 * wave + 1] = uw; upart[4 * wave + 2] = um; }
                flush_row = cu.row;
                flush_row_b = pair_task ? cu.row + 1 : cu.row;
                flush_ua = pair_task ? H0 - 1 : (a_on ? 2 * type : -1);          // unit index of q = 2 type + 1
                flush_ub = packed_task ? H0 - 1 : 2 * type + 1;
            }
        }
        cu = cu_next;
        ti = t_next;
    }
    __syncthreads();
    if (flush_row >= 0) flush_partials(flush_row, flush_row_b, flush_ua, flush_ub);
    // the last workgroup out clears the counters for the next launch (they are zeroed once, when the scratch is allocated): a memset in front of
    // every launch was a 5 us fill kernel per counter word
    if (tid_ == 0) {
        __threadfence();
        if (atomicAdd(next_task + 8, 1) == (int)gridDim.x - 1) {
#pragma unroll
            for (int i = 0; i < 9; ++i) next_task[i] = 0;
        }
    }
}

// ---- features of one frame from its UNIT-MAJOR row (and the previous frame's) -- kernels_wg.hpp's wg_feat_kernel for the rows above.
// the row element that holds bin k: complex unit q = 1 .. R0/2 - 1 delivers the bins q + R0 kappa (kappa < Q) and, as mirrors, the bins
// R0 - q + R0 (Q - 1 - kappa); the last unit the bins (R0 / 2) j
template <int R0, int Q>
__device__ __forceinline__ int idx_of(int k) {
    constexpr int H0 = R0 / 2;
    const int kr = (int)((unsigned)k / (unsigned)R0);
    const int rho = k - kr * R0;
    if (rho == 0) return (H0 - 1) * Q + 2 * kr;
    if (rho == H0) return (H0 - 1) * Q + 2 * kr + 1;
    return rho < H0 ? (rho - 1) * Q + kr : (R0 - rho - 1) * Q + Q - 1 - kr;
}

// The row is read as STREAMS whose elements are R0 (or R0 / 2) bins apart: residue stream rho (rho = 1 .. R0 - 1, rho != R0 / 2) = the bins
// rho + R0 n -- unit rho read forwards, or unit R0 - rho read backwards (its mirrored half) --, and the packed unit's bins (R0 / 2) n.  64
// consecutive elements of a residue stream (128 of the packed one) lie inside ONE natural block of 64 R0 bins, so a wave that takes 64 elements
// per round reads both rows coalesced, knows every element's bin from two constants, and its sum of squares IS that stream's share of the
// block's energy: the roll-off (:127-140) is located from the block energies and finished inside one block of 64 R0 bins, and no natural-order
// copy of the row is needed but for the bins the mel filters use (LDS, a few thousand).  The workgroup keeps 45 KB of LDS instead of the
// whole row's 160 KB: three of them share a CU (the sweep of one runs under the reductions and the mel walk of the others), and 599 frames are
// one round of workgroups instead of three.
constexpr int kFeatT = 512, kFeatW = kFeatT / 64;
constexpr int LCAP = 4096;                          // bins below LCAP are also kept in LDS, natural order (the mel filters' range for the usual rates)
template <int R0, int Q> struct FeatGeo {
    static constexpr int NF = R0 * Q / 2, SPB = R0, BLK = 64 * R0, NBLK = (NF + BLK - 1) / BLK, NIT = NBLK * SPB;
    static constexpr int OFF_SLOT_E = LCAP * 8;                           // double [NBLK]: the natural blocks' energies
    static constexpr int OFF_FINE = OFF_SLOT_E + 64 * 8;                  // double [BLK]: the squares of the block that holds the roll-off bin
    static constexpr int OFF_SMALL = OFF_FINE + BLK * 8;                  // fv [48], msp [40], red [kFeatW][16], slot [kFeatW], redi [kFeatW + 2]
    static constexpr int LDS = OFF_SMALL + (48 + 40 + kFeatW * 16 + kFeatW) * 8 + (kFeatW + 2) * 4 + 8;
    static_assert(NBLK <= 64, "one lane per natural block");
};
template <int R0, int Q> constexpr int feat_lds() { return (FeatGeo<R0, Q>::LDS + 15) / 16 * 16; }

__device__ __forceinline__ double bsum8(double v, double *slot, int lane, int wave) {
    v = wsum(v);
    if (lane == 0) slot[wave] = v;
    __syncthreads();
    const double r = ((slot[0] + slot[1]) + (slot[2] + slot[3])) + ((slot[4] + slot[5]) + (slot[6] + slot[7]));
    __syncthreads();
    return r;
}

template <int R0, int Q>
__global__ __launch_bounds__(kFeatT, 2) void wgs_feat_kernel(PlanDev P, const wg::FrameRef *__restrict__ frames, const ClipDev *__restrict__ clips,
                                                             int n_frames, const double *__restrict__ spec, const double *__restrict__ tfeat,
                                                             const double *__restrict__ psum, double *__restrict__ out) {
    typedef FeatGeo<R0, Q> G;
    constexpr int NF = G::NF, W = R0 * Q, H0 = R0 / 2, SPB = G::SPB, NBLK = G::NBLK, NIT = G::NIT, BLK = G::BLK;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // XCD-aware: workgroup b runs on XCD b % 8 and takes frame (b % 8) per + b / 8 (grid = 8 per): the workgroups that run side by side on an XCD
    // hold consecutive frames, so a row is fetched from HBM once -- as one frame's own row -- and found in that XCD's L2 as the next frame's
    // previous row
    const int per = (n_frames + 7) / 8, fi = (int)(blockIdx.x % 8) * per + (int)(blockIdx.x / 8);
    if (fi >= n_frames) return;
    const wg::FrameRef fr = frames[fi];
    if (fr.halo) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const ClipDev c = clips[fr.clip];
    double *low = reinterpret_cast<double *>(smem);
    double *slotE = reinterpret_cast<double *>(smem + G::OFF_SLOT_E);
    double *fine = reinterpret_cast<double *>(smem + G::OFF_FINE);
    double *fv = reinterpret_cast<double *>(smem + G::OFF_SMALL);             // [48]
    double *msp = fv + 48;               // [40]
    double *red = msp + 40;              // [kFeatW][16]
    double *slot = red + kFeatW * 16;    // [kFeatW]
    int *redi = reinterpret_cast<int *>(slot + kFeatW);      // [kFeatW + 2]
    const double *gc = spec + (long long)fr.row * NF;
    const double *prv = (fr.t == 0) ? gc : gc - NF;          // frames are laid out in clip order: the previous frame is the previous row
    auto X = [&](int k) -> double { return k < LCAP ? low[k] : gc[idx_of<R0, Q>(k)]; };
    double *oc = out + c.out_off;
    const long long Tc = c.T;
    const Tabs tb = tabs_global(P);
    // ---- the row's sums and maximum, and the previous row's sum, from the transform kernel's per-unit parts (fixed order: every frame gets the
    // same bits wherever it runs) -- centroid and the normalisations of spread and flux are known BEFORE the sweep
    double sX = 0.0, sIXr = 0.0, mx = 0.0, sXpr = 0.0;
    {
        const double *pc = psum + (long long)fr.row * H0 * 4, *pp = (fr.t == 0) ? pc : pc - H0 * 4;
#pragma unroll
        for (int u = 0; u < H0; ++u) { sX += pc[4 * u]; sIXr += pc[4 * u + 1]; mx = fmax(mx, pc[4 * u + 2]); sXpr += pp[4 * u]; }
    }
    const double f0 = P.fs / (2.0 * (double)NF);
    const double sIX = sIXr * f0;
    const double sXe = sX + (double)NF * kEps;                // np.sum(X + eps) (:118-119)
    const double sXp = sXpr + (double)NF * kEps;
    const double r = (mx == 0.0) ? 1.0 / kEps : fast_div(1.0, mx);
    const double den = sX * r + kEps;
    const double cen = fast_div(sIX * r, den);
    const double rX = fast_div(1.0, sXe), rXp = fast_div(1.0, sXp);
    // ---- ONE sweep (:57-124), one coalesced read of both rows: wave w takes the natural blocks jb = w, w + 8, ... -- all SPB streams of a block
    // at once: 2 SPB loads in flight per lane (the rows come from other CUs' stores: HBM latency), every stream's constants known at compile time
    // (a first version that dealt (block, stream) pairs to the waves spent 200 vector instructions per 64 elements, most of them on finding out
    // which stream it held: 77 us of this kernel's 108).  A block of 64 R0 bins meets at most two spectral-entropy blocks (:85-107): the
    // lanes' sums of squares go to the block they belong to through ONE per-lane accumulator that is flushed when the entropy block changes.
    const int LB = P.blk_f;
    double *ered = red;                   // [kFeatW][16]: entropy blocks 0..9, 10 = the tail; 11 spread, 12 flux
    if (lane < 16) ered[16 * wave + lane] = 0.0;
    double sSp = 0.0, sFl = 0.0;
    {
        double accE = 0.0;                // this lane's squares of the entropy block e_cur
        int e_cur = -1;
        auto flush = [&](int e, double acc) {
            const double v = wsum(acc);
            if (lane == 0 && e >= 0) ered[16 * wave + e] += v;
        };
        const double kf = (double)(R0 * lane);          // (bin + 1) f0 - cen = (kb + rho + 1 + R0 lane) f0 - cen
#pragma unroll 1
        for (int jb = wave; jb < ((kAblate & 32) ? 0 : NBLK); jb += kFeatW) {
            const int n = 64 * jb + lane;
            double xv[SPB], pv[SPB];
            const bool full = jb < NBLK - 1;          // (only the last block has bins beyond the row: wave-uniform)
#pragma unroll
            for (int s2 = 0; s2 < SPB; ++s2) {
                const bool packed = s2 >= R0 - 2;
                const int rho = (s2 < H0 - 1) ? s2 + 1 : s2 + 2;
                const bool fwd = packed || rho < H0;
                const int base = packed ? (H0 - 1) * Q : (fwd ? (rho - 1) * Q : (R0 - rho - 1) * Q + Q - 1);
                const int ns = packed ? n + 64 * jb + 64 * (s2 - (R0 - 2)) : n;          // element of the stream: 64 (2 jb + h) + lane for the packed halves
                const int k = packed ? H0 * ns : rho + R0 * ns;
                const int idx = (full || k < NF) ? (fwd ? base + ns : base - ns) : 0;
                xv[s2] = gc[idx];
                pv[s2] = prv[idx];
            }
            const int kb = BLK * jb;                                   // first bin of the block
            const int e_first = min((int)((unsigned)kb / (unsigned)LB), 10), e_last = min((int)((unsigned)(kb + BLK - 1) / (unsigned)LB), 10);
            const int bnd = (e_first + 1) * LB;                        // first bin of the next entropy block
            const bool stage_low = kb < LCAP;
            if (e_first != e_cur) { flush(e_cur, accE); accE = 0.0; e_cur = e_first; }
            double accR = 0.0, accLo = 0.0;
            const double c0 = (double)(kb + 1) * f0 - cen;
#pragma unroll
            for (int s2 = 0; s2 < SPB; ++s2) {
                const bool packed = s2 >= R0 - 2;
                const int rho = (s2 < H0 - 1) ? s2 + 1 : s2 + 2;
                const int ns = packed ? n + 64 * jb + 64 * (s2 - (R0 - 2)) : n;
                const int k = packed ? H0 * ns : rho + R0 * ns;
                const bool in = full || k < NF;
                const double Xv = in ? xv[s2] : 0.0, Pv = in ? pv[s2] : 0.0;
                if (stage_low && k < LCAP && in) low[k] = Xv;
                const double sq = Xv * Xv;
                accR += sq;
                if (e_first != e_last) accLo += (k < bnd) ? sq : 0.0;
                // (k + 1) f0 - cen with k - kb = rho + R0 lane (residue) / H0 (64 h + lane) (packed): a constant per stream + a constant per lane
                const double dv = packed ? fma((double)(H0 * (64 * (s2 - (R0 - 2)))) + 0.5 * kf, f0, c0) : fma((double)rho + kf, f0, c0);
                sSp = fma(dv * dv, Xv * r, sSp);
                const double df = Xv * rX - Pv * rXp;
                sFl = fma(df, df, sFl);
            }
            const double eb = wsum(accR);                 // the natural block's energy (the roll-off is located from these)
            if (lane == 0) slotE[jb] = eb;
            if (e_first == e_last) accE += accR;
            else { flush(e_cur, accE + accLo); accE = accR - accLo; e_cur = e_last; }
        }
        flush(e_cur, accE);
    }
    sSp = wsum(sSp); sFl = wsum(sFl);
    if (lane == 0) { ered[16 * wave + 11] = sSp; ered[16 * wave + 12] = sFl; }
    __syncthreads();          // (also: the low bins and the block energies are in place)
    double part[11];
#pragma unroll
    for (int i = 0; i < 11; ++i) {
        double a = 0.0;
#pragma unroll
        for (int w = 0; w < kFeatW; ++w) a += red[16 * w + i];
        part[i] = a;
    }
    sSp = 0.0; sFl = 0.0;
#pragma unroll
    for (int w = 0; w < kFeatW; ++w) { sSp += red[16 * w + 11]; sFl += red[16 * w + 12]; }
    double sP = part[10];
#pragma unroll
    for (int j = 0; j < 10; ++j) sP += part[j];
    if (P.mode == 2) {
        // ---- chromagram row (:356-359): pitch class by pitch class, one wave per class, all lanes on the class's bins
        for (int cls = wave; cls < 12; cls += kFeatW) {
            const int b = tb.ch_start[cls], e = tb.ch_start[cls + 1];
            double acc = 0.0;
            for (int i = b + lane; i < e; i += 64) { const double xv = X(tb.ch_src[i]); acc = fma(xv * xv, tb.ch_w[i], acc); }
            acc = wsum(acc);
            if (lane == 0) oc[(long long)fr.t * 12 + cls] = (sP == 0.0) ? acc / kEps : fast_div(acc, sP);
        }
        return;
    }
    double ent_f = 0.0;          // spectral entropy (:85-107)
#pragma unroll
    for (int j = 0; j < 10; ++j) {
        const double sh = fast_div(part[j], sP + kEps);
        ent_f -= sh * fast_log2(sh + kEps);
    }
    const double spread = fast_sqrt(fast_div(sSp, den));
    // ---- roll-off (:127-140): first k with cumsum(X^2)[k] + eps > 0.9 sum(X^2).  The natural blocks' energies (the streams' shares added in
    // stream order) locate the block; its 64 R0 squares come back in natural order (L2: this workgroup has just read them) and one wave walks them
    const double thr = 0.90 * sP;
    if (wave == 0 && !(kAblate & 128)) {
        const double En = (lane < NBLK) ? slotE[lane] : 0.0;
        const double cum = wscan_incl(En);
        const int jc = wmin_i((lane < NBLK && cum + kEps > thr) ? lane : 0x7fffffff);
        if (lane == (jc == 0x7fffffff ? 0 : jc)) { redi[kFeatW] = jc; slot[0] = cum - En; }
    }
    __syncthreads();
    int first = 0x7fffffff;
    {
        const int jc = redi[kFeatW];
        const double base = slot[0];
        if (jc != 0x7fffffff) {
            for (int t = tid; t < BLK; t += kFeatT) {
                const int k = BLK * jc + t;
                const double xv = (k < NF) ? gc[idx_of<R0, Q>(k)] : 0.0;
                fine[t] = xv * xv;
            }
        }
        __syncthreads();
        if (wave == 0 && jc != 0x7fffffff) {
            double own = 0.0;
            double xr[R0];
#pragma unroll
            for (int i = 0; i < R0; ++i) { xr[i] = fine[R0 * lane + i]; own += xr[i]; }
            double run = base + (wscan_incl(own) - own);
#pragma unroll
            for (int i = 0; i < R0; ++i) {
                run += xr[i];
                const int k = BLK * jc + R0 * lane + i;
                if (first == 0x7fffffff && k < NF && run + kEps > thr) first = k;
            }
            first = wmin_i(first);
            // (the block sums and this walk add in different orders: a crossing the walk misses by a rounding is the next block's first bin)
            if (first == 0x7fffffff && BLK * (jc + 1) < NF) first = BLK * (jc + 1);
            if (lane == 0) redi[kFeatW + 1] = first;
        }
        __syncthreads();
        first = (jc != 0x7fffffff) ? redi[kFeatW + 1] : 0x7fffffff;
    }
    // ---- MFCC (:236-254): wave w owns the filters w, w + 8, ..., walked together, all 64 lanes on a filter's bins
    {
        constexpr int NFW = 40 / kFeatW;
        int lo[NFW], cnt[NFW];
        const double *wv[NFW];
        double a[NFW];
        int maxc = 0;
#pragma unroll
        for (int j = 0; j < NFW; ++j) {
            const int m = wave + kFeatW * j;
            lo[j] = tb.mel_lo[m]; cnt[j] = (kAblate & 256) ? 0 : tb.mel_cnt[m]; wv[j] = tb.mel_w + tb.mel_off[m];
            a[j] = 0.0;
            maxc = max(maxc, cnt[j]);
        }
        for (int i = lane; i < maxc; i += 64) {
#pragma unroll
            for (int j = 0; j < NFW; ++j)
                if (i < cnt[j]) a[j] = fma(X(lo[j] + i), wv[j][i], a[j]);
        }
#pragma unroll
        for (int j = 0; j < NFW; ++j) a[j] = wsum(a[j]);
        double mine = a[0];
#pragma unroll
        for (int j = 1; j < NFW; ++j) mine = (lane == j) ? a[j] : mine;
        if (lane < NFW) msp[wave + kFeatW * lane] = fast_log10(mine + kEps);
    }
    // ---- chroma (:277-321): the pitch classes w and w + 8 of a wave, walked together
    {
        const int c0 = wave, c1 = wave + kFeatW;
        const bool two = c1 < 12;
        const int b0 = tb.ch_start[c0], e0 = (kAblate & 512) ? b0 : tb.ch_start[c0 + 1];
        const int b1 = two ? tb.ch_start[c1] : 0, e1 = (two && !(kAblate & 512)) ? tb.ch_start[c1 + 1] : 0;
        double acc0 = 0.0, acc1 = 0.0;
        const int n0 = e0 - b0, n1 = e1 - b1, nmax = max(n0, n1);
#pragma unroll 2
        for (int i = lane; i < nmax; i += 64) {
            if (i < n0) { const double xv = X(tb.ch_src[b0 + i]); acc0 = fma(xv * xv, tb.ch_w[b0 + i], acc0); }
            if (i < n1) { const double xv = X(tb.ch_src[b1 + i]); acc1 = fma(xv * xv, tb.ch_w[b1 + i], acc1); }
        }
        acc0 = wsum(acc0); acc1 = wsum(acc1);
        if (lane == 0) fv[21 + c0] = (sP == 0.0) ? acc0 / kEps : fast_div(acc0, sP);
        if (lane == 1 && two) fv[21 + c1] = (sP == 0.0) ? acc1 / kEps : fast_div(acc1, sP);
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
        fv[7] = (first == 0x7fffffff) ? 0.0 : (double)first / (double)NF;
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

// ---- host -------------------------------------------------------------------------------------------------------------------------
// (r0, points per sub-transform) of a window this family takes; r0 = 0: none
struct Sel { int r0, q; };
inline Sel wgs_select(int window) {
    if (window == 12 * S3675::Q) return {12, S3675::Q};          // 44 100: 1 s at 44.1 kHz
    if (window == 6 * S3675::Q) return {6, S3675::Q};            // 22 050: 1 s at 22.05 kHz, 0.5 s at 44.1 kHz
    if (window == 12 * S4000::Q) return {12, S4000::Q};          // 48 000: 1 s at 48 kHz
    if (window == 8 * S4000::Q) return {8, S4000::Q};            // 32 000: 1 s at 32 kHz
    if (window == 6 * S4000::Q) return {6, S4000::Q};            // 24 000: 1 s at 24 kHz, 0.5 s at 48 kHz
    return {0, 0};
}
inline int wgs_task_types(int r0) { return r0 == 12 ? task_types<12>() : (r0 == 8 ? task_types<8>() : task_types<6>()); }

}  // namespace wgs
}  // namespace paa
