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
//              the sign changes (integer sign codes; the sample before a pair is in the lane below: DPP wave_shr:1; what lane 0 of a wave
//              misses -- the last lane of the wave before it -- crosses in 2-bit codes through LDS behind the first exchange's barrier)
//   pass 1   : radix-R1 codelet (kernels_ct.hpp: 20 = 4 x 5 / 10 = 2 x 5 prime-factor forms, exact zeros for equal inputs), outputs
//              times W_N^(j k0) -- the powers of ONE table value, formed by multiplying -- stored group by group under the
//              remaining butterflies (group_pass)
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
//              filters as LANE JOBS: filter m's bins are dealt round-robin to nl[m] consecutive threads, at most sixteen bins each (all
//              threads of the workgroup carry about the same number), whose weights -- the reference's table -- are requested once per
//              frame, ahead of the sums; the threads' partial sums meet in LDS, eight lanes per filter add them (wave w: filters w,
//              w + 7, ...).  (One wave walking a filter with a load from L2 per step cost 11 k of 59 k cycles per frame.)  The chroma
//              gather lists live in registers (two classes per wave, one entry per lane); the logarithms of the two
//              entropies, the DCT and the chroma deviation on four different waves at once
//   The next frame's samples are touched (one load per line) before the feature stage starts.
// Nothing but the samples (read once + the overlap of the windows from L2) and the feature columns touches HBM; spectrogram plans write
// each row once from the registers.  Three passes instead of five, three exchanges through LDS instead of ten round trips.
// Replaces ShortTermFeatures.py:608-682 (+ helpers :22-140, :236-321), spectrogram (:415-422), chromagram (:349-359) for these windows.
#pragma once
#include "kernels_tri.hpp"          // tri::Cd (register codelets)

namespace paa {
namespace wgr {
#if defined(PAA_F800_TIMING) || defined(PAA_F800_TRACE)
using f800::g_phase_cycles;
using f800::g_wave_trace;
#endif

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
    static constexpr int OFF_EDGE = OFF_REDI + NW * 4;    // unsigned [NW][2]: packed sign codes of each wave's last pass-1 lane
    static constexpr int OFF_MELA = (OFF_EDGE + NW * 8 + 15) / 16 * 16;      // int2 [40]: first thread and number of threads of each mel filter
    static constexpr int OFF_DCT = OFF_MELA + 40 * 8;     // double [13][40]
    static constexpr int OFF_PART = N * 8;                // double [NT]: the threads' mel partial sums, in the buffer behind the spectrum row
    static constexpr int OFF_WS = OFF_DCT + 13 * 40 * 8;  // double2 [NW][3][64]: the waves' pair-sum scratch (pair_sums)
    static constexpr int LDS_BYTES = OFF_WS + NW * 3 * 64 * 16;
    static_assert(R1 % 10 == 0, "time-domain entropy blocks: static per register row");
    static_assert(J1 % 10 == 0, "spectral entropy blocks: whole scan chunks");
    static_assert(A1 >= (R2 - 1) * R3 + R3 && B2 >= R3 && A2 >= (R2 - 1) * B2 + R3, "exchange rows");
    static_assert(LDS_BYTES <= 160 * 1024, "one workgroup's LDS");
    static_assert(OFF_PART + NT * 8 <= BUF * 16, "the mel partial sums fit behind the spectrum row");
    static_assert(NW >= 5, "the final stage spreads over five waves");
};
// pads from scripts/dev/wgr_model.py (reads and writes of both exchanges at the conflict-free cycle count)
typedef Shape<20, 20, 20, 404, 401, 20> S16000;
typedef Shape<10, 20, 20, 404, 439, 22> S8000;

// the plan's tables (global memory, built by wgr_build_tab): the mel lane jobs (ShortTermFeatures.py:236-254; weights: the plan's own
// table, tables.hpp build_mel) and the chroma gather lists, one entry per lane (at most 64 per pitch class: one per semitone the bins
// reach)
constexpr int kMelPerLane = 16;     // bins of a mel lane job, at most
constexpr int kMaxThreads = 512;
struct WgrTab {
    int mel_job[kMaxThreads][4];    // thread t: {first bin, index of its weight, stride = threads of the filter, number of bins}: bins
                                    // kb + j stride, weights mel_w[eb + j stride], j < n
    int mel_fil[40][2];             // filter m: {first thread, threads}
    int ch_n[12];
    int ch_src[12][64];
    double ch_w[12][64];
};

// two consecutive samples as loaded (held across the feature stage as the NEXT frame's prefetch: integer types only)
template <typename T> struct Smp {
    ct::PairRaw<T> r;
    static constexpr bool kInt = true;
    static __device__ __forceinline__ Smp get(const T *p) { Smp s; s.r = ct::PairRaw<T>::get(p); return s; }
    __device__ __forceinline__ int i0() const { return r.x0(); }
    __device__ __forceinline__ int i1() const { return r.x1(); }
    __device__ __forceinline__ double d0() const { return (double)r.x0(); }
    __device__ __forceinline__ double d1() const { return (double)r.x1(); }
};
template <> struct Smp<double> {
    double2 v;
    static constexpr bool kInt = false;
    static __device__ __forceinline__ Smp get(const double *p) { Smp s; s.v = ct::PairLoad<double>::get(p); return s; }
    __device__ __forceinline__ int i0() const { return 0; }
    __device__ __forceinline__ int i1() const { return 0; }
    __device__ __forceinline__ double d0() const { return v.x; }
    __device__ __forceinline__ double d1() const { return v.y; }
};

__device__ __forceinline__ double2 csqr(double2 a) { return make_double2(fma(a.x, a.x, -a.y * a.y), 2.0 * (a.x * a.y)); }
// lane l receives the value of lane l - 1 (lane 0: `first`)
__device__ __forceinline__ int shr1(int v, int first) { return __builtin_amdgcn_update_dpp(first, v, 0x138, 0xF, 0xF, false); }

// Sums over the wave of NP <= 3 pairs of values at once: every lane parks its pairs in the wave's LDS scratch, lane (row i, column c) adds
// the four rows' values of pair i at column c (conflict-free: 16 consecutive 16-byte elements per read), a 16-lane DPP reduction finishes:
// all lanes of row i < NP return the totals of pair i.  37 + NP instructions instead of 29 per value on DPP alone.
template <int NP>
__device__ __forceinline__ double2 pair_sums(const double2 *v, double2 *ws, int lane) {
#pragma unroll
    for (int i = 0; i < NP; ++i) ws[64 * i + lane] = v[i];
    wsync();
    const int row = lane >> 4, i = row < NP ? row : NP - 1, c = lane & 15;
    const double2 *q = ws + 64 * i + c;
    const double2 a0 = q[0], a1 = q[16], a2 = q[32], a3 = q[48];
    wsync();
    double2 a = make_double2((a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y));
    a.x = group_sum(a.x);
    a.y = group_sum(a.y);
    return a;
}

// One pass of radix 20 = 4 x 5 / 10 = 2 x 5 (the prime-factor codelets of kernels_ct.hpp, exact zeros for equal inputs) whose outputs
// leave GROUP BY GROUP: the second stage's radix-5 butterfly g produces X[g], X[g + G], .. X[g + 4 G] (G = 4 / 2 groups); they are
// multiplied by W^q -- chain g: W^g, then times W^G per step, at most six roundings deep -- and handed to put(q, value) at once, so the
// LDS stores of a pass (1 820 cycles of the CU's store path per exchange when all waves store together behind the butterflies) start
// under the remaining butterflies, and no more than five outputs wait in registers for their twiddle
template <int R> struct GroupPass;
template <> struct GroupPass<20> {
    static constexpr int G = 4;
    static __device__ __forceinline__ void stage1(double2 *v) {
#pragma unroll
        for (int n2 = 0; n2 < 5; ++n2) {
            ct::dft4r(v[(4 * n2) % 20], v[(5 + 4 * n2) % 20], v[(10 + 4 * n2) % 20], v[(15 + 4 * n2) % 20]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    static __device__ __forceinline__ void stage2(double2 *v, int g) {
        ct::dft5r(v[(5 * g) % 20], v[(5 * g + 4) % 20], v[(5 * g + 8) % 20], v[(5 * g + 12) % 20], v[(5 * g + 16) % 20]);
    }
};
template <> struct GroupPass<10> {
    static constexpr int G = 2;
    static __device__ __forceinline__ void stage1(double2 *v) {
#pragma unroll
        for (int n2 = 0; n2 < 5; ++n2) {
            const double2 a = v[(2 * n2) % 10], b = v[(5 + 2 * n2) % 10];
            v[(2 * n2) % 10] = cadd(a, b);
            v[(5 + 2 * n2) % 10] = csub(a, b);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    static __device__ __forceinline__ void stage2(double2 *v, int g) {
        ct::dft5r(v[(5 * g) % 10], v[(5 * g + 2) % 10], v[(5 * g + 4) % 10], v[(5 * g + 6) % 10], v[(5 * g + 8) % 10]);
    }
};
template <int R, bool TW, typename Put>
__device__ __forceinline__ void group_pass(double2 *v, double2 w, Put put) {
    typedef GroupPass<R> GP;
    typedef tri::Cd<R> CD;
    constexpr int G = GP::G;
    GP::stage1(v);
    double2 wp[5];                  // W^1 .. W^G (wp[0] unused)
    if (TW) {
        // (opaque: the powers of a loop-invariant value would be hoisted out of the frame loop and spilled)
        asm volatile("" : "+v"(w.x), "+v"(w.y));
        wp[1] = w; wp[2] = csqr(w);
        if (G == 4) { wp[3] = cmul(wp[2], w); wp[4] = csqr(wp[2]); }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
        GP::stage2(v, g);
        double2 t = TW ? wp[g == 0 ? G : g] : make_double2(1.0, 0.0);
#pragma unroll
        for (int m = 0; m < 5; ++m) {
            const int q = g + G * m;
            double2 x = v[CD::pos(q)];
            if (TW && q > 0) {
                x = cmul(x, t);
                if (m < 4) t = cmul(t, wp[G]);
            }
            put(q, x);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// MODE 0: the 34 feature rows, 1: spectrogram rows, 2: chromagram rows
template <typename SH, typename T, int MODE>
__global__ __launch_bounds__(SH::NT) void wgr_kernel(PlanDev P, const T *__restrict__ sig, const ClipDev *__restrict__ clips,
                                                     const ClipNorm *__restrict__ norms, const Tile *__restrict__ runs, int n_runs,
                                                     const WgrTab *__restrict__ tab, double *__restrict__ out) {
    constexpr int R1 = SH::R1, R2 = SH::R2, R3 = SH::R3, A1 = SH::A1, A2 = SH::A2, B2 = SH::B2, N = SH::N, NF = SH::NF, W = SH::W;
    constexpr int J1 = SH::J1, J2 = SH::J2, J3 = SH::J3, NW = SH::NW, NJR = SH::NJR, C = SH::C;
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
    unsigned *edge = reinterpret_cast<unsigned *>(smem + SH::OFF_EDGE);
    int2 *melfil = reinterpret_cast<int2 *>(smem + SH::OFF_MELA);
    double *part = reinterpret_cast<double *>(smem + SH::OFF_PART);
    double *dct = reinterpret_cast<double *>(smem + SH::OFF_DCT);
    double2 *ws_all = reinterpret_cast<double2 *>(smem + SH::OFF_WS);
    constexpr bool INT_T = Smp<T>::kInt;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // ---- the thread's jobs (threads past a pass's job count shadow its last job; their stores are masked)
    const double sc = sample_scale<T>();
    const double invNf = 1.0 / (double)NF;
    const Tabs tb = tabs_global(P);
    const double f0 = P.fs / (2.0 * (double)NF);
    // ---- tables, once per workgroup: the mel constants to LDS, the wave's two chroma classes (w, w + NW) to registers
    int mj_kb = 0, mj_eb = 0, mj_st = 1, mj_n = 0;      // the thread's mel lane job
    if (MODE == 0) {
        if (tid < 40) melfil[tid] = make_int2(tab->mel_fil[tid][0], tab->mel_fil[tid][1]);
        mj_kb = tab->mel_job[tid][0]; mj_eb = tab->mel_job[tid][1]; mj_st = tab->mel_job[tid][2]; mj_n = tab->mel_job[tid][3];
        for (int i = tid; i < 13 * 40; i += SH::NT) dct[i] = tb.dct[(i / 40) * tb.dct_stride + i % 40];
    }
    __syncthreads();

    const int tile_id = blockIdx.x * NW + wave;          // (timing builds: the wave's slot in the trace)
    (void)tile_id;
    PAA_T0()
    for (int run = blockIdx.x; run < n_runs; run += gridDim.x) {
        const Tile tl = runs[run];
        const ClipDev c = clips[tl.clip];
        const ClipNorm nm = norms[tl.clip];
        const T *xc = sig + c.sample_off + P.frame_origin;
        double *oc = out + c.out_off;
        const long long Tc = c.T;
        const double inv = nm.inv;
        double sXp = 0.0;                                // sum of the previous frame's magnitudes
        double pm[2 * NJR];                              // ... and the magnitudes themselves (this thread's bins)
#pragma unroll
        for (int i = 0; i < 2 * NJR; ++i) pm[i] = 0.0;
        SignRule sr = {0, 0, 0};
        if (INT_T && MODE == 0) sr = sign_rule<T>(nm.mean);
        const int t_first = (MODE == 0 && tl.t0 > 0) ? tl.t0 - 1 : tl.t0, t_end = tl.t0 + tl.cnt;
        for (int t = t_first; t < t_end; ++t) {
            const bool halo = t < tl.t0;
            // (the job indices are formed again in every iteration from an opaque copy of the thread index: as loop invariants they -- and
            // the addresses, conversions and masks derived from them -- filled a hundred registers that were then spilled)
            int tq = tid;
            asm volatile("" : "+v"(tq));
            const int wv = tq >> 6, ln = tq & 63;              // (for LDS addresses: the scalar `wave` / `lane` forms were hoisted and spilled)
            double2 *ws = ws_all + 3 * 64 * wv;
            const bool a1 = tq < J1, a2 = tq < J2, a3 = tq < J3;
            const int j1 = a1 ? tq : J1 - 1;
            const int t2 = a2 ? tq : J2 - 1, k0_2 = t2 / R3, n2_2 = t2 - k0_2 * R3;
            const int u3 = a3 ? tq : J3 - 1, k1_3 = u3 / R1, k0_3 = u3 - k1_3 * R1;
            const int e1r = k0_2 * A1 + n2_2, e2w = k0_2 * A2 + n2_2, e2r = k0_3 * A2 + k1_3 * B2;
            // inactive pass-1 threads re-read the last job's samples with scale and mean 0: exact zeros, no energy
            const double scl = a1 ? sc : 0.0, meanl = a1 ? nm.mean : 0.0;
            const double mscale = a1 ? 0.5 * invNf : 0.0;
            // the two twiddle bases of the thread, fetched with the samples (L2) instead of living in eight registers across the frame
            const double2 w1 = P.tw[j1];                         // W_N^j
            const double2 w2 = P.tw[R1 * n2_2];                  // W_(R2 R3)^n2
            // ---------------- load + normalise (:567-570)
            Smp<T> smp[R1];
            {
                const T *x = xc + (long long)t * P.S + 2 * j1;
#pragma unroll
                for (int n0 = 0; n0 < R1; ++n0) smp[n0] = Smp<T>::get(x + 2 * n0 * J1);
            }
            double2 v[R1];
#pragma unroll
            for (int n0 = 0; n0 < R1; ++n0)
                v[n0] = make_double2(fma(smp[n0].d0(), scl, -meanl) * inv, fma(smp[n0].d1(), scl, -meanl) * inv);
#ifdef PAA_F800_TIMING
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
            PAA_TICK(0)
            // ---------------- time domain (:22-51) on the same registers
            int zc_w = 0;                                  // the wave's sign changes, but for lane 0's left neighbours
            unsigned pc0a = 0, pc0b = 0;                   // lane 0's first samples' codes, two bits per register row (wave-uniform)
            if (MODE == 0 && !halo) {
                double eb[10];
#pragma unroll
                for (int b = 0; b < 10; ++b) eb[b] = 0.0;
                const int last_lane = __builtin_amdgcn_readfirstlane((64 * wave + 63 < J1 - 1) ? 63 : J1 - 1 - 64 * wave);
                // sign codes (device_common.hpp: integer samples are compared with the clip mean in integer arithmetic); the sample before
                // a pair is the lane below's second one; lane 0 meets itself here and its true neighbour -- the last lane of the wave
                // before -- behind the next barrier
                int zc = 0;
                unsigned pc1a = 0, pc1b = 0;
#pragma unroll
                for (int n0 = 0; n0 < R1; ++n0) {
                    eb[n0 / SH::RB] += fma(v[n0].x, v[n0].x, v[n0].y * v[n0].y);
                    int c0, c1;
                    if constexpr (INT_T) { c0 = sgn1(smp[n0].i0(), sr); c1 = sgn1(smp[n0].i1(), sr); }
                    else { c0 = sgn1(v[n0].x); c1 = sgn1(v[n0].y); }
                    const int left = shr1(c1, c0);
                    sad_acc(zc, c0, left);
                    sad_acc(zc, c1, c0);
                    // (scalar copies of the two lanes whose codes cross a wave boundary; packed by the scalar unit)
                    const unsigned f0 = (unsigned)__builtin_amdgcn_readlane(c0, 0), f1 = (unsigned)__builtin_amdgcn_readlane(c1, last_lane);
                    if (n0 < 16) { pc0a |= f0 << (2 * n0); pc1a |= f1 << (2 * n0); }
                    else { pc0b |= f0 << (2 * (n0 - 16)); pc1b |= f1 << (2 * (n0 - 16)); }
                }
                if (lane == 0) { edge[2 * wv] = pc1a; edge[2 * wv + 1] = pc1b; }
                zc_w = wsum_i(a1 ? zc : 0);
                // the ten block energies: pairs (E0, E1) .. (E4, E5), then (E6, E7), (E8, E9)
                {
                    double2 pr[3] = {make_double2(eb[0], eb[1]), make_double2(eb[2], eb[3]), make_double2(eb[4], eb[5])};
                    const double2 s1 = pair_sums<3>(pr, ws, lane);
                    if ((lane & 15) == 0 && lane < 48) *reinterpret_cast<double2 *>(red + 16 * wv + 4 + 2 * (ln >> 4)) = s1;
                    double2 pq[2] = {make_double2(eb[6], eb[7]), make_double2(eb[8], eb[9])};
                    const double2 s2 = pair_sums<2>(pq, ws, lane);
                    if ((lane & 15) == 0 && lane < 32) *reinterpret_cast<double2 *>(red + 16 * wv + 10 + 2 * (ln >> 4)) = s2;
                }
            }
            PAA_TICK(1)
            __builtin_amdgcn_sched_barrier(0);             // (the time-domain stage interleaved with the first pass holds 60 registers more)
            // ---------------- pass 1: radix R1 over n0, outputs times W_N^(j k0)
            if (a1) {
                double2 *dst = buf + j1;
                group_pass<R1, true>(v, w1, [&](int q, double2 x) { dst[q * A1] = x; });
            }
            PAA_TICK(2)
            __syncthreads();
            if (MODE == 0 && !halo) {
                // lane 0's left neighbours: row n0 of the wave before's last lane -- for wave 0: row n0 - 1 of the LAST pass-1 thread (the
                // element before z[J1 n0] is z[J1 (n0 - 1) + J1 - 1]); the frame's first sample meets itself (:22-26 has W - 1 differences).
                // Lane n0 < R1 takes row n0.
                const int pwv = wave > 0 ? wave - 1 : NW - 1;
                unsigned la = edge[2 * pwv], lb = edge[2 * pwv + 1];
                if (wave == 0) { lb = (lb << 2) | (la >> 30); la = (la << 2) | (pc0a & 3u); }
                const int n0 = lane < R1 ? lane : 0, sh = 2 * (n0 & 15);
                const unsigned mine = ((n0 < 16 ? pc0a : pc0b) >> sh) & 3u, left = ((n0 < 16 ? la : lb) >> sh) & 3u;
                const int zb = wsum_i(lane < R1 ? abs((int)mine - (int)left) : 0);
                if (lane == 0) red[16 * wv + 14] = (double)((zc_w + zb) << (INT_T ? sr.sh : 0));
            }
            // ---------------- pass 2: radix R2 over n1 for (k0, n2), outputs times W_(R2 R3)^(n2 k1)
            double2 v2[R2];
#pragma unroll
            for (int r = 0; r < R2; ++r) v2[r] = buf[e1r + r * R3];
            __syncthreads();              // every thread has read its pass-2 inputs
            if (a2) {
                double2 *dst = buf + e2w;
                group_pass<R2, true>(v2, w2, [&](int q, double2 x) { dst[q * B2] = x; });
            }
            PAA_TICK(3)
            __syncthreads();
            // ---------------- pass 3: radix R3 over n2 for (k0, k1): Z[k0 + R1 k1 + R1 R2 k2], natural order into the buffer
            double2 v3[R3];
#pragma unroll
            for (int r = 0; r < R3; ++r) v3[r] = buf[e2r + r];
            // (opaque copy of the thread's first bin: everything derived from it below -- table addresses, (double)(k + 1) f0, ... -- is
            // formed again per frame instead of being hoisted out of the frame loop into 100 spilled registers)
            const int jf = j1;
            const bool first0 = jf == 0;                   // this thread's pair jj = 0 is k = 0: bins 0 and N / 2
            double2 pw[NJR];                               // w^k of the recombination, requested ahead of the last pass
#pragma unroll
            for (int jj = 0; jj < NJR; ++jj) pw[jj] = P.post[jf + J1 * jj];
            __syncthreads();              // every thread has read its pass-3 inputs
            if (a3) {
                double2 *dst = buf + u3;
                group_pass<R3, false>(v3, make_double2(1.0, 0.0), [&](int q, double2 x) { dst[q * J3] = x; });
            }
            PAA_TICK(4)
            __syncthreads();
            PAA_TICK(10)
            // ---------------- real-FFT recombination + |X| / num_fft (:617-621): pairs k = t + J1 jj and N - k (k = 0: bins 0 and N / 2)
            double mg[2 * NJR];
            // (two batches of pairs: all twenty elements in flight at once were 80 registers beside w^k, the magnitudes and the previous ones)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                constexpr int HB = (NJR + 1) / 2;
                const int j0 = h * HB, j1e = (h == 0) ? HB : NJR;
                double2 zk[NJR], zm[NJR];
#pragma unroll
                for (int jj = j0; jj < j1e; ++jj) {
                    const int k = jf + J1 * jj;
                    const bool k0 = (jj == 0) && first0;
                    zk[jj] = buf[k];
                    zm[jj] = buf[k0 ? N / 2 : N - k];
                }
#pragma unroll
                for (int jj = j0; jj < j1e; ++jj) {
                    const bool k0 = (jj == 0) && first0;
                    const double2 zh = k0 ? zk[jj] : zm[jj];          // (k = 0 pairs with itself)
                    // (2 E and 2 O: the halves ride in the scale -- exact; threads without bins scale by 0)
                    const double2 e = make_double2(zk[jj].x + zh.x, zk[jj].y - zh.y);
                    const double2 o = make_double2(zk[jj].y + zh.y, zh.x - zk[jj].x);
                    const double2 wo = cmul(pw[jj], o);
                    const double ar = e.x + wo.x, ai = e.y + wo.y;
                    // (bin N / 2: |Z[N / 2]| -- w^(N/2) = -i turns O into the imaginary part)
                    const double br = k0 ? 2.0 * zm[jj].x : e.x - wo.x, bi = k0 ? 2.0 * zm[jj].y : e.y - wo.y;
                    mg[2 * jj] = mag_sqrt(fma(ar, ar, ai * ai)) * mscale;
                    mg[2 * jj + 1] = mag_sqrt(fma(br, br, bi * bi)) * mscale;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            PAA_TICK(11)
            // (the previous frame's magnitudes stay in registers: twenty doubles.  Through blocks in global memory -- tried: no spills, the
            // same speed -- every store reached the HBM: 7.5 - 12 x the algorithmic traffic, profiles/r06_wgr_variants.txt.)
            if (MODE == 0 && halo) {
#pragma unroll
                for (int i = 0; i < 2 * NJR; ++i) pm[i] = mg[i];
            }
            // ---------------- the next frame's samples: one load per 128-byte line brings them to the L2 / L1 while the features are formed
            // (holding them in registers across the feature stage made the compiler spill them -- one exposed HBM latency per register)
            // (the loaded bytes are only "used" at the end of the iteration: nothing waits for them before)
            int touch = 0;
            {
                // (the run's last frame touches its own samples again: no branch around the loads)
                const char *xn = reinterpret_cast<const char *>(xc + (long long)(t + 1 < t_end ? t + 1 : t) * P.S);
                constexpr int kLines = W * (int)sizeof(T) / 128, kPer = (kLines + SH::NT - 1) / SH::NT;
#pragma unroll
                for (int u = 0; u < kPer; ++u) {
                    const int o = (tq + u * SH::NT) * 128;
                    touch |= *reinterpret_cast<const char *>(xn + (o < W * (int)sizeof(T) ? o : 0));
                }
            }
            PAA_TICK(5)
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
                asm volatile("" ::"v"(touch));
                continue;
            }
            // ---------------- sums over the thread's bins (:57-82, :110-124)
            double sXt = 0.0, sIXt = 0.0, mxt = 0.0;
            const double kd1 = (double)(jf + 1);           // k + 1 of the thread's first bin; N - k + 1 = (N + 2) - (k + 1)
#pragma unroll
            for (int jj = 0; jj < NJR; ++jj) {
                const double kl = kd1 + (double)(J1 * jj), kh = ((jj == 0) && first0) ? (double)(N / 2 + 1) : (double)(N + 2) - kl;
                sXt += mg[2 * jj] + mg[2 * jj + 1];
                sIXt = fma(kl, mg[2 * jj], sIXt);
                sIXt = fma(kh, mg[2 * jj + 1], sIXt);
                mxt = fmax(mxt, fmax(mg[2 * jj], mg[2 * jj + 1]));
            }
            {
                double2 pr[1] = {make_double2(sXt, sIXt)};
                const double2 st = pair_sums<1>(pr, ws, lane);
                sXt = st.x; sIXt = st.y;
            }
            if (MODE == 0) mxt = wmax_nonneg(mxt);
            __syncthreads();              // every pair has been read: the magnitudes may overwrite the buffer
            if (lane == 0) { red[16 * wv] = sXt; red[16 * wv + 1] = sIXt; red[16 * wv + 2] = mxt; }
            if (!halo && a1) {
#pragma unroll
                for (int jj = 0; jj < NJR; ++jj) {
                    const int k = jf + J1 * jj;
                    mags[k] = mg[2 * jj];
                    mags[((jj == 0) && first0) ? N / 2 : N - k] = mg[2 * jj + 1];
                }
            }
            // the weights of the thread's mel bins (frame-invariant, but sixteen doubles held across the passes would be spilled: the index
            // is opaque, the loads are issued here and land under the barrier and the scan)
            // ... and the thread's two chroma gather entries (classes w and w + NW of its wave; w + NW >= 12: class w once more, weight unused)
            int ch_s0, ch_s1;
            double ch_w0, ch_w1;
            {
                const int c1 = (wv + NW < 12) ? wv + NW : wv;
                ch_s0 = tab->ch_src[wv][ln]; ch_w0 = tab->ch_w[wv][ln];
                ch_s1 = tab->ch_src[c1][ln]; ch_w1 = (wv + NW < 12) ? tab->ch_w[c1][ln] : 0.0;
            }
            double mw[kMelPerLane];
            int mkb = mj_kb, mst = mj_st, mn = mj_n;
            asm volatile("" : "+v"(mkb), "+v"(mst), "+v"(mn));
            if (MODE == 0) {
                int eb = mj_eb;
                asm volatile("" : "+v"(eb));
                const double *wp_ = P.mel_w + eb;
#pragma unroll
                for (int j = 0; j < kMelPerLane; ++j) mw[j] = wp_[(j < mn ? j : 0) * mst];
            }
            __syncthreads();
            double sX = 0.0, sIX = 0.0, mx = 0.0;
#pragma unroll
            for (int w = 0; w < NW; ++w) { sX += red[16 * w]; sIX += red[16 * w + 1]; mx = fmax(mx, red[16 * w + 2]); }
            if (halo) {
                sXp = sX;
                __syncthreads();          // (red is rewritten by the next frame's time-domain stage)
                asm volatile("" ::"v"(touch));
                continue;
            }
            PAA_TICK(6)
            // ---------------- one scan of the row: chunk energies -> running energy at every chunk start
            double cs = 0.0;
            {
                const double2 *m2 = reinterpret_cast<const double2 *>(mags + jf * C);
#pragma unroll
                for (int i = 0; i < C / 2; ++i) {
                    const double2 mm = m2[i];
                    cs += mm.x * mm.x;
                    cs += mm.y * mm.y;
                }
                cs = a1 ? cs : 0.0;
            }
            const double incl = wscan_incl(cs);
            if (lane == 63) slot[wv] = incl;
            if (MODE == 0) {
                // ---------------- MFCC filter sums (:236-254), first half: the thread's lane job -- bins kb + j stride of ONE filter
                double a = 0.0;
#pragma unroll
                for (int j = 0; j < kMelPerLane; ++j) a = fma(mags[mkb + (j < mn ? j : 0) * mst], (j < mn) ? mw[j] : 0.0, a);
                part[tq] = a;
            }

            PAA_TICK(7)
            __syncthreads();
            double run_e = incl - cs, sP = 0.0;
#pragma unroll
            for (int w = 0; w < NW; ++w) { run_e += (w < wave) ? slot[w] : 0.0; sP += slot[w]; }
            if (MODE == 2) {
                // ---------------- chromagram row (:356-359)
                {
                    const double x0 = mags[ch_s0], x1 = mags[ch_s1];
                    double2 pr[1] = {make_double2((x0 * x0) * ch_w0, (x1 * x1) * ch_w1)};
                    const double2 ac = pair_sums<1>(pr, ws, lane);
                    const double mineq = (lane == 0) ? ac.x : ac.y;
                    if (lane == 0 || (lane == 1 && wave + NW < 12))
                        oc[(long long)t * 12 + wave + NW * lane] = (sP == 0.0) ? mineq / kEps : fast_div(mineq, sP);
                }
                __syncthreads();          // the row has been read: the next frame may write the buffer
                asm volatile("" ::"v"(touch));
                continue;
            }
            // ---------------- MFCC filter sums, second half: wave w adds the partial sums of the filters w, w + NW, ..: eight lanes per filter
            {
                const int jf_ = ln >> 3, i8 = ln & 7, m = wv + NW * jf_;
                const int2 fl = melfil[m < 40 ? m : 39];
                double a = 0.0;
                for (int c = i8; c < fl.y; c += 8) a += part[fl.x + c];
                a += dpp_mov<PAA_DPP_X1>(a);
                a += dpp_mov<PAA_DPP_X2>(a);
                a += dpp_mov<PAA_DPP_HM>(a);
                if (i8 == 0 && m < 40 && jf_ < SH::NFW) msp[m] = fast_log10(a + kEps);
            }
            // the running energy at the ten block boundaries (spectral entropy, :85-107)
            if (a1 && (tq % SH::CB) == 0) bnd[tq / SH::CB] = run_e;
            if (tid == 0) bnd[10] = sP;
            // ---------------- roll-off (:127-140): first k with cumsum(X^2)[k] + eps > 0.9 sum(X^2)
            {
                const double thr = 0.90 * sP;
                int first = 0x7fffffff;
                double rr = run_e;
                // (the chunk once more from LDS: its squares kept in registers since the scan were 40 of the 256)
                const double2 *m2 = reinterpret_cast<const double2 *>(mags + jf * C);
#pragma unroll
                for (int i = 0; i < C / 2; ++i) {
                    const double2 mm = m2[i];
                    rr += mm.x * mm.x;
                    first = (a1 && first == 0x7fffffff && rr + kEps > thr) ? jf * C + 2 * i : first;
                    rr += mm.y * mm.y;
                    first = (a1 && first == 0x7fffffff && rr + kEps > thr) ? jf * C + 2 * i + 1 : first;
                }
                first = mix::wmin_nonneg_i(first);
                if (lane == 0) redi[wv] = first;
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
                    const double kl = kd1 + (double)(J1 * jj), kh = ((jj == 0) && first0) ? (double)(N / 2 + 1) : (double)(N + 2) - kl;
                    const double dl = kl * f0 - cen, dh = kh * f0 - cen;
                    sSp = fma(dl * dl, mg[2 * jj] * r, sSp);
                    sSp = fma(dh * dh, mg[2 * jj + 1] * r, sSp);
                    const double fl = mg[2 * jj] * rX - pm[2 * jj] * rXp, fh = mg[2 * jj + 1] * rX - pm[2 * jj + 1] * rXp;
                    sFl = fma(fl, fl, sFl);
                    sFl = fma(fh, fh, sFl);
                }
                double2 pr[1] = {make_double2(sSp, sFl)};
                const double2 st = pair_sums<1>(pr, ws, lane);
                if (lane == 0) { red2[2 * wv] = st.x; red2[2 * wv + 1] = st.y; }
#pragma unroll
                for (int i = 0; i < 2 * NJR; ++i) pm[i] = mg[i];
            }
            // ---------------- chroma (:277-321): the pitch classes w and w + NW of a wave, one gather entry per lane (registers)
            {
                const double x0 = mags[ch_s0], x1 = mags[ch_s1];
                double2 pr[1] = {make_double2((x0 * x0) * ch_w0, (x1 * x1) * ch_w1)};
                const double2 ac = pair_sums<1>(pr, ws, lane);
                const double mineq = (lane == 0) ? ac.x : ac.y;
                if (lane == 0 || (lane == 1 && wave + NW < 12)) fv[21 + wv + NW * ln] = (sP == 0.0) ? mineq / kEps : fast_div(mineq, sP);
            }
            PAA_TICK(8)
            __syncthreads();
            // ---------------- the last mile, on five waves at once
            if (wave == 0) {
                if (lane < 13) {                                   // DCT (:250)
                    const double *m = dct + lane * 40;
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
            sXp = sX;
            asm volatile("" ::"v"(touch));
            PAA_TICK(9)
        }       // frames of the run
    }       // runs of this workgroup
    PAA_TEND()
}

// ---- host -----------------------------------------------------------------------------------------------------------------------
// 0: kernels_wg.hpp / kernels_big.hpp keep the window; else the shape's id
inline int wgr_shape_id(int window) {
    if (window == S16000::W) return 1;
    if (window == S8000::W) return 2;
    return 0;
}
inline const char *wgr_shape_name(int id) { return id == 1 ? "20x20x20" : "10x20x20"; }
inline int wgr_threads(int id) { return id == 1 ? S16000::NT : S8000::NT; }

// The plan's tables.  Mel lane jobs: filter m gets nl[m] = ceil(cnt[m] / L) consecutive threads, L the smallest bin count per thread
// (at most kMelPerLane) with which the filters' threads fit the workgroup; thread i of the filter takes its bins i, i + nl, i + 2 nl, ...
// (neighbouring threads read neighbouring bins and weights).  Chroma: the gather list of every pitch class padded to 64 entries of
// weight 0.  false: the filters need more threads than the workgroup has even at sixteen bins per thread (a window of more than
// two seconds' worth of mel bins), or a pitch class has more than 64 entries (cannot happen: one entry per semitone the bins reach) -- the
// caller keeps kernels_wg.hpp for the window.
inline bool wgr_build_tab(int nt, const MelTable *mel, const ChromaTable *chroma, WgrTab &t) {
    memset(&t, 0, sizeof(t));
    for (int i = 0; i < kMaxThreads; ++i) t.mel_job[i][2] = 1;
    if (mel) {
        int L = 0;
        for (int l = 1; l <= kMelPerLane && !L; ++l) {
            long need = 0;
            for (int m = 0; m < kNumMel; ++m) need += (mel->cnt[m] + l - 1) / l;
            if (need <= nt) L = l;
        }
        if (!L || nt > kMaxThreads) return false;
        int th = 0;
        for (int m = 0; m < kNumMel; ++m) {
            const int cnt = mel->cnt[m], nl = (cnt + L - 1) / L;
            t.mel_fil[m][0] = th; t.mel_fil[m][1] = nl;
            for (int i = 0; i < nl; ++i, ++th) {
                t.mel_job[th][0] = mel->lo[m] + i;
                t.mel_job[th][1] = mel->off[m] + i;
                t.mel_job[th][2] = nl;
                t.mel_job[th][3] = (cnt - i + nl - 1) / nl;
            }
        }
    }
    if (chroma) {
        for (int c = 0; c < 12; ++c) {
            const int b = chroma->class_start[c], n = chroma->class_start[c + 1] - b;
            if (n > 64) return false;
            t.ch_n[c] = n;
            for (int i = 0; i < n; ++i) { t.ch_src[c][i] = chroma->src[(size_t)(b + i)]; t.ch_w[c][i] = chroma->w[(size_t)(b + i)]; }
        }
    }
    return true;
}

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
                          long long n_runs, int num_cu, const WgrTab *d_tab, double *d_out, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&wgr_kernel<SH, T, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                SH::LDS_BYTES) != hipSuccess) return -1;
        attr_set = true;
    }
    const unsigned grid = (unsigned)std::min<long long>(n_runs, num_cu);
    hipLaunchKernelGGL((wgr_kernel<SH, T, MODE>), dim3(grid), dim3(SH::NT), (size_t)SH::LDS_BYTES, stream, P, (const T *)d_packed, clips,
                       norms, runs, (int)n_runs, d_tab, d_out);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace wgr
}  // namespace paa
