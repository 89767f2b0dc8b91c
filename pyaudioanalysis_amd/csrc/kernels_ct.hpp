// Register-FFT feature kernels for windows W = 2 RA RB, any step, any sample type (int16 / interleaved stereo int16 / float64),
// features / spectrogram / chromagram: the family behind the shapes the reference's own callers use besides the
// int16 800/400 headline (kernels_fast.hpp):
//     W = 800 = 2 x 25 x 16   float64 or stereo input at 50 ms / 16 kHz (audioBasicIO.py:167 hands every stereo file over as
//                             float64), and int16 at steps other than 400 / 800
//     W = 640 = 2 x 20 x 16   40 ms at 16 kHz: the CLI's spectrogram / chromagram / silence removal (audioAnalysis.py:66-81)
//     W = 400 = 2 x 25 x 8    50 ms at 8 kHz (audioTrainTest.py:28-29)
//     W = 320 = 2 x 10 x 16   20 ms at 16 kHz
// One wave = one run of consecutive frames of one clip, FOUR frames ("a quad") per iteration, 16 lanes per frame:
//   load    : lane (frame g, j < RB) fetches the RA complex samples z[j + RB r] = (x[2n], x[2n+1]) of its frame straight
//             from HBM/L2 -- per load instruction the RB lanes of a frame read one contiguous span -- and removes the clip
//             mean (d = x / 2^15 - mean; the 1 / peak factor is applied to the results, not to every sample)
//   time    : the same registers give the frame's energy, the ten entropy-block energies and the sign changes (the lane's
//             samples are pairs 2j + 2 RB r: block membership is static per register row, the sample before a pair is
//             in the lane below) -- 16-lane DPP reductions, results stay in registers until the feature stage
//   pass 1  : radix-RA DFT in registers (25 = 5 x 5 with twiddles; 20 = 4 x 5 and 10 = 2 x 5 as prime-factor transforms)
//   exchange: real plane, then imaginary plane, through the frame's spectrum slot (row stride RA or RA + 1: odd)
//   pass 2  : lane (g, p <= RA / 2): the two radix-RB DFTs of columns p and RA - p (conjugate twiddles, outputs rotated by
//             one), so Z[k] and Z[NC - k] meet in the lane: real-FFT recombination and |X| in registers
//   features: 16 lanes per frame, lane i owns bins [C i, C i + C); spectral-entropy blocks from the cumulative energy at
//             the block boundaries; mel lists / chroma lists / DCT exactly as in kernels_fast.hpp
//   store   : 64-byte chunks of eight frames per row, non-temporal (store_row_chunked of kernels_fast.hpp)
// Halo: a run with t0 > 0 starts ONE frame early (two with deltas: the delta of the flux reaches two spectra back); the
// extra frames ride in the first quad, so a run of 4 m - 1 frames costs exactly m iterations.
//
// Replaces ShortTermFeatures.py:608-682 (+ helpers :22-140, :236-321), and the loops of spectrogram (:415-422) /
// chromagram (:349-359), for these windows.
#pragma once
#include <algorithm>
#include <type_traits>
#include <vector>

#include "device_common.hpp"
#include "kernels_fast.hpp"
#include "tables.hpp"

namespace paa {
namespace ct {

using f800::dft4r;
using f800::dft5r;
#if defined(PAA_F800_TIMING) || defined(PAA_F800_TRACE)
using f800::g_phase_cycles;        // per-phase cycle accounting of diagnostic builds (PAA_T0 / PAA_TICK / PAA_TEND)
using f800::g_wave_trace;
#endif

// ---- shapes ---------------------------------------------------------------------------------------------------
template <int RA_, int RB_>
struct Shape {
    static constexpr int RA = RA_, RB = RB_;
    static constexpr int NC = RA * RB, W = 2 * NC, NF = NC;
    static constexpr int NP = RA / 2 + 1;                       // pass-2 lanes per frame: columns 0 .. RA / 2
    static constexpr int RAP = (RA % 2) ? RA : RA + 1;          // row stride of the exchange plane (odd: conflict-free)
    static constexpr int NFP = ((RAP * RB > NF ? RAP * RB : NF) + 1) & ~1;     // doubles per spectrum slot
    static constexpr int C = (NF + 15) / 16;                    // bins per lane in the feature stage
    static constexpr int LT = W / 10, LB = NF / 10;             // entropy blocks (samples / bins)
    static constexpr bool SCRATCH_IN_RING = 4 * (40 + 34) <= NF;    // msp[4][40] + fv[4][34] fit the previous-spectrum slot
    static_assert(RB == 16 || RB == 8, "pass 2 has radix-16 and radix-8 codelets");
    static_assert(NP <= 16, "the column pairs of a frame must fit its 16 lanes");
    static_assert(W % 20 == 0, "entropy blocks: even length, no tail (ShortTermFeatures.py:37-41)");
    static_assert(NF % 10 == 0 && C <= LB, "a lane's bins may contain at most one block boundary");
    static_assert(LT >= 2 * RB, "a register row of 2 RB samples may contain at most one block boundary");
};

// shared (per workgroup) LDS tables, laid out by the host (same scheme as f800::TabLayout)
struct TabLayout {
    int off_w0, off_k0, off_w1, off_k1, off_w2, off_k2;    // mel classes: filter i | filter 16+i | half of filter 32+(i&7)
    int off_chw, off_chk;                                   // chroma gather list of pitch class i (stride 12)
    int off_dct;                                            // 13 rows padded to 41 doubles
    int off_tw2, off_twp;                                   // double2 [RB][16]: W_NC^(r p) and W_W^(p + RA s) of lane p
    int off_sync;                                           // pacing: SIMD id [8], progress [8]
    int melN0, melN1, melN2, chN;                           // list lengths (multiples of 8)
    int mel_clamp;                                          // 1: some padded list reaches past the last bin
    int pace;
    double f0, rf0, r_half_fs, f0sq;                        // fs / W, its reciprocal, 2 / fs, f0^2
    int total;                                              // bytes, multiple of 16
};

// ---- register DFT codelets: run() transforms v[] in place, X[q] ends at v[pos(q)] -----------------------------
template <int R> struct Dft;
template <> struct Dft<25> {
    template <int SERIAL> static __device__ __forceinline__ void run(double2 *v) { f800::dft25<SERIAL>(v); }
    static constexpr int pos(int q) { return 5 * (q % 5) + q / 5; }
};
// 20 = 4 x 5, coprime: prime-factor transform, no twiddles.  Element (n1, n2) sits at v[(5 n1 + 4 n2) % 20]; after the
// two stages X[(5 k1 + 16 k2) % 20] sits at v[(5 k1 + 4 k2) % 20], i.e. X[q] at (5 (q % 4) + 4 (q % 5)) % 20.
template <> struct Dft<20> {
    template <int SERIAL> static __device__ __forceinline__ void run(double2 *v) {
#pragma unroll
        for (int n2 = 0; n2 < 5; ++n2) {
            dft4r(v[(4 * n2) % 20], v[(5 + 4 * n2) % 20], v[(10 + 4 * n2) % 20], v[(15 + 4 * n2) % 20]);
            if (SERIAL) __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int k1 = 0; k1 < 4; ++k1) {
            dft5r(v[(5 * k1) % 20], v[(5 * k1 + 4) % 20], v[(5 * k1 + 8) % 20], v[(5 * k1 + 12) % 20], v[(5 * k1 + 16) % 20]);
            if (SERIAL) __builtin_amdgcn_sched_barrier(0);
        }
    }
    static constexpr int pos(int q) { return (5 * (q % 4) + 4 * (q % 5)) % 20; }
};
// 10 = 2 x 5: element (n1, n2) at v[(5 n1 + 2 n2) % 10]; X[q] at (5 (q % 2) + 2 (q % 5)) % 10
template <> struct Dft<10> {
    template <int SERIAL> static __device__ __forceinline__ void run(double2 *v) {
#pragma unroll
        for (int n2 = 0; n2 < 5; ++n2) {
            const double2 a = v[(2 * n2) % 10], b = v[(5 + 2 * n2) % 10];
            v[(2 * n2) % 10] = cadd(a, b);
            v[(5 + 2 * n2) % 10] = csub(a, b);
        }
        if (SERIAL) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k1 = 0; k1 < 2; ++k1) {
            dft5r(v[(5 * k1) % 10], v[(5 * k1 + 2) % 10], v[(5 * k1 + 4) % 10], v[(5 * k1 + 6) % 10], v[(5 * k1 + 8) % 10]);
            if (SERIAL) __builtin_amdgcn_sched_barrier(0);
        }
    }
    static constexpr int pos(int q) { return (5 * (q % 2) + 2 * (q % 5)) % 10; }
};
template <> struct Dft<16> {
    template <int SERIAL> static __device__ __forceinline__ void run(double2 *v) { f800::dft16<SERIAL>(v); }
    static constexpr int pos(int q) { return 4 * (q % 4) + q / 4; }
};
// 8 = 4 (even) + 4 (odd) with the W8 twiddles on the odd half: X[k] at v[2 k], X[k + 4] at v[2 k + 1]
template <> struct Dft<8> {
    template <int SERIAL> static __device__ __forceinline__ void run(double2 *v) {
        const double h = 0.70710678118654752;
        dft4r(v[0], v[2], v[4], v[6]);
        if (SERIAL) __builtin_amdgcn_sched_barrier(0);
        dft4r(v[1], v[3], v[5], v[7]);
        if (SERIAL) __builtin_amdgcn_sched_barrier(0);
        v[3] = make_double2(h * (v[3].x + v[3].y), h * (v[3].y - v[3].x));      // W8   = (h, -h)
        v[5] = make_double2(v[5].y, -v[5].x);                                   // W8^2 = -i
        v[7] = make_double2(h * (v[7].y - v[7].x), -h * (v[7].x + v[7].y));     // W8^3 = (-h, -h)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const double2 e = v[2 * k], o = v[2 * k + 1];
            v[2 * k] = cadd(e, o);
            v[2 * k + 1] = csub(e, o);
        }
    }
    static constexpr int pos(int q) { return q < 4 ? 2 * q : 2 * (q - 4) + 1; }
};

// ---- two consecutive samples as one load (element alignment only: frames start anywhere) -----------------------
template <typename T> struct PairLoad;
template <> struct PairLoad<int16_t> {
    typedef short vec __attribute__((ext_vector_type(2), aligned(2)));
    static __device__ __forceinline__ double2 get(const int16_t *p) {
        const vec s = *reinterpret_cast<const vec *>(p);
        return make_double2((double)s.x, (double)s.y);
    }
};
template <> struct PairLoad<stereo16> {      // two stereo frames = 8 bytes, each summed L + R in the load
    typedef int vec __attribute__((ext_vector_type(2), aligned(4)));
    static __device__ __forceinline__ double2 get(const stereo16 *p) {
        const vec s = *reinterpret_cast<const vec *>(p);
        return make_double2((double)stereo_word_sum(s.x), (double)stereo_word_sum(s.y));
    }
};
template <> struct PairLoad<double> {
    typedef double vec __attribute__((ext_vector_type(2), aligned(8)));
    static __device__ __forceinline__ double2 get(const double *p) {
        const vec s = *reinterpret_cast<const vec *>(p);
        return make_double2(s.x, s.y);
    }
};

// the same pair as integers (int16 PCM: the two halves of one word; stereo: L + R of two frames); float64 has none
template <typename T> struct PairRaw;
template <> struct PairRaw<int16_t> {
    int w;
    static __device__ __forceinline__ PairRaw get(const int16_t *p) {
        typedef int w32 __attribute__((aligned(2)));
        PairRaw r; r.w = *reinterpret_cast<const w32 *>(p); return r;
    }
    __device__ __forceinline__ int x0() const { return (int)(short)(w & 0xffff); }
    __device__ __forceinline__ int x1() const { return w >> 16; }
};
template <> struct PairRaw<stereo16> {
    int a, b;
    static __device__ __forceinline__ PairRaw get(const stereo16 *p) {
        const PairLoad<stereo16>::vec s = *reinterpret_cast<const PairLoad<stereo16>::vec *>(p);
        PairRaw r; r.a = s.x; r.b = s.y; return r;
    }
    __device__ __forceinline__ int x0() const { return stereo_word_sum(a); }
    __device__ __forceinline__ int x1() const { return stereo_word_sum(b); }
};
template <> struct PairRaw<double> {
    static __device__ __forceinline__ PairRaw get(const double *) { return PairRaw(); }
    __device__ __forceinline__ int x0() const { return 0; }
    __device__ __forceinline__ int x1() const { return 0; }
};

// MODE 0: short-term features (DELTAS: 68 rows), 1: spectrogram rows, 2: chromagram rows.
// NW = waves per workgroup (8: two per SIMD, paced like the 800/400 kernel).
template <typename SH, typename T, int MODE, int DELTAS, int NW>
__global__ __launch_bounds__(64 * NW, NW / 4) void st_ct_kernel(PlanDev P, TabLayout L,
                                                               const unsigned char *__restrict__ blob,
                                                               const T *__restrict__ sig,
                                                               const ClipDev *__restrict__ clips,
                                                               const ClipNorm *__restrict__ norms,
                                                               const Tile *__restrict__ tiles, int n_tiles,
                                                               double *__restrict__ out) {
    constexpr int RA = SH::RA, RB = SH::RB, NC = SH::NC, W = SH::W, NF = SH::NF, NP = SH::NP, RAP = SH::RAP, NFP = SH::NFP;
    constexpr int C = SH::C, LT = SH::LT, LB = SH::LB, QUAD = 4, FV = 34;
    constexpr int HALO = (MODE == 0) ? (DELTAS ? 2 : 1) : 0;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    {
        const int4 *src4 = reinterpret_cast<const int4 *>(blob);
        int4 *dst4 = reinterpret_cast<int4 *>(smem);
        for (int n = threadIdx.x; n < L.total / 16; n += 64 * NW) dst4[n] = src4[n];
    }
    __syncthreads();       // the only workgroup-wide barrier (plus the pacing hand-shake below)
    const double *t_melw0 = reinterpret_cast<const double *>(smem + L.off_w0);
    const int *t_melk0 = reinterpret_cast<const int *>(smem + L.off_k0);
    const double *t_melw1 = reinterpret_cast<const double *>(smem + L.off_w1);
    const int *t_melk1 = reinterpret_cast<const int *>(smem + L.off_k1);
    const double *t_melw2 = reinterpret_cast<const double *>(smem + L.off_w2);
    const int *t_melk2 = reinterpret_cast<const int *>(smem + L.off_k2);
    const double *t_chw = reinterpret_cast<const double *>(smem + L.off_chw);
    const int *t_chk = reinterpret_cast<const int *>(smem + L.off_chk);
    const double *t_dct = reinterpret_cast<const double *>(smem + L.off_dct);
    const double2 *t_tw2 = reinterpret_cast<const double2 *>(smem + L.off_tw2);
    const double2 *t_twp = reinterpret_cast<const double2 *>(smem + L.off_twp);

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tile_id = blockIdx.x * NW + wave;
    // the two waves of a SIMD are paced against each other (see st_fast_800_kernel)
    volatile int *pace = reinterpret_cast<volatile int *>(smem + L.off_sync);
    int partner = wave;
    if (NW == 8) {
        const int my_simd = (int)((__builtin_amdgcn_s_getreg((31 << 11) | 4) >> 4) & 3u);     // HW_REG_HW_ID[5:4]
        if ((threadIdx.x & 63) == 0) {
            pace[wave] = my_simd;
            pace[8 + wave] = (tile_id < n_tiles) ? 0 : 0x7fffffff;
        }
        __syncthreads();
#pragma unroll
        for (int w = 0; w < 8; ++w) partner = (w != wave && pace[w] == my_simd) ? w : partner;
        partner = __builtin_amdgcn_readfirstlane(partner);
    }
    if (tile_id >= n_tiles) return;
    // per-wave LDS: 5 spectrum slots, the block-boundary scratch, msp / fv (inside the previous-spectrum slot when it fits)
    constexpr int WAVE_DOUBLES = 5 * NFP + QUAD * 12 + (SH::SCRATCH_IN_RING ? 0 : QUAD * (40 + FV));
    double *spec = reinterpret_cast<double *>(smem + L.total) + (size_t)wave * WAVE_DOUBLES;
    double *bnd = spec + 5 * NFP;                       // [4][12]: cumulative energy at the block boundaries 0 .. 10
    double *msp = bnd + QUAD * 12;
    double *fv = msp + QUAD * 40;

    const int lane = threadIdx.x & 63;
    int g = lane >> 4, i = lane & 15;
    const Tile tl = tiles[tile_id];
    const ClipDev c = clips[tl.clip];
    const ClipNorm nm = wave_clip_norm<T>(P, c, norms, tl.clip, lane);
    const T *xc = sig + c.sample_off + P.frame_origin;
    const long long Tc = c.T;
    double *oc = out + c.out_off;
    const int S = P.S;
    const double sc = sample_scale<T>();
    const double mean = nm.mean, inv = nm.inv;
    const double mscale = 0.5 * inv / (double)NF;         // E and O carry 1/2; X / len(X) (:621); y = d * inv
    const double f0 = L.f0, rf0 = L.rf0, r_half_fs = L.r_half_fs, f0sq = L.f0sq;
    constexpr bool INT_T = !std::is_same<T, double>::value;
    SignRule sr = {0, 0, 0};
    if (INT_T && MODE == 0) sr = sign_rule<T>(nm.mean);

    const int r0 = tl.t0, t_end = tl.t0 + tl.cnt;         // frames [r0, t_end) are this wave's to store
    int slot0 = 2;                   // slots of a quad: slot0 .. slot0+3 (mod 5) after the rotation at the loop head; previous = slot0-1
    double vlast = 0.0;              // lane l < 34: feature l of the frame before this quad
    double hold[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    double holdd[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    int n_done = 0;
    PAA_T0()
    for (int q0 = (r0 >= HALO) ? r0 - HALO : 0; q0 < t_end; q0 += QUAD) {
        slot0 = (slot0 + 4) % 5;
        if (NW != 4) asm volatile("" : "+v"(g), "+v"(i));
#define PAA_CT_PACE(half_)                                                                             \
        if (NW == 8 && L.pace != 0) {                                                                  \
            const int mine_ = 2 * n_done + (half_);                                                    \
            if (lane == 0) pace[8 + wave] = mine_;                                                     \
            const int other_ = __builtin_amdgcn_readfirstlane(pace[8 + partner]);                      \
            const int d_ = mine_ - other_;                                                             \
            if (d_ < 0) __builtin_amdgcn_s_setprio(3);                                                 \
            else if (d_ > 0) __builtin_amdgcn_s_setprio(0);                                            \
            else __builtin_amdgcn_s_setprio(1);                                                        \
        }
        PAA_CT_PACE(0)
        const int t = q0 + g;                                   // this group's frame
        const bool act1 = (RB == 16) || (i < RB);               // pass-1 / time-stage lanes
        const int pa = i, pb = (i == 0) ? 0 : RA - i;
        const bool act2 = i < NP;
        const int itw = min(i, NP - 1);                         // idle lanes re-read a valid table column
        const int ich = min(i, 11);

        // ---------------- load: z[j + RB r] of frame t, mean removed (frames past the clip's last repeat it: never stored)
        double2 v[RA];
        PairRaw<T> raw[(INT_T && MODE == 0) ? RA : 1];       // integer samples: converted row by row in the time-domain stage
        // idle lanes (RB = 8) re-read lane 0's samples with scale and mean 0: exact zeros, no energy, nothing to mask
        const double scl = act1 ? sc : 0.0, meanl = act1 ? mean : 0.0;
        {
            const long long tt = (t < Tc) ? t : Tc - 1;
            const T *xf = xc + tt * (long long)S + 2 * (act1 ? i : 0);
#pragma unroll
            for (int r = 0; r < RA; ++r) {
                if constexpr (INT_T && MODE == 0) {
                    raw[r] = PairRaw<T>::get(xf + 2 * RB * r);
                } else {
                    const double2 x = PairLoad<T>::get(xf + 2 * RB * r);
                    v[r] = make_double2(fma(x.x, scl, -meanl), fma(x.y, scl, -meanl));
                }
            }
        }
#ifdef PAA_F800_TIMING
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        PAA_TICK(0)
        // ---------------- time domain (ShortTermFeatures.py:22-51) on the same registers
        double e_tot = 0.0, ent_e = 0.0;
        int zc = 0;
        if (MODE == 0) {
            double eb[10];
#pragma unroll
            for (int b = 0; b < 10; ++b) eb[b] = 0.0;
            int prev_b = 0;
#pragma unroll
            for (int r = 0; r < RA; ++r) {
                int x0 = 0, x1 = 0;
                if constexpr (INT_T) {
                    x0 = raw[r].x0(); x1 = raw[r].x1();
                    v[r] = make_double2(fma((double)x0, scl, -meanl), fma((double)x1, scl, -meanl));
                }
                const double d0 = v[r].x, d1 = v[r].y;
                const double e = fma(d0, d0, d1 * d1);
                // samples 2 RB r + 2 i, + 1: block jlo for lanes below jth, jlo + 1 from there on (static per row)
                const int jlo = (2 * RB * r) / LT;
                const int jth = ((jlo + 1) * LT - 2 * RB * r) / 2;
                if (jth >= RB) {
                    eb[jlo] += e;
                } else {
                    eb[jlo] += (i < jth) ? e : 0.0;
                    eb[(jlo + 1 < 10) ? jlo + 1 : 9] += (i >= jth) ? e : 0.0;
                }
                // sign codes (device_common.hpp): only |differences| are summed
                int sa, sb;
                if constexpr (INT_T) { sa = sgn1(x0, sr); sb = sgn1(x1, sr); }
                else { sa = sgn1(d0); sb = sgn1(d1); }
                // the sample before the pair: the lane below; lane 0 takes the last active lane of the previous row (the frame's first
                // sample has no left one: it meets itself)
                const int from_left = __builtin_amdgcn_update_dpp(0, sb, 0x111, 0xF, 0xF, true);                  // row_shr:1
                int left;
                if (r == 0) {
                    left = (i == 0) ? sa : from_left;
                } else {
                    const int from_prev = (RB == 16) ? __builtin_amdgcn_update_dpp(0, prev_b, PAA_DPP_RM, 0xF, 0xF, true)
                                                     : __builtin_amdgcn_update_dpp(0, prev_b, PAA_DPP_HM, 0xF, 0xF, true);
                    left = (i == 0) ? from_prev : from_left;
                }
                sad_acc(zc, sb, sa);
                sad_acc(zc, sa, left);
                prev_b = sb;
            }
#pragma unroll
            for (int b = 0; b < 10; ++b) eb[b] = group_sum(eb[b]);
            zc = group_sum_i(act1 ? zc : 0) << (INT_T ? sr.sh : 0);      // (integer codes 1 / 2: differences count double)
            const double inv2 = inv * inv;
#pragma unroll
            for (int b = 0; b < 10; ++b) { eb[b] *= inv2; e_tot += eb[b]; }
            double mine = 0.0;
#pragma unroll
            for (int b = 0; b < 10; ++b) mine = (i == b) ? eb[b] : mine;
            const double se = fast_div(mine, e_tot + kEps);
            ent_e = group_sum((i < 10) ? -(se * fast_log2(se + kEps)) : 0.0);
        }
        PAA_TICK(1)
        // ---------------- pass 1: radix-RA over r
        if (act1) Dft<RA>::template run<(NW != 4)>(v);
        wsync();        // the previous quad's readers of the slots are done
        PAA_TICK(2)

        // exchange through the quad's 4 spectrum slots: real plane, then imaginary plane; element (j, q) at RAP j + q
        double ax[RB], ay[RB], bx[RB], by[RB];
        {
            double *pl = spec + ((slot0 + g) % 5) * NFP;
            if (act1) {
#pragma unroll
                for (int q = 0; q < RA; ++q) pl[RAP * i + q] = v[Dft<RA>::pos(q)].x;
            }
            wsync();
#pragma unroll
            for (int r = 0; r < RB; ++r) { ax[r] = pl[pa + RAP * r]; bx[r] = pl[pb + RAP * r]; }
            wsync();
            if (act1) {
#pragma unroll
                for (int q = 0; q < RA; ++q) pl[RAP * i + q] = v[Dft<RA>::pos(q)].y;
            }
            wsync();
#pragma unroll
            for (int r = 0; r < RB; ++r) { ay[r] = pl[pa + RAP * r]; by[r] = pl[pb + RAP * r]; }
            wsync();
        }
        PAA_TICK(3)
        // ---------------- pass 2 + real-FFT recombination + magnitude (ShortTermFeatures.py:617-621)
        if (act2) {
            double2 a[RB], b[RB];
            a[0] = make_double2(ax[0], ay[0]);
            b[0] = make_double2(bx[0], by[0]);
            double *sp = spec + ((slot0 + g) % 5) * NFP;
            // column RA - p (and the self-paired columns 0 and RA / 2) use the conjugate twiddles of column p, which leaves
            // their outputs rotated by one: Z[NC - k] for k = p + RA s is b[(RB - s) % RB]
#pragma unroll
            for (int rr = 1; rr < RB; rr += 3) {
                double2 wg[3];
#pragma unroll
                for (int u = 0; u < 3; ++u)
                    if (rr + u < RB) wg[u] = t_tw2[(rr + u) * 16 + itw];
#pragma unroll
                for (int u = 0; u < 3; ++u)
                    if (rr + u < RB) {
                        a[rr + u] = cmul(make_double2(ax[rr + u], ay[rr + u]), wg[u]);
                        b[rr + u] = cmul(make_double2(bx[rr + u], by[rr + u]), make_double2(wg[u].x, -wg[u].y));
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
            Dft<RB>::template run<1>(a);
            Dft<RB>::template run<1>(b);
#pragma unroll
            for (int r = 0; r < RB; ++r) asm volatile("" : "+v"(a[r].x), "+v"(a[r].y), "+v"(b[r].x), "+v"(b[r].y));
#pragma unroll
            for (int s0 = 0; s0 < RB; s0 += 2) {
                double2 wg[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) wg[u] = t_twp[(s0 + u) * 16 + itw];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int s = s0 + u;
                    const double2 zk = a[Dft<RB>::pos(s)];
                    const double2 zb = b[Dft<RB>::pos((RB - s) % RB)];
                    const int k = pa + RA * s;
                    // 2E = Z[k] + conj Z[NC-k],  2O = -i (Z[k] - conj Z[NC-k]);  X[k] = E + w^k O,  X[NC-k] = conj(E - w^k O)
                    const double2 e = make_double2(zk.x + zb.x, zk.y - zb.y);
                    const double2 o = make_double2(zk.y + zb.y, zb.x - zk.x);
                    const double2 tw = cmul(wg[u], o);
                    const double xr = e.x + tw.x, xi = e.y + tw.y, yr = e.x - tw.x, yi = e.y - tw.y;
                    const double mk = mag_sqrt(fma(xr, xr, xi * xi)) * mscale;
                    const double mm = mag_sqrt(fma(yr, yr, yi * yi)) * mscale;
                    sp[k] = mk;
                    // bin NC - k; the DC lane has no partner bin at s = 0 (the Nyquist bin is dropped): it stores |X[0]| twice
                    if (s == 0) sp[(i == 0) ? 0 : NF - k] = (i == 0) ? mk : mm;
                    else sp[NF - k] = mm;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        wsync();
        ++n_done;
        PAA_CT_PACE(-1)
        PAA_TICK(4)

        if (MODE == 1) {
            // ---------------- spectrogram rows (ShortTermFeatures.py:422): one row at a time with the whole wave
#pragma unroll 1
            for (int f = 0; f < QUAD; ++f) {
                const int tf = q0 + f;
                if (tf < r0 || tf >= t_end) continue;
                const double *cur = spec + ((slot0 + f) % 5) * NFP;
                double *row = oc + (long long)tf * NF;
                for (int k = lane; k < NF; k += 64) __builtin_nontemporal_store(cur[k], row + k);
            }
            wsync();
            continue;
        }
        // ---------------- spectrum sweeps: 16 lanes per frame, lane i owns bins [C i, C i + C)
        const double *cur = spec + ((slot0 + g) % 5) * NFP;
        const double *prv = (t == 0) ? cur : spec + ((slot0 + g + 4) % 5) * NFP;
        const int kb = C * i;
        double Xc[C], Xv[C];
#pragma unroll
        for (int m = 0; m < C; ++m) {
            if (16 * C == NF) {
                Xc[m] = cur[kb + m];
                Xv[m] = (MODE == 0) ? prv[kb + m] : 0.0;
            } else {
                const int k = min(kb + m, NF - 1);
                const double a = cur[k], b = (MODE == 0) ? prv[k] : 0.0;
                Xc[m] = (kb + m < NF) ? a : 0.0;
                Xv[m] = (kb + m < NF) ? b : 0.0;
            }
        }
        if (MODE == 0 && SH::SCRATCH_IN_RING) {
            wsync();       // the previous-spectrum slot becomes msp[] / fv[]
            msp = spec + ((slot0 + 4) % 5) * NFP;
            fv = msp + QUAD * 40;
        }
        double sXa = 0.0, sXb = 0.0, sMa = 0.0, sMb = 0.0, sVa = 0.0, sVb = 0.0, mx = 0.0, csa = 0.0, csb = 0.0;
#pragma unroll
        for (int m = 0; m + 1 < C; m += 2) {
            const double X0 = Xc[m], X1 = Xc[m + 1];
            sXa += X0; sXb += X1;
            sVa += Xv[m]; sVb += Xv[m + 1];
            sMa = fma((double)m, X0, sMa); sMb = fma((double)(m + 1), X1, sMb);
            csa = fma(X0, X0, csa); csb = fma(X1, X1, csb);
            mx = fmax(mx, fmax(X0, X1));
        }
        if (C & 1) {
            const double X0 = Xc[C - 1];
            sXa += X0; sVa += Xv[C - 1]; sMa = fma((double)(C - 1), X0, sMa); csa = fma(X0, X0, csa); mx = fmax(mx, X0);
        }
        const double cs = csa + csb;
        const double run_incl = group_scan_incl(cs);
        const double run_excl = run_incl - cs;
        const double sP = group_max(run_incl);              // total = the largest entry of a non-decreasing scan
        if (MODE == 2) {
            // ---------------- chromagram row (:356-359)
            double chroma = 0.0;
            for (int n = 0; n < L.chN; n += 8) {
                int kv[8];
                double wv[8], xv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { kv[u] = t_chk[(n + u) * 12 + ich]; wv[u] = t_chw[(n + u) * 12 + ich]; }
#pragma unroll
                for (int u = 0; u < 8; ++u) xv[u] = cur[kv[u]];
#pragma unroll
                for (int u = 0; u < 8; ++u) chroma = fma(xv[u] * xv[u], wv[u], chroma);
            }
            chroma = (sP == 0.0) ? chroma / kEps : fast_div(chroma, sP);
            if (i < 12 && t >= r0 && t < t_end) oc[(long long)t * 12 + i] = chroma;
            wsync();
            continue;
        }
        // ---------------- features (:57-140, :236-321)
        const double base_k = (double)(kb + 1);
        double sX = sXa + sXb;
        double sIX = f0 * fma(base_k, sX, sMa + sMb);            // sum (k + 1) f0 X
        double sXp = sVa + sVb;
        sX = group_sum(sX); sXp = group_sum(sXp);
        sIX = group_sum(sIX); mx = group_max(mx);
        const double sXe = sX + (double)NF * kEps;               // np.sum(X + eps) (:118-119)
        sXp += (double)NF * kEps;
        // spectral entropy (:85-107): cumulative energy at the block boundaries j LB, j = 0 .. 9, written by the lane whose
        // bins contain the boundary; boundary 10 is the total
        // -- and the roll-off (:127-140: first k with cumsum(X^2)[k] + eps > 0.9 sum(X^2)) from the same running energy: it
        // never decreases, so the first bin that qualifies is the number of bins that do not (the last bin below NF always
        // qualifies, bin 0 does when the total is 0: the zero bins past NF never decide)
        double *bg = bnd + 12 * g;
        int below = 0;
        {
            const int jb = (kb + LB - 1) / LB;                   // first boundary at or after the lane's first bin
            const int mb = jb * LB - kb;
            const double thr = 0.90 * sP;
            double run = run_excl, cumb = run_excl;
#pragma unroll
            for (int m = 0; m < C; ++m) {
                cumb = (m == mb) ? run : cumb;
                run = fma(Xc[m], Xc[m], run);
                below += (run + kEps > thr) ? 0 : 1;
            }
            if (mb < C && jb < 10 && kb < NF) bg[jb] = cumb;
        }
        wsync();
        double ent_f;
        {
            const int ib = min(i, 9);
            const double hi_ = (i >= 9) ? sP : bg[ib + 1], lo_ = bg[ib];
            const double sf = fast_div(hi_ - lo_, sP + kEps);
            ent_f = group_sum((i < 10) ? -(sf * fast_log2(sf + kEps)) : 0.0);
        }
        PAA_TICK(5)
        // centroid, spread, flux (:57-82, :110-124)
        const double r = (mx == 0.0) ? 1.0 / kEps : fast_div(1.0, mx);
        const double den = sX * r + kEps;
        const double rden = fast_div(1.0, den);
        const double cen = (sIX * r) * rden;
        const double rX = fast_div(1.0, sXe), rXp = fast_div(1.0, sXp);
        const double cb = base_k - cen * rf0;
        double sSa = 0.0, sSb = 0.0, sFa = 0.0, sFb = 0.0;
#pragma unroll
        for (int m = 0; m + 1 < C; m += 2) {
            const double d0 = cb + (double)m, d1 = cb + (double)(m + 1);
            sSa = fma(d0 * d0, Xc[m], sSa);
            sSb = fma(d1 * d1, Xc[m + 1], sSb);
            const double f0d = Xc[m] * rX - Xv[m] * rXp, f1d = Xc[m + 1] * rX - Xv[m + 1] * rXp;
            sFa = fma(f0d, f0d, sFa);
            sFb = fma(f1d, f1d, sFb);
        }
        if (C & 1) {
            const double d0 = cb + (double)(C - 1);
            sSa = fma(d0 * d0, Xc[C - 1], sSa);
            const double f0d = Xc[C - 1] * rX - Xv[C - 1] * rXp;
            sFa = fma(f0d, f0d, sFa);
        }
        double sSp = (sSa + sSb) * (f0sq * r), sFl = sFa + sFb;
        sSp = group_sum(sSp);
        sFl = group_sum(sFl);
        const double spread = fast_sqrt(sSp * rden);
        const int first = group_min_i((below < C) ? kb + below : 0x7fffffff);
        PAA_TICK(6)
        // MFCC (:236-254): per-lane padded mel lists: class 0 = filter i, class 1 = filter 16 + i, class 2 = one half of
        // filter 32 + (i & 7); the halves meet through a row rotation by 8
        double *mg = msp + 40 * g;
        {
            double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0;
            const int lo0 = t_melk0[i], lo1 = t_melk1[i], lo2 = t_melk2[i];
#define PAA_CT_MEL(acc, lo, N, tw, IDX)                                                                 \
            for (int n = 0; n < (N); n += 8) {                                                          \
                double xv_[8], wv_[8];                                                                  \
                _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                         \
                    xv_[u] = cur[IDX((lo) + n + u)];                                                    \
                    wv_[u] = (tw)[(n + u) * 16 + i];                                                    \
                }                                                                                       \
                double ea_ = 0.0, eb_ = 0.0;                                                            \
                _Pragma("unroll") for (int u = 0; u < 8; u += 2) {                                      \
                    ea_ = fma(xv_[u], wv_[u], ea_);                                                     \
                    eb_ = fma(xv_[u + 1], wv_[u + 1], eb_);                                             \
                }                                                                                       \
                acc += ea_ + eb_;                                                                       \
            }
#define PAA_CT_CLAMP(k) min((k), NF - 1)
#define PAA_CT_PLAIN(k) (k)
            if (L.mel_clamp) {
                PAA_CT_MEL(acc0, lo0, L.melN0, t_melw0, PAA_CT_CLAMP)
                PAA_CT_MEL(acc1, lo1, L.melN1, t_melw1, PAA_CT_CLAMP)
                PAA_CT_MEL(acc2, lo2, L.melN2, t_melw2, PAA_CT_CLAMP)
            } else {
                PAA_CT_MEL(acc0, lo0, L.melN0, t_melw0, PAA_CT_PLAIN)
                PAA_CT_MEL(acc1, lo1, L.melN1, t_melw1, PAA_CT_PLAIN)
                PAA_CT_MEL(acc2, lo2, L.melN2, t_melw2, PAA_CT_PLAIN)
            }
#undef PAA_CT_CLAMP
#undef PAA_CT_PLAIN
#undef PAA_CT_MEL
            acc2 += dpp_mov<0x128>(acc2);                    // row_ror:8
            mg[i] = fast_log10(acc0 + kEps);
            mg[16 + i] = fast_log10(acc1 + kEps);
            const double l2 = fast_log10(acc2 + kEps);
            if (i < 8) mg[32 + i] = l2;
        }
        PAA_TICK(7)
        // chroma (:277-321): lane i < 12 = pitch class i, padded gather list in ascending slot order
        double chroma = 0.0;
        {
            for (int n = 0; n < L.chN; n += 8) {
                int kv[8];
                double wv[8], xv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { kv[u] = t_chk[(n + u) * 12 + ich]; wv[u] = t_chw[(n + u) * 12 + ich]; }
#pragma unroll
                for (int u = 0; u < 8; ++u) xv[u] = cur[kv[u]];
#pragma unroll
                for (int u = 0; u < 8; ++u) chroma = fma(xv[u] * xv[u], wv[u], chroma);     // ascending slot order (:299-302)
            }
            chroma = (sP == 0.0) ? chroma / kEps : fast_div(chroma, sP);
            if (i >= 12) chroma = 0.0;
        }
        wsync();
        PAA_TICK(8)
        double *fg = fv + FV * g;
        if (i < 13) {
            const double *dm = t_dct + 41 * i;        // rows padded to 41 doubles: conflict-free across lanes
            double c0 = 0.0, c1 = 0.0, c2 = 0.0, c3 = 0.0;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                double dv[20], mv[20];
#pragma unroll
                for (int n = 0; n < 20; ++n) { dv[n] = dm[20 * h + n]; mv[n] = mg[20 * h + n]; }
#pragma unroll
                for (int n = 0; n < 20; n += 4) {
                    c0 = fma(dv[n], mv[n], c0);
                    c1 = fma(dv[n + 1], mv[n + 1], c1);
                    c2 = fma(dv[n + 2], mv[n + 2], c2);
                    c3 = fma(dv[n + 3], mv[n + 3], c3);
                }
            }
            fg[8 + i] = (c0 + c1) + (c2 + c3);
        }
        if (i < 12) fg[21 + i] = chroma;
        if (i == 15) {
            fg[0] = ((double)zc * 0.5) * (1.0 / (double)(W - 1));
            fg[1] = e_tot * (1.0 / (double)W);
            fg[2] = ent_e;
            fg[3] = cen * r_half_fs;
            fg[4] = spread * r_half_fs;
            fg[5] = ent_f;
            fg[6] = (t == 0) ? 0.0 : sFl;      // first frame: previous spectrum = itself (:624-625)
            fg[7] = (first == 0x7fffffff) ? 0.0 : (double)first * (1.0 / (double)NF);
        }
        {   // population std of the 12 chroma values (:667)
            const double m = group_sum((i < 12) ? chroma : 0.0) / 12.0;
            const double d = (i < 12) ? chroma - m : 0.0;
            const double var = group_sum(d * d) / 12.0;
            if (i == 14) fg[33] = fast_sqrt(var);
        }
        wsync();
        PAA_TICK(9)
        // ---------------- store: lane = feature row, 4 consecutive frames
        if (lane < kBase) {
            double vq[QUAD];
#pragma unroll
            for (int s = 0; s < QUAD; ++s) vq[s] = fv[FV * s + lane];
            const bool last_quad = q0 + QUAD >= t_end;
            f800::store_row_chunked(oc + (long long)lane * Tc, q0, r0, t_end, last_quad, hold, vq, P.debug);
            if (DELTAS) {
                const double dq[QUAD] = {(q0 == 0) ? 0.0 : vq[0] - vlast, vq[1] - vq[0], vq[2] - vq[1], vq[3] - vq[2]};
                f800::store_row_chunked(oc + (long long)(kBase + lane) * Tc, q0, r0, t_end, last_quad, holdd, dq, P.debug);
#pragma unroll
                for (int j = 0; j < 7; ++j) holdd[j] = (j < 3) ? holdd[j + 4] : dq[j - 3];
            }
#pragma unroll
            for (int j = 0; j < 7; ++j) hold[j] = (j < 3) ? hold[j + 4] : vq[j - 3];
            vlast = vq[QUAD - 1];
        }
        wsync();
        PAA_TICK(10)
    }
    PAA_TEND()
#undef PAA_CT_PACE
    if (NW == 8 && lane == 0) pace[8 + wave] = 0x7fffffff;
}

// per-wave LDS bytes of a shape (must match WAVE_DOUBLES in the kernel)
template <typename SH>
constexpr size_t wave_bytes() {
    return (size_t)(5 * SH::NFP + 4 * 12 + (SH::SCRATCH_IN_RING ? 0 : 4 * (40 + 34))) * 8;
}


// ---- host: which windows have an instance, LDS layout + table blob, launch --------------------------------------
struct CtLaunch {
    int shape = -1;                 // 0: 25 x 16 (W 800), 1: 20 x 16 (W 640), 2: 10 x 16 (W 320), 3: 25 x 8 (W 400)
    int waves = 8;
    size_t lds = 0;
    const char *name = "";
    TabLayout layout;
};

inline int ct_shape_of(int window) {
    switch (window) {
        case 800: return 0;
        case 640: return 1;
        case 320: return 2;
        case 400: return 3;
        default: return -1;
    }
}

template <typename SH>
inline void ct_fill(const FftPlan &fft, const MelTable *mel, const ChromaTable *chroma, double fs, CtLaunch &cl,
                    std::vector<unsigned char> &blob) {
    TabLayout &L = cl.layout;
    auto up8 = [](int n) { return std::max(8, (n + 7) / 8 * 8); };      // lists are unrolled by 8 on the device
    int c0 = 0, c1 = 0, c2 = 0, cc = 0;
    if (mel && !mel->w.empty()) {
        for (int m = 0; m < 16; ++m) c0 = std::max(c0, (int)mel->cnt[m]);
        for (int m = 16; m < 32; ++m) c1 = std::max(c1, (int)mel->cnt[m]);
        for (int m = 32; m < 40; ++m) c2 = std::max(c2, ((int)mel->cnt[m] + 1) / 2);
    }
    if (chroma && !chroma->src.empty())
        for (int c = 0; c < 12; ++c) cc = std::max(cc, (int)(chroma->class_start[c + 1] - chroma->class_start[c]));
    L.melN0 = up8(c0); L.melN1 = up8(c1); L.melN2 = up8(c2); L.chN = up8(cc);
    L.mel_clamp = 0;
    if (mel && !mel->w.empty())
        for (int i = 0; i < 16; ++i) {
            const int f2 = 32 + (i & 7), half = (mel->cnt[f2] + 1) / 2;
            const int lo2 = mel->lo[f2] + ((i < 8) ? 0 : half);
            if (mel->lo[i] + L.melN0 > SH::NF || mel->lo[16 + i] + L.melN1 > SH::NF || lo2 + L.melN2 > SH::NF) L.mel_clamp = 1;
        }
    int off = 0;
    auto take = [&off](int bytes) { const int o = off; off += (bytes + 15) / 16 * 16; return o; };
    L.off_w0 = take(L.melN0 * 16 * 8); L.off_k0 = take(16 * 4);
    L.off_w1 = take(L.melN1 * 16 * 8); L.off_k1 = take(16 * 4);
    L.off_w2 = take(L.melN2 * 16 * 8); L.off_k2 = take(16 * 4);
    L.off_chw = take(L.chN * 12 * 8); L.off_chk = take(L.chN * 12 * 4);
    L.off_dct = take(13 * 41 * 8);
    L.off_tw2 = take(SH::RB * 16 * 16);
    L.off_twp = take(SH::RB * 16 * 16);
    L.off_sync = take(16 * 4);
    L.total = off;
    {
        const char *pm = experiment_env("PAA_F800_PACE");
        L.pace = (pm && pm[0] == '0') ? 0 : 1;
    }
    L.f0 = fs / (2.0 * (double)SH::NF);
    L.rf0 = 1.0 / L.f0;
    L.r_half_fs = 1.0 / (fs / 2.0);
    L.f0sq = L.f0 * L.f0;
    blob.assign((size_t)L.total, 0);
    auto Wd = [&](int o) { return reinterpret_cast<double *>(blob.data() + o); };
    auto Ki = [&](int o) { return reinterpret_cast<int32_t *>(blob.data() + o); };
    for (int i = 0; i < 16; ++i) {
        if (mel && !mel->w.empty()) {
            const int f0 = i, f1 = 16 + i, f2 = 32 + (i & 7);
            Ki(L.off_k0)[i] = mel->lo[f0];
            Ki(L.off_k1)[i] = mel->lo[f1];
            for (int n = 0; n < mel->cnt[f0]; ++n) Wd(L.off_w0)[n * 16 + i] = mel->w[mel->off[f0] + n];
            for (int n = 0; n < mel->cnt[f1]; ++n) Wd(L.off_w1)[n * 16 + i] = mel->w[mel->off[f1] + n];
            const int half = (mel->cnt[f2] + 1) / 2;
            const int b = (i < 8) ? 0 : half, e = (i < 8) ? half : mel->cnt[f2];
            Ki(L.off_k2)[i] = mel->lo[f2] + b;
            for (int n = b; n < e; ++n) Wd(L.off_w2)[(n - b) * 16 + i] = mel->w[mel->off[f2] + n];
        }
        if (i < 12 && chroma && !chroma->src.empty())
            for (int n = chroma->class_start[i]; n < chroma->class_start[i + 1]; ++n) {
                Wd(L.off_chw)[(n - chroma->class_start[i]) * 12 + i] = chroma->w[n];
                Ki(L.off_chk)[(n - chroma->class_start[i]) * 12 + i] = chroma->src[n];
            }
    }
    double dct[kNumMfcc * kNumMel];
    build_dct(dct);
    for (int q = 0; q < 13; ++q)
        for (int n = 0; n < 40; ++n) Wd(L.off_dct)[q * 41 + n] = dct[q * 40 + n];
    for (int p = 0; p < 16; ++p) {
        const int pp = std::min(p, SH::NP - 1);
        for (int r = 0; r < SH::RB; ++r) {
            const int m2 = (r * pp) % SH::NC, mp = pp + SH::RA * r;
            Wd(L.off_tw2)[2 * (r * 16 + p)] = fft.tw[2 * m2];
            Wd(L.off_tw2)[2 * (r * 16 + p) + 1] = fft.tw[2 * m2 + 1];
            Wd(L.off_twp)[2 * (r * 16 + p)] = fft.post[2 * mp];
            Wd(L.off_twp)[2 * (r * 16 + p) + 1] = fft.post[2 * mp + 1];
        }
    }
    cl.waves = 8;       // two waves per SIMD; tables too long for that (very low sampling rates) -> ct_select declines
    cl.lds = (size_t)L.total + (size_t)cl.waves * wave_bytes<SH>();
}

typedef Shape<25, 16> S800;
typedef Shape<20, 16> S640;
typedef Shape<10, 16> S320;
typedef Shape<25, 8> S400;

// returns 1 when a register-FFT instance exists for this window (fills cl and the table blob), 0 otherwise
inline int ct_select(int window, int mode, double fs, const FftPlan &fft, const MelTable *mel, const ChromaTable *chroma,
                     CtLaunch &cl, std::vector<unsigned char> &blob) {
    const int sh = ct_shape_of(window);
    if (sh < 0 || !fft.even) return 0;
    cl.shape = sh;
    static const char *names[3][4] = {{"st_ct_25x16", "st_ct_20x16", "st_ct_10x16", "st_ct_25x8"},
                                      {"spectrogram_ct_25x16", "spectrogram_ct_20x16", "spectrogram_ct_10x16", "spectrogram_ct_25x8"},
                                      {"chromagram_ct_25x16", "chromagram_ct_20x16", "chromagram_ct_10x16", "chromagram_ct_25x8"}};
    cl.name = names[mode][sh];
    switch (sh) {
        case 0: ct_fill<S800>(fft, mel, chroma, fs, cl, blob); break;
        case 1: ct_fill<S640>(fft, mel, chroma, fs, cl, blob); break;
        case 2: ct_fill<S320>(fft, mel, chroma, fs, cl, blob); break;
        default: ct_fill<S400>(fft, mel, chroma, fs, cl, blob); break;
    }
    if (cl.lds > 160 * 1024) return 0;
    return 1;
}

#if !defined(PAA_NO_HOST_LAUNCHERS) || defined(PAA_LAUNCH_CT)      // (kernels are instantiated only in family_ct*.hip)
template <typename SH, typename T, int MODE, int DELTAS, int NW>
inline int ct_launch_one(const CtLaunch &cl, const PlanDev &P, const unsigned char *blob, const void *d_packed,
                         const ClipDev *clips, const ClipNorm *norms, const Tile *tiles, long long n_tiles, double *d_out,
                         hipStream_t stream) {
    static LdsAttrCache attr;
    if (!attr.covers(cl.lds)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&st_ct_kernel<SH, T, MODE, DELTAS, NW>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)cl.lds) != hipSuccess) return -1;
        attr.set(cl.lds);
    }
    const unsigned grid = (unsigned)((n_tiles + NW - 1) / NW);
    hipLaunchKernelGGL((st_ct_kernel<SH, T, MODE, DELTAS, NW>), dim3(grid), dim3(64 * NW), cl.lds, stream, P, cl.layout, blob,
                       (const T *)d_packed, clips, norms, tiles, (int)n_tiles, d_out);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

template <typename SH, typename T>
inline int ct_launch_mode(const CtLaunch &cl, const PlanDev &P, const unsigned char *blob, const void *d_packed,
                          const ClipDev *clips, const ClipNorm *norms, const Tile *tiles, long long n_tiles, double *d_out,
                          hipStream_t stream) {
#define PAA_CT_GO(MODE, DELTAS)                                                                                          \
    return ct_launch_one<SH, T, MODE, DELTAS, 8>(cl, P, blob, d_packed, clips, norms, tiles, n_tiles, d_out, stream);
    if (P.mode == 1) { PAA_CT_GO(1, 0) }
    if (P.mode == 2) { PAA_CT_GO(2, 0) }
    if (P.deltas) { PAA_CT_GO(0, 1) }
    PAA_CT_GO(0, 0)
#undef PAA_CT_GO
}

template <typename T>
inline int ct_launch_shape(const CtLaunch &cl, const PlanDev &P, const unsigned char *blob, const void *d_packed,
                           const ClipDev *clips, const ClipNorm *norms, const Tile *tiles, long long n_tiles, double *d_out,
                           hipStream_t stream) {
    switch (cl.shape) {
        case 0: return ct_launch_mode<S800, T>(cl, P, blob, d_packed, clips, norms, tiles, n_tiles, d_out, stream);
        case 1: return ct_launch_mode<S640, T>(cl, P, blob, d_packed, clips, norms, tiles, n_tiles, d_out, stream);
        case 2: return ct_launch_mode<S320, T>(cl, P, blob, d_packed, clips, norms, tiles, n_tiles, d_out, stream);
        case 3: return ct_launch_mode<S400, T>(cl, P, blob, d_packed, clips, norms, tiles, n_tiles, d_out, stream);
        default: return -1;
    }
}

// sample_kind 0: int16, 1: float64, 2: interleaved stereo int16 (summed in the loads)
inline int ct_launch(const CtLaunch &cl, int sample_kind, const PlanDev &P, const unsigned char *blob, const void *d_packed,
                     const ClipDev *clips, const ClipNorm *norms, const Tile *tiles, long long n_tiles, double *d_out,
                     hipStream_t stream) {
    if (sample_kind == 0) return ct_launch_shape<int16_t>(cl, P, blob, d_packed, clips, norms, tiles, n_tiles, d_out, stream);
    if (sample_kind == 2) return ct_launch_shape<stereo16>(cl, P, blob, d_packed, clips, norms, tiles, n_tiles, d_out, stream);
    return ct_launch_shape<double>(cl, P, blob, d_packed, clips, norms, tiles, n_tiles, d_out, stream);
}

#endif  // PAA_NO_HOST_LAUNCHERS

}  // namespace ct
}  // namespace paa
