// Bluestein (chirp-z) feature kernel: every window of up to 5 461 samples whose FFT length has a prime factor above 13 and
// that no register-FFT shape covers -- 0.030 x 22050 = 661 (prime), 1103, 46 ms at 16 kHz = 736 = 2^5 x 23 ... (the reference
// takes any int(window), ShortTermFeatures.py:563-564, :617).  Until round 6 these ran O(N p) Stockham passes in
// kernels_generic.hpp (1103 / 441: 6.6e5 frames/s, 800 x slower than the headline; this kernel: 5.2e7).
//
// The DFT of a real frame y[0 .. W) at the bins the reference keeps, k < Nf = W / 2 (:617-621), as a convolution
// (n k = (n^2 + k^2 - (k - n)^2) / 2):
//
//     X[k] = conj(c[k]) * sum_n (y[n] conj(c[n])) c[k - n],      c[m] = exp(i pi m^2 / W)
//
// so |X[k]| = |(a (*) b)[k]| with a[n] = y[n] conj(c[n]) and b[m] = c[m], m = -(W - 1) .. Nf - 1 -- a cyclic convolution of any
// length M >= W + Nf - 1: M is the next power of two, and the convolution is two power-of-two FFTs around a pointwise
// product with the precomputed FFT(b) / M.  |c[k]| = 1: no multiplication after the convolution.
//
// One wave = one run of consecutive frames of one clip, one frame at a time, the M complex points in the wave's LDS buffer
// (16 M bytes, XOR-swizzled: element e sits at e ^ ((e >> 4) & 15), conflict-free ds_read_b128 / ds_write_b128 for the
// strides of all passes at M = 256 / 2048 / 4096 and within 4/3 at 512 / 1024: scripts/dev/blu_model.py).  Three radix
// passes (16 / 8 / 4 codelets in registers; four at M = 8192), decimation in frequency on the way in, decimation in time on
// the way back, so that NO permutation is ever applied: FFT(b) is stored in the forward transform's digit-reversed order.
// Even windows run PACKED (W / 2 complex points, M >= W - 1) when that halves the convolution: see Shape.
//
//   load     : W samples -> y (normalised) as doubles at the front of the buffer; a frame whose samples are all equal (digital
//              silence) takes a shortcut: spectrum [W |y0| / Nf, 0, 0, ...] exactly, as the reference's pocketfft gives for a
//              constant frame (the other kernels get this from exact-zero codelets; a chirp convolution cannot)
//   time     : zero crossings, energy, energy entropy from y (kernels_mix.hpp's contiguous chunks; the ten block energies from
//              the running energy at the block boundaries: one wave scan instead of eleven reductions)
//   pass 0   : DIF radix R0 over the whole sequence, straight from y: element n = y[n] conj(c[n]) for n < W, 0 beyond
//   pass 1   : DIF radix R1 inside the R0 blocks
//   pass 2   : the innermost radix-R2 butterflies are transformed forward, multiplied by FFT(b) / M, conjugated and transformed
//              again in registers -- forward pass 2, product and the first pass back are ONE LDS round trip
//              (the way back is a forward transform of the conjugate: |conj z| = |z|, one set of twiddles)
//   pass 1'  : DIT radix R1 (input twiddles)
//   pass 0'  : DIT radix R0; only the outputs k < Nf are formed (the others are dead code in the codelet), |.| / Nf goes to
//              the frame's spectrum -- held in registers until every lane has read its operands (the spectrum overlaps the buffer)
//   features : kernels_mix.hpp's spectral stage (run-time Nf) with kernels_tri.hpp's lane-balanced mel sums / chroma gather and its
//              52-lane DCT; lane = feature row, a row's values wait in eight registers until a 64-byte aligned chunk is complete
//              (kernels_tri.hpp: row_put; stored at the top of the NEXT iteration: vmcnt counts loads and stores in one sequence) -- no
//              staging tile in LDS: 38 KB per wave at M = 2048, four waves per CU
// Twiddles are never loaded inside the frame loop (register-resident seeds W^k .. W^4k, products of at most three table values);
// the chirp values and FFT(b) / M are requested ahead of their use (see Seeds, chirp_prefetch, bp_load).
//
// Replaces the while loop at ShortTermFeatures.py:608-682 (+ helpers :22-140, :236-321) and the loops of spectrogram
// (:415-422) / chromagram (:349-359) for those windows.
#pragma once
#include <vector>

#include "kernels_mix.hpp"
#include "kernels_tri.hpp"        // row_put: a feature row's pending 64-byte chunk in registers

namespace paa {
namespace blu {
#if defined(PAA_F800_TIMING) || defined(PAA_F800_TRACE)
using f800::g_phase_cycles;
using f800::g_wave_trace;
#endif

template <int LOG2M> struct Sched;
// NW = waves per workgroup: it sets the register budget (512 / ceil(NW / 4) per lane).  More waves than these spilled to scratch --
// 288 bytes per lane at M = 1024 with eight waves, and 12 x the algorithmic bytes in HBM writes (profiles/r06a_blu_661_eight_waves_spill_summary.json)
// R1B > 1: a second middle pass (M = 8192 = 16 x 8 x 8 x 8: four passes; 128 KB of LDS, one wave per CU -- windows up to 5461 samples)
template <> struct Sched<8> { static constexpr int R0 = 4, R1 = 8, R1B = 1, R2 = 8, NW = 8; };
template <> struct Sched<9> { static constexpr int R0 = 8, R1 = 8, R1B = 1, R2 = 8, NW = 8; };
template <> struct Sched<10> { static constexpr int R0 = 16, R1 = 8, R1B = 1, R2 = 8, NW = 4; };
template <> struct Sched<11> { static constexpr int R0 = 16, R1 = 16, R1B = 1, R2 = 8, NW = 4; };
template <> struct Sched<12> { static constexpr int R0 = 16, R1 = 16, R1B = 1, R2 = 16, NW = 2; };
template <> struct Sched<13> { static constexpr int R0 = 16, R1 = 8, R1B = 8, R2 = 8, NW = 1; };

// PACKED: even windows as W / 2 complex points z[m] = y[2m] + i y[2m+1] (the real-input trick of every other kernel here): the convolution only has to
// hold 2 (W / 2) - 1 points, so M >= W - 1 instead of W + W / 2 - 1 -- half the length for 58 % of the even windows (736: 1024 instead of 2048).  The price:
// all W / 2 outputs Z[k] are needed as COMPLEX numbers (times conj c[k]) and Z[k], Z[W/2 - k] meet in one more pass over the buffer (pair_magnitudes).
template <int LOG2M_, bool PACKED_ = false>
struct Shape {
    static constexpr bool PACKED = PACKED_;
    static constexpr int LOG2M = LOG2M_, M = 1 << LOG2M_;
    static constexpr int R0 = Sched<LOG2M_>::R0, R1 = Sched<LOG2M_>::R1, R1B = Sched<LOG2M_>::R1B, R2 = Sched<LOG2M_>::R2;
    static constexpr int NW = Sched<LOG2M_>::NW;               // most waves per workgroup (sets the register budget)
    static constexpr int S0 = M / R0;                          // pass 0: span M, element stride S0, S0 butterflies
    static constexpr int SP1 = S0, S1 = SP1 / R1;              // pass 1: span S0, stride S1
    static constexpr int SP1B = S1, S1B = S1 / R1B;            // pass 1b (R1B > 1 only): span S1, stride S1B
    static constexpr int TW0 = 0, TW1 = (R0 - 1) * S0, TW1B = TW1 + (R1 - 1) * S1;      // twiddle tables [q - 1][k] of pass 0 / 1 / 1b
    static constexpr int NTW = TW1B + (R1B > 1 ? (R1B - 1) * S1B : 0);
    // outputs of the last pass that can be bins: k + q S0 < Nf <= (M + 1) / 3 (packed: all Z[k], k < W / 2 <= M / 2)
    static constexpr int QMAX = PACKED ? R0 / 2 : ((R0 == 16) ? 6 : (R0 == 8 ? 3 : 2));
    // WMAX = the longest window in samples, LMAX = the longest sequence in elements: W <= (2 M + 3) / 3 (M >= W + W / 2 - 1), packed W <= M (M >= W - 1, W
    // even) as W / 2 elements.  The rows of pass 0 from RZ on are zeros for every window
    static constexpr int WMAX = PACKED ? M : (2 * M + 3) / 3, LMAX = PACKED ? M / 2 : WMAX, RZ = (LMAX + S0 - 1) / S0;
    static constexpr int NB0 = S0 / 64;                        // pass-0 butterflies per lane
    // the chirp values of a frame are requested at the top of the frame when they fit the registers beside the frame's samples (two
    // butterflies per lane: 44 + 22 doubles); M = 4096 (four: 88 + 44) fetches them butterfly by butterfly inside pass 0
    static constexpr bool CHIRP_AHEAD = NB0 <= 2;
    static constexpr bool SEEDS_RESIDENT = NB0 <= 2;           // pass-0 twiddle seeds live in registers for the whole run (else: fetched per use)
    static constexpr int NSD0 = SEEDS_RESIDENT ? NB0 : 1;
    static constexpr int NCW = CHIRP_AHEAD ? NB0 : 1;
    static constexpr int NB2 = (M / R2 + 63) / 64, U2 = (NB2 >= 2 && R2 <= 8) ? 2 : 1;      // pass 2: butterflies per lane, in flight
    // pass 0 takes its samples from the staged frame in LDS (two butterflies per lane: the lane reads all of them before any lane writes
    // the buffer) or, with more butterflies per lane than registers for that, again from global memory (L2 hits), butterfly by butterfly
    static constexpr bool Y_FROM_LDS = NB0 <= 2;
    static_assert(R0 * R1 * R1B * R2 == M && S1B == R2, "three or four passes");
    static_assert(S0 % 64 == 0, "pass 0: every lane has the same number of butterflies");
};

struct BluLayout {
    int off_mello, off_melcnt, off_meloff, off_melw, off_dct, off_chstart, off_chsrc, off_chw;
    int table_bytes;     // feature tables at the front of the blob (mel, DCT, chroma), multiple of 256
    int lds_table_bytes; // ... copied to LDS: table_bytes, or 0 for M = 8192 (the 128 KB buffer leaves no room: the kernel reads them from global memory)
    int wave_bytes;      // per-wave region, multiple of 256
    int waves;
    int unit_bytes;      // U: one spectrum (Nf doubles) rounded to 256 bytes
    int buf_bytes;       // B: 16 M
    int log2m;
    int off_g_chirp;     // global part: double2 [W]: conj(c[n])
    int off_g_bp;        // double2 [M]: FFT(b) / M at the positions the DIF passes leave the bins
    int off_g_tw;        // double2 [NTW]: pass tables
    int off_g_post;      // packed: double2 [W / 4 + 1]: exp(-2 pi i k / W)
    int packed;          // 1: even window as W / 2 complex points (M >= W - 1)
    int off_g_meljob, off_g_chjob;      // int4 [64] each: the lane jobs of the mel sums / the chroma gather (kernels_tri.hpp: LaneJob)
    int total_bytes;
};

__device__ __forceinline__ constexpr int sw(int e) { return e ^ ((e >> 4) & 15); }
// the swizzle is linear over GF(2): for a compile-time offset c whose bits are clear in `base`, sw(base + c) = sw(base) ^ sw(c) --
// one XOR per element of a butterfly (base is the butterfly's first element: the offsets r S of its other elements fill bit
// fields that are zero in it) instead of shift / and / xor / add
#define PAA_BLU_AT(swbase, c) ((swbase) ^ sw(c))

// The twiddles of a butterfly, W^(q k) for q = 1 .. R - 1, depend on the lane only -- not on the frame: a lane keeps the SEEDS
// W^k, W^2k, W^3k, W^4k of its butterflies in registers for the whole run (loaded once from the pass tables) and forms the others
// as products of at most three table values (relative error <= 3 ulp): no twiddle loads inside the frame loop.  (The first
// version fetched 15 twiddles per radix-16 butterfly from the L2 in every pass: 92 loads of 1 KB per wave and frame, and
// one exposed L2 round trip per butterfly.)
struct Seeds {
    double2 w[4];           // W^(j k), j = 1 .. 4
};
__device__ __forceinline__ Seeds load_seeds(const double2 *__restrict__ tab, int stride, int k, int radix) {
    Seeds sd;
#pragma unroll
    for (int j = 0; j < 4; ++j) sd.w[j] = (j + 1 < radix) ? tab[j * stride + k] : make_double2(1.0, 0.0);
    return sd;
}
template <int R>
__device__ __forceinline__ void expand_twiddles(const Seeds &sd, double2 *w) {      // w[q] = W^(q k), q = 1 .. R - 1
#pragma unroll
    for (int q = 1; q < R && q <= 4; ++q) w[q] = sd.w[q - 1];
    if constexpr (R >= 8) {
        w[5] = cmul(sd.w[3], sd.w[0]);
        w[6] = cmul(sd.w[3], sd.w[1]);
        w[7] = cmul(sd.w[3], sd.w[2]);
    }
    if constexpr (R >= 16) {
        w[8] = cmul(sd.w[3], sd.w[3]);
        w[9] = cmul(w[8], sd.w[0]);
        w[10] = cmul(w[8], sd.w[1]);
        w[11] = cmul(w[8], sd.w[2]);
        w[12] = cmul(w[8], sd.w[3]);
        w[13] = cmul(w[12], sd.w[0]);
        w[14] = cmul(w[12], sd.w[1]);
        w[15] = cmul(w[12], sd.w[2]);
    }
}

// ---- pass 0 forward, fused with the chirp: y (doubles at the front of the buffer) -> buf.  The chirp values of a lane are
// requested at the top of the frame (chirp_prefetch: their L2 round trip runs under the frame load and the time-domain stage)
template <typename SH>
__device__ __forceinline__ void chirp_prefetch(const double2 *__restrict__ g_chirp, int W, int lane, double2 (&cw)[SH::NCW][SH::RZ]) {
    if constexpr (SH::CHIRP_AHEAD) {
#pragma unroll
        for (int u = 0; u < SH::NB0; ++u)
#pragma unroll
            for (int r = 0; r < SH::RZ; ++r) cw[u][r] = g_chirp[min(lane + 64 * u + r * SH::S0, W - 1)];      // (beyond W: y is 0)
    }
}
template <typename T, typename SH>
__device__ __forceinline__ void fwd_pass0(double2 *buf, const double2 *__restrict__ g_chirp, const double2 (&cw)[SH::NCW][SH::RZ],
                                          const Seeds *sd0, const double2 *__restrict__ g_tw, const T *__restrict__ x,
                                          const ClipNorm &nm, int L, int lane) {
    // L = elements of the sequence: W samples, or W / 2 sample pairs (packed)
    constexpr int R = SH::R0, S = SH::S0, NB = SH::NB0, RZ = SH::RZ;
    constexpr bool PK = SH::PACKED;
    const double *st = reinterpret_cast<const double *>(buf);
    const double2 *st2 = reinterpret_cast<const double2 *>(buf);
    double2 y[SH::Y_FROM_LDS ? NB : 1][RZ];          // (direct: .y unused)
    if constexpr (SH::Y_FROM_LDS) {
#pragma unroll
        for (int u = 0; u < NB; ++u)
#pragma unroll
            for (int r = 0; r < RZ; ++r) {
                const int n = lane + 64 * u + r * S;
                if constexpr (PK) {
                    const double2 z = st2[min(n, L - 1)];
                    y[u][r] = (n < L) ? z : make_double2(0.0, 0.0);
                } else {
                    y[u][r] = make_double2((n < L) ? st[min(n, L - 1)] : 0.0, 0.0);
                }
            }
        wsync();           // every lane has its samples: the buffer may be overwritten
    }
    const double sc = sample_scale<T>();
#pragma unroll
    for (int u = 0; u < NB; ++u) {
        const int k = lane + 64 * u;
        double2 v[R], w[R], cu[RZ], yu[RZ];
#pragma unroll
        for (int r = 0; r < RZ; ++r) {
            if constexpr (SH::CHIRP_AHEAD) cu[r] = cw[u < SH::NCW ? u : 0][r];
            else cu[r] = g_chirp[min(k + r * S, L - 1)];
            if constexpr (SH::Y_FROM_LDS) yu[r] = y[u < (SH::Y_FROM_LDS ? NB : 1) ? u : 0][r];
            else if constexpr (PK) yu[r] = ct::PairLoad<T>::get(x + 2 * min(k + r * S, L - 1));      // (the frame is in the L2: loaded for the time-domain stage)
            else yu[r] = make_double2(load_sample<T>(x + min(k + r * S, L - 1)), 0.0);
        }
        if constexpr (!SH::Y_FROM_LDS) {
#pragma unroll
            for (int r = 0; r < RZ; ++r) {
                const bool in = k + r * S < L;
                yu[r].x = in ? fma(yu[r].x, sc, -nm.mean) * nm.inv : 0.0;
                if constexpr (PK) yu[r].y = in ? fma(yu[r].y, sc, -nm.mean) * nm.inv : 0.0;
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (r < RZ) {
                const double2 a = yu[r < RZ ? r : 0], c = cu[r < RZ ? r : 0];
                if constexpr (PK) v[r] = cmul(a, c);
                else v[r] = make_double2(a.x * c.x, a.x * c.y);
            } else {
                v[r] = make_double2(0.0, 0.0);
            }
        }
        mix::Bfly<R>::run(v);
        // (M >= 4096: four and more butterflies per lane -- their seeds are fetched where they are used instead of living in registers)
        expand_twiddles<R>(SH::SEEDS_RESIDENT ? sd0[SH::SEEDS_RESIDENT ? u : 0] : load_seeds(g_tw + SH::TW0, S, k, R), w);
#pragma unroll
        for (int q = 1; q < R; ++q) v[mix::Bfly<R>::pos(q)] = cmul(v[mix::Bfly<R>::pos(q)], w[q]);
        const int swk = sw(k);          // (k < S: the fields of q S are clear)
#pragma unroll
        for (int q = 0; q < R; ++q) buf[PAA_BLU_AT(swk, q * S)] = v[mix::Bfly<R>::pos(q)];
        // (four butterflies per lane: the scheduler would fetch the operands of all of them first -- 1 KB of scratch per lane at M = 4096)
        if constexpr (NB > 2) __builtin_amdgcn_sched_barrier(0);
    }
    wsync();
}

// butterfly b of a pass with span SPAN and radix R: first element and offset inside the block
template <int R, int SPAN>
__device__ __forceinline__ void locate(int b, int &base, int &k) {
    constexpr int S = SPAN / R;
    k = b & (S - 1);
    base = (b / S) * SPAN + k;
}

// ---- pass 1: DIF (FWD: output twiddles) or DIT (!FWD: input twiddles) over the blocks of S0 elements
template <typename SH, bool FWD, int WHICH = 0>
__device__ __forceinline__ void pass1(double2 *buf, const Seeds &sd1, int lane) {
    constexpr int R = WHICH ? SH::R1B : SH::R1, SPAN = WHICH ? SH::SP1B : SH::SP1, S = WHICH ? SH::S1B : SH::S1, NBT = SH::M / R,
                  NB = (NBT + 63) / 64;
    constexpr int U = (NB >= 2 && R <= 8) ? 2 : 1;          // butterflies in flight per lane (radix 16: 32 registers of operands each)
    static_assert(S <= 64 && 64 % S == 0, "every butterfly of a lane has the same offset k = lane mod S: one set of twiddles");
    double2 w[R];
    expand_twiddles<R>(sd1, w);
#pragma unroll
    for (int u0 = 0; u0 < NB; u0 += U) {
        double2 v[U][R];
        int base[U], k[U];
        bool act[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int b = lane + 64 * (u0 + u);
            act[u] = (NBT % 64 == 0) || b < NBT;
            locate<R, SPAN>(act[u] ? b : NBT - 1, base[u], k[u]);
            base[u] = sw(base[u]);          // (base = blk SPAN + k, k < S: the fields of r S are clear)
#pragma unroll
            for (int r = 0; r < R; ++r) v[u][r] = buf[PAA_BLU_AT(base[u], r * S)];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!FWD) {
#pragma unroll
                for (int r = 1; r < R; ++r) v[u][r] = cmul(v[u][r], w[r]);
            }
            mix::Bfly<R>::run(v[u]);
            if (FWD) {
#pragma unroll
                for (int q = 1; q < R; ++q) v[u][mix::Bfly<R>::pos(q)] = cmul(v[u][mix::Bfly<R>::pos(q)], w[q]);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (act[u]) {
#pragma unroll
                for (int q = 0; q < R; ++q) buf[PAA_BLU_AT(base[u], q * S)] = v[u][mix::Bfly<R>::pos(q)];
            }
        if constexpr (NB > 2) __builtin_amdgcn_sched_barrier(0);
    }
    wsync();
}

// ---- pass 2 forward + product with FFT(b) / M + conjugate + pass 2 back: the R2 elements of a butterfly are contiguous.
// FFT(b) / M comes from the L2: the values of a batch of butterflies are requested one batch ahead (bp_load; the first batch
// before pass 1 starts), so that their round trip runs under a batch of butterflies instead of in front of it
template <typename SH>
__device__ __forceinline__ void bp_load(const double2 *__restrict__ g_bp, int lane, int u0, double2 (&bp)[SH::U2][SH::R2]) {
    constexpr int R = SH::R2, NBT = SH::M / R;
#pragma unroll
    for (int u = 0; u < SH::U2; ++u) {
        const int b = min(lane + 64 * (u0 + u), NBT - 1);
#pragma unroll
        for (int q = 0; q < R; ++q) bp[u][q] = g_bp[b * R + q];
    }
}
template <typename SH>
__device__ __forceinline__ void pass2_product(double2 *buf, const double2 *__restrict__ g_bp, double2 (&bp)[SH::U2][SH::R2], int lane) {
    constexpr int R = SH::R2, NBT = SH::M / R, NB = SH::NB2, U = SH::U2;
#pragma unroll
    for (int u0 = 0; u0 < NB; u0 += U) {
        double2 v[U][R], bp_next[U][R];
        int base[U];
        bool act[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int b = lane + 64 * (u0 + u);
            act[u] = (NBT % 64 == 0) || b < NBT;
            base[u] = sw((act[u] ? b : NBT - 1) * R);          // (multiples of R: the low bits are clear)
#pragma unroll
            for (int r = 0; r < R; ++r) v[u][r] = buf[PAA_BLU_AT(base[u], r)];
        }
        if (u0 + U < NB) bp_load<SH>(g_bp, lane, u0 + U, bp_next);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            mix::Bfly<R>::run(v[u]);
            double2 z[R];
#pragma unroll
            for (int q = 0; q < R; ++q) {
                const double2 p = cmul(v[u][mix::Bfly<R>::pos(q)], bp[u][q]);
                z[q] = make_double2(p.x, -p.y);
            }
            mix::Bfly<R>::run(z);
#pragma unroll
            for (int q = 0; q < R; ++q) v[u][q] = z[mix::Bfly<R>::pos(q)];
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (act[u]) {
#pragma unroll
                for (int q = 0; q < R; ++q) buf[PAA_BLU_AT(base[u], q)] = v[u][q];
            }
        if (u0 + U < NB) {
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int q = 0; q < R; ++q) bp[u][q] = bp_next[u][q];
        }
        if constexpr (NB > 4) __builtin_amdgcn_sched_barrier(0);
    }
    wsync();
}

// ---- pass 0 back (DIT, input twiddles) + |.| / Nf (ShortTermFeatures.py:617-621): bins k + q S0 < Nf only
template <typename SH>
__device__ __forceinline__ void back_pass0_magnitudes(const double2 *buf, double *cur, const Seeds *sd0, const double2 *__restrict__ g_tw,
                                                      int Nf, int lane) {
    constexpr int R = SH::R0, S = SH::S0, NB = S / 64, QM = SH::QMAX;
    const double invNf = 1.0 / (double)Nf;
    double res[NB][QM];
#pragma unroll
    for (int u = 0; u < NB; ++u) {
        const int k = lane + 64 * u;
        double2 v[R], w[R];
        const int swk = sw(k);
#pragma unroll
        for (int r = 0; r < R; ++r) v[r] = buf[PAA_BLU_AT(swk, r * S)];
        expand_twiddles<R>(SH::SEEDS_RESIDENT ? sd0[SH::SEEDS_RESIDENT ? u : 0] : load_seeds(g_tw + SH::TW0, S, k, R), w);
#pragma unroll
        for (int r = 1; r < R; ++r) v[r] = cmul(v[r], w[r]);
        mix::Bfly<R>::run(v);
#pragma unroll
        for (int q = 0; q < QM; ++q) {
            const double2 z = v[mix::Bfly<R>::pos(q)];
            res[u][q] = mag_sqrt(fma(z.x, z.x, z.y * z.y)) * invNf;
        }
        if constexpr (NB > 2) __builtin_amdgcn_sched_barrier(0);
    }
    wsync();           // every lane has read its operands: the spectrum may overwrite the buffer
#pragma unroll
    for (int u = 0; u < NB; ++u)
#pragma unroll
        for (int q = 0; q < QM; ++q) {
            const int idx = lane + 64 * u + q * S;
            if (idx < Nf) cur[idx] = res[u][q];
        }
    wsync();
}

// ---- packed windows: pass 0 back leaves Z[k] = conj(c[k]) conj(v[k]) IN PLACE (an in-place DIT butterfly writes where it read: no hazard), then the
// real-FFT recombination pairs bin k with W/2 - k: X[k] = E + w^k O, X[W/2 - k] = conj(E - w^k O), E = (Z[k] + conj Z[W/2 - k]) / 2,
// O = -i (Z[k] - conj Z[W/2 - k]) / 2, w = exp(-2 pi i / W) (g_post) -- |.| / Nf into the frame's spectrum, held in registers until every lane has read
template <typename SH>
__device__ __forceinline__ void back_pass0_packed(double2 *buf, double *cur, const Seeds *sd0, const double2 *__restrict__ g_tw,
                                                  const double2 *__restrict__ g_chirp, const double2 *__restrict__ g_post, int Nc, int lane) {
    constexpr int R = SH::R0, S = SH::S0, NB = S / 64, QM = SH::QMAX;
#pragma unroll
    for (int u = 0; u < NB; ++u) {
        const int k = lane + 64 * u;
        double2 v[R], w[R], cq[QM];
#pragma unroll
        for (int q = 0; q < QM; ++q) cq[q] = g_chirp[min(k + q * S, Nc - 1)];
        const int swk = sw(k);
#pragma unroll
        for (int r = 0; r < R; ++r) v[r] = buf[PAA_BLU_AT(swk, r * S)];
        expand_twiddles<R>(SH::SEEDS_RESIDENT ? sd0[SH::SEEDS_RESIDENT ? u : 0] : load_seeds(g_tw + SH::TW0, S, k, R), w);
#pragma unroll
        for (int r = 1; r < R; ++r) v[r] = cmul(v[r], w[r]);
        mix::Bfly<R>::run(v);
#pragma unroll
        for (int q = 0; q < QM; ++q) {
            const double2 z = v[mix::Bfly<R>::pos(q)];
            buf[PAA_BLU_AT(swk, q * S)] = cmul(cq[q], make_double2(z.x, -z.y));        // (indices >= Nc: never read)
        }
        if constexpr (NB > 2) __builtin_amdgcn_sched_barrier(0);
    }
    wsync();
    constexpr int NPL = SH::M / 256 + 1;             // pairs per lane: W/2 / 2 + 1 <= M / 4 + 1
    const int npairs = Nc / 2 + 1;
    const double hs = 0.5 / (double)Nc;              // E, O carry 1/2; X / len(X), len = Nf = W / 2 (:621)
    double r0[NPL], r1[NPL];
#pragma unroll
    for (int i0 = 0; i0 < NPL; i0 += 4) {
        double2 zk[4], zm[4], pw[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = min(lane + 64 * (i0 + i), npairs - 1);
            if (i0 + i < NPL) { zk[i] = buf[sw(k)]; zm[i] = buf[sw(k == 0 ? 0 : Nc - k)]; pw[i] = g_post[k]; }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i0 + i < NPL) {
                const double2 e = make_double2(zk[i].x + zm[i].x, zk[i].y - zm[i].y);
                const double2 o = make_double2(zk[i].y + zm[i].y, zm[i].x - zk[i].x);
                const double2 wo = cmul(pw[i], o);
                const double ar = e.x + wo.x, ai = e.y + wo.y, br = e.x - wo.x, bi = e.y - wo.y;
                r0[i0 + i] = mag_sqrt(fma(ar, ar, ai * ai)) * hs;
                r1[i0 + i] = mag_sqrt(fma(br, br, bi * bi)) * hs;
            }
    }
    wsync();           // every lane has read its operands: the spectrum may overwrite the buffer
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const int k = lane + 64 * i;
        if (k < npairs) {
            cur[k] = r0[i];
            if (k > 0 && Nc - k != k) cur[Nc - k] = r1[i];
        }
    }
    wsync();
}

// The samples of a frame: all NL = ceil(WMAX / 64) loads of the lane in flight at once (frame_request, index clamped at W - 1),
// then normalised + written to st[0 .. W) (frame_commit).
template <typename T, typename SH>
__device__ __forceinline__ void frame_request(const T *__restrict__ x, int W, int lane, double (&q)[(SH::WMAX + 63) / 64 + 1]) {
    constexpr int NL = (SH::WMAX + 63) / 64;
#pragma unroll
    for (int i = 0; i < NL; ++i) q[i] = load_sample<T>(x + min(lane + kWave * i, W - 1));
    q[NL] = load_sample<T>(x);          // (sample 0: every lane compares against it)
}
// returns (wave-uniform) whether all samples of the frame are equal; y0 = sample 0
template <typename T, typename SH>
__device__ __forceinline__ bool frame_commit(const PlanDev &P, const double (&q)[(SH::WMAX + 63) / 64 + 1], const ClipNorm &nm, double *st,
                                             int lane, double &y0) {
    constexpr int NL = (SH::WMAX + 63) / 64;
    const double sc = sample_scale<T>();
    const int W = P.W;
    const double first = fma(q[NL], sc, -nm.mean) * nm.inv;
    bool differs = false;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const int n = lane + kWave * i;
        const double yv = fma(q[i], sc, -nm.mean) * nm.inv;
        if (n < W) {
            differs |= (yv != first);
            st[n] = yv;
        }
    }
    y0 = first;
    wsync();
    return __ballot(differs) == 0ull;
}

// the same in chunks of sixteen loads per lane (the long windows of M >= 4096: 43 / 86 samples per lane would be 86 / 172 registers)
template <typename T, typename SH>
__device__ __forceinline__ bool frame_load_chunked(const PlanDev &P, const T *__restrict__ x, const ClipNorm &nm, double *st, int lane,
                                                   double &y0) {
    constexpr int NL = (SH::WMAX + 63) / 64, CH = 16;
    const double sc = sample_scale<T>();
    const int W = P.W;
    const double first = fma(load_sample<T>(x), sc, -nm.mean) * nm.inv;
    bool differs = false;
#pragma unroll
    for (int i0 = 0; i0 < NL; i0 += CH) {
        double q[CH];
#pragma unroll
        for (int i = 0; i < CH; ++i) q[i] = (i0 + i < NL) ? load_sample<T>(x + min(lane + kWave * (i0 + i), W - 1)) : 0.0;
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int n = lane + kWave * (i0 + i);
            const double yv = fma(q[i], sc, -nm.mean) * nm.inv;
            if (i0 + i < NL && n < W) {
                differs |= (yv != first);
                st[n] = yv;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    y0 = first;
    wsync();
    return __ballot(differs) == 0ull;
}

// zcr count, energy and energy entropy of the normalised frame (contiguous doubles), ShortTermFeatures.py:22-51.  Lane l owns the
// contiguous chunk of kernels_mix.hpp (make_chunk).  The ten block energies come from the RUNNING energy at the block boundaries
// (one wave scan of the lanes' chunk energies; the lane whose chunk contains boundary j writes the running value there to bnd[j]):
// kernels_mix.hpp reduces every block over the wave -- eleven reductions of ~25 instructions each per stage, a fifth of this
// kernel's frame at one wave per SIMD.
__device__ __forceinline__ TimeFeat time_features_real(const double *y, const mix::Chunk &ch, int W, int LT, double *bnd, int lane) {
    double own = 0.0;
    int zc = 0;
    auto sgn = [](double x) {
        const int hi = __double2hiint(x), lo = __double2loint(x);
        return (((hi & 0x7fffffff) | lo) != 0) ? ((hi >> 31) | 1) : 0;
    };
    int sprev = sgn(y[max(ch.kb - 1, 0)]);                 // (lane 0: sample 0 against itself counts nothing)
    // first boundary at or after the lane's first sample; the running energy BEFORE sample jb LT is bnd[jb]
    const int jb = (ch.kb + LT - 1) / LT;
    const int kbound = jb * LT;
    double part = 0.0;                                     // chunk energy before the boundary
    bool hit = false;
    auto one = [&](int n, double x) {
        if (n == kbound) { part = own; hit = true; }
        own = fma(x, x, own);
        const int sx = sgn(x);
        zc += abs(sx - sprev);
        sprev = sx;
    };
    int i = 0;
    for (; i + 4 <= ch.base; i += 4) {
        double v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = y[ch.kb + i + u];
#pragma unroll
        for (int u = 0; u < 4; ++u) one(ch.kb + i + u, v[u]);
    }
    for (int n = ch.kb + i; n < ch.ke; ++n) one(n, y[n]);
    const double incl = wscan_incl(own);
    const double before = incl - own;
    TimeFeat tf;
    tf.e_tot = readlane63(incl);
    if (hit && jb <= 10) bnd[jb] = before + part;
    if (10 * LT == W && lane == 0) bnd[10] = tf.e_tot;     // (the blocks tile the frame: no sample sits on boundary 10)
    tf.zc = wsum_i(zc);
    wsync();
    const int ib = min(lane, 9);
    const double s = fast_div(bnd[ib + 1] - bnd[ib], tf.e_tot + kEps);
    tf.ent_e = wsum((lane < 10) ? -(s * fast_log2(s + kEps)) : 0.0);
    wsync();
    return tf;
}

// the 34 base features of one frame into fv[0..33] (ShortTermFeatures.py:626-667): kernels_mix.hpp's frame_features_chunked with
// the block energies from the running energy at the block boundaries (see above; the roll-off needs that scan anyway), the mel sums
// and the chroma gather cut into lane jobs on all 64 lanes and the 13 x 40 DCT on 52 lanes (kernels_tri.hpp)
__device__ __forceinline__ void frame_features_blu(const PlanDev &P, const Tabs &tb, const TimeFeat &tf, const double *cur,
                                                   const double *prv, double *fv, double *msp, double *bnd, const mix::Chunk &ch,
                                                   const int4 mjob, const int4 cjob, int lane) {
    const int W = P.W, Nf = P.Nf, LB = P.blk_f;
    const double f0 = P.fs / (2.0 * (double)Nf);
    // ---------- sweep A over the lane's bins: sums, max, chunk energy (:57-107)
    double sX = 0.0, sXp = 0.0, sIX = 0.0, mx = 0.0, own = 0.0;
    auto sweep_a = [&](int k, double X, double Xp) {
        sX += X;
        sXp += Xp;
        sIX = fma((double)(k + 1), X, sIX);
        mx = fmax(mx, X);
        own = fma(X, X, own);
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
    const double incl = wscan_incl(own);
    const double before = incl - own;
    const double sP = readlane63(incl);                   // sum X^2 over all bins
    sX = wsum(sX);
    sXp = wsum(sXp);
    sIX = wsum(sIX) * f0;
    mx = wmax_nonneg(mx);
    const double sXe = sX + (double)Nf * kEps;              // np.sum(X + eps) (:118-119)
    sXp += (double)Nf * kEps;
    // ---------- centroid, then sweep B: spread + flux + roll-off + the running energy at the lane's block boundary (:57-140)
    const double r = (mx == 0.0) ? 1.0 / kEps : fast_div(1.0, mx);
    const double den = sX * r + kEps;
    const double cen = fast_div(sIX * r, den);
    const double rX = fast_div(1.0, sXe), rXp = fast_div(1.0, sXp);
    const double thr = 0.90 * sP;
    const int jb = (ch.kb + LB - 1) / LB;
    const int kbound = jb * LB;
    double sSp = 0.0, sFl = 0.0, run = before, cumb = 0.0;
    bool hit = false;
    int first = 0x7fffffff;
    auto sweep_b = [&](int k, double X, double Xp) {
        const double dv = (double)(k + 1) * f0 - cen;
        sSp = fma(dv * dv, X * r, sSp);
        const double df = X * rX - Xp * rXp;
        sFl = fma(df, df, sFl);
        if (k == kbound) { cumb = run; hit = true; }
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
    if (hit && jb <= 10) bnd[jb] = cumb;
    if (10 * LB == Nf && lane == 0) bnd[10] = sP;
    sSp = wsum(sSp);
    sFl = wsum(sFl);
    first = mix::wmin_nonneg_i(first);
    const double spread = fast_sqrt(fast_div(sSp, den));
    wsync();
    // spectral entropy (:85-107): lane j < 10 owns block j
    double ent_f;
    {
        const int ib = min(lane, 9);
        const double sf = fast_div(bnd[ib + 1] - bnd[ib], sP + kEps);
        ent_f = wsum((lane < 10) ? -(sf * fast_log2(sf + kEps)) : 0.0);
    }
    // ---------- MFCC (:236-254): mel band energies on all 64 lanes, log10 in lane m < 40, the 13 x 40 DCT on 52 lanes
    {
        const double e = tri::mel_sums_balanced(tb, cur, mjob, lane);
        if (lane < 40) msp[lane] = fast_log10(e + kEps);
    }
    // ---------- chroma (:277-321)
    double chroma = tri::chroma_sums_balanced(tb, cur, cjob, lane);
    chroma = (sP == 0.0) ? chroma / kEps : fast_div(chroma, sP);
    wsync();
    {
        // lane 4 q + part: ten terms of DCT row q; the four parts meet through two quad permutes
        const int q = min(lane >> 2, 12), part = lane & 3;
        const double *dm = tb.dct + q * tb.dct_stride + 10 * part;
        const double *mv = msp + 10 * part;
        double c0 = 0.0, c1 = 0.0;
#pragma unroll
        for (int n = 0; n < 10; n += 2) {
            c0 = fma(dm[n], mv[n], c0);
            c1 = fma(dm[n + 1], mv[n + 1], c1);
        }
        double cc = c0 + c1;
        cc += dpp_mov<PAA_DPP_X1>(cc);
        cc += dpp_mov<PAA_DPP_X2>(cc);
        if (lane < 52 && part == 0) fv[8 + q] = cc;
    }
    {   // population std of the 12 chroma values (:667): lanes 0..11 of the first row
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

template <typename T, int LOG2M, bool PK = false>
__global__ __launch_bounds__(64 * Sched<LOG2M>::NW) void st_blu_kernel(PlanDev P, BluLayout L, const unsigned char *__restrict__ blob,
                                                                       const T *__restrict__ sig, const ClipDev *__restrict__ clips,
                                                                       const ClipNorm *__restrict__ norms,
                                                                       const Tile *__restrict__ tiles, int n_tiles,
                                                                       double *__restrict__ out) {
    typedef Shape<LOG2M, PK> SH;
    extern __shared__ __attribute__((aligned(256))) unsigned char smem[];
    constexpr bool TABLES_IN_LDS = LOG2M < 13;
    if constexpr (TABLES_IN_LDS) {
        const int4 *src4 = reinterpret_cast<const int4 *>(blob);
        int4 *dst4 = reinterpret_cast<int4 *>(smem);
        for (int n = threadIdx.x; n < L.table_bytes / 16; n += blockDim.x) dst4[n] = src4[n];
        __syncthreads();       // the only workgroup-wide barrier
    }
    Tabs tb;
    tb.tw = nullptr; tb.post = nullptr;
    if constexpr (TABLES_IN_LDS) {
        tb.mel_lo = reinterpret_cast<const int *>(smem + L.off_mello);
        tb.mel_cnt = reinterpret_cast<const int *>(smem + L.off_melcnt);
        tb.mel_off = reinterpret_cast<const int *>(smem + L.off_meloff);
        tb.mel_w = reinterpret_cast<const double *>(smem + L.off_melw);
        tb.dct = reinterpret_cast<const double *>(smem + L.off_dct);
        tb.ch_start = reinterpret_cast<const int *>(smem + L.off_chstart);
        tb.ch_src = reinterpret_cast<const int *>(smem + L.off_chsrc);
        tb.ch_w = reinterpret_cast<const double *>(smem + L.off_chw);
    } else {
        tb.mel_lo = reinterpret_cast<const int *>(blob + L.off_mello);
        tb.mel_cnt = reinterpret_cast<const int *>(blob + L.off_melcnt);
        tb.mel_off = reinterpret_cast<const int *>(blob + L.off_meloff);
        tb.mel_w = reinterpret_cast<const double *>(blob + L.off_melw);
        tb.dct = reinterpret_cast<const double *>(blob + L.off_dct);
        tb.ch_start = reinterpret_cast<const int *>(blob + L.off_chstart);
        tb.ch_src = reinterpret_cast<const int *>(blob + L.off_chsrc);
        tb.ch_w = reinterpret_cast<const double *>(blob + L.off_chw);
    }
    tb.dct_stride = 41;
    const double2 *g_chirp = reinterpret_cast<const double2 *>(blob + L.off_g_chirp);
    const double2 *g_bp = reinterpret_cast<const double2 *>(blob + L.off_g_bp);
    const double2 *g_tw = reinterpret_cast<const double2 *>(blob + L.off_g_tw);
    const double2 *g_post = reinterpret_cast<const double2 *>(blob + L.off_g_post);          // (packed windows only)
    const int4 *g_meljob = reinterpret_cast<const int4 *>(blob + L.off_g_meljob), *g_chjob = reinterpret_cast<const int4 *>(blob + L.off_g_chjob);

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane_id = threadIdx.x & 63;
    const int tile_id = blockIdx.x * L.waves + wave;
    if (tile_id >= n_tiles) return;
    const int Nf = P.Nf, W = P.W;
    const int Lseq = PK ? W / 2 : W;          // elements of the convolved sequence: samples, or sample pairs (packed)
    int lane = lane_id;
    unsigned char *wb = smem + L.lds_table_bytes + wave * L.wave_bytes;
    double *fv = reinterpret_cast<double *>(wb + L.buf_bytes + L.unit_bytes);
    double *msp = fv + 48;
    double *bnd = msp + 40;

    const Tile tl = tiles[tile_id];
    const ClipDev c = clips[tl.clip];
    const ClipNorm nm = wave_clip_norm<T>(P, c, norms, tl.clip, lane);
    const T *x0 = sig + c.sample_off + P.frame_origin;
    const long long Tc = c.T;
    double *oc = out + c.out_off;

    const int hneed = (P.mode == 0) ? (P.deltas ? 2 : 1) : 0;
    const int h = min(hneed, tl.t0);
    double vprev = 0.0;
    int odd = 0;
    const int tend = tl.t0 + tl.cnt;
    // the twiddle seeds of this lane's butterflies (frame-invariant: see Seeds)
    Seeds sd0[SH::NSD0], sd1, sd1b;
#pragma unroll
    for (int u = 0; u < SH::NSD0; ++u) sd0[u] = load_seeds(g_tw + SH::TW0, SH::S0, lane + 64 * u, SH::R0);
    sd1 = load_seeds(g_tw + SH::TW1, SH::S1, lane & (SH::S1 - 1), SH::R1);
    sd1b = (SH::R1B > 1) ? load_seeds(g_tw + SH::TW1B, SH::S1B, lane & (SH::S1B - 1), SH::R1B) : sd1;
    tri::RowChunk rc, rcd;
#pragma unroll
    for (int i = 0; i < 8; ++i) { rc.h[i] = 0.0; rcd.h[i] = 0.0; }
    const mix::Chunk ch_t = mix::make_chunk(W, P.blk_t, lane_id), ch_f = mix::make_chunk(Nf, P.blk_f, lane_id);
    // a frame's row values are stored at the top of the NEXT iteration (row_put after frame_commit): vmcnt counts loads and stores
    // in one sequence on this chip, so stores issued at the end of an iteration made the next frame's first load wait for their
    // completion in HBM -- 5 k cycles per frame
    double v_pend = 0.0, d_pend = 0.0;
    int t_pend = -1;
    PAA_T0()
    for (int t = tl.t0 - h; t < tend; ++t, odd ^= 1) {
        // everything derived from the lane number and from the twiddle seeds is re-formed per frame: hoisted out of the loop, the
        // per-lane LDS addresses of all passes and the 15 twiddle products per butterfly are hundreds of loop-invariant registers
        // (they ended up in scratch: 850 bytes per lane, passes 0 - 2 at a third of their speed)
        lane = lane_id;
        asm volatile("" : "+v"(lane));
#pragma unroll
        for (int u = 0; u < SH::NSD0; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(sd0[u].w[j].x), "+v"(sd0[u].w[j].y));
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(sd1.w[j].x), "+v"(sd1.w[j].y));
        if constexpr (SH::R1B > 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(sd1b.w[j].x), "+v"(sd1b.w[j].y));
        }
        // even frames of the run: transform at the front, spectrum at the very front, previous spectrum behind the buffer;
        // odd frames: transform one unit further, spectrum in its last unit, previous spectrum at the very front
        double2 *buf = reinterpret_cast<double2 *>(wb + (odd ? L.unit_bytes : 0));
        double *cur = reinterpret_cast<double *>(wb + (odd ? L.buf_bytes : 0));
        double *prv = reinterpret_cast<double *>(wb + (odd ? 0 : L.buf_bytes));
        double *st = reinterpret_cast<double *>(buf);
        const T *x = x0 + (long long)t * P.S;
        double2 cw[SH::NCW][SH::RZ];
        chirp_prefetch<SH>(g_chirp, Lseq, lane, cw);
        double y0;
        bool silent;
        if constexpr ((SH::WMAX + 63) / 64 > 32) {
            PAA_TICK(8)
            silent = frame_load_chunked<T, SH>(P, x, nm, st, lane, y0);
        } else {
            double raw[(SH::WMAX + 63) / 64 + 1];
            frame_request<T, SH>(x, W, lane, raw);
            PAA_TICK(8)
            silent = frame_commit<T, SH>(P, raw, nm, st, lane, y0);
        }
        if (t_pend >= 0 && lane < kBase) {
            tri::row_put(rc, oc + (long long)lane * Tc, t_pend, tl.t0, tend, v_pend);
            if (P.deltas) tri::row_put(rcd, oc + (long long)(kBase + lane) * Tc, t_pend, tl.t0, tend, d_pend);
        }
        t_pend = -1;

        PAA_TICK(0)
        const bool want = (P.mode == 0) && ((t >= tl.t0) || (P.deltas && t == tl.t0 - 1));
        TimeFeat tf;
        tf.e_tot = 0.0; tf.ent_e = 0.0; tf.zc = 0;
        if (want) tf = time_features_real(st, ch_t, W, P.blk_t, bnd, lane);
        wsync();
        PAA_TICK(1)
        if (silent) {
            // all samples equal: X[0] = |sum y| / Nf, every other bin exactly 0 (what pocketfft returns for a constant frame)
            const double x0m = fabs((double)W * y0) / (double)Nf;
            for (int k = lane; k < Nf; k += kWave) cur[k] = (k == 0) ? x0m : 0.0;
            wsync();
        } else {
            fwd_pass0<T, SH>(buf, g_chirp, cw, sd0, g_tw, x, nm, Lseq, lane);
            PAA_TICK(2)
            double2 bp[SH::U2][SH::R2];
            bp_load<SH>(g_bp, lane, 0, bp);
            pass1<SH, true>(buf, sd1, lane);
            if constexpr (SH::R1B > 1) pass1<SH, true, 1>(buf, sd1b, lane);
            PAA_TICK(3)
            pass2_product<SH>(buf, g_bp, bp, lane);
            PAA_TICK(4)
            if constexpr (SH::R1B > 1) pass1<SH, false, 1>(buf, sd1b, lane);
            pass1<SH, false>(buf, sd1, lane);
            PAA_TICK(5)
            if constexpr (PK) back_pass0_packed<SH>(buf, cur, sd0, g_tw, g_chirp, g_post, Lseq, lane);
            else back_pass0_magnitudes<SH>(buf, cur, sd0, g_tw, Nf, lane);
            PAA_TICK(6)
        }

        if (P.mode == 1) {            // spectrogram row (ShortTermFeatures.py:422)
            double *row = oc + (long long)t * Nf;
            for (int k = lane; k < Nf; k += kWave) __builtin_nontemporal_store(cur[k], row + k);
        } else if (P.mode == 2) {     // chromagram row (:356-359)
            double p = 0.0;
            for (int k = lane; k < Nf; k += kWave) { const double X = cur[k]; p = fma(X, X, p); }
            p = wsum(p);
            double chv = tri::chroma_sums_balanced(tb, cur, g_chjob[lane], lane);
            chv = (p == 0.0) ? chv / kEps : fast_div(chv, p);
            if (lane < 12) oc[(long long)t * 12 + lane] = chv;
        } else {
            if (want) {
                frame_features_blu(P, tb, tf, cur, (t == 0) ? cur : prv, fv, msp, bnd, ch_f, g_meljob[lane], g_chjob[lane], lane);
                PAA_TICK(7)
                const double v = (lane < kBase) ? fv[lane] : 0.0;
                if (t >= tl.t0) { t_pend = t; v_pend = v; d_pend = (t == 0) ? 0.0 : v - vprev; }
                vprev = v;
            }
        }
        wsync();
        PAA_TICK(10)
    }
    if (t_pend >= 0 && lane < kBase) {
        tri::row_put(rc, oc + (long long)lane * Tc, t_pend, tl.t0, tend, v_pend);
        if (P.deltas) tri::row_put(rcd, oc + (long long)(kBase + lane) * Tc, t_pend, tl.t0, tend, d_pend);
    }
    PAA_TEND()
}

// ---- host: does the window take this kernel, LDS layout, tables ---------------------------------------------------------
inline int blu_log2m(int window) {
    const int Nf = window / 2, need = window + Nf - 1;
    for (int lg = 8; lg <= 13; ++lg)
        if ((1 << lg) >= need) return lg;
    return 0;
}
// even windows as W / 2 complex points: M >= W - 1, lengths 512 .. 4096 (0: not possible)
inline int blu_log2m_packed(int window) {
    if (window % 2) return 0;
    for (int lg = 9; lg <= 12; ++lg)
        if ((1 << lg) >= window - 1) return lg;
    return 0;
}
// radices of the DIF passes of M = 2^lg (the kernel's Sched)
// (r[0], r[1], r[2] [, r[3]]: pass 0, the middle pass(es), the last pass; r[3] = 0 for the three-pass lengths)
inline void blu_radices(int lg, int r[4]) {
    r[3] = 0;
    switch (lg) {
        case 8: r[0] = 4; r[1] = 8; r[2] = 8; break;
        case 9: r[0] = 8; r[1] = 8; r[2] = 8; break;
        case 10: r[0] = 16; r[1] = 8; r[2] = 8; break;
        case 11: r[0] = 16; r[1] = 16; r[2] = 8; break;
        case 12: r[0] = 16; r[1] = 16; r[2] = 16; break;
        default: r[0] = 16; r[1] = 8; r[2] = 8; r[3] = 8; break;
    }
}
// in-place radix-2 FFT in long double (host tables only: M <= 4096)
inline void blu_fft_ld(std::vector<long double> &re, std::vector<long double> &im) {
    const size_t n = re.size();
    for (size_t i = 1, j = 0; i < n; ++i) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) { std::swap(re[i], re[j]); std::swap(im[i], im[j]); }
    }
    const long double two_pi = 6.283185307179586476925286766559005768L;
    for (size_t len = 2; len <= n; len <<= 1) {
        for (size_t k = 0; k < len / 2; ++k) {
            const long double ang = -two_pi * (long double)k / (long double)len;
            const long double wr = cosl(ang), wi = sinl(ang);
            for (size_t i = k; i < n; i += len) {
                const size_t j = i + len / 2;
                const long double xr = re[j] * wr - im[j] * wi, xi = re[j] * wi + im[j] * wr;
                re[j] = re[i] - xr; im[j] = im[i] - xi;
                re[i] += xr; im[i] += xi;
            }
        }
    }
}

// 0: this window is not for the Bluestein kernel.  Fills the layout and the blob (LDS tables first, global tables behind them).
inline int blu_layout(const FftPlan &fft, const MelTable *mel, const ChromaTable *chroma, int F, BluLayout &L,
                      std::vector<unsigned char> *blob) {
    const int W = fft.window, Nf = W / 2;
    if (Nf < 64) return 0;                      // (chunks of the feature stages: at most two entropy blocks per lane)
    std::vector<int> radix;
    if (mix::mix_factor(fft.len, radix)) return 0;       // smooth lengths have their own kernels
    int lg = blu_log2m(W);
    // the packed form when it halves the convolution (or is the only one that fits)
    const int lgp = blu_log2m_packed(W);
    const bool packed = lgp && (!lg || lgp < lg);
    if (packed) lg = lgp;
    if (!lg) return 0;
    const int M = 1 << lg;
    memset(&L, 0, sizeof(L));
    L.log2m = lg;
    L.packed = packed ? 1 : 0;
    const int Lseq = packed ? W / 2 : W, Lout = packed ? W / 2 : Nf;          // elements in, outputs needed
    L.unit_bytes = (Nf * 8 + 255) / 256 * 256;
    L.buf_bytes = M * 16;
    const int FF = F > 0 ? F : 1;
    (void)FF;
    L.wave_bytes = (L.buf_bytes + L.unit_bytes + 48 * 8 + 40 * 8 + 12 * 8 + 255) / 256 * 256;      // + fv[48], msp[40], bnd[12]
    const size_t n_melw = mel ? mel->w.size() : 0, n_ch = chroma ? chroma->src.size() : 0;
    int off = 0;
    auto take = [&off](size_t bytes) { const int o = off; off += (int)((bytes + 15) / 16 * 16); return o; };
    L.off_mello = take(40 * 4);
    L.off_melcnt = take(40 * 4);
    L.off_meloff = take(40 * 4);
    L.off_melw = take(std::max<size_t>(n_melw, 1) * 8);
    L.off_dct = take(13 * 41 * 8);
    L.off_chstart = take(13 * 4);
    L.off_chsrc = take(std::max<size_t>(n_ch, 1) * 4);
    L.off_chw = take(std::max<size_t>(n_ch, 1) * 8);
    off = (off + 255) / 256 * 256;
    L.table_bytes = off;
    int rdx[4];
    blu_radices(lg, rdx);
    const int np = rdx[3] ? 4 : 3;
    const int S0 = M / rdx[0], S1 = S0 / rdx[1], S1B = rdx[3] ? S1 / rdx[2] : 0;
    const int ntw = (rdx[0] - 1) * S0 + (rdx[1] - 1) * S1 + (rdx[3] ? (rdx[2] - 1) * S1B : 0);
    L.off_g_chirp = take((size_t)Lseq * 16);
    L.off_g_post = take(packed ? (size_t)(W / 4 + 1) * 16 : 16);
    L.off_g_bp = take((size_t)M * 16);
    L.off_g_tw = take((size_t)ntw * 16);
    L.off_g_meljob = take(64 * 16);
    L.off_g_chjob = take(64 * 16);
    L.total_bytes = off;
    static const int max_waves[14] = {0, 0, 0, 0, 0, 0, 0, 0, 8, 8, 4, 4, 2, 1};          // (the kernel's Sched<>::NW)
    L.waves = max_waves[lg];
    L.lds_table_bytes = (lg < 13) ? L.table_bytes : 0;
    while (L.waves > 1 && (size_t)L.lds_table_bytes + (size_t)L.waves * L.wave_bytes > 160 * 1024) --L.waves;
    if ((size_t)L.lds_table_bytes + (size_t)L.waves * L.wave_bytes > 160 * 1024) return 0;
    if (!blob) return 1;
    blob->assign((size_t)L.total_bytes, 0);
    unsigned char *b = blob->data();
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
        int first[12], cnt[12];
        for (int c = 0; c < 12; ++c) { first[c] = chroma->class_start[c]; cnt[c] = chroma->class_start[c + 1] - chroma->class_start[c]; }
        tri::lane_jobs(12, first, first, cnt, reinterpret_cast<tri::LaneJob *>(b + L.off_g_chjob));
    }
    if (mel && !mel->w.empty())
        tri::lane_jobs(40, mel->lo.data(), mel->off.data(), mel->cnt.data(), reinterpret_cast<tri::LaneJob *>(b + L.off_g_meljob));
    const long double pi = 3.141592653589793238462643383279502884L;
    // c[m] = exp(i pi m^2 / Lseq): m^2 is reduced mod 2 Lseq in integers first
    auto chirp = [&](long long m, long double &cr, long double &ci) {
        const long long r = (m * m) % (2LL * Lseq);
        const long double ang = pi * (long double)r / (long double)Lseq;
        cr = cosl(ang); ci = sinl(ang);
    };
    if (packed) {
        double *gp = reinterpret_cast<double *>(b + L.off_g_post);
        for (int k = 0; k <= W / 4; ++k) {
            const long double ang = -2.0L * pi * (long double)k / (long double)W;
            gp[2 * k] = (double)cosl(ang);
            gp[2 * k + 1] = (double)sinl(ang);
        }
    }
    double *gc = reinterpret_cast<double *>(b + L.off_g_chirp);
    for (int n = 0; n < Lseq; ++n) {
        long double cr, ci;
        chirp(n, cr, ci);
        gc[2 * n] = (double)cr;
        gc[2 * n + 1] = (double)(-ci);                 // conj(c[n])
    }
    std::vector<long double> re((size_t)M, 0.0L), im((size_t)M, 0.0L);
    for (long long m = -(long long)(Lseq - 1); m <= (long long)Lout - 1; ++m) {
        long double cr, ci;
        chirp(m, cr, ci);
        const size_t idx = (size_t)(((m % M) + M) % M);
        re[idx] = cr; im[idx] = ci;
    }
    blu_fft_ld(re, im);
    // where the DIF passes leave bin k: digit q_p of k (least significant first) at weight M / (R_0 .. R_p)
    double *gb = reinterpret_cast<double *>(b + L.off_g_bp);
    for (int k = 0; k < M; ++k) {
        int rest = k, weight = M, pos = 0;
        for (int p = 0; p < np; ++p) {
            weight /= rdx[p];
            pos += (rest % rdx[p]) * weight;
            rest /= rdx[p];
        }
        gb[2 * pos] = (double)(re[(size_t)k] / (long double)M);
        gb[2 * pos + 1] = (double)(im[(size_t)k] / (long double)M);
    }
    const long double two_pi = 2.0L * pi;
    double *gt = reinterpret_cast<double *>(b + L.off_g_tw);
    size_t at = 0;
    const int spans[3] = {M, S0, S1}, strides[3] = {S0, S1, S1B};
    for (int p = 0; p < np - 1; ++p)
        for (int q = 1; q < rdx[p]; ++q)
            for (int k = 0; k < strides[p]; ++k) {
                const long double ang = -two_pi * (long double)(((long long)q * k) % spans[p]) / (long double)spans[p];
                gt[2 * at] = (double)cosl(ang);
                gt[2 * at + 1] = (double)sinl(ang);
                ++at;
            }
    return 1;
}
inline size_t blu_lds_bytes(const BluLayout &L) { return (size_t)L.lds_table_bytes + (size_t)L.waves * L.wave_bytes; }

}  // namespace blu
}  // namespace paa
