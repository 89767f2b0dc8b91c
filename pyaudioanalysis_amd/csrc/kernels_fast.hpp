// Specialised feature kernel for the headline configuration: int16 PCM, window 800, step 400
// (50 ms / 25 ms at 16 kHz -- BASELINE configs 1-4) or step 800 (the reference's own 50 ms / 50 ms default).
// Any sampling rate (tables are per fs).
//
// One wave = one run of consecutive frames of one clip (a workgroup is NW = 8 or 4 such waves that share the LDS
// tables and nothing else), processed FOUR frames ("a quad") per iteration:
//   stage   : 2000 raw int16 samples (4 frames, 50 % overlap) HBM -> LDS with 16 B/lane loads
//   time    : 50 lanes x 40-sample chunks: sum y^2 and sign changes (integer compares against the
//             clip mean); frames / 80-sample entropy blocks are sums of chunk partials
//   pass 1  : lane (frame f, j) : radix-25 DFT in registers of z[j + 16 r], z = y[2n] + i y[2n+1]
//             (real-input trick, 400-point complex FFT = 25 x 16)
//   exchange: one 4 x 400 double plane in LDS, real parts then imaginary parts
//   pass 2  : lane (frame f, p<13): two radix-16 DFTs (columns p and 25-p share conjugate twiddles),
//             the real-FFT recombination needs exactly Z[k] and Z[400-k], which live in the same
//             lane, so |X[k]|, |X[400-k]| are formed in registers and written once to LDS
//   features: 16 lanes per frame reduce the spectrum (4 frames at once): centroid/spread, entropy,
//             flux against the previous spectrum (kept in a rotating LDS slot), roll-off scan, sparse
//             mel -> log10 -> 13x40 DCT, chroma gather; deltas from the previous column in registers
//   store   : lane = feature row; the row's last seven values wait in registers until a 64-byte aligned chunk of eight
//             frames is complete, which is stored whole and non-temporally (store_row_chunked)
// Halo: a run with t0 > 0 starts its first quad one frame early (two with deltas) and does not store those frames.
//
// Replaces ShortTermFeatures.py:608-682 (+ helpers :22-140, :236-321) for this configuration.
#pragma once
#include <algorithm>
#include <vector>

#include "device_common.hpp"
#include "tables.hpp"

namespace paa {

struct FastTables {
    void *d_blob = nullptr;
};
namespace f800 {
// shared (per workgroup) LDS tables, laid out by the host
struct TabLayout {
    // per-lane padded lists, entry n of lane i at [n * 16 + i]  (conflict-free across the 16 lanes of a frame group)
    int off_w0, off_k0, off_w1, off_k1, off_w2, off_k2;   // mel classes: filter i | filter 16+i | half of filter 32+(i&7)
    int off_chw, off_chk;                                  // chroma gather list of pitch class i
    int off_dct;                                           // 13 rows padded to 41 doubles
    int off_tw2, off_twp;                                  // double2 [16][13]: W400^(r p) and W800^(p + 25 q) of lane p
    int off_sync;                                          // NW = 8 pacing: SIMD id [8], progress in half iterations [8] (ints)
    int melN0, melN1, melN2, chN;                          // list lengths (multiples of 8)
    int mel_clamp;                                         // 1: some padded list reaches past bin 399
    double f0, rf0, r_half_fs, f0sq;                       // fs / 800, its reciprocal, 2 / fs, f0^2 (host-computed: scalar registers)
    int fixed_lists;                                       // 1: lengths are exactly 8/16/16/8 without clamping
    int pace_mode;                                         // NW = 8: 0 none, 1 the two waves of a SIMD keep equal progress
    int total;                                             // bytes, multiple of 16
};
}  // namespace f800
struct FastLaunch {
    int run = 0;
    size_t lds = 0;
    const char *name = "";
    int variant = 0;
    int waves_per_cu = 4;
    f800::TabLayout layout;
};

inline void fast_tables_free(FastTables &t) {
    if (t.d_blob) (void)hipFree(t.d_blob);
    t.d_blob = nullptr;
}

namespace f800 {

constexpr int W = 800, NF = 400, QUAD = 4;
constexpr int RAW_PAD = 8;                          // raw[RAW_PAD + i]; raw[RAW_PAD - 1] = sample before
constexpr int CHUNK = 40;                           // time-domain partials are formed over 40-sample chunks
constexpr int FV_STRIDE = 34;
typedef short s16x2 __attribute__((ext_vector_type(2)));
constexpr int TW_STRIDE = 13;                       // twiddle tables: one column per ACTIVE pass-2 lane (p = 0..12)
constexpr int CH_STRIDE = 12;                       // chroma gather lists: one column per pitch class

// step-dependent geometry (S = 400: 50 % overlap, the BASELINE shape; S = 800: back-to-back frames, the
// reference's own default 50 ms / 50 ms)
template <int S, int NW = 4>
struct Geo {
    static_assert(S % CHUNK == 0 && S % 8 == 0, "step must be a multiple of the 40-sample chunk");
    static constexpr int RAW_N = (QUAD - 1) * S + W;            // samples per quad: 2000 / 3200
    static constexpr int NCHUNK = RAW_N / CHUNK;                // 50 / 80
    static constexpr int CPF = S / CHUNK;                       // chunks per frame step: 10 / 20
    static constexpr int NPRE = (RAW_N / 8 + 63) / 64;          // 16-byte prefetch registers per lane: 4 / 7
    // samples in front of raw[]: raw[PAD - 1] = the sample before the quad (16 bytes keep raw + PAD vector aligned).
    // The two-waves-per-SIMD layout of step 800 has no room for it (3200 samples fill the two spectrum slots exactly):
    // there lane 0 keeps that sample in a register (it owns chunk 0).
    static constexpr int PAD = (NW == 8 && S == 800) ? 0 : RAW_PAD;
    static constexpr int RAW_BYTES = (RAW_N + 2 * PAD) * 2;
    // NW = 8 (two waves per SIMD): the per-wave LDS must stay below 17.5 KB, so the transient buffers live inside the
    // spectrum ring: raw[] in two index-adjacent TARGET slots of the quad (dead until the exchange writes them),
    // msp[] / fv[] in the PREVIOUS-spectrum slot (dead once the flux operands are in registers)
    static constexpr bool ALIAS = (NW == 8);
    // LDS carve (bytes)
    static constexpr int OFF_SPEC = 0;                                   // 5 slots x 400 doubles
    static constexpr int OFF_RAW = OFF_SPEC + 5 * NF * 8;                // int16 raw[]; aliased by msp[4][40] later
    static constexpr int OFF_CE = ALIAS ? OFF_RAW : OFF_RAW + RAW_BYTES; // double cE[NCHUNK]
    static constexpr int OFF_CZ = OFF_CE + NCHUNK * 8;                   // int cZ[NCHUNK], cF[NCHUNK]
    static constexpr int OFF_FV = OFF_CZ + 2 * NCHUNK * 4;               // double fv[4][34]
    static constexpr int LDS_BYTES = ALIAS ? OFF_FV : OFF_FV + QUAD * FV_STRIDE * 8;
    static constexpr int WAVE_BYTES = ((LDS_BYTES + 15) / 16) * 16;
    static_assert(OFF_RAW % 16 == 0 && OFF_CE % 8 == 0 && OFF_FV % 8 == 0, "LDS alignment");
    static_assert(QUAD * 40 * 8 <= RAW_BYTES, "msp alias fits in the raw buffer");
    static_assert(!ALIAS || RAW_BYTES <= 2 * NF * 8, "raw[] fits in two spectrum slots");
    static_assert(!ALIAS || QUAD * 40 * 8 + QUAD * FV_STRIDE * 8 <= NF * 8, "msp + fv fit in one spectrum slot");
};

// ---- register DFTs ------------------------------------------------------------------------
// cos(144 deg) = -1/2 - cos(72 deg), so  a0 + c1 t1 + c2 t2 = (a0 - t2/2) + c1 (t1 - t2)  and
// a0 + c2 t1 + c1 t2 = (a0 - t1/2) - c1 (t1 - t2).  In this form five EQUAL inputs give exactly zero in the four
// non-DC outputs (t1 = t2 = 2 a0: every bracket is an exact zero), where the textbook form leaves a0 * 1e-16.  Frames of
// digital silence are constant after the clip mean is removed; the reference (pocketfft: radix-4 passes first, plain
// differences of equal numbers) returns exact zeros for their non-DC bins, and log10(E + eps) of a mel band resolves 1e-24.
__device__ __forceinline__ void dft5r(double2 &a0, double2 &a1, double2 &a2, double2 &a3, double2 &a4) {
    const double c1 = 0.30901699437494742410;
    const double s1 = 0.95105651629515357212, s2 = 0.58778525229247312917;
    const double2 t1 = cadd(a1, a4), t2 = cadd(a2, a3), t3 = csub(a1, a4), t4 = csub(a2, a3);
    const double2 d = csub(t1, t2);
    const double2 h1 = make_double2(fma(-0.5, t1.x, a0.x), fma(-0.5, t1.y, a0.y));
    const double2 h2 = make_double2(fma(-0.5, t2.x, a0.x), fma(-0.5, t2.y, a0.y));
    const double2 m1 = make_double2(fma(c1, d.x, h2.x), fma(c1, d.y, h2.y));
    const double2 m2 = make_double2(fma(-c1, d.x, h1.x), fma(-c1, d.y, h1.y));
    const double2 n1 = make_double2(fma(s2, t4.x, s1 * t3.x), fma(s2, t4.y, s1 * t3.y));
    const double2 n2 = make_double2(fma(-s1, t4.x, s2 * t3.x), fma(-s1, t4.y, s2 * t3.y));
    a0 = make_double2(a0.x + t1.x + t2.x, a0.y + t1.y + t2.y);
    a1 = sub_i(m1, n1);
    a4 = add_i(m1, n1);
    a2 = sub_i(m2, n2);
    a3 = add_i(m2, n2);
}
__device__ __forceinline__ void dft4r(double2 &a0, double2 &a1, double2 &a2, double2 &a3) {
    const double2 t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), t3 = csub(a1, a3);
    a0 = cadd(t0, t2);
    a2 = csub(t0, t2);
    a1 = sub_i(t1, t3);
    a3 = add_i(t1, t3);
}
// v[r], r = 5 r1 + r2  ->  result for output q stored at v[5 (q % 5) + q / 5]
// SERIAL = 1 (two waves per SIMD): every butterfly finishes in place before the next one starts.  Dependent FP64
// operations issue back to back on gfx950 (profiles/microbench_r02.txt), so nothing is lost, and the scheduler
// cannot run all butterflies of a stage side by side (which doubles the live registers).
template <int SERIAL>
__device__ __forceinline__ void dft25_tail(double2 *v);
template <int SERIAL = 0>
__device__ __forceinline__ void dft25(double2 *v) {
#pragma unroll
    for (int r2 = 0; r2 < 5; ++r2) {
        dft5r(v[r2], v[5 + r2], v[10 + r2], v[15 + r2], v[20 + r2]);
        if (SERIAL) __builtin_amdgcn_sched_barrier(0);
    }
    dft25_tail<SERIAL>(v);
}
// twiddles W25^(r2 q1) and the second stage
template <int SERIAL>
__device__ __forceinline__ void dft25_tail(double2 *v) {
#pragma unroll
    for (int q1 = 1; q1 < 5; ++q1)
#pragma unroll
        for (int r2 = 1; r2 < 5; ++r2) {
            const int m = r2 * q1;
            // literal constants (folded at compile time)
            const double cr = (m == 1) ? 0.96858316112863108 : (m == 2) ? 0.87630668004386358 : (m == 3) ? 0.72896862742141155
                            : (m == 4) ? 0.53582679497899666 : (m == 6) ? 0.06279051952931337 : (m == 8) ? -0.42577929156507272
                            : (m == 9) ? -0.63742398974868975 : (m == 12) ? -0.99211470131447788 : -0.63742398974868975;
            const double ci = (m == 1) ? -0.24868988716485479 : (m == 2) ? -0.48175367410171532 : (m == 3) ? -0.68454710592868873
                            : (m == 4) ? -0.84432792550201508 : (m == 6) ? -0.99802672842827156 : (m == 8) ? -0.90482705246601958
                            : (m == 9) ? -0.77051324277578925 : (m == 12) ? -0.12533323356430426 : 0.77051324277578925;
            v[5 * q1 + r2] = cmul(v[5 * q1 + r2], make_double2(cr, ci));
        }
    if (SERIAL) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q1 = 0; q1 < 5; ++q1) {
        dft5r(v[5 * q1], v[5 * q1 + 1], v[5 * q1 + 2], v[5 * q1 + 3], v[5 * q1 + 4]);
        if (SERIAL) __builtin_amdgcn_sched_barrier(0);
    }
}
#define PAA_DFT25_POS(q) (5 * ((q) % 5) + (q) / 5)

// First-stage radix-5 butterfly straight from five packed int16 pairs (lo = real, hi = imaginary part) minus the
// integer clip mean m: the sums and differences a1 +- a4, a2 +- a3 and a0 + a1 + .. + a4 are whole numbers, so they
// are formed in 32-bit integer arithmetic (half the issue cost of FP64 on gfx950) and converted once.  Same values as
// dft5r on the converted samples (every integer involved is exact in f64).
__device__ __forceinline__ void dft5r_first(int w0, int w1, int w2, int w3, int w4, int m, double2 &a0, double2 &a1,
                                            double2 &a2, double2 &a3, double2 &a4) {
    const double c1 = 0.30901699437494742410;
    const double s1 = 0.95105651629515357212, s2 = 0.58778525229247312917;
    const int x0r = (int)(short)(w0 & 0xffff), x0i = w0 >> 16, x1r = (int)(short)(w1 & 0xffff), x1i = w1 >> 16;
    const int x2r = (int)(short)(w2 & 0xffff), x2i = w2 >> 16, x3r = (int)(short)(w3 & 0xffff), x3i = w3 >> 16;
    const int x4r = (int)(short)(w4 & 0xffff), x4i = w4 >> 16;
    const int m2 = 2 * m;
    const int t1r = x1r + x4r - m2, t1i = x1i + x4i - m2, t2r = x2r + x3r - m2, t2i = x2i + x3i - m2;
    const int t3r = x1r - x4r, t3i = x1i - x4i, t4r = x2r - x3r, t4i = x2i - x3i;
    const int b0r = x0r - m, b0i = x0i - m;
    // the brackets of dft5r's exact-zero form, still in integers: d = t1 - t2, 2 h1 = 2 a0 - t1, 2 h2 = 2 a0 - t2
    const double2 D = make_double2((double)(t1r - t2r), (double)(t1i - t2i));
    const double2 H1 = make_double2(0.5 * (double)(2 * b0r - t1r), 0.5 * (double)(2 * b0i - t1i));
    const double2 H2 = make_double2(0.5 * (double)(2 * b0r - t2r), 0.5 * (double)(2 * b0i - t2i));
    const double2 T3 = make_double2((double)t3r, (double)t3i), T4 = make_double2((double)t4r, (double)t4i);
    const double2 M1 = make_double2(fma(c1, D.x, H2.x), fma(c1, D.y, H2.y));
    const double2 M2 = make_double2(fma(-c1, D.x, H1.x), fma(-c1, D.y, H1.y));
    const double2 N1 = make_double2(fma(s2, T4.x, s1 * T3.x), fma(s2, T4.y, s1 * T3.y));
    const double2 N2 = make_double2(fma(-s1, T4.x, s2 * T3.x), fma(-s1, T4.y, s2 * T3.y));
    a0 = make_double2((double)(b0r + t1r + t2r), (double)(b0i + t1i + t2i));
    a1 = sub_i(M1, N1);
    a4 = add_i(M1, N1);
    a2 = sub_i(M2, N2);
    a3 = add_i(M2, N2);
}
// dft25 whose first stage reads the packed samples w[r] (r = 5 r1 + r2 like v[])
template <int SERIAL>
__device__ __forceinline__ void dft25_packed(const int *w, int m, double2 *v) {
#pragma unroll
    for (int r2 = 0; r2 < 5; ++r2) {
        dft5r_first(w[r2], w[5 + r2], w[10 + r2], w[15 + r2], w[20 + r2], m, v[r2], v[5 + r2], v[10 + r2], v[15 + r2], v[20 + r2]);
        if (SERIAL) __builtin_amdgcn_sched_barrier(0);
    }
    dft25_tail<SERIAL>(v);
}

// v[r], r = 4 r1 + r2  ->  result for output q stored at v[4 (q % 4) + q / 4]
template <int SERIAL = 0>
__device__ __forceinline__ void dft16(double2 *v) {
    const double c = 0.92387953251128676, s = 0.38268343236508977, h = 0.70710678118654752;
#pragma unroll
    for (int r2 = 0; r2 < 4; ++r2) {
        dft4r(v[r2], v[4 + r2], v[8 + r2], v[12 + r2]);
        if (SERIAL) __builtin_amdgcn_sched_barrier(0);
    }
    // twiddles W16^(r2 q1) at index 4 q1 + r2
    v[5] = cmul(v[5], make_double2(c, -s));        // m = 1
    v[6] = make_double2(h * (v[6].x + v[6].y), h * (v[6].y - v[6].x));    // m = 2: (h, -h)
    v[7] = cmul(v[7], make_double2(s, -c));        // m = 3
    v[9] = make_double2(h * (v[9].x + v[9].y), h * (v[9].y - v[9].x));    // m = 2
    v[10] = make_double2(v[10].y, -v[10].x);       // m = 4: -i
    v[11] = make_double2(h * (v[11].y - v[11].x), -h * (v[11].x + v[11].y));   // m = 6: (-h, -h)
    v[13] = cmul(v[13], make_double2(s, -c));      // m = 3
    v[14] = make_double2(h * (v[14].y - v[14].x), -h * (v[14].x + v[14].y));   // m = 6
    v[15] = cmul(v[15], make_double2(-c, s));      // m = 9
    if (SERIAL) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q1 = 0; q1 < 4; ++q1) {
        dft4r(v[4 * q1], v[4 * q1 + 1], v[4 * q1 + 2], v[4 * q1 + 3]);
        if (SERIAL) __builtin_amdgcn_sched_barrier(0);
    }
}
#define PAA_DFT16_POS(q) (4 * ((q) % 4) + (q) / 4)

// value of lane 15 of the row in every lane (v_readlane-free: row_bcast is not available inside a row, so
// take the maximum of a non-negative non-decreasing scan instead: the inclusive scan's largest entry)
__device__ __forceinline__ double dpp_bcast15(double v) { return group_max(v); }

// fast_sqrt / mag_sqrt / fast_div / fast_log2 / fast_log10: device_common.hpp (shared with the other feature kernels)

// wave-uniform values computed with vector instructions: pin them into scalar registers
__device__ __forceinline__ double uni(double v) {
    const unsigned long long b = __double_as_longlong(v);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// optional per-phase cycle accounting (build with -DPAA_F800_TIMING; read with paa_debug_phase_cycles)
#ifdef PAA_F800_TIMING
// (internal linkage: every translation unit that instantiates kernels -- family_*.hip -- has its own copy; PAA_PHASE_READER
// below defines the unit's reader and paa_debug_phase_cycles adds the units up)
static __device__ unsigned long long g_phase_cycles[16];
static __device__ unsigned long long g_wave_trace[4096 * 4];      // per wave: realtime start / end (100 MHz), cycles, HW_ID | XCC << 32
#define PAA_T0() unsigned long long t_prev_ = __builtin_readcyclecounter(); unsigned long long t_acc_[12] = {0,0,0,0,0,0,0,0,0,0,0,0}; \
    const unsigned long long t_rt0_ = __builtin_amdgcn_s_memrealtime(), t_c0_ = t_prev_;
#define PAA_TICK(idx) { const unsigned long long t_now_ = __builtin_readcyclecounter(); t_acc_[idx] += t_now_ - t_prev_; t_prev_ = t_now_; }
#define PAA_TEND() if (lane == 0) { for (int z_ = 0; z_ < 12; ++z_) atomicAdd(&g_phase_cycles[z_], t_acc_[z_]); atomicAdd(&g_phase_cycles[15], 1ULL); \
    if (tile_id < 4096) { g_wave_trace[4 * tile_id] = t_rt0_; g_wave_trace[4 * tile_id + 1] = __builtin_amdgcn_s_memrealtime(); \
        g_wave_trace[4 * tile_id + 2] = __builtin_readcyclecounter() - t_c0_; \
        g_wave_trace[4 * tile_id + 3] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32); } }
#elif defined(PAA_F800_TRACE)
// light-weight variant: only the per-wave life times (two clock reads and four stores per wave)
static __device__ unsigned long long g_phase_cycles[16];
static __device__ unsigned long long g_wave_trace[4096 * 4];
#define PAA_T0() const unsigned long long t_rt0_ = __builtin_amdgcn_s_memrealtime(), t_c0_ = __builtin_readcyclecounter();
#define PAA_TICK(idx)
#define PAA_TEND() if (lane == 0) { atomicAdd(&g_phase_cycles[15], 1ULL); \
    if (tile_id < 4096) { g_wave_trace[4 * tile_id] = t_rt0_; g_wave_trace[4 * tile_id + 1] = __builtin_amdgcn_s_memrealtime(); \
        g_wave_trace[4 * tile_id + 2] = __builtin_readcyclecounter() - t_c0_; \
        g_wave_trace[4 * tile_id + 3] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32); } }
#else
#define PAA_T0()
#define PAA_TICK(idx)
#define PAA_TEND()
#endif
// reader of this translation unit's phase counters: adds them into acc16[16] and clears them; when trace != nullptr it also
// copies the per-wave trace of the unit's last launch (4 words per run) and returns the number of runs copied
#if defined(PAA_F800_TIMING) || defined(PAA_F800_TRACE)
#define PAA_PHASE_READER(NAME)                                                                                          \
    int NAME(unsigned long long *acc16, unsigned long long *trace, int max_waves) {                                     \
        unsigned long long host[16], zero[16] = {0};                                                                    \
        if (hipMemcpyFromSymbol(host, HIP_SYMBOL(f800::g_phase_cycles), sizeof(host)) != hipSuccess) return -1;          \
        for (int i = 0; i < 16; ++i) acc16[i] += host[i];                                                               \
        if (hipMemcpyToSymbol(HIP_SYMBOL(f800::g_phase_cycles), zero, sizeof(zero)) != hipSuccess) return -1;            \
        if (!trace) return 0;                                                                                           \
        const int n = max_waves < 4096 ? max_waves : 4096;                                                              \
        if (hipMemcpyFromSymbol(trace, HIP_SYMBOL(f800::g_wave_trace), (size_t)n * 4 * sizeof(unsigned long long)) != hipSuccess) return -1; \
        return n;                                                                                                       \
    }
#else
#define PAA_PHASE_READER(NAME) int NAME(unsigned long long *, unsigned long long *, int) { return 0; }
#endif

// Row store, in whole 64-byte chunks of the row.  The rows of a [F][T] slab are only 8-byte aligned (T is odd in general)
// and a wave produces four frames per row and iteration; stored as they come, the memory side sees partial-sector writes
// (measured with rocprofv3 WRITE_SIZE: 2.3x the output bytes; 1.76x with 32-byte aligned groups, because the L2 -- under the
// pressure of the sample stream -- evicts half-written 64-byte blocks; scripts/microbench/mb3.hip calibrates the counter and
// shows that 64-byte aligned 64-byte groups cost 1.02x).  So every lane keeps its row's last seven values h[0..6] (frames
// q0-7 .. q0-1; v[0..3] = the quad's frames q0 .. q0+3) and, once a chunk [g, g+8) with (row + g) 64-byte aligned is complete,
// stores it as a whole: with a = alignment of frame q0 in doubles, the chunk is the window w[s .. s+7], s = 7 - a, complete
// with this quad when s <= 3 -- half of the lanes store with every iteration.  Frames outside [lo, hi) belong to another wave
// (or do not exist): the first chunk of a run and the tail after its last quad fall back to 8-byte stores.
// (per-lane selects through v_cndmask on a ballot mask: written as `c ? w[j + 2] : w[j]` the optimiser recognises a
// dynamically indexed array and moves it to scratch memory)
__device__ __forceinline__ double sel64(unsigned long long take_a, double a, double b) {
    int lo, hi;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(lo) : "v"(__double2loint(b)), "v"(__double2loint(a)), "s"(take_a));
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(hi) : "v"(__double2hiint(b)), "v"(__double2hiint(a)), "s"(take_a));
    return __hiloint2double(hi, lo);
}
// the part [vlo, vhi) of one 64-byte aligned chunk, in the fewest naturally aligned pieces (32, 16, 8 bytes): a partial
// store costs one 32-byte write request whatever its size (measured), so a chunk shared by two runs is written in 2..4
// requests instead of 8
__device__ __forceinline__ void store_chunk_pieces(double *dst, const double (&o)[8], int vlo, int vhi, int dbg = 0) {
    // (-DPAA_EXPERIMENTS builds only: PAA_KERNEL_DEBUG bit 4 drops partial chunks, bit 8 whole ones -- traffic experiments)
    if (PAA_DEBUG_BIT(dbg, 4) && !(vlo <= 0 && vhi >= 8)) return;
    if (PAA_DEBUG_BIT(dbg, 8) && (vlo <= 0 && vhi >= 8)) return;
    (void)dbg;
    typedef double f64x4 __attribute__((ext_vector_type(4), aligned(32)));
    typedef double f64x2 __attribute__((ext_vector_type(2), aligned(16)));
    if (vlo <= 0 && vhi >= 8) {                           // the whole chunk: four 16-byte stores back to back
        __builtin_nontemporal_store(f64x4{o[0], o[1], o[2], o[3]}, reinterpret_cast<f64x4 *>(dst));
        __builtin_nontemporal_store(f64x4{o[4], o[5], o[6], o[7]}, reinterpret_cast<f64x4 *>(dst + 4));
        return;
    }
#pragma unroll
    for (int b4 = 0; b4 < 8; b4 += 4) {
        if (vlo <= b4 && vhi >= b4 + 4) {
            *reinterpret_cast<f64x4 *>(dst + b4) = f64x4{o[b4], o[b4 + 1], o[b4 + 2], o[b4 + 3]};
        } else {
#pragma unroll
            for (int b2 = b4; b2 < b4 + 4; b2 += 2) {
                if (vlo <= b2 && vhi >= b2 + 2) {
                    *reinterpret_cast<f64x2 *>(dst + b2) = f64x2{o[b2], o[b2 + 1]};
                } else {
                    if (vlo <= b2 && vhi > b2) dst[b2] = o[b2];
                    if (vlo <= b2 + 1 && vhi > b2 + 1) dst[b2 + 1] = o[b2 + 1];
                }
            }
        }
    }
}
__device__ __forceinline__ void store_row_chunked(double *row, int q0, int lo, int hi, bool last, const double (&h)[7],
                                                  const double (&v)[4], int dbg) {
    const double w[11] = {h[0], h[1], h[2], h[3], h[4], h[5], h[6], v[0], v[1], v[2], v[3]};      // w[j] = frame q0 - 7 + j
    const int s = 7 - (int)(((reinterpret_cast<uintptr_t>(row) >> 3) + (unsigned)q0) & 7);
    const bool emit = s <= 3;
    {
        const unsigned long long m1 = __builtin_amdgcn_ballot_w64((s & 2) != 0), m0 = __builtin_amdgcn_ballot_w64((s & 1) != 0);
        double x[9], o[8];
#pragma unroll
        for (int j = 0; j < 9; ++j) x[j] = sel64(m1, w[j + 2], w[j]);
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = sel64(m0, x[k + 1], x[k]);
        const int g = q0 - 7 + s;
        if (emit) store_chunk_pieces(row + g, o, lo - g, hi - g, dbg);
    }
    if (last) {                                           // the values still waiting for their chunk: w[ns ..], ns = 4 .. 11
        const int ns = emit ? s + 8 : s, g = q0 - 7 + ns, u = ns - 4;
        const unsigned long long m2 = __builtin_amdgcn_ballot_w64((u & 4) != 0), m1 = __builtin_amdgcn_ballot_w64((u & 2) != 0),
                                 m0 = __builtin_amdgcn_ballot_w64((u & 1) != 0);
        double z[7], y[7], o[8];                          // (indices past the window repeat w[10]: never stored, hi - g cuts them)
#pragma unroll
        for (int j = 0; j < 7; ++j) z[j] = sel64(m2, w[(j + 8 > 10) ? 10 : j + 8], w[j + 4]);
#pragma unroll
        for (int j = 0; j < 7; ++j) y[j] = sel64(m1, z[(j + 2 > 6) ? 6 : j + 2], z[j]);
#pragma unroll
        for (int j = 0; j < 7; ++j) o[j] = sel64(m0, y[(j + 1 > 6) ? 6 : j + 1], y[j]);
        o[7] = o[6];
        store_chunk_pieces(row + g, o, lo - g, hi - g, dbg);
    }
}

// FIXED = 1: the mel / chroma list lengths are the compile-time constants of the usual 16 kHz tables (8, 16, 16, 8,
// no clamping), which turns the whole feature stage into straight-line code the scheduler can interleave;
// FIXED = 0: run-time lengths from the layout (other sampling rates).
// NW = waves per workgroup: 4 (one wave per SIMD, up to 512 registers: loads are software-pipelined far ahead) or
// 8 (two per SIMD, <= 256 registers and 16.8 KB of LDS per wave: the partner wave hides the latencies instead, so table
// values are fetched right before their use and the transient LDS buffers live inside the spectrum ring).
template <int S, int DELTAS, int FIXED, int NW>
__global__ __launch_bounds__(64 * NW, NW / 4) void st_fast_800_kernel(PlanDev P, TabLayout L,
                                                                     const unsigned char *__restrict__ blob,
                                                                     const int16_t *__restrict__ sig,
                                                                     const ClipDev *__restrict__ clips,
                                                                     const ClipNorm *__restrict__ norms,
                                                                     const Tile *__restrict__ tiles, int n_tiles,
                                                                     double *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // ---------------- shared tables: the host-built blob has exactly the LDS layout; copy it once per workgroup
    {
        const int4 *src4 = reinterpret_cast<const int4 *>(blob);
        int4 *dst4 = reinterpret_cast<int4 *>(smem);
        for (int n = threadIdx.x; n < L.total / 16; n += 64 * NW) dst4[n] = src4[n];
    }
    __syncthreads();       // the only workgroup-wide barrier; from here on every wave runs on its own
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

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);    // wave-uniform: keeps per-wave constants in SGPRs
    const int tile_id = blockIdx.x * NW + wave;
    // NW = 8: the two waves of a SIMD are paced against each other.  The hardware issues oldest-first, so without pacing
    // the older wave of a pair runs at nearly its single-wave speed, finishes early and leaves the younger one alone on the
    // SIMD for a long tail.  Every iteration a wave publishes how many quads it still has to do and raises its priority
    // when it is behind its partner (s_setprio outranks age).  pace[0..7] = SIMD id, pace[8..15] = progress.
    // (Work stealing between the waves of a workgroup and a half-iteration pacing offset were measured and rejected:
    // scripts/experiments/r02_work_stealing_pace2.diff.)
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
    using G = Geo<S, NW>;
    constexpr int RAW_N = G::RAW_N, NCHUNK = G::NCHUNK, CPF = G::CPF;
    unsigned char *wbase = smem + L.total + wave * G::WAVE_BYTES;
    double *spec = reinterpret_cast<double *>(wbase + G::OFF_SPEC);
    // NW = 4: fixed buffers.  NW = 8: re-pointed every iteration into the spectrum ring (see Geo)
    int16_t *raw = reinterpret_cast<int16_t *>(wbase + (G::ALIAS ? 0 : G::OFF_RAW));
    double *msp = reinterpret_cast<double *>(wbase + (G::ALIAS ? 0 : G::OFF_RAW));   // alias: raw is dead by then
    double *cE = reinterpret_cast<double *>(wbase + G::OFF_CE);
    int *cZ = reinterpret_cast<int *>(wbase + G::OFF_CZ);
    int *cF = cZ + NCHUNK;
    double *fv = reinterpret_cast<double *>(wbase + (G::ALIAS ? 0 : G::OFF_FV));

    const int lane = threadIdx.x & 63;
    int g = lane >> 4, i = lane & 15;          // NW = 8 makes them opaque per iteration (see the loop head)
    PAA_T0()
    const Tile tl = tiles[tile_id];
    const ClipDev c = clips[tl.clip];
    const ClipNorm nm = wave_clip_norm<int16_t>(P, c, norms, tl.clip, lane);      // (formed from the partials in this prologue)
    const int16_t *xc = sig + c.sample_off;
    const long long Tc = c.T;
    double *oc = out + c.out_off;

    // everything below is wave-uniform (one tile per wave): the clip constants come precomputed from
    // clip_params_kernel (see ClipNorm) through scalar loads, the rest is pinned into scalar registers
    const double f0 = L.f0, rf0 = L.rf0, r_half_fs = L.r_half_fs, f0sq = L.f0sq;
    const int m_int = nm.m_int;
    const double delta_mu = nm.delta_mu;                                         // |.| <= 1/2
    const double y_scale2 = nm.y_scale2;                                         // y = (x' - delta) * inv / 2^15
    const double mag_scale = nm.mag_scale;                                       // 0.5: E and O carry a factor 1/2
    const double dc_shift = nm.dc_shift;
    // sign(x/2^15 - mean) = sign(x - mu) in packed 16-bit integers (mu lies inside the int16 range: it is a mean of int16)
    // as non-negative sign CODES (device_common.hpp: only |differences| are summed): code = clamp(sat(x - (zb - 1)), lo, 2),
    // lo = 0 when mu is a whole number (codes 0 / 1 / 2 = signs -1 / 0 / +1), 1 otherwise (codes 1 / 2: differences count
    // double, applied once per frame); |code - code'| of both halves is ONE v_sad_u16
    const bool mu_whole = nm.mu_whole != 0;
    const short zb1_ = (short)max(nm.zb - 1, -32768);       // (zb = -32768: every sample equals the mean, every code is equal)
    const s16x2 zc_b = {zb1_, zb1_};
    const s16x2 zc_lo = mu_whole ? (s16x2){0, 0} : (s16x2){1, 1};
    const s16x2 zc_two = {2, 2};
    const s16x2 zc_one = {1, 1};
    const int zc_shift = mu_whole ? 0 : 1;

    const int r0 = tl.t0, t_end = tl.t0 + tl.cnt;      // frames [r0, t_end) are this wave's to store
    // a run that starts inside a clip begins its first quad HALO frames early: the spectrum of frame r0 - 1 (flux of the first stored
    // frame) and, with deltas, the complete features of r0 - 1 (whose flux needs the spectrum of r0 - 2) come out of that quad, whose
    // other frames are already the run's own -- until round 5 a whole halo quad ran first (one iteration of 19 for nothing)
    constexpr int HALO = DELTAS ? 2 : 1;
    int q0 = r0 >= HALO ? r0 - HALO : 0;
    int slot0 = 2;                   // slots of a quad: slot0 .. slot0+3 (mod 5) after the rotation at the loop head; previous = slot0-1
    double vlast = 0.0;              // lane l < 34: feature l of the frame before this quad
    double hold[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};    // ... and the row's last seven values / deltas, waiting for
    double holdd[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};   // their 64-byte chunk (store_row_chunked)
    bool first_quad = true;          // the run's first iteration (its first HALO frames belong to the run before when r0 > 0)
    int n_done = 0;                  // quads of this run whose FFT stages are complete (pacing)

    // software prefetch of the next quad's samples (NPRE x 16 B per lane) into registers
    int4 pre[G::NPRE];
#pragma unroll
    for (int r = 0; r < G::NPRE; ++r) pre[r] = make_int4(0, 0, 0, 0);
    bool pre_ok;
#define PAA_F800_FETCH(q)                                                                              \
    {                                                                                                  \
        const long long base_ = (long long)(q) * S;                                                    \
        const int16_t *src_ = xc + base_;                                                              \
        pre_ok = ((reinterpret_cast<uintptr_t>(src_) & 15) == 0) && (c.n - base_ >= RAW_N);            \
        if (pre_ok) {                                                                                  \
            const int4 *s4_ = reinterpret_cast<const int4 *>(src_);                                    \
            _Pragma("unroll") for (int r_ = 0; r_ < G::NPRE; ++r_)                                     \
                if (lane + 64 * r_ < RAW_N / 8) pre[r_] = s4_[lane + 64 * r_];                         \
        }                                                                                              \
    }
    if (NW == 4) PAA_F800_FETCH(q0)

    int16_t before_next = 0;         // lane 0: the sample just before the NEXT quad (saved while raw[] still holds it)
    int16_t before_reg = 0;          // lane 0: the sample just before THIS quad
    for (;; first_quad = false) {
        // next quad
        if (!first_quad) q0 += QUAD;
        if (q0 >= t_end) break;
        slot0 = (slot0 + 4) % 5;
        // NW = 8: hide the lane indices from loop-invariant code motion -- the dozens of per-lane LDS addresses the
        // compiler would otherwise keep in registers across the whole iteration cost more than re-deriving them
        if (NW != 4) asm volatile("" : "+v"(g), "+v"(i));
        // pacing of the two waves of a SIMD: progress is counted in half iterations (loop head, end of the FFT) and the
        // pair is kept level
#define PAA_F800_PACE(half_)                                                                           \
        if (NW == 8 && L.pace_mode != 0) {                                                             \
            const int mine_ = 2 * n_done + (half_);                                                    \
            if (lane == 0) pace[8 + wave] = mine_;                                                     \
            const int other_ = __builtin_amdgcn_readfirstlane(pace[8 + partner]);                      \
            const int d_ = mine_ - other_;                                                             \
            if (d_ < 0) __builtin_amdgcn_s_setprio(3);                                                 \
            else if (d_ > 0) __builtin_amdgcn_s_setprio(0);                                            \
            else __builtin_amdgcn_s_setprio(1);                                                        \
        }
        PAA_F800_PACE(0)
        // pass-2 columns of this lane; the twiddles W400^(r p) and W800^(p + 25 q) sit in the shared LDS table
        const int pa = i, pb = (i == 0) ? 0 : 25 - i;
        const bool act = i < 13;
        const int itw = min(i, TW_STRIDE - 1), ich = min(i, CH_STRIDE - 1);     // idle lanes re-read a valid column
        if (G::ALIAS) {
            // previous-spectrum slot = slot0 - 1; the quad's four target slots are the others.  Two index-adjacent
            // target slots always exist: (3, 4) unless the previous slot is 3 or 4, then (0, 1)
            const int prev_slot = (slot0 + 4) % 5;
            raw = reinterpret_cast<int16_t *>(spec + ((prev_slot <= 2) ? 3 : 0) * NF);
            msp = spec + prev_slot * NF;
            fv = msp + QUAD * 40;
        }
        // ---------------- stage raw samples [q0*S - 1, q0*S + 2000)
        {
            const long long base = (long long)q0 * S;
            const int16_t *src = xc + base;
            // the sample before the quad: saved from the previous iteration's staging (one global load per run
            // instead of an exposed one per iteration)
            int16_t before = 0;
            if (lane == 0) before = (!first_quad) ? before_next : ((base > 0) ? src[-1] : src[0]);
            // NW = 8: no register prefetch across the loop edge (16 live registers through every stage would cost more
            // than the exposed load: the partner wave of the SIMD computes meanwhile)
            if (NW != 4) PAA_F800_FETCH(q0)
            if (pre_ok) {
                int4 *d4 = reinterpret_cast<int4 *>(raw + G::PAD);
#pragma unroll
                for (int r = 0; r < G::NPRE; ++r)
                    if (lane + 64 * r < RAW_N / 8) d4[lane + 64 * r] = pre[r];
            } else {
                const long long avail = c.n - base;
                for (int n = lane; n < RAW_N; n += 64) raw[G::PAD + n] = (n < avail) ? src[n] : (int16_t)0;
            }
            if (G::PAD > 0 && lane == 0) raw[G::PAD - 1] = before;
            before_reg = before;
            if (NW == 4 && q0 + QUAD < t_end) PAA_F800_FETCH(q0 + QUAD)
        }
        wsync();
        if (lane == 0) before_next = raw[G::PAD + QUAD * S - 1];
        PAA_TICK(0)

        // ---------------- time domain: chunk partials (ShortTermFeatures.py:22-51)
        // Two samples per 32-bit lane operation: v_dot2 gives x0^2 + x1^2 and x0 + x1, the sign of x - mu comes from
        // packed 16-bit saturating arithmetic:  s = clamp(sat(x - floor(mu)), lo, 1) * a + c  with (lo, a, c) =
        // (-1, 1, 0) when mu is a whole number (sign 0 exists) and (0, 2, -1) otherwise (x - floor(mu) >= 1 <=> +1).
        for (int ch = lane; ch < NCHUNK; ch += 64) {
            const int4 *p4 = reinterpret_cast<const int4 *>(raw + G::PAD + CHUNK * ch);
            // the dword before the chunk holds the previous sample in its upper half (chunk 0 without a pad: lane 0's register)
            s16x2 sp;
            if (G::PAD == 0 && ch == 0) sp = (s16x2){0, before_reg};
            else sp = __builtin_bit_cast(s16x2, reinterpret_cast<const int *>(raw + G::PAD + CHUNK * ch)[-1]);
            sp = __builtin_elementwise_min(__builtin_elementwise_max(__builtin_elementwise_sub_sat(sp, zc_b), zc_lo), zc_two);
            double sx2 = 0.0;            // sum x^2 (exact: every term is an integer below 2^31, the sum below 2^53)
            int sx = 0;                  // sum x
            int zacc = 0, zfirst = 0;
#pragma unroll
            for (int v4 = 0; v4 < CHUNK / 8; ++v4) {
                const int4 q = p4[v4];
                const int w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    const s16x2 cur = __builtin_bit_cast(s16x2, w[h]);
                    const s16x2 sg = __builtin_elementwise_min(__builtin_elementwise_max(__builtin_elementwise_sub_sat(cur, zc_b), zc_lo), zc_two);
                    // (code of the sample before the pair, code of its first sample)
                    const unsigned sgw = __builtin_bit_cast(unsigned, sg);
                    const unsigned shw = __builtin_amdgcn_alignbit(sgw, __builtin_bit_cast(unsigned, sp), 16);
                    if (v4 == 0 && h == 0) zfirst = abs((int)(sgw & 0xffffu) - (int)(shw & 0xffffu));
                    asm("v_sad_u16 %0, %1, %2, %0" : "+v"(zacc) : "v"(sgw), "v"(shw));
                    sp = sg;
                    sx2 += (double)(unsigned)__builtin_amdgcn_sdot2(cur, cur, 0, false);     // 2^31 for (-32768, -32768)
                    sx = __builtin_amdgcn_sdot2(cur, zc_one, sx, false);
                }
            }
            // sum (x - m_int)^2 and sum (x - m_int) in exact integer arithmetic, then the residual mean as before:
            // sum y^2 over the chunk = (inv/2^15)^2 * sum (x' - delta)^2 with x' = x - m_int, |delta| <= 1/2
            const double mi = nm.mi;
            const double e2 = fma(mi, fma((double)CHUNK, mi, -2.0 * (double)sx), sx2);
            const int s1 = sx - CHUNK * m_int;
            const double e = y_scale2 * fma(delta_mu, fma(-2.0, (double)s1, nm.chunk_dmu), e2);
            cE[ch] = e;
            cZ[ch] = zacc;
            cF[ch] = zfirst;
        }

        PAA_TICK(1)
        // ---------------- pass 1: radix-25 on z[j + 16 r], z = x[2n] + i x[2n+1] (raw integers, exact in f64)
        double2 v[25];
        {
            const int *r32 = reinterpret_cast<const int *>(raw + G::PAD) + (S / 2) * g + i;
            int w25[25];
#pragma unroll
            for (int r = 0; r < 25; ++r) w25[r] = r32[16 * r];
            dft25_packed<(NW != 4)>(w25, m_int, v);
        }
        wsync();
        PAA_TICK(2)      // raw + chunk partials complete; the previous quad's readers of the slots are done

        // exchange through the quad's 4 spectrum slots: real plane, then imaginary plane.
        // element (frame g, index 25 j + q)
        double ax[16], ay[16], bx[16], by[16];
        double2 w2[16];       // W400^(r p): NW = 4 requests them before the exchange so they land while it runs
        if (NW == 4) {
#pragma unroll
            for (int r = 1; r < 16; ++r) w2[r] = t_tw2[r * TW_STRIDE + itw];
        }
        {
            double *pl = spec + ((slot0 + g) % 5) * NF;
#pragma unroll
            for (int q = 0; q < 25; ++q) pl[25 * i + q] = v[PAA_DFT25_POS(q)].x;
            wsync();
#pragma unroll
            for (int r = 0; r < 16; ++r) { ax[r] = pl[pa + 25 * r]; bx[r] = pl[pb + 25 * r]; }
            wsync();
#pragma unroll
            for (int q = 0; q < 25; ++q) pl[25 * i + q] = v[PAA_DFT25_POS(q)].y;
            wsync();
#pragma unroll
            for (int r = 0; r < 16; ++r) { ay[r] = pl[pa + 25 * r]; by[r] = pl[pb + 25 * r]; }
            wsync();
        }

        PAA_TICK(3)
        // ---------------- pass 2 + real-FFT recombination + magnitude (ShortTermFeatures.py:617-621)
        if (act) {
            // column 25-p uses the conjugate twiddles and a one-step rotation of the outputs
            double2 a[16], b[16];
            a[0] = make_double2(ax[0], ay[0]);
            b[0] = make_double2(bx[0], by[0]);
            double *sp = spec + ((slot0 + g) % 5) * NF;
            // Z[k], k = p + 25 q ; Z[400 - k] = column (25-p), output (15 - q) -> rotated index (16 - q) % 16.
            // Lane 0 (p = 0) has b == a bit for bit (same column, unit twiddles), so no select is needed:
            // Z[400 - 25 q] = A[(16 - q) % 16] = b[...] there as well.
            // 2E = Z[k] + conj Z[400-k],  2O = -i (Z[k] - conj Z[400-k])
#define PAA_F800_BIN(q, wpq)                                                                               \
            {                                                                                              \
                const double2 zk = a[PAA_DFT16_POS(q)];                                                    \
                const double2 zb = b[PAA_DFT16_POS((16 - (q)) % 16)];                                      \
                const int k = pa + 25 * (q);                                                               \
                const double2 e = make_double2(zk.x + zb.x, zk.y - zb.y);                                  \
                const double2 o = make_double2(zk.y + zb.y, zb.x - zk.x);                                  \
                const double2 t = cmul((wpq), o);                                                          \
                double xr = e.x + t.x, xi = e.y + t.y;                                                     \
                const double yr = e.x - t.x, yi = e.y - t.y;                                               \
                /* DC bin (q = 0 in lane 0): remove the residual clip mean; straight-line selects, no branch */ \
                const bool dc_ = ((q) == 0) && (i == 0);                                                   \
                xr = dc_ ? xr - dc_shift : xr;                                                             \
                xi = dc_ ? 0.0 : xi;                                                                       \
                const double mk_ = mag_sqrt(fma(xr, xr, xi * xi)) * mag_scale;                             \
                sp[k] = mk_;                                                                               \
                /* bin 400 - k; the DC lane has no partner bin and stores |X[0]| to bin 0 a second time */ \
                const double mm_ = mag_sqrt(fma(yr, yr, yi * yi)) * mag_scale;                             \
                if ((q) == 0) sp[dc_ ? 0 : NF - k] = dc_ ? mk_ : mm_;                                      \
                else sp[NF - k] = mm_;                                                                     \
            }
            if (NW == 4) {
#pragma unroll
                for (int r = 1; r < 16; ++r) {
                    a[r] = cmul(make_double2(ax[r], ay[r]), w2[r]);
                    b[r] = cmul(make_double2(bx[r], by[r]), make_double2(w2[r].x, -w2[r].y));
                }
                double2 wp[16];   // W800^(p + 25 q): in flight during the two radix-16 transforms
#pragma unroll
                for (int q = 0; q < 16; ++q) wp[q] = t_twp[q * TW_STRIDE + itw];
                dft16<0>(a);
                dft16<0>(b);
#pragma unroll
                for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(a[r].x), "+v"(a[r].y), "+v"(b[r].x), "+v"(b[r].y));
#pragma unroll
                for (int q = 0; q < 16; ++q) PAA_F800_BIN(q, wp[q])
            } else {
                // two waves per SIMD: registers matter more than latency (the partner wave hides it), so the table
                // values are fetched in small groups right before their use and the scheduler may not hoist them
#pragma unroll
                for (int r0 = 1; r0 < 16; r0 += 3) {
                    double2 wg[3];
#pragma unroll
                    for (int u = 0; u < 3; ++u) wg[u] = t_tw2[(r0 + u) * TW_STRIDE + itw];
#pragma unroll
                    for (int u = 0; u < 3; ++u) {
                        a[r0 + u] = cmul(make_double2(ax[r0 + u], ay[r0 + u]), wg[u]);
                        b[r0 + u] = cmul(make_double2(bx[r0 + u], by[r0 + u]), make_double2(wg[u].x, -wg[u].y));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                dft16<1>(a);
                dft16<1>(b);
                // pin the transforms here: without it the optimiser sinks their second stages into the bin loop below
                // and both input sets stay live next to the partial results
#pragma unroll
                for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(a[r].x), "+v"(a[r].y), "+v"(b[r].x), "+v"(b[r].y));
#pragma unroll
                for (int q0_ = 0; q0_ < 16; q0_ += 2) {
                    double2 wg[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) wg[u] = t_twp[(q0_ + u) * TW_STRIDE + itw];
#pragma unroll
                    for (int u = 0; u < 2; ++u) PAA_F800_BIN(q0_ + u, wg[u])
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#undef PAA_F800_BIN
        }
        wsync();
        ++n_done;
        PAA_F800_PACE(-1)

        PAA_TICK(4)
        // ---------------- features: 16 lanes per frame (group g <-> frame q0 + g)
        // lane i owns bins [25 i, 25 i + 25) of its frame; current and previous spectrum are pulled into
        // registers once (two batched LDS bursts) and serve the sums, the spread/flux pass and the roll-off scan
        const int t = q0 + g;
        const double *cur = spec + ((slot0 + g) % 5) * NF;
        const double *prv = (t == 0) ? cur : spec + ((slot0 + g + 4) % 5) * NF;
        double Xc[25], Xv[25];
#pragma unroll
        for (int m = 0; m < 25; ++m) { Xc[m] = cur[25 * i + m]; Xv[m] = prv[25 * i + m]; }
        if (G::ALIAS) wsync();       // the previous-spectrum slot becomes msp[] / fv[] below
        // sums over the lane's 25 bins; two interleaved accumulator sets keep the dependent chains short.
        // sum(ind * X) with ind = (k+1) f0 is f0 * [(25 i + 1) * sum X + sum m X]   (small exact integers)
        // X^2 is accumulated in five 5-bin chunks: their sum is the lane total (roll-off scan), and cut at the lane's
        // 40-bin block boundary they give the spectral-entropy block energies (:85-107) without a second sweep.
        double sXa = 0.0, sXb = 0.0, sMa = 0.0, sMb = 0.0, sVa = 0.0, sVb = 0.0, mx = 0.0;
        double c5[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int m = 0; m < 24; m += 2) {
            const double X0 = Xc[m], X1 = Xc[m + 1];
            sXa += X0; sXb += X1;
            sVa += Xv[m]; sVb += Xv[m + 1];
            sMa = fma((double)m, X0, sMa); sMb = fma((double)(m + 1), X1, sMb);
            c5[m / 5] = fma(X0, X0, c5[m / 5]); c5[(m + 1) / 5] = fma(X1, X1, c5[(m + 1) / 5]);
            mx = fmax(mx, fmax(X0, X1));
        }
        sXa += Xc[24]; sVa += Xv[24]; sMa = fma(24.0, Xc[24], sMa); c5[4] = fma(Xc[24], Xc[24], c5[4]); mx = fmax(mx, Xc[24]);
        const double cs = ((c5[0] + c5[1]) + (c5[2] + c5[3])) + c5[4];
        // Lane i holds bins [25 i, 25 i + 25); block b holds bins [40 b, 40 b + 40).  The first kcut chunks of the lane
        // lie in block floor(25 i / 40), the rest in the next one; the pattern repeats every 8 lanes (200 bins):
        // kcut = 5 3 5 1 4 5 2 5.  "Home" lanes (i & 7 in {0,2,4,5,7}) are the first lane of a block: block energy =
        // own lower part + upper part of the lane before + lower part of the lane after when that one is not a home.
        double pblk;
        {
            const int i7 = i & 7;
            const int kcut = (0x52541535u >> (4 * i7)) & 7;
            const double pL = c5[0] + ((kcut > 1) ? c5[1] : 0.0) + ((kcut > 2) ? c5[2] : 0.0) + ((kcut > 3) ? c5[3] : 0.0) +
                              ((kcut > 4) ? c5[4] : 0.0);
            const double pH = ((kcut > 1) ? 0.0 : c5[1]) + ((kcut > 2) ? 0.0 : c5[2]) + ((kcut > 3) ? 0.0 : c5[3]) +
                              ((kcut > 4) ? 0.0 : c5[4]);
            const double from_prev = dpp_mov<0x111>(pH);        // row_shr:1, lane 0 of the row receives 0
            const double from_next = dpp_mov<0x101>(pL);        // row_shl:1, lane 15 receives 0
            const bool next_joins = (i7 == 0) || (i7 == 2) || (i7 == 5);      // lane i + 1 is not a home
            pblk = pL + from_prev + (next_joins ? from_next : 0.0);
        }
        const bool home = ((0xB5u >> (i & 7)) & 1u) != 0;          // i & 7 in {0, 2, 4, 5, 7}
        const double base_k = (double)(25 * i + 1);
        double sX = sXa + sXb;
        double sIX = f0 * fma(base_k, sX, sMa + sMb);
        double sXp = sVa + sVb;
        sX = group_sum(sX); sXp = group_sum(sXp);
        sIX = group_sum(sIX); mx = group_max(mx);
        // np.sum(X + eps) (:118-119) = sum X + 400 eps up to rounding
        const double sXe = sX + (double)NF * kEps;
        sXp += (double)NF * kEps;
        const double run_incl = group_scan_incl(cs);
        const double sP = dpp_bcast15(run_incl);            // total = inclusive scan at lane 15

        // energy entropy: 80-sample block i = chunks CPF g + 2 i, + 1 (:34-51)
        const double eblk = (i < 10) ? cE[CPF * g + 2 * i] + cE[CPF * g + 2 * i + 1] : 0.0;
        const double e_tot = group_sum(eblk);
        double ent_f, ent_e;
        {
            const double sf = fast_div(pblk, sP + kEps), se = fast_div(eblk, e_tot + kEps);
            ent_f = group_sum(home ? -(sf * fast_log2(sf + kEps)) : 0.0);
            ent_e = group_sum((i < 10) ? -(se * fast_log2(se + kEps)) : 0.0);
        }
        // zero crossings: 20 chunks of the frame minus the pair that straddles the frame start (:22-26)
        int zc = cZ[CPF * g + i] + ((i < 4) ? cZ[CPF * g + 16 + i] : 0) - ((i == 0) ? cF[CPF * g] : 0);
        zc = group_sum_i(zc) << zc_shift;

        PAA_TICK(5)
        // centroid, spread, flux (:57-82, :110-124)
        const double r = (mx == 0.0) ? 1.0 / kEps : fast_div(1.0, mx);
        const double den = sX * r + kEps;
        const double rden = fast_div(1.0, den);
        const double cen = (sIX * r) * rden;
        const double rX = fast_div(1.0, sXe), rXp = fast_div(1.0, sXp);
        // spread: sum (ind - C)^2 X / max = f0^2/max * sum ((k+1) - C/f0)^2 X
        const double cb = base_k - cen * rf0;
        double sSa = 0.0, sSb = 0.0, sFa = 0.0, sFb = 0.0;
#pragma unroll
        for (int m = 0; m < 24; m += 2) {
            const double d0 = cb + (double)m, d1 = cb + (double)(m + 1);
            sSa = fma(d0 * d0, Xc[m], sSa);
            sSb = fma(d1 * d1, Xc[m + 1], sSb);
            const double f0d = Xc[m] * rX - Xv[m] * rXp, f1d = Xc[m + 1] * rX - Xv[m + 1] * rXp;
            sFa = fma(f0d, f0d, sFa);
            sFb = fma(f1d, f1d, sFb);
        }
        {
            const double d0 = cb + 24.0;
            sSa = fma(d0 * d0, Xc[24], sSa);
            const double f0d = Xc[24] * rX - Xv[24] * rXp;
            sFa = fma(f0d, f0d, sFa);
        }
        double sSp = (sSa + sSb) * (f0sq * r), sFl = sFa + sFb;
        sSp = group_sum(sSp);
        sFl = group_sum(sFl);
        const double spread = fast_sqrt(sSp * rden);

        // roll-off (:127-140): first k with cumsum(X^2)[k] + eps > 0.9 sum(X^2).  The running energy never decreases, so the
        // first bin that qualifies is the number of bins that do not (one compare and one add-with-carry per bin)
        int below = 0;
        {
            const double thr = 0.90 * sP;
            double run = run_incl - cs;
#pragma unroll
            for (int m = 0; m < 25; ++m) {
                run = fma(Xc[m], Xc[m], run);
                below += (run + kEps > thr) ? 0 : 1;
            }
        }
        const int first = group_min_i((below < 25) ? 25 * i + below : 0x7fffffff);

        PAA_TICK(6)
        // MFCC (:236-254): per-lane padded mel lists (host-built): class 0 = filter i, class 1 = filter 16+i,
        // class 2 = one half of filter 32 + (i & 7); the halves meet through a row rotation by 8
        double *mg = msp + 40 * g;
        {
            // a filter covers consecutive bins, so only its first bin is tabulated (k*[i]); weights are
            // zero-padded to the class length and the bin index is clamped into the spectrum
            double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0;
            const int lo0 = t_melk0[i], lo1 = t_melk1[i], lo2 = t_melk2[i];
            // each trip issues its 16 LDS loads back to back (one wait), then runs two 4-long FMA chains
#define PAA_MEL_CLASS(acc, lo, N, tw, IDX, UNROLL)                                                      \
            UNROLL for (int n = 0; n < (N); n += 8) {                                                   \
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
#define PAA_IDX_CLAMP(k) min((k), NF - 1)
#define PAA_IDX_PLAIN(k) (k)
            if (FIXED) {
                PAA_MEL_CLASS(acc0, lo0, 8, t_melw0, PAA_IDX_PLAIN, _Pragma("unroll"))
                PAA_MEL_CLASS(acc1, lo1, 16, t_melw1, PAA_IDX_PLAIN, _Pragma("unroll"))
                PAA_MEL_CLASS(acc2, lo2, 16, t_melw2, PAA_IDX_PLAIN, _Pragma("unroll"))
            } else if (L.mel_clamp) {
                PAA_MEL_CLASS(acc0, lo0, L.melN0, t_melw0, PAA_IDX_CLAMP, )
                PAA_MEL_CLASS(acc1, lo1, L.melN1, t_melw1, PAA_IDX_CLAMP, )
                PAA_MEL_CLASS(acc2, lo2, L.melN2, t_melw2, PAA_IDX_CLAMP, )
            } else {        // every padded list stays inside the 400 bins
                PAA_MEL_CLASS(acc0, lo0, L.melN0, t_melw0, PAA_IDX_PLAIN, )
                PAA_MEL_CLASS(acc1, lo1, L.melN1, t_melw1, PAA_IDX_PLAIN, )
                PAA_MEL_CLASS(acc2, lo2, L.melN2, t_melw2, PAA_IDX_PLAIN, )
            }
#undef PAA_IDX_CLAMP
#undef PAA_IDX_PLAIN
#undef PAA_MEL_CLASS
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
            const int ch_len = FIXED ? 8 : L.chN;
            for (int n = 0; n < ch_len; n += 8) {
                int kv[8];
                double wv[8], xv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { kv[u] = t_chk[(n + u) * CH_STRIDE + ich]; wv[u] = t_chw[(n + u) * CH_STRIDE + ich]; }
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
        double *fg = fv + FV_STRIDE * g;
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
        {   // population std of the 12 chroma values (:667), by DPP inside the group
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
            for (int s = 0; s < QUAD; ++s) vq[s] = fv[FV_STRIDE * s + lane];
            const bool last_quad = q0 + QUAD >= t_end;
            store_row_chunked(oc + (long long)lane * Tc, q0, r0, t_end, last_quad, hold, vq, P.debug);
            if (DELTAS) {
                const double dq[QUAD] = {(q0 == 0) ? 0.0 : vq[0] - vlast, vq[1] - vq[0], vq[2] - vq[1], vq[3] - vq[2]};
                store_row_chunked(oc + (long long)(kBase + lane) * Tc, q0, r0, t_end, last_quad, holdd, dq, P.debug);
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
    if (NW == 8 && lane == 0) pace[8 + wave] = 0x7fffffff;
    PAA_TEND()
}

}  // namespace f800

// returns 1 when a specialised kernel exists for this configuration (and fills fl), 0 when
// the generic kernel must be used, < 0 on error
inline int fast_select(int window, int step, int sample_kind, double fs, FastTables &ft, const FftPlan &fft,
                       const MelTable &mel, const ChromaTable &chroma, FastLaunch &fl, int want_waves) {
    if (!(window == 800 && (step == 400 || step == 800) && sample_kind == 0)) return 0;
    int nw = (want_waves == 8) ? 8 : 4;             // 8 waves per workgroup = two per SIMD
    int wave_bytes = (step == 400) ? (nw == 8 ? f800::Geo<400, 8>::WAVE_BYTES : f800::Geo<400, 4>::WAVE_BYTES)
                                   : (nw == 8 ? f800::Geo<800, 8>::WAVE_BYTES : f800::Geo<800, 4>::WAVE_BYTES);
    f800::TabLayout &L = fl.layout;
    auto up4 = [](int n) { return std::max(8, (n + 7) / 8 * 8); };      // lists are unrolled by 8 on the device
    int c0 = 0, c1 = 0, c2 = 0, cc = 0;
    for (int m = 0; m < 16; ++m) c0 = std::max(c0, (int)mel.cnt[m]);
    for (int m = 16; m < 32; ++m) c1 = std::max(c1, (int)mel.cnt[m]);
    for (int m = 32; m < 40; ++m) c2 = std::max(c2, ((int)mel.cnt[m] + 1) / 2);
    for (int c = 0; c < 12; ++c) cc = std::max(cc, (int)(chroma.class_start[c + 1] - chroma.class_start[c]));
    L.melN0 = up4(c0); L.melN1 = up4(c1); L.melN2 = up4(c2); L.chN = up4(cc);
    L.mel_clamp = 0;
    for (int i = 0; i < 16; ++i) {
        const int f2 = 32 + (i & 7), half = (mel.cnt[f2] + 1) / 2;
        const int lo2 = mel.lo[f2] + ((i < 8) ? 0 : half);
        if (mel.lo[i] + L.melN0 > f800::NF || mel.lo[16 + i] + L.melN1 > f800::NF || lo2 + L.melN2 > f800::NF) L.mel_clamp = 1;
    }
    L.fixed_lists = (L.melN0 == 8 && L.melN1 == 16 && L.melN2 == 16 && L.chN == 8 && !L.mel_clamp) ? 1 : 0;
    int off = 0;
    auto take = [&off](int bytes) { const int o = off; off += (bytes + 15) / 16 * 16; return o; };
    // k0/k1/k2 hold only the FIRST bin of each lane's filter (a filter covers consecutive bins)
    L.off_w0 = take(L.melN0 * 16 * 8); L.off_k0 = take(16 * 4);
    L.off_w1 = take(L.melN1 * 16 * 8); L.off_k1 = take(16 * 4);
    L.off_w2 = take(L.melN2 * 16 * 8); L.off_k2 = take(16 * 4);
    L.off_chw = take(L.chN * f800::CH_STRIDE * 8);  L.off_chk = take(L.chN * f800::CH_STRIDE * 4);
    L.off_dct = take(13 * 41 * 8);
    L.off_tw2 = take(16 * f800::TW_STRIDE * 16);
    L.off_twp = take(16 * f800::TW_STRIDE * 16);
    L.off_sync = take(16 * 4);
    L.total = off;
    {
        const char *pm = experiment_env("PAA_F800_PACE");          // A/B switch of the pacing (default 1: 0.314 ms; 0: 0.329 on cfg2)
        L.pace_mode = (pm && pm[0] == '0') ? 0 : 1;
    }
    L.f0 = fs / (2.0 * (double)f800::NF);
    L.rf0 = 1.0 / L.f0;
    L.r_half_fs = 1.0 / (fs / 2.0);
    L.f0sq = L.f0 * L.f0;
#ifdef PAA_EXPERIMENTS
    if (nw == 8 && (size_t)L.total + (size_t)nw * wave_bytes > 160 * 1024) {      // tables too large: one wave per SIMD
        nw = 4;
        wave_bytes = (step == 400) ? f800::Geo<400, 4>::WAVE_BYTES : f800::Geo<800, 4>::WAVE_BYTES;
    }
#else
    // (very low sampling rates: mel / chroma lists too long for eight waves -- the 2 RA RB family or the mixed-radix kernel
    // takes the shape; the one-wave-per-SIMD instances exist only in -DPAA_EXPERIMENTS builds)
    if (nw != 8) return 0;
#endif
    if ((size_t)L.total + (size_t)nw * wave_bytes > 160 * 1024) return 0;   // generic kernel instead
    if (!ft.d_blob) {
        std::vector<unsigned char> blob((size_t)L.total, 0);
        auto W = [&](int o) { return reinterpret_cast<double *>(blob.data() + o); };
        auto K = [&](int o) { return reinterpret_cast<int32_t *>(blob.data() + o); };
        for (int i = 0; i < 16; ++i) {
            const int f0 = i, f1 = 16 + i, f2 = 32 + (i & 7);
            K(L.off_k0)[i] = mel.lo[f0];
            K(L.off_k1)[i] = mel.lo[f1];
            for (int n = 0; n < mel.cnt[f0]; ++n) W(L.off_w0)[n * 16 + i] = mel.w[mel.off[f0] + n];
            for (int n = 0; n < mel.cnt[f1]; ++n) W(L.off_w1)[n * 16 + i] = mel.w[mel.off[f1] + n];
            const int half = (mel.cnt[f2] + 1) / 2;
            const int b = (i < 8) ? 0 : half, e = (i < 8) ? half : mel.cnt[f2];
            K(L.off_k2)[i] = mel.lo[f2] + b;
            for (int n = b; n < e; ++n) W(L.off_w2)[(n - b) * 16 + i] = mel.w[mel.off[f2] + n];
            if (i < 12)
                for (int n = chroma.class_start[i]; n < chroma.class_start[i + 1]; ++n) {
                    W(L.off_chw)[(n - chroma.class_start[i]) * f800::CH_STRIDE + i] = chroma.w[n];
                    K(L.off_chk)[(n - chroma.class_start[i]) * f800::CH_STRIDE + i] = chroma.src[n];
                }
        }
        double dct[kNumMfcc * kNumMel];
        build_dct(dct);
        for (int q = 0; q < 13; ++q)
            for (int n = 0; n < 40; ++n) W(L.off_dct)[q * 41 + n] = dct[q * 40 + n];
        for (int p = 0; p < f800::TW_STRIDE; ++p)
            for (int r = 0; r < 16; ++r) {
                const int m2 = r * p, mp = p + 25 * r;
                W(L.off_tw2)[2 * (r * f800::TW_STRIDE + p)] = fft.tw[2 * m2];
                W(L.off_tw2)[2 * (r * f800::TW_STRIDE + p) + 1] = fft.tw[2 * m2 + 1];
                W(L.off_twp)[2 * (r * f800::TW_STRIDE + p)] = fft.post[2 * mp];
                W(L.off_twp)[2 * (r * f800::TW_STRIDE + p) + 1] = fft.post[2 * mp + 1];
            }
        if (hipMalloc(&ft.d_blob, blob.size()) != hipSuccess) return PAA_ERR_OOM;
        if (hipMemcpy(ft.d_blob, blob.data(), blob.size(), hipMemcpyHostToDevice) != hipSuccess) return PAA_ERR_HIP;
    }
    fl.name = (step == 400) ? (nw == 8 ? "st_fast_800_w8" : "st_fast_800") : (nw == 8 ? "st_fast_800_s800_w8" : "st_fast_800_s800");
    fl.lds = (size_t)L.total + (size_t)nw * wave_bytes;
    fl.variant = (step == 400) ? 800 : 1600;
    fl.run = 256;       // longest run (frames) given to one wave; the plan shrinks it to fill the chip
    fl.waves_per_cu = nw;
    return 1;
}

#if !defined(PAA_NO_HOST_LAUNCHERS) || defined(PAA_LAUNCH_FAST)      // (kernels are instantiated only in family_fast*.hip)
template <int S, int DELTAS, int FIXED, int NW>
inline int fast_launch_one(const FastLaunch &fl, const PlanDev &P, const unsigned char *blob, const void *d_packed,
                           const ClipDev *clips, const ClipNorm *norms, const Tile *tiles, long long n_tiles,
                           double *d_out, hipStream_t stream) {
    static LdsAttrCache attr;
    if (!attr.covers(fl.lds)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&f800::st_fast_800_kernel<S, DELTAS, FIXED, NW>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)fl.lds) != hipSuccess) return -1;
        attr.set(fl.lds);
    }
    const unsigned grid = (unsigned)((n_tiles + NW - 1) / NW);
    hipLaunchKernelGGL((f800::st_fast_800_kernel<S, DELTAS, FIXED, NW>), dim3(grid), dim3(64 * NW), fl.lds, stream,
                       P, fl.layout, blob, (const int16_t *)d_packed, clips, norms, tiles, (int)n_tiles, d_out);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

template <int S, int NW>
inline int fast_launch_step(const FastLaunch &fl, const PlanDev &P, const unsigned char *blob, const void *d_packed,
                            const ClipDev *clips, const ClipNorm *norms, const Tile *tiles, long long n_tiles,
                            double *d_out, hipStream_t stream) {
    if (fl.layout.fixed_lists)
        return P.deltas ? fast_launch_one<S, 1, 1, NW>(fl, P, blob, d_packed, clips, norms, tiles, n_tiles, d_out, stream)
                        : fast_launch_one<S, 0, 1, NW>(fl, P, blob, d_packed, clips, norms, tiles, n_tiles, d_out, stream);
    return P.deltas ? fast_launch_one<S, 1, 0, NW>(fl, P, blob, d_packed, clips, norms, tiles, n_tiles, d_out, stream)
                    : fast_launch_one<S, 0, 0, NW>(fl, P, blob, d_packed, clips, norms, tiles, n_tiles, d_out, stream);
}

inline int fast_launch(const FastLaunch &fl, const PlanDev &P, const FastTables &ft, const void *d_packed,
                       const ClipDev *clips, const ClipNorm *norms, const Tile *tiles, long long n_tiles,
                       double *d_out, hipStream_t stream) {
    if (!ft.d_blob) return -1;
    const unsigned char *blob = reinterpret_cast<const unsigned char *>(ft.d_blob);
    if (fl.variant == 800 && fl.waves_per_cu == 8)
        return fast_launch_step<400, 8>(fl, P, blob, d_packed, clips, norms, tiles, n_tiles, d_out, stream);
    if (fl.variant == 1600 && fl.waves_per_cu == 8)
        return fast_launch_step<800, 8>(fl, P, blob, d_packed, clips, norms, tiles, n_tiles, d_out, stream);
#ifdef PAA_EXPERIMENTS      // the one-wave-per-SIMD instances (NW = 4, ~340 registers) are the A/B baseline of scripts/ab_waves.sh
    if (fl.variant == 800) return fast_launch_step<400, 4>(fl, P, blob, d_packed, clips, norms, tiles, n_tiles, d_out, stream);
    if (fl.variant == 1600) return fast_launch_step<800, 4>(fl, P, blob, d_packed, clips, norms, tiles, n_tiles, d_out, stream);
#endif
    return -1;
}

#endif  // PAA_NO_HOST_LAUNCHERS
}  // namespace paa
