// Shared device-side definitions: plan / clip descriptors, wave reductions, small DFTs.
// gfx950 only: wavefront = 64 lanes, every workgroup of the feature kernels is ONE wave, so
// __syncthreads() is an LDS/VMEM wait plus a one-wave barrier.
#pragma once
#include <float.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace paa {

constexpr int kWave = 64;
constexpr int kBase = 34;          // base feature rows (ShortTermFeatures.py:580-585)
constexpr int kFlush = 8;          // frames staged in LDS before a row-segment store
constexpr double kEps = 2.220446049250313e-16;   // sys.float_info.epsilon (ShortTermFeatures.py:11)
// ablation bits of PlanDev::debug (scripts/reg_ablate.py, scripts/write_traffic_ab.sh): compiled in only with
// -DPAA_EXPERIMENTS; the default build has no code that could skip a stage or drop a store
#ifdef PAA_EXPERIMENTS
#define PAA_DEBUG_BIT(word, bit) (((word) & (bit)) != 0)
#else
#define PAA_DEBUG_BIT(word, bit) false
#endif

struct ClipDev {        // one per clip, built on the host
    long long sample_off;   // first sample in the packed buffer
    long long n;            // samples
    long long out_off;      // start of the [F][T] slab in the output (doubles)
    int T;                  // frames
    int stat_first;         // first statistics chunk of this clip
    int stat_count;
    int pad;
};

struct ClipNorm {       // written by clip_params_kernel
    double mean;        // mean(x / 2^15)
    double inv;         // 1 / (max|x/2^15 - mean| + 1e-10)
    // derived constants of the integer-input kernels (wave-uniform there: fetched with scalar loads).
    // The FFT runs on integers x - m_int (m_int = the clip mean rounded to a whole count; exact in f64).
    // y = (x/2^15 - mean) * inv is affine, so every bin scales by inv/2^15 and only the DC bin sees the
    // residual mean:  Y[0] = inv/2^15 * (X'[0] - W * (mu - m_int)).  Removing m_int first keeps the DC
    // component (and its round-off leakage into the other bins) below half a count per sample.
    double mu;          // clip mean in counts = mean * 2^15 (exact scaling)
    double delta_mu;    // mu - m_int, |.| <= 1/2
    double inv_sc;      // inv / 2^15: y = (x' - delta_mu) * inv_sc with x' = x - m_int
    double y_scale2;    // inv_sc^2
    double mi;          // (double)m_int
    double mag_scale;   // inv_sc * 0.5 / Nf: magnitude of a bin from the packed real-FFT recombination (E, O carry 1/2)
    double dc_shift;    // 2 W delta_mu: what the DC bin of the integer FFT carries too much
    double chunk_dmu;   // 40 delta_mu (time-domain partials are formed over 40-sample chunks)
    int m_int;          // nearbyint(mu)
    int zb;             // floor(mu) clamped into int16: sign(x/2^15 - mean) = sign(x - mu) in 16-bit arithmetic
    int mu_whole;       // 1: mu is a whole number (a sample can sit exactly on the mean: sign 0)
    int pad;
};

struct Tile {           // a run of consecutive frames of one clip = one workgroup
    int clip;
    int t0;
    int cnt;
    int pad;
};

namespace wg {
// one frame of a launch of the big-window kernels (kernels_wg.hpp, kernels_wgs.hpp): its clip, its index in the clip, the row of the
// spectrum scratch it writes, and whether it is only there to provide the previous spectrum of the next one (a chunk that starts
// inside a clip); task lists of the split transforms keep the sub-transform / task type in bits 8.. of `halo`
struct FrameRef {
    int clip, t, row, halo;
};
}  // namespace wg

struct StatChunk {      // a span of samples of one clip = one workgroup of clip_stats_kernel
    long long start;    // absolute sample index in the packed buffer
    int len;
    int clip;
};

struct PlanDev {
    int W, S, Nf, Nc, even;
    int n_pass;
    int radix[24];
    const double2 *tw;      // Nc
    const double2 *post;    // Nc (even only)
    const int *mel_lo, *mel_cnt, *mel_off;
    const double *mel_w;
    const double *dct;      // 13 x 40
    const int *ch_start;    // 13
    const int *ch_src;
    const double *ch_w;
    double fs;
    int deltas;
    int F;                  // 34 or 68
    int blk_t;              // floor(W / 10)
    int blk_f;              // floor(Nf / 10)
    int mode;               // 0 features, 1 spectrogram, 2 chromagram
    int frame_origin;       // first frame starts at this sample (0; W for spectrogram/chromagram)
    int debug;              // PAA_KERNEL_DEBUG bit mask (ablation experiments only; 0 in production)
    // the statistics partials of clip_stats_*: the kernels of the main shapes fold them into ClipNorm themselves, one wave
    // at a time in their prologue (norms_inline = 1), instead of waiting for clip_params_kernel -- 6 us and a kernel
    // boundary per step for a one-hour clip
    const void *st_sum, *st_min, *st_max;
    double st_scale;        // sample_scale of the plan's sample type
    int norms_inline;
    int pad_;
};

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// inclusive prefix sum across the wave
__device__ __forceinline__ double wave_scan_incl(double v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        double u = __shfl_up(v, o, 64);
        if (lane >= o) v += u;
    }
    return v;
}

__device__ __forceinline__ double2 cmul(double2 a, double2 b) {
    return make_double2(fma(a.x, b.x, -a.y * b.y), fma(a.x, b.y, a.y * b.x));
}
__device__ __forceinline__ double2 cadd(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ double2 csub(double2 a, double2 b) { return make_double2(a.x - b.x, a.y - b.y); }
// a - i*b and a + i*b
__device__ __forceinline__ double2 sub_i(double2 a, double2 b) { return make_double2(a.x + b.y, a.y - b.x); }
__device__ __forceinline__ double2 add_i(double2 a, double2 b) { return make_double2(a.x - b.y, a.y + b.x); }

__device__ __forceinline__ void dft2(double2 *v) {
    double2 a = v[0], b = v[1];
    v[0] = cadd(a, b);
    v[1] = csub(a, b);
}
__device__ __forceinline__ void dft3(double2 *v) {
    const double h = 0.86602540378443864676;   // sin(pi/3)
    double2 t = cadd(v[1], v[2]);
    double2 m = make_double2(v[0].x - 0.5 * t.x, v[0].y - 0.5 * t.y);
    double2 n = make_double2(h * (v[1].x - v[2].x), h * (v[1].y - v[2].y));
    v[0] = cadd(v[0], t);
    v[1] = sub_i(m, n);
    v[2] = add_i(m, n);
}
__device__ __forceinline__ void dft4(double2 *v) {
    double2 t0 = cadd(v[0], v[2]), t1 = csub(v[0], v[2]);
    double2 t2 = cadd(v[1], v[3]), t3 = csub(v[1], v[3]);
    v[0] = cadd(t0, t2);
    v[2] = csub(t0, t2);
    v[1] = sub_i(t1, t3);
    v[3] = add_i(t1, t3);
}
// radix 5 in the form that is EXACTLY zero in its four non-DC outputs for five equal inputs: cos 144 = -1/2 - cos 72, so
// X[1] = (a0 - t2/2) + c1 (t1 - t2) and X[2] = (a0 - t1/2) - c1 (t1 - t2) (the textbook form leaves a0 * 1e-16).  Frames of
// digital silence are constant once the clip mean is removed; the reference's FFT (pocketfft: plain sums and differences
// first) returns exact zeros for their non-DC bins, and log10(E + eps) of an empty mel band resolves 1e-24.
__device__ __forceinline__ void dft5(double2 *v) {
    const double c1 = 0.30901699437494742410;
    const double s1 = 0.95105651629515357212, s2 = 0.58778525229247312917;
    const double2 t1 = cadd(v[1], v[4]), t2 = cadd(v[2], v[3]), t3 = csub(v[1], v[4]), t4 = csub(v[2], v[3]);
    const double2 d = csub(t1, t2);
    const double2 h1 = make_double2(fma(-0.5, t1.x, v[0].x), fma(-0.5, t1.y, v[0].y));
    const double2 h2 = make_double2(fma(-0.5, t2.x, v[0].x), fma(-0.5, t2.y, v[0].y));
    const double2 m1 = make_double2(fma(c1, d.x, h2.x), fma(c1, d.y, h2.y));
    const double2 m2 = make_double2(fma(-c1, d.x, h1.x), fma(-c1, d.y, h1.y));
    const double2 n1 = make_double2(fma(s2, t4.x, s1 * t3.x), fma(s2, t4.y, s1 * t3.y));
    const double2 n2 = make_double2(fma(-s1, t4.x, s2 * t3.x), fma(-s1, t4.y, s2 * t3.y));
    v[0] = make_double2(v[0].x + t1.x + t2.x, v[0].y + t1.y + t2.y);
    v[1] = sub_i(m1, n1);
    v[4] = add_i(m1, n1);
    v[2] = sub_i(m2, n2);
    v[3] = add_i(m2, n2);
}

// ---- 16-lane group reductions on DPP (no LDS traffic; every lane ends with the same bits) ----------
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ int dpp_mov_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }
// quad_perm[1,0,3,2], quad_perm[2,3,0,1], row_half_mirror, row_mirror
#define PAA_DPP_X1 0xB1
#define PAA_DPP_X2 0x4E
#define PAA_DPP_HM 0x141
#define PAA_DPP_RM 0x140
__device__ __forceinline__ double group_sum(double v) {
    v += dpp_mov<PAA_DPP_X1>(v);
    v += dpp_mov<PAA_DPP_X2>(v);
    v += dpp_mov<PAA_DPP_HM>(v);
    v += dpp_mov<PAA_DPP_RM>(v);
    return v;
}
__device__ __forceinline__ double group_max(double v) {
    v = fmax(v, dpp_mov<PAA_DPP_X1>(v));
    v = fmax(v, dpp_mov<PAA_DPP_X2>(v));
    v = fmax(v, dpp_mov<PAA_DPP_HM>(v));
    v = fmax(v, dpp_mov<PAA_DPP_RM>(v));
    return v;
}
__device__ __forceinline__ int group_sum_i(int v) {
    v += dpp_mov_i<PAA_DPP_X1>(v);
    v += dpp_mov_i<PAA_DPP_X2>(v);
    v += dpp_mov_i<PAA_DPP_HM>(v);
    v += dpp_mov_i<PAA_DPP_RM>(v);
    return v;
}
__device__ __forceinline__ int group_min_i(int v) {
    v = min(v, dpp_mov_i<PAA_DPP_X1>(v));
    v = min(v, dpp_mov_i<PAA_DPP_X2>(v));
    v = min(v, dpp_mov_i<PAA_DPP_HM>(v));
    v = min(v, dpp_mov_i<PAA_DPP_RM>(v));
    return v;
}
// inclusive prefix sum over the 16-lane row: row_shr:n shifts zeros in (bound_ctrl)
__device__ __forceinline__ double group_scan_incl(double v) {
    v += dpp_mov<0x111>(v);
    v += dpp_mov<0x112>(v);
    v += dpp_mov<0x114>(v);
    v += dpp_mov<0x118>(v);
    return v;
}

// wave-local ordering point for LDS hand-offs between lanes of ONE wave: a wave's DS instructions execute
// in program order, so only the compiler has to be kept from moving memory operations across
__device__ __forceinline__ void wsync() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}


// ---- full-wave (64-lane) reductions on DPP: row reduction, then row_bcast15 / row_bcast31 carry the row
// totals forward; lane 63 ends with the total, which v_readlane broadcasts (identical bits in every lane)
__device__ __forceinline__ double readlane63(double v) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_mov_rows(double v) {      // masked-off rows receive 0
    int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROWMASK, 0xF, false);
    int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROWMASK, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wsum(double v) {
    v = group_sum(v);
    v += dpp_mov_rows<0x142, 0xA>(v);
    v += dpp_mov_rows<0x143, 0xC>(v);
    return readlane63(v);
}
__device__ __forceinline__ double wmax_nonneg(double v) {      // operands >= 0 (masked rows contribute 0)
    v = group_max(v);
    v = fmax(v, dpp_mov_rows<0x142, 0xA>(v));
    v = fmax(v, dpp_mov_rows<0x143, 0xC>(v));
    return readlane63(v);
}
__device__ __forceinline__ int wsum_i(int v) {
    v = group_sum_i(v);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int wmin_i(int v) {                 // via max of negated non-positive... keep simple: xor tree
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
    return v;
}
// inclusive prefix sum over the 64 lanes
__device__ __forceinline__ double wscan_incl(double v) {
    v = group_scan_incl(v);
    v += dpp_mov_rows<0x142, 0xA>(v);
    v += dpp_mov_rows<0x143, 0xC>(v);
    return v;
}

// ---- FP64 helpers of the feature stages: hardware seeds (v_rsq_f64 / v_rcp_f64) + Newton steps instead of libm's
// general-purpose sequences (which also handle sub-normals, infinities and the IEEE division corner cases)
// sqrt for x >= 0 to ~1 ulp: v_rsq_f64 seed + two coupled Newton steps (ocml's version adds scaling for
// sub-normal / huge arguments, which |X|^2 of a normalised frame never reaches)
__device__ __forceinline__ double fast_sqrt(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    double h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    const double d = fma(-g, g, x);
    g = fma(d, h, g);
    return (x > 0.0) ? g : 0.0;
}
// magnitude variant: one coupled Newton step (relative error ~ (rsq error)^2, far below the 1e-4 parity bound
// even for a 2^-20 seed); measured against the two-step version in tests/test_parity_gpu.py tolerances
__device__ __forceinline__ double mag_sqrt(double x) {
    // x is either exactly 0 or far above 1e-300 (squares of sums of integers and their round-off), so clamping the
    // seed's argument replaces the x > 0 select: 0 * rsq(1e-300) = 0 goes through the Newton step unchanged
    // (the clamp is an unsigned max on the high dword -- x >= 0 -- which costs half an FP64 issue slot)
    const unsigned hi_ = max((unsigned)__double2hiint(x), 0x01a56e1fu);         // high dword of 1e-300
    const double y = __builtin_amdgcn_rsq(__hiloint2double((int)hi_, __double2loint(x)));
    const double g = x * y;
    const double h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    return fma(g, r, g);
}

// a / b for finite b != 0 to ~1 ulp: v_rcp_f64 seed, two Newton steps, one residual correction (the IEEE
// division sequence -- div_scale / div_fmas / div_fixup -- is a ~150-cycle dependent chain per quotient)
__device__ __forceinline__ double fast_div(double a, double b) {
    double r = __builtin_amdgcn_rcp(b);
    r = fma(fma(-b, r, 1.0), r, r);
    r = fma(fma(-b, r, 1.0), r, r);
    const double q = a * r;
    return fma(fma(-b, q, a), r, q);
}

// log2(x) for finite x > 0 (callers add eps = 2^-52 first): x = m 2^e with m in [sqrt(1/2), sqrt(2)),
// ln m = 2 atanh(s), s = (m-1)/(m+1), |s| <= 0.1716, odd series to s^21 (truncation < 1e-18 relative).
// About 35 FP64 operations instead of ~115 in the generic libm path; error a few 1e-16 relative.
__device__ __forceinline__ double fast_log2(double x) {
    double m = __builtin_amdgcn_frexp_mant(x);          // [0.5, 1)
    int e = __builtin_amdgcn_frexp_exp(x);
    const bool lo = m < 0.70710678118654752440;
    m = lo ? m + m : m;
    e = lo ? e - 1 : e;
    const double num = m - 1.0, den = m + 1.0;
    double r = __builtin_amdgcn_rcp(den);
    r = fma(fma(-den, r, 1.0), r, r);
    r = fma(fma(-den, r, 1.0), r, r);
    double s = num * r;
    s = fma(fma(-den, s, num), r, s);
    const double z = s * s;
    double p = 1.0 / 21.0;
    p = fma(p, z, 1.0 / 19.0);
    p = fma(p, z, 1.0 / 17.0);
    p = fma(p, z, 1.0 / 15.0);
    p = fma(p, z, 1.0 / 13.0);
    p = fma(p, z, 1.0 / 11.0);
    p = fma(p, z, 1.0 / 9.0);
    p = fma(p, z, 1.0 / 7.0);
    p = fma(p, z, 1.0 / 5.0);
    p = fma(p, z, 1.0 / 3.0);
    p = fma(p, z, 1.0);
    // log2 m = (2 / ln 2) s p
    return fma(s * p, 2.8853900817779268147, (double)e);
}
__device__ __forceinline__ double fast_log10(double x) { return fast_log2(x) * 0.30102999566398119521; }

template <typename T> __device__ __forceinline__ double load_sample(const T *p);
template <> __device__ __forceinline__ double load_sample<int16_t>(const int16_t *p) { return (double)(*p); }
template <> __device__ __forceinline__ double load_sample<double>(const double *p) { return *p; }
// stereo16 = one interleaved stereo frame (L, R) of int16 PCM, fetched as one 32-bit word and summed in the load:
// stereo_to_mono (audioBasicIO.py:156-168) is mono = L/2 + R/2 exactly, so x / 2^15 = (L + R) / 2^16 -- the mono signal
// never exists in memory
struct alignas(4) stereo16 { int16_t l, r; };
__device__ __forceinline__ int stereo_word_sum(int w) { return (int)(short)(w & 0xffff) + (w >> 16); }
template <> __device__ __forceinline__ double load_sample<stereo16>(const stereo16 *p) {
    return (double)stereo_word_sum(*reinterpret_cast<const int *>(p));
}
template <typename T> __host__ __device__ constexpr double sample_scale() { return 1.0 / 32768.0; }
template <> __host__ __device__ constexpr double sample_scale<stereo16>() { return 1.0 / 65536.0; }

// Sign codes: only sums of |s_n - s_{n-1}| are ever used, so a sample's sign is kept as a small non-negative code whose
// differences are the sign differences (up to the factor `sh` applied once to the wave total).
// Integer samples (int16 PCM, or L + R of a stereo frame): sign(x sc - mean) = sign(x - mean / sc), decided in integer
// arithmetic: code = clamp(x - (zb - 1), lo, 2) with zb = floor(mean / sc); lo = 0 when mean / sc is a whole number (a sample
// can sit on the mean: codes 0 / 1 / 2 = signs -1 / 0 / +1), lo = 1 otherwise (codes 1 / 2, differences count double: sh = 1)
// -- two instructions per sample
struct SignRule {
    int zb1, lo, sh;       // wave-uniform
};
__device__ __forceinline__ int sgn1(int x, const SignRule &q) {
    int t;
    asm("v_med3_i32 %0, %1, %2, 2" : "=v"(t) : "v"(x - q.zb1), "v"(q.lo));
    return t;
}
__device__ __forceinline__ int sgn1(double d) { return ((d > 0.0) ? 2 : 1) - ((d < 0.0) ? 1 : 0); }      // float64 samples: sign + 1
// acc += |a - b| for a, b >= 0
__device__ __forceinline__ void sad_acc(int &acc, int a, int b) { asm("v_sad_u32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b)); }
// the rule of a clip for integer sample type T (mean = the clip mean of x sc; exact: sc is a power of two)
template <typename T>
__device__ __forceinline__ SignRule sign_rule(double mean) {
    const double thr = mean * (1.0 / sample_scale<T>());
    const double fl = floor(thr);
    const bool whole = (fl == thr);
    SignRule q;
    q.zb1 = (int)fl - 1;
    q.lo = whole ? 0 : 1;
    q.sh = __builtin_amdgcn_readfirstlane(whole ? 0 : 1);
    asm volatile("" : "+v"(q.zb1), "+v"(q.lo));        // (opaque: x - zb1 stays ONE subtraction)
    return q;
}

// ---- clip constants from the statistics partials of one clip (ShortTermFeatures.py:14-19, :567-570): the partials are
// folded lane-strided, then over a fixed xor tree (deterministic: every wave of every kernel gets the same bits); every lane
// returns the same ClipNorm.  SumT / MmT: long long / int for the integer sample types (exact), double / double for float64.
template <typename SumT, typename MmT>
__device__ __forceinline__ ClipNorm clip_norm_wave(const ClipDev &cd, const SumT *__restrict__ psum,
                                                   const MmT *__restrict__ pmin, const MmT *__restrict__ pmax, double sc,
                                                   int window, int lane) {
    SumT s = 0;
    double mn = DBL_MAX, mx = -DBL_MAX;
    const SumT *ps = psum + cd.stat_first;
    const MmT *pn = pmin + cd.stat_first, *px = pmax + cd.stat_first;
    int i = lane;
    for (; i + 192 < cd.stat_count; i += 256) {          // four loads of each array in flight (same summation order)
        SumT a[4];
        MmT b[4], c[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { a[u] = ps[i + 64 * u]; b[u] = pn[i + 64 * u]; c[u] = px[i + 64 * u]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) { s += a[u]; mn = fmin(mn, (double)b[u]); mx = fmax(mx, (double)c[u]); }
    }
    for (; i < cd.stat_count; i += 64) {
        s += ps[i];
        mn = fmin(mn, (double)pn[i]);
        mx = fmax(mx, (double)px[i]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_xor(s, o, 64);
        mn = fmin(mn, __shfl_xor(mn, o, 64));
        mx = fmax(mx, __shfl_xor(mx, o, 64));
    }
    ClipNorm nm;
    if (cd.n <= 0) { nm.mean = 0.0; nm.inv = 1.0; }
    else {
        nm.mean = ((double)s * sc) / (double)cd.n;
        const double peak = fmax(fabs(fma(mx, sc, -nm.mean)), fabs(fma(mn, sc, -nm.mean)));
        nm.inv = 1.0 / (peak + 1e-10);
    }
    nm.mu = nm.mean * 32768.0;
    nm.m_int = (int)fmin(fmax(nearbyint(nm.mu), -40000.0), 40000.0);
    nm.delta_mu = nm.mu - (double)nm.m_int;
    nm.inv_sc = nm.inv * (1.0 / 32768.0);
    nm.y_scale2 = nm.inv_sc * nm.inv_sc;
    nm.mi = (double)nm.m_int;
    nm.mag_scale = nm.inv_sc * (0.5 / (double)(window / 2));
    nm.dc_shift = 2.0 * (double)window * nm.delta_mu;
    nm.chunk_dmu = 40.0 * nm.delta_mu;
    const double mu_fl = floor(nm.mu);
    nm.zb = (int)fmin(fmax(mu_fl, -32768.0), 32767.0);
    nm.mu_whole = (mu_fl == nm.mu) ? 1 : 0;
    nm.pad = 0;
    return nm;
}
// wave-uniform values computed with vector instructions, pinned into scalar registers
__device__ __forceinline__ double uni_f64(double v) {
    const unsigned long long b = __double_as_longlong(v);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}
// the clip constants of a wave's clip: read (norms_inline = 0) or formed from the partials; wave-uniform either way
template <typename T>
__device__ __forceinline__ ClipNorm wave_clip_norm(const PlanDev &P, const ClipDev &cd, const ClipNorm *__restrict__ norms,
                                                   int clip, int lane) {
    if (!P.norms_inline) return norms[clip];
    ClipNorm nm;
    if constexpr (sizeof(T) == 8)
        nm = clip_norm_wave<double, double>(cd, (const double *)P.st_sum, (const double *)P.st_min, (const double *)P.st_max,
                                            P.st_scale, P.W, lane);
    else
        nm = clip_norm_wave<long long, int>(cd, (const long long *)P.st_sum, (const int *)P.st_min, (const int *)P.st_max,
                                            P.st_scale, P.W, lane);
    nm.mean = uni_f64(nm.mean); nm.inv = uni_f64(nm.inv); nm.mu = uni_f64(nm.mu); nm.delta_mu = uni_f64(nm.delta_mu);
    nm.inv_sc = uni_f64(nm.inv_sc); nm.y_scale2 = uni_f64(nm.y_scale2); nm.mi = uni_f64(nm.mi);
    nm.mag_scale = uni_f64(nm.mag_scale); nm.dc_shift = uni_f64(nm.dc_shift); nm.chunk_dmu = uni_f64(nm.chunk_dmu);
    nm.m_int = __builtin_amdgcn_readfirstlane(nm.m_int); nm.zb = __builtin_amdgcn_readfirstlane(nm.zb);
    nm.mu_whole = __builtin_amdgcn_readfirstlane(nm.mu_whole);
    return nm;
}

}  // namespace paa
