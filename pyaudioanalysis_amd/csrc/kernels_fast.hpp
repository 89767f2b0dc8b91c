// Specialised feature kernel for the headline configuration: int16 PCM, window 800, step 400
// (50 ms / 25 ms at 16 kHz -- BASELINE configs 1-4).  Any sampling rate (tables are per fs).
//
// One workgroup = ONE wave = one run of consecutive frames of one clip, processed FOUR frames
// ("a quad") per iteration:
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
//   store   : lane = feature row, 4 consecutive frames (32 B) per row
// Halo: a run with t0 > 0 first processes the quad t0-4..t0-1 without storing it.
//
// Replaces ShortTermFeatures.py:608-682 (+ helpers :22-140, :236-321) for this configuration.
#pragma once
#include "device_common.hpp"
#include "tables.hpp"

namespace paa {

struct FastTables {
    void *d_blob = nullptr;
};
struct FastLaunch {
    int run = 0;
    size_t lds = 0;
    const char *name = "";
    int variant = 0;
};

inline void fast_tables_free(FastTables &t) {
    if (t.d_blob) (void)hipFree(t.d_blob);
    t.d_blob = nullptr;
}

namespace f800 {

constexpr int W = 800, S = 400, NF = 400, QUAD = 4;
constexpr int RAW_N = (QUAD - 1) * S + W;          // 2000 samples per quad
constexpr int RAW_PAD = 8;                          // raw[RAW_PAD + i]; raw[RAW_PAD - 1] = sample before
constexpr int CHUNK = 40, NCHUNK = RAW_N / CHUNK;   // 50 chunks of 40 samples
constexpr int FV_STRIDE = 34;

// LDS carve (bytes)
constexpr int OFF_SPEC = 0;                                   // 5 slots x 400 doubles
constexpr int OFF_RAW = OFF_SPEC + 5 * NF * 8;                // int16 raw[2016]; aliased by msp[4][40] later
constexpr int OFF_CE = OFF_RAW + (RAW_N + 2 * RAW_PAD) * 2;   // double cE[50]
constexpr int OFF_CZ = OFF_CE + NCHUNK * 8;                   // int cZ[50], cF[50]
constexpr int OFF_FV = OFF_CZ + 2 * NCHUNK * 4;               // double fv[4][34]
constexpr int LDS_BYTES = OFF_FV + QUAD * FV_STRIDE * 8;
static_assert(OFF_RAW % 16 == 0 && OFF_CE % 8 == 0 && OFF_FV % 8 == 0, "LDS alignment");
static_assert(QUAD * 40 * 8 <= (RAW_N + 2 * RAW_PAD) * 2, "msp alias fits in the raw buffer");

// ---- register DFTs ------------------------------------------------------------------------
__device__ __forceinline__ void dft5r(double2 &a0, double2 &a1, double2 &a2, double2 &a3, double2 &a4) {
    const double c1 = 0.30901699437494742410, c2 = -0.80901699437494742410;
    const double s1 = 0.95105651629515357212, s2 = 0.58778525229247312917;
    const double2 t1 = cadd(a1, a4), t2 = cadd(a2, a3), t3 = csub(a1, a4), t4 = csub(a2, a3);
    const double2 m1 = make_double2(fma(c2, t2.x, fma(c1, t1.x, a0.x)), fma(c2, t2.y, fma(c1, t1.y, a0.y)));
    const double2 m2 = make_double2(fma(c1, t2.x, fma(c2, t1.x, a0.x)), fma(c1, t2.y, fma(c2, t1.y, a0.y)));
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
__device__ __forceinline__ void dft25(double2 *v) {
#pragma unroll
    for (int r2 = 0; r2 < 5; ++r2) dft5r(v[r2], v[5 + r2], v[10 + r2], v[15 + r2], v[20 + r2]);
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
#pragma unroll
    for (int q1 = 0; q1 < 5; ++q1) dft5r(v[5 * q1], v[5 * q1 + 1], v[5 * q1 + 2], v[5 * q1 + 3], v[5 * q1 + 4]);
}
#define PAA_DFT25_POS(q) (5 * ((q) % 5) + (q) / 5)

// v[r], r = 4 r1 + r2  ->  result for output q stored at v[4 (q % 4) + q / 4]
__device__ __forceinline__ void dft16(double2 *v) {
    const double c = 0.92387953251128676, s = 0.38268343236508977, h = 0.70710678118654752;
#pragma unroll
    for (int r2 = 0; r2 < 4; ++r2) dft4r(v[r2], v[4 + r2], v[8 + r2], v[12 + r2]);
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
#pragma unroll
    for (int q1 = 0; q1 < 4; ++q1) dft4r(v[4 * q1], v[4 * q1 + 1], v[4 * q1 + 2], v[4 * q1 + 3]);
}
#define PAA_DFT16_POS(q) (4 * ((q) % 4) + (q) / 4)

__device__ __forceinline__ double group_sum(double v) {      // over a 16-lane group
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double group_max(double v) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int group_sum_i(int v) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ int group_min_i(int v) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
    return v;
}

#ifndef PAA_F800_WAVES_PER_SIMD
#define PAA_F800_WAVES_PER_SIMD 2
#endif
template <int DELTAS>
__global__ __launch_bounds__(64, PAA_F800_WAVES_PER_SIMD) void st_fast_800_kernel(PlanDev P, const int16_t *__restrict__ sig,
                                                             const ClipDev *__restrict__ clips,
                                                             const ClipNorm *__restrict__ norms,
                                                             const Tile *__restrict__ tiles,
                                                             double *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *spec = reinterpret_cast<double *>(smem + OFF_SPEC);
    int16_t *raw = reinterpret_cast<int16_t *>(smem + OFF_RAW);
    double *msp = reinterpret_cast<double *>(smem + OFF_RAW);          // alias: raw is dead by then
    double *cE = reinterpret_cast<double *>(smem + OFF_CE);
    int *cZ = reinterpret_cast<int *>(smem + OFF_CZ);
    int *cF = cZ + NCHUNK;
    double *fv = reinterpret_cast<double *>(smem + OFF_FV);

    const int lane = threadIdx.x;
    const int g = lane >> 4, i = lane & 15;
    const Tile tl = tiles[blockIdx.x];
    const ClipDev c = clips[tl.clip];
    const ClipNorm nm = norms[tl.clip];
    const int16_t *xc = sig + c.sample_off;
    const long long Tc = c.T;
    double *oc = out + c.out_off;
    constexpr int F = DELTAS ? 68 : 34;

    const double sc = 1.0 / 32768.0;
    const double f0 = P.fs / (2.0 * (double)NF);
    const double half_fs = P.fs / 2.0;
    // integer sign thresholds: sign(x/2^15 - mean) = sign(x - mu), mu = mean * 2^15 (exact)
    const double mu = nm.mean * 32768.0;
    const int thr_pos = (int)fmin(fmax(floor(mu) + 1.0, -40000.0), 40000.0);   // x >= thr_pos  <=> positive
    const int thr_neg = (int)fmin(fmax(ceil(mu) - 1.0, -40000.0), 40000.0);    // x <= thr_neg  <=> negative

    const int t_end = tl.t0 + tl.cnt;
    int q0 = tl.t0 >= QUAD ? tl.t0 - QUAD : 0;
    int slot0 = 1;                   // slots of this quad: slot0 .. slot0+3 (mod 5); previous = slot0-1
    double vlast = 0.0;              // lane l < 34: feature l of the frame before this quad
    for (; q0 < t_end; q0 += QUAD, slot0 = (slot0 + 4) % 5) {
        // ---------------- stage raw samples [q0*S - 1, q0*S + 2000)
        {
            const long long base = (long long)q0 * S;
            const long long avail = c.n - base;        // samples of the clip from `base`
            const int16_t *src = xc + base;
            const bool aligned = ((reinterpret_cast<uintptr_t>(src) & 15) == 0);
            if (aligned && avail >= RAW_N) {
                const int4 *s4 = reinterpret_cast<const int4 *>(src);
                int4 *d4 = reinterpret_cast<int4 *>(raw + RAW_PAD);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int idx = lane + 64 * r;
                    if (idx < RAW_N / 8) d4[idx] = s4[idx];
                }
            } else {
                for (int n = lane; n < RAW_N; n += 64) raw[RAW_PAD + n] = (n < avail) ? src[n] : (int16_t)0;
            }
            if (lane == 0) raw[RAW_PAD - 1] = (base > 0) ? src[-1] : src[0];
        }
        __syncthreads();

        // ---------------- time domain: chunk partials (ShortTermFeatures.py:22-51)
        if (lane < NCHUNK) {
            const int4 *p4 = reinterpret_cast<const int4 *>(raw + RAW_PAD + CHUNK * lane);
            int prev = raw[RAW_PAD + CHUNK * lane - 1];
            int sprev = (prev >= thr_pos) - (prev <= thr_neg);
            double e = 0.0;
            int z = 0, zfirst = 0;
#pragma unroll
            for (int v4 = 0; v4 < CHUNK / 8; ++v4) {
                const int4 q = p4[v4];
                const int w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    const int xa = (int)(short)(w[h] & 0xffff), xb = w[h] >> 16;
                    const double ya = fma((double)xa, sc, -nm.mean) * nm.inv;
                    const double yb = fma((double)xb, sc, -nm.mean) * nm.inv;
                    e = fma(ya, ya, e);
                    e = fma(yb, yb, e);
                    const int sa = (xa >= thr_pos) - (xa <= thr_neg);
                    const int sb = (xb >= thr_pos) - (xb <= thr_neg);
                    const int da = abs(sa - sprev);
                    if (v4 == 0 && h == 0) zfirst = da;
                    z += da + abs(sb - sa);
                    sprev = sb;
                }
            }
            cE[lane] = e;
            cZ[lane] = z;
            cF[lane] = zfirst;
        }

        // ---------------- pass 1: radix-25 on z[j + 16 r]  (lane = frame g, column j = i)
        double2 v[25];
        {
            const int *r32 = reinterpret_cast<const int *>(raw + RAW_PAD) + (S / 2) * g + i;
#pragma unroll
            for (int r = 0; r < 25; ++r) {
                const int w = r32[16 * r];
                const int xa = (int)(short)(w & 0xffff), xb = w >> 16;
                v[r] = make_double2(fma((double)xa, sc, -nm.mean) * nm.inv, fma((double)xb, sc, -nm.mean) * nm.inv);
            }
        }
        dft25(v);
        __syncthreads();      // raw + chunk partials complete; previous quad's readers of the slots are done

        // exchange planes live in the 4 current slots (contiguous modulo the 5-slot ring)
        // element (frame g, index 25 j + q)
        double2 a[16], b[16];
        const int pa = i, pb = (i == 0) ? 0 : 25 - i;
        const bool act = i < 13;
        {
            double *pl = spec + ((slot0 + g) % 5) * NF;
#pragma unroll
            for (int q = 0; q < 25; ++q) pl[25 * i + q] = v[PAA_DFT25_POS(q)].x;
            __syncthreads();
            if (act) {
#pragma unroll
                for (int r = 0; r < 16; ++r) { a[r].x = pl[pa + 25 * r]; b[r].x = pl[pb + 25 * r]; }
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 25; ++q) pl[25 * i + q] = v[PAA_DFT25_POS(q)].y;
            __syncthreads();
            if (act) {
#pragma unroll
                for (int r = 0; r < 16; ++r) { a[r].y = pl[pa + 25 * r]; b[r].y = pl[pb + 25 * r]; }
            }
            __syncthreads();
        }

        // ---------------- pass 2 + real-FFT recombination + magnitude (ShortTermFeatures.py:617-621)
        if (act) {
            // twiddles W400^(r p); column 25-p uses the conjugates and a one-step output rotation
#pragma unroll
            for (int r = 1; r < 16; ++r) {
                const double2 w = P.tw[r * pa];
                a[r] = cmul(a[r], w);
                b[r] = cmul(b[r], make_double2(w.x, -w.y));
            }
            dft16(a);
            dft16(b);
            double *sp = spec + ((slot0 + g) % 5) * NF;
            const double invNf = 1.0 / (double)NF;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                // Z[k], k = p + 25 q ; Z[400 - k] = column (25-p), output (15 - q) -> rotated index (16 - q) % 16
                const double2 zk = a[PAA_DFT16_POS(q)];
                const int qm = (16 - q) % 16;
                const double2 zb = (i == 0) ? a[PAA_DFT16_POS(qm)] : b[PAA_DFT16_POS(qm)];
                const int k = pa + 25 * q;
                const double2 e = make_double2(0.5 * (zk.x + zb.x), 0.5 * (zk.y - zb.y));
                const double2 d = make_double2(0.5 * (zk.x - zb.x), 0.5 * (zk.y + zb.y));
                const double2 o = make_double2(d.y, -d.x);
                const double2 t = cmul(P.post[k], o);
                const double xr = e.x + t.x, xi = e.y + t.y;
                const double yr = e.x - t.x, yi = e.y - t.y;
                sp[k] = sqrt(fma(xr, xr, xi * xi)) * invNf;
                if (k > 0) sp[NF - k] = sqrt(fma(yr, yr, yi * yi)) * invNf;
            }
        }
        __syncthreads();

        // ---------------- features: 16 lanes per frame (group g <-> frame q0 + g)
        const int t = q0 + g;
        const double *cur = spec + ((slot0 + g) % 5) * NF;
        const double *prv = (t == 0) ? cur : spec + ((slot0 + g + 4) % 5) * NF;
        double sX = 0.0, sXe = 0.0, sXp = 0.0, sIX = 0.0, mx = 0.0, sP = 0.0;
#pragma unroll 5
        for (int m = 0; m < 25; ++m) {
            const int k = i + 16 * m;
            const double X = cur[k];
            sX += X;
            sXe += X + kEps;
            sXp += prv[k] + kEps;
            sIX = fma((double)(k + 1) * f0, X, sIX);
            mx = fmax(mx, X);
            sP = fma(X, X, sP);
        }
        sX = group_sum(sX); sXe = group_sum(sXe); sXp = group_sum(sXp);
        sIX = group_sum(sIX); sP = group_sum(sP); mx = group_max(mx);

        // spectral entropy: lane i < 10 sums block i of 40 bins (:85-107)
        double pblk = 0.0;
        if (i < 10) {
            const double2 *c2 = reinterpret_cast<const double2 *>(cur + 40 * i);
#pragma unroll 5
            for (int m = 0; m < 20; ++m) { const double2 x2 = c2[m]; pblk = fma(x2.x, x2.x, pblk); pblk = fma(x2.y, x2.y, pblk); }
        }
        // energy entropy: 80-sample block i = chunks 10 g + 2 i, + 1 (:34-51)
        const double eblk = (i < 10) ? cE[10 * g + 2 * i] + cE[10 * g + 2 * i + 1] : 0.0;
        const double e_tot = group_sum(eblk);
        double ent_f, ent_e;
        {
            const double sf = pblk / (sP + kEps), se = eblk / (e_tot + kEps);
            ent_f = group_sum((i < 10) ? -(sf * log2(sf + kEps)) : 0.0);
            ent_e = group_sum((i < 10) ? -(se * log2(se + kEps)) : 0.0);
        }
        // zero crossings: 20 chunks of the frame minus the pair that straddles the frame start (:22-26)
        int zc = cZ[10 * g + i] + ((i < 4) ? cZ[10 * g + 16 + i] : 0) - ((i == 0) ? cF[10 * g] : 0);
        zc = group_sum_i(zc);

        // centroid, spread, flux (:57-82, :110-124)
        const double r = (mx == 0.0) ? 1.0 / kEps : 1.0 / mx;
        const double den = sX * r + kEps;
        const double cen = (sIX * r) / den;
        const double rX = 1.0 / sXe, rXp = 1.0 / sXp;
        double sSp = 0.0, sFl = 0.0;
#pragma unroll 5
        for (int m = 0; m < 25; ++m) {
            const int k = i + 16 * m;
            const double X = cur[k];
            const double dv = (double)(k + 1) * f0 - cen;
            sSp = fma(dv * dv, X * r, sSp);
            const double df = __dmul_rn(X, rX) - __dmul_rn(prv[k], rXp);
            sFl = fma(df, df, sFl);
        }
        sSp = group_sum(sSp);
        sFl = group_sum(sFl);
        const double spread = sqrt(sSp / den);

        // roll-off (:127-140): lane i scans bins [25 i, 25 i + 25)
        int first = 0x7fffffff;
        {
            const double thr = 0.90 * sP;
            double cs = 0.0;
#pragma unroll 5
            for (int m = 0; m < 25; ++m) { const double X = cur[25 * i + m]; cs = fma(X, X, cs); }
            double run = cs;
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) { const double u = __shfl_up(run, o, 16); if (i >= o) run += u; }
            run -= cs;
            for (int m = 0; m < 25; ++m) {
                const double X = cur[25 * i + m];
                run = fma(X, X, run);
                if (run + kEps > thr) { first = 25 * i + m; break; }
            }
            first = group_min_i(first);
        }

        // MFCC (:236-254): filters i, i+16, i+32
        double *mg = msp + 40 * g;
#pragma unroll
        for (int fi = 0; fi < 3; ++fi) {
            const int m = i + 16 * fi;
            if (m < 40) {
                const int lo = P.mel_lo[m], cnt = P.mel_cnt[m];
                const double *w = P.mel_w + P.mel_off[m];
                double acc = 0.0;
                for (int n = 0; n < cnt; ++n) acc = fma(cur[lo + n], w[n], acc);
                mg[m] = log10(acc + kEps);
            }
        }
        // chroma (:277-321): lane i < 12 = pitch class i
        double chroma = 0.0;
        if (i < 12) {
            const int b0 = P.ch_start[i], b1 = P.ch_start[i + 1];
            for (int n = b0; n < b1; ++n) { const double x = cur[P.ch_src[n]]; chroma += (x * x) * P.ch_w[n]; }
            chroma = (sP == 0.0) ? chroma / kEps : chroma / sP;
        }
        __syncthreads();
        double *fg = fv + FV_STRIDE * g;
        if (i < 13) {
            const double *dm = P.dct + 40 * i;
            double acc = 0.0;
#pragma unroll 8
            for (int n = 0; n < 40; ++n) acc = fma(dm[n], mg[n], acc);
            fg[8 + i] = acc;
        }
        if (i < 12) fg[21 + i] = chroma;
        if (i == 15) {
            fg[0] = ((double)zc / 2.0) / (double)(W - 1);
            fg[1] = e_tot / (double)W;
            fg[2] = ent_e;
            fg[3] = cen / half_fs;
            fg[4] = spread / half_fs;
            fg[5] = ent_f;
            fg[6] = (t == 0) ? 0.0 : sFl;      // first frame: previous spectrum = itself (:624-625)
            fg[7] = (first == 0x7fffffff) ? 0.0 : (double)first / (double)NF;
        }
        {   // population std of the 12 chroma values (:667), by shuffles inside the group
            double m = group_sum((i < 12) ? chroma : 0.0) / 12.0;
            const double d = (i < 12) ? chroma - m : 0.0;
            const double var = group_sum(d * d) / 12.0;
            if (i == 14) fg[33] = sqrt(var);
        }
        __syncthreads();

        // ---------------- store: lane = feature row, 4 consecutive frames
        if (lane < kBase) {
            double vq[QUAD];
#pragma unroll
            for (int s = 0; s < QUAD; ++s) vq[s] = fv[FV_STRIDE * s + lane];
#pragma unroll
            for (int s = 0; s < QUAD; ++s) {
                const int ts = q0 + s;
                if (ts >= tl.t0 && ts < t_end) {
                    oc[(long long)lane * Tc + ts] = vq[s];
                    if (DELTAS) {
                        const double pv = (s == 0) ? vlast : vq[s - 1];
                        oc[(long long)(kBase + lane) * Tc + ts] = (ts == 0) ? 0.0 : vq[s] - pv;
                    }
                }
            }
            vlast = vq[QUAD - 1];
        }
        (void)F;
    }
}

}  // namespace f800

// returns 1 when a specialised kernel exists for this configuration (and fills fl), 0 when
// the generic kernel must be used, < 0 on error
inline int fast_select(int window, int step, int sample_kind, double fs, FastTables &ft, const FftPlan &fft,
                       FastLaunch &fl) {
    (void)fs; (void)ft; (void)fft;
    if (window == 800 && step == 400 && sample_kind == 0) {
        fl.name = "st_fast_800";
        fl.lds = f800::LDS_BYTES;
        fl.variant = 800;
        fl.run = 128;       // frames per run; the plan shrinks it to fill the chip
        return 1;
    }
    return 0;
}

inline int fast_launch(const FastLaunch &fl, const PlanDev &P, const FastTables &ft, const void *d_packed,
                       const ClipDev *clips, const ClipNorm *norms, const Tile *tiles, long long n_tiles,
                       double *d_out, hipStream_t stream) {
    (void)ft;
    if (fl.variant != 800) return -1;
    if (P.deltas)
        hipLaunchKernelGGL(f800::st_fast_800_kernel<1>, dim3((unsigned)n_tiles), dim3(64), fl.lds, stream, P,
                           (const int16_t *)d_packed, clips, norms, tiles, d_out);
    else
        hipLaunchKernelGGL(f800::st_fast_800_kernel<0>, dim3((unsigned)n_tiles), dim3(64), fl.lds, stream, P,
                           (const int16_t *)d_packed, clips, norms, tiles, d_out);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace paa
