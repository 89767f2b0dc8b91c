// Bluestein (chirp-z) feature kernel: every window whose FFT length has a prime factor above 13 and that no register-FFT
// shape covers -- 0.030 x 22050 = 661 (prime), 1103, 46 ms at 16 kHz = 736 = 2^5 x 23 ... (the reference takes any
// int(window), ShortTermFeatures.py:563-564, :617).  Until round 6 these ran O(N p) Stockham passes in kernels_generic.hpp
// (1103 / 441: 6.6e5 frames/s, 800 x slower than the headline).
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
// passes (16 / 8 / 4 codelets in registers), decimation in frequency on the way in, decimation in time on the way back, so
// that NO permutation is ever applied: FFT(b) is stored in the forward transform's digit-reversed order.
//
//   load     : W samples -> y (normalised) as doubles at the front of the buffer; a frame whose samples are all equal (digital
//              silence) takes a shortcut: spectrum [W |y0| / Nf, 0, 0, ...] exactly, as the reference's pocketfft gives for a
//              constant frame (the other kernels get this from exact-zero codelets; a chirp convolution cannot)
//   time     : zero crossings, energy, energy entropy from y (kernels_mix.hpp's contiguous chunks)
//   pass 0   : DIF radix R0 over the whole sequence, straight from y: element n = y[n] conj(c[n]) for n < W, 0 beyond
//   pass 1   : DIF radix R1 inside the R0 blocks
//   pass 2   : the innermost radix-R2 butterflies are transformed forward, multiplied by FFT(b) / M, conjugated and transformed
//              again in registers -- forward pass 2, product and the first pass back are ONE LDS round trip
//              (the way back is a forward transform of the conjugate: |conj z| = |z|, one set of twiddles)
//   pass 1'  : DIT radix R1 (input twiddles)
//   pass 0'  : DIT radix R0; only the outputs k < Nf are formed (the others are dead code in the codelet), |.| / Nf goes to
//              the frame's spectrum -- held in registers until every lane has read its operands (the spectrum overlaps the buffer)
//   features : kernels_mix.hpp's spectral stage (run-time Nf), rows staged [kFlush][F] and stored as row segments
//
// Replaces the while loop at ShortTermFeatures.py:608-682 (+ helpers :22-140, :236-321) and the loops of spectrogram
// (:415-422) / chromagram (:349-359) for those windows.
#pragma once
#include <vector>

#include "kernels_mix.hpp"

namespace paa {
namespace blu {

template <int LOG2M> struct Sched;
template <> struct Sched<8> { static constexpr int R0 = 4, R1 = 8, R2 = 8, NW = 16; };
template <> struct Sched<9> { static constexpr int R0 = 8, R1 = 8, R2 = 8, NW = 12; };
template <> struct Sched<10> { static constexpr int R0 = 16, R1 = 8, R2 = 8, NW = 8; };
template <> struct Sched<11> { static constexpr int R0 = 16, R1 = 16, R2 = 8, NW = 4; };
template <> struct Sched<12> { static constexpr int R0 = 16, R1 = 16, R2 = 16, NW = 2; };

template <int LOG2M_>
struct Shape {
    static constexpr int LOG2M = LOG2M_, M = 1 << LOG2M_;
    static constexpr int R0 = Sched<LOG2M_>::R0, R1 = Sched<LOG2M_>::R1, R2 = Sched<LOG2M_>::R2;
    static constexpr int NW = Sched<LOG2M_>::NW;               // most waves per workgroup (sets the register budget)
    static constexpr int S0 = M / R0;                          // pass 0: span M, element stride S0, S0 butterflies
    static constexpr int SP1 = S0, S1 = SP1 / R1;              // pass 1: span S0, stride S1
    static constexpr int TW0 = 0, TW1 = (R0 - 1) * S0, NTW = TW1 + (R1 - 1) * S1;      // twiddle tables [q - 1][k] of pass 0 / 1
    // outputs of the last pass that can be bins: k + q S0 < Nf <= (M + 1) / 3
    static constexpr int QMAX = (R0 == 16) ? 6 : (R0 == 8 ? 3 : 2);
    static_assert(R0 * R1 * R2 == M && S1 == R2, "three passes");
    static_assert(S0 % 64 == 0, "pass 0: every lane has the same number of butterflies");
};

struct BluLayout {
    int off_mello, off_melcnt, off_meloff, off_melw, off_dct, off_chstart, off_chsrc, off_chw;
    int table_bytes;     // LDS part of the blob, multiple of 256
    int wave_bytes;      // per-wave region, multiple of 256
    int waves;
    int unit_bytes;      // U: one spectrum (Nf doubles) rounded to 256 bytes
    int buf_bytes;       // B: 16 M
    int log2m;
    int off_g_chirp;     // global part: double2 [W]: conj(c[n])
    int off_g_bp;        // double2 [M]: FFT(b) / M at the positions the DIF passes leave the bins
    int off_g_tw;        // double2 [NTW]: pass tables
    int total_bytes;
};

__device__ __forceinline__ int sw(int e) { return e ^ ((e >> 4) & 15); }

// ---- pass 0 forward, fused with the chirp: y (doubles at the front of the buffer) -> buf
template <typename SH>
__device__ __forceinline__ void fwd_pass0(double2 *buf, const double2 *__restrict__ g_chirp, const double2 *__restrict__ g_tw,
                                          int W, int lane) {
    constexpr int R = SH::R0, S = SH::S0, NB = S / 64;
    const double *st = reinterpret_cast<const double *>(buf);
    double y[NB][R];
#pragma unroll
    for (int u = 0; u < NB; ++u)
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int n = lane + 64 * u + r * S;
            y[u][r] = (n < W) ? st[min(n, W - 1)] : 0.0;
        }
    wsync();           // every lane has its samples: the buffer may be overwritten
    const double2 *tw = g_tw + SH::TW0;
#pragma unroll
    for (int u = 0; u < NB; ++u) {
        const int k = lane + 64 * u;
        double2 v[R], w[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            // rows at or beyond W are zeros for every lane: no chirp loads (r S >= W is wave-uniform)
            if (r * S < W) {
                const double2 cw = g_chirp[min(k + r * S, W - 1)];
                v[r] = make_double2(y[u][r] * cw.x, y[u][r] * cw.y);
            } else {
                v[r] = make_double2(0.0, 0.0);
            }
        }
#pragma unroll
        for (int q = 1; q < R; ++q) w[q] = tw[(q - 1) * S + k];
        mix::Bfly<R>::run(v);
#pragma unroll
        for (int q = 1; q < R; ++q) v[mix::Bfly<R>::pos(q)] = cmul(v[mix::Bfly<R>::pos(q)], w[q]);
#pragma unroll
        for (int q = 0; q < R; ++q) buf[sw(k + q * S)] = v[mix::Bfly<R>::pos(q)];
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
template <typename SH, bool FWD>
__device__ __forceinline__ void pass1(double2 *buf, const double2 *__restrict__ g_tw, int lane) {
    constexpr int R = SH::R1, SPAN = SH::SP1, S = SH::S1, NBT = SH::M / R, NB = (NBT + 63) / 64;
    constexpr int U = (NB >= 2 && R <= 8) ? 2 : 1;          // butterflies in flight per lane (radix 16: 62 registers of operands each)
    const double2 *tw = g_tw + SH::TW1;
#pragma unroll
    for (int u0 = 0; u0 < NB; u0 += U) {
        double2 v[U][R], w[U][R];
        int base[U], k[U];
        bool act[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int b = lane + 64 * (u0 + u);
            act[u] = (NBT % 64 == 0) || b < NBT;
            locate<R, SPAN>(act[u] ? b : NBT - 1, base[u], k[u]);
#pragma unroll
            for (int r = 0; r < R; ++r) v[u][r] = buf[sw(base[u] + r * S)];
#pragma unroll
            for (int q = 1; q < R; ++q) w[u][q] = tw[(q - 1) * S + k[u]];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!FWD) {
#pragma unroll
                for (int r = 1; r < R; ++r) v[u][r] = cmul(v[u][r], w[u][r]);
            }
            mix::Bfly<R>::run(v[u]);
            if (FWD) {
#pragma unroll
                for (int q = 1; q < R; ++q) v[u][mix::Bfly<R>::pos(q)] = cmul(v[u][mix::Bfly<R>::pos(q)], w[u][q]);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (act[u]) {
#pragma unroll
                for (int q = 0; q < R; ++q) buf[sw(base[u] + q * S)] = v[u][mix::Bfly<R>::pos(q)];
            }
    }
    wsync();
}

// ---- pass 2 forward + product with FFT(b) / M + conjugate + pass 2 back: the R2 elements of a butterfly are contiguous
template <typename SH>
__device__ __forceinline__ void pass2_product(double2 *buf, const double2 *__restrict__ g_bp, int lane) {
    constexpr int R = SH::R2, NBT = SH::M / R, NB = (NBT + 63) / 64;
    constexpr int U = (NB >= 2 && R <= 8) ? 2 : 1;
#pragma unroll
    for (int u0 = 0; u0 < NB; u0 += U) {
        double2 v[U][R], bp[U][R];
        int base[U];
        bool act[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int b = lane + 64 * (u0 + u);
            act[u] = (NBT % 64 == 0) || b < NBT;
            base[u] = (act[u] ? b : NBT - 1) * R;
#pragma unroll
            for (int r = 0; r < R; ++r) v[u][r] = buf[sw(base[u] + r)];
#pragma unroll
            for (int q = 0; q < R; ++q) bp[u][q] = g_bp[base[u] + q];
        }
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
                for (int q = 0; q < R; ++q) buf[sw(base[u] + q)] = v[u][q];
            }
    }
    wsync();
}

// ---- pass 0 back (DIT, input twiddles) + |.| / Nf (ShortTermFeatures.py:617-621): bins k + q S0 < Nf only
template <typename SH>
__device__ __forceinline__ void back_pass0_magnitudes(const double2 *buf, double *cur, const double2 *__restrict__ g_tw, int Nf,
                                                      int lane) {
    constexpr int R = SH::R0, S = SH::S0, NB = S / 64, QM = SH::QMAX;
    const double2 *tw = g_tw + SH::TW0;
    const double invNf = 1.0 / (double)Nf;
    double res[NB][QM];
#pragma unroll
    for (int u = 0; u < NB; ++u) {
        const int k = lane + 64 * u;
        double2 v[R], w[R];
#pragma unroll
        for (int r = 0; r < R; ++r) v[r] = buf[sw(k + r * S)];
#pragma unroll
        for (int r = 1; r < R; ++r) w[r] = tw[(r - 1) * S + k];
#pragma unroll
        for (int r = 1; r < R; ++r) v[r] = cmul(v[r], w[r]);
        mix::Bfly<R>::run(v);
#pragma unroll
        for (int q = 0; q < QM; ++q) {
            const double2 z = v[mix::Bfly<R>::pos(q)];
            res[u][q] = mag_sqrt(fma(z.x, z.x, z.y * z.y)) * invNf;
        }
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

// load + normalise one frame as doubles at st[0 .. W); returns (wave-uniform) whether all samples are equal, y0 = sample 0
template <typename T>
__device__ __forceinline__ bool frame_load_real(const PlanDev &P, const T *__restrict__ x, const ClipNorm &nm, double *st, int lane,
                                                double &y0) {
    const double sc = sample_scale<T>();
    const int W = P.W;
    double first = 0.0;
    bool differs = false;
    int n = lane;
    {
        // (lane 0's first sample is sample 0; every lane compares against it after the broadcast below)
        const double q0 = load_sample<T>(x);
        first = fma(q0, sc, -nm.mean) * nm.inv;
    }
    for (; n + 3 * kWave < W; n += 4 * kWave) {
        double q[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) q[u] = load_sample<T>(x + n + kWave * u);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const double yv = fma(q[u], sc, -nm.mean) * nm.inv;
            differs |= (yv != first);
            st[n + kWave * u] = yv;
        }
    }
    for (; n < W; n += kWave) {
        const double yv = fma(load_sample<T>(x + n), sc, -nm.mean) * nm.inv;
        differs |= (yv != first);
        st[n] = yv;
    }
    y0 = first;
    wsync();
    return __ballot(differs) == 0ull;
}

// zcr count, energy and energy entropy of the normalised frame (contiguous doubles): kernels_mix.hpp's time_features_chunked
// on the plain layout (ShortTermFeatures.py:22-51)
__device__ __forceinline__ TimeFeat time_features_real(const double *y, const mix::Chunk &ch, int lane) {
    double ea = 0.0, eb = 0.0;
    int zc = 0;
    auto sgn = [](double x) {
        const int hi = __double2hiint(x), lo = __double2loint(x);
        return (((hi & 0x7fffffff) | lo) != 0) ? ((hi >> 31) | 1) : 0;
    };
    int sprev = sgn(y[max(ch.kb - 1, 0)]);                 // (lane 0: sample 0 against itself counts nothing)
    auto one = [&](int n, double x) {
        const double sq = x * x;
        const double sa = (n < ch.bound) ? sq : 0.0;
        ea += sa;
        eb += sq - sa;
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
    double eblk[10], e_tail;
    mix::block_sums(ch, ea, eb, eblk, e_tail);
    TimeFeat tf;
    tf.e_tot = e_tail;
#pragma unroll
    for (int j = 0; j < 10; ++j) tf.e_tot += eblk[j];
    tf.zc = wsum_i(zc);
    double num = 0.0;
#pragma unroll
    for (int j = 0; j < 10; ++j)
        if (lane == j) num = eblk[j];
    const double s = fast_div(num, tf.e_tot + kEps);
    tf.ent_e = wsum((lane < 10) ? -(s * fast_log2(s + kEps)) : 0.0);
    return tf;
}

template <typename T, int LOG2M>
__global__ __launch_bounds__(64 * Sched<LOG2M>::NW) void st_blu_kernel(PlanDev P, BluLayout L, const unsigned char *__restrict__ blob,
                                                                       const T *__restrict__ sig, const ClipDev *__restrict__ clips,
                                                                       const ClipNorm *__restrict__ norms,
                                                                       const Tile *__restrict__ tiles, int n_tiles,
                                                                       double *__restrict__ out) {
    typedef Shape<LOG2M> SH;
    extern __shared__ __attribute__((aligned(256))) unsigned char smem[];
    {
        const int4 *src4 = reinterpret_cast<const int4 *>(blob);
        int4 *dst4 = reinterpret_cast<int4 *>(smem);
        for (int n = threadIdx.x; n < L.table_bytes / 16; n += blockDim.x) dst4[n] = src4[n];
    }
    __syncthreads();       // the only workgroup-wide barrier
    Tabs tb;
    tb.tw = nullptr; tb.post = nullptr;
    tb.mel_lo = reinterpret_cast<const int *>(smem + L.off_mello);
    tb.mel_cnt = reinterpret_cast<const int *>(smem + L.off_melcnt);
    tb.mel_off = reinterpret_cast<const int *>(smem + L.off_meloff);
    tb.mel_w = reinterpret_cast<const double *>(smem + L.off_melw);
    tb.dct = reinterpret_cast<const double *>(smem + L.off_dct);
    tb.dct_stride = 41;
    tb.ch_start = reinterpret_cast<const int *>(smem + L.off_chstart);
    tb.ch_src = reinterpret_cast<const int *>(smem + L.off_chsrc);
    tb.ch_w = reinterpret_cast<const double *>(smem + L.off_chw);
    const double2 *g_chirp = reinterpret_cast<const double2 *>(blob + L.off_g_chirp);
    const double2 *g_bp = reinterpret_cast<const double2 *>(blob + L.off_g_bp);
    const double2 *g_tw = reinterpret_cast<const double2 *>(blob + L.off_g_tw);

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int tile_id = blockIdx.x * L.waves + wave;
    if (tile_id >= n_tiles) return;
    const int Nf = P.Nf, W = P.W, F = P.F > 0 ? P.F : 1;
    unsigned char *wb = smem + L.table_bytes + wave * L.wave_bytes;
    double *otile = reinterpret_cast<double *>(wb + L.buf_bytes + L.unit_bytes);
    double *fv = otile + kFlush * F;
    double *msp = fv + 48;

    const Tile tl = tiles[tile_id];
    const ClipDev c = clips[tl.clip];
    const ClipNorm nm = wave_clip_norm<T>(P, c, norms, tl.clip, lane);
    const T *x0 = sig + c.sample_off + P.frame_origin;
    const long long Tc = c.T;
    double *oc = out + c.out_off;

    const int hneed = (P.mode == 0) ? (P.deltas ? 2 : 1) : 0;
    const int h = min(hneed, tl.t0);
    double vprev = 0.0;
    int nslot = 0, tbase = tl.t0, odd = 0;
    const int tend = tl.t0 + tl.cnt;
    const mix::Chunk ch_t = mix::make_chunk(W, P.blk_t, lane), ch_f = mix::make_chunk(Nf, P.blk_f, lane);
    for (int t = tl.t0 - h; t < tend; ++t, odd ^= 1) {
        // even frames of the run: transform at the front, spectrum at the very front, previous spectrum behind the buffer;
        // odd frames: transform one unit further, spectrum in its last unit, previous spectrum at the very front
        double2 *buf = reinterpret_cast<double2 *>(wb + (odd ? L.unit_bytes : 0));
        double *cur = reinterpret_cast<double *>(wb + (odd ? L.buf_bytes : 0));
        double *prv = reinterpret_cast<double *>(wb + (odd ? 0 : L.buf_bytes));
        double *st = reinterpret_cast<double *>(buf);
        const T *x = x0 + (long long)t * P.S;
        double y0;
        const bool silent = frame_load_real<T>(P, x, nm, st, lane, y0);
        const bool want = (P.mode == 0) && ((t >= tl.t0) || (P.deltas && t == tl.t0 - 1));
        TimeFeat tf;
        tf.e_tot = 0.0; tf.ent_e = 0.0; tf.zc = 0;
        if (want) tf = time_features_real(st, ch_t, lane);
        int touched = 0;
        if (t + 1 < tend) {
            // pull the next frame's new samples towards the L2 / L1: their latency runs under this frame's transform
            const char *nb = reinterpret_cast<const char *>(x + W);
            const int nbytes = P.S * (int)sizeof(T);
            for (int o = 64 * lane; o + 4 <= nbytes; o += 64 * kWave) touched ^= *reinterpret_cast<const int *>(nb + o);
        }
        wsync();
        if (silent) {
            // all samples equal: X[0] = |sum y| / Nf, every other bin exactly 0 (what pocketfft returns for a constant frame)
            const double x0m = fabs((double)W * y0) / (double)Nf;
            for (int k = lane; k < Nf; k += kWave) cur[k] = (k == 0) ? x0m : 0.0;
            wsync();
        } else {
            fwd_pass0<SH>(buf, g_chirp, g_tw, W, lane);
            pass1<SH, true>(buf, g_tw, lane);
            pass2_product<SH>(buf, g_bp, lane);
            pass1<SH, false>(buf, g_tw, lane);
            back_pass0_magnitudes<SH>(buf, cur, g_tw, Nf, lane);
        }
        if (P.mode == 1) {            // spectrogram row (ShortTermFeatures.py:422)
            double *row = oc + (long long)t * Nf;
            for (int k = lane; k < Nf; k += kWave) __builtin_nontemporal_store(cur[k], row + k);
        } else if (P.mode == 2) {     // chromagram row (:356-359)
            double p = 0.0;
            for (int k = lane; k < Nf; k += kWave) { const double X = cur[k]; p = fma(X, X, p); }
            p = wsum(p);
            const double chv = chroma_class(tb, cur, p, lane);
            if (lane < 12) oc[(long long)t * 12 + lane] = chv;
        } else {
            if (want) {
                mix::frame_features_chunked(P, tb, tf, cur, (t == 0) ? cur : prv, fv, msp, ch_f, lane);
                const double v = (lane < kBase) ? fv[lane] : 0.0;
                if (t >= tl.t0) {
                    if (lane < kBase) {
                        otile[nslot * F + lane] = v;
                        if (P.deltas) otile[nslot * F + kBase + lane] = (t == 0) ? 0.0 : v - vprev;
                    }
                    ++nslot;
                }
                vprev = v;
            }
            if (nslot == kFlush || (t == tend - 1 && nslot > 0)) {
                wsync();
                // row segments: nslot consecutive frames of feature row f are contiguous in [F][T]
                for (int idx = lane; idx < F * kFlush; idx += kWave) {
                    const int f = idx / kFlush, i = idx % kFlush;
                    if (i < nslot) oc[(long long)f * Tc + tbase + i] = otile[i * F + f];
                }
                wsync();
                tbase += nslot;
                nslot = 0;
            }
        }
        asm volatile("" ::"v"(touched));      // (the prefetch loads retire here at the latest)
        wsync();
    }
}

// ---- host: does the window take this kernel, LDS layout, tables ---------------------------------------------------------
inline int blu_log2m(int window) {
    const int Nf = window / 2, need = window + Nf - 1;
    for (int lg = 8; lg <= 12; ++lg)
        if ((1 << lg) >= need) return lg;
    return 0;
}
// radices of the DIF passes of M = 2^lg (the kernel's Sched)
inline void blu_radices(int lg, int r[3]) {
    switch (lg) {
        case 8: r[0] = 4; r[1] = 8; r[2] = 8; break;
        case 9: r[0] = 8; r[1] = 8; r[2] = 8; break;
        case 10: r[0] = 16; r[1] = 8; r[2] = 8; break;
        case 11: r[0] = 16; r[1] = 16; r[2] = 8; break;
        default: r[0] = 16; r[1] = 16; r[2] = 16; break;
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
    const int lg = blu_log2m(W);
    if (!lg) return 0;
    const int M = 1 << lg;
    memset(&L, 0, sizeof(L));
    L.log2m = lg;
    L.unit_bytes = (Nf * 8 + 255) / 256 * 256;
    L.buf_bytes = M * 16;
    const int FF = F > 0 ? F : 1;
    L.wave_bytes = (L.buf_bytes + L.unit_bytes + kFlush * FF * 8 + 48 * 8 + 40 * 8 + 255) / 256 * 256;
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
    int rdx[3];
    blu_radices(lg, rdx);
    const int S0 = M / rdx[0], S1 = S0 / rdx[1];
    const int ntw = (rdx[0] - 1) * S0 + (rdx[1] - 1) * S1;
    L.off_g_chirp = take((size_t)W * 16);
    L.off_g_bp = take((size_t)M * 16);
    L.off_g_tw = take((size_t)ntw * 16);
    L.total_bytes = off;
    static const int max_waves[13] = {0, 0, 0, 0, 0, 0, 0, 0, 16, 12, 8, 4, 2};
    L.waves = max_waves[lg];
    while (L.waves > 1 && (size_t)L.table_bytes + (size_t)L.waves * L.wave_bytes > 160 * 1024) --L.waves;
    if ((size_t)L.table_bytes + (size_t)L.waves * L.wave_bytes > 160 * 1024) return 0;
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
    }
    const long double pi = 3.141592653589793238462643383279502884L;
    // c[m] = exp(i pi m^2 / W): m^2 is reduced mod 2 W in integers first
    auto chirp = [&](long long m, long double &cr, long double &ci) {
        const long long r = (m * m) % (2LL * W);
        const long double ang = pi * (long double)r / (long double)W;
        cr = cosl(ang); ci = sinl(ang);
    };
    double *gc = reinterpret_cast<double *>(b + L.off_g_chirp);
    for (int n = 0; n < W; ++n) {
        long double cr, ci;
        chirp(n, cr, ci);
        gc[2 * n] = (double)cr;
        gc[2 * n + 1] = (double)(-ci);                 // conj(c[n])
    }
    std::vector<long double> re((size_t)M, 0.0L), im((size_t)M, 0.0L);
    for (long long m = -(long long)(W - 1); m <= (long long)Nf - 1; ++m) {
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
        for (int p = 0; p < 3; ++p) {
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
    const int spans[2] = {M, S0}, strides[2] = {S0, S1};
    for (int p = 0; p < 2; ++p)
        for (int q = 1; q < rdx[p]; ++q)
            for (int k = 0; k < strides[p]; ++k) {
                const long double ang = -two_pi * (long double)(((long long)q * k) % spans[p]) / (long double)spans[p];
                gt[2 * at] = (double)cosl(ang);
                gt[2 * at + 1] = (double)sinl(ang);
                ++at;
            }
    return 1;
}
inline size_t blu_lds_bytes(const BluLayout &L) { return (size_t)L.table_bytes + (size_t)L.waves * L.wave_bytes; }

}  // namespace blu
}  // namespace paa
