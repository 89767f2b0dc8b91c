// Large windows -- beyond what one WAVE can hold in LDS (kernels_mix.hpp ends near 6 000 samples), e.g. the 1 s windows
// audioSegmentation.music_thumbnailing passes by default (audioSegmentation.py:1137: 16 000 samples at 16 kHz): ONE WORKGROUP
// per frame, the whole transform in LDS.
//
//   wg_spectrum_kernel  512 threads, one frame: samples -> normalised packed-complex sequence in LDS (Nc x 16 bytes: up to
//                       10 000 complex points, i.e. windows up to 20 000 samples even / 10 000 odd); zero-crossing count,
//                       energy and the ten entropy-block energies straight from that buffer (every wave a contiguous eighth of
//                       the frame); the in-place decimation-in-frequency passes of kernels_mix.hpp (radix 16 / 8 / 4 / 2 / 13 /
//                       11 / 7 / 5 / 3 butterflies in registers, two per lane in flight, one __syncthreads per pass, twiddles as the
//                       product of two LDS-resident factors W^(128 h) W^l -- no global load inside a pass); real-FFT recombination + |X| / num_fft read through the digit-reversal permutation and written
//                       ONCE to the frame's spectrum row in HBM (for spectrogram plans: straight into the output)
//   wg_feat_kernel      512 threads, one frame: the frame's spectrum row is staged in LDS (all loads in flight at once; the previous
//                       frame's row is swept from L2), then the 34 features with every sweep spread over the workgroup (block energies per block range, mel filters and
//                       chroma classes one wave at a time with all 64 lanes on the filter's bins)
//   wg_delta_kernel     rows 34..67 of every clip
//
// Three launches for ALL frames of ALL clips of a plan (kernels_big.hpp: about ten launches per clip and per chunk, every
// radix pass a round trip through HBM: 1.5 MB of traffic per 16 000-sample frame against the 16 KB + 64 KB this path moves).
// Windows whose transform does not fit the LDS (44 100 samples: 22 050 complex points = 353 KB) keep kernels_big.hpp.
// Replaces ShortTermFeatures.py:608-682 (+ helpers :22-140, :236-321), spectrogram (:415-422), chromagram (:349-359).
#pragma once
#include "kernels_mix.hpp"

namespace paa {
namespace wg {

// spectrum kernel: NT = 768 threads (twelve waves, three per SIMD, 121 registers) when every radix of the schedule is <= 8,
// NT = 512 (two per SIMD, 175 registers: the radix-16 / 13 / 11 butterflies fit) otherwise; 1024 threads spilled 31 registers
// at the 128 the hardware then allows and was no faster (scripts/rounds/r05/gpu_r05g.sh: 0.2022 / 0.2004 / 0.2082 ms per 1 199
// frames at 1024 / 768 / 512 threads)
constexpr int kFeatThreads = 512;
constexpr int kFeatWaves = kFeatThreads / 64;
#ifndef PAA_WG_ABLATE
#define PAA_WG_ABLATE 0            // timing builds of scripts/rounds/r05 only: bit mask of phases that are skipped
#endif
constexpr int kAblate = PAA_WG_ABLATE;
constexpr int kTwLo = 128;              // two-level twiddles: W^m = W^(128 (m >> 7)) W^(m & 127), both factors in LDS

struct WgLayout {
    int n_pass;
    int radix[mix::kMaxPass], span[mix::kMaxPass], tws[mix::kMaxPass];
    unsigned magic[mix::kMaxPass];
    int top;                        // elements per top-level block of the first pass (Nc / radix[0]); the LDS buffer holds one PAD
                                    // element after each: element e sits at e + e / top -- the digit-reversed reads of the magnitude
                                    // pass walk the top-level digit first, and top x 16 bytes is a multiple of the 256 bytes of one
                                    // bank sweep for the usual lengths (8000 = 8 x 1000: every lane of a read on two bank groups)
    unsigned magic_top;             // ceil(2^32 / top): e / top for e < 2^16
    int off_red, off_twlo, off_twhi, off_perm;      // byte offsets into the LDS behind the transform buffer
    int n_twhi;
    int perm_lds;                   // 1: the digit-reversal permutation fits the LDS beside the buffer
    int threads;                    // spectrum kernel: 768 when every radix is <= 8, else 512
    int lds_bytes;                  // spectrum kernel
    int feat_lds_bytes;             // feature kernel: the frame's spectrum row + 2 KB (2 KB alone when the row is not staged)
    int feat_staged;                // 1: the feature kernel stages the frame's row in LDS; 0: the row exceeds the LDS, it is read where it lies
    // split transforms (wg_split_kernel): the sequence is longer than the LDS -- its first radix-r0 pass runs straight from the
    // samples, one sub-transform q (of r0) at a time, and the schedule above describes ONE sub-transform of `sub` elements
    // (span[0] = sub, tws[p] = Nc / span[p]); 0: the whole transform is in LDS (wg_spectrum_kernel)
    int r0, sub;
    int off_cv;                     // [8] complex: W_r0^(r q) of the task
};
// (FrameRef -- one frame of the launch -- lives in device_common.hpp: kernels_wgs.hpp uses it too)

struct Tw2 {
    const double2 *lo, *hi;
    __device__ __forceinline__ double2 get(int m) const { return cmul(hi[m >> 7], lo[m & (kTwLo - 1)]); }
};

__device__ __forceinline__ double2 csqr(double2 a) { return make_double2(fma(a.x, a.x, -a.y * a.y), 2.0 * (a.x * a.y)); }

// U butterflies of one in-place DIF pass per lane (kernels_mix.hpp's dif_batch): butterfly b works on the R elements
// base + r * stride of its block; output q is multiplied by W_M^(q k) and goes back to base + q * stride.  Butterflies touch
// disjoint elements: no ordering inside a pass.  The twiddles of a butterfly are the powers of ONE table value W^(k tws) (the
// product of the two LDS factors), formed by squaring / multiplying (depth log2 R: a few ulp, far inside the gates) -- the R - 1
// table look-ups they replace were the larger half of a pass's instructions.  LDS addresses follow the padded layout (WgLayout::top).
template <int R, int U, bool FIRST, int NT>
__device__ __forceinline__ void wg_dif_batch(double2 *buf, int nb, int stride, int M, int tws, unsigned magic, unsigned magic_top,
                                             const Tw2 &tw, int b0) {
    double2 v[U][R];
    int base[U], t1[U];
    bool act[U];
    const int lstride = FIRST ? stride + 1 : stride;           // first pass: element r of a butterfly lies in top-level block r
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int b = b0 + NT * u;
        act[u] = b < nb;
        const int be = act[u] ? b : nb - 1;                     // (lanes past the end shadow a valid butterfly, stores masked)
        const int blk = (stride == 1) ? be : (int)__umulhi((unsigned)be, magic);
        const int k = be - __mul24(blk, stride);
        const int e0 = __mul24(blk, M) + k;
        base[u] = FIRST ? e0 + blk * R : e0 + (int)__umulhi((unsigned)e0, magic_top);      // (first pass: e0 / top = blk R -- blk > 0 only in
                                                                                           // the two sub-transforms of wg_split_kernel; later passes:
                                                                                           // a butterfly stays inside one top-level block)
        t1[u] = __mul24(k, tws);
#pragma unroll
        for (int r = 0; r < R; ++r) v[u][r] = buf[base[u] + r * lstride];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        mix::Bfly<R>::run(v[u]);
        if (stride > 1) {
            double2 wq[R];
            wq[1] = tw.get(t1[u]);
#pragma unroll
            for (int q = 2; q < R; ++q) wq[q] = (q % 2 == 0) ? csqr(wq[q / 2]) : cmul(wq[q / 2], wq[q - q / 2]);
#pragma unroll
            for (int q = 1; q < R; ++q) v[u][mix::Bfly<R>::pos(q)] = cmul(v[u][mix::Bfly<R>::pos(q)], wq[q]);
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if (act[u]) {
#pragma unroll
            for (int q = 0; q < R; ++q) buf[base[u] + q * lstride] = v[u][mix::Bfly<R>::pos(q)];
        }
    }
}
template <int R, int NT>
__device__ __forceinline__ void wg_dif_pass(double2 *buf, int Nc, int M, int tws, unsigned magic, unsigned magic_top, bool first,
                                            const Tw2 &tw, int tid) {
#ifndef PAA_WG_U8
#define PAA_WG_U8 0               // 1: A/B build of scripts/rounds/r05/gpu_r05ao.sh -- two radix-8 butterflies in flight per lane in the 768-thread
                                  // instance too (153 registers): 1.5-5 % SLOWER at 16 000 / 8 000, and wg_split_kernel spills
#endif
    constexpr int U = (NT <= 768 && R <= ((NT == 512 || PAA_WG_U8) ? 8 : 5)) ? 2 : 1;          // two butterflies in flight per lane where the registers allow it
    const int stride = M / R, nb = Nc / R;
    // (wave-uniform trip count: a wave whose first butterfly exists runs the batch)
    if (first) {
        for (int b0 = tid; (b0 & ~63) < nb; b0 += U * NT) wg_dif_batch<R, U, true, NT>(buf, nb, stride, M, tws, magic, magic_top, tw, b0);
    } else {
        for (int b0 = tid; (b0 & ~63) < nb; b0 += U * NT) wg_dif_batch<R, U, false, NT>(buf, nb, stride, M, tws, magic, magic_top, tw, b0);
    }
}

// all in-place passes of the schedule over the `n_el` elements of the buffer (one transform of span[0] elements, or the two
// sub-transforms of wg_split_kernel side by side); ends with a barrier
template <int NT>
__device__ __forceinline__ void wg_run_passes(double2 *buf, int n_el, const WgLayout &L, const Tw2 &tw, int tid) {
    const unsigned mtop = L.magic_top;
    for (int p = 0; p < ((kAblate & 4) ? 0 : L.n_pass); ++p) {
        const int M = L.span[p], ts = L.tws[p];
        const unsigned mg = L.magic[p];
        switch (L.radix[p]) {
            case 2: wg_dif_pass<2, NT>(buf, n_el, M, ts, mg, mtop, p == 0, tw, tid); break;
            case 3: wg_dif_pass<3, NT>(buf, n_el, M, ts, mg, mtop, p == 0, tw, tid); break;
            case 4: wg_dif_pass<4, NT>(buf, n_el, M, ts, mg, mtop, p == 0, tw, tid); break;
            case 5: wg_dif_pass<5, NT>(buf, n_el, M, ts, mg, mtop, p == 0, tw, tid); break;
            case 7: wg_dif_pass<7, NT>(buf, n_el, M, ts, mg, mtop, p == 0, tw, tid); break;
            case 8: wg_dif_pass<8, NT>(buf, n_el, M, ts, mg, mtop, p == 0, tw, tid); break;
            case 11: if constexpr (NT == 512) wg_dif_pass<11, NT>(buf, n_el, M, ts, mg, mtop, p == 0, tw, tid); break;
            case 13: if constexpr (NT == 512) wg_dif_pass<13, NT>(buf, n_el, M, ts, mg, mtop, p == 0, tw, tid); break;
            default: if constexpr (NT == 512) wg_dif_pass<16, NT>(buf, n_el, M, ts, mg, mtop, p == 0, tw, tid); break;
        }
        __syncthreads();
    }
}

// lane l receives the value of lane l - 1 (lane 0: `first`)
__device__ __forceinline__ int wg_shr1(int v, int first) { return __builtin_amdgcn_update_dpp(first, v, 0x138, 0xF, 0xF, false); }

// ---- time domain (ShortTermFeatures.py:22-51) of one frame by a workgroup of kWaves waves.  `get(p)` returns element p of the
// normalised sequence (a sample pair for even windows, (sample, 0) for odd ones).  Wave w owns the elements [w per, (w + 1) per);
// its samples meet at most three of the ten entropy blocks (or the tail the reference leaves out of them, block "10"): the energy
// of the first, of the last and of all of them are summed, the middle one is the rest.  Sign codes are formed once per sample;
// the left neighbour comes from the lane below (lane 0: the previous iteration's lane 63, or the element before the range).
// red[5 w ..]: first / middle / last block energy, zero-crossing count, first block index of wave w.
template <int kWaves, typename Get>
__device__ __forceinline__ void wg_time_partials(const PlanDev &P, int Nc, int lane, int wave, double *red, Get get) {
    const int LT = P.blk_t;
    const int spe = P.even ? 2 : 1;                       // samples per element
    const int per = (Nc + kWaves - 1) / kWaves;
    const int p0 = wave * per, p1 = min(Nc, p0 + per);
    const int b0 = min((p0 * spe) / LT, 10);
    const int bnd1 = (b0 < 10) ? (b0 + 1) * LT : 0x7fffffff, bnd2 = (b0 + 1 < 10) ? (b0 + 2) * LT : 0x7fffffff;
    double aA = 0.0, aC = 0.0, aT = 0.0;
    int zc = 0;
    int carry = 0;            // sign code of the sample before this iteration's first one
    bool have_left = false;
    if (p0 > 0 && p0 < p1) {
        const double2 zl = get(p0 - 1);
        const double vl = P.even ? zl.y : zl.x;
        carry = (vl > 0.0) - (vl < 0.0);
        have_left = true;
    }
    constexpr int TB = 4;            // elements per lane whose loads are in flight together (the sign chain below is sequential)
    for (int pq = p0; pq < p1; pq += 64 * TB) {
        double2 zz[TB];
#pragma unroll
        for (int u = 0; u < TB; ++u) zz[u] = get(min(pq + 64 * u + lane, p1 - 1));
#pragma unroll
        for (int u = 0; u < TB; ++u) {
        const int pb = pq + 64 * u;
        if (pb >= p1) break;
        const int p = pb + lane;
        const bool in = p < p1;
        const int pe = in ? p : p1 - 1;
        const double2 z = zz[u];
        const double v0 = in ? z.x : 0.0, v1 = (in && P.even) ? z.y : 0.0;
        const int n = pe * spe;
        const double e0 = v0 * v0, e1 = v1 * v1;
        aT += e0 + e1;
        aA += ((n < bnd1) ? e0 : 0.0) + ((n + 1 < bnd1) ? e1 : 0.0);
        aC += ((n >= bnd2) ? e0 : 0.0) + ((n + 1 >= bnd2) ? e1 : 0.0);
        const int c0 = (v0 > 0.0) - (v0 < 0.0), c1 = (v1 > 0.0) - (v1 < 0.0);
        const int last = P.even ? c1 : c0;                 // the element's last sample
        const int first_left = have_left ? carry : __builtin_amdgcn_readfirstlane(c0);      // (the frame's first sample meets itself)
        const int left = wg_shr1(last, first_left);
        if (in) zc += abs(c0 - left) + (P.even ? abs(c1 - c0) : 0);
        carry = __builtin_amdgcn_readlane(last, 63);
        have_left = true;
        }
    }
    aA = wsum(aA); aC = wsum(aC); aT = wsum(aT);
    zc = wsum_i(zc);
    if (lane == 0) { red[5 * wave] = aA; red[5 * wave + 1] = (aT - aA) - aC; red[5 * wave + 2] = aC; red[5 * wave + 3] = (double)zc; red[5 * wave + 4] = (double)b0; }
}
// (one wave, after a barrier) block j (lane j < 11) = the parts of every wave that fall into it, added in wave order
template <int kWaves>
__device__ __forceinline__ void wg_time_finish(const double *red, int lane, double *tfp) {
    double E = 0.0, zct = 0.0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) {
        const int bw = (int)red[5 * w + 4];
        E += (bw == lane) ? red[5 * w] : 0.0;
        E += (bw + 1 == lane) ? red[5 * w + 1] : 0.0;
        E += (bw + 2 == lane) ? red[5 * w + 2] : 0.0;
        zct += red[5 * w + 3];
    }
    const double e_tot = wsum((lane < 11) ? E : 0.0);
    const double s = fast_div(E, e_tot + kEps);
    const double ent = wsum((lane < 10) ? -(s * fast_log2(s + kEps)) : 0.0);
    if (lane == 0) { tfp[0] = e_tot; tfp[1] = ent; tfp[2] = zct; }
}

// P.mode decides where the row goes and whether the time-domain features are formed.  PERSISTENT: the transform buffer takes
// the CU's LDS, so one workgroup lives on a CU and nothing overlaps the latency chain at the head of a frame (frame record ->
// clip record -> first samples: four dependent trips to L2 / HBM) -- a workgroup therefore walks frames blockIdx.x,
// + gridDim.x, ..., loads the tables once, and fetches the next frame's records and touches its samples while it still
// computes the magnitudes of the current one
template <typename T, int NT>
__global__ __launch_bounds__(NT) void wg_spectrum_kernel(PlanDev P, WgLayout L, const unsigned short *__restrict__ perm_g,
                                                         const T *__restrict__ sig, const ClipDev *__restrict__ clips,
                                                         const ClipNorm *__restrict__ norms,
                                                         const FrameRef *__restrict__ frames, int n_frames,
                                                         double *__restrict__ spec, double *__restrict__ tfeat,
                                                         double *__restrict__ out) {
    constexpr int kThreads = NT, kWaves = NT / 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double2 *buf = reinterpret_cast<double2 *>(smem);
    double *red = reinterpret_cast<double *>(smem + L.off_red);        // [kWaves][5]
    double2 *twlo = reinterpret_cast<double2 *>(smem + L.off_twlo), *twhi = reinterpret_cast<double2 *>(smem + L.off_twhi);
    unsigned short *perm_l = reinterpret_cast<unsigned short *>(smem + L.off_perm);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int W = P.W, Nc = P.Nc, Nf = P.Nf;
    const double sc = sample_scale<T>();
    // ---- tables, once per workgroup: the two twiddle factors (and the permutation when it fits)
    if (tid < kTwLo) twlo[tid] = P.tw[tid < Nc ? tid : 0];
    if (tid >= 256 && tid - 256 < L.n_twhi) twhi[tid - 256] = P.tw[(tid - 256) * kTwLo];
    if (L.perm_lds) {
        const unsigned *src = reinterpret_cast<const unsigned *>(perm_g);
        unsigned *dst = reinterpret_cast<unsigned *>(perm_l);
        for (int i = tid; i < (Nc + 1) / 2; i += kThreads) dst[i] = src[i];
    }
    if ((int)blockIdx.x >= n_frames) return;
    // (only the fields that are used travel from frame to frame: whole records kept live across the loop went to scratch)
    struct Cur { long long x_off, out_off; double mean, inv; int t, row, halo, pad; };
    auto fetch = [&](int f) {
        const FrameRef r = frames[f];
        const ClipDev cd = clips[r.clip];
        const ClipNorm n_ = norms[r.clip];
        Cur q;
        q.x_off = cd.sample_off + P.frame_origin + (long long)r.t * P.S;
        q.out_off = cd.out_off; q.mean = n_.mean; q.inv = n_.inv; q.t = r.t; q.row = r.row; q.halo = r.halo; q.pad = 0;
        return q;
    };
    Cur cu = fetch(blockIdx.x);
    for (int fidx = blockIdx.x; fidx < n_frames; fidx += gridDim.x) {
    const T *x = sig + cu.x_off;
    struct { int t, row, halo; } fr = {cu.t, cu.row, cu.halo};
    struct { double mean, inv; } nm = {cu.mean, cu.inv};
    struct { long long out_off; } c = {cu.out_off};
    // ---- load: y = (x / 2^15 - mean) / (max|.| + 1e-10) (ShortTermFeatures.py:567-570); even windows packed two samples per point;
    // element e of the sequence sits at buf[e + e / top]
    const unsigned mtop = L.magic_top;
    if (kAblate & 1) {
        for (int p = tid; p < Nc; p += kThreads) buf[p + (int)__umulhi((unsigned)p, mtop)] = make_double2(1e-3 * (double)(p & 7), 1e-3);
    } else if (P.even) {
#pragma unroll 4
        for (int p = tid; p < Nc; p += kThreads) {
            const double2 xx = ct::PairLoad<T>::get(x + 2 * p);
            buf[p + (int)__umulhi((unsigned)p, mtop)] = make_double2(fma(xx.x, sc, -nm.mean) * nm.inv, fma(xx.y, sc, -nm.mean) * nm.inv);
        }
    } else {
#pragma unroll 4
        for (int n = tid; n < W; n += kThreads)
            buf[n + (int)__umulhi((unsigned)n, mtop)] = make_double2(fma(load_sample<T>(x + n), sc, -nm.mean) * nm.inv, 0.0);
    }
    __syncthreads();
    // ---- time domain (:22-51) straight from the buffer
    if (P.mode == 0 && !fr.halo && !(kAblate & 2))
        wg_time_partials<kWaves>(P, Nc, lane, wave, red, [&](int p) { return buf[p + (int)__umulhi((unsigned)p, mtop)]; });
    __syncthreads();          // every wave has read its samples: the passes may overwrite the buffer
    if (P.mode == 0 && !fr.halo && wave == 0) wg_time_finish<kWaves>(red, lane, tfeat + 3 * (long long)fr.row);
    // ---- in-place DIF passes (kernels_mix.hpp's butterflies; two waves per SIMD and two butterflies per lane hide the latency)
    const Tw2 tw = {twlo, twhi};
    wg_run_passes<NT>(buf, Nc, L, tw, tid);
    // ---- the next frame of this workgroup: its records now, a touch of its samples (one load per lane and 128-byte line of the
    // frame: they land in L2 / L1 while the magnitudes below are formed; the sum keeps the loads alive)
    Cur cu_next = cu;
    const int f_next = fidx + (int)gridDim.x;
    if (f_next < n_frames) {
        cu_next = fetch(f_next);
        const char *xn = reinterpret_cast<const char *>(sig + cu_next.x_off);
        const int bytes = W * (int)sizeof(T);
        int touch = 0;
        for (int o = tid * 128; o < bytes; o += kThreads * 128) touch += *reinterpret_cast<const volatile char *>(xn + o);
        asm volatile("" ::"v"(touch));
    }
    // ---- |X| / num_fft (:617-621) through the digit-reversal permutation, written once to the frame's row
    double *row = (P.mode == 1) ? out + c.out_off + (long long)fr.t * Nf : spec + (long long)fr.row * Nf;
    const double invNf = 1.0 / (double)Nf;
    const unsigned short *perm = L.perm_lds ? perm_l : perm_g;
    if (kAblate & 8) {
        if (tid == 0) row[0] = buf[5].x;
    } else if (P.even) {
        // bins k and Nc - k from one pair: X[k] = E + w^k O, X[Nc - k] = conj(E - w^k O), E = (Z[k] + conj Z[Nc - k]) / 2,
        // O = -i (Z[k] - conj Z[Nc - k]) / 2; four pairs in flight per lane
        const int npairs = Nc / 2 + 1;
        constexpr int MU = (NT == 512) ? 4 : 2;          // pairs in flight per lane
        for (int k0 = tid; (k0 & ~63) < npairs; k0 += MU * kThreads) {
            int kk[MU], pa[MU], pb[MU];
            double2 pw[MU], zk[MU], zm[MU];
#pragma unroll
            for (int u = 0; u < MU; ++u) {
                kk[u] = min(k0 + u * kThreads, npairs - 1);
                pw[u] = P.post[kk[u]];
                pa[u] = L.perm_lds ? perm_l[kk[u]] : perm_g[kk[u]];
                pb[u] = L.perm_lds ? perm_l[kk[u] == 0 ? 0 : Nc - kk[u]] : perm_g[kk[u] == 0 ? 0 : Nc - kk[u]];
            }
#pragma unroll
            for (int u = 0; u < MU; ++u) { zk[u] = buf[pa[u]]; zm[u] = buf[pb[u]]; }
#pragma unroll
            for (int u = 0; u < MU; ++u) {
                const int k = kk[u];
                const double2 e = make_double2(0.5 * (zk[u].x + zm[u].x), 0.5 * (zk[u].y - zm[u].y));
                const double2 o = make_double2(0.5 * (zk[u].y + zm[u].y), 0.5 * (zm[u].x - zk[u].x));
                const double2 wo = cmul(pw[u], o);
                const double ar = e.x + wo.x, ai = e.y + wo.y, br = e.x - wo.x, bi = e.y - wo.y;
                if (k0 + u * kThreads < npairs) {
                    row[k] = mag_sqrt(fma(ar, ar, ai * ai)) * invNf;
                    if (k > 0 && Nc - k != k) row[Nc - k] = mag_sqrt(fma(br, br, bi * bi)) * invNf;
                }
            }
        }
    } else {
#pragma unroll 4
        for (int k = tid; k < Nf; k += kThreads) {
            const double2 z = buf[perm[k]];
            row[k] = mag_sqrt(fma(z.x, z.x, z.y * z.y)) * invNf;
        }
    }
    __syncthreads();          // the buffer is free for the next frame
    cu = cu_next;
    }       // frames of this workgroup
}

// ---- sequences longer than one CU's LDS (44 100-sample windows: 22 050 complex points = 353 KB) ----------------------------------
// Decimation in frequency by r0 FIRST, straight from the samples: y_q[k] = W_Nc^(q k) sum_r z[k + r sub] W_r0^(r q) for k < sub is
// the input of sub-transform q, whose output kappa is Z[q + r0 kappa].  The real-FFT recombination pairs bin k with Nc - k, i.e.
// sub-transform q (output kappa) with sub-transform r0 - q (output sub - 1 - kappa; q = 0: with itself at sub - kappa).  A TASK is
// therefore one frame and the sub-transforms {q, r0 - q} -- both in LDS side by side, passes run over both at once -- or {0} or
// {r0 / 2} alone; a persistent workgroup walks tasks.  Magnitudes go straight to the frame's row (8-byte stores r0 doubles
// apart: the row's lines fill up in L2 from the tasks of the frame).  The time-domain features come from wg_time_kernel.
// FrameRef::halo of a task: bit 0 = halo frame, bits 8.. = q.
template <typename T, int NT>
__global__ __launch_bounds__(NT) void wg_split_kernel(PlanDev P, WgLayout L, const unsigned short *__restrict__ perm_g,
                                                      const T *__restrict__ sig, const ClipDev *__restrict__ clips,
                                                      const ClipNorm *__restrict__ norms,
                                                      const FrameRef *__restrict__ tasks, int n_tasks, int *next_task,
                                                      double *__restrict__ spec, double *__restrict__ out) {
    constexpr int kThreads = NT;
    __shared__ int s_next;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double2 *buf = reinterpret_cast<double2 *>(smem);
    double2 *twlo = reinterpret_cast<double2 *>(smem + L.off_twlo), *twhi = reinterpret_cast<double2 *>(smem + L.off_twhi);
    double2 *cv = reinterpret_cast<double2 *>(smem + L.off_cv);
    unsigned short *perm_l = reinterpret_cast<unsigned short *>(smem + L.off_perm);
    const int tid = threadIdx.x;
    const int Nc = P.Nc, Nf = P.Nf, S = L.sub, R0 = L.r0;
    const int pitch = S + L.radix[0];                    // padded elements per sub-transform
    const double sc = sample_scale<T>();
    if (tid < kTwLo) twlo[tid] = P.tw[tid];
    if (tid >= 256 && tid - 256 < L.n_twhi) twhi[tid - 256] = P.tw[(tid - 256) * kTwLo];
    if (L.perm_lds) {
        const unsigned *src = reinterpret_cast<const unsigned *>(perm_g);
        unsigned *dst = reinterpret_cast<unsigned *>(perm_l);
        for (int i = tid; i < (S + 1) / 2; i += kThreads) dst[i] = src[i];
    }
    const Tw2 tw = {twlo, twhi};
    const unsigned mtop = L.magic_top;
    const unsigned short *perm = L.perm_lds ? perm_l : perm_g;
    const double invNf = 1.0 / (double)Nf;
    // (only the fields that are used travel from task to task, like wg_spectrum_kernel's)
    struct Cur { long long x_off, out_off; double mean, inv; int t, row, q, pad; };
    auto fetch = [&](int ti) {
        const FrameRef r = tasks[ti];
        const ClipDev cd = clips[r.clip];
        const ClipNorm n_ = norms[r.clip];
        Cur c;
        c.x_off = cd.sample_off + P.frame_origin + (long long)r.t * P.S;
        c.out_off = cd.out_off; c.mean = n_.mean; c.inv = n_.inv; c.t = r.t; c.row = r.row; c.q = r.halo >> 8; c.pad = 0;
        return c;
    };
    if ((int)blockIdx.x >= n_tasks) return;
    Cur cu = fetch(blockIdx.x);
    // tasks are handed out through a counter (*next_task, zero at launch): pair tasks cost twice what single ones do and come
    // first in the list -- a fixed stride gave some workgroups nothing but pairs
    for (int ti = blockIdx.x; ti < n_tasks;) {
        const T *x = sig + cu.x_off;
        struct { double mean, inv; } nm = {cu.mean, cu.inv};
        struct { int t, row; } fr = {cu.t, cu.row};
        struct { long long out_off; } cd = {cu.out_off};
        const int qa = cu.q, qb = (qa == 0) ? 0 : R0 - qa;
        const bool two = qb != qa;
        __syncthreads();          // the previous task's magnitudes have been read (and the tables are in place)
        if (tid < R0) cv[tid] = P.tw[((tid * qa) % R0) * S];             // W_Nc^(S m) = W_r0^m
        if (tid == 64) s_next = (int)gridDim.x + atomicAdd(next_task, 1);          // (read after the barriers of the passes)
        __syncthreads();
        // ---- first pass from the samples: up to eight elements k + r sub of the normalised sequence per output
#pragma unroll 2
        for (int k = tid; k < S; k += kThreads) {
            double2 z[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                if (r < R0) {
                    const int e = k + r * S;
                    if (P.even) {
                        const double2 xx = ct::PairLoad<T>::get(x + 2 * e);
                        z[r] = make_double2(fma(xx.x, sc, -nm.mean) * nm.inv, fma(xx.y, sc, -nm.mean) * nm.inv);
                    } else {
                        z[r] = make_double2(fma(load_sample<T>(x + e), sc, -nm.mean) * nm.inv, 0.0);
                    }
                }
            }
            double ax = 0.0, ay = 0.0, bx = 0.0, by = 0.0;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                if (r < R0) {
                    const double2 c = cv[r];
                    const double p0 = z[r].x * c.x, p1 = z[r].y * c.y, p2 = z[r].x * c.y, p3 = z[r].y * c.x;
                    ax += p0 - p1; ay += p2 + p3;              // z c
                    bx += p0 + p1; by += p3 - p2;              // z conj(c): sub-transform r0 - q
                }
            }
            const int pos = k + (int)__umulhi((unsigned)k, mtop);
            buf[pos] = cmul(make_double2(ax, ay), tw.get(qa * k));
            if (two) buf[pitch + pos] = cmul(make_double2(bx, by), tw.get(qb * k));
        }
        __syncthreads();
        wg_run_passes<NT>(buf, two ? 2 * S : S, L, tw, tid);
        // ---- the next task of this workgroup: its records now, a touch of its samples (they land in L2 while the magnitudes are formed)
        Cur cu_next = cu;
        const int t_next = s_next;
        if (t_next < n_tasks) {
            cu_next = fetch(t_next);
            const char *xn = reinterpret_cast<const char *>(sig + cu_next.x_off);
            const int bytes = P.W * (int)sizeof(T);
            int touch = 0;
            for (int o = tid * 128; o < bytes; o += kThreads * 128) touch += *reinterpret_cast<const volatile char *>(xn + o);
            asm volatile("" ::"v"(touch));
        }
        // ---- magnitudes (:617-621): bin k = qa + r0 kappa from sub-transform a, its partner Nc - k from b
        double *row = (P.mode == 1) ? out + cd.out_off + (long long)fr.t * Nf : spec + (long long)fr.row * Nf;
        const double2 *bufb = two ? buf + pitch : buf;
        if (P.even) {
            // single sub-transforms pair with themselves: kappa <= its partner's index only
            const int n_k = two ? S : (qa == 0 ? S / 2 + 1 : (S + 1) / 2);
            for (int k0 = tid; (k0 & ~63) < n_k; k0 += 2 * kThreads) {
                int ka[2], lo[2];
                bool flip[2];
                double2 pw[2], zk[2], zm[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    ka[u] = min(k0 + u * kThreads, n_k - 1);
                    const int kb = (qa == 0) ? (ka[u] == 0 ? 0 : S - ka[u]) : S - 1 - ka[u];
                    const int k = qa + R0 * ka[u];
                    flip[u] = 2 * k > Nc;                         // the pair is (lo, Nc - lo) with lo <= Nc / 2
                    lo[u] = flip[u] ? Nc - k : k;
                    pw[u] = P.post[lo[u]];
                    zk[u] = buf[perm[ka[u]]];
                    zm[u] = bufb[perm[kb]];
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const double2 zl = flip[u] ? zm[u] : zk[u], zh = flip[u] ? zk[u] : zm[u];
                    const double2 e = make_double2(0.5 * (zl.x + zh.x), 0.5 * (zl.y - zh.y));
                    const double2 o = make_double2(0.5 * (zl.y + zh.y), 0.5 * (zh.x - zl.x));
                    const double2 wo = cmul(pw[u], o);
                    const double ar = e.x + wo.x, ai = e.y + wo.y, br = e.x - wo.x, bi = e.y - wo.y;
                    if (k0 + u * kThreads < n_k) {
                        const int l = lo[u];
                        row[l] = mag_sqrt(fma(ar, ar, ai * ai)) * invNf;
                        if (l > 0 && Nc - l != l) row[Nc - l] = mag_sqrt(fma(br, br, bi * bi)) * invNf;
                    }
                }
            }
        } else {
            // odd windows: one sample per element, X[k] = Z[k] for k < Nf
            for (int h = 0; h < (two ? 2 : 1); ++h) {
                const int q = h ? qb : qa;
                const double2 *bb = h ? bufb : buf;
                for (int ka = tid; ka < S; ka += kThreads) {
                    const int k = q + R0 * ka;
                    if (k < Nf) {
                        const double2 z = bb[perm[ka]];
                        row[k] = mag_sqrt(fma(z.x, z.x, z.y * z.y)) * invNf;
                    }
                }
            }
        }
        cu = cu_next;
        ti = t_next;
    }
}

// time-domain features of the frames of a split plan: one workgroup per frame, the samples read where they lie
constexpr int kTimeWaves = 16;
template <typename T>
__global__ __launch_bounds__(64 * kTimeWaves) void wg_time_kernel(PlanDev P, const T *__restrict__ sig, const ClipDev *__restrict__ clips,
                                                      const ClipNorm *__restrict__ norms, const FrameRef *__restrict__ frames,
                                                      double *__restrict__ tfeat) {
    __shared__ double red[kTimeWaves * 5];
    const FrameRef fr = frames[blockIdx.x];
    if (fr.halo) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const ClipDev cd = clips[fr.clip];
    const ClipNorm nm = norms[fr.clip];
    const T *x = sig + cd.sample_off + P.frame_origin + (long long)fr.t * P.S;
    const double sc = sample_scale<T>();
    const bool even = P.even;
    wg_time_partials<kTimeWaves>(P, P.Nc, lane, wave, red, [&](int p) {
        if (even) {
            const double2 xx = ct::PairLoad<T>::get(x + 2 * p);
            return make_double2(fma(xx.x, sc, -nm.mean) * nm.inv, fma(xx.y, sc, -nm.mean) * nm.inv);
        }
        return make_double2(fma(load_sample<T>(x + p), sc, -nm.mean) * nm.inv, 0.0);
    });
    __syncthreads();
    if (wave == 0) wg_time_finish<kTimeWaves>(red, lane, tfeat + 3 * (long long)fr.row);
}

// sum over the workgroup: every wave's total through LDS (slot[kFeatWaves]); all threads return the same bits
__device__ __forceinline__ double bsum8(double v, double *slot, int lane, int wave) {
    v = wsum(v);
    if (lane == 0) slot[wave] = v;
    __syncthreads();
    const double r = ((slot[0] + slot[1]) + (slot[2] + slot[3])) + ((slot[4] + slot[5]) + (slot[6] + slot[7]));
    __syncthreads();
    return r;
}

// the 34 features (or the 12 chromagram values) of one frame from its spectrum row and the previous frame's: both rows are
// staged in LDS first (16 independent loads per lane and row), every sweep then runs from there
template <bool STAGED>
__global__ __launch_bounds__(kFeatThreads) void wg_feat_kernel(PlanDev P, const FrameRef *__restrict__ frames,
                                                              const ClipDev *__restrict__ clips, const double *__restrict__ spec,
                                                              const double *__restrict__ tfeat, double *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const FrameRef fr = frames[blockIdx.x];
    if (fr.halo) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const ClipDev c = clips[fr.clip];
    const int Nf = P.Nf, W = P.W;
    double *cur_l = reinterpret_cast<double *>(smem);
    double *fv = cur_l + (STAGED ? Nf : 0);               // [48]
    double *msp = fv + 48;               // [40]
    double *red = msp + 40;              // [kFeatWaves][16]
    double *slot = red + kFeatWaves * 16;      // [kFeatWaves]
    int *redi = reinterpret_cast<int *>(slot + kFeatWaves);
    // the frame's own row is staged in LDS (it is swept five times, two of them as gathers); the previous frame's row -- two
    // coalesced sweeps -- is read where it lies (L2): 8 Nf bytes of LDS per workgroup, so two workgroups share a CU and the
    // staging of one runs under the sweeps of the other
    const double *gc = spec + (long long)fr.row * Nf;
    const double *prv = (fr.t == 0) ? gc : gc - Nf;          // frames are laid out in clip order: the previous frame is the previous row
    if (STAGED) {
#pragma unroll 8
        for (int k = tid; k < ((kAblate & 16) ? 64 : Nf); k += kFeatThreads) cur_l[k] = gc[k];
    }
    __syncthreads();
    // (a row beyond the LDS -- more than 20 000 bins -- is swept from L2 instead)
    const double *cur = STAGED ? cur_l : gc;
    double *oc = out + c.out_off;
    const long long Tc = c.T;
    const Tabs tb = tabs_global(P);
    // ---- sweep A (:57-107): sums, maximum, the ten block energies + the tail, block range by block range
    const int LB = P.blk_f;
    double part[15];          // 0..9 blocks, 10 tail, 11 sum X, 12 sum X_prev, 13 sum (k + 1) X, 14 max
#pragma unroll
    for (int i = 0; i < 15; ++i) part[i] = 0.0;
    if constexpr (!STAGED) {
        // the row lies in L2: ONE strided sweep with eight loads in flight (block by block, every short loop waited for its own
        // loads).  (Which thread adds which element differs from the staged form: last-bit differences between the two forms,
        // which never meet -- a window always takes the same one.)
        const unsigned mlb = (unsigned)(((1ULL << 32) + (unsigned)LB - 1) / (unsigned)LB);
#pragma unroll 4
        for (int k = tid; k < Nf; k += kFeatThreads) {
            const double X = cur[k];
            part[11] += X;
            part[12] += prv[k];
            part[13] = fma((double)(k + 1), X, part[13]);
            part[14] = fmax(part[14], X);
            const int j = min((int)__umulhi((unsigned)k, mlb), 10);
            const double sq = X * X;
#pragma unroll
            for (int i = 0; i < 11; ++i) part[i] += (i == j) ? sq : 0.0;
        }
    } else {
#pragma unroll
        for (int j = 0; j < 11; ++j) {
            const int lo = j * LB, hi = (kAblate & 32) ? lo : ((j < 10) ? lo + LB : Nf);
            double p = 0.0;
            for (int k = lo + tid; k < hi; k += kFeatThreads) {
                const double X = cur[k];
                part[11] += X;
                part[12] += prv[k];
                part[13] = fma((double)(k + 1), X, part[13]);
                part[14] = fmax(part[14], X);
                p = fma(X, X, p);
            }
            part[j] = p;
        }
    }
#pragma unroll
    for (int i = 0; i < 14; ++i) part[i] = wsum(part[i]);
    part[14] = wmax_nonneg(part[14]);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 15; ++i) red[16 * wave + i] = part[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 14; ++i) {
        double a = 0.0;
#pragma unroll
        for (int w = 0; w < kFeatWaves; ++w) a += red[16 * w + i];
        part[i] = a;
    }
    {
        double m = 0.0;
#pragma unroll
        for (int w = 0; w < kFeatWaves; ++w) m = fmax(m, red[16 * w + 14]);
        part[14] = m;
    }
    __syncthreads();
    double sP = part[10];
#pragma unroll
    for (int j = 0; j < 10; ++j) sP += part[j];
    if (P.mode == 2) {
        // ---- chromagram row (:356-359): pitch class by pitch class, one wave per class, all lanes on the class's bins
        for (int cls = wave; cls < 12; cls += kFeatWaves) {
            const int b = tb.ch_start[cls], e = tb.ch_start[cls + 1];
            double acc = 0.0;
            for (int i = b + lane; i < e; i += 64) { const double xv = cur[tb.ch_src[i]]; acc = fma(xv * xv, tb.ch_w[i], acc); }
            acc = wsum(acc);
            if (lane == 0) oc[(long long)fr.t * 12 + cls] = (sP == 0.0) ? acc / kEps : fast_div(acc, sP);
        }
        return;
    }
    const double f0 = P.fs / (2.0 * (double)Nf);
    const double sX = part[11], mx = part[14];
    const double sIX = part[13] * f0;
    const double sXe = sX + (double)Nf * kEps;                // np.sum(X + eps) (:118-119)
    const double sXp = part[12] + (double)Nf * kEps;
    // spectral entropy (:85-107)
    double ent_f = 0.0;
#pragma unroll
    for (int j = 0; j < 10; ++j) {
        const double s = fast_div(part[j], sP + kEps);
        ent_f -= s * fast_log2(s + kEps);
    }
    // ---- sweep B: spread and flux (:57-82, :110-124)
    const double r = (mx == 0.0) ? 1.0 / kEps : fast_div(1.0, mx);
    const double den = sX * r + kEps;
    const double cen = fast_div(sIX * r, den);
    const double rX = fast_div(1.0, sXe), rXp = fast_div(1.0, sXp);
    double sSp = 0.0, sFl = 0.0;
#pragma unroll 4
    for (int k = tid; k < ((kAblate & 32) ? 0 : Nf); k += kFeatThreads) {
        const double X = cur[k];
        const double dv = (double)(k + 1) * f0 - cen;
        sSp = fma(dv * dv, X * r, sSp);
        const double df = X * rX - prv[k] * rXp;
        sFl = fma(df, df, sFl);
    }
    sSp = bsum8(sSp, slot, lane, wave);
    sFl = bsum8(sFl, slot, lane, wave);
    const double spread = fast_sqrt(fast_div(sSp, den));
    // ---- roll-off (:127-140): first k with cumsum(X^2)[k] + eps > 0.9 sum(X^2); contiguous chunks of ODD length (a lane's
    // chunk starts an odd number of doubles after its neighbour's: no LDS bank is hit twice), workgroup-wide scan
    int first = 0x7fffffff;
    {
        const double thr = 0.90 * sP;
        const int cch = ((Nf + kFeatThreads - 1) / kFeatThreads) | 1;
        const int kb = min(tid * cch, Nf), ke = (kAblate & 64) ? kb : min(Nf, kb + cch);
        double cs = 0.0;
#pragma unroll 8
        for (int k = kb; k < ke; ++k) { const double X = cur[k]; cs = fma(X, X, cs); }
        const double incl = wscan_incl(cs);
        if (lane == 63) slot[wave] = incl;
        __syncthreads();
        double run = incl - cs;
#pragma unroll
        for (int w = 0; w < kFeatWaves; ++w) run += (w < wave) ? slot[w] : 0.0;
        for (int k = kb; k < ke; ++k) {
            const double X = cur[k];
            run = fma(X, X, run);
            if (run + kEps > thr) { first = k; break; }
        }
        first = wmin_i(first);
        if (lane == 0) redi[wave] = first;
        __syncthreads();
        first = redi[0];
#pragma unroll
        for (int w = 1; w < kFeatWaves; ++w) first = min(first, redi[w]);
    }
    // ---- MFCC (:236-254): wave w owns the filters w, w + 8, ..., all 64 lanes on a filter's bins -- the five filters of a wave
    // are walked TOGETHER (ten loads in flight per lane and step, five independent wave reductions, the five logarithms on
    // five lanes at once): filter after filter, every step waited for one weight from L2
    {
        constexpr int NFW = 40 / kFeatWaves;
        int lo[NFW], cnt[NFW];
        const double *wv[NFW];
        double a[NFW];
        int maxc = 0;
#pragma unroll
        for (int j = 0; j < NFW; ++j) {
            const int m = wave + kFeatWaves * j;
            lo[j] = tb.mel_lo[m]; cnt[j] = (kAblate & 128) ? 0 : tb.mel_cnt[m]; wv[j] = tb.mel_w + tb.mel_off[m];
            a[j] = 0.0;
            maxc = max(maxc, cnt[j]);
        }
        for (int i = lane; i < maxc; i += 64) {
#pragma unroll
            for (int j = 0; j < NFW; ++j)
                if (i < cnt[j]) a[j] = fma(cur[lo[j] + i], wv[j][i], a[j]);
        }
#pragma unroll
        for (int j = 0; j < NFW; ++j) a[j] = wsum(a[j]);
        double mine = a[0];
#pragma unroll
        for (int j = 1; j < NFW; ++j) mine = (lane == j) ? a[j] : mine;
        if (lane < NFW) msp[wave + kFeatWaves * lane] = fast_log10(mine + kEps);
    }
    // ---- chroma (:277-321): the pitch classes w and w + 8 of a wave, walked together
    {
        const int c0 = wave, c1 = wave + kFeatWaves;
        const bool two = c1 < 12;
        const int b0 = tb.ch_start[c0], e0 = (kAblate & 128) ? b0 : tb.ch_start[c0 + 1];
        const int b1 = two ? tb.ch_start[c1] : 0, e1 = (two && !(kAblate & 128)) ? tb.ch_start[c1 + 1] : 0;
        double acc0 = 0.0, acc1 = 0.0;
        const int n0 = e0 - b0, n1 = e1 - b1, nmax = max(n0, n1);
#pragma unroll 2
        for (int i = lane; i < nmax; i += 64) {
            if (i < n0) { const double xv = cur[tb.ch_src[b0 + i]]; acc0 = fma(xv * xv, tb.ch_w[b0 + i], acc0); }
            if (i < n1) { const double xv = cur[tb.ch_src[b1 + i]]; acc1 = fma(xv * xv, tb.ch_w[b1 + i], acc1); }
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
        fv[7] = (first == 0x7fffffff) ? 0.0 : (double)first / (double)Nf;
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

// rows 34..67 of every clip (ShortTermFeatures.py:668-680): blockIdx.y = clip
__global__ __launch_bounds__(256) void wg_delta_kernel(const ClipDev *__restrict__ clips, double *__restrict__ out) {
    const ClipDev c = clips[blockIdx.y];
    const long long Tc = c.T, idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)kBase * Tc) return;
    double *oc = out + c.out_off;
    const long long rowi = idx / Tc, t = idx % Tc;
    const double *rp = oc + rowi * Tc;
    oc[(kBase + rowi) * Tc + t] = (t == 0) ? 0.0 : rp[t] - rp[t - 1];
}

// ---- host: does the window fit, radix schedule, permutation ----------------------------------------------------------------
// fills the schedule of a transform of `n` elements whose twiddles come from the table of `Nc` (n = Nc, or one sub-transform)
inline void wg_schedule(int Nc, int n, const std::vector<int> &radix, WgLayout &L, std::vector<unsigned short> &perm) {
    L.top = n / radix[0];
    L.magic_top = (unsigned)(((1ULL << 32) + (unsigned)L.top - 1) / (unsigned)L.top);
    L.n_pass = (int)radix.size();
#ifndef PAA_WG_NT
#define PAA_WG_NT 0                 // timing builds of scripts/rounds/r05 only: force the workgroup size of the spectrum kernel
#endif
    L.threads = PAA_WG_NT ? PAA_WG_NT : 768;
    for (int r : radix) if (r > 8) L.threads = 512;
    int M = n;
    for (int p = 0; p < L.n_pass; ++p) {
        L.radix[p] = radix[p];
        L.span[p] = M;
        const unsigned stride = (unsigned)(M / radix[p]);
        L.magic[p] = stride > 1 ? (unsigned)((1ULL << 32) / stride) + 1u : 0u;
        L.tws[p] = Nc / M;
        M /= radix[p];
    }
    mix::mix_permutation(n, radix, perm);
    for (auto &pp : perm) pp = (unsigned short)(pp + pp / L.top);          // positions in the padded buffer
    perm.push_back(0);          // (the LDS copy moves whole 32-bit words)
}

// 1: the transform of this window runs in one workgroup's LDS -- whole (L.r0 = 0) or as r0 sub-transforms (L.r0 > 0); fills L and
// perm.  0: neither (a length with a prime factor above 13, or more than 32 768 points): kernels_big.hpp keeps it
inline int wg_layout(const FftPlan &fft, WgLayout &L, std::vector<unsigned short> &perm) {
    const int Nc = fft.len, Nf = fft.window / 2;
    if (Nc < 256 || Nc > 32768) return 0;
    std::vector<int> radix;
    if (!mix::mix_factor(Nc, radix)) return 0;
    memset(&L, 0, sizeof(L));
    auto up16 = [](size_t b) { return (b + 15) / 16 * 16; };
    constexpr size_t kLds = 160 * 1024;
    L.n_twhi = (Nc + kTwLo - 1) / kTwLo;
    if (L.n_twhi > 256) return 0;
    // the feature kernel: the row staged in LDS when it fits
    const size_t feat_small = up16((size_t)(48 + 40 + kFeatWaves * 16 + kFeatWaves) * 8 + kFeatWaves * 4);
    L.feat_staged = ((size_t)Nf * 8 + feat_small <= kLds) ? 1 : 0;
    L.feat_lds_bytes = (int)(feat_small + (L.feat_staged ? up16((size_t)Nf * 8) : 0));
    auto place = [&](size_t buf_elems, int n_perm) -> bool {
        size_t off = up16(buf_elems * 16);
        L.off_red = (int)off; off += up16((size_t)16 * 5 * 8);
        L.off_twlo = (int)off; off += (size_t)kTwLo * 16;
        L.off_twhi = (int)off; off += (size_t)L.n_twhi * 16;
        L.off_cv = (int)off; off += 8 * 16;
        L.off_perm = (int)off;
        L.perm_lds = (off + up16((size_t)n_perm * 2 + 4) <= kLds) ? 1 : 0;
        if (L.perm_lds) off += up16((size_t)n_perm * 2 + 4);
        L.lds_bytes = (int)off;
        return off <= kLds;
    };
    if (place((size_t)(Nc + radix[0]), Nc)) {
        wg_schedule(Nc, Nc, radix, L, perm);
        return 1;
    }
    // split: the first pass (radix r0 <= 8, any divisor) straight from the samples, two sub-transforms side by side in LDS;
    // fewest passes + a quarter pass per point of r0 (the first pass costs r0 complex multiply-adds per output)
    int best = 0;
    double best_cost = 1e30;
    for (int r0 = 2; r0 <= 8; ++r0) {
        if (Nc % r0) continue;
        std::vector<int> rs;
        const int S = Nc / r0;
        if (S < 256 || !mix::mix_factor(S, rs)) continue;
        if (!place((size_t)2 * (S + rs[0]), S)) continue;
        const double cost = (double)rs.size() + 0.25 * r0;
        if (cost < best_cost) { best_cost = cost; best = r0; }
    }
    if (!best) return 0;
    std::vector<int> rs;
    const int S = Nc / best;
    mix::mix_factor(S, rs);
    place((size_t)2 * (S + rs[0]), S);
    wg_schedule(Nc, S, rs, L, perm);
    L.r0 = best; L.sub = S;
    return 1;
}

}  // namespace wg
}  // namespace paa
