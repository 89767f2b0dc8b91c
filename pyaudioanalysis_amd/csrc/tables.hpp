// Host-side, frame-invariant tables, built by the reference's own rules (bug-compatible).
// Everything here runs once per (fs, window) and is cached by the plan.
//
//   mel bank ........ ShortTermFeatures.py:204-231  (bin axis k*fs/num_fft although the
//                     spectrum spans 0..fs/2 -- quirk kept)
//   DCT-II ortho .... scipy dct(type=2, norm='ortho')[:13] at ShortTermFeatures.py:253
//   chroma .......... ShortTermFeatures.py:257-302 (np.round half-even, negative slots wrap,
//                     last writer wins, divisor indexed by slot position)
//   FFT plan ........ replaces scipy.fftpack.fft at ShortTermFeatures.py:617
#pragma once
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <vector>

#include "../../include/paa_hip.h"

namespace paa {

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per device: the launchers' "already raised to N bytes" caches carry the
// generation they were filled in, and paa_shutdown starts a new one (a later paa_init may select another device)
inline std::atomic<int> &lds_attr_generation() {
    static std::atomic<int> g{0};
    return g;
}
struct LdsAttrCache {
    size_t bytes = 0;
    int generation = -1;
    bool covers(size_t need) const { return generation == lds_attr_generation().load() && bytes >= need; }
    void set(size_t b) { bytes = b; generation = lds_attr_generation().load(); }
};

// Experiment switches of the A/B scripts under scripts/ (PAA_KERNEL_DEBUG, PAA_RUN_CAP, PAA_NO_MIX, PAA_F800_WAVES,
// PAA_F800_PACE, PAA_MIX_*, PAA_HIP_FORCE_GENERIC) exist only in builds with -DPAA_EXPERIMENTS
// (PAA_HIPCC_FLAGS=-DPAA_EXPERIMENTS python -m pyaudioanalysis_amd._build): in the default build no environment variable
// can change which kernel runs or what it stores.
inline const char *experiment_env(const char *name) {
#ifdef PAA_EXPERIMENTS
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

constexpr int kNumMel = 40;
constexpr int kNumMfcc = 13;
constexpr int kNumLin = 13;
constexpr int kNumLog = 27;

struct MelTable {
    std::vector<int32_t> lo;      // first bin of filter m
    std::vector<int32_t> cnt;     // number of consecutive bins
    std::vector<int32_t> off;     // offset of its weights in w
    std::vector<double> w;
};

// the 42 frequency points of the triangles (ShortTermFeatures.py:207-211)
inline void mel_edges(double *edges) {
    for (int i = 0; i < kNumLin; ++i) edges[i] = 133.33 + (double)i * (200.0 / 3.0);
    for (int i = kNumLin; i < kNumMel + 2; ++i)
        edges[i] = edges[kNumLin - 1] * std::pow(1.0711703, (double)(i - kNumLin + 1));
}

// returns PAA_OK or PAA_ERR_MEL_INDEX (the reference indexes fbank[i][lid] out of range)
inline int build_mel(double fs, int nfft, MelTable &t) {
    double edges[kNumMel + 2];
    mel_edges(edges);
    t.lo.assign(kNumMel, 0);
    t.cnt.assign(kNumMel, 0);
    t.off.assign(kNumMel, 0);
    t.w.clear();
    for (int m = 0; m < kNumMel; ++m) {
        const double lo = edges[m], mid = edges[m + 1], hi = edges[m + 2];
        const double peak = 2.0 / (hi - lo);
        const long k_lo = (long)std::floor(lo * nfft / fs) + 1;
        const long k_mid = (long)std::floor(mid * nfft / fs) + 1;
        const long k_hi = (long)std::floor(hi * nfft / fs) + 1;
        if (k_hi > k_lo && k_hi - 1 >= nfft) return PAA_ERR_MEL_INDEX;
        const double up = peak / (mid - lo), dn = peak / (hi - mid);
        t.lo[m] = (int32_t)k_lo;
        t.off[m] = (int32_t)t.w.size();
        for (long k = k_lo; k < k_mid; ++k) {
            const double f = (double)k / (1.0 * nfft) * fs;
            t.w.push_back(up * (f - lo));
        }
        for (long k = k_mid; k < k_hi; ++k) {
            const double f = (double)k / (1.0 * nfft) * fs;
            t.w.push_back(dn * (hi - f));
        }
        t.cnt[m] = (int32_t)(t.w.size() - (size_t)t.off[m]);
    }
    return PAA_OK;
}

inline void build_dct(double *m13x40) {
    const double s = std::sqrt(2.0 / kNumMel);
    for (int k = 0; k < kNumMfcc; ++k)
        for (int n = 0; n < kNumMel; ++n) {
            double v = s * std::cos(M_PI * k * (2 * n + 1) / (2.0 * kNumMel));
            if (k == 0) v *= 1.0 / std::sqrt(2.0);
            m13x40[k * kNumMel + n] = v;
        }
}

struct ChromaTable {
    // entries grouped by pitch class (slot % 12), ascending slot inside a class: this is the
    // order in which np.sum(C2, axis=0) adds the rows of the (rows, 12) fold (:299-302)
    int32_t class_start[13];
    std::vector<int32_t> src;
    std::vector<double> w;
    // flat ascending-slot list (tests)
    std::vector<int32_t> flat_src, flat_slot;
    std::vector<double> flat_w;
};

inline int build_chroma(double fs, int nfft, ChromaTable &t) {
    std::vector<long> slot(nfft);
    long smax = -(1L << 40), smin = (1L << 40);
    for (int f = 0; f < nfft; ++f) {
        const double freq = ((double)(f + 1) * fs) / (double)(2 * nfft);
        slot[f] = (long)std::nearbyint(12.0 * std::log2(freq / 27.50));   // half-to-even like np.round
        if (slot[f] > smax) smax = slot[f];
        if (slot[f] < smin) smin = slot[f];
    }
    if (smax >= nfft) {
        bool any_over = false;
        for (int f = 0; f < nfft; ++f) any_over |= slot[f] > nfft;
        return any_over ? PAA_ERR_CHROMA_VALUE : PAA_ERR_CHROMA_INDEX;
    }
    if (smin < -(long)nfft) return PAA_ERR_CHROMA_INDEX;
    // number of bins sharing a bin's slot value (:267-272); slots are non-decreasing in f, so equal values
    // form runs (the quadratic count is kept for the impossible non-monotone case)
    std::vector<double> count(nfft, 0.0);
    bool monotone = true;
    for (int f = 1; f < nfft; ++f) monotone &= slot[f] >= slot[f - 1];
    if (monotone) {
        for (int f = 0; f < nfft;) {
            int e = f;
            while (e < nfft && slot[e] == slot[f]) ++e;
            for (int g = f; g < e; ++g) count[g] = (double)(e - f);
            f = e;
        }
    } else {
        for (int f = 0; f < nfft; ++f) {
            long c = 0;
            for (int g = 0; g < nfft; ++g) c += (slot[g] == slot[f]);
            count[f] = (double)c;
        }
    }
    std::vector<int32_t> owner(nfft, -1);
    auto wrap = [nfft](long s) { return (int)(s < 0 ? s + nfft : s); };
    for (int f = 0; f < nfft; ++f) owner[wrap(slot[f])] = f;   // later (higher) bins overwrite
    t.flat_src.clear(); t.flat_slot.clear(); t.flat_w.clear();
    for (int p = 0; p < nfft; ++p) {
        if (owner[p] < 0) continue;
        t.flat_src.push_back(owner[p]);
        t.flat_slot.push_back(p);
        t.flat_w.push_back(1.0 / count[wrap(slot[p])]);      // C /= count[slot] is position-indexed
    }
    t.src.clear(); t.w.clear();
    for (int c = 0; c < 12; ++c) {
        t.class_start[c] = (int32_t)t.src.size();
        for (size_t e = 0; e < t.flat_src.size(); ++e)
            if (t.flat_slot[e] % 12 == c) {
                t.src.push_back(t.flat_src[e]);
                t.w.push_back(t.flat_w[e]);
            }
    }
    t.class_start[12] = (int32_t)t.src.size();
    return PAA_OK;
}

// ---- FFT plan --------------------------------------------------------------------------
struct FftPlan {
    int window = 0;
    int even = 0;            // 1: real-input trick, complex length window/2
    int len = 0;             // complex FFT length
    std::vector<int32_t> radix;
    std::vector<double> tw;      // len complex: exp(-2 pi i j / len)
    std::vector<double> post;    // even: window/2 complex exp(-2 pi i k / window)
};

inline void build_fft_plan(int window, FftPlan &p) {
    p.window = window;
    p.even = (window % 2 == 0) ? 1 : 0;
    p.len = p.even ? window / 2 : window;
    p.radix.clear();
    int n = p.len;
    // any order is a valid Stockham schedule; small hard-coded radices first, the generic
    // O(R^2) prime passes last
    while (n % 4 == 0) { p.radix.push_back(4); n /= 4; }
    while (n % 2 == 0) { p.radix.push_back(2); n /= 2; }
    while (n % 3 == 0) { p.radix.push_back(3); n /= 3; }
    while (n % 5 == 0) { p.radix.push_back(5); n /= 5; }
    for (int f = 7; n > 1; f += 2) {
        if ((long)f * f > n) f = n;              // what is left is prime
        while (n % f == 0) { p.radix.push_back(f); n /= f; }
    }
    const long double two_pi = 6.283185307179586476925286766559005768L;
    p.tw.resize(2 * (size_t)p.len);
    for (int j = 0; j < p.len; ++j) {
        const long double a = -two_pi * (long double)j / (long double)p.len;
        p.tw[2 * j] = (double)cosl(a);
        p.tw[2 * j + 1] = (double)sinl(a);
    }
    p.post.clear();
    if (p.even) {
        p.post.resize(2 * (size_t)p.len);
        for (int k = 0; k < p.len; ++k) {
            const long double a = -two_pi * (long double)k / (long double)window;
            p.post[2 * k] = (double)cosl(a);
            p.post[2 * k + 1] = (double)sinl(a);
        }
    }
}

}  // namespace paa
