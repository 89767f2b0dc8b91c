// Truncated tail frames of chromagram(): the reference slices signal[p:p+window] past the end of the
// clip and FFTs whatever is left (ShortTermFeatures.py:349-355).  Those (few) frames are evaluated as a
// direct DFT of their true length L: X[k] = |sum_n y[n] exp(-2 pi i n k / L)| / num_fft, k < num_fft.
#pragma once
#include "kernels_generic.hpp"

namespace paa {

template <typename T>
__global__ __launch_bounds__(256) void chroma_tail_kernel(PlanDev P, const T *__restrict__ sig, long long pos0,
                                                           long long n_total, const ClipNorm *__restrict__ norms,
                                                           double *__restrict__ out, double *spill) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // the frame's spectrum: in LDS, or -- more than 20 000 bins -- in the caller's scratch (one row per tail frame)
    double *spec = spill ? spill + (long long)blockIdx.x * P.Nf : reinterpret_cast<double *>(smem);
    __shared__ double red[4];
    const long long pos = pos0 + (long long)blockIdx.x * P.S;
    const int L = (int)(n_total - pos);
    const ClipNorm nm = norms[0];
    const T *x = sig + pos;
    const double sc = sample_scale<T>();
    double p = 0.0;
    for (int k = threadIdx.x; k < P.Nf; k += 256) {
        double re = 0.0, im = 0.0;
        int m = 0;                                     // (n * k) mod L, kept exact
        for (int n = 0; n < L; ++n) {
            const double y = fma(load_sample<T>(x + n), sc, -nm.mean) * nm.inv;
            double s, c;
            sincospi(-2.0 * (double)m / (double)L, &s, &c);
            re = fma(y, c, re);
            im = fma(y, s, im);
            m += k;
            if (m >= L) m -= L;
        }
        const double X = sqrt(fma(re, re, im * im)) / (double)P.Nf;
        spec[k] = X;
        p = fma(X, X, p);
    }
    p = wave_sum(p);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = p;
    __syncthreads();
    const double sP = (red[0] + red[1]) + (red[2] + red[3]);
    if (threadIdx.x < 64) {
        const double ch = chroma_class(tabs_global(P), spec, sP, threadIdx.x);
        if (threadIdx.x < 12) out[(long long)blockIdx.x * 12 + threadIdx.x] = ch;
    }
}

// bytes of scratch launch_chroma_tail needs for `count` tail frames (0: their spectra fit the LDS)
inline size_t chroma_tail_spill_bytes(const PlanDev &P, int count) {
    return ((size_t)P.Nf * 8 + 16 > (size_t)160 * 1024) ? (size_t)count * (size_t)P.Nf * 8 : 0;
}
inline int launch_chroma_tail(const PlanDev &P, int sample_kind, const void *d_sig, long long pos0, long long n_total,
                              int count, const ClipNorm *norms, double *d_out, double *spill, hipStream_t stream) {
    size_t lds = (size_t)P.Nf * 8 + 16;
    if (lds > 160 * 1024) {
        if (!spill) return -2;
        lds = 16;
    } else {
        spill = nullptr;
    }
    if (lds > 64 * 1024) {
        const void *fn = sample_kind == 0 ? reinterpret_cast<const void *>(&chroma_tail_kernel<int16_t>)
                       : sample_kind == 2 ? reinterpret_cast<const void *>(&chroma_tail_kernel<stereo16>)
                                          : reinterpret_cast<const void *>(&chroma_tail_kernel<double>);
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
    }
    if (sample_kind == 0)
        hipLaunchKernelGGL(chroma_tail_kernel<int16_t>, dim3(count), dim3(256), lds, stream, P, (const int16_t *)d_sig,
                           pos0, n_total, norms, d_out, spill);
    else if (sample_kind == 2)        // interleaved stereo int16, summed in the loads (fused stereo_to_mono)
        hipLaunchKernelGGL(chroma_tail_kernel<stereo16>, dim3(count), dim3(256), lds, stream, P, (const stereo16 *)d_sig,
                           pos0, n_total, norms, d_out, spill);
    else
        hipLaunchKernelGGL(chroma_tail_kernel<double>, dim3(count), dim3(256), lds, stream, P, (const double *)d_sig,
                           pos0, n_total, norms, d_out, spill);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace paa
