// Batched onset probability of audioSegmentation.silence_removal (audioSegmentation.py:744-748): the reference calls
// svm.predict_proba() once per short-term frame on a scikit-learn SVC it trained a few lines earlier (:739).  This
// kernel evaluates that binary probabilistic SVC for every frame at once: standardise the frame's feature vector
// ((x - mean) / scale, :746), libsvm decision value from the support vectors (linear or RBF kernel), Platt sigmoid
// (sigmoid_predict), clip to [1e-7, 1 - 1e-7] and libsvm's multiclass_probability iteration for two classes -- the
// arithmetic of sklearn/svm/src/libsvm/svm.cpp (a third-party dependency of the reference, scikit-learn >= 0.24 per
// requirements.txt; its published algorithm is restated here, and the parity test runs the installed scikit-learn).
// Training stays with scikit-learn: the caller passes the trained model's arrays.
#pragma once
#include "device_common.hpp"

namespace paa {

// libsvm's multiclass_probability for k = 2 (pairwise r01 = p, r10 = 1 - p); returns the probability of class index 1.
// Kept operation by operation (no FMA contraction) so that the early-exit iteration follows libsvm's path.
#pragma clang fp contract(off)
__device__ __forceinline__ double libsvm_two_class_prob1(double r01) {
    const double r10 = 1.0 - r01;
    double Q00 = r10 * r10, Q11 = r01 * r01, Q01 = -r10 * r01;
    double p0 = 0.5, p1 = 0.5;
    const double eps = 0.005 / 2.0;
    for (int iter = 0; iter < 100; ++iter) {
        double Qp0 = 0.0, Qp1 = 0.0, pQp = 0.0;
        Qp0 += Q00 * p0; Qp0 += Q01 * p1;
        pQp += p0 * Qp0;
        Qp1 += Q01 * p0; Qp1 += Q11 * p1;
        pQp += p1 * Qp1;
        const double e0 = fabs(Qp0 - pQp), e1 = fabs(Qp1 - pQp);
        const double max_error = e1 > e0 ? e1 : e0;
        if (max_error < eps) break;
        {   // t = 0
            const double diff = (-Qp0 + pQp) / Q00;
            p0 += diff;
            pQp = (pQp + diff * (diff * Q00 + 2 * Qp0)) / (1 + diff) / (1 + diff);
            Qp0 = (Qp0 + diff * Q00) / (1 + diff);
            p0 /= (1 + diff);
            Qp1 = (Qp1 + diff * Q01) / (1 + diff);
            p1 /= (1 + diff);
        }
        {   // t = 1
            const double diff = (-Qp1 + pQp) / Q11;
            p1 += diff;
            pQp = (pQp + diff * (diff * Q11 + 2 * Qp1)) / (1 + diff) / (1 + diff);
            Qp0 = (Qp0 + diff * Q01) / (1 + diff);
            p0 /= (1 + diff);
            Qp1 = (Qp1 + diff * Q11) / (1 + diff);
            p1 /= (1 + diff);
        }
    }
    return p1;
}
#pragma clang fp contract(fast)

constexpr int kSvmMaxDims = 72;      // short-term feature vectors have 34 or 68 rows

// one thread per frame: feats is [n_dims][ld] (feature-major like the short-term matrix), frame t in column t
__global__ __launch_bounds__(256) void svm_binary_proba_kernel(const double *__restrict__ feats, int n_dims, long long ld,
                                                               long long n_frames, const double *__restrict__ mean,
                                                               const double *__restrict__ scale,
                                                               const double *__restrict__ sv, const double *__restrict__ coef,
                                                               int n_sv, double intercept, double gamma, double prob_a,
                                                               double prob_b, double *__restrict__ prob1) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_frames) return;
    double x[kSvmMaxDims];
#pragma unroll
    for (int d = 0; d < kSvmMaxDims; ++d) x[d] = (d < n_dims) ? (feats[(long long)d * ld + t] - mean[d]) / scale[d] : 0.0;
    double dec = 0.0;
    for (int i = 0; i < n_sv; ++i) {          // support vectors and coefficients are wave-uniform (scalar loads)
        const double *s = sv + (long long)i * n_dims;
        double k = 0.0;
        if (gamma > 0.0) {                    // RBF: exp(-gamma |sv - x|^2)
#pragma unroll
            for (int d = 0; d < kSvmMaxDims; ++d)
                if (d < n_dims) { const double df = s[d] - x[d]; k = fma(df, df, k); }
            k = exp(-gamma * k);
        } else {                              // linear: <sv, x>
#pragma unroll
            for (int d = 0; d < kSvmMaxDims; ++d)
                if (d < n_dims) k = fma(s[d], x[d], k);
        }
        dec = fma(coef[i], k, dec);
    }
    dec += intercept;                         // = sklearn's decision_function; libsvm's own decision value is -dec
    const double fApB = (-dec) * prob_a + prob_b;
    // sigmoid_predict, the numerically stable form of 1 / (1 + exp(fApB))
    double p = (fApB >= 0.0) ? exp(-fApB) / (1.0 + exp(-fApB)) : 1.0 / (1.0 + exp(fApB));
    p = fmin(fmax(p, 1e-7), 1.0 - 1e-7);
    prob1[t] = libsvm_two_class_prob1(p);
}

}  // namespace paa
