// FP64 matrix-core micro-benchmark (round 4, for the round-5 candidate "prime-factor passes as v_mfma_f64 GEMMs"):
//   1. how many cycles does a SIMD need per v_mfma_f64_16x16x4_f64 (back to back, four independent accumulators),
//      with one and with two waves on the SIMD?
//   2. does FP64 VECTOR work overlap with it -- inside one wave (F v_fma_f64 placed after every MFMA), and between two
//      waves of one SIMD (one wave issues only MFMAs, the other only v_fma_f64)?
// If the FP64 matrix pipe were the vector FP64 datapath under another name, (2) would show the SUM of both streams; if it
// is a pipe of its own the MAX.  Every wave records where it ran (HW_REG_HW_ID) and its own s_memtime interval, as mb2 does.
// Build: hipcc --offload-arch=gfx950 -O3 scripts/microbench/mb5_mfma_f64.hip -o scripts/microbench/mb5.out
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <map>
#include <vector>

typedef double double4_t __attribute__((ext_vector_type(4)));

struct Rec { unsigned hw_id, xcc, role, pad; unsigned long long t0, t1; };

__device__ __forceinline__ unsigned long long now() { return __builtin_readcyclecounter(); }
__device__ __forceinline__ unsigned hw_id() { return __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4); }     // HW_REG_HW_ID
__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20); }   // HW_REG_XCC_ID

#define FMA1(r) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(r) : "v"(fb), "v"(fc));

// F v_fma_f64 on eight independent chains (F <= 16)
template <int F>
__device__ __forceinline__ void fmas(double (&f)[8], double fb, double fc) {
#pragma unroll
    for (int i = 0; i < F; ++i) FMA1(f[i & 7])
}

// per iteration: 16 x (one MFMA on accumulator (i & 3), then F vector FMAs); the order is pinned with sched_barrier
template <int F>
__device__ __forceinline__ void body_mfma(double4_t (&acc)[4], double a, double b, double (&f)[8], double fb, double fc) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        acc[i & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i & 3], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (F > 0) {
            fmas<F>(f, fb, fc);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// role of a wave: 0 = MFMA stream with F FMAs after every MFMA, 1 = FMA-only stream (64 per iteration)
// split = 0: every wave has role 0; split = 1: the second half of the workgroup's waves (the second slot of every SIMD)
// has role 1 and runs n_it_fma iterations
template <int F>
__global__ void k_mix(Rec *rec, double seed, int n_it, int n_it_fma, int split) {
    const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const unsigned role = (split == 1 && wave >= nw / 2) ? 1u : (split == 2 ? 1u : 0u);
    double4_t acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = double4_t{seed, seed + i, seed + 2.0 * i, seed - i};
    double f[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = seed + i;
    const double a = 1.0 + 1e-9 * (threadIdx.x & 15), b = 1e-9 * (threadIdx.x >> 4), fb = 1.0000001, fc = 1e-9;
    const unsigned long long t0 = now();
    if (role == 0) {
#pragma unroll 1
        for (int it = 0; it < n_it; ++it) body_mfma<F>(acc, a, b, f, fb, fc);
    } else {
#pragma unroll 1
        for (int it = 0; it < n_it_fma; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r) fmas<16>(f, fb, fc);
        }
    }
    const unsigned long long t1 = now();
    if ((threadIdx.x & 63) == 0) {
        Rec r; r.hw_id = hw_id(); r.xcc = xcc_id(); r.role = role; r.pad = 0; r.t0 = t0; r.t1 = t1;
        rec[blockIdx.x * nw + wave] = r;
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += f[i];
    if (s == 12345.678) rec[0].pad = 1;
}

template <int F>
static void run(const char *name, Rec *d_rec, int waves_per_simd, int split, int n_it, int n_it_fma) {
    const int threads = 256 * waves_per_simd, blocks = 256, nw = threads / 64;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0.f;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k_mix<F>), dim3(blocks), dim3(threads), 0, 0, d_rec, 1.5, n_it, n_it_fma, split);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1);
    }
    std::vector<Rec> h((size_t)blocks * nw);
    hipMemcpy(h.data(), d_rec, h.size() * sizeof(Rec), hipMemcpyDeviceToHost);
    // per role: cycles per instruction as the wave sees it; SIMDs whose wave set is not the expected one are counted
    double cyc[2] = {0, 0}; long cnt[2] = {0, 0}; int bad = 0, simds = 0;
    std::map<unsigned long long, std::vector<unsigned>> by_simd;
    for (size_t i = 0; i < h.size(); ++i) {
        const Rec &r = h[i];
        const double insts = r.role == 0 ? 16.0 * n_it : 64.0 * n_it_fma;
        cyc[r.role] += (double)(r.t1 - r.t0) / insts; ++cnt[r.role];
        const unsigned simd = (r.hw_id >> 4) & 3, cu = (r.hw_id >> 8) & 15, sh = (r.hw_id >> 12) & 1, se = (r.hw_id >> 13) & 7;
        by_simd[((unsigned long long)(i / nw) << 32) | (r.xcc & 15) << 16 | se << 12 | sh << 8 | cu << 4 | simd].push_back(r.role);
    }
    for (auto &kv : by_simd) {
        ++simds;
        const int n1 = (int)std::count(kv.second.begin(), kv.second.end(), 1u);
        if ((int)kv.second.size() != waves_per_simd || (split == 1 && n1 * 2 != waves_per_simd)) ++bad;
    }
    printf("%-58s %d w/SIMD  kernel %8.3f ms |", name, waves_per_simd, ms);
    if (cnt[0]) printf(" MFMA wave: %7.2f cyc per MFMA%s", cyc[0] / cnt[0], F ? " (+F FMAs)" : "");
    if (cnt[1]) printf(" | FMA wave: %6.2f cyc per v_fma_f64", cyc[1] / cnt[1]);
    printf("   [SIMDs %d, unexpected wave sets %d]\n", simds, bad);
    hipEventDestroy(e0); hipEventDestroy(e1);
}

int main() {
    Rec *d_rec;
    hipMalloc(&d_rec, 256 * 16 * sizeof(Rec));
    const int N = 2048;
    printf("s_memtime ticks are shader cycles; 16 MFMAs (4 accumulators) or 64 v_fma_f64 (8 chains) per iteration\n");
    run<0>("MFMA only", d_rec, 1, 0, N, 0);
    run<0>("MFMA only", d_rec, 2, 0, N, 0);
    run<0>("FMA only", d_rec, 1, 2, 0, 4 * N);
    run<0>("FMA only", d_rec, 2, 2, 0, 4 * N);
    run<0>("split: slot 0 MFMA only, slot 1 FMA only (equal length)", d_rec, 2, 1, N, (int)(N * 16.0 * 64.0 / (64.0 * 4.45)));
    run<0>("split: slot 0 MFMA only, slot 1 FMA only (FMA shorter)", d_rec, 2, 1, N, 2 * N);
    run<4>("one wave: every MFMA followed by 4 v_fma_f64", d_rec, 1, 0, N, 0);
    run<8>("one wave: every MFMA followed by 8 v_fma_f64", d_rec, 1, 0, N, 0);
    run<12>("one wave: every MFMA followed by 12 v_fma_f64", d_rec, 1, 0, N, 0);
    run<16>("one wave: every MFMA followed by 16 v_fma_f64", d_rec, 1, 0, N, 0);
    run<8>("two waves: every MFMA followed by 8 v_fma_f64", d_rec, 2, 0, N, 0);
    run<16>("two waves: every MFMA followed by 16 v_fma_f64", d_rec, 2, 0, N, 0);
    printf("status: %s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
