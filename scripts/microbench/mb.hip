// Micro-benchmarks of the gfx950 facts the fast kernel's design depends on (one wave per SIMD):
// dependent / independent FP64 issue, DPP reduction chains, LDS latency, transcendental rates, FP64 MFMA rate.
// Build: hipcc --offload-arch=gfx950 -O3 scripts/microbench/mb.hip -o scripts/microbench/mb.out ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define N_IT 2048

__device__ __forceinline__ unsigned long long now() { return __builtin_readcyclecounter(); }

// result slot r of block 0 / wave 0 = cycles per instruction * 1000
#define REPORT(slot, cyc, count) if (threadIdx.x == 0 && blockIdx.x == 0) out[slot] = (double)(cyc) / (double)(count);

__global__ void k_dep_fma(double *out, double seed) {
    double a = seed + threadIdx.x, b = 1.0000001, c = 1e-9;
    unsigned long long t0 = now();
#pragma unroll 1
    for (int i = 0; i < N_IT; ++i) {
        asm volatile("v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n"
                     "v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n"
                     : "+v"(a) : "v"(b), "v"(c));
        asm volatile("v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n"
                     "v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n"
                     : "+v"(a) : "v"(b), "v"(c));
        asm volatile("v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n"
                     "v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n"
                     : "+v"(a) : "v"(b), "v"(c));
        asm volatile("v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n"
                     "v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n"
                     : "+v"(a) : "v"(b), "v"(c));
        asm volatile("v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n"
                     "v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n"
                     : "+v"(a) : "v"(b), "v"(c));
        asm volatile("v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n"
                     "v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n"
                     : "+v"(a) : "v"(b), "v"(c));
        asm volatile("v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n"
                     "v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n"
                     : "+v"(a) : "v"(b), "v"(c));
        asm volatile("v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n"
                     "v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n"
                     : "+v"(a) : "v"(b), "v"(c));
    }
    unsigned long long t1 = now();
    REPORT(0, t1 - t0, N_IT * 64)
    if (a == 12345.0) out[63] = a;
}

__global__ void k_indep_fma(double *out, double seed) {
    double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
    double b = 1.0000001, c = 1e-9;
    unsigned long long t0 = now();
#pragma unroll 1
    for (int i = 0; i < N_IT; ++i) {
        asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                     "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                     "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                     "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                     "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                     "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                     "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                     "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                     "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    }
    unsigned long long t1 = now();
    REPORT(1, t1 - t0, N_IT * 64)
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.0) out[63] = a0;
}

__global__ void k_dep_add(double *out, double seed) {
    double a = seed + threadIdx.x, c = 1e-9;
    unsigned long long t0 = now();
#pragma unroll 1
    for (int i = 0; i < N_IT; ++i) {
        asm volatile("v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n"
                     "v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n"
                     : "+v"(a) : "v"(c));
        asm volatile("v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n"
                     "v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n"
                     : "+v"(a) : "v"(c));
        asm volatile("v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n"
                     "v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n"
                     : "+v"(a) : "v"(c));
        asm volatile("v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n"
                     "v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n"
                     : "+v"(a) : "v"(c));
        asm volatile("v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n"
                     "v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n"
                     : "+v"(a) : "v"(c));
        asm volatile("v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n"
                     "v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n"
                     : "+v"(a) : "v"(c));
        asm volatile("v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n"
                     "v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n"
                     : "+v"(a) : "v"(c));
        asm volatile("v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n"
                     "v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n"
                     : "+v"(a) : "v"(c));
    }
    unsigned long long t1 = now();
    REPORT(2, t1 - t0, N_IT * 64)
    if (a == 12345.0) out[63] = a;
}

// two dependent chains interleaved
__global__ void k_dep2_fma(double *out, double seed) {
    double a = seed + threadIdx.x, a2 = seed * 2, b = 1.0000001, c = 1e-9;
    unsigned long long t0 = now();
#pragma unroll 1
    for (int i = 0; i < N_IT; ++i) {
        asm volatile("v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n"
                     "v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n"
                     : "+v"(a), "+v"(a2) : "v"(b), "v"(c));
        asm volatile("v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n"
                     "v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n"
                     : "+v"(a), "+v"(a2) : "v"(b), "v"(c));
        asm volatile("v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n"
                     "v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n"
                     : "+v"(a), "+v"(a2) : "v"(b), "v"(c));
        asm volatile("v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n"
                     "v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n"
                     : "+v"(a), "+v"(a2) : "v"(b), "v"(c));
        asm volatile("v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n"
                     "v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n"
                     : "+v"(a), "+v"(a2) : "v"(b), "v"(c));
        asm volatile("v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n"
                     "v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n"
                     : "+v"(a), "+v"(a2) : "v"(b), "v"(c));
        asm volatile("v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n"
                     "v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n"
                     : "+v"(a), "+v"(a2) : "v"(b), "v"(c));
        asm volatile("v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n"
                     "v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n"
                     : "+v"(a), "+v"(a2) : "v"(b), "v"(c));
    }
    unsigned long long t1 = now();
    REPORT(3, t1 - t0, N_IT * 64)
    if (a + a2 == 12345.0) out[63] = a;
}

// 16-lane DPP sum reduction of a double (4 steps x (2 dpp movs + add)), dependent reductions back to back
__device__ __forceinline__ double dppmov(double v, int ctrl_sel) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    switch (ctrl_sel) {
        case 0: lo = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xF, 0xF, true); break;
        case 1: lo = __builtin_amdgcn_update_dpp(0, lo, 0x4E, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0x4E, 0xF, 0xF, true); break;
        case 2: lo = __builtin_amdgcn_update_dpp(0, lo, 0x141, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0x141, 0xF, 0xF, true); break;
        default: lo = __builtin_amdgcn_update_dpp(0, lo, 0x140, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0x140, 0xF, 0xF, true); break;
    }
    return __hiloint2double(hi, lo);
}
__global__ void k_dpp_reduce(double *out, double seed) {
    double v = seed + threadIdx.x;
    unsigned long long t0 = now();
#pragma unroll 8
    for (int i = 0; i < N_IT; ++i) {
        v += dppmov(v, 0); v += dppmov(v, 1); v += dppmov(v, 2); v += dppmov(v, 3);
        v *= 0.0625;
    }
    unsigned long long t1 = now();
    REPORT(4, t1 - t0, N_IT)           // cycles per full 16-lane reduction (+1 mul)
    if (v == 12345.0) out[63] = v;
}

// LDS pointer chase: latency of a dependent ds_read_b32
__global__ void k_lds_latency(double *out, double seed) {
    __shared__ int chain[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) chain[i] = (i * 17 + 5) & 1023;
    __syncthreads();
    int p = threadIdx.x & 1023;
    unsigned long long t0 = now();
#pragma unroll 8
    for (int i = 0; i < N_IT; ++i) {
        p = chain[p]; p = chain[p]; p = chain[p]; p = chain[p];
    }
    unsigned long long t1 = now();
    REPORT(5, t1 - t0, N_IT * 4)
    if (p == 123456 + (int)seed) out[63] = p;
}

// LDS throughput: independent conflict-free ds_read_b64
__global__ void k_lds_tp(double *out, double seed) {
    __shared__ double buf[2048];
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) buf[i] = seed + i;
    __syncthreads();
    double acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
    const int l = threadIdx.x & 63;
    unsigned long long t0 = now();
#pragma unroll 8
    for (int i = 0; i < N_IT; ++i) {
        const int o = (i & 7) * 256;
        acc0 += buf[o + l]; acc1 += buf[o + 64 + l]; acc2 += buf[o + 128 + l]; acc3 += buf[o + 192 + l];
    }
    unsigned long long t1 = now();
    REPORT(6, t1 - t0, N_IT * 4)       // cycles per ds_read_b64 (+ one add each)
    if (acc0 + acc1 + acc2 + acc3 == 12345.0) out[63] = acc0;
}

__global__ void k_rsq(double *out, double seed) {
    double a0 = seed + 1, a1 = seed + 2, a2 = seed + 3, a3 = seed + 4;
    unsigned long long t0 = now();
#pragma unroll 1
    for (int i = 0; i < N_IT; ++i) {
        asm volatile("v_rsq_f64 %0, %0\n v_rsq_f64 %1, %1\n v_rsq_f64 %2, %2\n v_rsq_f64 %3, %3\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        asm volatile("v_rsq_f64 %0, %0\n v_rsq_f64 %1, %1\n v_rsq_f64 %2, %2\n v_rsq_f64 %3, %3\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        asm volatile("v_rsq_f64 %0, %0\n v_rsq_f64 %1, %1\n v_rsq_f64 %2, %2\n v_rsq_f64 %3, %3\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        asm volatile("v_rsq_f64 %0, %0\n v_rsq_f64 %1, %1\n v_rsq_f64 %2, %2\n v_rsq_f64 %3, %3\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        asm volatile("v_rsq_f64 %0, %0\n v_rsq_f64 %1, %1\n v_rsq_f64 %2, %2\n v_rsq_f64 %3, %3\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        asm volatile("v_rsq_f64 %0, %0\n v_rsq_f64 %1, %1\n v_rsq_f64 %2, %2\n v_rsq_f64 %3, %3\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        asm volatile("v_rsq_f64 %0, %0\n v_rsq_f64 %1, %1\n v_rsq_f64 %2, %2\n v_rsq_f64 %3, %3\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        asm volatile("v_rsq_f64 %0, %0\n v_rsq_f64 %1, %1\n v_rsq_f64 %2, %2\n v_rsq_f64 %3, %3\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
    }
    unsigned long long t1 = now();
    REPORT(7, t1 - t0, N_IT * 32)
    if (a0 + a1 + a2 + a3 == 12345.0) out[63] = a0;
}

// FP64 FMAs alternating with independent 32-bit integer adds: do the 4-cycle ops fill the FP64 issue gap?
__global__ void k_mix(double *out, double seed) {
    double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3;
    double b = 1.0000001, c = 1e-9;
    int i0 = threadIdx.x, i1 = 1, i2 = 2, i3 = 3, k = 7;
    unsigned long long t0 = now();
#pragma unroll 1
    for (int i = 0; i < N_IT; ++i) {
        asm volatile("v_fma_f64 %0, %0, %8, %9\n v_add_u32 %4, %4, %10\n v_fma_f64 %1, %1, %8, %9\n v_add_u32 %5, %5, %10\n"
                     "v_fma_f64 %2, %2, %8, %9\n v_add_u32 %6, %6, %10\n v_fma_f64 %3, %3, %8, %9\n v_add_u32 %7, %7, %10\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(b), "v"(c), "v"(k));
        asm volatile("v_fma_f64 %0, %0, %8, %9\n v_add_u32 %4, %4, %10\n v_fma_f64 %1, %1, %8, %9\n v_add_u32 %5, %5, %10\n"
                     "v_fma_f64 %2, %2, %8, %9\n v_add_u32 %6, %6, %10\n v_fma_f64 %3, %3, %8, %9\n v_add_u32 %7, %7, %10\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(b), "v"(c), "v"(k));
        asm volatile("v_fma_f64 %0, %0, %8, %9\n v_add_u32 %4, %4, %10\n v_fma_f64 %1, %1, %8, %9\n v_add_u32 %5, %5, %10\n"
                     "v_fma_f64 %2, %2, %8, %9\n v_add_u32 %6, %6, %10\n v_fma_f64 %3, %3, %8, %9\n v_add_u32 %7, %7, %10\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(b), "v"(c), "v"(k));
        asm volatile("v_fma_f64 %0, %0, %8, %9\n v_add_u32 %4, %4, %10\n v_fma_f64 %1, %1, %8, %9\n v_add_u32 %5, %5, %10\n"
                     "v_fma_f64 %2, %2, %8, %9\n v_add_u32 %6, %6, %10\n v_fma_f64 %3, %3, %8, %9\n v_add_u32 %7, %7, %10\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(b), "v"(c), "v"(k));
        asm volatile("v_fma_f64 %0, %0, %8, %9\n v_add_u32 %4, %4, %10\n v_fma_f64 %1, %1, %8, %9\n v_add_u32 %5, %5, %10\n"
                     "v_fma_f64 %2, %2, %8, %9\n v_add_u32 %6, %6, %10\n v_fma_f64 %3, %3, %8, %9\n v_add_u32 %7, %7, %10\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(b), "v"(c), "v"(k));
        asm volatile("v_fma_f64 %0, %0, %8, %9\n v_add_u32 %4, %4, %10\n v_fma_f64 %1, %1, %8, %9\n v_add_u32 %5, %5, %10\n"
                     "v_fma_f64 %2, %2, %8, %9\n v_add_u32 %6, %6, %10\n v_fma_f64 %3, %3, %8, %9\n v_add_u32 %7, %7, %10\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(b), "v"(c), "v"(k));
        asm volatile("v_fma_f64 %0, %0, %8, %9\n v_add_u32 %4, %4, %10\n v_fma_f64 %1, %1, %8, %9\n v_add_u32 %5, %5, %10\n"
                     "v_fma_f64 %2, %2, %8, %9\n v_add_u32 %6, %6, %10\n v_fma_f64 %3, %3, %8, %9\n v_add_u32 %7, %7, %10\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(b), "v"(c), "v"(k));
        asm volatile("v_fma_f64 %0, %0, %8, %9\n v_add_u32 %4, %4, %10\n v_fma_f64 %1, %1, %8, %9\n v_add_u32 %5, %5, %10\n"
                     "v_fma_f64 %2, %2, %8, %9\n v_add_u32 %6, %6, %10\n v_fma_f64 %3, %3, %8, %9\n v_add_u32 %7, %7, %10\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(b), "v"(c), "v"(k));
    }
    unsigned long long t1 = now();
    REPORT(10, t1 - t0, N_IT * 32)      // cycles per (fma_f64 + add_u32) pair
    if (a0 + a1 + a2 + a3 + i0 + i1 + i2 + i3 == 12345.0) out[63] = a0;
}
// 32-bit integer adds only
__global__ void k_int(double *out, double seed) {
    int i0 = threadIdx.x, i1 = 1, i2 = 2, i3 = (int)seed, k = 7;
    unsigned long long t0 = now();
#pragma unroll 1
    for (int i = 0; i < N_IT; ++i) {
        asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n"
                     "v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n"
                     : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(k));
        asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n"
                     "v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n"
                     : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(k));
        asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n"
                     "v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n"
                     : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(k));
        asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n"
                     "v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n"
                     : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(k));
        asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n"
                     "v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n"
                     : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(k));
        asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n"
                     "v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n"
                     : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(k));
        asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n"
                     "v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n"
                     : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(k));
        asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n"
                     "v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n"
                     : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(k));
    }
    unsigned long long t1 = now();
    REPORT(11, t1 - t0, N_IT * 64)
    if (i0 + i1 + i2 + i3 == 12345) out[63] = i0;
}
// v_mul_f64 / v_add_f64 independent, and f32 fma for comparison
__global__ void k_f32(double *out, double seed) {
    float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, b = 1.0000001f, c = 1e-9f;
    unsigned long long t0 = now();
#pragma unroll 1
    for (int i = 0; i < N_IT; ++i) {
        asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n"
                     "v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
        asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n"
                     "v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
        asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n"
                     "v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
        asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n"
                     "v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
        asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n"
                     "v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
        asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n"
                     "v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
        asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n"
                     "v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
        asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n"
                     "v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
    }
    unsigned long long t1 = now();
    REPORT(12, t1 - t0, N_IT * 64)
    if (a0 + a1 + a2 + a3 == 12345.0f) out[63] = a0;
}

typedef double f64x4 __attribute__((ext_vector_type(4)));
__global__ void k_mfma(double *out, double seed) {
    f64x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    double a = seed + threadIdx.x, b = 1.0 + threadIdx.x * 1e-3;
    unsigned long long t0 = now();
#pragma unroll 8
    for (int i = 0; i < N_IT; ++i) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    }
    unsigned long long t1 = now();
    REPORT(8, t1 - t0, N_IT * 4)
    if (c0[0] + c1[1] + c2[2] + c3[3] == 12345.0) out[63] = c0[0];
}

// dependent MFMA chain (same accumulator)
__global__ void k_mfma_dep(double *out, double seed) {
    f64x4 c0 = {0, 0, 0, 0};
    double a = seed + threadIdx.x, b = 1.0 + threadIdx.x * 1e-3;
    unsigned long long t0 = now();
#pragma unroll 8
    for (int i = 0; i < N_IT; ++i) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    }
    unsigned long long t1 = now();
    REPORT(9, t1 - t0, N_IT * 2)
    if (c0[0] == 12345.0) out[63] = c0[0];
}

int main() {
    double *d_out;
    hipMalloc(&d_out, 64 * sizeof(double));
    hipMemset(d_out, 0, 64 * sizeof(double));
    const char *names[] = {"dependent v_fma_f64 (cycles/instr)", "8 independent v_fma_f64 chains (cycles/instr)",
                           "dependent v_add_f64 (cycles/instr)", "2 interleaved dependent fma chains (cycles/instr)",
                           "16-lane DPP sum of a double, dependent (cycles/reduction)", "dependent ds_read_b32 (cycles)",
                           "independent ds_read_b64 + add (cycles/instr)", "v_rsq_f64 x4 independent (cycles/instr)",
                           "v_mfma_f64_16x16x4 x4 independent (cycles/instr)", "v_mfma_f64_16x16x4 dependent (cycles/instr)",
                           "v_fma_f64 + v_add_u32 alternating, independent (cycles/pair)", "v_add_u32 independent (cycles/instr)",
                           "v_fma_f32 independent (cycles/instr)"};
    for (int waves = 1; waves <= 2; ++waves) {        // 1 or 2 waves per SIMD (block = 256 or 512 threads, one block per CU)
        const int threads = 256 * waves;
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(k_dep_fma, dim3(256), dim3(threads), 0, 0, d_out, 1.5);
            hipLaunchKernelGGL(k_indep_fma, dim3(256), dim3(threads), 0, 0, d_out, 1.5);
            hipLaunchKernelGGL(k_dep_add, dim3(256), dim3(threads), 0, 0, d_out, 1.5);
            hipLaunchKernelGGL(k_dep2_fma, dim3(256), dim3(threads), 0, 0, d_out, 1.5);
            hipLaunchKernelGGL(k_dpp_reduce, dim3(256), dim3(threads), 0, 0, d_out, 1.5);
            hipLaunchKernelGGL(k_lds_latency, dim3(256), dim3(threads), 0, 0, d_out, 1.5);
            hipLaunchKernelGGL(k_lds_tp, dim3(256), dim3(threads), 0, 0, d_out, 1.5);
            hipLaunchKernelGGL(k_rsq, dim3(256), dim3(threads), 0, 0, d_out, 1.5);
            hipLaunchKernelGGL(k_mfma, dim3(256), dim3(threads), 0, 0, d_out, 1.5);
            hipLaunchKernelGGL(k_mfma_dep, dim3(256), dim3(threads), 0, 0, d_out, 1.5);
            hipLaunchKernelGGL(k_mix, dim3(256), dim3(threads), 0, 0, d_out, 1.5);
            hipLaunchKernelGGL(k_int, dim3(256), dim3(threads), 0, 0, d_out, 1.5);
            hipLaunchKernelGGL(k_f32, dim3(256), dim3(threads), 0, 0, d_out, 1.5);
            hipDeviceSynchronize();
        }
        std::vector<double> h(64);
        hipMemcpy(h.data(), d_out, 64 * sizeof(double), hipMemcpyDeviceToHost);
        printf("== %d wave(s) per SIMD (timed on wave 0 of block 0; s_memtime ticks)\n", waves);
        for (int i = 0; i < 13; ++i) printf("%-62s %8.2f\n", names[i], h[i]);
    }
    hipError_t e = hipGetLastError();
    printf("status: %s\n", hipGetErrorString(e));
    return 0;
}
