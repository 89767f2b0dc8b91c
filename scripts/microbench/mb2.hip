// Co-residency micro-benchmark (round 2): what do 1, 2, 3 or 4 waves on ONE SIMD get from the FP64 / integer VALU
// issue ports and from the LDS?  Every wave records where it ran (HW_REG_HW_ID: wave slot, SIMD, CU, SE; XCC id)
// and its own start / end s_memtime, so the table proves that the waves compared really shared a SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 scripts/microbench/mb2.hip -o scripts/microbench/mb2.out
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <map>
#include <vector>

#define N_IT 1024
#define PER_IT 64

struct Rec { unsigned hw_id, xcc; unsigned long long t0, t1; };

__device__ __forceinline__ unsigned long long now() { return __builtin_readcyclecounter(); }
__device__ __forceinline__ unsigned hw_id() { return __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4); }     // HW_REG_HW_ID
__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20); }   // HW_REG_XCC_ID

#define FMA8 "v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n" \
             "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
#define ADD8 "v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n" \
             "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
#define MIX8 "v_fma_f64 %0, %0, %8, %9\n v_add_u32 %10, %10, %11\n v_fma_f64 %1, %1, %8, %9\n v_add_u32 %10, %10, %11\n" \
             "v_fma_f64 %2, %2, %8, %9\n v_add_u32 %10, %10, %11\n v_fma_f64 %3, %3, %8, %9\n v_add_u32 %10, %10, %11\n"

// mode 0: 8 independent FP64 FMA chains; 1: integer adds; 2: FMA/int alternating; 3: ds_read_b64 stream + FP64 add;
// 4: dependent 16-lane DPP reduction chain (as in the feature stage)
template <int MODE>
__global__ void k_issue(Rec *rec, double seed, int n_it) {
    __shared__ double lds[4096];
    for (int n = threadIdx.x; n < 4096; n += blockDim.x) lds[n] = seed + n;
    __syncthreads();
    double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
    double b = 1.0000001, c = 1e-9;
    unsigned u0 = threadIdx.x, u1 = 1, u2 = 2, u3 = 3, u4 = 4, u5 = 5, u6 = 6, u7 = 7, inc = 3;
    const double *lp = lds + (threadIdx.x & 63);
    const unsigned long long t0 = now();
#pragma unroll 1
    for (int i = 0; i < n_it; ++i) {
        if (MODE == 0) {
#pragma unroll
            for (int r = 0; r < PER_IT / 8; ++r)
                asm volatile(FMA8 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        } else if (MODE == 1) {
#pragma unroll
            for (int r = 0; r < PER_IT / 8; ++r)
                asm volatile(ADD8 : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(inc));
        } else if (MODE == 2) {
#pragma unroll
            for (int r = 0; r < PER_IT / 8; ++r)
                asm volatile(MIX8 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                             : "v"(b), "v"(c), "v"(u0), "v"(inc));
        } else if (MODE == 3) {
#pragma unroll
            for (int r = 0; r < PER_IT / 8; ++r) {
                const double x0 = lp[64 * 0 + 512 * (r & 7)], x1 = lp[64 * 1 + 512 * (r & 7)], x2 = lp[64 * 2 + 512 * (r & 7)],
                             x3 = lp[64 * 3 + 512 * (r & 7)], x4 = lp[64 * 4 + 512 * (r & 7)], x5 = lp[64 * 5 + 512 * (r & 7)],
                             x6 = lp[64 * 6 + 512 * (r & 7)], x7 = lp[64 * 7 + 512 * (r & 7)];
                a0 += x0; a1 += x1; a2 += x2; a3 += x3; a4 += x4; a5 += x5; a6 += x6; a7 += x7;
                asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            }
        } else {
#pragma unroll
            for (int r = 0; r < PER_IT / 8; ++r) {       // 8 dependent 4-step reductions
#pragma unroll
                for (int z = 0; z < 8; ++z) {
                    int lo, hi;
#define STEP_(ctrl) lo = __builtin_amdgcn_update_dpp(0, __double2loint(a0), ctrl, 0xF, 0xF, true); \
                    hi = __builtin_amdgcn_update_dpp(0, __double2hiint(a0), ctrl, 0xF, 0xF, true); a0 += __hiloint2double(hi, lo);
                    STEP_(0xB1) STEP_(0x4E) STEP_(0x141) STEP_(0x140)
#undef STEP_
                    a0 = a0 * 0.0625;
                }
            }
        }
    }
    const unsigned long long t1 = now();
    if ((threadIdx.x & 63) == 0) {
        Rec r; r.hw_id = hw_id(); r.xcc = xcc_id(); r.t0 = t0; r.t1 = t1;
        rec[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = r;
    }
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.0 || u0 + u1 + u2 + u3 + u4 + u5 + u6 + u7 == 77u) rec[0].t1 = 0;
}

template <int MODE>
static void run(const char *name, Rec *d_rec, int ops_per_iter) {
    printf("-- %s\n", name);
    for (int waves_per_simd = 1; waves_per_simd <= 4; ++waves_per_simd) {
        const int threads = 256 * waves_per_simd, blocks = 256, nw = threads / 64;
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL((k_issue<MODE>), dim3(blocks), dim3(threads), 0, 0, d_rec, 1.5, N_IT);
            hipDeviceSynchronize();
        }
        std::vector<Rec> h((size_t)blocks * nw);
        hipMemcpy(h.data(), d_rec, h.size() * sizeof(Rec), hipMemcpyDeviceToHost);
        // block 0: print where its waves ran; all blocks: per-SIMD aggregate
        double sum_own = 0, sum_agg = 0; int n_simd = 0, bad = 0;
        for (int b = 0; b < blocks; ++b) {
            std::map<unsigned, std::vector<int>> by_simd;      // key = (xcc, se, cu, simd)
            for (int w = 0; w < nw; ++w) {
                const Rec &r = h[(size_t)b * nw + w];
                const unsigned simd = (r.hw_id >> 4) & 3, cu = (r.hw_id >> 8) & 15, sh = (r.hw_id >> 12) & 1, se = (r.hw_id >> 13) & 7;
                by_simd[(r.xcc & 15) << 16 | se << 12 | sh << 8 | cu << 4 | simd].push_back(w);
            }
            for (auto &kv : by_simd) {
                if ((int)kv.second.size() != waves_per_simd) ++bad;
                unsigned long long lo = ~0ull, hi = 0; double own = 0;
                for (int w : kv.second) {
                    const Rec &r = h[(size_t)b * nw + w];
                    lo = std::min(lo, r.t0); hi = std::max(hi, r.t1);
                    own += (double)(r.t1 - r.t0) / ((double)N_IT * ops_per_iter);
                }
                sum_own += own / kv.second.size();
                sum_agg += (double)(hi - lo) / ((double)N_IT * ops_per_iter * kv.second.size());
                ++n_simd;
            }
            if (b == 0) {
                printf("   block 0, %d waves:", nw);
                for (int w = 0; w < nw; ++w) {
                    const Rec &r = h[w];
                    printf(" [w%d simd%u slot%u cu%u xcc%u]", w, (r.hw_id >> 4) & 3, r.hw_id & 15, (r.hw_id >> 8) & 15, r.xcc & 15);
                }
                printf("\n");
            }
        }
        printf("   %d wave(s)/SIMD: cycles per instruction seen by one wave %7.2f | per instruction issued on the SIMD %7.2f"
               "   (SIMDs sampled %d, with unexpected wave count %d)\n",
               waves_per_simd, sum_own / n_simd, sum_agg / n_simd, n_simd, bad);
    }
}

int main() {
    Rec *d_rec;
    hipMalloc(&d_rec, 256 * 16 * sizeof(Rec));
    run<0>("v_fma_f64, 8 independent chains", d_rec, PER_IT);
    run<1>("v_add_u32, 8 independent chains", d_rec, PER_IT);
    run<2>("v_fma_f64 / v_add_u32 alternating (per instruction of either kind)", d_rec, PER_IT);
    run<3>("ds_read_b64 (conflict-free) + v_add_f64 (per load+add pair)", d_rec, PER_IT);
    run<4>("dependent 16-lane DPP sum of a double (per reduction: 8 dpp movs + 4 adds + 1 mul)", d_rec, PER_IT);
    printf("status: %s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
