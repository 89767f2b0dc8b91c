// Calibration of rocprofv3's WRITE_SIZE on gfx950 against KNOWN byte counts, in the store patterns the feature kernels
// use (MI355X_MICROARCH.md, HBM: "WRITE_SIZE is uncalibrated: calibrate on a known byte count in your own pattern").
// Every kernel writes the same [ROWS][T] float64 matrix (39.2 MB at the defaults), only the store pattern differs:
//   k_coalesced : 16 B per lane, consecutive lanes -> consecutive addresses (1 KB per wave instruction)
//   k_rows32    : one ROW per lane, 32-byte aligned groups of four frames, RUN frames per wave (the realigned row stores)
//   k_rows16x2  : one row per lane, two 16-byte stores per group at 8-byte alignment (rows of odd length: the old pattern)
//   k_rows32_gap: k_rows32 with ~15 us of idle time between groups and a concurrent streaming read (what the feature
//                 kernel does to the L2 between two groups of one row)
//   k_rows64    : one row per lane, 64 bytes (two aligned 32-byte stores back to back) every other group
// Build: hipcc --offload-arch=gfx950 -O3 scripts/microbench/mb3.hip -o scripts/microbench/mb3.out
// Run:   rocprofv3 --pmc WRITE_SIZE -d out -o pmc -- scripts/microbench/mb3.out   (and TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef double f64x4 __attribute__((ext_vector_type(4), aligned(32)));
typedef double f64x2 __attribute__((ext_vector_type(2), aligned(8)));
constexpr int ROWS = 34;

__global__ void k_coalesced(double *out, long long n) {
    const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 2;
    if (i + 1 < n) *reinterpret_cast<f64x2 *>(out + i) = f64x2{(double)i, 1.0};
}

// wave w owns frames [w * run, (w + 1) * run) of every row; lane = row.
// MODE 0: 32-byte groups back to back; 1: two 16-byte stores at 8-byte alignment; 3: 64 bytes every other group;
// MODE 2/4/5/6/7: GRP frames (32 / 64 / 128 bytes, or 16 bytes = half a sector for 7) per row, then a gap of ~15 us per four
// frames -- with (2, 5, 6, 7) or without (4) a streaming read through the same L2 in the gap
template <int MODE, int GRP>
__global__ void k_rows(double *out, long long T, int run, const float4 *stream, long long n_stream, float *sink) {
    const int lane = threadIdx.x & 63;
    const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long t0 = wave * run, t1 = (t0 + run < T) ? t0 + run : T;
    if (t0 >= T) return;
    float acc = 0.f;
    double *row = out + (long long)lane * T;
    constexpr bool GAP = MODE == 2 || MODE >= 4;
    for (long long t = t0; t + GRP <= t1; t += GRP) {
        if (lane < ROWS) {
            if (MODE == 1) {      // rows start 8 bytes past a 32-byte boundary
                *reinterpret_cast<f64x2 *>(row + t + 1) = f64x2{(double)t, 1.0};
                *reinterpret_cast<f64x2 *>(row + t + 3) = f64x2{2.0, 3.0};
            } else if (GRP == 2) {
                *reinterpret_cast<f64x2 *>(row + t) = f64x2{(double)t, 1.0};
            } else {
#pragma unroll
                for (int g = 0; g < GRP; g += 4) *reinterpret_cast<f64x4 *>(row + t + g) = f64x4{(double)t, 1.0, 2.0, 3.0};
            }
        }
        if (GAP) {                // ~15 us per four frames; the read stream is 800 B per frame as in the feature kernel
            if (MODE != 4) {
                const long long base = (wave * 4096 + (t - t0) * 64) % (n_stream - 4096);
                for (int k = 0; k < 3 * GRP; ++k) { float4 v = stream[base + k * 64 + lane]; acc += v.x + v.w; }
            }
            for (int k = 0; k < 10 * GRP; ++k) __builtin_amdgcn_s_sleep(127);
        }
    }
    if (acc == 123.456f) sink[0] = acc;
}

// The feature kernel's pattern itself: rows of ODD length (8-byte aligned only), four frames per iteration, every lane stores
// the CH-frame chunk of its row that (a) is CH*8-byte aligned and (b) completes with this iteration; chunks that cross the
// run's ends are skipped (the kernel writes them in pieces).  STREAM = read stream in the gap.
template <int CH, int STREAM>
__global__ void k_chunks(double *out, long long T, int run, const float4 *stream, long long n_stream, float *sink) {
    const int lane = threadIdx.x & 63;
    const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long t0 = wave * run, t1 = (t0 + run < T) ? t0 + run : T;
    if (t0 >= T) return;
    float acc = 0.f;
    const long long row_off = (long long)lane * T;
    double *row = out + row_off;
    for (long long q0 = t0; q0 < t1; q0 += 4) {
        const int s = (CH - 1) - (int)((row_off + q0) & (CH - 1));
        const long long g = q0 - (CH - 1) + s;
        if (lane < ROWS && s <= 3 && g >= t0 && g + CH <= t1) {
#pragma unroll
            for (int k = 0; k < CH; k += 4) *reinterpret_cast<f64x4 *>(row + g + k) = f64x4{(double)g, 1.0, 2.0, 3.0};
        }
        if (STREAM) {
            const long long base = (wave * 4096 + (q0 - t0) * 64) % (n_stream - 4096);
            for (int k = 0; k < 12; ++k) { float4 v = stream[base + k * 64 + lane]; acc += v.x + v.w; }
        }
        for (int k = 0; k < 40; ++k) __builtin_amdgcn_s_sleep(127);
    }
    if (acc == 123.456f) sink[0] = acc;
}
static long long chunk_bytes(int CH, long long T, int run) {
    long long n = 0;
    for (int lane = 0; lane < ROWS; ++lane)
        for (long long t0 = 0; t0 < T; t0 += run) {
            const long long t1 = (t0 + run < T) ? t0 + run : T;
            for (long long q0 = t0; q0 < t1; q0 += 4) {
                const int s = (CH - 1) - (int)(((long long)lane * T + q0) & (CH - 1));
                const long long g = q0 - (CH - 1) + s;
                if (s <= 3 && g >= t0 && g + CH <= t1) n += CH * 8;
            }
        }
    return n;
}

int main() {
    const long long T = 144000;
    const int run = 80;
    double *out; float4 *stream; float *sink;
    hipMalloc(&out, (64 * T + 8) * sizeof(double));
    const long long n_stream = 8 << 20;                   // 128 MB
    hipMalloc(&stream, n_stream * sizeof(float4));
    hipMalloc(&sink, 64);
    hipMemset(stream, 0, n_stream * sizeof(float4));
    hipMemset(out, 0, (64 * T + 8) * sizeof(double));
    hipDeviceSynchronize();
    const long long n = ROWS * T;
    const int waves = (int)((T + run - 1) / run), wg = 8;
    for (int rep = 0; rep < 3; ++rep) {
        k_coalesced<<<(unsigned)((n / 2 + 255) / 256), 256>>>(out, n);
#define ROWS_K(M, G) k_rows<M, G><<<(waves + wg - 1) / wg, 64 * wg>>>(out, T, run, stream, n_stream, sink);
        ROWS_K(0, 4) ROWS_K(1, 4) ROWS_K(3, 8) ROWS_K(2, 4) ROWS_K(4, 4) ROWS_K(5, 8) ROWS_K(6, 16) ROWS_K(7, 2)
        {
            const long long To = 143999;
            const int run2 = 72, waves2 = (int)((To + run2 - 1) / run2);
            k_chunks<8, 1><<<(waves2 + wg - 1) / wg, 64 * wg>>>(out, To, run2, stream, n_stream, sink);
            k_chunks<8, 0><<<(waves2 + wg - 1) / wg, 64 * wg>>>(out, To, run2, stream, n_stream, sink);
            k_chunks<16, 1><<<(waves2 + wg - 1) / wg, 64 * wg>>>(out, To, run2, stream, n_stream, sink);
            k_chunks<4, 1><<<(waves2 + wg - 1) / wg, 64 * wg>>>(out, To, run2, stream, n_stream, sink);
            if (rep == 0) printf("k_chunks known bytes (T %lld, run %d): CH=8 %lld, CH=16 %lld, CH=4 %lld\n", To, run2,
                                 chunk_bytes(8, To, run2), chunk_bytes(16, To, run2), chunk_bytes(4, To, run2));
        }
        hipDeviceSynchronize();
    }
    printf("known bytes per kernel: %lld (rows %d x T %lld x 8)\n", n * 8, ROWS, T);
    return 0;
}
