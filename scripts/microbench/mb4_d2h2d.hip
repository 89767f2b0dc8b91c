// How fast does a [34][T] float64 slab come back to PAGEABLE host memory in column ranges?  (host pipeline of one
// feature_extraction call: D2H of frame range k under the kernel of range k + 1 needs 2-D copies with the row pitch T * 8)
//   hipcc --offload-arch=gfx950 -O2 scripts/microbench/mb4_d2h2d.hip -o scripts/microbench/mb4.out && scripts/microbench/mb4.out
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t F = 34, T = 143999, bytes = F * T * 8;
    double *d = nullptr;
    CK(hipMalloc(&d, bytes));
    CK(hipMemset(d, 1, bytes));
    double *h = (double *)malloc(bytes), *hp = nullptr;
    memset(h, 0, bytes);
    CK(hipHostMalloc(&hp, bytes));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (int rep = 0; rep < 3; ++rep) {
        double t0 = now();
        CK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, s));
        CK(hipStreamSynchronize(s));
        double t1 = now();
        printf("1-D pageable            %.3f ms  %.1f GB/s\n", (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9);
        for (int R : {2, 4, 8}) {
            t0 = now();
            for (int r = 0; r < R; ++r) {
                const size_t a = T * r / R, b = T * (r + 1) / R;
                CK(hipMemcpy2DAsync(h + a, T * 8, d + a, T * 8, (b - a) * 8, F, hipMemcpyDeviceToHost, s));
            }
            CK(hipStreamSynchronize(s));
            t1 = now();
            printf("2-D pageable, %d ranges  %.3f ms  %.1f GB/s\n", R, (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9);
        }
        t0 = now();
        CK(hipMemcpyAsync(hp, d, bytes, hipMemcpyDeviceToHost, s));
        CK(hipStreamSynchronize(s));
        t1 = now();
        printf("1-D pinned              %.3f ms  %.1f GB/s\n", (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9);
        t0 = now();
        for (int r = 0; r < 4; ++r) {
            const size_t a = T * r / 4, b = T * (r + 1) / 4;
            CK(hipMemcpy2DAsync(hp + a, T * 8, d + a, T * 8, (b - a) * 8, F, hipMemcpyDeviceToHost, s));
        }
        CK(hipStreamSynchronize(s));
        t1 = now();
        printf("2-D pinned, 4 ranges    %.3f ms  %.1f GB/s\n", (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9);
        // row-range alternative: the slab as 4 groups of rows (contiguous 1-D copies)
        t0 = now();
        for (int r = 0; r < 4; ++r) {
            const size_t a = F * r / 4, b = F * (r + 1) / 4;
            CK(hipMemcpyAsync(h + a * T, d + a * T, (b - a) * T * 8, hipMemcpyDeviceToHost, s));
        }
        CK(hipStreamSynchronize(s));
        t1 = now();
        printf("1-D pageable, 4 row groups %.3f ms  %.1f GB/s\n", (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9);
    }
    return 0;
}
