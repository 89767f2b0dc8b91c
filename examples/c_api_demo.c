/* Minimal C client of libpaa_hip.so: the C ABI needs no Python.
 *   gcc -O2 -Iinclude examples/c_api_demo.c -o c_api_demo -Lpyaudioanalysis_amd -lpaa_hip -Wl,-rpath,$PWD/pyaudioanalysis_amd -lm
 *   ./c_api_demo                    a synthetic 2 s 440 Hz tone
 *   ./c_api_demo clip.raw [out.f64] raw 16 kHz mono int16 PCM (native byte order); with a second argument the whole
 *                                   68 x T matrix is written there as raw float64, row-major
 * With a GPU it extracts the 68 x T short-term matrix (ShortTermFeatures.feature_extraction(x, 16000, 800, 400), reference
 * ShortTermFeatures.py:543) and prints a few entries with 17 significant digits plus the sum of |F| over the matrix --
 * tests/test_parity_gpu.py::test_c_client_on_gpu feeds it a golden clip of the reference and compares; without a GPU it
 * reports the library's error (there is no CPU fallback) and exits with status 2. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "paa_hip.h"

int main(int argc, char **argv) {
    const int fs = 16000, window = 800, step = 400;
    int64_t n = 2 * fs;
    int16_t *x = NULL;
    if (argc > 1) {
        FILE *f = fopen(argv[1], "rb");
        if (!f) { printf("cannot open %s\n", argv[1]); return 1; }
        fseek(f, 0, SEEK_END);
        n = (int64_t)(ftell(f) / 2);
        fseek(f, 0, SEEK_SET);
        x = (int16_t *)malloc(sizeof(int16_t) * (size_t)(n > 0 ? n : 1));
        if (fread(x, 2, (size_t)n, f) != (size_t)n) { printf("short read\n"); fclose(f); free(x); return 1; }
        fclose(f);
    } else {
        x = (int16_t *)malloc(sizeof(int16_t) * (size_t)n);
        for (int64_t i = 0; i < n; ++i) x[i] = (int16_t)lrint(8000.0 * sin(2.0 * M_PI * 440.0 * (double)i / fs));
    }
    printf("%s, %d device(s)\n", paa_version(), paa_device_count());
    const int64_t T = paa_num_frames(n, window, step);
    double *F = (double *)malloc(sizeof(double) * 68 * (size_t)(T > 0 ? T : 1));
    const int rc = paa_st_features_i16(x, n, (double)fs, window, step, 1, F);
    if (rc != PAA_OK) {
        printf("paa_st_features_i16 -> %d: %s\n", rc, paa_last_error());
        free(F); free(x);
        return rc == PAA_ERR_HIP ? 2 : 1;
    }
    printf("frames %lld: zcr[0]=%.6f energy[0]=%.6f centroid[0]=%.6f mfcc_1[0]=%.4f\n", (long long)T, F[0 * T], F[1 * T],
           F[3 * T], F[8 * T]);
    /* entries (row, frame) a checker can compare with the reference: energy, centroid, mfcc_1, chroma_3, delta mfcc_2 */
    const int rows[5] = {1, 3, 8, 23, 34 + 9};
    const int64_t cols[3] = {0, T / 2, T - 1};
    for (int r = 0; r < 5; ++r)
        for (int c = 0; c < 3; ++c)
            printf("entry %d %lld %.17g\n", rows[r], (long long)cols[c], F[(int64_t)rows[r] * T + cols[c]]);
    long double sum = 0.0L;
    for (int64_t i = 0; i < 68 * T; ++i) sum += fabsl((long double)F[i]);
    printf("sum_abs %.17Lg\n", sum);
    if (argc > 2) {
        FILE *o = fopen(argv[2], "wb");
        if (!o || fwrite(F, 8, (size_t)(68 * T), o) != (size_t)(68 * T)) { printf("cannot write %s\n", argv[2]); return 1; }
        fclose(o);
    }
    paa_shutdown();
    free(F); free(x);
    return 0;
}
