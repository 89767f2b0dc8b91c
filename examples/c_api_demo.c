/* Minimal C client of libpaa_hip.so: the C ABI needs no Python.
 *   gcc -O2 -Iinclude examples/c_api_demo.c -o c_api_demo -Lpyaudioanalysis_amd -lpaa_hip -Wl,-rpath,$PWD/pyaudioanalysis_amd -lm
 * With a GPU it extracts the 68 x T short-term matrix of a synthetic 2 s clip and prints a few values; without
 * one it reports the library's error (there is no CPU fallback) and exits with status 2. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "paa_hip.h"

int main(void) {
    const int fs = 16000, window = 800, step = 400;
    const int64_t n = 2 * fs;
    int16_t *x = (int16_t *)malloc(sizeof(int16_t) * (size_t)n);
    for (int64_t i = 0; i < n; ++i) x[i] = (int16_t)lrint(8000.0 * sin(2.0 * M_PI * 440.0 * (double)i / fs));
    printf("%s, %d device(s)\n", paa_version(), paa_device_count());
    const int64_t T = paa_num_frames(n, window, step);
    double *F = (double *)malloc(sizeof(double) * 68 * (size_t)T);
    const int rc = paa_st_features_i16(x, n, (double)fs, window, step, 1, F);
    if (rc != PAA_OK) {
        printf("paa_st_features_i16 -> %d: %s\n", rc, paa_last_error());
        free(F); free(x);
        return rc == PAA_ERR_HIP ? 2 : 1;
    }
    printf("frames %lld: zcr[0]=%.6f energy[0]=%.6f centroid[0]=%.6f mfcc_1[0]=%.4f\n", (long long)T, F[0 * T], F[1 * T],
           F[3 * T], F[8 * T]);
    paa_shutdown();
    free(F); free(x);
    return 0;
}
