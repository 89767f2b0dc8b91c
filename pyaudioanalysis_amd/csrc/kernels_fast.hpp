// Specialised feature kernels (selected per window/step); see DESIGN.md.
#pragma once
#include "device_common.hpp"
#include "tables.hpp"

namespace paa {

struct FastTables {
    void *d_blob = nullptr;
};
struct FastLaunch {
    int run = 0;
    size_t lds = 0;
    const char *name = "";
    int variant = 0;
};

inline void fast_tables_free(FastTables &t) {
    if (t.d_blob) (void)hipFree(t.d_blob);
    t.d_blob = nullptr;
}

// returns 1 when a specialised kernel exists for this configuration (and fills fl), 0 when
// the generic kernel must be used, < 0 on error
inline int fast_select(int window, int step, int sample_kind, double fs, FastTables &ft, const FftPlan &fft,
                       FastLaunch &fl) {
    (void)window; (void)step; (void)sample_kind; (void)fs; (void)ft; (void)fft; (void)fl;
    return 0;
}

inline int fast_launch(const FastLaunch &fl, const PlanDev &P, const FastTables &ft, const void *d_packed,
                       const ClipDev *clips, const ClipNorm *norms, const Tile *tiles, long long n_tiles,
                       double *d_out, hipStream_t stream) {
    (void)fl; (void)P; (void)ft; (void)d_packed; (void)clips; (void)norms; (void)tiles; (void)n_tiles; (void)d_out; (void)stream;
    return -1;
}

}  // namespace paa
