// Three-pass register FFT (kernels_tri.hpp), first unit: the 50 ms windows at 48 / 44.1 kHz and config 5's feature matrix
// (2400, 2205, 1102) -- see family_launch.hpp.
#define PAA_NO_HOST_LAUNCHERS
#define PAA_LAUNCH_TRI
#define PAA_TRI_SHAPES_HERE(X) X(0, S2400) X(1, S2205) X(7, S1102)
#include <cstdlib>
#include <cstring>

#include "family_launch.hpp"

namespace paa {
namespace launch {
int tri_part_a(const tri::TriLaunch &tl, int sample_kind, const PlanDev &P, const unsigned char *blob, const void *d_packed,
               const ClipDev *clips, const ClipNorm *norms, const Tile *tiles, long long n_tiles, double *d_out,
               hipStream_t stream) {
    return tri::tri_launch(tl, sample_kind, P, blob, d_packed, clips, norms, tiles, n_tiles, d_out, stream);
}
// which unit holds a shape (tri_shape_of's numbering)
int tri(const tri::TriLaunch &tl, int sample_kind, const PlanDev &P, const unsigned char *blob, const void *d_packed,
        const ClipDev *clips, const ClipNorm *norms, const Tile *tiles, long long n_tiles, double *d_out, hipStream_t stream) {
    const bool here = tl.shape == 0 || tl.shape == 1 || tl.shape == 7;
    if (tl.shape >= 8) return tri_part_c(tl, sample_kind, P, blob, d_packed, clips, norms, tiles, n_tiles, d_out, stream);
    return here ? tri_part_a(tl, sample_kind, P, blob, d_packed, clips, norms, tiles, n_tiles, d_out, stream)
                : tri_part_b(tl, sample_kind, P, blob, d_packed, clips, norms, tiles, n_tiles, d_out, stream);
}
PAA_PHASE_READER(phase_tri_a)
}  // namespace launch
}  // namespace paa
