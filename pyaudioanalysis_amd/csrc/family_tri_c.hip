// Three-pass register FFT (kernels_tri.hpp), third unit: the power-of-two windows 1024, 2048, 512 -- see family_launch.hpp.
#define PAA_NO_HOST_LAUNCHERS
#define PAA_LAUNCH_TRI
#define PAA_TRI_SHAPES_HERE(X) X(8, S1024) X(9, S2048) X(10, S512) X(11, S256)
#include <cstdlib>
#include <cstring>

#include "family_launch.hpp"

namespace paa {
namespace launch {
int tri_part_c(const tri::TriLaunch &tl, int sample_kind, const PlanDev &P, const unsigned char *blob, const void *d_packed,
               const ClipDev *clips, const ClipNorm *norms, const Tile *tiles, long long n_tiles, double *d_out,
               hipStream_t stream) {
    return tri::tri_launch(tl, sample_kind, P, blob, d_packed, clips, norms, tiles, n_tiles, d_out, stream);
}
PAA_PHASE_READER(phase_tri_c)
}  // namespace launch
}  // namespace paa
