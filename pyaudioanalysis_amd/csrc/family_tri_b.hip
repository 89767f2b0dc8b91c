// Three-pass register FFT (kernels_tri.hpp), second unit: 1764, 1920, 1600, 1200 and the two-pass 551 -- see family_launch.hpp.
#define PAA_NO_HOST_LAUNCHERS
#define PAA_LAUNCH_TRI
#define PAA_TRI_SHAPES_HERE(X) X(2, S1764) X(3, S1920) X(4, S1600) X(5, S1200) X(6, S551)
#include <cstdlib>
#include <cstring>

#include "family_launch.hpp"

namespace paa {
namespace launch {
int tri_part_b(const tri::TriLaunch &tl, int sample_kind, const PlanDev &P, const unsigned char *blob, const void *d_packed,
               const ClipDev *clips, const ClipNorm *norms, const Tile *tiles, long long n_tiles, double *d_out,
               hipStream_t stream) {
    return tri::tri_launch(tl, sample_kind, P, blob, d_packed, clips, norms, tiles, n_tiles, d_out, stream);
}
PAA_PHASE_READER(phase_tri_b)
}  // namespace launch
}  // namespace paa
