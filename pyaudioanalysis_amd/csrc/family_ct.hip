// The 2 RA RB register-FFT family (kernels_ct.hpp: windows 800, 640, 400, 320) -- own translation unit, see family_launch.hpp.
#define PAA_NO_HOST_LAUNCHERS
#define PAA_LAUNCH_CT
#include <cstdlib>
#include <cstring>

#include "family_launch.hpp"

namespace paa {
namespace launch {
int ct(const ct::CtLaunch &cl, int sample_kind, const PlanDev &P, const unsigned char *blob, const void *d_packed,
       const ClipDev *clips, const ClipNorm *norms, const Tile *tiles, long long n_tiles, double *d_out, hipStream_t stream) {
    return ct::ct_launch(cl, sample_kind, P, blob, d_packed, clips, norms, tiles, n_tiles, d_out, stream);
}
PAA_PHASE_READER(phase_ct)
}  // namespace launch
}  // namespace paa
