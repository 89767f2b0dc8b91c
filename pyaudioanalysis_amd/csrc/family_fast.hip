// The 800-sample int16 kernels (kernels_fast.hpp) -- own translation unit, see family_launch.hpp.
#define PAA_NO_HOST_LAUNCHERS
#define PAA_LAUNCH_FAST
#include <cstdlib>
#include <cstring>

#include "family_launch.hpp"

namespace paa {
namespace launch {
int fast(const FastLaunch &fl, const PlanDev &P, const FastTables &ft, const void *d_packed, const ClipDev *clips,
         const ClipNorm *norms, const Tile *tiles, long long n_tiles, double *d_out, hipStream_t stream) {
    return fast_launch(fl, P, ft, d_packed, clips, norms, tiles, n_tiles, d_out, stream);
}
PAA_PHASE_READER(phase_fast)
}  // namespace launch
}  // namespace paa
