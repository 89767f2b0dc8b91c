// The workgroup-wide three-pass register transform with fused features (kernels_wgr.hpp: 16 000- and 8 000-sample windows) -- own
// translation unit, see family_launch.hpp.
#define PAA_NO_HOST_LAUNCHERS
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "family_launch.hpp"
#include "kernels_wgr.hpp"

namespace paa {
namespace launch {

template <typename SH, typename T>
static int wgr_modes(int mode, const PlanDev &P, const void *d_packed, const ClipDev *clips, const ClipNorm *norms, const Tile *runs,
                     long long n_runs, int num_cu, const wgr::WgrTab *d_tab, double *d_out, hipStream_t stream) {
    if (mode == 0) return wgr::wgr_launch_one<SH, T, 0>(P, d_packed, clips, norms, runs, n_runs, num_cu, d_tab, d_out, stream);
    if (mode == 1) return wgr::wgr_launch_one<SH, T, 1>(P, d_packed, clips, norms, runs, n_runs, num_cu, d_tab, d_out, stream);
    return wgr::wgr_launch_one<SH, T, 2>(P, d_packed, clips, norms, runs, n_runs, num_cu, d_tab, d_out, stream);
}
template <typename SH>
static int wgr_kinds(int sample_kind, int mode, const PlanDev &P, const void *d_packed, const ClipDev *clips, const ClipNorm *norms,
                     const Tile *runs, long long n_runs, int num_cu, const wgr::WgrTab *d_tab, double *d_out, hipStream_t stream) {
    if (sample_kind == 0) return wgr_modes<SH, int16_t>(mode, P, d_packed, clips, norms, runs, n_runs, num_cu, d_tab, d_out, stream);
    if (sample_kind == 2) return wgr_modes<SH, stereo16>(mode, P, d_packed, clips, norms, runs, n_runs, num_cu, d_tab, d_out, stream);
    return wgr_modes<SH, double>(mode, P, d_packed, clips, norms, runs, n_runs, num_cu, d_tab, d_out, stream);
}
int wgr(int shape_id, int sample_kind, int mode, const PlanDev &P, const void *d_packed, const ClipDev *clips, const ClipNorm *norms,
        const Tile *runs, long long n_runs, int num_cu, const wgr::WgrTab *d_tab, double *d_out, hipStream_t stream) {
    if (shape_id == 1) return wgr_kinds<wgr::S16000>(sample_kind, mode, P, d_packed, clips, norms, runs, n_runs, num_cu, d_tab, d_out, stream);
    if (shape_id == 2) return wgr_kinds<wgr::S8000>(sample_kind, mode, P, d_packed, clips, norms, runs, n_runs, num_cu, d_tab, d_out, stream);
    return -1;
}

PAA_PHASE_READER(phase_wgr)
}  // namespace launch
}  // namespace paa
