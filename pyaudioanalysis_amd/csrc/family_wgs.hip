// The real-input split of the long even windows on three register passes per sub-transform (kernels_wgs.hpp: 12 / 6 x 3675 samples = 44 100 /
// 22 050, 12 / 8 / 6 x 4000 = 48 000 / 32 000 / 24 000) -- own translation unit, see family_launch.hpp.
#define PAA_NO_HOST_LAUNCHERS
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "family_launch.hpp"
#include "kernels_wgs.hpp"

namespace paa {
namespace launch {

template <typename T, int R0, typename SH>
static int wgs_one(const PlanDev &P, const void *d_packed, const ClipDev *clips, const ClipNorm *norms, const wg::FrameRef *tasks, int n_tasks,
                   int *counter, int num_cu, double *spec, double *tfeat, double *psum, double *d_out, hipStream_t stream) {
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&wgs::wgs_kernel<T, R0, SH>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                SH::LDS_BYTES) != hipSuccess)
            return -1;
        attr = true;
    }
    // one workgroup per CU; a multiple of eight (one segment of the task list per XCD) whenever every workgroup of such a grid has a task
    unsigned grid = (unsigned)std::min(n_tasks, num_cu);
    if (grid >= 64) grid &= ~7u;
    if (grid == 0) return 0;
    hipLaunchKernelGGL((wgs::wgs_kernel<T, R0, SH>), dim3(grid), dim3(SH::NT), (size_t)SH::LDS_BYTES, stream, P, (const T *)d_packed, clips, norms,
                       tasks, n_tasks, counter, spec, tfeat, psum, d_out);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
template <int R0, typename SH>
static int wgs_kinds(int sample_kind, const PlanDev &P, const void *d_packed, const ClipDev *clips, const ClipNorm *norms,
                     const wg::FrameRef *tasks, int n_tasks, int *counter, int num_cu, double *spec, double *tfeat, double *psum,
                     double *d_out, hipStream_t stream) {
    if (sample_kind == 0) return wgs_one<int16_t, R0, SH>(P, d_packed, clips, norms, tasks, n_tasks, counter, num_cu, spec, tfeat, psum, d_out, stream);
    if (sample_kind == 2) return wgs_one<stereo16, R0, SH>(P, d_packed, clips, norms, tasks, n_tasks, counter, num_cu, spec, tfeat, psum, d_out, stream);
    return wgs_one<double, R0, SH>(P, d_packed, clips, norms, tasks, n_tasks, counter, num_cu, spec, tfeat, psum, d_out, stream);
}
#define PAA_WGS_ARGS sample_kind, P, d_packed, clips, norms, tasks, n_tasks, counter, num_cu, spec, tfeat, psum, d_out, stream
int wgs(int r0, int q, int sample_kind, const PlanDev &P, const void *d_packed, const ClipDev *clips, const ClipNorm *norms,
        const wg::FrameRef *tasks, int n_tasks, int *counter, int num_cu, double *spec, double *tfeat, double *psum, double *d_out,
        hipStream_t stream) {
    if (q == wgs::S3675::Q) {
        if (r0 == 12) return wgs_kinds<12, wgs::S3675>(PAA_WGS_ARGS);
        if (r0 == 6) return wgs_kinds<6, wgs::S3675>(PAA_WGS_ARGS);
    } else if (q == wgs::S4000::Q) {
        if (r0 == 12) return wgs_kinds<12, wgs::S4000>(PAA_WGS_ARGS);
        if (r0 == 8) return wgs_kinds<8, wgs::S4000>(PAA_WGS_ARGS);
        if (r0 == 6) return wgs_kinds<6, wgs::S4000>(PAA_WGS_ARGS);
    }
    return -1;
}

template <int R0, int Q>
static int wgs_feat_one(const PlanDev &P, const wg::FrameRef *frames, int n_frames, const ClipDev *clips, const double *spec,
                        const double *tfeat, const double *psum, double *d_out, hipStream_t stream) {
    if (n_frames <= 0) return 0;
    constexpr size_t lds = (size_t)wgs::feat_lds<R0, Q>();
    hipLaunchKernelGGL((wgs::wgs_feat_kernel<R0, Q>), dim3(8u * (unsigned)((n_frames + 7) / 8)), dim3(wgs::kFeatT), lds,
                       stream, P, frames, clips, n_frames, spec, tfeat, psum, d_out);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
int wgs_feat(int r0, int q, const PlanDev &P, const wg::FrameRef *frames, int n_frames, const ClipDev *clips, const double *spec,
             const double *tfeat, const double *psum, double *d_out, hipStream_t stream) {
    static_assert(wgs::feat_lds<12, wgs::S4000::Q>() <= 64 * 1024, "the feature kernel's LDS needs no attribute");
    if (q == wgs::S3675::Q) {
        if (r0 == 12) return wgs_feat_one<12, wgs::S3675::Q>(P, frames, n_frames, clips, spec, tfeat, psum, d_out, stream);
        if (r0 == 6) return wgs_feat_one<6, wgs::S3675::Q>(P, frames, n_frames, clips, spec, tfeat, psum, d_out, stream);
    } else if (q == wgs::S4000::Q) {
        if (r0 == 12) return wgs_feat_one<12, wgs::S4000::Q>(P, frames, n_frames, clips, spec, tfeat, psum, d_out, stream);
        if (r0 == 8) return wgs_feat_one<8, wgs::S4000::Q>(P, frames, n_frames, clips, spec, tfeat, psum, d_out, stream);
        if (r0 == 6) return wgs_feat_one<6, wgs::S4000::Q>(P, frames, n_frames, clips, spec, tfeat, psum, d_out, stream);
    }
    return -1;
}

}  // namespace launch
}  // namespace paa
