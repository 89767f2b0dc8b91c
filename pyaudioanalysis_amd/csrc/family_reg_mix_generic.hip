// The prime-factor register FFT (kernels_reg.hpp: rows of window 1102), the in-place mixed-radix kernel (kernels_mix.hpp) and
// the generic Stockham kernel (kernels_generic.hpp) -- own translation unit, see family_launch.hpp.
#define PAA_NO_HOST_LAUNCHERS
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "family_launch.hpp"

namespace paa {
namespace launch {

#if defined(PAA_EXPERIMENTS) || !PAA_TRI_1102_ROWS
template <typename T>
static int reg_one(const reg::RegLayout &rl, size_t lds, const PlanDev &P, const unsigned char *blob, const void *d_packed,
                   const ClipDev *clips, const ClipNorm *norms, const Tile *tiles, long long n_tiles, double *d_out,
                   hipStream_t stream) {
    using SH = reg::Shape1102;
    static LdsAttrCache attr;
    if (!attr.covers(lds)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&reg::st_reg_kernel<SH, T>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)std::max<size_t>(lds, 64 * 1024)) != hipSuccess)
            return -1;
        attr.set(std::max<size_t>(lds, 64 * 1024));
    }
    const unsigned grid = (unsigned)((n_tiles + rl.waves - 1) / rl.waves);
    hipLaunchKernelGGL((reg::st_reg_kernel<SH, T>), dim3(grid), dim3(64 * rl.waves), lds, stream, P, rl, blob,
                       (const T *)d_packed, clips, norms, tiles, (int)n_tiles, d_out);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
int reg(const reg::RegLayout &rl, size_t lds, int sample_kind, const PlanDev &P, const unsigned char *blob,
        const void *d_packed, const ClipDev *clips, const ClipNorm *norms, const Tile *tiles, long long n_tiles, double *d_out,
        hipStream_t stream) {
    if (sample_kind == 0) return reg_one<int16_t>(rl, lds, P, blob, d_packed, clips, norms, tiles, n_tiles, d_out, stream);
    if (sample_kind == 2) return reg_one<stereo16>(rl, lds, P, blob, d_packed, clips, norms, tiles, n_tiles, d_out, stream);
    return reg_one<double>(rl, lds, P, blob, d_packed, clips, norms, tiles, n_tiles, d_out, stream);
}
#else
// (the default build routes window 1102 to the three-pass family: st_reg is never dispatched, so it is not compiled in)
int reg(const reg::RegLayout &, size_t, int, const PlanDev &, const unsigned char *, const void *, const ClipDev *, const ClipNorm *,
        const Tile *, long long, double *, hipStream_t) {
    return -1;
}
#endif

template <typename T, int TWG, int LEAN>
static int mix_one(const mix::MixLayout &ml, size_t lds, const PlanDev &P, const unsigned char *blob, const void *d_packed,
                   const ClipDev *clips, const ClipNorm *norms, const Tile *tiles, long long n_tiles, double *d_out,
                   hipStream_t stream) {
    static LdsAttrCache attr;
    if (!attr.covers(lds)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&mix::st_mix_kernel<T, TWG, LEAN>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)std::max<size_t>(lds, 64 * 1024)) != hipSuccess)
            return -1;
        attr.set(std::max<size_t>(lds, 64 * 1024));
    }
    const unsigned grid = (unsigned)((n_tiles + ml.waves - 1) / ml.waves);
    hipLaunchKernelGGL((mix::st_mix_kernel<T, TWG, LEAN>), dim3(grid), dim3(64 * ml.waves), lds, stream, P, ml, blob,
                       (const T *)d_packed, clips, norms, tiles, (int)n_tiles, d_out);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
template <typename T>
static int mix_any(const mix::MixLayout &ml, size_t lds, const PlanDev &P, const unsigned char *blob, const void *d_packed,
                   const ClipDev *clips, const ClipNorm *norms, const Tile *tiles, long long n_tiles, double *d_out,
                   hipStream_t stream) {
#define PAA_MIX_GO(TWG, LEAN) return mix_one<T, TWG, LEAN>(ml, lds, P, blob, d_packed, clips, norms, tiles, n_tiles, d_out, stream);
    if (ml.lean && ml.pad_shift == 5) { if (ml.tw_global) { PAA_MIX_GO(1, 2) } PAA_MIX_GO(0, 2) }
    if (ml.lean) { if (ml.tw_global) { PAA_MIX_GO(1, 1) } PAA_MIX_GO(0, 1) }
    if (ml.tw_global) { PAA_MIX_GO(1, 0) }
    PAA_MIX_GO(0, 0)
#undef PAA_MIX_GO
}
int mix(const mix::MixLayout &ml, size_t lds, int sample_kind, const PlanDev &P, const unsigned char *blob,
        const void *d_packed, const ClipDev *clips, const ClipNorm *norms, const Tile *tiles, long long n_tiles, double *d_out,
        hipStream_t stream) {
    if (sample_kind == 0) return mix_any<int16_t>(ml, lds, P, blob, d_packed, clips, norms, tiles, n_tiles, d_out, stream);
    if (sample_kind == 2) return mix_any<stereo16>(ml, lds, P, blob, d_packed, clips, norms, tiles, n_tiles, d_out, stream);
    return mix_any<double>(ml, lds, P, blob, d_packed, clips, norms, tiles, n_tiles, d_out, stream);
}

template <typename T>
static int generic_one(const GenLayout &gl, size_t lds, const PlanDev &P, const unsigned char *blob, const void *d_packed,
                       const ClipDev *clips, const ClipNorm *norms, const Tile *tiles, long long n_tiles, double *d_out,
                       hipStream_t stream) {
    static LdsAttrCache attr;
    if (!attr.covers(lds)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&st_generic_kernel<T>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)std::max<size_t>(lds, 64 * 1024)) != hipSuccess)
            return -1;
        attr.set(std::max<size_t>(lds, 64 * 1024));
    }
    const unsigned grid = (unsigned)((n_tiles + gl.waves - 1) / gl.waves);
    hipLaunchKernelGGL(st_generic_kernel<T>, dim3(grid), dim3(64 * gl.waves), lds, stream, P, gl, blob, (const T *)d_packed,
                       clips, norms, tiles, (int)n_tiles, d_out);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
int generic(const GenLayout &gl, size_t lds, int sample_kind, const PlanDev &P, const unsigned char *blob,
            const void *d_packed, const ClipDev *clips, const ClipNorm *norms, const Tile *tiles, long long n_tiles,
            double *d_out, hipStream_t stream) {
    if (sample_kind == 0) return generic_one<int16_t>(gl, lds, P, blob, d_packed, clips, norms, tiles, n_tiles, d_out, stream);
    if (sample_kind == 2) return generic_one<stereo16>(gl, lds, P, blob, d_packed, clips, norms, tiles, n_tiles, d_out, stream);
    return generic_one<double>(gl, lds, P, blob, d_packed, clips, norms, tiles, n_tiles, d_out, stream);
}

PAA_PHASE_READER(phase_rmg)
}  // namespace launch
}  // namespace paa
