// The in-place mixed-radix kernel (kernels_mix.hpp) and the generic Stockham kernel (kernels_generic.hpp) -- own translation
// unit, see family_launch.hpp.  (The prime-factor kernel st_reg that gave the unit its name left the tree in round 6:
// scripts/experiments/kernels_reg.hpp.)
#define PAA_NO_HOST_LAUNCHERS
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "family_launch.hpp"

namespace paa {
namespace launch {

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
