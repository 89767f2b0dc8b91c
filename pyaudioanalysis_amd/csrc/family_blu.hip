// The Bluestein (chirp-z) kernel (kernels_blu.hpp: windows whose FFT length has a prime factor above 13) -- own translation unit,
// see family_launch.hpp.
#define PAA_NO_HOST_LAUNCHERS
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "family_launch.hpp"

namespace paa {
namespace launch {

template <typename T, int LOG2M, bool PK = false>
static int blu_one(const blu::BluLayout &bl, size_t lds, const PlanDev &P, const unsigned char *blob, const void *d_packed,
                   const ClipDev *clips, const ClipNorm *norms, const Tile *tiles, long long n_tiles, double *d_out,
                   hipStream_t stream) {
    static LdsAttrCache attr;
    if (!attr.covers(lds)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&blu::st_blu_kernel<T, LOG2M, PK>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)std::max<size_t>(lds, 64 * 1024)) != hipSuccess)
            return -1;
        attr.set(std::max<size_t>(lds, 64 * 1024));
    }
    const unsigned grid = (unsigned)((n_tiles + bl.waves - 1) / bl.waves);
    hipLaunchKernelGGL((blu::st_blu_kernel<T, LOG2M, PK>), dim3(grid), dim3(64 * bl.waves), lds, stream, P, bl, blob,
                       (const T *)d_packed, clips, norms, tiles, (int)n_tiles, d_out);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
template <typename T>
static int blu_any(const blu::BluLayout &bl, size_t lds, const PlanDev &P, const unsigned char *blob, const void *d_packed,
                   const ClipDev *clips, const ClipNorm *norms, const Tile *tiles, long long n_tiles, double *d_out,
                   hipStream_t stream) {
    if (bl.packed) {
        switch (bl.log2m) {
            case 9: return blu_one<T, 9, true>(bl, lds, P, blob, d_packed, clips, norms, tiles, n_tiles, d_out, stream);
            case 10: return blu_one<T, 10, true>(bl, lds, P, blob, d_packed, clips, norms, tiles, n_tiles, d_out, stream);
            case 11: return blu_one<T, 11, true>(bl, lds, P, blob, d_packed, clips, norms, tiles, n_tiles, d_out, stream);
            case 12: return blu_one<T, 12, true>(bl, lds, P, blob, d_packed, clips, norms, tiles, n_tiles, d_out, stream);
            default: return -1;
        }
    }
    switch (bl.log2m) {
        case 8: return blu_one<T, 8>(bl, lds, P, blob, d_packed, clips, norms, tiles, n_tiles, d_out, stream);
        case 9: return blu_one<T, 9>(bl, lds, P, blob, d_packed, clips, norms, tiles, n_tiles, d_out, stream);
        case 10: return blu_one<T, 10>(bl, lds, P, blob, d_packed, clips, norms, tiles, n_tiles, d_out, stream);
        case 11: return blu_one<T, 11>(bl, lds, P, blob, d_packed, clips, norms, tiles, n_tiles, d_out, stream);
        case 12: return blu_one<T, 12>(bl, lds, P, blob, d_packed, clips, norms, tiles, n_tiles, d_out, stream);
        case 13: return blu_one<T, 13>(bl, lds, P, blob, d_packed, clips, norms, tiles, n_tiles, d_out, stream);
        default: return -1;
    }
}
int blu(const blu::BluLayout &bl, size_t lds, int sample_kind, const PlanDev &P, const unsigned char *blob,
        const void *d_packed, const ClipDev *clips, const ClipNorm *norms, const Tile *tiles, long long n_tiles, double *d_out,
        hipStream_t stream) {
    if (sample_kind == 0) return blu_any<int16_t>(bl, lds, P, blob, d_packed, clips, norms, tiles, n_tiles, d_out, stream);
    if (sample_kind == 2) return blu_any<stereo16>(bl, lds, P, blob, d_packed, clips, norms, tiles, n_tiles, d_out, stream);
    return blu_any<double>(bl, lds, P, blob, d_packed, clips, norms, tiles, n_tiles, d_out, stream);
}

PAA_PHASE_READER(phase_blu)
}  // namespace launch
}  // namespace paa
