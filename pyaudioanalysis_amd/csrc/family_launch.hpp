// Launch entry points of the feature-kernel families.  Every family is its own translation unit (family_<name>.hip: its
// kernels are instantiated there and nowhere else, so the families compile in parallel); the host side of the library
// (paa_lib.hip and the lib_*.hpp units it is made of) sees only these functions and the families' host-side layout /
// selection code.  All of them queue ONE kernel on `stream` and return 0, or -1 when the launch failed (hipGetLastError
// has the reason).  sample_kind: 0 int16, 1 float64, 2 interleaved stereo int16 (summed in the kernels' loads).
#pragma once
#include <hip/hip_runtime.h>

#include "kernels_ct.hpp"
#include "kernels_fast.hpp"
#include "kernels_generic.hpp"
#include "kernels_mix.hpp"
#include "kernels_tri.hpp"
#include "kernels_blu.hpp"

namespace paa {
namespace wgr { struct WgrTab; }
namespace launch {

// kernels_fast.hpp: window 800, step 400 / 800, int16
int fast(const FastLaunch &fl, const PlanDev &P, const FastTables &ft, const void *d_packed, const ClipDev *clips,
         const ClipNorm *norms, const Tile *tiles, long long n_tiles, double *d_out, hipStream_t stream);
// kernels_ct.hpp: windows 2 RA RB (800, 640, 400, 320)
int ct(const ct::CtLaunch &cl, int sample_kind, const PlanDev &P, const unsigned char *blob, const void *d_packed,
       const ClipDev *clips, const ClipNorm *norms, const Tile *tiles, long long n_tiles, double *d_out, hipStream_t stream);
// kernels_tri.hpp: three-pass register FFT (2400, 2205, 1764, 1920, 1600, 1200, 1102 features, 551; 1024, 2048, 512); three units
int tri(const tri::TriLaunch &tl, int sample_kind, const PlanDev &P, const unsigned char *blob, const void *d_packed,
        const ClipDev *clips, const ClipNorm *norms, const Tile *tiles, long long n_tiles, double *d_out, hipStream_t stream);
int tri_part_a(const tri::TriLaunch &tl, int sample_kind, const PlanDev &P, const unsigned char *blob, const void *d_packed,
               const ClipDev *clips, const ClipNorm *norms, const Tile *tiles, long long n_tiles, double *d_out,
               hipStream_t stream);
int tri_part_b(const tri::TriLaunch &tl, int sample_kind, const PlanDev &P, const unsigned char *blob, const void *d_packed,
               const ClipDev *clips, const ClipNorm *norms, const Tile *tiles, long long n_tiles, double *d_out,
               hipStream_t stream);
int tri_part_c(const tri::TriLaunch &tl, int sample_kind, const PlanDev &P, const unsigned char *blob, const void *d_packed,
               const ClipDev *clips, const ClipNorm *norms, const Tile *tiles, long long n_tiles, double *d_out,
               hipStream_t stream);
// kernels_mix.hpp: in-place mixed-radix transform (every other length made of 2, 3, 5, 7, 11, 13)
int mix(const mix::MixLayout &ml, size_t lds, int sample_kind, const PlanDev &P, const unsigned char *blob,
        const void *d_packed, const ClipDev *clips, const ClipNorm *norms, const Tile *tiles, long long n_tiles, double *d_out,
        hipStream_t stream);
// kernels_blu.hpp: Bluestein convolution on power-of-two transforms (lengths with a prime factor above 13)
int blu(const blu::BluLayout &bl, size_t lds, int sample_kind, const PlanDev &P, const unsigned char *blob,
        const void *d_packed, const ClipDev *clips, const ClipNorm *norms, const Tile *tiles, long long n_tiles, double *d_out,
        hipStream_t stream);
// kernels_wgr.hpp: workgroup-wide three-pass register transform with fused features (16 000- / 8 000-sample windows); `runs`:
// runs of consecutive frames, one workgroup walks runs b, b + grid, ...
int wgr(int shape_id, int sample_kind, int mode, const PlanDev &P, const void *d_packed, const ClipDev *clips, const ClipNorm *norms,
        const Tile *runs, long long n_runs, int num_cu, const wgr::WgrTab *d_tab, double *d_out, hipStream_t stream);
// kernels_wgs.hpp: real-input split of the long even windows r0 x q samples (12 / 6 x 3675: 44 100, 22 050; 12 / 8 / 6 x 4000: 48 000, 32 000, 24 000),
// three register passes per sub-transform; `tasks`: (frame, task type) records handed out through `counter` (zero at the first launch; the kernel
// leaves it at zero); magnitudes go to the frames' UNIT-MAJOR rows of `spec` (spectrogram plans: d_out, natural order), the time-domain partials of
// a frame to `tfeat`, the units' sum X / sum (k + 1) X / max X to `psum` (4 doubles per row and unit, r0 / 2 units)
int wgs(int r0, int q, int sample_kind, const PlanDev &P, const void *d_packed, const ClipDev *clips, const ClipNorm *norms,
        const wg::FrameRef *tasks, int n_tasks, int *counter, int num_cu, double *spec, double *tfeat, double *psum, double *d_out,
        hipStream_t stream);
// ... and the features of those frames from the unit-major rows (one workgroup per frame)
int wgs_feat(int r0, int q, const PlanDev &P, const wg::FrameRef *frames, int n_frames, const ClipDev *clips, const double *spec,
             const double *tfeat, const double *psum, double *d_out, hipStream_t stream);
// kernels_generic.hpp: Stockham passes in LDS (what is left)
int generic(const GenLayout &gl, size_t lds, int sample_kind, const PlanDev &P, const unsigned char *blob,
            const void *d_packed, const ClipDev *clips, const ClipNorm *norms, const Tile *tiles, long long n_tiles,
            double *d_out, hipStream_t stream);


// timing builds (-DPAA_F800_TIMING / _TRACE): per-unit readers of the kernels' phase-cycle counters (kernels_fast.hpp:
// PAA_PHASE_READER); no-ops otherwise
int phase_fast(unsigned long long *acc16, unsigned long long *trace, int max_waves);
int phase_ct(unsigned long long *acc16, unsigned long long *trace, int max_waves);
int phase_tri_a(unsigned long long *acc16, unsigned long long *trace, int max_waves);
int phase_tri_b(unsigned long long *acc16, unsigned long long *trace, int max_waves);
int phase_tri_c(unsigned long long *acc16, unsigned long long *trace, int max_waves);
int phase_rmg(unsigned long long *acc16, unsigned long long *trace, int max_waves);
int phase_blu(unsigned long long *acc16, unsigned long long *trace, int max_waves);
int phase_wgr(unsigned long long *acc16, unsigned long long *trace, int max_waves);

}  // namespace launch
}  // namespace paa
