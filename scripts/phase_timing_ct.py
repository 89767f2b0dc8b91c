"""Per-phase cycle split of the st_ct kernels (needs a build with -DPAA_F800_TIMING: build it in the container with
    PAA_HIPCC_FLAGS=-DPAA_F800_TIMING python -c "from pyaudioanalysis_amd import _build as b; b.LIB=b.LIB.replace('.so','_timing.so'); print(b.build(force=True))"
and run on the GPU box with PAA_HIP_LIBRARY=pyaudioanalysis_amd/libpaa_hip_timing.so)."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "scripts"))
from pyaudioanalysis_amd import _ffi
from synth import synth_clip
from kernel_loop import CASES
lib = _ffi.lib(); _ffi.init(0)
names = ["load+wait", "time-domain", "pass1", "exchange", "pass2+mag", "sweepA+entropy", "spread/flux/rolloff", "mel", "chroma",
         "dct/fv", "store"]
for case in sys.argv[1:] or ["ct_640", "ct_800_f64", "ct_800_stereo", "ct_400", "ct_320"]:
    fs, W, S, seconds, clips, kind, mode, deltas = CASES[case]
    n = min(seconds, 100) * fs
    if kind == 0:
        x = synth_clip(5, n, fs)
    else:
        xs = synth_clip(5, n, fs, stereo=True)
        x = (xs[:, 1] / 2) + (xs[:, 0] / 2) if kind == 1 else xs
    reps = -(-seconds // min(seconds, 100)) * clips
    x = np.ascontiguousarray(np.tile(x, reps) if x.ndim == 1 else np.tile(x, (reps, 1)))
    offsets = np.arange(clips + 1, dtype=np.int64) * (x.shape[0] // clips)
    d_in = _ffi.DeviceBuffer.from_host(x.reshape(-1))
    plan = _ffi.Plan(offsets, fs, W, S, deltas=bool(deltas), sample_kind=kind, mode=mode)
    d_out = _ffi.DeviceBuffer(plan.out_doubles * 8)
    for _ in range(3): plan.execute(d_in, d_out)
    _ffi.sync()
    buf = (ctypes.c_uint64 * 16)()
    lib.paa_debug_phase_cycles(buf)
    for _ in range(5): plan.execute(d_in, d_out)
    _ffi.sync()
    lib.paa_debug_phase_cycles(buf)
    v = np.array(list(buf), dtype=np.float64)
    tot = max(v[:11].sum(), 1.0)
    print(case, plan.kernel_name, "frames", plan.total_frames, "waves", int(v[15]) // 5, "cycles/wave %.0f" % (tot / max(v[15], 1)))
    for nme, c in zip(names, v[:11]):
        print("   %-22s %6.2f %%   %.0f cycles/wave" % (nme, 100 * c / tot, c / max(v[15], 1)))
    plan.destroy()
