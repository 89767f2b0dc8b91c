"""Per-phase cycle split of the fused three-pass kernel (csrc/kernels_wgr.hpp).  Needs the timing build:
    PAA_HIPCC_FLAGS=-DPAA_F800_TIMING python -c "from pyaudioanalysis_amd import _build as b; b.LIB=b.LIB.replace('.so','_timing.so'); print(b.build(force=True))"
and runs on the GPU box with PAA_HIP_LIBRARY=pyaudioanalysis_amd/libpaa_hip_timing.so."""
import ctypes, sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "scripts")]
from pyaudioanalysis_amd import _ffi
from bench import SHAPES, shape_input
lib = _ffi.lib(); _ffi.init(0)
names = ["load", "time domain", "pass 1 + write", "barrier + read + pass 2", "barrier + write + barrier + read + pass 3",
         "barrier + write + barrier + recombination", "sums + barrier + row + barrier", "scan + mel", "barrier + roll-off / spread / flux / chroma",
         "barrier + last mile + barrier + store", "(recombination: barrier + write + barrier)", "(recombination: reads + magnitudes)"]
for case in sys.argv[1:] or ["big_16000", "big_16000_1h"]:
    fs, W, S, seconds, clips, kind, mode, deltas = SHAPES[case]
    x, offsets = shape_input(case)
    d_in = _ffi.DeviceBuffer.from_host(x)
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
    tot = max(v[:12].sum(), 1.0)
    waves = max(v[15], 1) / 5
    print(case, plan.kernel_name, "frames", plan.total_frames, "waves", int(waves), "cycles/wave %.0f" % (tot / max(v[15], 1)))
    for nme, c in zip(names, v[:12]):
        if c: print("   %-50s %6.2f %%   %.0f cycles per wave and launch" % (nme, 100 * c / tot, c / max(v[15], 1)))
    plan.destroy()
