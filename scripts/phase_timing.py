"""Per-phase cycle split of st_fast_800 (needs a build with PAA_HIPCC_FLAGS=-DPAA_F800_TIMING)."""
import ctypes, sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from pyaudioanalysis_amd import _ffi
from synth import synth_clip
lib = _ffi.lib(); _ffi.init(0)
x = synth_clip(2, 3600 * 16000)
d_in = _ffi.DeviceBuffer.from_host(x)
plan = _ffi.Plan(np.array([0, len(x)], dtype=np.int64), 16000, 800, 400, deltas=False)
d_out = _ffi.DeviceBuffer(plan.out_doubles * 8)
for _ in range(3): plan.execute(d_in, d_out)
_ffi.sync()
buf = (ctypes.c_uint64 * 16)()
lib.paa_debug_phase_cycles(buf)
for _ in range(5): plan.execute(d_in, d_out)
_ffi.sync()
lib.paa_debug_phase_cycles(buf)
v = np.array(list(buf), dtype=np.float64)
names = ["stage", "time-domain", "pass1 dft25", "exchange", "pass2+post", "sweepA+entropy", "spread/flux/rolloff", "mel", "chroma", "dct/fv", "store"]
tot = v[:11].sum()
print("waves", int(v[15]), "cycles/wave %.0f" % (tot / max(v[15], 1)))
for n, c in zip(names, v[:11]): print("%-22s %6.2f %%   %.0f cycles/wave-iteration" % (n, 100 * c / tot, c / max(v[15], 1) / 37.0))
