import ctypes, sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
os.environ["PAA_HIP_FAST_WS"] = "1"
from pyaudioanalysis_amd import _ffi
from synth import synth_clip
lib = _ffi.lib(); _ffi.init(0)
x = synth_clip(2, 3600 * 16000)
d_in = _ffi.DeviceBuffer.from_host(x)
plan = _ffi.Plan(np.array([0, len(x)], dtype=np.int64), 16000, 800, 400, deltas=False)
print(plan.kernel_name)
d_out = _ffi.DeviceBuffer(plan.out_doubles * 8)
for _ in range(3): plan.execute(d_in, d_out)
_ffi.sync()
buf = (ctypes.c_uint64 * 16)()
lib.paa_debug_phase_cycles(buf)
for _ in range(5): plan.execute(d_in, d_out)
_ffi.sync()
lib.paa_debug_phase_cycles(buf)
v = np.array(list(buf), dtype=np.float64)
nw = v[15]
print("waves", int(nw), "producer work %.0f wait %.0f | consumer work %.0f wait %.0f  (cycles per wave)" % (v[0]/(nw/2), v[2]/(nw/2), v[1]/(nw/2), v[3]/(nw/2)))
