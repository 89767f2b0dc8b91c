"""Where the host-to-host time of the drop-in API goes (1-hour and 10-minute clips): plan build, H2D, kernels, D2H."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from pyaudioanalysis_amd import _ffi, ShortTermFeatures
from synth import synth_clip
lib = _ffi.lib(); _ffi.init(0)
for seconds in (600, 3600):
    x = synth_clip(2, seconds * 16000)
    offs = np.array([0, len(x)], dtype=np.int64)
    def best(fn, reps=5):
        fn(); b = 1e9
        for _ in range(reps):
            t0 = time.perf_counter(); fn(); b = min(b, time.perf_counter() - t0)
        return b * 1e3
    t_plan = best(lambda: _ffi.Plan(offs, 16000, 800, 400, deltas=False).destroy())
    d_in = _ffi.DeviceBuffer(x.nbytes)
    t_h2d = best(lambda: _ffi.check(lib.paa_memcpy_h2d(d_in.ptr, x.ctypes.data, x.nbytes)))
    plan = _ffi.Plan(offs, 16000, 800, 400, deltas=False)
    d_out = _ffi.DeviceBuffer(plan.out_doubles * 8)
    def run(): plan.execute(d_in, d_out); _ffi.sync()
    t_exec = best(run)
    out = np.empty(plan.out_doubles)
    t_d2h = best(lambda: _ffi.check(lib.paa_memcpy_d2h(out.ctypes.data, d_out.ptr, out.nbytes)))
    t_alloc = best(lambda: np.empty((34, plan.total_frames)))
    t_all = best(lambda: ShortTermFeatures.feature_extraction(x, 16000, 800, 400, deltas=False))
    print("%4d s clip: plan %.3f | h2d %.3f (%.1f GB/s) | kernels %.3f | d2h %.3f (%.1f GB/s) | np.empty %.3f | sum %.3f | API call %.3f ms"
          % (seconds, t_plan, t_h2d, x.nbytes / t_h2d / 1e6, t_exec, t_d2h, out.nbytes / t_d2h / 1e6, t_alloc,
             t_plan + t_h2d + t_exec + t_d2h, t_all))
