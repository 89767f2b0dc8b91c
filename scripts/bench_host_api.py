"""Host NumPy in -> host NumPy out timing of the drop-in API (PCIe inclusive).  Prints frames/s for a few clip lengths."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from pyaudioanalysis_amd import ShortTermFeatures, _ffi
from synth import fast_noise_clip
_ffi.lib(); _ffi.init(0)
for seconds in (30, 600, 3600):
    x = fast_noise_clip(1, seconds * 16000)
    for deltas in (False, True):
        for _ in range(3): F, _n = ShortTermFeatures.feature_extraction(x, 16000, 800, 400, deltas)
        t0 = time.perf_counter(); reps = 10
        for _ in range(reps): F, _n = ShortTermFeatures.feature_extraction(x, 16000, 800, 400, deltas)
        dt = (time.perf_counter() - t0) / reps
        print("pin=%s  %5d s clip  rows %d  %.3f ms  %.1f M frames/s  (in %.1f MB, out %.1f MB)" % (
            os.environ.get("PAA_HIP_PIN", "1"), seconds, F.shape[0], 1e3 * dt, F.shape[1] / dt / 1e6, x.nbytes / 1e6, F.nbytes / 1e6))
