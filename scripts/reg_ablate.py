"""Ablation timing of the register-FFT kernel on config 5 (PAA_KERNEL_DEBUG: 1 = no time-domain stage, 2 = no feature stage).
Measured in round 2 (float64 mono input, 60 k frames): both FFT passes + exchange 0.34 ms, time-domain stage 0.13 ms, feature
stage 0.23 ms; the statistics kernels over the 212 MB float64 clip add 0.10 ms to the step."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from pyaudioanalysis_amd import _ffi
from synth import synth_clip
lib = _ffi.lib(); _ffi.init(0)
fs = 44100
xs = synth_clip(5, 100 * fs, fs=fs, stereo=True)
mono = np.ascontiguousarray(np.tile((xs[:, 1] / 2) + (xs[:, 0] / 2), 6))
d_in = _ffi.DeviceBuffer.from_host(mono)
for mode in (0, 1):
    plan = _ffi.Plan(np.array([0, len(mono)], dtype=np.int64), fs, 1102, 441, deltas=False, sample_kind=1, mode=mode)
    d_out = _ffi.DeviceBuffer(plan.out_doubles * 8)
    for _ in range(3): plan.execute(d_in, d_out)
    _ffi.sync()
    _ffi.check(lib.paa_prof_enable(1))
    _ffi.check(lib.paa_timer_start())
    for _ in range(10): plan.execute(d_in, d_out)
    ms = ctypes.c_float(); _ffi.check(lib.paa_timer_stop(ctypes.byref(ms)))
    kms = ctypes.c_double(); kn = ctypes.c_int64()
    _ffi.check(lib.paa_prof_read(ctypes.byref(kms), ctypes.byref(kn))); _ffi.check(lib.paa_prof_enable(0))
    print("   feature kernel alone %.4f ms" % (kms.value / max(1, kn.value)))
    print("debug", os.environ.get("PAA_KERNEL_DEBUG", "0"), plan.kernel_name, "%.4f ms" % (ms.value / 10), "%.3g frames/s" % (plan.total_frames / (ms.value / 10 * 1e-3)))
