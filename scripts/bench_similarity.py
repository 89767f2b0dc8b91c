"""Device-resident timing of the self-similarity / thumbnail kernels.  Prints one JSON line.
Algorithmic bytes: similarity writes 8 T^2 (reads 8 F T); the filter reads 8 T^2 and writes 8 R^2 twice
(diagonal sums, then masks)."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pyaudioanalysis_amd import _ffi
import ctypes as C

lib = _ffi.lib(); _ffi.init(0)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
F, M = 68, 20
rng = np.random.default_rng(1)
feats = rng.standard_normal((F, T)).cumsum(axis=1)
d_f = _ffi.DeviceBuffer.from_host(feats)
d_sim = _ffi.DeviceBuffer(T * T * 8)
R = T - M + 1
d_filt = _ffi.DeviceBuffer(R * R * 8)
pos = np.zeros(2, dtype=np.int64)


def timed(fn, reps=10):
    for _ in range(3): fn()
    _ffi.sync()
    ms = C.c_float()
    _ffi.check(lib.paa_timer_start())
    for _ in range(reps): fn()
    _ffi.check(lib.paa_timer_stop(C.byref(ms)))
    return ms.value / reps


t_sim = timed(lambda: _ffi.check(lib.paa_dev_self_similarity(d_f.ptr, F, T, T, d_sim.ptr)))
t_fil = timed(lambda: _ffi.check(lib.paa_dev_thumbnail_filter(d_sim.ptr, T, M, 10.0, 0.0, 1.0, d_filt.ptr, _ffi.as_i64p(pos))))
out = {"T": T, "F": F, "m_filter": M,
       "self_similarity_ms": t_sim, "self_similarity_GBps": 8.0 * T * T / t_sim / 1e6,
       "self_similarity_fp64_TFLOPs": 2.0 * T * T * F / t_sim / 1e9,
       "thumbnail_filter_ms": t_fil, "thumbnail_filter_GBps": (8.0 * T * T + 3 * 8.0 * R * R) / t_fil / 1e6}
print(json.dumps(out))
