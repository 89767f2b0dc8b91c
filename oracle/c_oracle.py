"""ctypes loader for the plain-C oracle (oracle/paa_oracle.c).  TEST INFRASTRUCTURE ONLY: imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the pyaudioanalysis_amd package."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libpaa_oracle.so")
_lib = None


def build(force=False):
    src = os.path.join(_HERE, "paa_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-B" if force else "-s"], check=True, capture_output=True)
    return _SO


def available():
    return os.path.exists(_SO)


def lib():
    global _lib
    if _lib is None:
        L = ctypes.CDLL(_SO)
        L.paa_c_feature_extraction.restype = ctypes.c_longlong
        L.paa_c_feature_extraction.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.c_longlong, ctypes.c_double,
                                               ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                               ctypes.POINTER(ctypes.c_double)]
        _lib = L
    return _lib


def feature_extraction(signal, fs, window, step, deltas=True):
    """Same contract as ShortTermFeatures.feature_extraction (ShortTermFeatures.py:543); returns the matrix."""
    window, step = int(window), int(step)
    sig = np.ascontiguousarray(np.double(signal))
    n_frames = (len(sig) - window) // step + 1 if len(sig) >= window else 0
    if n_frames < 1:
        raise ValueError("need at least one array to concatenate")
    out = np.empty((68 if deltas else 34, n_frames))
    rc = lib().paa_c_feature_extraction(sig.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), len(sig), float(fs),
                                        window, step, int(deltas), out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
    if rc == -6:
        raise ValueError("chroma slot beyond the spectrum")
    if rc in (-7, -8):
        raise IndexError("chroma / mel table index out of range")
    assert rc == n_frames, rc
    return out
