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
        f64p = ctypes.POINTER(ctypes.c_double)
        L.paa_c_spectrogram.restype = ctypes.c_longlong
        L.paa_c_spectrogram.argtypes = [f64p, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, f64p]
        L.paa_c_chromagram.restype = ctypes.c_longlong
        L.paa_c_chromagram.argtypes = [f64p, ctypes.c_longlong, ctypes.c_double, ctypes.c_int, ctypes.c_int, f64p]
        L.paa_c_mid_statistics.restype = ctypes.c_longlong
        L.paa_c_mid_statistics.argtypes = [f64p, ctypes.c_int, ctypes.c_longlong, ctypes.c_longlong, ctypes.c_longlong, f64p]
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


def _f64(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def spectrogram(signal, window, step):
    """ShortTermFeatures.spectrogram (:389-452) without the axes; returns the (rows, window // 2) matrix."""
    window, step = int(window), int(step)
    sig = np.ascontiguousarray(np.double(signal))
    rows = int((len(sig) - window) / step) + 1
    out = np.zeros((max(rows, 1), window // 2))
    got = lib().paa_c_spectrogram(_f64(sig), len(sig), window, step, _f64(out))
    assert got == rows, (got, rows)
    return out


def chromagram(signal, fs, window, step):
    """ShortTermFeatures.chromagram (:324-386) without the axes; returns the (rows, 12) matrix."""
    window, step = int(window), int(step)
    sig = np.ascontiguousarray(np.double(signal))
    rows = int((len(sig) - step - window) / step) + 1
    out = np.zeros((max(rows, 1), 12))
    got = lib().paa_c_chromagram(_f64(sig), len(sig), float(fs), window, step, _f64(out))
    assert got == rows, (got, rows)
    return out


def mid_statistics(short_features, ratio, step_ratio):
    """MidTermFeatures.py:110-126 on a (F, T) matrix."""
    st = np.ascontiguousarray(short_features, dtype=np.float64)
    F, T = st.shape
    M = (T + step_ratio - 1) // step_ratio
    out = np.empty((2 * F, M))
    got = lib().paa_c_mid_statistics(_f64(st), F, T, int(ratio), int(step_ratio), _f64(out))
    assert got == M, (got, M)
    return out
