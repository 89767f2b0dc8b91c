"""Drop-in for pyAudioAnalysis.MidTermFeatures.mid_feature_extraction (reference:
pyAudioAnalysis/MidTermFeatures.py:87-127) plus a batched many-clip form.  HIP only, no CPU path.
"""
import numpy as np

from . import ShortTermFeatures, _ffi

eps = 0.00000001                          # MidTermFeatures.py:13


def _ratios(mid_window, mid_step, short_window, short_step):
    """Python round() (banker's) exactly where the reference applies it (:100-102).

    feature_extraction int()-truncates short_window/short_step internally (:563-564) but the ratios
    use the caller's (possibly float) values, so they do here too.
    """
    mid_window_ratio = round((mid_window - (short_window - short_step)) / short_step)
    mt_step_ratio = int(round(mid_step / short_step))
    return int(mid_window_ratio), mt_step_ratio


def _mid_names(short_names):
    return [s + "_mean" for s in short_names] + [s + "_std" for s in short_names]   # :113-114


def mid_feature_extraction(signal, sampling_rate, mid_window, mid_step, short_window, short_step):
    """Mid-term feature extraction (reference :87-127).

    RETURNS (mid_features [136 x M], short_features [68 x T], mid_feature_names)
    """
    ratio, step_ratio = _ratios(mid_window, mid_step, short_window, short_step)
    if step_ratio < 1:
        raise ValueError("mid_step / short_step rounds to 0: the reference never terminates "
                         "(MidTermFeatures.py:102,124)")
    window, step = int(short_window), int(short_step)
    kind, sig = _ffi.classify_signal(signal)
    short_names = ShortTermFeatures._feature_names(True)            # deltas always on (:93-95)
    lib = _ffi.lib()
    T = int(lib.paa_num_frames(sig.shape[0], window, step)) if window >= 1 and step >= 1 else 0
    if T < 1:
        raise ValueError("need at least one array to concatenate")  # ShortTermFeatures.py:684
    M = int(lib.paa_num_mid_windows(T, step_ratio))
    st = np.empty((len(short_names), T), dtype=np.float64)
    mid = np.empty((2 * len(short_names), M), dtype=np.float64)
    if kind == 0:
        rc = lib.paa_mid_features_i16(_ffi.as_i16p(sig), sig.shape[0], float(sampling_rate), window, step,
                                      ratio, step_ratio, _ffi.as_f64p(mid), _ffi.as_f64p(st))
    else:
        rc = lib.paa_mid_features_f64(_ffi.as_f64p(sig), sig.shape[0], float(sampling_rate), window, step,
                                      ratio, step_ratio, _ffi.as_f64p(mid), _ffi.as_f64p(st))
    _ffi.check(rc)
    return mid, st, _mid_names(short_names)


def mid_feature_extraction_batch(signals, sampling_rate, mid_window, mid_step, short_window, short_step,
                                 return_short=False):
    """Many int16 clips in one launch -> list of (136, M_c) arrays (+ list of (68, T_c)), names."""
    ratio, step_ratio = _ratios(mid_window, mid_step, short_window, short_step)
    if step_ratio < 1:
        raise ValueError("mid_step / short_step rounds to 0: the reference never terminates")
    window, step = int(short_window), int(short_step)
    clips = [np.ascontiguousarray(s, dtype=np.int16) for s in signals]
    if not clips:
        raise ValueError("need at least one clip")
    short_names = ShortTermFeatures._feature_names(True)
    F = len(short_names)
    lib = _ffi.lib()
    lens = np.array([c.shape[0] for c in clips], dtype=np.int64)
    offsets = np.zeros(len(clips) + 1, dtype=np.int64)
    np.cumsum(lens, out=offsets[1:])
    packed = np.concatenate(clips) if len(clips) > 1 else clips[0]
    T = np.array([int(lib.paa_num_frames(int(n), window, step)) for n in lens], dtype=np.int64)
    if np.any(T < 1):
        raise ValueError("need at least one array to concatenate")
    M = np.array([int(lib.paa_num_mid_windows(int(t), step_ratio)) for t in T], dtype=np.int64)
    mid_off = np.zeros(len(clips), dtype=np.int64)
    np.cumsum(2 * F * M[:-1], out=mid_off[1:])
    mid = np.empty(int(2 * F * M.sum()), dtype=np.float64)
    st = st_off = None
    if return_short:
        st_off = np.zeros(len(clips), dtype=np.int64)
        np.cumsum(F * T[:-1], out=st_off[1:])
        st = np.empty(int(F * T.sum()), dtype=np.float64)
    _ffi.check(lib.paa_mid_features_batch_i16(
        _ffi.as_i16p(packed), _ffi.as_i64p(offsets), len(clips), float(sampling_rate), window, step,
        ratio, step_ratio, _ffi.as_f64p(mid), _ffi.as_i64p(mid_off),
        _ffi.as_f64p(st) if return_short else None, _ffi.as_i64p(st_off) if return_short else None))
    mids = [mid[int(o):int(o) + 2 * F * int(m)].reshape(2 * F, int(m)) for o, m in zip(mid_off, M)]
    if return_short:
        sts = [st[int(o):int(o) + F * int(t)].reshape(F, int(t)) for o, t in zip(st_off, T)]
        return mids, sts, _mid_names(short_names)
    return mids, _mid_names(short_names)
