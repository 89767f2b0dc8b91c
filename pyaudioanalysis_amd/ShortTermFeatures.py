"""Drop-in for pyAudioAnalysis.ShortTermFeatures (reference: pyAudioAnalysis/ShortTermFeatures.py).

Same names, argument meaning, return layout and exception types as the reference for
feature_extraction (:543-685), spectrogram (:389-452) and chromagram (:324-386); the arithmetic
runs in hand-written HIP kernels on the MI355X through libpaa_hip.so.  There is no CPU path.
"""
import sys

import numpy as np

from . import _ffi

eps = sys.float_info.epsilon            # ShortTermFeatures.py:11

_CHROMA_NAMES = ['A', 'A#', 'B', 'C', 'C#', 'D', 'D#', 'E', 'F', 'F#', 'G', 'G#']   # :283-284


def _feature_names(deltas=True):
    """Exact strings of ShortTermFeatures.py:590-604."""
    names = ["zcr", "energy", "energy_entropy"]
    names += ["spectral_centroid", "spectral_spread"]
    names.append("spectral_entropy")
    names.append("spectral_flux")
    names.append("spectral_rolloff")
    names += ["mfcc_{0:d}".format(i) for i in range(1, 14)]
    names += ["chroma_{0:d}".format(i) for i in range(1, 13)]
    names.append("chroma_std")
    if deltas:
        names = names + ["delta " + f for f in names]
    return names


def _spec_call(lib, stem, kind, sig):
    """entry point and sample pointer of spectrogram / chromagram for a classified signal: int16 mono, float64, or
    interleaved stereo int16 (summed on the device: the float64 mono copy of audioBasicIO.stereo_to_mono is never made)"""
    if kind == 0:
        return getattr(lib, "paa_%s_i16" % stem), _ffi.as_i16p(sig)
    if kind == 2:
        return getattr(lib, "paa_%s_stereo_i16" % stem), _ffi.as_i16p(sig)
    return getattr(lib, "paa_%s_f64" % stem), _ffi.as_f64p(sig)


def feature_extraction(signal, sampling_rate, window, step, deltas=True):
    """Short-term windowing and feature extraction (reference :543-685).

    ARGUMENTS
        signal:         the input signal samples (1-D; int16 PCM or anything np.double() accepts; an (n, 2)
                        int16 array is taken as stereo and reduced to mono on the device like
                        audioBasicIO.stereo_to_mono)
        sampling_rate:  the sampling freq (in Hz)
        window:         the short-term window size (in samples; floats are int()-truncated, :563)
        step:           the short-term window step (in samples)
        deltas:         (opt) True/False if delta features are to be computed
    RETURNS
        features (numpy.ndarray):   (n_feats x numOfShortTermWindows) float64, C-contiguous
        feature_names (list of str)
    """
    window = int(window)
    step = int(step)
    kind, sig = _ffi.classify_signal(signal)
    names = _feature_names(deltas)
    lib = _ffi.lib()
    n_frames = int(lib.paa_num_frames(sig.shape[0], window, step)) if window >= 1 and step >= 1 else 0
    if n_frames < 1:
        # np.concatenate([]) at ShortTermFeatures.py:684
        raise ValueError("need at least one array to concatenate")
    out = _ffi.result_array((len(names), n_frames))
    if kind == 0:
        rc = lib.paa_st_features_i16(_ffi.as_i16p(sig), sig.shape[0], float(sampling_rate), window, step,
                                     1 if deltas else 0, _ffi.as_f64p(out))
    elif kind == 2:       # (n, 2) int16: stereo_to_mono fused on the device
        rc = lib.paa_st_features_stereo_i16(_ffi.as_i16p(sig), sig.shape[0], float(sampling_rate), window, step,
                                            1 if deltas else 0, _ffi.as_f64p(out))
    else:
        rc = lib.paa_st_features_f64(_ffi.as_f64p(sig), sig.shape[0], float(sampling_rate), window, step,
                                     1 if deltas else 0, _ffi.as_f64p(out))
    _ffi.check(rc)
    return out, names


def _batch_clips(signals):
    """-> (contiguous 1-D clips of ONE dtype, True for int16 / False for float64)"""
    arrays = [np.asarray(s) for s in signals]
    if not arrays:
        raise ValueError("need at least one clip")
    if any(a.ndim != 1 for a in arrays):
        raise ValueError("batched clips must be one-dimensional (reduce stereo with audioBasicIO.stereo_to_mono first)")
    if all(a.dtype == np.int16 for a in arrays):
        return [np.ascontiguousarray(a) for a in arrays], True
    return [np.ascontiguousarray(np.double(a)) for a in arrays], False


def feature_extraction_batch(signals, sampling_rate, window, step, deltas=True):
    """Many clips in one launch: list of 1-D arrays -> list of (F, T_c) arrays + names.  int16 clips travel as int16;
    if any clip has another dtype the whole batch goes through np.double() like the reference's :567 (8 B/sample).

    This is what MidTermFeatures.directory_feature_extraction (:140-221) does one file at a time.
    """
    window = int(window)
    step = int(step)
    clips, as_i16 = _batch_clips(signals)
    names = _feature_names(deltas)
    lib = _ffi.lib()
    lens = np.array([c.shape[0] for c in clips], dtype=np.int64)
    offsets = np.zeros(len(clips) + 1, dtype=np.int64)
    np.cumsum(lens, out=offsets[1:])
    packed = np.concatenate(clips) if len(clips) > 1 else clips[0]
    frames = np.array([int(lib.paa_num_frames(int(n), window, step)) for n in lens], dtype=np.int64)
    if np.any(frames < 1):
        raise ValueError("need at least one array to concatenate")
    F = len(names)
    out_off = np.zeros(len(clips), dtype=np.int64)
    np.cumsum(F * frames[:-1], out=out_off[1:])
    out = _ffi.result_array((int(F * frames.sum()),))
    fn = lib.paa_st_features_batch_i16 if as_i16 else lib.paa_st_features_batch_f64
    _ffi.check(fn(_ffi.as_i16p(packed) if as_i16 else _ffi.as_f64p(packed), _ffi.as_i64p(offsets), len(clips),
                  float(sampling_rate), window, step, 1 if deltas else 0, _ffi.as_f64p(out), _ffi.as_i64p(out_off)))
    res = [out[int(o):int(o) + F * int(t)].reshape(F, int(t)) for o, t in zip(out_off, frames)]
    return res, names


def spectrogram(signal, sampling_rate, window, step, plot=False, show_progress=False):
    """Short-term FFT magnitude (reference :389-452).  Returns (specgram [T x num_fft], time_axis, freq_axis).

    Frame i starts at sample window + i*step and trailing rows stay zero, as in the reference
    (:413-422).  `show_progress` is accepted and ignored; the reference's print of the shape (:451) is kept.
    """
    window = int(window)
    step = int(step)
    kind, sig = _ffi.classify_signal(signal)
    lib = _ffi.lib()
    num_fft = int(window / 2)
    rows = int(lib.paa_spectrogram_rows(sig.shape[0], window, step, None)) if window >= 1 and step >= 1 else 0
    if rows < 1:
        raise ValueError("negative dimensions are not allowed")     # np.zeros((<=0, num_fft)) at :413
    specgram = _ffi.result_array((rows, num_fft))
    fn, ptr = _spec_call(lib, "spectrogram", kind, sig)
    _ffi.check(fn(ptr, sig.shape[0], float(sampling_rate), window, step, _ffi.as_f64p(specgram)))
    freq_axis = [float((f + 1) * sampling_rate) / (2 * num_fft) for f in range(specgram.shape[1])]
    time_axis = [float(t * step) / sampling_rate for t in range(specgram.shape[0])]
    if plot:
        _plot_image(specgram.transpose()[::-1, :], "freq (Hz)")
    print(specgram.shape)
    return specgram, time_axis, freq_axis


def chromagram(signal, sampling_rate, window, step, plot=False, show_progress=False):
    """Chromagram (reference :324-386).  Returns (chromogram [T x 12], time_axis, freq_axis)."""
    window = int(window)
    step = int(step)
    kind, sig = _ffi.classify_signal(signal)
    lib = _ffi.lib()
    rows = int(lib.paa_chromagram_rows(sig.shape[0], window, step, None)) if window >= 1 and step >= 1 else 0
    if rows < 1:
        raise ValueError("negative dimensions are not allowed")     # np.zeros at :347
    chromogram = np.empty((rows, 12), dtype=np.float64)
    fn, ptr = _spec_call(lib, "chromagram", kind, sig)
    _ffi.check(fn(ptr, sig.shape[0], float(sampling_rate), window, step, _ffi.as_f64p(chromogram)))
    freq_axis = list(_CHROMA_NAMES)
    time_axis = [(t * step) / sampling_rate for t in range(chromogram.shape[0])]
    if plot:
        _plot_image(chromogram.transpose()[::-1, :], "chroma")
    return chromogram, time_axis, freq_axis


def _plot_image(img, ylabel):
    import matplotlib.pyplot as plt
    fig, ax = plt.subplots()
    im = plt.imshow(img, aspect="auto")
    ax.set_xlabel('time (frames)')
    ax.set_ylabel(ylabel)
    im.set_cmap('jet')
    plt.colorbar()
    plt.show()
