"""GPU drop-ins for the two matrix-heavy functions of pyAudioAnalysis.audioSegmentation.

    self_similarity_matrix(feature_vectors)                      audioSegmentation.py:40-55
    music_thumbnailing(signal, sampling_rate, short_window=1.0,  audioSegmentation.py:1096-1190
                       short_step=0.5, thumb_size=10.0, limit_1=0, limit_2=1)

Same names, argument meaning and return values as the reference.  Everything numeric runs in libpaa_hip.so
(standardisation, FP64 matrix-core Gram matrix, diagonal filter, masks, arg-max); there is no CPU fallback.
`music_thumbnailing` keeps the short-term features and the similarity matrix in HBM: only the filtered matrix it
returns comes back to the host.  The rest of audioSegmentation (HMM / SVM segmentation, diarisation, plotting) is
out of scope (control plane / third-party models).
"""
import ctypes as C

import numpy as np

from . import _ffi
from . import audioBasicIO


def self_similarity_matrix(feature_vectors):
    """(nDims x nVectors) feature matrix -> (nVectors x nVectors) cosine self-similarity of the standardised
    columns (audioSegmentation.py:40-55)."""
    F = np.ascontiguousarray(np.asarray(feature_vectors, dtype=np.float64))
    if F.ndim != 2 or F.shape[0] < 1 or F.shape[1] < 1:
        raise ValueError("feature_vectors must be a non-empty (nDims x nVectors) matrix")
    sim = np.empty((F.shape[1], F.shape[1]))
    _ffi.check(_ffi.lib().paa_self_similarity_f64(_ffi.as_f64p(F), F.shape[0], F.shape[1], _ffi.as_f64p(sim)))
    return sim


def _grow_thumbnail(filtered, rows, cols, m_filter):
    # extend the arg-max cell along its diagonal until it spans m_filter cells (:1167-1182)
    i1 = i2 = int(rows)
    j1 = j2 = int(cols)
    last_r, last_c = filtered.shape[0] - 2, filtered.shape[1] - 2
    while i2 - i1 < m_filter:
        if i1 <= 0 or j1 <= 0 or i2 >= last_r or j2 >= last_c:
            break
        if filtered[i1 - 1, j1 - 1] > filtered[i2 + 1, j2 + 1]:
            i1, j1 = i1 - 1, j1 - 1
        else:
            i2, j2 = i2 + 1, j2 + 1
    return i1, i2, j1, j2


def music_thumbnailing(signal, sampling_rate, short_window=1.0, short_step=0.5, thumb_size=10.0,
                       limit_1=0, limit_2=1):
    """Returns (A1, A2, B1, B2, filtered similarity matrix): start / end of the two thumbnail instances in seconds
    (audioSegmentation.py:1096-1190)."""
    lib = _ffi.lib()
    signal = audioBasicIO.stereo_to_mono(signal)
    kind, sig = _ffi.classify_signal(signal)
    window, step = int(sampling_rate * short_window), int(sampling_rate * short_step)     # int() of :563-564
    n_frames = int(lib.paa_num_frames(len(sig), window, step)) if window > 0 and step > 0 else 0
    if n_frames < 1:
        raise ValueError("need at least one array to concatenate")          # ShortTermFeatures.py:684
    m_filter = int(round(thumb_size / short_step))                           # :1142
    R = int(lib.paa_thumbnail_rows(n_frames, m_filter))
    if R < 1:
        raise ValueError("%d feature vectors are fewer than the thumbnail filter length %d (the reference's "
                         "convolve2d silently swaps its operands in that case)" % (n_frames, m_filter))
    offsets = np.array([0, len(sig)], dtype=np.int64)
    plan = _ffi.Plan(offsets, sampling_rate, window, step, deltas=True, sample_kind=kind)
    d_in = d_st = d_sim = d_filt = None
    try:
        d_in = _ffi.DeviceBuffer.from_host(sig)
        d_st = _ffi.DeviceBuffer(plan.out_doubles * 8)
        plan.execute(d_in, d_st)
        d_sim = _ffi.DeviceBuffer(n_frames * n_frames * 8)
        _ffi.check(lib.paa_dev_self_similarity(d_st.ptr, plan.F, n_frames, n_frames, d_sim.ptr))
        d_filt = _ffi.DeviceBuffer(R * R * 8)
        pos = np.zeros(2, dtype=np.int64)
        _ffi.check(lib.paa_dev_thumbnail_filter(d_sim.ptr, n_frames, m_filter, 5.0 / short_step, float(limit_1),
                                                float(limit_2), d_filt.ptr, _ffi.as_i64p(pos)))
        filtered = d_filt.to_host(np.float64, R * R).reshape(R, R)
    finally:
        for b in (d_in, d_st, d_sim, d_filt):
            if b is not None:
                b.free()
        plan.destroy()
    i1, i2, j1, j2 = _grow_thumbnail(filtered, pos[0], pos[1], m_filter)
    return short_step * i1, short_step * i2, short_step * j1, short_step * j2, filtered
