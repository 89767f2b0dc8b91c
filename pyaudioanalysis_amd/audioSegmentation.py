"""GPU drop-ins for the numerically heavy functions of pyAudioAnalysis.audioSegmentation.

    self_similarity_matrix(feature_vectors)                      audioSegmentation.py:40-55
    music_thumbnailing(signal, sampling_rate, short_window=1.0,  audioSegmentation.py:1096-1190
                       short_step=0.5, thumb_size=10.0, limit_1=0, limit_2=1)
    silence_removal(signal, sampling_rate, st_win, st_step,      audioSegmentation.py:672-815
                    smooth_window=0.5, weight=0.5, plot=False)

Same names, argument meaning and return values as the reference.  Everything numeric runs in libpaa_hip.so
(standardisation, FP64 matrix-core Gram matrix, diagonal filter, masks, arg-max; short-term features and the per-frame
SVM probability of silence_removal); there is no CPU fallback for those.  `music_thumbnailing` keeps the short-term
features and the similarity matrix in HBM: only the filtered matrix it returns comes back to the host.
`silence_removal` trains its two-class SVM exactly where the reference does -- with scikit-learn (:739,
audioTrainTest.train_svm) -- and replaces the per-frame predict_proba loop (:744-748) by one kernel over all frames
(svm_onset_probability).  The rest of audioSegmentation (HMM segmentation, diarisation, plotting) is out of scope
(control plane / third-party models).
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
    sim = _ffi.result_array((F.shape[1], F.shape[1]))
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


# ---------------------------------------------------------------------------------------------------------
# silence removal (reference :672-815)
# ---------------------------------------------------------------------------------------------------------
def smooth_moving_avg(signal, window=11):
    """Box-filter smoothing of a 1-D sequence whose ends are continued by point reflection about the first / last
    sample (what the reference's helper computes, audioSegmentation.py:25-37): out[i] is the mean of `width`
    consecutive entries of the extended sequence, the block ending (width - 1) // 2 entries after position i.
    width = int(window); sequences are returned untouched for width < 3 (:31-32); ValueError for anything that is
    not one-dimensional or is shorter than the width (:27-30)."""
    width = int(window)
    seq = signal
    if seq.ndim != 1:
        raise ValueError("smooth_moving_avg needs a one-dimensional sequence, got %d dimensions" % seq.ndim)
    n = seq.size
    if n < width:
        raise ValueError("Input vector needs to be bigger than window size.")
    if width < 3:
        return signal
    # point reflections: `width` entries in front (seq[width-1] .. seq[0] mirrored about seq[0]), width - 1 behind
    lead = 2.0 * seq[0] - seq[:width][::-1]
    trail = 2.0 * seq[-1] - seq[n - width + 1:][::-1]
    extended = np.concatenate((lead, seq, trail))
    box = np.full(width, 1.0) / float(width)
    # 'valid' block means m[k] = mean(extended[k : k + width]); out[i] uses the block that ends at extended index
    # width + i + (width - 1) // 2, i.e. k = i + 1 + (width - 1) // 2  (np.convolve(.., 'same') of the reference, cropped)
    first = 1 + (width - 1) // 2
    return np.convolve(extended, box, mode="valid")[first:first + n]


def svm_onset_probability(st_feats, mean, std, svm):
    """svm.predict_proba((st_feats[:, i] - mean) / std)[0][1] for every frame i (reference :744-748) in one kernel.

    svm: a TRAINED binary scikit-learn SVC with probability=True (kernel 'linear' or 'rbf'); only its arrays are read
    (support_vectors_, dual_coef_, intercept_, probA_, probB_, kernel, _gamma)."""
    F = np.ascontiguousarray(np.asarray(st_feats, dtype=np.float64))
    if F.ndim != 2:
        raise ValueError("st_feats must be (n_feats x n_frames)")
    kernel = getattr(svm, "kernel", "linear")
    if kernel not in ("linear", "rbf"):
        raise NotImplementedError("SVC kernel %r (the reference trains linear SVMs here, audioTrainTest.py:152)" % (kernel,))
    if len(getattr(svm, "classes_", (0, 1))) != 2:
        raise ValueError("a two-class SVC is required")
    sv = np.ascontiguousarray(svm.support_vectors_, dtype=np.float64)
    coef = np.ascontiguousarray(np.asarray(svm.dual_coef_, dtype=np.float64).reshape(-1))
    if kernel == "linear":
        # <sum_i coef_i sv_i, x>: fold the support vectors into one weight vector (same value up to rounding order)
        sv = np.ascontiguousarray((coef[:, None] * sv).sum(axis=0, keepdims=True))
        coef = np.ones(1)
    mean = np.ascontiguousarray(np.asarray(mean, dtype=np.float64).reshape(-1))
    std = np.ascontiguousarray(np.asarray(std, dtype=np.float64).reshape(-1))
    if sv.shape[1] != F.shape[0] or mean.shape[0] != F.shape[0] or std.shape[0] != F.shape[0]:
        raise ValueError("feature dimension mismatch between st_feats, the scaler and the SVM")
    prob = np.empty(F.shape[1])
    gamma = float(svm._gamma) if kernel == "rbf" else 0.0
    _ffi.check(_ffi.lib().paa_svm_binary_proba_f64(
        _ffi.as_f64p(F), F.shape[0], F.shape[1], _ffi.as_f64p(mean), _ffi.as_f64p(std), _ffi.as_f64p(sv),
        _ffi.as_f64p(coef), sv.shape[0], float(np.asarray(svm.intercept_).reshape(-1)[0]), gamma,
        float(np.asarray(svm.probA_).reshape(-1)[0]), float(np.asarray(svm.probB_).reshape(-1)[0]), _ffi.as_f64p(prob)))
    return prob


def _train_onset_svm(low_energy, high_energy):
    """Steps of reference :727-739: features_to_matrix, StandardScaler, train_svm(.., 1.0) -- scikit-learn's job."""
    try:
        import sklearn.svm
        from sklearn.preprocessing import StandardScaler
    except ImportError as exc:       # the reference needs it at the same place
        raise ImportError("silence_removal trains its SVM with scikit-learn (audioSegmentation.py:733-739): %s" % exc)
    features = np.vstack([low_energy, high_energy])
    labels = np.append(np.zeros(low_energy.shape[0]), np.ones(high_energy.shape[0]))     # audioTrainTest.features_to_matrix
    scaler = StandardScaler()
    features_norm = scaler.fit_transform(features)
    svm = sklearn.svm.SVC(C=1.0, kernel='linear', probability=True, gamma='auto')         # audioTrainTest.py:152-154
    svm.fit(features_norm, labels)
    return svm, scaler.mean_, scaler.scale_


def _energy_extremes(st_feats):
    """Columns (frames) of the short-term matrix that fall into the quietest / loudest tenth by energy (row 1):
    thresholds are the means of the lowest and of the highest tenth of the sorted energies -- the loudest frame itself
    left out of the upper mean, as in the reference -- plus 1e-15 (audioSegmentation.py:713-728)."""
    energy = st_feats[1]
    ranked = np.sort(energy)
    tenth = int(ranked.size / 10)
    quiet_limit = ranked[:tenth].mean() + 1e-15
    loud_limit = ranked[-tenth:-1].mean() + 1e-15
    return st_feats[:, energy <= quiet_limit], st_feats[:, energy >= loud_limit]


def _onset_threshold(prob, weight):
    """Weighted mix of the mean of the lowest tenth and the mean of the highest tenth of the smoothed probabilities
    (audioSegmentation.py:753-759; the low part is scaled before it is averaged, the high part after)."""
    ranked = np.sort(prob)
    tenth = int(ranked.size / 10)
    return ((1 - weight) * ranked[:tenth]).mean() + weight * ranked[-tenth:].mean()


def _onset_segments(active, st_step, min_duration=0.2):
    """Frame indices above the threshold -> [start, end] limits in seconds (audioSegmentation.py:764-791).

    Indices whose neighbours are at most 2 frames apart belong to one segment; the list is cut where np.diff exceeds 2.
    A segment is kept when it lasts longer than min_duration seconds (:786-790) -- which also removes every one-frame
    segment, so the reference's habit of never opening a segment at the very last index (:770-771) changes nothing."""
    active = np.asarray(active)
    if active.size == 0:
        return []
    cuts = np.flatnonzero(np.diff(active) > 2) + 1
    firsts = active[np.concatenate(([0], cuts))]
    lasts = active[np.concatenate((cuts - 1, [active.size - 1]))]
    limits = [[lo * st_step, hi * st_step] for lo, hi in zip(firsts, lasts)]
    return [seg for seg in limits if seg[1] - seg[0] > min_duration]


def _plot_segments(signal, sampling_rate, prob, st_step, seg_limits):
    # waveform and probability curve with the segment limits marked (:793-811)
    import matplotlib.pyplot as plt
    curves = ((np.arange(signal.shape[0]) / float(sampling_rate), signal, 'Signal'),
              (np.arange(prob.shape[0]) * st_step, prob, 'svm Probability'))
    for row, (xs, ys, title) in enumerate(curves):
        plt.subplot(2, 1, row + 1)
        plt.plot(xs, ys)
        for lo, hi in seg_limits:
            plt.axvline(x=lo, color='red')
            plt.axvline(x=hi, color='red')
        plt.title(title)
    plt.show()


def silence_removal(signal, sampling_rate, st_win, st_step, smooth_window=0.5, weight=0.5, plot=False):
    """Event detection (silence removal), reference :672-815.  Returns the list of [start, end] segments in seconds.

    signal, sampling_rate: the audio; st_win, st_step: short-term window and step in SECONDS; smooth_window: length of
    the probability smoothing in seconds; weight in (0, 1): the higher, the stricter (values outside are pulled to
    0.01 / 0.99, :697-700)."""
    from . import ShortTermFeatures as stf
    weight = min(max(weight, 0.01), 0.99) if not 0 < weight < 1 else weight
    mono = audioBasicIO.stereo_to_mono(signal)
    st_feats, _ = stf.feature_extraction(mono, sampling_rate, st_win * sampling_rate, st_step * sampling_rate)   # :707-710
    quiet, loud = _energy_extremes(st_feats)
    svm, mean, std = _train_onset_svm(quiet.T, loud.T)                                     # :730-739, scikit-learn
    # onset probability of every frame in one kernel (replaces the predict_proba loop :741-748), then smoothing (:751)
    prob = smooth_moving_avg(svm_onset_probability(st_feats, mean, std, svm), smooth_window / st_step)
    active = np.flatnonzero(prob > _onset_threshold(prob, weight))                         # :761
    seg_limits = _onset_segments(active, st_step)
    if plot:
        _plot_segments(mono, sampling_rate, prob, st_step, seg_limits)
    return seg_limits
