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
    """Moving average with reflected ends (reference :25-37)."""
    window = int(window)
    if signal.ndim != 1:
        raise ValueError("")
    if signal.size < window:
        raise ValueError("Input vector needs to be bigger than window size.")
    if window < 3:
        return signal
    s = np.r_[2 * signal[0] - signal[window - 1::-1], signal, 2 * signal[-1] - signal[-1:-window:-1]]
    w = np.ones(window, 'd')
    y = np.convolve(w / w.sum(), s, mode='same')
    return y[window:-window + 1]


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


def silence_removal(signal, sampling_rate, st_win, st_step, smooth_window=0.5, weight=0.5, plot=False):
    """Event detection (silence removal), reference :672-815.  Returns the list of [start, end] segments in seconds."""
    from . import ShortTermFeatures as stf
    if weight >= 1:
        weight = 0.99
    if weight <= 0:
        weight = 0.01
    # Step 1: feature extraction (:707-710)
    signal = audioBasicIO.stereo_to_mono(signal)
    st_feats, _ = stf.feature_extraction(signal, sampling_rate, st_win * sampling_rate, st_step * sampling_rate)
    # Step 2: binary SVM of low vs high energy frames (:712-739)
    st_energy = st_feats[1, :]
    en = np.sort(st_energy)
    st_windows_fraction = int(len(en) / 10)
    low_threshold = np.mean(en[0:st_windows_fraction]) + 1e-15
    high_threshold = np.mean(en[-st_windows_fraction:-1]) + 1e-15
    low_energy = st_feats[:, np.where(st_energy <= low_threshold)[0]]
    high_energy = st_feats[:, np.where(st_energy >= high_threshold)[0]]
    svm, mean, std = _train_onset_svm(low_energy.T, high_energy.T)
    # Step 3: onset probability of every frame (:741-751) -- one kernel instead of a predict_proba call per frame
    prob_on_set = svm_onset_probability(st_feats, mean, std, svm)
    prob_on_set = smooth_moving_avg(prob_on_set, smooth_window / st_step)
    # Step 4A: threshold as a weighted average of the top and bottom 10 % (:753-762)
    prog_on_set_sort = np.sort(prob_on_set)
    nt = int(prog_on_set_sort.shape[0] / 10)
    threshold = (np.mean((1 - weight) * prog_on_set_sort[0:nt]) + weight * np.mean(prog_on_set_sort[-nt::]))
    max_indices = np.where(prob_on_set > threshold)[0]
    # Step 4B: group frame indices to onset segments (:764-783)
    index = 0
    seg_limits = []
    while index < len(max_indices):
        cur_cluster = [max_indices[index]]
        if index == len(max_indices) - 1:
            break
        while max_indices[index + 1] - cur_cluster[-1] <= 2:
            cur_cluster.append(max_indices[index + 1])
            index += 1
            if index == len(max_indices) - 1:
                break
        index += 1
        seg_limits.append([cur_cluster[0] * st_step, cur_cluster[-1] * st_step])
    # Step 5: drop very small segments (:785-791)
    min_duration = 0.2
    seg_limits = [s_lim for s_lim in seg_limits if s_lim[1] - s_lim[0] > min_duration]
    if plot:       # waveform and probability curve with the segment limits marked (:793-811)
        import matplotlib.pyplot as plt
        curves = ((np.arange(signal.shape[0]) / float(sampling_rate), signal, 'Signal'),
                  (np.arange(prob_on_set.shape[0]) * st_step, prob_on_set, 'svm Probability'))
        for row, (xs, ys, title) in enumerate(curves):
            plt.subplot(2, 1, row + 1)
            plt.plot(xs, ys)
            for lo, hi in seg_limits:
                plt.axvline(x=lo, color='red')
                plt.axvline(x=hi, color='red')
            plt.title(title)
        plt.show()
    return seg_limits
