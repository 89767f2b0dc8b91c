"""CPU oracle: a NumPy restatement of the pyAudioAnalysis short-/mid-term path.

TEST INFRASTRUCTURE ONLY -- not a product path and never a fallback.
Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may
import this module (as the checker / the reported CPU baseline).  The product
package `pyaudioanalysis_amd` never imports it and fails loudly when the HIP
library is missing.

Parity status: PINNED.  The reference's own tests hold shapes only
(pytests/test_feature_extraction.py:10-29), so value parity is pinned against
outputs of the unmodified reference executed in the build container:
tests/golden/*.npz, produced by oracle/make_golden.py (committed) and checked
by tests/test_oracle_vs_golden.py.

What is restated (reference file:line, relative to /root/reference/pyAudioAnalysis):
  clip normalisation .............. ShortTermFeatures.py:14-19, 567-570
  framing / |FFT|/Nf .............. ShortTermFeatures.py:608-621
  zcr / energy / energy entropy ... ShortTermFeatures.py:22-51
  centroid+spread / spec. entropy . ShortTermFeatures.py:57-107
  flux / roll-off ................. ShortTermFeatures.py:110-140, 624-625, 682
  mel bank / MFCC ................. ShortTermFeatures.py:191-254
  chroma tables / chroma / std .... ShortTermFeatures.py:257-321, 667
  deltas, names, output layout .... ShortTermFeatures.py:590-604, 668-685
  spectrogram / chromagram ........ ShortTermFeatures.py:324-452
  mid-term mean/std ............... MidTermFeatures.py:87-127
  stereo -> mono .................. audioBasicIO.py:156-168
  self-similarity / thumbnailing .. audioSegmentation.py:40-55, 1096-1190  (StandardScaler from
                                    scikit-learn and pdist('cosine') / convolve2d from SciPy are third-party
                                    dependencies that are not in /root/reference: their published algorithms
                                    are restated below and pinned by tests/golden/thumb_*.npz / sim_*.npz)
The FFT itself is SciPy's pocketfft (scipy>=1.6.3, requirements.txt:3) in the
reference; it is not in /root/reference, so this oracle calls scipy.fft.fft (the
same pocketfft entry, bit-identical to scipy.fftpack.fft on real input -- checked
when the goldens were generated; numpy.fft differs by ~1e-14) at the reference's
call sites.

Structure differs from the reference on purpose (frame-invariant tables are
built once, the chroma scatter is expressed as a gather list); arithmetic per
frame follows the reference operation by operation.
"""
import sys

import numpy as np
import scipy.fft

EPS = sys.float_info.epsilon          # ShortTermFeatures.py:11
N_BASE = 34                           # 8 time/spectral + 13 mfcc + 13 chroma   (:580-585)
N_MFCC = 13
N_MEL = 40                            # 13 linear + 27 log filters               (:191-192,204)
ROLLOFF_C = 0.90                      # (:653)
N_ENTROPY_BLOCKS = 10                 # (:34,85)


# --------------------------------------------------------------------------
# names
# --------------------------------------------------------------------------
def feature_names(deltas=True):
    """Exact strings of ShortTermFeatures.py:590-604."""
    base = ["zcr", "energy", "energy_entropy", "spectral_centroid",
            "spectral_spread", "spectral_entropy", "spectral_flux",
            "spectral_rolloff"]
    base += ["mfcc_%d" % i for i in range(1, N_MFCC + 1)]
    base += ["chroma_%d" % i for i in range(1, 13)]
    base += ["chroma_std"]
    if deltas:
        base = base + ["delta " + s for s in base]
    return base


def mid_feature_names():
    """MidTermFeatures.py:113-114 (always over the 68 delta-augmented rows, :93-95)."""
    st = feature_names(True)
    return [s + "_mean" for s in st] + [s + "_std" for s in st]


# --------------------------------------------------------------------------
# frame-invariant tables
# --------------------------------------------------------------------------
def mel_bank(fs, nfft):
    """Dense (40, nfft) triangular bank, ShortTermFeatures.py:204-231.

    Quirk kept: the bin-frequency axis is k*fs/nfft although the nfft bins of
    the magnitude spectrum span 0..fs/2 (:215).
    """
    n_lin, n_log = 13, 27
    edges = np.zeros(N_MEL + 2)
    edges[:n_lin] = 133.33 + np.arange(n_lin) * (200 / 3)
    edges[n_lin:] = edges[n_lin - 1] * 1.0711703 ** np.arange(1, n_log + 3)
    peak = 2.0 / (edges[2:] - edges[:-2])
    axis = np.arange(nfft) / (1.0 * nfft) * fs
    bank = np.zeros((N_MEL, nfft))
    for m in range(N_MEL):
        lo, mid, hi = edges[m], edges[m + 1], edges[m + 2]
        k_lo = int(np.floor(lo * nfft / fs)) + 1
        k_mid = int(np.floor(mid * nfft / fs)) + 1
        k_hi = int(np.floor(hi * nfft / fs)) + 1
        up = np.arange(k_lo, k_mid, dtype=int)
        dn = np.arange(k_mid, k_hi, dtype=int)
        # out-of-range bins raise IndexError in the reference (:230-231) too
        bank[m][up] = (peak[m] / (mid - lo)) * (axis[up] - lo)
        bank[m][dn] = (peak[m] / (hi - mid)) * (hi - axis[dn])
    return bank


def dct_matrix():
    """Rows 0..12 of the orthonormal DCT-II of length 40 (scipy dct(type=2, norm='ortho'), :253)."""
    n = np.arange(N_MEL)
    k = np.arange(N_MFCC)[:, None]
    mat = np.sqrt(2.0 / N_MEL) * np.cos(np.pi * k * (2 * n + 1) / (2.0 * N_MEL))
    mat[0] *= 1.0 / np.sqrt(2.0)
    return mat


def chroma_bins(fs, nfft):
    """(slot_of_bin, count_of_bin) exactly as chroma_features_init, :257-274."""
    freqs = np.array([((f + 1) * fs) / (2 * nfft) for f in range(nfft)])
    slot = np.round(12.0 * np.log2(freqs / 27.50)).astype(int)
    count = np.zeros(nfft)
    for u in np.unique(slot):
        idx = np.nonzero(slot == u)
        count[idx] = idx[0].shape
    return slot, count


def chroma_gather(fs, nfft):
    """Gather form of the scatter at ShortTermFeatures.py:286-302.

    Returns (src_bin, weight, pitch_class) arrays, one entry per written slot,
    in ascending slot order.  `C[slot] = spec` keeps the LAST (highest) bin
    writing a slot, negative slots wrap (numpy indexing), and the division
    `C /= count[slot]` is indexed by slot POSITION (:289).  Raises the
    reference's exception types when max(slot) >= nfft (:290-294).
    """
    slot, count = chroma_bins(fs, nfft)
    if slot.max() >= nfft:
        first_over = np.nonzero(slot > nfft)[0]
        if first_over.size == 0:
            raise IndexError("index 0 is out of bounds for axis 0 with size 0")
        raise ValueError("shape mismatch: value array of shape (%d,) could not be "
                         "broadcast to indexing result" % nfft)
    owner = {}
    for k in range(nfft):                      # later bins overwrite earlier ones
        owner[int(slot[k]) % nfft] = k
    pos = np.array(sorted(owner), dtype=np.int64)
    src = np.array([owner[p] for p in pos], dtype=np.int64)
    weight = 1.0 / count[slot[pos]]            # divisor indexed by slot position
    # NB: count[slot[pos]] uses numpy indexing, so slot[pos] may itself be negative/wrap
    return src, weight, (pos % 12).astype(np.int64), pos


class Tables:
    """All frame-invariant data for one (fs, window)."""

    def __init__(self, fs, window):
        self.fs = fs
        self.window = int(window)
        self.nfft = int(self.window / 2)                      # :575
        self.mel = mel_bank(fs, self.nfft)
        self.dct = dct_matrix()
        self.ch_src, self.ch_w, self.ch_class, self.ch_pos = chroma_gather(fs, self.nfft)
        self.freq_ramp = np.arange(1, self.nfft + 1) * (fs / (2.0 * self.nfft))   # :59-60


# --------------------------------------------------------------------------
# per-clip / per-frame arithmetic
# --------------------------------------------------------------------------
def stereo_to_mono(signal):
    """audioBasicIO.py:156-168."""
    signal = np.asarray(signal)
    if signal.ndim == 2:
        if signal.shape[1] == 1:
            return signal.flatten()
        if signal.shape[1] == 2:
            return (signal[:, 1] / 2) + (signal[:, 0] / 2)
    return signal


def normalize_clip(signal):
    """x/2^15, remove clip mean, divide by clip max|.| + 1e-10 (:567-570, 14-19)."""
    x = np.double(signal) / (2.0 ** 15)
    x = x - x.mean()
    return x / (np.abs(x).max() + 1e-10)


def magnitude_spectrum(frame, nfft):
    """|FFT(frame)|[0:nfft] / nfft, rectangular window (:617-621)."""
    return np.abs(scipy.fft.fft(frame))[0:nfft] / nfft


def _block_entropy(v2, total):
    """-sum s log2(s+eps) over 10 leading blocks of floor(len/10) (:34-51, 85-107)."""
    blk = int(np.floor(len(v2) / N_ENTROPY_BLOCKS))
    sub = v2[:blk * N_ENTROPY_BLOCKS].reshape(N_ENTROPY_BLOCKS, blk).sum(axis=1)
    s = sub / (total + EPS)
    return -np.sum(s * np.log2(s + EPS))


def frame_vector(x, X, X_prev, tab):
    """The 34 base features of one frame (ShortTermFeatures.py:626-667)."""
    fs, nfft = tab.fs, tab.nfft
    out = np.zeros(N_BASE)
    w = len(x)
    # time domain
    out[0] = (np.sum(np.abs(np.diff(np.sign(x)))) / 2) / np.float64(w - 1.0)
    x2 = x ** 2
    e_tot = np.sum(x2)
    out[1] = e_tot / np.float64(w)
    out[2] = _block_entropy(x2, e_tot)
    # centroid / spread
    peak = X.max()
    Xn = X / EPS if peak == 0 else X / peak
    den = np.sum(Xn) + EPS
    cen = np.sum(tab.freq_ramp * Xn) / den
    spr = np.sqrt(np.sum(((tab.freq_ramp - cen) ** 2) * Xn) / den)
    out[3] = cen / (fs / 2.0)
    out[4] = spr / (fs / 2.0)
    # spectral entropy
    P = X ** 2
    p_tot = np.sum(P)
    out[5] = _block_entropy(P, p_tot)
    # flux
    out[6] = np.sum((X / np.sum(X + EPS) - X_prev / np.sum(X_prev + EPS)) ** 2)
    # roll-off
    above = np.nonzero(np.cumsum(P) + EPS > ROLLOFF_C * p_tot)[0]
    out[7] = np.float64(above[0]) / float(nfft) if len(above) > 0 else 0.0
    # mfcc
    out[8:21] = tab.dct @ np.log10(np.dot(X, tab.mel.T) + EPS)
    # chroma: fold gathered bins into 12 pitch classes, in slot order
    # (np.sum over the (rows,12) reshape adds rows in ascending order, :299-302)
    vals = P[tab.ch_src] * tab.ch_w
    chroma = np.zeros(12)
    rows = int(np.ceil(nfft / 12.0))
    grid = np.zeros((rows * 12,))
    grid[tab.ch_pos] = vals
    chroma = grid.reshape(rows, 12).sum(axis=0)
    chroma = chroma / EPS if p_tot == 0 else chroma / p_tot
    out[21:33] = chroma
    out[33] = chroma.std()
    return out


def feature_extraction(signal, sampling_rate, window, step, deltas=True):
    """Oracle for ShortTermFeatures.feature_extraction (:543-685)."""
    window, step = int(window), int(step)
    x_all = normalize_clip(signal)
    n = len(x_all)
    tab = Tables(sampling_rate, window)
    cols = []
    pos = 0
    prev_X = None
    prev_v = None
    while pos + window - 1 < n:
        x = x_all[pos:pos + window]
        pos += step
        X = magnitude_spectrum(x, tab.nfft)
        if prev_X is None:
            prev_X = X.copy()
        v = frame_vector(x, X, prev_X, tab)
        if deltas:
            d = v - prev_v if prev_v is not None else np.zeros(N_BASE)
            cols.append(np.concatenate((v, d)))
            prev_v = v
        else:
            cols.append(v)
        prev_X = X
    if not cols:
        raise ValueError("need at least one array to concatenate")
    return np.ascontiguousarray(np.stack(cols, axis=1)), feature_names(deltas)


def feature_extraction_reference_cost(signal, sampling_rate, window, step, deltas=True):
    """The same numbers as feature_extraction() above, produced with the reference's COST STRUCTURE -- for bench.py's
    cpu_baseline only.  The Python reference cannot travel to the GPU box (no /root/reference there), and the ports above
    are 8-18x faster than it because they hoist what it recomputes; this variant puts those per-frame costs back:
      * the chroma tables are rebuilt for EVERY frame (chroma_features calls chroma_features_init, ShortTermFeatures.py:281
        -> :257-274: a Python list comprehension over the bins, np.unique, one np.nonzero per distinct slot) and the chroma
        vector goes through the dense scatter / divide / pad / reshape / column-sum of :286-302;
      * the MFCCs go through scipy.fftpack.dct(type=2, norm='ortho') per frame (:253);
      * per-frame results are (F, 1) columns collected in a Python list and joined by one np.concatenate (:684).
    The mel bank is built once per call (:578), as there.  Calibrated against the unmodified reference in the build
    container (profiles/r06_reference_cost_port.json, scripts/reference_cpu_baseline.py)."""
    from scipy.fftpack import dct as fftpack_dct, fft as fftpack_fft
    window, step = int(window), int(step)
    x_all = normalize_clip(signal)
    n = len(x_all)
    fs = sampling_rate
    nfft = int(window / 2)
    tab = Tables(fs, window)
    cols = []
    pos = 0
    prev_X = None
    prev_v = None
    while pos + window - 1 < n:
        x = x_all[pos:pos + window]
        pos += step
        X = abs(fftpack_fft(x))[0:nfft] / nfft
        if prev_X is None:
            prev_X = X.copy()
        v = frame_vector(x, X, prev_X, tab)
        # ---- the reference's per-frame chroma: tables rebuilt, dense scatter (identical values, its cost)
        slot, count = chroma_bins(fs, nfft)
        P = X ** 2
        dense = np.zeros((nfft,))
        dense[slot] = P
        dense /= count[slot]
        rows = int(np.ceil(nfft / 12.0))
        padded = np.zeros((rows * 12,))
        padded[:nfft] = dense
        folded = np.sum(padded.reshape(rows, 12), axis=0)
        p_tot = P.sum()
        folded = folded / EPS if p_tot == 0 else folded / p_tot
        v[21:33] = folded
        v[33] = folded.std()
        # ---- its MFCC call: scipy's DCT on the log mel spectrum
        v[8:21] = fftpack_dct(np.log10(np.dot(X, tab.mel.T) + EPS), type=2, norm="ortho", axis=-1)[:N_MFCC]
        col = np.zeros((2 * N_BASE if deltas else N_BASE, 1))
        col[:N_BASE, 0] = v
        if deltas and prev_v is not None:
            col[N_BASE:, 0] = v - prev_v
        prev_v = v
        cols.append(col)
        prev_X = X.copy()
    if not cols:
        raise ValueError("need at least one array to concatenate")
    return np.concatenate(cols, 1), feature_names(deltas)


def mid_ratios(mid_window, mid_step, short_window, short_step):
    """(ratio, step_ratio) exactly as MidTermFeatures.py:100-102 (Python round)."""
    ratio = round((mid_window - (short_window - short_step)) / short_step)
    step_ratio = int(round(mid_step / short_step))
    return ratio, step_ratio


def mid_statistics(short_features, ratio, step_ratio):
    """Mean / population std over sliding mid windows (MidTermFeatures.py:110-126)."""
    nrows, T = short_features.shape
    starts = list(range(0, T, step_ratio))
    mid = np.zeros((2 * nrows, len(starts)))
    for i in range(nrows):
        row = short_features[i]
        for m, c in enumerate(starts):
            seg = row[c:min(c + ratio, T)]
            mid[i, m] = np.mean(seg)
            mid[i + nrows, m] = np.std(seg)
    return np.nan_to_num(mid)


def mid_feature_extraction(signal, sampling_rate, mid_window, mid_step,
                           short_window, short_step):
    """Oracle for MidTermFeatures.mid_feature_extraction (:87-127)."""
    st, _ = feature_extraction(signal, sampling_rate, short_window, short_step)
    ratio, step_ratio = mid_ratios(mid_window, mid_step, short_window, short_step)
    if step_ratio < 1:
        raise ValueError("mid_step shorter than half a short step: the reference loops forever")
    return mid_statistics(st, ratio, step_ratio), st, mid_feature_names()


def spectrogram(signal, sampling_rate, window, step):
    """Oracle for ShortTermFeatures.spectrogram (:389-452): frame i starts at window + i*step."""
    window, step = int(window), int(step)
    x_all = normalize_clip(signal)
    n = len(x_all)
    nfft = int(window / 2)
    out = np.zeros((int((n - window) / step) + 1, nfft))
    for i, p in enumerate(range(window, n - window + 1, step)):
        out[i, :] = magnitude_spectrum(x_all[p:p + window], nfft)
    freq_axis = [float((f + 1) * sampling_rate) / (2 * nfft) for f in range(nfft)]
    time_axis = [float(t * step) / sampling_rate for t in range(out.shape[0])]
    return out, time_axis, freq_axis


CHROMA_NAMES = ['A', 'A#', 'B', 'C', 'C#', 'D', 'D#', 'E', 'F', 'F#', 'G', 'G#']   # :283-284


def chromagram(signal, sampling_rate, window, step):
    """Oracle for ShortTermFeatures.chromagram (:324-386); the last frame may be truncated."""
    window, step = int(window), int(step)
    x_all = normalize_clip(signal)
    n = len(x_all)
    nfft = int(window / 2)
    tab = Tables(sampling_rate, window)
    out = np.zeros((int((n - step - window) / step) + 1, 12))
    for i, p in enumerate(range(window, n - step, step)):
        x = x_all[p:p + window]
        X = np.abs(scipy.fft.fft(x))[0:nfft]
        X = X / len(X)
        P = X ** 2
        if len(P) < nfft:
            # a truncated last frame shorter than num_fft: the reference's scatter `C[num_chroma] = spec` (:288-293) fails with a
            # shape mismatch -- ValueError, not the IndexError this gather form would raise (checked against the live reference)
            raise ValueError("shape mismatch: value array of shape (%d,) could not be broadcast to indexing result of shape (%d,)"
                             % (len(P), nfft))
        rows = int(np.ceil(nfft / 12.0))
        grid = np.zeros((rows * 12,))
        grid[tab.ch_pos] = P[tab.ch_src] * tab.ch_w
        c = grid.reshape(rows, 12).sum(axis=0)
        tot = P.sum()
        out[i, :] = c / EPS if tot == 0 else c / tot
    time_axis = [(t * step) / sampling_rate for t in range(out.shape[0])]
    return out, time_axis, list(CHROMA_NAMES)


# --------------------------------------------------------------------------
# comparison policy shared by every parity test
# --------------------------------------------------------------------------
MFCC_ROWS = tuple(range(8, 21))


# --------------------------------------------------------------------------
# self-similarity matrix and music thumbnailing
# --------------------------------------------------------------------------
def standardize_rows(feature_vectors):
    """StandardScaler().fit_transform(F.T).T (audioSegmentation.py:51-52).

    scikit-learn (>=0.24 rule, installed 1.7.2): mean = sum/n; variance by the corrected two-pass formula
    [sum (x-m)^2 - (sum (x-m))^2/n] / n; a feature with var <= n*eps*var + (n*mean*eps)^2 counts as constant
    and gets scale 1 instead of sqrt(var)."""
    X = np.asarray(feature_vectors, dtype=np.float64)
    n = X.shape[1]
    mean = X.sum(axis=1) / n
    d = X - mean[:, None]
    corr = d.sum(axis=1)
    var = ((d * d).sum(axis=1) - corr * corr / n) / n
    constant = var <= n * EPS * var + (n * mean * EPS) ** 2
    scale = np.sqrt(var)
    scale[constant] = 1.0
    return d / scale[:, None]


def self_similarity_matrix(feature_vectors):
    """audioSegmentation.py:40-55: 1 - squareform(pdist(Z.T, 'cosine')) on the standardised vectors.

    SciPy's cosine distance: 1 - clip(u.v / (|u| |v|), -1, 1) with |u| = sqrt(sum u^2); squareform puts exact
    zeros on the diagonal, so the similarity diagonal is exactly 1 (also for zero vectors, whose off-diagonal
    entries are NaN = 0/0 -- kept)."""
    Z = standardize_rows(feature_vectors)
    norms = np.sqrt((Z * Z).sum(axis=0))
    with np.errstate(invalid="ignore", divide="ignore"):
        cos = (Z.T @ Z) / (norms[:, None] * norms[None, :])
    big = np.abs(cos) > 1.0
    cos[big] = np.copysign(1.0, cos[big])
    sim = 1.0 - (1.0 - cos)
    np.fill_diagonal(sim, 1.0)
    return sim


def thumbnail_filter(sim_matrix, m_filter, short_step, limit_1=0, limit_2=1):
    """audioSegmentation.py:1146-1163: diagonal moving sum (convolve2d with eye(M), 'valid'), masking of the
    near-diagonal band and the lower triangle with the global minimum, limit masks.  Returns the masked matrix."""
    T = sim_matrix.shape[0]
    R = T - m_filter + 1
    if R < 1:
        raise ValueError("fewer feature vectors (%d) than the thumbnail filter length (%d)" % (T, m_filter))
    out = np.zeros((R, R))
    for k in range(m_filter):
        out += sim_matrix[k:k + R, k:k + R]
    min_sm = np.min(out)
    i = np.arange(R)
    band = (np.abs(i[:, None] - i[None, :]) < 5.0 / short_step) | (i[:, None] > i[None, :])
    out[band] = min_sm
    out[0:int(limit_1 * R), :] = min_sm
    out[:, 0:int(limit_1 * R)] = min_sm
    out[int(limit_2 * R):, :] = min_sm
    out[:, int(limit_2 * R):] = min_sm
    return out


def thumbnail_grow(filtered, m_filter):
    """audioSegmentation.py:1165-1182: start at the arg-max and grow along the diagonal to m_filter cells."""
    rows, cols = np.unravel_index(filtered.argmax(), filtered.shape)
    i1 = i2 = int(rows)
    j1 = j2 = int(cols)
    while i2 - i1 < m_filter:
        if i1 <= 0 or j1 <= 0 or i2 >= filtered.shape[0] - 2 or j2 >= filtered.shape[1] - 2:
            break
        if filtered[i1 - 1, j1 - 1] > filtered[i2 + 1, j2 + 1]:
            i1 -= 1
            j1 -= 1
        else:
            i2 += 1
            j2 += 1
    return i1, i2, j1, j2


def music_thumbnailing(signal, sampling_rate, short_window=1.0, short_step=0.5, thumb_size=10.0,
                       limit_1=0, limit_2=1):
    """audioSegmentation.py:1096-1190.  Returns (A1, A2, B1, B2, filtered similarity matrix)."""
    signal = stereo_to_mono(signal)
    st, _ = feature_extraction(signal, sampling_rate, sampling_rate * short_window, sampling_rate * short_step)
    sim = self_similarity_matrix(st)
    m_filter = int(round(thumb_size / short_step))
    filt = thumbnail_filter(sim, m_filter, short_step, limit_1, limit_2)
    i1, i2, j1, j2 = thumbnail_grow(filt, m_filter)
    return short_step * i1, short_step * i2, short_step * j1, short_step * j2, filt


# ---- beat extraction (MidTermFeatures.py:18-84 + utilities.peakdet, utilities.py:33-102): checker of the GPU beat_kernel ----
def _peak_positions(v, delta):
    """Positions of the maxima of Billauer's peakdet (utilities.py:33-102): a maximum is recorded once the signal has dropped
    by more than delta below the running maximum, a minimum once it rose by more than delta above the running minimum."""
    peaks = []
    lo, hi = np.inf, -np.inf
    hi_pos = 0
    seek_max = True
    for k in range(len(v)):
        cur = v[k]
        if cur > hi:
            hi, hi_pos = cur, k
        if cur < lo:
            lo = cur
        if seek_max:
            if cur < hi - delta:
                peaks.append(hi_pos)
                lo = cur
                seek_max = False
        elif cur > lo + delta:
            hi, hi_pos = cur, k
            seek_max = True
    return peaks


def beat_extraction(short_features, window_size):
    """(bpm, confidence) of a short-term matrix, restating MidTermFeatures.py:18-84 (pinned on tests/golden/directory_small.npz)."""
    rows = [0, 1, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18]      # :30-31
    max_beat_time = int(round(2.0 / window_size))
    hist_all = np.zeros((max_beat_time,))
    n_frames = short_features.shape[1]
    edges = np.arange(0.5, max_beat_time + 1.5)
    for r in rows:
        v = np.asarray(short_features[r, :])
        with np.errstate(invalid="ignore"):
            thr = 2.0 * (np.abs(v[0:-1] - v[1::])).mean() if n_frames > 1 else np.nan   # :37-38 (mean of nothing: NaN)
        if thr <= 0:
            thr = 0.0000000000000001
        pos = _peak_positions(v, thr)
        gaps = [pos[j + 1] - pos[j] for j in range(len(pos) - 1)]
        counts, _ = np.histogram(gaps, edges)
        hist_all += counts.astype(float) / n_frames
    centers = (edges[0:-1] + edges[1::]) / 2.0
    best = np.argmax(hist_all)
    bpm = (60 / (centers * window_size))[best]
    ratio = hist_all[best] / (hist_all.sum() + 0.00000001)                       # MidTermFeatures.py:13 eps
    return bpm, ratio


def ill_conditioned_mfcc_frames(signal, sampling_rate, window, step, factor=1e4):
    """Frames whose MFCCs the reference itself computes from FFT round-off.

    mfcc = DCT(log10(E_i + eps)), E_i = mel energy (:252).  When a mel band is (numerically) EMPTY --
    E_i below factor*eps, e.g. a pure tone sitting exactly on an FFT bin, where every other bin holds
    only round-off of order 1e-17 -- d log10(E_i+eps)/dE_i = 1/(eps ln10) = 2e15 and the result is a
    function of the FFT implementation's rounding, not of the signal.  Exact digital silence (E_i == 0
    on both sides) is NOT flagged.  Returns a bool mask over frames; a frame following a flagged frame
    is flagged too (its delta row subtracts the flagged value).
    """
    window, step = int(window), int(step)
    x = normalize_clip(signal)
    tab = Tables(sampling_rate, window)
    T = (len(x) - window) // step + 1
    mask = np.zeros(max(T, 0), dtype=bool)
    for t in range(T):
        X = magnitude_spectrum(x[t * step:t * step + window], tab.nfft)
        E = np.dot(X, tab.mel.T)
        mask[t] = bool(np.any((E > 0) & (E < factor * EPS)))
    out = mask.copy()
    out[1:] |= mask[:-1]
    return out


def mixed_tolerance_violations(got, ref, rel=1e-4, row_abs=1e-6, abs_floor=1e-9):
    """Count entries with |d| > rel*|ref| + row_abs*scale(row) + abs_floor (rows = axis 0).

    rel is the north_star tolerance (1e-4 relative).  row_abs covers entries that cross zero or
    sit ~1e-35 on silent frames, where element-wise relative error is undefined;
    scale(row) = max|ref_row|, except that the 13 MFCC rows (and their deltas / mid-term
    statistics) share ONE scale, max|ref| over the MFCC group: they are one orthonormal DCT of one
    log-mel vector, so an absolute perturbation of the log-mel energies (FFT round-off against the
    eps = 2.2e-16 floor on bins the signal leaves exactly empty) lands on every coefficient alike.
    abs_floor covers rows the reference emits as exact zeros (e.g. MFCC 2..13 of an all-zero clip,
    where scipy's DCT cancels exactly and any other summation order leaves ~1e-14).
    """
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    if got.shape != ref.shape:
        raise AssertionError("shape %s != %s" % (got.shape, ref.shape))
    if ref.ndim == 2:
        scale = np.max(np.abs(ref), axis=1, keepdims=True)
        nrows = ref.shape[0]
        if nrows % N_BASE == 0 and nrows // N_BASE in (1, 2, 4):
            for blk in range(nrows // N_BASE):
                rows = [blk * N_BASE + r for r in MFCC_ROWS]
                scale[rows] = scale[rows].max()
    else:
        scale = np.max(np.abs(ref))
    bad = np.abs(got - ref) > rel * np.abs(ref) + row_abs * scale + abs_floor
    bad |= ~np.isfinite(got)
    return int(bad.sum()), bad
