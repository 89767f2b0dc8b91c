"""Checker helpers shared by the -m gpu tests and bench.py's --check leg (test infrastructure, like everything under
oracle/: the product package never imports this).

reference_matrix: the full reference matrix of a clip -- the plain-C oracle (oracle/paa_oracle.c, ~45 k frames/s) for every
    frame, the NumPy oracle (same pocketfft as the reference) for frames of DIGITAL SILENCE.
ill_mask: frames whose MFCCs the reference itself computes from FFT round-off (paa_oracle.ill_conditioned_mfcc_frames,
    vectorised).
contract_violations: the north-star gate |d| <= 1e-4 |ref| + 1e-6 scale(row) + 1e-9, with the documented 1e-5 row term for
    the MFCC rows of ill-conditioned frames.
ill_info / IllInfo: the exception is BOUNDED -- flagged frames are either digitally silent (their MFCCs are then checked at
    1e-9 against the analytic value, silent_mfcc) or count against a per-test budget (default: 1e-3 of the frames).
"""
import numpy as np

import c_oracle
import paa_oracle as O

REL, ROW, FLOOR = 1e-4, 1e-6, 1e-9                 # the contract (tests/test_parity_gpu.py uses these names)
MFCC_ALL = [r + b for b in (0, 34) for r in O.MFCC_ROWS]


def ill_mask(signal, fs, window, step, factor=1e4):
    """paa_oracle.ill_conditioned_mfcc_frames (a numerically empty mel band: the reference's own MFCCs are a function of
    its FFT's round-off), vectorised over frames.  Frames of digital silence are re-examined one by one through the
    oracle's own single-frame FFT call: whether pocketfft returns exact zeros for a CONSTANT frame or leaves 1e-17 in the
    non-DC bins depends on the constant (and its batched transform need not round like its single one); when it leaves
    something, mfcc_2.. of the reference read 5e-8 instead of 0 and the frame is flagged like any other empty band."""
    window, step = int(window), int(step)          # (the reference's callers pass floats: 0.050 * fs; :563-564 truncates)
    x = O.normalize_clip(signal)
    tab = O.Tables(fs, window)
    frames = np.lib.stride_tricks.sliding_window_view(x, window)[::step]
    mask = np.zeros(len(frames), dtype=bool)
    for a in range(0, len(frames), 4096):
        X = np.abs(np.fft.fft(frames[a:a + 4096], axis=1))[:, :tab.nfft] / tab.nfft
        E = X @ tab.mel.T
        mask[a:a + 4096] = np.any((E > 0) & (E < factor * O.EPS), axis=1)
    raw = np.lib.stride_tricks.sliding_window_view(np.asarray(signal, dtype=np.float64), window)[::step]
    for t in np.flatnonzero(raw.max(axis=1) == raw.min(axis=1)):
        E = np.dot(O.magnitude_spectrum(x[t * step:t * step + window], tab.nfft), tab.mel.T)
        mask[t] = bool(np.any((E > 0) & (E < factor * O.EPS)))
    out = mask.copy()
    out[1:] |= mask[:-1]
    return out


class IllInfo:
    """What the checker knows about the frames whose reference MFCCs are a function of FFT round-off:
    mask    -- ill_mask(): the frames (and their successors) that get the documented 1e-5 row term on their MFCC rows;
    silent  -- frames of DIGITAL SILENCE (all raw samples equal): their spectrum is exactly [2|c|, 0, 0, ...] on the GPU, so every
               mel energy is exactly 0 and the MFCCs are the analytic vector silent_mfcc() -- checked at 1e-9, whatever the
               reference's pocketfft left in its non-DC bins;
    other   -- flagged frames that are neither silent nor the successor of a silent frame (a tone sitting exactly on an FFT
               bin, ...): only these escape a tight MFCC check, and their number is BOUNDED per test (budget_ok)."""

    def __init__(self, mask, silent):
        self.mask = np.asarray(mask, dtype=bool)
        self.silent = np.asarray(silent, dtype=bool)
        related = self.silent.copy()
        related[1:] |= self.silent[:-1]
        self.other = self.mask & ~related

    def counts(self):
        return {"frames": int(len(self.mask)), "flagged": int(self.mask.sum()), "silent": int(self.silent.sum()),
                "flagged_not_silent": int(self.other.sum())}

    def budget_ok(self, max_other_share=1e-3, max_other_abs=0):
        """flagged frames that are not explained by digital silence: at most max(max_other_abs, share * frames)"""
        return int(self.other.sum()) <= max(int(max_other_abs), int(np.floor(max_other_share * len(self.mask))))


def silent_mask(signal, window, step):
    raw = np.lib.stride_tricks.sliding_window_view(np.asarray(signal, dtype=np.float64), int(window))[::int(step)]
    return raw.max(axis=1) == raw.min(axis=1)


def ill_info(signal, fs, window, step, factor=1e4):
    return IllInfo(ill_mask(signal, fs, window, step, factor), silent_mask(signal, window, step))


def silent_mfcc():
    """MFCCs of a frame whose 40 mel energies are exactly 0: the orthonormal DCT-II of the constant log10(eps) (:252-253)."""
    out = np.zeros(13)
    out[0] = np.log10(O.EPS) * np.sqrt(40.0)
    return out


def silent_mfcc_violations(got, info, tol=1e-9):
    """entries of the MFCC rows (8..20) of digitally silent frames that differ from silent_mfcc() by more than tol"""
    if info is None or not info.silent.any() or got.shape[0] not in (34, 68):
        return 0
    sub = got[8:21][:, info.silent]
    return int((np.abs(sub - silent_mfcc()[:, None]) > tol).sum())


def reference_matrix(mono, fs, window, step, deltas):
    """The full reference matrix: the plain-C oracle (45 k frames/s) for every frame, except the frames of DIGITAL SILENCE
    (all samples equal), which come from the NumPy oracle.  On such frames the reference's FFT (pocketfft) returns exact
    zeros for the non-DC bins and log10(E + eps) of the mel bands resolves that; oracle/paa_oracle.c has its own DFT and
    leaves 1e-17 there (-99.00180463 instead of -99.00180475 in mfcc_1) -- the NumPy oracle, which runs the same
    pocketfft as the reference and is pinned to it at 1e-9, is the authority for those frames.  Delta rows are
    differences of the base rows (:668-680) and are re-formed around the patched frames."""
    ref = c_oracle.feature_extraction(mono, fs, window, step, deltas)
    x = np.asarray(mono, dtype=np.float64)
    frames = np.lib.stride_tricks.sliding_window_view(x, window)[::step]
    silent = np.flatnonzero(frames.max(axis=1) == frames.min(axis=1))
    if len(silent):
        xn = O.normalize_clip(mono)
        tab = O.Tables(fs, window)
        spec = lambda t: O.magnitude_spectrum(xn[t * step:t * step + window], tab.nfft)          # noqa: E731
        for t in silent:
            X = spec(t)
            ref[:34, t] = O.frame_vector(xn[t * step:t * step + window], X, X if t == 0 else spec(t - 1), tab)
        if deltas:
            for t in sorted(set(silent) | set(silent + 1)):
                if t < ref.shape[1]:
                    ref[34:, t] = 0.0 if t == 0 else ref[:34, t] - ref[:34, t - 1]
    return ref


def contract_violations(got, ref, ill=None):
    """(count, mask) of entries outside the contract; MFCC rows of ill-conditioned frames get 1e-5 of the group scale.
    ill: bool mask over frames, or an IllInfo (its mask is used; the caller applies budget_ok / silent_mfcc_violations)."""
    if isinstance(ill, IllInfo):
        ill = ill.mask
    nbad, bad = O.mixed_tolerance_violations(got, ref, REL, ROW, FLOOR)
    if nbad and ill is not None and ill.any() and ref.shape[0] in (34, 68):
        _, loose = O.mixed_tolerance_violations(got, ref, REL, 1e-5, FLOOR)
        rows = [r for r in MFCC_ALL if r < ref.shape[0]]
        sub = bad[rows]
        sub[:, ill] = loose[rows][:, ill]
        bad[rows] = sub
        nbad = int(bad.sum())
    return nbad, bad
