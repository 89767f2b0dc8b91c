"""Parity at BASELINE.json's FULL sizes for the configurations the golden / seeded tests only cover small
(configs[2..4]): the whole batch runs through the C ABI, a strided sample of its units is checked against the CPU
oracle with the tight gate of test_parity_gpu, and size-independent properties cover the rest (identical clips give
identical slabs wherever they sit, rows the reference leaves zero stay zero, deltas are exact differences).  -m gpu."""
import numpy as np
import pytest

import paa_oracle as O
from pyaudioanalysis_amd import MidTermFeatures, ShortTermFeatures, _ffi
from synth import synth_clip
from test_parity_gpu import assert_parity

pytestmark = pytest.mark.gpu
FS = 16000


def test_config4_shard_12500_clips(gpu_lib):
    """One GPU's share of config 4: 12 500 x 10 s clips (16 distinct seeded clips, tiled), 800/400, 34 rows."""
    pool = [synth_clip(40000 + i, 10 * FS) for i in range(16)]
    clips = [pool[k % 16] for k in range(12500)]
    res, names = ShortTermFeatures.feature_extraction_batch(clips, FS, 800, 400, deltas=False)
    assert len(res) == 12500 and len(names) == 34 and res[0].shape == (34, 399)
    for k in range(16, 12500):                       # position in the batch never changes a clip's bits
        assert np.array_equal(res[k], res[k % 16]), k
    for k in (0, 5, 1037, 6250, 12499):              # distinct clips at distinct places against the oracle
        ref, _ = O.feature_extraction(clips[k], FS, 800, 400, deltas=False)
        assert_parity(res[k], ref, "config 4 clip %d" % k, sig=(clips[k], FS, 800, 400))


def test_config3_1000_clips_mid_term(gpu_lib):
    """Config 3: mid_feature_extraction (1.0 s / 1.0 s over 50 ms / 25 ms) on 1000 x 30 s clips in one batch."""
    pool = [synth_clip(3000 + i, 30 * FS) for i in range(8)]
    clips = [pool[k % 8] for k in range(1000)]
    mids, sts, names = MidTermFeatures.mid_feature_extraction_batch(clips, FS, FS, FS, 800, 400, return_short=True)
    assert len(mids) == 1000 and mids[0].shape == (136, 30) and sts[0].shape == (68, 1199) and len(names) == 136
    for k in range(8, 1000):
        assert np.array_equal(mids[k], mids[k % 8]), k
    for k in (0, 3, 501, 999):
        ref_mid, ref_st, _ = O.mid_feature_extraction(clips[k], FS, FS, FS, 800, 400)
        sig = (clips[k], FS, 800, 400)
        assert_parity(sts[k], ref_st, "config 3 short %d" % k, sig=sig)
        assert_parity(mids[k], ref_mid, "config 3 mid %d" % k, sig=sig)


@pytest.fixture(scope="module")
def cfg5_clip():
    fs = 44100
    xs = synth_clip(5, 600 * fs, fs=fs, stereo=True)          # SEED = 5, 600 s, stereo int16
    mono = O.stereo_to_mono(xs)
    return fs, xs, mono, O.normalize_clip(mono)


def test_config5_features_600s_stereo(gpu_lib, cfg5_clip):
    """Config 5: 44.1 kHz stereo -> mono fused on the device, window 25 ms / step 10 ms (1102 / 441)."""
    fs, xs, mono, xn = cfg5_clip
    W, S = int(0.025 * fs), int(0.010 * fs)
    F, _ = ShortTermFeatures.feature_extraction(xs, fs, 0.025 * fs, 0.010 * fs)
    T = (len(mono) - W) // S + 1
    assert F.shape == (68, T) and np.all(np.isfinite(F))
    assert np.array_equal(F[34:, 1:], F[:34, 1:] - F[:34, :-1]) and np.all(F[34:, 0] == 0.0)
    tab = O.Tables(fs, W)
    frames = [0, 1, 2, 77, T // 3, T // 2, T - 2, T - 1]
    ref = np.empty((34, len(frames)))
    for n, t in enumerate(frames):
        fr = xn[t * S:t * S + W]
        X = O.magnitude_spectrum(fr, tab.nfft)
        Xp = X if t == 0 else O.magnitude_spectrum(xn[(t - 1) * S:(t - 1) * S + W], tab.nfft)
        ref[:, n] = O.frame_vector(fr, X, Xp, tab)
    assert_parity(np.ascontiguousarray(F[:34, frames]), ref, "config 5 sampled frames")


def test_config5_spectrogram_chromagram_600s(gpu_lib, cfg5_clip, capsys):
    """Config 5's own path: spectrogram (:389-452) and chromagram (:324-386) of the 600 s clip, including the rows the
    reference allocates but never fills and the truncated tail frame of the chromagram."""
    fs, xs, mono, xn = cfg5_clip
    W, S = 1102, 441
    n = len(mono)
    spec, t_ax, f_ax = ShortTermFeatures.spectrogram(xs, fs, W, S)
    assert "(%d, %d)" % spec.shape in capsys.readouterr().out
    rows = int((n - W) / S) + 1
    starts = list(range(W, n - W + 1, S))
    assert spec.shape == (rows, W // 2) and len(t_ax) == rows and len(f_ax) == W // 2
    assert np.all(spec[len(starts):] == 0.0)                     # allocated, never filled (:413-415)
    pick = [0, 1, 2, len(starts) // 2, len(starts) - 2, len(starts) - 1]
    ref = np.stack([O.magnitude_spectrum(xn[starts[i]:starts[i] + W], W // 2) for i in pick])
    assert_parity(np.ascontiguousarray(spec[pick].T), np.ascontiguousarray(ref.T), "config 5 spectrogram rows")

    chroma, ct_ax, cnames = ShortTermFeatures.chromagram(xs, fs, W, S)
    crows = int((n - S - W) / S) + 1
    cstarts = list(range(W, n - S, S))
    assert chroma.shape == (crows, 12) and len(ct_ax) == crows and cnames == O.CHROMA_NAMES
    assert len(xn[cstarts[-1]:cstarts[-1] + W]) < W              # the last frame really is truncated (:349-355)
    assert np.all(chroma[len(cstarts):] == 0.0)
    tab = O.Tables(fs, W)
    pick = [0, 1, len(cstarts) // 2, len(cstarts) - 3, len(cstarts) - 2, len(cstarts) - 1]
    ref = np.zeros((len(pick), 12))
    for m, i in enumerate(pick):
        x = xn[cstarts[i]:cstarts[i] + W]
        X = np.abs(np.fft.fft(x))[0:W // 2]
        X = X / len(X)
        P = X ** 2
        grid = np.zeros((int(np.ceil((W // 2) / 12.0)) * 12,))
        grid[tab.ch_pos] = P[tab.ch_src] * tab.ch_w
        c = grid.reshape(-1, 12).sum(axis=0)
        ref[m] = c / O.EPS if P.sum() == 0 else c / P.sum()
    assert_parity(np.ascontiguousarray(chroma[pick].T), np.ascontiguousarray(ref.T), "config 5 chromagram rows")


# ---------------------------------------------------------------------------------------------------------
# full-matrix checks at BASELINE sizes against the plain-C oracle (oracle/paa_oracle.c: ~45 k frames/s on one core)
# ---------------------------------------------------------------------------------------------------------
def _ill_mask(signal, fs, window, step):
    from test_ct_kernels_gpu import ill_info
    return ill_info(signal, fs, window, step)


def _reference_matrix(mono, fs, window, step, deltas):
    from test_ct_kernels_gpu import reference_matrix          # C oracle + the NumPy oracle on digitally silent frames
    return reference_matrix(mono, fs, window, step, deltas)


@pytest.mark.parametrize("deltas", [False, True], ids=["34rows", "68rows"])
def test_config2_one_hour_full_matrix(gpu_lib, deltas):
    """BASELINE config 2 itself: the whole (34 | 68) x 143 999 matrix of the seeded 1-hour clip against the C oracle,
    contract and tight gate on every entry (ShortTermFeatures.py:608-682)."""
    import c_oracle
    x = synth_clip(2, 3600 * FS)
    F, _ = ShortTermFeatures.feature_extraction(x, FS, 800, 400, deltas)
    assert F.shape == (68 if deltas else 34, 143999)
    ref = _reference_matrix(x, FS, 800, 400, deltas)
    assert_parity(F, ref, "config 2 full matrix", ill=_ill_mask(x, FS, 800, 400))


def test_config5_full_matrices(gpu_lib, cfg5_clip, capsys):
    """Config 5: every frame of the 600 s stereo clip's feature matrix, and every row of the spectrogram and chromagram of
    its first 60 s, against the C oracle."""
    import c_oracle
    fs, xs, mono, xn = cfg5_clip
    W, S = 1102, 441
    F, _ = ShortTermFeatures.feature_extraction(xs, fs, W, S, deltas=False)
    ref = _reference_matrix(mono, fs, W, S, False)
    assert_parity(F, ref, "config 5 features, all frames", ill=_ill_mask(mono, fs, W, S))
    n60 = 60 * fs
    spec, _, _ = ShortTermFeatures.spectrogram(xs[:n60], fs, W, S)
    capsys.readouterr()
    assert_parity(np.ascontiguousarray(spec.T), np.ascontiguousarray(c_oracle.spectrogram(mono[:n60], W, S).T),
                  "config 5 spectrogram, 60 s, all rows")
    chroma, _, _ = ShortTermFeatures.chromagram(xs[:n60], fs, W, S)
    assert_parity(np.ascontiguousarray(chroma.T), np.ascontiguousarray(c_oracle.chromagram(mono[:n60], fs, W, S).T),
                  "config 5 chromagram, 60 s, all rows")


def _plan_matrix(x, deltas, shape):
    plan = _ffi.Plan(np.array([0, len(x)], dtype=np.int64), FS, 800, 400, deltas=deltas, sample_kind=0)
    try:
        d_in = _ffi.DeviceBuffer.from_host(x)
        d_out = _ffi.DeviceBuffer(plan.out_doubles * 8)
        plan.execute(d_in, d_out)
        _ffi.sync()
        return d_out.to_host(np.float64, plan.out_doubles).reshape(shape)
    finally:
        plan.destroy()


def test_ranged_host_call_equals_the_single_launch_bit_for_bit(gpu_lib):
    """A long clip handed to the host-buffer API is computed in four consecutive frame ranges whose columns are copied back
    while the next range computes (csrc/lib_host_api.hpp: run_host_st, shorter runs per wave).  Every bit must equal the
    device-resident plan's single launch over the same clip (ShortTermFeatures.py:608-682 is the same arithmetic per frame
    whatever the tiling), with and without deltas, and on both sides of the 65 536-frame threshold."""
    x = synth_clip(2, 1700 * FS)                                                        # 67 999 frames: the ranged path
    for deltas in (False, True):
        F, _ = ShortTermFeatures.feature_extraction(x, FS, 800, 400, deltas)
        assert F.shape[1] == 67999
        assert np.array_equal(F, _plan_matrix(x, deltas, F.shape))
    y = x[:65535 * 400 + 400]                                                           # 65 535 frames: one launch
    Fy, _ = ShortTermFeatures.feature_extraction(y, FS, 800, 400, True)
    assert Fy.shape[1] == 65535
    assert np.array_equal(Fy, _plan_matrix(y, True, Fy.shape))


@pytest.mark.parametrize("fs,window,step,deltas,family", [
    (16000, 640, 320, True, "st_ct_20x16"),            # 2 RA RB family: runs after a clip's first start one or two frames early (halo inside)
    (48000, 2400, 1200, True, "st_tri_20x20x3"),       # three-pass family, with deltas
    (16000, 1024, 512, False, "st_tri_8x8x8"),         # ... a power-of-two shape of round 5
    (16000, 256, 128, False, "st_tri_4x4x8"),          # ... and of round 6 (st_mix until then)
    (16000, 390, 195, False, "st_mix"),                # in-place mixed-radix kernel (195 = 3 x 5 x 13, lean instance)
    (22050, 661, 330, True, "st_blu_1024"),            # Bluestein kernel (round 6): row chunks in registers, stored one frame late
    (11025, 551, 275, True, "st_tri_r29x19"),          # shared prime butterflies
])
def test_ranged_host_call_on_the_other_families(gpu_lib, fs, window, step, deltas, family):
    """The four-range copy-back pipeline of the host-buffer API applies to every kernel family (advisor, round 4): a clip of
    more than 65 536 frames through ct / tri / mix / blu equals the device-resident plan's single launch bit for bit, and the profiling
    events count the ranged launches."""
    import ctypes
    n = 65999 * step + window                                                           # 66 000 frames
    x = np.tile(synth_clip(31, 60 * fs, fs), -(-n // (60 * fs)))[:n]
    _ffi.check(gpu_lib.paa_prof_enable(1))
    F, _ = ShortTermFeatures.feature_extraction(x, fs, window, step, deltas)
    ms, cnt = ctypes.c_double(), ctypes.c_int64()
    _ffi.check(gpu_lib.paa_prof_read(ctypes.byref(ms), ctypes.byref(cnt)))
    _ffi.check(gpu_lib.paa_prof_enable(0))
    assert F.shape[1] == 66000 and cnt.value >= 2, cnt.value          # several ranged launches, every one bracketed
    plan = _ffi.Plan(np.array([0, len(x)], dtype=np.int64), fs, window, step, deltas=deltas, sample_kind=0)
    try:
        assert plan.kernel_name == family
        d_in = _ffi.DeviceBuffer.from_host(x)
        d_out = _ffi.DeviceBuffer(plan.out_doubles * 8)
        plan.execute(d_in, d_out)
        _ffi.sync()
        assert np.array_equal(F, d_out.to_host(np.float64, plan.out_doubles).reshape(F.shape))
    finally:
        plan.destroy()
