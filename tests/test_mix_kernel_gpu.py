"""The in-place mixed-radix kernel (csrc/kernels_mix.hpp: every window whose FFT length factors into 2, 3, 5, 7, 11, 13
and that no register-FFT kernel covers -- 50 ms at 44.1 / 48 kHz, 40 ms at 44.1 / 48 kHz, 1024, odd windows, ...) against
the plain-C oracle on every frame, through the C ABI.  -m gpu.  (ShortTermFeatures.py:608-682, :389-452, :324-386)"""
import numpy as np
import pytest

import c_oracle
import paa_oracle as O
from pyaudioanalysis_amd import ShortTermFeatures, _ffi
from synth import synth_clip
from test_ct_kernels_gpu import ill_info, make_signal, reference_matrix
from test_parity_gpu import assert_parity

pytestmark = pytest.mark.gpu


def test_plans_dispatch_the_mixed_radix_kernel(gpu_lib):
    def name(fs, w, s, kind=0, mode=0):
        plan = _ffi.Plan(np.array([0, 20 * fs], dtype=np.int64), fs, w, s, deltas=False, sample_kind=kind, mode=mode)
        try:
            return plan.kernel_name
        finally:
            plan.destroy()
    assert name(48000, 2400, 1200) == "st_tri_20x20x3"         # 50 ms at 48 kHz (audioTrainTest.py:28-29): kernels_tri.hpp
    assert name(44100, 2205, 1102) == "st_tri_r21x21x5"        # 50 ms at 44.1 kHz: odd window, real-input three-pass FFT
    assert name(48000, 2400, 1200, mode=1) == "spectrogram_tri_20x20x3"
    assert name(44100, 2205, 1102, kind=2, mode=2) == "chromagram_tri_r21x21x5"
    assert name(96000, 4800, 2400) == "st_mix"                 # 50 ms at 96 kHz stays with the in-place transform
    assert name(44100, 1764, 1764, mode=1) == "spectrogram_tri_21x21x2"    # the CLI's 40 ms at 44.1 kHz (audioAnalysis.py:71)
    assert name(48000, 1920, 1920, mode=2) == "chromagram_tri_20x16x3"
    assert name(32000, 1600, 800) == "st_tri_20x20x2"          # 50 ms at 32 kHz
    assert name(24000, 1200, 600) == "st_tri_20x10x3"          # 50 ms at 24 kHz / 25 ms at 48 kHz
    # 50 ms at 11.025 kHz and 25 ms at 22.05 kHz (audioTrainTest.py:28-29): 551 = 19 x 29, odd -> real-input two-pass FFT
    assert name(11025, 551, 275) == "st_tri_r29x19"
    assert name(22050, 551, 220, kind=1) == "st_tri_r29x19"
    assert name(44100, 1755, 877, mode=1) == "spectrogram_mix"   # other lengths made of 2, 3, 5, 7, 11, 13 stay mixed-radix
    # power-of-two windows: three-pass register FFT since round 5 (8 x 8 x 8, 16 x 16 x 4, 4 x 8 x 8) and round 6 (256 = 4 x 4 x 8); 4096 stays mixed-radix
    assert name(16000, 1024, 512, kind=1) == "st_tri_8x8x8"
    assert name(44100, 2048, 1024) == "st_tri_16x16x4"
    assert name(16000, 512, 256, kind=2) == "st_tri_4x8x8"
    assert name(16000, 1024, 512, mode=1) == "spectrogram_tri_8x8x8"
    assert name(44100, 2048, 512, mode=2) == "chromagram_tri_16x16x4"
    assert name(16000, 256, 128) == "st_tri_4x4x8"             # 16 ms at 16 kHz: three-pass register FFT since round 6
    assert name(16000, 256, 64, kind=2, mode=1) == "spectrogram_tri_4x4x8"
    assert name(44100, 4096, 2048) == "st_mix"
    assert name(44100, 1102, 441) == "st_tri_r19x29x2"         # config 5's features: real-input 19 x 29 x 2, the radix-29 butterflies shared by three lanes
    assert name(44100, 1102, 441, kind=2, mode=1) == "spectrogram_tri_r19x29x2"   # its rows too since round 5 (one slot, sixteen waves per CU)
    assert name(44100, 1102, 441, mode=2) == "chromagram_tri_r19x29x2"
    assert name(16000, 800, 400) == "st_fast_800_w8"
    assert name(22050, 1103, 441) == "st_blu_2048"             # 1103 is prime: Bluestein convolution of length 2048 (round 6; st_generic until then)


CASES = [
    # fs, window, step, kind, seconds, deltas
    (48000, 2400, 1200, "i16", 20, True),      # radices 8 4 4 ... 5 5 3, twiddles from global memory
    (48000, 2400, 2400, "stereo", 15, False),
    (48000, 2400, 480, "f64", 6, True),
    (44100, 2205, 1102, "i16", 20, True),      # odd: 2205 = 3 3 5 7 7
    (44100, 2205, 2205, "f64", 10, False),
    (44100, 1764, 882, "i16", 15, True),       # 882 = 2 3 3 7 7
    (48000, 1920, 960, "stereo", 10, True),    # 960 = 8 8 5 3
    (32000, 1600, 800, "i16", 15, False),      # 50 ms at 32 kHz
    (16000, 1024, 512, "i16", 20, True),       # 512 = 8 x 8 x 8: three-pass register FFT (round 5)
    (16000, 1024, 1024, "stereo", 20, False),
    (16000, 1024, 333, "f64", 12, True),
    (16000, 512, 256, "f64", 10, True),        # 256 = 4 x 8 x 8; entropy blocks of 51 samples (a pair straddles the boundary)
    (16000, 512, 256, "i16", 20, True),
    (8000, 512, 512, "stereo", 20, False),
    (44100, 2048, 1024, "i16", 20, True),      # 1024 = 16 x 16 x 4
    (44100, 2048, 441, "f64", 8, False),
    (48000, 2048, 2048, "stereo", 15, True),
    (22050, 1100, 550, "i16", 10, False),      # 550 = 2 5 5 11
    (11025, 551, 275, "i16", 20, True),        # 50 ms at 11.025 kHz: odd, 19 x 29 -> two-pass real-input kernel (kernels_tri.hpp)
    (22050, 551, 220, "f64", 15, True),        # 25 ms at 22.05 kHz
    (22050, 551, 551, "stereo", 15, False),
    (24000, 1200, 600, "i16", 10, True),       # 600 = 20 x 10 x 3
    (32000, 1600, 1600, "f64", 10, True),
    (16000, 390, 200, "i16", 10, True),        # 195 = 3 5 13
    (16000, 1001, 500, "unit", 10, False),     # odd: 7 11 13
    (16000, 256, 128, "i16", 5, True),         # 128 = 4 x 4 x 8 (round 6); entropy blocks of 25 samples, tail of 6
    (16000, 256, 256, "stereo", 8, False),
    (22050, 256, 100, "f64", 4, True),
    (96000, 4800, 2400, "i16", 6, True),       # 50 ms at 96 kHz: 2400 complex points, 70 KB per wave (two waves per CU)
    (44100, 3465, 1000, "f64", 4, False),      # odd and large: 3 3 5 7 11, 73 KB per wave
    (44100, 6000, 3000, "stereo", 5, False),   # the edge of the magnitude pass's register slots (1501 pairs)         # small window: 128 = 8 4 4
]


@pytest.mark.parametrize("fs,window,step,kind,seconds,deltas", CASES,
                         ids=["%d_%d_%d_%s_%ds_%s" % (c[0], c[1], c[2], c[3], c[4], "d" if c[5] else "n") for c in CASES])
def test_full_matrix_against_c_oracle(gpu_lib, fs, window, step, kind, seconds, deltas):
    sig, mono = make_signal(kind, 9000 + window + step, seconds, fs)
    F, names = ShortTermFeatures.feature_extraction(sig, fs, window, step, deltas)
    ref = reference_matrix(mono, fs, window, step, deltas)
    assert F.shape == ref.shape and len(names) == ref.shape[0]
    assert_parity(F, ref, "%s %d/%d @%d" % (kind, window, step, fs), ill=ill_info(mono, fs, window, step))
    if deltas:
        assert np.array_equal(F[34:, 1:], F[:34, 1:] - F[:34, :-1]) and np.all(F[34:, 0] == 0.0)
        G, _ = ShortTermFeatures.feature_extraction(sig, fs, window, step, False)
        assert np.array_equal(G, F[:34])


@pytest.mark.parametrize("fs,window,step,kind", [(48000, 2400, 1200, "i16"), (44100, 2205, 1102, "stereo"),
                                                  (44100, 1764, 1764, "i16"), (48000, 1920, 1920, "f64"),
                                                  (16000, 1024, 300, "i16"), (11025, 551, 275, "i16"),
                                                  (32000, 1600, 800, "stereo"), (24000, 1200, 1200, "f64"),
                                                  (44100, 2048, 1024, "stereo"), (16000, 512, 256, "i16"), (16000, 1024, 512, "f64"),
                                                  (16000, 256, 128, "i16"), (16000, 256, 64, "stereo")])
def test_spectrogram_chromagram_full_against_c_oracle(gpu_lib, capsys, fs, window, step, kind):
    sig, mono = make_signal(kind, 9100 + window, 12.7, fs)
    spec, t_ax, f_ax = ShortTermFeatures.spectrogram(sig, fs, window, step)
    capsys.readouterr()
    ref = c_oracle.spectrogram(mono, window, step)
    assert spec.shape == ref.shape and len(t_ax) == ref.shape[0] and len(f_ax) == window // 2
    assert_parity(np.ascontiguousarray(spec.T), np.ascontiguousarray(ref.T), "spectrogram %s %d/%d" % (kind, window, step))
    chroma, ct_ax, cnames = ShortTermFeatures.chromagram(sig, fs, window, step)
    cref = c_oracle.chromagram(mono, fs, window, step)
    assert chroma.shape == cref.shape and cnames == O.CHROMA_NAMES
    assert_parity(np.ascontiguousarray(chroma.T), np.ascontiguousarray(cref.T), "chromagram %s %d/%d" % (kind, window, step))


def test_degenerate_clips_and_ragged_batches(gpu_lib):
    fs, W, S = 48000, 2400, 1200
    cases = {
        "zeros": np.zeros(5 * W, dtype=np.int16),
        "one_window": synth_clip(91, W, fs),
        "w_plus_s_minus_1": synth_clip(92, W + S - 1, fs),
        "dc": np.full(4 * W, 1234, dtype=np.int16),
    }
    # (no full-scale square wave at fs / 2 here: all of its energy sits in the Nyquist bin the reference drops, and with
    # radix-3 / radix-5 factors in the transform bins 0 .. Nf-1 hold nothing but the FFT's round-off -- the reference's own
    # centroid and spread of that frame are functions of pocketfft's rounding; the 2 RA RB family's test keeps the case)
    x = synth_clip(93, 3 * fs, fs).copy()
    x[fs:2 * fs] = 0
    cases["silence_inside"] = x
    for label, sig in cases.items():
        F, _ = ShortTermFeatures.feature_extraction(sig, fs, W, S)
        ref, _ = O.feature_extraction(sig, fs, W, S)
        assert_parity(F, ref, label, sig=(sig, fs, W, S))
    with pytest.raises(ValueError):
        ShortTermFeatures.feature_extraction(synth_clip(94, W - 1, fs), fs, W, S)
    lens = [W, 50000, 2 * W - 1, 96000, W + S, 7 * W + 3]
    clips = [synth_clip(9600 + i, n, fs) for i, n in enumerate(lens)]
    res, _ = ShortTermFeatures.feature_extraction_batch(clips, fs, W, S, deltas=True)
    for c, r in zip(clips, res):
        single, _ = ShortTermFeatures.feature_extraction(c, fs, W, S)
        assert np.array_equal(single, r)
        assert_parity(r, reference_matrix(c, fs, W, S, True), "ragged batch", ill=ill_info(c, fs, W, S))


@pytest.mark.parametrize("fs,window,step,kind", [(44100, 2205, 1102, "i16"), (44100, 1102, 441, "stereo"), (44100, 1102, 441, "i16"),
                                                  (48000, 2400, 1200, "stereo"), (48000, 2400, 1200, "i16"),
                                                  (11025, 551, 275, "stereo"), (44100, 1764, 882, "f64"), (16000, 800, 400, "i16"),
                                                  (16000, 640, 320, "stereo"), (16000, 1024, 512, "i16"), (16000, 512, 256, "stereo"),
                                                  (44100, 2048, 1024, "i16"), (16000, 256, 128, "i16"), (16000, 256, 128, "stereo")])
def test_samples_that_sit_on_a_whole_number_mean(gpu_lib, fs, window, step, kind):
    """The kernels decide sign(x / 2^15 - mean) of integer PCM in integer arithmetic (x against floor(mean 2^15), with a
    separate rule when the clip mean is a whole count: then samples can sit exactly ON the mean and np.sign gives 0,
    ShortTermFeatures.py:22-26).  A mirrored clip (a, -a) + c has the whole-number mean c and every fifth sample on it; the
    zero-crossing row is discrete -- assert_parity's tight gate counts any flip."""
    rng = np.random.default_rng(window + step)
    n = 6 * fs // 2
    def mirrored(c, amp):
        a = rng.integers(-amp, amp + 1, n)
        a[rng.random(n) < 0.2] = 0
        return np.concatenate([a, -a]) + c
    if kind == "i16":
        sig = mirrored(-7, 3000).astype(np.int16)
        mono = sig
    else:
        left, right = mirrored(40, 2500), mirrored(-27, 2500)       # L + R has the whole-number mean 13; mono mean 6.5
        sig = np.stack([left, right], axis=1).astype(np.int16)
        mono = O.stereo_to_mono(sig)
        if kind == "f64":
            sig = mono
    assert float(np.mean(np.double(mono) * (2.0 if kind != "i16" else 1.0))) in (-7.0, 13.0)
    F, _ = ShortTermFeatures.feature_extraction(sig, fs, window, step, False)
    ref = reference_matrix(mono, fs, window, step, False)
    assert_parity(F, ref, "whole mean %s %d/%d" % (kind, window, step), ill=ill_info(mono, fs, window, step))
    counts = lambda row: np.rint(row * 2.0 * (window - 1))        # zcr = sum |diff(sign)| / 2 / (W - 1): whole numbers
    assert np.array_equal(counts(F[0]), counts(ref[0])) and np.abs(F[0] * 2.0 * (window - 1) - counts(F[0])).max() < 1e-6
