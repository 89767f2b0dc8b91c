"""The register-FFT family for windows 2 RA RB (csrc/kernels_ct.hpp): 800 (float64 / stereo / any step), 640, 400, 320
-- the shapes the reference's own callers use besides the int16 800/400 headline (audioAnalysis.py:66-81,
audioBasicIO.py:167, audioTrainTest.py:28-29).  Every case is a FULL-matrix comparison of the HIP path (through the C ABI)
with the plain-C oracle (oracle/paa_oracle.c, pinned to the reference's goldens), contract + tight gate, on clips long
enough to span many runs of the kernel's tiling.  -m gpu."""
import numpy as np
import pytest

import c_oracle
import paa_oracle as O
from pyaudioanalysis_amd import MidTermFeatures, ShortTermFeatures, _ffi
from synth import synth_clip
from test_parity_gpu import assert_parity

pytestmark = pytest.mark.gpu


from checks import ill_info, reference_matrix          # noqa: E402,F401  (shared with bench.py's --check leg)


def make_signal(kind, seed, seconds, fs):
    """kind: 'i16' mono PCM, 'f64' what stereo_to_mono returns for a stereo file (.5 fractions), 'stereo' (n, 2) int16,
    'unit' a float signal in [-1, 1] (not a multiple of anything)"""
    n = int(seconds * fs)
    if kind == "i16":
        x = synth_clip(seed, n, fs)
        return x, x
    xs = synth_clip(seed, n, fs, stereo=True)
    if kind == "stereo":
        return xs, O.stereo_to_mono(xs)
    if kind == "f64":
        m = O.stereo_to_mono(xs)
        return m, m
    u = O.stereo_to_mono(xs) / 32768.0 * np.pi / 3
    return u, u


def test_plans_dispatch_the_register_fft_family(gpu_lib):
    """plan.kernel_name for the shapes VERDICT r02 lists -- none of them may fall to st_generic."""
    def name(fs, w, s, kind=0, mode=0):
        plan = _ffi.Plan(np.array([0, 20 * fs], dtype=np.int64), fs, w, s, deltas=False, sample_kind=kind, mode=mode)
        try:
            return plan.kernel_name
        finally:
            plan.destroy()
    assert name(16000, 640, 640) == "st_ct_20x16"
    assert name(16000, 800, 400, kind=1) == "st_ct_25x16"
    assert name(16000, 800, 400, kind=2) == "st_ct_25x16"
    assert name(16000, 800, 160) == "st_ct_25x16"               # int16 at a step the 800/400 kernel does not have
    assert name(16000, 800, 400) == "st_fast_800_w8"            # the headline shape keeps its own kernel
    assert name(8000, 400, 200) == "st_ct_25x8"
    assert name(16000, 320, 160) == "st_ct_10x16"
    assert name(16000, 640, 640, mode=1) == "spectrogram_ct_20x16"
    assert name(16000, 640, 640, mode=2) == "chromagram_ct_20x16"


CASES = [
    # fs, window, step, kind, seconds, deltas
    (16000, 640, 640, "i16", 30, True),        # the CLI's 40 ms / 40 ms
    (16000, 640, 320, "i16", 20, False),
    (16000, 640, 441, "stereo", 61, True),     # odd step: every other frame starts on an odd sample (unaligned pairs)
    (16000, 640, 640, "f64", 61, False),
    (16000, 800, 400, "f64", 61, True),        # what audioBasicIO.stereo_to_mono hands the reference for a stereo file
    (16000, 800, 400, "stereo", 61, True),     # the same file with the channel sum fused on the device
    (16000, 800, 400, "unit", 20, False),      # arbitrary float samples
    (16000, 800, 800, "f64", 30, True),
    (16000, 800, 160, "i16", 20, True),        # 50 ms / 10 ms
    (16000, 800, 333, "i16", 20, False),
    (16000, 800, 1000, "i16", 30, True),       # step > window (gaps between frames)
    (8000, 400, 200, "i16", 60, True),         # 50 ms / 25 ms at 8 kHz
    (8000, 400, 400, "f64", 30, False),
    (8000, 400, 80, "stereo", 20, True),
    (16000, 320, 160, "i16", 30, True),        # 20 ms / 10 ms
    (16000, 320, 320, "f64", 20, False),
    (22050, 640, 320, "i16", 20, True),        # another sampling rate: other mel / chroma tables
    (44100, 800, 441, "i16", 10, False),
]


@pytest.mark.parametrize("fs,window,step,kind,seconds,deltas", CASES,
                         ids=["%d_%d_%d_%s_%ds_%s" % (c[0], c[1], c[2], c[3], c[4], "d" if c[5] else "n") for c in CASES])
def test_full_matrix_against_c_oracle(gpu_lib, fs, window, step, kind, seconds, deltas):
    sig, mono = make_signal(kind, 7000 + window + step, seconds, fs)
    F, names = ShortTermFeatures.feature_extraction(sig, fs, window, step, deltas)
    ref = reference_matrix(mono, fs, window, step, deltas)
    assert F.shape == ref.shape and len(names) == ref.shape[0]
    assert_parity(F, ref, "%s %d/%d @%d" % (kind, window, step, fs), ill=ill_info(mono, fs, window, step))
    if deltas:
        assert np.array_equal(F[34:, 1:], F[:34, 1:] - F[:34, :-1]) and np.all(F[34:, 0] == 0.0)
        G, _ = ShortTermFeatures.feature_extraction(sig, fs, window, step, False)
        assert np.array_equal(G, F[:34])                                     # the 34-row kernel gives the same bits


@pytest.mark.parametrize("fs,window,step,kind", [(16000, 640, 640, "i16"), (16000, 640, 640, "stereo"),
                                                  (16000, 800, 400, "f64"), (8000, 400, 200, "i16"),
                                                  (16000, 320, 160, "stereo"), (16000, 640, 441, "i16")])
def test_spectrogram_chromagram_full_against_c_oracle(gpu_lib, capsys, fs, window, step, kind):
    sig, mono = make_signal(kind, 7100 + window, 25.3, fs)
    spec, t_ax, f_ax = ShortTermFeatures.spectrogram(sig, fs, window, step)
    capsys.readouterr()
    ref = c_oracle.spectrogram(mono, window, step)
    assert spec.shape == ref.shape and len(t_ax) == ref.shape[0] and len(f_ax) == window // 2
    assert_parity(np.ascontiguousarray(spec.T), np.ascontiguousarray(ref.T), "spectrogram %s %d/%d" % (kind, window, step))
    chroma, ct_ax, cnames = ShortTermFeatures.chromagram(sig, fs, window, step)
    cref = c_oracle.chromagram(mono, fs, window, step)
    assert chroma.shape == cref.shape and cnames == O.CHROMA_NAMES
    assert_parity(np.ascontiguousarray(chroma.T), np.ascontiguousarray(cref.T), "chromagram %s %d/%d" % (kind, window, step))


def test_degenerate_clips_through_the_family(gpu_lib):
    """all-zero clip, exactly one window, constant DC, full-scale square wave, a clip with digital silence inside"""
    fs, W, S = 16000, 640, 640
    cases = {
        "zeros": np.zeros(5 * W, dtype=np.int16),
        "one_window": synth_clip(81, W, fs),
        "w_plus_s_minus_1": synth_clip(82, W + S - 1, fs),
        "dc": np.full(4 * W, 1234, dtype=np.int16),
        "square": np.tile(np.array([32767, -32768], dtype=np.int16), 3 * W // 2),
    }
    x = synth_clip(83, 6 * fs, fs).copy()
    x[2 * fs:3 * fs] = 0
    cases["silence_inside"] = x
    for label, sig in cases.items():
        F, _ = ShortTermFeatures.feature_extraction(sig, fs, W, S)
        ref, _ = O.feature_extraction(sig, fs, W, S)
        # (the square wave at fs / 2 is BUILT to be ill-conditioned: all of its energy sits in the Nyquist bin the reference drops,
        # every kept bin holds FFT round-off only -- it states its own allowance for the bounded MFCC exception)
        assert_parity(F, ref, label, sig=(sig, fs, W, S), max_other_share=1.0 if label == "square" else 1e-3)
    z, _ = ShortTermFeatures.feature_extraction(cases["zeros"], fs, W, S, deltas=False)
    assert abs(z[8, 0] - 40.0 * np.log10(O.EPS) / np.sqrt(40.0)) < 1e-9 and np.all(np.abs(z[9:21]) < 1e-12)   # mfcc of silence
    with pytest.raises(ValueError):
        ShortTermFeatures.feature_extraction(synth_clip(84, W - 1, fs), fs, W, S)


def test_ragged_batches_and_mid_term_through_the_family(gpu_lib):
    fs, W, S = 16000, 640, 320
    lens = [W, 5000, 16000, 640 * 7 + 3, 48000, 2 * W - 1, 31999]
    clips = [synth_clip(8600 + i, n, fs) for i, n in enumerate(lens)]
    res, _ = ShortTermFeatures.feature_extraction_batch(clips, fs, W, S, deltas=True)
    for c, r in zip(clips, res):
        single, _ = ShortTermFeatures.feature_extraction(c, fs, W, S)
        assert np.array_equal(single, r)
        assert_parity(r, reference_matrix(c, fs, W, S, True), "ragged batch", ill=ill_info(c, fs, W, S))
    # float64 clips batch too (what the directory walkers do with stereo / non-int16 files)
    fclips = [O.stereo_to_mono(synth_clip(8700 + i, n, fs, stereo=True)) for i, n in enumerate(lens[1:5])]
    fres, _ = ShortTermFeatures.feature_extraction_batch(fclips, fs, 800, 400, deltas=True)
    for c, r in zip(fclips, fres):
        single, _ = ShortTermFeatures.feature_extraction(c, fs, 800, 400)
        assert np.array_equal(single, r)
    mids, _ = MidTermFeatures.mid_feature_extraction_batch(fclips, fs, fs, fs, 800, 400)
    for c, m in zip(fclips, mids):
        ref_mid, ref_st, _ = O.mid_feature_extraction(c, fs, fs, fs, 800, 400)
        assert_parity(m, ref_mid, "float64 mid batch", sig=(c, fs, 800, 400))
