"""The real-input split of the 12 x 3675- / 6 x 3675-sample windows (csrc/kernels_wgs.hpp: 44 100 and 22 050 samples -- the 1 s window
audioSegmentation.music_thumbnailing passes to feature_extraction at 44.1 / 22.05 kHz, audioSegmentation.py:1134-1138) against the NumPy
oracle on every frame and row, through the C ABI: every sample type, features / spectrogram / chromagram, ragged batches, silent clips,
more frames than one chunk of the spectrum scratch holds, both gates.  -m gpu."""
import numpy as np
import pytest

import paa_oracle as O
from pyaudioanalysis_amd import ShortTermFeatures, _ffi
from synth import synth_clip
from test_ct_kernels_gpu import make_signal
from test_parity_gpu import assert_parity

pytestmark = pytest.mark.gpu


def kernel_name(fs, w, s, kind=0, mode=0):
    plan = _ffi.Plan(np.array([0, 4 * w], dtype=np.int64), fs, w, s, deltas=False, sample_kind=kind, mode=mode)
    try:
        return plan.kernel_name
    finally:
        plan.destroy()


def test_plans_dispatch_the_real_input_split(gpu_lib):
    assert kernel_name(44100, 44100, 22050) == "st_wgs_12x3675"
    assert kernel_name(44100, 44100, 44100, kind=2) == "st_wgs_12x3675"
    assert kernel_name(44100, 44100, 11025, kind=1, mode=1) == "spectrogram_wgs_12x3675"
    assert kernel_name(44100, 44100, 22050, mode=2) == "chromagram_wgs_12x3675"
    assert kernel_name(22050, 22050, 11025) == "st_wgs_6x3675"
    assert kernel_name(22050, 22050, 7000, kind=1) == "st_wgs_6x3675"
    assert kernel_name(44100, 22050, 11025, kind=2, mode=1) == "spectrogram_wgs_6x3675"        # 0.5 s at 44.1 kHz
    assert kernel_name(48000, 44100, 22050) == "st_wgs_12x3675"                                # any sampling rate: the tables are per (fs, window)
    assert kernel_name(44100, 44102, 22050) != "st_wgs_12x3675"                                # neighbours keep their kernels
    assert kernel_name(48000, 48000, 24000) == "st_wgs_12x4000"                                # 8 x 20 x 25 points per sub-transform
    assert kernel_name(32000, 32000, 16000, kind=2, mode=2) == "chromagram_wgs_8x4000"
    assert kernel_name(24000, 24000, 12000, kind=1) == "st_wgs_6x4000"


@pytest.mark.parametrize("kind,fs,window,step,seconds,deltas", [
    ("i16", 44100, 44100, 22050, 9.0, True),         # 17 frames x 3 tasks {1,2} {3,4} {5,packed}
    ("stereo", 44100, 44100, 30000, 7.3, False),     # interleaved stereo samples summed in the loads, a step that is no multiple of anything
    ("f64", 44100, 44100, 44100, 6.0, True),         # float64 samples (stereo_to_mono's .5 fractions), no overlap
    ("i16", 22050, 22050, 11025, 9.0, True),         # 6 x 3675: tasks {1,2} {packed}
    ("stereo", 22050, 22050, 5000, 4.0, False),
    ("f64", 22050, 22050, 22050, 7.0, False),
    ("unit", 44100, 44100, 22050, 4.0, False),       # a float signal in [-1, 1]
    ("i16", 32000, 44100, 22050, 5.0, False),        # the window at another sampling rate: other mel / chroma tables
    ("i16", 44100, 22050, 11025, 5.0, True),         # 0.5 s at 44.1 kHz
    ("i16", 48000, 48000, 24000, 7.0, True),         # 12 x 4000: sub-transforms of 8 x 20 x 25 points
    ("stereo", 48000, 48000, 48000, 5.0, False),
    ("f64", 48000, 48000, 20000, 4.0, False),
    ("i16", 32000, 32000, 16000, 7.0, True),         # 8 x 4000: tasks {1,2} {3,packed}
    ("stereo", 32000, 32000, 9000, 4.0, False),
    ("f64", 32000, 32000, 32000, 5.0, True),
    ("i16", 24000, 24000, 12000, 6.0, True),         # 6 x 4000
    ("f64", 48000, 24000, 12000, 3.0, False),
])
def test_full_matrix_against_oracle(gpu_lib, kind, fs, window, step, seconds, deltas):
    sig, mono = make_signal(kind, 8800 + window + step, seconds, fs)
    ref, _ = O.feature_extraction(mono, fs, window, step, deltas)
    got, names = ShortTermFeatures.feature_extraction(sig, fs, window, step, deltas)
    assert got.shape == ref.shape and len(names) == ref.shape[0]
    assert_parity(got, ref, "%s %d/%d@%d" % (kind, window, step, fs), sig=(mono, fs, window, step))
    if deltas:
        assert np.array_equal(got[34:, 1:], got[:34, 1:] - got[:34, :-1]) and np.all(got[34:, 0] == 0.0)
    again, _ = ShortTermFeatures.feature_extraction(sig, fs, window, step, deltas)
    assert np.array_equal(got, again)          # (the task counters reset themselves; the sums are formed in a fixed order)


@pytest.mark.parametrize("kind,fs,window,step", [("i16", 44100, 44100, 22050), ("stereo", 44100, 44100, 17000), ("f64", 22050, 22050, 11025),
                                                  ("stereo", 22050, 22050, 22050), ("i16", 48000, 48000, 24000), ("stereo", 32000, 32000, 16000),
                                                  ("f64", 24000, 24000, 7000)])
def test_spectrogram_and_chromagram_rows(gpu_lib, capsys, kind, fs, window, step):
    """Spectrogram rows go to the output in natural order straight from the transform kernel; chromagram rows come from the unit-major
    scratch rows; the chromagram's truncated tail frame (the reference FFTs what is left, :349-355) keeps its own kernel."""
    sig, mono = make_signal(kind, 9900 + step, 6.4, fs)
    spec, t_ax, f_ax = ShortTermFeatures.spectrogram(sig, fs, window, step)
    capsys.readouterr()
    ref, _, _ = O.spectrogram(mono, fs, window, step)
    assert spec.shape == ref.shape and len(f_ax) == window // 2 and len(t_ax) == ref.shape[0]
    assert_parity(np.ascontiguousarray(spec.T), np.ascontiguousarray(ref.T), "spectrogram %s %d/%d" % (kind, window, step))
    # chromagram: frames start at `window`, the last one is truncated (the reference FFTs what is left and fails when that is shorter than
    # num_fft, :349-355, :288) -- the package follows the oracle either way; then a clip cut so that the truncated tail is long enough
    for cut in (0, step // 3, (2 * step) // 3):
        sg, mn = sig[:len(mono) - cut], mono[:len(mono) - cut]
        try:
            cref, _, _ = O.chromagram(mn, fs, window, step)
        except (ValueError, IndexError) as exc:          # (a tail shorter than num_fft: ValueError; longer but below the chroma tables' reach: IndexError)
            with pytest.raises(type(exc)):
                ShortTermFeatures.chromagram(sg, fs, window, step)
            continue
        chroma, _, cnames = ShortTermFeatures.chromagram(sg, fs, window, step)
        assert chroma.shape == cref.shape and cnames == O.CHROMA_NAMES
        assert_parity(np.ascontiguousarray(chroma.T), np.ascontiguousarray(cref.T), "chromagram %s %d/%d cut %d" % (kind, window, step, cut))


def test_ragged_batch_equals_single_clips_and_silent_clips(gpu_lib):
    """One plan for a ragged batch (a clip of exactly one window, a digitally silent one, a constant one among them): bit-identical to the
    single-clip calls, and the silent frames keep their exact spectrum (assert_parity holds their MFCCs to the analytic vector)."""
    fs, W, S = 44100, 44100, 22050
    lens = [W, 3 * W + 17, 2 * W - 1, 5 * W, W + S, 4 * W]
    clips = [synth_clip(7100 + i, n, fs) for i, n in enumerate(lens)]
    clips[3] = np.zeros(lens[3], dtype=np.int16)
    clips[5] = clips[5].copy()
    clips[5][W:3 * W] = 1234                                 # two frames of a constant: zero after the mean only if the clip were constant --
    res, _ = ShortTermFeatures.feature_extraction_batch(clips, fs, W, S, deltas=True)       # here a DC step: every complex unit sees equal samples
    for c, r in zip(clips, res):
        single, _ = ShortTermFeatures.feature_extraction(c, fs, W, S)
        assert np.array_equal(single, r)
        ref, _ = O.feature_extraction(c, fs, W, S)
        assert_parity(r, ref, "ragged batch of 1 s windows", sig=(c, fs, W, S))
    fs, W, S = 22050, 22050, 11025
    clips = [make_signal("f64", 7200 + i, sec, fs)[0] for i, sec in enumerate([1.0, 3.7, 2.2])]
    res, _ = ShortTermFeatures.feature_extraction_batch(clips, fs, W, S, deltas=False)
    for c, r in zip(clips, res):
        single, _ = ShortTermFeatures.feature_extraction(c, fs, W, S, deltas=False)
        assert np.array_equal(single, r)
        ref, _ = O.feature_extraction(c, fs, W, S, deltas=False)
        assert_parity(r, ref, "float64 batch of 1 s windows", sig=(c, fs, W, S))


def test_more_frames_than_one_chunk_of_the_scratch(gpu_lib):
    """The spectrum scratch holds at most 1 GiB of rows (6 087 frames of 22 050 bins): a longer clip runs chunk after chunk, the second one
    starting with a halo frame.  The clip is periodic (period = 7 steps), so frames one period apart see the same samples and the same clip
    constants: every column equals the one a period before it, across the chunk boundary too; columns around the boundary also against the
    oracle."""
    fs, W, S = 44100, 44100, 22050
    rng = np.random.default_rng(4242)
    period = 7
    block = (rng.standard_normal(period * S) * 3000).astype(np.int16)
    n_rep = 900
    x = np.tile(block, n_rep)                                # 138.9 M samples: 6 298 frames
    F, _ = ShortTermFeatures.feature_extraction(x, fs, W, S, deltas=False)
    T = F.shape[1]
    assert T == (len(x) - W) // S + 1 and T > 6087 + 50
    assert np.array_equal(F[:, period:T - period], F[:, 2 * period:T])          # (column 0 has flux 0: the comparison starts one period in)
    xn = O.normalize_clip(x)
    tab = O.Tables(fs, W)
    for t in (6085, 6086, 6087, 6088):
        fr = xn[t * S:t * S + W]
        X = O.magnitude_spectrum(fr, tab.nfft)
        Xp = O.magnitude_spectrum(xn[(t - 1) * S:(t - 1) * S + W], tab.nfft)
        v = O.frame_vector(fr, X, Xp, tab)
        assert np.allclose(F[:34, t], v, rtol=1e-9, atol=1e-10)


@pytest.mark.parametrize("fs,window,step", [(22050, 22050, 11025), (44100, 44100, 22050), (32000, 32000, 16000)])
def test_many_short_clips_in_one_plan(gpu_lib, fs, window, step):
    """240 clips of one to five frames in one plan: more workgroups than tasks per XCD segment at the ends of the list, clip boundaries between
    almost every pair of frames (r0 = 6: the packed units of two frames share a task only inside a clip), a silent clip among them -- bit-identical
    to the single-clip calls, a sample of them against the oracle."""
    rng = np.random.default_rng(window)
    lens = [window + step * int(rng.integers(0, 5)) + int(rng.integers(0, step)) for _ in range(240)]
    clips = [synth_clip(7600 + i, n, fs) for i, n in enumerate(lens)]
    clips[17] = np.zeros(lens[17], dtype=np.int16)
    res, _ = ShortTermFeatures.feature_extraction_batch(clips, fs, window, step, deltas=True)
    assert len(res) == len(clips)
    for i, (c, r) in enumerate(zip(clips, res)):
        if i % 8 == 0 or i == 17:
            single, _ = ShortTermFeatures.feature_extraction(c, fs, window, step)
            assert np.array_equal(single, r), i
        if i % 40 == 0 or i == 17:
            ref, _ = O.feature_extraction(c, fs, window, step)
            assert_parity(r, ref, "clip %d of 240" % i, sig=(c, fs, window, step))
    again, _ = ShortTermFeatures.feature_extraction_batch(clips, fs, window, step, deltas=True)
    assert all(np.array_equal(a, b) for a, b in zip(res, again))
