"""CPU tests of the host-side glue of audioSegmentation.silence_removal (row f4): smoothing, thresholds and the grouping
of frame indices into segments.  The GPU pieces (short-term features, per-frame SVM probability) are replaced by their
CPU checkers here -- the oracle's feature_extraction and scikit-learn's predict_proba -- so that the glue can be compared
with the UNMODIFIED reference running in the same process (skipped where /root/reference is absent)."""
import numpy as np
import pytest

import load_reference
import paa_oracle as O
from pyaudioanalysis_amd import audioSegmentation
from synth import synth_clip


def test_smooth_moving_avg_known_answers():
    # away from the ends a straight line is a fixed point of a centred box filter (odd widths); an even width
    # averages one more sample behind the point than ahead of it: half a step lower
    line = 0.25 * np.arange(40) - 3.0
    for w in (3, 4, 7, 10):
        want = line if w % 2 else line - 0.125
        got = audioSegmentation.smooth_moving_avg(line, w)
        assert np.allclose(got[w:-w], want[w:-w], rtol=0, atol=1e-12)       # (the mirrored ends repeat the end sample)
    v = np.sin(np.arange(200) * 0.1) + 0.1 * np.cos(np.arange(200) * 1.7)
    assert audioSegmentation.smooth_moving_avg(v, 2) is v
    # interior points: plain mean of `width` neighbours, the block ending (width - 1) // 2 samples after the point
    for w in (3, 10, 25):
        got = audioSegmentation.smooth_moving_avg(v, w)
        assert got.shape == v.shape
        for i in (40, 99, 150):
            hi = i + (w - 1) // 2
            assert abs(got[i] - v[hi - w + 1:hi + 1].mean()) < 1e-13
    with pytest.raises(ValueError):
        audioSegmentation.smooth_moving_avg(v[:5], 11)
    with pytest.raises(ValueError):
        audioSegmentation.smooth_moving_avg(v.reshape(2, 100), 5)


def test_onset_segments_known_answers():
    seg = audioSegmentation._onset_segments
    assert seg(np.array([], dtype=np.int64), 0.05) == []
    assert seg(np.array([7]), 0.05) == []                                   # one frame lasts 0 s
    # gaps of 1 and 2 frames join, a gap of 3 cuts; [3..12] lasts 0.45 s, [20..22] only 0.1 s, the lone 40 nothing
    got = seg(np.array([3, 4, 6, 8, 9, 10, 12, 20, 21, 22, 40]), 0.05)
    assert len(got) == 1 and got[0][0] == 3 * 0.05 and got[0][1] == 12 * 0.05
    got = seg(np.array([0, 1, 2, 3, 4, 5, 9, 10, 11, 12, 13, 14, 15]), 0.05)
    assert got == [[0 * 0.05, 5 * 0.05], [9 * 0.05, 15 * 0.05]]
    # strictly longer than 0.2 s (binary-exact step so that the comparison is not a rounding matter)
    assert seg(np.arange(8, 12), 0.0625) == []                               # 3 steps = 0.1875 s
    assert seg(np.arange(8, 13), 0.0625) == [[0.5, 0.75]]                    # 4 steps = 0.25 s
    assert seg(np.arange(8, 12), 0.0625, min_duration=0.1875) == []          # strict ">"


def test_onset_threshold_mixes_the_extreme_tenths():
    p = np.linspace(0.0, 1.0, 101)
    thr = audioSegmentation._onset_threshold(p, 0.25)
    assert abs(thr - (0.75 * p[:10].mean() + 0.25 * p[-10:].mean())) < 1e-15


@pytest.mark.skipif(not load_reference.reference_available(), reason="needs the reference tree")
@pytest.mark.parametrize("seed,st_win,st_step,smooth,weight", [(1, 0.050, 0.050, 0.5, 0.5), (2, 0.020, 0.020, 1.0, 0.3),
                                                              (3, 0.050, 0.025, 0.5, 0.7), (4, 0.050, 0.050, 0.3, 1.5)])
def test_glue_matches_live_reference(monkeypatch, seed, st_win, st_step, smooth, weight):
    pytest.importorskip("sklearn.svm")
    ref_seg = load_reference.load_segmentation()
    fs = 16000
    x = synth_clip(500 + seed, 6 * fs).copy()
    rng = np.random.default_rng(seed)
    for _ in range(4):                                                      # digital silence between bursts
        a = int(rng.integers(0, 5 * fs))
        x[a:a + int(rng.integers(fs // 4, fs))] = 0

    from pyaudioanalysis_amd import ShortTermFeatures as product_stf
    monkeypatch.setattr(product_stf, "feature_extraction",
                        lambda sig, rate, win, step, deltas=True: O.feature_extraction(sig, rate, win, step, deltas))

    def predict_loop(st_feats, mean, std, svm):
        return np.array([svm.predict_proba(((st_feats[:, i] - mean) / std).reshape(1, -1))[0][1]
                         for i in range(st_feats.shape[1])])
    monkeypatch.setattr(audioSegmentation, "svm_onset_probability", predict_loop)

    np.random.seed(99)
    want = ref_seg.silence_removal(x, fs, st_win, st_step, smooth, weight)
    np.random.seed(99)
    got = audioSegmentation.silence_removal(x, fs, st_win, st_step, smooth, weight)
    want = np.array(want, dtype=np.float64).reshape(-1, 2)
    got = np.array(got, dtype=np.float64).reshape(-1, 2)
    assert got.shape == want.shape and got.shape[0] >= 1, (got, want)
    assert np.allclose(got, want, rtol=0, atol=1e-9), (got, want)
