"""Row f4 (remainder): the per-frame SVM probability loop of audioSegmentation.silence_removal (:744-748) as one kernel,
and the silence_removal drop-in against outputs of the unmodified reference (tests/golden/silence_*.npz).  -m gpu.
scikit-learn is the reference's own dependency for this function (it trains the SVM at :739) and is used here the
same way: to train, and -- in the test -- as the oracle of predict_proba."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR
from pyaudioanalysis_amd import ShortTermFeatures, audioSegmentation
from synth import synth_clip

pytestmark = pytest.mark.gpu
sklearn_svm = pytest.importorskip("sklearn.svm")


@pytest.mark.parametrize("kernel", ["linear", "rbf"])
def test_onset_probability_equals_predict_proba(gpu_lib, kernel):
    x = synth_clip(321, 6 * 16000).copy()
    x[16000:30000] = 0
    st, _ = ShortTermFeatures.feature_extraction(x, 16000, 800, 800)
    energy = st[1]
    lo, hi = st[:, energy <= np.quantile(energy, 0.2)].T, st[:, energy >= np.quantile(energy, 0.8)].T
    feats = np.vstack([lo, hi])
    labels = np.append(np.zeros(len(lo)), np.ones(len(hi)))
    mean, std = feats.mean(axis=0), feats.std(axis=0)
    std[std == 0] = 1.0
    svm = sklearn_svm.SVC(C=1.0, kernel=kernel, probability=True, gamma='auto', random_state=7).fit((feats - mean) / std, labels)
    got = audioSegmentation.svm_onset_probability(st, mean, std, svm)
    ref = np.array([svm.predict_proba(((st[:, i] - mean) / std).reshape(1, -1))[0][1] for i in range(st.shape[1])])
    assert got.shape == ref.shape
    assert np.max(np.abs(got - ref)) < 1e-10, np.max(np.abs(got - ref))


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN_DIR, "silence_*.npz"))),
                         ids=lambda p: os.path.splitext(os.path.basename(p))[0])
def test_silence_removal_matches_reference(gpu_lib, path):
    with np.load(path, allow_pickle=False) as z:
        g = {k: z[k] for k in z.files}
    np.random.seed(int(g["seed"]))          # the SVM's probability calibration draws from NumPy's global state
    segs = audioSegmentation.silence_removal(g["signal"], int(g["fs"]), float(g["st_win"]), float(g["st_step"]),
                                             float(g["smooth_window"]), float(g["weight"]))
    got = np.array(segs, dtype=np.float64).reshape(-1, 2)
    ref = g["segments"]
    assert got.shape == ref.shape, (got, ref)
    assert np.allclose(got, ref, rtol=0, atol=1e-9), (got, ref)


def test_smooth_moving_avg_matches_reference_formula():
    v = np.sin(np.arange(200) * 0.1) + 0.1 * np.cos(np.arange(200) * 1.7)
    for w in (2, 3, 10, 25):
        got = audioSegmentation.smooth_moving_avg(v, w)
        if w < 3:
            assert got is v
            continue
        s = np.r_[2 * v[0] - v[w - 1::-1], v, 2 * v[-1] - v[-1:-w:-1]]
        ref = np.convolve(np.ones(w) / w, s, mode='same')[w:-w + 1]
        assert np.array_equal(got, ref)
    with pytest.raises(ValueError):
        audioSegmentation.smooth_moving_avg(v[:5], 11)
