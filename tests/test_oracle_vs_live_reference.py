"""The oracle against the UNMODIFIED reference running in this container, on seeded random shapes the committed goldens
do not cover (the goldens pin fixed files; this sweeps sampling rates, windows, steps and clip lengths).  CPU only, and
only where /root/reference exists (the build container): skipped on the GPU box, where nothing may read it."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import load_reference            # noqa: E402
import paa_oracle as O           # noqa: E402
from synth import synth_clip     # noqa: E402

pytestmark = pytest.mark.skipif(not load_reference.reference_available(), reason="reference tree not present")


def _cases():
    rng = np.random.default_rng(20260921)
    cases = []
    for k in range(10):
        fs = int(rng.choice([8000, 11025, 16000, 22050, 44100]))
        window = int(rng.integers(int(0.010 * fs), int(0.060 * fs)))
        step = int(rng.integers(max(1, window // 4), window + 1))
        n = int(rng.integers(window, 12 * window))
        cases.append((k, fs, window, step, n))
    cases += [(100, 16000, 800, 400, 800), (101, 16000, 800, 800, 2399), (102, 44100, 1102, 441, 6000)]
    return cases


@pytest.mark.parametrize("seed,fs,window,step,n", _cases())
def test_feature_extraction_and_rows_match_the_running_reference(seed, fs, window, step, n, capsys):
    ref_st, ref_mt, _ = load_reference.load()
    x = synth_clip(7000 + seed, n, fs)
    # (tiny windows at low sampling rates make the reference's own chroma / mel set-up raise: same exception type)
    try:
        ref, ref_names = ref_st.feature_extraction(x.astype(np.float64), fs, window, step)
    except Exception as exc:
        with pytest.raises(type(exc)):
            O.feature_extraction(x, fs, window, step)
        return
    got, names = O.feature_extraction(x, fs, window, step)
    assert list(names) == list(ref_names) and got.shape == ref.shape
    assert O.mixed_tolerance_violations(got, ref, rel=1e-9, row_abs=1e-9, abs_floor=1e-12)[0] == 0
    spec, _, _ = ref_st.spectrogram(x.astype(np.float64), fs, window, step)
    capsys.readouterr()
    got_spec = O.spectrogram(x, fs, window, step)[0]
    assert got_spec.shape == spec.shape and np.max(np.abs(got_spec - spec)) <= 1e-12 * max(1.0, np.max(np.abs(spec)))
    if got.shape[1] >= 4:
        mid_w, mid_s = 3 * step + window, 2 * step           # a few short-term frames per mid-term window
        rmid, rst, rnames = ref_mt.mid_feature_extraction(x.astype(np.float64), fs, mid_w, mid_s, window, step)
        gmid, gst, gnames = O.mid_feature_extraction(x, fs, mid_w, mid_s, window, step)
        assert list(gnames) == list(rnames) and gmid.shape == rmid.shape
        assert O.mixed_tolerance_violations(gmid, rmid, rel=1e-9, row_abs=1e-9, abs_floor=1e-12)[0] == 0


@pytest.mark.parametrize("fs,window,step,n", [(16000, 800, 400, 800 + 400 * 7 + 150), (16000, 800, 400, 800 + 400 * 7 + 390), (16000, 800, 300, 4000),
                                              (44100, 44100, 17000, 282240), (44100, 44100, 17000, 276574), (22050, 1103, 441, 9000)])
def test_chromagram_truncated_tail_matches_the_running_reference(fs, window, step, n):
    """chromagram (:324-386) FFTs what is left of a truncated last frame: a tail of at least num_fft samples gives a row, a shorter one fails
    in the reference's scatter (:288-293) -- the oracle returns the same rows or raises the same exception TYPE (round 6: it raised IndexError
    where the reference raises ValueError; found by tests/test_wgs_kernel_gpu.py, whose kernels had it right)."""
    ref_st, _, _ = load_reference.load()
    x = synth_clip(7300 + step, n, fs)
    try:
        ref, _, _ = ref_st.chromagram(x.astype(np.float64), fs, window, step)
    except Exception as exc:
        with pytest.raises(type(exc)):
            O.chromagram(x, fs, window, step)
        return
    got, _, _ = O.chromagram(x, fs, window, step)
    assert got.shape == ref.shape and np.max(np.abs(got - ref)) <= 1e-12
