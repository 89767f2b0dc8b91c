"""Host-side logic of pyaudioanalysis_amd.audioSegmentation that needs no GPU: the diagonal growth of the
thumbnail (audioSegmentation.py:1167-1182), argument validation, and the no-CPU-fallback rule."""
import numpy as np
import pytest

import paa_oracle as O
from conftest import golden_files, golden_id, load_golden
from pyaudioanalysis_amd import _ffi, audioSegmentation


@pytest.mark.parametrize("path", golden_files("thumb"), ids=golden_id)
def test_grow_matches_reference_positions(path):
    """Starting from the arg-max of the reference's own filtered matrix, the growth loop must land on its limits."""
    g = load_golden(path)
    filt = g["filtered"]
    ss, th = float(g["short_step"]), float(g["thumb_size"])
    m = int(round(th / ss))
    r, c = np.unravel_index(filt.argmax(), filt.shape)
    i1, i2, j1, j2 = audioSegmentation._grow_thumbnail(filt, r, c, m)
    assert [ss * i1, ss * i2, ss * j1, ss * j2] == list(g["pos"])


def test_grow_matches_oracle_on_random_matrices():
    rng = np.random.default_rng(7)
    for n, m in ((30, 6), (64, 20), (9, 4), (5, 10)):
        a = rng.standard_normal((n, n))
        r, c = np.unravel_index(a.argmax(), a.shape)
        assert audioSegmentation._grow_thumbnail(a, r, c, m) == O.thumbnail_grow(a, m)


def test_thumbnail_rows_and_argument_checks():
    lib = _ffi.lib()
    assert int(lib.paa_thumbnail_rows(100, 20)) == 81
    assert int(lib.paa_thumbnail_rows(20, 20)) == 1
    assert int(lib.paa_thumbnail_rows(19, 20)) == 0
    assert int(lib.paa_thumbnail_rows(10, 0)) == 0
    with pytest.raises(ValueError):
        audioSegmentation.self_similarity_matrix(np.zeros((0, 4)))
    with pytest.raises(ValueError):                 # shorter than one 1-second window (ShortTermFeatures.py:684)
        audioSegmentation.music_thumbnailing(np.zeros(1000, dtype=np.int16), 16000)
    with pytest.raises(ValueError):                 # 3 feature vectors < 20-cell filter
        audioSegmentation.music_thumbnailing(np.zeros(2 * 16000, dtype=np.int16), 16000)


def test_no_cpu_fallback_for_similarity():
    if _ffi.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(_ffi.HipLibraryError):
        audioSegmentation.self_similarity_matrix(np.ones((4, 8)))
    with pytest.raises(_ffi.HipLibraryError):
        audioSegmentation.music_thumbnailing(np.zeros(30 * 16000, dtype=np.int16), 16000)
