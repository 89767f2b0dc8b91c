"""One-off robustness sweep on the GPU box (not part of the test suite): N random (fs, window, step, sample type, deltas)
shapes, whole feature matrix against the NumPy oracle with the gates of tests/test_parity_gpu.py; prints which kernel ran
each shape and every failure.    python scripts/random_sweep.py [N] [first seed]"""
import collections, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import paa_oracle as O
from pyaudioanalysis_amd import ShortTermFeatures, _ffi
from synth import synth_clip
from test_parity_gpu import assert_parity

n, first = int(sys.argv[1]) if len(sys.argv) > 1 else 200, int(sys.argv[2]) if len(sys.argv) > 2 else 0
_ffi.lib(); _ffi.init(0)
kernels, bad = collections.Counter(), []
for seed in range(first, first + n):
    rng = np.random.default_rng(50000 + seed)
    fs = int(rng.choice([8000, 11025, 12000, 16000, 22050, 24000, 32000, 44100, 48000, 96000]))
    if rng.random() < 0.5:          # the windows people use: 20 / 25 / 30 / 40 / 50 / 64 ms, powers of two
        window = int(rng.choice([int(fs * ms / 1000) for ms in (20, 25, 30, 40, 50, 64)] + [256, 512, 1024, 2048]))
    else:
        window = int(rng.integers(max(200, fs // 60), fs // 12))
    if window / 2 < 12 * np.log2((fs / 2) / 27.5) + 2:
        window = int(fs // 20)
    step = int(rng.choice([window // 4, window // 2, window, int(rng.integers(window // 5, window + window // 4))]))
    kind = int(rng.integers(0, 3))
    deltas = bool(rng.integers(0, 2))
    seconds = float(rng.choice([0.7, 1.3, 2.9]))
    xs = synth_clip(60000 + seed, int(fs * seconds) + window, fs=fs, stereo=True)
    mono = O.stereo_to_mono(xs)
    sig = xs[:, 0].copy() if kind == 0 else (mono if kind == 1 else xs)
    ref_in = sig if kind != 2 else mono
    try:
        ref, _ = O.feature_extraction(ref_in, fs, window, step, deltas)
    except (ValueError, IndexError) as exc:
        try:
            ShortTermFeatures.feature_extraction(sig, fs, window, step, deltas)
            bad.append((seed, fs, window, step, kind, "no %s raised" % type(exc).__name__))
        except type(exc):
            pass
        continue
    plan = _ffi.Plan(np.array([0, len(ref_in)], dtype=np.int64), fs, window, step, deltas=deltas, sample_kind=kind)
    kname = plan.kernel_name
    kernels[kname] += 1
    plan.destroy()
    try:
        got, _ = ShortTermFeatures.feature_extraction(sig, fs, window, step, deltas)
        assert_parity(got, ref, "seed %d" % seed, O.ill_conditioned_mfcc_frames(ref_in, fs, window, step))
    except Exception as exc:
        bad.append((seed, kname, fs, window, step, kind, repr(exc)[:160]))
    if seed % 3 == 0:               # spectrogram / chromagram rows of the same clip (modes 1 / 2 of the same kernels)
        import contextlib, io
        try:
            try:
                sref = O.spectrogram(ref_in, fs, window, step)[0]
                cref = O.chromagram(ref_in, fs, window, step)[0]
            except (ValueError, IndexError):
                continue
            with contextlib.redirect_stdout(io.StringIO()):
                sg = ShortTermFeatures.spectrogram(sig, fs, window, step)[0]
            cg = ShortTermFeatures.chromagram(sig, fs, window, step)[0]
            kernels["rows:" + kname] += 1
            assert_parity(np.ascontiguousarray(sg.T), np.ascontiguousarray(sref.T), "spectrogram seed %d" % seed)
            assert_parity(np.ascontiguousarray(cg.T), np.ascontiguousarray(cref.T), "chromagram seed %d" % seed)
        except Exception as exc:
            bad.append((seed, "rows:" + kname, fs, window, step, kind, repr(exc)[:160]))
print("kernels:", dict(kernels))
print("failures:", len(bad), dict(collections.Counter(b[1] for b in bad)))
for b in bad[:20]:
    print("  ", b)
