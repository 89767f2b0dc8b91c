"""How close is the HIP path to the reference goldens / the oracle really?  Prints, per case, the worst
|got - ref| / (|ref| + row scale) over well-conditioned frames (continuous rows), the same over the frames
paa_oracle.ill_conditioned_mfcc_frames flags, and the number of discrete-row (ZCR / roll-off) flips.
Run on the GPU box; calibrates the tight gate of tests/test_parity_gpu.py."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import paa_oracle as O
from conftest import golden_files, golden_id, load_golden
from pyaudioanalysis_amd import ShortTermFeatures, _ffi
from synth import synth_clip

DISCRETE = (0, 7, 34, 41)


def margin(got, ref, ill=None):
    scale = np.max(np.abs(ref), axis=1, keepdims=True)
    if ref.shape[0] in (34, 68):
        for blk in range(ref.shape[0] // 34):
            rows = [blk * 34 + r for r in O.MFCC_ROWS]
            scale[rows] = scale[rows].max()
    e = np.abs(got - ref) / (np.abs(ref) + scale + 1e-300)
    cont = np.ones(ref.shape[0], bool)
    flips = 0
    if ref.shape[0] in (34, 68):
        for r in DISCRETE:
            if r < ref.shape[0]:
                cont[r] = False
                flips += int(np.sum(got[r] != ref[r]) if r in (0, 7) else 0)
    ill = np.zeros(ref.shape[1], bool) if ill is None else ill
    well = e[cont][:, ~ill]
    bad = e[cont][:, ill]
    w = float(well.max()) if well.size else 0.0
    wi = np.unravel_index(np.argmax(e * cont[:, None] * (~ill)[None, :]), e.shape) if well.size else (0, 0)
    return w, (float(bad.max()) if bad.size else 0.0), flips, int(ill.sum()), wi


_ffi.lib(); _ffi.init(0)
rows = []
for path in golden_files("st"):
    g = load_golden(path)
    F, _ = ShortTermFeatures.feature_extraction(g["signal"], g["fs"], g["window"], g["step"], g["deltas"])
    ill = O.ill_conditioned_mfcc_frames(g["signal"], g["fs"], g["window"], g["step"])
    rows.append((golden_id(path),) + margin(F, g["features"], ill))
for fs, window, step, seconds in [(16000, 800, 400, 4.0), (16000, 800, 800, 2.0), (16000, 400, 160, 1.0), (16000, 801, 401, 1.0),
                                  (16000, 1024, 512, 1.5), (22050, 1103, 441, 1.0), (44100, 1102, 441, 1.0), (8000, 400, 200, 1.5),
                                  (8000, 800, 400, 2.0), (7000, 800, 400, 2.0), (44100, 800, 400, 1.0), (22050, 800, 800, 1.0),
                                  (48000, 2400, 1200, 1.0), (16000, 640, 640, 2.0), (16000, 320, 160, 1.0)]:
    x = synth_clip(100 + window, int(seconds * fs), fs=fs)
    ref, _ = O.feature_extraction(x, fs, window, step, True)
    got, _ = ShortTermFeatures.feature_extraction(x, fs, window, step, True)
    ill = O.ill_conditioned_mfcc_frames(x, fs, window, step)
    rows.append(("seeded %d/%d@%d" % (window, step, fs),) + margin(got, ref, ill))
for path in golden_files("spec"):
    g = load_golden(path)
    import io, contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        S, _, _ = ShortTermFeatures.spectrogram(g["signal"], g["fs"], g["window"], g["step"])
    C, _, _ = ShortTermFeatures.chromagram(g["signal"], g["fs"], g["window"], g["step"])
    rows.append((golden_id(path) + " spec",) + margin(S.T, g["specgram"].T))
    rows.append((golden_id(path) + " chroma",) + margin(C.T, g["chromagram"].T))
print("%-44s %12s %12s %6s %6s  worst(row,frame)" % ("case", "well-cond", "ill-cond", "flips", "n_ill"))
for r in sorted(rows, key=lambda r: -r[1]):
    print("%-44s %12.3e %12.3e %6d %6d  %s" % (r[0], r[1], r[2], r[3], r[4], tuple(int(v) for v in r[5])))
