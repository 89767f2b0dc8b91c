#!/usr/bin/env python
"""The small kernels no bench shape runs on its own -- beat_kernel, svm_binary_proba_kernel, expand_deltas_kernel,
chroma_tail_kernel, the delta kernels of the big-window paths and kernels_big.hpp's radix passes through HBM -- each called a
few times through the product's own entry points, so that ONE rocprofv3 kernel trace lists every kernel the library ships
(VERDICT r04 item 6).  Prints one JSON line with what was run.

    rocprofv3 --kernel-trace --stats -d out -o trace -- python scripts/aux_kernels_loop.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)
from pyaudioanalysis_amd import ShortTermFeatures, MidTermFeatures, audioSegmentation, _ffi          # noqa: E402
from synth import synth_clip                                                                          # noqa: E402


def main():
    lib = _ffi.lib()
    _ffi.init(0)
    reps = 5
    fs = 16000
    x = synth_clip(11, 120 * fs)
    ran = {}
    F, _ = ShortTermFeatures.feature_extraction(x, fs, 800, 400, deltas=False)
    for _ in range(reps):
        bpm, ratio = MidTermFeatures.beat_extraction(F, 0.025)                      # beat_kernel (MidTermFeatures.py:18-84)
    ran["beat_extraction"] = {"frames": int(F.shape[1]), "bpm": float(bpm)}
    # silence_removal (audioSegmentation.py:672-815): a clip with silent gaps; sklearn trains, the per-frame loop is svm_binary_proba_kernel
    y = x[:30 * fs].copy()
    y[5 * fs:8 * fs] //= 200
    y[15 * fs:19 * fs] //= 200
    for _ in range(reps):
        seg = audioSegmentation.silence_removal(y, fs, 0.05, 0.05, 0.5, 0.3)
    ran["silence_removal"] = {"segments": len(seg)}
    # expand_deltas_kernel: base rows of a ragged batch -> 68-row slabs on the device
    lens = [800, 1200, 16000, 400 * 5000 + 800, 2400]
    clips = [synth_clip(8100 + i, n, fs) for i, n in enumerate(lens)]
    offs = np.concatenate(([0], np.cumsum(lens))).astype(np.int64)
    d_in = _ffi.DeviceBuffer.from_host(np.concatenate(clips))
    p34 = _ffi.Plan(offs, fs, 800, 400, deltas=False)
    d34 = _ffi.DeviceBuffer(p34.out_doubles * 8)
    d68 = _ffi.DeviceBuffer(2 * p34.out_doubles * 8)
    frames = np.array([(n - 800) // 400 + 1 for n in lens], dtype=np.int64)
    for _ in range(reps):
        p34.execute(d_in, d34)
        _ffi.check(lib.paa_dev_expand_deltas(d34.ptr, _ffi.as_i64p(frames), len(frames), d68.ptr))
    _ffi.sync()
    p34.destroy()
    ran["expand_deltas"] = {"clips": len(lens), "frames": int(frames.sum())}
    # chroma_tail_kernel: chromagram() FFTs the truncated last frames (ShortTermFeatures.py:349-355)
    for _ in range(reps):
        C, _, _ = ShortTermFeatures.chromagram(x[:5 * fs + 777], fs, 800, 300)
    ran["chromagram_tail"] = {"rows": int(C.shape[0])}
    # the delta kernels of the big-window paths, and the HBM radix passes (a window of 80 000 samples: 40 000 complex points)
    for _ in range(reps):
        A, _ = ShortTermFeatures.feature_extraction(x[:40 * fs], fs, 16000, 8000, deltas=True)      # wg_delta_kernel
        B, _ = ShortTermFeatures.feature_extraction(x[:40 * fs], fs, 80000, 40000, deltas=True)     # big_* kernels
    ran["big_windows"] = {"frames_16000": int(A.shape[1]), "frames_80000": int(B.shape[1])}
    print(json.dumps({"case": "aux_kernels", "repetitions": reps, "ran": ran}))


if __name__ == "__main__":
    main()
