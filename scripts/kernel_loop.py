#!/usr/bin/env python
"""One plan of a named case, executed N times with samples and results resident in HBM: the workload that
scripts/profile_kernel.sh puts under rocprofv3 (kernel trace / counter passes) for the NON-headline kernels.
Prints one JSON line: kernel name, frames, HIP-event time per launch, SURVEY 8d's algorithmic bytes per frame.

    python scripts/kernel_loop.py --case reg_spectrogram --launches 100
"""
import argparse
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)
from pyaudioanalysis_amd import _ffi          # noqa: E402
from synth import synth_clip                  # noqa: E402

# name: (fs, window, step, seconds of one clip, clips, sample kind [0 int16, 1 f64, 2 int32 sums], mode, deltas)
CASES = {
    "reg_features": (44100, 1102, 441, 600, 1, 0, 0, 0),
    "reg_features_stereo": (44100, 1102, 441, 600, 1, 2, 0, 0),
    "reg_spectrogram": (44100, 1102, 441, 600, 1, 0, 1, 0),
    "reg_spectrogram_stereo": (44100, 1102, 441, 600, 1, 2, 1, 0),
    "reg_chromagram": (44100, 1102, 441, 600, 1, 0, 2, 0),
    "ct_640": (16000, 640, 640, 3600, 1, 0, 0, 0),
    "ct_640_spectrogram": (16000, 640, 640, 3600, 1, 0, 1, 0),
    "ct_800_f64": (16000, 800, 400, 3600, 1, 1, 0, 0),
    "ct_800_stereo": (16000, 800, 400, 3600, 1, 2, 0, 0),
    "ct_400": (8000, 400, 200, 3600, 2, 0, 0, 0),
    "ct_320": (16000, 320, 160, 1800, 1, 0, 0, 0),
    "generic_2400": (48000, 2400, 1200, 1200, 1, 0, 0, 0),
    "generic_2205": (44100, 2205, 1102, 1200, 1, 0, 0, 0),
    "mid_stats": (16000, 800, 400, 30, 1000, 0, 0, 1),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", required=True, choices=sorted(CASES))
    ap.add_argument("--launches", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    args = ap.parse_args()
    fs, W, S, seconds, clips, kind, mode, deltas = CASES[args.case]
    lib = _ffi.lib()
    _ffi.init(0)
    base_s = min(seconds, 100)                       # synthesise at most 100 s, tile the rest
    reps = -(-seconds // base_s)
    n = base_s * fs
    if kind == 0:
        x = np.tile(synth_clip(5, n, fs), reps)
    else:
        xs = synth_clip(5, n, fs, stereo=True)
        x = np.tile((xs[:, 1] / 2) + (xs[:, 0] / 2), reps) if kind == 1 else np.tile(xs[:, 0].astype(np.int32) + xs[:, 1], reps)
    x = np.ascontiguousarray(np.tile(x, clips))
    per = len(x) // clips
    offsets = np.arange(clips + 1, dtype=np.int64) * per
    d_in = _ffi.DeviceBuffer.from_host(x)
    plan = _ffi.Plan(offsets, fs, W, S, deltas=bool(deltas), sample_kind=kind, mode=mode)
    d_out = _ffi.DeviceBuffer(plan.out_doubles * 8)
    d_mid = None
    if args.case == "mid_stats":
        d_mid = _ffi.DeviceBuffer(plan.mid_doubles(40) * 8)

    def step():
        plan.execute(d_in, d_out)
        if d_mid is not None:
            plan.mid_execute(d_out, 39, 40, d_mid)
    for _ in range(args.warmup):
        step()
    _ffi.sync()
    _ffi.check(lib.paa_timer_start())
    for _ in range(args.launches):
        step()
    ms = ctypes.c_float()
    _ffi.check(lib.paa_timer_stop(ctypes.byref(ms)))
    in_bytes = {0: 2, 1: 8, 2: 4}[kind] * S
    out_bytes = 8 * (plan.F if mode != 0 else (68 if deltas else 34))
    per_frame = in_bytes + out_bytes
    res = {"case": args.case, "kernel": plan.kernel_name, "frames": plan.total_frames, "launches": args.launches,
           "ms_per_step": ms.value / args.launches, "frames_per_s": plan.total_frames / (ms.value / args.launches * 1e-3),
           "algorithmic_bytes_per_frame": per_frame, "algorithmic_bytes_per_launch": per_frame * plan.total_frames,
           "sample_kind": kind, "mode": mode, "window": W, "step": S, "fs": fs}
    res["achieved_GBps"] = res["algorithmic_bytes_per_launch"] / (res["ms_per_step"] * 1e-3) / 1e9
    print(json.dumps(res))


if __name__ == "__main__":
    main()
