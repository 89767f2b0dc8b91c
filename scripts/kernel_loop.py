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

from bench import SHAPES as CASES, shape_input, shape_bytes_per_frame          # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", required=True, choices=sorted(CASES))
    ap.add_argument("--launches", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    args = ap.parse_args()
    fs, W, S, seconds, clips, kind, mode, deltas = CASES[args.case]
    lib = _ffi.lib()
    _ffi.init(0)
    x, offsets = shape_input(args.case)
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
    per_frame = shape_bytes_per_frame(args.case, plan.F if mode != 0 else (68 if deltas else 34))
    res = {"case": args.case, "kernel": plan.kernel_name, "frames": plan.total_frames, "launches": args.launches,
           "ms_per_step": ms.value / args.launches, "frames_per_s": plan.total_frames / (ms.value / args.launches * 1e-3),
           "algorithmic_bytes_per_frame": per_frame, "algorithmic_bytes_per_launch": per_frame * plan.total_frames,
           "sample_kind": kind, "mode": mode, "window": W, "step": S, "fs": fs}
    res["achieved_GBps"] = res["algorithmic_bytes_per_launch"] / (res["ms_per_step"] * 1e-3) / 1e9
    print(json.dumps(res))


if __name__ == "__main__":
    main()
