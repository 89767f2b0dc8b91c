#!/usr/bin/env python
"""Launch geometry of the hot kernel (st_fast_800, one wave per run of a multiple of four frames, eight runs per workgroup): a
one-hour clip is 143 999 frames = 2000 runs of 72 = 250 workgroups on 256 CUs.  How much would full occupancy of the last six CUs
be worth?  Times the same plan on clips of 2000 x 72, 2048 x 72 (256 workgroups, 2.4 % more frames) and 2048 x 68 frames with the
profiling events around the feature kernel (paa_prof_*), statistics pass excluded.

    python scripts/experiments/run_geometry.py [frames ...]

Round 5, pass u (equal runs): 2000 x 72 frames 0.2752 ms, 2048 x 72 frames 0.2747 ms, 2048 x 68 frames 0.2554 ms -> the plan now re-cuts
a one-round clip into num_cu x 8 runs of two lengths (lib_plan.hpp: balanced_runs); PAA_HIP_LIBRARY selects the A/B build.
"""
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)
from pyaudioanalysis_amd import _ffi          # noqa: E402
import synth                                   # noqa: E402


def main():
    lib = _ffi.lib()
    _ffi.init(0)
    base = synth.synth_clip(2, 3700 * 16000)
    out = []
    sizes = [int(a) for a in sys.argv[1:]] or [2000 * 72 - 1, 2000 * 72, 2048 * 72, 2048 * 68, 2048 * 72 + 1]
    for frames in sizes:
        n = (frames - 1) * 400 + 800
        x = base[:n]
        d_in = _ffi.DeviceBuffer.from_host(x)
        plan = _ffi.Plan(np.array([0, n], dtype=np.int64), 16000, 800, 400, deltas=False, sample_kind=0)
        d_out = _ffi.DeviceBuffer(plan.out_doubles * 8)
        for _ in range(60):
            plan.execute(d_in, d_out)
        _ffi.sync()
        _ffi.check(lib.paa_prof_enable(1))
        for _ in range(40):
            plan.execute(d_in, d_out)
        _ffi.sync()
        ms, cnt = ctypes.c_double(), ctypes.c_int64()
        _ffi.check(lib.paa_prof_read(ctypes.byref(ms), ctypes.byref(cnt)))
        _ffi.check(lib.paa_prof_enable(0))
        k_ms = ms.value / max(cnt.value, 1)
        out.append({"frames": plan.total_frames, "kernel": plan.kernel_name, "kernel_ms": k_ms, "pairs": cnt.value,
                    "frames_per_s_kernel": plan.total_frames / (k_ms * 1e-3)})
        plan.destroy()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
