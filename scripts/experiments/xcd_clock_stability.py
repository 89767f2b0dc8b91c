#!/usr/bin/env python
"""How stable is the XCD-to-XCD clock spread under the hot kernel?  (round 5: the eight XCDs of an MI355X ran the same 578 k-cycle
waves at 2.13-2.27 GHz, scripts/rounds/r05/gpu_r05aa.sh.)  Needs the -DPAA_F800_TRACE build (PAA_HIP_LIBRARY): per launch the
per-XCD median clock (wave cycles / wave life) and the time at which the XCD's last workgroup ended; launches sampled back to
back, every 100 launches, and after idle gaps.

    PAA_HIP_LIBRARY=pyaudioanalysis_amd/libpaa_hip_trace.so python scripts/experiments/xcd_clock_stability.py
"""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)
from pyaudioanalysis_amd import _ffi          # noqa: E402
from synth import synth_clip                  # noqa: E402


def main():
    lib = _ffi.lib()
    _ffi.init(0)
    x = synth_clip(2, 3600 * 16000)
    d_in = _ffi.DeviceBuffer.from_host(x)
    plan = _ffi.Plan(np.array([0, len(x)], dtype=np.int64), 16000, 800, 400, deltas=False)
    d_out = _ffi.DeviceBuffer(plan.out_doubles * 8)
    tr = (ctypes.c_uint64 * (4096 * 4))()

    def sample(tag):
        plan.execute(d_in, d_out)
        _ffi.sync()
        n = lib.paa_debug_wave_trace(tr, 4096)
        t = np.array(list(tr), dtype=np.uint64).reshape(-1, 4)[:2048]
        t0, t1, cyc, hw = t[:, 0].astype(np.float64), t[:, 1].astype(np.float64), t[:, 2].astype(np.float64), t[:, 3]
        life = (t1 - t0) / 100.0
        xcc = ((hw >> np.uint64(32)) & np.uint64(15)).astype(int)
        end = (t1 - t0.min()) / 100.0
        clk = [float(np.median(cyc[xcc == q] / (life[xcc == q] * 1e3))) for q in range(8)]
        ends = [float(end[xcc == q].max()) for q in range(8)]
        wg_xcc = xcc.reshape(-1, 8)[:, 0]
        print(tag, "clock GHz", " ".join("%.3f" % c for c in clk), "| end us", " ".join("%.0f" % e for e in ends),
              "| wg0->xcc %d, per-xcc workgroups %s" % (wg_xcc[0], np.bincount(wg_xcc, minlength=8)), flush=True)

    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < 1.0:
        for _ in range(20):
            plan.execute(d_in, d_out)
        _ffi.sync()
    for k in range(4):
        sample("back-to-back %d" % k)
    for k in range(4):
        for _ in range(100):
            plan.execute(d_in, d_out)
        sample("after 100 more %d" % k)
    for gap in (0.05, 0.5, 2.0):
        time.sleep(gap)
        sample("after %.2f s idle   " % gap)
        sample("  and the next one  ")
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < 3.0:
        for _ in range(20):
            plan.execute(d_in, d_out)
        _ffi.sync()
    for k in range(3):
        sample("after 3 s of load %d" % k)


if __name__ == "__main__":
    main()
