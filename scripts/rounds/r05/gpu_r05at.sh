#!/bin/bash
# round 5, GPU pass at: 16 x 16 x 4 (2048 samples) with the two self-paired pass-3 jobs folded into lane 0 (128 jobs: two rounds instead
# of three) and sixteen-wave row instances: parity of every three-pass test, loops of the 2048 shapes
out=gpurun_out/r05at; mkdir -p $out
(timeout 600 python -m pytest tests/test_mix_kernel_gpu.py tests/test_parity_gpu.py -m gpu -q --no-header --maxfail=20 2>&1 | tail -15) > $out/tests.log
grep -n "passed\|failed" $out/tests.log | tail -3; grep -n "^FAILED" $out/tests.log | head
timeout 200 python scripts/kernel_loop.py --case w2048 --launches 60 | cut -c1-200
python - <<'PY'
import sys, os, ctypes, numpy as np
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "oracle")]
from pyaudioanalysis_amd import _ffi
from synth import synth_clip
lib = _ffi.lib(); _ffi.init(0)
x = synth_clip(5, 600 * 44100, 44100)
d_in = _ffi.DeviceBuffer.from_host(x)
for mode in (1, 2):
    plan = _ffi.Plan(np.array([0, len(x)], dtype=np.int64), 44100, 2048, 1024, deltas=False, sample_kind=0, mode=mode)
    d_out = _ffi.DeviceBuffer(plan.out_doubles * 8)
    for _ in range(20): plan.execute(d_in, d_out)
    _ffi.sync(); _ffi.check(lib.paa_timer_start())
    for _ in range(60): plan.execute(d_in, d_out)
    ms = ctypes.c_float(); _ffi.check(lib.paa_timer_stop(ctypes.byref(ms)))
    print(plan.kernel_name, plan.total_frames, "frames", "%.4f ms per step" % (ms.value / 60), "%.3g frames/s" % (plan.total_frames / (ms.value / 60 * 1e-3)))
    plan.destroy()
PY
