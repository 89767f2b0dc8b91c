#!/usr/bin/env python
"""Calibrate oracle/paa_oracle.py::feature_extraction_reference_cost against the UNMODIFIED reference, in the build container
(the only place /root/reference exists): same clip, same process, alternating runs, one thread.  Writes
profiles/r06_reference_cost_port.json -- bench.py's cpu_baseline.reference_cost_port quotes the ratio recorded here beside the
cost port's rate on the GPU box's host.

    OMP_NUM_THREADS=1 python scripts/reference_cost_calibration.py

TEST / BENCH INFRASTRUCTURE: nothing in the product package imports this."""
import json
import os
import platform
import sys
import time

os.environ.setdefault("OMP_NUM_THREADS", "1")
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import scipy

import cpu_bench
import load_reference
import paa_oracle as O
from synth import synth_clip


def main():
    ref_st, _, _ = load_reference.load()
    fs = 16000
    x = synth_clip(2, 60 * fs, fs)
    rounds = []
    for _ in range(3):
        t0 = time.perf_counter()
        F_ref, _ = ref_st.feature_extraction(x, fs, 800, 400, deltas=False)
        t_ref = time.perf_counter() - t0
        t0 = time.perf_counter()
        F_port, _ = O.feature_extraction_reference_cost(x, fs, 800, 400, deltas=False)
        t_port = time.perf_counter() - t0
        rounds.append((t_ref, t_port))
    t_ref = min(r[0] for r in rounds)
    t_port = min(r[1] for r in rounds)
    T = F_ref.shape[1]
    rec = {"what": "unmodified reference vs oracle/paa_oracle.py::feature_extraction_reference_cost, first 60 s of the seed-2 clip, "
                   "16 kHz, 800/400, deltas off, one thread, best of 3 alternating runs",
           "where": "build container", "cpu_model": cpu_bench.cpu_model(), "python": platform.python_version(),
           "numpy": np.__version__, "scipy": scipy.__version__, "frames": int(T),
           "reference_frames_per_s": T / t_ref, "cost_port_frames_per_s": T / t_port,
           "cost_port_over_reference": (T / t_port) / (T / t_ref),
           "max_abs_diff_port_vs_reference": float(np.max(np.abs(F_ref - F_port))),
           "rounds_seconds_reference_port": rounds}
    out = os.path.join(ROOT, "profiles", "r06_reference_cost_port.json")
    json.dump(rec, open(out, "w"), indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
