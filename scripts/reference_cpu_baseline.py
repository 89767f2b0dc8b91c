#!/usr/bin/env python
"""Time the UNMODIFIED reference (tyiannak/pyAudioAnalysis under /root/reference, imported through
oracle/load_reference.py) on SURVEY 8d's seeded inputs, single process / single thread, in the BUILD CONTAINER, and
write profiles/reference_cpu_r03.json.  bench.py embeds that file under cpu_baseline.reference, labelled as measured in
the build container (the GPU box has no /root/reference, so the reference itself cannot be timed there; the ports in
oracle/ are timed on the GPU box's host instead).

    OMP_NUM_THREADS=1 python scripts/reference_cpu_baseline.py [--out profiles/reference_cpu_r03.json]

TEST / BENCH INFRASTRUCTURE: nothing in the product package imports this."""
import argparse
import contextlib
import io
import json
import os
import platform
import sys
import time

os.environ.setdefault("OMP_NUM_THREADS", "1")
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
os.environ.setdefault("MKL_NUM_THREADS", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import scipy

import cpu_bench
import load_reference
from synth import synth_clip


def timed(fn, repeat=1):
    best = None
    out = None
    for _ in range(repeat):
        t0 = time.perf_counter()
        out = fn()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return best, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "reference_cpu_r03.json"))
    ap.add_argument("--cfg34-clips", type=int, default=8, help="clips of the config 3 / 4 samples")
    args = ap.parse_args()
    ref_st, ref_mt, ref_io = load_reference.load()
    res = {"what": "unmodified reference (pyAudioAnalysis 0.3.14 sources under /root/reference), one process, one thread",
           "where": "build container (NOT the GPU box's host)", "cpu_model": cpu_bench.cpu_model(),
           "usable_cores": cpu_bench.usable_cores(), "python": platform.python_version(), "numpy": np.__version__,
           "scipy": scipy.__version__, "omp_num_threads": os.environ.get("OMP_NUM_THREADS"), "entries": {}}
    E = res["entries"]

    def entry(name, seconds, frames, workload, **kw):
        E[name] = dict(seconds=seconds, frames=int(frames), frames_per_s=frames / seconds, workload=workload, **kw)
        print("%-22s %8.2f s  %9d frames  %10.1f frames/s" % (name, seconds, frames, frames / seconds), flush=True)

    fs = 16000
    # config 2: first 60 s of the 1-hour clip (seed 2), 800/400, 34 rows
    x = synth_clip(2, 60 * fs, fs)
    dt, (F, _) = timed(lambda: ref_st.feature_extraction(x, fs, 800, 400, deltas=False))
    entry("cfg2_60s_34rows", dt, F.shape[1], "first 60 s of seed-2 clip, 16 kHz, window 800 / step 400, deltas off")
    dt, (F, _) = timed(lambda: ref_st.feature_extraction(x, fs, 800, 400, deltas=True))
    entry("cfg2_60s_68rows", dt, F.shape[1], "same, deltas on (the reference's default)")
    # config 3 sample: 30 s clips, mid-term 1.0 s / 1.0 s over 50 ms / 25 ms
    clips = [synth_clip(3000 + i, 30 * fs, fs) for i in range(args.cfg34_clips)]
    dt, outs = timed(lambda: [ref_mt.mid_feature_extraction(c, fs, fs, fs, 800, 400) for c in clips])
    entry("cfg3_sample", dt, sum(o[1].shape[1] for o in outs),
          "%d clips x 30 s (seeds 3000..), mid_feature_extraction 1.0 s / 1.0 s over 800 / 400" % len(clips),
          clips=len(clips), clips_per_s=len(clips) / dt)
    # config 4 sample: 10 s clips, 34 rows
    clips = [synth_clip(40000 + i, 10 * fs, fs) for i in range(args.cfg34_clips)]
    dt, outs = timed(lambda: [ref_st.feature_extraction(c, fs, 800, 400, deltas=False) for c in clips])
    entry("cfg4_sample", dt, sum(o[0].shape[1] for o in outs),
          "%d clips x 10 s (seeds 40000..), window 800 / step 400, deltas off" % len(clips), clips=len(clips),
          clips_per_s=len(clips) / dt)
    # config 5: 20 s of 44.1 kHz stereo -> mono (the reference's stereo_to_mono), 1102 / 441
    fs5 = 44100
    xs = synth_clip(5, 20 * fs5, fs=fs5, stereo=True)
    dt_mono, mono = timed(lambda: ref_io.stereo_to_mono(xs))
    dt, (F, _) = timed(lambda: ref_st.feature_extraction(mono, fs5, 1102, 441, deltas=False))
    entry("cfg5_features_20s", dt + dt_mono, F.shape[1], "20 s of seed-5 stereo clip, stereo_to_mono + window 1102 / step 441, "
          "deltas off", stereo_to_mono_seconds=dt_mono)
    with contextlib.redirect_stdout(io.StringIO()):
        dt, (S, _, _) = timed(lambda: ref_st.spectrogram(mono, fs5, 1102, 441))
    entry("cfg5_spectrogram_20s", dt + dt_mono, S.shape[0], "same clip, spectrogram", stereo_to_mono_seconds=dt_mono)
    dt, (Cg, _, _) = timed(lambda: ref_st.chromagram(mono, fs5, 1102, 441))
    entry("cfg5_chromagram_20s", dt + dt_mono, Cg.shape[0], "same clip, chromagram", stereo_to_mono_seconds=dt_mono)
    # the ports bench.py times on the GPU box, here on the same core for the ratio port / reference
    try:
        one = cpu_bench.single_core(synth_clip(2, 60 * fs, fs), fs, 800, 400, 2399)
        res["ports_on_this_core"] = {"numpy_port_frames_per_s": one["numpy_port"], "c_port_frames_per_s": one["c_port"],
                                     "frames": one["frames"]}
    except Exception as exc:
        res["ports_on_this_core"] = {"error": repr(exc)}
    with open(args.out, "w") as f:
        json.dump(res, f, indent=1)
    print("wrote", args.out)


if __name__ == "__main__":
    main()
