"""CPU legs of bench.py (TEST / BENCH INFRASTRUCTURE: bench.py's cpu_baseline only -- never imported by the package).

single_core(): the NumPy port of the reference loop (oracle/paa_oracle.py) and the plain-C port (oracle/paa_oracle.c)
on a bounded prefix of the bench clip, one thread.
all_cores(): os.cpu_count() single-threaded processes of the C port over seeded 10 s clips (config 4's unit), the "all
host cores" figure SURVEY 8d asks for.  Workers are FORKED, so bench.py calls this before it loads HIP."""
import os
import sys
import time

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
if _HERE not in sys.path:
    sys.path.insert(0, _HERE)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def usable_cores():
    """CPUs this process may really use: the affinity mask capped by the cgroup CPU quota (cpu.max), which is what
    bounds a container -- os.cpu_count() alone reports the host's logical CPUs."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(-(-int(quota) // int(period)))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def single_core(x, fs, window, step, budget_frames):
    import paa_oracle as O
    n = min(len(x), window + step * (budget_frames - 1))
    t0 = time.perf_counter()
    F, _ = O.feature_extraction(x[:n], fs, window, step, deltas=False)
    dt_np = time.perf_counter() - t0
    out = {"numpy_port": F.shape[1] / dt_np, "numpy_port_seconds": dt_np, "c_port": None, "c_port_seconds": 0.0,
           "frames": int(F.shape[1]), "seconds_of_audio": n / float(fs)}
    try:
        import c_oracle
        if c_oracle.available():
            t0 = time.perf_counter()
            Fc = c_oracle.feature_extraction(x[:n], fs, window, step, deltas=False)
            out["c_port_seconds"] = time.perf_counter() - t0
            out["c_port"] = Fc.shape[1] / out["c_port_seconds"]
    except Exception as exc:  # the C oracle is optional test infrastructure
        out["c_port_error"] = repr(exc)
    return out


def reference_cost(x, fs, window, step, budget_frames=4000):
    """The cost-faithful restatement of the reference loop (paa_oracle.feature_extraction_reference_cost: chroma tables rebuilt
    per frame, scipy DCT per frame, list + concatenate -- what the unmodified Python reference does, which cannot travel to
    this host) on a prefix of the bench clip, one thread; with the ratio to the real reference measured in the build container."""
    import json
    import paa_oracle as O
    n = min(len(x), window + step * (budget_frames - 1))
    t0 = time.perf_counter()
    F, _ = O.feature_extraction_reference_cost(x[:n], fs, window, step, deltas=False)
    dt = time.perf_counter() - t0
    out = {"value": F.shape[1] / dt, "unit": "frames/s", "cores": 1, "frames": int(F.shape[1]), "seconds": dt,
           "what": "oracle restatement with the reference's per-frame costs (not the reference itself)"}
    try:
        cal = json.load(open(os.path.join(os.path.dirname(_HERE), "profiles", "r06_reference_cost_port.json")))
        out["over_reference_in_build_container"] = cal["cost_port_over_reference"]
    except Exception:
        pass
    return out


def _worker(args):
    seed, fs, window, step, clip_seconds, reps = args
    os.environ["OMP_NUM_THREADS"] = "1"
    import c_oracle
    from synth import synth_clip
    x = synth_clip(seed, int(clip_seconds * fs), fs)
    frames = 0
    t0 = time.perf_counter()
    for _ in range(reps):
        frames += c_oracle.feature_extraction(x, fs, window, step, deltas=False).shape[1]
    return frames, time.perf_counter() - t0


def all_cores(fs, window, step, clip_seconds=10.0, target_seconds=6.0, per_core_rate=4.0e4, timeout=120.0):
    """-> dict with the aggregate frames/s of os.cpu_count() single-threaded C-port workers (None when unavailable).
    A ProcessPoolExecutor on the fork context (256 spawned interpreters take minutes on a cold box): a worker that dies
    breaks the pool with an exception (no respawn loop), and the whole leg is bounded by `timeout` seconds.  Call it
    before the process creates a HIP context."""
    import numpy  # noqa: F401  (loaded in the parent so that the forked workers share it)
    import synth  # noqa: F401
    import concurrent.futures as cf
    import multiprocessing as mp
    import c_oracle
    if not c_oracle.available():
        return None
    n = usable_cores()
    frames_per_clip = (int(clip_seconds * fs) - window) // step + 1
    reps = max(1, int(target_seconds * per_core_rate / frames_per_clip))
    t0 = time.perf_counter()
    with cf.ProcessPoolExecutor(max_workers=n, mp_context=mp.get_context("fork")) as pool:
        list(pool.map(_warm, range(n), timeout=timeout))          # start every worker before the clock runs
        t1 = time.perf_counter()
        res = list(pool.map(_worker, [(40000 + i, fs, window, step, clip_seconds, reps) for i in range(n)],
                            timeout=timeout))
        wall = time.perf_counter() - t1
    frames = sum(r[0] for r in res)
    return {"value": frames / wall, "unit": "frames/s", "cores": n, "host_logical_cpus": os.cpu_count(),
            "cores_note": "affinity mask capped by the cgroup quota (/sys/fs/cgroup/cpu.max)",
            "kind": "port (oracle/paa_oracle.c, one single-threaded process per core)",
            "frames": int(frames), "wall_seconds": wall, "slowest_worker_seconds": max(r[1] for r in res),
            "pool_start_seconds": t1 - t0,
            "sample": "%d workers x %d passes over one seeded %g s clip each (%d frames per pass), deltas off"
                      % (n, reps, clip_seconds, frames_per_clip)}


def _warm(_):
    import c_oracle
    c_oracle.lib()
    return os.getpid()
