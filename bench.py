#!/usr/bin/env python
"""bench.py -- short-term frames/s of the 34-feature extractor on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" = one pass of the hot path (clip statistics -> features) over one batch of synthetic
16 kHz int16 PCM that is already resident in HBM; the [34][T] float64 result stays in HBM.
Workload per rank (weak scaling): BASELINE config 2 -- one synthetic 1-hour 16 kHz mono clip,
window 50 ms / step 25 ms (800/400 samples), 143 999 frames, full 34-feature vector.
With N > 1 every rank processes its own 1-hour clip (seed 2 + rank) and the [34][T] blocks are
gathered to rank 0 with RCCL (paa_comm_gather_f64) inside the timed step.

torch is used for process-group plumbing only (rendezvous, barrier, max over ranks): the compute
path is ctypes -> libpaa_hip.so.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

FS, WINDOW, STEP = 16000, 800, 400
HBM_PEAK_GBS = 8000.0                 # MI355X_MICROARCH.md: 8 TB/s spec
FP64_VALU_PEAK_TFLOPS = 78.6          # datasheet; informational only


def make_workload(name, rank, seconds):
    """-> (packed int16, offsets int64, description)."""
    from synth import synth_clip
    if name == "cfg2":
        n = int(seconds * FS)
        x = synth_clip(2 + rank, n, FS)
        return x, np.array([0, n], dtype=np.int64), "cfg2: 1 clip x %d s, 16 kHz mono int16, 800/400, 34 features" % seconds
    if name == "cfg4":
        # many 10 s clips; a pool of 64 distinct seeded clips is tiled to the batch size
        n_clips = int(seconds)
        n = 10 * FS
        pool = [synth_clip(40000 + 1000 * rank + i, n, FS) for i in range(min(64, n_clips))]
        packed = np.concatenate([pool[i % len(pool)] for i in range(n_clips)])
        return packed, np.arange(n_clips + 1, dtype=np.int64) * n, "cfg4-shard: %d clips x 10 s (64 distinct, tiled)" % n_clips
    raise SystemExit("unknown workload " + name)


def cpu_baseline(x, budget_frames):
    """Both CPU oracles (the NumPy port of the reference loop and the plain-C port) on a bounded prefix of the
    same clip; `value` is the faster of the two so that the GPU/CPU ratio is the conservative one."""
    import paa_oracle as O
    n = min(len(x), WINDOW + STEP * (budget_frames - 1))
    t0 = time.perf_counter()
    F, _ = O.feature_extraction(x[:n], FS, WINDOW, STEP, deltas=False)
    dt_np = time.perf_counter() - t0
    rate_np = F.shape[1] / dt_np
    rate_c, dt_c = None, 0.0
    try:
        import c_oracle
        if c_oracle.available():
            t0 = time.perf_counter()
            Fc = c_oracle.feature_extraction(x[:n], FS, WINDOW, STEP, deltas=False)
            dt_c = time.perf_counter() - t0
            rate_c = Fc.shape[1] / dt_c
    except Exception as exc:  # the C oracle is optional test infrastructure
        print("C oracle not timed:", exc, file=sys.stderr)
    best = max(rate_np, rate_c or 0.0)
    return {"value": best, "unit": "frames/s", "cores": 1, "kind": "port",
            "numpy_port": rate_np, "c_port": rate_c,
            "sample": "first %.0f s of the same clip (%d frames; %.1f s of CPU for oracle/paa_oracle.py, %.1f s for "
                      "oracle/paa_oracle.c), deltas off, single thread; host has %d cores"
                      % (n / FS, F.shape[1], dt_np, dt_c, os.cpu_count() or 0)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="cfg2", choices=["cfg2", "cfg4"])
    ap.add_argument("--seconds", type=float, default=3600.0, help="cfg2: clip length; cfg4: number of clips")
    ap.add_argument("--deltas", type=int, default=0, help="1: 68-row output (reference default), 0: the 34-feature metric")
    ap.add_argument("--no-gather", action="store_true", help="N>1: skip the RCCL gather to rank 0")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=144000, help="frames of the clip prefix timed on the CPU (about 8 s of CPU)")
    ap.add_argument("--check", type=int, default=1, help="verify a few frames against the oracle after timing")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)

    from pyaudioanalysis_amd import _ffi
    lib = _ffi.lib()
    if _ffi.device_count() < 1:
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    _ffi.init(local_rank % _ffi.device_count())

    x, offsets, desc = make_workload(args.workload, rank, args.seconds)
    d_in = _ffi.DeviceBuffer.from_host(x)
    plan = _ffi.Plan(offsets, FS, WINDOW, STEP, deltas=bool(args.deltas), sample_kind=0)
    F = plan.F
    d_out = _ffi.DeviceBuffer(plan.out_doubles * 8)
    d_out2 = _ffi.DeviceBuffer(plan.out_doubles * 8) if world > 1 else d_out   # N>1: the gather of step k overlaps step k+1
    frames = plan.total_frames

    gather = world > 1 and not args.no_gather
    gather_note = None
    d_all = None
    counts = np.full(world, plan.out_doubles, dtype=np.int64)
    comm = None
    if gather:
        from pyaudioanalysis_amd import distributed as D

        def bcast(payload):
            obj = [payload]
            dist.broadcast_object_list(obj, src=0)
            return obj[0]
        try:
            comm = D.RcclGather(world, rank, bcast)
            if rank == 0:
                d_all = _ffi.DeviceBuffer(int(counts.sum()) * 8)
            ok = True
        except Exception as exc:       # keep the scaling run alive, say what happened
            ok = False
            gather_note = "RCCL init failed on rank %d: %s" % (rank, exc)
        all_ok = [None] * world
        dist.all_gather_object(all_ok, ok)
        gather = all(all_ok)

    bufs = [d_out, d_out2]
    state = {"k": 0}

    def step():
        buf = bufs[state["k"] & 1]
        state["k"] += 1
        plan.execute(d_in, buf)
        if gather:
            # runs on the library's communication stream; the next write of `buf` waits for it
            comm.gather(buf, counts, 0, d_all)

    def device_sync():
        _ffi.sync()                      # both library streams (compute + communication)
        if dist is not None:             # N > 1 also drains torch's view of this rank's device
            try:
                import torch
                if torch.cuda.is_available():
                    torch.cuda.set_device(local_rank % torch.cuda.device_count())
                    torch.cuda.synchronize()
            except Exception:
                pass

    def barrier():
        device_sync()
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    barrier()
    _ffi.check(lib.paa_prof_enable(1))
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    device_sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
        import torch
        tt = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    import ctypes
    kms = ctypes.c_double()
    kn = ctypes.c_int64()
    _ffi.check(lib.paa_prof_read(ctypes.byref(kms), ctypes.byref(kn)))
    _ffi.check(lib.paa_prof_enable(0))

    # N > 1: also time the same steps without the gather (reported beside the headline value)
    value_no_gather = None
    if gather:
        saved, gather = gather, False
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        _ffi.sync()
        el2 = time.perf_counter() - t1
        dist.barrier()
        import torch
        tt = torch.tensor([el2], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        value_no_gather = frames * world * args.steps / float(tt.item())
        gather = saved

    if rank == 0:
        total_frames = frames * world
        ms_per_step = 1e3 * elapsed / args.steps
        value = total_frames * args.steps / elapsed
        bytes_per_frame = 2 * STEP + 8 * F               # int16 in once + f64 out once
        k_avg_ms = kms.value / max(1, kn.value)
        achieved = bytes_per_frame * frames / (k_avg_ms * 1e-3) / 1e9 if k_avg_ms > 0 else 0.0
        result = {
            "metric": "short-term frames/sec (34-feat, 16 kHz, 50 ms/25 ms) + HBM GB/s vs peak",
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic (oracle/synth.py, seed 2+rank)",
            "config": {"workload": desc, "frames_per_step_per_gpu": int(frames), "window": WINDOW, "step": STEP,
                       "rows": F, "kernel": plan.kernel_name,
                       "multi_gpu": ("one clip per rank, RCCL gather to rank 0" if gather else
                                     ("one clip per rank, no gather" if world > 1 else "single GPU"))},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": plan.kernel_name, "kernel_avg_ms": k_avg_ms, "launches_timed": int(kn.value),
                         "algorithmic_bytes_per_frame": bytes_per_frame,
                         "note": "path is FP64 VALU/LDS bound (about 40 flop/B); HBM fraction is reported as the "
                                 "metric asks, see DESIGN.md"},
        }
        if gather_note:
            result["config"]["gather_note"] = gather_note
        if value_no_gather is not None:
            result["config"]["frames_per_s_without_gather"] = value_no_gather
            result["config"]["gather_bytes_per_step_into_root"] = int(plan.out_doubles * 8 * (world - 1))
        # HBM traffic of the feature kernel from the committed rocprofv3 PMC pass of this same command
        # (FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950, WRITE_SIZE as reported), per launch
        try:
            prof = json.load(open(os.path.join(ROOT, "profiles", "latest_traffic.json")))
            if prof.get("kernel") == plan.kernel_name and prof.get("frames") == int(frames) and prof.get("rows") == F:
                result["roofline"]["traffic"] = prof["hbm_bytes_per_launch"]
                result["roofline"]["traffic_over_algorithmic"] = prof["hbm_bytes_per_launch"] / float(bytes_per_frame * frames)
                result["roofline"]["traffic_source"] = prof.get("source")
        except Exception:
            pass
        # informational: executed FP64 work against the vector FP64 peak (the roofline that actually binds)
        result["roofline"]["fp64_valu"] = {"algorithmic_kflop_per_frame": 45.0,
                                           "achieved_tflops": 45.0e3 * frames / (k_avg_ms * 1e-3) / 1e12 if k_avg_ms > 0 else 0.0,
                                           "peak_tflops": FP64_VALU_PEAK_TFLOPS}
        if args.check:
            import paa_oracle as O
            got = d_out.to_host(np.float64, plan.out_doubles).reshape(-1)
            T0 = int(lib.paa_num_frames(int(offsets[1] - offsets[0]), WINDOW, STEP))
            slab = got[:F * T0].reshape(F, T0)
            xn = O.normalize_clip(x[offsets[0]:offsets[1]])
            tab = O.Tables(FS, WINDOW)
            worst = 0
            for t in (0, 1, 63, 64, 65, T0 // 2, T0 - 1):
                fr = xn[t * STEP:t * STEP + WINDOW]
                X = O.magnitude_spectrum(fr, tab.nfft)
                Xp = X if t == 0 else O.magnitude_spectrum(xn[(t - 1) * STEP:(t - 1) * STEP + WINDOW], tab.nfft)
                v = O.frame_vector(fr, X, Xp, tab)
                nb, _ = O.mixed_tolerance_violations(slab[:34, t:t + 1], v[:, None], 1e-4, 1e-5, 1e-8)
                worst += nb
            result["parity_spot_check"] = "ok" if worst == 0 else "FAILED (%d entries)" % worst
        if not args.no_cpu_baseline and world == 1:
            result["cpu_baseline"] = cpu_baseline(x[offsets[0]:offsets[1]], args.cpu_frames)
        elif not args.no_cpu_baseline:
            result["cpu_baseline"] = None
        print(json.dumps(result))
        sys.stdout.flush()
    if comm is not None:
        comm.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
