#!/usr/bin/env python
"""bench.py -- short-term frames/s of the 34-feature extractor on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...     (any launcher that exports
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT will do; bench.py itself does not import torch)

A "step" = one pass of the hot path (clip statistics -> clip constants -> features) over one batch of synthetic
16 kHz int16 PCM that is already resident in HBM; the [34][T] float64 result stays in HBM.

N = 1 (the headline): BASELINE config 2 -- one synthetic 1-hour 16 kHz mono clip, window 50 ms / step 25 ms
  (800/400 samples), 143 999 frames, full 34-feature vector.  The steps rotate over THREE distinct clips (3 x 115 MB of
  input: more than the 256 MiB Infinity Cache) so that the input really comes from HBM.
N > 1 (default): BASELINE config 4, strong scaling -- 100 000 synthetic 10 s clips (64 distinct seeded clips, tiled)
  split into contiguous ranges with distributed.partition_by_frames, every rank extracts its range and the [34][399]
  slabs are gathered to rank 0 with RCCL (paa_comm_gather_f64, own stream, overlapping the next step) inside the timed
  region; the same job is also timed without the gather and with the cheap gather (68 short-term rows stay in HBM, the
  (136, 10) mid-term matrices of 1.0 s / 1.0 s go to rank 0: config 3's product on config 4's clips).  --workload cfg2
  gives one 1-hour clip per rank instead (weak scaling).

The line also carries (N = 1): the other BASELINE configurations kernel-resident (config.others: config 3 mid-term
batch, a config-4 shard, config 5 features / spectrogram / chromagram starting from the interleaved stereo int16 samples,
the shapes of the reference's other callers -- 40 ms windows, float64 / stereo input at 50 ms, 8 kHz and 48 kHz), the
host-to-host rate of the drop-in API, a >= 2 s sustained loop, a full-matrix parity check against the plain-C oracle,
the CPU ports on one core and on all host cores of THIS box, and the unmodified reference's own timings (measured in the
build container, profiles/reference_cpu_r03.json: /root/reference does not exist on the GPU box).

The control plane for N > 1 (rendezvous, barrier, max over ranks) is pyaudioanalysis_amd/_rendezvous.py (TCP sockets,
standard library); the compute path is ctypes -> libpaa_hip.so -> RCCL.  Prints ONE JSON line on rank 0.  Nothing here
reads /root/reference.
"""
import argparse
import ctypes
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

FS, WINDOW, STEP = 16000, 800, 400
HBM_PEAK_GBS = 8000.0                 # MI355X_MICROARCH.md: 8 TB/s spec
FP64_VALU_PEAK_TFLOPS = 78.6          # datasheet; informational only
KFLOP_PER_FRAME = 45.0                # ~35-55 kflop of FP64 per 800/400 frame
CFG4_TOTAL_CLIPS = 100000
PROFILE_ROUND = "r06"                 # profiles/latest_traffic.json must come from this round's PMC pass of the headline
                                      # kernel (scripts/profile.sh r04): an older file is reported, flagged traffic_stale
XGMI_LINK_GBS = 76.8                  # one xGMI link, one direction (7 links into the root at N = 8)


# the shapes of the reference's other callers, kernel-resident (also the cases of scripts/kernel_loop.py):
# name: (fs, window, step, seconds of one clip, clips, sample kind [0 int16, 1 float64, 2 interleaved stereo int16], mode
#        [0 features, 1 spectrogram, 2 chromagram], deltas)
SHAPES = {
    "reg_features": (44100, 1102, 441, 600, 1, 0, 0, 0),
    "reg_features_stereo": (44100, 1102, 441, 600, 1, 2, 0, 0),
    "reg_spectrogram": (44100, 1102, 441, 600, 1, 0, 1, 0),
    "reg_spectrogram_stereo": (44100, 1102, 441, 600, 1, 2, 1, 0),
    "reg_chromagram": (44100, 1102, 441, 600, 1, 0, 2, 0),
    "reg_chromagram_stereo": (44100, 1102, 441, 600, 1, 2, 2, 0),
    "ct_640": (16000, 640, 640, 3600, 1, 0, 0, 0),                 # the CLI's 40 ms windows (audioAnalysis.py:71,80)
    "ct_640_spectrogram": (16000, 640, 640, 3600, 1, 0, 1, 0),
    "ct_640_chromagram": (16000, 640, 640, 3600, 1, 0, 2, 0),
    "ct_800_f64": (16000, 800, 400, 3600, 1, 1, 0, 0),             # what stereo_to_mono hands on (audioBasicIO.py:167)
    "ct_800_stereo": (16000, 800, 400, 3600, 1, 2, 0, 0),
    "ct_400": (8000, 400, 200, 3600, 2, 0, 0, 0),                  # 50 ms at 8 kHz (audioTrainTest.py:28-29)
    "ct_320": (16000, 320, 160, 1800, 1, 0, 0, 0),
    "w1024": (16000, 1024, 512, 3600, 1, 0, 0, 0),                 # power-of-two windows: three-pass register FFT 8 x 8 x 8 ...
    "w2048": (44100, 2048, 1024, 1200, 1, 0, 0, 0),                # ... 16 x 16 x 4
    "w512": (16000, 512, 256, 3600, 1, 0, 0, 0),                   # ... 4 x 8 x 8
    "w1024_68": (16000, 1024, 512, 3600, 1, 0, 0, 1),
    "w1024_spectrogram": (16000, 1024, 512, 3600, 1, 0, 1, 0),
    "w2400": (48000, 2400, 1200, 1200, 1, 0, 0, 0),                # 50 ms at 48 kHz
    "w2205": (44100, 2205, 1102, 1200, 1, 0, 0, 0),                # 50 ms at 44.1 kHz (odd window)
    "w1764": (44100, 1764, 1764, 1200, 1, 0, 0, 0),                # the CLI's 40 ms at 44.1 kHz (audioAnalysis.py:71,80)
    "w1920": (48000, 1920, 1920, 1200, 1, 0, 0, 0),                # 40 ms at 48 kHz
    "w551_11k": (11025, 551, 275, 3600, 1, 0, 0, 0),               # 50 ms at 11.025 kHz (odd: 19 x 29)
    "w551_22k": (22050, 551, 220, 1800, 1, 0, 0, 0),               # 25 ms at 22.05 kHz
    "w2400_68": (48000, 2400, 1200, 1200, 1, 0, 0, 1),             # 50 ms at 48 kHz with deltas (what mid-term extraction runs)
    "w2205_stereo_68": (44100, 2205, 1102, 1200, 1, 2, 0, 1),
    "mid_stats": (16000, 800, 400, 30, 1000, 0, 0, 1),
    "fast_s800": (16000, 800, 800, 30, 1000, 0, 0, 1),             # 50 ms / 50 ms (audioTrainTest.py:28-29), 1000 clips, 68 rows
    "mix_4800": (96000, 4800, 2400, 300, 1, 0, 0, 0),              # 50 ms at 96 kHz: the in-place mixed-radix kernel, full instance
    "mix_256": (16000, 256, 128, 3600, 1, 0, 0, 0),                # 16 ms windows: its lean, skewed instance
    "blu_1103": (22050, 1103, 441, 3600, 1, 0, 0, 0),               # a prime window: Bluestein convolution of length 2048 (st_generic until round 5)
    "blu_661": (22050, 661, 220, 3600, 1, 0, 0, 0),                 # 0.030 x 22050 = 661 (prime): convolution length 1024
    "blu_736": (16000, 736, 368, 1800, 1, 0, 0, 0),                # 46 ms at 16 kHz = 2^5 x 23: packed as 368 complex points, length 1024
    "blu_3002": (44100, 3002, 1501, 600, 1, 0, 0, 0),              # 2 x 19 x 79: packed, length 4096 (the direct form would need 8192)
    "blu_1103_spectrogram": (22050, 1103, 441, 1800, 1, 0, 1, 0),
    "blu_2203": (44100, 2203, 1100, 1200, 1, 0, 0, 0),             # a prime 50 ms window at 44.1 kHz: convolution length 4096
    "blu_202": (16000, 202, 101, 1800, 1, 0, 0, 0),                # 2 x 101: length 512
    "blu_4001": (96000, 4001, 2000, 600, 1, 0, 0, 0),              # a prime window at 96 kHz: length 8192, one wave per CU
    "big_16000": (16000, 16000, 8000, 600, 1, 0, 0, 0),            # music_thumbnailing's 1 s window (audioSegmentation.py:1137)
    "big_16000_1h": (16000, 16000, 8000, 3600, 1, 0, 0, 0),        # ... on the one-hour clip (7 199 frames: 28 per CU)
    "big_16000_68": (16000, 16000, 8000, 600, 1, 0, 0, 1),         # ... with deltas, as music_thumbnailing calls it
    "big_8000_batch": (16000, 8000, 4000, 30, 200, 0, 0, 0),       # 0.5 s windows, 200 clips x 30 s in one plan
    "big_44100": (44100, 44100, 22050, 300, 1, 0, 0, 0),           # 1 s at 44.1 kHz: 22 050 complex points do not fit the LDS: real-input split, 6 x 3675 points
    "big_44100_20min": (44100, 44100, 22050, 1200, 1, 0, 0, 0),    # ... a 20-minute recording (2 399 frames: 28 tasks per CU)
    "big_22050": (22050, 22050, 11025, 1200, 1, 0, 0, 0),          # 1 s at 22.05 kHz: 3 x 3675 points
    "big_48000": (48000, 48000, 24000, 600, 1, 0, 0, 0),           # 1 s at 48 kHz: 6 x 4000 points (8 x 20 x 25 per sub-transform)
    "big_32000": (32000, 32000, 16000, 600, 1, 0, 0, 0),           # 1 s at 32 kHz: 4 x 4000 points
}
SAMPLE_BYTES = {0: 2, 1: 8, 2: 4}


def shape_input(name):
    """(host sample array, offsets) of a SHAPES entry: at most 100 s synthesised (seed 5), tiled to the length."""
    from synth import synth_clip
    fs, W, S, seconds, clips, kind, mode, deltas = SHAPES[name]
    base_s = min(seconds, 100)
    reps = -(-seconds // base_s) * clips
    n = base_s * fs
    if kind == 0:
        x = np.tile(synth_clip(5, n, fs), reps)
    else:
        xs = synth_clip(5, n, fs, stereo=True)
        x = np.tile((xs[:, 1] / 2) + (xs[:, 0] / 2), reps) if kind == 1 else np.tile(xs, (reps, 1))
    per = x.shape[0] // clips
    return np.ascontiguousarray(x).reshape(-1), np.arange(clips + 1, dtype=np.int64) * per


def shape_bytes_per_frame(name, rows):
    """SURVEY 8d's accounting: the samples of one step in once, the rows of one frame out once."""
    fs, W, S, seconds, clips, kind, mode, deltas = SHAPES[name]
    return SAMPLE_BYTES[kind] * S + 8 * rows


def replicate_on_device(ffi, pool_host, n_units):
    """Device buffer holding n_units copies-in-rotation of the pool's units (pool_host: [n_pool, unit] array): the pool
    goes over PCIe once, the tiling is device-to-device."""
    lib = ffi.lib()
    n_pool, unit = pool_host.shape
    unit_bytes = unit * pool_host.itemsize
    d_pool = ffi.DeviceBuffer.from_host(pool_host.reshape(-1))
    d_all = ffi.DeviceBuffer(n_units * unit_bytes)
    done = 0
    while done < n_units:
        cnt = min(n_pool, n_units - done)
        ffi.check(lib.paa_memcpy_d2d(ctypes.c_void_p(d_all.ptr.value + done * unit_bytes), d_pool.ptr, cnt * unit_bytes))
        done += cnt
    ffi.sync()
    return d_all


def timed(ffi, fn, steps, warmup):
    """HIP-event time per call of fn() on the library stream (ms)."""
    lib = ffi.lib()
    for _ in range(warmup):
        fn()
    ffi.sync()
    ffi.check(lib.paa_timer_start())
    for _ in range(steps):
        fn()
    ms = ctypes.c_float()
    ffi.check(lib.paa_timer_stop(ctypes.byref(ms)))
    return ms.value / steps


def other_configs(ffi, steps=10):
    """BASELINE configs 3, 4 (one GPU's shard) and 5, kernel-resident, with SURVEY 8d's algorithmic bytes per frame."""
    from synth import synth_clip
    out = {}

    def entry(frames, ms, bytes_per_frame, kernel, extra=None):
        e = {"frames_per_step": int(frames), "ms_per_step": ms, "frames_per_s": frames / (ms * 1e-3),
             "algorithmic_bytes_per_frame": bytes_per_frame,
             "achieved_GBps": bytes_per_frame * frames / (ms * 1e-3) / 1e9, "kernel": kernel}
        e["hbm_frac"] = e["achieved_GBps"] / HBM_PEAK_GBS
        if extra:
            e.update(extra)
        return e

    # ---- config 3: 1000 x 30 s, short 800/400 with deltas (what mid_feature_extraction computes), mid 1.0 s / 1.0 s
    n3 = 30 * FS
    pool = np.stack([synth_clip(3000 + i, n3, FS) for i in range(8)])
    d_in = replicate_on_device(ffi, pool, 1000)
    plan = ffi.Plan(np.arange(1001, dtype=np.int64) * n3, FS, WINDOW, STEP, deltas=True)
    d_st = ffi.DeviceBuffer(plan.out_doubles * 8)
    d_mid = ffi.DeviceBuffer(plan.mid_doubles(40) * 8)

    def step3():
        plan.execute(d_in, d_st)
        plan.mid_execute(d_st, 39, 40, d_mid)          # ratio / step ratio of 1.0 s / 1.0 s over 50 ms / 25 ms (:100-102)
    ms = timed(ffi, step3, steps, 2)
    out["cfg3"] = entry(plan.total_frames, ms, 2 * STEP + 8 * 68 + 8 * 136 * 30 / 1199.0, plan.kernel_name,
                        {"workload": "1000 clips x 30 s (8 distinct, tiled), 68 short-term rows + (136, 30) mid-term per clip",
                         "clips_per_s": 1000 / (ms * 1e-3)})
    plan.destroy()
    del d_in, d_st, d_mid
    # ---- config 4, one GPU's shard: 12 500 x 10 s
    n4 = 10 * FS
    pool = np.stack([synth_clip(40000 + i, n4, FS) for i in range(64)])
    d_in = replicate_on_device(ffi, pool, 12500)
    plan = ffi.Plan(np.arange(12501, dtype=np.int64) * n4, FS, WINDOW, STEP, deltas=False)
    d_out = ffi.DeviceBuffer(plan.out_doubles * 8)
    ms = timed(ffi, lambda: plan.execute(d_in, d_out), steps, 2)
    out["cfg4_shard"] = entry(plan.total_frames, ms, 2 * STEP + 8 * 34, plan.kernel_name,
                              {"workload": "12 500 clips x 10 s (64 distinct, tiled), 34 rows: one GPU's share of config 4"})
    plan.destroy()
    del d_in, d_out
    # ---- the reference's own default shape for classification / segmentation (50 ms / 50 ms: window 800, step 800,
    # audioTrainTest.py:28-29): 1000 clips x 30 s, 68 rows
    plan = ffi.Plan(np.arange(1001, dtype=np.int64) * n3, FS, WINDOW, WINDOW, deltas=True)
    d_in = replicate_on_device(ffi, np.stack([synth_clip(3000 + i, n3, FS) for i in range(8)]), 1000)
    d_out = ffi.DeviceBuffer(plan.out_doubles * 8)
    ms = timed(ffi, lambda: plan.execute(d_in, d_out), steps, 2)
    out["step800_68rows"] = entry(plan.total_frames, ms, 2 * WINDOW + 8 * 68, plan.kernel_name,
                                  {"workload": "1000 clips x 30 s (8 distinct, tiled), window 800 / step 800, 68 rows"})
    plan.destroy()
    del d_in, d_out
    # ---- config 5: 44.1 kHz stereo -> mono, window 25 ms / step 10 ms (1102 / 441), 600 s.  The resident input is the
    # interleaved stereo int16 buffer as it comes from the file (1 764 B per frame step): stereo_to_mono
    # (audioBasicIO.py:156-168) happens in the kernels' sample loads, inside the timed step.  Beside it: int16 mono and the
    # float64 mono array the reference itself would hand over.
    def run_shape(name, key, what, launches=None):
        fs_, W_, S_, seconds, clips, kind, mode, deltas = SHAPES[name]
        x, offs = shape_input(name)
        d_in = ffi.DeviceBuffer.from_host(x)
        plan = ffi.Plan(offs, fs_, W_, S_, deltas=bool(deltas), sample_kind=kind, mode=mode)
        d_out = ffi.DeviceBuffer(plan.out_doubles * 8)
        ms = timed(ffi, lambda: plan.execute(d_in, d_out), launches or 4 * steps, 2 if launches else 8)   # (sub-millisecond steps: 40 of them)
        rows = plan.F if mode != 0 else (68 if deltas else 34)
        out[key] = entry(plan.total_frames, ms, shape_bytes_per_frame(name, rows), plan.kernel_name,
                         {"workload": what, "fs": fs_, "window": W_, "step": S_,
                          "samples": ("int16 mono", "float64 mono", "interleaved stereo int16")[kind]})
        plan.destroy()

    cfg5 = "600 s of 44.1 kHz audio (100 s seeded stereo clip, tiled x6), window 1102 / step 441, "
    run_shape("reg_features_stereo", "cfg5_features", cfg5 + "34 feature rows from the interleaved stereo samples")
    run_shape("reg_spectrogram_stereo", "cfg5_spectrogram", cfg5 + "551 spectrogram rows from the interleaved stereo samples")
    run_shape("reg_chromagram_stereo", "cfg5_chromagram", cfg5 + "12 chromagram rows from the interleaved stereo samples")
    run_shape("reg_features", "cfg5_features_mono_i16", cfg5 + "int16 mono resident (left channel)")
    run_shape("reg_spectrogram", "cfg5_spectrogram_mono_i16", cfg5 + "int16 mono resident (left channel)")
    # ---- the shapes of the reference's other callers
    run_shape("ct_640", "w640_step640", "1 h at 16 kHz, 40 ms / 40 ms (the CLI's spectrogram / chromagram window), features")
    run_shape("ct_640_spectrogram", "w640_spectrogram", "1 h at 16 kHz, 40 ms / 40 ms, spectrogram rows")
    run_shape("ct_640_chromagram", "w640_chromagram", "1 h at 16 kHz, 40 ms / 40 ms, chromagram rows")
    run_shape("ct_800_f64", "w800_float64", "1 h at 16 kHz, 800 / 400, float64 mono samples (stereo_to_mono's output)")
    run_shape("ct_800_stereo", "w800_stereo", "1 h at 16 kHz, 800 / 400, interleaved stereo int16 samples")
    run_shape("ct_400", "w400_8kHz", "2 x 1 h at 8 kHz, 50 ms / 25 ms (400 / 200)")
    run_shape("w1024", "w1024_16kHz", "1 h at 16 kHz, window 1024 / step 512 (power of two: three-pass register FFT 8 x 8 x 8)")
    run_shape("w1024_68", "w1024_16kHz_68rows", "1 h at 16 kHz, 1024 / 512, 68 rows")
    run_shape("w1024_spectrogram", "w1024_spectrogram", "1 h at 16 kHz, 1024 / 512, spectrogram rows")
    run_shape("w2048", "w2048_44kHz", "20 min at 44.1 kHz, window 2048 / step 1024 (16 x 16 x 4)")
    run_shape("w512", "w512_16kHz", "1 h at 16 kHz, window 512 / step 256 (4 x 8 x 8; entropy blocks of 51 samples)")
    run_shape("w2400", "w2400_48kHz", "20 min at 48 kHz, 50 ms / 25 ms (2400 / 1200)")
    run_shape("w2205", "w2205_44kHz", "20 min at 44.1 kHz, 50 ms / 25 ms (2205 / 1102, odd window)")
    run_shape("w2400_68", "w2400_48kHz_68rows", "20 min at 48 kHz, 2400 / 1200, 68 rows (what mid-term extraction runs)")
    run_shape("w2205_stereo_68", "w2205_44kHz_stereo_68rows", "20 min at 44.1 kHz, 2205 / 1102, interleaved stereo int16, 68 rows")
    run_shape("w1764", "w1764_44kHz", "20 min at 44.1 kHz, 40 ms / 40 ms (1764 / 1764: the CLI's window at 44.1 kHz)")
    run_shape("w1920", "w1920_48kHz", "20 min at 48 kHz, 40 ms / 40 ms (1920 / 1920)")
    run_shape("w551_11k", "w551_11kHz", "1 h at 11.025 kHz, 50 ms / 25 ms (551 / 275, odd window 19 x 29)")
    run_shape("w551_22k", "w551_22kHz", "30 min at 22.05 kHz, 25 ms / 10 ms (551 / 220)")
    # windows whose FFT length has a prime factor above 13 (the reference takes any int(window), :563-564): Bluestein kernel
    run_shape("blu_1103", "w1103_22kHz", "1 h at 22.05 kHz, window 1103 (prime) / step 441: chirp convolution of length 2048")
    run_shape("blu_661", "w661_22kHz", "1 h at 22.05 kHz, 30 ms / 10 ms (661, prime / 220): convolution length 1024")
    run_shape("blu_736", "w736_16kHz", "30 min at 16 kHz, 46 ms / 23 ms (736 = 2^5 x 23 / 368): packed as 368 complex points, convolution length 1024")
    run_shape("blu_1103_spectrogram", "w1103_spectrogram", "30 min at 22.05 kHz, 1103 / 441, spectrogram rows")
    run_shape("mix_256", "w256_16kHz", "1 h at 16 kHz, window 256 / step 128")
    # the window of music_thumbnailing (audioSegmentation.py:1137: 1 s / 0.5 s): beyond the LDS envelope, passes through HBM
    run_shape("big_16000", "w16000_16kHz", "10 min at 16 kHz, 1 s / 0.5 s (16000 / 8000): fused three-pass kernel, one run of frames per workgroup, transform in registers", launches=20)
    run_shape("big_16000_1h", "w16000_16kHz_1h", "1 h at 16 kHz, 16000 / 8000", launches=10)
    run_shape("big_16000_68", "w16000_16kHz_68rows", "10 min at 16 kHz, 16000 / 8000, 68 rows", launches=20)
    run_shape("big_8000_batch", "w8000_batch", "200 clips x 30 s at 16 kHz, 8000 / 4000, one plan", launches=20)
    run_shape("big_44100", "w44100_44kHz", "5 min at 44.1 kHz, 1 s / 0.5 s (44100 / 22050): real-input split, six transforms of 3675 points on register passes", launches=20)
    run_shape("big_44100_20min", "w44100_44kHz_20min", "20 min at 44.1 kHz, 44100 / 22050", launches=10)
    run_shape("big_22050", "w22050_22kHz", "20 min at 22.05 kHz, 1 s / 0.5 s (22050 / 11025): three transforms of 3675 points", launches=20)
    run_shape("big_48000", "w48000_48kHz", "10 min at 48 kHz, 1 s / 0.5 s (48000 / 24000): real-input split, six transforms of 4000 points", launches=20)
    run_shape("big_32000", "w32000_32kHz", "10 min at 32 kHz, 1 s / 0.5 s (32000 / 16000): four transforms of 4000 points", launches=20)
    return out


def host_to_host(ffi, clip):
    """The drop-in API as a caller sees it: NumPy in -> NumPy out, PCIe included (fresh output array per call)."""
    from pyaudioanalysis_amd import ShortTermFeatures
    res = {}
    for label, seconds, reps in (("10_min_clip", 600, 5), ("1_hour_clip", 3600, 3)):
        if len(clip) < seconds * FS:
            continue
        x = np.ascontiguousarray(clip[:seconds * FS])
        ShortTermFeatures.feature_extraction(x, FS, WINDOW, STEP, deltas=False)
        best = 1e9
        for _ in range(reps):
            t0 = time.perf_counter()
            F, _ = ShortTermFeatures.feature_extraction(x, FS, WINDOW, STEP, deltas=False)
            best = min(best, time.perf_counter() - t0)
        res[label] = {"frames": int(F.shape[1]), "ms": best * 1e3, "frames_per_s": F.shape[1] / best,
                      "pcie_GBps": (x.nbytes + F.nbytes) / best / 1e9}
    res["note"] = "ShortTermFeatures.feature_extraction(int16 ndarray) -> fresh (34, T) float64 ndarray, best of a few calls"
    return res


def visible_devices():
    """HIP devices this host shows, counted in a throw-away child (the launcher itself never loads the HIP runtime)."""
    import subprocess
    code = ("import sys; sys.path.insert(0, %r); from pyaudioanalysis_amd import _ffi; _ffi.lib(); "
            "print('DEVICES', _ffi.device_count())" % ROOT)
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    for ln in res.stdout.splitlines():
        if ln.startswith("DEVICES "):
            return int(ln.split()[1])
    raise SystemExit("bench.py: cannot count the HIP devices (libpaa_hip.so missing / not loadable?):\n%s%s" % (res.stdout, res.stderr))


def self_launch(args):
    """`python bench.py --gpus N` without a launcher's environment: spawn the N ranks here -- one process per GPU, RANK /
    LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set as torchrun would, rank r on device r -- and wait for them; rank
    0 prints the one JSON line on the stdout it inherits.  Fewer than N devices: a job with the RCCL gather cannot run (two
    ranks on one device never leave ncclCommInitRank) and is refused HERE, loudly, with a non-zero exit -- the line never claims
    GPUs that did not run; with --no-gather the ranks may share devices (what the one-GPU test box exercises: partitioning,
    control plane, per-rank plans), and the line says which device every rank used (config.devices)."""
    import socket
    import subprocess
    n = args.gpus
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    have = visible_devices()
    if have < 1:
        sys.stderr.write("bench.py: no HIP device visible (this bench has no CPU path)\n")
        return 2
    if have < n and not args.no_gather:
        sys.stderr.write("bench.py: --gpus %d asked for, %d HIP device(s) visible: the %d-rank job with the RCCL gather needs one "
                         "device per rank; not running a smaller job under that name (use --no-gather to time the ranks' "
                         "compute on shared devices)\n" % (n, have, n))
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   PAA_BENCH_SELF_LAUNCHED="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        pending = set(range(n))
        while pending:
            for r in sorted(pending):
                code = procs[r].poll()
                if code is None:
                    continue
                pending.discard(r)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
                    sys.stderr.write("bench.py: rank %d exited with %d; stopping the other ranks\n" % (r, code))
                    for q in pending:
                        procs[q].terminate()
            time.sleep(0.1)
    finally:
        for p_ in procs:
            if p_.poll() is None:
                p_.kill()
    return rc


def compact(entry):
    """[frames/s, ms per step, fraction of the HBM peak, kernel] of a config.others entry, four significant digits"""
    def r4(v):
        return float("%.4g" % v)
    return [r4(entry["frames_per_s"]), r4(entry["ms_per_step"]), r4(entry["hbm_frac"]), entry["kernel"]]


LINE_LIMIT = 7600                     # the driver keeps the last 8 018 characters of stdout: the ONE line must fit whole


def _r(v, digits=5):
    """numbers of the line rounded to `digits` significant digits (the full-precision record goes to the side file)"""
    if isinstance(v, bool) or v is None:
        return v
    if isinstance(v, float):
        return float("%.*g" % (digits, v))
    if isinstance(v, dict):
        return {k: _r(x, digits) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_r(x, digits) for x in v]
    return v


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def format_line(result, side_path=None, limit=LINE_LIMIT):
    """The ONE JSON line of the run, at most `limit` characters (BENCH_r05: a 20.5 KB line did not fit the driver's 8 KB tail
    and was recorded as unparsed).  `result` is the full record -- it goes to `side_path` (named in the line) untouched; the
    line keeps the contract's keys, the whole `roofline` and `cpu_baseline` objects with their numbers and short provenance
    strings, the parity check, and ends with the compact `configs` table.  Sections are dropped from the END of a fixed
    priority list until the line fits, never the contract keys."""
    cfg = result.get("config", {})
    line = {k: result[k] for k in ("metric", "value", "unit", "n_gpus", "ranks", "steps", "warmup", "ms_per_step", "higher_is_better",
                                   "scaling", "vs_baseline", "dtype", "data") if k in result}
    line["config"] = _pick(cfg, ("workload", "kernel", "rows", "window", "step", "frames_per_step_job", "frames_per_step_rank0", "clips_in_job",
                                 "input_buffers_rotated", "devices", "distinct_devices", "rccl_ranks", "launched_by", "multi_gpu",
                                 "gather_note", "value_is"))
    roof = result.get("roofline") or {}
    r = _pick(roof, ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_avg_ms", "launches_timed",
                     "algorithmic_bytes_per_frame", "traffic_over_algorithmic", "traffic_round", "traffic_stale"))
    if roof.get("traffic") is not None:
        r["traffic_source"] = "committed PMC pass of this command, not this run"
    fv = roof.get("fp64_valu")
    if fv:
        r["fp64_valu"] = _pick(fv, ("achieved_tflops", "peak_tflops", "frac", "issued_kflop_per_frame_measured", "issued_frac",
                                    "sustained_clock_ghz", "issued_frac_at_sustained_clock"))
    r["note"] = "FP64-VALU bound (~40 flop/B); HBM fraction reported as the metric asks"
    line["roofline"] = r
    cb = result.get("cpu_baseline")
    if cb:
        c = _pick(cb, ("value", "unit", "cores", "kind", "sample", "numpy_port", "c_port", "cpu_model", "host_cores",
                       "reference_cost_port", "reference_note"))
        if isinstance(cb.get("all_cores"), dict):
            c["all_cores"] = _pick(cb["all_cores"], ("value", "cores", "kind", "error"))
        ref = cb.get("reference")
        if isinstance(ref, dict) and "frames_per_s" in ref:
            c["reference_in_build_container"] = {"frames_per_s_cfg2": (ref["frames_per_s"] or {}).get("cfg2_60s_34rows"),
                                                 "cores": 1, "cpu_model": ref.get("cpu_model"), "source": ref.get("source")}
        line["cpu_baseline"] = c
    elif "cpu_baseline" in result:
        line["cpu_baseline"] = None
    if "parity_check" in result:
        line["parity_check"] = _pick(result["parity_check"], ("status", "violations", "entries", "max_abs_diff", "gate"))
    if "sustained" in result:
        line["sustained"] = _pick(result["sustained"], ("seconds", "steps", "frames_per_s"))
    h2h = result.get("host_to_host")
    if isinstance(h2h, dict):
        line["host_to_host"] = {k: _pick(v, ("frames_per_s", "ms", "pcie_GBps")) for k, v in h2h.items() if isinstance(v, dict)}
    if side_path:
        line["full_record"] = side_path
    # N > 1: the three rates of the job and what the short gather can reach, in one compact object near the tail
    scale_keys = ("frames_per_s", "frames_per_s_without_gather", "frames_per_s_mid_gather", "frames_per_s_short_gather",
                  "gather_bytes_per_step_into_root", "rccl_ranks", "distinct_devices")
    if any(k in cfg for k in scale_keys[:4]):
        sc = _pick(cfg, scale_keys)
        es = cfg.get("expected_speedup_short_gather")
        if es:
            sc["expected_speedup_short_gather"] = _pick(es, ("speedup_over_one_gpu", "ideal", "bound"))
        line["scale"] = sc
    if "configs" in result:
        line["configs_columns"] = result.get("configs_columns")
        line["configs"] = result["configs"]
    line = _r(line)
    droppable = ["host_to_host", "sustained", ("cpu_baseline", "reference_in_build_container"), ("config", "launched_by"),
                 ("config", "multi_gpu"), ("roofline", "note"), ("cpu_baseline", "sample")]
    text = json.dumps(line, separators=(",", ":"))
    while len(text) > limit and droppable:
        key = droppable.pop(0)
        if isinstance(key, tuple):
            if isinstance(line.get(key[0]), dict):
                line[key[0]].pop(key[1], None)
        else:
            line.pop(key, None)
        text = json.dumps(line, separators=(",", ":"))
    if len(text) > limit and isinstance(line.get("configs"), dict):
        # still too long: the configs table loses its non-BASELINE rows, last first
        keep = ("cfg2", "cfg2_job", "cfg4_job", "cfg3", "cfg4_shard", "cfg5_features", "cfg5_spectrogram", "cfg5_chromagram")
        for k in [k for k in reversed(list(line["configs"])) if k not in keep]:
            del line["configs"][k]
            text = json.dumps(line, separators=(",", ":"))
            if len(text) <= limit:
                break
    return text


def write_side_record(result, world):
    """The full record (verbose config.others, every provenance string) beside the line: gpurun_out/ when the tree has one
    or can have one (it is what travels back from a GPU box), else the working directory.  Returns the path written, or None."""
    name = "bench_full_n%d.json" % world
    for d in (os.path.join(ROOT, "gpurun_out"), os.getcwd()):
        try:
            os.makedirs(d, exist_ok=True)
            path = os.path.join(d, name)
            with open(path, "w") as fh:
                json.dump(result, fh, indent=1)
            return os.path.relpath(path, ROOT) if path.startswith(ROOT) else path
        except OSError:
            continue
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--workload", default=None, choices=["cfg2", "cfg4"],
                    help="default: cfg2 at N = 1 (headline), cfg4 strong scaling at N > 1")
    ap.add_argument("--seconds", type=float, default=3600.0, help="cfg2: clip length")
    ap.add_argument("--clips", type=int, default=CFG4_TOTAL_CLIPS, help="cfg4: clips in the whole job")
    ap.add_argument("--deltas", type=int, default=0, help="1: 68-row output (reference default), 0: the 34-feature metric")
    ap.add_argument("--no-gather", action="store_true", help="N>1: skip the RCCL gather to rank 0")
    ap.add_argument("--gather", default="short", choices=["short", "mid"],
                    help="N>1: what `value` measures -- short: the [34][T] slabs travel to rank 0 (the north star's gather; "
                         "link-bound, see expected_speedup_short_gather); mid: 68 rows are computed and kept, the (136, 10) "
                         "mid-term matrices travel (the near-linear configuration, DESIGN section 6)")
    ap.add_argument("--comm-timeout", type=int, default=180, help="N>1: seconds allowed for the RCCL rendezvous")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--event-every", type=int, default=4,
                    help="HIP-event pair around every n-th feature-kernel launch of the timed region (roofline.kernel_avg_ms)")
    ap.add_argument("--no-extras", action="store_true", help="skip config.others, host_to_host and the sustained loop")
    ap.add_argument("--cpu-frames", type=int, default=60000, help="frames of the clip prefix timed on one CPU core")
    ap.add_argument("--sustain-seconds", type=float, default=2.5)
    ap.add_argument("--check", type=int, default=1, help="verify the clip against the oracle after timing")
    ap.add_argument("--prewarm-seconds", type=float, default=0.3,
                    help="untimed device work before the --warmup steps (the engine clock of a fresh box ramps over the "
                         "first ~0.2 s of load; reported in the line)")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # bare `python bench.py --gpus N`: no launcher exported RANK / WORLD_SIZE -- this process becomes the launcher
        raise SystemExit(self_launch(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: the line would claim GPUs that did not run" % (args.gpus, world))
    workload = args.workload or ("cfg2" if world == 1 else "cfg4")

    group = None                        # control plane: TCP sockets on the launcher's environment (no torch)
    if world > 1:
        # several processes that touch GPUs on this driver stack need dmabuf IPC (RCCL's peer mappings fail with
        # "hipIpcGetMemHandle: invalid argument" in the legacy mode); must be in place before the HIP runtime loads
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        from pyaudioanalysis_amd._rendezvous import SocketGroup
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        group = SocketGroup(rank=rank, world_size=world, timeout=max(60.0, float(args.comm_timeout)))

    # the all-cores CPU leg runs FIRST, before this process loads HIP, so that its workers can simply be forked
    cpu_all = None
    if world == 1 and not args.no_cpu_baseline:
        try:
            import cpu_bench
            cpu_all = cpu_bench.all_cores(FS, WINDOW, STEP)
        except Exception as exc:
            cpu_all = {"error": repr(exc)}

    from pyaudioanalysis_amd import _ffi
    from pyaudioanalysis_amd import distributed as D
    from synth import synth_clip
    lib = _ffi.lib()
    if _ffi.device_count() < 1:
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    _ffi.init(local_rank % _ffi.device_count())
    bus = ctypes.create_string_buffer(64)
    _ffi.check(lib.paa_device_bus_id(bus, 64))
    devices = group.all_gather(bus.value.decode()) if group else [bus.value.decode()]      # PCI bus id of every rank's device

    # ------------------------------------------------------------------ workload
    host_clip = None                    # rank 0 keeps what the spot check and the CPU baseline need
    if workload == "cfg2":
        n = int(args.seconds * FS)
        seeds = [2 + rank, 1002 + rank, 2002 + rank]          # three distinct clips: inputs rotate through > 256 MiB
        d_inputs = []
        for sd in seeds:
            x = synth_clip(sd, n, FS)
            if host_clip is None:
                host_clip = x
            d_inputs.append(_ffi.DeviceBuffer.from_host(x))
        offsets = np.array([0, n], dtype=np.int64)
        desc = "cfg2: 1 clip x %d s, 16 kHz mono int16, 800/400, %d rows; 3 distinct clips rotate (seeds %s)" % (
            args.seconds, 68 if args.deltas else 34, seeds)
        scaling = "weak"
        job_clips = world
    else:
        n = 10 * FS
        frames_all = np.full(args.clips, (n - WINDOW) // STEP + 1, dtype=np.int64)
        a, b = D.partition_by_frames(frames_all, world)[rank]
        my_clips = b - a
        pool = np.stack([synth_clip(40000 + i, n, FS) for i in range(64)])
        # clip k of the job is pool[k % 64]: rotate the pool so that this rank's first clip lines up
        pool = np.roll(pool, -(a % 64), axis=0)
        host_clip = pool[0].copy()
        d_inputs = [replicate_on_device(_ffi, pool, my_clips)]
        offsets = np.arange(my_clips + 1, dtype=np.int64) * n
        desc = "cfg4: %d clips x 10 s in the job (64 distinct seeded clips, tiled), %d on this rank, 800/400, %d rows" % (
            args.clips, my_clips, 68 if args.deltas else 34)
        scaling = "strong"
        job_clips = args.clips

    # N > 1 with the gather and 68 rows: every rank computes and ships the 34 BASE rows, rank 0 re-forms rows 34..67 on its device
    # (paa_dev_expand_deltas: exact differences of consecutive columns, bit-identical to the 68-row kernel) -- half the bytes
    # on the xGMI links into the root
    ship_base = world > 1 and not args.no_gather and bool(args.deltas)
    plan = _ffi.Plan(offsets, FS, WINDOW, STEP, deltas=bool(args.deltas) and not ship_base, sample_kind=0)
    F = plan.F
    frames = plan.total_frames
    d_out = _ffi.DeviceBuffer(plan.out_doubles * 8)
    d_out2 = _ffi.DeviceBuffer(plan.out_doubles * 8) if world > 1 else d_out   # N>1: the gather of step k overlaps step k+1

    # frames / result sizes of every rank (equal for cfg2; cfg4 follows the partition)
    sizes = group.all_gather((int(frames), int(plan.out_doubles))) if group else [(int(frames), int(plan.out_doubles))]
    total_frames = sum(s[0] for s in sizes)
    counts = np.array([s[1] for s in sizes], dtype=np.int64)

    gather = world > 1 and not args.no_gather
    gather_note = None
    comm_hung = False
    d_all = None
    comm = None
    if gather:
        def bcast(payload):
            return group.broadcast(payload, 0)
        # the communicator is created on a helper thread with a deadline: a rendezvous that never completes (one rank
        # missing, a fabric problem) must cost the scaling run its gather, not the whole run
        box = {}

        def init_comm():
            try:
                box["comm"] = D.RcclGather(world, rank, bcast, group.all_gather)
            except Exception as exc:
                box["err"] = exc
        th = threading.Thread(target=init_comm, daemon=True)
        th.start()
        th.join(args.comm_timeout)
        if th.is_alive():
            ok = False
            comm_hung = True
            gather_note = "RCCL init did not return within %d s on rank %d" % (args.comm_timeout, rank)
        elif "err" in box:     # keep the scaling run alive, say what happened
            ok = False
            gather_note = "RCCL init failed on rank %d: %s" % (rank, box["err"])
        else:
            comm = box["comm"]
            if rank == 0:
                d_all = _ffi.DeviceBuffer(int(counts.sum()) * 8)
            ok = True
        gather = all(group.all_gather(ok))
        if ship_base and not gather:         # no exchange after all: the ranks produce their 68 rows themselves
            ship_base = False
            plan.destroy()
            plan = _ffi.Plan(offsets, FS, WINDOW, STEP, deltas=True, sample_kind=0)
            F = plan.F
            d_out = _ffi.DeviceBuffer(plan.out_doubles * 8)
            d_out2 = _ffi.DeviceBuffer(plan.out_doubles * 8)
    d_full = frames_job = None
    if ship_base and rank == 0:
        # frames of every clip of the job in rank order (contiguous clip ranges): the layout of the gathered base rows
        per_clip = (10 * FS - WINDOW) // STEP + 1 if workload == "cfg4" else int(frames)
        n_job = int(job_clips)
        frames_job = np.full(n_job, per_clip, dtype=np.int64)
        assert int(frames_job.sum()) * 34 == int(counts.sum())
        d_full = _ffi.DeviceBuffer(2 * int(counts.sum()) * 8)

    bufs = [d_out, d_out2]
    state = {"k": 0}

    def step():
        k = state["k"]
        state["k"] = k + 1
        buf = bufs[k & 1]
        plan.execute(d_inputs[k % len(d_inputs)], buf)
        if gather:
            # runs on the library's communication stream; the next write of `buf` waits for it
            comm.gather(buf, counts, 0, d_all)
            if d_full is not None:       # root: delta rows of the whole job, queued behind the gather on the communication stream
                _ffi.check(lib.paa_dev_expand_deltas(d_all.ptr, _ffi.as_i64p(frames_job), len(frames_job), d_full.ptr))

    def device_sync():
        _ffi.sync()                      # both library streams (compute + communication): everything this process queued

    def barrier():
        device_sync()
        if group is not None:
            group.barrier()

    def max_over_ranks(seconds):
        return seconds if group is None else float(group.all_max(float(seconds)))

    # a fresh box idles at a low engine clock: run the step untimed for a fixed wall time first, so that a short
    # --warmup does not decide what is measured (the --warmup steps follow as asked)
    if args.prewarm_seconds > 0:
        saved_gather, gather = gather, False          # (time-based: no collective in here, ranks may differ in count)
        t_pre = time.perf_counter()
        while time.perf_counter() - t_pre < args.prewarm_seconds:
            for _ in range(20):
                step()
            device_sync()
        gather = saved_gather
    for _ in range(args.warmup):
        step()
    barrier()
    # every 4th launch of the feature kernel is bracketed by a HIP-event pair on the library stream (an event pair on every
    # launch costs the step ~4 %: it is an ordering point between back-to-back kernels)
    _ffi.check(lib.paa_prof_enable(args.event_every))
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    device_sync()
    elapsed = time.perf_counter() - t0
    if group is not None:
        group.barrier()
    elapsed = max_over_ranks(elapsed)
    kms = ctypes.c_double()
    kn = ctypes.c_int64()
    _ffi.check(lib.paa_prof_read(ctypes.byref(kms), ctypes.byref(kn)))
    _ffi.check(lib.paa_prof_enable(0))

    # N > 1: also time the same steps without the gather, and the cheap gather (reported beside the headline value)
    value_no_gather = None
    mid_gather = None
    if gather:
        saved, gather = gather, False
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        device_sync()
        el2 = time.perf_counter() - t1
        group.barrier()
        value_no_gather = total_frames * args.steps / max_over_ranks(el2)
        gather = saved
        # the cheap gather: 68 short-term rows stay in HBM, the (136, M) mid-term matrices (1.0 s / 1.0 s) travel
        plan68 = _ffi.Plan(offsets, FS, WINDOW, STEP, deltas=True, sample_kind=0)
        d_st68 = _ffi.DeviceBuffer(plan68.out_doubles * 8)
        n_mid = plan68.mid_doubles(40)
        d_mids = [_ffi.DeviceBuffer(max(n_mid, 1) * 8) for _ in range(2)]
        mid_counts = np.array(group.all_gather(int(n_mid)), dtype=np.int64)
        d_mid_all = _ffi.DeviceBuffer(max(int(mid_counts.sum()), 1) * 8) if rank == 0 else None

        def mid_step(k):
            plan68.execute(d_inputs[k % len(d_inputs)], d_st68)
            plan68.mid_execute(d_st68, 39, 40, d_mids[k & 1])
            comm.gather(d_mids[k & 1], mid_counts, 0, d_mid_all)
        for k in range(min(args.warmup, 3)):
            mid_step(k)
        barrier()
        t1 = time.perf_counter()
        for k in range(args.steps):
            mid_step(k)
        device_sync()
        el3 = time.perf_counter() - t1
        group.barrier()
        el3 = max_over_ranks(el3)
        mid_gather = {"frames_per_s": total_frames * args.steps / el3, "ms_per_step": 1e3 * el3 / args.steps,
                      "bytes_per_step_into_root": int(mid_counts[1:].sum() * 8),
                      "what": "68 short-term rows per frame computed and kept in HBM, (136, 10) mid-term matrices "
                              "(1.0 s / 1.0 s) gathered to rank 0"}
        plan68.destroy()

    # a sustained loop of the same step (>= 2 s of continuous GPU work: visible to an external smi sampler)
    sustained = None
    if not args.no_extras and world == 1:
        t1 = time.perf_counter()
        n_sus = 0
        while time.perf_counter() - t1 < args.sustain_seconds:
            for _ in range(50):
                step()
            _ffi.sync()
            n_sus += 50
        sustained = {"seconds": time.perf_counter() - t1, "steps": n_sus}
        sustained["frames_per_s"] = frames * n_sus / sustained["seconds"]

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = total_frames * args.steps / elapsed
        bytes_per_frame = 2 * STEP + 8 * F               # int16 in once + f64 out once
        k_avg_ms = kms.value / max(1, kn.value)
        achieved = bytes_per_frame * frames / (k_avg_ms * 1e-3) / 1e9 if k_avg_ms > 0 else 0.0
        result = {
            "metric": "short-term frames/sec (34-feat, 16 kHz, 50 ms/25 ms) + HBM GB/s vs peak",
            # n_gpus = the DEVICES that ran (advisor, round 5): ranks that share a device (--no-gather on a box with fewer
            # GPUs than ranks) do not make it a multi-GPU figure -- `ranks` says how many processes, `scaling` is null then
            "value": value, "unit": "frames/s", "n_gpus": len(set(devices)), "ranks": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": (scaling if len(set(devices)) == world else None), "vs_baseline": None,
            "prewarm_seconds": args.prewarm_seconds,
            "dtype": "f64", "data": "synthetic (oracle/synth.py; SURVEY 8d seeds)",
            "config": {"workload": desc, "frames_per_step_job": int(total_frames),
                       "frames_per_step_rank0": int(frames), "clips_in_job": int(job_clips),
                       "window": WINDOW, "step": STEP, "rows": 68 if args.deltas else 34, "kernel": plan.kernel_name,
                       "rows_computed_per_rank": F,
                       "input_buffers_rotated": len(d_inputs),
                       "devices": devices, "distinct_devices": len(set(devices)),
                       "rccl_ranks": (world if comm is not None and gather else None),
                       "launched_by": ("bench.py itself (no launcher environment)" if os.environ.get("PAA_BENCH_SELF_LAUNCHED")
                                       else ("launcher environment (RANK / WORLD_SIZE)" if world > 1 else "single process")),
                       "multi_gpu": ("contiguous clip ranges per rank (partition_by_frames), RCCL gather of the slabs to "
                                     "rank 0 overlapped with the next step" if (gather and workload == "cfg4") else
                                     "one clip per rank, RCCL gather to rank 0" if gather else
                                     ("no gather" if world > 1 else "single GPU"))},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": plan.kernel_name, "kernel_avg_ms": k_avg_ms, "launches_timed": int(kn.value),
                         "event_pair_every_nth_launch": int(args.event_every),
                         "algorithmic_bytes_per_frame": bytes_per_frame,
                         "note": "path is FP64 VALU/LDS bound (about 40 flop/B); HBM fraction is reported as the "
                                 "metric asks, see DESIGN.md"},
        }
        if ship_base:
            result["config"]["gather_rows"] = ("34 base rows per frame travel; rank 0 re-forms rows 34..67 on its device "
                                               "(paa_dev_expand_deltas) inside the timed region")
        if gather_note:
            result["config"]["gather_note"] = gather_note
        if value_no_gather is not None:
            result["config"]["frames_per_s"] = value
            result["config"]["frames_per_s_without_gather"] = value_no_gather
            result["config"]["gather_bytes_per_step_into_root"] = int(counts[1:].sum() * 8)
        if mid_gather is not None:
            result["config"]["frames_per_s_mid_gather"] = mid_gather["frames_per_s"]
            result["config"]["mid_gather"] = mid_gather
            if args.gather == "mid":          # the documented near-linear configuration is the headline of this run
                result["config"]["frames_per_s_short_gather"] = value
                result["value"] = mid_gather["frames_per_s"]
                result["ms_per_step"] = mid_gather["ms_per_step"]
                result["config"]["value_is"] = "mid-term gather (--gather mid); the short gather is reported beside it"
        if value_no_gather is not None and world > 1:
            # what the short gather can reach at best: every peer's share of a step rides its own xGMI link
            # into rank 0, perfectly overlapped with the next step's kernels -- so that a reader of the first real
            # SCALE_r*.json can tell the design limit from a defect
            bytes_step = float(counts.sum()) * 8.0
            t1 = total_frames / (value_no_gather / world)             # one GPU doing the whole job, from the no-gather rate
            t_link = bytes_step / world / (XGMI_LINK_GBS * 1e9)       # one peer's block over one link
            result["config"]["expected_speedup_short_gather"] = {
                "speedup_over_one_gpu": t1 / max(t1 / world, t_link), "ideal": world,
                "compute_s_per_step_one_gpu": t1, "link_s_per_step": t_link, "link_GBps": XGMI_LINK_GBS,
                "bound": "xgmi link into the root" if t_link > t1 / world else "compute",
                "note": "arithmetic, not a measurement: per-GPU compute rate from this run's no-gather leg, one xGMI link "
                        "per peer at %.1f GB/s, gather of step k fully overlapped with step k+1" % XGMI_LINK_GBS}
        if sustained:
            result["sustained_frames_per_s"] = sustained["frames_per_s"]
            result["sustained"] = sustained
        # HBM traffic of the feature kernel: NOT measured in this run -- taken from the committed rocprofv3 PMC pass of
        # this same command (FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950, WRITE_SIZE as reported)
        try:
            prof = json.load(open(os.path.join(ROOT, "profiles", "latest_traffic.json")))
            if prof.get("kernel") == plan.kernel_name and prof.get("frames") == int(frames) and prof.get("rows") == F:
                prof_round = str(prof.get("round") or str(prof.get("source", ""))[:3])
                result["roofline"]["traffic_round"] = prof_round
                result["roofline"]["traffic_stale"] = prof_round != PROFILE_ROUND      # the file is older than this round's build
                result["roofline"]["traffic"] = prof["hbm_bytes_per_launch"]
                result["roofline"]["traffic_from_profile"] = prof["hbm_bytes_per_launch"]
                result["roofline"]["traffic_over_algorithmic"] = prof["hbm_bytes_per_launch"] / float(bytes_per_frame * frames)
                result["roofline"]["traffic_source"] = "not measured in this run: " + str(prof.get("source"))
        except Exception:
            pass
        # informational: executed FP64 work against the vector FP64 peak (the roofline that actually binds)
        result["roofline"]["fp64_valu"] = {"algorithmic_kflop_per_frame": KFLOP_PER_FRAME,
                                           "achieved_tflops": KFLOP_PER_FRAME * 1e3 * frames / (k_avg_ms * 1e-3) / 1e12 if k_avg_ms > 0 else 0.0,
                                           "peak_tflops": FP64_VALU_PEAK_TFLOPS}
        result["roofline"]["fp64_valu"]["frac"] = result["roofline"]["fp64_valu"]["achieved_tflops"] / FP64_VALU_PEAK_TFLOPS
        # the measured count: SQ_INSTS_VALU of the committed PMC pass x the FP64 share and flop weights of the kernel's ISA
        # histogram (scripts/fp64_executed.py): wave-instructions x 64 lanes, idle lanes included
        try:
            ex = json.load(open(os.path.join(ROOT, "profiles", "%s_fast800_fp64_executed.json" % PROFILE_ROUND)))
            if plan.kernel_name == "st_fast_800_w8" and F == 34 and ex.get("frames") == int(frames) and k_avg_ms > 0:
                fv = result["roofline"]["fp64_valu"]
                fv["issued_kflop_per_frame_measured"] = ex["issued_kflop_per_frame"]
                fv["issued_tflops"] = ex["issued_fp64_flop_per_launch"] / (k_avg_ms * 1e-3) / 1e12
                fv["issued_frac"] = fv["issued_tflops"] / FP64_VALU_PEAK_TFLOPS
                fv["issued_source"] = "profiles/%s_fast800_fp64_executed.json (counter pass of this command x ISA histogram)" % PROFILE_ROUND
        except Exception:
            pass
        # the data sheet's FP64 peak needs 2.4 GHz; the clock this kernel sustains was read from its waves' own cycle counters
        # (profiles/<round>_fast800_clock.json): the same fractions against the peak at THAT clock, beside the ones above
        try:
            ck = json.load(open(os.path.join(ROOT, "profiles", "%s_fast800_clock.json" % PROFILE_ROUND)))
            ghz = float(ck["sustained_clock_ghz"])
            if plan.kernel_name == ck.get("kernel") and ck.get("frames") == int(frames) and 0.5 < ghz <= float(ck.get("data_sheet_clock_ghz", 2.4)):
                fv = result["roofline"]["fp64_valu"]
                peak_at = FP64_VALU_PEAK_TFLOPS * ghz / float(ck.get("data_sheet_clock_ghz", 2.4))
                fv["sustained_clock_ghz"] = ghz
                fv["peak_tflops_at_sustained_clock"] = peak_at
                fv["frac_at_sustained_clock"] = fv["achieved_tflops"] / peak_at
                if "issued_tflops" in fv:
                    fv["issued_frac_at_sustained_clock"] = fv["issued_tflops"] / peak_at
                fv["clock_source"] = "not measured in this run: " + str(ck.get("source"))
        except Exception:
            pass
        if args.check:
            import paa_oracle as O
            # the buffer the last step wrote holds clip (k-1) % len(d_inputs); re-run clip 0 into d_out for the check
            plan.execute(d_inputs[0], d_out)
            _ffi.sync()
            T0 = int(lib.paa_num_frames(int(offsets[1] - offsets[0]), WINDOW, STEP))
            slab = d_out.to_host(np.float64, F * T0).reshape(F, T0)
            ref, checker, ill = None, None, None
            gate = "|d| <= 1e-4 |ref| + 1e-6 scale(row) + 1e-9"
            try:
                import c_oracle
                if c_oracle.available():      # every frame of the clip: the plain-C oracle does ~45 k frames/s
                    import checks             # (the -m gpu tests' own reference: C oracle, NumPy oracle on silent frames)
                    clip0 = host_clip[:int(offsets[1] - offsets[0])]
                    ref = checks.reference_matrix(clip0, FS, WINDOW, STEP, False)
                    ill = checks.ill_mask(clip0, FS, WINDOW, STEP)
                    checker = ("oracle/paa_oracle.c, all %d frames; digitally silent frames from oracle/paa_oracle.py (the "
                               "reference's pocketfft); %d frames with a numerically empty mel band get the documented "
                               "1e-5 row term on their MFCC rows" % (T0, int(ill.sum())))
            except Exception:
                ref = None
            if ref is None:                   # no C compiler on the box: a few frames through the NumPy oracle
                xn = O.normalize_clip(host_clip)
                tab = O.Tables(FS, WINDOW)
                pick = [0, 1, 63, 64, 65, T0 // 2, T0 - 1]
                cols = []
                for t in pick:
                    fr = xn[t * STEP:t * STEP + WINDOW]
                    X = O.magnitude_spectrum(fr, tab.nfft)
                    Xp = X if t == 0 else O.magnitude_spectrum(xn[(t - 1) * STEP:(t - 1) * STEP + WINDOW], tab.nfft)
                    cols.append(O.frame_vector(fr, X, Xp, tab))
                ref = np.stack(cols, axis=1)
                slab = np.ascontiguousarray(slab[:, pick])
                checker = "oracle/paa_oracle.py, frames %s" % pick
                worst, _ = O.mixed_tolerance_violations(slab[:34], ref[:34], 1e-4, 1e-6, 1e-9)
            else:
                # the contract gate of the north star exactly as tests/test_parity_gpu.py applies it
                worst, _ = checks.contract_violations(slab[:34], ref[:34], ill)
            result["parity_check"] = {"status": "ok" if worst == 0 else "FAILED", "violations": int(worst),
                                      "entries": int(ref[:34].size), "checker": checker, "gate": gate,
                                      "max_abs_diff": float(np.max(np.abs(slab[:34] - ref[:34])))}
            result["parity_spot_check"] = "ok" if worst == 0 else "FAILED (%d entries)" % worst
        if not args.no_extras and world == 1:
            try:
                result["config"]["others"] = other_configs(_ffi)
            except Exception as exc:
                result["config"]["others"] = {"error": repr(exc)}
            try:
                result["host_to_host"] = host_to_host(_ffi, host_clip)
            except Exception as exc:
                result["host_to_host"] = {"error": repr(exc)}
        if not args.no_cpu_baseline and world == 1:
            import cpu_bench
            one = cpu_bench.single_core(host_clip, FS, WINDOW, STEP, args.cpu_frames)
            best = max(one["numpy_port"], one["c_port"] or 0.0)
            cb = {"value": best, "unit": "frames/s", "cores": 1, "kind": "port",
                  "numpy_port": one["numpy_port"], "c_port": one["c_port"], "cpu_model": cpu_bench.cpu_model(),
                  "host_cores": os.cpu_count(),
                  "sample": "first %.0f s of the bench clip (%d frames; %.1f s of CPU for oracle/paa_oracle.py, %.1f s for "
                            "oracle/paa_oracle.c), deltas off, single thread" % (
                                one["seconds_of_audio"], one["frames"], one["numpy_port_seconds"], one["c_port_seconds"]),
                  "ports_vs_reference": "both ports hoist the frame-invariant tables (mel bank, chroma map, DCT) that "
                                        "the reference rebuilds per call / per frame and are 8-18x faster than it"}
            try:
                cb["reference_cost_port"] = cpu_bench.reference_cost(host_clip, FS, WINDOW, STEP)
                cb["reference_note"] = ("kind stays 'port': the Python reference may not travel to this host in any form; "
                                        "reference_cost_port = its cost structure restated, within 3 % of it where both run")
            except Exception as exc:
                cb["reference_cost_port"] = {"error": repr(exc)}
            # the UNMODIFIED reference, timed by scripts/reference_cpu_baseline.py in the build container (the GPU box has
            # no /root/reference): embedded so that the stated baseline travels with the line
            try:
                rec = json.load(open(os.path.join(ROOT, "profiles", "reference_cpu_r03.json")))
                cb["reference"] = {"where": "measured in the build container, not on this host",
                                   "cpu_model": rec.get("cpu_model"), "numpy": rec.get("numpy"), "scipy": rec.get("scipy"),
                                   "cores": 1, "source": "profiles/reference_cpu_r03.json",
                                   "frames_per_s": {k: v.get("frames_per_s") for k, v in rec.get("entries", {}).items()},
                                   "ports_on_that_core": rec.get("ports_on_this_core")}
            except Exception as exc:
                cb["reference"] = {"error": repr(exc)}
            cb["all_cores"] = cpu_all
            result["cpu_baseline"] = cb
        elif not args.no_cpu_baseline:
            result["cpu_baseline"] = None
        # every BASELINE configuration once more, compact and LAST in the line (what a truncated tail of the line keeps):
        # name -> [frames/s, ms per step, fraction of the HBM peak (SURVEY 8d's algorithmic bytes), kernel]
        configs = {workload + ("_job" if world > 1 else ""): [float("%.4g" % result["value"]), float("%.4g" % result["ms_per_step"]),
                                                              float("%.4g" % result["roofline"]["frac"]), plan.kernel_name]}
        others = result["config"].get("others") or {}
        for key, e in others.items():
            if isinstance(e, dict) and "frames_per_s" in e:
                configs[key] = compact(e)
        for key in ("cfg3", "cfg4_shard", "cfg5_features", "cfg5_spectrogram", "cfg5_chromagram"):
            if key in configs:             # flat scalars beside the nested record
                result["config"][key + "_frames_per_s"] = configs[key][0]
                result["config"][key + "_hbm_frac"] = configs[key][2]
        result["configs_columns"] = ["frames_per_s", "ms_per_step", "hbm_frac", "kernel"]
        result["configs"] = configs
        print(format_line(result, write_side_record(result, world)))
        sys.stdout.flush()
    if comm is not None:
        comm.close()
    if group is not None:
        group.close()
    if comm_hung:                      # a helper thread is still inside RCCL: do not wait for it at interpreter exit
        sys.stdout.flush()
        os._exit(0)


if __name__ == "__main__":
    main()
