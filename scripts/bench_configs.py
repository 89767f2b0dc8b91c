"""Throughput of the other BASELINE configs (parity-test cases, not the bench line) -- run on the GPU box.
Prints one JSON object; kernel-resident timing (inputs/outputs in HBM) with HIP events."""
import ctypes, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from pyaudioanalysis_amd import _ffi, MidTermFeatures
from synth import synth_clip
import paa_oracle as O

lib = _ffi.lib(); _ffi.init(0)
out = {}


def timed(fn, reps=10, warm=3):
    for _ in range(warm): fn()
    _ffi.sync()
    lib.paa_timer_start()
    for _ in range(reps): fn()
    ms = ctypes.c_float()
    lib.paa_timer_stop(ctypes.byref(ms))
    return ms.value / reps


def plan_case(name, packed, offsets, fs, W, S, deltas, kind):
    d_in = _ffi.DeviceBuffer.from_host(packed)
    plan = _ffi.Plan(offsets, fs, W, S, deltas=deltas, sample_kind=kind)
    d_out = _ffi.DeviceBuffer(plan.out_doubles * 8)
    ms = timed(lambda: plan.execute(d_in, d_out))
    out[name] = {"kernel": plan.kernel_name, "frames": plan.total_frames, "ms": ms,
                 "frames_per_s": plan.total_frames / (ms * 1e-3)}
    return plan, d_in, d_out

# cfg3: 1000 x 30 s clips, mid 1.0/1.0 s over 800/400 (64 distinct clips tiled)
pool = [synth_clip(3000 + i, 30 * 16000) for i in range(64)]
n_clips = 1000
packed = np.concatenate([pool[i % 64] for i in range(n_clips)])
offsets = np.arange(n_clips + 1, dtype=np.int64) * (30 * 16000)
plan, d_in, d_out = plan_case("cfg3_short_term_68rows", packed, offsets, 16000, 800, 400, True, 0)
d_mid = _ffi.DeviceBuffer(plan.mid_doubles(40) * 8)
ms = timed(lambda: (plan.execute(d_in, d_out), plan.mid_execute(d_out, 39, 40, d_mid)))
out["cfg3_mid_term_total"] = {"clips": n_clips, "ms": ms, "clips_per_s": n_clips / (ms * 1e-3),
                              "st_frames_per_s": plan.total_frames / (ms * 1e-3)}
# spot parity of one clip's mid-term block against the oracle
mid = d_mid.to_host(np.float64, 136 * 30).reshape(136, 30)
ref_mid, _, _ = O.mid_feature_extraction(pool[0], 16000, 16000, 16000, 800, 400)
out["cfg3_mid_term_total"]["parity_violations"] = O.mixed_tolerance_violations(mid, ref_mid)[0]
del plan, d_in, d_out, d_mid

# cfg4 shard: 12 500 x 10 s clips (one GPU's share of 100 000 on 8 GPUs)
pool = [synth_clip(40000 + i, 160000) for i in range(64)]
n_clips = 12500
packed = np.concatenate([pool[i % 64] for i in range(n_clips)])
offsets = np.arange(n_clips + 1, dtype=np.int64) * 160000
plan, d_in, d_out = plan_case("cfg4_shard_12500_clips_34rows", packed, offsets, 16000, 800, 400, False, 0)
del plan, d_in, d_out

# cfg5: 44.1 kHz stereo -> mono (float64), 1102/441: features (generic kernel)
xs = synth_clip(5, 44100 * 600, fs=44100, stereo=True)
mono = O.stereo_to_mono(xs)
plan_case("cfg5_features_1102_441_f64_600s", mono, np.array([0, len(mono)], dtype=np.int64), 44100, 1102, 441, True, 1)
# reference default 50 ms / 50 ms
xl = synth_clip(8, 3600 * 16000)
plan_case("st_800_800_int16_3600s", xl, np.array([0, len(xl)], dtype=np.int64), 16000, 800, 800, True, 0)
plan_case("st_640_320_int16_3600s", xl, np.array([0, len(xl)], dtype=np.int64), 16000, 640, 320, True, 0)
plan_case("st_400_160_int16_3600s", xl, np.array([0, len(xl)], dtype=np.int64), 16000, 400, 160, True, 0)
x = synth_clip(7, 600 * 16000)
plan_case("st_800_800_int16_600s", x, np.array([0, len(x)], dtype=np.int64), 16000, 800, 800, True, 0)
plan_case("st_640_320_int16_600s", x, np.array([0, len(x)], dtype=np.int64), 16000, 640, 320, True, 0)
# host-to-host (PCIe inclusive) of the headline shape, 10 min clip
from pyaudioanalysis_amd import ShortTermFeatures
t0 = time.perf_counter(); F, _ = ShortTermFeatures.feature_extraction(x, 16000, 800, 400, deltas=False); dt = time.perf_counter() - t0
t0 = time.perf_counter(); F, _ = ShortTermFeatures.feature_extraction(x, 16000, 800, 400, deltas=False); dt = time.perf_counter() - t0
out["host_to_host_800_400_600s_34rows"] = {"frames": F.shape[1], "ms": dt * 1e3, "frames_per_s": F.shape[1] / dt}
print(json.dumps(out, indent=1))
