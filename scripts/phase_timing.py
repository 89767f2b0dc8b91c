"""Per-phase cycle split and per-wave life times of st_fast_800 (needs a build with PAA_HIPCC_FLAGS=-DPAA_F800_TIMING)."""
import ctypes, sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from pyaudioanalysis_amd import _ffi
from synth import synth_clip
lib = _ffi.lib(); _ffi.init(0)
x = synth_clip(2, 3600 * 16000)
d_in = _ffi.DeviceBuffer.from_host(x)
plan = _ffi.Plan(np.array([0, len(x)], dtype=np.int64), 16000, 800, 400, deltas=False)
d_out = _ffi.DeviceBuffer(plan.out_doubles * 8)
import time
# a fresh box idles at a low engine clock: run the plan for a fixed wall time first (PAA_PHASE_PREWARM seconds, default 0.5),
# so that the clock the wave trace reports below is the sustained one
t_pre = time.perf_counter()
while time.perf_counter() - t_pre < float(os.environ.get("PAA_PHASE_PREWARM", "0.5")):
    for _ in range(20): plan.execute(d_in, d_out)
    _ffi.sync()
buf = (ctypes.c_uint64 * 16)()
lib.paa_debug_phase_cycles(buf)
for _ in range(5): plan.execute(d_in, d_out)
_ffi.sync()
lib.paa_debug_phase_cycles(buf)
v = np.array(list(buf), dtype=np.float64)
names = ["stage", "time-domain", "pass1 dft25", "exchange", "pass2+post", "sweepA+entropy", "spread/flux/rolloff", "mel", "chroma", "dct/fv", "store"]
tot = v[:11].sum()
print(plan.kernel_name, "waves", int(v[15]), "cycles/wave %.0f" % (tot / max(v[15], 1)), "steals per launch %.0f" % (v[14] / 5))
tot = max(tot, 1.0)
for n, c in zip(names, v[:11]): print("%-22s %6.2f %%   %.0f cycles/wave" % (n, 100 * c / tot, c / max(v[15], 1)))
tr = (ctypes.c_uint64 * (4096 * 4))()
n = lib.paa_debug_wave_trace(tr, 4096)
if n > 0:
    t = np.array(list(tr), dtype=np.uint64).reshape(-1, 4)
    nw = min(int(v[15]) // 5, 4096)
    t = t[:nw]
    t0 = t[:, 0].astype(np.float64); t1 = t[:, 1].astype(np.float64)
    base = t0.min()
    print("waves traced", nw, "kernel span (first start -> last end) %.1f us" % ((t1.max() - base) / 100.0))
    print("start offsets us: min %.2f median %.2f p90 %.2f max %.2f" % tuple(np.percentile((t0 - base) / 100.0, [0, 50, 90, 100])))
    life = (t1 - t0) / 100.0
    print("wave life us: min %.1f median %.1f p90 %.1f max %.1f" % tuple(np.percentile(life, [0, 50, 90, 100])))
    cyc = t[:, 2].astype(np.float64)
    print("wave cycles: min %.0f median %.0f max %.0f  -> clock %.2f GHz" % (cyc.min(), np.median(cyc), cyc.max(), np.median(cyc / (life * 1e3))))
    hw = t[:, 3]
    cu = (hw >> np.uint64(8)) & np.uint64(15); se = (hw >> np.uint64(13)) & np.uint64(7); sh = (hw >> np.uint64(12)) & np.uint64(1); xcc = (hw >> np.uint64(32)) & np.uint64(15)
    key = (xcc.astype(np.int64) << 12) | (se.astype(np.int64) << 8) | (sh.astype(np.int64) << 4) | cu.astype(np.int64)
    uniq, counts = np.unique(key, return_counts=True)
    print("distinct CUs used", len(uniq), "waves per CU: min %d max %d" % (counts.min(), counts.max()), "histogram", np.bincount(counts))
    late = np.argsort(t1)[-8:]
    print("last finishers: start us", np.round((t0[late] - base) / 100.0, 1), "life", np.round(life[late], 1), "tile", late)
if n > 0:
    simd = ((hw >> np.uint64(4)) & np.uint64(3)).astype(int)
    slot = (hw & np.uint64(15)).astype(int)
    widx = np.arange(nw) % (8 if "w8" in plan.kernel_name else 4)
    for w in range(widx.max() + 1):
        m = widx == w
        print("wave %d of the workgroup: life us mean %.1f min %.1f max %.1f | simd %s slot %s" % (w, life[m].mean(), life[m].min(), life[m].max(), np.bincount(simd[m], minlength=4), np.bincount(slot[m])[:6]))
    for b in (0, 1, 100, 249):
        sl = slice(8 * b, 8 * b + 8) if "w8" in plan.kernel_name else slice(4 * b, 4 * b + 4)
        print("block", b, "life", np.round(life[sl], 0), "simd", simd[sl], "slot", slot[sl], "xcc", xcc[sl][:1])
    # where the spread of the wave lives comes from: per XCD, per workgroup, per iteration count (round 5)
    if "w8" in plan.kernel_name and nw % 8 == 0:
        wg_life = life.reshape(-1, 8)
        wg_end = ((t1 - base) / 100.0).reshape(-1, 8).max(axis=1)
        wg_xcc = xcc.reshape(-1, 8)[:, 0].astype(int)
        wg_cyc = cyc.reshape(-1, 8).mean(axis=1)
        print("workgroup end us: min %.1f p10 %.1f median %.1f p90 %.1f max %.1f" % tuple(np.percentile(wg_end, [0, 10, 50, 90, 100])))
        for xq in sorted(set(wg_xcc)):
            m = wg_xcc == xq
            print("xcc %d: %3d workgroups, end us mean %.1f min %.1f max %.1f | cycles/wave mean %.0f | clock %.3f GHz" % (
                xq, m.sum(), wg_end[m].mean(), wg_end[m].min(), wg_end[m].max(), wg_cyc[m].mean(),
                np.median((cyc.reshape(-1, 8)[m] / (wg_life[m] * 1e3)))))
        order = np.argsort(wg_end)
        print("slowest workgroups", order[-6:], "xcc", wg_xcc[order[-6:]], "fastest", order[:6], "xcc", wg_xcc[order[:6]])
