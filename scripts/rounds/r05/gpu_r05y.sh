#!/bin/bash
# round 5, GPU pass y (fifth consolidation, after the hot kernel's halo moved inside the first quad and its runs were balanced):
# the whole -m gpu suite, the default bench line, the headline profile passes (scripts/profile.sh r05: kernel trace + four counter
# passes), the wave trace of the -DPAA_F800_TRACE build (sustained clock), counter passes of the two other cases that run the hot kernel
out=gpurun_out/r05y; mkdir -p $out
(timeout 1200 python -m pytest tests -m gpu -q --no-header --durations=5 --maxfail=30 2>&1 | tail -60) > $out/tests.log
grep -n "passed\|failed" $out/tests.log | tail -3
timeout 900 python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err
python - <<'PY'
import json
try:
    d = json.loads(open('gpurun_out/r05y/bench.json').read().strip().splitlines()[-1])
    print('value', d['value'], 'ms', d['ms_per_step'], 'kernel ms', d['roofline']['kernel_avg_ms'], d.get('parity_check', {}).get('status'))
    for k, v in d['configs'].items(): print(k, v)
except Exception as e:
    print('bench parse failed', e); print(open('gpurun_out/r05y/bench.err').read()[-2000:])
PY
timeout 900 bash scripts/profile.sh r05 > $out/profile.log 2>&1
python scripts/summarize_prof.py gpurun_out/prof_r05 gpurun_out/r05_fast800_w8_summary.json > $out/summarize.log 2>&1
rm -rf gpurun_out/prof_r05/trace gpurun_out/prof_r05/pmc1 gpurun_out/prof_r05/pmc2 gpurun_out/prof_r05/pmc3 gpurun_out/prof_r05/pmc4
tail -3 $out/summarize.log
(echo "# scripts/phase_timing.py with the -DPAA_F800_TRACE build (per-wave life times only), headline plan (1-hour clip, 800/400), after 0.5 s of untimed launches"
 PAA_HIP_LIBRARY=$PWD/pyaudioanalysis_amd/libpaa_hip_trace.so timeout 300 python scripts/phase_timing.py) > $out/wave_trace.txt 2> $out/wave_trace.err
head -12 $out/wave_trace.txt; tail -2 $out/wave_trace.err
for c in fast_s800 mid_stats; do
  timeout 300 bash scripts/profile_kernel.sh r05 $c 40 > $out/prof_$c.log 2>&1
done
ls gpurun_out/r05_*_summary.json | wc -l
