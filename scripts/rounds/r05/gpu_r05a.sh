#!/bin/bash
# round 5, GPU pass a: the whole -m gpu suite on the build with the power-of-two three-pass shapes, the base-row gather, the GPU
# beat_extraction entry and the bounded MFCC exception; the default bench line (driver-style: 20 steps); the headline profile
# passes (scripts/profile.sh r05: kernel trace + four counter passes)
out=gpurun_out/r05a; mkdir -p $out
(timeout 1200 python -m pytest tests -m gpu -q --no-header --durations=8 --maxfail=30 2>&1 | tail -80) > $out/tests.log
tail -6 $out/tests.log
timeout 600 python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err
python - <<'PY'
import json
try:
    d = json.loads(open('gpurun_out/r05a/bench.json').read().strip().splitlines()[-1])
    print('value', d['value'], 'ms', d['ms_per_step'], 'kernel ms', d['roofline']['kernel_avg_ms'], d.get('parity_check', {}).get('status'))
    for k, v in d['configs'].items(): print(k, v)
except Exception as e:
    print('bench parse failed', e); print(open('gpurun_out/r05a/bench.err').read()[-2000:])
PY
timeout 900 bash scripts/profile.sh r05 > $out/profile.log 2>&1
python scripts/summarize_prof.py gpurun_out/prof_r05 gpurun_out/r05_fast800_w8_summary.json > $out/summarize.log 2>&1
rm -rf gpurun_out/prof_r05/trace gpurun_out/prof_r05/pmc1 gpurun_out/prof_r05/pmc2 gpurun_out/prof_r05/pmc3 gpurun_out/prof_r05/pmc4
tail -3 $out/summarize.log
