#!/bin/bash
# round 5, GPU pass a: the whole -m gpu suite on the build with the power-of-two three-pass shapes, the base-row gather, the GPU
# beat_extraction entry and the bounded MFCC exception; then the default bench line (driver-style: 20 steps) and the self-launched
# two-rank run without the gather
out=gpurun_out/r05a; mkdir -p $out
(timeout 1500 python -m pytest tests -m gpu -q --no-header --durations=8 -x 2>&1 | tail -40) > $out/tests.log
tail -5 $out/tests.log
timeout 600 python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err
python - <<'PY'
import json
try:
    d = json.loads(open('gpurun_out/r05a/bench.json').read().strip().splitlines()[-1])
    print('value', d['value'], 'ms', d['ms_per_step'], 'kernel ms', d['roofline']['kernel_avg_ms'])
    for k, v in d['configs'].items(): print(k, v)
except Exception as e:
    print('bench parse failed', e); print(open('gpurun_out/r05a/bench.err').read()[-2000:])
PY
