#!/bin/bash
# round 5, GPU pass l (second consolidation): the whole -m gpu suite on the build with the shared prime butterflies (551 / 1102)
# and the final workgroup-per-frame kernels, the default bench line, and the counter passes of the kernels that changed since pass h
out=gpurun_out/r05l; mkdir -p $out
(timeout 1200 python -m pytest tests -m gpu -q --no-header --durations=5 --maxfail=30 2>&1 | tail -60) > $out/tests.log
tail -5 $out/tests.log
timeout 900 python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err
python - <<'PY'
import json
try:
    d = json.loads(open('gpurun_out/r05l/bench.json').read().strip().splitlines()[-1])
    print('value', d['value'], 'ms', d['ms_per_step'], 'kernel ms', d['roofline']['kernel_avg_ms'], d.get('parity_check', {}).get('status'))
    for k, v in d['configs'].items(): print(k, v)
except Exception as e:
    print('bench parse failed', e); print(open('gpurun_out/r05l/bench.err').read()[-2000:])
PY
for c in reg_features_stereo w551_11k w2400 w2205 w1764 w1920 big_16000; do
  timeout 300 bash scripts/profile_kernel.sh r05 $c 40 > $out/prof_$c.log 2>&1
done
ls gpurun_out/r05_*_summary.json | wc -l
